set -u
R=$(pwd); O=$R/gpurun_out/r6K; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  n=$(echo $C | cut -d' ' -f1)
  rm -rf $O/p_$n
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/p_$n -o p -- python $R/tools/conv_bench.py --dtype h2 --w16 --tiles T245,T243,T245:321 --reps 2 --shapes "1x1 1152->384,pose 1x1 768->384" > $O/log_$n.txt 2>&1; echo "$n rc=$?"
done
python - $O <<'PY'
import csv, sys, glob, collections
root = sys.argv[1]
for d in sorted(glob.glob(root + "/p_*")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    t = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    if not f: print(d, "no data"); continue
    dur = {int(r["Dispatch_Id"]): (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6 for r in csv.DictReader(open(t[0]))}
    agg = collections.defaultdict(float); name = {}
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "conv_h2s" in k or "conv_h2_1p" in k:
            agg[(int(r["Dispatch_Id"]), r["Counter_Name"])] += float(r["Counter_Value"]); name[int(r["Dispatch_Id"])] = k
    for i in sorted(name):
        c = {k: v for (dd, k), v in agg.items() if dd == i}
        print(i, name[i].replace("void padel::", "")[:40], f"{dur.get(i, 0):.3f} ms", {k: f"{v:.4g}" for k, v in c.items()})
PY
