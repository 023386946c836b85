# HBM traffic of the default (v5 tap) conv kernel: FETCH_SIZE and WRITE_SIZE in separate --pmc passes (kernel-trace only)
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
timeout 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmct_$c -o p -- python $R/tools/conv_bench.py --tiles auto --reps 1 --shapes "m.P4.bneck,m.head0,m.c2f.cv2" > $R/gpurun_out/pmct_$c.log 2>&1
python - "$R/gpurun_out/pmct_$c/p_counter_collection.csv" $c <<'PY'
import csv, sys, collections
agg = collections.defaultdict(float); name = {}
for r in csv.DictReader(open(sys.argv[1])):
    if "conv_tap" in r["Kernel_Name"]:
        agg[int(r["Dispatch_Id"])] += float(r["Counter_Value"]); name[int(r["Dispatch_Id"])] = r["Kernel_Name"]
for d in sorted(agg): print(sys.argv[2], "dispatch", d, name[d][:58], "%.1f MB (counter unit KiB)" % (agg[d] * 1024 / 1e6))
PY
done
