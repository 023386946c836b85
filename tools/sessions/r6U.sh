set -u
R=$(pwd); O=$R/gpurun_out/r6U; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/p_$C
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/p_$C -o p -- python $R/bench.py --steps 1 --warmup 1 --quick --engine-only --no-roofline --traffic none --dump-ops $O/ops_c3.csv > $O/log_$C.json 2> $O/log_$C.err; echo "$C rc=$?"
done
python - $O <<'PY'
import csv, sys, glob, collections
root = sys.argv[1]
out = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{root}/p_{C}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: [0.0, set()])
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == C:
            k = r["Kernel_Name"].replace("void padel::", "").split("(")[0]
            agg[k][0] += float(r["Counter_Value"]); agg[k][1].add(r["Dispatch_Id"])
    out[C] = agg
with open(f"{root}/traffic_by_kernel.csv", "w") as fo:
    fo.write("kernel,launches,fetch_MB_x2,write_MB\n")
    for k in sorted(out["FETCH_SIZE"], key=lambda k: -out["FETCH_SIZE"][k][0]):
        fe = out["FETCH_SIZE"][k]; wr = out["WRITE_SIZE"].get(k, [0.0, set()])
        fo.write(f"\"{k}\",{len(fe[1])},{fe[0] * 1024 * 2 / 1e6:.1f},{wr[0] * 1024 / 1e6:.1f}\n")
print(open(f"{root}/traffic_by_kernel.csv").read())
PY
find $O -name '*.csv' -path '*p_*' -delete 2>/dev/null; rm -rf $O/p_FETCH_SIZE $O/p_WRITE_SIZE
