#!/bin/bash
# HBM traffic of the conv kernel: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (they do not
# fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots"), kernel-trace only.
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -o p -- python $R/tools/conv_bench.py --one 0 0 --reps 2 --shapes "m.P4.bneck,m.head0,m.c2f.cv2" > $R/gpurun_out/pmc_$c.log 2>&1
done
python - <<PY
import csv
for c in ("FETCH_SIZE","WRITE_SIZE"):
    rows=[r for r in csv.DictReader(open("$R/gpurun_out/pmc_%s/p_counter_collection.csv"%c)) if "conv" in r["Kernel_Name"]]
    for r in rows: print(c, r["Dispatch_Id"], r["Kernel_Name"][:48], r["Grid_Size"], r["Counter_Value"])
PY
