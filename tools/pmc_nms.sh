#!/bin/bash
# instruction mix / wait breakdown of nms_kernel and decode_kernel on the bench's three graphs (one --pmc pass, kernel-trace only)
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; OUT=${1:-$R/gpurun_out/pmc_nms}
rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD \
    --output-format csv -d "$OUT" -o p -- python $R/bench.py --steps 1 --warmup 1 --quick --engine-only --no-roofline > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "pass rc=$?"
python - "$OUT" <<'PY'
import csv, sys, glob, collections
root = sys.argv[1]
f = glob.glob(f"{root}/**/*counter_collection.csv", recursive=True)
t = glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True)
agg = collections.defaultdict(float); name = {}
for r in csv.DictReader(open(f[0])):
    if "nms_kernel" in r["Kernel_Name"] or "decode_kernel" in r["Kernel_Name"]:
        agg[(int(r["Dispatch_Id"]), r["Counter_Name"])] += float(r["Counter_Value"]); name[int(r["Dispatch_Id"])] = r["Kernel_Name"]
dur = {}
for r in csv.DictReader(open(t[0])):
    dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
for d in sorted(name):
    c = collections.defaultdict(float, {k: v for (dd, k), v in agg.items() if dd == d})
    print(f"d{d} {name[d][:32]} {dur.get(d, 0):8.1f} us  wave_cyc {c['SQ_WAVE_CYCLES']:.3g} wait_any {c['SQ_WAIT_ANY'] / max(c['SQ_WAVE_CYCLES'], 1):.2f} "
          f"wait_inst {c['SQ_WAIT_INST_ANY'] / max(c['SQ_WAVE_CYCLES'], 1):.2f} active {c['SQ_ACTIVE_INST_ANY'] / max(c['SQ_WAVE_CYCLES'], 1):.2f} "
          f"VALU {c['SQ_INSTS_VALU']:.3g} SALU {c['SQ_INSTS_SALU']:.3g} LDS {c['SQ_INSTS_LDS']:.3g} VMEM_RD {c['SQ_INSTS_VMEM_RD']:.3g}")
PY
