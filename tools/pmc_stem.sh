#!/bin/bash
# what bounds the stem kernel: VALU / MFMA / memory counters of stem_mfma_kernel over the bench's engine-only step (2 --pmc passes)
mkdir -p gpurun_out; R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp
cd /tmp
run() { n=$1; shift; rm -rf $R/gpurun_out/pmcs$n; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmcs$n -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --engine-only --no-compare > $R/gpurun_out/pmcs$n.log 2>&1; echo "pass $n rc=$?"; }
run 1 GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES
run 2 GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum TCC_EA_RDREQ_sum
python - "$R/gpurun_out" <<'PY'
import csv, sys, collections, glob
root = sys.argv[1]
for n in (1, 2):
    f = glob.glob(f"{root}/pmcs{n}/**/*counter_collection.csv", recursive=True)
    t = glob.glob(f"{root}/pmcs{n}/**/*kernel_trace.csv", recursive=True)
    if not f: print("pass", n, "no data"); continue
    agg = collections.defaultdict(float); name = {}
    for r in csv.DictReader(open(f[0])):
        if "stem_mfma" in r["Kernel_Name"]:
            agg[(int(r["Dispatch_Id"]), r["Counter_Name"])] += float(r["Counter_Value"]); name[int(r["Dispatch_Id"])] = r["Kernel_Name"]
    dur = {}
    for r in csv.DictReader(open(t[0])):
        if "stem_mfma" in r["Kernel_Name"]:
            dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    for d in sorted(name):
        c = collections.defaultdict(float, {k: v for (dd, k), v in agg.items() if dd == d})
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        print(f"pass {n} d{d} {dur.get(d,0):.3f} ms clk {cyc/(max(dur.get(d,1),1e-9)*1e-3)/1e9:.2f} GHz", {k: f"{v:.3g}" for k, v in c.items()})
        if n == 1:
            print(f"    VALU busy per SIMD {c['SQ_ACTIVE_INST_VALU']/(cyc*1024):.2f}  MFMA busy {c['SQ_VALU_MFMA_BUSY_CYCLES']/(cyc*1024):.2f}  wait_inst {c['SQ_WAIT_INST_ANY']/max(c['SQ_WAVE_CYCLES'],1):.2f}  active {c['SQ_ACTIVE_INST_ANY']/max(c['SQ_WAVE_CYCLES'],1):.2f}  waves/SIMD {c['SQ_WAVE_CYCLES']/(cyc*1024):.1f}")
PY
