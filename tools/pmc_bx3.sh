#!/bin/bash
# matrix-pipe utilisation, wait breakdown and LDS behaviour of the default (bf16x3) conv kernels: two --pmc passes,
# kernel-trace only, on the yolov8m P4 / P3 bottleneck layers (tools/conv_bench.py tile "B" = default choice)
mkdir -p gpurun_out; R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp
cd /tmp
run() { n=$1; shift; rm -rf $R/gpurun_out/pmcb$n; timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmcb$n -o p -- python $R/tools/conv_bench.py --tiles B --reps 2 --shapes "m.P4.bneck,m.P3.bneck,m.P2.bneck,m.c2f.cv2 1x1 576" > $R/gpurun_out/pmcb$n.log 2>&1; echo "pass $n rc=$?"; }
run 1 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
run 2 GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
python - "$R/gpurun_out" <<'PY'
import csv, sys, collections, glob
root = sys.argv[1]
for n in (1, 2):
    f = glob.glob(f"{root}/pmcb{n}/**/*counter_collection.csv", recursive=True)
    t = glob.glob(f"{root}/pmcb{n}/**/*kernel_trace.csv", recursive=True)
    if not f: print("pass", n, "no data"); continue
    agg = collections.defaultdict(float); name = {}
    for r in csv.DictReader(open(f[0])):
        if "conv_bx3" in r["Kernel_Name"]:
            agg[(int(r["Dispatch_Id"]), r["Counter_Name"])] += float(r["Counter_Value"]); name[int(r["Dispatch_Id"])] = r["Kernel_Name"]
    dur = {}
    for r in csv.DictReader(open(t[0])):
        if "conv_bx3" in r["Kernel_Name"]:
            dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    seen = set()
    for d in sorted(name):
        c = {k: v for (dd, k), v in agg.items() if dd == d}
        key = (name[d], round(dur.get(d, 0), 2))
        cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8.0
        if n == 1:
            print(f"d{d} {name[d][13:58]} {dur.get(d,0):.3f} ms clk {cyc/(dur.get(d,1)*1e-3)/1e9:.2f} GHz MfmaUtil {c['SQ_VALU_MFMA_BUSY_CYCLES']/(cyc*1024):.3f} "
                  f"wait_inst {c['SQ_WAIT_INST_ANY']/c['SQ_WAVE_CYCLES']:.2f} wait_any {c['SQ_WAIT_ANY']/c['SQ_WAVE_CYCLES']:.2f} active {c['SQ_ACTIVE_INST_ANY']/c['SQ_WAVE_CYCLES']:.2f} "
                  f"VALU {c['SQ_INSTS_VALU']:.3g} SALU {c['SQ_INSTS_SALU']:.3g} wave_cyc {c['SQ_WAVE_CYCLES']:.3g}")
        else:
            print(f"d{d} {name[d][13:58]} {dur.get(d,0):.3f} ms LDS conflict {c.get('SQ_LDS_BANK_CONFLICT',0):.3g} / idx_active {c.get('SQ_LDS_IDX_ACTIVE',0):.3g} "
                  f"insts_lds {c.get('SQ_INSTS_LDS',0):.3g} wait_inst_lds {c.get('SQ_WAIT_INST_LDS',0):.3g} vmem_rd {c.get('SQ_INSTS_VMEM_RD',0):.3g} "
                  f"active_valu {c.get('SQ_ACTIVE_INST_VALU',0):.3g} active_lds {c.get('SQ_ACTIVE_INST_LDS',0):.3g} gui/xcd {cyc:.3g}")
PY
