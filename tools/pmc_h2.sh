#!/bin/bash
# matrix-pipe utilisation, wait breakdown, LDS and L2 behaviour of the h2 conv kernels (default arithmetic since round 3):
# three --pmc passes, kernel-trace only, over tools/conv_bench.py --dtype h2 (tile "auto" = the per-layer default); W16=1: weights that
# are fp16 numbers, i.e. the two-product kernels (PA_CONV_W_SINGLE)
mkdir -p gpurun_out; R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp
SHAPES=${1:-"m.P4.bneck,m.P3.bneck,m.P2.bneck,pose.P2.bneck,m.L3 96,m.c2f.cv2 1x1 576,m.c2f.cv1 1x1 96->96 P2,m.c2f.cv2 1x1 1152"}
cd /tmp
run() { n=$1; shift; rm -rf $R/gpurun_out/pmch$n; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmch$n -o p -- python $R/tools/conv_bench.py --dtype h2 ${W16:+--w16} --tiles auto --reps 2 --shapes "$SHAPES" > $R/gpurun_out/pmch$n.log 2>&1; echo "pass $n rc=$?"; }
run 1 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
run 2 GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run 3 GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM TCC_HIT_sum TCC_MISS_sum
python - "$R/gpurun_out" <<'PY'
import csv, sys, collections, glob
root = sys.argv[1]
for n in (1, 2, 3):
    f = glob.glob(f"{root}/pmch{n}/**/*counter_collection.csv", recursive=True)
    t = glob.glob(f"{root}/pmch{n}/**/*kernel_trace.csv", recursive=True)
    if not f: print("pass", n, "no data"); continue
    agg = collections.defaultdict(float); name = {}
    for r in csv.DictReader(open(f[0])):
        if "conv_h2" in r["Kernel_Name"]:
            agg[(int(r["Dispatch_Id"]), r["Counter_Name"])] += float(r["Counter_Value"]); name[int(r["Dispatch_Id"])] = r["Kernel_Name"]
    dur = {}
    for r in csv.DictReader(open(t[0])):
        if "conv_h2" in r["Kernel_Name"]:
            dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    for d in sorted(name):
        c = collections.defaultdict(float, {k: v for (dd, k), v in agg.items() if dd == d})
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        nm = name[d].replace("void padel::", "")[:44]
        if n == 1:
            print(f"d{d} {nm} {dur.get(d,0):.3f} ms clk {cyc/(dur.get(d,1)*1e-3)/1e9:.2f} GHz MfmaUtil {c['SQ_VALU_MFMA_BUSY_CYCLES']/(cyc*1024):.3f} "
                  f"wait_inst {c['SQ_WAIT_INST_ANY']/c['SQ_WAVE_CYCLES']:.2f} wait_any {c['SQ_WAIT_ANY']/c['SQ_WAVE_CYCLES']:.2f} active {c['SQ_ACTIVE_INST_ANY']/c['SQ_WAVE_CYCLES']:.2f} "
                  f"VALU {c['SQ_INSTS_VALU']:.3g} SALU {c['SQ_INSTS_SALU']:.3g} wave_cyc {c['SQ_WAVE_CYCLES']:.3g}")
        elif n == 2:
            print(f"d{d} {nm} {dur.get(d,0):.3f} ms LDS conflict {c['SQ_LDS_BANK_CONFLICT']:.3g} / idx_active {c['SQ_LDS_IDX_ACTIVE']:.3g} = lds busy {c['SQ_LDS_IDX_ACTIVE']/(cyc*256):.2f} of CU-cycles "
                  f"insts_lds {c['SQ_INSTS_LDS']:.3g} wait_inst_lds {c['SQ_WAIT_INST_LDS']:.3g} vmem_rd {c['SQ_INSTS_VMEM_RD']:.3g} "
                  f"active_valu {c['SQ_ACTIVE_INST_VALU']:.3g} active_lds {c['SQ_ACTIVE_INST_LDS']:.3g}")
        else:
            tot = c['TCC_HIT_sum'] + c['TCC_MISS_sum']
            print(f"d{d} {nm} {dur.get(d,0):.3f} ms waves {c['SQ_WAVES']:.3g} sq_busy {c['SQ_BUSY_CYCLES']:.3g} vmem_wr {c['SQ_INSTS_VMEM_WR']:.3g} active_vmem {c['SQ_ACTIVE_INST_VMEM']:.3g} "
                  f"inst_cyc_vmem {c['SQ_INST_CYCLES_VMEM']:.3g} L2 hit {c['TCC_HIT_sum']/max(tot,1):.3f} of {tot:.3g}")
PY
