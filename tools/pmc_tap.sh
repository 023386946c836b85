# MFMA utilisation of the default (v5 tap) conv kernel: one --pmc pass, kernel-trace only
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
cd /tmp
timeout 150 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmct -o p -- python $R/tools/conv_bench.py --tiles auto --reps 2 --shapes "m.P4.bneck,m.P3.bneck" > $R/gpurun_out/pmct.log 2>&1
python - "$R/gpurun_out/pmct/p_counter_collection.csv" "$R/gpurun_out/pmct/p_kernel_trace.csv" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(float); name = {}
for r in rows:
    if "conv_tap" in r["Kernel_Name"]:
        agg[(int(r["Dispatch_Id"]), r["Counter_Name"])] += float(r["Counter_Value"]); name[int(r["Dispatch_Id"])] = r["Kernel_Name"]
dur = {}
for r in csv.DictReader(open(sys.argv[2])):
    if "conv_tap" in r["Kernel_Name"]:
        dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
for d in sorted(name):
    c = {k: v for (dd, k), v in agg.items() if dd == d}
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    print(f"dispatch {d} {name[d][:60]} {dur.get(d, 0):.3f} ms  clock {cyc / (dur.get(d, 1) * 1e-3) / 1e9:.2f} GHz  MfmaUtil {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):.3f}"
          f"  wait_inst {c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']:.2f} wait_any {c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.2f} active {c['SQ_ACTIVE_INST_ANY'] / c['SQ_WAVE_CYCLES']:.2f}"
          f"  VALU {c['SQ_INSTS_VALU']:.3g} SALU {c['SQ_INSTS_SALU']:.3g}")
PY
