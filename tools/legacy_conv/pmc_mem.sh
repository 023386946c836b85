# memory-path counters of the 64x96 LDS conv tile (separate --pmc passes, kernel-trace only)
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(TCP|TCC|TA|TD)_[A-Za-z0-9_]+" | sort -u > $R/gpurun_out/mem_counters.txt
cd /tmp
export PADEL_CONV_LDS_VARIANT=7
run() { n=$1; shift; timeout 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmcm$n -o p -- python $R/tools/conv_bench.py --one 0 0 --reps 2 --shapes "m.P4.bneck" > $R/gpurun_out/pmcm$n.log 2>&1; echo "pass $n rc=$?"; }
run 1 GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run 2 GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_sum
run 3 GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
run 4 GRBM_GUI_ACTIVE TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
for n in 1 2 3 4; do f=$(ls $R/gpurun_out/pmcm$n/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(float)
for r in rows:
    if "conv_lds_kernel" in r["Kernel_Name"]:
        agg[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
last = max(d for d, _ in agg)
print({c: v for (d, c), v in agg.items() if d == last})
PY
done
tail -2 $R/gpurun_out/pmcm1.log
