#!/bin/bash
# HBM traffic of the bench's dominant kernel, per launch: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc
# passes (kernel-trace only — MI355X_MICROARCH.md "rocprofv3 PMC slots": both do not fit one pass), over the bench's
# own command line in engine-only mode, then summarised into profiles/r3_traffic.json (read by bench.py for
# roofline.traffic).  Usage (on the GPU box, from the repo root):  bash tools/pmc_bench_traffic.sh [c3|c2]
WL=${1:-c3}
IMPL=${2:-h2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd /tmp
python $R/bench.py --workload $WL --impl $IMPL --steps 1 --warmup 1 --no-cpu-baseline --engine-only --no-compare --dump-ops $R/gpurun_out/ops_$WL.csv > $R/gpurun_out/bench_ops_$WL.json 2> $R/gpurun_out/bench_ops_$WL.err
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_bench_$C
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_bench_$C -o p -- \
      python $R/bench.py --workload $WL --impl $IMPL --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --engine-only --no-compare > $R/gpurun_out/pmc_bench_$C.log 2>&1
  echo "pass $C rc=$?"
done
cd $R
python tools/pmc_traffic_summary.py $WL gpurun_out/ops_$WL.csv gpurun_out/pmc_bench_FETCH_SIZE gpurun_out/pmc_bench_WRITE_SIZE $IMPL
