#!/bin/bash
# round 3, GPU call AE (final kernel set): HBM traffic of the 3x3 kernels of the final kernel set (PMC FETCH_SIZE / WRITE_SIZE passes) and the
# rocprofv3 --kernel-trace --stats summary of the bench command
mkdir -p gpurun_out/r3ae
bash tools/pmc_bench_traffic.sh c3 h2 > gpurun_out/r3ae/pmc_traffic.log 2>&1
tail -5 gpurun_out/r3ae/pmc_traffic.log
cp profiles/r3_traffic.json gpurun_out/r3ae/r3_traffic.json
export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3ae/stats -o s -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-frames --no-reference-default --engine-only --no-compare > $R/gpurun_out/r3ae/stats.log 2>&1
echo "stats rc=$?"
cd $R
f=$(find gpurun_out/r3ae/stats -name "*kernel_stats.csv" | head -1); echo $f; head -12 "$f" | cut -c1-200
