#!/bin/bash
# round 3, GPU call AK: rocprofv3 --kernel-trace --stats of the bench command with the final build
mkdir -p gpurun_out/r3ak
export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3ak/stats -o s -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-frames --no-reference-default --engine-only --no-compare > $R/gpurun_out/r3ak/stats.log 2>&1
echo "stats rc=$?"
cd $R
f=$(find gpurun_out/r3ak/stats -name "*kernel_stats.csv" | head -1); head -8 "$f" | cut -c1-160
