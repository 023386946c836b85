#!/bin/bash
# round 3, GPU call W: stem store-coalescing ceiling probe (wrong output layout; timing of kind-1 ops only)
mkdir -p gpurun_out/r3w
cp tools/probe_build/libpadel_hip.so padel_analytics_amd/libpadel_hip.so
timeout 600 python bench.py --engine-only --no-compare --no-cpu-baseline --no-host-frames --no-reference-default --dump-ops gpurun_out/r3w/ops_probe.csv > gpurun_out/r3w/bench_probe.json 2> gpurun_out/r3w/bench_probe.err
echo "rc=$?"
grep -E "^(players|ball|pose),1," gpurun_out/r3w/ops_probe.csv
