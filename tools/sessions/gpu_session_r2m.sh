#!/bin/bash
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
cd $R
timeout 300 python tools/conv_bench.py --reps 3 --tiles B220,B420,B520,B620,B720,B820,B920,B1020,B1120 --shapes "m.P3.bneck,m.P4.bneck,m.head0" > gpurun_out/conv_probe_bx3_r2m2.txt 2>&1; cat gpurun_out/conv_probe_bx3_r2m2.txt
