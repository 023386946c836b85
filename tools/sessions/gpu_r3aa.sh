#!/bin/bash
# round 3, GPU call AA: wide patch kernel (tiles 341..343): h2 unit tests + speed against the 8 x 16 tiles on the few-channel layers
mkdir -p gpurun_out/r3aa
timeout 600 python -m pytest tests/test_gpu_h2.py -m gpu -q -x -k conv_variants > gpurun_out/r3aa/pytest_h2.txt 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r3aa/status.txt
tail -12 gpurun_out/r3aa/pytest_h2.txt
timeout 600 python tools/conv_bench.py --dtype h2 --reps 3 --shapes "m.P2.bneck,pose.P2.bneck,n.P2.bneck,n.P3.bneck" --tiles auto,T303,T313,T304,T341,T342,T343 > gpurun_out/r3aa/sweep_h2w.txt 2>&1
cat gpurun_out/r3aa/sweep_h2w.txt
