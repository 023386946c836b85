#!/bin/bash
# round 3, GPU call I: the quad patch kernel (tile 323): h2 unit tests (bitwise against the other tiles under the
# column-major tap order), then speed against tile 303
mkdir -p gpurun_out/r3i
timeout 600 python -m pytest tests/test_gpu_h2.py -m gpu -q -x > gpurun_out/r3i/pytest_h2.txt 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r3i/status.txt
tail -15 gpurun_out/r3i/pytest_h2.txt
timeout 600 python tools/conv_bench.py --dtype h2 --reps 3 --shapes "m.P4.bneck,pose.P3.bneck,m.P5.bneck,pose.head0,m.P3.bneck,m.head0" --tiles auto,T303,T323,T213 > gpurun_out/r3i/sweep_h2q.txt 2>&1
echo "sweep rc=$?" | tee -a gpurun_out/r3i/status.txt
cat gpurun_out/r3i/sweep_h2q.txt
