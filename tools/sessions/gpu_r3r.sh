#!/bin/bash
# round 3, GPU call R: fp16 quad patch kernel (tiles 323 / 324 / 326): unit tests + speed against the 8 x 16 tiles
mkdir -p gpurun_out/r3r
timeout 600 python -m pytest tests/test_gpu_fp16.py -m gpu -q -x > gpurun_out/r3r/pytest_fp16.txt 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r3r/status.txt
tail -12 gpurun_out/r3r/pytest_fp16.txt
timeout 600 python tools/conv_bench.py --dtype f16 --reps 3 --shapes "m.P4.bneck,pose.P3.bneck,m.P5.bneck,pose.head0,m.P3.bneck,m.head0,m.P2.bneck,pose.P2.bneck" --tiles auto,T303,T304,T306,T323,T324,T326 > gpurun_out/r3r/sweep_p16q.txt 2>&1
cat gpurun_out/r3r/sweep_p16q.txt
