#!/bin/bash
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_ball.py -m gpu -q -rf --tb=short -k "upsample or ball or session or median or tracker" 2>&1 | tail -8
timeout 300 python tools/tracknet_bench.py 2>/dev/null | tail -1
