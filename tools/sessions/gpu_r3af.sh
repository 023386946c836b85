#!/bin/bash
# round 3, GPU call AF: stem with unconditional (clamped) pixel loads: graph-level tests of the three bench graphs + per-op times
mkdir -p gpurun_out/r3af
timeout 600 python -m pytest tests/test_gpu_bench_config.py tests/test_gpu_fp16.py -m gpu -q -k "not conv16" > gpurun_out/r3af/pytest.txt 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r3af/pytest.txt
timeout 600 python bench.py --engine-only --no-compare --no-cpu-baseline --no-host-frames --no-reference-default --dump-ops gpurun_out/r3af/ops_c3.csv > gpurun_out/r3af/bench.json 2> gpurun_out/r3af/bench.err
grep -E "^(players|ball|pose),1," gpurun_out/r3af/ops_c3.csv
python -c "
import json; d=json.load(open('gpurun_out/r3af/bench.json')); print(d['engine_only']['value'], d['roofline']['all_kernels_ms_per_step'], d['roofline']['other_ms_per_step'])"
