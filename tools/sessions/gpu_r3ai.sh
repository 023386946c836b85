#!/bin/bash
mkdir -p gpurun_out/r3ai
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-host-frames --no-reference-default > gpurun_out/r3ai/bench_c3_driver_cmdline.json 2> gpurun_out/r3ai/err.txt
python -c "
import json; d=json.load(open('gpurun_out/r3ai/bench_c3_driver_cmdline.json')); print(d['value'], d['ms_per_step'], d['engine_only']['value'])"
