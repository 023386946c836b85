#!/bin/bash
mkdir -p gpurun_out/r3v
timeout 600 python tools/runner_overlap_probe.py > gpurun_out/r3v/overlap_f32.txt 2> gpurun_out/r3v/overlap_f32.err; tail -3 gpurun_out/r3v/overlap_f32.err; cat gpurun_out/r3v/overlap_f32.txt
timeout 600 python tools/runner_overlap_probe.py --half > gpurun_out/r3v/overlap_f16.txt 2> gpurun_out/r3v/overlap_f16.err; cat gpurun_out/r3v/overlap_f16.txt
