#!/bin/bash
mkdir -p gpurun_out/r3y
timeout 600 python tools/two_stream_probe.py 2> gpurun_out/r3y/probe.err | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" > gpurun_out/r3y/two_stream_h2.txt; cat gpurun_out/r3y/two_stream_h2.txt; tail -3 gpurun_out/r3y/probe.err
timeout 600 python tools/two_stream_probe.py --half 2> gpurun_out/r3y/probe16.err | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" > gpurun_out/r3y/two_stream_f16.txt; cat gpurun_out/r3y/two_stream_f16.txt
