#!/bin/bash
# round 3, GPU call N: sustained fp16 MFMA ceiling of the chip (no memory traffic), zero vs random operands
mkdir -p gpurun_out/r3n
timeout 300 tools/mfma_f16_ubench > gpurun_out/r3n/mfma_f16_ubench.txt 2>&1
cat gpurun_out/r3n/mfma_f16_ubench.txt
