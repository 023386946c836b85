set -u
O=gpurun_out/r6B; mkdir -p $O
timeout 600 python tools/conv_bench.py --dtype h2 --w16 --tiles T243,T244,T244:65 --reps 7 --shapes "1x1" > $O/h2s_probe.txt 2>&1; grep -v amdgpu.ids $O/h2s_probe.txt | head -10
