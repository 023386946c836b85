set -u
O=gpurun_out/r6D; mkdir -p $O
timeout 600 python tools/conv_bench.py --dtype h2 --w16 --tiles T243,T245 --reps 7 --shapes "pose 1x1,1x1 1152,1x1 192->192" > $O/h2s_wide2.txt 2>&1; grep -v amdgpu.ids $O/h2s_wide2.txt | head -14
