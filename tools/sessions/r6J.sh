set -u
O=gpurun_out/r6J; mkdir -p $O
timeout 600 python tools/conv_bench.py --dtype h2 --w16 --tiles T245,T245:65,T245:321,T245:577,T245:4161 --reps 7 --shapes "1x1 1152->384,pose 1x1 768->384,pose 1x1 1152->576,pose 1x1 384->384" > $O/h2s_probe_halfw.txt 2>&1; grep -v amdgpu.ids $O/h2s_probe_halfw.txt | head -16
