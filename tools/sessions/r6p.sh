set -u
O=gpurun_out/r6p; mkdir -p $O
timeout 600 python tools/conv_bench.py --dtype h2 --w16 --tiles T324:9,T324:41,T324:73,T324:105 --reps 7 --shapes "P3.bneck,P4.bneck,P5.bneck,head0" > $O/cache.txt 2>&1; grep -v amdgpu.ids $O/cache.txt | head -10
