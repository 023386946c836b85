set -u
O=gpurun_out/r6o; mkdir -p $O
# tuning word: bits 5.. = ABL >> 4: 16 -> 32+9, 2048 -> 4096+9
PADEL_LIB=tools/ab/libpadel_hip_probes.so timeout 600 python tools/conv_bench.py --dtype h2 --w16 --tiles T324:9,T324:41,T324:4105,T324:489,T324:2025 --reps 7 --shapes "m.P3.bneck,m.P4.bneck" > $O/ablate_h2r.txt 2>&1; grep -v amdgpu.ids $O/ablate_h2r.txt | head -4
