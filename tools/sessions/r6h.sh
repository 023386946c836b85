set -u
O=gpurun_out/r6h; mkdir -p $O
PADEL_LIB=tools/ab/libpadel_hip_probes.so timeout 600 python tools/conv_bench.py --dtype h2 --w16 --tiles T324,T324:3,T324:241,T324:257,T324:497,T324:513,T324:1009 --reps 7 --shapes "m.P3.bneck,m.P4.bneck" > $O/ablate_h2r.txt 2>&1; grep -v amdgpu.ids $O/ablate_h2r.txt | head -12
