set -u
O=gpurun_out/r6i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_h2.py -m gpu -q -x -k "two_product or conv_variants or persistent" > $O/pytest.txt 2>&1; echo "tests rc=$?"; tail -3 $O/pytest.txt
timeout 600 python tools/conv_bench.py --dtype h2 --w16 --tiles T323,T324,T324:5 --reps 7 --shapes "P3.bneck,P4.bneck,P5.bneck,head0,players.P" > $O/persist.txt 2>&1; grep -v amdgpu.ids $O/persist.txt | head -14
PADEL_LIB=tools/ab/libpadel_hip_probes.so timeout 600 python tools/conv_bench.py --dtype h2 --w16 --tiles T324,T324:17,T324:241,T324:1009 --reps 7 --shapes "m.P3.bneck,m.P4.bneck" > $O/ablate_h2r.txt 2>&1; grep -v amdgpu.ids $O/ablate_h2r.txt | head -4
PADEL_LIB=tools/ab/libpadel_hip_probes.so timeout 300 python tools/timeline_probe.py --kernel h2r --out "$O/timeline_h2r_192.txt" > /dev/null 2>"$O/timeline_192.err"; echo "tl192 rc=$?"
PADEL_LIB=tools/ab/libpadel_hip_probes.so timeout 300 python tools/timeline_probe.py --kernel h2r --cin 96 --cout 96 --hw 96x160 --out "$O/timeline_h2r_96.txt" > /dev/null 2>"$O/timeline_96.err"; echo "tl96 rc=$?"
sed -n 1,4p $O/timeline_h2r_192.txt; sed -n 45,64p $O/timeline_h2r_192.txt
