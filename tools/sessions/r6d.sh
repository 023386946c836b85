set -u
O=gpurun_out/r6d; mkdir -p $O
PADEL_LIB=tools/ab/libpadel_hip_probes.so timeout 300 python tools/timeline_probe.py --kernel h2r --out "$O/timeline_h2r_192.txt" > /dev/null 2>"$O/timeline_192.err"; echo "tl192 rc=$?"
PADEL_LIB=tools/ab/libpadel_hip_probes.so timeout 300 python tools/timeline_probe.py --kernel h2r --cin 96 --cout 96 --hw 96x160 --out "$O/timeline_h2r_96.txt" > /dev/null 2>"$O/timeline_96.err"; echo "tl96 rc=$?"
head -75 $O/timeline_h2r_192.txt; tail -5 $O/timeline_192.err
