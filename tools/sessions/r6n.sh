set -u
O=gpurun_out/r6n; mkdir -p $O
PADEL_LIB=tools/ab/libpadel_hip_probes.so timeout 300 python tools/timeline_probe.py --kernel h2r --tune 9 --out "$O/timeline_h2r_192.txt" > /dev/null 2>"$O/timeline_192.err"; echo "tl192 rc=$?"
PADEL_LIB=tools/ab/libpadel_hip_probes.so timeout 300 python tools/timeline_probe.py --kernel h2r --tune 9 --cin 96 --cout 96 --hw 96x160 --out "$O/timeline_h2r_96.txt" > /dev/null 2>"$O/timeline_96.err"; echo "tl96 rc=$?"
sed -n 1,66p $O/timeline_h2r_192.txt | cut -c1-180
