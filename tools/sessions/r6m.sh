set -u
O=gpurun_out/r6m; mkdir -p $O
for t in 1 9 17; do
PADEL_LIB=tools/ab/libpadel_hip_probes.so timeout 300 python tools/timeline_probe.py --kernel h2r --tune $t --out "$O/timeline_h2r_192_t$t.txt" > /dev/null 2>"$O/timeline_192.err"; echo "tl192 t$t rc=$?"
grep -A 12 "phase of the two" $O/timeline_h2r_192_t$t.txt | cut -c1-260
done
