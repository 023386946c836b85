set -u
O=gpurun_out/r5m; mkdir -p $O
for lib in padel_analytics_amd/libpadel_hip.so tools/ab/libpadel_hip_stemprobe1.so tools/ab/libpadel_hip_stemprobe2.so tools/ab/libpadel_hip_stemprobe4.so tools/ab/libpadel_hip_stemprobe5.so; do
  n=$(basename $lib .so)
  PADEL_LIB=$lib timeout 300 python bench.py --steps 2 --warmup 1 --quick --engine-only --traffic none --dump-ops $O/ops_$n.csv > $O/bench_$n.json 2> $O/bench_$n.err
  echo "$n rc=$? stem+L1: $(grep -E '^[a-z]+,1,3,' $O/ops_$n.csv | awk -F, '{printf "%s %.3f ms  ", $1, $10}')"
done
