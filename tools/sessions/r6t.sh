set -u
O=gpurun_out/r6t; mkdir -p $O
timeout 600 python tools/tracknet_bench.py --frames 264 --feed 64 --dump-ops $O/tracknet_ops.csv > $O/tracknet_bench.json 2> $O/tracknet_bench.err; echo "tracknet rc=$?"; cat $O/tracknet_bench.json
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/rocprof" -o tn -- python "$GRAFT_REPO_ROOT/tools/tracknet_bench.py" --frames 264 --feed 64 > "$GRAFT_REPO_ROOT/$O/rocprof_tracknet.json" 2> "$GRAFT_REPO_ROOT/$O/rocprof_tracknet.err" ); echo "rocprof rc=$?"
find "$O/rocprof" -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} "$O/tracknet_kernel_stats.csv"
find "$O/rocprof" -type f ! -name '*kernel_stats.csv' -delete 2>/dev/null
head -14 "$O/tracknet_kernel_stats.csv" | cut -c1-200
python - <<'PY'
import csv, collections
rows=list(csv.DictReader(open('gpurun_out/r6t/tracknet_ops.csv')))
agg=collections.defaultdict(lambda:[0,0.0,0.0])
for r in rows:
    k=(r['kind'],r['ksize'],r['M'],r['cin'],r['cout'],r['stride'],r['bm'],r['bn'])
    a=agg[k]; a[0]+=1; a[1]+=float(r['ms']); a[2]+=float(r['flops'])
tot=sum(a[1] for a in agg.values())
for k,a in sorted(agg.items(), key=lambda kv:-kv[1][1])[:25]:
    print(k, a[0], '%.3f ms  %.1f%%  %.0f TF/s'%(a[1], 100*a[1]/tot, a[2]/a[1]/1e9 if a[1] else 0))
print('total ms', tot)
PY
