set -u
O=gpurun_out/r6y; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_h2.py -m gpu -q -x -k "stem" > $O/pytest.txt 2>&1; echo "tests rc=$?"; tail -3 $O/pytest.txt
for t in 1 17 9; do
PADEL_CONV_TUNE=$t timeout 600 python bench.py --steps 3 --warmup 1 --quick --engine-only --traffic none --dump-ops $O/ops_t$t.csv > $O/bench_t$t.json 2> $O/bench_t$t.err; echo "bench t$t rc=$?"
python - <<PY
import csv
rows=[r for r in csv.DictReader(open('$O/ops_t$t.csv')) if r['kind']=='1']
print('tune $t stem+l1 ms:', {r['tracker']: round(float(r['ms']),3) for r in rows})
import json; d=json.load(open('$O/bench_t$t.json')); print(' engine-only', d['value'], d['ms_per_step'])
PY
done
