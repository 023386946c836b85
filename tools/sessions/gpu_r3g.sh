#!/bin/bash
# round 3, GPU call G: whole GPU suite, the full default bench line, c2, and the fp16 configurations for the record
mkdir -p gpurun_out/r3g
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 > gpurun_out/r3g/pytest_gpu.txt 2>&1
echo "pytest gpu rc=$?" | tee -a gpurun_out/r3g/status.txt
grep -E "passed|failed|FAILED" gpurun_out/r3g/pytest_gpu.txt | tail -12
cp gpurun_out/parity_report.json gpurun_out/r3g/parity_report.json 2>/dev/null
cp gpurun_out/parity_report_fp16.json gpurun_out/r3g/parity_report_fp16.json 2>/dev/null
cp gpurun_out/config4_report.json gpurun_out/r3g/config4_report.json 2>/dev/null
timeout 1200 python bench.py --dump-ops gpurun_out/r3g/ops_c3.csv > gpurun_out/r3g/bench_c3.json 2> gpurun_out/r3g/bench_c3.err
echo "bench c3 rc=$?" | tee -a gpurun_out/r3g/status.txt
cat gpurun_out/r3g/bench_c3.json
timeout 600 python bench.py --workload c2 --no-cpu-baseline --no-host-frames > gpurun_out/r3g/bench_c2.json 2> gpurun_out/r3g/bench_c2.err
echo "bench c2 rc=$?" | tee -a gpurun_out/r3g/status.txt
timeout 600 python bench.py --workload c4 --no-cpu-baseline --no-host-frames > gpurun_out/r3g/bench_c4.json 2> gpurun_out/r3g/bench_c4.err
echo "bench c4 rc=$?" | tee -a gpurun_out/r3g/status.txt
timeout 600 python bench.py --impl bx3 --no-cpu-baseline --no-host-frames --no-reference-default > gpurun_out/r3g/bench_c3_bx3.json 2> gpurun_out/r3g/bench_c3_bx3.err
echo "bench c3 bx3 rc=$?" | tee -a gpurun_out/r3g/status.txt
python -c "
import json
for n in ('c2','c4','c3_bx3'):
    d=json.load(open('gpurun_out/r3g/bench_%s.json'%n)); print(n, d['value'], d['ms_per_step'], d['engine_only']['value'], d['roofline']['achieved'], d['roofline']['frac'])
"
