#!/bin/bash
# round 3, final GPU call: whole GPU suite + smoke + the default bench line with the final build (stem / pool changes included)
mkdir -p gpurun_out/r3ah
timeout 1200 python -m pytest tests -m gpu -q --maxfail=10 > gpurun_out/r3ah/pytest_gpu.txt 2>&1
echo "pytest gpu rc=$?" | tee -a gpurun_out/r3ah/status.txt
grep -E "passed|failed|FAILED" gpurun_out/r3ah/pytest_gpu.txt | tail -8
for f in parity_report.json parity_report_fp16.json config4_report.json; do cp gpurun_out/$f gpurun_out/r3ah/$f 2>/dev/null; done
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r3ah/smoke.txt 2>&1
echo "smoke rc=$?" | tee -a gpurun_out/r3ah/status.txt
timeout 900 python bench.py --dump-ops gpurun_out/r3ah/ops_c3.csv > gpurun_out/r3ah/bench_c3.json 2> gpurun_out/r3ah/bench_c3.err
echo "bench c3 rc=$?" | tee -a gpurun_out/r3ah/status.txt
python -c "
import json
d=json.load(open('gpurun_out/r3ah/bench_c3.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], d['engine_only']['value'], d['host_frames']['sequential_frames_per_s'], d['host_frames']['fanout_frames_per_s'], d['reference_default']['value'], r['achieved'], r['frac'], r['conv1x1'], r['all_kernels_ms_per_step'], r['other_ms_per_step'], d['parity']['linf_px_vs_fp64'], d['parity']['low_noise_heads']['linf_px_vs_fp32_oracle'], d['cpu_baseline']['value'])
"
