#!/bin/bash
# round 3, closing GPU call: whole GPU suite, smoke, the full default bench line, c2 / c4 / bx3 lines, per-layer PMC of the
# default kernels, the TrackNet tracker alone
mkdir -p gpurun_out/r3z
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 > gpurun_out/r3z/pytest_gpu.txt 2>&1
echo "pytest gpu rc=$?" | tee -a gpurun_out/r3z/status.txt
grep -E "passed|failed|FAILED" gpurun_out/r3z/pytest_gpu.txt | tail -12
for f in parity_report.json parity_report_fp16.json config4_report.json; do cp gpurun_out/$f gpurun_out/r3z/$f 2>/dev/null; done
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r3z/smoke.txt 2>&1
echo "smoke rc=$?" | tee -a gpurun_out/r3z/status.txt
timeout 1200 python bench.py --dump-ops gpurun_out/r3z/ops_c3.csv > gpurun_out/r3z/bench_c3.json 2> gpurun_out/r3z/bench_c3.err
echo "bench c3 rc=$?" | tee -a gpurun_out/r3z/status.txt
cat gpurun_out/r3z/bench_c3.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-host-frames --no-reference-default > gpurun_out/r3z/bench_c3_driver_cmdline.json 2> gpurun_out/r3z/bench_c3_driver.err
timeout 600 python bench.py --workload c2 --no-cpu-baseline --no-host-frames > gpurun_out/r3z/bench_c2.json 2> gpurun_out/r3z/bench_c2.err
timeout 600 python bench.py --workload c4 --no-cpu-baseline --no-host-frames --dump-ops gpurun_out/r3z/ops_c4.csv > gpurun_out/r3z/bench_c4.json 2> gpurun_out/r3z/bench_c4.err
timeout 600 python bench.py --impl bx3 --no-cpu-baseline --no-host-frames --no-reference-default > gpurun_out/r3z/bench_c3_bx3.json 2> gpurun_out/r3z/bench_c3_bx3.err
python -c "
import json
for n in ('c3_driver_cmdline','c2','c4','c3_bx3'):
    d=json.load(open('gpurun_out/r3z/bench_%s.json'%n)); print(n, d['value'], d['ms_per_step'], d['engine_only']['value'], d['roofline']['achieved'], d['roofline']['frac'])
"
timeout 300 python tools/tracknet_bench.py > gpurun_out/r3z/tracknet_bench.json 2> gpurun_out/r3z/tracknet_bench.err; tail -2 gpurun_out/r3z/tracknet_bench.json | cut -c1-400
bash tools/pmc_h2.sh > gpurun_out/r3z/conv_h2_pmc.txt 2>&1; grep -c MfmaUtil gpurun_out/r3z/conv_h2_pmc.txt
