#!/bin/bash
# One parameterised GPU session script (round 4; replaces the per-call scripts of rounds 1-3, whose list is kept in
# tools/sessions/INDEX.md).  Usage on the GPU box, from the repository root:
#     bash tools/gpu_session.sh <tag> <stage> [<stage> ...]
# Every stage writes under gpurun_out/<tag>/ and appends "<stage> rc=<n>" to gpurun_out/<tag>/status.txt.
# Stages:
#   tests_h2      the h2 / conv kernel tests (bitwise across tiles, vs fp64)          ~2 min
#   tests_all     the whole -m gpu suite                                              ~6 min
#   smoke         __graft_entry__.smoke()
#   ubench        tools/mfma_f16_ubench (16x16x32 and 32x32x16 f16 MFMA ceilings)
#   sweep         tools/conv_bench.py --dtype h2 for every library under tools/ab/ and the product library
#   timeline      tools/timeline_probe.py --kernel h2q with tools/ab/libpadel_hip_probes.so (192->192 and 96->96)
#   pmc_nms       instruction mix / wait breakdown of nms_kernel and decode_kernel (tools/pmc_nms.sh)
#   tiles         tools/conv_bench.py --dtype h2 --tiles $TILES (default auto,T323,T303) on $SWEEP_ARGS shapes
#   tests_sel     python -m pytest $PYTEST_SEL -m gpu (any selection)
#   bench_c2q / bench_c4q   the other configs with --quick --steps 5 (runner + engine-only + roofline only)
#   bench_c2d / bench_c4d   the same with the driver's --steps 20 --warmup 5
#   tests_post    ball / known-answer (decode, NMS) / runner / bench-config suites
#   bench_driver  the driver's command line: python bench.py --gpus 1 --steps 20 --warmup 5
#   tests_f16     the fp16 kernel tests + BASELINE configs[0] / [3] / [4] tests
#   replay        engine-only c3 with 16 / 32 frames per pass over the op list (MALL residency experiment)
#   bench         python bench.py --dump-ops (default command line: c3)
#   bench_short   python bench.py --steps 5 --warmup 2 --dump-ops, engine-only extras skipped where the flag exists
#   bench_c2 / bench_c4   the other single-GPU configs
#   stats         rocprofv3 --kernel-trace --stats of the bench command (summary copied to gpurun_out/<tag>/)
#   pmc           tools/pmc_h2.sh (MFMA busy / wait breakdown / LDS conflicts / L2 hit rate of the h2 conv kernels; three --pmc passes)
set -u
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
note() { echo "$1 rc=$2" | tee -a "$OUT/status.txt"; }
for stage in "$@"; do
  case $stage in
    tests_h2)
      timeout 900 python -m pytest tests/test_gpu_h2.py tests/test_gpu_conv.py -m gpu -q -x > "$OUT/pytest_h2.txt" 2>&1; note $stage $?
      tail -3 "$OUT/pytest_h2.txt" ;;
    tests_all)
      timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 > "$OUT/pytest_gpu.txt" 2>&1; note $stage $?
      grep -E "passed|failed|FAILED|Error" "$OUT/pytest_gpu.txt" | tail -12
      for f in parity_report.json parity_report_fp16.json config4_report.json; do cp gpurun_out/$f "$OUT/$f" 2>/dev/null; done ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/smoke.txt" 2>&1; note $stage $?
      tail -2 "$OUT/smoke.txt" ;;
    ubench)
      timeout 300 tools/mfma_f16_ubench > "$OUT/mfma_f16_ubench.txt" 2>&1; note $stage $?
      cat "$OUT/mfma_f16_ubench.txt" ;;
    sweep)
      for lib in tools/ab/libpadel_hip_r*.so padel_analytics_amd/libpadel_hip.so; do
        [ -f "$lib" ] || continue
        name=$(basename "$lib" .so)
        PADEL_LIB=$lib timeout 600 python tools/conv_bench.py --dtype h2 --tiles auto --reps 5 ${SWEEP_ARGS:-} > "$OUT/sweep_$name.txt" 2>&1; note "sweep:$name" $?
        echo "== $name"; cat "$OUT/sweep_$name.txt" | tail -24
      done ;;
    timeline)
      PADEL_LIB=tools/ab/libpadel_hip_probes.so timeout 300 python tools/timeline_probe.py --kernel h2q --out "$OUT/timeline_h2q_192.txt" > /dev/null 2>"$OUT/timeline_192.err"; note "timeline:192" $?
      PADEL_LIB=tools/ab/libpadel_hip_probes.so timeout 300 python tools/timeline_probe.py --kernel h2q --cin 96 --cout 96 --hw 96x160 --out "$OUT/timeline_h2q_96.txt" > /dev/null 2>"$OUT/timeline_96.err"; note "timeline:96" $?
      head -60 "$OUT/timeline_h2q_192.txt"; head -20 "$OUT/timeline_h2q_96.txt"; grep -A3 'epilogue' "$OUT/timeline_h2q_96.txt" | head -5 ;;
    bench)
      timeout 1200 python bench.py --dump-ops "$OUT/ops_c3.csv" > "$OUT/bench_c3.json" 2> "$OUT/bench_c3.err"; note $stage $?
      python tools/bench_summary.py "$OUT/bench_c3.json" ;;
    bench_short)
      timeout 900 python bench.py --steps 5 --warmup 2 --quick --dump-ops "$OUT/ops_c3.csv" > "$OUT/bench_c3_short.json" 2> "$OUT/bench_c3_short.err"; note $stage $?
      python tools/bench_summary.py "$OUT/bench_c3_short.json" ;;
    bench_c2)
      timeout 900 python bench.py --workload c2 --dump-ops "$OUT/ops_c2.csv" > "$OUT/bench_c2.json" 2> "$OUT/bench_c2.err"; note $stage $?
      python tools/bench_summary.py "$OUT/bench_c2.json" ;;
    bench_c4)
      timeout 900 python bench.py --workload c4 --dump-ops "$OUT/ops_c4.csv" > "$OUT/bench_c4.json" 2> "$OUT/bench_c4.err"; note $stage $?
      python tools/bench_summary.py "$OUT/bench_c4.json" ;;
    tests_sel)
      timeout 1200 python -m pytest ${PYTEST_SEL:-tests} -m gpu -q --maxfail=5 > "$OUT/pytest_sel.txt" 2>&1; note $stage $?
      grep -E "passed|failed|FAILED|Error" "$OUT/pytest_sel.txt" | tail -12 ;;
    bench_c2d)
      timeout 600 python bench.py --workload c2 --steps 20 --warmup 5 --quick --dump-ops "$OUT/ops_c2.csv" > "$OUT/bench_c2.json" 2> "$OUT/bench_c2.err"; note $stage $?
      python tools/bench_summary.py "$OUT/bench_c2.json" ;;
    bench_c4d)
      timeout 600 python bench.py --workload c4 --steps 20 --warmup 5 --quick --dump-ops "$OUT/ops_c4.csv" > "$OUT/bench_c4.json" 2> "$OUT/bench_c4.err"; note $stage $?
      python tools/bench_summary.py "$OUT/bench_c4.json" ;;
    bench_c2q)
      timeout 600 python bench.py --workload c2 --steps 5 --warmup 2 --quick > "$OUT/bench_c2_quick.json" 2> "$OUT/bench_c2_quick.err"; note $stage $?
      python tools/bench_summary.py "$OUT/bench_c2_quick.json" ;;
    bench_c4q)
      timeout 600 python bench.py --workload c4 --steps 5 --warmup 2 --quick > "$OUT/bench_c4_quick.json" 2> "$OUT/bench_c4_quick.err"; note $stage $?
      python tools/bench_summary.py "$OUT/bench_c4_quick.json" ;;
    pmc_nms)
      timeout 900 bash tools/pmc_nms.sh "$GRAFT_REPO_ROOT/$OUT/pmc_nms" > "$OUT/pmc_nms.txt" 2>&1; note $stage $?
      cat "$OUT/pmc_nms.txt" | tail -14; rm -rf "$OUT/pmc_nms" ;;
    tiles)
      timeout 600 python tools/conv_bench.py --dtype h2 --tiles ${TILES:-auto,T323,T303} --reps 7 ${SWEEP_ARGS:-} > "$OUT/tiles.txt" 2>&1; note $stage $?
      grep -v 'amdgpu.ids' "$OUT/tiles.txt" | head -30 ;;
    tests_post)
      timeout 900 python -m pytest tests/test_gpu_ball.py tests/test_gpu_known_answers.py tests/test_gpu_nms_stress.py tests/test_gpu_runner.py tests/test_gpu_bench_config.py -m gpu -q -x > "$OUT/pytest_post.txt" 2>&1; note $stage $?
      tail -5 "$OUT/pytest_post.txt" ;;
    bench_driver)
      timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_c3_driver_cmdline.json" 2> "$OUT/bench_c3_driver_cmdline.err"; note $stage $?
      python tools/bench_summary.py "$OUT/bench_c3_driver_cmdline.json" ;;
    tests_f16)
      timeout 900 python -m pytest tests/test_gpu_fp16.py tests/test_gpu_baseline_configs.py -m gpu -q -x > "$OUT/pytest_f16.txt" 2>&1; note $stage $?
      tail -5 "$OUT/pytest_f16.txt"; cp gpurun_out/config4_report.json "$OUT/" 2>/dev/null ;;
    replay)
      for n in 16 32; do
        timeout 600 python bench.py --steps 5 --warmup 2 --quick --engine-only --no-roofline --replay $n > "$OUT/bench_replay$n.json" 2> "$OUT/bench_replay$n.err"; note "replay:$n" $?
        python -c "import json;d=json.load(open('$OUT/bench_replay$n.json'));print('replay $n', d['value'], d['ms_per_step'])"
      done ;;
    stats)
      ( cd /tmp && export TMPDIR=/tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/rocprof" -o c3 -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --quick > "$GRAFT_REPO_ROOT/$OUT/rocprof_bench.json" 2> "$GRAFT_REPO_ROOT/$OUT/rocprof_bench.err" ); note $stage $?
      find "$OUT/rocprof" -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} "$OUT/c3_kernel_stats.csv"
      find "$OUT/rocprof" -type f ! -name '*kernel_stats.csv' -delete 2>/dev/null
      head -12 "$OUT/c3_kernel_stats.csv" ;;
    pmc)
      timeout 1200 bash tools/pmc_h2.sh > "$OUT/conv_h2_pmc.txt" 2>&1; note "pmc:h2" $?
      grep -v '^pass' "$OUT/conv_h2_pmc.txt" | head -40 ;;
    *) echo "unknown stage $stage"; note "$stage" 99 ;;
  esac
done
cat "$OUT/status.txt"
