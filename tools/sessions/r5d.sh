set -u
mkdir -p gpurun_out/r5d
bash tools/gpu_session.sh r5d tests_all bench_short
