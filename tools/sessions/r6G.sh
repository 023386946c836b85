set -u
O=gpurun_out/r6G; mkdir -p $O
bash tools/gpu_session.sh r6G tests_h2 bench_short
