set -u
bash tools/gpu_session.sh r5n tests_all smoke bench_driver
