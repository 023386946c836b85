set -u
export W16=1
bash tools/gpu_session.sh r7b tests_all smoke bench_driver stats
