set -u
export W16=1
bash tools/gpu_session.sh r6P tests_all smoke bench_driver stats bench_c2d bench_c4d pmc
