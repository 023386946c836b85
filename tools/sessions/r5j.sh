set -u
bash tools/gpu_session.sh r5j tests_all smoke bench_driver stats pmc bench_c2d bench_c4d
