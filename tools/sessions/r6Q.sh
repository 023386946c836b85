set -u
export W16=1
export PYTEST_SEL="tests/test_gpu_runner.py tests/test_gpu_pipeline.py tests/test_gpu_bench_config.py tests/test_gpu_baseline_configs.py"
bash tools/gpu_session.sh r6Q tests_sel bench_driver bench_c2d
