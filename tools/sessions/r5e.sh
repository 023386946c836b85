set -u
O=gpurun_out/r5e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_yolo_parity.py -m gpu -q -x -k "pipeline or ticket or pose_parity or letterbox or netin or resize or bicubic" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
# verdict runs: the second fp32 implementation (conv_lds.hip) against the fp32 tap kernels; the fp16 fused SPPF
timeout 300 python tools/conv_bench.py --dtype f32 --tiles auto,T7,L7,L9 --reps 5 --shapes m.P3.bneck,m.P4.bneck,576,1152 > $O/f32_tap_vs_lds.txt 2>&1; grep -v amdgpu.ids $O/f32_tap_vs_lds.txt | head -8
for v in 1 5; do
  PADEL_FUSE_SPPF=$v timeout 400 python bench.py --workload c4 --steps 5 --warmup 2 --quick --engine-only --traffic none --dump-ops $O/ops_c4_sppf$v.csv > $O/bench_c4_sppf$v.json 2> $O/bench_c4_sppf$v.err
  echo "c4 fuse_sppf=$v rc=$? $(python -c "import json;d=json.load(open('$O/bench_c4_sppf$v.json'));print(d['value'], d['ms_per_step'])") pools: $(grep ',3,5,' $O/ops_c4_sppf$v.csv | awk -F, '{printf "%s %.3f ms  ", $1, $10}')"
done
bash tools/gpu_session.sh r5e bench_driver
