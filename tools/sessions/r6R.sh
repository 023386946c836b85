set -u
O=gpurun_out/r6R; mkdir -p $O
for d in 8 1 8; do
  PADEL_HOST_QUEUE_DEPTH=$d timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-reference-default --traffic none > $O/bench_d$d.json 2> $O/bench_d$d.err
  python - $O/bench_d$d.json $d <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); e = d["value_eager_objects"]
print("depth", sys.argv[2], "value", d["value"], "lazy", d["value_lazy_objects"]["value"], "engine-only", d["engine_only"]["value"],
      "realistic eager/lazy", e["realistic_detections"]["value"], e["realistic_detections"]["value_lazy_objects"], "all records", e["timed_checkpoints_all_records"]["value"],
      "host", d["host_frames"]["sequential_frames_per_s"], d["host_frames"]["fanout_frames_per_s"], "per tracker", d["config"]["runner_seconds_per_tracker_rank0"])
PY
done
