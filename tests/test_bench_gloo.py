"""``bench.py --gpus N`` without a GPU: ``main()`` runs under gloo with world 2 over the fake engine of
tests/fake_engine.py, started exactly the way the driver starts it (one process per rank, RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* in the environment).  Checked: ONE JSON line, from rank 0 only; ``n_gpus``, ``steps``, ``warmup`` echo the
command line; ``value`` = world x B x K / (the max-over-ranks time the line reports as ms_per_step); ranks != 0 build
their models EMPTY and end up with rank 0's weights through the broadcast hook alone."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_main_world2_gloo_fake_engine(tmp_path):
    world, B, K, Wm = 2, 4, 2, 1
    port = str(_free_port())
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", str(world), "--steps", str(K), "--warmup", str(Wm), "--fake-engine",
           "--batch", str(B), "--height", "72", "--width", "128", "--scales", "players=n,ball=n,pose=n"]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen(cmd, cwd=tmp_path, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{se[-3000:]}"
    assert outs[1][0].strip() == "", "only rank 0 prints"
    lines = [l for l in outs[0][0].splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["fake_engine"] is True and d["n_gpus"] == world and d["steps"] == K and d["warmup"] == Wm
    assert d["metric"].startswith("frames/sec") and d["unit"] == "frames/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["config"]["frames_per_gpu_per_step"] == B
    # value = whole-job frames / (max over ranks of the timed region)
    assert abs(d["value"] - world * B * K / (d["ms_per_step"] * K / 1e3)) <= 0.02 * d["value"]
    # N > 1: `value` is the SHARDED runner (one clip of world x K x B frames, TrackingRunner(distributed=True): packed gather to
    # rank 0, ByteTrack there); the independent-replica number stays beside it
    rep = d["replica_runners"]
    assert abs(rep["value"] - world * B * K / (rep["ms_per_step"] * K / 1e3)) <= 0.02 * rep["value"]
    assert "distributed=True" in d["config"]["timed_path_n_gpus"]
    assert "strings_over_120_chars" not in d, d.get("strings_over_120_chars")          # the driver cuts longer strings in the fields it keeps
    assert d["config"]["tracked_players_rank0"] > 0                     # rank 0 holds the merged results of ALL shards
    assert d["config"]["frames_with_results_rank0"] == {"players": world * K * B, "ball": world * K * B, "pose": world * K * B}
    assert set(d["config"]["runner_seconds_per_tracker_rank0"]) == {"players_tracker", "ball_tracker", "players_keypoints_tracker"}
    e = d["engine_only"]
    assert abs(e["value"] - world * B * K / (e["ms_per_step"] * K / 1e3)) <= 0.02 * e["value"]
    # three trackers x (1 warm-up + K) fake steps of 10 ms each, at least: the timed region is real
    assert e["ms_per_step"] >= 3 * 10.0 * 0.9
    r0, r1 = d["ranks"]
    assert (r0["rank"], r1["rank"]) == (0, 1)
    assert sum("model created empty=False" in x for x in r0["log"]) == 3 and not any("empty=True" in x for x in r0["log"])
    assert sum("model created empty=True" in x for x in r1["log"]) == 3, r1["log"]
    assert sum("bcast root=0 had_weights=False" in x for x in r1["log"]) == 3 and sum("bcast root=0 had_weights=True" in x for x in r0["log"]) == 3
    assert any("comm_init nranks=2 rank=1" in x for x in r1["log"])
    # round 6: the sharded runner's gathers travel over the ENGINE's communicator (pa_engine_gather on the GPU box), not torch's
    assert any("gather_bytes via the engine's communicator" in x for x in r0["log"]) and any("gather_bytes via the engine's communicator" in x for x in r1["log"])
    assert r1["weight_checksums"] == r0["weight_checksums"] and all(v != 0 for v in r0["weight_checksums"].values())
    # the sharded runner's device stage goes through the two-call form (submit_sample / collect_sample -> yolo_submit / yolo_wait)
    assert "submit" in r0["log"] and "submit" in r1["log"]
