"""N>1 path on CPU (gloo, one OS process per rank, launched like torch.distributed.run would).

* plumbing: frames shard into contiguous blocks, the weight blob reaches every rank bit-exactly through the
  one-time broadcast, results gather back in global frame order, the RCCL unique id travels through the store;
* the sharded TrackingRunner: the REAL tracker host logic over a fake engine must give, for
  world 2 and 3 (uneven shards), exactly the predictions of the single-process run — ByteTrack ids assigned on
  rank 0 in global frame order, the TrackNet 7-frame halo on both sides of every shard boundary, the background
  median shared from rank 0, InpaintNet over the gathered trajectory."""
import json
import socket
import subprocess
import sys
from pathlib import Path

import pytest

from padel_analytics_amd import dist as D
from tests import synth  # noqa: F401  (registers the synthetic:// frame source)

WORKER = str(Path(__file__).with_name("dist_worker.py"))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(mode, world, out):
    port = str(_free_port())
    procs = [subprocess.Popen([sys.executable, WORKER, mode, str(r), str(world), port, str(out)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, log) in enumerate(zip(procs, logs)):
        assert p.returncode == 0, f"rank {r} failed:\n{log[-3000:]}"
    return json.loads(Path(out).read_text())


def test_shard_ranges_cover_exactly():
    for n in (0, 1, 7, 64, 255, 256):
        for w in (1, 2, 3, 8):
            r = [D.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_broadcast_and_gather_world2(tmp_path):
    import numpy as np
    got = _launch("blob", 2, tmp_path / "blob.json")
    want = np.random.default_rng(0).normal(size=1000).astype(np.float32)[:5].tolist()
    assert got["head"] == want
    assert got["order"] == list(range(37))
    assert got["packed_ok"] is True            # dist.gather_arrays: (counts, rows) per rank, an empty shard, a second dtype


@pytest.fixture(scope="module")
def single_process_reference(tmp_path_factory):
    return _launch("runner", 1, tmp_path_factory.mktemp("w1") / "w1.json")


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_runner_equals_single_process(single_process_reference, tmp_path, world):
    ref = single_process_reference
    got = _launch("runner", world, tmp_path / f"w{world}.json")
    for variant in ("tracknet", "detect"):
        for name in ("players_tracker", "players_keypoints_tracker", "ball_tracker"):
            a, b = ref[variant][name], got[variant][name]
            assert len(a) == len(b) == 45, (variant, name, len(a), len(b))
            for i, (x, y) in enumerate(zip(a, b)):
                assert x == y, f"{variant}/{name} frame {i}: world {world} differs from the single-process run"
    # the reference run is not trivial: players carry ids, balls are seen, keypoints exist
    assert any(p["id"] for fr in ref["tracknet"]["players_tracker"] for p in fr)
    assert sum(b["visibility"] for b in ref["tracknet"]["ball_tracker"]) > 5
    assert sum(b["visibility"] for b in ref["detect"]["ball_tracker"]) > 5


def test_sharded_runner_agrees_on_the_full_range_fallback(tmp_path):
    """One rank's models leave the fp16 range in the middle of its shard: nobody hangs, every rank repeats its
    shard on the bf16x3 path and the merged predictions are those of a run that was on that path from the first frame."""
    want = _launch("fullrange", 1, tmp_path / "full.json")
    got = _launch("overflow", 2, tmp_path / "over.json")
    plain = _launch("runner", 1, tmp_path / "plain.json")
    for variant in ("tracknet", "detect"):
        for name in ("players_tracker", "players_keypoints_tracker", "ball_tracker"):
            assert got[variant][name] == want[variant][name], (variant, name)
    # the fake arithmetic is visible: the h2 run differs, so a result merged from two arithmetics could not have passed
    assert plain["detect"]["players_tracker"] != want["detect"]["players_tracker"]


def test_runner_clamps_end_beyond_the_clip():
    """n_available is what the generator will yield: an `end` past the last frame must not inflate the shards."""
    from padel_analytics_amd.trackers import TrackingRunner
    src = "synthetic://?n=20&h=36&w=64&fps=30&seed=1"
    assert TrackingRunner([], src, "out.mp4", start=0, end=50).n_available == 20
    assert TrackingRunner([], src, "out.mp4", start=5, end=50).n_available == 15
    assert TrackingRunner([], src, "out.mp4", start=5, end=12).n_available == 7
    assert TrackingRunner([], src, "out.mp4", start=5).n_available == 15
    assert TrackingRunner([], src, "out.mp4", start=30, end=50).n_available == 0
