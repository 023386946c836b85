"""Worker of tests/test_distributed_gloo.py (one process per rank, gloo).  Modes:
  blob   — broadcast_blob / shard_range / gather_results plumbing;
  overflow / fullrange — the runner mode with models that carry the fp32-equivalent arithmetic switch (h2 -> bx3): in
           "overflow" ONE rank's models leave the fp16 range in the middle of their shard (a batch tracker switches itself
           inside the offending batch, the stream tracker raises engine.RangeOverflow); every rank must then repeat its
           shard on the full-range path, and the merged predictions must equal a run that was on that path from the start
           ("fullrange") — ADVICE r3: the other ranks used to block in the gather for ever;
  runner — the REAL tracker host logic (TrackingRunner, PlayerTracker + PolygonZone + native ByteTrack,
           PlayerKeypointsTracker, BallTracker with the 7-frame TrackNet halo + InpaintNet, BallDetectTracker) over
           a FAKE engine whose per-frame outputs are a deterministic function of the frame pixels and whose ball
           session reproduces the stream semantics of pa_ball_feed (head means / weighted steady state / tail
           means, via oracle/ball_ref.py:ensemble).  Rank 0 writes the serialized predictions as JSON.
Usage: python tests/dist_worker.py MODE RANK WORLD PORT OUT"""
import json
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from tests import synth  # noqa: F401  (registers the synthetic:// frame source)


def frame_seed(f) -> int:
    a = np.asarray(f)
    return int(a[::7, ::5].astype(np.int64).sum() % 2147483647)


OVERFLOW_RANK = -1        # "overflow" mode: the rank whose models leave the fp16 range once
MY_RANK = 0


class FakeYOLO:
    """Stands in for padel_analytics_amd.yolo.YOLO: same infer_frames contract, outputs keyed on frame content."""

    def __init__(self, model_path, engine=None, half=False):
        self.task = "pose" if "pose" in str(model_path) else "detect"
        self.kpt_shape = (13, 3) if self.task == "pose" else None
        self.names = {0: "person"}
        self.calls = 0

    def to(self, device): return self

    # the arithmetic switch of yolo.YOLO, present only in the overflow / fullrange modes (FakeYOLO.fp32_mode is set there)
    def set_fp32_mode(self, mode): self.fp32_mode = mode

    def _shift(self):
        """What the arithmetic does to the numbers (visible, so that a merge of two arithmetics cannot pass): the second
        call of the overflowing rank's h2 model switches to bx3 like yolo.YOLO.infer_frames does."""
        if not hasattr(self, "fp32_mode"):
            return 0.0
        self.calls += 1
        if self.fp32_mode == "h2" and MY_RANK == OVERFLOW_RANK and self.calls == 2:
            self.fp32_mode, self.fell_back = "bx3", True
        return 0.25 if self.fp32_mode == "bx3" else 0.0

    def infer_frames(self, frames, conf, iou, imgsz, classes=None, max_det=300, *, channel_reverse, pil_stretch=False, reuse_outputs=False):
        frames = list(frames)
        n = len(frames)
        h, w = frames[0].shape[:2]
        shift = self._shift()
        boxes = np.zeros((n, max_det, 6), np.float32)
        counts = np.zeros(n, np.int32)
        nk = 39 if self.task == "pose" else 0
        kpts = np.zeros((n, max_det, nk), np.float32) if nk else None
        for i, f in enumerate(frames):
            rng = np.random.default_rng(frame_seed(f))
            # 5 slowly drifting "players" (position from the low bits of the frame statistics) + content-keyed extras
            base = np.array([[0.15, 0.3], [0.35, 0.6], [0.55, 0.4], [0.75, 0.7], [0.5, 0.85]]) * [w, h]
            k = min(max_det, 5 + int(rng.integers(0, 3)))
            c = np.concatenate([base + rng.normal(0, 1.5, base.shape), rng.uniform([0, 0], [w, h], (max(k - 5, 0), 2))])[:k]
            wh = np.array([w * 0.08, h * 0.25])
            boxes[i, :k, :2] = c - wh / 2 + shift
            boxes[i, :k, 2:4] = c + wh / 2 + shift
            boxes[i, :k, 4] = np.sort(rng.uniform(max(conf, 0.3), 0.95, k))[::-1]
            counts[i] = 0 if (self.task == "detect" and max_det == 1 and rng.random() < 0.2) else k
            if nk:
                kpts[i, :k] = rng.uniform(0, imgsz, (k, nk)).astype(np.float32)
                kpts[i, :k, 2::3] = rng.uniform(0, 1, (k, 13))
        return boxes, kpts, counts, (h, w), int(imgsz), 1 if pil_stretch else 0


class FakeModel:
    overflows_left = 0        # "overflow" mode: how many h2 TrackNet runs of this process still leave the fp16 range

    def __init__(self, engine, graph, blob=None, **kw): self.max_batch = 64; self.graph = graph
    def set_max_batch(self, n): self.max_batch = int(n)
    def close(self): pass

    def take_overflow(self):
        from padel_analytics_amd import graph as G
        if FakeModel.overflows_left > 0 and self.graph.dtype == G.DTYPE_H2:
            FakeModel.overflows_left -= 1
            return True
        return False


class FakeBallSession:
    """pa_ball_feed's stream semantics on scalar 'heat maps': window g slot s = f(frame g+s, g)."""

    def __init__(self, model, h, w):
        from padel_analytics_amd import graph as G
        self.max_feed = model.max_batch
        self.h, self.w = h, w
        self.u, self.bg = [], 0.0
        self.shift = 0.0 if model.graph.dtype == G.DTYPE_H2 else 0.013       # the arithmetic is visible in the heat maps

    def set_background(self, med): self.bg = float(np.asarray(med, np.float64).mean()) / 255.0; self.u = []

    def background_from_frames(self, frames, want_median=False, n=None):
        rgb = np.asarray(frames)[..., ::-1]
        med = np.median(rgb, 0).astype(np.uint8)
        self.set_background(med)
        return med if want_median else None

    def feed(self, frames, flush=False, want_heat=False, want_rects=False, want_masks=True, n=None):
        from oracle import ball_ref
        if frames is not None:
            self.u += [(frame_seed(f) % 1000) / 1000.0 + self.shift for f in frames]
        rects = np.zeros((0, 4), np.int32)
        if flush and len(self.u) >= 8:
            F = len(self.u)
            y = np.zeros((F - 7, 8, 1, 1), np.float32)
            for g in range(F - 7):
                for s in range(8):
                    # a window's output for its slot s depends on the whole window (like TrackNet's): frame g+s,
                    # the window's first and last frame, and the background
                    y[g, s] = np.float32(0.5 * self.u[g + s] + 0.2 * self.u[g] + 0.15 * self.u[g + 7] + 0.15 * self.bg)
            heat = ball_ref.ensemble(y)[:, 0, 0]
            rects = np.array([[int(v * 400), int(v * 200), 10, 12] if v > 0.5 else [0, 0, 0, 0] for v in heat], np.int32)
        return None, None, rects

    def close(self): pass


def run_runner(rank, world, out, mode="runner"):
    import tempfile
    global OVERFLOW_RANK, MY_RANK
    MY_RANK = rank
    from oracle import tracknet_ref as tr
    from padel_analytics_amd import checkpoint, detections as D, engine as E, yolo
    from padel_analytics_amd.trackers import (BallDetectTracker, BallTracker, PlayerKeypointsTracker, PlayerTracker,
                                              TrackingRunner)
    from padel_analytics_amd.trackers import ball_detect_tracker, players_keypoints_tracker, players_tracker
    for mod in (players_tracker, players_keypoints_tracker, ball_detect_tracker, yolo):
        mod.YOLO = FakeYOLO
    E.Model, E.BallSession = FakeModel, FakeBallSession
    E.default_engine = lambda *a, **k: None
    if mode in ("overflow", "fullrange"):
        FakeYOLO.fp32_mode, FakeYOLO.half, FakeYOLO.fell_back = "h2", False, False
        if mode == "overflow":
            OVERFLOW_RANK = world - 1
    src = "synthetic://?n=45&h=72&w=128&fps=30&seed=3"
    tmp = Path(tempfile.mkdtemp())
    checkpoint.save_checkpoint(tmp / "tracknet.pt", tr.synth_tracknet_state_dict(9), "tracknet",
                               param_dict={"seq_len": 8, "bg_mode": "concat"})
    checkpoint.save_checkpoint(tmp / "inpaint.pt", tr.synth_inpaintnet_state_dict(4), "inpaintnet", param_dict={"seq_len": 16})
    zone = D.PolygonZone(np.array([[10, 10], [118, 10], [118, 66], [10, 66]]), frame_resolution_wh=(128, 72))
    res = {}
    for ball_cls in ("tracknet", "detect"):
        ball = (BallTracker(str(tmp / "tracknet.pt"), str(tmp / "inpaint.pt"), batch_size=8, median_max_sample_num=20)
                if ball_cls == "tracknet" else BallDetectTracker("ball.pt", batch_size=8))
        trackers = [PlayerTracker("players.pt", zone, batch_size=8), PlayerKeypointsTracker("pose.pt", 640, batch_size=8), ball]
        if mode == "fullrange":
            for t in trackers:
                t.use_full_range()
                assert t.full_range
        elif mode == "overflow":
            assert not any(t.full_range for t in trackers)
            FakeModel.overflows_left = 1 if rank == OVERFLOW_RANK else 0
        runner = TrackingRunner(trackers, src, tmp / "out.mp4", distributed=world > 1)
        runner.run()
        if mode == "overflow":
            assert all(t.full_range for t in trackers), "every rank ends on the full-range arithmetic"
        if rank == 0:
            res[ball_cls] = {str(t): [o.serialize() for o in t.results.predictions] for t in trackers}
    if rank == 0:
        Path(out).write_text(json.dumps(res))


def run_blob(rank, world, out):
    from padel_analytics_amd import dist as D
    n = 37
    blob = np.random.default_rng(0).normal(size=1000).astype(np.float32) if rank == 0 else None
    got = D.broadcast_blob(blob, 1000, src=0)
    lo, hi = D.shard_range(n, rank, world)
    allr = D.gather_results([(i, float(got[i])) for i in range(lo, hi)], dst=0)
    med = D.broadcast_array(np.arange(24, dtype=np.uint8).reshape(2, 4, 3) if rank == 0 else None, (2, 4, 3), np.uint8)
    uid = D.share_unique_id(lambda: b"\x07" * 128)
    assert med.tolist() == np.arange(24).reshape(2, 4, 3).tolist() and uid == b"\x07" * 128
    # the packed gather of the sharded runner: ragged per-frame arrays as (counts, rows), uneven and EMPTY shards, mixed dtypes
    from padel_analytics_amd.trackers.tracker import pack_ragged, unpack_ragged
    items = [np.full((i % 4, 6), i, np.float32) for i in range(lo, hi)] if rank != 1 else []     # rank 1 contributes nothing
    parts = D.gather_arrays(pack_ragged(items) + [np.arange(rank + 2, dtype=np.int64)], dst=0)
    packed_ok = None
    if rank == 0:
        back = [x for a in parts for x in unpack_ragged(a[:2])]
        want = [np.full((i % 4, 6), i, np.float32) for r in range(world) if r != 1 for i in range(*D.shard_range(n, r, world))]
        packed_ok = (len(back) == len(want) and all(a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b) for a, b in zip(back, want))
                     and [a[2].tolist() for a in parts] == [list(range(r + 2)) for r in range(world)])
    else:
        assert parts is None
    if rank == 0:
        Path(out).write_text(json.dumps({"head": got[:5].tolist(), "order": [a[0] for a in allr], "packed_ok": packed_ok}))


def main():
    mode, rank, world, port, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    if world > 1:
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    if mode == "blob":
        run_blob(rank, world, out)
    else:
        run_runner(rank, world, out, mode)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
