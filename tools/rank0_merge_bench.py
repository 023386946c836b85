#!/usr/bin/env python
"""What rank 0 does per step of the sharded runner at N ranks, measured WITHOUT GPUs: N gloo processes, every
rank holds bench-size partial results of its 64-frame batch — players: ~95 boxes per frame, pose: ~283 persons x 13 keypoints per
frame (the synthetic checkpoints' densities), ball: one row — and the loop of ``TrackingRunner._predict_sharded`` is replayed
without the device stage: ``pack_partials`` -> ``dist.gather_arrays`` (gloo here; ``pa_engine_gather`` over RCCL on the GPU box)
-> ``unpack_partials`` -> ``merge_partials`` (ByteTrack over ALL frames in global order for the players, the result containers
with their array-backed objects for the pose tracker) on rank 0.

    python tools/rank0_merge_bench.py --ranks 8 --steps 5          # spawns the ranks itself (127.0.0.1)

Prints, per tracker, rank 0's seconds per step (one step = N x 64 frames) for pack / gather / unpack / merge, next to the device
step it has to stay under (68.5 ms per 64 frames and GPU at round 5: N GPUs deliver N x 64 frames per ~68 ms)."""
import argparse
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

B = 64


def make_partials(kind: str, rng, first_frame: int):
    if kind == "players":                             # ~95 boxes that persist from frame to frame (the bench's synthetic rectangles do) and drift
        out = []
        base = np.random.default_rng(7)               # the same scene on every rank: tracks continue across shard borders
        n0 = 95
        cx0, cy0 = base.uniform(100, 1180, n0), base.uniform(100, 620, n0)
        w, h = base.uniform(30, 80, n0), base.uniform(60, 160, n0)
        vx, vy = base.uniform(-1.5, 1.5, n0), base.uniform(-1.0, 1.0, n0)
        for f in range(B):
            t = first_frame + f
            keep = rng.uniform(0, 1, n0) < 0.97        # a few detections drop out per frame
            cx, cy = (cx0 + vx * t)[keep], (cy0 + vy * t)[keep]
            out.append(np.stack([cx - w[keep] / 2, cy - h[keep] / 2, cx + w[keep] / 2, cy + h[keep] / 2,
                                 rng.uniform(0.5, 0.95, int(keep.sum())), np.zeros(int(keep.sum()))], 1).astype(np.float32))
        return out
    if kind == "pose":
        out = []
        for f in range(B):
            n = int(rng.integers(270, 296))
            xy = rng.uniform(0, 1280, (n, 26))
            out.append(np.concatenate([xy, np.tile([1.0, 0.5625], 13)[None, :]], 0))
        return out
    return [(float(rng.uniform(0, 1280)), float(rng.uniform(0, 720)), 1) for _ in range(B)]          # BallDetectTracker: (x, y, visibility)


def worker(a):
    import torch.distributed as dist
    from padel_analytics_amd import dist as D
    from padel_analytics_amd.trackers import PlayerTracker, PlayerKeypointsTracker, BallDetectTracker
    from padel_analytics_amd.trackers.tracker import relaxed_gc
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{a.port}", rank=rank, world_size=world)
    rng = np.random.default_rng(100 + rank)

    class _Model:                                     # what merge_partials reads of the model object
        kpt_shape = (13, 3)

    trackers = {}
    pt = object.__new__(PlayerTracker)
    pt.batch_size = B
    trackers["players"] = pt
    kt = object.__new__(PlayerKeypointsTracker)
    kt.model = _Model()
    trackers["pose"] = kt
    trackers["ball"] = object.__new__(BallDetectTracker)
    rows = {}
    for name, tr in trackers.items():
        acc = {"pack": 0.0, "gather": 0.0, "unpack": 0.0, "merge": 0.0, "bytes_per_rank": 0}
        if name == "players":
            tr.byte_track = None
            if rank == 0:
                try:                                  # the native ByteTrack (libpadel_hip.so's host code loads without a GPU)
                    from padel_analytics_amd.engine import NativeByteTrack
                    tr.byte_track = NativeByteTrack(frame_rate=30)
                except Exception as exc:              # measured without the association step rather than not at all
                    print(f"rank 0: native ByteTrack not available ({exc!r}): merge timed without it", file=sys.stderr)
        for step in range(a.steps + 1):
            partial = make_partials(name, rng, step * world * B + rank * B)
            dist.barrier()
            t0 = time.perf_counter()
            packed = tr.pack_partials(partial)
            t1 = time.perf_counter()
            parts = D.gather_arrays(packed, dst=0)
            t2 = time.perf_counter()
            if rank == 0:
                allp = [x for p in parts for x in tr.unpack_partials(p)]
                t3 = time.perf_counter()
                with relaxed_gc():
                    try:
                        merged = tr.merge_partials(allp)
                    except Exception as exc:
                        if step == 0:
                            print(f"{name}: merge_partials not runnable here ({exc!r}); timing the containers only", file=sys.stderr)
                        merged = allp
                t4 = time.perf_counter()
                assert len(merged) == world * B
                if step > 0:                          # step 0 warms up (imports, gloo buffers)
                    acc["pack"] += t1 - t0; acc["gather"] += t2 - t1; acc["unpack"] += t3 - t2; acc["merge"] += t4 - t3
                    acc["bytes_per_rank"] = int(sum(p.nbytes for p in packed))
        if rank == 0:
            rows[name] = {k: (round(1e3 * v / a.steps, 3) if k != "bytes_per_rank" else v) for k, v in acc.items()}
            rows[name]["total_ms_per_step"] = round(sum(rows[name][k] for k in ("pack", "gather", "unpack", "merge")), 3)
    if rank == 0:
        tot = round(sum(r["total_ms_per_step"] for r in rows.values()), 3)
        out = {"ranks": world, "frames_per_step": world * B, "steps": a.steps, "ms_per_step_rank0": rows, "sum_ms_per_step": tot,
               "device_step_ms_it_must_stay_under": a.device_ms,
               "transport": "gloo over loopback (pa_engine_gather over RCCL / xGMI on the GPU box)", "cpus": os.cpu_count()}
        print(json.dumps(out))
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--port", type=int, default=29571)
    ap.add_argument("--device-ms", type=float, default=68.5)
    ap.add_argument("--worker", action="store_true")
    a = ap.parse_args()
    if a.worker:
        return worker(a)
    procs = []
    for r in range(a.ranks):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(a.ranks), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, __file__, "--worker", "--ranks", str(a.ranks), "--steps", str(a.steps), "--port", str(a.port),
                                       "--device-ms", str(a.device_ms)], env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate(timeout=900)
    for p in procs[1:]:
        p.wait(timeout=900)
    print(out.strip())


if __name__ == "__main__":
    main()
