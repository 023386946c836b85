#!/usr/bin/env python
"""Host cost of building the result objects of a dense 64-frame step (283 persons + 95 players per frame, the bench's
synthetic checkpoints) and what CPython's cyclic collector adds to it — the measurement behind trackers.tracker.relaxed_gc.
No GPU.    python tools/eager_objects_bench.py"""
import gc
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from padel_analytics_amd.trackers.players_keypoints_tracker import PlayersKeypoints
from padel_analytics_amd.trackers.players_tracker import Players
from padel_analytics_amd.trackers.tracker import relaxed_gc

rng = np.random.default_rng(0)
xy = rng.uniform(0, 1280, (64, 283, 13, 2)).astype(np.float32)
rows = rng.uniform(0, 1000, (64, 95, 6)).astype(np.float32)
ids = np.arange(1, 96)


def step(keep):
    for i in range(64):
        keep.append(PlayersKeypoints(xy=xy[i], ratio=(1.0, 0.5625)).players_keypoints)
        keep.append(Players(rows=rows[i], ids=ids).players)


def run(label):
    keep, ts = [], []
    for _ in range(20):
        t = time.perf_counter()
        step(keep)
        ts.append(1e3 * (time.perf_counter() - t))
    print(f"{label:34s} mean {np.mean(ts):6.1f} ms / step   max {np.max(ts):6.1f}   ({64 * 378} objects per step, 20 steps kept alive)")
    del keep
    gc.collect()


run("default collector thresholds")
with relaxed_gc():
    run("inside relaxed_gc()")
gc.disable()
run("collector disabled (floor)")
gc.enable()
