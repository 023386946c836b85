"""ByteTrack known answers (reference call sites players_tracker.py:311,367-369).

``tests/golden/bytetrack_golden.json`` is produced by an INDEPENDENT scalar implementation of the published
algorithm (``tests/golden/make_bytetrack_golden.py``: per-track Kalman objects, no stacked numpy); the product's
vectorised tracker must reproduce its keep-sets and ids on every frame of every scenario.  The scripted scenarios
carry hand-checkable events, asserted literally below (supervision 0.21-0.23 semantics, documented in
padel_analytics_amd/bytetrack.py: frame-1 tracks are confirmed at birth, later tracks at their first matched
update = "activation after 2 frames", public ids assigned at confirmation, lost tracks expire after
frame_rate/30*lost_track_buffer frames, first-association threshold 0.8 on the fused IoU x score cost)."""
import json
from pathlib import Path

import numpy as np
import pytest

from padel_analytics_amd import bytetrack
from padel_analytics_amd.detections import Detections

GOLD = json.loads((Path(__file__).parent / "golden" / "bytetrack_golden.json").read_text())


def _run(name):
    sc = GOLD[name]
    bt = bytetrack.ByteTrack(frame_rate=30, **sc["params"])
    out = []
    for f in sc["frames"]:
        n = len(f["conf"])
        det = Detections(np.array(f["xyxy"], np.float32).reshape(-1, 4), np.array(f["conf"], np.float32),
                         np.zeros(n, int))
        got = bt.update_with_detections(det)
        out.append(got)
    return sc, out


@pytest.mark.parametrize("name", sorted(GOLD))
def test_matches_independent_implementation(name):
    sc, out = _run(name)
    for i, (f, got) in enumerate(zip(sc["frames"], out)):
        assert got.tracker_id.tolist() == f["ids"], f"{name} frame {i}: ids {got.tracker_id.tolist()} != {f['ids']}"
        want = np.array(f["xyxy"], np.float32).reshape(-1, 4)[f["keep"]]
        assert np.array_equal(got.xyxy, want), f"{name} frame {i}: kept boxes differ (the detector's own boxes are returned)"


def test_hand_checked_events():
    ids = lambda name: [f["ids"] for f in GOLD[name]["frames"]]
    # a box first seen on frame 3 is reported from frame 4 on (unconfirmed for one frame), with the next dense id
    assert ids("two_players")[:5] == [[1, 2], [1, 2], [1, 2], [1, 2, 3], [1, 2, 3]]
    # 5 missing frames < lost_track_buffer 30: the same id comes back
    o = ids("occlusion")
    assert o[4] == [1, 2] and all(x == [1] for x in o[5:10]) and o[10] == [1, 2]
    # lost_track_buffer 3: missing for 4 frames -> expired; the box returns as a NEW track, confirmed one frame later
    e = ids("expiry")
    assert e[3] == [1, 2] and all(x == [1] for x in e[4:9]) and e[9] == [1, 3]
    # score .2 (< .25 activation threshold, > .1) keeps its track through the second association
    assert all(x == [1, 2] for x in ids("low_score"))
    # one-frame false positives never consume a public id
    s = ids("spurious")
    assert s[6] == [1] and s[7] == [1, 2] and max(max(x) for x in s) == 2


def test_state_does_not_grow():
    """the removed list held every track ever removed; now only the current frame's."""
    rng = np.random.default_rng(0)
    bt = bytetrack.ByteTrack(frame_rate=30, lost_track_buffer=2)
    for f in range(200):
        b = rng.uniform(0, 500, (6, 2))
        xyxy = np.concatenate([b, b + 50], 1).astype(np.float32)
        bt.update_with_detections(Detections(xyxy, np.full(6, 0.9, np.float32), np.zeros(6, int)))
    assert len(bt.removed) <= 20 and len(bt.lost) <= 40


# ---- the host-native (C++) tracker inside libpadel_hip.so: same fixtures, plus a long random equivalence run
def _native():
    from padel_analytics_amd import engine as E
    return E


@pytest.mark.parametrize("name", sorted(GOLD))
def test_native_matches_golden(name):
    E = _native()
    sc = GOLD[name]
    bt = E.NativeByteTrack(frame_rate=30, **sc["params"])
    nf = len(sc["frames"])
    stride = max(len(f["conf"]) for f in sc["frames"]) + 1
    boxes = np.zeros((nf, stride, 6), np.float32)
    counts = np.zeros(nf, np.int32)
    for i, f in enumerate(sc["frames"]):
        n = len(f["conf"])
        counts[i] = n
        boxes[i, :n, :4] = np.array(f["xyxy"], np.float32).reshape(-1, 4)
        boxes[i, :n, 4] = f["conf"]
    # half the frames in one call, the rest one by one: batching must not matter
    ids = np.concatenate([bt.update_batch(boxes[:nf // 2], counts[:nf // 2])] +
                         [bt.update_batch(boxes[i:i + 1], counts[i:i + 1]) for i in range(nf // 2, nf)])
    for i, f in enumerate(sc["frames"]):
        keep = np.nonzero(ids[i, :counts[i]] >= 0)[0].tolist()
        assert keep == f["keep"], f"{name} frame {i}"
        assert ids[i, keep].tolist() == f["ids"], f"{name} frame {i}"
    assert (ids[np.arange(stride)[None] >= counts[:, None]] == -1).all()


def test_native_equals_python_on_a_dense_random_stream():
    """~120 boxes per frame (what the synthetic bench weights produce), 64 frames, zone mask on: ids identical."""
    E = _native()
    rng = np.random.default_rng(11)
    n, nf, stride = 120, 64, 300
    pos = rng.uniform([50, 50], [1230, 670], (n, 2))
    vel = rng.uniform(-4, 4, (n, 2))
    boxes = np.zeros((nf, stride, 6), np.float32)
    counts = np.zeros(nf, np.int32)
    keep = np.zeros((nf, stride), np.uint8)
    for f in range(nf):
        pos = pos + vel + rng.normal(0, 0.7, pos.shape)
        alive = rng.random(n) > 0.1
        k = int(alive.sum())
        wh = np.stack([30 + np.arange(n) % 40, 80 + np.arange(n) % 60], 1)[alive]
        boxes[f, :k, :2] = pos[alive] - wh / 2
        boxes[f, :k, 2:4] = pos[alive] + wh / 2
        boxes[f, :k, 4] = rng.uniform(0.12, 0.97, k)
        counts[f] = k
        keep[f, :k] = rng.random(k) > 0.3
    ids_native = E.NativeByteTrack(frame_rate=30).update_batch(boxes, counts, keep)
    bt = bytetrack.ByteTrack(frame_rate=30)
    for f in range(nf):
        sel = np.nonzero(keep[f, :counts[f]])[0]
        det = Detections(boxes[f, sel, :4], boxes[f, sel, 4], np.zeros(len(sel), int))
        got = bt.update_with_detections(det)
        want = {tuple(boxes[f, i, :4].tolist()): int(ids_native[f, i]) for i in sel if ids_native[f, i] >= 0}
        have = {tuple(b.tolist()): int(t) for b, t in zip(got.xyxy, got.tracker_id)}
        assert have == want, f"frame {f}"


@pytest.mark.parametrize("mode", ["grid", "crowd", "duplicates", "gaps"])
def test_native_equals_python_on_ties_crowds_and_duplicates(mode):
    """Round 4 reorganised the native solver (first pass folded into the setup, visited lists, a separate pass for the pick
    under scipy's tie rule) and the cost builders: streams built to hit TIES — boxes on a grid (identical IoUs), scores
    rounded to quarters, exact duplicates of boxes, a crowd where most pairs overlap, frames without detections (tracks go
    lost and come back: more tracks than detections, the transposed problem) — must give the ids of the Python twin."""
    E = _native()
    rng = np.random.default_rng({"grid": 3, "crowd": 4, "duplicates": 5, "gaps": 6}[mode])
    n, nf, stride = 48, 40, 64
    pos = rng.uniform([50, 50], [1230, 670], (n, 2))
    vel = rng.uniform(-4, 4, (n, 2))
    if mode == "grid":
        pos, vel = np.round(pos / 40) * 40, vel * 0
    if mode == "crowd":
        pos = rng.uniform([500, 300], [700, 420], (n, 2))
    boxes = np.zeros((nf, stride, 6), np.float32)
    counts = np.zeros(nf, np.int32)
    for f in range(nf):
        pos = pos + vel + (0 if mode == "grid" else rng.normal(0, 0.7, pos.shape))
        alive = rng.random(n) > 0.1
        if mode == "gaps" and f % 7 in (3, 4):
            alive[:] = False
        k = int(alive.sum())
        wh = np.stack([30 + np.arange(n) % 40, 80 + np.arange(n) % 60], 1)[alive] if mode != "grid" else np.full((k, 2), [40.0, 90.0])
        boxes[f, :k, :2] = pos[alive] - wh / 2
        boxes[f, :k, 2:4] = pos[alive] + wh / 2
        sc = rng.uniform(0.05, 0.99, k)
        boxes[f, :k, 4] = np.round(sc * 4) / 4 if mode in ("grid", "duplicates") else sc
        if mode == "duplicates" and k > 3:
            boxes[f, 1, :4] = boxes[f, 0, :4]
            boxes[f, 3, :5] = boxes[f, 2, :5]
        counts[f] = k
    ids_native = E.NativeByteTrack(frame_rate=30, lost_track_buffer=5).update_batch(boxes, counts)
    bt = bytetrack.ByteTrack(frame_rate=30, lost_track_buffer=5)
    n_ids = 0
    for f in range(nf):
        k = counts[f]
        got = bt.update_with_detections(Detections(boxes[f, :k, :4], boxes[f, :k, 4], np.zeros(k, int)))
        # duplicates: compare as multisets of (box, id) pairs in output order of the kept rows
        want = sorted((tuple(boxes[f, i, :4].tolist()), int(ids_native[f, i])) for i in range(k) if ids_native[f, i] >= 0)
        have = sorted((tuple(b.tolist()), int(t)) for b, t in zip(got.xyxy, got.tracker_id))
        assert have == want, f"{mode}: frame {f}"
        n_ids += len(want)
    assert n_ids > nf
