"""GPU end-to-end: TrackingRunner drives PlayerTracker + PlayerKeypointsTracker + BallTracker over a synthetic
clip exactly like the reference's main.py/runner.py (sequential trackers, batch sampler, JSON caches), and the
returned objects match the oracle run through the same host glue."""
import json

import numpy as np
import pytest
import torch

from oracle import synth_weights, tracknet_ref as tr, yolov8_ref as ref
from padel_analytics_amd import checkpoint, detections as D, video
from padel_analytics_amd.trackers import (BallTracker, PlayerKeypointsTracker, PlayerTracker, Players, PlayersKeypoints,
                                          TrackingRunner)
from tests import parity
from tests import synth  # noqa: F401  (registers the synthetic:// frame source)

pytestmark = pytest.mark.gpu


def test_runner_end_to_end(gpu_engine, tmp_path):
    src = "synthetic://?n=20&h=360&w=640&fps=30&seed=5"
    frames = list(video.get_video_frames_generator(src))
    # checkpoints (synthetic, calibrated on the clip)
    srcs = [f[..., ::-1] for f in frames[:4]]
    sd_p = synth_weights.calibrated_state_dict("n", 80, None, ref.preprocess(srcs, 640), 0.5, seed=3)
    checkpoint.save_checkpoint(tmp_path / "players.pt", sd_p, "detect", 80, None, "n", {0: "person"})
    from PIL import Image
    pil = [np.asarray(Image.fromarray(f[..., ::-1].copy()).resize((640, 640)))[..., ::-1] for f in frames[:4]]
    sd_k = synth_weights.calibrated_state_dict("n", 1, (13, 3), ref.preprocess(pil, 640), 0.25, seed=4)
    checkpoint.save_checkpoint(tmp_path / "pose.pt", sd_k, "pose", 1, (13, 3), "n", {0: "person"})
    checkpoint.save_checkpoint(tmp_path / "tracknet.pt", tr.synth_tracknet_state_dict(9), "tracknet",
                               param_dict={"seq_len": 8, "bg_mode": "concat"})
    zone = D.PolygonZone(np.array([[40, 40], [600, 40], [600, 340], [40, 340]]), frame_resolution_wh=(640, 360))
    players = PlayerTracker(str(tmp_path / "players.pt"), zone, batch_size=8, save_path=tmp_path / "players.json")
    pose = PlayerKeypointsTracker(str(tmp_path / "pose.pt"), 640, batch_size=8, load_path=None, save_path=tmp_path / "pose.json")
    ball = BallTracker(str(tmp_path / "tracknet.pt"), None, batch_size=8, median_max_sample_num=20, save_path=tmp_path / "ball.json")
    runner = TrackingRunner([players, pose, ball], src, tmp_path / "out.mp4")
    runner.run()
    assert set(runner.timings) == {"players_tracker", "players_keypoints_tracker", "ball_tracker"}
    assert len(players) == len(pose) == len(ball) == 20
    # JSON caches round-trip through the reference wire format
    for name, cls in (("players.json", Players), ("pose.json", PlayersKeypoints)):
        data = json.loads((tmp_path / name).read_text())
        assert len(data) == 20 and len(cls.from_json(data[0])) == len(data[0])
    # a second runner with load_path set skips inference (runner.py:187-191)
    players2 = PlayerTracker(str(tmp_path / "players.pt"), zone, batch_size=8, load_path=tmp_path / "players.json")
    assert len(players2) == 20
    # detections (before zone / ByteTrack) agree with the oracle on the first batch
    res = players.model.predict_frames(np.stack(frames[:8]), 0.5, 0.7, 640, classes=[0], channel_reverse=False)
    r32 = ref.predict(ref.YoloV8Ref(sd_p, 80, None), [f[..., ::-1] for f in frames[:8]], 0.5, 0.7, 640, classes=[0])
    n = len(res)
    boxes = np.zeros((n, 300, 6), np.float32); counts = np.zeros(n, np.int32)
    for i, r in enumerate(res):
        counts[i] = len(r.boxes); boxes[i, :counts[i]] = r.boxes.data
    rep = parity.compare_batch(r32, boxes, None, counts, 0.5, 0.7)
    assert rep["worst_px"] < 0.1 and rep["n"] > 0
    # every kept player is inside the zone and carries a ByteTrack id after the first frames
    ids = [p.id for pl in players.results.predictions[3:] for p in pl]
    assert all(i is not None for i in ids)


def test_court_keypoints_tracker_yolo(gpu_engine, tmp_path):
    """model_type='yolo': a 12-keypoint YOLOv8-pose graph with max_det=12 through the same engine."""
    from PIL import Image
    from padel_analytics_amd.trackers import KeypointsTracker
    frames = list(video.get_video_frames_generator("synthetic://?n=4&h=360&w=640&seed=9"))
    pil = [np.asarray(Image.fromarray(f[..., ::-1].copy()).resize((640, 640)))[..., ::-1] for f in frames]
    sd = synth_weights.calibrated_state_dict("n", 1, (12, 2), ref.preprocess(pil, 640), 0.5, seed=6, frac=0.02)
    checkpoint.save_checkpoint(tmp_path / "court.pt", sd, "pose", 1, (12, 2), "n", {0: "court"})
    t = KeypointsTracker(str(tmp_path / "court.pt"), batch_size=4, model_type="yolo")
    out = t.predict_and_update(iter(frames))
    assert len(out) == 4
    r = ref.predict(ref.YoloV8Ref(sd, 1, (12, 2)), pil, 0.5, 0.7, 640, classes=None, max_det=12)
    for kp, rr in zip(out, r):
        if len(rr["boxes"]) == 0:
            assert len(kp) == 0
            continue
        assert sorted(k.id for k in kp) == list(range(12))
        want = rr["kpts"][0]
        for i in range(12):
            k = kp[KeypointsTracker.POINTS_MAPPER[i]]
            assert abs(k.xy[0] - want[i, 0] * 640 / 640) < 0.1 and abs(k.xy[1] - want[i, 1] * 360 / 640) < 0.1


def test_recycled_output_arrays_hold_the_same_results(gpu_engine):
    """``Model.yolo_infer(reuse_outputs=True)`` (what the trackers' batch loops ask for): results land in one of OUT_RING
    page-locked sets the model recycles — same numbers as fresh arrays, a set stays untouched while OUT_RING - 1 further
    calls run, and is the one handed out again after that."""
    from padel_analytics_amd import engine as E, graph as G, yolo_arch
    m = E.Model(gpu_engine, G.build_yolov8(yolo_arch.synth_state_dict("n", 1, (13, 3), seed=2, cls_bias=1.0), 1, (13, 3), dtype=E.graph_dtype()))
    m.set_max_batch(3)
    kw = dict(imgsz=320, conf=0.25, iou=0.7)
    batches = [synth.synthetic_frames(3, 180, 320, seed=20 + i) for i in range(E.Model.OUT_RING + 1)]
    fresh = [m.yolo_infer(b, 3, 180, 320, **kw) for b in batches]
    held = []
    for i, b in enumerate(batches):
        got = m.yolo_infer(b, 3, 180, 320, reuse_outputs=True, **kw)
        for a, w in zip(got, fresh[i]):
            assert np.array_equal(a, w)
        for j, (g_old, i_old) in enumerate(held[-(E.Model.OUT_RING - 1):]):          # still what they were
            for a, w in zip(g_old, fresh[i_old]):
                assert np.array_equal(a, w), (i, i_old)
        held.append((got, i))
    assert held[E.Model.OUT_RING][0][0] is held[0][0][0]                          # the first set came round again
    assert int(sum(f[2].sum() for f in fresh)) > 0
    m.close()


def test_sharded_runner_on_the_real_engine_equals_the_sequential_run(gpu_engine, tmp_path):
    """``TrackingRunner(distributed=True)`` — the path ``bench.py --gpus N`` times since round 5 — on the real engine with a
    world of one (no process group: every collective degenerates, everything else is the real thing): ``predict_partial``
    through the two-call device stage over HBM-resident frames, the partials packed to arrays and unpacked again
    (``Tracker.pack_partials``), ByteTrack / containers in ``merge_partials`` — must serialise to exactly what the
    reference-order run produces, for the players, pose and ball-detector trackers, with eager objects on."""
    from padel_analytics_amd import trackers as T
    from padel_analytics_amd.trackers import BallDetectTracker
    from padel_analytics_amd import yolo_arch
    frames = synth.synthetic_frames(13, 360, 640, seed=15)
    clip = video.DeviceClip(gpu_engine, frames)                              # 13 frames, batch 5: a short last batch
    checkpoint.save_checkpoint(tmp_path / "players.pt", yolo_arch.synth_state_dict("n", 80, None, seed=3, cls_bias=0.5), "detect", 80, None, "n", {0: "person"})
    checkpoint.save_checkpoint(tmp_path / "pose.pt", yolo_arch.synth_state_dict("n", 1, (13, 3), seed=4, cls_bias=0.5), "pose", 1, (13, 3), "n", {0: "person"})
    checkpoint.save_checkpoint(tmp_path / "ball.pt", yolo_arch.synth_state_dict("n", 1, None, seed=5, cls_bias=0.5), "detect", 1, None, "n", {0: "ball"})
    zone = D.PolygonZone(np.array([[40, 40], [600, 40], [600, 340], [40, 340]]), frame_resolution_wh=(640, 360))

    def run(**kw):
        players = PlayerTracker(str(tmp_path / "players.pt"), zone, batch_size=5)
        pose = PlayerKeypointsTracker(str(tmp_path / "pose.pt"), 640, batch_size=5)
        ball = BallDetectTracker(str(tmp_path / "ball.pt"), batch_size=5)
        r = TrackingRunner([players, pose, ball], clip, tmp_path / "out.mp4", **kw)
        r.run()
        out = {str(t): [o.serialize() for o in t.results] for t in (players, pose, ball)}
        for t in (players, pose, ball):
            t.model.close()
        return out

    T.set_eager_objects(True)
    try:
        want = run()
        got = run(distributed=True)
    finally:
        T.set_eager_objects(False)
        clip.free()
    assert set(got) == set(want)
    for k in want:
        assert len(want[k]) == 13
        assert json.dumps(got[k]) == json.dumps(want[k]), k
    assert sum(len(f) for f in want["players_tracker"]) > 0 and sum(len(f) for f in want["players_keypoints_tracker"]) > 0
