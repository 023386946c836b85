"""The device stage in two halves (pa_yolo_submit / pa_yolo_wait, round 4): batch k + 1 is queued on the GPU before batch k is
collected.  Results must be those of the synchronous call — at the C-ABI, through the trackers' batch loop, and when an h2
model overflows in the middle of a pipelined clip (every batch then comes from the bf16x3 model, like the synchronous path)."""
import numpy as np
import pytest

from padel_analytics_amd import checkpoint, detections as D, engine as E, graph as G, video, yolo_arch
from padel_analytics_amd.trackers import PlayerKeypointsTracker, PlayerTracker
from tests import synth

pytestmark = pytest.mark.gpu


def test_submit_wait_equals_infer_with_two_tickets_in_flight(gpu_engine):
    kpt = (13, 3)
    m = E.Model(gpu_engine, G.build_yolov8(yolo_arch.synth_state_dict("n", 1, kpt, seed=2, cls_bias=1.0), 1, kpt, dtype=E.graph_dtype()))
    m.set_max_batch(4)
    h, w = 180, 320
    frames = synth.synthetic_frames(12, h, w, seed=31)
    clip = video.DeviceClip(gpu_engine, frames)
    kw = dict(imgsz=320, conf=0.25, iou=0.7, classes=[0])
    views = [clip.buffer.view(i * 4 * clip.frame_bytes, 4 * clip.frame_bytes) for i in range(3)]
    want = [m.yolo_infer(v, 4, h, w, **kw) for v in views]
    t0 = m.yolo_submit(views[0], 4, h, w, **kw)
    t1 = m.yolo_submit(views[1], 4, h, w, **kw)
    got0 = m.yolo_wait(t0)
    t2 = m.yolo_submit(views[2], 4, h, w, **kw)
    got = [got0, m.yolo_wait(t1), m.yolo_wait(t2)]
    for g, wnt in zip(got, want):
        assert g[3] is False
        for a, b in zip(g[:3], wnt):
            assert np.array_equal(a, b)
    assert sum(int(x[2].sum()) for x in want) > 0
    # a ticket can be collected once; PA_MAX_INFLIGHT bounds what is outstanding
    with pytest.raises(E.EngineError):
        m.yolo_wait(t2)
    ts = [m.yolo_submit(views[i % 3], 4, h, w, **kw) for i in range(E.Model.MAX_INFLIGHT)]
    with pytest.raises(E.EngineError):
        m.yolo_submit(views[0], 4, h, w, **kw)
    # a synchronous call drains the stream: the outstanding tickets are complete, collecting them still works
    again = m.yolo_infer(views[1], 4, h, w, **kw)
    for a, b in zip(again, want[1]):
        assert np.array_equal(a, b)
    for i, t in enumerate(ts):
        g = m.yolo_wait(t)
        for a, b in zip(g[:3], want[i % 3]):
            assert np.array_equal(a, b)
    clip.free()
    m.close()


def test_close_with_a_ticket_in_flight_then_reuse_the_pages(gpu_engine):
    """``Model.close()`` drains the stream BEFORE it unregisters the page-locked result sets — a ticket still
    queued copies into exactly those pages.  Close with tickets out (what the overflow fallback does), drop the model, then
    run a second model whose fresh page-locked sets may reuse the freed memory: its results must be the synchronous ones, and
    ``YOLO.discard_frames`` of a token whose model is gone must do nothing."""
    kpt = (13, 3)
    sd = yolo_arch.synth_state_dict("n", 1, kpt, seed=2, cls_bias=1.0)
    h, w = 180, 320
    frames = synth.synthetic_frames(8, h, w, seed=41)
    clip = video.DeviceClip(gpu_engine, frames)
    kw = dict(imgsz=320, conf=0.25, iou=0.7, classes=[0])
    views = [clip.buffer.view(i * 4 * clip.frame_bytes, 4 * clip.frame_bytes) for i in range(2)]
    ref = E.Model(gpu_engine, G.build_yolov8(sd, 1, kpt, dtype=E.graph_dtype()))
    ref.set_max_batch(4)
    want = [tuple(np.array(a) for a in ref.yolo_infer(v, 4, h, w, **kw)) for v in views]
    ref.close()
    for rep in range(3):
        m = E.Model(gpu_engine, G.build_yolov8(sd, 1, kpt, dtype=E.graph_dtype()))
        m.set_max_batch(4)
        t0 = m.yolo_submit(views[0], 4, h, w, **kw)
        t1 = m.yolo_submit(views[1], 4, h, w, **kw)
        if rep == 2:
            del t0, t1
            del m                                        # dropped without close(): __del__ drains before it unpins
        else:
            m.close()                                    # tickets t0 / t1 still in flight
            assert m.handle is None and not m._out_ring
        m2 = E.Model(gpu_engine, G.build_yolov8(sd, 1, kpt, dtype=E.graph_dtype()))
        m2.set_max_batch(4)
        for v, wnt in zip(views, want):
            got = m2.yolo_infer(v, 4, h, w, reuse_outputs=True, **kw)
            for a, b in zip(got, wnt):
                assert np.array_equal(a, b)
        m2.close()
    clip.free()


def _trackers(tmp_path, sd_p, sd_k, batch):
    checkpoint.save_checkpoint(tmp_path / "players.pt", sd_p, "detect", 80, None, "n", {0: "person"})
    checkpoint.save_checkpoint(tmp_path / "pose.pt", sd_k, "pose", 1, (13, 3), "n", {0: "person"})
    zone = D.PolygonZone(np.array([[40, 40], [600, 40], [600, 340], [40, 340]]), frame_resolution_wh=(640, 360))
    players = PlayerTracker(str(tmp_path / "players.pt"), zone, batch_size=batch)
    players.video_info_post_init(video.VideoInfo(width=640, height=360, fps=30, total_frames=20))
    pose = PlayerKeypointsTracker(str(tmp_path / "pose.pt"), 640, batch_size=batch)
    return players, pose


def _run(tracker, clip, monkeypatch=None, sync=False):
    tracker.restart()
    if sync:
        monkeypatch.setattr(type(tracker), "submit_sample", lambda self, sample, **kw: None)
    tracker.predict_and_update(clip.frames())
    out = [o.serialize() for o in tracker.results]
    if sync:
        monkeypatch.undo()
    return out


def test_tracker_batch_loop_pipelined_equals_synchronous(gpu_engine, tmp_path, monkeypatch):
    frames = synth.synthetic_frames(20, 360, 640, seed=8)
    clip = video.DeviceClip(gpu_engine, frames)
    sd_p = yolo_arch.synth_state_dict("n", 80, None, seed=3, cls_bias=0.5)
    sd_k = yolo_arch.synth_state_dict("n", 1, (13, 3), seed=4, cls_bias=0.5)
    players, pose = _trackers(tmp_path, sd_p, sd_k, batch=6)          # 6 + 6 + 6 + 2 frames: a short last batch
    for t in (players, pose):
        calls = []
        orig = type(t).collect_sample
        monkeypatch.setattr(type(t), "collect_sample", lambda self, token, _o=orig, _c=calls: (_c.append(1), _o(self, token))[1])
        piped = _run(t, clip)
        monkeypatch.undo()
        assert len(calls) == 4, "the two-call device stage did not run"
        sync = _run(t, clip, monkeypatch, sync=True)
        assert piped == sync and len(piped) == 20
        assert sum(len(x) for x in piped) > 0
        t.to("cpu")
    clip.free()


def test_overflow_in_the_middle_of_a_pipelined_clip(gpu_engine, tmp_path, monkeypatch):
    """A checkpoint whose stem output leaves the fp16 range: the first collected ticket reports it, the model is rebuilt on
    the bf16x3 kernels, that batch and the one already queued behind it are computed again, the rest of the clip runs on the
    new model — same objects as a tracker that was bf16x3 from the start."""
    if E.fp32_mode() != "h2":
        pytest.skip("h2 is not the default arithmetic here")
    frames = synth.synthetic_frames(20, 360, 640, seed=9)
    clip = video.DeviceClip(gpu_engine, frames)
    sd_p = yolo_arch.synth_state_dict("n", 80, None, seed=3, cls_bias=0.5)
    sd_p["model.0.conv.weight"] = sd_p["model.0.conv.weight"] * 3.0e5          # SiLU(x) ~ x beyond 65504
    sd_p["model.1.conv.weight"] = sd_p["model.1.conv.weight"] / 3.0e5          # (the rest of the network sees ordinary values)
    sd_k = yolo_arch.synth_state_dict("n", 1, (13, 3), seed=4, cls_bias=0.5)
    players, _ = _trackers(tmp_path, sd_p, sd_k, batch=6)
    assert players.model.fp32_mode == "h2"
    piped = _run(players, clip)
    assert players.model.fell_back and players.model.fp32_mode == "bx3"
    players.to("cpu")
    ref_players, _ = _trackers(tmp_path, sd_p, sd_k, batch=6)
    ref_players.model.set_fp32_mode("bx3")
    want = _run(ref_players, clip, monkeypatch, sync=True)
    assert piped == want and len(piped) == 20
    ref_players.to("cpu")
    clip.free()
