"""Host-side plugin API (no GPU): sampler / predict_and_update protocol, JSON prediction caches, result
objects, frame sources, PolygonZone, ByteTrack — mirrors how the reference's runner drives trackers
(trackers/tracker.py:280-330, trackers/runner.py:175-236)."""
import json

import numpy as np
import pytest

from padel_analytics_amd import bytetrack, detections as D, video
from padel_analytics_amd.trackers import tracker as T
from padel_analytics_amd.trackers.players_keypoints_tracker import PlayerKeypoint, PlayerKeypoints, PlayersKeypoints
from padel_analytics_amd.trackers.players_tracker import Player, Players
from tests import synth  # noqa: F401  (registers the synthetic:// frame source)


class _Obj(T.Object):
    def __init__(self, v): self.v = v
    @classmethod
    def from_json(cls, x): return cls(x["v"])
    def serialize(self): return {"v": self.v}


class _BatchTracker(T.Tracker):
    batch_size = 4
    def __init__(self, **kw):
        self.calls = []
        super().__init__(**kw)
    def video_info_post_init(self, vi): return self
    def object(self): return _Obj
    def draw_kwargs(self): return {}
    def restart(self): self.results.restart()
    def __str__(self): return "dummy_tracker"
    def predict_sample(self, sample, **kw):
        self.calls.append(len(sample))
        return [_Obj(int(f.sum())) for f in sample]
    def predict_frames(self, gen, **kw): raise T.NoPredictFrames()


class _StreamTracker(_BatchTracker):
    def predict_sample(self, sample, **kw): raise T.NoPredictSample()
    def predict_frames(self, gen, **kw): return [_Obj(i) for i, _ in enumerate(gen)]


def test_sampler_protocol_and_last_short_batch():
    t = _BatchTracker()
    frames = (np.full((2, 2, 3), i, np.uint8) for i in range(10))
    res = t.predict_and_update(frames, total_frames=10)
    assert t.calls == [4, 4, 2] and len(res) == 10 and res.counter == 3
    assert [o.v for o in res] == [12 * i for i in range(10)]
    assert len(res.sample_predictions) == 2


def test_stream_tracker_uses_predict_frames():
    t = _StreamTracker()
    res = t.predict_and_update(iter([np.zeros((1, 1, 3), np.uint8)] * 5))
    assert len(res) == 5


def test_prediction_cache_roundtrip(tmp_path):
    p = tmp_path / "cache.json"
    t = _BatchTracker(save_path=p)
    t.predict_and_update(np.zeros((1, 1, 3), np.uint8) for _ in range(3))
    t.save_predictions()
    t2 = _BatchTracker(load_path=p)
    assert len(t2) == 3 and [o.v for o in t2.results] == [0, 0, 0]


def test_players_json_wire_format():
    det = D.Detections(np.array([[10.6, 20.2, 50.9, 120.4]]), np.array([0.9]), np.array([0]), np.array([7]))
    pl = Player(det)
    assert pl.id == 7 and pl.top_left == (10, 20) and pl.bottom_right == (50, 120)
    assert pl.feet == (30, 120) and pl.midpoint == (30, 70)
    s = Players([pl]).serialize()
    assert set(s[0]) == {"id", "xyxy", "projection", "class_id", "confidence"}
    back = Players.from_json(json.loads(json.dumps(s)))
    assert back[0].id == 7 and np.allclose(back[0].xyxy, pl.xyxy)
    assert Player(D.Detections(np.zeros((1, 4)), np.ones(1), np.zeros(1, int), None)).id is None


def test_keypoints_json_wire_format():
    kps = PlayerKeypoints([PlayerKeypoint(i, n, (float(i), 2.0 * i)) for i, n in enumerate(PlayerKeypoints.KEYPOINTS_NAMES)])
    assert len(kps) == 13 and kps["head"].id == 5 and kps["head"].asint() == (5, 10)
    s = PlayersKeypoints([kps]).serialize()
    assert list(s[0]) == ["player_keypoints"] and set(s[0]["player_keypoints"][0]) == {"id", "name", "xy"}
    back = PlayersKeypoints.from_json(json.loads(json.dumps(s)))
    assert back[0]["left_elbow"].xy == [12.0, 24.0] or tuple(back[0]["left_elbow"].xy) == (12.0, 24.0)


def test_synthetic_video_source(tmp_path):
    vi = video.VideoInfo.from_video_path("synthetic://?n=5&h=48&w=64&fps=25&seed=3")
    assert (vi.width, vi.height, vi.fps, vi.total_frames, vi.resolution_wh) == (64, 48, 25, 5, (64, 48))
    fr = list(video.get_video_frames_generator("synthetic://?n=5&h=48&w=64&seed=3", start=1, end=4))
    assert len(fr) == 3 and fr[0].shape == (48, 64, 3) and fr[0].dtype == np.uint8
    again = list(video.get_video_frames_generator("synthetic://?n=5&h=48&w=64&seed=3", start=1, end=4))
    assert all(np.array_equal(a, b) for a, b in zip(fr, again))
    np.save(tmp_path / "clip.npy", np.stack(fr))
    assert video.VideoInfo.from_video_path(tmp_path / "clip.npy").total_frames == 3
    assert np.array_equal(next(video.get_video_frames_generator(tmp_path / "clip.npy")), fr[0])


def test_polygon_zone_bottom_center_trigger():
    zone = D.PolygonZone(np.array([[100, 100], [300, 100], [300, 300], [100, 300]]), frame_resolution_wh=(400, 400))
    det = D.Detections(np.array([[150, 50, 250, 200], [0, 0, 50, 50], [250, 250, 420, 450], [280, 10, 320, 100]]),
                       np.ones(4), np.zeros(4, int))
    inside = zone.trigger(det)
    # anchors: (200,200) in; (25,50) out; clipped (325,400) out; (300,100) on the outline -> in
    assert inside.tolist() == [True, False, False, True]
    assert zone.mask.shape == (401, 401)
    assert len(det[inside]) == 2


def test_bytetrack_ids_are_stable_and_new_tracks_need_two_frames():
    bt = bytetrack.ByteTrack(frame_rate=30)
    boxes = np.array([[100, 100, 150, 220], [400, 120, 450, 230]], np.float32)
    ids = []
    for f in range(6):
        det = D.Detections(boxes + f * 2.0, np.array([0.9, 0.8], np.float32), np.zeros(2, int))
        out = bt.update_with_detections(det)
        ids.append(out.tracker_id.tolist())
        assert np.allclose(out.xyxy, (boxes + f * 2.0)[: len(out)])          # detector boxes, not Kalman boxes
    assert ids[0] == [1, 2] and all(i == [1, 2] for i in ids)
    # a detection appearing later is unconfirmed on its first frame (dropped), confirmed on the next
    extra = np.array([[700, 300, 760, 420]], np.float32)
    det = D.Detections(np.vstack([boxes + 12, extra]), np.array([0.9, 0.8, 0.95], np.float32), np.zeros(3, int))
    assert bt.update_with_detections(det).tracker_id.tolist() == [1, 2]
    det = D.Detections(np.vstack([boxes + 14, extra + 1]), np.array([0.9, 0.8, 0.95], np.float32), np.zeros(3, int))
    assert bt.update_with_detections(det).tracker_id.tolist() == [1, 2, 3]
    bt.reset()
    assert bt.update_with_detections(D.Detections(boxes, np.array([0.9, 0.8]), np.zeros(2, int))).tracker_id.tolist() == [1, 2]
    assert len(bt.update_with_detections(D.Detections.empty())) == 0


def test_court_keypoints_fixed_short_circuit(tmp_path):
    from padel_analytics_amd.trackers import Keypoint, Keypoints, KeypointsTracker
    fixed = Keypoints([Keypoint(id=i, xy=(10.0 * i, 5.0 * i)) for i in (3, 1, 2)])
    assert [k.id for k in fixed] == [1, 2, 3] and fixed[2].asint() == (20, 10)
    assert Keypoints.from_json(json.loads(json.dumps(fixed.serialize())))[3].xy == [30.0, 15.0]
    t = KeypointsTracker("missing.pt", batch_size=4, model_type="yolo", fixed_keypoints_detection=fixed)
    t.to("cuda")                                            # no model is touched in the shipped configuration
    out = t.predict_and_update(np.zeros((2, 2, 3), np.uint8) for _ in range(5))
    assert len(out) == 5 and all(o is fixed for o in out)
    with pytest.raises(NotImplementedError):
        KeypointsTracker("x.pt", 4, model_type="resnet").predict_frames(iter([]))
