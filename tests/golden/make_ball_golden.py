"""Generates tests/golden/ball_golden.npz and tests/golden/objects_golden.json by importing the reference's OWN
modules in THIS container (the reference never travels: only the numbers below are committed):

* ``trackers/ball_tracker/ball_tracker.py``: ``get_ensemble_weight`` (:68-97), ``generate_inpaint_mask`` (:100-136) on
  60 random visibility patterns + the hand-written edge cases, and the temporal-ensemble loop of
  ``BallTracker.predict_frames`` (:421-523) driven end to end — the real ``DataLoader`` batching, the real
  ``y_pred_buffer`` algebra, head / steady-state / tail branches — with a stub window iterable, a stub TrackNet that
  returns seeded window outputs, and a stub ``predict_modified`` that captures the ensembled heat maps;
* ``Ball`` / ``Player`` / ``PlayerKeypoint(s)`` result objects: ``serialize`` output and the derived integer
  properties (``players_tracker.py:14-98``, ``players_keypoints_tracker.py:14-135``, ``ball_tracker.py:139-175``).

The modules' top-level imports need packages that are not installed here (cv2, supervision, ultralytics, parse) and
the reference's ``trackers/__init__.py`` pulls all of them in; they are replaced by permissive placeholder modules
for the import only, and the ``trackers`` package is registered as a bare namespace over the reference directory so
its ``__init__`` never runs.  Heat maps are 6x10 instead of 288x512 (class attributes overridden on a subclass):
the buffer algebra does not depend on the map size.

    python tests/golden/make_ball_golden.py        # needs /root/reference
"""
import json
import sys
import types
from pathlib import Path

import numpy as np
import torch

REF = "/root/reference"
H, W = 6, 10


class _Anything(types.ModuleType):
    """Placeholder module: any attribute is a dummy class (only ever used as an annotation / base / never called)."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = type(name, (), {"__init__": lambda self, *a, **k: self.__dict__.update(k)})
        setattr(self, name, cls)
        return cls


def _placeholders():
    for name in ("cv2", "supervision", "ultralytics", "parse", "utils", "utils.converters"):
        sys.modules[name] = _Anything(name)
    sys.modules["utils"].converters = sys.modules["utils.converters"]
    pkg = types.ModuleType("trackers")
    pkg.__path__ = [REF + "/trackers"]
    sys.modules["trackers"] = pkg
    for sub in ("ball_tracker", "players_tracker", "players_keypoints_tracker"):
        m = types.ModuleType(f"trackers.{sub}")
        m.__path__ = [f"{REF}/trackers/{sub}"]
        sys.modules[f"trackers.{sub}"] = m


def ensemble_cases(bt):
    """Drive BallTracker.predict_frames' loop (:421-523) for several (video length, batch size) pairs."""
    out = {}
    captured = []

    def capture(y_pred, img_scaler, WIDTH, HEIGHT):
        captured.append(y_pred.numpy().copy())
        return {"x": [], "y": [], "visibility": []}

    bt.predict_modified = capture

    class StubIterable(torch.utils.data.IterableDataset):
        n_windows = 0

        def __init__(self, **kw): pass

        def __iter__(self):
            for g in range(self.n_windows):
                yield np.full((27, H, W), g, np.float64)       # the stub TrackNet reads the window index back

    bt.BallTrajectoryIterable = StubIterable

    class Tiny(bt.BallTracker):
        HEIGHT, WIDTH = H, W

        def __init__(self): pass

    for case, (T, batch) in enumerate([(8, 3), (9, 4), (15, 8), (20, 8), (37, 5), (23, 1), (64, 16)]):
        rng = np.random.default_rng(100 + case)
        y = rng.uniform(0, 1, (T - 7, 8, H, W)).astype(np.float32)
        t = Tiny()
        t.video_info = types.SimpleNamespace(width=W * 2, height=H * 3)
        t.tracknet_seq_len, t.median, t.median_max_sample_num, t.batch_size, t.inpaintnet = 8, None, 400, batch, None
        t.tracknet = lambda x, y=y: torch.from_numpy(y[x[:, 0, 0, 0].long().numpy()])
        StubIterable.n_windows = T - 7
        captured.clear()
        try:
            t.predict_frames(iter(()), total_frames=T)
        except KeyError:
            pass            # without an InpaintNet the reference's final assembly raises
        heat = np.concatenate(captured)[:, 0]
        assert heat.shape == (T, H, W), heat.shape
        out[f"ens{case}_T"] = np.int32(T)
        out[f"ens{case}_batch"] = np.int32(batch)
        out[f"ens{case}_y"] = y
        out[f"ens{case}_heat"] = heat
    out["n_ens"] = np.int32(7)
    return out


def main():
    _placeholders()
    sys.path.insert(0, REF)
    import importlib
    bt = importlib.import_module("trackers.ball_tracker.ball_tracker")
    npz = ensemble_cases(bt)
    for L in (8, 16, 5):
        for mode in ("weight", "average"):
            npz[f"w_{mode}_{L}"] = bt.get_ensemble_weight(L, mode).numpy()
    rng = np.random.default_rng(0)
    cases = [([1, 1, 0, 0, 1, 1], [100, 100, 0, 0, 100, 100]), ([0, 0, 1, 1], [0, 0, 90, 90]), ([1, 0, 0, 1], [80, 0, 0, 80]),
             ([1, 1, 1, 0, 0], [70, 70, 70, 0, 0]), ([1, 1, 0, 1, 0, 0, 1], [5, 5, 0, 5, 0, 0, 99]), ([1] * 5, [50] * 5),
             ([0] * 5, [0] * 5), ([1], [40]), ([0], [0]), ([1, 0], [90, 0]), ([0, 1], [0, 90])]
    for _ in range(60):
        n = int(rng.integers(1, 40))
        v = (rng.uniform(size=n) > 0.4).astype(int)
        y = np.where(v == 1, rng.integers(0, 200, n), 0)
        cases.append((v.tolist(), y.tolist()))
    masks = []
    for v, y in cases:
        m = bt.generate_inpaint_mask({"y": y, "visibility": v}, th_h=36.0)
        masks.append({"visibility": v, "y": y, "th_h": 36.0, "mask": [int(a) for a in m]})
    np.savez_compressed(Path(__file__).with_name("ball_golden.npz"), **npz)

    # ---- result objects
    pt = importlib.import_module("trackers.players_tracker.players_tracker")
    pk = importlib.import_module("trackers.players_keypoints_tracker.players_keypoints_tracker")
    objs = {"inpaint_masks": masks, "ball": [], "player": [], "player_keypoints": []}
    for fr, xy, vis in ((0, (0, 0), 0), (17, (512, 300), 1), (3, (12.5, 7.25), 1)):
        b = bt.Ball(frame=fr, xy=xy, visibility=vis)
        objs["ball"].append({"args": {"frame": fr, "xy": list(xy), "visibility": vis}, "serialize": json.loads(json.dumps(b.serialize())),
                             "asint": list(b.asint())})
    for xyxy, tid, cid, conf in (([10.6, 20.2, 50.9, 120.4], 7, 0, 0.9), ([0.0, 0.0, 1279.99, 719.5], 1, 0, 0.51),
                                 ([333.3, 100.0, 400.7, 333.9], 0, 0, 0.75), ([5.5, 6.5, 7.5, 9.5], None, 2, 0.3)):
        det = types.SimpleNamespace(xyxy=np.array([xyxy], np.float32), confidence=np.array([conf], np.float32),
                                    class_id=np.array([cid]), tracker_id=None if tid is None else np.array([tid]))
        p = pt.Player(det)
        objs["player"].append({"xyxy": xyxy, "tracker_id": tid, "class_id": cid, "confidence": conf,
                               "serialize": json.loads(json.dumps(p.serialize())), "top_left": list(p.top_left),
                               "bottom_right": list(p.bottom_right), "height": p.height, "width": p.width,
                               "midpoint": list(p.midpoint), "feet": list(p.feet)})
    names = pk.PlayerKeypoints.KEYPOINTS_NAMES
    kps = pk.PlayerKeypoints([pk.PlayerKeypoint(id=i, name=n, xy=(1.5 * i, 100.0 - 2.25 * i)) for i, n in enumerate(names)])
    objs["player_keypoints"] = {"names": list(names), "serialize": json.loads(json.dumps(kps.serialize())),
                                "asint_3": list(kps[names[3]].asint())}
    Path(__file__).with_name("objects_golden.json").write_text(json.dumps(objs))
    print("wrote ball_golden.npz", {k: v.shape for k, v in npz.items() if k.endswith("heat")}, "and objects_golden.json")


if __name__ == "__main__":
    main()
