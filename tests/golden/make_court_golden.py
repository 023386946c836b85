"""Generates tests/golden/court_golden.json by importing the reference's ``analytics/projected_court.py`` in THIS
container (the reference never travels: only the numbers below are committed).

The module's top-level imports need packages that are not installed here (cv2, supervision) and the reference's own
``trackers`` package (which imports ultralytics); they are replaced by empty placeholder modules for the import only.
Nothing of them is executed: the golden covers exactly the pure-Python / numpy arithmetic of
``ProjectedCourt.__init__`` (court geometry, projected_court.py:217-324), ``ProjectedCourtKeypoints.keypoints``
(:108-148) and ``ProjectedCourt.project_point`` (:473-502).  ``cv2.findHomography`` (:469) is NOT covered — that part
of the restatement stays "parity unpinned".

    python tests/golden/make_court_golden.py        # needs /root/reference
"""
import json, sys, types
from dataclasses import dataclass
from pathlib import Path

import numpy as np

REF = "/root/reference"


def _placeholders():
    for name in ("cv2", "supervision"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["supervision"].VideoInfo = object           # only used as a type annotation
    t = types.ModuleType("trackers")

    @dataclass
    class Keypoint:
        id: int
        xy: tuple

    for n in ("Player", "Players", "Keypoints", "Ball"):
        setattr(t, n, type(n, (), {}))
    t.Keypoint = Keypoint
    sys.modules["trackers"] = t
    a = types.ModuleType("analytics.data_analytics")
    a.DataAnalytics = type("DataAnalytics", (), {})
    pkg = types.ModuleType("analytics")
    pkg.__path__ = [REF + "/analytics"]
    sys.modules["analytics"] = pkg
    sys.modules["analytics.data_analytics"] = a


def main():
    _placeholders()
    sys.path.insert(0, REF)
    import importlib
    pc = importlib.import_module("analytics.projected_court")
    out = {"geometry": [], "project_point": []}
    for (w, h) in ((1280, 720), (1920, 1080), (640, 640), (854, 480), (3840, 2160)):
        vi = types.SimpleNamespace(width=w, height=h)
        court = pc.ProjectedCourt(vi)
        entry = {"width": w, "height": h,
                 "background": [list(court.background_position.top_left), list(court.background_position.bottom_right)],
                 "court": [list(court.court_position.top_left), list(court.court_position.bottom_right)],
                 "origin": list(court.court_keypoints.origin)}
        for n in (12, 18, 22):
            entry[f"keypoints_{n}"] = [[k.id, list(k.xy)] for k in court.court_keypoints.keypoints(number_keypoints=n)]
        out["geometry"].append(entry)
    rng = np.random.default_rng(7)
    court = pc.ProjectedCourt(types.SimpleNamespace(width=1280, height=720))
    for _ in range(8):
        H = np.eye(3) + rng.normal(0, 0.2, (3, 3)) * np.array([[1, 1, 50], [1, 1, 50], [1e-3, 1e-3, 0]])
        pt = (int(rng.integers(0, 1280)), int(rng.integers(0, 720)))
        x, y = court.project_point(pt, H)
        out["project_point"].append({"H": H.tolist(), "point": list(pt), "projected": [float(x), float(y)]})
    Path(__file__).with_name("court_golden.json").write_text(json.dumps(out, indent=1))
    print("wrote", len(out["geometry"]), "geometries,", len(out["project_point"]), "projections")


if __name__ == "__main__":
    main()
