"""Pins the ball-path oracle (oracle/ball_ref.py) and the product's host pieces to golden outputs OF THE REFERENCE
ITSELF: ``tests/golden/ball_golden.npz`` / ``objects_golden.json`` were produced by importing the reference's
``trackers/ball_tracker/ball_tracker.py`` and result-object modules in the build container
(``tests/golden/make_ball_golden.py``) — the real ``predict_frames`` ensemble loop (:421-523) with its DataLoader
batching and 7-deep prediction buffer, ``get_ensemble_weight`` (:68-97), ``generate_inpaint_mask`` (:100-136),
``Ball`` / ``Player`` / ``PlayerKeypoints`` serialisation.  Plus the heat-map decode rules (cv2 contour order stays
unpinned: cv2 is not installable)."""
import json
from pathlib import Path

import numpy as np
import pytest

from oracle import ball_ref as br
from padel_analytics_amd import inpaint as ip
from padel_analytics_amd.detections import Detections
from padel_analytics_amd.trackers.ball_tracker import Ball, predict_location
from padel_analytics_amd.trackers.players_keypoints_tracker import PlayerKeypoint, PlayerKeypoints, PlayersKeypoints
from padel_analytics_amd.trackers.players_tracker import Player, Players

GOLD = np.load(Path(__file__).parent / "golden" / "ball_golden.npz")
OBJ = json.loads((Path(__file__).parent / "golden" / "objects_golden.json").read_text())


def test_ensemble_weight_matches_reference():
    assert np.array_equal(br.ensemble_weight(8), GOLD["w_weight_8"])
    assert np.array_equal(br.inpaint_ensemble_weight(16), GOLD["w_weight_16"])
    for L in (8, 16, 5):
        assert np.array_equal(ip.ensemble_weight(L), GOLD[f"w_weight_{L}"]), L


@pytest.mark.parametrize("case", range(int(GOLD["n_ens"])))
def test_ensemble_matches_reference_predict_frames_loop(case):
    """Oracle and product ensembles vs the heat maps the reference's own loop produced for the same window outputs
    (video lengths 8..64, DataLoader batch sizes 1..16: head / steady-state / tail branches, buffer carry-over)."""
    y, want = GOLD[f"ens{case}_y"], GOLD[f"ens{case}_heat"]
    T = int(GOLD[f"ens{case}_T"])
    got = br.ensemble(y)
    assert got.shape == want.shape == (T,) + y.shape[2:]
    assert np.array_equal(got, want), f"oracle ensemble differs from the reference (T={T}, batch={int(GOLD[f'ens{case}_batch'])})"
    prod = ip.temporal_ensemble(y, ip.ensemble_weight(8))
    assert np.abs(prod - want).max() <= 1.2e-7          # same terms, numpy summation order


def test_result_objects_match_reference_serialisation():
    for g in OBJ["ball"]:
        a = g["args"]
        b = Ball(frame=a["frame"], xy=tuple(a["xy"]), visibility=a["visibility"])
        assert json.loads(json.dumps(b.serialize())) == g["serialize"] and list(b.asint()) == g["asint"]
        assert Ball.from_json(b.serialize()).serialize() == b.serialize()
    for g in OBJ["player"]:
        det = Detections(np.array([g["xyxy"]], np.float32), np.array([g["confidence"]], np.float32),
                         np.array([g["class_id"]]), None if g["tracker_id"] is None else np.array([g["tracker_id"]]))
        p = Player(det)
        assert json.loads(json.dumps(p.serialize())) == g["serialize"]
        for k in ("top_left", "bottom_right", "midpoint", "feet"):
            assert list(getattr(p, k)) == g[k], k
        assert (p.height, p.width) == (g["height"], g["width"])
        # the array-backed (lazy) container serialises to the same bytes as the reference's objects
        rows = np.array([g["xyxy"] + [g["confidence"], g["class_id"]]], np.float32)
        ids = None if g["tracker_id"] is None else np.array([g["tracker_id"]])
        assert json.loads(json.dumps(Players(rows=rows, ids=ids).serialize())) == [g["serialize"]]
    g = OBJ["player_keypoints"]
    assert PlayerKeypoints.KEYPOINTS_NAMES == g["names"]
    kps = PlayerKeypoints([PlayerKeypoint(id=i, name=n, xy=(1.5 * i, 100.0 - 2.25 * i)) for i, n in enumerate(g["names"])])
    assert json.loads(json.dumps(kps.serialize())) == g["serialize"] and list(kps[g["names"][3]].asint()) == g["asint_3"]
    xy = np.array([[[1.5 * i, 100.0 - 2.25 * i] for i in range(13)]], np.float32)
    assert json.loads(json.dumps(PlayersKeypoints(xy=xy).serialize())) == [g["serialize"]]


def test_decode_rules():
    m = np.zeros((288, 512), np.uint8)
    assert br.predict_location(m) == (0, 0, 0, 0) and predict_location(m) == (0, 0, 0, 0)
    m[10:14, 20:29] = 255            # 9 x 4 = 36
    m[100:103, 300:303] = 255        # 3 x 3 = 9
    m[103, 303] = 255                # diagonal neighbour: 8-connected -> same component, rect 4 x 4 = 16
    assert br.predict_location(m) == (20, 10, 9, 4) == predict_location(m)
    heat = np.zeros((2, 288, 512), np.float32)
    heat[0, 10:14, 20:29] = 0.9
    heat[1, 0, 0] = 0.5              # not > 0.5
    x, y, v = br.decode_heat(heat, (1280 / 512, 720 / 288))
    assert (x[0], y[0], v[0]) == (int(24 * 2.5), int(12 * 2.5), 1) and (x[1], y[1], v[1]) == (0, 0, 0)


def test_decode_tie_rule_hand_derived():
    """Equal-area rectangles: the chosen cv2.findContours order — reverse raster discovery — applied by hand
    in tests/known_answers.py:BALL_TIE_CASES; oracle and product host code must return exactly those rectangles (the device
    kernel meets the same cases in tests/test_gpu_ball.py)."""
    from tests.known_answers import BALL_TIE_CASES
    for why, mask, want in BALL_TIE_CASES:
        assert br.predict_location(mask) == want, (why, br.predict_location(mask))
        assert predict_location(mask) == want, (why, predict_location(mask))
