"""GPU parity: HIP engine (through the C-ABI) vs the CPU oracle on the same seeded frames/weights.

Bar (BASELINE.json north_star): class ids bit-exact, box / keypoint coordinates within 1e-3 px after
NMS.  Threshold-adjacent decisions (|score-conf| or |IoU-iou| within float noise) can legitimately
flip; the harness checks that the chosen seeds keep a margin."""
import numpy as np
import pytest
import torch

from oracle import yolov8_ref as ref
from padel_analytics_amd import engine as E, graph as G, synth
from tests.helpers import calibrated_state_dict

pytestmark = pytest.mark.gpu

TOL_PX = 1e-3


def _run_engine(eng, sd, nc, kpt, frames, **kw):
    g = G.build_yolov8(sd, nc, kpt)
    m = E.Model(eng, g)
    m.set_max_batch(max(1, min(8, len(frames))))
    n, h, w, _ = frames.shape
    out = m.yolo_infer(frames, n, h, w, **kw)
    return m, out


def _compare(res_ref, boxes, kpts, counts, kpt_shape=None):
    worst = 0.0
    for i, r in enumerate(res_ref):
        nref = len(r["boxes"])
        assert counts[i] == nref, f"image {i}: {counts[i]} detections vs oracle {nref} (conf margin {r['conf_margin']:.2e})"
        if nref == 0:
            continue
        got = boxes[i, :nref]
        assert np.array_equal(got[:, 5], r["boxes"][:, 5]), "class ids differ"
        worst = max(worst, float(np.abs(got[:, :4] - r["boxes"][:, :4]).max()))
        assert np.abs(got[:, 4] - r["boxes"][:, 4]).max() < 1e-5
        if kpt_shape is not None:
            gk = kpts[i, :nref].reshape(nref, *kpt_shape)
            worst = max(worst, float(np.abs(gk[..., :2] - r["kpts"][..., :2]).max()))
            if kpt_shape[1] == 3:
                assert np.abs(gk[..., 2] - r["kpts"][..., 2]).max() < 1e-5
    return worst


@pytest.mark.parametrize("scale,hw,nf", [("n", (720, 1280), 4), ("n", (640, 640), 3), ("n", (1080, 1920), 2),
                                         ("m", (720, 1280), 2), ("n", (480, 854), 2)])
def test_detect_parity(gpu_engine, scale, hw, nf):
    frames = synth.synthetic_frames(nf, hw[0], hw[1], seed=3)
    # players path: frames reach the network in their own (BGR) channel order,
    # i.e. upstream is handed the RGB-converted arrays and flips them back
    srcs = [f[..., ::-1] for f in frames]
    sd = calibrated_state_dict(scale, 80, None, srcs, 640, 0.5, seed=5)
    oracle = ref.YoloV8Ref(sd, 80, None)
    res = ref.predict(oracle, srcs, conf=0.5, iou=0.7, imgsz=640, classes=[0])
    assert sum(len(r["boxes"]) for r in res) > 0, "calibration produced no detections"
    m, (boxes, kpts, counts) = _run_engine(gpu_engine, sd, 80, None, frames, imgsz=640, conf=0.5, iou=0.7,
                                           classes=[0], channel_reverse=False)
    # raw head maps first: localises a failure to the conv stack vs decode/NMS
    with torch.no_grad():
        im = ref.preprocess([f[..., ::-1] for f in frames], 640)
        det, _ = oracle.head_raw(oracle.features(im))
    for l in range(3):
        hd = m.read_head(l, min(len(frames), 8))[..., :144]
        want = det[l].permute(0, 2, 3, 1).numpy()[:hd.shape[0]]
        err = np.abs(hd - want).max() / max(1.0, np.abs(want).max())
        assert err < 2e-5, f"head level {l}: rel err {err:.3e}"
    worst = _compare(res, boxes, kpts, counts)
    assert worst <= TOL_PX, f"box L-inf {worst:.3e} px"
    m.close()


@pytest.mark.parametrize("scale,S,kpt", [("n", 640, (13, 3)), ("n", 1280, (13, 3)), ("n", 640, (13, 2)), ("m", 640, (13, 3))])
def test_pose_parity(gpu_engine, scale, S, kpt):
    from PIL import Image
    frames = synth.synthetic_frames(2, 720, 1280, seed=7)
    # pose path: BGR->RGB, PIL bicubic stretch to SxS (players_keypoints_tracker.py:260-266); the PIL
    # image is converted RGB->BGR by upstream and flipped back in preprocess -> true RGB
    pil = [np.asarray(Image.fromarray(f[..., ::-1].copy()).resize((S, S))) for f in frames]
    srcs = [p[..., ::-1] for p in pil]
    sd = calibrated_state_dict(scale, 1, kpt, srcs, S, 0.25, seed=11)
    oracle = ref.YoloV8Ref(sd, 1, kpt)
    res = ref.predict(oracle, srcs, conf=0.25, iou=0.7, imgsz=S, classes=[0])
    assert sum(len(r["boxes"]) for r in res) > 0
    m, (boxes, kpts, counts) = _run_engine(gpu_engine, sd, 1, kpt, frames, imgsz=S, conf=0.25, iou=0.7, classes=[0],
                                           pre_mode=E.PRE_PIL_STRETCH, channel_reverse=True)
    worst = _compare(res, boxes, kpts, counts, kpt)
    assert worst <= TOL_PX, f"box/kpt L-inf {worst:.3e} px"
    m.close()
