"""The hand-derived known answers of tests/known_answers.py through the CPU oracle (oracle/yolov8_ref.py): pins the
restatement of the [upstream] Detect / Pose inference branch, ops.non_max_suppression and scale_boxes / scale_coords /
LetterBox geometry to numbers worked out from the published formulas."""
import numpy as np
import pytest
import torch

from oracle import yolov8_ref as ref
from tests import known_answers as KA


def _oracle_post(nc, kpt_shape, heads, conf, iou, classes):
    """heads (n, H, W, c) NHWC -> what YOLO.predict returns for 720p sources."""
    m = ref.YoloV8Ref({}, nc, kpt_shape)
    det = [torch.from_numpy(h[..., :64 + nc]).permute(0, 3, 1, 2).contiguous() for h in heads]
    kpt = [torch.from_numpy(h[..., 64 + nc:64 + nc + (kpt_shape[0] * kpt_shape[1] if kpt_shape else 0)]).permute(0, 3, 1, 2).contiguous()
           for h in heads] if kpt_shape else []
    pred = m.decode(det, kpt)
    out = ref.non_max_suppression(pred, conf, iou, classes, 300, nc=nc)
    res = []
    for d in out:
        d = d.clone()
        d[:, :4] = ref.scale_boxes((384, 640), d[:, :4], (KA.H0, KA.W0))
        k = None
        if kpt_shape:
            k = ref.scale_coords((384, 640), d[:, 6:].view(len(d), *kpt_shape), (KA.H0, KA.W0)).numpy()
        res.append((d[:, :6].numpy(), k))
    return res


def _check_rows(got, want, tag):
    assert len(got) == len(want), f"{tag}: {len(got)} rows, expected {len(want)}"
    for i, (g, w) in enumerate(zip(got, want)):
        assert np.array_equal(g[:4], np.asarray(w[:4], np.float32)), f"{tag} row {i}: box {g[:4]} != {w[:4]}"
        assert abs(float(g[4]) - w[4]) < 2e-7 and int(g[5]) == w[5], f"{tag} row {i}: score / class {g[4:6]} != {w[4:6]}"


@pytest.mark.parametrize("builder,classes", [(KA.detect_cases, None), (KA.detect_cases_class_filter, [1])])
def test_detect_known_answers(builder, classes):
    nc, heads, exp = builder()
    res = _oracle_post(nc, None, heads, 0.5, 0.7, classes)
    for i in range(len(res)):
        _check_rows(res[i][0], exp[i], f"image {i}")


def test_pose_known_answers():
    nc, kshape, heads, rows, ek = KA.pose_cases()
    (boxes, kpts), = _oracle_post(nc, kshape, heads, 0.25, 0.7, [0])
    _check_rows(boxes, rows, "pose")
    assert np.allclose(kpts[0][:, :2], ek[:, :2], rtol=0, atol=0), kpts[0][:3]
    assert np.allclose(kpts[0][:, 2], ek[:, 2], atol=2e-7)
    xy = ref.keypoints_xy(torch.from_numpy(kpts)).numpy()[0]
    assert xy[1].tolist() == [0.0, 0.0] and xy[0].tolist() == [328.0, 160.0] and xy[2].tolist() == [1280.0, 720.0]
    assert (xy[3:] == 0).all()                       # visibility sigmoid(-20) < 0.5: zeroed


@pytest.mark.parametrize("hw,resized,pads,net,gain,pad_xy", KA.GEOMETRY, ids=[f"{g[0][0]}x{g[0][1]}" for g in KA.GEOMETRY])
def test_letterbox_and_scale_boxes_geometry(hw, resized, pads, net, gain, pad_xy):
    nw, nh, top, bottom, left, right = ref.letterbox_geometry(hw[0], hw[1], KA.IMGSZ, auto=True)
    assert (nw, nh) == resized and (top, bottom, left, right) == pads
    assert (nh + top + bottom, nw + left + right) == net
    # scale_boxes: a box given in network pixels maps back through exactly (x - pad_x) / gain
    b = torch.tensor([[pad_xy[0] + 10.0, pad_xy[1] + 20.0, pad_xy[0] + 110.0, pad_xy[1] + 220.0]])
    got = ref.scale_boxes(net, b, hw)[0].numpy()
    g32 = np.float32(gain) if gain in (0.5, 1.0, 0.8, 0.64) else np.float32(min(net[0] / hw[0], net[1] / hw[1]))
    want = np.minimum(np.array([10.0, 20.0, 110.0, 220.0], np.float32) / g32, np.array([hw[1], hw[0], hw[1], hw[0]], np.float32))
    assert np.allclose(got, want, rtol=1e-6), (got, want)
