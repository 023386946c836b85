"""The hand-derived known answers of tests/known_answers.py through the HIP decode / NMS kernels (postproc.hip) via
``pa_yolo_postprocess``: the product's Detect / Pose inference branch, NMS and scale_boxes / scale_coords on
caller-supplied head maps, against numbers worked out from the published formulas (no oracle in between)."""
import numpy as np
import pytest

from padel_analytics_amd import engine as E, graph as G, yolo_arch
from tests import known_answers as KA

pytestmark = pytest.mark.gpu


def _model(eng, nc, kpt, n):
    m = E.Model(eng, G.build_yolov8(yolo_arch.synth_state_dict("n", nc, kpt, seed=0), nc, kpt, dtype=E.graph_dtype()))
    m.set_max_batch(n)
    return m


def _check_rows(boxes, count, want, tag):
    assert count == len(want), f"{tag}: {count} rows, expected {len(want)}"
    for i, w in enumerate(want):
        g = boxes[i]
        assert np.array_equal(g[:4], np.asarray(w[:4], np.float32)), f"{tag} row {i}: box {g[:4]} != {w[:4]}"
        assert abs(float(g[4]) - w[4]) < 3e-7 and int(g[5]) == w[5], f"{tag} row {i}: score / class {g[4:6]} != {w[4:6]}"
    assert (boxes[count:] == 0).all(), f"{tag}: rows beyond the count must be zero"


@pytest.mark.parametrize("builder,classes", [(KA.detect_cases, None), (KA.detect_cases_class_filter, [1])], ids=["all", "classes=[1]"])
def test_detect_known_answers(gpu_engine, builder, classes):
    nc, heads, exp = builder()
    m = _model(gpu_engine, nc, None, len(heads[0]))
    assert [h.shape[1:] for h in heads] == [tuple(s) for s in m.head_shapes(KA.H0, KA.W0, KA.IMGSZ)]
    boxes, _, counts = m.yolo_postprocess(heads, KA.H0, KA.W0, imgsz=KA.IMGSZ, conf=0.5, iou=0.7, classes=classes)
    m.close()
    for i in range(len(counts)):
        _check_rows(boxes[i], int(counts[i]), exp[i], f"image {i}")


def test_detect_max_det_truncates_after_nms(gpu_engine):
    nc, heads, exp = KA.detect_cases()
    m = _model(gpu_engine, nc, None, len(heads[0]))
    boxes, _, counts = m.yolo_postprocess(heads, KA.H0, KA.W0, imgsz=KA.IMGSZ, conf=0.5, iou=0.7, max_det=2)
    m.close()
    for i in range(len(counts)):
        _check_rows(boxes[i], int(counts[i]), exp[i][:2], f"image {i} (max_det 2)")


def test_pose_known_answers(gpu_engine):
    nc, kshape, heads, rows, ek = KA.pose_cases()
    m = _model(gpu_engine, nc, kshape, 1)
    boxes, kpts, counts = m.yolo_postprocess(heads, KA.H0, KA.W0, imgsz=KA.IMGSZ, conf=0.25, iou=0.7, classes=[0])
    m.close()
    _check_rows(boxes[0], int(counts[0]), rows, "pose")
    k = kpts[0, 0].reshape(13, 3)
    assert np.array_equal(k[:, :2], ek[:, :2].astype(np.float32)), k[:3]
    assert np.allclose(k[:, 2], ek[:, 2], atol=3e-7)
