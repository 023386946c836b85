"""decode + NMS kernels (pa_yolo_postprocess) against the oracle's decode + non_max_suppression on random head maps with a
controlled number of candidates per image: a few dozen to ~4 800 of 5 040 anchors at 640, 11 600 / 18 300 of 19 320 at 1280 —
more than one chunk of 4 096 ranked candidates staged in LDS, the 16 384-key LDS sort and the HBM sort beyond it, max_det
saturation early in the ranking and heavy overlap (few survivors among thousands of candidates).  Kept sets, their order and
the rescaled boxes must equal the oracle's."""
import numpy as np
import pytest
import torch

from oracle import yolov8_ref as ref
from padel_analytics_amd import engine as E, graph as G, yolo_arch
from tests import known_answers as KA

pytestmark = pytest.mark.gpu

CASES = [(0.01, 3.0, 300, 640), (0.05, 3.0, 300, 640), (0.2, 3.0, 300, 640), (0.2, 0.5, 300, 640), (0.6, 3.0, 300, 640),
         (0.95, 3.0, 300, 640), (0.95, 0.5, 300, 640), (0.95, 0.15, 300, 640), (0.2, 3.0, 50, 640), (0.05, 6.0, 300, 640),
         (0.6, 0.5, 300, 1280), (0.95, 0.5, 300, 1280), (0.95, 0.15, 300, 1280)]


@pytest.mark.parametrize("frac,spread,max_det,imgsz", CASES)
def test_postprocess_matches_oracle_on_random_heads(gpu_engine, frac, spread, max_det, imgsz):
    nc = 1
    m = E.Model(gpu_engine, G.build_yolov8(yolo_arch.synth_state_dict("n", nc, None, seed=0), nc, None, dtype=E.graph_dtype()))
    m.set_max_batch(2)
    rng = np.random.default_rng(int(frac * 1000) + int(spread * 10) + max_det + imgsz)
    shapes = m.head_shapes(KA.H0, KA.W0, imgsz)
    assert imgsz != 640 or [tuple(x[:2]) for x in shapes] == [x[:2] for x in KA.LEVELS]
    heads = [np.zeros((2, hh, ww, 68), np.float32) for (hh, ww, _) in shapes]
    for hd in heads:
        hd[..., :64] = rng.normal(0, spread, hd[..., :64].shape)
        logit = rng.normal(0, 2, hd[..., 64].shape)
        thr = np.quantile(logit, 1 - frac)
        hd[..., 64] = np.where(logit > thr, np.abs(logit) * 0.3 + 0.1, -20.0)
        hd[..., 65:] = -20.0
    boxes, _, counts = m.yolo_postprocess(heads, KA.H0, KA.W0, imgsz=imgsz, conf=0.5, iou=0.7, max_det=max_det)
    m.close()
    mo = ref.YoloV8Ref({}, nc, None)
    det = [torch.from_numpy(h[..., :65]).permute(0, 3, 1, 2).contiguous() for h in heads]
    out, cands = ref.non_max_suppression(mo.decode(det, []), 0.5, 0.7, None, max_det, nc=nc, return_candidates=True)
    for i in range(2):
        d = out[i].clone()
        d[:, :4] = ref.scale_boxes((shapes[0][0] * 8, shapes[0][1] * 8), d[:, :4], (KA.H0, KA.W0))
        want = d[:, :6].numpy()
        got = boxes[i, :counts[i]]
        assert len(want) == len(got), f"image {i}: {len(cands[i])} candidates, oracle kept {len(want)}, engine kept {counts[i]}"
        assert np.allclose(want, got, atol=2e-3), f"image {i}: first differing rank {int(np.argmax(~np.isclose(want, got, atol=2e-3).all(1)))}"
    if frac >= 0.95:
        assert max(len(c) for c in cands) > (4096 if imgsz == 640 else 16384), "the case is meant to span chunks / leave the LDS sort"
    elif imgsz == 1280:
        assert 8192 < max(len(c) for c in cands) <= 16384, "the case is meant for the upper half of the LDS sort"
