#!/usr/bin/env python
"""GPU diagnostic: pa_yolo_postprocess (decode + NMS kernels) against the oracle's decode + non_max_suppression on random
head maps with a controlled number of candidates per image (exercises the bit-mask NMS across its row blocks, the
sequential fallback, and max_det saturation)."""
import sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import yolov8_ref as ref
from padel_analytics_amd import engine as E, graph as G, yolo_arch
from tests import known_answers as KA

eng = E.default_engine(0)
nc = 1
m = E.Model(eng, G.build_yolov8(yolo_arch.synth_state_dict("n", nc, None, seed=0), nc, None, dtype=E.graph_dtype()))
m.set_max_batch(2)
rng = np.random.default_rng(0)
for frac, spread, max_det in [(0.01, 3.0, 300), (0.05, 3.0, 300), (0.2, 3.0, 300), (0.2, 0.5, 300), (0.6, 3.0, 300), (0.95, 3.0, 300), (0.2, 3.0, 50), (0.05, 6.0, 300)]:
    heads = KA.blank_heads(2, 68)
    for hd in heads:
        hd[..., :64] = rng.normal(0, spread, hd[..., :64].shape)
        logit = rng.normal(0, 2, hd[..., 64].shape)
        thr = np.quantile(logit, 1 - frac)
        hd[..., 64] = np.where(logit > thr, np.abs(logit) * 0.3 + 0.1, -20.0)
        hd[..., 65:] = -20.0
    boxes, _, counts = m.yolo_postprocess(heads, KA.H0, KA.W0, imgsz=640, conf=0.5, iou=0.7, max_det=max_det)
    mo = ref.YoloV8Ref({}, nc, None)
    det = [torch.from_numpy(h[..., :65]).permute(0, 3, 1, 2).contiguous() for h in heads]
    pred = mo.decode(det, [])
    out, cands = ref.non_max_suppression(pred, 0.5, 0.7, None, max_det, nc=nc, return_candidates=True)
    for i in range(2):
        d = out[i].clone()
        d[:, :4] = ref.scale_boxes((384, 640), d[:, :4], (KA.H0, KA.W0))
        want = d[:, :6].numpy()
        got = boxes[i, :counts[i]]
        same = len(want) == len(got) and np.allclose(want, got, atol=2e-3)
        first_bad = -1
        if not same:
            for k in range(min(len(want), len(got))):
                if not np.allclose(want[k], got[k], atol=2e-3):
                    first_bad = k
                    break
        print(f"frac {frac} spread {spread} max_det {max_det} image {i}: candidates {len(cands[i])} oracle kept {len(want)} engine kept {counts[i]} {'OK' if same else 'MISMATCH at rank ' + str(first_bad)}")
        if not same and first_bad >= 0:
            print("   want", want[first_bad], "got", got[first_bad])
m.close()
