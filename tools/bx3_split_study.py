#!/usr/bin/env python
"""CPU study (no GPU): does the bf16x3 operand split by TRUNCATION explain the engine's excess whole-graph error over
the fp32 oracle's own noise floor, and does a ROUND-TO-NEAREST split remove it?

The oracle's convs are replaced by an emulation of the engine's arithmetic: both operands split exactly into three
bf16-valued fp32 tensors, the six kept cross products (each exact in fp32) evaluated as six fp32 convs and summed
smallest first.  Accumulation order differs from the MFMA's, so this reproduces the *bias* of the dropped terms, not
the engine bit for bit.  Output: L-inf / RMS px after NMS vs the fp64 evaluation for {plain fp32, trunc split, RN
split}, on the low-noise-head graphs of tests/test_gpu_yolo_parity.py.

    python tools/bx3_split_study.py [m|n] > profiles/bx3_split_study_r3.txt
"""
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from oracle import synth_weights, yolov8_ref as ref          # noqa: E402
from padel_analytics_amd import synth                        # noqa: E402
from tests import synth
from tests import parity                                     # noqa: E402


def trunc_bf16(x):
    return (x.view(torch.int32) & -65536).view(torch.float32)


def rn_bf16(x):
    u = x.view(torch.int32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & -65536).view(torch.float32)


def split3(x, rnd):
    h = rnd(x)
    r = x - h
    m = rnd(r)
    l = r - m
    assert torch.equal(rnd(l), l), "third term must be exact"
    return h, m, l


class Bx3Ref(ref.YoloV8Ref):
    def __init__(self, *a, rnd=trunc_bf16, **k):
        super().__init__(*a, **k)
        self.rnd = rnd

    def _conv6(self, x, w, b, s, pad):
        xh, xm, xl = split3(x.contiguous(), self.rnd)
        wh, wm, wl = split3(w.contiguous(), self.rnd)
        acc = None
        for (a_, w_) in ((xl, wh), (xh, wl), (xm, wm), (xm, wh), (xh, wm), (xh, wh)):     # smallest first, like the kernels
            t = F.conv2d(a_, w_, None, stride=s, padding=pad)
            acc = t if acc is None else acc + t
        return acc + b.view(1, -1, 1, 1)

    def _conv(self, x, prefix, k, s):
        if prefix == "model.0":                                                            # the stem runs on the fp32 MFMA
            return super()._conv(x, prefix, k, s)
        if prefix not in self._fused:
            self._fused[prefix] = ref.fuse_conv_bn(self.sd, prefix)
        w, b = self._fused[prefix]
        return F.silu(self._conv6(x, w, b, s, k // 2))

    def _branch(self, x, br, l):
        p = f"model.22.{br}.{l}"
        x = self._conv(x, f"{p}.0", 3, 1)
        x = self._conv(x, f"{p}.1", 3, 1)
        return self._conv6(x, ref._t(self.sd, f"{p}.2.weight").float(), ref._t(self.sd, f"{p}.2.bias").float(), 1, 0)


def arrays(res, nk=0):
    n = len(res)
    boxes = np.zeros((n, 300, 6), np.float32)
    kpts = np.zeros((n, 300, nk), np.float32) if nk else None
    counts = np.zeros(n, np.int32)
    for i, r in enumerate(res):
        counts[i] = len(r["boxes"])
        boxes[i, :counts[i]] = r["boxes"]
        if nk and counts[i]:
            kpts[i, :counts[i]] = r["kpts"].reshape(counts[i], -1)
    return boxes, kpts, counts


def run(scale, seeds):
    torch.set_num_threads(8)
    for fseed, wseed in seeds:
        frames = synth.synthetic_frames(3, 720, 1280, seed=fseed)
        srcs = [f[..., ::-1] for f in frames]
        im = ref.preprocess(list(srcs), 640)
        sd = synth_weights.calibrated_state_dict(scale, 80, None, im, 0.5, wseed)
        for l in range(3):
            for nm in ("weight", "bias"):
                k = f"model.22.cv2.{l}.2.{nm}"
                sd[k] = (sd[k] * np.float32(0.02)).astype(np.float16).astype(np.float32)
        kw = dict(conf=0.5, iou=0.7, imgsz=640, classes=[0])
        r64 = ref.predict(ref.YoloV8Ref(sd, 80, None, dtype=torch.float64), srcs, **kw)
        rows = {"fp32 oracle": ref.predict(ref.YoloV8Ref(sd, 80, None), srcs, **kw),
                "bx3 trunc": ref.predict(Bx3Ref(sd, 80, None, rnd=trunc_bf16), srcs, **kw),
                "bx3 RN": ref.predict(Bx3Ref(sd, 80, None, rnd=rn_bf16), srcs, **kw)}
        print(f"detect-{scale}-tight frames seed {fseed} weights seed {wseed}: {sum(len(r['boxes']) for r in r64)} detections")
        for name, r in rows.items():
            b, k, c = arrays(r)
            s = parity.compare_batch(r64, b, k, c, 0.5, 0.7)
            print(f"   {name:12s} vs fp64: L-inf {s['worst_px']:.3e} px   RMS {s['rms_px']:.3e} px   n {s['n']}")


if __name__ == "__main__":
    scale = sys.argv[1] if len(sys.argv) > 1 else "m"
    run(scale, [(13, 17), (3, 5), (21, 23)])
