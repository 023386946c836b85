#!/usr/bin/env python
"""CPU study for a cheaper SiLU epilogue (DESIGN.md §5 "next"): how much rounding error may the activation carry
before the network-level parity criterion (RMS error vs the fp64 evaluation <= 1.5 x the fp32 oracle's own) is at
risk?  The fp32 oracle is re-run with every SiLU output perturbed by a uniformly random relative error of up to
+-k ulp (a stand-in for `v * rcp(1 + exp2(-v*log2e))`-style formulas built from 1-ulp hardware transcendentals) and
compared, like the engine is in tests/, with the fp64 evaluation of the same weights.

    python tools/silu_noise_study.py > profiles/silu_noise_study_r1.txt
"""
import sys
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import synth_weights, yolov8_ref as ref          # noqa: E402  (tuning tool: the oracle is the subject here)
from padel_analytics_amd import synth                         # noqa: E402
from tests import synth
from tests import parity                                      # noqa: E402


class NoisyRef(ref.YoloV8Ref):
    def __init__(self, *a, ulps=0.0, seed=0, **k):
        super().__init__(*a, **k)
        self.ulps = ulps
        self.gen = torch.Generator().manual_seed(seed)

    def _conv(self, x, prefix, k, s):
        y = super()._conv(x, prefix, k, s)
        if self.ulps:
            y = y * (1.0 + (torch.rand(y.shape, generator=self.gen) * 2 - 1) * (self.ulps * 2.0 ** -24))
        return y


def as_arrays(res):
    n = len(res)
    boxes = np.zeros((n, 300, 6), np.float32)
    counts = np.zeros(n, np.int32)
    for i, r in enumerate(res):
        counts[i] = len(r["boxes"])
        boxes[i, :counts[i]] = r["boxes"]
    return boxes, counts


def main():
    torch.set_num_threads(16)
    frames = synth.synthetic_frames(2, 360, 640, seed=3)
    srcs = [f[..., ::-1] for f in frames]
    print("scale  ulps   RMS px vs fp64   worst px   ratio to the exact-SiLU fp32 oracle   detections")
    for scale in ("n", "s"):
        sd = synth_weights.calibrated_state_dict(scale, 80, None, ref.preprocess(srcs, 640), 0.5, seed=5)
        r64 = ref.predict(ref.YoloV8Ref(sd, 80, None, dtype=torch.float64), srcs, 0.5, 0.7, 640, classes=[0])
        base = None
        for ulps in (0.0, 1.0, 2.0, 4.0, 8.0, 16.0):
            rms, worst, n = [], [], 0
            for seed in range(3 if ulps else 1):
                rr = ref.predict(NoisyRef(sd, 80, None, ulps=ulps, seed=seed), srcs, 0.5, 0.7, 640, classes=[0])
                b, c = as_arrays(rr)
                try:
                    rep = parity.compare_batch(r64, b, None, c, 0.5, 0.7)
                except AssertionError as e:
                    print(f"{scale:5s} {ulps:5.1f}   detection sets differ beyond a threshold-adjacent flip: {str(e)[:80]}")
                    continue
                rms.append(rep["rms_px"]); worst.append(rep["worst_px"]); n = rep["n"]
            if not rms:
                continue
            if base is None:
                base = float(np.mean(rms))
            print(f"{scale:5s} {ulps:5.1f}   {np.mean(rms):.3e}        {np.max(worst):.3e}   {np.mean(rms) / base:5.2f}                                   {n}")


if __name__ == "__main__":
    main()
