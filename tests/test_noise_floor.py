"""Pins (on CPU) the statement DESIGN.md §4 and the GPU parity criterion rest on: on the synthetic
checkpoints, the reference algorithm evaluated in fp32 is itself only reproducible to more than the 1e-3 px
bar — the oracle in fp32 and the same oracle in fp64 differ by ~1e-2 px on box coordinates while agreeing on
every class id and (away from threshold-adjacent decisions) on the detection set."""
import numpy as np
import torch

from oracle import synth_weights, yolov8_ref as ref
from tests import synth
from tests import parity


def test_fp32_oracle_vs_fp64_oracle_exceeds_1e3_px():
    frames = synth.synthetic_frames(2, 360, 640, seed=3)
    srcs = [f[..., ::-1] for f in frames]
    sd = synth_weights.calibrated_state_dict("n", 80, None, ref.preprocess(srcs, 640), 0.5, seed=5)
    r32 = ref.predict(ref.YoloV8Ref(sd, 80, None), srcs, 0.5, 0.7, 640, classes=[0])
    r64 = ref.predict(ref.YoloV8Ref(sd, 80, None, dtype=torch.float64), srcs, 0.5, 0.7, 640, classes=[0])
    n = len(frames)
    boxes = np.zeros((n, 300, 6), np.float32)
    counts = np.zeros(n, np.int32)
    for i, r in enumerate(r64):
        counts[i] = len(r["boxes"])
        boxes[i, :counts[i]] = r["boxes"]
    rep = parity.compare_batch(r32, boxes, None, counts, 0.5, 0.7)
    assert rep["n"] >= 10
    assert rep["worst_px"] > 1e-3, "fp32 noise floor unexpectedly below the 1e-3 px bar"
    assert rep["worst_px"] < 0.2 and rep["worst_score"] < 1e-3
