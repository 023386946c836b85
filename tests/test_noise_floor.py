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


def test_least_squares_heads_detect_the_rectangles_and_keep_a_floor_above_1e3_px():
    """Round 6: heads fitted by ridge regression on the clip's own rectangles (oracle/synth_weights.py:
    fitted_state_dict — DFL logits peaked at the true distances, class logit high inside a rectangle, keypoints at fixed places)
    instead of random last convs.  The fit is real: the kept boxes overlap the painted rectangles far more than the random heads'
    do.  And the conjecture it was built to test does NOT hold: the reference algorithm in fp32 and in fp64 still differ by more
    than 1e-3 px on these heads (yolov8n, 640 wide; 10 x more on yolov8m) — the bar sits inside the fp32 evaluation noise of the
    graph whatever the last layer is."""
    from tests import helpers
    frames, rects = synth.synthetic_frames(2, 360, 640, seed=3, return_rects=True)
    srcs = [f[..., ::-1] for f in frames]
    sd_rand = synth_weights.calibrated_state_dict("n", 80, None, ref.preprocess(srcs, 640), 0.5, seed=5)
    sd, rep = helpers.fitted_state_dict("n", 80, None, srcs, rects, (360, 640), 640, 0.5, seed=5)
    for l in range(3):
        assert rep[f"level{l}"]["positives"] >= 16 and rep[f"level{l}"]["r2_dfl"] > 0.05 and rep[f"level{l}"]["r2_cls"] > 0.05, rep
    for k, v in sd.items():
        if k.startswith("model.22.") and k.endswith((".2.weight", ".2.bias")):
            assert np.array_equal(np.asarray(v, np.float32).astype(np.float16).astype(np.float32), v), f"{k}: not fp16 numbers"
    r32 = ref.predict(ref.YoloV8Ref(sd, 80, None), srcs, 0.5, 0.7, 640, classes=[0])
    r64 = ref.predict(ref.YoloV8Ref(sd, 80, None, dtype=torch.float64), srcs, 0.5, 0.7, 640, classes=[0])
    rr = ref.predict(ref.YoloV8Ref(sd_rand, 80, None), srcs, 0.5, 0.7, 640, classes=[0])
    iou_fit = np.concatenate([helpers.best_iou_with_rects(r["boxes"], rects[i]) for i, r in enumerate(r32)])
    iou_rand = np.concatenate([helpers.best_iou_with_rects(r["boxes"], rects[i]) for i, r in enumerate(rr)])
    assert len(iou_fit) >= 10 and iou_fit.mean() > 0.3 and iou_fit.mean() > 1.5 * iou_rand.mean(), (iou_fit.mean(), iou_rand.mean())
    n = len(frames)
    boxes = np.zeros((n, 300, 6), np.float32)
    counts = np.zeros(n, np.int32)
    for i, r in enumerate(r64):
        counts[i] = len(r["boxes"])
        boxes[i, :counts[i]] = r["boxes"]
    floor = parity.compare_batch(r32, boxes, None, counts, 0.5, 0.7)
    print("least-squares heads: mean best IoU", iou_fit.mean(), "random heads", iou_rand.mean(), "fp32-vs-fp64 floor", floor["worst_px"], "px")
    assert 5e-4 < floor["worst_px"] < 0.05 and floor["worst_score"] < 1e-3
