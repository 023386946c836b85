"""Shared test helpers."""
from __future__ import annotations

import numpy as np
import torch

from oracle import synth_weights, yolov8_ref as ref


def calibrated_state_dict(scale, nc, kpt_shape, sources, imgsz, conf, seed=0, frac=0.01):
    """Calibrated synthetic weights for `sources` (list/array of HWC u8 images in upstream's BGR
    convention, i.e. exactly what is handed to ref.predict)."""
    im = ref.preprocess(list(sources[:2]), imgsz)
    return synth_weights.calibrated_state_dict(scale, nc, kpt_shape, im, conf, seed, frac)


def rects_in_network_pixels(rects, frame_hw, imgsz, stretch=False):
    """The painted rectangles of tests/synth.py (frame pixels) in network-input pixels: the letterbox of the detect path
    (``ref.letterbox_geometry``, auto padding) or the pose path's stretch to imgsz x imgsz."""
    h0, w0 = frame_hw
    if stretch:
        sx, sy, left, top = imgsz / w0, imgsz / h0, 0.0, 0.0
    else:
        nw, nh, top, _, left, _ = ref.letterbox_geometry(h0, w0, imgsz, True)
        sx, sy = nw / w0, nh / h0
    return [[(x0 * sx + left, y0 * sy + top, x1 * sx + left, y1 * sy + top) for (x0, y0, x1, y1) in rr] for rr in rects]


def fitted_state_dict(scale, nc, kpt_shape, sources, rects, frame_hw, imgsz, conf, seed=0, stretch=False, ridge=1e-2):
    """Checkpoint with least-squares heads (oracle/synth_weights.py:fitted_state_dict) for `sources` (what is handed to
    ref.predict) and the rectangles painted into the frames they were made from.  -> (state dict, fit report)."""
    im = ref.preprocess(list(sources), imgsz)
    return synth_weights.fitted_state_dict(scale, nc, kpt_shape, im, rects_in_network_pixels(rects, frame_hw, imgsz, stretch),
                                           conf, seed, ridge=ridge)


def best_iou_with_rects(boxes, rects):
    """Per box (x1, y1, x2, y2, ...) the largest IoU with any rectangle (x0, y0, x1, y1)."""
    out = []
    for b in boxes:
        best = 0.0
        for (x0, y0, x1, y1) in rects:
            iw, ih = min(b[2], x1) - max(b[0], x0), min(b[3], y1) - max(b[1], y0)
            if iw > 0 and ih > 0:
                inter = iw * ih
                best = max(best, inter / ((b[2] - b[0]) * (b[3] - b[1]) + (x1 - x0) * (y1 - y0) - inter))
        out.append(best)
    return np.asarray(out)
