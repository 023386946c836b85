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
