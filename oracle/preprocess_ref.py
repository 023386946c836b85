"""ORACLE — test infrastructure only.  numpy restatement of Pillow's 8-bit bicubic ``Image.resize``
(``ImagingResample``: ``precompute_coeffs`` + ``normalize_coeffs_8bpc`` + horizontal then vertical pass),
the host op at ``players_keypoints_tracker.py:264-266`` and ``ball_tracker/iterable.py:80,188``.

PINNED: tests/test_preprocess_ref.py checks it bit-exactly against Pillow itself (importable here) on the
sizes the reference uses (720x1280 -> 1280x1280, 1080x1920 -> 1280x1280, -> 512x288).  The engine's device
kernel (csrc/kernels_misc.hip:resample_pass_kernel, tables from engine.cpp:pil_coeffs) follows the same
arithmetic and is checked against Pillow on the GPU (tests/test_gpu_preprocess.py)."""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_coeffs(in_size: int, out_size: int):
    """-> (bounds (out,2) [xmin, count], coefs (out,ksize) int32, ksize)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(w)
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def _pass(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    bounds, kk, ksize = pil_coeffs(img.shape[axis], out_size)
    a = np.moveaxis(img.astype(np.int64), axis, 0)
    out = np.empty((out_size,) + a.shape[1:], np.int64)
    for o in range(out_size):
        lo, n = bounds[o]
        acc = np.full(a.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for j in range(n):
            acc += a[lo + j] * int(kk[o, j])
        out[o] = acc >> PRECISION_BITS
    return np.moveaxis(np.clip(out, 0, 255).astype(np.uint8), 0, axis)


def pil_resize_bicubic_u8(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """img HxWxC uint8 -> out_h x out_w x C; horizontal pass first, a pass is skipped when its size is unchanged."""
    if img.shape[1] != out_w:
        img = _pass(img, out_w, 1)
    if img.shape[0] != out_h:
        img = _pass(img, out_h, 0)
    return img
