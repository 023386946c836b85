"""TEST INFRASTRUCTURE — data-calibrated synthetic YOLOv8 checkpoints.

There are no real weights offline (reference `.gitignore:8`, `README.md:31-32`).  Purely random
BatchNorm statistics make a 60-90 layer SiLU network either blow up or collapse to its biases, which
is useless for parity work, so — like a trained checkpoint — the running mean/var of every BatchNorm
are set to the statistics its conv actually produces on a calibration batch, walking the graph once
with the oracle.  Finally the class-logit biases are shifted so that ~`frac` of the anchors pass the
confidence threshold, which makes decode + NMS do real work.

Weight *synthesis* only: nothing measured or shipped runs through this file.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch
import torch.nn.functional as F

from oracle import yolov8_ref as ref
from padel_analytics_amd import yolo_arch


def _r16(a):
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


class _Calibrator(ref.YoloV8Ref):
    """Walks the graph like the oracle but sets BN running stats from the data before fusing."""

    def _conv(self, x, prefix, k, s):
        w = ref._t(self.sd, f"{prefix}.conv.weight").float()
        y = F.conv2d(x, w, None, stride=s, padding=k // 2)
        mu = y.mean(dim=(0, 2, 3))
        var = y.var(dim=(0, 2, 3), unbiased=False).clamp_min(1e-4)
        self.sd[f"{prefix}.bn.running_mean"] = _r16(mu.numpy())
        self.sd[f"{prefix}.bn.running_var"] = _r16(var.numpy())
        self._fused.pop(prefix, None)
        return super()._conv(x, prefix, k, s)


@torch.no_grad()
def calibrated_state_dict(scale: str, nc: int, kpt_shape: Optional[tuple], calib_input: torch.Tensor,
                          conf: float, seed: int = 0, frac: float = 0.01, target_class: int = 0):
    """calib_input: (B,3,H,W) fp32 network input (already preprocessed)."""
    sd = yolo_arch.synth_state_dict(scale, nc, kpt_shape, seed, cls_bias=0.0, gain=2.0)
    cal = _Calibrator(sd, nc, kpt_shape)
    det, _ = cal.head_raw(cal.features(calib_input))
    # make `target_class` the arg-max on most anchors that pass (reference filters classes=[0])
    logit_t = torch.cat([d[:, 64 + target_class].reshape(-1) for d in det]).numpy()
    target = float(np.log(conf / (1 - conf)))
    delta_t = target - float(np.quantile(logit_t, 1.0 - frac))
    if nc > 1:
        other = torch.cat([torch.cat([d[:, 64:64 + target_class], d[:, 64 + target_class + 1:64 + nc]], 1).amax(1).reshape(-1)
                           for d in det]).numpy()
        delta_o = (target - 1.0) - float(np.quantile(other, 1.0 - frac))
    for l in range(3):
        k = f"model.22.cv3.{l}.2.bias"
        b = np.asarray(sd[k], np.float32).copy()
        if nc > 1:
            b += np.float32(delta_o)
        b[target_class] = sd[k][target_class] + np.float32(delta_t)
        sd[k] = _r16(b)
    return sd


# ------------------------------------------------------------------------------------------------------------------------------
# Least-squares heads (round 6): a checkpoint with trained-like head statistics without any download.  The random last convs of
# the three head branches are replaced by the closed-form ridge regression of what a trained head would output on the clip's own
# rectangles: DFL logits peaked at the true left / top / right / bottom distances, a class logit that is high inside a rectangle and
# low outside, keypoints at fixed places of the rectangle.  Setup only (untimed, CPU, fp64); the fitted weights are rounded to
# fp16 numbers like every checkpoint tensor.

# 13 keypoints at fixed relative places (u, v) of a rectangle: a stick figure (head, shoulders, elbows, wrists, hips, knees, feet)
KPT_PATTERN = np.array([(0.50, 0.08), (0.32, 0.22), (0.68, 0.22), (0.22, 0.38), (0.78, 0.38), (0.18, 0.52), (0.82, 0.52),
                        (0.38, 0.55), (0.62, 0.55), (0.36, 0.74), (0.64, 0.74), (0.35, 0.93), (0.65, 0.93)], np.float64)
DFL_LOGIT_FLOOR = -8.0       # logits of far bins (trained heads keep them in a modest range)


def _ridge(phi: np.ndarray, t: np.ndarray, ridge: float):
    """argmin |[phi 1] w - t|^2 + lambda |w|^2 (no penalty on the bias), lambda = ridge x the mean feature energy x rows."""
    n, c = phi.shape
    a = np.concatenate([phi, np.ones((n, 1))], 1)
    g = a.T @ a
    lam = ridge * float(np.trace(g[:c, :c])) / c
    g[np.arange(c), np.arange(c)] += lam
    w = np.linalg.solve(g, a.T @ t)
    return w[:c].T.copy(), w[c].copy()               # (out, c), (out,)


@torch.no_grad()
def fitted_state_dict(scale: str, nc: int, kpt_shape: Optional[tuple], calib_input: torch.Tensor, rects_net, conf: float,
                      seed: int = 0, ridge: float = 1e-2, frac: float = 0.01, target_class: int = 0):
    """``rects_net``: per image of ``calib_input`` the rectangles (x0, y0, x1, y1) in NETWORK-INPUT pixels, in paint order.
    Returns the state dict and a report (positives and R^2 of the fits per level)."""
    sd = calibrated_state_dict(scale, nc, kpt_shape, calib_input, conf, seed, frac, target_class)
    o = ref.YoloV8Ref(sd, nc, kpt_shape, dtype=torch.float64)
    feats = o.features(calib_input.double())
    report = {}
    bins = np.arange(ref.REG_MAX, dtype=np.float64)
    for l, f in enumerate(feats):
        stride = (8, 16, 32)[l]
        B, _, H, W = f.shape
        ax = (np.arange(W) + 0.5)[None, :].repeat(H, 0)                  # anchor centres in cells
        ay = (np.arange(H) + 0.5)[:, None].repeat(W, 1)
        owner = np.full((B, H, W, 4), np.nan)                            # the topmost rectangle under every anchor centre (cells)
        for b in range(B):
            for (x0, y0, x1, y1) in rects_net[b]:
                r = np.array([x0, y0, x1, y1], np.float64) / stride
                inside = (ax > r[0]) & (ax < r[2]) & (ay > r[1]) & (ay < r[3])
                owner[b][inside] = r
        owner = owner.reshape(-1, 4)
        axf, ayf = np.tile(ax.reshape(-1), B), np.tile(ay.reshape(-1), B)
        ltrb = np.stack([axf - owner[:, 0], ayf - owner[:, 1], owner[:, 2] - axf, owner[:, 3] - ayf], 1)
        pos = np.isfinite(ltrb).all(1) & (np.nan_to_num(ltrb, nan=1e9).max(1) < ref.REG_MAX - 1.01)
        hid = {}
        for br in ("cv2", "cv3") + (("cv4",) if kpt_shape else ()):
            p = f"model.22.{br}.{l}"
            h = o._conv(o._conv(f, f"{p}.0", 3, 1), f"{p}.1", 3, 1)
            hid[br] = h.permute(0, 2, 3, 1).reshape(B * H * W, -1).numpy()
        rep = {"anchors": int(B * H * W), "positives": int(pos.sum())}
        r2 = lambda t, y: float(1.0 - ((t - y) ** 2).sum() / max(((t - t.mean(0)) ** 2).sum(), 1e-30))
        if pos.sum() >= 16:
            # DFL: per side 16 logits, -(bin - distance)^2 clipped from below
            d = ltrb[pos]                                                # (P, 4)
            t = np.maximum(DFL_LOGIT_FLOOR, -(bins[None, None, :] - d[:, :, None]) ** 2).reshape(len(d), 4 * ref.REG_MAX)
            w, b = _ridge(hid["cv2"][pos], t, ridge)
            rep["r2_dfl"] = r2(t, hid["cv2"][pos] @ w.T + b)
            sd[f"model.22.cv2.{l}.2.weight"] = _r16(w.reshape(w.shape[0], -1, 1, 1))
            sd[f"model.22.cv2.{l}.2.bias"] = _r16(b)
            if kpt_shape:
                nkp, ndim = kpt_shape
                rw, rh = owner[pos, 2] - owner[pos, 0], owner[pos, 3] - owner[pos, 1]
                kx = owner[pos, 0][:, None] + KPT_PATTERN[None, :nkp, 0] * rw[:, None]          # cells
                ky = owner[pos, 1][:, None] + KPT_PATTERN[None, :nkp, 1] * rh[:, None]
                t = np.zeros((int(pos.sum()), nkp, ndim))
                t[..., 0] = (kx - axf[pos][:, None] + 0.5) / 2.0         # decode: (raw * 2 + anchor - 0.5) * stride
                t[..., 1] = (ky - ayf[pos][:, None] + 0.5) / 2.0
                if ndim == 3:
                    t[..., 2] = 2.0                                       # visibility logit
                t = t.reshape(len(t), -1)
                w, b = _ridge(hid["cv4"][pos], t, ridge)
                rep["r2_kpt"] = r2(t, hid["cv4"][pos] @ w.T + b)
                sd[f"model.22.cv4.{l}.2.weight"] = _r16(w.reshape(w.shape[0], -1, 1, 1))
                sd[f"model.22.cv4.{l}.2.bias"] = _r16(b)
        # class logits over ALL anchors: the target class high inside a rectangle, everything else low
        t = np.full((B * H * W, nc), -6.0)
        t[:, target_class] = np.where(pos, 2.2, -4.6)
        w, b = _ridge(hid["cv3"], t, ridge)
        rep["r2_cls"] = r2(t[:, target_class], hid["cv3"] @ w[target_class] + b[target_class])
        sd[f"model.22.cv3.{l}.2.weight"] = _r16(w.reshape(w.shape[0], -1, 1, 1))
        sd[f"model.22.cv3.{l}.2.bias"] = _r16(b)
        report[f"level{l}"] = rep
    # ~frac of the anchors above the confidence threshold, as calibrated_state_dict does for the random heads
    o2 = ref.YoloV8Ref(sd, nc, kpt_shape, dtype=torch.float64)
    logit = torch.cat([o2._branch(f, "cv3", l)[:, target_class].reshape(-1) for l, f in enumerate(feats)]).numpy()
    delta = float(np.log(conf / (1 - conf))) - float(np.quantile(logit, 1.0 - frac))
    for l in range(3):
        k = f"model.22.cv3.{l}.2.bias"
        b = np.asarray(sd[k], np.float32).copy()
        b[target_class] += np.float32(delta)
        sd[k] = _r16(b)
    report["cls_bias_shift"] = delta
    return sd, report
