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
