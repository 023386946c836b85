"""YOLOv8 detect / pose architecture table and checkpoint naming.

The reference never defines the network itself: ``players_tracker.py:303`` and
``players_keypoints_tracker.py:238`` call ``ultralytics.YOLO(model_path)`` and the
graph comes out of the pickled checkpoint.  This module restates the published
YOLOv8 layer table so the engine can (a) build its op list
from a plain ``state_dict`` and (b) manufacture seeded synthetic checkpoints with
the exact Ultralytics key names / shapes (there are no real ``.pt`` files offline).

Nothing in here touches a device.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Optional

import numpy as np

# (depth, width, max_channels) — SURVEY.md Appendix A "Scales".
SCALES = {
    "n": (0.33, 0.25, 1024),
    "s": (0.33, 0.50, 1024),
    "m": (0.67, 0.75, 768),
    "l": (1.00, 1.00, 512),
    "x": (1.00, 1.25, 512),
}

REG_MAX = 16
BN_EPS = 1e-3          # ultralytics initialize_weights() sets BatchNorm2d.eps = 1e-3
STRIDES = (8, 16, 32)


def make_divisible(x: float, divisor: int = 8) -> int:
    return int(math.ceil(x / divisor) * divisor)


@dataclass(frozen=True)
class ArchDims:
    """Channel widths / repeat counts of one YOLOv8 scale."""
    scale: str
    c1: int   # "64"   stem
    c2: int   # "128"  P2
    c3: int   # "256"  P3
    c4: int   # "512"  P4
    c5: int   # "1024" P5
    n3: int   # repeats of the n=3 C2f blocks
    n6: int   # repeats of the n=6 C2f blocks


def arch_dims(scale: str) -> ArchDims:
    depth, width, max_ch = SCALES[scale]
    c = lambda x: make_divisible(min(x, max_ch) * width, 8)
    n = lambda r: max(round(r * depth), 1)
    return ArchDims(scale, c(64), c(128), c(256), c(512), c(1024), n(3), n(6))


def head_dims(d: ArchDims, nc: int, kpt_shape: Optional[tuple]) -> tuple:
    """(c2 box-branch width, c3 cls-branch width, c4 kpt-branch width or 0, nk)."""
    ch0 = d.c3
    c2 = max(16, ch0 // 4, REG_MAX * 4)
    c3 = max(ch0, min(nc, 100))
    if kpt_shape is None:
        return c2, c3, 0, 0
    nk = int(kpt_shape[0]) * int(kpt_shape[1])
    return c2, c3, max(ch0 // 4, nk), nk


# --------------------------------------------------------------------------------------
# state_dict specification (Ultralytics key names)
# --------------------------------------------------------------------------------------

def _conv_bn(spec, prefix, cin, cout, k):
    spec[f"{prefix}.conv.weight"] = (cout, cin, k, k)
    spec[f"{prefix}.bn.weight"] = (cout,)
    spec[f"{prefix}.bn.bias"] = (cout,)
    spec[f"{prefix}.bn.running_mean"] = (cout,)
    spec[f"{prefix}.bn.running_var"] = (cout,)
    spec[f"{prefix}.bn.num_batches_tracked"] = ()


def _c2f(spec, prefix, cin, cout, n):
    c = cout // 2
    _conv_bn(spec, f"{prefix}.cv1", cin, 2 * c, 1)
    _conv_bn(spec, f"{prefix}.cv2", (2 + n) * c, cout, 1)
    for j in range(n):
        _conv_bn(spec, f"{prefix}.m.{j}.cv1", c, c, 3)
        _conv_bn(spec, f"{prefix}.m.{j}.cv2", c, c, 3)


def yolov8_state_spec(scale: str, nc: int, kpt_shape: Optional[tuple] = None) -> "OrderedDict[str, tuple]":
    """Ordered ``name -> shape`` of every tensor in an Ultralytics YOLOv8 detect/pose state_dict."""
    d = arch_dims(scale)
    s: "OrderedDict[str, tuple]" = OrderedDict()
    _conv_bn(s, "model.0", 3, d.c1, 3)
    _conv_bn(s, "model.1", d.c1, d.c2, 3)
    _c2f(s, "model.2", d.c2, d.c2, d.n3)
    _conv_bn(s, "model.3", d.c2, d.c3, 3)
    _c2f(s, "model.4", d.c3, d.c3, d.n6)
    _conv_bn(s, "model.5", d.c3, d.c4, 3)
    _c2f(s, "model.6", d.c4, d.c4, d.n6)
    _conv_bn(s, "model.7", d.c4, d.c5, 3)
    _c2f(s, "model.8", d.c5, d.c5, d.n3)
    _conv_bn(s, "model.9.cv1", d.c5, d.c5 // 2, 1)
    _conv_bn(s, "model.9.cv2", (d.c5 // 2) * 4, d.c5, 1)
    _c2f(s, "model.12", d.c5 + d.c4, d.c4, d.n3)
    _c2f(s, "model.15", d.c4 + d.c3, d.c3, d.n3)
    _conv_bn(s, "model.16", d.c3, d.c3, 3)
    _c2f(s, "model.18", d.c3 + d.c4, d.c4, d.n3)
    _conv_bn(s, "model.19", d.c4, d.c4, 3)
    _c2f(s, "model.21", d.c4 + d.c5, d.c5, d.n3)
    c2, c3, c4, nk = head_dims(d, nc, kpt_shape)
    chs = (d.c3, d.c4, d.c5)
    for branch, width, nout in (("cv2", c2, 4 * REG_MAX), ("cv3", c3, nc)):
        for l, ch in enumerate(chs):
            _conv_bn(s, f"model.22.{branch}.{l}.0", ch, width, 3)
            _conv_bn(s, f"model.22.{branch}.{l}.1", width, width, 3)
            s[f"model.22.{branch}.{l}.2.weight"] = (nout, width, 1, 1)
            s[f"model.22.{branch}.{l}.2.bias"] = (nout,)
    s["model.22.dfl.conv.weight"] = (1, REG_MAX, 1, 1)
    if kpt_shape is not None:
        for l, ch in enumerate(chs):
            _conv_bn(s, f"model.22.cv4.{l}.0", ch, c4, 3)
            _conv_bn(s, f"model.22.cv4.{l}.1", c4, c4, 3)
            s[f"model.22.cv4.{l}.2.weight"] = (nk, c4, 1, 1)
            s[f"model.22.cv4.{l}.2.bias"] = (nk,)
    return s


def count_parameters(spec) -> int:
    """Trainable-parameter count the way Ultralytics reports it (BN running stats excluded)."""
    total = 0
    for k, shp in spec.items():
        if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            continue
        total += int(np.prod(shp)) if len(shp) else 1
    return total


def infer_arch_from_state_dict(sd) -> dict:
    """Recover (scale, nc, kpt nk) from tensor shapes of a YOLOv8 state_dict."""
    c1 = int(sd["model.0.conv.weight"].shape[0])
    c5 = int(sd["model.7.conv.weight"].shape[0])
    n3 = 0
    while f"model.2.m.{n3}.cv1.conv.weight" in sd:
        n3 += 1
    scale = None
    for sc in SCALES:
        d = arch_dims(sc)
        if d.c1 == c1 and d.c5 == c5 and d.n3 == n3:
            scale = sc
            break
    if scale is None:
        raise ValueError(f"unrecognised YOLOv8 scale (stem={c1}, P5={c5}, n3={n3})")
    nc = int(sd["model.22.cv3.0.2.weight"].shape[0])
    nk = int(sd["model.22.cv4.0.2.weight"].shape[0]) if "model.22.cv4.0.2.weight" in sd else 0
    return {"scale": scale, "nc": nc, "nk": nk}


# --------------------------------------------------------------------------------------
# conv inventory (for FLOP accounting: bench roofline + known-answer tests)
# --------------------------------------------------------------------------------------

def conv_inventory(scale: str, nc: int, kpt_shape: Optional[tuple], net_h: int, net_w: int):
    """List of (name, cin, cout, k, stride, hout, wout) for every conv of the graph at a
    given network-input size.  2*MAC summed over it reproduces the published GFLOPs
    (SURVEY.md §8(c) known-answer 1)."""
    d = arch_dims(scale)
    out = []

    def conv(name, cin, cout, k, s, h, w):
        ho, wo = (h + s - 1) // s if k == 3 else h // s, (w + s - 1) // s if k == 3 else w // s
        out.append((name, cin, cout, k, s, ho, wo))
        return ho, wo

    def c2f(name, cin, cout, n, h, w):
        c = cout // 2
        conv(f"{name}.cv1", cin, 2 * c, 1, 1, h, w)
        for j in range(n):
            conv(f"{name}.m.{j}.cv1", c, c, 3, 1, h, w)
            conv(f"{name}.m.{j}.cv2", c, c, 3, 1, h, w)
        conv(f"{name}.cv2", (2 + n) * c, cout, 1, 1, h, w)

    h, w = conv("model.0", 3, d.c1, 3, 2, net_h, net_w)
    h, w = conv("model.1", d.c1, d.c2, 3, 2, h, w)
    c2f("model.2", d.c2, d.c2, d.n3, h, w)
    h3, w3 = conv("model.3", d.c2, d.c3, 3, 2, h, w)
    c2f("model.4", d.c3, d.c3, d.n6, h3, w3)
    h4, w4 = conv("model.5", d.c3, d.c4, 3, 2, h3, w3)
    c2f("model.6", d.c4, d.c4, d.n6, h4, w4)
    h5, w5 = conv("model.7", d.c4, d.c5, 3, 2, h4, w4)
    c2f("model.8", d.c5, d.c5, d.n3, h5, w5)
    conv("model.9.cv1", d.c5, d.c5 // 2, 1, 1, h5, w5)
    conv("model.9.cv2", (d.c5 // 2) * 4, d.c5, 1, 1, h5, w5)
    c2f("model.12", d.c5 + d.c4, d.c4, d.n3, h4, w4)
    c2f("model.15", d.c4 + d.c3, d.c3, d.n3, h3, w3)
    conv("model.16", d.c3, d.c3, 3, 2, h3, w3)
    c2f("model.18", d.c3 + d.c4, d.c4, d.n3, h4, w4)
    conv("model.19", d.c4, d.c4, 3, 2, h4, w4)
    c2f("model.21", d.c4 + d.c5, d.c5, d.n3, h5, w5)
    c2, c3, c4, nk = head_dims(d, nc, kpt_shape)
    branches = [("cv2", c2, 4 * REG_MAX), ("cv3", c3, nc)]
    if kpt_shape is not None:
        branches.append(("cv4", c4, nk))
    for l, (ch, hh, ww) in enumerate(((d.c3, h3, w3), (d.c4, h4, w4), (d.c5, h5, w5))):
        for br, width, nout in branches:
            conv(f"model.22.{br}.{l}.0", ch, width, 3, 1, hh, ww)
            conv(f"model.22.{br}.{l}.1", width, width, 3, 1, hh, ww)
            conv(f"model.22.{br}.{l}.2", width, nout, 1, 1, hh, ww)
    return out


def conv_flops(inv) -> float:
    """2 * MACs over a conv inventory."""
    return float(sum(2 * cin * cout * k * k * ho * wo for (_, cin, cout, k, _, ho, wo) in inv))


# --------------------------------------------------------------------------------------
# synthetic checkpoints (SURVEY.md §8(d) recipe)
# --------------------------------------------------------------------------------------

# conv init variance = gain / fan_in.  The running BN statistics are random (not matched to the data),
# so activation scale drifts exponentially with depth; these per-scale gains keep features O(1..10)
# and head logits O(10) through the 60-90 conv layers (measured with the oracle).
SYNTH_GAIN = {"n": 1.8, "s": 1.6, "m": 1.4, "l": 1.3, "x": 1.3}


def synth_state_dict(scale: str, nc: int, kpt_shape: Optional[tuple] = None, seed: int = 0,
                     cls_bias: float = -4.0, gain: Optional[float] = None) -> "OrderedDict[str, np.ndarray]":
    """Seeded random weights with Ultralytics names/shapes.

    conv ~ N(0, gain/fan_in) (SYNTH_GAIN); BN gamma~U(.8,1.6), beta~U(-.3,.3), mean~U(-.2,.2), var~U(.5,1.5);
    every value is rounded through fp16 the way real checkpoints are stored.  ``cls_bias``
    shifts the classification logits so only a small fraction of anchors pass ``conf``
    (bench/tests calibrate it with the oracle).
    """
    rng = np.random.default_rng(seed)
    gain = SYNTH_GAIN[scale] if gain is None else gain
    spec = yolov8_state_spec(scale, nc, kpt_shape)
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    r16 = lambda a: a.astype(np.float16).astype(np.float32)
    for name, shp in spec.items():
        if name.endswith("num_batches_tracked"):
            sd[name] = np.array(0, dtype=np.int64)
        elif name.endswith("dfl.conv.weight"):
            sd[name] = np.arange(REG_MAX, dtype=np.float32).reshape(shp)
        elif name.endswith("conv.weight") or (name.endswith(".2.weight")):
            fan_in = shp[1] * shp[2] * shp[3]
            sd[name] = r16(rng.normal(0.0, math.sqrt(gain / fan_in), size=shp).astype(np.float32))
        elif name.endswith("bn.weight"):
            sd[name] = r16(rng.uniform(0.8, 1.6, size=shp).astype(np.float32))
        elif name.endswith("bn.bias"):
            sd[name] = r16(rng.uniform(-0.3, 0.3, size=shp).astype(np.float32))
        elif name.endswith("running_mean"):
            sd[name] = r16(rng.uniform(-0.2, 0.2, size=shp).astype(np.float32))
        elif name.endswith("running_var"):
            sd[name] = r16(rng.uniform(0.5, 1.5, size=shp).astype(np.float32))
        elif name.endswith(".2.bias"):
            if ".cv3." in name:
                sd[name] = r16(np.full(shp, cls_bias, dtype=np.float32)
                               + rng.uniform(-0.5, 0.5, size=shp).astype(np.float32))
            elif ".cv2." in name:
                sd[name] = r16(rng.uniform(0.5, 1.5, size=shp).astype(np.float32))
            else:
                sd[name] = r16(rng.uniform(-0.5, 0.5, size=shp).astype(np.float32))
        else:
            raise AssertionError(name)
    return sd
