"""Host-side graph builder: ``state_dict`` -> (buffer table, op list, packed fp32 weight blob).

This is the one-time work ``YOLO(model_path)`` + ``AutoBackend(fuse=True)`` do upstream
(``players_tracker.py:303``, SURVEY.md Appendix A "Checkpoint"): fold BatchNorm into the conv
(``W' = W * g/sqrt(var+eps)``, ``b' = beta - g*mu/sqrt(var+eps)``, fp32), then lay the weights out the
way the gfx950 kernels read them (``csrc/conv_igemm.hip``):

* conv weights ``[Npad][Ktot]`` with K ordered (32-channel chunk, 3x3 tap, 16-channel half) so that
  one k-step of the implicit GEMM is one contiguous 64-byte run per output channel;
* torch.cat / chunk / nn.Upsample never materialise: producers write channel slices of the
  consumer's concat buffer (C2f, SPPF, the FPN/PAN concats, the Detect/Pose head map).

Nothing here touches a device; the result is handed to ``libpadel_hip.so`` through
``pa_model_create`` (include/padel_hip.h).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import os

import numpy as np

from . import yolo_arch

# mirror of include/padel_hip.h
OP_STEM, OP_CONV, OP_SPPF_POOL, OP_UPSAMPLE2X, OP_MAXPOOL2 = 1, 2, 3, 4, 5
ACT_NONE, ACT_SILU, ACT_RELU, ACT_SIGMOID, ACT_LEAKY = 0, 1, 2, 3, 4
TASK_DETECT, TASK_POSE, TASK_TRACKNET = 0, 1, 2
DTYPE_F32, DTYPE_F16, DTYPE_H2 = 0, 1, 2


def pad16(c: int) -> int:
    return (c + 15) // 16 * 16


def kstep_order(cin: int, ksize: int, kc: int = 16):
    """[(tap, c0)] in the order the conv kernels walk K: k-steps of ``kc`` channels (one 64-byte run: 16 fp32 or
    32 fp16), grouped (2*kc-channel chunk, tap, half); cin must be a multiple of kc."""
    taps = ksize * ksize
    steps = []
    for c in range(cin // (2 * kc)):
        for tap in range(taps):
            for half in (0, 1):
                steps.append((tap, c * 2 * kc + half * kc))
    if cin % (2 * kc):
        for tap in range(taps):
            steps.append((tap, (cin // (2 * kc)) * 2 * kc))
    return steps


def pack_conv_weight(w: np.ndarray, kc: int = 16) -> np.ndarray:
    """(Cout, Cin, k, k) fp32 with Cout multiple of 16, Cin multiple of kc -> [Cout][Ktot] in kernel K order."""
    cout, cin, k, _ = w.shape
    assert cout % 16 == 0 and cin % kc == 0
    steps = kstep_order(cin, k, kc)
    out = np.empty((cout, len(steps) * kc), np.float32)
    for i, (tap, c0) in enumerate(steps):
        out[:, i * kc:(i + 1) * kc] = w[:, c0:c0 + kc, tap // k, tap % k]
    return out


def split_bf16x3(w: np.ndarray):
    """fp32 -> three uint16 arrays (hi, mid, lo bf16 bit patterns) with hi + mid + lo == w EXACTLY: each part is the
    truncation of what the previous parts left (8 + 8 + 8 significand bits)."""
    w = np.ascontiguousarray(w, np.float32)
    b = w.view(np.uint32)
    hi = b & np.uint32(0xFFFF0000)
    r = w - hi.view(np.float32)
    mid = r.view(np.uint32) & np.uint32(0xFFFF0000)
    lo = (r - mid.view(np.float32)).view(np.uint32)
    assert not (lo & np.uint32(0xFFFF)).any(), "the third part of an fp32 value always fits bf16"
    return (hi >> 16).astype(np.uint16), (mid >> 16).astype(np.uint16), (lo >> 16).astype(np.uint16)


# k-slot order of one 32-channel k-step of the bf16x3 kernels: lane group q holds channels 4q..4q+3 of the first
# 16-channel sub-row and 16+4q..16+4q+3 of the second (what two 16-byte fragment reads give it; csrc/conv_tap_bx3.hip)
BX3_PERM = np.array([4 * (s // 8) + (s % 8) if s % 8 < 4 else 16 + 4 * (s // 8) + (s % 8 - 4) for s in range(32)])


def bx3_ksteps(cin: int, k: int):
    """K-steps of the bf16x3 kernels: each is a list of 32 (channel, tap) slots in sub-row order (16 + 16), None =
    zero padding.  Full 32-channel chunks walk their taps; the last 16 channels of a cin % 32 == 16 layer pair TAPS in
    a 3x3 conv (tap 2t | tap 2t+1: 5 steps instead of 9 half-empty ones) and stay half-empty in a 1x1."""
    steps = []
    nfull = cin // 32
    for c in range(nfull):
        for tap in range(k * k):
            steps.append([(c * 32 + i, tap) for i in range(32)])
    if cin % 32:
        c0 = nfull * 32
        if k == 3:
            for t in range(5):
                steps.append([(c0 + i, 2 * t) for i in range(16)] +
                             [((c0 + i, 2 * t + 1) if 2 * t + 1 < 9 else None) for i in range(16)])
        else:
            for tap in range(k * k):
                steps.append([(c0 + i, tap) for i in range(16)] + [None] * 16)
    return steps


def pack_conv_weight_bx3(w: np.ndarray) -> np.ndarray:
    """(Cout, Cin, k, k) fp32, Cout % 16 == 0, Cin % 16 == 0 -> uint16 [Cout][k-step][hi|mid|lo][32] for the bf16x3
    kernels (k-steps: ``bx3_ksteps``; the 32 slots of a step are stored in the lane order BX3_PERM)."""
    cout, cin, k, _ = w.shape
    assert cout % 16 == 0 and cin % 16 == 0
    steps = bx3_ksteps(cin, k)
    out = np.empty((cout, len(steps), 3, 32), np.uint16)
    for s, slots in enumerate(steps):
        blk = np.zeros((cout, 32), np.float32)
        for i, sl in enumerate(slots):
            if sl is not None:
                blk[:, i] = w[:, sl[0], sl[1] // k, sl[1] % k]
        hi, mid, lo = split_bf16x3(blk[:, BX3_PERM])
        out[:, s, 0], out[:, s, 1], out[:, s, 2] = hi, mid, lo
    return out


# ---- "h2": fp32 values as PAIRS of fp16 numbers, x ~ h + m / 2048 (22-23 significant bits; csrc/h2_common.h) ----------
# Activations of a DTYPE_H2 graph live in HBM in this form, 4 bytes per channel like fp32: per pixel and 16-channel GROUP
# 64 bytes = [h of the 16 channels | m of the 16 channels].  The producer's epilogue encodes once; every consumer reads
# ready-made fp16 MFMA operands and evaluates a * w with THREE products (ah*wh + (ah*wm + am*wh) / 2048) instead of the
# six of the bf16x3 scheme.
H2_MAX = 65504.0
H2_RSCALE = 2048.0


def h2_split(x: np.ndarray):
    """fp32 array -> (h, m) float16 arrays with h = RN16(x), m = RN16((x - h) * 2048); |x| is clamped to the fp16 range
    (the device encoder raises the model's overflow flag in that case, engine.py re-runs on the bf16x3 path)."""
    xs = np.clip(np.asarray(x, np.float32), -H2_MAX, H2_MAX)
    h = xs.astype(np.float16)
    m = ((xs - h.astype(np.float32)) * np.float32(H2_RSCALE)).astype(np.float16)
    return h, m


def h2_value(h: np.ndarray, m: np.ndarray) -> np.ndarray:
    """The fp32 value a pair stands for (exact: the two parts do not overlap)."""
    return h.astype(np.float32) + m.astype(np.float32) * np.float32(1.0 / H2_RSCALE)


def h2_encode_nhwc(x: np.ndarray) -> np.ndarray:
    """(..., C) fp32 with C % 16 == 0 -> (..., C) uint32-sized words holding the group layout, returned as float32 view
    of shape (..., C): per 16-channel group 16 h halves then 16 m halves."""
    x = np.asarray(x, np.float32)
    c = x.shape[-1]
    assert c % 16 == 0
    h, m = h2_split(x.reshape(x.shape[:-1] + (c // 16, 16)))
    return np.ascontiguousarray(np.concatenate([h, m], axis=-1)).view(np.float32).reshape(x.shape)


def h2_decode_nhwc(e: np.ndarray) -> np.ndarray:
    e = np.ascontiguousarray(e, np.float32)
    c = e.shape[-1]
    hm = e.view(np.float16).reshape(e.shape[:-1] + (c // 16, 32))
    return h2_value(hm[..., :16], hm[..., 16:]).reshape(e.shape)


def h2_row_scale(w2d: np.ndarray) -> np.ndarray:
    """Per output channel power of two s with max |w| * s in [2^12, 2^13): keeps both planes of a weight row in the
    normal fp16 range whatever the magnitude of the folded weights; all-zero (padding) and denormal rows get 1."""
    mx = np.abs(w2d).max(axis=1)
    s = np.ones_like(mx, dtype=np.float32)
    # rows whose largest weight is below the smallest normal fp32 number (dead BN-folded channels: gamma ~ 0) count as
    # all-zero: their scale would leave the fp32 range (inf * 0 = NaN planes, 1 / inf = 0).  The exponent is clamped so that
    # both s and 1 / s are normal fp32 numbers for every other row
    nz = mx >= np.float32(np.finfo(np.float32).tiny)
    e = np.clip(12.0 - np.floor(np.log2(mx[nz].astype(np.float64))), -100.0, 100.0)
    s[nz] = np.exp2(e).astype(np.float32)
    return s


def pack_conv_weight_h2(w: np.ndarray):
    """(Cout, Cin, k, k) fp32, Cout % 16 == 0, Cin % 16 == 0 -> (uint16 [Cout][k-step][h|m][32], fp32 [Cout] = 1 / row
    scale).  K-steps are ``bx3_ksteps`` (tap pairing for the 16-channel tail of a 3x3) with the taps of a 3x3 walked
    COLUMN-major (tap t = (ky, kx) = (t % 3, t // 3), csrc/h2_common.h:h2_tap_ky); the 32 slots of a step are in natural
    order: an MFMA lane group q holds slots 8q..8q+7."""
    cout, cin, k, _ = w.shape
    assert cout % 16 == 0 and cin % 16 == 0
    sc = h2_row_scale(w.reshape(cout, -1))
    steps = bx3_ksteps(cin, k)
    out = np.empty((cout, len(steps), 2, 32), np.uint16)
    for s, slots in enumerate(steps):
        blk = np.zeros((cout, 32), np.float32)
        for i, sl in enumerate(slots):
            if sl is not None:
                blk[:, i] = w[:, sl[0], sl[1] % k, sl[1] // k]       # taps column-major: k-step t = (ky, kx) = (t % 3, t // 3)
        h, m = h2_split(blk * sc[:, None])
        out[:, s, 0], out[:, s, 1] = h.view(np.uint16), m.view(np.uint16)
    return out, (1.0 / sc).astype(np.float32)


def h2_weights_single(planes: np.ndarray) -> bool:
    """True if the correction (m) plane of packed h2 weights is all zero: every weight is an fp16 number times its row's
    power of two (include/padel_hip.h PA_CONV_W_SINGLE)."""
    return not bool((planes[:, :, 1, :] & 0x7FFF).any())


def fold_bn_split(sd, prefix: str, eps: float):
    """The pieces of ``fold_bn`` kept apart: (raw conv weight, per-channel BatchNorm scale, folded bias) — the same fp32
    operations in the same order for scale and bias; ``fold_bn``'s weight is ``raw * scale`` rounded to fp32."""
    w = np.asarray(sd[f"{prefix}.conv.weight"], np.float32)
    g = np.asarray(sd[f"{prefix}.bn.weight"], np.float32)
    beta = np.asarray(sd[f"{prefix}.bn.bias"], np.float32)
    mu = np.asarray(sd[f"{prefix}.bn.running_mean"], np.float32)
    var = np.asarray(sd[f"{prefix}.bn.running_var"], np.float32)
    scale = (g / np.sqrt(np.float32(eps) + var)).astype(np.float32)
    bf = (beta - (g * mu) / np.sqrt(var + np.float32(eps))).astype(np.float32)
    return w, scale, bf


def fp16_exact(w: np.ndarray) -> bool:
    w = np.asarray(w, np.float32)
    return bool(np.array_equal(w.astype(np.float16).astype(np.float32), w))


def fold_bn(sd, prefix: str, eps: float):
    """Conv+BN fold in fp32, same operation order as ultralytics' fuse_conv_and_bn."""
    w = np.asarray(sd[f"{prefix}.conv.weight"], np.float32)
    g = np.asarray(sd[f"{prefix}.bn.weight"], np.float32)
    beta = np.asarray(sd[f"{prefix}.bn.bias"], np.float32)
    mu = np.asarray(sd[f"{prefix}.bn.running_mean"], np.float32)
    var = np.asarray(sd[f"{prefix}.bn.running_var"], np.float32)
    scale = g / np.sqrt(np.float32(eps) + var)
    wf = (w.reshape(w.shape[0], -1) * scale[:, None]).reshape(w.shape).astype(np.float32)
    bf = (beta - (g * mu) / np.sqrt(var + np.float32(eps))).astype(np.float32)
    return wf, bf


@dataclass
class Graph:
    task: int
    nc: int = 0
    nk: int = 0
    kpt_dim: int = 0
    bufs: list = field(default_factory=list)      # (level, channels)
    ops: list = field(default_factory=list)       # dicts mirroring pa_op_desc
    chunks: list = field(default_factory=list)    # weight blob pieces
    n_floats: int = 0
    head_buf: tuple = (-1, -1, -1)
    in_channels: int = 0
    out_channels: int = 0
    dtype: int = DTYPE_F32                        # storage type of activations / conv weights (pa_dtype)
    bx3: bool = True                              # fp32 graphs: also pack the bf16x3 (3-way split) weight planes

    @property
    def kalign(self) -> int:
        """Channel granularity of a conv's input slice: one 64-byte k-step (16 fp32 / 32 fp16 channels)."""
        return 32 if self.dtype == DTYPE_F16 else 16

    def padk(self, c: int) -> int:
        return (c + self.kalign - 1) // self.kalign * self.kalign

    # ---- construction helpers
    def buf(self, level: int, channels: int) -> int:
        self.bufs.append((level, channels))
        return len(self.bufs) - 1

    def _add(self, arr: np.ndarray) -> int:
        arr = np.ascontiguousarray(arr, np.float32).reshape(-1)
        off = self.n_floats
        padn = (-arr.size) % 4
        self.chunks.append(arr)
        if padn:
            self.chunks.append(np.zeros(padn, np.float32))
        self.n_floats += arr.size + padn
        return off

    def conv(self, src, dst, w, b, k, s, act, res=None, out_width=None, out_scale=None):
        """src = (buf, choff, width read), dst = (buf, choff).  ``w`` is (cout, cin, k, k) with the
        real channel counts; input channels are zero-padded up to the slice width, output channels up
        to ``out_width`` (those rows are zero, so the kernel writes act(0) there).
        ``out_scale`` (h2 graphs only): per-output-channel factor applied to the accumulated sum before the bias — BatchNorm's
        scale kept out of the weights, so that a checkpoint's fp16 weights stay fp16 numbers (PA_CONV_W_SINGLE)."""
        sb, so, sw = src
        cout, cin = w.shape[:2]
        if sw % self.kalign:
            # the slice is narrower than a whole number of k-steps: read on into the neighbouring channels of the
            # same buffer under ZERO weights (they hold finite activations, or the buffer is widened with pad
            # channels that stay zero / finite) — fp16 k-steps are 32 channels wide, yolov8's C2f halves are not
            sw = self.padk(sw)
            lvl, ch = self.bufs[sb]
            if so + sw > ch:
                self.bufs[sb] = (lvl, so + sw)
        assert sw % self.kalign == 0 and sw >= cin, (sw, cin)
        ow = cout if out_width is None else out_width
        npad = pad16(ow)
        wp = np.zeros((npad, sw, k, k), np.float32)
        wp[:cout, :cin] = w
        bp = np.zeros(npad, np.float32)
        bp[:cout] = b
        w3_off = 0
        flags = 0
        assert out_scale is None or self.dtype == DTYPE_H2, "out_scale: h2 graphs only"
        if self.dtype == DTYPE_F16:
            w_off = self._add(np.ascontiguousarray(pack_conv_weight(wp, 32).astype(np.float16)).view(np.float32))
        elif self.dtype == DTYPE_H2:
            planes, inv_scale = pack_conv_weight_h2(wp)
            if out_scale is not None:             # (1 / row scale is a power of two: the product is the scale's own bits)
                osc = np.ones(npad, np.float32)
                osc[:cout] = np.asarray(out_scale, np.float32)
                inv_scale = (inv_scale * osc).astype(np.float32)
            flags = FLAG_W_SINGLE if h2_weights_single(planes) else 0
            w_off = self._add(np.ascontiguousarray(planes).view(np.float32))
            w3_off = self._add(inv_scale)         # `reserved` of an h2 conv: its per-output-channel output scale (1 / row scale [x BN scale])
        else:
            w_off = self._add(pack_conv_weight(wp))
            if self.bx3:            # the same weights pre-split for the bf16x3 kernels (engine tuning impl=2)
                w3_off = self._add(np.ascontiguousarray(pack_conv_weight_bx3(wp)).view(np.float32))
        b_off = self._add(bp)
        self.ops.append(dict(kind=OP_CONV, in_buf=sb, in_choff=so, cin=sw, out_buf=dst[0], out_choff=dst[1],
                             cout=ow, ksize=k, stride=s, act=act, res_buf=-1 if res is None else res[0],
                             res_choff=0 if res is None else res[1], npad=npad, w_off=w_off, b_off=b_off, reserved=w3_off,
                             flags=flags))

    def blob(self) -> np.ndarray:
        return np.concatenate(self.chunks) if self.chunks else np.zeros(0, np.float32)

    def conv_flops(self, net_h: int, net_w: int) -> float:
        """Real (unpadded would need the spec; this counts the op list as executed) 2*MAC per image."""
        total = 0.0
        for o in self.ops:
            if o["kind"] not in (OP_CONV, OP_STEM):
                continue
            lvl = self.bufs[o["out_buf"]][0]
            hw = (net_h >> lvl) * (net_w >> lvl)
            kk = 27 if o["kind"] == OP_STEM else o["cin"] * o["ksize"] ** 2
            total += 2.0 * hw * o["cout"] * kk
        return total


FLAG_W_SINGLE = 1          # pa_op_desc.flags: PA_CONV_W_SINGLE (include/padel_hip.h)
# tests / tuning (PADEL_UNFOLDED_BN=0): fold BatchNorm into the weights of h2 graphs even when the checkpoint's weights are fp16 numbers
UNFOLDED_BN: bool = os.environ.get("PADEL_UNFOLDED_BN", "1") != "0"

# tests / tuning (PADEL_HEAD_SPLIT=0|1): force the Detect / Pose head's first convs merged (False) or per branch (True)
HEAD_SPLIT: Optional[bool] = {"0": False, "1": True}.get(os.environ.get("PADEL_HEAD_SPLIT", ""))


def build_yolov8(sd, nc: int, kpt_shape: Optional[tuple] = None, dtype: str = "f32") -> Graph:
    """YOLOv8 detect / pose graph (SURVEY.md Appendix A layer table) over the engine's op set.

    ``dtype="f16"`` (BASELINE configs[4]): same graph with fp16 activations and conv weights (fp32 accumulate, fp32
    biases, fp32 Detect/Pose head maps); BatchNorm is folded in fp32 first, then the folded weights are rounded to
    fp16 once — what ``model.half()`` after ``fuse()`` does upstream."""
    info = yolo_arch.infer_arch_from_state_dict(sd)
    d = yolo_arch.arch_dims(info["scale"])
    assert info["nc"] == nc, (info, nc)
    c2h, c3h, c4h, nk = yolo_arch.head_dims(d, nc, kpt_shape)
    assert nk == info["nk"], (nk, info)
    g = Graph(task=TASK_POSE if kpt_shape else TASK_DETECT, nc=nc, nk=nk, kpt_dim=int(kpt_shape[1]) if kpt_shape else 0,
              dtype={"f32": DTYPE_F32, "f16": DTYPE_F16, "h2": DTYPE_H2}[dtype])
    eps = yolo_arch.BN_EPS
    fuse = lambda p: fold_bn(sd, p, eps)

    # h2 graphs: a checkpoint whose conv weights are fp16 numbers (Ultralytics stores `model.half()`) keeps them that way —
    # BatchNorm's scale goes into the conv's per-channel output scale instead of into the weights, the correction plane of
    # the packed weights is zero and the kernels run two products per operand pair instead of three (PA_CONV_W_SINGLE).
    # sum(w a) * scale + shift instead of sum(fl32(w * scale) a) + shift: the same value to fp32 rounding (the head maps move
    # by 0.12-0.18 x the fp32 oracle's own distance from its fp64 evaluation: DESIGN.md 3.5).  Any other weights: folded as before.
    split_bn = g.dtype == DTYPE_H2 and UNFOLDED_BN

    def parts(prefix):
        """-> (weights to pack, bias, out_scale | None)"""
        if split_bn:
            w, sc, b = fold_bn_split(sd, prefix, eps)
            if fp16_exact(w):
                return w, b, sc
        w, b = fuse(prefix)
        return w, b, None

    def cbs(prefix, src, dst, k, s, res=None, out_width=None):
        w, b, sc = parts(prefix)
        g.conv(src, dst, w, b, k, s, ACT_SILU, res, out_width, out_scale=sc)

    def nblocks(i):
        n = 0
        while f"model.{i}.m.{n}.cv1.conv.weight" in sd:
            n += 1
        return n

    def c2f(i, src, cout, shortcut, level, dst):
        n = nblocks(i)
        c = cout // 2
        assert c % 16 == 0
        cat = g.buf(level, (2 + n) * c)
        ct = g.padk(c)                  # the scratch between a bottleneck's two convs is written at k-step width
        tmp = g.buf(level, ct)
        cbs(f"model.{i}.cv1", src, (cat, 0), 1, 1)
        for j in range(n):
            cbs(f"model.{i}.m.{j}.cv1", (cat, (1 + j) * c, c), (tmp, 0), 3, 1, out_width=ct if ct != c else None)
            cbs(f"model.{i}.m.{j}.cv2", (tmp, 0, ct), (cat, (2 + j) * c), 3, 1,
                res=(cat, (1 + j) * c) if shortcut else None)
        cbs(f"model.{i}.cv2", (cat, 0, (2 + n) * c), dst, 1, 1)

    c1, c2, c3, c4, c5 = d.c1, d.c2, d.c3, d.c4, d.c5
    # stem: straight from the u8 network input
    c1p = g.padk(c1)               # stem output at k-step width: zero rows -> SiLU(0) = 0 in the pad channels
    b0 = g.buf(1, c1p)
    w0, bias0 = fuse("model.0")
    w0p = np.zeros((c1p, 27), np.float32)
    w0p[:c1] = np.ascontiguousarray(w0.transpose(0, 2, 3, 1)).reshape(c1, 27)          # [cout][ky][kx][c]
    b0p = np.zeros(c1p, np.float32)
    b0p[:c1] = bias0
    w_off = g._add(w0p)
    b_off = g._add(b0p)
    g.ops.append(dict(kind=OP_STEM, in_buf=0, in_choff=0, cin=3, out_buf=b0, out_choff=0, cout=c1p, ksize=3, stride=2,
                      act=ACT_SILU, res_buf=-1, res_choff=0, npad=c1p, w_off=w_off, b_off=b_off))
    b1 = g.buf(2, c2)
    cbs("model.1", (b0, 0, c1p), (b1, 0), 3, 2)
    b2 = g.buf(2, c2)
    c2f(2, (b1, 0, c2), c2, True, 2, (b2, 0))
    b3 = g.buf(3, c3)
    cbs("model.3", (b2, 0, c2), (b3, 0), 3, 2)
    cat14 = g.buf(3, c4 + c3)      # [upsample(model.12) | model.4]
    c2f(4, (b3, 0, c3), c3, True, 3, (cat14, c4))
    b5 = g.buf(4, c4)
    cbs("model.5", (cat14, c4, c3), (b5, 0), 3, 2)
    cat11 = g.buf(4, c5 + c4)      # [upsample(model.9) | model.6]
    c2f(6, (b5, 0, c4), c4, True, 4, (cat11, c5))
    b7 = g.buf(5, c5)
    cbs("model.7", (cat11, c5, c4), (b7, 0), 3, 2)
    b8 = g.buf(5, c5)
    c2f(8, (b7, 0, c5), c5, True, 5, (b8, 0))
    cat20 = g.buf(5, c4 + c5)      # [model.19 | model.9]
    ch = c5 // 2
    cat9 = g.buf(5, 4 * ch)
    cbs("model.9.cv1", (b8, 0, c5), (cat9, 0), 1, 1)
    g.ops.append(dict(kind=OP_SPPF_POOL, in_buf=cat9, in_choff=0, cin=ch, out_buf=cat9, out_choff=ch, cout=3 * ch,
                      ksize=5, stride=1, act=0, res_buf=-1, res_choff=0, npad=0, w_off=0, b_off=0))
    cbs("model.9.cv2", (cat9, 0, 4 * ch), (cat20, c4), 1, 1)

    def upsample(src, dst):
        g.ops.append(dict(kind=OP_UPSAMPLE2X, in_buf=src[0], in_choff=src[1], cin=src[2], out_buf=dst[0],
                          out_choff=dst[1], cout=src[2], ksize=0, stride=0, act=0, res_buf=-1, res_choff=0, npad=0,
                          w_off=0, b_off=0))

    upsample((cat20, c4, c5), (cat11, 0))
    cat17 = g.buf(4, c3 + c4)      # [model.16 | model.12]
    c2f(12, (cat11, 0, c5 + c4), c4, False, 4, (cat17, c3))
    upsample((cat17, c3, c4), (cat14, 0))
    b15 = g.buf(3, c3)
    c2f(15, (cat14, 0, c4 + c3), c3, False, 3, (b15, 0))
    cbs("model.16", (b15, 0, c3), (cat17, 0), 3, 2)
    b18 = g.buf(4, c4)
    c2f(18, (cat17, 0, c3 + c4), c4, False, 4, (b18, 0))
    cbs("model.19", (b18, 0, c4), (cat20, 0), 3, 2)
    b21 = g.buf(5, c5)
    c2f(21, (cat20, 0, c4 + c5), c5, False, 5, (b21, 0))

    # Detect / Pose head: the first 3x3 convs of the box / cls / kpt branches share their input, so
    # they run as ONE conv with concatenated (16-padded) output slices — except, on h2 graphs, at P3 / P4 when the merged
    # width is a whole number of neither 48- nor 64-channel tiles (the 13-keypoint pose head of the m scale: 64 + 192 + 48 = 304
    # channels = 19 fragments on seven 48-channel tiles): there each branch is its own conv into its slice of the same
    # buffer, the 192-channel class branch on the 96-channel quad tiles (conv_patch_h2q.hip), the others on the 64- / 48-
    # channel patch tiles.  Measured on the bench's 64 x 1280^2 pose batch (profiles/r4o_head_split.txt): P3 4.55 -> 4.23 ms,
    # P4 2.24 -> 2.14, P5 (few tiles per conv) 0.96 -> 1.05; a detect head of 64 + 192 = 256 channels, which fills its tiles,
    # loses 4-7 % at 640^2 when split.  Same numbers either way (a row of the weight matrix does not know its neighbours);
    # HEAD_SPLIT forces one or the other for the tests.
    def split_head(lvl, tot):
        if HEAD_SPLIT is not None:
            return HEAD_SPLIT
        return g.dtype == DTYPE_H2 and lvl <= 4 and tot % 64 != 0 and tot % 48 != 0
    head_cs = (64 + nc + nk + 3) // 4 * 4
    heads = []
    branches = [("cv2", c2h, 4 * yolo_arch.REG_MAX, 0), ("cv3", c3h, nc, 64)]
    if kpt_shape:
        branches.append(("cv4", c4h, nk, 64 + nc))
    for l, (feat, chn, lvl) in enumerate(((b15, c3, 3), (b18, c4, 4), (b21, c5, 5))):
        widths = [g.padk(wd) for (_, wd, _, _) in branches]
        tot = sum(widths)
        wcat = np.zeros((tot, chn, 3, 3), np.float32)
        bcat = np.zeros(tot, np.float32)
        scat = np.ones(tot, np.float32)
        offs = []
        o = 0
        br_parts = [parts(f"model.22.{br}.{l}.0") for (br, _, _, _) in branches]
        if any(p[2] is None for p in br_parts):           # one branch has to be folded: fold them all (one conv, one rule)
            br_parts = [fuse(f"model.22.{br}.{l}.0") + (None,) for (br, _, _, _) in branches]
        for (br, wd, _, _), pw, (w, b, sc) in zip(branches, widths, br_parts):
            wcat[o:o + wd] = w
            bcat[o:o + wd] = b
            if sc is not None:
                scat[o:o + wd] = sc
            offs.append(o)
            o += pw
        use_sc = br_parts[0][2] is not None
        h0 = g.buf(lvl, tot)
        if split_head(lvl, tot):
            for (br, wd, _, _), pw, o in zip(branches, widths, offs):
                g.conv((feat, 0, chn), (h0, o), wcat[o:o + wd], bcat[o:o + wd], 3, 1, ACT_SILU, out_width=pw,
                       out_scale=scat[o:o + wd] if use_sc else None)
        else:
            g.conv((feat, 0, chn), (h0, 0), wcat, bcat, 3, 1, ACT_SILU, out_scale=scat if use_sc else None)
        hd = g.buf(lvl, head_cs)
        for (br, wd, nout, hoff), pw, o in zip(branches, widths, offs):
            h1 = g.buf(lvl, pw)
            cbs(f"model.22.{br}.{l}.1", (h0, o, pw), (h1, 0), 3, 1, out_width=pw)
            w = np.asarray(sd[f"model.22.{br}.{l}.2.weight"], np.float32)
            b = np.asarray(sd[f"model.22.{br}.{l}.2.bias"], np.float32)
            g.conv((h1, 0, pw), (hd, hoff), w, b, 1, 1, ACT_NONE)
        heads.append(hd)
    g.head_buf = tuple(heads)
    return g


TRACKNET_BN_EPS = 1e-5      # nn.BatchNorm2d default (reference models.py:9)


def build_tracknet(sd, dtype: str = "f32") -> Graph:
    """TrackNetV3 U-Net (reference ``trackers/ball_tracker/models.py:45-74``) over the engine's op set.

    Buffer 0 is the fp32 NHWC input with the 27 channels (background + 8 frames x RGB) zero-padded to 32.
    ``torch.cat([Upsample(x), skip])`` (:66,:68,:70) is a concat buffer whose first slice is written by the
    upsample op and whose second slice is written directly by the encoder block that produces the skip."""
    g = Graph(task=TASK_TRACKNET, dtype={"f32": DTYPE_F32, "h2": DTYPE_H2}[dtype])
    in_dim = int(np.asarray(sd["down_block_1.conv_1.conv.weight"]).shape[1])
    out_dim = int(np.asarray(sd["predictor.weight"]).shape[0])
    cin0 = pad16(in_dim)
    g.in_channels = cin0

    def block(prefix, src, dst):
        w, b = fold_bn(sd, prefix, TRACKNET_BN_EPS)
        g.conv(src, dst, w, b, 3, 1, ACT_RELU)

    def pool(src, dst):
        g.ops.append(dict(kind=OP_MAXPOOL2, in_buf=src[0], in_choff=src[1], cin=src[2], out_buf=dst[0], out_choff=dst[1],
                          cout=src[2], ksize=2, stride=2, act=0, res_buf=-1, res_choff=0, npad=0, w_off=0, b_off=0))

    def up(src, dst):
        g.ops.append(dict(kind=OP_UPSAMPLE2X, in_buf=src[0], in_choff=src[1], cin=src[2], out_buf=dst[0],
                          out_choff=dst[1], cout=src[2], ksize=0, stride=0, act=0, res_buf=-1, res_choff=0, npad=0,
                          w_off=0, b_off=0))

    x0 = g.buf(0, cin0)
    cat3 = g.buf(0, 128 + 64)        # [up(up_block_2) | x1]
    cat2 = g.buf(1, 256 + 128)       # [up(up_block_1) | x2]
    cat1 = g.buf(2, 512 + 256)       # [up(bottleneck) | x3]
    t = g.buf(0, 64)
    block("down_block_1.conv_1", (x0, 0, cin0), (t, 0))
    block("down_block_1.conv_2", (t, 0, 64), (cat3, 128))
    p1 = g.buf(1, 64)
    pool((cat3, 128, 64), (p1, 0))
    t = g.buf(1, 128)
    block("down_block_2.conv_1", (p1, 0, 64), (t, 0))
    block("down_block_2.conv_2", (t, 0, 128), (cat2, 256))
    p2 = g.buf(2, 128)
    pool((cat2, 256, 128), (p2, 0))
    ta, tb = g.buf(2, 256), g.buf(2, 256)
    block("down_block_3.conv_1", (p2, 0, 128), (ta, 0))
    block("down_block_3.conv_2", (ta, 0, 256), (tb, 0))
    block("down_block_3.conv_3", (tb, 0, 256), (cat1, 512))
    p3 = g.buf(3, 256)
    pool((cat1, 512, 256), (p3, 0))
    ba, bb, bc = g.buf(3, 512), g.buf(3, 512), g.buf(3, 512)
    block("bottleneck.conv_1", (p3, 0, 256), (ba, 0))
    block("bottleneck.conv_2", (ba, 0, 512), (bb, 0))
    block("bottleneck.conv_3", (bb, 0, 512), (bc, 0))
    up((bc, 0, 512), (cat1, 0))
    block("up_block_1.conv_1", (cat1, 0, 768), (ta, 0))
    block("up_block_1.conv_2", (ta, 0, 256), (tb, 0))
    u1 = g.buf(2, 256)
    block("up_block_1.conv_3", (tb, 0, 256), (u1, 0))
    up((u1, 0, 256), (cat2, 0))
    t = g.buf(1, 128)
    u2 = g.buf(1, 128)
    block("up_block_2.conv_1", (cat2, 0, 384), (t, 0))
    block("up_block_2.conv_2", (t, 0, 128), (u2, 0))
    up((u2, 0, 128), (cat3, 0))
    t = g.buf(0, 64)
    u3 = g.buf(0, 64)
    block("up_block_3.conv_1", (cat3, 0, 192), (t, 0))
    block("up_block_3.conv_2", (t, 0, 64), (u3, 0))
    out = g.buf(0, out_dim if out_dim % 4 == 0 else pad16(out_dim))
    g.conv((u3, 0, 64), (out, 0), np.asarray(sd["predictor.weight"], np.float32),
           np.asarray(sd["predictor.bias"], np.float32), 1, 1, ACT_SIGMOID)
    g.head_buf = (out, -1, -1)
    g.out_channels = out_dim
    return g


INPAINT_LAYERS = ("down_1", "down_2", "down_3", "buttleneck.conv_1", "buttleneck.conv_2", "up_1", "up_2", "up_3")


def build_inpaintnet(sd, dtype: str = "f32") -> Graph:
    """InpaintNet (reference ``trackers/ball_tracker/models.py:101-130``: Conv1d(k=3, same) + LeakyReLU U-Net over
    length-L coordinate sequences, sigmoid output) over the engine's op set (round 4).

    A sequence is one image row: NHWC ``(windows, 1, L, C)``.  ``Conv1d(k=3, padding=same)`` along the sequence is a 3x3
    convolution whose only non-zero kernel row is the middle one (``w2d[:, :, 1, kx] = w1d[:, :, kx]``; the rows above and
    below the single image row are padding anyway).  ``torch.cat([x, skip], 1)`` (:117-122) are concat buffers written in
    place by their producers, like everywhere else.  Buffer 0: the fp32 input ``[x, y, mask]`` zero-padded to 16 channels;
    the head buffer: 2 sigmoid outputs (padded to 4)."""
    g = Graph(task=TASK_TRACKNET, dtype={"f32": DTYPE_F32, "h2": DTYPE_H2}[dtype])
    g.in_channels = 16

    def c1d(name, src, dst, act=ACT_LEAKY, out_width=None):
        w1 = np.asarray(sd[f"{name}.weight"], np.float32)               # (cout, cin, 3)
        w2 = np.zeros(w1.shape[:2] + (3, 3), np.float32)
        w2[:, :, 1, :] = w1
        g.conv(src, dst, w2, np.asarray(sd[f"{name}.bias"], np.float32), 3, 1, act, None, out_width)

    x0 = g.buf(0, 16)
    cat3 = g.buf(0, 64 + 32)         # [up_2 | x1]
    cat2 = g.buf(0, 128 + 64)        # [up_1 | x2]
    cat1 = g.buf(0, 256 + 128)       # [bottleneck | x3]
    c1d("down_1.conv", (x0, 0, 16), (cat3, 64))
    c1d("down_2.conv", (cat3, 64, 32), (cat2, 128))
    c1d("down_3.conv", (cat2, 128, 64), (cat1, 256))
    t = g.buf(0, 256)
    c1d("buttleneck.conv_1.conv", (cat1, 256, 128), (t, 0))
    c1d("buttleneck.conv_2.conv", (t, 0, 256), (cat1, 0))
    c1d("up_1.conv", (cat1, 0, 384), (cat2, 0))
    c1d("up_2.conv", (cat2, 0, 192), (cat3, 0))
    u3 = g.buf(0, 32)
    c1d("up_3.conv", (cat3, 0, 96), (u3, 0))
    out = g.buf(0, 4)
    c1d("predictor", (u3, 0, 32), (out, 0), act=ACT_SIGMOID)
    g.head_buf = (out, -1, -1)
    g.out_channels = 2
    return g
