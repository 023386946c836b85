"""TEST INFRASTRUCTURE: a slow torch-CPU interpreter of the engine's op list (padel_analytics_amd.graph).

It un-packs the weight blob exactly the way csrc/conv_igemm.hip indexes it, so running it against the
oracle proves the host logic (BN fold, K-order packing, concat-by-slice wiring) without a GPU.  It is
never imported by the product package."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from padel_analytics_amd import graph as G


def unpack_conv(blob, o, f16: bool = False):
    """Weights of a conv op back in (npad, cin, k, k) order; fp16 graphs keep them as halves inside the fp32 blob
    words, 32 channels per k-step (csrc/conv_tap16.hip)."""
    k, cin, npad = o["ksize"], o["cin"], o["npad"]
    kc = 32 if f16 else 16
    steps = G.kstep_order(cin, k, kc)
    n = npad * len(steps) * kc
    if f16:
        wp = blob[o["w_off"]:o["w_off"] + n // 2].view(np.float16).astype(np.float32).reshape(npad, len(steps) * kc)
    else:
        wp = blob[o["w_off"]:o["w_off"] + n].reshape(npad, len(steps) * kc)
    w = np.zeros((npad, cin, k, k), np.float32)
    for i, (tap, c0) in enumerate(steps):
        w[:, c0:c0 + kc, tap // k, tap % k] = wp[:, i * kc:(i + 1) * kc]
    b = blob[o["b_off"]:o["b_off"] + npad]
    return w, b


def unpack_conv_h2(blob, o):
    """Weights of a DTYPE_H2 conv back in (npad, cin, k, k) order, as the VALUE its fp16 pairs stand for:
    (h + m / 2048) / row scale (csrc/h2_common.h); also returns the bias."""
    k, cin, npad = o["ksize"], o["cin"], o["npad"]
    steps = G.bx3_ksteps(cin, k)
    n = npad * len(steps) * 64
    planes = blob[o["w_off"]:o["w_off"] + n // 2].view(np.float16).reshape(npad, len(steps), 2, 32)
    inv = blob[o["reserved"]:o["reserved"] + npad]
    val = G.h2_value(planes[:, :, 0], planes[:, :, 1]) * inv[:, None, None]
    w = np.zeros((npad, cin, k, k), np.float32)
    for s, slots in enumerate(steps):
        for i, sl in enumerate(slots):
            if sl is not None:
                w[:, sl[0], sl[1] % k, sl[1] // k] = val[:, s, i]        # taps column-major (graph.py:pack_conv_weight_h2)
    return w, blob[o["b_off"]:o["b_off"] + npad]


def h2_round(x: torch.Tensor) -> torch.Tensor:
    """What an h2 buffer holds after a value was stored in it: h + m / 2048 of the fp16 pair (22-23 bits)."""
    h, m = G.h2_split(x.numpy())
    return torch.from_numpy(G.h2_value(h, m))


def act(x, a):
    if a == G.ACT_SILU:
        return F.silu(x)
    if a == G.ACT_RELU:
        return F.relu(x)
    if a == G.ACT_SIGMOID:
        return torch.sigmoid(x)
    if a == G.ACT_LEAKY:
        return F.leaky_relu(x, 0.01)
    return x


@torch.no_grad()
def run(graph: G.Graph, net_in: torch.Tensor = None, buf0: torch.Tensor = None, stale: float = 0.0):
    """net_in: (B,3,H,W) fp32 in [0,1] for YOLO graphs; buf0: (B,C,H,W) for TrackNet graphs.
    Returns the list of buffers as NCHW tensors."""
    blob = graph.blob()
    ref = net_in if net_in is not None else buf0
    B, _, H, W = ref.shape
    f16 = getattr(graph, "dtype", 0) == G.DTYPE_F16
    h2 = getattr(graph, "dtype", 0) == G.DTYPE_H2
    heads = set(graph.head_buf)
    # fp16 graphs: what is written to a non-head buffer is rounded to fp16 (storage), arithmetic stays fp32 —
    # the engine's fp16 path up to the order of the fp32 accumulation.  `stale` fills the buffers first, to prove
    # that pad channels read under zero weights / never-written channels cannot leak into results.
    bufs = [torch.full((B, c, H >> l, W >> l), float(stale)) for (l, c) in graph.bufs]
    if buf0 is not None:
        bufs[0][:] = h2_round(buf0) if h2 else buf0

    def store(bi, lo, val):
        if f16 and bi not in heads:
            val = val.half().float()
        if h2 and bi not in heads:
            val = h2_round(val)
        bufs[bi][:, lo:lo + val.shape[1]] = val

    for o in graph.ops:
        kd = o["kind"]
        if kd == G.OP_STEM:
            w = torch.from_numpy(blob[o["w_off"]:o["w_off"] + o["cout"] * 27].reshape(o["cout"], 3, 3, 3).transpose(0, 3, 1, 2).copy())
            b = torch.from_numpy(blob[o["b_off"]:o["b_off"] + o["cout"]].copy())
            store(o["out_buf"], o["out_choff"], F.silu(F.conv2d(net_in, w, b, stride=2, padding=1)))
        elif kd == G.OP_CONV:
            w, b = unpack_conv_h2(blob, o) if h2 else unpack_conv(blob, o, f16)
            x = bufs[o["in_buf"]][:, o["in_choff"]:o["in_choff"] + o["cin"]]
            y = act(F.conv2d(x, torch.from_numpy(w), torch.from_numpy(b.copy()), stride=o["stride"], padding=o["ksize"] // 2), o["act"])
            y = y[:, :o["cout"]]
            if o["res_buf"] >= 0:
                y = y + bufs[o["res_buf"]][:, o["res_choff"]:o["res_choff"] + o["cout"]]
            store(o["out_buf"], o["out_choff"], y)
        elif kd == G.OP_SPPF_POOL:
            c = o["cin"]
            t = bufs[o["in_buf"]]
            for k in range(3):
                src = t[:, o["in_choff"] + k * c:o["in_choff"] + (k + 1) * c]
                t[:, o["in_choff"] + (k + 1) * c:o["in_choff"] + (k + 2) * c] = F.max_pool2d(src, 5, 1, 2)
        elif kd == G.OP_UPSAMPLE2X:
            x = bufs[o["in_buf"]][:, o["in_choff"]:o["in_choff"] + o["cin"]]
            bufs[o["out_buf"]][:, o["out_choff"]:o["out_choff"] + o["cin"]] = F.interpolate(x, scale_factor=2.0, mode="nearest")
        elif kd == G.OP_MAXPOOL2:
            x = bufs[o["in_buf"]][:, o["in_choff"]:o["in_choff"] + o["cin"]]
            bufs[o["out_buf"]][:, o["out_choff"]:o["out_choff"] + o["cin"]] = F.max_pool2d(x, 2, 2)
        else:
            raise AssertionError(kd)
    return bufs
