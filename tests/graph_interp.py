"""TEST INFRASTRUCTURE: a slow torch-CPU interpreter of the engine's op list (padel_analytics_amd.graph).

It un-packs the weight blob exactly the way csrc/conv_igemm.hip indexes it, so running it against the
oracle proves the host logic (BN fold, K-order packing, concat-by-slice wiring) without a GPU.  It is
never imported by the product package."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from padel_analytics_amd import graph as G


def unpack_conv(blob, o):
    k, cin, npad = o["ksize"], o["cin"], o["npad"]
    steps = G.kstep_order(cin, k)
    wp = blob[o["w_off"]:o["w_off"] + npad * len(steps) * 16].reshape(npad, len(steps) * 16)
    w = np.zeros((npad, cin, k, k), np.float32)
    for i, (tap, c0) in enumerate(steps):
        w[:, c0:c0 + 16, tap // k, tap % k] = wp[:, i * 16:(i + 1) * 16]
    b = blob[o["b_off"]:o["b_off"] + npad]
    return w, b


def act(x, a):
    if a == G.ACT_SILU:
        return F.silu(x)
    if a == G.ACT_RELU:
        return F.relu(x)
    if a == G.ACT_SIGMOID:
        return torch.sigmoid(x)
    return x


@torch.no_grad()
def run(graph: G.Graph, net_in: torch.Tensor = None, buf0: torch.Tensor = None):
    """net_in: (B,3,H,W) fp32 in [0,1] for YOLO graphs; buf0: (B,C,H,W) for TrackNet graphs.
    Returns the list of buffers as NCHW tensors."""
    blob = graph.blob()
    ref = net_in if net_in is not None else buf0
    B, _, H, W = ref.shape
    bufs = [torch.zeros(B, c, H >> l, W >> l) for (l, c) in graph.bufs]
    if buf0 is not None:
        bufs[0][:] = buf0
    for o in graph.ops:
        kd = o["kind"]
        if kd == G.OP_STEM:
            w = torch.from_numpy(blob[o["w_off"]:o["w_off"] + o["cout"] * 27].reshape(o["cout"], 3, 3, 3).transpose(0, 3, 1, 2).copy())
            b = torch.from_numpy(blob[o["b_off"]:o["b_off"] + o["cout"]].copy())
            bufs[o["out_buf"]][:, o["out_choff"]:o["out_choff"] + o["cout"]] = F.silu(F.conv2d(net_in, w, b, stride=2, padding=1))
        elif kd == G.OP_CONV:
            w, b = unpack_conv(blob, o)
            x = bufs[o["in_buf"]][:, o["in_choff"]:o["in_choff"] + o["cin"]]
            y = act(F.conv2d(x, torch.from_numpy(w), torch.from_numpy(b.copy()), stride=o["stride"], padding=o["ksize"] // 2), o["act"])
            y = y[:, :o["cout"]]
            if o["res_buf"] >= 0:
                y = y + bufs[o["res_buf"]][:, o["res_choff"]:o["res_choff"] + o["cout"]]
            bufs[o["out_buf"]][:, o["out_choff"]:o["out_choff"] + o["cout"]] = y
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
