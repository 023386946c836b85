"""ORACLE — test infrastructure only.  torch-CPU restatement of the reference's in-tree networks
``/root/reference/trackers/ball_tracker/models.py``: ``Conv2DBlock`` :5-17 (3x3 'same' conv without bias +
BatchNorm2d(eps 1e-5) + ReLU), ``TrackNet`` :45-74 (U-Net 27->8 with three 2x2 max-pools, nearest x2
upsampling, [upsampled | skip] concats, 1x1 predictor + sigmoid) and ``InpaintNet`` :101-130 (Conv1d U-Net).

PINNED: tests/test_tracknet_ref.py checks it against golden outputs produced by the reference's own
``models.py`` (tests/golden/make_tracknet_golden.py, run in this container) and against the known answers
11 341 000 / 520 610 parameters, 104 state-dict tensors."""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
DOUBLE = {"down_block_1": (27, 64), "down_block_2": (64, 128), "up_block_2": (384, 128), "up_block_3": (192, 64)}
TRIPLE = {"down_block_3": (128, 256), "bottleneck": (256, 512), "up_block_1": (768, 256)}
ORDER = ["down_block_1", "down_block_2", "down_block_3", "bottleneck", "up_block_1", "up_block_2", "up_block_3"]


def tracknet_spec(in_dim=27, out_dim=8):
    spec = OrderedDict()
    for blk in ORDER:
        cin, cout = DOUBLE.get(blk) or TRIPLE[blk]
        if blk == "down_block_1":
            cin = in_dim
        for i in range(2 if blk in DOUBLE else 3):
            p = f"{blk}.conv_{i + 1}"
            spec[f"{p}.conv.weight"] = (cout, cin if i == 0 else cout, 3, 3)
            for nm in ("weight", "bias", "running_mean", "running_var"):
                spec[f"{p}.bn.{nm}"] = (cout,)
            spec[f"{p}.bn.num_batches_tracked"] = ()
    spec["predictor.weight"] = (out_dim, 64, 1, 1)
    spec["predictor.bias"] = (out_dim,)
    return spec


def synth_tracknet_state_dict(seed=0, in_dim=27, out_dim=8):
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    for k, shp in tracknet_spec(in_dim, out_dim).items():
        if k.endswith("num_batches_tracked"):
            sd[k] = np.array(0, np.int64)
        elif k.endswith("conv.weight") or k == "predictor.weight":
            sd[k] = rng.normal(0, math.sqrt(2.0 / (shp[1] * shp[2] * shp[3])), shp).astype(np.float32)
        elif k.endswith("bn.weight"):
            sd[k] = rng.uniform(0.8, 1.2, shp).astype(np.float32)
        elif k.endswith("running_var"):
            sd[k] = rng.uniform(0.8, 1.25, shp).astype(np.float32)
        elif k == "predictor.bias":
            sd[k] = rng.uniform(-3.0, -1.0, shp).astype(np.float32)
        else:
            sd[k] = rng.uniform(-0.1, 0.1, shp).astype(np.float32)
    return sd


INPAINT = [("down_1", 3, 32), ("down_2", 32, 64), ("down_3", 64, 128), ("buttleneck.conv_1", 128, 256),
           ("buttleneck.conv_2", 256, 256), ("up_1", 384, 128), ("up_2", 192, 64), ("up_3", 96, 32)]


def synth_inpaintnet_state_dict(seed=0):
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    for name, cin, cout in INPAINT:
        sd[f"{name}.conv.weight"] = rng.normal(0, math.sqrt(2.0 / (cin * 3)), (cout, cin, 3)).astype(np.float32)
        sd[f"{name}.conv.bias"] = rng.uniform(-0.1, 0.1, (cout,)).astype(np.float32)
    sd["predictor.weight"] = rng.normal(0, math.sqrt(2.0 / 96), (2, 32, 3)).astype(np.float32)
    sd["predictor.bias"] = rng.uniform(-0.1, 0.1, (2,)).astype(np.float32)
    return sd


def _t(sd, k, dtype=torch.float32):
    v = sd[k]
    return (v if torch.is_tensor(v) else torch.from_numpy(np.asarray(v))).to(dtype)


class TrackNetRef:
    def __init__(self, sd, dtype=torch.float32):
        self.sd, self.dtype = sd, dtype

    def _block(self, x, p):
        # conv -> BN (inference) -> ReLU, unfused like models.py:13-17
        d = self.dtype
        y = F.conv2d(x, _t(self.sd, f"{p}.conv.weight", d), None, padding=1)
        y = F.batch_norm(y, _t(self.sd, f"{p}.bn.running_mean", d), _t(self.sd, f"{p}.bn.running_var", d),
                         _t(self.sd, f"{p}.bn.weight", d), _t(self.sd, f"{p}.bn.bias", d), False, 0.0, BN_EPS)
        return F.relu(y)

    def _multi(self, x, blk):
        for i in range(2 if blk in DOUBLE else 3):
            x = self._block(x, f"{blk}.conv_{i + 1}")
        return x

    @torch.no_grad()
    def forward(self, x):
        x = x.to(self.dtype)
        up = lambda t: F.interpolate(t, scale_factor=2.0, mode="nearest")
        x1 = self._multi(x, "down_block_1")
        x2 = self._multi(F.max_pool2d(x1, 2, 2), "down_block_2")
        x3 = self._multi(F.max_pool2d(x2, 2, 2), "down_block_3")
        x = self._multi(F.max_pool2d(x3, 2, 2), "bottleneck")
        x = self._multi(torch.cat([up(x), x3], 1), "up_block_1")
        x = self._multi(torch.cat([up(x), x2], 1), "up_block_2")
        x = self._multi(torch.cat([up(x), x1], 1), "up_block_3")
        x = F.conv2d(x, _t(self.sd, "predictor.weight", self.dtype), _t(self.sd, "predictor.bias", self.dtype))
        return torch.sigmoid(x)


class InpaintNetRef:
    def __init__(self, sd):
        self.sd = sd

    def _c(self, x, p):
        return F.leaky_relu(F.conv1d(x, _t(self.sd, f"{p}.conv.weight"), _t(self.sd, f"{p}.conv.bias"), padding=1))

    @torch.no_grad()
    def forward(self, x, m):
        x = torch.cat([x, m], 2).permute(0, 2, 1)
        x1 = self._c(x, "down_1")
        x2 = self._c(x1, "down_2")
        x3 = self._c(x2, "down_3")
        x = self._c(self._c(x3, "buttleneck.conv_1"), "buttleneck.conv_2")
        x = self._c(torch.cat([x, x3], 1), "up_1")
        x = self._c(torch.cat([x, x2], 1), "up_2")
        x = self._c(torch.cat([x, x1], 1), "up_3")
        x = torch.sigmoid(F.conv1d(x, _t(self.sd, "predictor.weight"), _t(self.sd, "predictor.bias"), padding=1))
        return x.permute(0, 2, 1)
