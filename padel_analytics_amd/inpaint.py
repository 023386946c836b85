"""InpaintNet trajectory repair (0.52 M parameters over length-16 coordinate sequences).
The network runs on the device since round 4 (``InpaintNetDevice``: the engine's conv kernels over one-row images); its numpy
fp32 twin ``InpaintNetHost`` remains for CPU tests; mask generation, windows, blending and the temporal ensemble are host
bookkeeping in both cases.

Restates, from the reference's ball tracker:
* ``generate_inpaint_mask`` (``ball_tracker.py:100-136``): runs of invisible frames that are bounded by
  visible frames lower than ``th_h`` pixels from the top get mask 1 (with the reference's edge rules: a run
  starting at frame 0 only needs the closing frame; a run starting at frame 1 is never masked; a run reaching
  the end of the clip is never masked);
* the coordinate windows of ``BallTrajectoryDataset`` (``dataset.py:387-429,493-503``): length-L windows
  sliding by 1, x / w and y / h normalisation in fp32;
* ``InpaintNet`` (``models.py:101-130``): Conv1d(k=3, same) + LeakyReLU U-Net, sigmoid output;
* blend / threshold / temporal ensemble / ``predict`` (``ball_tracker.py:572-657``, ``predict.py:91-146``).
"""
from __future__ import annotations

import math

import numpy as np


def generate_inpaint_mask(y, vis, th_h: float) -> np.ndarray:
    y = np.asarray(y)
    vis = np.asarray(vis)
    n = len(vis)
    mask = np.zeros(n, dtype=y.dtype if y.dtype.kind == "f" else np.int64)
    i = 0
    while i < n:
        while i < n - 1 and vis[i] == 1:      # first invisible frame (stops at n-1 regardless)
            i += 1
        j = i
        while j < n - 1 and vis[j] == 0:      # first visible frame after the run (or n-1)
            j += 1
        if j == i:
            break
        if i == 0:
            if y[j] > th_h:
                mask[:j] = 1
        elif i > 1 and y[i - 1] > th_h and y[j] > th_h:
            mask[i:j] = 1
        i = j
    return mask


def ensemble_weight(seq_len: int) -> np.ndarray:
    w = np.ones(seq_len, np.float32)
    for i in range(math.ceil(seq_len / 2)):
        w[i] = i + 1
        w[seq_len - i - 1] = i + 1
    return w / w.sum(dtype=np.float32)


def temporal_ensemble(seq: np.ndarray, weight: np.ndarray) -> np.ndarray:
    """seq: (S, L, ...) per-window predictions (window s covers frames s..s+L-1) -> (S + L - 1, ...):
    frame g < L-1: mean of the available windows; L-1 <= g < S: sum_k weight[k] * seq[g-(L-1)+k][L-1-k];
    the L-1 tail frames: means of the remaining windows (ball_tracker.py:449-509 / :600-650).
    All frames at once; per frame the L terms are added in the order k = 0 .. L-1 (the loop of the reference)."""
    S, L = seq.shape[:2]
    T = S + L - 1
    zero = np.zeros((L - 1,) + seq.shape[1:], np.float32)
    buf = np.concatenate([zero, seq.astype(np.float32), zero], 0)          # row r <-> window r-(L-1)
    k = np.arange(L)
    rows = buf[np.arange(T)[:, None] + k[None, :], (L - 1 - k)[None, :]]    # (T, L, ...): window k of frame g (zeros where none)
    wb = weight.reshape((1, L) + (1,) * (seq.ndim - 2)).astype(np.float32)
    g = np.arange(T)
    mid = (g >= L - 1) & (g < S)
    terms = np.where(mid.reshape((T, 1) + (1,) * (seq.ndim - 2)), rows * wb, rows)
    acc = terms[:, 0].copy()
    for j in range(1, L):
        acc = acc + terms[:, j]
    # head: g + 1 windows, tail frame S - 1 + fi: L - fi windows  (S < L - 1: the head rule wins, like the loops did)
    div = np.ones(T, np.float32)
    head = g < min(L - 1, S)
    div[head] = (g[head] + 1).astype(np.float32)
    tail = g >= S
    div[tail] = (L - (g[tail] - S + 1)).astype(np.float32)
    out = acc.copy()
    sel = head | tail
    out[sel] = acc[sel] / div[sel].reshape((-1,) + (1,) * (seq.ndim - 2))
    return out


class InpaintNetHost:
    """numpy fp32 forward of the reference's InpaintNet (state_dict keys of models.py)."""

    def __init__(self, state_dict):
        self.sd = {k: np.asarray(v, np.float32) for k, v in state_dict.items()}

    def _conv(self, x, name, act=True):
        w, b = self.sd[f"{name}.weight"], self.sd[f"{name}.bias"]
        xp = np.pad(x, ((0, 0), (0, 0), (1, 1)))
        L = x.shape[2]
        cols = np.stack([xp[:, :, t:t + L] for t in range(3)], 2)            # (N, Ci, 3, L)
        y = np.einsum("oik,nikl->nol", w, cols, optimize=True).astype(np.float32) + b[None, :, None]
        return np.where(y >= 0, y, np.float32(0.01) * y).astype(np.float32) if act else y

    def forward(self, coor: np.ndarray, mask: np.ndarray) -> np.ndarray:
        x = np.concatenate([coor, mask], 2).astype(np.float32).transpose(0, 2, 1)   # (N, 3, L)
        x1 = self._conv(x, "down_1.conv")
        x2 = self._conv(x1, "down_2.conv")
        x3 = self._conv(x2, "down_3.conv")
        x = self._conv(self._conv(x3, "buttleneck.conv_1.conv"), "buttleneck.conv_2.conv")
        x = self._conv(np.concatenate([x, x3], 1), "up_1.conv")
        x = self._conv(np.concatenate([x, x2], 1), "up_2.conv")
        x = self._conv(np.concatenate([x, x1], 1), "up_3.conv")
        x = self._conv(x, "predictor", act=False)
        x = (1.0 / (1.0 + np.exp(-x.astype(np.float32)))).astype(np.float32)
        return x.transpose(0, 2, 1)


class InpaintNetDevice:
    """The same forward on the HIP engine: ``graph.build_inpaintnet`` — eight Conv1d + LeakyReLU
    layers and the sigmoid predictor as 3x3 convolutions over one-row images, in the engine's fp32-equivalent arithmetic.
    Same ``forward(coor, mask)`` contract as ``InpaintNetHost`` (which stays as its CPU twin for the tests); windows are
    processed ``max_windows`` at a time."""

    def __init__(self, state_dict, engine=None, max_windows: int = 4096):
        from . import engine as E, graph as G
        self._E, self._G = E, G
        self.sd = {k: np.asarray(v, np.float32) for k, v in state_dict.items()}
        self.engine = engine
        self.max_windows = int(max_windows)
        self.mode = E.fp32_mode()
        self._model = None

    def _ensure(self):
        if self._model is None:
            E, G = self._E, self._G
            self._model = E.Model(self.engine or E.default_engine(), G.build_inpaintnet(self.sd, dtype=E.graph_dtype(self.mode)))
            self._model.set_max_batch(self.max_windows)
        return self._model

    def forward(self, coor: np.ndarray, mask: np.ndarray) -> np.ndarray:
        n, L = coor.shape[:2]
        x = np.zeros((n, 1, L, 16), np.float32)
        x[:, 0, :, :2] = coor
        x[:, 0, :, 2] = mask[..., 0]
        out = np.empty((n, L, 2), np.float32)
        for lo in range(0, n, self.max_windows):
            m = self._ensure()
            y = m.tracknet_infer(x[lo:lo + self.max_windows])
            if self.mode == "h2" and m.take_overflow():          # cannot happen on normalised coordinates; kept for symmetry
                self.close()
                self.mode = "bx3"
                y = self._ensure().tracknet_infer(x[lo:lo + self.max_windows])
            out[lo:lo + self.max_windows] = y[:, 0, :, :2]
        return out

    def close(self) -> None:
        if self._model is not None:
            self._model.close()
            self._model = None


def inpaint_trajectory(xs, ys, vs, img_w: int, img_h: int, net: InpaintNetHost, seq_len: int,
                       width: int = 512, height: int = 288):
    """TrackNet coordinates (source pixels, ints) -> repaired (x, y, visibility) per frame.
    Frames the windows cannot cover (clips shorter than seq_len) come back as None."""
    T = len(xs)
    if T < seq_len:
        return [None] * T
    coor_th = 50.0 / math.sqrt(height ** 2 + width ** 2)
    mask = generate_inpaint_mask(np.asarray(ys), np.asarray(vs), th_h=img_h * 0.05).astype(np.float32)
    cx = np.asarray(xs, np.float32) / np.float32(img_w)
    cy = np.asarray(ys, np.float32) / np.float32(img_h)
    S = T - seq_len + 1
    idx = np.arange(S)[:, None] + np.arange(seq_len)[None, :]
    coor = np.stack([cx[idx], cy[idx]], 2).astype(np.float32)                # (S, L, 2)
    m = mask[idx][..., None].astype(np.float32)                              # (S, L, 1)
    out = net.forward(coor, m)
    out = out * m + coor * (1 - m)
    low = (out[:, :, 0] < coor_th) & (out[:, :, 1] < coor_th)
    out[low] = 0.0
    ens = temporal_ensemble(out, ensemble_weight(seq_len))                   # (T, 2)
    low = (ens[:, 0] < coor_th) & (ens[:, 1] < coor_th)
    ens[low] = 0.0
    w_scaler, h_scaler = img_w / width, img_h / height
    res = []
    for g in range(T):
        px, py = int(ens[g, 0] * width * w_scaler), int(ens[g, 1] * height * h_scaler)
        res.append((px, py, 0 if (px == 0 and py == 0) else 1))
    return res
