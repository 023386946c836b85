"""ORACLE — test infrastructure only (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).

CPU restatement (torch fp32 + numpy) of what ``ultralytics.YOLO.predict`` does for the
reference's two YOLOv8 call sites:

* detect: ``/root/reference/trackers/players_tracker/players_tracker.py:351-359``
* pose:   ``/root/reference/trackers/players_keypoints_tracker/players_keypoints_tracker.py:285-292``

PARITY UNPINNED: the arithmetic lives in third-party packages that are neither vendored in
the reference nor installed here (``ultralytics`` 8.3.x line, ``torchvision.ops.nms``,
``opencv-python`` — all un-pinned in ``requirements.txt:1-11``) and the reference holds no
tests or golden vectors for this boundary.  This file restates the
published algorithms; it is pinned only by known answers:
parameter counts / GFLOPs of the graph (tests/test_arch_known_answers.py) and, for the
Pillow resize used on the pose path, bit-exactness against Pillow itself
(oracle/preprocess_ref.py).

Nothing in the product package imports this module.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

REG_MAX = 16
BN_EPS = 1e-3
MAX_WH = 7680
MAX_NMS = 30000


# ------------------------------------------------------------------------------------ graph

def _t(sd, k):
    v = sd[k]
    return v if torch.is_tensor(v) else torch.from_numpy(np.asarray(v))


def fuse_conv_bn(sd, prefix):
    """ultralytics.utils.torch_utils.fuse_conv_and_bn: W' = diag(g/sqrt(v+eps)) W,
    b' = beta - g*mu/sqrt(v+eps)."""
    w = _t(sd, f"{prefix}.conv.weight").float()
    g = _t(sd, f"{prefix}.bn.weight").float()
    b = _t(sd, f"{prefix}.bn.bias").float()
    mu = _t(sd, f"{prefix}.bn.running_mean").float()
    var = _t(sd, f"{prefix}.bn.running_var").float()
    scale = g.div(torch.sqrt(BN_EPS + var))
    wf = (w.view(w.shape[0], -1) * scale[:, None]).view_as(w)
    bf = b - g.mul(mu).div(torch.sqrt(var + BN_EPS))
    return wf, bf


class YoloV8Ref:
    """Fused-BN YOLOv8 detect/pose forward on CPU fp32 (Conv = conv + BN + SiLU)."""

    def __init__(self, state_dict, nc: int, kpt_shape: Optional[tuple] = None, dtype=torch.float32):
        """dtype=torch.float64 evaluates the SAME fp32-folded weights in double precision: the
        rounding-free value every fp32 implementation (this oracle included) approximates."""
        self.sd = state_dict
        self.dtype = dtype
        self.nc = nc
        self.kpt_shape = tuple(kpt_shape) if kpt_shape else None
        self._fused = {}
        self.n_blocks = {}
        for i in (2, 4, 6, 8, 12, 15, 18, 21):
            n = 0
            while f"model.{i}.m.{n}.cv1.conv.weight" in state_dict:
                n += 1
            self.n_blocks[i] = n

    def _conv(self, x, prefix, k, s):
        if prefix not in self._fused:
            w, b = fuse_conv_bn(self.sd, prefix)
            self._fused[prefix] = (w.to(self.dtype), b.to(self.dtype))
        w, b = self._fused[prefix]
        return F.silu(F.conv2d(x, w, b, stride=s, padding=k // 2))

    def _c2f(self, x, i, shortcut):
        p = f"model.{i}"
        y = list(self._conv(x, f"{p}.cv1", 1, 1).chunk(2, 1))
        for j in range(self.n_blocks[i]):
            z = self._conv(self._conv(y[-1], f"{p}.m.{j}.cv1", 3, 1), f"{p}.m.{j}.cv2", 3, 1)
            y.append(y[-1] + z if shortcut else z)
        return self._conv(torch.cat(y, 1), f"{p}.cv2", 1, 1)

    def _sppf(self, x):
        y = [self._conv(x, "model.9.cv1", 1, 1)]
        for _ in range(3):
            y.append(F.max_pool2d(y[-1], 5, 1, 2))
        return self._conv(torch.cat(y, 1), "model.9.cv2", 1, 1)

    def features(self, x):
        x = self._conv(x, "model.0", 3, 2)
        x = self._conv(x, "model.1", 3, 2)
        x = self._c2f(x, 2, True)
        x = self._conv(x, "model.3", 3, 2)
        x4 = self._c2f(x, 4, True)
        x = self._conv(x4, "model.5", 3, 2)
        x6 = self._c2f(x, 6, True)
        x = self._conv(x6, "model.7", 3, 2)
        x = self._c2f(x, 8, True)
        x9 = self._sppf(x)
        x = torch.cat([F.interpolate(x9, scale_factor=2.0, mode="nearest"), x6], 1)
        x12 = self._c2f(x, 12, False)
        x = torch.cat([F.interpolate(x12, scale_factor=2.0, mode="nearest"), x4], 1)
        x15 = self._c2f(x, 15, False)
        x = torch.cat([self._conv(x15, "model.16", 3, 2), x12], 1)
        x18 = self._c2f(x, 18, False)
        x = torch.cat([self._conv(x18, "model.19", 3, 2), x9], 1)
        x21 = self._c2f(x, 21, False)
        return [x15, x18, x21]

    def _branch(self, x, br, l):
        p = f"model.22.{br}.{l}"
        x = self._conv(x, f"{p}.0", 3, 1)
        x = self._conv(x, f"{p}.1", 3, 1)
        return F.conv2d(x, _t(self.sd, f"{p}.2.weight").float().to(self.dtype),
                        _t(self.sd, f"{p}.2.bias").float().to(self.dtype))

    def head_raw(self, feats):
        """Per level raw head maps: (B, 64+nc, H, W) and, for pose, (B, nk, H, W)."""
        det, kpt = [], []
        for l, f in enumerate(feats):
            det.append(torch.cat((self._branch(f, "cv2", l), self._branch(f, "cv3", l)), 1))
            if self.kpt_shape:
                kpt.append(self._branch(f, "cv4", l))
        return det, kpt

    @staticmethod
    def make_anchors(feats, strides=(8, 16, 32), offset=0.5):
        pts, st = [], []
        dt = feats[0].dtype
        for f, s in zip(feats, strides):
            h, w = f.shape[2:]
            sx = torch.arange(w, dtype=dt) + offset
            sy = torch.arange(h, dtype=dt) + offset
            sy, sx = torch.meshgrid(sy, sx, indexing="ij")
            pts.append(torch.stack((sx, sy), -1).view(-1, 2))
            st.append(torch.full((h * w, 1), float(s), dtype=dt))
        return torch.cat(pts).transpose(0, 1), torch.cat(st).transpose(0, 1)   # (2,A), (1,A)

    def decode(self, det, kpt):
        """Detect/Pose inference branch: (B, 4+nc[+nk], A)."""
        bs = det[0].shape[0]
        no = self.nc + 4 * REG_MAX
        x_cat = torch.cat([d.reshape(bs, no, -1) for d in det], 2)
        anchors, strides = self.make_anchors(det)
        box, cls = x_cat.split((4 * REG_MAX, self.nc), 1)
        a = box.shape[-1]
        prob = box.view(bs, 4, REG_MAX, a).transpose(2, 1).softmax(1)           # (B,16,4,A)
        proj = torch.arange(REG_MAX, dtype=box.dtype).view(1, REG_MAX, 1, 1)
        dist = (prob * proj).sum(1)                                              # (B,4,A)
        lt, rb = dist.chunk(2, 1)
        x1y1 = anchors.unsqueeze(0) - lt
        x2y2 = anchors.unsqueeze(0) + rb
        dbox = torch.cat(((x1y1 + x2y2) / 2, x2y2 - x1y1), 1) * strides
        y = torch.cat((dbox, cls.sigmoid()), 1)
        if not self.kpt_shape:
            return y
        nk = self.kpt_shape[0] * self.kpt_shape[1]
        ndim = self.kpt_shape[1]
        k = torch.cat([t.reshape(bs, nk, -1) for t in kpt], -1).clone()
        if ndim == 3:
            k[:, 2::3] = k[:, 2::3].sigmoid()
        k[:, 0::ndim] = (k[:, 0::ndim] * 2.0 + (anchors[0] - 0.5)) * strides
        k[:, 1::ndim] = (k[:, 1::ndim] * 2.0 + (anchors[1] - 0.5)) * strides
        return torch.cat([y, k], 1)

    @torch.no_grad()
    def forward(self, x):
        det, kpt = self.head_raw(self.features(x.to(self.dtype)))
        return self.decode(det, kpt)


# --------------------------------------------------------------------------- preprocessing

def cv2_resize_linear_u8(img: np.ndarray, dst_w: int, dst_h: int) -> np.ndarray:
    """Restatement of ``cv2.resize(img, (dst_w, dst_h), interpolation=cv2.INTER_LINEAR)`` on u8.

    * exact 2x2 decimation takes OpenCV's area-fast path: ``(a+b+c+d+2)>>2``;
    * otherwise 11-bit fixed-point separable bilinear (INTER_RESIZE_COEF_BITS = 11):
      horizontal pass in int32, vertical pass
      ``(((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2``.
    """
    sh, sw = img.shape[:2]
    if (sw, sh) == (dst_w, dst_h):
        return img.copy()
    if sw == 2 * dst_w and sh == 2 * dst_h:
        a = img.astype(np.int32)
        s = a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2
        return (s >> 2).astype(np.uint8)

    def coeffs(src, dst):
        scale = src / dst
        idx = np.empty(dst, np.int64)
        c0 = np.empty(dst, np.int64)
        c1 = np.empty(dst, np.int64)
        for d in range(dst):
            f = (d + 0.5) * scale - 0.5
            s = int(math.floor(f))
            f -= s
            if s < 0:
                s, f = 0, 0.0
            if s >= src - 1:
                s, f = src - 1, 0.0
            idx[d] = s
            # saturate_cast<short>(x * 2048) with round-half-to-even (cvRound)
            c0[d] = int(np.rint((1.0 - f) * 2048.0))
            c1[d] = int(np.rint(f * 2048.0))
        return idx, c0, c1

    xi, xa0, xa1 = coeffs(sw, dst_w)
    yi, yb0, yb1 = coeffs(sh, dst_h)
    a = img.astype(np.int64)
    x1 = np.minimum(xi + 1, sw - 1)
    rows = a[:, xi] * xa0[None, :, None] + a[:, x1] * xa1[None, :, None]          # (sh, dw, C)
    y1 = np.minimum(yi + 1, sh - 1)
    s0 = rows[yi] >> 4
    s1 = rows[y1] >> 4
    out = (((yb0[:, None, None] * s0) >> 16) + ((yb1[:, None, None] * s1) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def letterbox_geometry(h0: int, w0: int, imgsz: int = 640, auto: bool = True, stride: int = 32):
    """ultralytics LetterBox arithmetic.  Returns (new_w, new_h, top, bottom, left, right)."""
    r = min(imgsz / h0, imgsz / w0)
    new_w, new_h = int(round(w0 * r)), int(round(h0 * r))
    dw, dh = imgsz - new_w, imgsz - new_h
    if auto:
        dw, dh = dw % stride, dh % stride
    dw /= 2
    dh /= 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return new_w, new_h, top, bottom, left, right


def letterbox_u8(img: np.ndarray, imgsz: int = 640, auto: bool = True, stride: int = 32) -> np.ndarray:
    h0, w0 = img.shape[:2]
    nw, nh, top, bottom, left, right = letterbox_geometry(h0, w0, imgsz, auto, stride)
    if (w0, h0) != (nw, nh):
        img = cv2_resize_linear_u8(img, nw, nh)
    out = np.full((nh + top + bottom, nw + left + right, 3), 114, np.uint8)
    out[top:top + nh, left:left + nw] = img
    return out


def preprocess(sources: Sequence[np.ndarray], imgsz: int) -> torch.Tensor:
    """LoadPilAndNumpy + BasePredictor.preprocess: ndarrays are BGR by convention;
    letterbox (auto iff all shapes equal) -> stack -> [..., ::-1] -> BCHW -> fp32 / 255."""
    same = len({s.shape for s in sources}) == 1
    ims = [letterbox_u8(s, imgsz, auto=same) for s in sources]
    im = np.stack(ims)[..., ::-1].transpose(0, 3, 1, 2)
    im = torch.from_numpy(np.ascontiguousarray(im)).float()
    im /= 255
    return im


# --------------------------------------------------------------------------- post-processing

def nms_torchvision(boxes: torch.Tensor, scores: torch.Tensor, thr: float, flip=None) -> torch.Tensor:
    """torchvision.ops.nms CPU kernel: stable descending sort, suppress when IoU > thr.
    ``flip``: optional set of (i, j) candidate-index pairs whose suppress decision is inverted (used by
    the parity harness to prove that a count mismatch is a threshold-adjacent IoU decision)."""
    n = boxes.shape[0]
    if n == 0:
        return torch.empty((0,), dtype=torch.int64)
    b = boxes.numpy()
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    order = torch.sort(scores, stable=True, descending=True)[1].numpy()
    suppressed = np.zeros(n, bool)
    keep = []
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        xx1 = np.maximum(x1[i], x1[rest])
        yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest])
        yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(b.dtype.type(0), xx2 - xx1)
        h = np.maximum(b.dtype.type(0), yy2 - yy1)
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (areas[i] + areas[rest] - inter)
        sup = ovr > b.dtype.type(thr)
        if flip:
            for (fi, fj) in flip:
                if fi == i:
                    sup[rest == fj] = ~sup[rest == fj]
        suppressed[rest[sup]] = True
    return torch.as_tensor(np.asarray(keep, dtype=np.int64))


def non_max_suppression(prediction: torch.Tensor, conf_thres: float, iou_thres: float,
                        classes=None, max_det: int = 300, nc: int = 0, return_candidates: bool = False):
    """ultralytics.utils.ops.non_max_suppression (multi_label=False, agnostic=False, no time limit).
    With ``return_candidates`` also returns, per image, the pre-NMS candidate rows."""
    bs = prediction.shape[0]
    nc = nc or (prediction.shape[1] - 4)
    nm = prediction.shape[1] - nc - 4
    mi = 4 + nc
    xc = prediction[:, 4:mi].amax(1) > conf_thres
    prediction = prediction.transpose(-1, -2).clone()
    xy, wh = prediction[..., :2].clone(), prediction[..., 2:4] / 2
    prediction[..., :2] = xy - wh
    prediction[..., 2:4] = xy + wh
    cls_t = None if classes is None else torch.tensor(classes, dtype=prediction.dtype)
    out = [torch.zeros((0, 6 + nm), dtype=prediction.dtype)] * bs
    cands = [torch.zeros((0, 6 + nm), dtype=prediction.dtype)] * bs
    for xi, x in enumerate(prediction):
        x = x[xc[xi]]
        if not x.shape[0]:
            continue
        box, cls, mask = x.split((4, nc, nm), 1)
        conf, j = cls.max(1, keepdim=True)
        x = torch.cat((box, conf, j.to(box.dtype), mask), 1)[conf.view(-1) > conf_thres]
        if cls_t is not None:
            x = x[(x[:, 5:6] == cls_t).any(1)]
        n = x.shape[0]
        if not n:
            continue
        if n > MAX_NMS:
            x = x[x[:, 4].argsort(descending=True)[:MAX_NMS]]
        c = x[:, 5:6] * MAX_WH
        i = nms_torchvision(x[:, :4] + c, x[:, 4], iou_thres)[:max_det]
        out[xi] = x[i]
        cands[xi] = x
    return (out, cands) if return_candidates else out


def scale_boxes(net_hw, boxes: torch.Tensor, orig_hw) -> torch.Tensor:
    gain = min(net_hw[0] / orig_hw[0], net_hw[1] / orig_hw[1])
    pad = (round((net_hw[1] - orig_hw[1] * gain) / 2 - 0.1), round((net_hw[0] - orig_hw[0] * gain) / 2 - 0.1))
    boxes = boxes.clone()
    boxes[..., 0] -= pad[0]
    boxes[..., 1] -= pad[1]
    boxes[..., 2] -= pad[0]
    boxes[..., 3] -= pad[1]
    boxes[..., :4] /= gain
    boxes[..., 0].clamp_(0, orig_hw[1])
    boxes[..., 1].clamp_(0, orig_hw[0])
    boxes[..., 2].clamp_(0, orig_hw[1])
    boxes[..., 3].clamp_(0, orig_hw[0])
    return boxes


def scale_coords(net_hw, coords: torch.Tensor, orig_hw) -> torch.Tensor:
    gain = min(net_hw[0] / orig_hw[0], net_hw[1] / orig_hw[1])
    pad = ((net_hw[1] - orig_hw[1] * gain) / 2, (net_hw[0] - orig_hw[0] * gain) / 2)
    coords = coords.clone()
    coords[..., 0] -= pad[0]
    coords[..., 1] -= pad[1]
    coords[..., 0] /= gain
    coords[..., 1] /= gain
    coords[..., 0].clamp_(0, orig_hw[1])
    coords[..., 1].clamp_(0, orig_hw[0])
    return coords


def keypoints_xy(kpts: torch.Tensor) -> torch.Tensor:
    """ultralytics.engine.results.Keypoints: zero points with visibility < 0.5, return xy."""
    k = kpts.clone()
    if k.shape[-1] == 3:
        m = k[..., 2] < 0.5
        k[..., :2][m] = 0
    return k[..., :2]


# --------------------------------------------------------------------------- predict()

@torch.no_grad()
def predict(model: YoloV8Ref, sources: Sequence[np.ndarray], conf: float, iou: float, imgsz: int,
            classes=None, max_det: int = 300, heads=None):
    """-> list of dicts {boxes (n,6) [x1,y1,x2,y2,conf,cls], kpts (n,K,ndim) | None, margins}.

    ``margins`` carries the smallest |score - conf| over all anchors of the image and is used by
    the parity harness to detect threshold-adjacent decisions.  ``heads``: the (det, kpt) raw head maps of
    ``model.head_raw(model.features(preprocess(sources, imgsz)))`` if the caller already holds them (the tests look at both)."""
    im = preprocess(sources, imgsz)
    pred = model.forward(im) if heads is None else model.decode(*heads)
    dets, cands = non_max_suppression(pred, conf, iou, classes, max_det, nc=model.nc, return_candidates=True)
    res = []
    for i, d in enumerate(dets):
        h0, w0 = sources[i].shape[:2]
        d = d.clone()
        d[:, :4] = scale_boxes(im.shape[2:], d[:, :4], (h0, w0))
        k = None
        if model.kpt_shape:
            k = d[:, 6:].view(len(d), *model.kpt_shape) if len(d) else d[:, 6:].view(0, *model.kpt_shape)
            k = scale_coords(im.shape[2:], k, (h0, w0))
        sc = pred[i, 4:4 + model.nc].amax(0)
        res.append({"boxes": d[:, :6].numpy(), "kpts": None if k is None else k.numpy(),
                    "conf_margin": float((sc - conf).abs().min()), "cands": cands[i].numpy(),
                    "net_hw": tuple(im.shape[2:]), "orig_hw": (h0, w0)})
    return res
