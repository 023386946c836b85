"""ORACLE — test infrastructure only.  CPU restatement of the reference's ball path around TrackNet:

* ``trackers/ball_tracker/iterable.py``: background median of the first ``median_range`` RGB frames
  (:59-81: ``np.median`` -> ``astype('uint8')`` truncation -> Pillow bicubic resize to 512x288), 8-frame
  windows sliding by 1 (:153-165), per-frame Pillow resize + CHW stacking + ``/255`` in float64 (:167-199);
* ``trackers/ball_tracker/ball_tracker.py:421-523``: temporal ensemble of the 8 overlapping window outputs
  (``get_ensemble_weight`` :68-97 -> [1,2,3,4,4,3,2,1]/20; incomplete head: plain mean of the available
  windows; tail: plain means);
* ``trackers/ball_tracker/predict.py``: ``predict_modified`` :149-221 / ``predict_location`` :7-39 / ``to_img``
  :42-54: threshold 0.5 -> external contours -> bounding rects -> first rect of maximal w*h ->
  ``int(x+w/2), int(y+h/2)`` -> ``int(c*scaler)``; visibility 0 iff both coordinates are 0.

The window stream is the *intended* contiguous one (the reference drops the 7
windows that straddle the median boundary for clips longer than ``median_range``).

PINNED: TrackNet itself against the reference's own models.py (tracknet_ref.py goldens); the Pillow resize
bit-exactly against Pillow; ``ensemble`` / ``ensemble_weight`` / ``generate_inpaint_mask_ref`` bit-exactly against
outputs of the reference's own ``ball_tracker.py`` (its real ``predict_frames`` loop, ``get_ensemble_weight``,
``generate_inpaint_mask``) generated in the build container (tests/golden/make_ball_golden.py -> ball_golden.npz,
objects_golden.json; tests/test_ball_ref.py, tests/test_inpaint.py).  UNPINNED: ``cv2.findContours`` ordering (ties between equal-area rectangles) —
restated as 8-connected components in reverse raster-discovery order; cv2 is not installable here.
"""
from __future__ import annotations

import math

import numpy as np
import torch
from PIL import Image
from scipy import ndimage

HEIGHT, WIDTH, SEQ = 288, 512, 8


def ensemble_weight(seq_len: int = SEQ) -> np.ndarray:
    w = np.ones(seq_len, np.float32)
    for i in range(math.ceil(seq_len / 2)):
        w[i] = i + 1
        w[seq_len - i - 1] = i + 1
    return (torch.from_numpy(w) / torch.from_numpy(w).sum()).numpy()


def median_background(frames_bgr) -> np.ndarray:
    """-> (3, 288, 512) uint8 (iterable.py:59-81)."""
    rgb = np.array([f[..., ::-1] for f in frames_bgr])
    med = np.median(rgb, 0).astype("uint8")
    return np.moveaxis(np.array(Image.fromarray(med).resize((WIDTH, HEIGHT))), -1, 0)


def resize_frame(frame_bgr: np.ndarray) -> np.ndarray:
    """BGR frame -> (3, 288, 512) uint8 RGB, Pillow bicubic (iterable.py:160,188-189)."""
    return np.moveaxis(np.array(Image.fromarray(np.ascontiguousarray(frame_bgr[..., ::-1])).resize((WIDTH, HEIGHT))), -1, 0)


def window_input(median_chw: np.ndarray, frames_chw) -> np.ndarray:
    """(27, 288, 512) float32 exactly as ``x.float()`` of the float64 ``frames /= 255.`` (ball_tracker.py:440)."""
    x = np.concatenate([median_chw.astype(np.float64)] + [f.astype(np.float64) for f in frames_chw], 0)
    x /= 255.0
    return x.astype(np.float32)


def ensemble(y: np.ndarray) -> np.ndarray:
    """y: (Nw, 8, H, W) fp32 window outputs -> (Nw + 7, H, W) fp32 per-frame heat maps.
    Literal index algebra of ball_tracker.py:421-509 (buffer rows = windows, slot = 7 - k)."""
    nw = y.shape[0]
    w = torch.from_numpy(ensemble_weight())
    yt = torch.from_numpy(y)
    zero = torch.zeros((7,) + tuple(y.shape[1:]), dtype=torch.float32)
    buf = torch.cat([zero, yt, zero], 0)             # row r <-> window r - 7
    si = torch.arange(8)
    fi = torch.arange(7, -1, -1)
    out = []
    for g in range(nw):                               # frame g = first frame of window g
        rows = buf[si + g, fi]
        out.append(rows.sum(0) / (g + 1) if g < 7 else (rows * w[:, None, None]).sum(0))
    for frame_i in range(1, 8):                       # the 7 tail frames
        rows = buf[si + (nw - 1) + frame_i, fi]
        out.append(rows.sum(0) / (8 - frame_i))
    return torch.stack(out).numpy()


def predict_location(mask_u8: np.ndarray):
    """Largest bounding rectangle among the 8-connected foreground components (predict.py:7-39)."""
    if mask_u8.max() == 0:
        return 0, 0, 0, 0
    lab, n = ndimage.label(mask_u8 > 0, structure=np.ones((3, 3)))
    rects = []
    for sl in ndimage.find_objects(lab):
        rects.append((sl[1].start, sl[0].start, sl[1].stop - sl[1].start, sl[0].stop - sl[0].start))
    rects = rects[::-1]                               # cv2 returns the last-discovered contour first
    best = 0
    for i in range(1, len(rects)):
        if rects[i][2] * rects[i][3] > rects[best][2] * rects[best][3]:
            best = i
    return rects[best]


def decode_heat(heat: np.ndarray, img_scaler, threshold: float = 0.5):
    """(T, H, W) fp32 -> lists x, y, visibility (predict_modified)."""
    xs, ys, vs = [], [], []
    for h in heat:
        m = ((h > threshold) * 255).astype("uint8")
        x, y, w, hh = predict_location(m)
        cx, cy = int(x + w / 2), int(y + hh / 2)
        cx, cy = int(cx * img_scaler[0]), int(cy * img_scaler[1])
        xs.append(cx); ys.append(cy); vs.append(0 if (cx == 0 and cy == 0) else 1)
    return xs, ys, vs


def track(frames_bgr, tracknet, median_chw=None, batch: int = 4):
    """Whole-clip oracle: frames -> (x, y, visibility) per frame + the per-frame heat maps.
    ``tracknet``: callable (N,27,288,512) fp32 tensor -> (N,8,288,512)."""
    frames_bgr = list(frames_bgr)
    T = len(frames_bgr)
    if median_chw is None:
        median_chw = median_background(frames_bgr)
    small = [resize_frame(f) for f in frames_bgr]
    ys = []
    for g0 in range(0, T - 7, batch):
        xb = np.stack([window_input(median_chw, small[g:g + 8]) for g in range(g0, min(g0 + batch, T - 7))])
        ys.append(tracknet(torch.from_numpy(xb)).numpy())
    heat = ensemble(np.concatenate(ys))
    h0, w0 = frames_bgr[0].shape[:2]
    x, y, v = decode_heat(heat, (w0 / WIDTH, h0 / HEIGHT))
    return x, y, v, heat


# ------------------------------------------------------------------------------------------------
# InpaintNet stage (ball_tracker.py:100-136, :525-673; dataset.py:387-429,493-503; predict.py:91-146)

def generate_inpaint_mask_ref(y, vis_pred, th_h: float):
    """Transcription of the control flow of ball_tracker.py:112-136."""
    y = np.array(y)
    vis_pred = np.array(vis_pred)
    inpaint_mask = np.zeros_like(y)
    i = j = 0
    while j < len(vis_pred):
        while i < len(vis_pred) - 1 and vis_pred[i] == 1:
            i += 1
        j = i
        while j < len(vis_pred) - 1 and vis_pred[j] == 0:
            j += 1
        if j == i:
            break
        elif i == 0 and y[j] > th_h:
            inpaint_mask[:j] = 1
        elif (i > 1 and y[i - 1] > th_h) and (j < len(vis_pred) and y[j] > th_h):
            inpaint_mask[i:j] = 1
        i = j
    return inpaint_mask.tolist()


def inpaint_stage_ref(xs, ys, vs, img_w, img_h, inpaint_net, seq_len, batch_size=4):
    """Streaming transcription of ball_tracker.py:525-673 with torch; returns {frame: (x, y, vis)} plus the
    pre-truncation pixel values (for tolerance-aware comparison)."""
    coor_th = 50.0 / math.sqrt(HEIGHT ** 2 + WIDTH ** 2)
    img_scaler = (img_w / WIDTH, img_h / HEIGHT)
    inpaint = generate_inpaint_mask_ref(ys, vs, th_h=img_h * 0.05)
    T = len(xs)
    ids, coors, masks = [], [], []
    for i in range(T):
        if i + seq_len <= T:
            ids.append([(0, i + f) for f in range(seq_len)])
            c = np.array([(xs[i + f], ys[i + f]) for f in range(seq_len)], np.float32)
            c[:, 0] = c[:, 0] / img_w
            c[:, 1] = c[:, 1] / img_h
            coors.append(c)
            masks.append(np.array([inpaint[i + f] for f in range(seq_len)], np.float32).reshape(-1, 1))
    weight = torch.from_numpy(inpaint_ensemble_weight(seq_len))
    num_sample, sample_count = len(ids), 0
    buffer_size = seq_len - 1
    si, fi = torch.arange(seq_len), torch.arange(seq_len - 1, -1, -1)
    buf = torch.zeros((buffer_size, seq_len, 2), dtype=torch.float32)
    out, raw = {}, {}
    for b0 in range(0, num_sample, batch_size):
        i_b = torch.tensor(ids[b0:b0 + batch_size])
        coor_pred = torch.from_numpy(np.stack(coors[b0:b0 + batch_size])).float()
        m = torch.from_numpy(np.stack(masks[b0:b0 + batch_size])).float()
        ci = inpaint_net(coor_pred, m)
        ci = ci * m + coor_pred * (1 - m)
        th = (ci[:, :, 0] < coor_th) & (ci[:, :, 1] < coor_th)
        ci[th] = 0.0
        buf = torch.cat((buf, ci), 0)
        e_i, e_c = [], []
        for s in range(i_b.shape[0]):
            if sample_count < buffer_size:
                c = buf[si + s, fi].sum(0)
                c /= (sample_count + 1)
            else:
                c = (buf[si + s, fi] * weight[:, None]).sum(0)
            e_i.append(int(i_b[s][0][1])); e_c.append(c)
            sample_count += 1
            if sample_count == num_sample:
                buf = torch.cat((buf, torch.zeros((buffer_size, seq_len, 2))), 0)
                for frame_i in range(1, seq_len):
                    c = buf[si + s + frame_i, fi].sum(0)
                    c /= (seq_len - frame_i)
                    e_i.append(int(i_b[-1][frame_i][1])); e_c.append(c)
        e_c = torch.stack(e_c)
        th = (e_c[:, 0] < coor_th) & (e_c[:, 1] < coor_th)
        e_c[th] = 0.0
        for f_i, c in zip(e_i, e_c.numpy()):
            fx, fy = c[0] * WIDTH * img_scaler[0], c[1] * HEIGHT * img_scaler[1]
            px, py = int(fx), int(fy)
            out[f_i] = (px, py, 0 if (px == 0 and py == 0) else 1)
            raw[f_i] = (float(fx), float(fy))
        buf = buf[-buffer_size:]
    return out, raw


def inpaint_ensemble_weight(seq_len):
    w = np.ones(seq_len, np.float32)
    for i in range(math.ceil(seq_len / 2)):
        w[i] = i + 1
        w[seq_len - i - 1] = i + 1
    return (torch.from_numpy(w) / torch.from_numpy(w).sum()).numpy()
