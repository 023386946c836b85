"""Parity harness: HIP engine vs CPU oracle, robust to fp32-noise-level decision flips.

Two fp32 evaluations of the same 60-90 layer graph differ by ~1e-5 relative on the head logits
(measured: torch oneDNN vs torch native vs fp64), so a candidate whose score is within
that noise of `conf`, or a pair whose IoU is within it of `iou`, can legitimately be decided either way.
The harness therefore
  1. matches engine detections to oracle detections one-to-one (same class, nearest box),
  2. if the sets differ, searches the oracle's own candidate list for threshold-adjacent decisions,
     re-runs the oracle NMS with those decisions flipped and requires the engine's set to equal one of
     the outcomes (anything else is a real bug and fails),
  3. reports the L-inf coordinate error over matched detections.
"""
from __future__ import annotations

import itertools

import numpy as np
import torch

from oracle import yolov8_ref as ref

ADJ = 2e-4   # |IoU - thr| or |score - conf| below this is "threshold-adjacent" (fp32 noise on logits ~5e-4)


def _iou_matrix(b):
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    area = (x2 - x1) * (y2 - y1)
    xx1 = np.maximum(x1[:, None], x1[None]); yy1 = np.maximum(y1[:, None], y1[None])
    xx2 = np.minimum(x2[:, None], x2[None]); yy2 = np.minimum(y2[:, None], y2[None])
    inter = np.maximum(0, xx2 - xx1) * np.maximum(0, yy2 - yy1)
    with np.errstate(divide="ignore", invalid="ignore"):
        return inter / (area[:, None] + area[None] - inter)


def _rescaled(cands, net_hw, orig_hw):
    return ref.scale_boxes(net_hw, torch.from_numpy(cands[:, :4].copy()), orig_hw).numpy()


def match(ref_boxes, got_boxes, tol_match=0.25):
    """one-to-one nearest matching on xyxy with equal class; returns (pairs, ref_unmatched, got_unmatched)."""
    pairs, used = [], set()
    for i, rb in enumerate(ref_boxes):
        best, bj = tol_match, -1
        for j, gb in enumerate(got_boxes):
            if j in used or gb[5] != rb[5]:
                continue
            d = float(np.abs(gb[:4] - rb[:4]).max())
            if d < best:
                best, bj = d, j
        if bj >= 0:
            used.add(bj)
            pairs.append((i, bj))
    ru = [i for i in range(len(ref_boxes)) if i not in {p[0] for p in pairs}]
    gu = [j for j in range(len(got_boxes)) if j not in used]
    return pairs, ru, gu


def explain_by_flips(r, got_boxes, conf, iou, max_det, max_flips=3):
    """Try to reproduce the engine's kept set by flipping threshold-adjacent decisions of the oracle."""
    c = r["cands"]
    if len(c) == 0:
        return False, "no oracle candidates"
    scaled = _rescaled(c, r["net_hw"], r["orig_hw"])
    ious = _iou_matrix((c[:, :4] + c[:, 5:6] * ref.MAX_WH).astype(np.float64))
    ii, jj = np.where(np.abs(ious - iou) < ADJ)
    pairs = sorted({(int(a), int(b)) for a, b in zip(ii, jj) if a != b})
    # NMS visits (higher score i, lower score j); keep both orientations, the visit order decides
    fuzzy = pairs[:16]
    if not fuzzy:
        return False, "no threshold-adjacent IoU pair among oracle candidates"
    ct = torch.from_numpy(c)
    for k in range(1, max_flips + 1):
        for combo in itertools.combinations(fuzzy, k):
            keep = ref.nms_torchvision(ct[:, :4] + ct[:, 5:6] * ref.MAX_WH, ct[:, 4], iou, flip=set(combo))[:max_det].numpy()
            kb = np.concatenate([scaled[keep], c[keep, 4:6]], 1)
            if len(kb) != len(got_boxes):
                continue
            pr, ru, gu = match(kb, got_boxes)
            if not ru and not gu:
                return True, f"explained by flipping IoU-adjacent pair(s) {combo} (|IoU-thr|<{ADJ})"
    return False, f"{len(fuzzy)} adjacent pairs but no flip combination reproduces the engine's set"


def compare_image(r, got_boxes, got_kpts, conf, iou, max_det=300, kpt_shape=None):
    """-> dict(worst_px, flips:str|None).  Raises AssertionError on a real mismatch."""
    rb = r["boxes"]
    pairs, ru, gu = match(rb, got_boxes)
    note = None
    if ru or gu:
        ok, note = explain_by_flips(r, got_boxes, conf, iou, max_det)
        assert ok, (f"detection sets differ (oracle {len(rb)}, engine {len(got_boxes)}, unmatched oracle {ru[:5]}, "
                    f"unmatched engine {gu[:5]}; conf margin {r['conf_margin']:.2e}): {note}")
    worst = 0.0
    worst_score = 0.0
    sq, cnt = 0.0, 0
    for i, j in pairs:
        assert got_boxes[j, 5] == rb[i, 5], "class ids differ"
        d = (got_boxes[j, :4].astype(np.float64) - rb[i, :4].astype(np.float64))
        sq += float((d * d).sum()); cnt += 4
        worst = max(worst, float(np.abs(got_boxes[j, :4] - rb[i, :4]).max()))
        worst_score = max(worst_score, float(abs(got_boxes[j, 4] - rb[i, 4])))
        if kpt_shape is not None and r["kpts"] is not None:
            gk = got_kpts[j].reshape(*kpt_shape)
            rk = r["kpts"][i]
            worst = max(worst, float(np.abs(gk[..., :2] - rk[..., :2]).max()))
            dk = (gk[..., :2].astype(np.float64) - rk[..., :2].astype(np.float64))
            sq += float((dk * dk).sum()); cnt += dk.size
            if kpt_shape[1] == 3:
                worst_score = max(worst_score, float(np.abs(gk[..., 2] - rk[..., 2]).max()))
    return {"worst_px": worst, "worst_score": worst_score, "flips": note, "n": len(pairs), "sq": sq, "cnt": cnt}


def compare_batch(res_ref, boxes, kpts, counts, conf, iou, max_det=300, kpt_shape=None):
    out = {"worst_px": 0.0, "worst_score": 0.0, "flips": [], "n": 0, "rms_px": 0.0}
    sq, cnt = 0.0, 0
    for i, r in enumerate(res_ref):
        gb = boxes[i, :counts[i]]
        gk = None if kpts is None else kpts[i, :counts[i]]
        s = compare_image(r, gb, gk, conf, iou, max_det, kpt_shape)
        out["worst_px"] = max(out["worst_px"], s["worst_px"])
        out["worst_score"] = max(out["worst_score"], s["worst_score"])
        out["n"] += s["n"]
        sq += s["sq"]; cnt += s["cnt"]
        if s["flips"]:
            out["flips"].append((i, s["flips"]))
    out["rms_px"] = float(np.sqrt(sq / max(cnt, 1)))
    return out
