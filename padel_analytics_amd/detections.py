"""Minimal stand-ins for the two ``supervision`` types the players tracker touches
(``players_tracker.py:363-369``; ``main.py:108-119``): ``Detections`` and ``PolygonZone``.

``supervision`` is not installable here and is un-pinned upstream; the semantics below
follow the 0.2x line that ``main.py:118`` (``frame_resolution_wh=``) implies:

* ``Detections.from_ultralytics`` copies ``boxes.xyxy/conf/cls`` (+ ``boxes.id`` when tracking);
* ``PolygonZone.trigger`` keeps detections whose BOTTOM-CENTRE anchor (box clipped to the frame, then
  ``ceil`` to int) falls on a non-zero pixel of the rasterised polygon; the mask is ``(h+1, w+1)``.
  The rasteriser is a restatement of ``cv2.fillPoly`` (even-odd scanline fill at integer rows plus the
  polygon outline): *parity unpinned* — OpenCV's fixed-point edge walker cannot be checked offline.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np


@dataclass
class Detections:
    xyxy: np.ndarray
    confidence: Optional[np.ndarray] = None
    class_id: Optional[np.ndarray] = None
    tracker_id: Optional[np.ndarray] = None

    def __post_init__(self):
        self.xyxy = np.asarray(self.xyxy, dtype=np.float32).reshape(-1, 4)

    def __len__(self) -> int:
        return len(self.xyxy)

    @classmethod
    def empty(cls) -> "Detections":
        return cls(np.empty((0, 4), np.float32), np.empty((0,), np.float32), np.empty((0,), int))

    @classmethod
    def from_ultralytics(cls, result) -> "Detections":
        b = result.boxes
        tid = None if getattr(b, "id", None) is None else np.asarray(b.id).astype(int)
        return cls(np.asarray(b.xyxy, np.float32), np.asarray(b.conf, np.float32), np.asarray(b.cls).astype(int), tid)

    def __getitem__(self, idx) -> "Detections":
        if isinstance(idx, (int, np.integer)):
            idx = [int(idx)]
        idx = np.asarray(idx)
        pick = lambda a: None if a is None else np.asarray(a)[idx]
        return Detections(self.xyxy[idx], pick(self.confidence), pick(self.class_id), pick(self.tracker_id))


def polygon_to_mask(polygon: np.ndarray, resolution_wh) -> np.ndarray:
    """Restatement of ``cv2.fillPoly(mask, [polygon], 1)`` on a (h, w) uint8 mask."""
    w, h = resolution_wh
    mask = np.zeros((h, w), np.uint8)
    pts = np.asarray(polygon, dtype=np.int64).reshape(-1, 2)
    n = len(pts)
    if n == 0:
        return mask
    # interior: even-odd rule on integer scanlines (half-open edge rule avoids double counting vertices)
    ys = np.arange(max(0, pts[:, 1].min()), min(h - 1, pts[:, 1].max()) + 1)
    for y in ys:
        xs = []
        for i in range(n):
            (x0, y0), (x1, y1) = pts[i], pts[(i + 1) % n]
            if y0 == y1:
                continue
            if (y >= min(y0, y1)) and (y < max(y0, y1)):
                xs.append(x0 + (y - y0) * (x1 - x0) / (y1 - y0))
        xs.sort()
        for a, b in zip(xs[0::2], xs[1::2]):
            xa, xb = int(np.ceil(a)), int(np.floor(b))
            if xb >= xa:
                mask[y, max(xa, 0):min(xb, w - 1) + 1] = 1
    # outline (fillPoly also draws the edges)
    for i in range(n):
        (x0, y0), (x1, y1) = pts[i], pts[(i + 1) % n]
        steps = int(max(abs(x1 - x0), abs(y1 - y0)))
        for t in range(steps + 1):
            x = int(round(x0 + (x1 - x0) * t / max(steps, 1)))
            y = int(round(y0 + (y1 - y0) * t / max(steps, 1)))
            if 0 <= x < w and 0 <= y < h:
                mask[y, x] = 1
    return mask


class PolygonZone:
    def __init__(self, polygon: np.ndarray, frame_resolution_wh, triggering_position: str = "bottom_center"):
        self.polygon = np.asarray(polygon).astype(int)
        self.frame_resolution_wh = tuple(frame_resolution_wh)
        self.triggering_position = triggering_position
        w, h = self.frame_resolution_wh
        self.mask = polygon_to_mask(self.polygon, (w + 1, h + 1))
        self.current_count = 0

    def trigger(self, detections: Detections) -> np.ndarray:
        inside = self.trigger_boxes(detections.xyxy)
        self.current_count = int(inside.sum())
        return inside

    def trigger_boxes(self, xyxy: np.ndarray) -> np.ndarray:
        """``trigger`` on a bare (m, 4) box array (the tracker calls it once per batch of frames)."""
        if len(xyxy) == 0:
            return np.zeros((0,), bool)
        w, h = self.frame_resolution_wh
        b = np.array(xyxy, dtype=np.float32).reshape(-1, 4)
        b[:, [0, 2]] = b[:, [0, 2]].clip(0, w)
        b[:, [1, 3]] = b[:, [1, 3]].clip(0, h)
        if self.triggering_position == "bottom_center":
            ax, ay = (b[:, 0] + b[:, 2]) / 2, b[:, 3]
        elif self.triggering_position == "center":
            ax, ay = (b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2
        else:
            raise ValueError(self.triggering_position)
        ax, ay = np.ceil(ax).astype(int), np.ceil(ay).astype(int)
        return self.mask[ay, ax].astype(bool)
