"""Ball tracker on a YOLOv8 detect model — SURVEY.md §8 row a16 (no reference counterpart: the reference's
``BallTracker`` is TrackNetV3, ``ball_tracker.py:208-708``; BASELINE's configs name "players + ball YOLOv8 detect").

Same plugin contract and the same result type as the reference's ball tracker: one ``Ball(frame, xy, visibility)``
per frame (``ball_tracker.py:139-163``), ``__str__() == "ball_tracker"`` so it is a drop-in for ``BallTracker`` in
``TrackingRunner``.  Per frame: the nc=1 detector's top-1 box after NMS (rows come back sorted by confidence, so
``max_det=1`` is exactly the best surviving box); ``xy`` = box centre in source pixels, ``visibility`` = 1 iff the
frame has a detection, otherwise ``xy = (0, 0)``, ``visibility = 0`` — the convention of the reference's
``predict_modified`` ("visibility 0 iff both coordinates are 0", ``predict.py:149-221``).
"""
from __future__ import annotations

from pathlib import Path
from typing import Iterable, Optional, Type

import numpy as np

from ..yolo import YOLO
from .ball_tracker import Ball
from .tracker import NoPredictFrames, Object, Tracker


class BallDetectTracker(Tracker):
    CONF = 0.25
    IOU = 0.7
    IMGSZ = 640
    streams = False

    def __init__(self, model_path: str, batch_size: int, conf: Optional[float] = None,
                 load_path: Optional[str | Path] = None, save_path: Optional[str | Path] = None, half: bool = False):
        super().__init__(load_path=load_path, save_path=save_path)
        self.model = YOLO(model_path, half=half)
        if self.model.task != "detect":
            raise ValueError(f"{model_path}: BallDetectTracker needs a YOLOv8 detect checkpoint")
        self.batch_size = batch_size
        if conf is not None:
            self.CONF = float(conf)
        self._next_frame = 0

    def video_info_post_init(self, video_info) -> "BallDetectTracker": return self

    def object(self) -> Type[Object]: return Ball

    def draw_kwargs(self) -> dict: return {}

    def __str__(self) -> str: return "ball_tracker"

    def restart(self) -> None:
        self.results.restart()
        self._next_frame = 0

    def to(self, device: str) -> None:
        self.model.to(device)

    def infer_sample(self, sample, **kwargs):
        # players path convention: raw BGR frames reach the network in their own channel order (App. C #1)
        boxes, _, counts, _, _, _ = self.model.infer_frames(sample, self.CONF, self.IOU, self.IMGSZ, classes=None,
                                                            max_det=1, channel_reverse=False, reuse_outputs=self._reuse_outputs)
        return boxes, counts

    def submit_sample(self, sample, **kwargs):
        if not hasattr(self.model, "submit_frames"):
            return None
        return self.model.submit_frames(sample, self.CONF, self.IOU, self.IMGSZ, classes=None, max_det=1, channel_reverse=False)

    def collect_sample(self, token):
        boxes, _, counts, _, _, _ = self.model.collect_frames(token)
        return boxes, counts

    @staticmethod
    def top1_to_xyv(boxes: np.ndarray, counts: np.ndarray) -> list:
        """(n, >=1, 6) boxes + (n,) counts -> [(x, y, visibility)]: centre of the best box, (0, 0, 0) without one."""
        out = []
        for i in range(len(counts)):
            if counts[i] > 0:
                x1, y1, x2, y2 = (float(v) for v in boxes[i, 0, :4])
                out.append(((x1 + x2) / 2, (y1 + y2) / 2, 1))
            else:
                out.append((0.0, 0.0, 0))
        return out

    def post_sample(self, raw, **kwargs) -> list:
        preds = []
        for x, y, v in self.top1_to_xyv(*raw):
            preds.append(Ball(frame=self._next_frame, xy=(x, y), visibility=v))
            self._next_frame += 1
        return preds

    def predict_sample(self, sample: Iterable[np.ndarray], **kwargs) -> list:
        return self.post_sample(self.infer_sample(sample, **kwargs), **kwargs)

    def predict_frames(self, frame_generator, **kwargs):
        raise NoPredictFrames()

    # sharded: frame numbers are global, assigned on rank 0 after the gather
    def predict_partial(self, frame_generator, *, first_frame: int = 0, head_context: int = 0, tail_context: int = 0,
                        **kwargs) -> list:
        out = []
        for raw in self._raw_batches(frame_generator):
            out += self.top1_to_xyv(*raw)
        return out

    def merge_partials(self, partials: list, **kwargs) -> list:
        return [Ball(frame=i, xy=(x, y), visibility=v) for i, (x, y, v) in enumerate(partials)]

    # on the wire: one (n, 3) float64 array per rank (x, y exactly; visibility 0 / 1)
    def pack_partials(self, partials: list):
        return [np.array(partials, np.float64).reshape(-1, 3)]

    def unpack_partials(self, arrays: list) -> list:
        return [(float(x), float(y), int(v)) for x, y, v in arrays[0]]
