"""Frame sources: the reference reads video through ``supervision`` (``sv.VideoInfo.from_video_path``
``trackers/runner.py:52``, ``sv.get_video_frames_generator`` ``runner.py:215-220``), which needs OpenCV.
Neither is available here or on the GPU box, and the checkout ships no video, so this module provides the
same two calls over (a) ``.npy`` frame stacks (N,H,W,3 uint8 BGR, memory-mapped), (b) seeded synthetic
clips ``synthetic://?n=64&h=720&w=1280&fps=30&seed=0`` and (c) real video files when ``cv2`` happens to be
importable.  Frames are HWC uint8 **BGR**, exactly what supervision yields."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterator, Optional
from urllib.parse import parse_qs, urlparse

import numpy as np


@dataclass
class VideoInfo:
    width: int
    height: int
    fps: int
    total_frames: Optional[int] = None

    @property
    def resolution_wh(self) -> tuple:
        return self.width, self.height

    @classmethod
    def from_video_path(cls, video_path) -> "VideoInfo":
        p = str(video_path)
        if p.startswith("synthetic://"):
            q = _query(p)
            return cls(q["w"], q["h"], q["fps"], q["n"])
        if p.endswith(".npy"):
            a = np.load(p, mmap_mode="r")
            return cls(int(a.shape[2]), int(a.shape[1]), 30, int(a.shape[0]))
        cv2 = _cv2()
        cap = cv2.VideoCapture(p)
        if not cap.isOpened():
            raise FileNotFoundError(f"Could not open video at {p}")
        info = cls(int(cap.get(cv2.CAP_PROP_FRAME_WIDTH)), int(cap.get(cv2.CAP_PROP_FRAME_HEIGHT)),
                   int(cap.get(cv2.CAP_PROP_FPS)), int(cap.get(cv2.CAP_PROP_FRAME_COUNT)))
        cap.release()
        return info


def _query(p: str) -> dict:
    q = {k: int(v[0]) for k, v in parse_qs(urlparse(p).query).items()}
    return {"n": q.get("n", 64), "h": q.get("h", 720), "w": q.get("w", 1280), "fps": q.get("fps", 30),
            "seed": q.get("seed", 0)}


def _cv2():
    try:
        import cv2
        return cv2
    except ImportError as e:
        raise RuntimeError("reading encoded video needs opencv-python, which is not installed; use a .npy frame "
                           "stack or a synthetic:// source") from e


def get_video_frames_generator(source_path, stride: int = 1, start: int = 0, end: Optional[int] = None) -> Iterator[np.ndarray]:
    p = str(source_path)
    if p.startswith("synthetic://"):
        from . import synth
        q = _query(p)
        stop = q["n"] if end is None else min(end, q["n"])
        for i in range(start, stop, stride):
            yield synth.synthetic_frames(1, q["h"], q["w"], seed=q["seed"] * 100003 + i)[0]
        return
    if p.endswith(".npy"):
        a = np.load(p, mmap_mode="r")
        stop = a.shape[0] if end is None else min(end, a.shape[0])
        for i in range(start, stop, stride):
            yield np.ascontiguousarray(a[i])
        return
    cv2 = _cv2()
    cap = cv2.VideoCapture(p)
    total = int(cap.get(cv2.CAP_PROP_FRAME_COUNT))
    stop = total if end is None else min(end, total)
    cap.set(cv2.CAP_PROP_POS_FRAMES, start)
    i = start
    while i < stop:
        ok, frame = cap.read()
        if not ok:
            break
        if (i - start) % stride == 0:
            yield frame
        i += 1
    cap.release()
