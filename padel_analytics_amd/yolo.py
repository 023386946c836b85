"""``YOLO`` — the inner drop-in boundary: a duck type of ``ultralytics.YOLO`` covering exactly what the
reference's trackers use (SURVEY.md §8(b) "Inner"): ``YOLO(model_path)``, ``.to(device)``,
``.predict(source, conf, iou, imgsz, device, classes, max_det) -> list[Results]`` with
``result.boxes.xyxy/.conf/.cls/.id``, ``result.names`` and ``result.keypoints.xy``
(call sites: ``players_tracker.py:303,338-339,351-359``, ``players_keypoints_tracker.py:238,285-299``).

Everything numeric happens in ``libpadel_hip.so`` (letterbox / PIL-bicubic on device, fp32 MFMA
convolutions, decode, NMS, box/keypoint rescale); this file only marshals arrays.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np

from . import checkpoint, engine as E, graph as G


class Boxes:
    def __init__(self, data: np.ndarray, orig_shape):
        self.data = data                      # (n, 6) x1,y1,x2,y2,conf,cls
        self.orig_shape = orig_shape
        self.id = None                        # predict() never tracks (is_track False)

    @property
    def xyxy(self): return self.data[:, :4]

    @property
    def conf(self): return self.data[:, 4]

    @property
    def cls(self): return self.data[:, 5]

    def __len__(self): return len(self.data)


class Keypoints:
    """``xy`` zeroes points whose visibility is < 0.5 when the checkpoint predicts one (ndim == 3)."""

    def __init__(self, data: np.ndarray, orig_shape):
        self.data = data                      # (n, K, ndim)
        self.orig_shape = orig_shape

    @property
    def xy(self) -> np.ndarray:
        k = self.data[..., :2].copy()
        if self.data.shape[-1] == 3:
            k[self.data[..., 2] < 0.5] = 0
        return k

    @property
    def conf(self): return self.data[..., 2] if self.data.shape[-1] == 3 else None

    def __len__(self): return len(self.data)


class Results:
    def __init__(self, orig_shape, names, boxes: np.ndarray, keypoints: Optional[np.ndarray] = None):
        self.orig_shape = orig_shape
        self.names = names
        self.boxes = Boxes(boxes, orig_shape)
        self.keypoints = None if keypoints is None else Keypoints(keypoints, orig_shape)

    def __len__(self): return len(self.boxes)


class YOLO:
    def __init__(self, model_path, engine: Optional[E.Engine] = None):
        self.ckpt = checkpoint.load_checkpoint(model_path)
        if self.ckpt.task not in ("detect", "pose"):
            raise ValueError(f"{model_path}: not a YOLOv8 detect/pose checkpoint (task {self.ckpt.task})")
        self.task = self.ckpt.task
        self.names = self.ckpt.names or {i: str(i) for i in range(self.ckpt.nc)}
        self.kpt_shape = self.ckpt.kpt_shape
        self.graph = G.build_yolov8(self.ckpt.state_dict, self.ckpt.nc, self.kpt_shape)
        self._engine = engine
        self._model: Optional[E.Model] = None
        self.max_batch = 64

    # ---- device placement ("cuda" == the HIP engine; there is no CPU execution path)
    def to(self, device) -> "YOLO":
        dev = str(device)
        if dev.startswith("cuda") or dev.isdigit():
            self._ensure_model()
        elif dev == "cpu":
            if self._model is not None:       # runner.py:230 parks models on the host after a tracker ran
                self._model.close()
                self._model = None
        else:
            raise ValueError(f"unknown device {device!r}")
        return self

    def _ensure_model(self) -> E.Model:
        if self._model is None:
            eng = self._engine or E.default_engine()
            self._model = E.Model(eng, self.graph)
            self._model.set_max_batch(self.max_batch)
        return self._model

    def set_max_batch(self, n: int):
        self.max_batch = int(n)
        if self._model is not None:
            self._model.set_max_batch(self.max_batch)

    # ---- inference
    def predict(self, source, conf: float = 0.25, iou: float = 0.7, imgsz: int = 640, device=None,
                classes: Optional[Sequence[int]] = None, max_det: int = 300, **_ignored) -> list:
        """source: list of HWC uint8 ndarrays (treated as BGR, like upstream) or PIL images.
        The whole list is one batch."""
        frames, reverse = _as_batch(source)
        return self._run(frames, conf, iou, imgsz, classes, max_det, E.PRE_LETTERBOX, reverse)

    def predict_frames(self, frames, conf, iou, imgsz, classes=None, max_det=300, *, channel_reverse: bool,
                       pil_stretch: bool = False) -> list:
        """Fast path used by the trackers: ``frames`` are the raw BGR video frames; the reference's host
        ``processor`` (BGR2RGB, PIL resize) is folded into the device preprocessing:
        ``channel_reverse`` = network channel c reads frame channel 2-c; ``pil_stretch`` = Pillow
        bicubic resize to imgsz x imgsz first (players_keypoints_tracker.py:260-266)."""
        frames = frames if isinstance(frames, (np.ndarray, E.DeviceBuffer)) else np.stack(list(frames))
        return self._run(frames, conf, iou, imgsz, classes, max_det,
                         E.PRE_PIL_STRETCH if pil_stretch else E.PRE_LETTERBOX, channel_reverse)

    def _run(self, frames: np.ndarray, conf, iou, imgsz, classes, max_det, pre_mode, reverse) -> list:
        m = self._ensure_model()
        n, h, w, _ = frames.shape
        boxes, kpts, counts = m.yolo_infer(frames, n, h, w, imgsz=int(imgsz), conf=float(conf), iou=float(iou),
                                           classes=classes, max_det=int(max_det), pre_mode=pre_mode,
                                           channel_reverse=reverse, letterbox_auto=True)
        oshape = (imgsz, imgsz) if pre_mode == E.PRE_PIL_STRETCH else (h, w)
        out = []
        for i in range(n):
            c = int(counts[i])
            k = None
            if kpts is not None:
                k = kpts[i, :c].reshape(c, *self.kpt_shape).copy()
            out.append(Results(oshape, self.names, boxes[i, :c].copy(), k))
        return out

    __call__ = predict


def _as_batch(source):
    """-> (N,H,W,3) uint8 array in the order the arrays were given + whether channels must be reversed so
    the network sees RGB.  ndarray sources are BGR by upstream convention -> reverse; PIL images are RGB."""
    if isinstance(source, np.ndarray) and source.ndim == 4:
        return source, True
    if isinstance(source, np.ndarray):
        source = [source]
    source = list(source)
    if not source:
        raise ValueError("empty source")
    if all(isinstance(s, np.ndarray) for s in source):
        shapes = {s.shape for s in source}
        if len(shapes) != 1:
            raise ValueError("all frames of one predict() call must share a shape")
        return np.stack(source), True
    arrs = [np.asarray(s.convert("RGB")) for s in source]    # PIL
    if len({a.shape for a in arrs}) != 1:
        raise ValueError("all images of one predict() call must share a shape")
    return np.stack(arrs), False
