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
    def __init__(self, model_path, engine: Optional[E.Engine] = None, half: bool = False, fp32_mode: Optional[str] = None):
        """``half=True`` is upstream's ``predict(half=True)`` / ``model.half()``: fp16 activations and weights with
        fp32 accumulation (BASELINE configs[4]).  The reference leaves it False (players_tracker.py:351-359), and
        only the fp32 path meets the parity bar; the fp16 path reports its own error.

        ``fp32_mode`` (``half=False`` only): "h2" — activations as fp16 pairs, three MFMA products per operand pair
        (default, ``engine.fp32_mode()``); "bx3" — exact bf16 triples, six products.  An h2 model whose activations
        leave the fp16 range (|x| > 65504) raises its overflow flag: the call is repeated on a bx3 model, which this
        object then keeps using."""
        self.ckpt = checkpoint.load_checkpoint(model_path)
        if self.ckpt.task not in ("detect", "pose"):
            raise ValueError(f"{model_path}: not a YOLOv8 detect/pose checkpoint (task {self.ckpt.task})")
        self.task = self.ckpt.task
        self.names = self.ckpt.names or {i: str(i) for i in range(self.ckpt.nc)}
        self.kpt_shape = self.ckpt.kpt_shape
        self.half = bool(half)
        self.fp32_mode = fp32_mode or E.fp32_mode()
        self.graph = G.build_yolov8(self.ckpt.state_dict, self.ckpt.nc, self.kpt_shape, dtype=self._graph_dtype())
        self._engine = engine
        self._model: Optional[E.Model] = None
        self.max_batch = 64
        #: set when an h2 model overflowed and this object rebuilt itself on the bx3 kernels (infer_frames); callers that
        #: hold the old E.Model handle (bench, tests) or coordinate several ranks (TrackingRunner) read it
        self.fell_back = False

    def _graph_dtype(self) -> str:
        return "f16" if self.half else ("h2" if self.fp32_mode == "h2" else "f32")

    def _rebuild(self) -> None:
        self.close()
        self.graph = G.build_yolov8(self.ckpt.state_dict, self.ckpt.nc, self.kpt_shape, dtype=self._graph_dtype())

    def set_half(self, half: bool) -> None:
        """Switch precision (rebuilds the packed graph; the HBM-resident model is re-created on next use)."""
        if bool(half) != self.half:
            self.half = bool(half)
            self._rebuild()

    def set_fp32_mode(self, mode: str) -> None:
        if mode not in ("h2", "bx3"):
            raise ValueError(mode)
        if mode != self.fp32_mode:
            self.fp32_mode = mode
            if not self.half:
                self._rebuild()

    # ---- device placement ("cuda" == the HIP engine; there is no CPU execution path)
    def to(self, device) -> "YOLO":
        dev = str(device)
        if dev.startswith("cuda") or dev.isdigit():
            self._ensure_model()
        elif dev == "cpu":
            # runner.py:230 parks each tracker's model on the host after it ran (the reference targets 8 GB cards,
            # README.md:38-39).  With 288 GB of HBM the weights and the activation plan stay resident by default:
            # re-planning per run would cost more than the parking saves.  PADEL_RELEASE_ON_CPU=1 makes .to("cpu") give
            # the HBM back (weights + arena: close()), like the reference; the next .to("cuda") re-uploads.
            import os
            if os.environ.get("PADEL_RELEASE_ON_CPU") == "1":
                self.close()
        else:
            raise ValueError(f"unknown device {device!r}")
        return self

    def _ensure_model(self, empty: bool = False) -> E.Model:
        if self._model is None:
            eng = self._engine or E.default_engine()
            self._model = E.Model(eng, self.graph, empty=empty)
            self._model.set_max_batch(self.max_batch)
        return self._model

    def attach(self, engine: Optional[E.Engine] = None, *, receive_weights: bool = False) -> E.Model:
        """Create the HBM-resident model now, on ``engine``.  ``receive_weights=True``: this rank did not load the
        checkpoint — the blob is allocated empty and must be filled by ``broadcast_weights`` (multi-GPU: only rank 0
        reads the .pt)."""
        if engine is not None:
            self._engine = engine
        return self._ensure_model(empty=receive_weights)

    def broadcast_weights(self, root: int = 0) -> None:
        """One-time RCCL broadcast of the packed weight blob, HBM to HBM, over the engine's communicator."""
        m = self._ensure_model()
        m.engine.bcast_weights(m, root)

    def close(self) -> None:
        if self._model is not None:
            self._model.close()
            self._model = None

    def set_max_batch(self, n: int):
        self.max_batch = int(n)
        if self._model is not None:
            self._model.set_max_batch(self.max_batch)

    # ---- inference
    def predict(self, source, conf: float = 0.25, iou: float = 0.7, imgsz: int = 640, device=None,
                classes: Optional[Sequence[int]] = None, max_det: int = 300, half: Optional[bool] = None,
                **_ignored) -> list:
        """source: list of HWC uint8 ndarrays (treated as BGR, like upstream) or PIL images.
        The whole list is one batch."""
        if half is not None:
            self.set_half(half)
        frames, reverse = _as_batch(source)
        return self._run(frames, conf, iou, imgsz, classes, max_det, E.PRE_LETTERBOX, reverse)

    def predict_frames(self, frames, conf, iou, imgsz, classes=None, max_det=300, *, channel_reverse: bool,
                       pil_stretch: bool = False) -> list:
        """Fast path used by the trackers: ``frames`` are the raw BGR video frames; the reference's host
        ``processor`` (BGR2RGB, PIL resize) is folded into the device preprocessing:
        ``channel_reverse`` = network channel c reads frame channel 2-c; ``pil_stretch`` = Pillow
        bicubic resize to imgsz x imgsz first (players_keypoints_tracker.py:260-266)."""
        return self._results(*self.infer_frames(frames, conf, iou, imgsz, classes, max_det, channel_reverse=channel_reverse,
                                                pil_stretch=pil_stretch))

    def infer_frames(self, frames, conf, iou, imgsz, classes=None, max_det=300, *, channel_reverse: bool,
                     pil_stretch: bool = False, reuse_outputs: bool = False) -> tuple:
        """The device stage alone: -> (boxes (n,max_det,6), kpts (n,max_det,nk) | None, counts (n,), (h, w), imgsz,
        pre_mode).  ``frames``: (n,h,w,3) uint8 array, a list of such frames, or a list of ``video.DeviceFrame``
        handles of one contiguous range of a clip that is already in HBM (no upload).  ``reuse_outputs``: see
        ``engine.Model.yolo_infer`` (a set is handed out again ``engine.Model.OUT_RING`` calls later)."""
        from . import video
        pre_mode = E.PRE_PIL_STRETCH if pil_stretch else E.PRE_LETTERBOX
        dev = None if isinstance(frames, np.ndarray) else video.device_batch(frames)
        if dev is not None:
            src, n, h, w = dev
        else:
            src = video.host_batch(frames)
            n, h, w, _ = src.shape
        kw = dict(imgsz=int(imgsz), conf=float(conf), iou=float(iou), classes=classes, max_det=int(max_det),
                  pre_mode=pre_mode, channel_reverse=channel_reverse, letterbox_auto=True, reuse_outputs=reuse_outputs)
        m = self._ensure_model()
        boxes, kpts, counts = m.yolo_infer(src, n, h, w, **kw)
        if self.graph.dtype == G.DTYPE_H2 and m.take_overflow():
            # an activation left the fp16 range: this checkpoint needs the full-range arithmetic from now on
            print(f"padel_analytics_amd: activations beyond the fp16 range — switching this model to the bf16x3 path")
            self.set_fp32_mode("bx3")
            self.fell_back = True
            boxes, kpts, counts = self._ensure_model().yolo_infer(src, n, h, w, **kw)
        return boxes, kpts, counts, (h, w), int(imgsz), pre_mode

    # ---- the device stage in two halves (pa_yolo_submit / pa_yolo_wait): the trackers' batch loops submit batch k + 1 before
    # collecting batch k, so the GPU does not idle while the host unpacks results and prepares the next call
    def submit_frames(self, frames, conf, iou, imgsz, classes=None, max_det=300, *, channel_reverse: bool,
                      pil_stretch: bool = False):
        """-> a token for ``collect_frames``, or None when the two-call form does not apply (frames not in HBM, profiling
        on, a fake engine): the caller then uses ``infer_frames``."""
        from . import video
        dev = None if isinstance(frames, np.ndarray) else video.device_batch(frames)
        m = self._ensure_model()
        if dev is None or not hasattr(m, "yolo_submit") or getattr(m.engine, "profiling", False) or getattr(m.engine, "timeline", False):
            return None                              # (pa_yolo_submit refuses while profiling / collecting a timeline)
        src, n, h, w = dev
        if n > m.max_batch:
            return None
        pre_mode = E.PRE_PIL_STRETCH if pil_stretch else E.PRE_LETTERBOX
        ticket = m.yolo_submit(src, n, h, w, imgsz=int(imgsz), conf=float(conf), iou=float(iou), classes=classes,
                               max_det=int(max_det), pre_mode=pre_mode, channel_reverse=channel_reverse, letterbox_auto=True)
        return dict(ticket=ticket, model=m, frames=frames, args=(conf, iou, imgsz, classes, max_det),
                    kw=dict(channel_reverse=channel_reverse, pil_stretch=pil_stretch), ret=((h, w), int(imgsz), pre_mode))

    def collect_frames(self, token) -> tuple:
        """Results of a ``submit_frames`` token, in ``infer_frames``' form.  If an activation left the fp16 range (now or in
        an earlier ticket, which replaced the model) the batch is computed again on the bf16x3 model, like ``infer_frames``."""
        m = token["model"]
        if m is not self._model:                     # the model this was submitted to is gone (overflow fallback in between)
            return self.infer_frames(token["frames"], *token["args"], reuse_outputs=True, **token["kw"])
        boxes, kpts, counts, ovf = m.yolo_wait(token["ticket"])
        if ovf and self.graph.dtype == G.DTYPE_H2:
            print(f"padel_analytics_amd: activations beyond the fp16 range — switching this model to the bf16x3 path")
            self.set_fp32_mode("bx3")                # (closing the model drains its stream: later tickets die with it)
            self.fell_back = True
            return self.infer_frames(token["frames"], *token["args"], reuse_outputs=True, **token["kw"])
        return (boxes, kpts, counts) + token["ret"]

    def discard_frames(self, token) -> None:
        """Give up a ``submit_frames`` token whose results nobody wants (a batch loop that ends early): wait for the ticket
        if the model it was submitted to is still the live one, nothing otherwise — never a fresh inference, never a model
        re-upload (closing a model drained its stream and its tickets died with it)."""
        m = token["model"]
        if m is self._model and getattr(m, "handle", None):
            m.yolo_wait(token["ticket"])

    def _run(self, frames: np.ndarray, conf, iou, imgsz, classes, max_det, pre_mode, reverse) -> list:
        return self._results(*self.infer_frames(frames, conf, iou, imgsz, classes, max_det, channel_reverse=reverse,
                                                pil_stretch=pre_mode == E.PRE_PIL_STRETCH))

    def _results(self, boxes, kpts, counts, hw, imgsz, pre_mode) -> list:
        h, w = hw
        n = len(counts)
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
