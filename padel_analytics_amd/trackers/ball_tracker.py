"""Ball tracker — drop-in for the reference's ``trackers/ball_tracker/ball_tracker.py`` (``Ball`` :139-205,
``BallTracker`` :208-708), TrackNet stage.

Hot path (``predict_frames`` :373-523): background median of the first ``median_max_sample_num`` frames ->
every frame Pillow-resized to 512x288 -> 8-frame sliding windows (+ background = 27 channels) -> TrackNet ->
temporal ensemble of the 8 overlapping outputs -> threshold .5 -> largest bounding rectangle of the connected
components -> centre scaled to source pixels.  On the GPU (``pa_ball_*``): resize, window assembly, network,
ensemble, threshold and the connected-component rectangle pick (``ball_locate_kernel``; the scipy.ndimage
``predict_location`` below is only the fallback for masks with more than 12 288 foreground pixels).  On the host:
nothing numeric: the background median of the first ``median_max_sample_num`` frames is a per-thread LDS
histogram kernel too (``median_kernel``, K11).

Deliberate differences: the window stream is contiguous across the median boundary
(#10: the reference drops 7 windows for clips longer than ``median_range``); the tracker completes without an
InpaintNet (#3: the reference raises KeyError); nothing hard-codes ``.cuda()``.  The InpaintNet stage
(:525-673) runs on the host in numpy (``padel_analytics_amd/inpaint.py``; 0.5 M parameters over 16-long
coordinate sequences).
"""
from __future__ import annotations

from pathlib import Path
from typing import Iterable, Optional, Type

import numpy as np
from scipy import ndimage

from .. import checkpoint, engine as E, graph as G, inpaint, video
from .tracker import NoPredictSample, Object, Tracker


class Ball(Object):
    def __init__(self, frame: int, xy: tuple, visibility: int, projection: Optional[tuple] = None):
        super().__init__()
        self.frame = frame
        self.xy = xy
        self.visibility = visibility
        self.projection = projection

    @classmethod
    def from_json(cls, x: dict): return cls(**x)

    def serialize(self) -> dict:
        return {"frame": self.frame, "xy": self.xy, "visibility": self.visibility, "projection": self.projection}

    def asint(self) -> tuple: return tuple(int(v) for v in self.xy)


def predict_location(mask_u8: np.ndarray) -> tuple:
    """predict.py:7-39: bounding rectangle of maximal w*h among the 8-connected foreground components; on ties
    the first one in cv2.findContours order (taken as reverse raster-discovery order — cv2 is unavailable)."""
    if mask_u8.max() == 0:
        return 0, 0, 0, 0
    lab, _ = ndimage.label(mask_u8 > 0, structure=np.ones((3, 3)))
    rects = [(sl[1].start, sl[0].start, sl[1].stop - sl[1].start, sl[0].stop - sl[0].start)
             for sl in ndimage.find_objects(lab)][::-1]
    best = 0
    for i in range(1, len(rects)):
        if rects[i][2] * rects[i][3] > rects[best][2] * rects[best][3]:
            best = i
    return rects[best]


class BallTracker(Tracker):
    EVAL_MODE = "weight"
    TRAJECTORY_LENGTH = 8
    HEIGHT = 288
    WIDTH = 512
    # a frame's ensembled heat map needs the 8 windows that contain it: 7 frames before and 7 after
    temporal_context = (TRAJECTORY_LENGTH - 1, TRAJECTORY_LENGTH - 1)
    streams = True

    def __init__(self, tracking_model_path: str, inpainting_model_path: Optional[str], batch_size: int,
                 median_max_sample_num: int = 1800, median: Optional[np.ndarray] = None,
                 load_path: Optional[str | Path] = None, save_path: Optional[str | Path] = None):
        super().__init__(load_path=load_path, save_path=save_path)
        ck = checkpoint.load_checkpoint(tracking_model_path)
        if ck.task != "tracknet":
            raise ValueError(f"{tracking_model_path}: not a TrackNet checkpoint")
        self.tracknet_seq_len = int(ck.param_dict.get("seq_len", 8))
        assert self.tracknet_seq_len == self.TRAJECTORY_LENGTH
        self.bg_mode = ck.param_dict.get("bg_mode", "concat")
        assert self.bg_mode == "concat", "only bg_mode='concat' (27 input channels) is wired, like the reference (:402,:443)"
        self._state_dict = ck.state_dict
        self.fp32_mode = E.fp32_mode()           # "h2" (fp16 pairs, 3 products) or "bx3"; see yolo.YOLO
        self.graph = G.build_tracknet(ck.state_dict, dtype="h2" if self.fp32_mode == "h2" else "f32")
        self.inpaintnet = None
        if inpainting_model_path:
            ick = checkpoint.load_checkpoint(inpainting_model_path)
            if ick.task != "inpaintnet":
                raise ValueError(f"{inpainting_model_path}: not an InpaintNet checkpoint")
            self.inpaintnet_seq_len = int(ick.param_dict.get("seq_len", 16))
            self.inpaintnet = inpaint.InpaintNetHost(ick.state_dict)     # CPU twin (tests; boxes without a GPU cannot get here)
            self._inpaint_sd = ick.state_dict
        self._inpaint_dev = None                                         # the device network (round 4), created on first use
        self.batch_size = batch_size
        self.median_max_sample_num = median_max_sample_num
        self.median = median
        self.video_info = None
        self._model: Optional[E.Model] = None
        self._engine: Optional[E.Engine] = None

    def video_info_post_init(self, video_info) -> "BallTracker":
        self.video_info = video_info
        return self

    def object(self) -> Type[Object]: return Ball

    def draw_kwargs(self) -> dict: return {}

    def __str__(self) -> str: return "ball_tracker"

    def restart(self) -> None: self.results.restart()

    @property
    def full_range(self) -> bool:
        return self.graph.dtype != G.DTYPE_H2

    def use_full_range(self) -> None:
        if self.graph.dtype == G.DTYPE_H2:
            if self._model is not None:
                self._model.close()
                self._model = None
            self.fp32_mode = "bx3"
            self.graph = G.build_tracknet(self._state_dict, dtype="f32")

    def to(self, device: str) -> None:
        if str(device).startswith("cuda"):
            if self._model is None:
                self._model = E.Model(self._engine or E.default_engine(), self.graph)
                self._model.set_max_batch(max(1, self.batch_size))
        else:
            if self._model is not None:
                self._model.close()
                self._model = None
            if self._inpaint_dev is not None:
                self._inpaint_dev.close()

    def _inpaint_net(self):
        """InpaintNet on the HIP engine when there is one (K12), else its numpy twin (CPU-only test boxes)."""
        if self.DEVICE != "cuda":
            return self.inpaintnet
        if self._inpaint_dev is None:
            self._inpaint_dev = inpaint.InpaintNetDevice(self._inpaint_sd, self._engine)
        return self._inpaint_dev

    def predict_sample(self, sample: Iterable[np.ndarray], **kwargs):
        raise NoPredictSample()

    def predict_frames(self, frame_generator: Iterable[np.ndarray], total_frames: int = None, **kwargs) -> list:
        return self.merge_partials(self.predict_partial(frame_generator))

    # ---- background (iterable.py:59-78): np.median of the first median_max_sample_num frames, on the device
    def compute_median(self, frames) -> np.ndarray:
        """(h, w, 3) uint8 RGB median of a list of BGR frames (host arrays or HBM-resident DeviceFrames).  In sharded
        runs rank 0 computes it over the head of the clip and every rank receives it (``self.median``)."""
        self.to("cuda")
        h0, w0 = frames[0].shape[:2]
        sess = E.BallSession(self._model, h0, w0)
        try:
            dev = video.device_batch(frames)
            if dev is not None:
                return sess.background_from_frames(dev[0], want_median=True, n=dev[1])
            return sess.background_from_frames(video.host_batch(frames), want_median=True)
        finally:
            sess.close()

    def predict_partial(self, frame_generator: Iterable[np.ndarray], *, first_frame: int = 0, head_context: int = 0,
                        tail_context: int = 0, **kwargs) -> list:
        """TrackNet stage over one contiguous range of the clip -> [(x, y, visibility)] per OWNED frame.

        ``frame_generator`` yields ``head_context`` frames before the owned range and ``tail_context`` after it.  The
        session treats what it is fed as a stream with its own head (plain means over the windows seen so far) and
        tail; the outputs of the context frames are exactly the ones that differ from the unsharded stream, and
        they are dropped: an owned frame with >= 7 frames fed before it gets the full weighted ensemble, and one
        within 7 frames of the real start / end of the clip gets the reference's head / tail means because there
        the context is empty (``head_context == 0`` / ``tail_context == 0``)."""
        self.to("cuda")
        it = iter(frame_generator)
        head = []
        median = self.median
        if median is None:                       # iterable.py:59-74 (RGB frames, np.median, uint8 truncation)
            if first_frame != 0:
                raise ValueError("a shard that does not start at frame 0 needs the clip's background median (set .median)")
            for f in it:
                head.append(f)
                if len(head) == self.median_max_sample_num:
                    break
            if not head:
                return []
        first = head[0] if head else next(it, None)
        if first is None:
            return []
        if not head:
            head = [first]
        h0, w0 = first.shape[:2]
        self._last_hw = (h0, w0)
        w_scaler, h_scaler = w0 / self.WIDTH, h0 / self.HEIGHT
        sess = E.BallSession(self._model, h0, w0)
        if median is None:                       # K11: np.median(frames_rgb, 0).astype(uint8) on the device
            dev = video.device_batch(head)
            if dev is not None:
                sess.background_from_frames(dev[0], n=dev[1])
            else:
                sess.background_from_frames(video.host_batch(head))
        else:
            sess.set_background(median)
        out = []

        def consume(res):
            masks, _, rects = res
            for i, (x, y, w, h) in enumerate(rects.tolist()):
                if w < 0:                          # foreground overflowed the device list: host fallback on the mask
                    x, y, w, h = predict_location(masks[i])
                cx, cy = int(x + w / 2), int(y + h / 2)
                cx, cy = int(cx * w_scaler), int(cy * h_scaler)
                out.append((cx, cy, 0 if (cx == 0 and cy == 0) else 1))

        def chunks():
            buf = []
            for src in (head, it):
                for f in src:
                    buf.append(f)
                    if len(buf) == sess.max_feed:
                        yield buf
                        buf = []
            if buf:
                yield buf

        n_fed = 0
        for c in chunks():
            n_fed += len(c)
            dev = video.device_batch(c)
            if dev is not None:
                consume(sess.feed(dev[0], want_rects=True, n=dev[1]))
            else:
                consume(sess.feed(video.host_batch(c), want_rects=True))
        consume(sess.feed(None, flush=True, want_rects=True))
        sess.close()
        if self.graph.dtype == G.DTYPE_H2 and self._model.take_overflow():
            # activations beyond the fp16 range: the stream has been consumed, so the caller (TrackingRunner) restarts
            # this tracker; from now on it runs the full-range bf16x3 arithmetic
            self.use_full_range()
            raise E.RangeOverflow("TrackNet activations left the fp16 range: tracker switched to the bf16x3 path, run it again")
        if len(out) < n_fed:                       # fewer than 8 frames fed: no window, no detection (reference :688-696)
            out += [None] * (n_fed - len(out))
        return out[head_context:n_fed - tail_context]

    merge_is_host_only = False          # the InpaintNet runs on the device

    def merge_partials(self, partials: list, **kwargs) -> list:
        """InpaintNet trajectory repair over the WHOLE clip (:525-673) + ``Ball`` objects with global frame numbers."""
        n_total = len(partials)
        have = [p for p in partials if p is not None]
        xs, ys, vs = [p[0] for p in have], [p[1] for p in have], [p[2] for p in have]
        if self.inpaintnet is not None and len(xs) == n_total and n_total:
            h0, w0 = self._frame_hw()
            fixed = inpaint.inpaint_trajectory(xs, ys, vs, w0, h0, self._inpaint_net(), self.inpaintnet_seq_len,
                                               self.WIDTH, self.HEIGHT)
            if fixed and fixed[0] is not None:
                xs, ys, vs = [f[0] for f in fixed], [f[1] for f in fixed], [f[2] for f in fixed]
            else:                                  # clip shorter than one InpaintNet window: nothing is covered
                xs, ys, vs = [], [], []
        out = []
        for i in range(n_total):
            if i < len(xs):
                out.append(Ball(frame=i, xy=(xs[i], ys[i]), visibility=vs[i]))
            else:                                  # clips shorter than one window (reference :688-696)
                print(f"{self}: missing detection frame {i}")
                out.append(Ball(frame=i, xy=(0.0, 0.0), visibility=0))
        return out

    # on the wire: one (n, 4) float64 array per rank — x, y, visibility, 1 (0: the frame has no output, None)
    def pack_partials(self, partials: list):
        a = np.zeros((len(partials), 4), np.float64)
        for i, p in enumerate(partials):
            if p is not None:
                a[i] = (p[0], p[1], p[2], 1.0)
        return [a]

    def unpack_partials(self, arrays: list) -> list:
        out = []
        for x, y, v, have in arrays[0]:
            out.append((self._wire_num(x), self._wire_num(y), int(v)) if have else None)
        return out

    @staticmethod
    def _wire_num(v: float):
        return int(v) if float(v).is_integer() else float(v)

    def _frame_hw(self) -> tuple:
        if getattr(self, "_last_hw", None):
            return self._last_hw
        if self.video_info is None:
            raise ValueError("BallTracker with an InpaintNet needs video_info_post_init() (frame size normalises the coordinates)")
        return self.video_info.height, self.video_info.width
