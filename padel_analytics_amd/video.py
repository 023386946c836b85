"""Frame sources: the reference reads video through ``supervision`` (``sv.VideoInfo.from_video_path``
``trackers/runner.py:52``, ``sv.get_video_frames_generator`` ``runner.py:215-220``), which needs OpenCV.
Neither is available here or on the GPU box, and the checkout ships no video, so this module provides the
same two calls over (a) ``.npy`` frame stacks (N,H,W,3 uint8 BGR, memory-mapped), (b) URL-style sources whose
scheme somebody registered with ``register_source`` (the tests and the bench register ``synthetic://?n=64&h=720&w=1280&
fps=30&seed=0``: tests/synth.py — generated frames are test infrastructure, not product) and (c) real video files when
``cv2`` happens to be importable, (d) ``DeviceClip`` objects: a clip already resident in HBM (the bench's device-resident mode; the
runner's fan-out mode uploads each batch once and hands the same ``DeviceFrame`` handles to every tracker).
Frames are HWC uint8 **BGR**, exactly what supervision yields."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterator, Optional

import numpy as np


# scheme -> (info(url) -> VideoInfo, frames(url, start, end, stride) -> iterator of HWC uint8 BGR frames)
_SOURCES: dict = {}


def register_source(scheme: str, info, frames) -> None:
    """Make ``scheme://...`` paths readable by ``VideoInfo.from_video_path`` / ``get_video_frames_generator``."""
    _SOURCES[scheme] = (info, frames)


def _scheme(p: str):
    i = p.find("://")
    if i <= 0:
        return None
    if p[:i] not in _SOURCES:
        raise FileNotFoundError(f"no frame source registered for '{p[:i]}://' (video.register_source; the synthetic clips of "
                                f"the tests register themselves on `import tests.synth`)")
    return _SOURCES[p[:i]]


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
        if isinstance(video_path, (DeviceClip, ArrayClip)):
            return cls(video_path.w, video_path.h, video_path.fps, video_path.total_frames)
        p = str(video_path)
        src = _scheme(p)
        if src is not None:
            return src[0](p)
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


def _cv2():
    try:
        import cv2
        return cv2
    except ImportError as e:
        raise RuntimeError("reading encoded video needs opencv-python, which is not installed; use a .npy frame "
                           "stack or a registered frame source") from e


class DeviceFrame:
    """Handle of one HWC uint8 BGR frame that lives in HBM (frame ``index`` of ``clip.buffer``).  Quacks like the
    ndarray the trackers expect as far as they look at it on the host (``.shape``)."""
    __slots__ = ("clip", "index")

    def __init__(self, clip: "DeviceClip", index: int):
        self.clip, self.index = clip, index

    @property
    def shape(self) -> tuple:
        return (self.clip.h, self.clip.w, 3)


class DeviceClip:
    """``n`` frames resident in HBM, presented as a clip of ``n * repeat`` frames (frame i = stored frame i % n)."""

    def __init__(self, engine, frames: Optional[np.ndarray] = None, *, shape: Optional[tuple] = None, repeat: int = 1,
                 fps: int = 30):
        if frames is not None:
            frames = np.ascontiguousarray(frames, np.uint8)
            shape = frames.shape
        self.n, self.h, self.w = int(shape[0]), int(shape[1]), int(shape[2])
        self.repeat, self.fps = int(repeat), int(fps)
        self.frame_bytes = self.h * self.w * 3
        self.buffer = engine.alloc(self.n * self.frame_bytes)
        if frames is not None:
            self.buffer.upload(frames)

    @property
    def total_frames(self) -> int:
        return self.n * self.repeat

    def alias(self, repeat: int) -> "DeviceClip":
        """The same frames in HBM (no copy, not an owner: never ``free()`` it) presented with another ``repeat`` — the
        bench's sharded leg shows every rank a clip of world x K x n frames of which it only ever reads its own shard."""
        c = object.__new__(DeviceClip)
        c.__dict__.update(self.__dict__)
        c.repeat = int(repeat)
        return c

    def upload(self, frames: np.ndarray, first: int = 0, copy_stream: bool = True) -> None:
        """Overwrite stored frames [first, first + len(frames)) (prefetch thread of the runner's fan-out mode)."""
        frames = np.ascontiguousarray(frames, np.uint8)
        assert frames.shape[1:] == (self.h, self.w, 3) and first + len(frames) <= self.n
        self.buffer.view(first * self.frame_bytes, frames.nbytes).upload(frames, copy_stream=copy_stream)

    def frames(self, start: int = 0, end: Optional[int] = None, stride: int = 1) -> Iterator[DeviceFrame]:
        stop = self.total_frames if end is None else min(end, self.total_frames)
        for i in range(start, stop, stride):
            yield DeviceFrame(self, i % self.n)

    def free(self) -> None:
        self.buffer.free()


class ArrayClip:
    """Host-memory clip: ``frames`` (n, h, w, 3) uint8 BGR presented as ``n * repeat`` frames (bench: the
    PCIe-inclusive rate; tests)."""

    def __init__(self, frames: np.ndarray, repeat: int = 1, fps: int = 30):
        self.array = np.ascontiguousarray(frames, np.uint8)
        self.n, self.h, self.w = self.array.shape[:3]
        self.repeat, self.fps = int(repeat), int(fps)
        self._pinned_by = None

    def pin(self, engine) -> "ArrayClip":
        """Page-lock the clip (a decoder writing into pinned memory): batches of consecutive frames then go up without a
        staging copy (``host_batch``) and at PCIe speed."""
        if self._pinned_by is None:
            engine.pin(self.array)
            self._pinned_by = engine
        elif self._pinned_by is not engine:
            raise ValueError("ArrayClip is page-locked through another engine: unpin() it first")
        return self

    def unpin(self) -> None:
        if self._pinned_by is not None:
            eng, self._pinned_by = self._pinned_by, None
            eng.unpin(self.array)

    # a page-locked range must be unregistered before numpy frees it (a stale registration makes later registrations /
    # copies of recycled pages fail): `with ArrayClip(...).pin(eng) as clip:` or let the finaliser do it
    def __enter__(self) -> "ArrayClip":
        return self

    def __exit__(self, *exc) -> None:
        self.unpin()

    def __del__(self):
        try:
            self.unpin()
        except Exception:
            pass

    @property
    def total_frames(self) -> int:
        return self.n * self.repeat

    def frames(self, start: int = 0, end: Optional[int] = None, stride: int = 1) -> Iterator[np.ndarray]:
        stop = self.total_frames if end is None else min(end, self.total_frames)
        for i in range(start, stop, stride):
            yield self.array[i % self.n]


def device_batch(sample):
    """If ``sample`` is a list of DeviceFrame handles of ONE clip with consecutive stored indices, return
    (DeviceBuffer view over exactly those frames, n, h, w); None for host frames.  Anything else is an error: a
    device batch must be one contiguous range (batch sizes that divide the stored clip length always are)."""
    if isinstance(sample, np.ndarray) or not len(sample) or not isinstance(sample[0], DeviceFrame):
        return None
    clip, i0 = sample[0].clip, sample[0].index
    for k, f in enumerate(sample):
        if not isinstance(f, DeviceFrame) or f.clip is not clip or f.index != i0 + k:
            raise ValueError("a batch of device-resident frames must be a contiguous range of one DeviceClip")
    n = len(sample)
    return clip.buffer.view(i0 * clip.frame_bytes, n * clip.frame_bytes), n, clip.h, clip.w


def host_batch(sample) -> np.ndarray:
    """(n, h, w, 3) uint8 array of a batch of host frames.  Frames that are consecutive in memory — views of ONE
    contiguous array: an ``ArrayClip``, a memory-mapped .npy stack, a decoder's ring — come back as a view over them: no
    staging copy, and if that memory is page-locked the upload runs straight from it; anything else is stacked."""
    if isinstance(sample, np.ndarray):
        return sample
    sample = list(sample)
    f0 = sample[0]
    if isinstance(f0, np.ndarray) and f0.flags.c_contiguous and f0.dtype == np.uint8 and f0.base is not None:
        fb, p0 = f0.nbytes, f0.ctypes.data
        root = f0.base
        if fb and all(isinstance(f, np.ndarray) and f.base is root and f.shape == f0.shape and f.flags.c_contiguous and
                      f.ctypes.data == p0 + k * fb for k, f in enumerate(sample)):
            return np.lib.stride_tricks.as_strided(f0, shape=(len(sample),) + f0.shape, strides=(fb,) + f0.strides, writeable=False)
    return np.stack(sample)


def get_video_frames_generator(source_path, stride: int = 1, start: int = 0, end: Optional[int] = None) -> Iterator[np.ndarray]:
    if isinstance(source_path, (DeviceClip, ArrayClip)):
        yield from source_path.frames(start, end, stride)
        return
    p = str(source_path)
    src = _scheme(p)
    if src is not None:
        yield from src[1](p, start, end, stride)
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
