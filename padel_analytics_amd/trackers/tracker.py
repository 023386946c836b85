"""Tracker plugin API — the outer drop-in boundary.

Mirrors the abstractions of the reference's ``trackers/tracker.py`` (``Object`` :30-63,
``TrackingResults`` :66-119, ``Tracker`` :122-330) with the same names, signatures, printed messages
and error conventions, so ``TrackingRunner`` and user code keep working unchanged:

* a tracker that works on batches raises ``NoPredictFrames`` from ``predict_frames`` and is fed lists of
  ``batch_size`` frames (last one short) by ``predict_and_update`` (reference :290-326);
* a tracker that works on the whole stream raises ``NoPredictSample`` from ``predict_sample``;
* predictions are cached as JSON through ``save_predictions`` / ``load_predictions`` (:200-241).

Additions that do not change what a caller sees:

* a tracker may split ``predict_sample`` into ``infer_sample`` (device stage, returns raw arrays) and
  ``post_sample`` (host stage, builds the objects); ``predict_and_update`` then runs the host stage of batch k
  on a worker thread while the device stage of batch k+1 is in flight (ctypes releases the GIL);
* ``predict_partial`` / ``merge_partials`` are the two halves of a prediction when the clip is sharded over GPUs
  (``TrackingRunner(distributed=True)``): the stateless per-frame part runs on the rank that owns the frames, the
  sequential part (ByteTrack ids, InpaintNet trajectory repair) on rank 0 after the gather.

Written from scratch; host logic only (the numeric engines live behind ``padel_analytics_amd.engine``).
"""
from __future__ import annotations

import sys
import threading

import json
from abc import ABC, abstractmethod
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass, field
from pathlib import Path
from typing import Iterable, Iterator, Optional, Type

import numpy as np


class NoPredictSample(Exception):
    """The tracker predicts from the frame generator, not from samples (reference tracker.py:15-20)."""


class NoPredictFrames(Exception):
    """The tracker predicts from samples, not from a frame generator (reference tracker.py:22-27)."""


class Object(ABC):
    """One frame's worth of tracked objects (players, keypoints, ball)."""

    @classmethod
    def from_json(cls, x):
        raise NotImplementedError

    def serialize(self):
        raise NotImplementedError

    def draw(self, frame: np.ndarray, **kwargs) -> np.ndarray:
        """Annotation is outside the hot path (OUT OF SCOPE); frames pass through."""
        return frame


@dataclass
class TrackingResults:
    predictions: list = field(default_factory=list)
    sample_predictions: list = field(default_factory=list)
    counter: int = 0

    def load(self, predictions: list) -> None:
        self.predictions = predictions
        self.sample_predictions = []
        self.counter = 0

    def update(self, predictions: list) -> None:
        self.predictions += predictions
        self.sample_predictions = predictions
        self.counter += 1

    def restart(self) -> None:
        self.predictions = []
        self.sample_predictions = []
        self.counter = 0

    def __len__(self) -> int:
        return len(self.predictions)

    def __getitem__(self, i: int):
        return self.predictions[i]

    def __iter__(self) -> Iterator:
        return iter(self.predictions)


def _sampler(generator: Iterable[np.ndarray], sequence_length: int) -> Iterator[list]:
    """Chop a frame stream into lists of ``sequence_length`` frames; the last one may be short."""
    w: list = []
    for x in generator:
        w.append(x)
        if len(w) == sequence_length:
            yield w
            w = []
    if w:
        yield w


class relaxed_gc:
    """While a batch loop builds result objects by the ten thousand (hundreds of Player / PlayerKeypoints per frame on a dense
    scene), CPython's cyclic collector — a young collection every 700 allocations, full collections that walk every result
    object alive so far (110-150 ms pauses with the GIL held once a clip's results have accumulated) — cost 3 x the
    construction itself and stalled the thread that feeds the GPU (measured: 34.7 -> 12.3 ms of host time per 64-frame step,
    tools/eager_objects_bench.py).  The result objects are acyclic: reference counting frees them.  Inside the context the
    young-generation threshold is raised (never lowered); the previous thresholds come back on exit."""

    YOUNG = 100_000
    _lock = threading.Lock()
    _depth = 0                      # contexts alive in the process (any thread): the first one in raises, the last one out restores
    _saved = None

    def __enter__(self):
        import gc
        cls = relaxed_gc
        with cls._lock:
            if cls._depth == 0:
                cls._saved = gc.get_threshold()
                gc.set_threshold(max(cls._saved[0], cls.YOUNG), *cls._saved[1:])
            cls._depth += 1
        return self

    def __exit__(self, *exc):
        import gc
        cls = relaxed_gc
        with cls._lock:
            cls._depth -= 1
            if cls._depth == 0 and cls._saved is not None:
                gc.set_threshold(*cls._saved)
                cls._saved = None
        return False


def _own_arrays(raw):
    """A device stage's result with its arrays copied (they may be a model's recycled page-locked sets)."""
    if isinstance(raw, np.ndarray):
        return raw.copy()
    if isinstance(raw, (tuple, list)):
        return type(raw)(_own_arrays(x) for x in raw)
    return raw


def pack_ragged(items: list) -> Optional[list]:
    """[ndarray (k_i, ...)] with one dtype / trailing shape -> [counts (n,) int32, rows (sum k_i, ...)]; None if the items
    are anything else."""
    if not all(isinstance(x, np.ndarray) and x.ndim >= 1 for x in items):
        return None
    if not items:
        return [np.zeros(0, np.int32), np.zeros((0,), np.float32)]
    tail, dt = items[0].shape[1:], items[0].dtype
    if any(x.shape[1:] != tail or x.dtype != dt for x in items):
        return None
    return [np.array([len(x) for x in items], np.int32), np.concatenate(items, axis=0) if items else np.zeros((0,) + tail, dt)]


def unpack_ragged(arrays: list) -> list:
    counts, rows = arrays
    if len(counts) == 0:
        return []
    return np.split(rows, np.cumsum(counts)[:-1])


class Tracker(ABC):
    batch_size: int
    #: True while a loop that bounds the number of live result sets runs (``_predict_batches``): ``infer_sample`` may then ask
    #: the engine for its recycled output arrays
    _reuse_outputs: bool = False
    #: frames of temporal context a shard needs before / after its own frames (TrackNet windows: 7 / 7)
    temporal_context: tuple = (0, 0)
    #: True for trackers that consume the whole stream in predict_frames (TrackNet), False for batch trackers
    streams: Optional[bool] = None

    def __init__(self, load_path: Optional[str | Path] = None, save_path: Optional[str | Path] = None) -> None:
        self.results = TrackingResults()
        self.load_path = load_path
        self.save_path = save_path
        self.load_predictions()

    @abstractmethod
    def video_info_post_init(self, video_info) -> "Tracker": ...

    @abstractmethod
    def object(self) -> Type[Object]: ...

    @abstractmethod
    def draw_kwargs(self) -> dict: ...

    @property
    def DEVICE(self) -> str:
        """"cuda" when the HIP engine sees a GPU (ROCm devices are "cuda" to callers), else "cpu".
        There is no CPU execution path: on "cpu" the numeric calls raise EngineUnavailable."""
        from .. import engine
        try:
            return "cuda" if engine.load_library().pa_device_count() > 0 else "cpu"
        except engine.EngineUnavailable:
            return "cpu"

    @abstractmethod
    def restart(self) -> None: ...

    def __len__(self) -> int:
        return len(self.results)

    @abstractmethod
    def __str__(self) -> str: ...

    def save_predictions(self) -> None:
        if self.save_path:
            print(f"{self.__str__()}: Saving predictions ...")
            parsable = [obj.serialize() for obj in self.results.predictions]
            with open(self.save_path, "w") as f:
                json.dump(parsable, f)
            print(f"{self.__str__()}: {self.__len__()} predictions saved.")

    def load_predictions(self) -> None:
        if self.load_path:
            print(f"{self.__str__()}: Loading predictions ...")
            with open(self.load_path, "r") as f:
                parsable = json.load(f)
            self.results.load([self.object().from_json(o) for o in parsable])
        print(f"{self.__str__()}: {self.__len__()} predictions loaded.")

    def to(self, device: str) -> None:
        """Move the tracker's model(s) to ``device`` (weights go to HBM on "cuda")."""

    # ---- arithmetic of the fp32-equivalent path (engine.fp32_mode): the default "h2" covers |x| <= 65504 and switches a
    # model to the full-range "bx3" kernels when an activation leaves that range.  A sharded run must take that decision
    # for ALL ranks at once (TrackingRunner._predict_sharded), so trackers expose it
    @property
    def full_range(self) -> bool:
        """True when the tracker's model(s) cannot overflow any more (bx3 / fp16 models, or nothing to switch)."""
        m = getattr(self, "model", None)
        if m is not None and hasattr(m, "fp32_mode"):
            return bool(getattr(m, "half", False)) or m.fp32_mode != "h2"
        return True

    def use_full_range(self) -> None:
        """Put the tracker on the full-range arithmetic (rebuilds the packed graph; the HBM model follows on next use)."""
        m = getattr(self, "model", None)
        if m is not None and hasattr(m, "set_fp32_mode"):
            m.set_fp32_mode("bx3")

    @abstractmethod
    def predict_sample(self, sample: Iterable[np.ndarray], **kwargs) -> Optional[list]: ...

    @abstractmethod
    def predict_frames(self, frame_generator: Iterable[np.ndarray], **kwargs) -> Optional[list]: ...

    # ---- optional two-stage form of predict_sample (device stage / host stage)
    def infer_sample(self, sample: list, **kwargs):
        raise NotImplementedError

    def post_sample(self, raw, **kwargs) -> list:
        raise NotImplementedError

    # ---- optional two-call form of infer_sample: submit_sample enqueues the device stage and returns a token at once (None:
    # not possible for this sample — use infer_sample), collect_sample(token) returns what infer_sample would have
    def submit_sample(self, sample: list, **kwargs):
        return None

    def collect_sample(self, token):
        raise NotImplementedError

    def discard_sample(self, token) -> None:
        """Give up a submitted batch (its results are not wanted): only releases the ticket (``YOLO.discard_frames``)."""
        model = getattr(self, "model", None)
        if hasattr(model, "discard_frames"):
            model.discard_frames(token)
        else:
            self.collect_sample(token)

    def _has_stages(self) -> bool:
        return type(self).infer_sample is not Tracker.infer_sample and type(self).post_sample is not Tracker.post_sample

    #: host stages a batch loop may have queued behind its device stage (``TrackingRunner`` raises it: the host stage of a
    #: detector with many tracks is slower than its device stage, and what queues up drains beside the NEXT tracker's device work)
    host_queue_depth = 1

    def _predict_batches(self, frame_generator, update, defer: Optional[list] = None, **kwargs) -> None:
        """Batch loop of reference tracker.py:319-326.  With the two-stage form, post_sample(batch k) runs on one
        worker thread (in batch order) while infer_sample(batch k+1) occupies the GPU.

        ``defer`` (a list): host stages still queued when the device loop ends are not waited for here — a ``finish()`` callable
        that collects them (in order, through ``update``) is appended instead, for the caller to run once the next tracker's
        device work is under way.  Results are identical either way: one worker, batch order."""
        if not self._has_stages():
            for sample in _sampler(frame_generator, self.batch_size):
                update(self.predict_sample(sample, **kwargs))
            return
        # The device stage blocks inside the C library with the GIL released; when it returns, this thread has to take the
        # GIL back from the worker, which hands it over only every sys.getswitchinterval() (5 ms by default — the length of
        # a whole device stage of the small models): a short interval for the duration of the loop
        interval = sys.getswitchinterval()
        sys.setswitchinterval(min(interval, 2e-4))
        # at most three result sets are alive here (the batch in the host stage, the submitted batch being collected, the batch
        # submitted behind it): the device stage may hand out the model's recycled page-locked arrays (engine.Model.OUT_RING).
        # A deeper host queue owns copies of what it holds.
        self._reuse_outputs = True
        depth = max(1, int(self.host_queue_depth))
        submitted = None                                    # the token of the batch that is queued on the GPU and not collected yet
        pool = ThreadPoolExecutor(max_workers=1)
        pending = []
        handed_over = False
        try:
            with relaxed_gc():
                def host_stage(raw):
                    if depth > 1:
                        raw = _own_arrays(raw)
                    pending.append(pool.submit(self.post_sample, raw, **kwargs))
                    while len(pending) > depth:             # keep `depth` host stages queued behind the device stage
                        update(pending.pop(0).result())

                # Trackers with the two-call device stage (submit_sample / collect_sample: pa_yolo_submit / pa_yolo_wait) have
                # batch k + 1 queued on the GPU before batch k is collected; the others run one synchronous call per batch.
                for sample in _sampler(frame_generator, self.batch_size):
                    token = self.submit_sample(sample, **kwargs)
                    if submitted is not None:
                        done, submitted = submitted, token       # (what is in flight is always in `submitted`: the finally drains it)
                        host_stage(self.collect_sample(done))
                        if token is None:
                            host_stage(self.infer_sample(sample, **kwargs))
                        continue
                    if token is None:
                        host_stage(self.infer_sample(sample, **kwargs))
                    else:
                        submitted = token
                if submitted is not None:
                    token, submitted = submitted, None
                    host_stage(self.collect_sample(token))

                def finish():
                    try:
                        with relaxed_gc():
                            while pending:
                                update(pending.pop(0).result())
                    finally:
                        pool.shutdown(wait=True, cancel_futures=True)

                if defer is not None and depth > 1 and pending:
                    defer.append(finish)
                    handed_over = True
                else:
                    finish()
        finally:
            self._reuse_outputs = False
            sys.setswitchinterval(interval)
            self._drain(submitted)
            if not handed_over:
                pool.shutdown(wait=True, cancel_futures=True)

    def _raw_batches(self, frame_generator, **kwargs):
        """Yield ``infer_sample``'s result for every batch of the stream, in order — with the next batch already submitted
        (``submit_sample``) where the tracker has the two-call device stage.  For loops that finish with a batch's arrays
        before asking for the next one (sharded ``predict_partial``): at most two result sets are alive."""
        self._reuse_outputs = True
        submitted = None
        try:
            for sample in _sampler(frame_generator, self.batch_size):
                token = self.submit_sample(sample, **kwargs)
                if submitted is not None:
                    done, submitted = submitted, token       # (the new token is tracked before control leaves this frame)
                    token = None
                    yield self.collect_sample(done)
                    if submitted is None:
                        yield self.infer_sample(sample, **kwargs)
                    continue
                if token is None:
                    yield self.infer_sample(sample, **kwargs)
                else:
                    submitted = token
            if submitted is not None:
                token, submitted = submitted, None
                yield self.collect_sample(token)
        finally:
            self._reuse_outputs = False
            self._drain(submitted)

    def _drain(self, token) -> None:
        """A loop that ends early (an exception in a host stage, a consumer that stops iterating) must not leave its submitted
        batch uncollected: the ticket would stay in flight on the model (PA_MAX_INFLIGHT of them block further submits)."""
        if token is not None:
            try:
                self.discard_sample(token)
            except Exception as exc:                         # the loop is already unwinding: report, do not mask what ended it
                print(f"{self.__str__()}: could not release the batch still in flight ({exc!r})")

    def predict_and_update(self, frame_generator: Iterable[np.ndarray], defer: Optional[list] = None, **kwargs) -> TrackingResults:
        """``defer``: see ``_predict_batches`` — the results are complete once every callable appended to it has run."""
        tail = [] if defer is not None else None
        try:
            predictions = self.predict_frames(frame_generator, **kwargs)
            self.results.predictions = predictions
        except NoPredictFrames:
            self._predict_batches(frame_generator, self.results.update, defer=tail, **kwargs)
        if tail:
            def finish():
                for f in tail:
                    f()
                print(f"{self.__str__()}: {len(self.results)} predictions.")
            defer.append(finish)
        else:
            print(f"{self.__str__()}: {len(self.results)} predictions.")
        return self.results

    # ---- sharded prediction: frames [first, first + n) of the clip live on this rank
    def predict_partial(self, frame_generator: Iterable[np.ndarray], *, first_frame: int = 0, head_context: int = 0,
                        tail_context: int = 0, **kwargs) -> list:
        """Stateless part of the prediction for one contiguous shard.  ``frame_generator`` yields the shard's
        frames preceded by ``head_context`` and followed by ``tail_context`` frames of temporal context (zero for
        per-frame trackers).  Returns one picklable item per OWNED frame, in frame order."""
        assert head_context == 0 and tail_context == 0, "per-frame trackers need no temporal context"
        out: list = []
        try:
            out = self.predict_frames(frame_generator, **kwargs)
        except NoPredictFrames:
            self._predict_batches(frame_generator, out.extend, **kwargs)
        return out

    #: ``merge_partials`` touches neither the GPU nor shared state of other trackers: the sharded runner may run it on a
    #: worker thread of rank 0 while that rank's GPU already works on the next tracker's shard (False: run it in line)
    merge_is_host_only = True

    def merge_partials(self, partials: list, **kwargs) -> list:
        """Sequential part, on rank 0, over the partials of ALL frames in global frame order -> final objects."""
        return partials

    # ---- the partials on the wire: arrays, not pickled objects (the gather is then O(bytes): dist.gather_arrays)
    def pack_partials(self, partials: list) -> Optional[list]:
        """-> a list of ndarrays that ``unpack_partials`` turns back into the same partials, or None (the runner then falls
        back to one pickled buffer per rank).  Default: per-frame items that are all ndarrays of one dtype and trailing shape
        travel as (counts, rows)."""
        return pack_ragged(partials)

    def unpack_partials(self, arrays: list) -> list:
        return unpack_ragged(arrays)
