"""TrackingRunner — drop-in for the tracker loop of the reference's ``trackers/runner.py``
(``__init__`` :37-79, ``run`` :175-236): trackers run one after the other over the whole clip, each is
moved to the device, timed with ``timeit`` exactly where the reference times it (:222-228), moved back,
and its predictions are cached as JSON; a tracker with cached predictions is skipped (:187-191).

Additions (all off by default, the reference's sequential semantics stay the default):

* ``run()`` records ``self.timings`` (seconds / frames per tracker) and prints frames/s;
* ``distributed=True``: one process per GPU under ``torch.distributed``; every tracker's frames
  are split into contiguous shards ``dist.shard_range`` (plus the tracker's ``temporal_context`` — TrackNet's 7-frame
  halo), each rank runs the stateless part on its shard (``Tracker.predict_partial``), the per-frame partials are
  gathered to rank 0 in frame order and the sequential part (ByteTrack ids, InpaintNet) runs there
  (``Tracker.merge_partials``).  The ball tracker's background median is computed once on rank 0 and broadcast.
  Results and JSON caches live on rank 0;
* ``fanout=True``: ONE pass over the video feeds every tracker — each batch of frames is
  uploaded to HBM once (the next batch uploads on the engine's copy stream while this one computes) and all
  batch trackers consume the same device-resident frames; stream trackers (TrackNet) get the same handles.  The
  reference decodes and uploads the clip once per tracker (:215-220).

``draw_and_collect_data`` (video encode, homography, analytics: SURVEY.md §1 L3') is out of scope."""
from __future__ import annotations

import threading
import os
import timeit
from pathlib import Path
from typing import Optional

import numpy as np

from .. import dist as D, video
from .tracker import NoPredictFrames, Tracker, _sampler


class TrackingRunner:
    def __init__(self, trackers: list, video_path: str | Path, inference_path: str | Path, start: int = 0,
                 end: Optional[int] = None, collect_data: bool = False, *, distributed: bool = False,
                 fanout: bool = False, engine=None, host_queue_depth: Optional[int] = None) -> None:
        self.video_path = video_path
        self.inference_path = inference_path
        self.start = start
        self.stride = 1
        self.end = end
        self.video_info = video.VideoInfo.from_video_path(video_path)
        self.total_frames = self.video_info.total_frames if end is None else end - start
        # frames actually available from `start` on (the reference reports the whole clip's length when end is None,
        # runner.py:52; the sharded path must split what the generator really yields)
        # (an `end` beyond the clip is clamped: the shards must add up to what the generator yields, or the last ranks come
        # up short and the others wait in the gather for ever)
        self.n_available = max(0, (self.video_info.total_frames if end is None else min(end, self.video_info.total_frames)) - start)
        self.trackers = {}
        for tracker in trackers:
            self.trackers[str(tracker)] = tracker.video_info_post_init(self.video_info)
        self.collect_data = collect_data
        self.data_analytics = None
        self.timings: dict = {}
        self.distributed = distributed
        self.fanout = fanout
        self.engine = engine
        # host stages (zone, ByteTrack, result objects) a batch tracker may have queued behind its device stage; what is still
        # queued when its device loop ends drains beside the NEXT tracker's device work (1: the loop waits for its own tail)
        self.host_queue_depth = int(os.environ.get("PADEL_HOST_QUEUE_DEPTH", 8)) if host_queue_depth is None else host_queue_depth
        self._tails: list = []

    def restart(self) -> None:
        for tracker in self.trackers.values():
            tracker.restart()

    def draw_and_collect_data(self) -> None:
        print("runner: drawing / data collection is outside the hot path of this build (skipped)")

    def _frames(self, lo: int = 0, hi: Optional[int] = None):
        """Frames [start + lo, start + hi) of the clip (hi None: to self.end)."""
        end = self.end if hi is None else self.start + hi
        return video.get_video_frames_generator(self.video_path, start=self.start + lo, stride=self.stride, end=end)

    def run(self) -> None:
        print(f"runner: Running {self.total_frames} frames")
        self._merges = []                  # sharded mode, rank 0: sequential stages still running behind the next tracker's shard
        try:
            if self.fanout:
                self._run_fanout()
            else:
                for tracker in self.trackers.values():
                    # results (and the prediction caches) live on rank 0 only: its decision to skip a tracker is the one
                    # every rank follows, or the others would wait in the sharded path's collectives for ever
                    stored = D.broadcast_flag(len(tracker) != 0) if self.distributed else len(tracker) != 0
                    if stored:
                        print(f"{tracker.__str__()}: {len(tracker)} predictions stored")
                        continue
                    tracker.to(tracker.DEVICE)
                    print(f"{str(tracker)}: Running on {tracker.DEVICE} ...")
                    t0 = timeit.default_timer()
                    tail: list = []
                    if self.distributed:
                        self._predict_sharded(tracker)
                    else:
                        tracker.host_queue_depth = max(1, int(self.host_queue_depth))
                        self._predict(tracker, defer=tail)
                    t1 = timeit.default_timer()
                    tracker.to("cpu")
                    if self._merges and self._merges[-1][0] is tracker:
                        self._merges[-1] += (t0, t1)      # reported when the merge is done: shard time and the merge's own span apart
                        continue
                    # the trackers before this one had this tracker's whole device loop to finish their queued host stages
                    self._join_tails()
                    if tail:
                        self._tails.append((tracker, tail, t0, t1))
                        continue
                    self._report(tracker, t0, t1)
                    if not self.distributed or D.rank() == 0:
                        tracker.save_predictions()
        finally:
            #  a tracker that raises must not leave finished merges unsaved or the worker pool alive; a merge's own
            # exception surfaces here, after the merges before it were saved
            try:
                self._join_tails()
            finally:
                self._join_merges()
        self.draw_and_collect_data()

    def _predict(self, tracker: Tracker, defer: Optional[list] = None) -> None:
        """One tracker over the whole clip.  A stream tracker on the h2 arithmetic whose activations left the fp16
        range has switched itself to the full-range path and asks to be run again (engine.RangeOverflow); batch
        trackers repeat the offending batch themselves (yolo.YOLO.infer_frames).  ``defer``: Tracker._predict_batches."""
        from .. import engine as E
        try:
            tracker.predict_and_update(self._frames(), defer=defer, total_frames=self.total_frames)
        except E.RangeOverflow as ex:
            print(f"{str(tracker)}: {ex}")
            tracker.restart()
            tracker.to(tracker.DEVICE)
            tracker.predict_and_update(self._frames(), defer=defer, total_frames=self.total_frames)

    def _join_tails(self) -> None:
        """Collect the host stages that were still queued when their trackers' device loops ended.  ``timings``: "seconds" is the
        device loop's span, "host_tail_seconds" what was left to wait for here — after the next tracker's loop, normally nothing."""
        tails, self._tails = self._tails, []
        first_error = None
        for tracker, fins, t0, t1 in tails:
            ts = timeit.default_timer()
            try:
                for f in fins:
                    f()
            except BaseException as exc:           # keep collecting: the tails behind this one are finished work too
                first_error = first_error or exc
                continue
            self._report(tracker, t0, t1)
            self.timings[str(tracker)].update(host_tail_seconds=timeit.default_timer() - ts, host_tail_overlapped=True)
            if not self.distributed or D.rank() == 0:
                tracker.save_predictions()
        if first_error is not None:
            raise first_error

    def _report(self, tracker, t0, t1) -> None:
        print(f"{str(tracker)}: {t1 - t0} inference time.")
        n = len(tracker)
        self.timings[str(tracker)] = {"seconds": t1 - t0, "frames": n}
        if t1 > t0:
            print(f"{str(tracker)}: {n / (t1 - t0):.1f} frames/s")

    # ------------------------------------------------------------------ sharded over GPUs
    def _predict_sharded(self, tracker: Tracker) -> None:
        rank, world = D.rank(), D.world_size()
        n = self.n_available
        lo, hi = D.shard_range(n, rank, world)
        ch, ct = tracker.temporal_context
        head, tail = min(ch, lo), min(ct, n - hi)
        if ch or ct:
            self._share_background(tracker)
        from .. import engine as E

        def attempt():
            if hi <= lo:
                return [], False
            try:
                return tracker.predict_partial(self._frames(lo - head, hi + tail), first_frame=lo, head_context=head,
                                               tail_context=tail, total_frames=hi - lo), False
            except E.RangeOverflow as ex:                  # a stream tracker switched itself to the full-range kernels
                print(f"{str(tracker)}: {ex}")
                return None, True

        was_full = tracker.full_range
        partial, over = attempt()
        # h2 -> bx3 is decided for ALL ranks at once: a rank whose shard overflowed (stream trackers raise, batch trackers
        # switch inside the offending batch) must not leave the others waiting in the gather, and the merged result must
        # come from one arithmetic.  Every rank enters this collective
        if D.any_flag(over or (not was_full and tracker.full_range)):
            print(f"{str(tracker)}: activations left the fp16 range on some rank — every rank repeats its shard on the bf16x3 path")
            tracker.use_full_range()
            tracker.to(tracker.DEVICE)
            partial, over = attempt()
            assert not over
        assert len(partial) == hi - lo, (str(tracker), len(partial), lo, hi)
        # the partials travel as arrays (counts + rows per rank: two collectives, O(bytes)); trackers whose partials are not
        # arrays fall back to one pickled buffer per rank.  Every rank must take the same branch: agreed by a collective
        packed = tracker.pack_partials(partial)
        if not D.any_flag(packed is None):
            parts = D.gather_arrays(packed, dst=0)
            allp = None if parts is None else [x for a in parts for x in tracker.unpack_partials(a)]
        else:
            allp = D.gather_results(partial, dst=0)
        if rank == 0:
            def merge():
                from .tracker import relaxed_gc
                ts = timeit.default_timer()
                with relaxed_gc():
                    tracker.results.predictions = tracker.merge_partials(allp)
                print(f"{tracker.__str__()}: {len(tracker.results)} predictions.")
                return ts, timeit.default_timer()
            # The sequential stage (ByteTrack ids over ALL frames in global order: ~60 us per frame of host C++) is rank 0's
            # alone.  Where it touches no GPU it runs on a worker thread while rank 0's GPU starts on the next tracker's shard —
            # otherwise the other ranks would wait for it at the next gather; run() joins before it returns
            if tracker.merge_is_host_only and world > 1:
                if getattr(self, "_merge_pool", None) is None:
                    from concurrent.futures import ThreadPoolExecutor
                    self._merge_pool = ThreadPoolExecutor(max_workers=1)
                self._merges.append((tracker, self._merge_pool.submit(merge)))
            else:
                merge()

    def _join_merges(self) -> None:
        """Collect rank 0's deferred sequential stages.  ``timings``: "seconds" is the tracker's own shard (device stage + gather),
        "merge_seconds" the merge's own span, which ran BESIDE the next tracker's shard ("merge_overlapped") — the two are not
        added: the next tracker's "seconds" already covers that wall time (no double counting)."""
        merges, self._merges = getattr(self, "_merges", []), []
        pool, self._merge_pool = getattr(self, "_merge_pool", None), None
        first_error = None
        try:
            for item in merges:
                tracker, fut = item[0], item[1]
                try:
                    ts, te = fut.result()
                except BaseException as exc:       # keep collecting: the merges behind this one are finished work
                    first_error = first_error or exc
                    continue
                if len(item) >= 4:
                    self._report(tracker, item[2], item[3])
                    self.timings[str(tracker)].update(merge_seconds=te - ts, merge_overlapped=True)
                else:                               # the tracker loop was left before its bookkeeping: report the merge alone
                    self._report(tracker, ts, te)
                tracker.save_predictions()
        finally:
            if pool is not None:
                pool.shutdown()
        if first_error is not None:
            raise first_error

    def _share_background(self, tracker) -> None:
        """TrackNet's background median (iterable.py:59-81) is a property of the clip's first frames: rank 0
        computes it (on its GPU), every rank receives the (h, w, 3) uint8 array."""
        if getattr(tracker, "median", None) is not None or not hasattr(tracker, "compute_median"):
            return
        med = None
        if D.rank() == 0:
            nmed = min(tracker.median_max_sample_num, self.n_available)
            med = tracker.compute_median(list(self._frames(0, nmed)))
        tracker.median = D.broadcast_array(med, (self.video_info.height, self.video_info.width, 3), np.uint8, src=0)

    # ------------------------------------------------------------------ one decode / one upload for all trackers
    def _run_fanout(self) -> None:
        todo = [t for t in self.trackers.values() if len(t) == 0]
        for t in self.trackers.values():
            if len(t) != 0:
                print(f"{t.__str__()}: {len(t)} predictions stored")
        if not todo:
            return
        batch = [t for t in todo if self._is_batch_tracker(t)]
        stream = [t for t in todo if t not in batch]
        bs = max([t.batch_size for t in batch] or [max(t.batch_size for t in stream)])
        for t in todo:
            t.to(t.DEVICE)
            print(f"{str(t)}: Running on {t.DEVICE} (fan-out, {bs} frames per upload) ...")
        t0 = timeit.default_timer()
        from concurrent.futures import ThreadPoolExecutor

        def batches():
            """Batches of frames resident in HBM: device clips pass through; host frames are uploaded once per
            batch into one of two staging clips, the next upload overlapping this batch's compute."""
            gen = self._frames()
            first = next(gen, None)
            if first is None:
                return
            if isinstance(first, video.DeviceFrame):
                def chain():
                    yield first
                    yield from gen
                yield from _sampler(chain(), bs)
                return
            from .. import engine as E
            eng = self.engine or E.default_engine()
            h, w = first.shape[:2]
            clips = [video.DeviceClip(eng, shape=(bs, h, w, 3)) for _ in range(2)]

            def chain():
                yield first
                yield from gen

            def stage(k, frames):
                c = clips[k % len(clips)]
                c.upload(video.host_batch(frames), copy_stream=True)
                return [video.DeviceFrame(c, i) for i in range(len(frames))]

            with ThreadPoolExecutor(max_workers=1) as up:
                fut, k = None, 0
                for frames in _sampler(chain(), bs):
                    nxt = up.submit(stage, k, frames)
                    k += 1
                    if fut is not None:
                        yield fut.result()
                    fut = nxt
                if fut is not None:
                    yield fut.result()
            for c in clips:
                c.free()

        import sys
        interval = sys.getswitchinterval()                 # see Tracker._predict_batches: the device stage must get the GIL back quickly
        sys.setswitchinterval(min(interval, 2e-4))
        from .tracker import relaxed_gc
        try:
            with relaxed_gc(), ThreadPoolExecutor(max_workers=1) as post:
                pending = []
                for sample in batches():
                    for t in batch:
                        for sub in _sampler(iter(sample), t.batch_size):
                            if t._has_stages():
                                raw = t.infer_sample(sub)
                                pending.append((t, post.submit(t.post_sample, raw)))
                            else:
                                t.results.update(t.predict_sample(sub))
                    while len(pending) > len(batch):           # host stages trail the device stages by one batch
                        t, f = pending.pop(0)
                        t.results.update(f.result())
                for t, f in pending:
                    t.results.update(f.result())
        finally:
            sys.setswitchinterval(interval)
        # stream trackers (TrackNet needs the clip's background median before its first window): their own pass —
        # over the same HBM-resident handles when the clip lives in HBM, else a second read of the source
        for t in stream:
            self._predict(t)
        t1 = timeit.default_timer()
        for t in todo:
            t.to("cpu")
            print(f"{t.__str__()}: {len(t.results)} predictions.")
            t.save_predictions()       # (the trackers share every upload: only the combined time below is meaningful)
        self.timings["__fanout__"] = {"seconds": t1 - t0, "frames": self.total_frames}
        print(f"runner: fan-out pass {t1 - t0} inference time, {self.total_frames / max(t1 - t0, 1e-9):.1f} frames/s (all trackers)")

    @staticmethod
    def _is_batch_tracker(t: Tracker) -> bool:
        """Batch trackers answer predict_frames with NoPredictFrames (reference tracker.py:315-326)."""
        if getattr(t, "streams", None) is not None:
            return not t.streams
        try:
            t.predict_frames(iter(()))
        except NoPredictFrames:
            return True
        except Exception:
            return False
        return False
