"""Tracker._predict_batches (host logic, no GPU): the device stage in two halves — batch k + 1 submitted before batch k is
collected — keeps batch order, bounds the number of live result sets, mixes with batches that have to take the synchronous
call, and a tracker without the two-call form runs as before."""
import numpy as np

from padel_analytics_amd.trackers.tracker import NoPredictFrames, Tracker


class _Toy(Tracker):
    """Frames are integers; a 'detection' is frame * 10.  submit_sample declines every batch whose first frame is in `sync`."""
    batch_size = 4
    streams = False

    def __init__(self, two_call=True, sync=()):
        super().__init__()
        self.two_call, self.sync = two_call, set(sync)
        self.log, self.live, self.max_live = [], 0, 0

    def video_info_post_init(self, video_info): return self
    def object(self): return int
    def draw_kwargs(self): return {}
    def __str__(self): return "toy"
    def restart(self): self.results.restart()
    def predict_sample(self, sample, **kw): return self.post_sample(self.infer_sample(sample))
    def predict_frames(self, frame_generator, **kw): raise NoPredictFrames()

    def _take(self):
        self.live += 1
        self.max_live = max(self.max_live, self.live)

    def infer_sample(self, sample, **kw):
        assert self._reuse_outputs
        self.log.append(("infer", sample[0]))
        self._take()
        return list(sample)

    def submit_sample(self, sample, **kw):
        if not self.two_call or sample[0] in self.sync:
            return None
        self.log.append(("submit", sample[0]))
        self._take()
        return {"sample": list(sample)}

    def collect_sample(self, token):
        self.log.append(("collect", token["sample"][0]))
        return token["sample"]

    def post_sample(self, raw, **kw):
        out = [f * 10 for f in raw]
        self.live -= 1
        return out


def test_two_call_loop_keeps_order_and_submits_ahead():
    t = _Toy()
    t.predict_and_update(iter(range(18)))                      # 4 + 4 + 4 + 4 + 2
    assert t.results.predictions == [f * 10 for f in range(18)]
    ev = t.log
    assert ev[:3] == [("submit", 0), ("submit", 4), ("collect", 0)]            # batch 1 is queued before batch 0 is collected
    assert [e for e in ev if e[0] == "collect"] == [("collect", f) for f in (0, 4, 8, 12, 16)]
    for k in (4, 8, 12, 16):
        assert ev.index(("submit", k)) < ev.index(("collect", k - 4))
    assert t.max_live <= 3 and not t._reuse_outputs            # what engine.Model.OUT_RING = 4 has to cover


def test_batches_that_decline_the_two_call_form_run_synchronously_in_place():
    t = _Toy(sync={4, 12})
    t.predict_and_update(iter(range(20)))
    assert t.results.predictions == [f * 10 for f in range(20)]
    assert ("infer", 4) in t.log and ("infer", 12) in t.log and ("submit", 4) not in t.log
    assert t.log.index(("collect", 0)) < t.log.index(("infer", 4)) < t.log.index(("submit", 8))
    assert t.max_live <= 3


def test_tracker_without_the_two_call_form():
    t = _Toy(two_call=False)
    t.predict_and_update(iter(range(9)))
    assert t.results.predictions == [f * 10 for f in range(9)]
    assert [e[0] for e in t.log] == ["infer"] * 3


def test_raw_batches_for_sharded_loops():
    t = _Toy(sync={8})
    got = []
    for raw in t._raw_batches(iter(range(14))):
        got += raw
        t.live -= 1                                           # the consumer is done with the batch before asking for the next
    assert got == list(range(14)) and t.max_live <= 2 and not t._reuse_outputs
    assert t.log.index(("submit", 4)) < t.log.index(("collect", 0)) < t.log.index(("infer", 8)) < t.log.index(("submit", 12))


def test_a_failing_host_stage_does_not_leave_a_batch_in_flight():
    class Boom(_Toy):
        def post_sample(self, raw, **kw):
            if raw[0] == 4:
                raise RuntimeError("host stage failed")
            return super().post_sample(raw)

    t = Boom()
    try:
        t.predict_and_update(iter(range(20)))
        raise AssertionError("the host stage's exception must surface")
    except RuntimeError as ex:
        assert "host stage failed" in str(ex)
    submitted = [e[1] for e in t.log if e[0] == "submit"]
    collected = [e[1] for e in t.log if e[0] == "collect"]
    assert submitted and sorted(submitted) == sorted(collected), (submitted, collected)       # every ticket was waited for
    assert not t._reuse_outputs


def test_a_consumer_that_stops_early_drains_the_look_ahead():
    t = _Toy()
    it = t._raw_batches(iter(range(20)))
    assert next(it) == [0, 1, 2, 3]
    it.close()                                                 # e.g. an exception in the sharded loop's body
    submitted = [e[1] for e in t.log if e[0] == "submit"]
    collected = [e[1] for e in t.log if e[0] == "collect"]
    assert submitted == [0, 4] and sorted(collected) == [0, 4] and not t._reuse_outputs


def test_drain_releases_the_ticket_without_a_fresh_inference():
    """the early-exit path gives the submitted batch up through ``discard_sample`` -> ``YOLO.discard_frames`` — a
    wait on the ticket when its model is still the live one, nothing at all when that model was replaced or closed; never
    ``collect_frames`` (which would run a whole inference, re-creating the HBM model, to throw the result away)."""
    from padel_analytics_amd import yolo as Y

    class _M:
        handle = 1
        def __init__(self): self.waited = []
        def yolo_wait(self, ticket): self.waited.append(ticket); return None, None, None, False

    class _Y:
        discard_frames = Y.YOLO.discard_frames
        def __init__(self): self._model = _M()
        def collect_frames(self, token): raise AssertionError("the drain must not collect")
        def infer_frames(self, *a, **k): raise AssertionError("the drain must not infer")

    class WithModel(_Toy):
        def __init__(self):
            super().__init__()
            self.model = _Y()
        def submit_sample(self, sample, **kw):
            tok = super().submit_sample(sample, **kw)
            tok.update(ticket=("t", sample[0]), model=self.model._model)
            return tok

    t = WithModel()
    live = t.model._model
    it = t._raw_batches(iter(range(20)))
    assert next(it) == [0, 1, 2, 3]
    it.close()
    assert live.waited == [("t", 4)]                           # the look-ahead ticket was waited for, nothing else ran
    t2 = WithModel()
    it = t2._raw_batches(iter(range(20)))
    next(it)
    old, t2.model._model = t2.model._model, _M()               # the overflow fallback replaced the model in between
    it.close()
    assert old.waited == [] and t2.model._model.waited == []
    t3 = WithModel()
    it = t3._raw_batches(iter(range(20)))
    next(it)
    t3.model._model.handle = None                              # ... or closed it
    it.close()
    assert t3.model._model.waited == []


def test_relaxed_gc_is_reentrant_across_threads_and_restores_the_thresholds():
    """The sharded runner's merge thread and a batch loop on the main thread may both be inside ``relaxed_gc``: the first one in
    raises the young-generation threshold, the LAST one out restores what the first one found."""
    import gc
    import threading
    from padel_analytics_amd.trackers.tracker import relaxed_gc
    before = gc.get_threshold()
    inside, leave = threading.Event(), threading.Event()

    def worker():
        with relaxed_gc():
            inside.set()
            leave.wait(10)

    t = threading.Thread(target=worker)
    t.start()
    inside.wait(10)
    with relaxed_gc():
        assert gc.get_threshold()[0] >= relaxed_gc.YOUNG
    assert gc.get_threshold()[0] >= relaxed_gc.YOUNG          # the worker is still inside: not restored yet
    leave.set()
    t.join()
    assert gc.get_threshold() == before


class _Slow(_Toy):
    """The host stage is the slow one (a detector with many tracks): arrays out of a two-set 'ring' that the device stage
    overwrites on every other call, like engine.Model's recycled page-locked result sets."""

    def __init__(self, delay=0.01):
        super().__init__()
        self.delay = delay
        self.ring = [np.zeros(4, np.int64), np.zeros(4, np.int64)]
        self.calls = 0
        self.post_threads = set()

    def collect_sample(self, token):
        s = token["sample"]
        a = self.ring[self.calls % 2]
        self.calls += 1
        a[:] = -1
        a[:len(s)] = s
        self.log.append(("collect", s[0]))
        return (a, len(s))

    def post_sample(self, raw, **kw):
        import threading, time
        time.sleep(self.delay)
        self.post_threads.add(threading.get_ident())
        a, n = raw
        return [int(f) * 10 for f in a[:n]]


def test_a_deeper_host_queue_owns_its_arrays_and_hands_its_tail_over():
    """Round 6: with ``host_queue_depth`` > 1 the device loop does not wait for a slow host stage — the queued batches hold
    COPIES of the recycled arrays, keep batch order, and what is still queued when the device loop ends is handed to the
    caller (``defer``), who collects it later: same results as the depth-1 loop."""
    import time
    ref = _Slow(delay=0.0)
    ref.predict_and_update(iter(range(30)))
    assert ref.results.predictions == [f * 10 for f in range(30)]

    t = _Slow(delay=0.02)
    t.host_queue_depth = 6
    tail = []
    t0 = time.perf_counter()
    t.predict_and_update(iter(range(30)), defer=tail)             # 8 batches x 20 ms of host stage
    dt_loop = time.perf_counter() - t0
    assert len(tail) == 1 and len(t.results.predictions) < 30     # the device loop is done, the host stages are not
    assert dt_loop < 0.12, dt_loop                                # it did not wait for 8 x 20 ms
    tail[0]()
    assert t.results.predictions == ref.results.predictions       # order kept, no array overwritten under a queued batch
    assert len(t.post_threads) == 1 and not t._reuse_outputs
    # without a `defer` list the loop collects its own tail
    t2 = _Slow(delay=0.0)
    t2.host_queue_depth = 6
    t2.predict_and_update(iter(range(30)))
    assert t2.results.predictions == ref.results.predictions


def test_runner_collects_a_tracker_tail_after_the_next_tracker_loop(tmp_path):
    """TrackingRunner: the first tracker's queued host stages drain beside the second tracker's device loop and are collected
    after it; results and per-tracker timings are complete when run() returns."""
    from padel_analytics_amd import video
    from padel_analytics_amd.trackers.runner import TrackingRunner

    n = 30
    video.register_source("ints", lambda p: video.VideoInfo(8, 8, 30, n), lambda p, start, end, stride: iter(range(start, n if end is None else min(end, n), stride)))

    class A(_Slow):
        def __str__(self): return "a"

    class B(_Slow):
        def __str__(self): return "b"

    a, b = A(delay=0.02), B(delay=0.0)
    for t in (a, b):
        t.to = lambda device: None
        t.save_predictions = lambda: None
    r = TrackingRunner([a, b], "ints://clip", tmp_path / "out.mp4", host_queue_depth=6)
    r.run()
    assert a.results.predictions == [f * 10 for f in range(n)] and b.results.predictions == a.results.predictions
    assert r.timings["a"]["frames"] == n and r.timings["a"].get("host_tail_overlapped") is True
    assert r.timings["b"]["frames"] == n
    # depth 1: the old schedule, same results
    a1, b1 = A(delay=0.0), B(delay=0.0)
    for t in (a1, b1):
        t.to = lambda device: None
        t.save_predictions = lambda: None
    r1 = TrackingRunner([a1, b1], "ints://clip", tmp_path / "out.mp4", host_queue_depth=1)
    r1.run()
    assert a1.results.predictions == a.results.predictions and "host_tail_overlapped" not in r1.timings["a"]
