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
