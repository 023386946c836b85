#!/usr/bin/env python
"""Does the host stage of a batch tracker really run under the device stage of the next batch?  (tuning tool, GPU only)

Times, for the bench's players / ball / pose trackers on a device-resident clip: the device stage alone (infer_sample),
the host stage alone (post_sample on the saved raw outputs) and the two-stage loop Tracker._predict_batches."""
import contextlib, sys, tempfile, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench
from padel_analytics_amd import engine as E, video
from tests import synth
from padel_analytics_amd.trackers.tracker import _sampler

B, H, W, NB = 64, 720, 1280, 8
half = "--half" in sys.argv
eng = E.Engine(0)
eng.comm_init(E.comm_unique_id(), 1, 0)
frames = synth.synthetic_frames(B, H, W, seed=1000)
clip = video.DeviceClip(eng, frames, repeat=NB)
tmp = tempfile.mkdtemp(prefix="padel_probe_")
with contextlib.redirect_stdout(sys.stderr):
    trackers, _ = bench.build_trackers(["players", "ball", "pose"], frames, 0, B, H, W, eng, tmp, half=half)
info = video.VideoInfo.from_video_path(clip)
for name, t in trackers.items():
    t.video_info_post_init(info)
    t.to(t.DEVICE)
    batches = list(_sampler(clip.frames(), B))
    t.infer_sample(batches[0])
    eng.synchronize()
    t0 = time.perf_counter()
    raws = [t.infer_sample(s) for s in batches]
    t1 = time.perf_counter()
    outs = [t.post_sample(r) for r in raws]
    t2 = time.perf_counter()
    t.restart()
    got = []
    with contextlib.redirect_stdout(sys.stderr):
        t3 = time.perf_counter()
        t._predict_batches(clip.frames(), got.extend)
        t4 = time.perf_counter()
    print(f"{name:8s} per batch of {B}: device stage {1e3 * (t1 - t0) / NB:7.2f} ms   host stage {1e3 * (t2 - t1) / NB:7.2f} ms   "
          f"two-stage loop {1e3 * (t4 - t3) / NB:7.2f} ms   (sum {1e3 * (t2 - t0) / NB:.2f}, max {1e3 * max(t1 - t0, t2 - t1) / NB:.2f})")
