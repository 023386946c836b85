#!/usr/bin/env python
"""How much does a second stream buy?  (tuning tool, GPU only)

Two engines (= two HIP streams) on one GPU, the same model on each; batches of a device-resident clip go alternately
to the two from two host threads (pa_yolo_infer releases the GIL).  Prints frames/s with one and with two streams
for the bench's three graphs."""
import contextlib, sys, tempfile, threading, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench
from padel_analytics_amd import checkpoint, engine as E, video, yolo
from tests import synth

B, H, W, NB = 64, 720, 1280, 8
half = "--half" in sys.argv
engs = [E.Engine(0), E.Engine(0)]
frames = synth.synthetic_frames(B, H, W, seed=1000)
clip = video.DeviceClip(engs[0], frames, repeat=NB)
batches = [list(clip.frames(i * B, (i + 1) * B)) for i in range(NB)]
tmp = tempfile.mkdtemp(prefix="padel_probe_")
for name in ("players", "ball", "pose"):
    cfg = bench.TRACKERS[name]
    with contextlib.redirect_stdout(sys.stderr):
        sd = bench.make_state_dict(name, cfg, frames)
    path = Path(tmp) / f"{name}.pt"
    checkpoint.save_checkpoint(path, sd, "pose" if cfg["kpt"] else "detect", cfg["nc"], cfg["kpt"], cfg["scale"], {0: "person"})
    models = [yolo.YOLO(str(path), engine=e, half=half) for e in engs]
    for m in models:
        m.set_max_batch(B)
    kw = dict(conf=cfg["conf"], iou=0.7, imgsz=cfg["imgsz"], classes=cfg["classes"], channel_reverse=cfg["rev"], pil_stretch=cfg["pre"] == "pil")
    ref = [models[0].infer_frames(b, **kw) for b in batches[:2]]
    models[1].infer_frames(batches[0], **kw)
    t0 = time.perf_counter()
    for b in batches:
        models[0].infer_frames(b, **kw)
    t1 = time.perf_counter()
    outs = [None] * NB
    def work(k):
        for i in range(k, NB, 2):
            outs[i] = models[k].infer_frames(batches[i], **kw)
    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    t2 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    t3 = time.perf_counter()
    import numpy as np
    same = all(np.array_equal(outs[i][0], ref[0][0]) and np.array_equal(outs[i][2], ref[0][2]) for i in range(NB))
    print(f"{name:8s} one stream {1e3 * (t1 - t0) / NB:7.2f} ms/batch   two streams {1e3 * (t3 - t2) / NB:7.2f} ms/batch   "
          f"gain {(t1 - t0) / (t3 - t2):.3f}x   results identical: {same}")
    for m in models:
        m.close()
