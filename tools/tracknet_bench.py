#!/usr/bin/env python
"""Throughput of the reference's ball tracker (TrackNetV3, 227.6 GFLOP per frame) through the streaming
device session (pa_ball_*): frames/s for a synthetic 1280x720 clip, host frames in, masks out, plus the
conv roofline of the TrackNet graph.  GPU only.

    python tools/tracknet_bench.py [--frames 72] [--feed 8]
"""
import argparse, json, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=72)
    ap.add_argument("--feed", type=int, default=8)
    ap.add_argument("--dump-ops", default="", help="per-op CSV of one profiled pass (kind, ksize, M, cout, cin, stride, tile, ms, flops)")
    a = ap.parse_args()
    from oracle import tracknet_ref as tr          # seeded synthetic TrackNet weights (setup only)
    from padel_analytics_amd import engine as E, graph as G
    from tests import synth
    eng = E.default_engine(0)
    frames = synth.synthetic_frames(a.frames, 720, 1280, seed=77)
    g = G.build_tracknet(tr.synth_tracknet_state_dict(3), dtype=E.graph_dtype())
    m = E.Model(eng, g)
    m.set_max_batch(a.feed)
    sess = E.BallSession(m, 720, 1280)
    med = np.median(np.array([f[..., ::-1] for f in frames[:16]]), 0).astype("uint8")

    def run(profile=False):
        sess.set_background(med)
        eng.set_profiling(profile)
        n, recs = 0, []
        for i in range(0, a.frames, a.feed):
            n += len(sess.feed(frames[i:i + a.feed])[0])
            if profile:
                recs += m.profile_rows()
        n += len(sess.feed(None, flush=True)[0])
        eng.set_profiling(False)
        return n, recs

    run()
    t0 = time.perf_counter()
    n, _ = run()
    dt = time.perf_counter() - t0
    _, recs = run(profile=True)
    if a.dump_ops:
        with open(a.dump_ops, "w") as f:
            f.write("kind,ksize,M,cout,cin,stride,bm,bn,ms,flops\n")
            for r in recs:
                f.write(f"{r['kind']},{r['ksize']},{r['M']},{r['cout']},{r['cin']},{r['stride']},{r['mf']},{r['nf']},{r['ms']:.5f},{r['flops']:.0f}\n")
    conv = [r for r in recs if r["kind"] == 2]
    ms = sum(r["ms"] for r in conv); fl = sum(r["flops"] for r in conv)
    print(json.dumps({"tracker": "ball_tracker (TrackNetV3 27->8 @288x512, one window per frame)", "frames": n,
                      "frames_per_s": round(n / dt, 1), "ms_per_frame": round(1e3 * dt / n, 3),
                      "conv_tflops": round(fl / ms / 1e9, 1), "conv_ms_per_frame": round(ms / (a.frames - 7), 3),
                      "all_kernels_ms_per_frame": round(sum(r["ms"] for r in recs) / (a.frames - 7), 3),
                      "non_conv_ms_per_frame": {str(k): round(sum(r["ms"] for r in recs if r["kind"] == k) / (a.frames - 7), 4) for k in sorted({r["kind"] for r in recs}) if k != 2},
                      "input": "host frames (H2D inside the timed region), masks D2H", "arithmetic": E.fp32_mode()}))
