#!/usr/bin/env python
"""Single-conv microbenchmark through the C-ABI (tuning tool, GPU only).

Builds a one-op graph (fp32 NHWC input buffer -> conv kxk -> output buffer), runs it through pa_tracknet_infer with
profiling on and prints TFLOP/s per (shape, tile).  Kernel / tile selection goes through pa_engine_set_tuning
(no environment variables, one process).

    python tools/conv_bench.py                                  # sweep: auto + every tap tile
    python tools/conv_bench.py --tiles auto,T7,T13,Ap3          # Tn tap tile n, Ap3 = auto, 1x1 PD 3
    python tools/conv_bench.py --shapes P3.bneck,1x1 --reps 5
"""
import argparse
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

# (name, batch, H, W, cin, cout, k, stride) — layer shapes of yolov8m/n @384x640 and pose-m @1280
SHAPES = [
    ("m.P3.bneck 96->96", 64, 96, 160, 96, 96, 3, 1),
    ("m.P4.bneck 192->192", 64, 48, 80, 192, 192, 3, 1),
    ("m.P5.bneck 288->288", 64, 24, 40, 288, 288, 3, 1),
    ("m.P2.bneck 48->48", 64, 192, 320, 48, 48, 3, 1),
    ("m.head0.P3 192->256", 64, 96, 160, 192, 256, 3, 1),
    ("m.L3 96->192 s2", 64, 192, 320, 96, 192, 3, 2),
    ("pose.P3.bneck 96->96", 64, 160, 160, 96, 96, 3, 1),
    ("pose.P2.bneck 48->48", 16, 320, 320, 48, 48, 3, 1),
    ("pose.head0.P3 192->304", 64, 160, 160, 192, 304, 3, 1),
    ("pose.L1 48->96 s2", 16, 640, 640, 48, 96, 3, 2),
    ("m.c2f.cv2 1x1 576->192", 64, 96, 160, 576, 192, 1, 1),
    ("m.c2f.cv1 1x1 96->96 P2", 64, 192, 320, 96, 96, 1, 1),
    ("pose.c2f.cv2 1x1 192->96 P2", 16, 320, 320, 192, 96, 1, 1),
    ("m.c2f.cv1 1x1 192->192 P3", 64, 96, 160, 192, 192, 1, 1),
    ("m.c2f.cv2 1x1 1152->384", 64, 48, 80, 1152, 384, 1, 1),
    ("n.P3.bneck 32->32", 64, 96, 160, 32, 32, 3, 1),
    ("n.P4.bneck 64->64", 64, 48, 80, 64, 64, 3, 1),
    ("n.P2.bneck 16->16", 64, 192, 320, 16, 16, 3, 1),
    ("n.P5.bneck 128->128", 64, 12, 20, 128, 128, 3, 1),
    # the players graph's real map sizes (384 x 640 network input): where the 8 x 16 patches do not tile the map
    ("players.P3 96->96 48x80", 64, 48, 80, 96, 96, 3, 1),
    ("players.P4 192->192 24x40", 64, 24, 40, 192, 192, 3, 1),
    ("players.P5 288->288 12x20", 64, 12, 20, 288, 288, 3, 1),
    ("players.head 192->256 48x80", 64, 48, 80, 192, 256, 3, 1),
    # the pose graph's long-K 1x1 layers (C2f cv2 / SPPF / FPN joins at P3-P5, 1280^2)
    ("pose 1x1 768->384 P4", 64, 80, 80, 768, 384, 1, 1),
    ("pose 1x1 960->384 P4", 64, 80, 80, 960, 384, 1, 1),
    ("pose 1x1 1152->576 P5", 64, 40, 40, 1152, 576, 1, 1),
    ("pose 1x1 576->384 P4", 64, 80, 80, 576, 384, 1, 1),
    ("pose 1x1 384->384 P4", 64, 80, 80, 384, 384, 1, 1),
    ("pose 1x1 384->192 P3", 64, 160, 160, 384, 192, 1, 1),
    ("pose 1x1 576->192 P3", 64, 160, 160, 576, 192, 1, 1),
    # TrackNetV3 (fp32 checkpoint: three products) at 288 x 512 and the 64-channel layers of the pose heads
    ("tn 64->64 288x512", 16, 288, 512, 64, 64, 3, 1),
    ("tn 128->128 144x256", 32, 144, 256, 128, 128, 3, 1),
    ("tn 256->256 72x128", 64, 72, 128, 256, 256, 3, 1),
    ("tn 512->512 36x64", 64, 36, 64, 512, 512, 3, 1),
    ("pose.head 192->64 P3", 64, 160, 160, 192, 64, 3, 1),
    # stride-2 3x3 layers (backbone L3 / L5 / L7, the PAN's downsampling convs) of the pose graph (1280^2) and the players graph
    ("s2 pose.L3 96->192", 64, 320, 320, 96, 192, 3, 2),
    ("s2 pose.L5 192->384", 64, 160, 160, 192, 384, 3, 2),
    ("s2 pose.L7 384->576", 64, 80, 80, 384, 576, 3, 2),
    ("s2 pose.pan 192->192", 64, 160, 160, 192, 192, 3, 2),
    ("s2 pose.pan 384->384", 64, 80, 80, 384, 384, 3, 2),
    ("s2 players.L3 96->192", 64, 96, 160, 96, 192, 3, 2),
    ("s2 players.L5 192->384", 64, 48, 80, 192, 384, 3, 2),
    ("pose.head 64->64 P3", 64, 160, 160, 64, 64, 3, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--shapes", default="")
    ap.add_argument("--w16", action="store_true", help="h2: weights that are fp16 numbers (the two-product kernels: PA_CONV_W_SINGLE)")
    ap.add_argument("--act", type=int, default=1, help="0 none, 1 SiLU, 2 ReLU (epilogue cost probe)")
    ap.add_argument("--tiles", default="auto,T6,T7,T9,T10,T11,T13,T14,T15,T20")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f16", "h2"], help="f16: conv_tap16 kernels (tiles Tn = fp16 tile ids 6,7,9,11,12,20,30,31,32); h2: fp16-pair kernels (tiles Tn = 207,209,211,213,220,225,239,243,303,304,313,323,341-343)")
    a = ap.parse_args()
    from padel_analytics_amd import engine as E, graph as G
    eng = E.default_engine(0)
    eng.set_profiling(True)
    shapes = [s for s in SHAPES if (not a.shapes or any(t in s[0] for t in a.shapes.split(",")))]
    tiles = a.tiles.split(",")
    table = {}
    rng = np.random.default_rng(0)
    for (name, B, H, W, cin, cout, k, s) in shapes:
        f16 = a.dtype == "f16"
        h2 = a.dtype == "h2"
        g = G.Graph(task=G.TASK_TRACKNET, dtype=G.DTYPE_F16 if f16 else G.DTYPE_H2 if h2 else G.DTYPE_F32)
        cin_p = g.padk(cin)
        b0 = g.buf(0, cin_p)
        b1 = g.buf(1 if s == 2 else 0, g.padk(cout))
        w = rng.normal(0, (2.0 / (cin * k * k)) ** 0.5, (cout, cin, k, k)).astype(np.float32)
        if a.w16:
            w = w.astype(np.float16).astype(np.float32)
        g.conv((b0, 0, cin_p), (b1, 0), w, np.zeros(cout, np.float32), k, s, a.act, out_width=g.padk(cout) if (f16 or h2) else None)
        if f16 or h2:                     # the measured conv writes fp16; a tiny fp32 head keeps pa_tracknet_infer's contract
            hd = g.buf(1 if s == 2 else 0, 16)
            g.conv((b1, 0, g.padk(cout)), (hd, 0), np.zeros((1, g.padk(cout), 1, 1), np.float32), np.zeros(1, np.float32), 1, 1, 0)
            g.head_buf = (hd, -1, -1)
        else:
            g.head_buf = (b1, -1, -1)
        cin = cin_p
        m = E.Model(eng, g)
        m.set_max_batch(B)
        x = rng.normal(0, 1, (B, H, W, cin)).astype(np.float32)
        for t in tiles:
            kw = dict(impl=0, variant=-1, tap_pd=2)        # "auto" / Tn / Apn: the fp32-MFMA tap kernels
            kw["tune"] = 1
            if t.startswith("T"):                         # Tn = tile n; Tn:m = tile n with tuning word m (probe builds: ablation bits)
                tt = t[1:].split(":")
                kw.update(variant=int(tt[0]))
                if len(tt) > 1:
                    kw["tune"] = int(tt[1])
            elif t.startswith("B"):                       # bf16x3 kernels: B = auto tile, Bn = tile n
                kw.update(impl=2, variant=int(t[1:]) if len(t) > 1 else -1)
            elif t.startswith("Ap"):
                kw.update(tap_pd=int(t[2:]))
            eng.set_tuning(**kw)
            best, r = 1e9, None
            try:
                for _ in range(a.reps):
                    m.tracknet_infer(x)
                    r = [p for p in m.profile_rows() if p["kind"] == 2][0]      # the conv under test is the first conv op
                    best = min(best, r["ms"])
                table.setdefault(name, {})[t] = (r["flops"] / best / 1e9, r["mf"], r["nf"])
            except E.EngineError:
                pass
        m.close()
    eng.set_tuning(impl=2, variant=-1, tap_pd=2, tune=1)
    print("%-30s" % "shape" + "".join("%9s" % t for t in tiles))
    for name, row in table.items():
        print("%-30s" % name + "".join("%9s" % (("%.1f" % row[t][0]) if t in row else "-") for t in tiles))
    print("auto picks:", {n: f"{r['auto'][1]}x{r['auto'][2]}" for n, r in table.items() if "auto" in r})


if __name__ == "__main__":
    main()
