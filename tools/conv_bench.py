#!/usr/bin/env python
"""Single-conv microbenchmark through the C-ABI (tuning tool, GPU only).

Builds a one-op graph (fp32 NHWC input buffer -> conv kxk -> output buffer), runs it through
pa_tracknet_infer with profiling on and prints TFLOP/s per (shape, tile).  Tile override via the
PADEL_CONV_MF / PADEL_CONV_NF environment variables (read once per process), so the sweep re-execs itself.

    python tools/conv_bench.py                  # sweep
    python tools/conv_bench.py --one 2 4        # one tile config over all shapes (used by the sweep)
"""
import argparse, json, os, subprocess, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

# (name, batch, H, W, cin, cout, k, stride) — layer shapes of yolov8m/n @384x640 and pose @1280
SHAPES = [
    ("m.P3.bneck 96->96", 64, 96, 160, 96, 96, 3, 1),
    ("m.P4.bneck 192->192", 64, 48, 80, 192, 192, 3, 1),
    ("m.P5.bneck 288->288", 64, 24, 40, 288, 288, 3, 1),
    ("m.P2.bneck 48->48", 64, 192, 320, 48, 48, 3, 1),
    ("m.head0.P3 192->256", 64, 96, 160, 192, 256, 3, 1),
    ("m.L3 96->192 s2", 64, 192, 320, 96, 192, 3, 2),
    ("m.c2f.cv2 1x1 576->192", 64, 96, 160, 576, 192, 1, 1),
    ("m.c2f.cv1 1x1 96->96 P2", 64, 192, 320, 96, 96, 1, 1),
    ("n.P3.bneck 32->32", 64, 96, 160, 32, 32, 3, 1),
    ("n.P4.bneck 64->64", 64, 48, 80, 64, 64, 3, 1),
    ("n.P2.bneck 16->16", 64, 192, 320, 16, 16, 3, 1),
]


def run_one(mf, nf, shapes, reps, act=1):
    from padel_analytics_amd import engine as E, graph as G
    eng = E.default_engine(0)
    eng.set_profiling(True)
    rng = np.random.default_rng(0)
    out = []
    for (name, B, H, W, cin, cout, k, s) in shapes:
        g = G.Graph(task=G.TASK_TRACKNET)
        b0 = g.buf(0, cin)
        b1 = g.buf(1 if s == 2 else 0, G.pad16(cout))
        w = rng.normal(0, (2.0 / (cin * k * k)) ** 0.5, (cout, cin, k, k)).astype(np.float32)
        g.conv((b0, 0, cin), (b1, 0), w, np.zeros(cout, np.float32), k, s, act)
        g.head_buf = (b1, -1, -1)
        m = E.Model(eng, g)
        m.set_max_batch(B)
        x = rng.normal(0, 1, (B, H, W, cin)).astype(np.float32)
        best = 1e9
        for _ in range(reps):
            m.tracknet_infer(x)
            r = [p for p in m.profile_rows() if p["kind"] == 2][0]
            best = min(best, r["ms"])
        out.append(dict(name=name, mf=r["mf"], nf=r["nf"], ms=best, tf=r["flops"] / best / 1e9))
        m.close()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--one", nargs=2, type=int)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--shapes", default="")
    ap.add_argument("--act", type=int, default=1, help="0 none, 1 SiLU, 2 ReLU (epilogue cost probe)")
    ap.add_argument("--tiles", default="d4x3,d4x4,L0,L1,L2,L3,L4,L5,L6,L7,L8,L9,L10,L11",
                    help="dMxN = register-direct kernel with MFxNF fragments; Lk = LDS kernel variant k; auto")
    a = ap.parse_args()
    shapes = [s for s in SHAPES if (not a.shapes or any(t in s[0] for t in a.shapes.split(",")))]
    if a.one:
        print("RESULT " + json.dumps(run_one(a.one[0], a.one[1], shapes, a.reps, a.act)))
        sys.exit(0)
    table = {}
    for t in ["auto"] + a.tiles.split(","):
        env = dict(os.environ)
        mf, nf = 0, 0
        if t.startswith("d"):
            mf, nf = map(int, t[1:].split("x"))
            env["PADEL_CONV_IMPL"] = "direct"
            env["PADEL_CONV_MF"], env["PADEL_CONV_NF"] = str(mf), str(nf)
        elif t.startswith("P"):
            env["PADEL_CONV_LDS_VARIANT"] = t[1:]
            env["PADEL_CONV_PIPE"] = "1"
        elif t.startswith("A"):                       # default kernel choice with 1x1 prefetch distance N: ApN
            env["PADEL_CONV_TAP_PD"] = t[2:]
        elif t.startswith("T"):                       # v5 tap-unrolled DMA ring (3x3, cin % 32 == 0), same variant ids
            v, _, tune = t[1:].partition("t")
            env["PADEL_CONV_LDS_VARIANT"] = v
            env["PADEL_CONV_TAP"] = "1"
        elif t.startswith("R"):                       # v4 LDS-DMA ring, same variant ids
            v, _, tune = t[1:].partition("t")
            env["PADEL_CONV_LDS_VARIANT"] = v
            env["PADEL_CONV_RING"] = "1"
            if tune:
                env["PADEL_CONV_TUNE"] = tune
        elif t.startswith("L"):
            body, _, tune = t[1:].partition("t")
            v, _, kb = body.partition("k")
            env["PADEL_CONV_LDS_VARIANT"] = v
            if kb:
                env["PADEL_CONV_KB"] = kb
            if tune:
                env["PADEL_CONV_TUNE"] = tune
        p = subprocess.run([sys.executable, __file__, "--one", str(mf), str(nf), "--reps", str(a.reps), "--shapes", a.shapes, "--act", str(a.act)],
                           env=env, capture_output=True, text=True)
        for line in p.stdout.splitlines():
            if line.startswith("RESULT "):
                for r in json.loads(line[7:]):
                    table.setdefault(r["name"], {})[t] = (r["tf"], r["mf"], r["nf"])
        if p.returncode != 0:
            print(t, "failed:", p.stderr[-300:])
    tiles = ["auto"] + a.tiles.split(",")
    print("%-26s" % "shape" + "".join("%9s" % t for t in tiles))
    for name, row in table.items():
        print("%-26s" % name + "".join("%9s" % (("%.1f" % row[t][0]) if t in row else "-") for t in tiles))
    print("auto picks:", {n: f"{r['auto'][1]}x{r['auto'][2]}" for n, r in table.items() if "auto" in r})
