"""Diagnostic (GPU): why engine(B=64)[i] != engine(B=2)[i] bitwise.  Compares raw head maps of frames [0:2]."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench
from padel_analytics_amd import engine as E, graph as G
from tests import synth

name = sys.argv[1] if len(sys.argv) > 1 else "ball"
frames = synth.synthetic_frames(64, 720, 1280, seed=1000)
cfg = bench.TRACKERS[name]
sd = bench.make_state_dict(name, cfg, frames)
eng = E.default_engine(0)
g = G.build_yolov8(sd, cfg["nc"], cfg["kpt"])


def heads(B, fr, **tune):
    eng.set_tuning(impl=2, variant=-1, alias=1, graph=0)
    eng.set_tuning(**tune)
    m = E.Model(eng, g)
    m.set_max_batch(B)
    n, h, w, _ = fr.shape
    out = m.yolo_infer(fr, n, h, w, imgsz=cfg["imgsz"], conf=cfg["conf"], iou=0.7, classes=cfg["classes"],
                       pre_mode=E.PRE_PIL_STRETCH if cfg["pre"] == "pil" else E.PRE_LETTERBOX, channel_reverse=cfg["rev"])
    hd = [m.read_head(l, 2) for l in range(3)]
    m.close()
    return hd, out


def cmp(tag, a, b):
    s = []
    for l in range(3):
        d = np.abs(a[0][l] - b[0][l])
        s.append(f"L{l}: ndiff {int((d > 0).sum())}/{d.size} max {d.max():.3e}")
    same_out = all(np.array_equal(x[:2], y[:2]) for x, y in zip(a[1], b[1]) if x is not None)
    print(f"{tag:44s} {' | '.join(s)} | outputs equal {same_out}", flush=True)


a64 = heads(64, frames)
cmp("B=64 run twice", a64, heads(64, frames))
a2 = heads(2, frames[:2])
cmp("B=2 run twice", a2, heads(2, frames[:2]))
cmp("B=64 vs B=2 (default)", a64, a2)
cmp("B=64 vs B=4", a64, heads(4, frames[:4]))
cmp("B=64 vs B=2, alias=0 both", heads(64, frames, alias=0), heads(2, frames[:2], alias=0))
for v in (7, 11):
    cmp(f"B=64 vs B=2, tap variant {v} both", heads(64, frames, variant=v), heads(2, frames[:2], variant=v))
cmp("B=2: tap auto vs tap variant 7", a2, heads(2, frames[:2], variant=7))
cmp("B=64: tap auto vs tap variant 7", a64, heads(64, frames, variant=7))
