"""GPU tests of the fp16 path (BASELINE configs[4]; VERDICT r1 row g): fp16 activations / weights, fp32 accumulation on
v_mfma_f32_16x16x32_f16 (csrc/conv_tap16.hip), fp32 Detect/Pose heads.

* conv unit test: every fp16 tile variant against an fp64 conv2d of the SAME fp16-rounded operands (so the only
  error left is fp32 accumulation order + the fp16 rounding of the stored output), bitwise equal across tiles, on
  shapes that exercise the 32-channel K tail, the chunk -> tail wrap, stride 2, 1x1, partial channel tiles, the M
  tail, the fused residual and the fp32-output epilogue;
* whole graphs: head maps against the CPU interpretation of the same op list with fp16 storage rounding
  (tests/graph_interp.py), and detections after NMS against the fp32 oracle with the path's OWN reported L-inf (the
  reference runs half=False; there is no 1e-3 px claim for fp16)."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import yolov8_ref as ref
from padel_analytics_amd import engine as E, graph as G
from tests import synth
from tests import graph_interp, parity
from tests.test_gpu_yolo_parity import _calib

pytestmark = pytest.mark.gpu

# (B, H, W, cin, cout, k, stride, act, residual)
CASES = [
    (2, 24, 40, 64, 64, 3, 1, G.ACT_SILU, False),
    (3, 20, 36, 96, 80, 3, 1, G.ACT_SILU, True),      # full chunk -> 32-channel tail wrap, partial N tile, residual
    (2, 32, 48, 32, 32, 3, 2, G.ACT_RELU, False),     # tail block only, stride 2
    (1, 16, 24, 96, 96, 1, 1, G.ACT_NONE, False),     # 1x1
    (2, 12, 20, 64, 144, 3, 1, G.ACT_SIGMOID, False),
    (1, 36, 28, 288, 48, 1, 1, G.ACT_SILU, True),
    (1, 8, 12, 576, 192, 3, 1, G.ACT_SILU, False),    # long K
    (3, 17, 23, 160, 96, 3, 1, G.ACT_SILU, True),     # two chunks + tail, odd spatial size
    (2, 16, 16, 64, 39, 1, 1, G.ACT_NONE, False),     # cout not a multiple of 4: element-wise epilogue
    (2, 20, 40, 64, 192, 3, 1, G.ACT_SILU, True),     # two 96-channel tiles, 16-row tiles with a partial last row block (quad patch kernel)
]
VARIANTS = (6, 7, 9, 11, 12, 20, 30, 31, 32, 46, 47, 49, 51, 60, 70, 71, 72)
# fp16 patch kernel (csrc/conv_patch16.hip; stride-1 3x3 only, elsewhere these ids fall back to tap tiles): it walks K as
# (32-channel chunk, tap) instead of (64-channel chunk, tap, half), so its fp32 sums round differently from the tap
# kernels' — bitwise equal among its own tiles, same error bound against fp64
PATCH_VARIANTS = (303, 304, 306, 323, 324, 326)       # 32x: the quad kernel (16 x 16 pixels per workgroup)


def _run(eng, case, x16, w, b, wr):
    B, H, W, cin, cout, k, s, act, use_res = case
    g = G.Graph(task=G.TASK_TRACKNET, dtype=G.DTYPE_F16)
    b0 = g.buf(0, cin)
    lvl = 1 if s == 2 else 0
    cp = g.padk(cout)
    b1 = g.buf(lvl, cp)                                  # fp16 output of the conv under test
    res = None
    if use_res:
        b2 = g.buf(lvl, cp)
        g.conv((b0, 0, cin), (b2, 0), wr, np.zeros(cout, np.float32), 1, s, G.ACT_NONE, out_width=cp)
        res = (b2, 0)
    g.conv((b0, 0, cin), (b1, 0), w, b, k, s, act, res=res, out_width=cp)
    hd = g.buf(lvl, G.pad16(cout))                        # fp32 head: exact copy through an identity 1x1
    g.conv((b1, 0, cp), (hd, 0), np.eye(cout, cp, dtype=np.float32)[:, :, None, None], np.zeros(cout, np.float32), 1, 1, G.ACT_NONE)
    g.head_buf = (hd, -1, -1)
    m = E.Model(eng, g)
    m.set_max_batch(B)
    y = m.tracknet_infer(x16)[..., :cout]
    m.close()
    return y


@pytest.mark.parametrize("case", CASES, ids=[f"h{i}" for i in range(len(CASES))])
def test_conv16_variants(gpu_engine, case):
    B, H, W, cin, cout, k, s, act, use_res = case
    rng = np.random.default_rng(cin * 131 + cout)
    x16 = rng.normal(0, 1, (B, H, W, cin)).astype(np.float16)
    w = rng.normal(0, (2.0 / (cin * k * k)) ** 0.5, (cout, cin, k, k)).astype(np.float16).astype(np.float32)
    b = rng.normal(0, 0.5, cout).astype(np.float32)
    wr = rng.normal(0, (1.0 / cin) ** 0.5, (cout, cin, 1, 1)).astype(np.float16).astype(np.float32)
    xt = torch.from_numpy(x16.astype(np.float32)).permute(0, 3, 1, 2).double()
    want = F.conv2d(xt, torch.from_numpy(w).double(), torch.from_numpy(b).double(), stride=s, padding=k // 2)
    want = {G.ACT_SILU: F.silu, G.ACT_RELU: F.relu, G.ACT_SIGMOID: torch.sigmoid, G.ACT_NONE: lambda t: t}[act](want)
    if use_res:
        want = want + F.conv2d(xt, torch.from_numpy(wr).double(), stride=s).half().double()    # the residual is stored as fp16
    want = want.permute(0, 2, 3, 1).numpy()
    scale = max(1.0, float(np.abs(want).max()))
    outs = {}
    try:
        for v in VARIANTS:
            gpu_engine.set_tuning(variant=v)
            for rep in range(2):
                outs[f"H{v}.{rep}"] = _run(gpu_engine, case, x16, w, b, wr)
        outs_p = {}
        for v in PATCH_VARIANTS:
            gpu_engine.set_tuning(variant=v)
            for rep in range(2):
                outs_p[f"P{v}.{rep}"] = _run(gpu_engine, case, x16, w, b, wr)
        gpu_engine.set_tuning(variant=-1)
        auto = _run(gpu_engine, case, x16, w, b, wr)
    finally:
        gpu_engine.set_tuning(variant=-1)
    patch_case = k == 3 and s == 1
    if not patch_case:
        outs.update(outs_p)                      # the ids fell back to tap tiles: same family
        outs_p = {}
    for fam in (outs, outs_p):
        if not fam:
            continue
        ref_name, r0 = next(iter(fam.items()))
        for name, y in fam.items():
            assert y.shape == want.shape
            err = float(np.abs(y - want).max()) / scale
            assert err < 1.5e-3, f"{name}: rel err {err:.2e} vs fp64 conv2d of the fp16 operands (fp16 output rounding is 4.9e-4)"
            assert np.array_equal(y, r0), f"{name} differs bitwise from {ref_name} (max {np.abs(y - r0).max():.3e})"
    assert any(np.array_equal(auto, next(iter(fam.values()))) for fam in (outs, outs_p) if fam), "auto matches neither family"


@pytest.mark.parametrize("scale,nc,kpt,hw,S,pre", [("n", 80, None, (720, 1280), 640, "lb"), ("m", 80, None, (720, 1280), 640, "lb"),
                                                  ("n", 1, (13, 3), (720, 1280), 640, "pil"), ("m", 1, (13, 3), (720, 1280), 1280, "pil")])
def test_fp16_graph_vs_emulation_and_fp32_oracle(gpu_engine, scale, nc, kpt, hw, S, pre):
    from PIL import Image
    frames = synth.synthetic_frames(2, hw[0], hw[1], seed=5)
    conf = 0.5 if kpt is None else 0.25
    if pre == "pil":
        pil = [np.asarray(Image.fromarray(f[..., ::-1].copy()).resize((S, S))) for f in frames]
        srcs = [p[..., ::-1] for p in pil]
    else:
        srcs = [f[..., ::-1] for f in frames]
    import bench
    cfg = dict(scale=scale, nc=nc, kpt=kpt, imgsz=S, conf=conf, pre="pil" if pre == "pil" else "letterbox")
    sd = bench.make_state_dict(f"fp16-{scale}-{nc}-{S}", cfg, frames)
    g16 = G.build_yolov8(sd, nc, kpt, dtype="f16")
    m = E.Model(gpu_engine, g16)
    m.set_max_batch(2)
    boxes, kpts, counts = m.yolo_infer(frames, 2, hw[0], hw[1], imgsz=S, conf=conf, iou=0.7, classes=[0],
                                       pre_mode=E.PRE_PIL_STRETCH if pre == "pil" else E.PRE_LETTERBOX, channel_reverse=pre == "pil")
    heads = [m.read_head(l, 2) for l in range(3)]
    m.close()
    # (1) head maps vs the CPU interpretation of the same fp16 op list (one frame, cropped for speed when large)
    x = ref.preprocess(srcs[:1], S)
    if S <= 640:
        bufs = graph_interp.run(g16, net_in=x)
        nch = 64 + nc + (kpt[0] * kpt[1] if kpt else 0)
        for l in range(3):
            want = bufs[g16.head_buf[l]][0, :nch].permute(1, 2, 0).numpy()
            got = heads[l][0, ..., :nch]
            err = float(np.abs(got - want).max()) / max(1.0, float(np.abs(want).max()))
            # two fp16 evaluations that differ only in fp32 accumulation order: rounding flips of the stored halves
            # compound through ~80 layers of these ill-conditioned synthetic nets (measured 1e-3 .. 2.2e-2)
            assert err < 6e-2, f"head level {l}: rel err {err:.3e} vs the fp16 emulation"
    # (2) detections vs the fp32 oracle: own L-inf, reported (no 1e-3 claim); most detections must match
    r32 = ref.predict(ref.YoloV8Ref(sd, nc, kpt), srcs, conf, 0.7, S, classes=[0])
    tot, matched, worst, sq, cnt = 0, 0, 0.0, 0.0, 0
    for i, r in enumerate(r32):
        gb = boxes[i, :counts[i]]
        pairs, ru, gu = parity.match(r["boxes"], gb, tol_match=8.0)
        tot += len(r["boxes"]); matched += len(pairs)
        for a_, b_ in pairs:
            d = np.abs(gb[b_, :4] - r["boxes"][a_, :4])
            worst = max(worst, float(d.max())); sq += float((d.astype(np.float64) ** 2).sum()); cnt += 4
    # (the synthetic calibrated networks amplify perturbations by ~1e5 px per unit relative error — their fp32 noise
    # floor alone is ~1e-2 px, tests/test_noise_floor.py — so fp16's 5e-4 rounding moves boxes by pixels and flips
    # threshold-adjacent detections; trained weights are far better conditioned.  What is asserted is that the path
    # reproduces the bulk of the detections; the L-inf is REPORTED.)
    assert tot > 0 and matched >= 0.7 * tot, (matched, tot)
    rep = {"detections": tot, "matched": matched, "linf_px_vs_fp32_oracle": worst, "rms_px": (sq / max(cnt, 1)) ** 0.5}
    print(f"fp16 {scale} nc={nc} kpt={kpt} S={S}: {rep}")
    out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out")
    if os.path.isdir(out):
        p = os.path.join(out, "parity_report_fp16.json")
        old = json.load(open(p)) if os.path.exists(p) else {}
        old[f"{scale}-nc{nc}-{'pose' if kpt else 'detect'}-{S}"] = rep
        json.dump(old, open(p, "w"), indent=1)
    assert rep["rms_px"] < 4.0, rep


def test_fused_sppf_fp16_matches_three_pool_launches(gpu_engine):
    """sppf_f16_kernel (the default of fp16 graphs since round 5: 0.34 -> 0.24 ms per c4 step, profiles/r5e_sppf_f16.txt; fuse_sppf=0: three launches): same head maps (as numbers: a maximum does not depend on the walk, the sign of a zero
    may) and the same detections as the three pool5_kernel launches."""
    from padel_analytics_amd import yolo_arch
    from tests import synth
    for scale, (h, w), imgsz in (("n", (360, 640), 640), ("m", (720, 1280), 1280)):
        frames = synth.synthetic_frames(2, h, w, seed=12)
        sd = yolo_arch.synth_state_dict(scale, 80, None, seed=5, cls_bias=-1.0)
        m = E.Model(gpu_engine, G.build_yolov8(sd, 80, None, dtype="f16"))
        m.set_max_batch(2)
        try:
            gpu_engine.set_tuning(fuse_sppf=0)
            b0, _, c0 = m.yolo_infer(frames, 2, h, w, imgsz=imgsz, conf=0.25, iou=0.7)
            h0 = [m.read_head(l, 2) for l in range(3)]
            gpu_engine.set_tuning(fuse_sppf=1)
            b1, _, c1 = m.yolo_infer(frames, 2, h, w, imgsz=imgsz, conf=0.25, iou=0.7)
            h1 = [m.read_head(l, 2) for l in range(3)]
        finally:
            gpu_engine.set_tuning(fuse_sppf=1)
            m.close()
        for l in range(3):
            assert np.array_equal(h0[l], h1[l]), f"{scale}: head {l} differs"
        assert np.array_equal(c0, c1) and np.array_equal(b0, b1) and int(c0.sum()) > 0
