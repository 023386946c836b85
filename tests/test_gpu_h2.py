"""GPU unit tests of the h2 arithmetic (csrc/h2_common.h): activations as fp16 pairs x ~ h + m / 2048 written by the
producer, three f16 MFMA products per operand pair, correction products in their own accumulator.

* every tile of the tap kernels (conv_tap_h2.hip) and of the patch kernel (conv_patch_h2.hip) on the conv shapes of
  tests/test_gpu_conv.py against torch conv2d in fp64 ON THE ORIGINAL fp32 OPERANDS — the bound (L-inf 3e-6 relative)
  and the admission criterion (RMS error <= 1.25 x the fp32-MFMA kernels') are the ones the bf16x3 kernels were
  admitted under, and they include what the 22-23-bit operand representation costs;
* bitwise equality among the tiles that share the accumulation scheme;
* values far below the fp16 normal range (the MFMA must not flush fp16 subnormals), the overflow flag, the absorbed
  upsample, the helper kernels (SPPF pools, MaxPool2d(2, 2), upsample) — exact on pairs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from padel_analytics_amd import engine as E, graph as G
from tests.test_gpu_conv import CASES

pytestmark = pytest.mark.gpu

H2_TILES = (207, 209, 211, 213, 220, 225, 239, 243, 244, 245, 246, 247, 248, 303, 304, 313, 323, 324, 325, 341, 342, 343)      # 324 / 325: the register-weights quad kernels (round 6; 96-channel tiles, two-product layers only / 64-channel tiles, two or three products); 239 / 243: tap tiles with the three-stage activation ring (conv_tap_h2p.hip); 31x: software-pipelined patch schedule; 323: the quad patch kernel; 34x: the wide patch kernel (cin 16 / 32 / 48)
H2_SINGLE_LEVEL = ()            # (the 6-fragment patch tile 306 — main product accumulated in ONE level — was removed in round 5)

def _graph(case, w, b, wr, dtype):
    B, H, W, cin, cout, k, s, act, use_res = case
    g = G.Graph(task=G.TASK_TRACKNET, dtype=dtype)
    b0 = g.buf(0, cin)
    lvl = 1 if s == 2 else 0
    res = None
    if use_res:
        b2 = g.buf(lvl, G.pad16(cout))
        g.conv((b0, 0, cin), (b2, 0), wr, np.zeros(cout, np.float32), 1, s, G.ACT_NONE)
        res = (b2, 0)
    b1 = g.buf(lvl, G.pad16(cout))
    g.conv((b0, 0, cin), (b1, 0), w, b, k, s, act, res=res)
    g.head_buf = (b1, -1, -1)          # head buffer: written as plain fp32
    return g


def _run(eng, case, x, w, b, wr, dtype=G.DTYPE_H2, want_flag=False):
    m = E.Model(eng, _graph(case, w, b, wr, dtype))
    m.set_max_batch(case[0])
    y = m.tracknet_infer(x)[..., :case[4]]
    flag = m.take_overflow()
    m.close()
    return (y, flag) if want_flag else y


def _data(case, xscale=1.0, wscale=1.0):
    B, H, W, cin, cout, k, s, act, use_res = case
    rng = np.random.default_rng(cin * 131 + cout)
    x = (rng.normal(0, 1, (B, H, W, cin)) * xscale).astype(np.float32)
    w = (rng.normal(0, (2.0 / (cin * k * k)) ** 0.5, (cout, cin, k, k)) * wscale).astype(np.float32)
    b = (rng.normal(0, 0.5, cout) * xscale * wscale).astype(np.float32)
    wr = (rng.normal(0, (1.0 / cin) ** 0.5, (cout, cin, 1, 1)) * wscale).astype(np.float32)
    return x, w, b, wr


def _want(case, x, w, b, wr):
    B, H, W, cin, cout, k, s, act, use_res = case
    xt = torch.from_numpy(x).permute(0, 3, 1, 2).double()
    want = F.conv2d(xt, torch.from_numpy(w).double(), torch.from_numpy(b).double(), stride=s, padding=k // 2)
    want = {G.ACT_SILU: F.silu, G.ACT_RELU: F.relu, G.ACT_SIGMOID: torch.sigmoid, G.ACT_NONE: lambda t: t}[act](want)
    if use_res:
        want = want + F.conv2d(xt, torch.from_numpy(wr).double(), stride=s)
    return want.permute(0, 2, 3, 1).numpy()


@pytest.mark.parametrize("case", CASES, ids=[f"c{i}" for i in range(len(CASES))])
def test_h2_conv_variants(gpu_engine, case):
    x, w, b, wr = _data(case)
    want = _want(case, x, w, b, wr)
    scale = max(1.0, float(np.abs(want).max()))
    outs = {}
    try:
        for v in H2_TILES:
            gpu_engine.set_tuning(variant=v)
            for rep in range(2):                       # LDS-DMA ring: a DMA / barrier race is not deterministic
                outs[f"H{v}.{rep}"] = _run(gpu_engine, case, x, w, b, wr)
        gpu_engine.set_tuning(variant=-1)
        outs["H.auto"] = _run(gpu_engine, case, x, w, b, wr)
        gpu_engine.set_tuning(alias=0)
        outs["H.auto.noalias"] = _run(gpu_engine, case, x, w, b, wr)
        gpu_engine.set_tuning(impl=0, variant=-1, alias=1)
        y32 = _run(gpu_engine, case, x, w, b, wr, dtype=G.DTYPE_F32)           # fp32-input MFMA kernels
        gpu_engine.set_tuning(impl=2)
        y3 = _run(gpu_engine, case, x, w, b, wr, dtype=G.DTYPE_F32)            # bf16x3
    finally:
        gpu_engine.set_tuning(impl=2, variant=-1, alias=1)
    k, cin = case[5], case[3]
    patch_applies = k == 3 and case[6] == 1
    ref_name, ref = "H220.0", outs["H220.0"]
    for name, y in outs.items():
        assert y.shape == want.shape
        err = float(np.abs(y - want).max()) / scale
        assert err < 3e-6, f"{name}: rel err {err:.2e} vs fp64 conv2d"
        single = patch_applies and any(name.startswith(f"H{v}.") for v in H2_SINGLE_LEVEL)
        if not single:
            assert np.array_equal(y, ref), f"{name} differs bitwise from {ref_name} (max {np.abs(y - ref).max():.3e})"
    rms = lambda y: float(np.sqrt(np.mean((y - want) ** 2)))
    print(f"case {case}: RMS error vs fp64  fp32-MFMA {rms(y32):.3e}  bf16x3 {rms(y3):.3e}  h2 {rms(ref):.3e}")
    assert rms(ref) <= 1.25 * rms(y32) + 1e-9, (rms(ref), rms(y32))


@pytest.mark.parametrize("case", [CASES[0], CASES[1], CASES[3], CASES[5], CASES[7], CASES[8], CASES[11], CASES[12], CASES[13]],
                         ids=["3x3", "3x3-tail-res", "1x1", "1x1-res", "s2-res", "odd-size", "quad-192", "1x1-long-K", "s2-odd-13-fragments"])
def test_h2_two_product_mode_on_fp16_weights(gpu_engine, case):
    """Round 5: a checkpoint's conv weights are fp16 numbers (Ultralytics stores ``model.half()``); with BatchNorm's scale kept
    in the conv's per-channel OUTPUT scale instead of multiplied into them (``Graph.conv(out_scale=)``), the packed weights'
    correction plane is all zero (``PA_CONV_W_SINGLE``) and the kernels run TWO products per operand pair instead of three
    (wh x ah and wh x am; the skipped wm x ah is exactly zero).  Checked on every tile: the flag is set by the packer; the
    two-product kernels give BITWISE the results of the three-product kernels on the same blob (tuning ``w_single=0``); against
    fp64 conv2d with the exact weights w x scale the error stays inside the h2 bound (3e-6) and the RMS inside the admission
    criterion against the fp32-input MFMA kernels run on the FOLDED fp32 weights (what the reference's fused model holds)."""
    B, H, W, cin, cout, k, s, act, use_res = case
    rng = np.random.default_rng(cin * 17 + cout + k)
    x = rng.normal(0, 1, (B, H, W, cin)).astype(np.float32)
    w = rng.normal(0, (2.0 / (cin * k * k)) ** 0.5, (cout, cin, k, k)).astype(np.float16).astype(np.float32)      # fp16 numbers
    scale = rng.uniform(0.5, 2.0, cout).astype(np.float32)
    b = rng.normal(0, 0.5, cout).astype(np.float32)
    wr = rng.normal(0, (1.0 / cin) ** 0.5, (cout, cin, 1, 1)).astype(np.float32)
    w_fold = (w.reshape(cout, -1) * scale[:, None]).reshape(w.shape).astype(np.float32)                        # fold_bn's product
    xt = torch.from_numpy(x).permute(0, 3, 1, 2).double()
    w_exact = torch.from_numpy(w).double() * torch.from_numpy(scale).double()[:, None, None, None]
    want = F.conv2d(xt, w_exact, torch.from_numpy(b).double(), stride=s, padding=k // 2)
    want = {G.ACT_SILU: F.silu, G.ACT_RELU: F.relu, G.ACT_SIGMOID: torch.sigmoid, G.ACT_NONE: lambda t: t}[act](want)
    if use_res:
        want = want + F.conv2d(xt, torch.from_numpy(wr).double(), stride=s)
    want = want.permute(0, 2, 3, 1).numpy()
    sc = max(1.0, float(np.abs(want).max()))

    def graph(dtype, weights, out_scale):
        g = G.Graph(task=G.TASK_TRACKNET, dtype=dtype)
        b0 = g.buf(0, cin)
        lvl = 1 if s == 2 else 0
        res = None
        if use_res:
            b2 = g.buf(lvl, G.pad16(cout))
            g.conv((b0, 0, cin), (b2, 0), wr, np.zeros(cout, np.float32), 1, s, G.ACT_NONE)
            res = (b2, 0)
        b1 = g.buf(lvl, G.pad16(cout))
        g.conv((b0, 0, cin), (b1, 0), weights, b, k, s, act, res=res, out_scale=out_scale)
        g.head_buf = (b1, -1, -1)
        return g

    def run(g):
        m = E.Model(gpu_engine, g)
        m.set_max_batch(B)
        y = m.tracknet_infer(x)[..., :cout]
        assert not m.take_overflow()
        m.close()
        return y

    g2 = graph(G.DTYPE_H2, w, scale)
    assert g2.ops[-1]["flags"] & G.FLAG_W_SINGLE, "fp16 weights + out_scale must pack with an all-zero m plane"
    g3 = graph(G.DTYPE_H2, w_fold, None)
    assert not (g3.ops[-1]["flags"] & G.FLAG_W_SINGLE), "folded weights are not fp16 numbers"
    outs = {}
    try:
        for v in H2_TILES + (-1,):
            gpu_engine.set_tuning(variant=v, w_single=1)
            outs[f"two-product H{v}"] = run(g2)
            gpu_engine.set_tuning(variant=v, w_single=0)
            outs[f"three-product kernels, same blob H{v}"] = run(g2)
        gpu_engine.set_tuning(variant=-1, w_single=1)
        y_fold = run(g3)                                                                  # the round-4 path: folded weights, 3 products
        gpu_engine.set_tuning(impl=0, variant=-1)
        y32 = run(graph(G.DTYPE_F32, w_fold, None))                                       # fp32-input MFMA kernels on the folded weights
    finally:
        gpu_engine.set_tuning(impl=2, variant=-1, w_single=1)
    ref_name, ref = next(iter(outs.items()))
    for name, y in outs.items():
        err = float(np.abs(y - want).max()) / sc
        assert err < 3e-6, f"{name}: rel err {err:.2e} vs fp64 conv2d on the exact weights"
        assert np.array_equal(y, ref), f"{name} differs bitwise from {ref_name} (max {np.abs(y - ref).max():.3e})"
    rms = lambda y: float(np.sqrt(np.mean((y - want) ** 2)))
    print(f"case {case}: RMS vs fp64(w x scale)  two-product {rms(ref):.3e}  folded weights, three products {rms(y_fold):.3e}  fp32-MFMA on folded weights {rms(y32):.3e}")
    assert rms(ref) <= 1.25 * rms(y32) + 1e-9, (rms(ref), rms(y32))


@pytest.mark.parametrize("xs,ws", [(1e-4, 1.0), (3e-6, 1e-3), (200.0, 1e-5), (1.0, 64.0)], ids=["tiny-x", "tiny-x-w", "big-x-tiny-w", "big-w"])
def test_h2_dynamic_range(gpu_engine, xs, ws):
    """fp16 subnormal parts (|x| < 6.1e-5: h is subnormal or 0, m carries the value) must survive the MFMA, and the
    per-row weight scale must keep tiny / large weights exact: same relative error as at unit scale."""
    case = (2, 24, 40, 32, 64, 3, 1, G.ACT_NONE, False)
    x, w, b, wr = _data(case, xs, ws)
    want = _want(case, x, w, b, wr)
    scale = float(np.abs(want).max())
    for v in (220, 303):
        gpu_engine.set_tuning(variant=v)
        try:
            y, flag = _run(gpu_engine, case, x, w, b, wr, want_flag=True)
        finally:
            gpu_engine.set_tuning(variant=-1)
        assert not flag
        err = float(np.abs(y - want).max()) / scale
        print(f"x scale {xs} w scale {ws} tile {v}: rel err {err:.2e}")
        assert err < 2e-5 if xs < 1e-5 else err < 3e-6, f"tile {v}: rel err {err:.2e}"


def test_h2_overflow_flag(gpu_engine):
    """|x| > 65504 cannot be an fp16 pair: the encoder clamps, the model's flag goes up once and is cleared by the read."""
    case = (1, 16, 24, 32, 32, 3, 1, G.ACT_NONE, False)
    x, w, b, wr = _data(case)
    g = G.Graph(task=G.TASK_TRACKNET, dtype=G.DTYPE_H2)
    b0 = g.buf(0, 32)
    mid = g.buf(0, 32)
    out = g.buf(0, 32)
    g.conv((b0, 0, 32), (mid, 0), (w * 3e4).astype(np.float32), b, 3, 1, G.ACT_NONE)     # outputs ~ 1e5: do not fit
    g.conv((mid, 0, 32), (out, 0), w, b, 3, 1, G.ACT_NONE)
    g.head_buf = (out, -1, -1)
    m = E.Model(gpu_engine, g)
    m.set_max_batch(1)
    y = m.tracknet_infer(x)
    assert np.isfinite(y).all()
    assert m.take_overflow() and not m.take_overflow()
    y2 = m.tracknet_infer((x * 1e-3).astype(np.float32))
    assert np.isfinite(y2).all() and not m.take_overflow()
    xin = x.copy()
    xin[0, 3, 5, 7] = 1e6                                                               # the input encoder raises it too
    m.tracknet_infer(xin)
    assert m.take_overflow()
    m.close()


@pytest.mark.parametrize("shape", [(2, 24, 40, 64, 32, 80, 1), (1, 18, 28, 32, 48, 96, 1),
                                   (2, 24, 48, 64, 32, 80, 3), (1, 16, 32, 32, 96, 48, 3)],
                         ids=["up64+32", "up32+48", "3x3-up64+32", "3x3-up32+96"])
def test_h2_upsample_absorbed(gpu_engine, shape):
    """Upsample(2) + cat in front of a stride-1 conv read from the coarse map (h2 1x1 tap kernel, 3x3 patch kernel):
    bitwise equal to running the upsample kernel."""
    B, H, W, c_up, c_skip, cout, k = shape
    rng = np.random.default_rng(c_up * 7 + c_skip + k)
    cin0 = 32
    x = rng.normal(0, 1, (B, H, W, cin0)).astype(np.float32)
    w_dn = rng.normal(0, (2.0 / (cin0 * 9)) ** 0.5, (c_up, cin0, 3, 3)).astype(np.float32)
    w_sk = rng.normal(0, (2.0 / cin0) ** 0.5, (c_skip, cin0, 1, 1)).astype(np.float32)
    w = rng.normal(0, (2.0 / ((c_up + c_skip) * k * k)) ** 0.5, (cout, c_up + c_skip, k, k)).astype(np.float32)
    b = rng.normal(0, 0.5, cout).astype(np.float32)

    def run(**tuning):
        g = G.Graph(task=G.TASK_TRACKNET, dtype=G.DTYPE_H2)
        b0 = g.buf(0, cin0)
        coarse = g.buf(1, c_up)
        cat = g.buf(0, c_up + c_skip)
        out = g.buf(0, G.pad16(cout))
        g.conv((b0, 0, cin0), (coarse, 0), w_dn, np.zeros(c_up, np.float32), 3, 2, G.ACT_SILU)
        g.ops.append(dict(kind=G.OP_UPSAMPLE2X, in_buf=coarse, in_choff=0, cin=c_up, out_buf=cat, out_choff=0, cout=c_up,
                          ksize=0, stride=0, act=0, res_buf=-1, res_choff=0, npad=0, w_off=0, b_off=0))
        g.conv((b0, 0, cin0), (cat, c_up), w_sk, np.zeros(c_skip, np.float32), 1, 1, G.ACT_NONE)
        g.conv((cat, 0, c_up + c_skip), (out, 0), w, b, k, 1, G.ACT_SILU)
        g.head_buf = (out, -1, -1)
        gpu_engine.set_tuning(**tuning)
        gpu_engine.set_profiling(True)
        m = E.Model(gpu_engine, g)
        m.set_max_batch(B)
        y = m.tracknet_infer(x)[..., :cout]
        n_up = sum(1 for r in m.profile_rows() if r["kind"] == G.OP_UPSAMPLE2X)
        m.close()
        gpu_engine.set_profiling(False)
        return y, n_up

    absorbing = (220, 209, 213, 207, 243, 239) if k == 1 else (303, 304)
    keeping = () if k == 1 else (220,)
    try:
        ref, n_up = run(variant=220 if k == 1 else 303, fold_up=0)
        assert n_up == 1
        for v in absorbing + keeping:
            y, n_up = run(variant=v, fold_up=1)
            assert n_up == (0 if v in absorbing else 1), f"variant {v}: {n_up} upsample launches"
            if v in H2_SINGLE_LEVEL:
                assert float(np.abs(y - ref).max()) < 2e-6 * max(1.0, float(np.abs(ref).max()))
            else:
                assert np.array_equal(y, ref), f"variant {v}: absorbed upsample differs (max {np.abs(y - ref).max():.3e})"
    finally:
        gpu_engine.set_tuning(variant=-1, fold_up=1)


def test_h2_helper_kernels(gpu_engine):
    """MaxPool2d(2,2), the (unabsorbed) upsample and the SPPF pools on h2 buffers: the winning PAIR is copied, so the
    result is exactly the pooling of the values the pairs stand for."""
    B, H, W, c = 2, 24, 40, 32
    rng = np.random.default_rng(5)
    x = rng.normal(0, 3, (B, H, W, c)).astype(np.float32)
    xr = G.h2_decode_nhwc(G.h2_encode_nhwc(x))                       # what the input buffer holds
    g = G.Graph(task=G.TASK_TRACKNET, dtype=G.DTYPE_H2)
    b0 = g.buf(0, c)
    small = g.buf(1, c)
    cat = g.buf(0, 4 * c)
    op = lambda kind, src, dst, cin, cout, k, s: dict(kind=kind, in_buf=src[0], in_choff=src[1], cin=cin, out_buf=dst[0], out_choff=dst[1],
                                                      cout=cout, ksize=k, stride=s, act=0, res_buf=-1, res_choff=0, npad=0, w_off=0, b_off=0)
    g.ops.append(op(G.OP_MAXPOOL2, (b0, 0), (small, 0), c, c, 2, 2))
    g.ops.append(op(G.OP_UPSAMPLE2X, (small, 0), (cat, 0), c, c, 0, 0))          # its reader is a pool: never absorbed
    g.ops.append(op(G.OP_SPPF_POOL, (cat, 0), (cat, c), c, 3 * c, 5, 1))
    g.head_buf = (cat, -1, -1)                                        # read back raw: the pools wrote PAIRS into it
    m = E.Model(gpu_engine, g)
    m.set_max_batch(B)
    got = G.h2_decode_nhwc(m.tracknet_infer(x))
    m.close()
    t = torch.from_numpy(xr).permute(0, 3, 1, 2)
    ys = [F.interpolate(F.max_pool2d(t, 2, 2), scale_factor=2.0, mode="nearest")]
    for _ in range(3):
        ys.append(F.max_pool2d(ys[-1], 5, 1, 2))
    want = torch.cat(ys, 1).permute(0, 2, 3, 1).numpy()
    assert np.array_equal(got, want), float(np.abs(got - want).max())


@pytest.mark.parametrize("mode", ["h2", "bx3"])
def test_stale_arena_bytes_cannot_reach_results(gpu_engine, mode):
    """activation buffers alias inside one arena, so whatever a conv reads beyond what its producers wrote
    (pad channels, partial groups, absorbed-upsample slices) would be stale data of another layer — NaN if that layer
    overflowed.  The h2 / fp32 kernels read no such byte: fill the arena with NaN patterns between two identical
    inferences (YOLOv8 detect with its concat buffers and absorbed upsamples; the TrackNet U-Net) — results unchanged."""
    from oracle import tracknet_ref as tr
    from padel_analytics_amd import yolo_arch
    from tests import synth
    frames = synth.synthetic_frames(2, 360, 640, seed=4)
    sd = yolo_arch.synth_state_dict("n", 80, None, seed=2, cls_bias=-1.0)
    m = E.Model(gpu_engine, G.build_yolov8(sd, 80, None, dtype=E.graph_dtype(mode)))
    m.set_max_batch(2)
    kw = dict(imgsz=640, conf=0.25, iou=0.7)
    b0, _, c0 = m.yolo_infer(frames, 2, 360, 640, **kw)
    h0 = [m.read_head(l, 2) for l in range(3)]
    m.fill_arena(0xFF)
    b1, _, c1 = m.yolo_infer(frames, 2, 360, 640, **kw)
    h1 = [m.read_head(l, 2) for l in range(3)]
    assert np.array_equal(c0, c1) and np.array_equal(b0, b1)
    for a, b in zip(h0, h1):
        assert np.isfinite(b).all() and np.array_equal(a, b)
    m.close()
    g = G.build_tracknet(tr.synth_tracknet_state_dict(2), dtype=E.graph_dtype(mode))
    t = E.Model(gpu_engine, g)
    t.set_max_batch(1)
    x = np.random.default_rng(0).uniform(0, 1, (1, 64, 96, 32)).astype(np.float32)
    x[..., 27:] = 0
    y0 = t.tracknet_infer(x)
    t.fill_arena(0xFF)
    y1 = t.tracknet_infer(x)
    assert np.isfinite(y1).all() and np.array_equal(y0, y1)
    t.close()


@pytest.mark.parametrize("nc,kpt", [(80, None), (1, (13, 3))], ids=["detect", "pose"])
def test_head_branches_as_separate_convs_match_the_merged_conv(gpu_engine, nc, kpt, monkeypatch):
    """graph.HEAD_SPLIT (automatic on h2 m / l / x graphs): the first 3x3 conv of every Detect / Pose branch as its own op into
    its slice of the shared buffer, so that every branch gets the tile that fits its width.  A row of the weight matrix does
    not know its neighbours and results do not depend on the tile: head maps and detections bitwise those of the merged conv."""
    from padel_analytics_amd import yolo_arch
    from tests import synth
    h, w = 360, 640
    frames = synth.synthetic_frames(2, h, w, seed=4)
    sd = yolo_arch.synth_state_dict("m", nc, kpt, seed=6, cls_bias=-1.0)
    res = []
    for split in (False, True):
        monkeypatch.setattr(G, "HEAD_SPLIT", split)
        m = E.Model(gpu_engine, G.build_yolov8(sd, nc, kpt, dtype="h2"))
        m.set_max_batch(2)
        try:
            b, k, c = m.yolo_infer(frames, 2, h, w, imgsz=640, conf=0.25, iou=0.7)
            heads = [m.read_head(l, 2) for l in range(3)]
            assert not m.take_overflow()
        finally:
            m.close()
        res.append((b, k, c, heads))
    (b0, k0, c0, h0), (b1, k1, c1, h1) = res
    for l in range(3):
        assert np.array_equal(h0[l].view(np.uint32), h1[l].view(np.uint32)), f"head {l} differs"
    assert np.array_equal(c0, c1) and np.array_equal(b0.view(np.uint32), b1.view(np.uint32)) and int(c0.sum()) > 0
    if kpt:
        assert np.array_equal(k0.view(np.uint32), k1.view(np.uint32))


@pytest.mark.parametrize("scale,hw,imgsz", [("n", (360, 640), 640), ("s", (180, 320), 288), ("m", (720, 1280), 1280), ("n", (1080, 1920), 1920)],
                         ids=["n-640", "s-288", "m-1280", "n-1920"])
def test_fused_sppf_matches_three_pool_launches(gpu_engine, scale, hw, imgsz):
    """Tuning "fuse_sppf" (sppf_h2_kernel, on by default): the three chained MaxPool2d(5, 1, 2) of SPPF as one kernel — keys in
    LDS, a row pass and a column pass per level.  Every h2 max-pool orders pairs the same way (value, then the pair's bits), so
    the separable walk picks the pairs the 25-tap walks pick: head maps and detections are bitwise those of the three
    launches (P5 maps of 12 x 20, 5 x 9 (partial everything), 23 x 40 and 34 x 60: the largest map that fits the LDS planes)."""
    from padel_analytics_amd import yolo_arch
    from tests import synth
    h, w = hw
    frames = synth.synthetic_frames(2, h, w, seed=12)
    sd = yolo_arch.synth_state_dict(scale, 80, None, seed=5, cls_bias=-1.0)
    m = E.Model(gpu_engine, G.build_yolov8(sd, 80, None, dtype="h2"))
    m.set_max_batch(2)
    kw = dict(imgsz=imgsz, conf=0.25, iou=0.7)
    gpu_engine.set_profiling(True)
    try:
        gpu_engine.set_tuning(fuse_sppf=0)
        b0, _, c0 = m.yolo_infer(frames, 2, h, w, **kw)
        h0 = [m.read_head(l, 2) for l in range(3)]
        assert not m.take_overflow()
        gpu_engine.set_tuning(fuse_sppf=1)
        b1, _, c1 = m.yolo_infer(frames, 2, h, w, **kw)
        h1 = [m.read_head(l, 2) for l in range(3)]
        assert not m.take_overflow()
    finally:
        gpu_engine.set_tuning(fuse_sppf=1)           # the default
        gpu_engine.set_profiling(False)
        m.close()
    for l in range(3):
        assert np.array_equal(h0[l].view(np.uint32), h1[l].view(np.uint32)), f"head {l} differs"
    assert np.array_equal(c0, c1) and np.array_equal(b0.view(np.uint32), b1.view(np.uint32))
    assert int(c1.sum()) > 0


@pytest.mark.parametrize("scale,hw,imgsz", [("n", (360, 640), 640), ("s", (180, 320), 288), ("m", (180, 320), 288), ("m", (360, 640), 640)],
                         ids=["n-640", "s-288", "m-288", "m-640"])
def test_fused_stem_layer1_matches_unfused(gpu_engine, scale, hw, imgsz):
    """Tuning "fuse_stem" (csrc/stem_l1_h2.hip, on by default since round 4): model.0 + model.1 as one kernel — the stem map stays in
    LDS (c = 16: tail-only K walk; 32: one chunk; 48: chunk + tail with the LDS region reused; partial tiles at 288).  Layer 1 is
    the arithmetic of the kernel it replaces; the stem phase runs on the f16 matrix pipe since round 5 — sum (w / 255) v over the
    exact input bytes v with w / 255 as an fp16 pair, two MFMAs per fragment, instead of sum w (v / 255) on the fp32-input MFMA:
    the same numbers to fp32 rounding, so the head maps of the two runs agree like two fp32 evaluations of one graph do
    (measured 2e-6 .. 6e-6 of the largest head value) and the detections are the same set within a fraction of the noise floor."""
    from padel_analytics_amd import yolo_arch
    from tests import synth
    h, w = hw
    frames = synth.synthetic_frames(3, h, w, seed=9)
    sd = yolo_arch.synth_state_dict(scale, 80, None, seed=3, cls_bias=-1.0)
    m = E.Model(gpu_engine, G.build_yolov8(sd, 80, None, dtype="h2"))
    m.set_max_batch(3)
    kw = dict(imgsz=imgsz, conf=0.25, iou=0.7)
    gpu_engine.set_profiling(True)
    try:
        gpu_engine.set_tuning(fuse_stem=0)
        b0, _, c0 = m.yolo_infer(frames, 3, h, w, **kw)
        h0 = [m.read_head(l, 3) for l in range(3)]
        n_convs0 = sum(1 for r in m.profile_rows() if r["kind"] == 2)
        assert not m.take_overflow()
        gpu_engine.set_tuning(fuse_stem=1)
        b1, _, c1 = m.yolo_infer(frames, 3, h, w, **kw)
        h1 = [m.read_head(l, 3) for l in range(3)]
        n_convs1 = sum(1 for r in m.profile_rows() if r["kind"] == 2)
        assert not m.take_overflow()
    finally:
        gpu_engine.set_tuning(fuse_stem=1)           # the default
        gpu_engine.set_profiling(False)
        m.close()
    assert n_convs1 == n_convs0 - 1, "the fused kernel did not run (layer 1 was launched on its own)"
    for l, (x, y) in enumerate(zip(h0, h1)):
        assert np.isfinite(y).all()
        rel = float(np.abs(x - y).max()) / max(1.0, float(np.abs(x).max()))
        print(f"fused vs unfused stem, {scale} {hw}: head {l} rel diff {rel:.2e}")
        assert rel <= 3e-5, f"head {l}: relative difference {rel:.3e}"
    assert np.array_equal(c0, c1) and int(c0.sum()) > 0
    assert np.array_equal(b0[..., 5], b1[..., 5]) and float(np.abs(b0[..., :4] - b1[..., :4]).max()) <= 2e-2 and float(np.abs(b0[..., 4] - b1[..., 4]).max()) <= 1e-4


@pytest.mark.parametrize("shape", [(10, 50, 70, 64, 192, True), (13, 41, 100, 96, 96, False), (4, 64, 96, 192, 288, True)], ids=["2-chunks", "3-chunks", "6-chunks"])
def test_h2r_persistent_workgroups_walk_many_tiles(gpu_engine, shape):
    """Round 6: the register-weights quad kernel (conv_patch_h2r.hip, tile 324) runs 2 PERSISTENT workgroups per CU; a workgroup
    requests its next tile's first patch chunk during the current tile's second-last chunk.  The tile tests above have fewer
    tiles than workgroups; here every workgroup walks several tiles (partial patches in y and x, several channel tiles, residual):
    bitwise the quad kernel (323), the 48-channel patch tile (303) and its own one-workgroup-per-tile form (tuning bit 2), twice."""
    B, H, W, cin, cout, use_res = shape
    rng = np.random.default_rng(cin + cout + H)
    x = rng.normal(0, 1, (B, H, W, cin)).astype(np.float32)
    w = rng.normal(0, (2.0 / (cin * 9)) ** 0.5, (cout, cin, 3, 3)).astype(np.float16).astype(np.float32)
    scale = rng.uniform(0.5, 2.0, cout).astype(np.float32)
    b = rng.normal(0, 0.5, cout).astype(np.float32)
    wr = rng.normal(0, (1.0 / cin) ** 0.5, (cout, cin, 1, 1)).astype(np.float32)
    tiles = B * ((H + 7) // 8) * ((W + 15) // 16) * ((cout + 95) // 96)
    assert tiles > 512, "the point of the test: more tiles than persistent workgroups"

    def run():
        g = G.Graph(task=G.TASK_TRACKNET, dtype=G.DTYPE_H2)
        b0 = g.buf(0, cin)
        res = None
        if use_res:
            b2 = g.buf(0, G.pad16(cout))
            g.conv((b0, 0, cin), (b2, 0), wr, np.zeros(cout, np.float32), 1, 1, G.ACT_NONE)
            res = (b2, 0)
        b1 = g.buf(0, G.pad16(cout))
        g.conv((b0, 0, cin), (b1, 0), w, b, 3, 1, G.ACT_SILU, res=res, out_scale=scale)
        assert g.ops[-1]["flags"] & G.FLAG_W_SINGLE
        g.head_buf = (b1, -1, -1)
        m = E.Model(gpu_engine, g)
        m.set_max_batch(B)
        y = m.tracknet_infer(x)[..., :cout]
        assert not m.take_overflow()
        m.close()
        return y

    outs = {}
    try:
        for name, kw in (("303", dict(variant=303)), ("323", dict(variant=323)), ("324", dict(variant=324)), ("324 again", dict(variant=324)),
                         ("324 one workgroup per tile", dict(variant=324, tune=5)), ("325", dict(variant=325)), ("325 again", dict(variant=325)),
                         ("325 three products", dict(variant=325, w_single=0)), ("325 one workgroup per tile", dict(variant=325, tune=5))):
            gpu_engine.set_tuning(**{"tune": 1, **kw})
            outs[name] = run()
    finally:
        gpu_engine.set_tuning(variant=-1, tune=1, w_single=1)
    xt = torch.from_numpy(x).permute(0, 3, 1, 2).double()
    want = F.silu(F.conv2d(xt, torch.from_numpy(w).double() * torch.from_numpy(scale).double()[:, None, None, None], torch.from_numpy(b).double(), padding=1))
    if use_res:
        want = want + F.conv2d(xt, torch.from_numpy(wr).double())
    want = want.permute(0, 2, 3, 1).numpy()
    ref = outs["303"]
    assert float(np.abs(ref - want).max()) / max(1.0, float(np.abs(want).max())) < 3e-6
    for name, y in outs.items():
        assert np.array_equal(y, ref), f"{name} differs from the 48-channel patch tile (max {np.abs(y - ref).max():.3e}, {int((y != ref).sum())} values)"


def test_w_single_promise_is_checked_against_the_blob(gpu_engine):
    """PA_CONV_W_SINGLE is the caller's promise (public C-ABI) that a conv's packed m plane is all zero.  A blob that
    breaks it must fail the call — the two-product kernels would otherwise skip a non-zero product silently."""
    cin, cout = 64, 96
    rng = np.random.default_rng(5)
    w = rng.normal(0, 0.05, (cout, cin, 3, 3)).astype(np.float32)                 # not fp16 numbers: the m plane is populated
    g = G.Graph(task=G.TASK_TRACKNET, dtype=G.DTYPE_H2)
    b0, b1 = g.buf(0, cin), g.buf(0, cout)
    g.conv((b0, 0, cin), (b1, 0), w, np.zeros(cout, np.float32), 3, 1, G.ACT_SILU)
    g.head_buf = (b1, -1, -1)
    assert not (g.ops[-1]["flags"] & G.FLAG_W_SINGLE)
    g.ops[-1]["flags"] |= G.FLAG_W_SINGLE                                          # the lie
    m = E.Model(gpu_engine, g)
    m.set_max_batch(1)
    try:
        with pytest.raises(E.EngineError, match="PA_CONV_W_SINGLE"):
            m.tracknet_infer(rng.normal(0, 1, (1, 16, 16, cin)).astype(np.float32))
    finally:
        m.close()


def test_engine_gather_bytes_single_rank(gpu_engine):
    """ABI v5: pa_engine_gather_sizes / pa_engine_gather — with one rank (all a 1-GPU box has) the host-copy form; the N > 1 form
    is covered by the gloo stand-in (tests/test_bench_gloo.py) and by the driver's multi-GPU bench."""
    buf = np.arange(1000, dtype=np.uint8)
    got = gpu_engine.gather_bytes(buf, 0)
    assert len(got) == 1 and np.array_equal(got[0], buf)
    assert gpu_engine.gather_bytes(np.zeros(0, np.uint8), 0)[0].size == 0


@pytest.mark.parametrize("cin,cout,w16", [(48, 48, True), (32, 64, False), (16, 32, True), (48, 96, False)], ids=["48-48-two", "32-64-three", "16-32-two", "48-96-three"])
def test_h2v_register_weight_wide_kernel_equals_the_wide_kernel(gpu_engine, cin, cout, w16):
    """Round 6: conv_patch_h2v.hip (the wide patch tile with the weights global -> VGPR, no per-step barrier) against
    conv_patch_h2w.hip (tuning bit 3) and the 48-channel patch tile, bitwise, on partial tiles with a residual, twice."""
    B, H, W = 3, 37, 50
    rng = np.random.default_rng(cin * 7 + cout)
    x = rng.normal(0, 1, (B, H, W, cin)).astype(np.float32)
    w = rng.normal(0, (2.0 / (cin * 9)) ** 0.5, (cout, cin, 3, 3)).astype(np.float32)
    scale = None
    if w16:
        w = w.astype(np.float16).astype(np.float32)
        scale = rng.uniform(0.5, 2.0, cout).astype(np.float32)
    b = rng.normal(0, 0.5, cout).astype(np.float32)
    wr = rng.normal(0, (1.0 / cin) ** 0.5, (cout, cin, 1, 1)).astype(np.float32)

    def run():
        g = G.Graph(task=G.TASK_TRACKNET, dtype=G.DTYPE_H2)
        b0 = g.buf(0, cin)
        b2 = g.buf(0, G.pad16(cout))
        g.conv((b0, 0, cin), (b2, 0), wr, np.zeros(cout, np.float32), 1, 1, G.ACT_NONE)
        b1 = g.buf(0, G.pad16(cout))
        g.conv((b0, 0, cin), (b1, 0), w, b, 3, 1, G.ACT_SILU, res=(b2, 0), out_scale=scale)
        assert bool(g.ops[-1]["flags"] & G.FLAG_W_SINGLE) == w16
        g.head_buf = (b1, -1, -1)
        m = E.Model(gpu_engine, g)
        m.set_max_batch(B)
        y = m.tracknet_infer(x)[..., :cout]
        m.close()
        return y

    outs = {}
    try:
        for nf in (1, 2, 3):
            for name, tune in ((f"h2v nf {nf}", 1), (f"h2v nf {nf} again", 1), (f"h2w nf {nf}", 9)):
                gpu_engine.set_tuning(variant=340 + nf, tune=tune)
                outs[name] = run()
        gpu_engine.set_tuning(variant=303, tune=1)
        outs["303"] = run()
    finally:
        gpu_engine.set_tuning(variant=-1, tune=1)
    ref = outs["303"]
    for name, y in outs.items():
        assert np.array_equal(y, ref), f"{name} differs from the 48-channel patch tile (max {np.abs(y - ref).max():.3e}, {int((y != ref).sum())} values)"


def test_profile_rows_flag_the_convs_that_read_a_residual(gpu_engine):
    """bench.py prices a conv's HBM traffic against input + output + weights + the residual input it adds: the per-op profile
    rows (pa_model_profile_text, 11th column) say which convs have one.  yolov8n: every second 3x3 of the backbone's bottlenecks
    (C2f with shortcut: model.2 / 4 / 6 / 8 -> 1 + 2 + 2 + 1 blocks), none in the neck, no 1x1."""
    from padel_analytics_amd import yolo_arch
    from tests import synth
    frames = synth.synthetic_frames(1, 180, 320, seed=2)
    sd = yolo_arch.synth_state_dict("n", 80, None, seed=3, cls_bias=-1.0)
    m = E.Model(gpu_engine, G.build_yolov8(sd, 80, None, dtype="h2"))
    m.set_max_batch(1)
    gpu_engine.set_profiling(True)
    try:
        m.yolo_infer(frames, 1, 180, 320, imgsz=320, conf=0.25, iou=0.7)
        rows = m.profile_rows()
    finally:
        gpu_engine.set_profiling(False)
        m.close()
    convs = [r for r in rows if r["kind"] == G.OP_CONV]
    assert convs and all("res" in r for r in convs)
    with_res = [r for r in convs if r["res"]]
    assert len(with_res) == 6, [(r["ksize"], r["cin"], r["cout"]) for r in with_res]
    assert all(r["ksize"] == 3 and r["stride"] == 1 and r["cin"] == r["cout"] for r in with_res)
