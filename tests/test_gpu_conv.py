"""GPU unit test of the fp32-storage conv kernels through the C-ABI (the h2 kernels — the default arithmetic since round 3
— have their own file, tests/test_gpu_h2.py): every tile variant of the fp32-input MFMA generations (tap-unrolled LDS-DMA
kernels conv_tap.hip; the register-staged LDS kernel of round 1 that used to cross-check them is retired: tools/legacy_conv/conv_lds.hip) and of the bf16x3 kernels
(conv_tap_bx3.hip / conv_patch_bx3.hip: the full-range fallback of h2) against torch.nn.functional.conv2d (fp64 CPU), on
shapes that exercise stride 2, 1x1, the 16-channel K tail (cin % 32 == 16) including the full-chunk -> tail wrap of the tap
kernel's request ring, partial channel tiles (cout = 80 -> 5 fragments), the M tail, the fused residual and every
activation; each family BITWISE equal across its own tiles (same K order + same accumulation blocks)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from padel_analytics_amd import engine as E, graph as G

pytestmark = pytest.mark.gpu

# (B, H, W, cin, cout, k, stride, act, residual)
CASES = [
    (2, 24, 40, 32, 64, 3, 1, G.ACT_SILU, False),
    (3, 20, 36, 48, 80, 3, 1, G.ACT_SILU, True),      # full chunk -> K-tail wrap + partial N tile + residual
    (2, 32, 48, 16, 16, 3, 2, G.ACT_RELU, False),     # stride 2, smallest channels (tail block only)
    (1, 16, 24, 96, 96, 1, 1, G.ACT_NONE, False),     # 1x1, no activation
    (2, 12, 20, 64, 144, 3, 1, G.ACT_SIGMOID, False),
    (1, 36, 28, 288, 48, 1, 1, G.ACT_SILU, True),
    (1, 8, 12, 576, 192, 3, 1, G.ACT_SILU, False),    # long K: exercises several accumulation blocks
    (2, 32, 48, 64, 96, 3, 2, G.ACT_SILU, True),      # stride 2 with cin % 32 == 0, residual at the output resolution
    (3, 17, 23, 96, 96, 3, 1, G.ACT_SILU, True),      # odd spatial size: M tail + borders in every tile, residual
    (2, 16, 20, 80, 48, 3, 1, G.ACT_SILU, False),     # two full chunks + tail (cin 80), 48 outputs (128x48 tile)
    (2, 18, 26, 16, 32, 3, 1, G.ACT_SILU, False),     # stride 1 with the tail block only (n-scale's 16 channels), partial patches
    (2, 20, 24, 64, 192, 3, 1, G.ACT_SILU, True),     # two full 96-channel tiles (quad patch kernel: both channel halves), partial patches in y and x
    (1, 20, 27, 688, 96, 1, 1, G.ACT_SILU, False),    # 1x1 with long K: 21 full chunks + a 16-channel tail = three accumulation blocks (9 + 9 + 4), M tail
    (3, 18, 22, 96, 208, 3, 2, G.ACT_SILU, True),     # stride 2 to an odd map (9 x 11 outputs: M tail, every border), 13 fragments = two 192-channel tiles, residual
]

TAP_VARIANTS = (6, 7, 9, 10, 11, 12, 13, 14, 15, 20)
BX3_VARIANTS = (6, 7, 9, 11, 12, 13, 14, 20, 25, 206, 207, 209, 211, 220, 225, 213, 303, 304, 306)      # bf16x3 kernels (conv_tap_bx3.hip): fp32 accuracy, own rounding


def _run(eng, case, x, w, b, wr):
    """Graph: [op0: 1x1 conv (stride s, no act) x -> residual buffer]  op1: the conv under test (+ residual)."""
    B, H, W, cin, cout, k, s, act, use_res = case
    g = G.Graph(task=G.TASK_TRACKNET)
    b0 = g.buf(0, cin)
    lvl = 1 if s == 2 else 0
    b1 = g.buf(lvl, G.pad16(cout))
    res = None
    if use_res:
        b2 = g.buf(lvl, G.pad16(cout))
        g.conv((b0, 0, cin), (b2, 0), wr, np.zeros(cout, np.float32), 1, s, G.ACT_NONE)
        res = (b2, 0)
    g.conv((b0, 0, cin), (b1, 0), w, b, k, s, act, res=res)
    g.head_buf = (b1, -1, -1)
    m = E.Model(eng, g)
    m.set_max_batch(B)
    y = m.tracknet_infer(x)[..., :cout]
    m.close()
    return y


@pytest.mark.parametrize("case", CASES, ids=[f"c{i}" for i in range(len(CASES))])
def test_conv_variants(gpu_engine, case):
    B, H, W, cin, cout, k, s, act, use_res = case
    rng = np.random.default_rng(cin * 131 + cout)
    x = rng.normal(0, 1, (B, H, W, cin)).astype(np.float32)
    w = rng.normal(0, (2.0 / (cin * k * k)) ** 0.5, (cout, cin, k, k)).astype(np.float32)
    b = rng.normal(0, 0.5, cout).astype(np.float32)
    wr = rng.normal(0, (1.0 / cin) ** 0.5, (cout, cin, 1, 1)).astype(np.float32)
    xt = torch.from_numpy(x).permute(0, 3, 1, 2)
    want = F.conv2d(xt.double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), stride=s, padding=k // 2)
    want = {G.ACT_SILU: F.silu, G.ACT_RELU: F.relu, G.ACT_SIGMOID: torch.sigmoid, G.ACT_NONE: lambda t: t}[act](want)
    if use_res:
        want = want + F.conv2d(xt.double(), torch.from_numpy(wr).double(), stride=s)
    want = want.permute(0, 2, 3, 1).numpy()
    scale = max(1.0, float(np.abs(want).max()))
    outs = {}
    try:
        for v in TAP_VARIANTS:                            # LDS-DMA ring: twice, a DMA / barrier race is not deterministic
            gpu_engine.set_tuning(impl=0, variant=v)
            for rep in range(2):
                outs[f"T{v}.{rep}"] = _run(gpu_engine, case, x, w, b, wr)
        gpu_engine.set_tuning(impl=0, variant=7, tap_pd=3)   # 1x1 tap kernel with prefetch distance 3
        outs["T7.pd3"] = _run(gpu_engine, case, x, w, b, wr)
        gpu_engine.set_tuning(impl=0, variant=-1, tap_pd=2)
        outs["auto"] = _run(gpu_engine, case, x, w, b, wr)
        gpu_engine.set_tuning(graph=1)
        outs["auto.graph"] = _run(gpu_engine, case, x, w, b, wr)
        gpu_engine.set_tuning(graph=0, alias=0)
        outs["auto.noalias"] = _run(gpu_engine, case, x, w, b, wr)
    finally:
        gpu_engine.set_tuning(impl=2, variant=-1, tap_pd=2, graph=0, alias=1)
    ref_name, ref = next(iter(outs.items()))
    for name, y in outs.items():
        assert y.shape == want.shape
        err = float(np.abs(y - want).max()) / scale
        assert err < 3e-6, f"{name}: rel err {err:.2e} vs fp64 conv2d"
        assert np.array_equal(y, ref), f"{name} differs bitwise from {ref_name} (max {np.abs(y - ref).max():.3e})"
    # ---- bf16x3: same accuracy bar against fp64, bitwise equal among its own tiles, and its RMS error not worse
    # than the fp32 MFMA kernels' (the admission criterion for making it the default)
    outs3 = {}
    try:
        for v in BX3_VARIANTS:
            gpu_engine.set_tuning(impl=2, variant=v)
            for rep in range(2):
                outs3[f"B{v}.{rep}"] = _run(gpu_engine, case, x, w, b, wr)
        gpu_engine.set_tuning(impl=2, variant=-1)
        outs3["B.auto"] = _run(gpu_engine, case, x, w, b, wr)
    finally:
        gpu_engine.set_tuning(impl=2, variant=-1)
    n3, r3 = next(iter(outs3.items()))
    for name, y in outs3.items():
        err = float(np.abs(y - want).max()) / scale
        assert err < 3e-6, f"{name}: rel err {err:.2e} vs fp64 conv2d"
        assert np.array_equal(y, r3), f"{name} differs bitwise from {n3} (max {np.abs(y - r3).max():.3e})"
    rms32 = float(np.sqrt(np.mean((ref - want) ** 2)))
    rms3 = float(np.sqrt(np.mean((r3 - want) ** 2)))
    print(f"case {case}: RMS error vs fp64  fp32-MFMA {rms32:.3e}  bf16x3 {rms3:.3e}")
    assert rms3 <= 1.25 * rms32 + 1e-9, (rms3, rms32)


@pytest.mark.parametrize("shape", [(2, 24, 40, 64, 32, 80, 1), (1, 18, 28, 32, 48, 96, 1), (3, 16, 16, 96, 16, 48, 1),
                                   (2, 24, 48, 64, 32, 80, 3), (1, 16, 32, 32, 96, 48, 3), (2, 40, 16, 96, 32, 64, 3)],
                         ids=["up64+32", "up32+48", "up96+16", "3x3-up64+32", "3x3-up32+96", "3x3-up96+32"])
def test_upsample_absorbed(gpu_engine, shape):
    """Upsample(2) + cat in front of a stride-1 conv is never materialised — the bf16x3 1x1 kernel (YOLOv8's
    FPN joins) and the 3x3 patch kernel (TrackNet's decoder blocks) read the first channels at [y >> 1][x >> 1] of the
    coarse map (csrc/engine.cpp:find_upsample_folds).  Same arithmetic on the same values: bitwise equal to running the
    upsample kernel, for every tile that has the absorbing instantiation; tiles without it keep the upsample kernel."""
    B, H, W, c_up, c_skip, cout, k = shape
    rng = np.random.default_rng(c_up * 7 + c_skip + k)
    cin0 = 32
    x = rng.normal(0, 1, (B, H, W, cin0)).astype(np.float32)
    w_dn = rng.normal(0, (2.0 / (cin0 * 9)) ** 0.5, (c_up, cin0, 3, 3)).astype(np.float32)
    w_sk = rng.normal(0, (2.0 / cin0) ** 0.5, (c_skip, cin0, 1, 1)).astype(np.float32)
    w = rng.normal(0, (2.0 / ((c_up + c_skip) * k * k)) ** 0.5, (cout, c_up + c_skip, k, k)).astype(np.float32)
    b = rng.normal(0, 0.5, cout).astype(np.float32)

    def run(**tuning):
        g = G.Graph(task=G.TASK_TRACKNET)
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

    # tile ids with an absorbing instantiation: every 1x1 id maps to one; 3x3: the patch kernel (30x) only
    absorbing = (-1, 220, 209, 213, 7) if k == 1 else (-1, 303, 304, 306)
    keeping = () if k == 1 else (220, 213)
    try:
        ref, n_up = run(impl=2, variant=-1, fold_up=0)
        assert n_up == 1
        for v in absorbing + keeping:
            y, n_up = run(impl=2, variant=v, fold_up=1)
            if v == -1 and k == 3:
                assert n_up in (0, 1)                   # the per-layer heuristic may prefer a tap tile on a tiny map
            else:
                assert n_up == (0 if v in absorbing else 1), f"variant {v}: {n_up} upsample launches"
            assert np.array_equal(y, ref), f"variant {v}: absorbed upsample differs (max {np.abs(y - ref).max():.3e})"
        y, n_up = run(impl=0, variant=-1, fold_up=1)        # the fp32-MFMA kernels keep the upsample kernel
        assert n_up == 1
    finally:
        gpu_engine.set_tuning(impl=2, variant=-1, fold_up=1)
