"""GPU unit test of the conv kernels through the C-ABI: every tile variant of all five kernel generations — the
register-direct kernel (conv_igemm.hip), the LDS kernel (conv_lds.hip), the pipelined LDS kernel (conv_pipe.hip), the
LDS-DMA ring (conv_ring.hip) and the default tap-unrolled kernels (conv_tap.hip) — against
torch.nn.functional.conv2d (fp64 CPU), on
shapes that exercise stride 2, 1x1, the 16-channel K tail (cin % 32 == 16), partial channel tiles
(cout = 80 -> 5 fragments), the M tail, the fused residual and every activation; and against each other
BITWISE (same K order + same accumulation blocks => identical results)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from padel_analytics_amd import engine as E, graph as G

pytestmark = pytest.mark.gpu

# (B, H, W, cin, cout, k, stride, act, residual)
CASES = [
    (2, 24, 40, 32, 64, 3, 1, G.ACT_SILU, False),
    (3, 20, 36, 48, 80, 3, 1, G.ACT_SILU, True),      # K tail + partial N tile + residual
    (2, 32, 48, 16, 16, 3, 2, G.ACT_RELU, False),     # stride 2, smallest channels
    (1, 16, 24, 96, 96, 1, 1, G.ACT_NONE, False),     # 1x1, no activation
    (2, 12, 20, 64, 144, 3, 1, G.ACT_SIGMOID, False),
    (1, 36, 28, 288, 48, 1, 1, G.ACT_SILU, True),
    (1, 8, 12, 576, 192, 3, 1, G.ACT_SILU, False),    # long K: exercises several accumulation blocks
    (2, 32, 48, 64, 96, 3, 2, G.ACT_SILU, False),     # stride 2 with cin % 32 == 0 (tap-unrolled kernel's stride path)
    (3, 17, 23, 96, 96, 3, 1, G.ACT_SILU, True),      # odd spatial size: M tail + borders in every tile, residual
]


def _run(eng, case, x, w, b, res):
    B, H, W, cin, cout, k, s, act, use_res = case
    g = G.Graph(task=G.TASK_TRACKNET)
    b0 = g.buf(0, cin)
    b1 = g.buf(1 if s == 2 else 0, G.pad16(cout))
    if use_res:
        # residual lives in the output buffer's sibling: emulate with a second input buffer written by a 1x1 identity?
        # simpler: residual slice = the input slice itself (cin == cout not required: use first `cout` channels)
        pass
    g.conv((b0, 0, cin), (b1, 0), w, b, k, s, act, res=(b0, 0) if use_res else None)
    g.head_buf = (b1, -1, -1)
    m = E.Model(eng, g)
    m.set_max_batch(B)
    y = m.tracknet_infer(x)[..., :cout]
    m.close()
    return y


def _setenv(**kw):
    for k in ("PADEL_CONV_IMPL", "PADEL_CONV_MF", "PADEL_CONV_NF", "PADEL_CONV_LDS_VARIANT", "PADEL_CONV_KB", "PADEL_CONV_PIPE", "PADEL_CONV_RING", "PADEL_CONV_TAP"):
        os.environ.pop(k, None)
    os.environ.update({k: str(v) for k, v in kw.items()})


@pytest.mark.parametrize("case", CASES, ids=[f"c{i}" for i in range(len(CASES))])
def test_conv_variants(gpu_engine, case):
    B, H, W, cin, cout, k, s, act, use_res = case
    if use_res and (s != 1 or cout > cin):
        pytest.skip("residual case needs same spatial size and cout <= cin")
    rng = np.random.default_rng(cin * 131 + cout)
    x = rng.normal(0, 1, (B, H, W, cin)).astype(np.float32)
    w = rng.normal(0, (2.0 / (cin * k * k)) ** 0.5, (cout, cin, k, k)).astype(np.float32)
    b = rng.normal(0, 0.5, cout).astype(np.float32)
    xt = torch.from_numpy(x).permute(0, 3, 1, 2)
    want = F.conv2d(xt.double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), stride=s, padding=k // 2)
    want = {G.ACT_SILU: F.silu, G.ACT_RELU: F.relu, G.ACT_SIGMOID: torch.sigmoid, G.ACT_NONE: lambda t: t}[act](want)
    if use_res:
        want = want + xt[:, :cout].double()
    want = want.permute(0, 2, 3, 1).numpy()
    scale = max(1.0, float(np.abs(want).max()))
    outs = {}
    try:
        for mf, nf in ((1, 1), (2, 3), (4, 2), (4, 3), (4, 4), (2, 5), (1, 6)):
            _setenv(PADEL_CONV_IMPL="direct", PADEL_CONV_MF=mf, PADEL_CONV_NF=nf)
            outs[f"d{mf}x{nf}"] = _run(gpu_engine, case, x, w, b, None)
        for v in range(13):
            _setenv(PADEL_CONV_LDS_VARIANT=v)
            outs[f"L{v}"] = _run(gpu_engine, case, x, w, b, None)
        for v in (0, 1, 6, 7, 9, 10, 11):             # v3: 3-stage LDS ring, double-buffered fragments
            _setenv(PADEL_CONV_LDS_VARIANT=v, PADEL_CONV_PIPE=1)
            outs[f"P{v}"] = _run(gpu_engine, case, x, w, b, None)
        for v in (1, 4, 6, 7, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19):   # v4 (13..19: 8/16-wave workgroups): LDS-DMA ring (global_load_lds, counted vmcnt, raw barrier)
            _setenv(PADEL_CONV_LDS_VARIANT=v, PADEL_CONV_RING=1)
            for rep in range(2):                       # twice: a DMA/barrier race would not be deterministic
                outs[f"R{v}.{rep}"] = _run(gpu_engine, case, x, w, b, None)
        if True:                                       # v5: tap-unrolled DMA ring (buffer addressing, zeros by range check)
            for v in (6, 7, 9, 10, 11, 12, 13, 14, 15, 20):
                _setenv(PADEL_CONV_LDS_VARIANT=v, PADEL_CONV_TAP=1)
                for rep in range(2):
                    outs[f"T{v}.{rep}"] = _run(gpu_engine, case, x, w, b, None)
        for v in (1, 7, 9, 11):                        # two k-steps per barrier (32-wide LDS stages)
            _setenv(PADEL_CONV_LDS_VARIANT=v, PADEL_CONV_KB=2)
            outs[f"L{v}k2"] = _run(gpu_engine, case, x, w, b, None)
        _setenv()
        outs["auto"] = _run(gpu_engine, case, x, w, b, None)
    finally:
        _setenv()
    ref_name, ref = next(iter(outs.items()))
    for name, y in outs.items():
        assert y.shape == want.shape
        err = float(np.abs(y - want).max()) / scale
        assert err < 3e-6, f"{name}: rel err {err:.2e} vs fp64 conv2d"
        assert np.array_equal(y, ref), f"{name} differs bitwise from {ref_name} (max {np.abs(y - ref).max():.3e})"
