"""Host logic without a GPU: BN fold + K-order weight packing + concat-by-slice wiring of the engine's op
list, checked by interpreting the op list on CPU (tests/graph_interp.py) against the oracle."""
import numpy as np
import pytest
import torch

from oracle import yolov8_ref as ref
from padel_analytics_amd import graph as G, yolo_arch
from tests import graph_interp


def test_kstep_order_and_pack_roundtrip():
    for cin, k in ((16, 3), (32, 3), (48, 3), (96, 1), (80, 3)):
        steps = G.kstep_order(cin, k)
        assert len(steps) == cin // 16 * k * k
        assert sorted(steps) == sorted((t, c) for t in range(k * k) for c in range(0, cin, 16))
        w = np.random.default_rng(0).normal(size=(32, cin, k, k)).astype(np.float32)
        p = G.pack_conv_weight(w)
        o = dict(ksize=k, cin=cin, npad=32, w_off=0, b_off=0)
        wu, _ = graph_interp.unpack_conv(np.concatenate([p.reshape(-1), np.zeros(32, np.float32)]), o)
        assert np.array_equal(wu, w)


@pytest.mark.parametrize("scale,nc,kpt", [("n", 80, None), ("n", 1, (13, 3)), ("s", 2, (13, 2))])
def test_graph_matches_oracle(scale, nc, kpt):
    sd = yolo_arch.synth_state_dict(scale, nc, kpt, seed=1, gain=1.0)
    g = G.build_yolov8(sd, nc, kpt)
    x = torch.rand(1, 3, 64, 96)
    bufs = graph_interp.run(g, net_in=x)
    m = ref.YoloV8Ref(sd, nc, kpt)
    with torch.no_grad():
        det, kp = m.head_raw(m.features(x))
    for l in range(3):
        want = det[l] if not kpt else torch.cat([det[l], kp[l]], 1)
        got = bufs[g.head_buf[l]][:, :want.shape[1]]
        assert float((got - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))
    # every conv slice obeys the kernel's alignment contract
    for o in g.ops:
        if o["kind"] == G.OP_CONV:
            assert o["cin"] % 16 == 0 and o["in_choff"] % 4 == 0 and o["npad"] % 16 == 0 and o["w_off"] % 4 == 0


def test_conv_flops_accounting():
    sd = yolo_arch.synth_state_dict("n", 80, None, seed=0)
    g = G.build_yolov8(sd, 80, None)
    # executed FLOPs (with the 16-padding of head widths) are within 1% of the algorithmic count
    algo = yolo_arch.conv_flops(yolo_arch.conv_inventory("n", 80, None, 384, 640))
    assert abs(g.conv_flops(384, 640) / algo - 1) < 0.01


def test_graph_layout_is_weight_independent():
    """bench.py / dist: ranks != 0 build the op list from uncalibrated weights and receive rank 0's blob over
    RCCL — the layout (ops, buffers, offsets, blob length) must depend on shapes only."""
    for scale, nc, kpt in (("n", 1, None), ("n", 1, (13, 3))):
        a = G.build_yolov8(yolo_arch.synth_state_dict(scale, nc, kpt, seed=0), nc, kpt)
        b = G.build_yolov8(yolo_arch.synth_state_dict(scale, nc, kpt, seed=9, cls_bias=1.5, gain=0.7), nc, kpt)
        assert a.n_floats == b.n_floats and a.bufs == b.bufs and a.ops == b.ops and a.head_buf == b.head_buf
        assert not np.array_equal(a.blob(), b.blob())


# ---- fp16 graphs (BASELINE configs[4]): 32-channel k-steps, padded slices, fp16 weight words inside the fp32 blob
def test_fp16_pack_roundtrip():
    for cin, k in ((32, 3), (64, 3), (96, 3), (160, 1)):
        steps = G.kstep_order(cin, k, 32)
        assert sorted(steps) == sorted((t, c) for t in range(k * k) for c in range(0, cin, 32))
        w = np.random.default_rng(0).normal(size=(16, cin, k, k)).astype(np.float16).astype(np.float32)
        g = G.Graph(task=G.TASK_DETECT, dtype=G.DTYPE_F16)
        b0, b1 = g.buf(0, cin), g.buf(0, 16)
        g.conv((b0, 0, cin), (b1, 0), w, np.zeros(16, np.float32), k, 1, G.ACT_NONE)
        wu, _ = graph_interp.unpack_conv(g.blob(), g.ops[0], f16=True)
        assert np.array_equal(wu, w)


@pytest.mark.parametrize("scale,nc,kpt", [("n", 80, None), ("m", 1, (13, 3)), ("s", 1, None)])
def test_fp16_graph_wiring(scale, nc, kpt):
    """The fp16 op list computes the same network: interpreted on CPU with fp16 storage rounding it agrees with the
    fp32 oracle to fp16 accuracy, every conv slice obeys the fp16 kernel's contract, and garbage in pad / not yet
    written channels (buffers pre-filled with a large finite value) does not reach any result."""
    sd = yolo_arch.synth_state_dict(scale, nc, kpt, seed=1, gain=1.0)
    g = G.build_yolov8(sd, nc, kpt, dtype="f16")
    x = torch.rand(1, 3, 64, 96)
    bufs = graph_interp.run(g, net_in=x)
    dirty = graph_interp.run(g, net_in=x, stale=1000.0)
    m = ref.YoloV8Ref(sd, nc, kpt)
    with torch.no_grad():
        det, kp = m.head_raw(m.features(x))
    for l in range(3):
        want = det[l] if not kpt else torch.cat([det[l], kp[l]], 1)
        got = bufs[g.head_buf[l]][:, :want.shape[1]]
        assert float((got - want).abs().max()) <= 3e-2 * max(1.0, float(want.abs().max())), l
        assert torch.equal(dirty[g.head_buf[l]][:, :want.shape[1]], got), "stale pad channels leaked into the head"
    for o in g.ops:
        if o["kind"] == G.OP_CONV:
            lvl, ch = g.bufs[o["in_buf"]]
            assert o["cin"] % 32 == 0 and o["in_choff"] % 8 == 0 and o["in_choff"] + o["cin"] <= ch and o["npad"] % 16 == 0
        if o["kind"] in (G.OP_SPPF_POOL, G.OP_UPSAMPLE2X):
            assert o["cin"] % 8 == 0 and o["in_choff"] % 8 == 0 and o["out_choff"] % 8 == 0


def test_bf16x3_split_is_exact_and_packed_in_lane_order():
    rng = np.random.default_rng(3)
    w = (rng.normal(size=(16, 48, 3, 3)) * 10.0 ** rng.integers(-6, 3, (16, 48, 3, 3))).astype(np.float32)
    hi, mid, lo = G.split_bf16x3(w)
    back = sum((p.astype(np.uint32) << 16).view(np.float32).astype(np.float64) for p in (hi, mid, lo))
    assert np.array_equal(back.astype(np.float32), w) and np.array_equal(back, w.astype(np.float64))
    packed = G.pack_conv_weight_bx3(w)          # cin 48 -> one full chunk x 9 taps + 5 tap-paired tail steps
    steps = G.bx3_ksteps(48, 3)
    assert packed.shape == (16, 14, 3, 32) and len(steps) == 14
    val = sum((packed[:, :, p].astype(np.uint32) << 16).view(np.float32).astype(np.float64) for p in range(3))   # (16, 14, 32)
    seen = set()
    for si, slots in enumerate(steps):
        for pos in range(32):
            sl = slots[G.BX3_PERM[pos]]
            want = np.zeros(16, np.float32) if sl is None else w[:, sl[0], sl[1] // 3, sl[1] % 3]
            assert np.array_equal(val[:, si, pos], want.astype(np.float64)), (si, pos)
            if sl is not None:
                seen.add(sl)
    assert seen == {(c, t) for c in range(48) for t in range(9)}          # every (channel, tap) exactly once
    assert steps[9][:16] == [(32 + i, 0) for i in range(16)] and steps[9][16:] == [(32 + i, 1) for i in range(16)]
    assert steps[13][16:] == [None] * 16
    assert len(G.bx3_ksteps(48, 1)) == 2 and len(G.bx3_ksteps(16, 3)) == 5 and len(G.bx3_ksteps(96, 3)) == 27
    assert sorted(G.BX3_PERM.tolist()) == list(range(32))


def test_h2_pairs_and_column_major_weight_pack():
    """h2 (csrc/h2_common.h): a value as an fp16 pair h + m / 2048 keeps 22-23 bits; weight rows are scaled by a power of
    two so that both planes sit in the normal fp16 range; the k-steps of a 3x3 walk the taps COLUMN-major (k-step t of a
    chunk = tap (ky, kx) = (t % 3, t // 3)) and the 16-channel tail pairs consecutive taps of that order."""
    rng = np.random.default_rng(5)
    x = (rng.normal(size=4096) * 10.0 ** rng.integers(-3, 4, 4096)).astype(np.float32)
    x = x[np.abs(x) < 6.0e4]
    h, m = G.h2_split(x)
    back = G.h2_value(h, m)
    big = np.abs(x) > 1.2e-4
    assert np.all(np.abs(back[big] - x[big]) <= np.abs(x[big]) * 2.0 ** -22)
    assert np.all(np.abs(back[~big] - x[~big]) <= 3e-11)
    w = (rng.normal(size=(32, 48, 3, 3)) * 10.0 ** rng.integers(-4, 2, (32, 1, 1, 1))).astype(np.float32)
    planes, inv = G.pack_conv_weight_h2(w)                    # cin 48: one full chunk x 9 taps + 5 tap-paired tail steps
    assert planes.shape == (32, 14, 2, 32) and inv.shape == (32,)
    sc = 1.0 / inv
    assert np.all(np.log2(sc) == np.round(np.log2(sc)))       # powers of two: the scaling itself is exact
    top = np.abs(w.reshape(32, -1)).max(1) * sc
    assert np.all((top >= 2.0 ** 12) & (top < 2.0 ** 13))
    val = G.h2_value(planes[:, :, 0].view(np.float16), planes[:, :, 1].view(np.float16)) * inv[:, None, None]      # (32, 14, 32)
    for t in range(9):                                        # full chunk, k-step t: channels 0..31 at tap (t % 3, t // 3)
        want = w[:, :32, t % 3, t // 3]
        assert np.all(np.abs(val[:, t] - want) <= np.abs(want) * 2.0 ** -21 + 1e-12), t
    for jt in range(5):                                       # tail: slots 0..15 = channels 32..47 at tap 2 jt, 16..31 = at tap 2 jt + 1
        ta, tb = 2 * jt, 2 * jt + 1
        assert np.all(np.abs(val[:, 9 + jt, :16] - w[:, 32:, ta % 3, ta // 3]) <= np.abs(w[:, 32:, ta % 3, ta // 3]) * 2.0 ** -21 + 1e-12)
        if tb < 9:
            assert np.all(np.abs(val[:, 9 + jt, 16:] - w[:, 32:, tb % 3, tb // 3]) <= np.abs(w[:, 32:, tb % 3, tb // 3]) * 2.0 ** -21 + 1e-12)
        else:
            assert np.all(val[:, 9 + jt, 16:] == 0.0)
    o = dict(ksize=3, cin=48, npad=32, w_off=0, reserved=planes.size // 2, b_off=planes.size // 2 + 32)
    blob = np.concatenate([planes.reshape(-1).view(np.float32), inv, np.zeros(32, np.float32)])
    wu, _ = graph_interp.unpack_conv_h2(blob, o)
    assert np.all(np.abs(wu - w) <= np.abs(w) * 2.0 ** -21 + 1e-12)


def test_h2_row_scale_survives_denormal_and_huge_rows():
    """A dead BN-folded channel (weights ~ 1e-40) used to get the scale 2^12 / 2^-133 = inf: NaN planes, 1 / s = 0
    (ADVICE r3).  Denormal rows count as all-zero; every scale and its inverse are normal fp32 numbers."""
    w = np.zeros((16, 16, 1, 1), np.float32)
    w[0] = 1e-40
    w[1] = 3.0e38
    w[2] = 1e-30
    w[3, 0] = 1.0
    sc = G.h2_row_scale(w.reshape(16, -1))
    assert np.all(np.isfinite(sc)) and np.all(sc > 0) and np.all(np.isfinite(1.0 / sc)) and np.all(1.0 / sc > 0)
    assert sc[0] == 1.0 and sc[4] == 1.0 and sc[3] == 2.0 ** 12
    planes, inv = G.pack_conv_weight_h2(w)
    assert np.all(np.isfinite(inv)) and np.all(inv > 0)
    val = G.h2_value(planes[:, :, 0].view(np.float16), planes[:, :, 1].view(np.float16))
    assert np.all(np.isfinite(val))
    back = val[:, 0, :16] * inv[:, None]
    assert np.all(back[0] == 0.0)                                   # the denormal row contributes nothing (as in fp16 it cannot)
    assert np.all(np.abs(back[2] - 1e-30) <= 1e-30 * 2.0 ** -21)     # 2^100 keeps a 1e-30 row in the normal fp16 range
    assert np.all(np.abs(back[3, 0] - 1.0) <= 2.0 ** -21)


@pytest.mark.parametrize("nc,kpt", [(80, None), (1, (13, 3))])
def test_head_first_convs_per_branch_equal_the_merged_conv(nc, kpt, monkeypatch):
    """Round 4: h2 graphs of the m / l / x scales run the first 3x3 conv of each Detect / Pose branch as its own op into its
    slice of the shared buffer (tile fit, graph.py).  Same head maps as the merged conv, and the automatic choice is what the
    comment in graph.py says."""
    sd = yolo_arch.synth_state_dict("s", nc, kpt, seed=2, gain=1.0)
    x = torch.rand(1, 3, 64, 96)
    outs, n_convs = [], []
    for split in (False, True):
        monkeypatch.setattr(G, "HEAD_SPLIT", split)
        g = G.build_yolov8(sd, nc, kpt)
        bufs = graph_interp.run(g, net_in=x)
        outs.append([bufs[g.head_buf[l]].clone() for l in range(3)])
        n_convs.append(sum(1 for o in g.ops if o["kind"] == G.OP_CONV))
    nbr = 3 if kpt else 2
    assert n_convs[1] == n_convs[0] + 3 * (nbr - 1)
    for a, b in zip(*outs):          # (the CPU interpreter's conv2d blocks differently for different widths: last-bit
        assert float((a - b).abs().max()) <= 4e-6 * max(1.0, float(a.abs().max()))       # differences; bitwise on the GPU: tests/test_gpu_h2.py)
    monkeypatch.setattr(G, "HEAD_SPLIT", None)
    count = lambda g: sum(1 for o in g.ops if o["kind"] == G.OP_CONV)
    # automatic: only h2, only P3 / P4, only a merged width that is not a whole number of 48- or 64-channel tiles (m pose: 304)
    sd_m = yolo_arch.synth_state_dict("m", nc, kpt, seed=2)
    extra = 2 * (nbr - 1) if kpt else 0
    assert count(G.build_yolov8(sd_m, nc, kpt, dtype="h2")) == count(G.build_yolov8(sd_m, nc, kpt, dtype="f32")) + extra
    sd_s = yolo_arch.synth_state_dict("s", nc, kpt, seed=2)           # 64 + 128 (+ 48) channels: whole 64- / 48-channel tiles
    assert count(G.build_yolov8(sd_s, nc, kpt, dtype="h2")) == count(G.build_yolov8(sd_s, nc, kpt, dtype="f32"))


def test_unfolded_batchnorm_keeps_fp16_checkpoint_weights_single_plane():
    """Round 5 (PA_CONV_W_SINGLE): an h2 YOLOv8 graph built from a checkpoint whose conv weights are fp16 numbers keeps them
    unfolded — raw weights x row power of two in the h plane, an all-zero m plane, BatchNorm's scale in the per-channel output
    scale — and flags every such conv; one weight that is not an fp16 number sends THAT conv back to the folded packing; other
    dtypes never flag; PADEL_UNFOLDED_BN=0 folds everything.  The packed pieces reproduce fold_bn's numbers: h / rowscale x
    output scale == w x BN scale exactly (products of a 11-bit and a 24-bit number fit a double)."""
    from padel_analytics_amd import yolo_arch
    sd = yolo_arch.synth_state_dict("n", 80, None, seed=3, cls_bias=-1.0)
    assert all(G.fp16_exact(np.asarray(v)) for k, v in sd.items() if k.endswith("conv.weight"))
    g = G.build_yolov8(sd, 80, None, dtype="h2")
    convs = [o for o in g.ops if o["kind"] == G.OP_CONV]
    assert convs and all(o["flags"] & G.FLAG_W_SINGLE for o in convs)
    assert not any(o.get("flags", 0) for o in G.build_yolov8(sd, 80, None, dtype="f32").ops)
    assert not any(o.get("flags", 0) for o in G.build_yolov8(sd, 80, None, dtype="f16").ops)
    # the first conv after the stem (model.1): its packed planes and output scale against fold_bn
    blob = g.blob()
    o = convs[0]
    w_raw, bn_scale, bf = G.fold_bn_split(sd, "model.1", yolo_arch.BN_EPS)
    wf, bf2 = G.fold_bn(sd, "model.1", yolo_arch.BN_EPS)
    assert np.array_equal(bf, bf2)
    cout, cin = w_raw.shape[:2]
    nsteps = len(G.bx3_ksteps(o["cin"], 3))
    planes = blob[o["w_off"]:o["w_off"] + o["npad"] * nsteps * 2 * 32 // 2].view(np.uint16).reshape(o["npad"], nsteps, 2, 32)
    assert not (planes[:, :, 1] & 0x7FFF).any()
    osc = blob[o["reserved"]:o["reserved"] + o["npad"]]
    h = planes[:, :, 0].view(np.float16).astype(np.float64)
    # k-step t of the full chunk = tap (ky, kx) = (t % 3, t // 3), slots = channels (cin = 16 here: the tail pairs taps; use the sum)
    got = (h * osc[:, None, None].astype(np.float64)).reshape(o["npad"], -1).sum(1)[:cout]
    want = (w_raw.astype(np.float64) * bn_scale.astype(np.float64)[:, None, None, None]).reshape(cout, -1).sum(1)
    assert np.allclose(got, want, rtol=1e-12, atol=1e-12)
    assert np.allclose(got, wf.astype(np.float64).reshape(cout, -1).sum(1), rtol=2e-6)          # fold_bn's fp32 products: same to fp32 rounding
    # one weight off the fp16 grid: that conv folds, the others stay single
    sd2 = dict(sd)
    w2 = np.asarray(sd["model.3.conv.weight"], np.float32).copy()
    w2.flat[0] = np.float32(w2.flat[0]) * np.float32(1.0 + 2.0 ** -20)
    sd2["model.3.conv.weight"] = w2
    ops2 = [o2 for o2 in G.build_yolov8(sd2, 80, None, dtype="h2").ops if o2["kind"] == G.OP_CONV]
    assert sum(1 for o2 in ops2 if not (o2["flags"] & G.FLAG_W_SINGLE)) == 1
    # a head branch off the fp16 grid: the first head convs of that level share one rule — merged (one conv) or split per branch
    # (two convs on a detect head), they are all folded
    sd3 = dict(sd)
    w3 = np.asarray(sd["model.22.cv3.0.0.conv.weight"], np.float32).copy()
    w3.flat[5] = np.float32(w3.flat[5]) * np.float32(1.0 + 2.0 ** -20)
    sd3["model.22.cv3.0.0.conv.weight"] = w3
    hs = G.HEAD_SPLIT
    try:
        for split, n_folded in ((False, 1), (True, 2)):
            G.HEAD_SPLIT = split
            ops3 = [o3 for o3 in G.build_yolov8(sd3, 80, None, dtype="h2").ops if o3["kind"] == G.OP_CONV]
            assert sum(1 for o3 in ops3 if not (o3["flags"] & G.FLAG_W_SINGLE)) == n_folded, (split, [o3["flags"] for o3 in ops3])
    finally:
        G.HEAD_SPLIT = hs
    old = G.UNFOLDED_BN
    try:
        G.UNFOLDED_BN = False
        # (the head's last 1x1 convs have no BatchNorm: their weights are the checkpoint's fp16 numbers either way)
        assert all(o2["act"] == G.ACT_NONE and o2["ksize"] == 1 for o2 in G.build_yolov8(sd, 80, None, dtype="h2").ops
                   if o2["kind"] == G.OP_CONV and o2["flags"])
    finally:
        G.UNFOLDED_BN = old
