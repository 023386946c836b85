"""InpaintNet stage on the host: the numpy network against the reference's own models.py golden, the mask
generator against masks produced by the reference's own generate_inpaint_mask (including its edge quirks), and the
whole trajectory repair against the oracle's streaming restatement for several batch sizes."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import ball_ref as br, tracknet_ref as tr
from padel_analytics_amd import inpaint as ip

GOLD = np.load(Path(__file__).parent / "golden" / "tracknet_golden.npz")


def test_inpaintnet_host_matches_reference_golden():
    net = ip.InpaintNetHost(tr.synth_inpaintnet_state_dict(int(GOLD["seed"]) + 2))
    y = net.forward(GOLD["coor"], GOLD["mask"])
    assert y.shape == GOLD["yi"].shape and np.abs(y - GOLD["yi"]).max() < 2e-6


def test_inpaint_mask_matches_reference_goldens():
    """generate_inpaint_mask (ball_tracker.py:100-136): oracle restatement and product vs the masks the reference's
    own function returned for 71 visibility patterns (tests/golden/make_ball_golden.py), edge quirks included."""
    import json
    cases = json.loads((Path(__file__).parent / "golden" / "objects_golden.json").read_text())["inpaint_masks"]
    assert len(cases) >= 60
    for c in cases:
        v, y, want = c["visibility"], c["y"], c["mask"]
        assert br.generate_inpaint_mask_ref(y, v, th_h=c["th_h"]) == want, (v, y)
        assert ip.generate_inpaint_mask(np.array(y), np.array(v), th_h=c["th_h"]).tolist() == want, (v, y)


def test_ensemble_weight_and_generic_ensemble():
    assert np.allclose(ip.ensemble_weight(16), br.inpaint_ensemble_weight(16))
    assert np.allclose(ip.ensemble_weight(8), br.ensemble_weight(8))
    y = np.random.default_rng(1).uniform(0, 1, (11, 8, 3, 5)).astype(np.float32)
    assert np.allclose(ip.temporal_ensemble(y, ip.ensemble_weight(8)), br.ensemble(y), atol=1e-7)


@pytest.mark.parametrize("T,batch", [(40, 4), (23, 8), (16, 3)])
def test_trajectory_repair_matches_streaming_transcription(T, batch):
    rng = np.random.default_rng(T)
    w, h = 1280, 720
    vis = (rng.uniform(size=T) > 0.3).astype(int)
    xs = np.where(vis == 1, rng.integers(50, w - 50, T), 0).tolist()
    ys = np.where(vis == 1, rng.integers(60, h - 50, T), 0).tolist()
    sd = tr.synth_inpaintnet_state_dict(4)
    ref_net = tr.InpaintNetRef(sd)
    want, raw = br.inpaint_stage_ref(xs, ys, vis.tolist(), w, h, ref_net.forward, 16, batch)
    got = ip.inpaint_trajectory(xs, ys, vis.tolist(), w, h, ip.InpaintNetHost(sd), 16)
    assert len(got) == T == len(want)
    n_masked = sum(ip.generate_inpaint_mask(np.array(ys), vis, th_h=36.0))
    for g in range(T):
        fx, fy = raw[g]
        near = min(abs(fx - round(fx)), abs(fy - round(fy))) < 1e-2          # int() truncation boundary
        if not near:
            assert got[g] == want[g], (g, got[g], want[g], raw[g])
    assert n_masked >= 0


def test_inpaintnet_device_graph_host_logic():
    """graph.build_inpaintnet (the device form of InpaintNet, round 4) interpreted on the CPU (tests/graph_interp.py unpacks
    the weight blob the way the kernels index it): Conv1d(k=3) as the middle row of a 3x3 over one-row images, concats as
    channel slices, LeakyReLU / sigmoid op codes — against the torch oracle, for fp32 and h2 storage; stale buffer contents
    (pad channels, never-written slices) must not reach the output."""
    import torch
    from oracle import tracknet_ref as tr
    from padel_analytics_amd import graph as G
    from tests import graph_interp
    sd = tr.synth_inpaintnet_state_dict(4)
    rng = np.random.default_rng(1)
    S, L = 9, 16
    coor = rng.uniform(0, 1, (S, L, 2)).astype(np.float32)
    mask = (rng.uniform(size=(S, L, 1)) > 0.6).astype(np.float32)
    want = tr.InpaintNetRef(sd).forward(torch.from_numpy(coor), torch.from_numpy(mask)).numpy()        # (S, L, 2)
    x = np.zeros((S, 16, 1, L), np.float32)                                                            # NCHW of (S, 1, L, 16)
    x[:, 0, 0], x[:, 1, 0], x[:, 2, 0] = coor[..., 0], coor[..., 1], mask[..., 0]
    for dtype, tol in (("f32", 2e-6), ("h2", 5e-6)):
        g = G.build_inpaintnet(sd, dtype=dtype)
        assert len(g.ops) == 9 and [o["act"] for o in g.ops] == [G.ACT_LEAKY] * 8 + [G.ACT_SIGMOID]
        for stale in (0.0, 1000.0):
            bufs = graph_interp.run(g, buf0=torch.from_numpy(x), stale=stale)
            got = bufs[g.head_buf[0]][:, :2, 0].numpy().transpose(0, 2, 1)
            assert np.abs(got - want).max() < tol, (dtype, stale, np.abs(got - want).max())
