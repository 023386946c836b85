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
