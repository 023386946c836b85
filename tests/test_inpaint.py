"""InpaintNet stage on the host: the numpy network against the reference's own models.py golden, the mask
generator against the transcription of the reference loop (including its edge quirks), and the whole
trajectory repair against the streaming transcription for several batch sizes."""
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


def test_inpaint_mask_matches_reference_loop():
    rng = np.random.default_rng(0)
    cases = [([1, 1, 0, 0, 1, 1], [100, 100, 0, 0, 100, 100]), ([0, 0, 1, 1], [0, 0, 90, 90]), ([1, 0, 0, 1], [80, 0, 0, 80]),
             ([1, 1, 1, 0, 0], [70, 70, 70, 0, 0]), ([1, 1, 0, 1, 0, 0, 1], [5, 5, 0, 5, 0, 0, 99]), ([1] * 5, [50] * 5), ([0] * 5, [0] * 5)]
    for _ in range(200):
        n = int(rng.integers(1, 40))
        v = (rng.uniform(size=n) > 0.4).astype(int)
        y = np.where(v == 1, rng.integers(0, 200, n), 0)
        cases.append((v.tolist(), y.tolist()))
    for v, y in cases:
        want = br.generate_inpaint_mask_ref(y, v, th_h=36.0)
        got = ip.generate_inpaint_mask(np.array(y), np.array(v), th_h=36.0).tolist()
        assert got == want, (v, y)


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
