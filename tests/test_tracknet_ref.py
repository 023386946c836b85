"""TrackNet / InpaintNet: the oracle restatement is pinned to golden outputs of the reference's own
models.py (tests/golden/make_tracknet_golden.py), and the engine's TrackNet graph (BN fold, 27->32 channel
padding, concat-by-slice, pools, upsamples) is checked against the oracle on CPU."""
from pathlib import Path

import numpy as np
import torch

from oracle import tracknet_ref as tr
from padel_analytics_amd import graph as G
from tests import graph_interp

GOLD = np.load(Path(__file__).parent / "golden" / "tracknet_golden.npz")


def test_known_answers():
    spec = tr.tracknet_spec()
    n = sum(int(np.prod(s)) for k, s in spec.items() if not k.endswith(("running_mean", "running_var", "num_batches_tracked")))
    assert n == 11341000 == int(GOLD["n_params"])
    assert len(spec) == 104 == int(GOLD["n_tensors"])
    sdi = tr.synth_inpaintnet_state_dict(0)
    assert sum(v.size for v in sdi.values()) == 520610 == int(GOLD["n_params_inpaint"])


def test_tracknet_oracle_matches_reference_golden():
    sd = tr.synth_tracknet_state_dict(int(GOLD["seed"]))
    y = tr.TrackNetRef(sd).forward(torch.from_numpy(GOLD["x"])).numpy()
    assert y.shape == GOLD["y"].shape
    assert np.abs(y - GOLD["y"]).max() < 2e-6


def test_inpaintnet_oracle_matches_reference_golden():
    sd = tr.synth_inpaintnet_state_dict(int(GOLD["seed"]) + 2)
    y = tr.InpaintNetRef(sd).forward(torch.from_numpy(GOLD["coor"]), torch.from_numpy(GOLD["mask"])).numpy()
    assert np.abs(y - GOLD["yi"]).max() < 2e-6


def test_tracknet_graph_matches_oracle():
    sd = tr.synth_tracknet_state_dict(7)
    g = G.build_tracknet(sd)
    assert g.bufs[0] == (0, 32) and g.bufs[g.head_buf[0]][1] == 8
    x = torch.rand(1, 27, 16, 32)
    xin = torch.zeros(1, 32, 16, 32)
    xin[:, :27] = x
    bufs = graph_interp.run(g, buf0=xin)
    want = tr.TrackNetRef(sd).forward(x)
    got = bufs[g.head_buf[0]]
    assert float((got - want).abs().max()) < 5e-6
    algo = sum(2.0 * np.prod(s) for k, s in tr.tracknet_spec().items() if k.endswith(("conv.weight", "predictor.weight"))) * 288 * 512
    # 227.6 GFLOP per 288x512 window — levels shrink the later layers, so recount per op
    fl = g.conv_flops(288, 512)
    assert abs(fl / 1e9 - 227.61) / 227.61 < 0.03      # + the 27->32 input padding of the first layer
