"""Generates tests/golden/tracknet_golden.npz by importing the REFERENCE's own
/root/reference/trackers/ball_tracker/models.py (by file path; the package import needs ultralytics) in
this container.  Only numeric arrays are stored: input, outputs and the seed — the weights are re-created
from the seed by `oracle.tracknet_ref.synth_tracknet_state_dict` / `synth_inpaintnet_state_dict`, which
this script also uses to fill the reference modules, so the fixture stays a few hundred kB.

    python tests/golden/make_tracknet_golden.py
"""
import importlib.util
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import tracknet_ref as tr  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_models", "/root/reference/trackers/ball_tracker/models.py")
ref_models = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_models)

SEED = 1234
torch.manual_seed(0)
net = ref_models.TrackNet(in_dim=27, out_dim=8).eval()
sd = tr.synth_tracknet_state_dict(SEED)
missing = net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
x = torch.from_numpy(np.random.default_rng(SEED + 1).uniform(0, 1, (2, 27, 32, 64)).astype(np.float32))
with torch.no_grad():
    y = net(x)
n_params = sum(p.numel() for p in net.parameters())

inp = ref_models.InpaintNet().eval()
sdi = tr.synth_inpaintnet_state_dict(SEED + 2)
inp.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sdi.items()}, strict=True)
rng = np.random.default_rng(SEED + 3)
coor = torch.from_numpy(rng.uniform(0, 1, (3, 16, 2)).astype(np.float32))
mask = torch.from_numpy((rng.uniform(0, 1, (3, 16, 1)) > 0.7).astype(np.float32))
with torch.no_grad():
    yi = inp(coor, mask)
n_params_i = sum(p.numel() for p in inp.parameters())

np.savez_compressed(Path(__file__).with_name("tracknet_golden.npz"), seed=SEED, x=x.numpy(), y=y.numpy(),
                    n_params=n_params, n_tensors=len(net.state_dict()), coor=coor.numpy(), mask=mask.numpy(),
                    yi=yi.numpy(), n_params_inpaint=n_params_i)
print("TrackNet params", n_params, "state tensors", len(net.state_dict()), "InpaintNet params", n_params_i, "y", tuple(y.shape))
