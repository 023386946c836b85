"""The loader a user with a real ``yolov8m.pt`` hits first (reference ``players_tracker.py:303``: ``YOLO(model_path)``;
``ball_tracker.py:253-274``: the TrackNetV3 dict format).  ultralytics is not installable here, so the test BUILDS an
Ultralytics-shaped pickle: a fake ``ultralytics.nn.tasks`` / ``ultralytics.nn.modules`` package with nn.Module classes
of the right nesting (``model.{i}.conv / .bn / .cv1 / .m.{j} ...``), fp16 tensors, ``yaml``, ``names``, ``ema`` —
pickles it, REMOVES the fake package from ``sys.modules`` and loads the file the way the product does.  Also: the stub
unpickler resolves an exact allowlist only (a reduce-to-torch-function pickle must come back inert)."""
import io
import os
import pickle
import sys
import types

import numpy as np
import pytest
import torch
import torch.nn as nn

from padel_analytics_amd import checkpoint, graph as G, yolo_arch


def _install_fake_ultralytics():
    """Module classes named and nested like upstream's; state_dict keys come out as ``model.{i}...``."""
    pkg = types.ModuleType("ultralytics")
    nnm = types.ModuleType("ultralytics.nn")
    tasks = types.ModuleType("ultralytics.nn.tasks")
    mods = types.ModuleType("ultralytics.nn.modules")
    pkg.nn, nnm.tasks, nnm.modules = nnm, tasks, mods

    def cls(name, module):
        c = type(name, (nn.Module,), {"__module__": module.__name__})
        setattr(module, name, c)
        return c

    for n in ("Conv", "C2f", "Bottleneck", "SPPF", "Concat", "Detect", "Pose", "DFL"):
        cls(n, mods)
    for n in ("DetectionModel", "PoseModel"):
        cls(n, tasks)
    sys.modules.update({"ultralytics": pkg, "ultralytics.nn": nnm, "ultralytics.nn.tasks": tasks, "ultralytics.nn.modules": mods})
    return tasks, mods


def _remove_fake_ultralytics():
    for k in [k for k in sys.modules if k == "ultralytics" or k.startswith("ultralytics.")]:
        del sys.modules[k]


def _module_tree(sd, tasks, mods, pose):
    """nn.Module tree whose parameters / buffers reproduce ``sd`` (keys ``model.{i}.a.b.weight``) in fp16."""
    kinds = {"conv": nn.Conv2d, "bn": nn.BatchNorm2d}
    root = (tasks.PoseModel if pose else tasks.DetectionModel)()
    nn.Module.__init__(root)

    def child(parent, name):
        if name in parent._modules:
            return parent._modules[name]
        m = nn.Module.__new__(mods.Conv if not name.isdigit() else mods.C2f)
        nn.Module.__init__(m)
        parent.add_module(name, m)
        return m

    for key, val in sd.items():
        parts = key.split(".")
        node = root
        for p in parts[:-1]:
            node = child(node, p)
        t = torch.from_numpy(np.asarray(val))
        if t.is_floating_point():
            t = t.half()
        leaf = parts[-1]
        if leaf in ("weight", "bias") and parts[-2] != "bn":
            node.register_parameter(leaf, nn.Parameter(t, requires_grad=False))
        elif leaf in ("weight", "bias"):
            node.register_parameter(leaf, nn.Parameter(t, requires_grad=False))
        else:
            node.register_buffer(leaf, t)
    return root


@pytest.mark.parametrize("scale,nc,kpt,use_ema", [("n", 80, None, True), ("n", 1, (13, 3), False), ("s", 2, (12, 2), True)])
def test_ultralytics_style_pickle_roundtrip(tmp_path, scale, nc, kpt, use_ema):
    sd = yolo_arch.synth_state_dict(scale, nc, kpt, seed=3)
    sd["model.0.bn.num_batches_tracked"] = np.asarray(7, np.int64)          # upstream state_dicts carry these
    tasks, mods = _install_fake_ultralytics()
    try:
        model = _module_tree(sd, tasks, mods, pose=kpt is not None)
        model.yaml = {"nc": nc, "scale": scale, **({"kpt_shape": list(kpt)} if kpt else {})}
        model.names = {i: f"c{i}" for i in range(nc)}
        model.stride = torch.tensor([8.0, 16.0, 32.0])
        ck = {"epoch": -1, "best_fitness": None, "model": None if use_ema else model, "ema": model if use_ema else None,
              "updates": 0, "optimizer": None, "train_args": {"imgsz": 640}, "date": "2025-01-14", "version": "8.3.0"}
        if use_ema:
            ck["model"] = model
        path = tmp_path / "yolov8_fake.pt"
        torch.save(ck, str(path))
    finally:
        _remove_fake_ultralytics()
    assert "ultralytics" not in sys.modules
    with pytest.raises(Exception):
        torch.load(str(path), map_location="cpu", weights_only=True)        # the safe loader alone cannot read it
    got = checkpoint.load_checkpoint(path)
    assert got.task == ("pose" if kpt else "detect") and got.nc == nc and got.scale == scale
    assert got.kpt_shape == (tuple(kpt) if kpt else None)
    assert got.names == {i: f"c{i}" for i in range(nc)}
    want_keys = {k for k in sd}
    assert set(got.state_dict) == want_keys, (sorted(want_keys - set(got.state_dict))[:5], sorted(set(got.state_dict) - want_keys)[:5])
    for k, v in sd.items():
        w = np.asarray(v)
        g = got.state_dict[k]
        if np.issubdtype(w.dtype, np.floating):
            assert g.dtype == np.float32                                     # .float() like upstream
            assert np.array_equal(g, w.astype(np.float16).astype(np.float32)), k
        else:
            assert np.array_equal(g, w), k
    # and the graph builder accepts what came back (strict key / shape match, SURVEY §8(c) known answer 6)
    g = G.build_yolov8(got.state_dict, got.nc, got.kpt_shape)
    assert g.nc == nc and len(g.ops) > 50


def test_ambiguous_pose_head_without_yaml_is_rejected(tmp_path):
    sd = yolo_arch.synth_state_dict("n", 1, (12, 2), seed=1)                # nk = 24: 12 x 2 or 8 x 3
    tasks, mods = _install_fake_ultralytics()
    try:
        model = _module_tree(sd, tasks, mods, pose=True)
        torch.save({"model": model, "ema": None}, str(tmp_path / "court.pt"))
    finally:
        _remove_fake_ultralytics()
    with pytest.raises(ValueError, match="ambiguous"):
        checkpoint.load_checkpoint(tmp_path / "court.pt")


def test_tracknetv3_dict_format(tmp_path):
    """ball_tracker.py:253-274: ``torch.load(path)`` -> {"param_dict": {...seq_len, bg_mode}, "model": state_dict}."""
    from oracle import tracknet_ref as tr
    sd = tr.synth_tracknet_state_dict(2)
    tsd = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    torch.save({"epoch": 29, "param_dict": {"seq_len": 8, "bg_mode": "concat", "model_name": "TrackNet"}, "model": tsd,
                "optimizer": {}, "scheduler": {}}, str(tmp_path / "TrackNet_best.pt"))
    ck = checkpoint.load_checkpoint(tmp_path / "TrackNet_best.pt")
    assert ck.task == "tracknet" and ck.param_dict["seq_len"] == 8 and ck.param_dict["bg_mode"] == "concat"
    assert set(ck.state_dict) == set(sd)
    for k in sd:
        assert np.array_equal(ck.state_dict[k], np.asarray(sd[k]))
    g = G.build_tracknet(ck.state_dict)
    assert g.in_channels == 32 and g.out_channels == 8
    isd = tr.synth_inpaintnet_state_dict(1) if hasattr(tr, "synth_inpaintnet_state_dict") else None
    if isd is not None:
        torch.save({"param_dict": {"seq_len": 16}, "model": {k: torch.from_numpy(np.asarray(v)) for k, v in isd.items()}},
                   str(tmp_path / "InpaintNet_best.pt"))
        ick = checkpoint.load_checkpoint(tmp_path / "InpaintNet_best.pt")
        assert ick.task == "inpaintnet" and ick.param_dict["seq_len"] == 16


class _Evil:
    def __init__(self, fn, args):
        self.fn, self.args = fn, args

    def __reduce__(self):
        return self.fn, self.args


@pytest.mark.parametrize("target", ["os.system", "torch.utils.collect_env.run", "torch.hub.load", "builtins.eval",
                                    "numpy.testing._private.utils.runstring", "torch.load", "subprocess.check_output"])
def test_stub_unpickler_never_calls_a_reduce_target(tmp_path, target):
    """A pickle whose __reduce__ names a callable under torch.* / numpy.* / os comes back as an inert stub: nothing runs."""
    import importlib
    modname, fn = target.rsplit(".", 1)
    f = getattr(importlib.import_module(modname), fn)
    marker = tmp_path / "pwned"
    payload = {"model": _Evil(f, (f"touch {marker}",)), "ema": None}
    buf = io.BytesIO()
    pickle.dump(payload, buf, protocol=2)
    buf.seek(0)
    obj = checkpoint._StubUnpickler(buf).load()
    assert not marker.exists()
    assert isinstance(obj["model"], checkpoint._Bag) and type(obj["model"]).__name__ == fn
    # the same through the public loader (legacy non-zip torch.save framing is not needed: load_checkpoint falls back to
    # the stub pickle module whenever weights_only=True refuses the file)
    path = tmp_path / "evil.pt"
    torch.save(payload, str(path))
    with pytest.raises(ValueError):
        checkpoint.load_checkpoint(path)                                     # no tensors recoverable -> rejected
    assert not marker.exists()
