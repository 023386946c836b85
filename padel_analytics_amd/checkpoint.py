"""Checkpoint I/O: turn a ``.pt`` file into a flat fp32 ``state_dict`` + model metadata.

Two on-disk formats are accepted:

* **Ultralytics pickles** (what ``YOLO(model_path)`` receives at
  ``players_tracker.py:303`` / ``players_keypoints_tracker.py:238``): a dict whose
  ``"ema"``/``"model"`` entry is a pickled ``ultralytics.nn.tasks.DetectionModel`` /
  ``PoseModel`` in fp16.  ``ultralytics`` is not importable here, so the unpickler maps
  every ``ultralytics.*`` class to a generic attribute bag and the tensors are recovered
  by walking ``_modules`` / ``_parameters`` / ``_buffers`` (SURVEY.md §7 "Hard parts").
* **plain dict checkpoints** written by :func:`save_checkpoint` (used for the seeded
  synthetic weights: there are no real weights offline) and TrackNet-style
  ``{"param_dict": ..., "model": state_dict}`` files (``ball_tracker.py:253-265``).

PyTorch is used only as the tensor (de)serialiser.
"""
from __future__ import annotations

import pickle
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Optional

import numpy as np
import torch

from . import yolo_arch

COCO_PERSON_NAMES = {0: "person"}


@dataclass
class Checkpoint:
    state_dict: "OrderedDict[str, np.ndarray]"
    task: str                               # "detect" | "pose" | "tracknet" | "inpaintnet"
    nc: int = 0
    kpt_shape: Optional[tuple] = None
    scale: Optional[str] = None
    names: dict = field(default_factory=dict)
    param_dict: dict = field(default_factory=dict)


# ----------------------------------------------------------------------------- stub unpickle

class _Bag:
    """Stand-in for any class the pickle references that is not importable."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        elif isinstance(state, tuple) and len(state) == 2 and isinstance(state[1], dict):
            self.__dict__.update(state[1])


def _make_stub(module, name):
    return type(name, (_Bag,), {"__module__": module})


# Globals the stub unpickler resolves for real: an EXACT (module, name) allowlist of tensor / container plumbing.
# Everything else a pickle names — ultralytics.* and torch.nn.* classes, but also os.system, builtins.eval or any
# callable that merely lives under torch / numpy (torch.hub.load, torch.utils.collect_env.run,
# numpy.testing._private.utils.runstring ...) — becomes an inert attribute bag, so a crafted .pt cannot execute code
# through this loader (upstream's plain torch.load(weights_only=False) would).  nn.Module instances come back as bags
# whose __dict__ still holds _parameters / _buffers / _modules: all _walk_module needs.
_TORCH_STORAGES = {"FloatStorage", "HalfStorage", "DoubleStorage", "BFloat16Storage", "LongStorage", "IntStorage",
                   "ShortStorage", "CharStorage", "ByteStorage", "BoolStorage"}
_TORCH_DTYPES = {"float16", "float32", "float64", "bfloat16", "int8", "int16", "int32", "int64", "uint8", "bool",
                 "half", "float", "double", "long", "int", "short"}
_SAFE_GLOBALS = {
    ("collections", "OrderedDict"),
    ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"),
    ("torch._utils", "_rebuild_parameter"), ("torch._utils", "_rebuild_parameter_with_state"),
    ("torch", "Size"), ("torch", "device"), ("torch", "Tensor"), ("torch.nn.parameter", "Parameter"),
    ("torch.storage", "UntypedStorage"), ("torch.storage", "TypedStorage"),
    ("numpy", "ndarray"), ("numpy", "dtype"),
    ("numpy.core.multiarray", "_reconstruct"), ("numpy.core.multiarray", "scalar"),
    ("numpy._core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "scalar"),
    ("_codecs", "encode"),
} | {("torch", n) for n in _TORCH_STORAGES | _TORCH_DTYPES}
_SAFE_BUILTINS = {"set", "frozenset", "list", "dict", "tuple", "slice", "complex", "bytearray", "range", "object",
                  "int", "float", "bool", "str", "bytes"}


class _StubUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _SAFE_GLOBALS or (module == "builtins" and name in _SAFE_BUILTINS):
            try:
                return super().find_class(module, name)
            except Exception:
                return _make_stub(module, name)
        return _make_stub(module, name)


class _StubPickleModule:
    """``pickle_module`` for ``torch.load`` (needs Unpickler + load)."""
    __name__ = "padel_stub_pickle"
    Unpickler = _StubUnpickler

    @staticmethod
    def load(f, **kw):
        return _StubUnpickler(f, **kw).load()


def _walk_module(obj, prefix, out):
    d = getattr(obj, "__dict__", {})
    for group in ("_parameters", "_buffers"):
        for k, v in (d.get(group) or {}).items():
            if v is not None and torch.is_tensor(v):
                out[f"{prefix}{k}"] = v
    for k, m in (d.get("_modules") or {}).items():
        if m is not None:
            _walk_module(m, f"{prefix}{k}.", out)


def _to_np(t):
    if torch.is_tensor(t):
        t = t.detach().cpu()
        if t.is_floating_point():
            t = t.float()          # checkpoints are fp16; upstream does model.float() first
        return t.numpy()
    return np.asarray(t)


# ----------------------------------------------------------------------------- public API

def save_checkpoint(path, state_dict, task, nc=0, kpt_shape=None, scale=None, names=None,
                    param_dict=None) -> None:
    """Write a plain-dict checkpoint (synthetic weights)."""
    sd = OrderedDict((k, torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray)
                      else torch.as_tensor(v)) for k, v in state_dict.items())
    torch.save({
        "padel_format": 1, "task": task, "nc": int(nc),
        "kpt_shape": tuple(kpt_shape) if kpt_shape else None, "scale": scale,
        "names": dict(names or {}), "param_dict": dict(param_dict or {}), "state_dict": sd,
    }, str(path))


def load_checkpoint(path) -> Checkpoint:
    """Load either checkpoint flavour; raises ``ValueError`` on an unrecognised layout."""
    # plain-dict checkpoints (ours, TrackNetV3's) load under torch's safe unpickler; only Ultralytics pickles
    # (whole nn.Modules) need the stub unpickler, which never resolves a non-tensor global (see above)
    try:
        obj = torch.load(str(path), map_location="cpu", weights_only=True)
    except Exception:
        obj = torch.load(str(path), map_location="cpu", weights_only=False,
                         pickle_module=_StubPickleModule)
    if isinstance(obj, dict) and obj.get("padel_format") == 1:
        sd = OrderedDict((k, _to_np(v)) for k, v in obj["state_dict"].items())
        return Checkpoint(sd, obj["task"], obj.get("nc", 0), obj.get("kpt_shape"),
                          obj.get("scale"), obj.get("names") or {}, obj.get("param_dict") or {})
    if isinstance(obj, dict) and "param_dict" in obj and "model" in obj and isinstance(obj["model"], dict):
        # TrackNetV3 release format (ball_tracker.py:253-265)
        sd = OrderedDict((k, _to_np(v)) for k, v in obj["model"].items())
        task = "inpaintnet" if any(k.startswith("buttleneck") for k in sd) else "tracknet"
        return Checkpoint(sd, task, param_dict=dict(obj["param_dict"]))
    if isinstance(obj, dict) and ("model" in obj or "ema" in obj):
        model = obj.get("ema") or obj.get("model")
        tensors: dict = {}
        _walk_module(model, "", tensors)
        if not tensors:
            raise ValueError(f"{path}: could not recover tensors from the pickled model")
        sd = OrderedDict((k, _to_np(v)) for k, v in tensors.items())
        info = yolo_arch.infer_arch_from_state_dict(sd)
        yaml = getattr(model, "yaml", None) or {}
        kpt_shape = yaml.get("kpt_shape") if isinstance(yaml, dict) else None
        if info["nk"] and not kpt_shape:
            # no model.yaml in the pickle: nk = K * ndim is only decidable when exactly one of ndim 2 / 3 divides
            # it (13 x 3 = 39 is; the 12-keypoint 2-D court model's 24 is NOT: it would read as 8 x 3)
            nk = info["nk"]
            if nk % 3 == 0 and nk % 2 == 0:
                raise ValueError(f"{path}: pose head with {nk} outputs and no kpt_shape in the checkpoint is ambiguous "
                                 f"({nk // 2} x 2 or {nk // 3} x 3); re-save it with save_checkpoint(..., kpt_shape=...)")
            kpt_shape = (nk // 3, 3) if nk % 3 == 0 else (nk // 2, 2)
        names = getattr(model, "names", None) or {}
        return Checkpoint(sd, "pose" if info["nk"] else "detect", info["nc"],
                          tuple(kpt_shape) if kpt_shape else None, info["scale"], dict(names))
    raise ValueError(f"{path}: unrecognised checkpoint layout")


def make_synthetic_yolo(path, scale, nc, kpt_shape=None, seed=0, cls_bias=-4.0, names=None) -> None:
    """Seeded synthetic YOLOv8 detect/pose checkpoint (SURVEY.md §8(d) weight recipe)."""
    sd = yolo_arch.synth_state_dict(scale, nc, kpt_shape, seed, cls_bias)
    if names is None:
        names = {i: ("person" if i == 0 else f"class{i}") for i in range(nc)}
    save_checkpoint(path, sd, "pose" if kpt_shape else "detect", nc, kpt_shape, scale, names)
