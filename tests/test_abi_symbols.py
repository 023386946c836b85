"""The C-ABI library loads without a GPU and exports every function include/padel_hip.h declares."""
import ctypes
import re
from pathlib import Path

from padel_analytics_amd import engine as E

ROOT = Path(__file__).resolve().parents[1]


def _declared():
    txt = (ROOT / "include" / "padel_hip.h").read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pa_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    assert E.lib_path().exists(), "libpadel_hip.so not built (run __graft_entry__.build())"
    lib = ctypes.CDLL(str(E.lib_path()))
    names = _declared()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} missing from libpadel_hip.so"
    assert sorted(E.ABI_SYMBOLS) == names
    lib.pa_abi_version.restype = ctypes.c_int
    assert lib.pa_abi_version() == 5


def test_no_gpu_fails_loudly():
    """Without a GPU the engine must raise, never fall back to a CPU path."""
    lib = E.load_library()
    if lib.pa_device_count() > 0:
        return
    import pytest
    with pytest.raises(E.EngineUnavailable):
        E.Engine(0)
