import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_engine():
    """The HIP engine on cuda:0 — fails loudly (no skip, no CPU fallback) if the extension is missing."""
    from padel_analytics_amd import engine
    return engine.default_engine(0)


# The oracle-parity files run FIRST under -m gpu: the driver's step has a time limit, and whatever it cuts off
# should be the kernel-variant sweeps, not the parity statements.  Within a file pytest's order is kept.
_GPU_FILE_ORDER = ("test_gpu_yolo_parity.py", "test_gpu_bench_config.py", "test_gpu_baseline_configs.py", "test_gpu_ball.py",
                   "test_gpu_known_answers.py", "test_gpu_runner.py", "test_gpu_pipeline.py", "test_gpu_nms_stress.py",
                   "test_gpu_h2.py", "test_gpu_conv.py", "test_gpu_fp16.py")


def pytest_collection_modifyitems(session, config, items):
    rank = {name: i for i, name in enumerate(_GPU_FILE_ORDER)}
    items.sort(key=lambda it: rank.get(Path(str(it.fspath)).name, len(rank)))      # stable: CPU files keep their places behind
