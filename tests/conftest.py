import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
PKG = os.path.join(ROOT, "multilateral-temporal-view-pyramid-transformer-for-video-inpainting-detection_amd")
for p in (ROOT, GOLDEN, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


# the oracle (CPU torch) is the slow half of every whole-model test: a GPU box reports all the host's cores while a job owns a share
# of them, and torch's default of one thread per reported core oversubscribes that share
torch.set_num_threads(max(1, min(16, os.cpu_count() or 8)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def have_gpu() -> bool:
    return torch.cuda.is_available()


@pytest.fixture(scope="session")
def ops_golden():
    return np.load(os.path.join(GOLDEN, "ops.npz"))


@pytest.fixture(scope="session")
def full_golden():
    return np.load(os.path.join(GOLDEN, "full_model.npz"))


@pytest.fixture(scope="session")
def full_golden_t9():
    """B=1, T=9 (config 4's temporal length at 224x224) from the real reference: tests/golden/gen_goldens_t9.py."""
    return np.load(os.path.join(GOLDEN, "full_model_t9.npz"))


@pytest.fixture(scope="session")
def train_golden():
    return np.load(os.path.join(GOLDEN, "train_tail.npz"))


@pytest.fixture(scope="session")
def index_golden():
    return np.load(os.path.join(GOLDEN, "index_maps.npz"))


def golden_input(store, name):
    """Regenerate a seeded input recorded by gen_goldens.inp()."""
    from weight_fill import seeded_randn
    ss = [int(v) for v in store[name + "/seed_shape"]]
    return seeded_randn(ss[0], *ss[1:])


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max |a-b| / max |b|  — the 'relative' measure used for the 1e-3 fp32 parity bar."""
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def rms_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """RMS of the difference over the RMS of the reference: a per-tensor measure that one large element cannot hide behind
    (rel_err normalises the largest difference by the largest reference value)."""
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-30))


def check_digest(t: torch.Tensor, store, name, tol):
    from weight_fill import digest
    s, v = digest(t)
    gs, gv = store[name + "/stats"], store[name + "/samples"]
    scale = float(gs[3]) + 1e-30
    assert np.abs(v - gv).max() / scale < tol, f"{name} samples differ"
    assert abs(s[2] - gs[2]) / (gs[2] + 1e-30) < tol, f"{name} l2 differs"
