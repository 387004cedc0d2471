import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

import __graft_entry__ as ge  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return ge.load_package()


@pytest.fixture(scope="session")
def po():
    return ge.load_oracle()


@pytest.fixture(scope="session")
def model_small(tmp_path_factory, pkg, po):
    """hidden=128 synthetic model written in the reference's file format; (path, oracle model, file tensors)."""
    d = tmp_path_factory.mktemp("model128")
    path = str(d / "ggml-model-synth128-u8.bin.gz")
    pkg.ggml.write_model(path, pkg.ggml.synth_weights(128, seed=3), 128)
    hidden, targets = pkg.ggml.read_model(path)
    return path, po.Model.load(path), targets


def rel_l2(a, b):
    a = np.asarray(a, np.float64) if not np.iscomplexobj(a) else np.asarray(a, np.complex128)
    b = np.asarray(b, np.float64) if not np.iscomplexobj(b) else np.asarray(b, np.complex128)
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-30))


def sdr_db(ref, est):
    ref = np.asarray(ref, np.float64)
    est = np.asarray(est, np.float64)
    return float(10 * np.log10(np.sum(ref ** 2) / max(np.sum((ref - est) ** 2), 1e-300)))
