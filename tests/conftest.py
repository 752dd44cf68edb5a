"""Shared fixtures.  `-m "not gpu"` runs on the CPU-only build container; `-m gpu` needs an MI355X."""
import importlib
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent
GOLDEN = REPO / "tests" / "golden"
for p in (str(REPO), str(GOLDEN)):
    if p not in sys.path:
        sys.path.insert(0, p)

PKG = "diffusion-motion-inbetweening_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


@pytest.fixture(scope="session")
def condmdi():
    """The product package (hyphenated directory name -> importlib)."""
    return importlib.import_module(PKG)


def sub(name):
    return importlib.import_module(f"{PKG}.{name}")


@pytest.fixture(scope="session")
def cases():
    return importlib.import_module("cases")


def load_golden(name):
    return np.load(GOLDEN / f"{name}.npz")


def check_fingerprint(cases_mod, name, inputs):
    fp = load_golden(name)["fingerprint"]
    assert np.array_equal(fp, cases_mod.fingerprint(inputs)), \
        f"inputs of golden case {name} no longer regenerate bit-identically (numpy RNG drift?)"


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def max_abs(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))
