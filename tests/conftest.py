import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle runs B = 1 convolutions: on the GPU box's 256 logical CPUs torch's default thread count makes them
    # 3-7 x SLOWER than 16 threads do (bench.py's cpu_baseline ladder: 16 threads 0.35 utt/s, 128 threads 0.05)
    try:
        import torch
        torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    except Exception:
        pass


# Arithmetic modes: the product library has "f32", "f32w", "bf16x6"; the measured alternatives ("bf16x3", "bf16x3p", "f16x3",
# "bf16x6w") exist only in the experiment build (csrc: make exp) and their tests run only when that library is loaded.
PRODUCT_MODES = ("f32", "f32w", "bf16x6")
ALT_MODES_BUILT = os.environ.get("MISONET_LIB_PATH", "").endswith("_exp.so")


def modes(*names):
    """the subset of `names` that the loaded library implements"""
    return [m for m in names if m in PRODUCT_MODES or ALT_MODES_BUILT]


needs_alt_modes = pytest.mark.skipif(not ALT_MODES_BUILT, reason="mode of the experiment build (MISONET_LIB_PATH=.../libmisonet_hip_exp.so)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def rel_l2(a, b):
    a = np.asarray(a); b = np.asarray(b)
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-30))


def mag_parity(x_hat, x, tol=1e-3):
    """SURVEY.md 8(d) parity metric on complex-spectrogram magnitudes.

    Returns (rel_l2 of magnitudes, fraction of bins violating | |x^|-|x| | <= tol*|x| + tol*median|x|)."""
    mh, m = np.abs(np.asarray(x_hat)), np.abs(np.asarray(x))
    r = float(np.linalg.norm((mh - m).ravel()) / max(np.linalg.norm(m.ravel()), 1e-30))
    bad = np.abs(mh - m) > tol * m + tol * np.median(m)
    return r, float(bad.mean())


@pytest.fixture(scope="session")
def sd1():
    from misonet_amd import weights as W
    return W.make_state_dict(W.miso1_spec(), seed=0)


@pytest.fixture(scope="session")
def sd3():
    from misonet_amd import weights as W
    return W.make_state_dict(W.miso3_spec(), seed=1)
