import os
import sys

import numpy as np
import pytest

# torch wheels bundle their own HIP runtime (torch/lib/libamdhip64.so): whichever of torch / libfd_hip.so is loaded first decides
# which runtime the process uses, and torch initialised AFTER the system runtime reports "No HIP GPUs are available".  The GPU
# tests that stage frames in HBM through torch therefore need torch first -- exactly the order bench.py uses.
try:
    import torch  # noqa: F401
except Exception:   # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def synth():
    from featuredetection_amd import synth
    return synth


@pytest.fixture(scope="session")
def capi():
    from featuredetection_amd import capi
    capi.lib()  # fails loudly when the HIP extension is not built
    return capi


@pytest.fixture(scope="session")
def ctx(capi):
    c = capi.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def frame640(synth):
    return synth.make_frame(640, 480, seed=20260927)


@pytest.fixture(scope="session")
def small_models(synth, oracle, frame640):
    """A small but complete cascade: 20x20 WVM (6 per level x 5 levels) + 96-SV RBF SVM."""
    gray = oracle.bgr2gray(frame640)
    rng = np.random.default_rng(5)
    calib = synth.random_patches(gray[::4, ::4].copy(), 20, 20, 4000, rng)
    wvm = synth.make_wvm(11, n_per=6, n_levels=5, calib_patches=calib, min_survivors=64)
    eq = synth.histeq64_np(synth.random_patches(gray[::4, ::4].copy(), 20, 20, 600, rng))
    svm = synth.make_svm_u8(3, eq, nsv=96, calib=eq[96:], positive_fraction=0.4)
    return wvm, svm
