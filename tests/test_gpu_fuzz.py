"""Seeded differential fuzzing (tools/fuzz_parity.py) as part of the GPU suite: randomised frame sizes, pyramid
parameters, patch shapes, strides, ROIs, model shapes and filter parameters, HIP path vs the CPU oracle."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fuzz():
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "tools", "fuzz_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("kind", ["pyramid", "cascade", "frames", "hist", "fhog", "aggregated", "svm", "hog_svm", "rvm", "whi", "sdm", "hog_fused", "batch_tail", "batch_group"])
def test_fuzzed_parity(capi, ctx, oracle, kind):
    fz = _fuzz()
    bad, ran = [], 0
    for i in range(6):
        r = fz.CASES[kind](np.random.default_rng([20260927, sorted(fz.CASES).index(kind), i]), ctx)
        ran += 1
        if r is not None and not r.startswith(("skip:", "note:")):
            bad.append((i, r))
    assert not bad, bad
    assert ran == 6
