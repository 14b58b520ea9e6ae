"""Robustness of the library's process-level machinery: two caller threads with their own contexts share the
process-wide host-stage queue, and the FD_* environment knobs are clamped so that nonsense values change speed, never results."""
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FF = dict(inc=float(np.float32(0.92)), min_scale=float(np.float32(0.05)), max_scale=float(np.float32(0.16)))


def test_two_caller_threads_with_their_own_contexts(capi, ctx, synth, small_models):
    """Contexts are single-threaded by contract, but a process may run several: two caller threads, each with its own context, pyramid
    and model handles, keep ticket calls (fd_detect_five_stage_frames_begin / _end) in flight at the same time; the library's
    process-wide queue threads serve both.  Every result equals the one a single thread gets."""
    wvm, svm = small_models
    NF, ROUNDS = 5, 6
    frames = [[synth.make_frame(320, 240, seed=7000 + 100 * t + i) for i in range(NF)] for t in range(2)]
    ref = []
    for t in range(2):   # single-threaded reference on the session context
        p = capi.Pyramid(ctx, **FF)
        p.set_frames(NF)
        p.update_frames(images=frames[t])
        w, s = capi.Wvm(ctx, wvm), capi.Svm(ctx, svm)
        ref.append(capi.detect_five_stage_frames(ctx, p, w, s, NF))
        w.close(); s.close(); p.close()
    assert sum(len(d) for r in ref for d, _ in r) > 0
    errors = []

    def worker(t):
        try:
            c = capi.Context(0)
            sets = []
            for _ in range(2):   # two calls in flight per thread
                p = capi.Pyramid(c, **FF)
                p.set_frames(NF)
                sets.append((p, capi.Wvm(c, wvm), capi.Svm(c, svm)))
            for r in range(ROUNDS):
                tickets = []
                for p, w, s in sets:
                    p.update_frames(images=frames[t])
                    tickets.append(capi.FiveStageFrames(c, p, w, s, NF))
                for tk in tickets:
                    res = tk.end()
                    for (d, st), (dr, sr) in zip(res, ref[t]):
                        if d.tobytes() != dr.tobytes() or not np.array_equal(st, sr):
                            errors.append("thread %d round %d differs" % (t, r))
            for p, w, s in sets:
                w.close(); s.close(); p.close()
            c.close()
        except Exception as e:   # noqa: BLE001
            errors.append("thread %d: %r" % (t, e))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join(timeout=300)
    assert not any(x.is_alive() for x in th), "a caller thread hangs"
    assert not errors, errors


NONSENSE = {
    "FD_ASYNC_THREADS": "-7", "FD_BATCH_THREADS": "0", "FD_BATCH_STREAMS": "9999", "FD_WVM_GRID_PER_CU": "-3", "FD_WVM_ROUNDS": "100000000",
    "FD_WVD_ROUNDS": "-1", "FD_WVM_POS_CAP": "banana", "FD_WVM_DEEP_CAP": "-12", "FD_WVB_PHASES": ",,x,0,-4,1,1,99999", "FD_WVM_DEEPB_PER_CU": "0",
    "FD_WVB_EXIT_PER_CU": "-2", "FD_SVM_KERNEL": "77", "FD_WVM_DEEP_WAVES": "3", "FD_WVB_ADAPT": "maybe", "FD_PYR_FUSED": "", "FD_WVD_K": "-5", "FD_WVB_PREP_LANES": "-9", "FD_SVM_WAVES": "7",
}


def test_nonsense_environment_knobs_do_not_change_results():
    """every FD_* tuning knob is read through a clamp: with absurd values the smoke run (which compares the HIP path with the oracle:
    all windows of a WVM run, a five-stage run, a multi-frame ticket call) still passes in a fresh process"""
    env = dict(os.environ)
    env.update(NONSENSE)
    code = "import torch, __graft_entry__ as g; g.smoke(); print('SMOKE_OK')"
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SMOKE_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


@pytest.mark.parametrize("k", [2, 5, 16])
def test_prefilter_column_walks_of_every_length(k):
    """k_wvm_prefilter's lanes walk down their column through K windows, sliding the histogram; the library picks K per launch (1 for the
    small pyramids of most tests, 3 for the 64-frame headline, 16 on 1080p layers).  With K pinned (FD_WVD_K, read once per process) the
    production path must still give the exact path's positives byte for byte -- window steps 1, 2 and 3 (two and three rows leave and enter
    per window), ragged row groups, a roi, six patch sizes -- and the 64-frame headline check against the oracle, the threshold-tie cases
    of all five patch sizes and the late-rejecting models must still pass."""
    env = dict(os.environ)
    env["FD_WVD_K"] = str(k)
    env["FD_WVB_PREP_LANES"] = "1"   # and stage B's lane == window prepare kernel for every queue, however short (default: from 32 K windows)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "tests/test_gpu_fullsize.py", "tests/test_gpu_cascade_hardening.py", "-k",
                        "production_path_equals_exact or headline_workload_against_the_oracle or exact_threshold_ties or rejection_profiles"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " passed" in r.stdout and "failed" not in r.stdout, (r.stdout[-3000:], r.stderr[-2000:])
