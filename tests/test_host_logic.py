"""Host-side logic of the product library that needs no GPU: OverlapElimination and block NMS through
the C ABI against the oracle, the exported symbol table against include/fd_hip.h, and the loud
failure when no device is present."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _random_dets(oracle, capi, rng, n, w=640, h=480, tie=False):
    o = np.zeros(n, oracle.DET_DTYPE)
    o["cx"] = rng.integers(0, w, n)
    o["cy"] = rng.integers(0, h, n)
    o["w"] = rng.choice([120, 135, 147, 160, 200], n)
    o["h"] = o["w"]
    o["prob"] = 0.5 if tie else rng.random(n)
    c = np.zeros(n, capi.DET_DTYPE)
    for f in ("cx", "cy", "w", "h"):
        c[f] = o[f]
    c["probability"] = o["prob"]
    return o, c


def test_symbols_match_header(capi):
    hdr = open(os.path.join(ROOT, "include", "fd_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(fd_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) > 30
    lib = capi.lib()
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, "symbols declared in include/fd_hip.h but not exported: %s" % missing
    assert set(capi._SIGS) == names
    # measurement hooks live in their own header, outside the drop-in boundary
    bh = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "fd_hip_bench.h")).read(), flags=re.S)
    bnames = set(re.findall(r"\b(fd_[a-z0-9_]+)\s*\(", bh))
    assert bnames == set(capi._BENCH_SIGS) and not (bnames & names)
    assert all(hasattr(lib, n) for n in bnames)
    assert "fd_bench" not in hdr


@pytest.mark.parametrize("dist,ratio", [(5.0, 0.0), (0.5, 0.0), (20.0, 0.8), (1.0, 1.5)])
def test_overlap_elimination_matches_oracle(oracle, capi, dist, ratio):
    rng = np.random.default_rng(int(dist * 10 + ratio * 100))
    for n in (0, 1, 2, 17, 300):
        o, c = _random_dets(oracle, capi, rng, n, w=200, h=150)
        assert np.array_equal(oracle.overlap_elimination(o, dist, ratio), capi.overlap_elimination(c, dist, ratio))


def test_overlap_elimination_large_clustered(oracle, capi):
    """The hashed-grid implementation against the reference-shaped O(n^2) loop at the sizes config 3 produces
    (thousands of WVM positives, clustered, also with negative coordinates)."""
    rng = np.random.default_rng(77)
    n = 6000
    centres = rng.integers(-40, 1900, (60, 2))
    pick = rng.integers(0, 60, n)
    o, c = _random_dets(oracle, capi, rng, n, w=1920, h=1080)
    o["cx"] = centres[pick, 0] + rng.integers(-15, 16, n)
    o["cy"] = centres[pick, 1] + rng.integers(-15, 16, n)
    o["prob"] = np.round(rng.random(n), 3)   # many equal probabilities
    for f in ("cx", "cy"):
        c[f] = o[f]
    c["probability"] = o["prob"]
    for dist, ratio in ((5.0, 0.0), (0.05, 0.7), (12.5, 0.0)):
        assert np.array_equal(oracle.overlap_elimination(o, dist, ratio), capi.overlap_elimination(c, dist, ratio))


def test_overlap_elimination_extreme_extents_use_the_hash_grid(oracle, capi):
    """Centres spread over +-2^30 make the dense cell grid too large: the open-addressed fallback must give the same result."""
    rng = np.random.default_rng(5)
    n = 3000
    o, c = _random_dets(oracle, capi, rng, n, w=1920, h=1080)
    far = rng.integers(0, n, n // 7)
    o["cx"][far] = rng.integers(-2 ** 30, 2 ** 30, len(far))
    o["cy"][far[::2]] = rng.integers(-2 ** 30, 2 ** 30, len(far[::2]))
    o["prob"] = np.round(rng.random(n), 2)
    for f in ("cx", "cy"):
        c[f] = o[f]
    c["probability"] = o["prob"]
    for dist, ratio in ((5.0, 0.0), (0.3, 0.5)):
        assert np.array_equal(oracle.overlap_elimination(o, dist, ratio), capi.overlap_elimination(c, dist, ratio))


def test_block_nms_random_maps_with_out_of_range_and_nonpositive_entries(oracle, capi):
    """sorted-vector block NMS against the dense restatement on random sparse maps: detections outside the image and with
    probability <= 0 must be ignored exactly like the dense map does (the map keeps its 0)."""
    rng = np.random.default_rng(12)
    for trial in range(40):
        W, H = int(rng.integers(40, 400)), int(rng.integers(40, 300))
        n = int(rng.integers(1, 2500))
        d = np.zeros(n, capi.DET_DTYPE)
        d["cx"] = rng.integers(-3, W + 3, n); d["cy"] = rng.integers(-3, H + 3, n); d["w"] = 20; d["h"] = 20
        d["probability"] = np.round(rng.random(n), 2) - 0.05
        pmap = np.zeros((H, W), np.float32)
        for q in d:
            if 0 <= q["cx"] < W and 0 <= q["cy"] < H and pmap[q["cy"], q["cx"]] < q["probability"]:
                pmap[q["cy"], q["cx"]] = q["probability"]
        for masked in (True, False):
            mask = ((pmap > np.float32(0.3)) * 255).astype(np.uint8) if masked else None
            ys, xs = np.nonzero(oracle.block_nms(pmap, 35, mask))
            got = capi.block_nms(d, W, H, 35, masked)
            assert np.array_equal(np.asarray(got).reshape(-1, 2), np.stack([xs, ys], 1).astype(np.int32).reshape(-1, 2)), (trial, masked)


@pytest.mark.parametrize("mtype", [0, 1, 2])
def test_iou_nms_matches_oracle(oracle, capi, mtype):
    """detection::NonMaximumSuppression (IoU clustering) through the C ABI against the oracle, plus the defining property
    for MAX_SCORE: the kept boxes are the cluster heads and no two of them overlap more than the threshold allows at the
    time they were picked."""
    rng = np.random.default_rng(40 + mtype)
    for n, thr in ((0, 0.3), (1, 0.3), (50, 0.3), (400, 0.5), (400, 0.0), (30, 1.0)):
        b = np.zeros(n, capi.BOX_DTYPE)
        centres = rng.integers(0, 500, (max(n // 8, 1), 2))
        pick = rng.integers(0, len(centres), n)
        b["x"] = centres[pick, 0] + rng.integers(-12, 13, n)
        b["y"] = centres[pick, 1] + rng.integers(-12, 13, n)
        b["w"] = rng.integers(20, 80, n)
        b["h"] = rng.integers(20, 80, n)
        b["score"] = np.round(rng.random(n) + 0.1, 2).astype(np.float32)   # ties included
        got = capi.nms_iou(b, thr, mtype)
        so, bo = oracle.nms_iou(b["score"], np.stack([b["x"], b["y"], b["w"], b["h"]], 1) if n else np.zeros((0, 4), np.int32), thr, mtype)
        assert len(got) == len(so)
        assert np.array_equal(got["score"], so)
        assert np.array_equal(np.stack([got["x"], got["y"], got["w"], got["h"]], 1) if len(got) else np.zeros((0, 4), np.int32), bo)
        if mtype == 0 and thr < 1.0 and n:
            assert np.all(np.diff(got["score"]) <= 0)   # cluster heads come out best first
    with pytest.raises(RuntimeError):
        capi.nms_iou(np.array([(1.0, 0, 0, 10, 10)], capi.BOX_DTYPE), 1.5)


def test_overlap_elimination_ties(oracle, capi):
    rng = np.random.default_rng(9)
    o, c = _random_dets(oracle, capi, rng, 200, w=120, h=90, tie=True)
    assert np.array_equal(oracle.overlap_elimination(o, 5.0, 0.0), capi.overlap_elimination(c, 5.0, 0.0))


@pytest.mark.parametrize("masked", [True, False])
def test_block_nms_matches_dense_oracle(oracle, capi, masked):
    rng = np.random.default_rng(4)
    for n, tie in ((0, False), (1, False), (40, False), (40, True), (400, False), (400, True)):
        W, H = 320, 240
        o, c = _random_dets(oracle, capi, rng, n, w=W, h=H, tie=tie)
        pmap = np.zeros((H, W), np.float32)
        for d in o:
            if pmap[d["cy"], d["cx"]] < d["prob"]:
                pmap[d["cy"], d["cx"]] = d["prob"]
        mask = ((pmap > np.float32(0.3)) * 255).astype(np.uint8) if masked else None
        dense = oracle.block_nms(pmap, 35, mask)
        ys, xs = np.nonzero(dense)
        got = capi.block_nms(c, W, H, 35, masked)
        assert np.array_equal(got, np.stack([xs, ys], 1).astype(np.int32).reshape(-1, 2)), (n, tie)


def test_no_silent_cpu_fallback(capi):
    """On a box without a GPU every compute entry point must fail loudly (FD_ERR_HIP)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.FdError):
        capi.Context(0)


def test_c_abi_header_is_plain_c_and_the_frame_loop_binding_compiles(tmp_path):
    """include/fd_hip.h is the drop-in boundary: it must be valid C99 on its own, and the frame-loop binding INTEGRATION.md shows for
    the multi-frame entry points must compile against it (syntax only: no GPU, no linking)."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("gcc not on PATH")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    inc = os.path.join(root, "include")
    for h in ("fd_hip.h", "fd_hip_bench.h"):
        r = subprocess.run([gcc, "-fsyntax-only", "-x", "c", "-std=c99", "-Wall", "-Werror", os.path.join(inc, h)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    src = tmp_path / "frame_loop.c"
    src.write_text("""
#include <stdint.h>
#include "fd_hip.h"
int frame_loop(fd_ctx* ctx, fd_pyramid* pyr, const fd_wvm* wvm, const fd_svm* svm, const uint8_t* const* frames) {
    fd_five_stage_frames* t = 0;
    static fd_detection out[32 * 256];
    int32_t counts[32], stages[32 * 4];
    int rc = fd_pyramid_set_frames(pyr, 32);
    if (rc) return rc;
    rc = fd_pyramid_update_frames(pyr, frames, 32, 640, 480, 3, 1);
    if (rc) return rc;
    rc = fd_detect_five_stage_frames_begin(ctx, pyr, wvm, svm, 5.f, 0.f, 1, 1, 0, &t);
    if (rc) return rc;
    return fd_detect_five_stage_frames_end(ctx, t, out, 256, counts, stages);
}
""")
    r = subprocess.run([gcc, "-fsyntax-only", "-std=c99", "-Wall", "-Werror", "-I", inc, str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@pytest.mark.parametrize("shape", [(20, 20, 14, 3, 6, (2, 8)), (24, 24, 30, 2, 6, (2, 8)), (32, 24, 9, 3, 12, (1, 8)), (19, 21, 9, 4, 6, (2, 8)),
                                   (16, 24, 20, 2, 16, (6, 8)), (7, 5, 3, 8, 6, (2, 8)), (20, 20, 40, 1, 6, (2, 8)),
                                   # numUsedFilters that is not a multiple of numFiltersPerLevel: classes with different numbers of
                                   # generations (k_wvb_chain2 computes a level's operand tile from (class, generation) in closed form)
                                   (20, 20, 14, 3, 6, (2, 8), 37), (24, 24, 30, 2, 6, (2, 8), 47), (32, 24, 9, 3, 12, (1, 8), 22),
                                   (20, 20, 14, 20, 6, (2, 4), 269), (16, 24, 20, 2, 16, (6, 8), 21)])
def test_stage_b_tables_reproduce_the_rect_sums(capi, synth, shape):
    """The dense stage B (wvm_stageb.hpp) replaces the rect lookups on the integral image (WvmClassifier.cpp:277-306) by an int8
    contraction against per-(level, grey value) coverage counts.  The host-built operand tables, read with the kernel's own
    addressing, must give exactly the rect sums computed rect by rect."""
    pw, ph, nper, nlev, cntval, rr = shape[:6]
    wvm = synth.make_wvm(5, fw=pw, fh=ph, n_per=nper, n_levels=nlev, cntval=cntval, rect_range=rr)
    if len(shape) > 6:
        wvm["num_used"] = shape[6]
    rng = np.random.default_rng(pw * 100 + ph)
    patches = rng.integers(0, 256, (40, ph, pw), dtype=np.uint8)
    patches[0] = 255
    patches[1] = 0
    got = capi.wvb_rect_sums(wvm, patches)
    if wvm["num_used"] <= 16:
        assert got is None   # the one-wave-per-window stage finishes such models alone
        return
    sums, gens = got
    assert gens[0] == 0 and gens[-1] == -(-wvm["num_used"] // nper) and all(a < b for a, b in zip(gens, gens[1:]))
    ii = np.zeros((len(patches), ph + 1, pw + 1), np.int64)
    ii[:, 1:, 1:] = patches.astype(np.int64).cumsum(1).cumsum(2)
    c = 0
    for k in range(wvm["num_used"]):
        v0, v1 = wvm["val_off"][k], wvm["val_off"][k + 1]
        for v in range(v0 + 1, v1):
            want = np.zeros(len(patches), np.int64)
            for r in range(wvm["rec_off"][v], wvm["rec_off"][v + 1]):
                x1, y1, x2, y2 = (int(q) for q in wvm["rects"][r])
                want += ii[:, y2 + 1, x2 + 1] - ii[:, y1, x2 + 1] - ii[:, y2 + 1, x1] + ii[:, y1, x1]
            assert np.array_equal(sums[:, c], want), (k, v - v0)
            c += 1
    assert c == sums.shape[1]


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_prefilter_plan_covers_every_window_once(capi, seed):
    """k_wvm_prefilter's lanes walk down a column through K windows.  The host's plan (fd_debug_wvd_plan, no GPU): K from the cost
    model (1 when the vertical step leaves nothing to slide, more when there are more tiles than wavefront slots), the tile list
    contiguous, and the kernel's task -> (column, row group) mapping reaches every window of every layer exactly once."""
    rng = np.random.default_rng(seed)
    for _ in range(40):
        n = int(rng.integers(1, 14))
        nx = rng.integers(1, 400, n).astype(np.int32)
        ny = rng.integers(1, 300, n).astype(np.int32)
        frames = int(rng.choice([1, 1, 8, 64]))
        sy = int(rng.choice([1, 1, 2, 3, 12]))
        ph = int(rng.choice([16, 20, 24]))
        slots = int(rng.choice([256, 3072]))
        k, first = capi.wvd_plan(nx, ny, frames, sy, ph, slots)
        assert 1 <= k <= 16
        if 2 * sy > ph:
            assert k == 1
        assert first[0] == 0
        for i in range(n):
            g = -(-int(ny[i]) // k)
            assert first[i + 1] - first[i] == -(-int(nx[i]) * g // 64), (i, k)
            seen = np.zeros((int(ny[i]), int(nx[i])), np.int32)
            t = np.arange(int(nx[i]) * g)
            col, grp = t % int(nx[i]), t // int(nx[i])
            for s in range(k):
                row = grp * k + s
                ok = row < int(ny[i])
                np.add.at(seen, (row[ok], col[ok]), 1)
            assert np.all(seen == 1)
    # one tile's worth of work per slot at most: a single small frame is not walked at all, a big launch gets long columns
    assert capi.wvd_plan([77, 45], [53, 29], 1, 1, 20, 3072)[0] == 1
    assert capi.wvd_plan([1901], [1061], 1, 1, 24, 3072)[0] >= 8
    assert capi.wvd_plan([1901], [1061], 64, 1, 24, 3072)[0] >= 12


def test_probability_order_margin_of_the_device_overlap_elimination():
    """k_fs_oe (csrc/fs_tail.hpp) orders a frame's positives by the fp32 cascade output and claims that this is the order of
    the reference's double probabilities 1 / (1 + exp(A + B x)) whenever two sorted neighbours pass its gap test (relative distance of
    the two 1 + e values > 1e-14); everything else goes back to the host.  The claim rests on the host's libm: sweep pairs of fp32
    outputs -- adjacent floats and small multiples of an ulp apart, from the steep part of the logistic into its saturation -- and
    check, with THIS host's exp, that every pair the kernel would accept has strictly ordered probabilities, and that the rule does
    give up where the doubles collapse."""
    import math
    A, B = 0.00556, -2.95   # ProbabilisticWvmClassifier.hpp:36
    accepted = rejected = collapsed_accepted = 0
    for base in np.concatenate([np.linspace(-3.0, 14.0, 3001), np.linspace(11.0, 13.5, 2001)]).astype(np.float32):
        for k in (1, 2, 3, 5, 9, 33, 1025):
            x1 = np.float32(base)
            x2 = x1
            for _ in range(min(k, 40)):
                x2 = np.nextafter(x2, np.float32(np.inf))
            if k > 40:
                x2 = np.float32(x1 + np.float32(k) * np.spacing(x1))
            if x2 == x1:
                continue
            if x2 < x1:   # np.spacing carries the sign of its argument
                x1, x2 = x2, x1
            t1, t2 = A + B * float(x1), A + B * float(x2)
            ex = math.exp(min(t1, t2))
            gap = ex / (1.0 + ex) * abs(B) * abs(float(x1) - float(x2))
            p1 = 1.0 / (1.0 + math.exp(t1))   # wvm_probability (csrc/wvm.hip) = ProbabilisticWvmClassifier.cpp:52, evaluated in double
            p2 = 1.0 / (1.0 + math.exp(t2))
            if gap > 1e-14:   # the kernel keeps the fp32 order: B < 0, so the larger output must have the larger probability
                accepted += 1
                assert p2 > p1, (float(x1), float(x2), p1, p2, gap)
            else:
                rejected += 1
            collapsed_accepted += int(p1 == p2 and gap > 1e-14)
    assert accepted > 10000 and rejected > 100 and collapsed_accepted == 0
