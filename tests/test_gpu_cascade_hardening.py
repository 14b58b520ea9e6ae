"""Parity hardening of the production WVM cascade (VERDICT r02 task 1 + 3): the dense pre-filter rejects on an error-bounded
inequality and stage B evaluates the survivors as dense contractions, so the places where a single ulp decides are tested
explicitly: thresholds placed exactly ON a real window's filter output (and one ulp either side), sums of squares around 2^24,
kernel arguments beyond the fast-exp range, weights with heavy cancellation, late-rejecting models, and the headline workload
itself against the oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
FF = dict(inc=float(np.float32(0.92)), min_scale=float(np.float32(0.05)), max_scale=float(np.float32(0.16)))  # FaceFrontal.cfg
EAR = dict(inc=float(np.float32(0.9)), min_scale=float(np.float32(0.5)), max_scale=float(np.float32(0.7)))


def _pyr_pair(oracle, capi, ctx, frame, **kw):
    po = oracle.Pyramid(**kw)
    po.update(frame)
    pg = capi.Pyramid(ctx, **kw)
    pg.update(frame)
    return po, pg


def _eq_patches(oracle, po, pw, ph, idx):
    wins = po.windows(pw, ph, 1, 1)
    layers = {}
    out = []
    for i in idx:
        l, x, y = int(wins[i, 0]), int(wins[i, 1]), int(wins[i, 2])
        if l not in layers:
            layers[l] = po.layer(l)
        out.append(oracle.histeq64(layers[l][y:y + ph, x:x + pw].copy()))
    return out


def _level_output(oracle, model, k, patch_eq):
    """res_k of one window (WvmClassifier.cpp:335-341), from the oracle: a copy of the model that stops at level k and never rejects"""
    m = dict(model)
    m["thresholds"] = np.full(model["num_filters"], -3e38, np.float32)
    m["num_used"] = k + 1
    lv, f = oracle.Wvm(m).eval(patch_eq)
    assert lv == k
    return np.float32(f)


def _check_all_paths(oracle, capi, ctx, po, pg, model, tag):
    """production path (pre-filter + stage B) == exact path (per-window outputs) == oracle"""
    wo, wg = oracle.Wvm(model), capi.Wvm(ctx, model)
    pos_o, lv_o, fo_o = oracle.sliding_wvm(po, wo, 1, 1)
    pos_e, lv_e, fo_e = capi.detect_wvm(ctx, pg, wg, 1, 1, want_all=True)
    pos_p, _, _ = capi.detect_wvm(ctx, pg, wg, 1, 1, want_all=False)
    wg.close()
    assert np.array_equal(lv_e, lv_o), tag
    assert np.array_equal(fo_e, fo_o), tag
    for pos in (pos_e, pos_p):
        assert len(pos) == len(pos_o), (tag, len(pos), len(pos_o))
        for f in ("cx", "cy", "w", "h", "layer", "lx", "ly"):
            assert np.array_equal(pos[f], pos_o[f]), (tag, f)
        assert np.array_equal(pos["score"], pos_o["fout"]), tag
    return lv_o, fo_o


CASES = {
    # name: (pyramid, frame size, patch w, h, n_per, n_levels, make_wvm extras, brighten)
    "face20_L14": (FF, (320, 240), 20, 20, 14, 3, {}, False),
    "ear16x24_L16": (EAR, (200, 150), 16, 24, 20, 2, {}, False),
    "nose32x24_sxx_2p24": (EAR, (220, 160), 32, 24, 16, 2, {}, True),
    "face20_big_basis": (FF, (320, 240), 20, 20, 14, 2, dict(r=0.8), False),
    "face20_cancellation": (FF, (320, 240), 20, 20, 14, 2, dict(hk_scale=3000.0), False),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_production_cascade_at_exact_threshold_ties(oracle, capi, ctx, synth, name):
    """Thresholds of the pre-filter's levels are set to the exact fp32 filter output of a real window, and to the next float above
    and below it: the window must pass / pass / be rejected exactly like in the reference (`fout >= threshold`,
    WvmClassifier.cpp:139-141), in the production path, the exact path and the oracle alike."""
    kw, size, pw, ph, nper, nlev, extra, bright = CASES[name]
    frame = synth.make_frame(size[0], size[1], seed=77)
    if bright:   # large equalised values everywhere: sum of squares of a 32x24 patch around and above 2^24
        frame = np.clip(frame.astype(np.int32) // 3 + 170, 0, 255).astype(np.uint8)
    po, pg = _pyr_pair(oracle, capi, ctx, frame, **kw)
    gray = oracle.bgr2gray(frame)
    rng = np.random.default_rng(9)
    calib = synth.random_patches(gray[::2, ::2].copy(), pw, ph, 3000, rng)
    base = synth.make_wvm(41, fw=pw, fh=ph, n_per=nper, n_levels=nlev, calib_patches=calib, min_survivors=40, **extra)
    nwin = len(po.windows(pw, ph, 1, 1))
    assert nwin >= 512   # below that the pre-filter is not used
    if bright:
        eq = _eq_patches(oracle, po, pw, ph, range(0, nwin, max(1, nwin // 64)))
        assert max(int((e.astype(np.int64) ** 2).sum()) for e in eq) >= 1 << 24
    L = min(16, nper, base["num_used"] - 1)
    levels = sorted({0, L // 2, L - 1})
    # 32 windows per case, each at one of the three levels (rotating); every window's exact output as the threshold ("tie": the window
    # passes), and the next float above it ("above": the exact cascade rejects the window at this level while the pre-filter, whose
    # bound cannot tell a one-ulp difference, lets it through by its guard band only -- stage B has to do the rejecting); the first
    # windows also with the next float below.
    picks = rng.choice(nwin, 32, replace=False)
    patches = _eq_patches(oracle, po, pw, ph, picks)
    ran = 0
    for i, (wi, pe) in enumerate(zip(picks, patches)):
        k = levels[i % len(levels)]
        res = _level_output(oracle, base, k, pe)
        if not np.isfinite(res):
            continue
        variants = [("tie", res), ("above", np.nextafter(res, np.float32(np.inf)))]
        if i < 3:
            variants.append(("below", np.nextafter(res, np.float32(-np.inf))))
        for which, thr in variants:
            m = dict(base)
            t = base["thresholds"].copy()
            t[:k] = -3e38          # the window reaches level k
            t[k] = thr
            m["thresholds"] = t
            lv_o, fo_o = _check_all_paths(oracle, capi, ctx, po, pg, m, (name, k, int(wi), which))
            if which == "above":
                assert lv_o[wi] == k and fo_o[wi] == res
            else:
                assert lv_o[wi] > k
            ran += 1
    assert ran >= 60
    pg.close()


PROFILES = {
    # (ii) of VERDICT r02 task 3: 0.9 pass rate per filter down to ~0.1 % survivors; (iii): rejections only at level-group ends
    "late_reject_0.9": dict(pass_rate=0.9, min_survivors=12),
    "group_end_only": dict(pass_rate=0.5, min_survivors=12, reject_every=14),
}


@pytest.mark.parametrize("profile", sorted(PROFILES))
def test_rejection_profiles_production_equals_exact_equals_oracle(oracle, capi, ctx, synth, frame640, profile):
    """Models that keep rejecting deep into the cascade hand stage B a large part of the windows; its phases then thin the list out
    between generations.  Every window's (level, output) and every positive must still equal the oracle's."""
    gray = oracle.bgr2gray(frame640)
    calib = synth.random_patches(gray[::4, ::4].copy(), 20, 20, 12000, np.random.default_rng(1))
    model = synth.make_wvm(7, n_per=14, n_levels=8, calib_patches=calib, **PROFILES[profile])
    po, pg = _pyr_pair(oracle, capi, ctx, frame640, **FF)
    lv, _ = _check_all_paths(oracle, capi, ctx, po, pg, model, profile)
    wg = capi.Wvm(ctx, model)   # one handle, several runs: the phase plan settles, the positives stay
    runs = [capi.detect_wvm(ctx, pg, wg, 1, 1, want_all=False)[0].tobytes() for _ in range(4)]
    assert all(r == runs[0] for r in runs)
    wg.close()
    deep = int((lv >= 16).sum())
    assert deep > 0.05 * len(lv), "the profile should send a sizeable share of the windows to stage B (%d of %d)" % (deep, len(lv))
    if profile == "group_end_only":
        assert all(k % 14 == 13 for k in set(lv.tolist()))   # exits only at the last filter of a level group
    else:
        assert len(set(lv.tolist())) > 20   # exits spread over many levels
    pg.close()


def test_stage_b_queue_overflow_is_reported(oracle, capi, ctx, synth, frame640):
    """more queued windows than the stage-B state holds: an error (FD_ERR_CAPACITY), not a silently shorter result"""
    model = synth.make_wvm(3, n_per=14, n_levels=2)   # thresholds -1e30: every window runs all 28 filters
    _, pg = _pyr_pair(oracle, capi, ctx, frame640, **FF)
    wg = capi.Wvm(ctx, model)
    os.environ["FD_WVM_DEEP_CAP"] = "1000"
    try:
        with pytest.raises(capi.FdError) as e:
            capi.detect_wvm(ctx, pg, wg, 1, 1, want_all=False)
        assert "FD_WVM_DEEP_CAP" in str(e.value)
    finally:
        del os.environ["FD_WVM_DEEP_CAP"]
    pos, _, _ = capi.detect_wvm(ctx, pg, wg, 1, 1, want_all=False)   # the handle is usable afterwards
    assert len(pos) == 16185
    wg.close(); pg.close()


def test_stage_b_state_grows_with_the_queue(oracle, capi, ctx, synth, small_models):
    """A model whose first 27 filters reject nothing sends every window of a 24-frame call to stage B (388 k > the default state of
    2^18 windows): the library grows the state and runs stage B again -- same detections as frame-by-frame calls."""
    _, svm = small_models
    frames = [synth.make_frame(640, 480, seed=900 + i) for i in range(24)]
    gray = oracle.bgr2gray(frames[0])
    calib = synth.random_patches(gray[::4, ::4].copy(), 20, 20, 4000, np.random.default_rng(4))
    model = synth.make_wvm(19, n_per=14, n_levels=2, calib_patches=calib)
    po = oracle.Pyramid(**FF)
    po.update(frames[0])
    t = np.full(28, -3e38, np.float32)
    model["thresholds"] = t
    _, lv0, fo0 = oracle.sliding_wvm(po, oracle.Wvm(model), 1, 1)
    assert (lv0 == 27).all()
    t[27] = np.quantile(fo0, 0.99)   # ~160 WVM positives per frame, everything else leaves at the very last filter
    wg, sg = capi.Wvm(ctx, model), capi.Svm(ctx, svm)
    single = capi.Pyramid(ctx, **FF)
    ref = []
    for f in frames[:3] + frames[-2:]:
        single.update(f)
        ref.append(capi.detect_five_stage(ctx, single, wg, sg, cap=1024))
    multi = capi.Pyramid(ctx, **FF)
    multi.set_frames(len(frames))
    multi.update_frames(images=frames)
    for _ in range(2):   # the first call grows the state, the second finds it large enough
        res = capi.detect_five_stage_frames(ctx, multi, wg, sg, len(frames), cap=1024)
        got = res[:3] + res[-2:]
        assert sum(int(s[0]) for _, s in got) > 0
        for (d, s), (dr, sr) in zip(got, ref):
            assert np.array_equal(s, sr) and d.tobytes() == dr.tobytes()
    wg.close(); sg.close(); single.close(); multi.close()


@pytest.mark.parametrize("profile", ["group", "late"])
def test_long_queue_kernels_of_stage_b_against_short_queue_kernels_and_the_oracle(oracle, capi, ctx, synth, profile):
    """With hundreds of thousands of queued windows stage B switches kernels / forms -- k_wvb_prepare_lanes (from 32 K windows),
    k_wvb_chain2 with all class quarters of a tile in one unit and the level slots outside the phase skipped (from 1024 tiles, round
    5).  The bench's heavy-queue models (bench.cascade_models: 280 filters; 'group' rejects at the end of a level
    group only, 'late' keeps rejecting through ~65 filters) on 32 frames in ONE call queue 200-330 K windows; every frame's stage
    counts and detections must equal those of single-frame calls (short queues: the other kernels) and, for three frames, the
    oracle's."""
    import bench
    wvm_m, svm_m = bench.cascade_models(profile)
    frames, _ = synth.make_frames_varied(32, 640, 480, seed=4242, scene_len=8)
    wg, sg = capi.Wvm(ctx, wvm_m), capi.Svm(ctx, svm_m)
    multi = capi.Pyramid(ctx, **FF)
    multi.set_frames(len(frames))
    multi.update_frames(images=frames)
    res = None
    for _ in range(2):   # the first call may grow the state and settles the phase plan, the second runs on the settled plan
        res = capi.detect_five_stage_frames(ctx, multi, wg, sg, len(frames), cap=2048)
    assert wg.last_queue_length() > 150000, wg.last_queue_length()
    single = capi.Pyramid(ctx, **FF)
    w1 = capi.Wvm(ctx, wvm_m)   # its own handle: grids and phase plan of a short queue
    for fi, f in enumerate(frames):
        single.update(f)
        d, st = capi.detect_five_stage(ctx, single, w1, sg, cap=2048)
        assert np.array_equal(res[fi][1], st) and res[fi][0].tobytes() == d.tobytes(), fi
    po, wo, so = oracle.Pyramid(**FF), oracle.Wvm(wvm_m), oracle.Svm(svm_m)
    for fi in (0, 13, 31):
        po.update(frames[fi])
        do, sto = oracle.five_stage(po, wo, so)
        assert np.array_equal(res[fi][1], sto), (fi, res[fi][1], sto)
        for f in ("cx", "cy", "w", "h"):
            assert np.array_equal(res[fi][0][f], do[f])
    wg.close(); w1.close(); sg.close(); multi.close(); single.close()


def test_stage_b_dense_equals_rect_lookup_kernels(capi, ctx, synth, oracle, frame640):
    """FD_WVM_STAGEB=old keeps the rect-lookup stage-B kernel (k_wvm_deep): both must deliver the same positive records"""
    import bench
    wvm_m, _ = bench.cascade_models()
    _, pg = _pyr_pair(oracle, capi, ctx, frame640, **FF)
    res = {}
    for mode in ("new", "old"):
        if mode == "old":
            os.environ["FD_WVM_STAGEB"] = "old"
        try:
            wg = capi.Wvm(ctx, wvm_m)
        finally:
            os.environ.pop("FD_WVM_STAGEB", None)
        res[mode] = (capi.detect_wvm(ctx, pg, wg, 1, 1, want_all=False)[0], capi.detect_wvm(ctx, pg, wg, 1, 1, want_all=True))
        # the dense stage B adapts its phase plan and grids to what the previous runs of the handle saw: same result every time
        for _ in range(3):
            assert capi.detect_wvm(ctx, pg, wg, 1, 1, want_all=False)[0].tobytes() == res[mode][0].tobytes()
        wg.close()
    assert len(res["new"][0]) > 0
    assert res["new"][0].tobytes() == res["old"][0].tobytes()
    assert np.array_equal(res["new"][1][1], res["old"][1][1]) and np.array_equal(res["new"][1][2].view(np.uint32), res["old"][1][2].view(np.uint32))
    pg.close()


def test_headline_workload_against_the_oracle(oracle, capi, ctx, synth):
    """The bench's headline call, verbatim: bench.cascade_models() (280-filter FaceFrontal WVM + 1024-SV SVM), 64 frames of 640x480
    in one multi-frame pyramid, fd_detect_five_stage_frames_begin / _end.  Detections and stage counts of frames spread over the
    call are compared with oracle.five_stage."""
    import bench
    wvm_m, svm_m = bench.cascade_models()
    NF = 64
    frames = [synth.make_frame(640, 480, seed=20260927 + i % 8) for i in range(NF)]   # the bench cycles through 8 frames
    pm = capi.Pyramid(ctx, **FF)
    pm.set_frames(NF)
    pm.update_frames(images=frames)
    wg, sg = capi.Wvm(ctx, wvm_m), capi.Svm(ctx, svm_m)
    res = capi.FiveStageFrames(ctx, pm, wg, sg, NF).end()
    wo, so = oracle.Wvm(wvm_m), oracle.Svm(svm_m)
    po = oracle.Pyramid(**FF)
    total = 0
    for f in (0, 5, 18, 39, 63):
        po.update(frames[f])
        do, sto = oracle.five_stage(po, wo, so)
        dg, stg = res[f]
        assert np.array_equal(stg, sto), (f, stg, sto)
        assert len(dg) == len(do)
        for fld in ("cx", "cy", "w", "h"):
            assert np.array_equal(dg[fld], do[fld]), (f, fld)
        assert np.allclose(dg["probability"], do["prob"], rtol=1e-4, atol=0), f   # SVM distance tolerance of north_star
        total += len(dg)
    assert total > 0
    # frames with the same content give the same result wherever they sit in the call
    for f in range(8, NF):
        assert res[f][0].tobytes() == res[f % 8][0].tobytes() and np.array_equal(res[f][1], res[f % 8][1])
    wg.close(); sg.close(); pm.close()


def test_single_frame_scores_queued_behind_the_cascade(oracle, capi, ctx, synth, monkeypatch):
    """fd_detect_five_stage on one frame: the SVM scores of ALL WVM positives are queued behind the cascade and the host waits once
    (five_stage.hpp).  Byte-identical to the two-round-trip order (FD_FS_SPEC=0) on a sequence whose positive counts jump -- a blank
    frame makes the next launch cover 64 slots only, so the busy frame behind it must fall back -- and equal to the oracle.  The hook
    says which order produced each result."""
    import bench
    wvm_m, svm_m = bench.cascade_models()
    frames, _ = synth.make_frames_varied(6, 640, 480, seed=977, scene_len=2)
    blank = np.full((480, 640, 3), 128, np.uint8)
    seq = [frames[0], frames[0], blank, frames[3], frames[3], frames[5], blank, blank, frames[1]]
    pg, ph = capi.Pyramid(ctx, **FF), capi.Pyramid(ctx, **FF)
    wg, wh, sg = capi.Wvm(ctx, wvm_m), capi.Wvm(ctx, wvm_m), capi.Svm(ctx, svm_m)
    states, expect, prev = [], [], -1
    for i, fr in enumerate(seq):
        pg.update(fr)
        dg, stg = capi.detect_five_stage(ctx, pg, wg, sg)
        states.append(wg.last_spec_state())
        covered = 1024 if prev < 0 else max(64, 2 * prev + 64)   # the launch covers twice the previous frame's positives (five_stage.hpp)
        expect.append(0 if int(stg[0]) <= covered else 1)
        prev = int(stg[0])
        monkeypatch.setenv("FD_FS_SPEC", "0")
        ph.update(fr)
        dh, sth = capi.detect_five_stage(ctx, ph, wh, sg)
        assert wh.last_spec_state() == -1
        monkeypatch.delenv("FD_FS_SPEC")
        assert dg.tobytes() == dh.tobytes() and np.array_equal(stg, sth), i
        if i in (0, 3):
            po = oracle.Pyramid(**FF)
            po.update(fr)
            do, sto = oracle.five_stage(po, oracle.Wvm(wvm_m), oracle.Svm(svm_m), 5.0, 0.0, 1, 1, None)
            assert len(dg) == len(do) and list(stg) == list(sto), i
            for fld in ("cx", "cy", "w", "h"):
                assert np.array_equal(dg[fld], do[fld]), (i, fld)
        if i == 3:
            assert stg[0] > 64, "the busy frame behind the blank one must outgrow the 64-slot launch"
    assert states == expect and states[3] == 1 and states[8] == 1 and states.count(0) >= 5, (states, expect)
    for o_ in (wg, wh, sg, pg, ph):
        o_.close()


def test_device_overlap_elimination_equals_the_host_path(oracle, capi, ctx, synth, monkeypatch):
    """Stages 2-3 on the device (csrc/fs_tail.hpp: k_fs_oe + the counted SVM launch) against the host-driven tail (FD_FS_TAIL=0) and
    the oracle, on frames whose number of WVM positives differs ~7x (synth.make_frames_varied: the bench's headline content), through
    the multi-frame ticket API and the single-frame entry point.  The hook says which path produced the result."""
    import bench
    wvm_m, svm_m = bench.cascade_models()
    NF = 24
    frames, _ = synth.make_frames_varied(NF, 640, 480, seed=4242, scene_len=3)
    pm = capi.Pyramid(ctx, **FF)
    pm.set_frames(NF)
    pm.update_frames(images=frames)
    wg, sg = capi.Wvm(ctx, wvm_m), capi.Svm(ctx, svm_m)
    res = capi.FiveStageFrames(ctx, pm, wg, sg, NF).end()
    assert wg.last_tail_state() == 0, "the device tail should have produced this result"
    res2 = capi.FiveStageFrames(ctx, pm, wg, sg, NF).end()   # second run: the SVM launch is sized from the first
    monkeypatch.setenv("FD_FS_TAIL", "0")
    wh = capi.Wvm(ctx, wvm_m)
    ref = capi.FiveStageFrames(ctx, pm, wh, sg, NF).end()
    assert wh.last_tail_state() == -1
    monkeypatch.delenv("FD_FS_TAIL")
    npos = []
    for f in range(NF):
        assert res[f][0].tobytes() == ref[f][0].tobytes() and np.array_equal(res[f][1], ref[f][1]), f
        assert res2[f][0].tobytes() == ref[f][0].tobytes() and np.array_equal(res2[f][1], ref[f][1]), f
        npos.append(int(ref[f][1][0]))
    assert max(npos) >= 3 * max(1, min(npos)), npos   # the frames really differ
    wo, so = oracle.Wvm(wvm_m), oracle.Svm(svm_m)
    po = oracle.Pyramid(**FF)
    p1 = capi.Pyramid(ctx, **FF)
    monkeypatch.setenv("FD_FS_TAIL", "1")   # single-frame calls take the host tail by default (it is faster for ~150 positives)
    for f in (0, 7, 13, 23):
        po.update(frames[f])
        do, sto = oracle.five_stage(po, wo, so)
        dg, stg = res[f]
        assert np.array_equal(stg, sto), (f, stg, sto)
        for fld in ("cx", "cy", "w", "h"):
            assert np.array_equal(dg[fld], do[fld]), (f, fld)
        p1.update(frames[f])
        d1, st1 = capi.detect_five_stage(ctx, p1, wg, sg)   # single-frame entry point: one workgroup does the elimination
        assert wg.last_tail_state() == 0
        assert np.array_equal(st1, sto) and d1.tobytes() == dg.tobytes(), f
    for h in (wg, wh, sg, pm, p1):
        h.close()


@pytest.mark.parametrize("reps,expect", [((2, 3), 1), ((4, 5), 2)])
def test_device_overlap_elimination_gives_up_on_ties(oracle, capi, ctx, synth, monkeypatch, reps, expect):
    """The reference orders the positives with std::sort on the probability; what it does with equal keys is a property of that sort,
    which the device kernel cannot reproduce.  A frame of repeated tiles on a scale-1 pyramid gives windows with identical pixels, hence
    identical outputs: k_fs_oe must flag the frame (state 1) and the host's elimination must produce the oracle's detections.  The larger
    frame has more positives than the kernel holds per frame (state 2): same fallback."""
    monkeypatch.setenv("FD_FS_TAIL", "1")   # the device tail also for this single-frame call
    rng = np.random.default_rng(3)
    tile = synth.make_frame(48, 40, seed=11)
    frame = np.tile(tile, (reps[0], reps[1], 1))
    gray = oracle.bgr2gray(frame)
    calib = synth.random_patches(gray, 20, 20, 3000, rng)
    wvm_m = synth.make_wvm(31, n_per=6, n_levels=4, calib_patches=calib, min_survivors=200)
    eq = synth.histeq64_np(synth.random_patches(gray, 20, 20, 400, rng))
    svm_m = synth.make_svm_u8(32, eq, nsv=64, calib=eq[64:], positive_fraction=0.5)
    kw = dict(octave_layers=1, min_scale=1.0, max_scale=1.0)
    po, pg = _pyr_pair(oracle, capi, ctx, frame, **kw)
    wg, sg = capi.Wvm(ctx, wvm_m), capi.Svm(ctx, svm_m)
    dg, stg = capi.detect_five_stage(ctx, pg, wg, sg)
    state = wg.last_tail_state()
    do, sto = oracle.five_stage(po, oracle.Wvm(wvm_m), oracle.Svm(svm_m))
    assert sto[0] >= 8, "the test needs WVM positives"
    assert state == expect, (state, sto)
    assert np.array_equal(stg, sto), (stg, sto)
    for fld in ("cx", "cy", "w", "h"):
        assert np.array_equal(dg[fld], do[fld]), fld
    wg.close(); sg.close(); pg.close()


def test_detect_image_equals_update_then_detect(oracle, capi, ctx, synth, small_models):
    """fd_detect_five_stage_image (Detector::detect(const Mat&), Detector.hpp:59): the pyramid update and the detection in one call give
    what the two calls give, for host and device-resident images, frame after frame on the same handles."""
    import torch
    wvm, svm = small_models
    pg = capi.Pyramid(ctx, **FF)
    pg2 = capi.Pyramid(ctx, **FF)
    wg, sg = capi.Wvm(ctx, wvm), capi.Svm(ctx, svm)
    one = capi.FiveStageImage(ctx, pg, wg, sg)
    po = oracle.Pyramid(**FF)
    for seed in (5, 6, 7):
        frame = synth.make_frame(640, 480, seed=seed)
        d1, st1 = one.detect(frame)
        d1, st1 = d1.copy(), st1.copy()
        dev = torch.from_numpy(frame).cuda()
        d3, st3 = one.detect_device(dev.data_ptr(), 640, 480, 3)
        pg2.update(frame)
        d2, st2 = capi.detect_five_stage(ctx, pg2, wg, sg)
        assert np.array_equal(st1, st2) and d1.tobytes() == d2.tobytes()
        assert np.array_equal(st3, st2) and d3.tobytes() == d2.tobytes()
        po.update(frame)
        do, sto = oracle.five_stage(po, oracle.Wvm(wvm), oracle.Svm(svm))
        assert np.array_equal(st1, sto)
    for h in (wg, sg, pg, pg2):
        h.close()


@pytest.mark.parametrize("shape,nsv", [((20, 20), 1024), ((20, 20), 700), ((24, 32), 330), ((16, 24), 40)])
def test_u8_svm_kernel_with_8_and_16_wavefronts_gives_the_same_bits(capi, ctx, synth, shape, nsv):
    """a single frame's positives are scored by k_svm_u8_rbf_mfma<16> when the launch covers at most 2048 vectors and by <8>
    otherwise or on the two-round-trip path -- the same frame may see either, so the two must give identical sums (one partial per tile
    of 32 support vectors, added in a fixed pairwise order).  Compared directly on identical inputs: 32 / 22 / 11 / 2 support-vector
    tiles (not multiples of 16) and 13 / 24 / 12 k-steps (24 > 16: vectors of 768 bytes), vector counts that are not multiples of 32."""
    ph, pw = shape
    rng = np.random.default_rng(nsv)
    pool = rng.integers(0, 256, (nsv + 400, ph, pw), dtype=np.uint8)
    m = synth.make_svm_u8(17, pool, nsv=nsv, calib=pool[nsv:])
    s = capi.Svm(ctx, m)
    for n in (1, 33, 777, 2048):
        feats = rng.integers(0, 256, (n, ph * pw), dtype=np.uint8)
        o8, o16 = capi.svm_u8_both(ctx, s, feats)
        assert o8.tobytes() == o16.tobytes(), (shape, nsv, n)
        assert np.array_equal(o8, s.distance(feats))   # ... and the production launcher's result
    s.close()


def _big_tail_models(synth, oracle):
    """three detectors of different patch shapes with WVMs that leave thousands of positives per 960x540 frame"""
    gray = oracle.bgr2gray(synth.make_frame(640, 480, seed=20260927))
    out = []
    for di, name in enumerate(("LeftEyeCenter", "NoseTip", "CenterLipUpperOuter")):
        inc, mn, mx, pw, ph, nper, nlev = synth.DETECTOR_CFGS[name]
        src = gray[::2, ::2]
        calib = synth.random_patches(src.copy(), pw, ph, 6000, np.random.default_rng(100 + di))
        wm = synth.make_wvm(70 + di, fw=pw, fh=ph, n_per=nper, n_levels=min(nlev, 3), calib_patches=calib, min_survivors=60)
        eq = synth.histeq64_np(synth.random_patches(src.copy(), pw, ph, 456, np.random.default_rng(200 + di)))
        sm = synth.make_svm_u8(300 + di, eq, nsv=256, calib=eq[256:])
        out.append((name, (inc, mn, mx), wm, sm))
    return out


@pytest.mark.parametrize("content", ["plain", "tiled"])
def test_batch_device_tail_equals_the_host_tail(oracle, capi, ctx, synth, monkeypatch, content):
    """The jobs of a batch (fd_detect_five_stage_batch: config 3 / 5) run their overlap elimination on the device whatever the number of
    positives (fs_tail.hpp: k_fs_oe_big -- radix sort, painted map in device memory, 256 candidates per step).  Against the host stages
    (FD_FS_TAIL=0: std::sort + hostalgo.cpp) the detections must be the same BYTES, on frames with thousands of positives per job.
    "tiled": the frame is two identical halves, so every positive has a twin with the same fp32 output far away -- ties that do not
    overlap are harmless for the set of survivors and are handled on the device; a job whose SVM positives contain a tied pair, or whose
    tied elements overlap, is handed to the host stages (the hook says which path produced a job's result); either way the bytes agree.
    One job is also compared with the oracle."""
    frame = synth.make_frame(960, 540, seed=3)
    if content == "tiled":
        frame = np.ascontiguousarray(np.concatenate([frame[:, :480], frame[:, :480]], axis=1))
    models = _big_tail_models(synth, oracle)
    pyr = capi.Pyramid(ctx, inc=float(np.float32(0.9)), min_scale=float(np.float32(0.5)), max_scale=float(np.float32(0.7)))
    pyr.update(frame)
    handles = [(capi.Wvm(ctx, wm), capi.Svm(ctx, sm)) for _, _, wm, sm in models]
    jobs = [(pyr, w, s) for w, s in handles]
    monkeypatch.setenv("FD_FS_TAIL", "0")
    ref = capi.detect_five_stage_batch(ctx, jobs, cap=1 << 14)
    assert all(w.last_tail_state() == -1 for w, _ in handles)
    monkeypatch.setenv("FD_FS_TAIL", "1")
    got = capi.detect_five_stage_batch(ctx, jobs, cap=1 << 14)
    states = [w.last_tail_state() for w, _ in handles]
    got2 = capi.detect_five_stage_batch(ctx, jobs, cap=1 << 14)   # second run: the SVM launch is sized from the first
    for (name, *_), (dr, sr), (dg, sg), (dg2, sg2) in zip(models, ref, got, got2):
        assert sr[0] > 1024, (name, sr)   # more positives than k_fs_oe's LDS arrays hold
        assert np.array_equal(sg, sr) and dg.tobytes() == dr.tobytes(), (name, sg, sr)
        assert np.array_equal(sg2, sr) and dg2.tobytes() == dr.tobytes(), name
    assert all(st in (0, 1, 0x200) for st in states), states   # 0: the device's result; 1: overlapping ties; 0x200: tied SVM positives
    if content == "plain":
        assert 0 in states, states
    # the oracle on one job
    po = oracle.Pyramid(inc=float(np.float32(0.9)), min_scale=float(np.float32(0.5)), max_scale=float(np.float32(0.7)))
    po.update(frame)
    name, _, wm, sm = models[0]
    do, so = oracle.five_stage(po, oracle.Wvm(wm), oracle.Svm(sm), cap=1 << 14)
    assert np.array_equal(got[0][1], so), (got[0][1], so)
    for fld in ("cx", "cy", "w", "h"):
        assert np.array_equal(got[0][0][fld], do[fld]), fld
    for w, s in handles:
        w.close(); s.close()
    pyr.close()
