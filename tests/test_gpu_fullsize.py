"""GPU parity at BASELINE's own sizes (VERDICT r01 task 2): config 3 with all 15 detectors at their full filter counts on a
1920x1080 frame, config 4 at B = 256, config 2 with the oracle on whole 64-window MFMA tiles and on every positive.
Everything goes through the C ABI; the oracle is the checker."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
FF_ = dict(inc=float(np.float32(0.92)), min_scale=float(np.float32(0.05)), max_scale=float(np.float32(0.16)))  # FaceFrontal.cfg


def _models(synth, oracle, nsv):
    """the 15 ffpDetectApp detectors with their cfg-implied filter counts (same recipe as bench.py's ffp15 workload)"""
    gray = oracle.bgr2gray(synth.make_frame(640, 480, seed=20260927))
    out = []
    for di, (name, (inc, mn, mx, pw, ph, nper, nlev)) in enumerate(sorted(synth.DETECTOR_CFGS.items())):
        src = gray[::4, ::4] if mx < 0.3 else gray[::2, ::2]
        calib = synth.random_patches(src.copy(), pw, ph, 6000, np.random.default_rng(100 + di))
        wm = synth.make_wvm(50 + di, fw=pw, fh=ph, n_per=nper, n_levels=nlev, calib_patches=calib, min_survivors=24)
        eq = synth.histeq64_np(synth.random_patches(src.copy(), pw, ph, nsv + 200, np.random.default_rng(200 + di)))
        sm = synth.make_svm_u8(300 + di, eq, nsv=nsv, calib=eq[nsv:])
        out.append((name, (inc, mn, mx), wm, sm, pw, ph))
    return out


def _kw(key):
    return dict(inc=float(np.float32(key[0])), min_scale=float(np.float32(key[1])), max_scale=float(np.float32(key[2])))


def test_config3_all_detectors_full_size(oracle, capi, ctx, synth):
    """One 1080p frame, the 15 detectors (7..20 levels x 14..30 filters each: 98..280 filters) as ONE batch over 4 shared pyramids.
    * the batch is deterministic and equals 15 single fd_detect_five_stage calls; stage counts are non-increasing,
    * the three 20x20 face detectors AND one detector with a non-square patch on the large layers (LeftEyeCenter, 32x16, 2.7 M
      windows): the complete five-stage result equals the CPU oracle's on the full frame (stage counts, boxes, order, scores),
    * one detector of every patch shape (20x20, 32x16, 32x24, 16x24, 24x24): the WVM positives of the production path
      (dense pre-filter + exact cascade) equal {windows whose exact (level, fout) is positive}, and the exact path's (level,
      fout) equals the oracle on a strided sample AND on every WVM positive (bit-exact)."""
    models = _models(synth, oracle, nsv=1024)   # the bench's model size (1024 support vectors each)
    frame = synth.make_frame(1920, 1080, seed=20260927)
    pyrs, dets = {}, []
    for name, key, wm, sm, pw, ph in models:
        if key not in pyrs:
            pyrs[key] = capi.Pyramid(ctx, **_kw(key))
            pyrs[key].update(frame)
        dets.append((name, key, pyrs[key], capi.Wvm(ctx, wm), capi.Svm(ctx, sm), wm, sm, pw, ph))
    assert len(pyrs) == 4
    assert sum(p.window_count(pw, ph, 1, 1) for _, _, p, _, _, _, _, pw, ph in dets) == 32113402
    jobs = [(p, w, s) for _, _, p, w, s, _, _, _, _ in dets]
    r1 = capi.detect_five_stage_batch(ctx, jobs, cap=1 << 14)
    r2 = capi.detect_five_stage_batch(ctx, jobs, cap=1 << 14)
    npos_total = 0
    for (name, *_), (d1, s1), (d2, s2) in zip(dets, r1, r2):
        assert d1.tobytes() == d2.tobytes() and np.array_equal(s1, s2), name
        assert s1[0] >= s1[1] >= s1[2] >= s1[3] == len(d1), (name, s1)
        assert np.all((d1["cx"] >= 0) & (d1["cx"] < 1920) & (d1["cy"] >= 0) & (d1["cy"] < 1080)), name
        npos_total += int(s1[0])
    assert npos_total > 1000
    for i in (0, 3, 7, 14):   # batch == single call
        name, key, p, w, s = dets[i][:5]
        ds, ss = capi.detect_five_stage(ctx, p, w, s, cap=1 << 14)
        assert ds.tobytes() == r1[i][0].tobytes() and np.array_equal(ss, r1[i][1]), name

    # ---- the 20x20 face detectors and a 32x16 detector against the complete oracle cascade on the full frame
    opyr = {}
    for i, (name, key, p, w, s, wm, sm, pw, ph) in enumerate(dets):
        if not (name.startswith("Face") or name == "LeftEyeCenter"):
            continue
        if key not in opyr:
            opyr[key] = oracle.Pyramid(**_kw(key))
            opyr[key].update(frame)
        do, so = oracle.five_stage(opyr[key], oracle.Wvm(wm), oracle.Svm(sm), cap=1 << 14)
        dg, sg = r1[i]
        assert np.array_equal(sg, so), (name, sg, so)
        for f in ("cx", "cy", "w", "h", "layer", "lx", "ly", "level"):
            assert np.array_equal(dg[f], do[f]), (name, f)
        assert np.allclose(dg["score"], do["fout"], rtol=1e-4, atol=1e-6), name   # SVM distance (fp64 sum reordered)

    # ---- one detector per patch shape: production-path positives == exact path == oracle
    checked = 0
    for want in ("FaceFrontal", "LeftEyeCenter", "NoseTip", "LeftEarCenter", "LeftLipCorner"):
        i = [k for k, d_ in enumerate(dets) if d_[0] == want][0]
        name, key, p, w, s, wm, sm, pw, ph = dets[i]
        pos_fast, _, _ = capi.detect_wvm(ctx, p, w, 1, 1, want_all=False, cap=1 << 17)
        pos_all, lv, fo = capi.detect_wvm(ctx, p, w, 1, 1, want_all=True, cap=1 << 17)
        assert pos_fast.tobytes() == pos_all.tobytes(), name
        assert len(pos_fast) == r1[i][1][0], name
        F = wm["num_filters"]
        ispos = (lv == F - 1) & (fo >= wm["thresholds"][F - 1])
        wins = p.windows(pw, ph, 1, 1)
        pidx = np.nonzero(ispos)[0]
        assert len(pidx) == len(pos_all) and np.array_equal(wins[pidx][:, :3], np.stack([pos_all["layer"], pos_all["lx"], pos_all["ly"]], 1)), name
        if key not in opyr:
            opyr[key] = oracle.Pyramid(**_kw(key))
            opyr[key].update(frame)
        layers = [opyr[key].layer(k) for k in range(len(opyr[key].layers()))]
        wo = oracle.Wvm(wm)
        sample = np.unique(np.concatenate([np.arange(0, len(wins), max(1, len(wins) // 600)), pidx]))
        for k in sample:
            lp, lx, ly = wins[k][:3]
            l_, f_ = wo.eval(oracle.histeq64(np.ascontiguousarray(layers[lp][ly:ly + ph, lx:lx + pw])))
            assert (l_, np.float32(f_)) == (lv[k], fo[k]), (name, int(k))
        checked += len(sample)
    assert checked > 3000
    for d_ in dets:
        d_[3].close(); d_[4].close()
    for p in pyrs.values():
        p.close()


def test_grouped_prefilter_equals_separate_launches(oracle, capi, ctx, synth, monkeypatch):
    """Detectors of a batch that scan the same windows (one pyramid, one patch size: seven 24x24, two 16x24, two 20x20 of the
    fifteen) share ONE dense pre-filter launch (k_wvm_prefilter_group, wvm_dense_group.hpp).  The batch with groups equals the batch
    with FD_WVM_GROUP=0 (fifteen k_wvm_prefilter launches) byte for byte -- detections, order, scores, stage counts -- on two frames,
    and a group of two with different cascade depths (L = 16 and L = 8 dense levels) equals its members' single calls."""
    models = _models(synth, oracle, nsv=256)
    pyrs, dets = {}, []
    for name, key, wm, sm, pw, ph in models:
        if key not in pyrs:
            pyrs[key] = capi.Pyramid(ctx, **_kw(key))
        dets.append((name, pyrs[key], capi.Wvm(ctx, wm), capi.Svm(ctx, sm)))
    jobs = [(p, w, s) for _, p, w, s in dets]
    for seed, (W, H) in ((5, (960, 540)), (6, (1280, 720))):
        frame = synth.make_frame(W, H, seed=seed)
        for p in pyrs.values():
            p.update(frame)
        monkeypatch.setenv("FD_WVM_GROUP", "0")
        r0 = capi.detect_five_stage_batch(ctx, jobs, cap=1 << 14)
        monkeypatch.delenv("FD_WVM_GROUP")
        r1 = capi.detect_five_stage_batch(ctx, jobs, cap=1 << 14)
        r2 = capi.detect_five_stage_batch(ctx, jobs, cap=1 << 14)
        npos = 0
        for (name, *_), (d0, s0), (d1, s1), (d2, s2) in zip(dets, r0, r1, r2):
            assert np.array_equal(s0, s1) and np.array_equal(s1, s2), (name, s0, s1, s2)
            assert d0.tobytes() == d1.tobytes() == d2.tobytes(), name
            npos += int(s0[0])
        assert npos > 200
    # the ticket entry points run the per-detector host stages as tasks of the batch queue (three frames in flight) or, with
    # FD_BATCH_ASYNC=0, inside _end: the same bytes as the blocking call either way
    frames = [synth.make_frame(960, 540, seed=20 + k) for k in range(3)]
    sets = []
    for k in range(3):
        ps, js = {}, []
        for name, key, wm, sm, pw, ph in models:
            if key not in ps:
                ps[key] = capi.Pyramid(ctx, **_kw(key))
            js.append((ps[key], capi.Wvm(ctx, wm), capi.Svm(ctx, sm)))
        sets.append((ps, js))
    want = []
    for k in range(3):
        for p in pyrs.values():
            p.update(frames[k])
        want.append(capi.detect_five_stage_batch(ctx, jobs, cap=1 << 14))
    for mode in ("1", "0", "1"):
        monkeypatch.setenv("FD_BATCH_ASYNC", mode)
        tickets = []
        for k in range(3):
            for p in sets[k][0].values():
                p.update(frames[k])
            tickets.append(capi.FiveStageBatch(ctx, sets[k][1], cap=1 << 14))
        for k in range(3):
            got = tickets[k].end()
            for (name, *_), (dw, sw), (dg, sg) in zip(dets, want[k], got):
                assert np.array_equal(sw, sg) and dw.tobytes() == dg.tobytes(), (mode, k, name)
    monkeypatch.delenv("FD_BATCH_ASYNC")
    for ps, js in sets:
        for _, w_, s_ in js:
            w_.close(); s_.close()
        for p in ps.values():
            p.close()
    # two 24x24 models of different depth in one group
    gray = oracle.bgr2gray(synth.make_frame(640, 480, seed=20260927))
    calib = synth.random_patches(gray[::2, ::2].copy(), 24, 24, 4000, np.random.default_rng(9))
    wa = synth.make_wvm(71, fw=24, fh=24, n_per=20, n_levels=4, calib_patches=calib, min_survivors=24)
    wb = synth.make_wvm(72, fw=24, fh=24, n_per=8, n_levels=6, calib_patches=calib, min_survivors=24)
    eq = synth.histeq64_np(synth.random_patches(gray[::2, ::2].copy(), 24, 24, 456, np.random.default_rng(10)))
    sm = synth.make_svm_u8(73, eq, nsv=256, calib=eq[256:])
    p = capi.Pyramid(ctx, **_kw((0.9, 0.5, 0.7)))
    p.update(synth.make_frame(800, 600, seed=11))
    ha, hb, s1_, s2_ = capi.Wvm(ctx, wa), capi.Wvm(ctx, wb), capi.Svm(ctx, sm), capi.Svm(ctx, sm)
    rb = capi.detect_five_stage_batch(ctx, [(p, ha, s1_), (p, hb, s2_)], cap=1 << 14)
    for (w_, s_), (db, sb) in zip(((ha, s1_), (hb, s2_)), rb):
        ds, ss = capi.detect_five_stage(ctx, p, w_, s_, cap=1 << 14)
        assert ds.tobytes() == db.tobytes() and np.array_equal(ss, sb)
    assert int(rb[0][1][0]) + int(rb[1][1][0]) > 0
    for h in (ha, hb, s1_, s2_, p):
        h.close()
    for d_ in dets:
        d_[2].close(); d_[3].close()
    for p_ in pyrs.values():
        p_.close()


def test_abandoned_batch_ticket_is_ended_and_handles_stay_usable(capi, ctx, synth, small_models):
    """Between fd_five_stage_batch_begin and _end the library's queue threads write the jobs' outputs; a ticket that is dropped without
    end() is ended by the binding (the job arrays stay alive until then), and the same handles run the next batch with the same result."""
    wm, sm = small_models
    frame = synth.make_frame(640, 480, seed=31)
    p = capi.Pyramid(ctx, **FF_)
    p.update(frame)
    hs = [(capi.Wvm(ctx, wm), capi.Svm(ctx, sm)) for _ in range(3)]
    jobs = [(p, w, s) for w, s in hs]
    want = capi.detect_five_stage_batch(ctx, jobs, cap=4096)
    t = capi.FiveStageBatch(ctx, jobs, cap=4096)
    del t   # never ended by the caller
    import gc
    gc.collect()
    got = capi.FiveStageBatch(ctx, jobs, cap=4096).end()
    for (dw, sw), (dg, sg) in zip(want, got):
        assert np.array_equal(sw, sg) and dw.tobytes() == dg.tobytes()
    for w, s in hs:
        w.close(); s.close()
    p.close()


@pytest.mark.parametrize("step,roi", [(1, None), (2, None), (1, (200, 100, 700, 500)), (3, (-30, -20, 400, 300))])
def test_wvm_production_path_equals_exact_path(oracle, capi, ctx, synth, step, roi):
    """fd_detect_wvm without per-window outputs takes the production path (dense pre-filter on the matrix pipe, exact cascade on
    what it lets through); with them every window runs the exact cascade.  The positives must be byte-identical."""
    gray = oracle.bgr2gray(synth.make_frame(640, 480, seed=20260927))
    frame = synth.make_frame(960, 540, seed=77)
    for name in ("FaceFrontal", "FaceLeftProfile", "RightEyeCenter", "NoseTip", "RightEarCenter", "CenterLipUpperOuter"):
        inc, mn, mx, pw, ph, nper, nlev = synth.DETECTOR_CFGS[name]
        src = gray[::4, ::4] if mx < 0.3 else gray[::2, ::2]
        calib = synth.random_patches(src.copy(), pw, ph, 4000, np.random.default_rng(5))
        wm = synth.make_wvm(91, fw=pw, fh=ph, n_per=nper, n_levels=min(nlev, 4), calib_patches=calib, min_survivors=48)
        p = capi.Pyramid(ctx, **_kw((inc, mn, mx)))
        p.update(frame)
        w = capi.Wvm(ctx, wm)
        a, _, _ = capi.detect_wvm(ctx, p, w, step, step, roi=roi, want_all=False, cap=1 << 17)
        b, lv, fo = capi.detect_wvm(ctx, p, w, step, step, roi=roi, want_all=True, cap=1 << 17)
        assert len(lv) > 0 and len(b) > 0, name
        assert a.tobytes() == b.tobytes(), name
        w.close(); p.close()


def test_sdm_fit_batch_256(oracle, capi, ctx, synth):
    """BASELINE config 4 at its own batch size: 256 faces x 68 landmarks x 4 steps (fills whole 16-row f64 MFMA tiles of the regressor).
    All 256 shapes finite, identical faces give identical shapes, 16 faces spread over the batch within 1e-4 of the oracle."""
    model = synth.make_sdm(9, L=68, S=4)
    B = 256
    base = np.stack([synth.make_frame(256, 256, seed=100 + i, channels=1) for i in range(32)])
    imgs = np.concatenate([base] * (B // 32))
    boxes = np.array([[48, 48, 160, 160]] * B, np.int32)
    boxes[1::7] = [40, 56, 150, 170]
    sg = capi.Sdm(ctx, model)
    shapes, status = sg.fit(imgs, boxes)
    assert shapes.shape == (B, 136) and np.all(np.isfinite(shapes)) and np.all(status == 0)
    for i in range(0, B - 32 * 7, 1):   # same image + same box => same shape, wherever it sits in the batch
        j = i + 32 * 7
        if np.array_equal(boxes[i], boxes[j]):
            assert np.array_equal(shapes[i], shapes[j]), (i, j)
    worst = 0.0
    for i in list(range(0, B, 17)) + [B - 1]:
        st, ref = oracle.sdm_fit(imgs[i], model, boxes[i])
        assert st == 0
        assert np.allclose(shapes[i], ref, rtol=1e-4, atol=1e-4), (i, np.abs(shapes[i] - ref).max())
        worst = max(worst, float(np.max(np.abs(shapes[i] - ref) / np.maximum(np.abs(ref), 1.0))))
    print("sdm B=256: worst relative landmark error %.3e" % worst)
    sg.close()


def test_config2_full_size_tiles_and_positives(oracle, capi, ctx, synth):
    """Config 2 (640x480, 278,142 windows, HOG-324 + RBF-SVM 1024 SV) at full size: the oracle on the first and the last 64-window
    tile of the MFMA kernel (tile edges, the partially filled last tile) and on EVERY positive; per-score relative error reported
    beside the sum|coeff|-relative one (the fp64 sum over support vectors has cancellation: the bound that holds is relative to
    the terms, 1e-4 * sum|coeff|; the per-score figure is printed and asserted only where |score| is not tiny)."""
    kw = dict(octave_layers=5, min_scale=1 / 16, max_scale=1.0)
    pg = capi.Pyramid(ctx, **kw)
    pg.set_layer_filter(1, bins=9)
    pg.update(synth.make_frame(640, 480, seed=32))
    hp = capi.hog_params(20, 20, 2, 2, 9, 5, 2, False)
    feats2 = capi.extract_hog(ctx, pg, hp)
    m = synth.make_svm_f32(6, feats2, nsv=1024, gamma=0.5, positive_fraction=0.01)
    del feats2
    sg = capi.Svm(ctx, m)
    frame = synth.make_frame(640, 480, seed=31)
    pg.update(frame)
    dets, dist = capi.detect_hog_svm(ctx, pg, sg, hp)
    N = 278142
    assert len(dist) == N
    # the asynchronous product entry points deliver the same detections
    run = capi.HogSvmRun(ctx, pg, sg, hp, cap=1 << 14)
    assert run.end().tobytes() == dets.tobytes()
    pos = np.nonzero(dist >= float(np.float32(m["threshold"])))[0]
    assert len(pos) == len(dets) > 100
    po = oracle.Pyramid(**kw)
    po.set_layer_filter(1, bins=9)
    po.update(frame)
    layers = [po.layer(i) for i in range(len(po.layers()))]
    wins = pg.windows(20, 20, 2, 2)
    so = oracle.Svm(m)
    idx = np.unique(np.concatenate([np.arange(0, 64), np.arange(N - (N % 64 or 64) - 64, N), pos]))
    fo = np.stack([oracle.hog_filter(np.ascontiguousarray(layers[wins[i][0]][wins[i][2]:wins[i][2] + 20, wins[i][1]:wins[i][1] + 20]), 9, 5, 2)
                   for i in idx])
    do = so.distance(fo)
    scale = float(np.abs(m["coeff"]).sum())
    err = np.abs(dist[idx] - do)
    assert err.max() <= 1e-4 * scale
    big = np.abs(do) > 1e-3 * scale
    rel = err[big] / np.abs(do[big])
    print("config 2 full size: %d windows vs oracle, max |err| %.3e = %.2e of sum|coeff|; per-score relative error max %.3e (median %.1e) over %d scores"
          % (len(idx), err.max(), err.max() / scale, rel.max(), np.median(rel), big.sum()))
    assert rel.max() <= 1e-4
    # positives: same set as the oracle's decision on these windows (no score within the error band of the threshold flips silently)
    thr = float(np.float32(m["threshold"]))
    flips = (do >= thr) != (dist[idx] >= thr)
    assert np.all(np.abs(do[flips] - thr) <= 1e-4 * scale)
    sg.close(); pg.close(); po.close()


def test_batch_on_pool_streams_waits_for_the_pyramid_update(oracle, capi, ctx, synth):
    """The batch entry points launch the cascades on a pool of streams while fd_pyramid_update runs on the context's stream: the
    cascades must wait for the update they read (event recorded by the update).  Two 1080p frames alternate through
    update_device + fd_five_stage_batch_begin/_end without any host synchronisation in between; every repetition must return
    the detections of a fully synchronous run."""
    import torch
    models = [m for m in _models(synth, oracle, nsv=128) if m[0] in ("FaceLeftProfile", "LeftEarCenter", "LeftEyeOuterCorner", "NoseTip", "LeftLipCorner")]
    frames = [synth.make_frame(1920, 1080, seed=s) for s in (11, 12)]
    dfr = [torch.from_numpy(f).cuda() for f in frames]
    torch.cuda.synchronize()
    pyrs, dets = {}, []
    for name, key, wm, sm, pw, ph in models:
        if key not in pyrs:
            pyrs[key] = capi.Pyramid(ctx, **_kw(key))
        dets.append((pyrs[key], capi.Wvm(ctx, wm), capi.Svm(ctx, sm)))
    ref = []
    for f in frames:   # synchronous reference: host update, context synchronised, single calls
        for p in pyrs.values():
            p.update(f)
        ctx.synchronize()
        ref.append([capi.detect_five_stage(ctx, p, w, s, cap=1 << 13) for p, w, s in dets])
    assert sum(len(d) for d, _ in ref[0]) > 10
    for it in range(8):
        k = it % 2
        for p in pyrs.values():
            p.update_device(dfr[k].data_ptr(), 1920, 1080, 3)
        res = capi.FiveStageBatch(ctx, dets, cap=1 << 13).end()
        for (d, s), (dr, sr) in zip(res, ref[k]):
            assert np.array_equal(s, sr), (it, s, sr)
            assert d.tobytes() == dr.tobytes(), it
    for p, w, s in dets:
        w.close(); s.close()
    for p in pyrs.values():
        p.close()


@pytest.mark.parametrize("nframes,size", [(5, (640, 480)), (16, (320, 240)), (3, (960, 540)), (64, (320, 240)), (11, (321, 243))])
def test_multi_frame_pyramid_equals_single_frames(oracle, capi, ctx, synth, small_models, nframes, size):
    """fd_pyramid_set_frames / fd_pyramid_update_frames / fd_detect_five_stage_frames: n frames in one pyramid, one launch per pyramid
    stage, ONE cascade run and ONE SVM launch per call.  Every frame's layers and detections (boxes, order, scores, stage counts)
    equal the single-frame entry points' -- and through them the oracle's (frame 0 is checked against the oracle directly)."""
    wvm, svm = small_models
    W, H = size
    frames = [synth.make_frame(W, H, seed=500 + i) for i in range(nframes)]
    wg, sg = capi.Wvm(ctx, wvm), capi.Svm(ctx, svm)
    single = capi.Pyramid(ctx, **FF_)
    ref = []
    layers0 = None
    for i, f in enumerate(frames):
        single.update(f)
        ref.append(capi.detect_five_stage(ctx, single, wg, sg, cap=256))
        if i == nframes - 1:
            layers0 = [single.layer(k) for k in range(len(single.layers()))]
    multi = capi.Pyramid(ctx, **FF_)
    multi.set_frames(nframes)
    multi.update_frames(images=frames)
    assert multi.layers() == single.layers()
    for k in range(len(layers0)):
        assert np.array_equal(multi.frame_layer(nframes - 1, k), layers0[k]), k
    res = capi.detect_five_stage_frames(ctx, multi, wg, sg, nframes, cap=256)
    assert sum(len(d) for d, _ in res) > 0
    for f, ((d, s), (dr, sr)) in enumerate(zip(res, ref)):
        assert np.array_equal(s, sr), (f, s, sr)
        assert d.tobytes() == dr.tobytes(), f
    # gray frames resident in HBM take the same path
    import torch
    gray = [oracle.bgr2gray(f) for f in frames]
    dg = [torch.from_numpy(g).cuda() for g in gray]
    multi.update_frames(device_ptrs=[t.data_ptr() for t in dg], w=W, h=H, ch=1)
    res2 = capi.detect_five_stage_frames(ctx, multi, wg, sg, nframes, cap=256)
    for (d, s), (dr, sr) in zip(res2, ref):
        assert d.tobytes() == dr.tobytes() and np.array_equal(s, sr)
    # frame 0 against the oracle
    po = oracle.Pyramid(**FF_)
    po.update(frames[0])
    do, so = oracle.five_stage(po, oracle.Wvm(wvm), oracle.Svm(svm))
    assert np.array_equal(res[0][1], so)
    for fld in ("cx", "cy", "w", "h"):
        assert np.array_equal(res[0][0][fld], do[fld])
    # the single-frame entry points refuse a multi-frame pyramid
    with pytest.raises(capi.FdError):
        capi.detect_five_stage(ctx, multi, wg, sg)
    wg.close(); sg.close(); single.close(); multi.close()


def test_multi_frame_calls_in_flight_on_two_contexts(capi, ctx, synth, small_models):
    """fd_detect_five_stage_frames_begin / _end with two calls in flight (each on its own context, pyramid and handles): the
    pipelined results equal the synchronous ones."""
    wvm, svm = small_models
    NF = 6
    sets = []
    for k in range(2):
        c_ = ctx if k == 0 else capi.Context(0)
        p_ = capi.Pyramid(c_, **FF_)
        p_.set_frames(NF)
        sets.append((c_, p_, capi.Wvm(c_, wvm), capi.Svm(c_, svm)))
    batches = [[synth.make_frame(320, 240, seed=900 + 10 * b + i) for i in range(NF)] for b in range(4)]
    ref = []
    for b, frames in enumerate(batches):
        c_, p_, w_, s_ = sets[0]
        p_.update_frames(images=frames)
        ref.append(capi.detect_five_stage_frames(c_, p_, w_, s_, NF))
    runs, got = [None, None], [None] * 4
    for b, frames in enumerate(batches):
        k = b % 2
        c_, p_, w_, s_ = sets[k]
        if runs[k] is not None:
            got[runs[k][0]] = runs[k][1].end()
        p_.update_frames(images=frames)
        runs[k] = (b, capi.FiveStageFrames(c_, p_, w_, s_, NF))
    for r in runs:
        got[r[0]] = r[1].end()
    for b in range(4):
        for (d, s), (dr, sr) in zip(got[b], ref[b]):
            assert d.tobytes() == dr.tobytes() and np.array_equal(s, sr), b
    for c_, p_, w_, s_ in sets:
        w_.close(); s_.close(); p_.close()
    sets[1][0].close()


def test_multi_frame_ticket_reports_errors_and_stays_usable(capi, ctx, synth, small_models):
    """The host stages of fd_detect_five_stage_frames_begin run on the library's queue threads: what they fail with (here: more
    detections than the caller's capacity) is reported by _end, the handles stay usable, and end_flat() returns the same
    detections as end()."""
    wvm, svm = small_models
    NF = 4
    frames = [synth.make_frame(320, 240, seed=1300 + i) for i in range(NF)]
    p_ = capi.Pyramid(ctx, **FF_)
    p_.set_frames(NF)
    w_, s_ = capi.Wvm(ctx, wvm), capi.Svm(ctx, svm)
    p_.update_frames(images=frames)
    ref = capi.detect_five_stage_frames(ctx, p_, w_, s_, NF, cap=256)
    most = max(len(d) for d, _ in ref)
    assert most >= 2
    run = capi.FiveStageFrames(ctx, p_, w_, s_, NF, cap=most - 1)
    with pytest.raises(capi.FdError):
        run.end()
    p_.update_frames(images=frames)
    dets, fidx, stages = capi.FiveStageFrames(ctx, p_, w_, s_, NF, cap=256).end_flat()
    assert len(dets) == sum(len(d) for d, _ in ref) and np.array_equal(np.bincount(fidx, minlength=NF), [len(d) for d, _ in ref])
    assert dets.tobytes() == b"".join(d.tobytes() for d, _ in ref)
    assert np.array_equal(stages, np.stack([s for _, s in ref]))
    w_.close(); s_.close(); p_.close()


def test_config5_image_against_the_oracle(oracle, capi, ctx, synth):
    """BASELINE config 5's content: images generated on the device from (seed, image index) (bench.device_frames, what `bench.py
    --workload config5` shards over the ranks).  Image 7 of the job through the 15 detectors as one batch (the job's per-image call);
    the three 20x20 face detectors and one 24x24 detector (2.7 M windows, the busiest kind on this content) are compared with the
    complete oracle cascade on the full 1920x1080 frame: stage counts, boxes, order, scores."""
    import torch
    import bench
    dev = torch.device("cuda", 0)
    img = bench.device_frames(np.array([7]), 1920, 1080, dev, seed0=20260928)[0]   # config 5's seed
    frame = img.cpu().numpy()
    models = _models(synth, oracle, nsv=1024)
    pyrs, dets = {}, []
    for name, key, wm, sm, pw, ph in models:
        if key not in pyrs:
            pyrs[key] = capi.Pyramid(ctx, **_kw(key))
            pyrs[key].update_device(img.data_ptr(), 1920, 1080, 3)
        dets.append((name, key, pyrs[key], capi.Wvm(ctx, wm), capi.Svm(ctx, sm), wm, sm))
    res = capi.detect_five_stage_batch(ctx, [(p, w, s) for _, _, p, w, s, _, _ in dets], cap=1 << 14)
    assert sum(int(st[0]) for _, st in res) > 10000   # busy content: tens of thousands of WVM positives over the 15 detectors
    opyr = {}
    checked = 0
    for i, (name, key, p, w, s, wm, sm) in enumerate(dets):
        if not (name.startswith("Face") or name == "LeftLipCorner"):
            continue
        if key not in opyr:
            opyr[key] = oracle.Pyramid(**_kw(key))
            opyr[key].update(frame)
        do, so = oracle.five_stage(opyr[key], oracle.Wvm(wm), oracle.Svm(sm), cap=1 << 14)
        dg, sg = res[i]
        assert np.array_equal(sg, so), (name, sg, so)
        for f in ("cx", "cy", "w", "h", "layer", "lx", "ly", "level"):
            assert np.array_equal(dg[f], do[f]), (name, f)
        assert np.allclose(dg["score"], do["fout"], rtol=1e-4, atol=1e-6), name
        checked += 1
    assert checked == 4
    for d_ in dets:
        d_[3].close(); d_[4].close()
    for p in pyrs.values():
        p.close()
