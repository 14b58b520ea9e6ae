#!/usr/bin/env python
"""Seeded differential fuzzing of the HIP path (through the C ABI) against the CPU oracle on randomised geometry:
frame sizes, pyramid parameters, patch sizes, strides, ROIs, model shapes, histogram / FHOG parameters.
Test infrastructure (uses oracle/); run on the GPU box:  python tools/fuzz_parity.py --cases 60 --seed 1
Prints one line per case and a summary; exit code 1 on any mismatch."""
import argparse
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from featuredetection_amd import capi, synth  # noqa: E402
from oracle import pyoracle as O  # noqa: E402


def rand_pyr_kw(rng, small=False):
    if rng.random() < 0.5:
        inc = float(np.float32(rng.choice([0.92, 0.9, 0.85, 0.8])))
        mn = float(np.float32(rng.uniform(0.08, 0.3)))
        mx = float(np.float32(min(1.0, mn * rng.uniform(1.2, 3.0))))
        return dict(inc=inc, min_scale=mn, max_scale=mx)
    octl = int(rng.integers(1, 6))
    mn = float(rng.uniform(0.1, 0.4))
    return dict(octave_layers=octl, min_scale=mn, max_scale=float(min(1.0, mn * rng.uniform(1.5, 3.5))))


def rand_frame(rng, channels=3):
    w, h = int(rng.integers(97, 420)), int(rng.integers(81, 330))
    f = synth.make_frame(w, h, seed=int(rng.integers(1 << 30)))
    if channels == 1:
        return O.bgr2gray(f)
    return f


def rand_roi(rng, w, h):
    if rng.random() < 0.4:
        return None
    x, y = int(rng.integers(-30, w - 10)), int(rng.integers(-30, h - 10))
    return (x, y, int(rng.integers(30, w + 40)), int(rng.integers(30, h + 40)))


def same_geometry(g, o):
    for f in ("cx", "cy", "w", "h", "layer", "lx", "ly"):
        if not np.array_equal(g[f], o[f]):
            return "geometry field %s differs" % f
    return None


def case_pyramid(rng, ctx):
    frame = rand_frame(rng, channels=int(rng.choice([1, 3])))
    kw = rand_pyr_kw(rng)
    po = O.Pyramid(**kw); po.update(frame)
    pg = capi.Pyramid(ctx, **kw); pg.update(frame)
    try:
        lo, lg = po.layers(), pg.layers()
        if lo != lg:
            return "layer tables differ: %s vs %s" % (lo, lg)
        STATS['layers'] += len(lo)
        for i in range(len(lo)):
            if not np.array_equal(pg.layer(i), po.layer(i)):
                return "layer %d pixels differ (%s, frame %s)" % (i, kw, frame.shape)
        pw, ph = int(rng.integers(5, 33)), int(rng.integers(5, 33))
        sx, sy = int(rng.integers(1, 5)), int(rng.integers(1, 5))
        roi = rand_roi(rng, frame.shape[1], frame.shape[0])
        if not np.array_equal(pg.windows(pw, ph, sx, sy, roi), po.windows(pw, ph, sx, sy, roi)):
            return "window enumeration differs (%dx%d step %d,%d roi %s)" % (pw, ph, sx, sy, roi)
    finally:
        pg.close(); po.close()
    return None


STATS = dict(layers=0, windows=0, detections=0, features=0, cells=0, candidates=0)

SIZES = [(20, 20), (24, 24), (16, 24), (32, 16), (32, 24), (19, 21), (7, 5), (12, 30), (31, 9)]


def case_cascade(rng, ctx):
    frame = rand_frame(rng)
    gray = O.bgr2gray(frame)
    pw, ph = SIZES[int(rng.integers(len(SIZES)))]
    kw = rand_pyr_kw(rng)
    nper = int(rng.choice([1, 3, 6, 10, 20, 33]))
    nlev = int(rng.integers(1, 5)) if nper < 20 else int(rng.integers(1, 3))
    src = np.ascontiguousarray(gray[::2, ::2])
    if src.shape[0] <= ph + 2 or src.shape[1] <= pw + 2:
        src = gray
    calib = synth.random_patches(src, pw, ph, 1500, rng)
    wvm = synth.make_wvm(int(rng.integers(1 << 20)), fw=pw, fh=ph, n_per=nper, n_levels=nlev, calib_patches=calib,
                         min_survivors=int(rng.integers(8, 64)))
    eq = synth.histeq64_np(synth.random_patches(src, pw, ph, 260, rng))
    svm = synth.make_svm_u8(int(rng.integers(1 << 20)), eq, nsv=int(rng.choice([17, 64, 100])), calib=eq[100:], positive_fraction=0.4)
    po = O.Pyramid(**kw); po.update(frame)
    pg = capi.Pyramid(ctx, **kw); pg.update(frame)
    wo, so = O.Wvm(wvm), O.Svm(svm)
    wg, sg = capi.Wvm(ctx, wvm), capi.Svm(ctx, svm)
    desc = "%dx%d nper %d lev %d pyr %s frame %s" % (pw, ph, nper, nlev, kw, frame.shape)
    try:
        sx, sy = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        pos_o, lv_o, fo_o = O.sliding_wvm(po, wo, sx, sy)
        pos_g, lv_g, fo_g = capi.detect_wvm(ctx, pg, wg, sx, sy, want_all=True)
        if len(lv_o) != len(lv_g):
            return "window count %d vs %d (%s)" % (len(lv_g), len(lv_o), desc)
        if len(lv_o) == 0:
            return None
        STATS['windows'] += len(lv_o)
        if not np.array_equal(lv_g, lv_o):
            return "cascade levels differ at %d windows (%s)" % (int((lv_g != lv_o).sum()), desc)
        if not np.array_equal(fo_g, fo_o):
            return "fp32 filter outputs differ at %d windows (%s)" % (int((fo_g != fo_o).sum()), desc)
        e = same_geometry(pos_g, pos_o)
        if e:
            return e + " (" + desc + ")"
        roi = rand_roi(rng, frame.shape[1], frame.shape[0])
        dist, ratio = (5.0, 0.0) if rng.random() < 0.5 else (float(rng.uniform(0.2, 0.9)), float(rng.uniform(0.3, 0.9)))
        do, sto = O.five_stage(po, wo, so, dist, ratio, sx, sy, roi)
        dg, stg = capi.detect_five_stage(ctx, pg, wg, sg, dist, ratio, sx, sy, roi)
        STATS['detections'] += len(do)
        if not np.array_equal(stg, sto):
            return "stage counts %s vs %s (roi %s, %s)" % (stg, sto, roi, desc)
        e = same_geometry(dg, do)
        if e:
            return "five-stage " + e + " (" + desc + ")"
        if not np.array_equal(dg["probability"], do["prob"]):
            # SVM distances agree to 1e-4 relative; probabilities follow
            if not np.allclose(dg["probability"], do["prob"], rtol=1e-4, atol=1e-7):
                return "five-stage probabilities differ (%s)" % desc
    finally:
        wg.close(); sg.close(); pg.close(); po.close()
    return None


def case_frames(rng, ctx):
    """multi-frame pyramids: fd_pyramid_update_frames + fd_detect_five_stage_frames (ticket entry points, host stages on the queue
    threads) against the oracle's per-frame five-stage detector; every layer of one random frame against the oracle's pyramid"""
    w, h = int(rng.integers(97, 420)), int(rng.integers(81, 330))
    nf = int(rng.integers(2, 12))
    frames = [synth.make_frame(w, h, seed=int(rng.integers(1 << 30))) for _ in range(nf)]
    gray = O.bgr2gray(frames[0])
    pw, ph = SIZES[int(rng.integers(len(SIZES)))]
    kw = rand_pyr_kw(rng)
    nper = int(rng.choice([3, 6, 10, 20]))
    nlev = int(rng.integers(1, 4))
    src = np.ascontiguousarray(gray[::2, ::2])
    if src.shape[0] <= ph + 2 or src.shape[1] <= pw + 2:
        src = gray
    calib = synth.random_patches(src, pw, ph, 1500, rng)
    wvm = synth.make_wvm(int(rng.integers(1 << 20)), fw=pw, fh=ph, n_per=nper, n_levels=nlev, calib_patches=calib,
                         min_survivors=int(rng.integers(8, 64)))
    eq = synth.histeq64_np(synth.random_patches(src, pw, ph, 260, rng))
    svm = synth.make_svm_u8(int(rng.integers(1 << 20)), eq, nsv=int(rng.choice([17, 64, 100])), calib=eq[100:], positive_fraction=0.4)
    po = O.Pyramid(**kw)
    pg = capi.Pyramid(ctx, **kw)
    wo, so = O.Wvm(wvm), O.Svm(svm)
    wg, sg = capi.Wvm(ctx, wvm), capi.Svm(ctx, svm)
    desc = "%d frames %dx%d, patch %dx%d nper %d lev %d pyr %s" % (nf, w, h, pw, ph, nper, nlev, kw)
    try:
        pg.set_frames(nf)
        pg.update_frames(images=frames)
        sx, sy = int(rng.integers(1, 3)), int(rng.integers(1, 3))
        dist, ratio = (5.0, 0.0) if rng.random() < 0.5 else (float(rng.uniform(0.2, 0.9)), float(rng.uniform(0.3, 0.9)))
        res = capi.FiveStageFrames(ctx, pg, wg, sg, nf, oe_dist=dist, oe_ratio=ratio, sx=sx, sy=sy, cap=4096).end()
        fchk = int(rng.integers(nf))
        for f in range(nf):
            po.update(frames[f])
            if f == fchk:
                if po.layers() != pg.layers():
                    return "layer tables differ (%s)" % desc
                for li in range(len(po.layers())):
                    STATS['layers'] += 1
                    if not np.array_equal(pg.frame_layer(f, li), po.layer(li)):
                        return "layer %d of frame %d differs (%s)" % (li, f, desc)
            do, sto = O.five_stage(po, wo, so, dist, ratio, sx, sy, None)
            dg, stg = res[f]
            STATS['detections'] += len(do)
            if not np.array_equal(stg, sto):
                return "frame %d: stage counts %s vs %s (%s)" % (f, stg, sto, desc)
            e = same_geometry(dg, do)
            if e:
                return "frame %d: %s (%s)" % (f, e, desc)
    finally:
        wg.close(); sg.close(); pg.close(); po.close()
    return None


def case_hist(rng, ctx):
    frame = rand_frame(rng)
    kw = rand_pyr_kw(rng)
    kind = int(rng.integers(0, 4))
    lbp = rng.random() < 0.35 and kind in (1, 3)
    interp = bool(rng.random() < 0.5)
    if lbp:
        lt = int(rng.integers(0, 4))
        bins = {0: 256, 1: 59, 2: 16, 3: 16}[lt]
        lf = dict(kind=2, lbp_type=lt)
        sau = False
    else:
        bins = int(rng.integers(4, 13))
        sau = bool(rng.random() < 0.3) and kind in (0, 2)
        if sau:
            bins = 2 * (bins // 2) or 2
        lf = dict(kind=1, bins=bins, signed_gradients=sau, interpolate=bool(rng.random() < 0.5))
    if kind in (0, 1):
        cell, cell_h = int(rng.integers(3, 9)), int(rng.integers(3, 9))
        block, block_h = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        nx, ny = int(rng.integers(block, block + 3)), int(rng.integers(block_h, block_h + 3))
        pw, ph = cell * nx, cell_h * ny
        hpk = dict(kind=kind, bins=bins, cell=cell, cell_h=cell_h, block=block, block_h=block_h, interpolate=interp)
        okw = dict(bins=bins, cell=cell, cell_h=cell_h, block=block, block_h=block_h, interpolate=interp)
        if kind == 0:
            hpk["signed_and_unsigned"] = sau; okw["signed_and_unsigned"] = sau
            fname = "hog_filter"
        else:
            norm = int(rng.integers(0, 5))
            conc = bool(rng.random() < 0.5)
            hpk.update(normalization=norm, concatenate=conc); okw.update(normalization=norm, concatenate=conc)
            fname = "spatial_histogram"
    else:
        levels = int(rng.integers(1, 4))
        pw = ph = int(rng.choice([16, 20, 24, 32]))
        hpk = dict(kind=kind, bins=bins, levels=levels, interpolate=interp)
        okw = dict(bins=bins, levels=levels, interpolate=interp)
        if kind == 2:
            hpk["signed_and_unsigned"] = sau; okw["signed_and_unsigned"] = sau
            fname = "pyramid_hog"
        else:
            norm = int(rng.integers(0, 5))
            hpk["normalization"] = norm; okw["normalization"] = norm
            fname = "spatial_pyramid_histogram"
    if pw > 64 or ph > 64:
        return None
    desc = "%s %s layer %s patch %dx%d pyr %s" % (fname, hpk, lf, pw, ph, kw)
    po = O.Pyramid(**kw); po.set_layer_filter(**lf); po.update(frame)
    pg = capi.Pyramid(ctx, **kw); pg.set_layer_filter(**lf); pg.update(frame)
    try:
        sx, sy = int(rng.integers(2, 6)), int(rng.integers(2, 6))
        try:
            hp = capi.hist_params(pw=pw, ph=ph, sx=sx, sy=sy, **hpk)
            fg = capi.extract_hist(ctx, pg, hp)
        except capi.FdError as e:   # unsupported / invalid combination: the oracle must reject it too, or it is a limit of the backend
            return "skip:" + str(e)[:80]
        layers = [po.layer(i) for i in range(len(po.layers()))]
        wins = po.windows(pw, ph, sx, sy)
        if len(wins) == 0:
            return None if len(fg) == 0 else "features for no windows"
        if len(fg) == 0:
            return "no features for %d windows (%s)" % (len(wins), desc)
        fo = np.stack([getattr(O, fname)(np.ascontiguousarray(layers[lp][ly:ly + ph, lx:lx + pw]), **okw) for lp, lx, ly, *_ in wins])
        STATS['features'] += int(fo.size)
        if fg.shape != fo.shape:
            return "shape %s vs %s (%s)" % (fg.shape, fo.shape, desc)
        if not np.array_equal(fg, fo):
            if not np.allclose(fg, fo, rtol=1e-6, atol=1e-9):
                return "features differ, max abs %g (%s)" % (float(np.abs(fg - fo).max()), desc)
            return "note:tolerance-equal only"
    finally:
        pg.close(); po.close()
    return None


def case_fhog(rng, ctx):
    gray = rand_frame(rng, channels=int(rng.choice([1, 3])))   # CV_8UC1 or CV_8UC3
    cell = int(rng.integers(2, 11))
    ub = int(rng.integers(2, 19))
    ib, ic = bool(rng.random() < 0.5), bool(rng.random() < 0.5)
    alpha = float(rng.choice([0.2, 0.1, 0.5]))
    fo = O.fhog(gray, cell, ub, ib, ic, alpha)
    fg = capi.fhog(ctx, gray, cell_size=cell, unsigned_bins=ub, interpolate_bins=ib, interpolate_cells=ic, alpha=alpha)
    STATS['cells'] += int(fo.shape[0] * fo.shape[1])
    if fo.shape != fg.shape:
        return "fhog shape %s vs %s" % (fg.shape, fo.shape)
    if not np.array_equal(fo, fg):
        return "fhog differs at %d values (cell %d bins %d ib %s ic %s, image %s)" % (int((fo != fg).sum()), cell, ub, ib, ic, gray.shape)
    return None


def case_aggregated(rng, ctx):
    ch = int(rng.choice([1, 3]))
    frame = rand_frame(rng, channels=ch)
    cell = int(rng.choice([4, 6, 8]))
    ww, wh = int(rng.integers(3, 8)), int(rng.integers(3, 8))
    ib, ic = bool(rng.random() < 0.5), bool(rng.random() < 0.7)
    wts = np.random.default_rng(int(rng.integers(1 << 30))).normal(0, 0.1, (wh, ww, 31)).astype(np.float32)
    octl = int(rng.integers(2, 7))
    minw = int(rng.choice([0, ww * cell * 2]))
    wsc, hsc = float(rng.choice([1.0, 0.8])), float(rng.choice([1.0, 1.2]))
    nms, mtype = float(rng.choice([0.3, 0.5])), int(rng.integers(0, 2))
    bias = 0.05
    okw = dict(cell_size=cell, interpolate_bins=ib, interpolate_cells=ic, octave_layers=octl, min_window_width=minw, width_scale=wsc, height_scale=hsc)
    r = O.aggregated_candidates(frame, wts, bias, -1e30, **okw)
    if r is None:   # fewer than two layers: the backend must refuse too
        try:
            det = capi.Aggregated(ctx, wts, bias, 0.0, nms_overlap=nms, nms_type=mtype, **okw)
            try:
                det.detect(frame)
            finally:
                det.close()
        except capi.FdError:
            return None
        return "oracle rejects the pyramid but the backend accepts"
    so, co = r
    if len(so) == 0:
        return None
    thr = float(np.quantile(so, 0.97))
    det = capi.Aggregated(ctx, wts, bias, thr, nms_overlap=nms, nms_type=mtype, **okw)
    try:
        fin, cand = det.detect(frame)
        so2, co2 = O.aggregated_candidates(frame, wts, bias, thr, **okw)
        STATS['candidates'] += len(so2)
        if len(cand) != len(so2):
            return "candidates %d vs %d" % (len(cand), len(so2))
        gb = np.stack([cand["x"], cand["y"], cand["w"], cand["h"]], 1) if len(cand) else np.zeros((0, 4), np.int32)
        if not np.array_equal(gb, np.asarray(co2).reshape(-1, 4)) or not np.array_equal(cand["score"], so2):
            return "candidate boxes / scores differ"
        fs, fb = O.nms_iou(so2, co2, nms, mtype)
        fgb = np.stack([fin["x"], fin["y"], fin["w"], fin["h"]], 1) if len(fin) else np.zeros((0, 4), np.int32)
        if len(fs) != len(fin) or not np.array_equal(fin["score"], fs) or not np.array_equal(fgb, np.asarray(fb).reshape(-1, 4)):
            return "final detections differ (%d vs %d)" % (len(fin), len(fs))
    finally:
        det.close()
    return None


def case_svm(rng, ctx):
    kernel, dtype = int(rng.integers(0, 4)), int(rng.integers(0, 2))
    nsv, dim, n = int(rng.integers(1, 700)), int(rng.integers(1, 1100)), int(rng.integers(1, 300))
    if dtype == 0:
        sv = rng.integers(0, 256, (nsv, dim), dtype=np.uint8)
        x = rng.integers(0, 256, (n, dim), dtype=np.uint8)
        p0 = {2: 0.04 / 65025.0 * 400 / dim, 1: 1.0 / 65025.0}.get(kernel, 0.0)
    else:
        sv = rng.random((nsv, dim)).astype(np.float32) * 0.2
        x = rng.random((n, dim)).astype(np.float32) * 0.2
        p0 = {2: 0.5, 1: 0.3}.get(kernel, 0.0)
    m = dict(kernel=kernel, p0=p0, p1=0.7, p2=int(rng.integers(1, 4)), dtype=dtype, sv=sv, coeff=rng.normal(0, 1, nsv).astype(np.float32),
             bias=np.float32(0.25), threshold=0.0)
    do = O.Svm(m).distance(x)
    sg = capi.Svm(ctx, m)
    try:
        dg = sg.distance(x)
    finally:
        sg.close()
    STATS['features'] += n * nsv
    scale = np.abs(m["coeff"]).sum() * (np.abs(do).max() / max(np.abs(m["coeff"]).sum(), 1e-30) if kernel in (0, 1, 3) else 1.0)
    tol = 1e-12 if dtype == 0 else 1e-5
    if not np.all(np.abs(dg - do) <= 1e-4 * np.abs(do) + tol * max(scale, 1.0)):
        return "svm distances differ: max %g (kernel %d dtype %d nsv %d dim %d n %d)" % (float(np.abs(dg - do).max()), kernel, dtype, nsv, dim, n)
    return None


def case_hog_svm(rng, ctx):
    """config-2 path (HOG features in the fragment layout + MFMA RBF SVM) on random geometry"""
    frame, frame2 = rand_frame(rng), rand_frame(rng)
    kw = rand_pyr_kw(rng)
    cell, block = int(rng.integers(3, 7)), int(rng.integers(1, 3))
    ncell = int(rng.integers(block, 5))
    pw = ph = cell * ncell
    bins = int(rng.integers(4, 10))
    sx, sy = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    po = O.Pyramid(**kw); po.set_layer_filter(1, bins=bins); po.update(frame2)
    pg = capi.Pyramid(ctx, **kw); pg.set_layer_filter(1, bins=bins)
    try:
        _, _, feats2 = O.sliding_hog_svm(po, None, pw, ph, sx, sy, bins, cell, block, want_feats=10 ** 9)
        if feats2 is None or len(feats2) < 40:
            return None
        nsv = int(rng.choice([33, 64, 150, 300]))
        m = synth.make_svm_f32(int(rng.integers(1 << 20)), feats2, nsv=min(nsv, len(feats2)), gamma=float(rng.choice([0.5, 2.0])), positive_fraction=0.05)
        po.update(frame)
        so = O.Svm(m)
        dets_o, dist_o, feats = O.sliding_hog_svm(po, so, pw, ph, sx, sy, bins, cell, block, want_feats=10 ** 9)
        pg.update(frame)
        hp = capi.hog_params(pw=pw, ph=ph, sx=sx, sy=sy, bins=bins, cell=cell, block=block, signed_and_unsigned=False)
        fg = capi.extract_hog(ctx, pg, hp)
        if len(dist_o) == 0:
            return None if len(fg) == 0 else "features without windows"
        STATS['windows'] += len(dist_o)
        if fg.shape != feats.shape or not np.array_equal(fg, feats):
            return "HOG features differ (cell %d block %d bins %d patch %d)" % (cell, block, bins, pw)
        sg = capi.Svm(ctx, m)
        try:
            dets_g, dist_g = capi.detect_hog_svm(ctx, pg, sg, hp)
        finally:
            sg.close()
        err = np.abs(dist_g - dist_o)
        scale = np.abs(m["coeff"]).sum()
        if err.max() > 1e-5 * scale or err.max() > 1e-4 * max(1.0, np.abs(dist_o).max()):
            return "HOG+SVM distances differ: %g (scale %g, F %d, nsv %d)" % (float(err.max()), float(scale), feats.shape[1], len(m["coeff"]))
        safe = np.abs(dist_o - m["threshold"]) > 1e-4
        pos_o, pos_g = np.nonzero(dist_o >= m["threshold"])[0], np.nonzero(dist_g >= m["threshold"])[0]
        if not np.array_equal(pos_o[safe[pos_o]], pos_g[safe[pos_g]]):
            return "positives differ away from the threshold"
        if np.array_equal(pos_o, pos_g):
            e = same_geometry(dets_g, dets_o)
            if e:
                return "HOG+SVM " + e
    finally:
        pg.close(); po.close()
    return None


def case_rvm(rng, ctx):
    frame = rand_frame(rng)
    kw = rand_pyr_kw(rng)
    space = int(rng.integers(0, 3))
    scale, shift = [(1.0, 0.0), (1.0 / 255.0, 0.0), (0.5, -3.0)][int(rng.integers(0, 3))]
    pw, ph = SIZES[int(rng.integers(len(SIZES)))]
    sx, sy = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    po = O.Pyramid(**kw); po.update(frame)
    pg = capi.Pyramid(ctx, **kw); pg.update(frame)
    try:
        layers = [po.layer(i) for i in range(len(po.layers()))]
        wins = po.windows(pw, ph, sx, sy)
        if len(wins) < 60:
            return None
        pat = np.stack([np.ascontiguousarray(layers[lp][ly:ly + ph, lx:lx + pw]) for lp, lx, ly, *_ in wins])
        if space == 1:
            pat = np.stack([O.histeq64(p_) for p_ in pat])
        elif space == 2:
            pat = np.stack([O.equalize_hist(p_) for p_ in pat])
        feats = pat.reshape(len(pat), -1).astype(np.float32) * np.float32(scale) + np.float32(shift)
        nf = int(rng.integers(1, 45))
        m = synth.make_rvm(int(rng.integers(1 << 20)), feats[::3], pw, ph, n_filters=nf, kernel=int(rng.choice([2, 2, 1, 3, 0])))
        ro, rg = O.Rvm(m), capi.Rvm(ctx, m)
        try:
            lo, do = ro.eval(feats)
            dets, lg, dg = capi.detect_rvm(ctx, pg, rg, feature_space=space, conv_scale=scale, conv_shift=shift, sx=sx, sy=sy)
            STATS['windows'] += len(lo)
            if len(lg) != len(lo) or not np.array_equal(lg, lo):
                return "RVM levels differ (%dx%d space %d filters %d)" % (pw, ph, space, nf)
            # fp64 chain with the device exp (an ulp off libm now and then): 1e-12 relative to the size of the terms
            atol = 1e-12 * float(np.abs(do).max())
            if not np.allclose(dg, do, rtol=1e-12, atol=atol):
                bad = np.nonzero(~np.isclose(dg, do, rtol=1e-12, atol=atol))[0]
                return "RVM distances differ at %d of %d windows, e.g. %r vs %r at level %d (%dx%d space %d scale %g shift %g filters %d kernel %d)" % (
                    len(bad), len(do), float(dg[bad[0]]), float(do[bad[0]]), int(lo[bad[0]]), pw, ph, space, scale, shift, nf, m["kernel"])
            pos = np.nonzero((lo == nf - 1) & (do >= m["thresholds"][nf - 1]))[0]
            if len(dets) != len(pos):
                return "RVM detections %d vs %d" % (len(dets), len(pos))
        finally:
            rg.close(); ro.close()
    finally:
        pg.close(); po.close()
    return None


def case_whi(rng, ctx):
    gray = rand_frame(rng, channels=1)
    h, w = int(rng.integers(8, 33)), int(rng.integers(8, 33))
    n = 24
    ys, xs = rng.integers(0, gray.shape[0] - h, n), rng.integers(0, gray.shape[1] - w, n)
    patches = np.stack([gray[y:y + h, x:x + w] for y, x in zip(ys, xs)])
    patches[0] = int(rng.integers(0, 256))
    try:
        eq_g = capi.equalize_hist_batch(ctx, patches)
    except capi.FdError as e:
        return "skip:" + str(e)[:80]
    eq_o = np.stack([O.equalize_hist(p_) for p_ in patches])
    STATS['features'] += int(patches.size)
    if not np.array_equal(eq_g, eq_o):
        return "equalizeHist differs (%dx%d)" % (w, h)
    alpha, cutoff = float(rng.choice([1.0, 0.5, 2.0])), float(rng.choice([0.390625, 0.0, 0.25]))
    wg = capi.whi_batch(ctx, patches, alpha, cutoff)
    wo = np.stack([O.whi(p_, alpha, cutoff) for p_ in patches])
    if not np.allclose(wg, wo, rtol=1e-6, atol=1e-9):
        return "whi chain differs: %g (%dx%d alpha %g cutoff %g)" % (float(np.abs(wg - wo).max()), w, h, alpha, cutoff)
    return None


def case_sdm(rng, ctx):
    gray = synth.make_frame(256, 256, seed=int(rng.integers(1 << 30)), channels=1)
    npts = int(rng.integers(1, 80))
    if rng.random() < 0.4:
        # the extractor's own geometry (non-adaptive): working images of 4..48 pixels, 3..16 orientations, both HOG variants
        nc = int(rng.integers(1, 7))
        cs = 2 * int(rng.integers(2, max(3, 48 // (2 * nc) + 1)))
        nb, variant = int(rng.integers(3, 17)), int(rng.integers(0, 2))
        half = nc * (cs // 2)
        px = rng.uniform(half + 1, 254 - half, npts).astype(np.float32)
        py = rng.uniform(half + 1, 254 - half, npts).astype(np.float32)
        kw = dict(variant=variant, num_cells=nc, cell_size=cs, num_bins=nb)
        do = O.sdm_descriptors(gray, px, py, 0, **kw)
        if do is None:
            return None
        try:
            dg = capi.sdm_descriptors(ctx, gray, px, py, 0, **kw)
        except capi.FdError as e:
            if "LDS budget" in str(e):   # a documented limit of the kernel (fd_hip.h), not a mismatch
                return None
            raise
        STATS['features'] += int(do.size)
        if dg.shape != do.shape or not np.array_equal(dg, do):
            return "SDM descriptors differ (non-adaptive %d cells x %d px, %d bins, variant %d, %d points)" % (nc, cs, nb, variant, npts)
        return None
    px = rng.uniform(2, 254, npts).astype(np.float32)
    py = rng.uniform(2, 254, npts).astype(np.float32)
    wsh = int(rng.integers(6, 40))
    do = O.sdm_descriptors(gray, px, py, wsh)
    if do is None:   # a window leaves the image where the reference throws
        try:
            capi.sdm_descriptors(ctx, gray, px, py, wsh)
        except capi.FdError:
            return None
        return "oracle rejects the points but the backend accepts (wsh %d)" % wsh
    dg = capi.sdm_descriptors(ctx, gray, px, py, wsh)
    STATS['features'] += int(do.size)
    if dg.shape != do.shape or not np.array_equal(dg, do):
        return "SDM descriptors differ (wsh %d, %d points)" % (wsh, npts)
    return None


def case_hog_fused(rng, ctx):
    """config 2's shape (20x20 patches, 5-pixel cells, 2x2 blocks, 9 bins) through k_hog_svm_fused on random frames, pyramids, strides and
    support-vector counts: against the oracle, and against the two-kernel path (FD_HOG_FUSED=0)"""
    frame, frame2 = rand_frame(rng), rand_frame(rng)
    kw = rand_pyr_kw(rng)
    sx, sy = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    po = O.Pyramid(**kw); po.set_layer_filter(1, bins=9); po.update(frame2)
    pg = capi.Pyramid(ctx, **kw); pg.set_layer_filter(1, bins=9)
    keep = os.environ.get("FD_HOG_FUSED")
    try:
        _, _, feats2 = O.sliding_hog_svm(po, None, 20, 20, sx, sy, 9, 5, 2, want_feats=10 ** 9)
        if feats2 is None or len(feats2) < 40:
            return None
        nsv = int(rng.choice([33, 64, 257, 700, 1024]))
        m = synth.make_svm_f32(int(rng.integers(1 << 20)), feats2, nsv=min(nsv, len(feats2)), gamma=float(rng.choice([0.5, 2.0])), positive_fraction=0.05)
        po.update(frame)
        dets_o, dist_o, _ = O.sliding_hog_svm(po, O.Svm(m), 20, 20, sx, sy, 9, 5, 2)
        if len(dist_o) == 0:
            return None
        pg.update(frame)
        hp = capi.hog_params(pw=20, ph=20, sx=sx, sy=sy, bins=9, cell=5, block=2, signed_and_unsigned=False)
        sg = capi.Svm(ctx, m)
        try:
            os.environ["FD_HOG_FUSED"] = "1"
            dets_g, dist_g = capi.detect_hog_svm(ctx, pg, sg, hp)
            os.environ["FD_HOG_FUSED"] = "0"
            dets_2, dist_2 = capi.detect_hog_svm(ctx, pg, sg, hp)
        finally:
            sg.close()
        STATS['windows'] += len(dist_o)
        scale = np.abs(m["coeff"]).sum()
        if len(dist_g) != len(dist_o):
            return "window count %d vs %d" % (len(dist_g), len(dist_o))
        err = np.abs(dist_g - dist_o)
        if err.max() > 1e-5 * scale or err.max() > 1e-4 * max(1.0, np.abs(dist_o).max()):
            return "fused HOG+SVM distances differ from the oracle: %g (scale %g, nsv %d, %d windows)" % (float(err.max()), float(scale), len(m["coeff"]), len(dist_o))
        if np.abs(dist_g - dist_2).max() > 2e-7 * scale:
            return "fused vs two-kernel path: %g (scale %g)" % (float(np.abs(dist_g - dist_2).max()), float(scale))
        safe = np.abs(dist_o - m["threshold"]) > 1e-4
        pos_o, pos_g = np.nonzero(dist_o >= m["threshold"])[0], np.nonzero(dist_g >= m["threshold"])[0]
        if not np.array_equal(pos_o[safe[pos_o]], pos_g[safe[pos_g]]):
            return "positives differ away from the threshold"
        if np.array_equal(pos_o, pos_g):
            e = same_geometry(dets_g, dets_o)
            if e:
                return "fused HOG+SVM " + e
    finally:
        if keep is None:
            os.environ.pop("FD_HOG_FUSED", None)
        else:
            os.environ["FD_HOG_FUSED"] = keep
        pg.close(); po.close()
    return None


def case_batch_tail(rng, ctx):
    """the jobs of a batch with their overlap elimination on the device (k_fs_oe_big, FD_FS_TAIL=1) against the host stages (FD_FS_TAIL=0):
    random frames, 1..4 detectors of random patch shapes on one pyramid, WVMs that leave hundreds to thousands of positives; one job
    also against the oracle"""
    frame = rand_frame(rng)
    if rng.random() < 0.3:   # repeated content: ties with equal outputs far apart
        half = frame[:, : frame.shape[1] // 2]
        frame = np.ascontiguousarray(np.concatenate([half, half], axis=1))
    gray = O.bgr2gray(frame)
    kw = rand_pyr_kw(rng)
    njobs = int(rng.integers(1, 5))
    pg = capi.Pyramid(ctx, **kw); pg.update(frame)
    po = O.Pyramid(**kw); po.update(frame)
    handles, models = [], []
    keep = os.environ.get("FD_FS_TAIL")
    try:
        for _ in range(njobs):
            pw, ph = SIZES[int(rng.integers(5))]   # the sizes with a dense pre-filter (the device tail needs the production cascade)
            src = np.ascontiguousarray(gray[::2, ::2])
            if src.shape[0] <= ph + 2 or src.shape[1] <= pw + 2:
                src = gray
            calib = synth.random_patches(src, pw, ph, 1500, rng)
            wvm = synth.make_wvm(int(rng.integers(1 << 20)), fw=pw, fh=ph, n_per=int(rng.choice([10, 14, 20])), n_levels=int(rng.integers(2, 4)), calib_patches=calib,
                                 min_survivors=int(rng.integers(30, 400)))
            eq = synth.histeq64_np(synth.random_patches(src, pw, ph, 260, rng))
            svm = synth.make_svm_u8(int(rng.integers(1 << 20)), eq, nsv=int(rng.choice([33, 64, 100])), calib=eq[100:], positive_fraction=0.4)
            models.append((wvm, svm))
            handles.append((capi.Wvm(ctx, wvm), capi.Svm(ctx, svm)))
        jobs = [(pg, w, s) for w, s in handles]
        os.environ["FD_FS_TAIL"] = "0"
        ref = capi.detect_five_stage_batch(ctx, jobs, cap=1 << 14)
        os.environ["FD_FS_TAIL"] = "1"
        got = capi.detect_five_stage_batch(ctx, jobs, cap=1 << 14)
        states = [w.last_tail_state() for w, _ in handles]
        for ji, ((dr, sr), (dg, sg)) in enumerate(zip(ref, got)):
            STATS['detections'] += len(dr)
            STATS['candidates'] += int(sr[0])
            if not np.array_equal(sr, sg) or dr.tobytes() != dg.tobytes():
                return "job %d of %d: device tail %s vs host stages %s (state %s, frame %s)" % (ji, njobs, sg.tolist(), sr.tolist(), states[ji], frame.shape)
        do, so = O.five_stage(po, O.Wvm(models[0][0]), O.Svm(models[0][1]), cap=1 << 14)
        if not np.array_equal(got[0][1], so):
            return "stage counts %s vs the oracle's %s" % (got[0][1].tolist(), so.tolist())
        e = same_geometry(got[0][0], do)
        if e:
            return "batch " + e
    finally:
        if keep is None:
            os.environ.pop("FD_FS_TAIL", None)
        else:
            os.environ["FD_FS_TAIL"] = keep
        for w, s in handles:
            w.close(); s.close()
        pg.close(); po.close()
    return None


def case_batch_group(rng, ctx):
    """detectors of a batch that scan the same windows share one pre-filter launch (k_wvm_prefilter_group): 2..8 detectors of ONE random
    patch shape (+ sometimes a detector of another shape) on one pyramid, random depths (4..16 dense levels), random frames -- the batch
    with groups against FD_WVM_GROUP=0, the ticket entry points (host stages on the batch queue) against the blocking call, and one
    member against the oracle"""
    frame = rand_frame(rng)
    gray = O.bgr2gray(frame)
    kw = rand_pyr_kw(rng)
    pw, ph = [(20, 20), (24, 24), (16, 24), (32, 16)][int(rng.integers(4))]   # the shapes with a group kernel
    nmem = int(rng.integers(2, 9))
    pg = capi.Pyramid(ctx, **kw); pg.update(frame)
    po = O.Pyramid(**kw); po.update(frame)
    handles, models = [], []
    keep = {k: os.environ.get(k) for k in ("FD_WVM_GROUP", "FD_BATCH_ASYNC")}
    try:
        shapes = [(pw, ph)] * nmem + ([SIZES[int(rng.integers(5))]] if rng.random() < 0.4 else [])
        for (w_, h_) in shapes:
            src = np.ascontiguousarray(gray[::2, ::2])
            if src.shape[0] <= h_ + 2 or src.shape[1] <= w_ + 2:
                src = gray
            calib = synth.random_patches(src, w_, h_, 1200, rng)
            wvm = synth.make_wvm(int(rng.integers(1 << 20)), fw=w_, fh=h_, n_per=int(rng.choice([4, 8, 14, 20, 30])), n_levels=int(rng.integers(2, 5)),
                                 calib_patches=calib, min_survivors=int(rng.integers(20, 200)))
            eq = synth.histeq64_np(synth.random_patches(src, w_, h_, 200, rng))
            svm = synth.make_svm_u8(int(rng.integers(1 << 20)), eq, nsv=int(rng.choice([33, 64])), calib=eq[64:], positive_fraction=0.4)
            models.append((wvm, svm))
            handles.append((capi.Wvm(ctx, wvm), capi.Svm(ctx, svm)))
        jobs = [(pg, w, s) for w, s in handles]
        if pg.window_count(pw, ph, 1, 1) < 512:
            return "skip: fewer than 512 windows (no dense pre-filter)"
        os.environ["FD_WVM_GROUP"] = "0"
        ref = capi.detect_five_stage_batch(ctx, jobs, cap=1 << 14)
        os.environ.pop("FD_WVM_GROUP")
        got = capi.detect_five_stage_batch(ctx, jobs, cap=1 << 14)
        os.environ["FD_BATCH_ASYNC"] = "1" if rng.random() < 0.7 else "0"
        tick = capi.FiveStageBatch(ctx, jobs, cap=1 << 14).end()
        for ji, ((dr, sr), (dg, sg), (dt, st_)) in enumerate(zip(ref, got, tick)):
            STATS['detections'] += len(dr)
            STATS['candidates'] += int(sr[0])
            if not np.array_equal(sr, sg) or dr.tobytes() != dg.tobytes():
                return "job %d of %d (%dx%d x %d): grouped %s vs separate %s" % (ji, len(jobs), pw, ph, nmem, sg.tolist(), sr.tolist())
            if not np.array_equal(sr, st_) or dr.tobytes() != dt.tobytes():
                return "job %d of %d: ticket %s vs blocking %s" % (ji, len(jobs), st_.tolist(), sr.tolist())
        mi = int(rng.integers(nmem))
        do, so = O.five_stage(po, O.Wvm(models[mi][0]), O.Svm(models[mi][1]), cap=1 << 14)
        if not np.array_equal(got[mi][1], so):
            return "member %d: stage counts %s vs the oracle's %s" % (mi, got[mi][1].tolist(), so.tolist())
        e = same_geometry(got[mi][0], do)
        if e:
            return "group member " + e
    finally:
        for k, v in keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        for w, s in handles:
            w.close(); s.close()
        pg.close(); po.close()
    return None


CASES = dict(pyramid=case_pyramid, cascade=case_cascade, frames=case_frames, hist=case_hist, fhog=case_fhog, aggregated=case_aggregated, svm=case_svm,
             hog_svm=case_hog_svm, rvm=case_rvm, whi=case_whi, sdm=case_sdm, hog_fused=case_hog_fused, batch_tail=case_batch_tail, batch_group=case_batch_group)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--kinds", default=",".join(CASES))
    ap.add_argument("--seconds", type=float, default=240.0, help="stop starting new cases after this long")
    ap.add_argument("--only", type=int, default=-1, help="run just this case index")
    args = ap.parse_args()
    ctx = capi.Context(0)
    kinds = args.kinds.split(",")
    bad, skipped, notes = [], 0, 0
    t0 = time.time()
    for i in range(args.cases):
        if time.time() - t0 > args.seconds:
            print("time limit after %d cases" % i)
            break
        if args.only >= 0 and i != args.only:
            continue
        kind = kinds[i % len(kinds)]
        rng = np.random.default_rng([args.seed, i])
        try:
            r = CASES[kind](rng, ctx)
        except Exception:
            r = "EXCEPTION " + traceback.format_exc().strip().splitlines()[-1][:200]
        if r is None:
            status = "ok"
        elif r.startswith("skip:"):
            status = r; skipped += 1
        elif r.startswith("note:"):
            status = r; notes += 1
        else:
            status = "MISMATCH " + r
            bad.append((i, kind, r))
        print("case %3d %-10s %s" % (i, kind, status), flush=True)
    print("fuzz summary: %d mismatches, %d skipped, %d tolerance-only, seed %d; compared %s" % (len(bad), skipped, notes, args.seed, STATS))
    for b in bad:
        print("  ", b)
    ctx.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
