"""Generates tests/golden/*.npz.  Run in the build container (needs /root/reference for oracle/_ref):

    python tests/golden/make_golden.py

ref_*.npz hold outputs of the REFERENCE's own translation units (hog.c, IImg.cpp, svm.cpp compiled
from /root/reference by oracle/Makefile): they pin the oracle to the reference.  orc_*.npz hold
oracle outputs for the parts of the path the reference cannot pin (no tests, no models, OpenCV
absent): they pin the oracle against drift between rounds and travel to the GPU box.
Fixtures are data only (inputs + expected outputs)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O  # noqa: E402
from featuredetection_amd import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    rng = np.random.default_rng(20260927)
    r = O.ref()
    assert r is not None, "oracle/_ref/libfdref.so missing (needs /root/reference)"
    # ---- reference hog.c: 30x30 and odd-size float patches, both variants
    imgs = [rng.uniform(0, 255, (30, 30)).astype(np.float32), np.rint(rng.uniform(0, 255, (30, 30))).astype(np.float32),
            synth.make_frame(64, 48, seed=3, channels=1)[:30, :30].astype(np.float32),
            rng.uniform(0, 255, (24, 36)).astype(np.float32)]
    cases = [(0, 10, 9, 1), (1, 10, 9, 1), (2, 10, 9, 0), (3, 6, 4, 1), (2, 10, 9, 1), (3, 8, 9, 0)]
    hog = {"n": np.int32(len(cases))}
    for i, (ii, cell, nori, var) in enumerate(cases):
        hog["img%d" % i] = imgs[ii]
        hog["par%d" % i] = np.array([cell, nori, var], np.int32)
        hog["out%d" % i] = O.ref_vlhog(imgs[ii], cell, nori, var)
    np.savez_compressed(os.path.join(OUT, "ref_vlhog.npz"), **hog)
    # ---- reference IImg.cpp
    patches = [rng.integers(0, 256, (20, 20), dtype=np.uint8), np.full((20, 20), 255, np.uint8),
               rng.integers(0, 256, (24, 32), dtype=np.uint8), np.full((24, 32), 255, np.uint8)]
    ii = {"n": np.int32(len(patches))}
    import ctypes as C
    for i, p in enumerate(patches):
        ii["patch%d" % i] = p
        for sqr in (0, 1):
            out = np.empty(p.shape, np.float32)
            r.ref_iimg(p.ctypes.data_as(C.c_void_p), p.shape[1], p.shape[0], sqr, out.ctypes.data_as(C.c_void_p))
            ii["out%d_%d" % (i, sqr)] = out
    np.savez_compressed(os.path.join(OUT, "ref_iimg.npz"), **ii)
    # ---- reference libsvm: decision values for the four kernels
    nsv, dim = 24, 40
    sv = rng.uniform(0, 1, (nsv, dim))
    x = rng.uniform(0, 1, (6, dim))
    coef = rng.normal(0, 1, nsv)
    rho = 0.37
    svm = dict(sv=sv, x=x, coef=coef, rho=np.float64(rho))
    for name, (kt, deg, gamma, c0) in dict(linear=(0, 0, 0, 0), poly=(1, 3, 0.5, 1.0), rbf=(2, 0, 0.7, 0), hik=(5, 0, 0, 0)).items():
        svm["dec_" + name] = np.array([r.ref_svm_decision(kt, deg, gamma, c0, nsv, dim, sv.ctypes.data_as(C.c_void_p),
                                                           coef.ctypes.data_as(C.c_void_p), rho,
                                                           np.ascontiguousarray(xi).ctypes.data_as(C.c_void_p)) for xi in x])
        svm["par_" + name] = np.array([kt, deg, gamma, c0], np.float64)
    np.savez_compressed(os.path.join(OUT, "ref_libsvm.npz"), **svm)
    # ---- oracle regression vectors (small cascade on a 160x120 frame)
    frame = synth.make_frame(160, 120, seed=77)
    gray = O.bgr2gray(frame)
    calib = synth.random_patches(gray, 20, 20, 3000, np.random.default_rng(1))
    wvm = synth.make_wvm(21, n_per=5, n_levels=4, calib_patches=calib, min_survivors=48)
    eq = synth.histeq64_np(synth.random_patches(gray, 20, 20, 400, np.random.default_rng(2)))
    svmm = synth.make_svm_u8(4, eq, nsv=64, calib=eq[64:], positive_fraction=0.5)
    pyr = O.Pyramid(octave_layers=4, min_scale=0.4, max_scale=1.0)
    pyr.update(frame)
    w = O.Wvm(wvm)
    s = O.Svm(svmm)
    pos, lv, fo = O.sliding_wvm(pyr, w)
    dets, stages = O.five_stage(pyr, w, s)
    pyr2 = O.Pyramid(octave_layers=3, min_scale=0.3, max_scale=1.0)
    pyr2.set_layer_filter(1, bins=9)
    pyr2.update(frame)
    _, _, feats = O.sliding_hog_svm(pyr2, None, 20, 20, 2, 2, 9, 5, 2, want_feats=10 ** 9)
    g = dict(frame=frame, layer_sizes=np.array([[l["index"], l["w"], l["h"]] for l in pyr.layers()], np.int32),
             layer_sums=np.array([int(pyr.layer(i).astype(np.int64).sum()) for i in range(len(pyr.layers()))], np.int64),
             last_layer=pyr.layer(len(pyr.layers()) - 1), wvm_level=lv, wvm_fout=fo, wvm_pos=pos, five=dets, stages=stages,
             hog_feat_head=feats[:64], hog_feat_sum=np.float64(feats.astype(np.float64).sum()), hog_n=np.int64(len(feats)))
    for k, v in wvm.items():
        g["wvm__" + k] = np.asarray(v)
    for k, v in svmm.items():
        g["svm__" + k] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, "orc_cascade_160x120.npz"), **g)
    next_rows()
    print("golden fixtures written to", OUT)


def next_rows():
    """orc_next_rows_128x96.npz: oracle regression vectors of the SURVEY 8(f) rows (FHOG on gray / BGR images, the aggregated
    detector, the cascaded RVM, the whitening chain, one HistogramFilter variant) on a 128x96 frame."""
    frame = synth.make_frame(128, 96, seed=91)
    gray = O.bgr2gray(frame)
    g = dict(frame=frame, gray=gray)
    g["fhog_gray"] = O.fhog(gray, 8, 9, False, True, 0.2)
    g["fhog_bgr"] = O.fhog(frame, 8, 9, False, True, 0.2)
    g["fhog_gray_c4_b6_ib"] = O.fhog(gray, 4, 6, True, False, 0.2)
    wts = np.random.default_rng(5).normal(0, 0.1, (4, 5, 31)).astype(np.float32)
    so, co = O.aggregated_candidates(frame, wts, 0.05, -1e30, cell_size=8, octave_layers=4)
    thr = float(np.float32(np.quantile(so, 0.9)))
    sc, cc = O.aggregated_candidates(frame, wts, 0.05, thr, cell_size=8, octave_layers=4)
    fs, fb = O.nms_iou(sc, cc, 0.3, 0)
    g.update(agg_weights=wts, agg_threshold=np.float32(thr), agg_cand_scores=sc, agg_cand_boxes=np.asarray(cc, np.int32).reshape(-1, 4),
             agg_final_scores=fs, agg_final_boxes=np.asarray(fb, np.int32).reshape(-1, 4))
    # cascaded RVM on HistEq64 windows of a small pyramid
    kw = dict(octave_layers=2, min_scale=0.4, max_scale=0.8)
    pyr = O.Pyramid(**kw)
    pyr.update(frame)
    layers = [pyr.layer(i) for i in range(len(pyr.layers()))]
    wins = pyr.windows(20, 20, 2, 2)
    pat = np.stack([O.histeq64(np.ascontiguousarray(layers[lp][ly:ly + 20, lx:lx + 20])) for lp, lx, ly, *_ in wins])
    feats = pat.reshape(len(pat), -1).astype(np.float32)
    rvm = synth.make_rvm(13, feats[::3], 20, 20, n_filters=12, kernel=2)
    lo, do = O.Rvm(rvm).eval(feats)
    g.update(rvm_level=lo, rvm_dist=do)
    for k, v in rvm.items():
        g["rvm__" + k] = np.asarray(v)
    # whitening chain and equalizeHist on a few patches
    pp = np.stack([gray[y:y + 20, x:x + 20] for y, x in ((0, 0), (10, 30), (40, 77), (76, 108))])
    g.update(whi_patches=pp, whi_out=np.stack([O.whi(q, 1.0, 0.390625) for q in pp]), eqhist_out=np.stack([O.equalize_hist(q) for q in pp]))
    # interpolating HogFilter with signed + unsigned bins on the bin image of the last pyramid layer
    pyr2 = O.Pyramid(**kw)
    pyr2.set_layer_filter(kind=1, bins=8, signed_gradients=True, interpolate=True)
    pyr2.update(frame)
    L = pyr2.layer(len(pyr2.layers()) - 1)
    g["hist_hog_interp"] = np.stack([O.hog_filter(np.ascontiguousarray(L[y:y + 20, x:x + 20]), 8, 5, 2, True, True) for y, x in ((0, 0), (7, 11), (15, 20))])
    np.savez_compressed(os.path.join(OUT, "orc_next_rows_128x96.npz"), **g)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "next_rows":
        next_rows()
    else:
        main()
