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
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
