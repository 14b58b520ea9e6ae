#!/usr/bin/env python
"""tests/golden/make_sdm_real.py -- BUILD CONTAINER ONLY (reads /root/reference): re-packs the one trained model the reference ships,
detect-landmarks/share/models/SDM_Model_HOG_Zhenhua_11012014.txt (22 landmarks, 5 regressors of 3169 / 3169 / 1409 / 1409 / 353 rows x
44 columns, header lines 47-48, 3218, 6388, 7798, 9208), into the fields of fd_sdm_model, runs the CPU oracle on 32 seeded crops and
writes tests/golden/sdm_real_11012014.npz = {model, inputs' seeds, per-step oracle shapes, per-step reference (hog.c) descriptors of
two faces}.  Data only: the regressor coefficients (as integers in 1e-6 units, exactly the six decimals of the text file), no source.

What had to be decided, because the file is in a format the reference's current loader cannot parse (`numHogScales` / `scale i
rows r cols c cellSize s numBins b` instead of `numCascadeSteps` + four header lines per step, SdmLandmarkModel.cpp:161-210) and because
optimize() as compiled (`if (true) { // adaptive`, SdmLandmarkModel.hpp:209) always extracts 3x3x31 = 279 values per landmark, which no
row count of this file matches (22 x 279 + 1 = 6139):

* The fit runs the NON-ADAPTIVE branch of optimize() (:236-238, :246-248; the `else` arms upstream compiles out): getDescriptors(image,
  points) with the extractor's own parameters, shape += delta without the face-size factor (fd_sdm_model.desc_params, oracle:
  orc_sdm_optimize_fixed).
* Descriptor geometry.  Rows = 22 x cells^2 x 16 + 1 with cells = 3, 3, 2, 2, 1 and 16 = 3 x numBins + 4 (UoCTTI, numBins 4): the file's
  "cellSize" column {3,3,2,2,1} is Zhenhua's Matlab convention = number of HOG cells across the patch
  (detect-landmarks/share/scripts/SDMMatToTxt.m:22 `params = { {3, 4}, {3, 4}, {2, 4}, {2, 4}, {1, 4} }`, DescriptorExtractor.hpp:141-142
  "cellSize: has nothing to do with HOG. It's rather the number of HOG cells we want").  Read literally as VlHogDescriptorExtractor(
  Uoctti, numCells 3, cellSize 3, numBins 4) the patch is 2 x 3 x (3 / 2) = 6 pixels and vl_hog yields (6 + 1) / 3 = 2 cells: 64 values,
  not 144 -- so the literal reading cannot be what the regressors were trained on.  The HOG cell size in PIXELS is not recorded anywhere
  in the tree.  This fixture therefore uses numCells = {3,3,2,2,1} with an even pixel cell size C per step, for which patchWidthHalf =
  numCells x C / 2 and vl_hog's (numCells x C + C / 2) / C = numCells cells: C = {10, 8, 8, 6, 6} (patches 30, 24, 16, 12, 6 pixels,
  shrinking with the step like the paper's schedule).  The regressor VALUES and shapes are the reference's; the pixel scale is ours.
* Mean shape.  The file stores the mean in the pixel coordinates of the training images (x 160..327, y 158..327); alignRigid
  (SdmLandmarkModel.hpp:156-192) expects it in [-0.5, 0.5]^2 of the face box.  Normalised here as (x - cx) / s, (y - cy) / s with (cx,
  cy) the centre of the landmarks' bounding box and s = 1.25 x its larger side (1.25: "faceboxScaleFactor" of the commented-out
  alignment at :176).

Usage (build container): python tests/golden/make_sdm_real.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
SRC = "/root/reference/detect-landmarks/share/models/SDM_Model_HOG_Zhenhua_11012014.txt"
OUT = os.path.join(ROOT, "tests", "golden", "sdm_real_11012014.npz")

NUM_CELLS = [3, 3, 2, 2, 1]
CELL_PX = [10, 8, 8, 6, 6]
NUM_BINS = 4
NFACES = 32
FACE_BOX = [48, 48, 160, 160]


def parse(path):
    lines = open(path).read().splitlines()
    assert lines[0].startswith("#")
    L = int(lines[1].split()[1])
    mean_px = np.array([float(v) for v in lines[2:2 + 2 * L]], np.float64)
    k = 2 + 2 * L
    S = int(lines[k].split()[1])
    assert lines[k].split()[0] == "numHogScales"
    k += 1
    R_micro, hdr = [], []
    for s in range(S):
        t = lines[k].split()   # scale i rows r cols c cellSize s numBins b
        assert t[0] == "scale" and int(t[1]) == s
        rows, cols, cs, nb = int(t[3]), int(t[5]), int(t[7]), int(t[9])
        hdr.append((rows, cols, cs, nb))
        # six decimals in the text: keep them as exact integers (1e-6 units)
        m = np.array([[int(round(float(v) * 1e6)) for v in lines[k + 1 + r].split()] for r in range(rows)], np.int32)
        assert m.shape == (rows, cols)
        R_micro.append(m)
        k += 1 + rows
    return L, mean_px, hdr, R_micro


def unpack_model(g):
    """the dict capi.Sdm / oracle.sdm_fit take, from the committed arrays (also used by the tests)"""
    S = int(g["S"])
    # the float the reference's lexical_cast<float> yields for the six-decimal text: nearest fp32 of the decimal
    R = [(g["R%d_micro" % s].astype(np.float64) / 1e6).astype(np.float32) for s in range(S)]
    return dict(L=int(g["L"]), S=S, mean=g["mean"].astype(np.float32), R=R, variant=1,
                desc_params=np.stack([g["num_cells"], g["cell_px"], np.full(S, int(g["num_bins"]))], 1).astype(np.int32).ravel())


def main():
    from featuredetection_amd import synth
    from oracle import pyoracle as O
    L, mean_px, hdr, R_micro = parse(SRC)
    assert L == 22 and [h[0] for h in hdr] == [3169, 3169, 1409, 1409, 353] and all(h[1] == 44 for h in hdr)
    assert [h[2] for h in hdr] == NUM_CELLS and all(h[3] == NUM_BINS for h in hdr)
    for s, (rows, _, _, _) in enumerate(hdr):
        assert rows == L * NUM_CELLS[s] ** 2 * (3 * NUM_BINS + 4) + 1
    x, y = mean_px[:L], mean_px[L:]
    cx, cy = (x.min() + x.max()) / 2, (y.min() + y.max()) / 2
    side = 1.25 * max(x.max() - x.min(), y.max() - y.min())
    mean = np.concatenate([(x - cx) / side, (y - cy) / side]).astype(np.float32)
    g = dict(L=L, S=len(hdr), mean=mean, mean_px=mean_px, num_cells=np.array(NUM_CELLS, np.int32), cell_px=np.array(CELL_PX, np.int32),
             num_bins=NUM_BINS, face_box=np.array(FACE_BOX, np.int32), nfaces=NFACES, frame_seed0=7100)
    for s, m in enumerate(R_micro):
        g["R%d_micro" % s] = m
    model = unpack_model(g)
    # oracle: 32 seeded crops, the shape after every cascade step (a prefix model per step), final status
    imgs = [synth.make_frame(256, 256, seed=7100 + i, channels=1) for i in range(NFACES)]
    steps = np.zeros((NFACES, model["S"] + 1, 2 * L), np.float32)
    for i, im in enumerate(imgs):
        for s in range(model["S"] + 1):
            sub = dict(model, S=s, R=model["R"][:s], desc_params=model["desc_params"][:3 * s])
            if s == 0:
                sh = model["mean"].copy()
                O.lib().orc_sdm_align_rigid(sh.ctypes.data, L, np.array(FACE_BOX, np.int32).ctypes.data)
                st = 0
            else:
                st, sh = O.sdm_fit(im, sub, FACE_BOX)
            assert st == 0, (i, s, st)
            steps[i, s] = sh
    g["oracle_shapes"] = steps
    # the reference's own hog.c (oracle/_ref) on the patches of faces 0 and 17 at every step: crop (DescriptorExtractor.hpp:156-178) ->
    # vl_hog -> per-plane transpose and stack (:198-205)
    assert O.ref() is not None, "oracle/_ref/libfdref.so missing: run make -C oracle in the build container"
    for f in (0, 17):
        for s in range(model["S"]):
            nc, cp = NUM_CELLS[s], CELL_PX[s]
            pwh = nc * (cp // 2)
            sh = steps[f, s]
            out = []
            for i in range(L):
                px, py = int(np.rint(sh[i])), int(np.rint(sh[i + L]))   # cvRound = rint (half to even)
                assert px - pwh >= 0 and py - pwh >= 0 and px + pwh < 256 and py + pwh < 256
                roi = imgs[f][py - pwh:py + pwh, px - pwh:px + pwh].astype(np.float32)
                h = O.ref_vlhog(roi, cp, NUM_BINS, 1)   # [dd][hh][ww]
                out.append(np.stack([pl.T.reshape(-1) for pl in h]).reshape(-1))
            g["ref_desc_f%d_s%d" % (f, s)] = np.stack(out).astype(np.float32)
    np.savez_compressed(OUT, **g)
    mv = np.abs(np.diff(steps, axis=1)).max(axis=(0, 2))
    print("wrote %s (%.2f MB): max |landmark move| per step %s px; |R| mean %.3f" %
          (OUT, os.path.getsize(OUT) / 1e6, np.round(mv, 2).tolist(), np.mean([np.abs(r).mean() for r in model["R"]])))


if __name__ == "__main__":
    main()
