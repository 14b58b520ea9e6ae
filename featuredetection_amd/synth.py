"""Seeded synthetic frames and classifier models.

The reference ships no WVM/SVM model files (they are Matlab .mat files on the author's disk,
ffpDetectApp/FaceFrontal.cfg:7-16), so models are synthesised here in the unit conventions of the
reference loaders (0..255 grey values, gamma / 65025 etc.).  Pure numpy; used by tests, bench.py
and __graft_entry__.smoke().  Nothing here depends on oracle/.
"""
import numpy as np

# ffpDetectApp/*.cfg: name -> (incrementalScaleFactor, minScale, maxScale, patch w, patch h, nPerLevel, levels)
DETECTOR_CFGS = {
    "FaceFrontal": (0.92, 0.05, 0.16, 20, 20, 14, 20),
    "FaceLeftProfile": (0.9, 0.09, 0.25, 20, 20, 14, 7),
    "FaceRightProfile": (0.9, 0.09, 0.25, 20, 20, 14, 7),
    "LeftEarCenter": (0.9, 0.5, 0.7, 16, 24, 20, 10),
    "RightEarCenter": (0.9, 0.5, 0.7, 16, 24, 20, 10),
    "LeftEyeCenter": (0.9, 0.5, 0.7, 32, 16, 20, 8),
    "RightEyeCenter": (0.85, 0.5, 0.7, 32, 16, 20, 8),
    "LeftEyeOuterCorner": (0.9, 0.5, 0.7, 24, 24, 20, 11),
    "RightEyeOuterCorner": (0.9, 0.5, 0.7, 24, 24, 20, 11),
    "LeftLipCorner": (0.9, 0.5, 0.7, 24, 24, 30, 8),
    "RightLipCorner": (0.9, 0.5, 0.7, 24, 24, 30, 8),
    "LeftNoseCorner": (0.9, 0.5, 0.7, 24, 24, 30, 8),
    "RightNoseCorner": (0.9, 0.5, 0.7, 24, 24, 30, 8),
    "CenterLipUpperOuter": (0.9, 0.5, 0.7, 24, 24, 30, 8),
    "NoseTip": (0.9, 0.5, 0.7, 32, 24, 30, 7),
}


def _blur(a, sigma):
    """separable Gaussian blur, reflect border (numpy only)"""
    r = int(3 * sigma + 0.5)
    k = np.exp(-0.5 * (np.arange(-r, r + 1) / sigma) ** 2)
    k /= k.sum()
    p = np.pad(a, ((r, r), (0, 0)), mode="reflect")
    a = sum(k[i] * p[i:i + a.shape[0]] for i in range(2 * r + 1))
    p = np.pad(a, ((0, 0), (r, r)), mode="reflect")
    return sum(k[i] * p[:, i:i + a.shape[1]] for i in range(2 * r + 1))


def make_frame(width=640, height=480, seed=20260927, channels=3):
    """Smooth noise base + 3 pasted high-contrast blobs (SURVEY.md 8(d) config 1 recipe)."""
    rng = np.random.default_rng(seed)
    planes = []
    for _ in range(channels):
        base = _blur(rng.random((height, width)), 8.0)
        base = (base - base.min()) / max(base.max() - base.min(), 1e-12)
        fine = rng.random((height, width))
        planes.append(0.8 * base + 0.2 * fine)
    img = np.stack(planes, axis=-1)
    for _ in range(3):
        s = int(rng.integers(64, 161))
        s = min(s, height - 2, width - 2)
        y0 = int(rng.integers(0, height - s))
        x0 = int(rng.integers(0, width - s))
        yy, xx = np.mgrid[0:s, 0:s]
        blob = 0.5 + 0.5 * np.sin(yy / s * rng.uniform(4, 12)) * np.cos(xx / s * rng.uniform(4, 12))
        img[y0:y0 + s, x0:x0 + s, :] = 0.15 * img[y0:y0 + s, x0:x0 + s, :] + 0.85 * blob[..., None]
    out = np.clip(np.rint(img * 255), 0, 255).astype(np.uint8)
    return out if channels == 3 else out[..., 0]


def make_frames_varied(n, width=640, height=480, seed=20260927, scene_len=32, nbase=9):
    """n distinct frames as a run of short 'scenes' (bench.py's headline content): smooth noise fields drifting like a camera pan,
    per-frame fine noise, and 0..10 high-contrast blobs that move through the scene.  The busy-ness (blob count, fine-noise share,
    contrast) is drawn per scene, so consecutive multi-frame calls hand the cascade different numbers of surviving windows.
    Returns (list of HxWx3 uint8 frames, per-frame busy-ness in [0, 1])."""
    rng = np.random.default_rng(seed)
    bases = []
    for i in range(nbase):
        b = _blur(rng.random((height, width)), (4.0, 8.0, 14.0)[i % 3])
        bases.append(((b - b.min()) / max(b.max() - b.min(), 1e-12)).astype(np.float32))
    frames, busy = [], []
    yy, xx = np.mgrid[0:160, 0:160].astype(np.float32)
    while len(frames) < n:
        b = float(rng.random())
        nblob = int(round(b * 10))
        fine = np.float32(0.05 + 0.35 * rng.random())
        gain = np.float32(0.5 + 0.5 * rng.random())          # contrast of the smooth part
        idx = rng.integers(0, nbase, 3)
        vel = rng.integers(-3, 4, (3, 2))
        blobs = [dict(s=int(rng.integers(48, 161)), y=float(rng.uniform(0, height - 161)), x=float(rng.uniform(0, width - 161)),
                      vy=float(rng.uniform(-2, 2)), vx=float(rng.uniform(-2, 2)), fy=float(rng.uniform(4, 12)), fx=float(rng.uniform(4, 12)))
                 for _ in range(nblob)]
        for t in range(scene_len):
            if len(frames) >= n:
                break
            img = np.empty((height, width, 3), np.float32)
            noise = rng.random((height, width, 3), dtype=np.float32)
            for c in range(3):
                base = np.roll(bases[idx[c]], (int(vel[c, 0]) * t, int(vel[c, 1]) * t), axis=(0, 1))
                img[..., c] = (np.float32(1) - fine) * (np.float32(0.5) + gain * (base - np.float32(0.5))) + fine * noise[..., c]
            for bl in blobs:
                s = bl["s"]
                y0 = int(min(max(bl["y"] + bl["vy"] * t, 0), height - s - 1))
                x0 = int(min(max(bl["x"] + bl["vx"] * t, 0), width - s - 1))
                blob = 0.5 + 0.5 * np.sin(yy[:s, :s] / s * bl["fy"]) * np.cos(xx[:s, :s] / s * bl["fx"])
                img[y0:y0 + s, x0:x0 + s, :] = 0.15 * img[y0:y0 + s, x0:x0 + s, :] + 0.85 * blob[..., None]
            frames.append(np.clip(np.rint(img * 255), 0, 255).astype(np.uint8))
            busy.append(b)
    return frames, np.asarray(busy)


def bgr2gray_np(img):
    """cv::cvtColor(BGR2GRAY) on 8-bit images: (B * 1868 + G * 9617 + R * 4899 + 8192) >> 14; used to draw calibration
    patches for the synthetic models (bench.py must not need the oracle to build them)"""
    a = np.asarray(img, np.uint8).astype(np.uint32)
    return ((a[..., 0] * 1868 + a[..., 1] * 9617 + a[..., 2] * 4899 + 8192) >> 14).astype(np.uint8)


def histeq64_np(patches):
    """Vectorised HistEq64 used for threshold calibration / SV synthesis; tests/test_oracle_golden.py also checks the C++ oracle against it."""
    p = np.asarray(patches, np.uint8)
    n = p.shape[0]
    flat = p.reshape(n, -1)
    d = flat.shape[1]
    bins = flat >> 2
    hist = np.zeros((n, 64), np.float32)
    np.add.at(hist, (np.repeat(np.arange(n), d), bins.ravel()), 1.0)
    pdf = hist * np.float32(255.0 / d)
    cdf = np.cumsum(pdf, axis=1, dtype=np.float32)
    lut = np.floor(cdf.astype(np.float64) + 0.5).astype(np.uint8)
    return np.take_along_axis(lut, bins.astype(np.int64), axis=1).reshape(p.shape)


def random_patches(frame_gray, pw, ph, n, rng):
    H, W = frame_gray.shape
    ys = rng.integers(0, H - ph, n)
    xs = rng.integers(0, W - pw, n)
    return np.stack([frame_gray[y:y + ph, x:x + pw] for y, x in zip(ys, xs)])


def make_wvm(seed, fw=20, fh=20, n_per=14, n_levels=20, r=0.04, calib_patches=None, pass_rate=0.65, min_survivors=32,
             cntval=6, rect_range=(2, 8), reject_every=1, hk_scale=1.0):
    """Synthetic wavelet reduced vector machine in the layout of fd_wvm_model / orc_wvm_desc.

    Structure follows the cfg-implied one (n_per x n_levels filters); thresholds are calibrated so that
    ~pass_rate of the surviving calibration patches pass each filter while at least min_survivors
    remain, after that every survivor passes (mimics a cascade).
    Rejection profiles (the rejection rate per level drives throughput): pass_rate / min_survivors
    set how fast and how far the cascade thins out; reject_every = n rejects only at every n-th filter (e.g. n_per:
    only the last filter of a level group has a real threshold).  hk_scale scales the hyperplane weights (large
    values give sums with heavy cancellation)."""
    rng = np.random.default_rng(seed)
    F = n_per * n_levels
    d = fw * fh
    val_off = np.arange(F + 1, dtype=np.int32) * cntval
    val = rng.uniform(0, 255, F * cntval)
    rec_off = [0]
    rects = []
    for k in range(F):
        for v in range(cntval):
            nrec = 0 if v == 0 else int(rng.integers(rect_range[0], rect_range[1] + 1))
            for _ in range(nrec):
                x1 = int(rng.integers(0, fw)); x2 = int(rng.integers(x1, min(fw, x1 + max(2, fw // 2))))
                y1 = int(rng.integers(0, fh)); y2 = int(rng.integers(y1, min(fh, y1 + max(2, fh // 2))))
                rects.append((x1, y1, x2, y2))
            rec_off.append(len(rects))
    rects = np.asarray(rects, np.uint8).reshape(-1, 4)
    rec_off = np.asarray(rec_off, np.int32)
    # later approximation levels hold residuals: small values around zero (cumulative vector stays in range)
    for k in range(n_per, F):
        v0 = val_off[k]
        val[v0:v0 + cntval] = (val[v0:v0 + cntval] - 127.5) / (2.0 ** (k // n_per))
    # dense residual images r_k and cumulative approximated vectors P_k
    dense = np.zeros((F, fh, fw), np.float64)
    for k in range(F):
        v0 = val_off[k]
        dense[k] += val[v0]
        for v in range(1, cntval):
            for i in range(rec_off[v0 + v], rec_off[v0 + v + 1]):
                x1, y1, x2, y2 = rects[i]
                dense[k, y1:y2 + 1, x1:x2 + 1] += val[v0 + v] - val[v0]
    P = np.zeros_like(dense)
    for k in range(F):
        P[k] = dense[k] + (P[k - n_per] if k >= n_per else 0)
    pp = (P.reshape(F, -1) ** 2).sum(1)
    basis = np.float32(r / 65025.0)
    hk = np.zeros((F, F), np.float32)
    for k in range(F):
        hk[k, :k + 1] = (hk_scale * rng.normal(0, 1.0 / np.sqrt(k + 1), k + 1)).astype(np.float32)
    bias = np.float32(0.0)
    thresholds = np.full(F, -1e30, np.float32)
    if calib_patches is not None and len(calib_patches):
        X = histeq64_np(calib_patches).reshape(len(calib_patches), -1).astype(np.float64)
        sxx = (X ** 2).sum(1)
        xp = X @ P.reshape(F, -1).T
        norm = sxx[:, None] - 2 * xp + pp[None, :]
        K = np.exp(-float(basis) * norm)
        res = K @ hk.astype(np.float64).T - float(bias)  # res[:, k] = sum_{p<=k} w[k][p] K[p]
        alive = np.ones(len(X), bool)
        for k in range(F):
            vals = res[alive, k]
            if vals.size >= min_survivors and (k + 1) % reject_every == 0:
                thr = np.quantile(vals, 1.0 - pass_rate)
            else:
                thr = (vals.min() - 1.0) if vals.size else -1e30
            thresholds[k] = np.float32(thr)
            alive &= res[:, k] >= thresholds[k]
    return dict(filter_w=fw, filter_h=fh, num_filters=F, num_used=F, num_per_level=n_per, basis_param=float(basis),
                bias=float(bias), thresholds=thresholds, hk_weights=hk, pp=pp.astype(np.float64), val_off=val_off,
                val=val.astype(np.float64), rec_off=rec_off, rects=rects, logistic_a=0.00556, logistic_b=-2.95)


def make_svm_u8(seed, patches_eq, nsv=1024, r=0.04, positive_fraction=0.3, calib=None):
    """RBF SVM on u8 HistEq64 patches (second cascade stage): SVs = equalised random patches,
    gamma = r/65025 (SvmClassifier.cpp:264), bias chosen so that ~positive_fraction of calib is positive."""
    rng = np.random.default_rng(seed)
    sv = np.asarray(patches_eq[:nsv], np.uint8).reshape(nsv, -1)
    coeff = rng.normal(0, 1, nsv).astype(np.float32)
    gamma = float(np.float32(r / 65025.0))
    bias = 0.0
    if calib is not None and len(calib):
        X = np.asarray(calib, np.uint8).reshape(len(calib), -1).astype(np.float64)
        S = sv.astype(np.float64)
        d2 = (X ** 2).sum(1)[:, None] + (S ** 2).sum(1)[None, :] - 2 * X @ S.T
        dist = np.exp(-gamma * d2) @ coeff.astype(np.float64)
        bias = float(np.quantile(dist, 1.0 - positive_fraction))
    return dict(kernel=2, p0=gamma, p1=0.0, p2=0.0, dtype=0, sv=sv, coeff=coeff, bias=np.float32(bias), threshold=0.0,
                logistic_a=0.00556, logistic_b=-2.95)


def make_svm_f32(seed, feats, nsv=1024, gamma=0.5, positive_fraction=0.01, kernel=2):
    """SVM on f32 feature vectors (config 2: HOG-324): SVs drawn from real feature vectors of a second
    frame, coefficients ~N(0,1), bias such that ~positive_fraction of `feats` is positive."""
    rng = np.random.default_rng(seed)
    feats = np.asarray(feats, np.float32)
    idx = rng.choice(len(feats), nsv, replace=len(feats) < nsv)
    sv = feats[idx].copy()
    coeff = rng.normal(0, 1, nsv).astype(np.float32)
    sub = feats[rng.choice(len(feats), min(len(feats), 4096), replace=False)].astype(np.float64)
    S = sv.astype(np.float64)
    if kernel == 2:
        d2 = (sub ** 2).sum(1)[:, None] + (S ** 2).sum(1)[None, :] - 2 * sub @ S.T
        dist = np.exp(-gamma * np.maximum(d2, 0)) @ coeff.astype(np.float64)
    elif kernel == 3:
        dist = np.minimum(sub[:, None, :], S[None, :, :]).sum(-1) @ coeff.astype(np.float64)
    else:
        dist = (sub @ S.T) @ coeff.astype(np.float64)
    bias = float(np.quantile(dist, 1.0 - positive_fraction))
    return dict(kernel=kernel, p0=gamma, p1=0.0, p2=0.0, dtype=1, sv=sv, coeff=coeff, bias=np.float32(bias), threshold=0.0,
                logistic_a=0.00556, logistic_b=-2.95)


def make_sdm(seed, L=68, S=4, feat_per_landmark=279, sigma=1e-3):
    """SDM model: mean shape = L points on a 0.6-scale template in
    [-0.5,0.5]^2, R_s ~ N(0, sigma^2) of shape (L*279 + 1) x 2L."""
    rng = np.random.default_rng(seed)
    ang = np.linspace(0, 2 * np.pi, L, endpoint=False)
    rad = 0.3 * (0.6 + 0.4 * rng.random(L))
    mean = np.concatenate([rad * np.cos(ang), rad * np.sin(ang)]).astype(np.float32)
    # landmarks 8,9 (eyes) above 11,12 (mouth) so that the anchor distance is well defined
    mean[8], mean[9], mean[8 + L], mean[9 + L] = -0.12, 0.12, -0.12, -0.12
    mean[11], mean[12], mean[11 + L], mean[12 + L] = -0.1, 0.1, 0.18, 0.18
    F = L * feat_per_landmark
    R = [rng.normal(0, sigma, (F + 1, 2 * L)).astype(np.float32) for _ in range(S)]
    return dict(L=L, S=S, mean=mean, R=R, variant=1)


# ---- model / image files read by the C++ host layer (featuredetection_amd/host) -----------------------
def save_wvm(path, m):
    """Binary FDWVM1 file (WvmClassifier::loadFromFile): the reference's Matlab .mat models are absent."""
    import struct
    F = int(m["num_filters"])
    val = np.ascontiguousarray(m["val"], np.float64)
    rects = np.ascontiguousarray(m["rects"], np.uint8).reshape(-1, 4)
    with open(path, "wb") as f:
        f.write(b"FDWVM1\0\0")
        f.write(struct.pack("<7i", int(m["filter_w"]), int(m["filter_h"]), F, int(m["num_used"]), int(m["num_per_level"]), len(val), len(rects)))
        f.write(struct.pack("<2f", float(m["basis_param"]), float(m["bias"])))
        f.write(np.ascontiguousarray(m["thresholds"], np.float32).tobytes())
        f.write(np.ascontiguousarray(m["hk_weights"], np.float32).tobytes())
        f.write(np.ascontiguousarray(m["pp"], np.float64).tobytes())
        f.write(np.ascontiguousarray(m["val_off"], np.int32).tobytes())
        f.write(val.tobytes())
        f.write(np.ascontiguousarray(m["rec_off"], np.int32).tobytes())
        f.write(rects.tobytes())


def make_rvm(seed, feats, fw, fh, n_filters=24, kernel=2, gamma=None, pass_rate=0.6, min_survivors=24):
    """Synthetic cascaded reduced-vector machine (RvmClassifier): reduced set vectors = smoothed training vectors,
    lower-triangular coefficients, thresholds calibrated on `feats` (f32 [n][fw*fh], already in the classifier's
    feature space) so that ~pass_rate of the survivors pass each level while at least min_survivors remain.
    The calibration follows the reference's cached evaluation (d_k = d_{k-1} + c[k][k] K_k)."""
    rng = np.random.default_rng(seed)
    feats = np.ascontiguousarray(feats, np.float32).reshape(len(feats), -1)
    d = fw * fh
    assert feats.shape[1] == d
    idx = rng.choice(len(feats), n_filters, replace=len(feats) < n_filters)
    sv = (0.85 * feats[idx] + 0.15 * feats.mean(0, keepdims=True)).astype(np.float32)
    if gamma is None:
        d2 = ((feats[:200, None, :].astype(np.float64) - sv[None, :8].astype(np.float64)) ** 2).sum(-1)
        gamma = float(np.float32(1.0 / np.median(d2)))
    coeff = np.zeros(n_filters * (n_filters + 1) // 2, np.float32)
    for k in range(n_filters):
        coeff[k * (k + 1) // 2: k * (k + 1) // 2 + k + 1] = rng.normal(0, 1, k + 1).astype(np.float32)
    bias = np.float32(0.1)
    X = feats.astype(np.float64)
    if kernel == 2:
        ssd = np.stack([((feats - sv[k]) ** 2).sum(1, dtype=np.float32) for k in range(n_filters)], 1)   # ~fp32 like the reference
        K = np.exp(-gamma * ssd.astype(np.float64))
        p0, p1, p2 = gamma, 0.0, 0.0
    elif kernel == 1:
        p0, p1, p2 = 1.0 / (d * 255.0 * 255.0), 0.5, 2.0
        K = (p0 * (X @ sv.astype(np.float64).T) + p1) ** 2
    elif kernel == 3:
        K = np.minimum(X[:, None, :], sv[None].astype(np.float64)).sum(-1)
        coeff /= np.float32(K.mean())
        p0 = p1 = p2 = 0.0
    else:
        K = X @ sv.astype(np.float64).T
        coeff /= np.float32(np.abs(K).mean())
        p0 = p1 = p2 = 0.0
    thr = np.full(n_filters, -1e30, np.float32)
    alive = np.ones(len(feats), bool)
    dist = np.full(len(feats), -float(bias))
    for k in range(n_filters):
        dist = dist + float(coeff[k * (k + 1) // 2 + k]) * K[:, k]
        vals = dist[alive]
        if alive.sum() * pass_rate >= min_survivors:
            t = np.quantile(vals, 1.0 - pass_rate)
            # keep the threshold away from any sample (fp64 exp differs by an ulp between libm and the device)
            gaps = np.sort(vals)
            j = np.searchsorted(gaps, t)
            j = min(max(j, 1), len(gaps) - 1)
            thr[k] = np.float32(0.5 * (gaps[j - 1] + gaps[j]))
            alive &= dist >= thr[k]
    return dict(kernel=kernel, p0=p0, p1=p1, p2=p2, filter_w=fw, filter_h=fh, sv=sv, coeff=coeff, thresholds=thr, bias=bias, num_used=0,
                logistic_a=0.2, logistic_b=-1.5)


def save_rvm(path, m):
    """Binary FDRVM1 file (RvmClassifier::loadFromFile of the host layer): the reference's Matlab .mat models are absent."""
    import struct
    sv = np.ascontiguousarray(m["sv"], np.float32)
    with open(path, "wb") as f:
        f.write(b"FDRVM1\0\0")
        f.write(struct.pack("<5i", int(m["kernel"]), int(m["filter_w"]), int(m["filter_h"]), sv.shape[0], int(m.get("num_used", 0))))
        f.write(struct.pack("<3d", float(m.get("p0", 0)), float(m.get("p1", 0)), float(m.get("p2", 0))))
        f.write(struct.pack("<f", float(m["bias"])))
        f.write(sv.tobytes())
        f.write(np.ascontiguousarray(m["coeff"], np.float32).tobytes())
        f.write(np.ascontiguousarray(m["thresholds"], np.float32).tobytes())


def save_svm_text(path, m, rows=None, cols=None):
    """Text format of SvmClassifier::store + 'Logistic a b' (SvmClassifier.cpp:68-107, ProbabilisticSvmClassifier.cpp:65-68)."""
    sv = np.asarray(m["sv"])
    n, d = sv.shape
    rows = rows or 1
    cols = cols or d // rows
    depth = 0 if sv.dtype == np.uint8 else 5
    kname = {0: "Linear", 1: "Polynomial %d %.17g %.17g" % (int(m.get("p2", 2)), m.get("p1", 0.0), m.get("p0", 1.0)),
             2: "RBF %.17g" % m.get("p0", 0.0), 3: "HIK"}[int(m["kernel"])]
    with open(path, "w") as f:
        f.write("Kernel %s\n" % kname)
        f.write("Bias %.9g\n" % float(m["bias"]))
        f.write("Coefficients %d\n" % n)
        for c in np.asarray(m["coeff"], np.float32):
            f.write("%.9g\n" % c)
        f.write("SupportVectors %d %d %d 1 %d\n" % (n, rows, cols, depth))
        for s in sv:
            f.write(" ".join(("%d" % v) if depth == 0 else ("%.9g" % v) for v in s) + "\n")
        f.write("Logistic %.17g %.17g\n" % (m.get("logistic_a", 0.00556), m.get("logistic_b", -2.95)))


def save_pnm(path, img):
    img = np.ascontiguousarray(img, np.uint8)
    with open(path, "wb") as f:
        if img.ndim == 2:
            f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
            f.write(img.tobytes())
        else:  # BGR in memory -> RGB in the file
            f.write(b"P6\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
            f.write(img[..., ::-1].tobytes())


def save_sdm_text(path, m):
    """Text format of SdmLandmarkModel::save (SdmLandmarkModel.cpp:98-128): vlhog-uoctti descriptors, adaptive (empty parameter line) or
    with `numCells n cellSize c numBins b` per step when the model carries desc_params (SdmLandmarkModel.cpp:188-204)."""
    dp = None if m.get("desc_params") is None else np.asarray(m["desc_params"], np.int64).reshape(-1, 3)
    L = int(m["L"])
    with open(path, "w") as f:
        f.write("# synthetic SDM model\n")
        f.write("numLandmarks %d\n" % L)
        for i in range(L):
            f.write("lm%d\n" % i)
        for v in np.asarray(m["mean"], np.float32):
            f.write("%.9g\n" % v)
        f.write("numCascadeSteps %d\n" % int(m["S"]))
        for s, R in enumerate(m["R"]):
            f.write("cascadeStep %d rows %d cols %d\n" % (s, R.shape[0], R.shape[1]))
            f.write("descriptorType vlhog-uoctti\n")
            f.write("descriptorPostprocessing none\n")
            f.write("descriptorParameters \n" if dp is None else "descriptorParameters numCells %d cellSize %d numBins %d\n" % tuple(dp[s]))
            for row in np.asarray(R, np.float32):
                f.write(" ".join("%.9g" % v for v in row) + " \n")
