"""Stand-alone ImageFilter::applyTo(Mat) forms of the filters the detection kernels fuse (VERDICT r01 "What's missing" 5): every one
through the C ABI against the oracle on the same inputs; integer work bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ksize", [1, 3, 5, 7, -1])
def test_gradient_and_binning_images(oracle, capi, ctx, frame640, ksize):
    """GradientFilter(ksize) -> CV_8UC2, GradientBinningFilter(bins, signed, interpolate) -> CV_8UC2 / CV_8UC4: bit-exact, and the
    two stand-alone filters chained equal the fused layer filter of the pyramid."""
    gray = oracle.bgr2gray(frame640)[:203, :317].copy()   # odd sizes
    go = oracle.gradient_filter(gray, ksize)
    gg = capi.gradient_image(ctx, gray, ksize)
    assert gg.shape == go.shape and np.array_equal(gg, go)
    for bins, sg, ip in ((9, False, False), (18, True, False), (9, False, True), (12, True, True)):
        bo = oracle.gradient_binning(go, bins, sg, ip)
        bg = capi.gradient_binning_image(ctx, gg, bins, sg, ip)
        assert bg.shape == bo.shape and np.array_equal(bg, bo), (bins, sg, ip)
    # one-layer pyramid at scale 1 with the fused layer filter == the two stand-alone filters
    pg = capi.Pyramid(ctx, octave_layers=1, min_scale=1.0, max_scale=1.0)
    pg.set_layer_filter(capi.FD_LAYER_GRADBIN, bins=9, grad_kernel=ksize)
    pg.update(gray)
    assert np.array_equal(pg.layer(0), capi.gradient_binning_image(ctx, gg, 9))
    pg.close()
    # GradientFilter's blurKernelSize (cv::blur before the derivatives), odd and even box sizes, also on a tiny image where the
    # reflected border is wider than the image
    for blur in (3, 4, 5):
        assert np.array_equal(capi.gradient_image(ctx, gray, ksize, blur), oracle.gradient_filter(gray, ksize, blur)), blur
    tiny = gray[:3, :5].copy()
    assert np.array_equal(capi.gradient_image(ctx, tiny, ksize, 3), oracle.gradient_filter(tiny, ksize, 3))
    assert np.array_equal(capi.gradient_image(ctx, tiny, ksize), oracle.gradient_filter(tiny, ksize))


@pytest.mark.parametrize("lbp_type", [0, 1, 2, 3])
def test_lbp_image(oracle, capi, ctx, frame640, lbp_type):
    gray = oracle.bgr2gray(frame640)[:131, :97].copy()
    assert np.array_equal(capi.lbp_image(ctx, gray, lbp_type), oracle.lbp(gray, lbp_type))


def test_histogram_patch_filters_per_mat(oracle, capi, ctx, frame640):
    """HogFilter / SpatialHistogramFilter / PyramidHogFilter / SpatialPyramidHistogramFilter::applyTo(Mat) on single bin-image
    patches (fd_hist_patch_batch) equal the oracle's filters; batches of patches equal the per-patch results."""
    gray = oracle.bgr2gray(frame640)
    grad = oracle.gradient_filter(gray[100:300, 100:400].copy(), 1)
    bin2 = oracle.gradient_binning(grad, 9, False, False)
    bin4 = oracle.gradient_binning(grad, 9, False, True)
    lbp = oracle.lbp(gray[100:300, 100:400].copy(), 1)
    rng = np.random.default_rng(4)

    def patches(img, n, pw, ph):
        ys, xs = rng.integers(0, img.shape[0] - ph, n), rng.integers(0, img.shape[1] - pw, n)
        return np.stack([img[y:y + ph, x:x + pw] for y, x in zip(ys, xs)])
    cases = [
        (bin2, 20, 20, dict(kind=0, bins=9, cell=5, block=2), lambda p: oracle.hog_filter(p, 9, 5, 2)),
        (bin2, 20, 20, dict(kind=0, bins=9, cell=5, block=2, interpolate=True), lambda p: oracle.hog_filter(p, 9, 5, 2, interpolate=True)),
        (bin4, 24, 16, dict(kind=0, bins=9, cell=4, block=2), lambda p: oracle.hog_filter(p, 9, 4, 2)),
        (lbp, 20, 20, dict(kind=1, bins=59, cell=5, block=2, normalization=2), lambda p: oracle.spatial_histogram(p, 59, 5, 2, normalization=2)),
        (bin2, 32, 32, dict(kind=2, bins=9, levels=3), lambda p: oracle.pyramid_hog(p, 9, 3)),
        (lbp, 16, 16, dict(kind=3, bins=59, levels=2, normalization=1), lambda p: oracle.spatial_pyramid_histogram(p, 59, 2, normalization=1)),
    ]
    for img, pw, ph, hk, ofn in cases:
        P = patches(img, 37, pw, ph)
        hp = capi.hist_params(pw=pw, ph=ph, sx=1, sy=1, **hk)
        fo = np.stack([np.asarray(ofn(np.ascontiguousarray(p)), np.float32).ravel() for p in P])
        fg = capi.hist_patch_batch(ctx, P, hp)
        assert fg.shape == fo.shape, hk
        assert np.array_equal(fg, fo), hk
        one = capi.hist_patch_batch(ctx, P[5:6], hp)   # the per-Mat form: n = 1
        assert np.array_equal(one[0], fo[5])
    with pytest.raises(capi.FdError):   # HOG on a one-channel (LBP) image
        capi.hist_patch_batch(ctx, patches(lbp, 2, 20, 20), capi.hist_params(kind=0, bins=9, cell=5, block=2))


def test_whi_chain_single_filters(oracle, capi, ctx, frame640):
    """WhiteningFilter, ConversionFilter and UnitNormFilter one by one (their per-Mat applyTo) and composed: the composition equals
    the fused chain kernel bit for bit."""
    gray = oracle.bgr2gray(frame640)
    rng = np.random.default_rng(9)
    P = np.stack([gray[y:y + 20, x:x + 20] for y, x in zip(rng.integers(0, 400, 24), rng.integers(0, 600, 24))])
    wo = np.stack([oracle.whitening(np.ascontiguousarray(p)) for p in P])
    wg = capi.whitening_batch(ctx, P)
    assert np.array_equal(wg, wo)
    eq = capi.equalize_hist_batch(ctx, wg)
    cv = capi.convert_batch(ctx, eq, 1.0 / 127.5, -1.0, to_f32=True)
    assert np.array_equal(cv, eq.astype(np.float32) * np.float32(1.0 / 127.5) + np.float32(-1.0))
    un = capi.unit_norm_batch(ctx, cv.reshape(len(cv), -1), 4).reshape(cv.shape)
    assert np.array_equal(un, capi.whi_batch(ctx, P))               # == the fused chain
    assert np.array_equal(un, np.stack([oracle.whi(np.ascontiguousarray(p)) for p in P]).reshape(un.shape))
    # convertTo to CV_8U saturates and rounds half to even; f32 source
    f = np.array([[-3.2, 0.5, 1.5, 2.5, 254.5, 255.5, 300.0]], np.float32)
    assert capi.convert_batch(ctx, f, 1.0, 0.0, to_f32=False).tolist() == [[0, 0, 2, 2, 254, 255, 255]]
    # L1 / INF norms
    v = rng.normal(0, 1, (5, 77)).astype(np.float32)
    for nt, fn in ((2, lambda a: np.abs(a.astype(np.float64)).sum()), (1, lambda a: np.abs(a.astype(np.float64)).max())):
        g = capi.unit_norm_batch(ctx, v, nt)
        exp = np.stack([a * np.float32(1.0 / (fn(a) + np.float64(np.float32(1e-4)))) for a in v])
        assert np.array_equal(g, exp)
