"""The [OpenCV-recall] semantics table (oracle/cvprims.h Semantics, msorb_extractor_set_semantics): Gaussian taps, resize
rounding variant, fastAtan2 contraction.  Every variant of the ORACLE is checked against a definition-level numpy restatement
here (CPU); every variant of the KERNELS against the oracle in test_semantics_variants_gpu.  A pin mismatch on real OpenCV
(tools/pin_opencv.py) then is a switch, not a rewrite."""
import numpy as np
import pytest

from msorb import synth

ALT_TAPS = [18, 34, 49, 55, 49, 34, 18]   # a float-kernel build's rounding: sum 257


def np_resize(src, drows, dcols, single_stage):
    def taps(dn, sn):
        scale = 1.0 / (float(dn) / sn)
        f = ((np.arange(dn) + 0.5) * scale - 0.5).astype(np.float32)
        i = np.floor(f).astype(np.int64)
        f = (f - i).astype(np.float32)
        f[i < 0] = 0
        i[i < 0] = 0
        hi = i >= sn - 1
        f[hi] = 0
        i[hi] = sn - 1
        return i, np.minimum(i + 1, sn - 1), np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64), np.rint(f * np.float32(2048)).astype(np.int64)
    x0, x1, ca0, ca1 = taps(dcols, src.shape[1])
    y0, y1, cb0, cb1 = taps(drows, src.shape[0])
    s = src.astype(np.int64)
    H = s[:, x0] * ca0 + s[:, x1] * ca1
    if single_stage:
        return ((H[y0] * cb0[:, None] + H[y1] * cb1[:, None] + (1 << 21)) >> 22).astype(np.uint8)
    return ((((cb0[:, None] * (H[y0] >> 4)) >> 16) + ((cb1[:, None] * (H[y1] >> 4)) >> 16) + 2) >> 2).astype(np.uint8)


def np_gauss(src, k):
    k = np.asarray(k, np.int64)

    def refl(n):
        p = np.arange(-3, n + 3)
        p = np.where(p < 0, -p, p)
        return np.where(p >= n, 2 * (n - 1) - p, p)
    s = src.astype(np.int64)
    h = sum(k[t] * s[:, refl(src.shape[1])[t:t + src.shape[1]]] for t in range(7))
    v = sum(k[t] * h[refl(src.shape[0])[t:t + src.shape[0]], :] for t in range(7))
    return np.minimum((v + 32768) >> 16, 255).astype(np.uint8)   # saturate_cast<uchar>


def np_atan2(y, x, fma):
    """cv::fastAtan2's polynomial with every operation a correctly rounded float32 operation; fma: each Horner step a*b+c rounded
    once (evaluated in float64, which holds a float32 product + float32 addend exactly enough: 24+24 bit products are exact in 53)."""
    f = np.float32
    scale = f(180 / 3.1415926535897932384626433832795)
    p1, p3, p5, p7 = [f(f(c) * scale) for c in (0.9997878412794807, -0.3258083974640975, 0.1555786518463281, -0.04432655554792128)]
    eps = f(2.2204460492503131e-16)
    ax, ay = abs(f(x)), abs(f(y))

    def step(a, b, c):
        if fma:
            return f(np.float64(a) * np.float64(b) + np.float64(c))
        return f(f(a * b) + c)

    def poly(c, c2):
        return f(step(step(step(p7, c2, p5), c2, p3), c2, p1) * c)
    if ax >= ay:
        c = f(ay / f(ax + eps))
        a = poly(c, f(c * c))
    else:
        c = f(ax / f(ay + eps))
        a = f(f(90.0) - poly(c, f(c * c)))
    if x < 0:
        a = f(f(180.0) - a)
    if y < 0:
        a = f(f(360.0) - a)
    return a


@pytest.fixture
def restore(oracle):
    yield
    oracle.set_semantics()


def test_oracle_variants_against_definition_level_numpy(oracle, restore):
    rng = np.random.default_rng(7)
    img = synth.image(13, 110, 161)
    noisy = np.clip(img.astype(np.int64) + rng.integers(-40, 41, img.shape), 0, 255).astype(np.uint8)
    moments = [(int(a), int(b)) for a, b in rng.integers(-200000, 200000, (400, 2))] + [(0, 5), (5, 0), (-5, 0), (0, -5), (7, 7), (0, 0)]
    n_diff = {"resize": 0, "gauss": 0, "atan2": 0}
    for taps, single, fma in ((None, False, False), (ALT_TAPS, False, False), (None, True, False), (None, False, True), (ALT_TAPS, True, True)):
        oracle.set_semantics(taps, single, fma)
        k = taps or [18, 34, 48, 56, 48, 34, 18]
        for im in (img, noisy):
            for dr, dc in ((92, 134), (109, 160)):
                assert np.array_equal(oracle.resize_linear_u8(im, dr, dc), np_resize(im, dr, dc, single)), (taps, single, fma)
            assert np.array_equal(oracle.gaussian7(im), np_gauss(im, k)), (taps, single, fma)
        for m01, m10 in moments:
            got = np.float32(oracle.fast_atan2(float(m01), float(m10)))
            want = np_atan2(float(m01), float(m10), fma)
            assert got.tobytes() == np.float32(want).tobytes(), (m01, m10, fma)
    # the variants are real variants: each changes some output
    oracle.set_semantics()
    base = (oracle.resize_linear_u8(noisy, 92, 134), oracle.gaussian7(noisy), [oracle.fast_atan2(float(a), float(b)) for a, b in moments])
    oracle.set_semantics(ALT_TAPS, True, True)
    alt = (oracle.resize_linear_u8(noisy, 92, 134), oracle.gaussian7(noisy), [oracle.fast_atan2(float(a), float(b)) for a, b in moments])
    assert (base[0] != alt[0]).any() and (base[1] != alt[1]).any() and base[2] != alt[2]
    assert int(np.abs(base[0].astype(int) - alt[0].astype(int)).max()) <= 1     # single- vs two-stage rounding: at most one grey level
    with pytest.raises(ValueError):
        oracle.set_semantics([40, 40, 40, 40, 40, 40, 40])                      # 16-bit horizontal sums would overflow


@pytest.mark.gpu
@pytest.mark.parametrize("taps,single,fma", [(ALT_TAPS, False, False), (None, True, False), (None, False, True), (ALT_TAPS, True, True),
                                             ([16, 32, 48, 64, 48, 32, 16], False, False)])
def test_semantics_variants_gpu(msorb_mod, oracle, restore, taps, single, fma):
    """Kernels under msorb_extractor_set_semantics vs the oracle under the same table: pyramid levels, blurred levels, keypoints
    (angle bit patterns) and descriptors, per frame and batched, and back to the defaults afterwards."""
    import torch
    cfg = synth.KITTI
    imgs = [synth.image(70 + i, cfg["rows"], cfg["cols"]) for i in range(3)]
    ex = msorb_mod.ORBextractor(1500, 1.2, 8, 20, 7)
    ref = oracle.OracleExtractor(1500, 1.2, 8, 20, 7)
    try:
        mono0, k0, d0 = ex(imgs[0])                       # defaults first
        ex.set_semantics(taps, single, fma)
        oracle.set_semantics(taps, single, fma)
        changed = False
        for im in imgs:
            mono, kps, desc = ex(im)
            rmono, rkps, rdesc = ref(im)
            assert mono == rmono and len(kps) == len(rkps) > 500
            assert np.array_equal(kps.view(np.uint8), rkps.view(np.uint8)), "keypoints differ from the oracle under this variant"
            assert np.array_equal(desc, rdesc)
            for l in (0, 1, 4, 7):
                assert np.array_equal(ex.debug_level(0, l), ref.level(l))
                assert np.array_equal(ex.debug_level(0, l, blurred=True), ref.level(l, blurred=True))
            if im is imgs[0]:
                changed = len(kps) != len(k0) or not np.array_equal(desc, d0) or not np.array_equal(kps.view(np.uint8), k0.view(np.uint8))
        assert changed, "the variant did not change anything: not exercised"
        # batched (the batch kernels must route to the variant-capable forms too)
        d_img = torch.from_numpy(np.stack(imgs * 6)).cuda()          # 18 images: the batch kernel path (>= 16)
        counts, monos, d_kps, d_desc = ex.extract_batch(d_img)
        for i in (0, 1, 2, 17):
            rmono, rkps, rdesc = ref(imgs[i % 3])
            got = msorb_mod.keypoints_from_device(d_kps, counts)[i]
            assert np.array_equal(got.view(np.uint8), rkps.view(np.uint8)) and np.array_equal(d_desc[i, :counts[i]].cpu().numpy(), rdesc)
        ex.set_semantics()
        oracle.set_semantics()
        mono1, k1, d1 = ex(imgs[0])
        assert np.array_equal(k1.view(np.uint8), k0.view(np.uint8)) and np.array_equal(d1, d0)
    finally:
        ex.close()
