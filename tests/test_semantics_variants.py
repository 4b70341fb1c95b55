"""The [OpenCV-recall] semantics table (oracle/cvprims.h Semantics, msorb_extractor_set_semantics): Gaussian taps, resize
rounding variant, fastAtan2 contraction, and the contraction of the rotated rBRIEF tap (brief_tap: the one float expression of
ORBextractor.cc itself whose rounding the reference's compiler decides, :117-119 built -O3 -march=native).  Every variant of the
ORACLE is checked against a definition-level numpy restatement here (CPU); every variant of the KERNELS against the oracle in test_semantics_variants_gpu.  A pin mismatch on real OpenCV
(tools/pin_opencv.py) then is a switch, not a rewrite."""
import importlib.util
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

from msorb import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

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


# ---- brief_tap: cvRound(x*b + y*a), cvRound(x*a - y*b) under the three contractions a build of the reference can have -------------
def _kit():
    spec = importlib.util.spec_from_file_location("pin_opencv", os.path.join(ROOT, "tools", "pin_opencv.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _pattern_points():
    txt = re.sub(r"//.*", "", open(os.path.join(ROOT, "ms-slam_amd", "csrc", "orb_pattern.inc")).read())
    return np.array([int(x) for x in re.findall(r"-?\d+", txt)], np.int64).reshape(512, 2)


def _probe_cases():
    rows = []
    for line in open(os.path.join(ROOT, "tools", "probe_brief_tap_cases.inc")):
        m = re.match(r"\{0x([0-9a-f]+)u, 0x([0-9a-f]+)u, (\d+)\},\s*//\s*(\d)", line)
        if m:
            rows.append((int(m.group(1), 16), int(m.group(2), 16), int(m.group(3)), int(m.group(4))))
    return rows


def _tap_definition(mode, x, y, a, b):
    """One tap from first principles with Python's exact rationals: every product and sum exact, one rounding to float32 where the
    convention rounds.  (row, col)."""
    from fractions import Fraction as Fr

    def rnd32(q):
        # correctly rounded float32 of an exact rational: compare against the two neighbouring float32 values
        d = np.float32(float(q))
        cands = {float(d), float(np.nextafter(d, np.float32(np.inf))), float(np.nextafter(d, np.float32(-np.inf)))}
        best = sorted(cands, key=lambda v: (abs(Fr(v) - q), int(np.float32(v).view(np.uint32)) & 1))[0]
        return Fr(best)
    X, Y, A, B = Fr(int(x)), Fr(int(y)), Fr(float(a)), Fr(float(b))
    if mode == 0:
        r, c = rnd32(X * B + rnd32(Y * A)), rnd32(X * A - rnd32(Y * B))
    elif mode == 1:
        r, c = rnd32(Y * A + rnd32(X * B)), rnd32(rnd32(X * A) - Y * B)
    else:
        r, c = rnd32(rnd32(X * B) + rnd32(Y * A)), rnd32(rnd32(X * A) - rnd32(Y * B))

    def cv_round(q):   # round half to even
        fl = q.numerator // q.denominator
        fr = q - fl
        return fl + (1 if fr > Fr(1, 2) or (fr == Fr(1, 2) and fl % 2) else 0)
    return cv_round(r), cv_round(c)


def test_brief_tap_three_conventions_agree_across_oracle_kit_and_definition(oracle):
    """oracle/orb_extractor_oracle.cc rotated_tap == tools/pin_opencv.py rotated_taps == exact rational arithmetic, on the probe's
    discriminating inputs (tools/probe_brief_tap_cases.inc: pattern points and angles on which the conventions sample different
    pixels) and on random ones; and the table is what its comments say it is."""
    kit, pts, cases = _kit(), _pattern_points(), _probe_cases()
    assert len(cases) >= 60 and {c[3] for c in cases} == {1, 2, 3}
    rng = np.random.default_rng(3)
    rand = [(int(np.float32(np.cos(t)).view(np.uint32)), int(np.float32(np.sin(t)).view(np.uint32)), int(p), 0)
            for t, p in zip(rng.uniform(0, 2 * np.pi, 300), rng.integers(0, 512, 300))]
    for a_bits, b_bits, p, mask in cases + rand:
        a, b = np.array([a_bits, b_bits], np.uint32).view(np.float32)
        x, y = pts[p]
        got = [oracle.rotated_tap(m, x, y, a, b) for m in range(3)]
        for m in range(3):
            rr, qq = kit.rotated_taps(np.array([x]), np.array([y]), a, b, m)
            assert got[m] == (int(rr[0]), int(qq[0])) == _tap_definition(m, x, y, a, b), (hex(a_bits), hex(b_bits), p, m)
        if mask:    # bit 0: convention 1 samples another pixel than convention 0; bit 1: convention 2 does
            assert (got[1] != got[0]) == bool(mask & 1) and (got[2] != got[0]) == bool(mask & 2)
            assert max(abs(got[m][0] - got[0][0]) + abs(got[m][1] - got[0][1]) for m in (1, 2)) == 1      # the neighbouring pixel, never further


def test_brief_tap_flip_count_sweep(oracle):
    """SURVEY section 7 "hard parts": how many (pattern point, angle) pairs change their pixel with the compiler's contraction?  All
    512 pattern points x every angle fastAtan2 returns on the integer moment lattice |m01|, |m10| <= 300 and on 300 000 hashed
    moments of a real patch's magnitude.  (The full sweep — lattice +-1000, 4 M moments, 3.1 G pairs — is tools/brief_tap_sweep.py,
    result in profiles/round6_brief_tap_sweep.json: 2.0e-7 of the pairs, 3.8e-5 of the angles, 0.08 keypoints per 2000-keypoint frame.)"""
    g = np.arange(-300, 301, dtype=np.float32)
    yy, xx = np.meshgrid(g, g, indexing="ij")
    lattice = np.unique(oracle.fast_atan2_n(yy, xx))
    rng = np.random.default_rng(1)
    m = rng.integers(-3_000_000, 3_000_001, (2, 300_000)).astype(np.float32)
    moments = np.unique(oracle.fast_atan2_n(m[0], m[1]))
    assert len(lattice) > 200_000 and len(moments) > 290_000 and lattice.min() >= 0 and lattice.max() < 360
    total = {"pairs": 0, "angles": 0, 1: 0, 2: 0, "a1": 0, "a2": 0}
    for angles in (lattice, moments):
        r = oracle.brief_tap_sweep(angles)
        assert r["pairs"] == 512 * len(angles)
        total["pairs"] += r["pairs"]
        total["angles"] += r["angles"]
        for k in (1, 2):
            total[k] += r["flips_vs0"][k]
            total[f"a{k}"] += r["angles_vs0"][k]
        # the examples are real: re-evaluated one by one through the single-tap entry
        pts = _pattern_points()
        for a_bits, p, mask in r["examples"][:50]:
            ang = np.array([a_bits], np.uint32).view(np.float32)[0]
            a, b = oracle.cos_sin(float(ang))
            got = [oracle.rotated_tap(mm, pts[p][0], pts[p][1], a, b) for mm in range(3)]
            assert (got[1] != got[0]) == bool(mask & 1) and (got[2] != got[0]) == bool(mask & 2)
    print("brief_tap sweep:", total)
    # exposure: present (the conventions ARE different functions) and tiny (a few pairs in 10^7; a few angles in 10^5)
    assert total[1] > 0 and total[2] > 0
    assert total[1] / total["pairs"] < 2e-6 and total[2] / total["pairs"] < 2e-6
    assert total["a1"] / total["angles"] < 3e-4 and total["a2"] / total["angles"] < 3e-4


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_probe_brief_tap_reports_the_contraction_of_the_build(tmp_path):
    """tools/probe_brief_tap.cc — what a maintainer compiles with the reference's own flags to learn brief_tap — answers 0 for an
    FMA target under g++'s default contraction (what -O3 -march=native is on any x86-64 since 2013: CMakeLists.txt:10-13) and 2 for
    -ffp-contract=off and for a target without FMA, in both of its shapes (expression alone, inside the descriptor loop)."""
    src = os.path.join(ROOT, "tools", "probe_brief_tap.cc")
    for flags, want in ((["-O3", "-march=x86-64-v3"], 0), (["-O3", "-march=x86-64-v3", "-ffp-contract=off"], 2), (["-O3", "-march=x86-64"], 2), (["-O2", "-mfma"], 0)):
        exe = str(tmp_path / "probe")
        subprocess.check_call(["g++", *flags, "-std=c++17", src, "-o", exe])
        r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0 and f"brief_tap = {want} " in r.stdout, (flags, r.stdout)
        assert r.stdout.count(" of 96 inputs") + r.stdout.count(" of 9") >= 2
    kit = _kit()
    assert kit.probe_brief_tap(flags=("-O3", "-march=x86-64-v3")) == 0
    assert kit.probe_brief_tap(flags=("-O3", "-march=x86-64")) == 2


def test_oracle_descriptor_follows_brief_tap(oracle, restore):
    """Whole-extractor level: keypoints never depend on the convention; the descriptors do on the images listed here (found by
    scanning synthetic frames with the oracle: about one frame in forty carries a descriptor bit that depends on it)."""
    cfg = synth.KITTI
    ref = oracle.OracleExtractor(2000, 1.2, 8, 20, 7)
    for mode, seed in BRIEF_TAP_SEEDS:
        img = synth.image(seed, cfg["rows"], cfg["cols"])
        oracle.set_semantics()
        _, k0, d0 = ref(img)
        oracle.set_semantics(brief_tap=mode)
        _, k1, d1 = ref(img)
        assert np.array_equal(k0.view(np.uint8), k1.view(np.uint8))
        diff = np.flatnonzero((d0 != d1).any(1))
        assert 1 <= len(diff) <= 3, (mode, seed, len(diff))
        assert all(int(np.unpackbits(d0[i] ^ d1[i]).sum()) <= 2 for i in diff)      # a moved tap changes the bits of its own tests only
    with pytest.raises(ValueError):
        oracle.set_semantics(brief_tap=3)


# (brief_tap, synth.image seed at KITTI size, 2000 features): frames on which that convention changes at least one descriptor
BRIEF_TAP_SEEDS = [(1, 1047), (1, 1122), (1, 1398), (2, 1122), (2, 1398)]   # 5 hits in 400 frames x 2 conventions (/tmp scan, 2000 features)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 2])
def test_brief_tap_variants_gpu(msorb_mod, oracle, restore, mode):
    """describe_kernel<kTap> under msorb_extractor_set_semantics(brief_tap) vs the oracle under the same convention, per frame and
    batched, on frames whose descriptors DO depend on it (so the variant is exercised, not merely selected), and back."""
    import torch
    cfg = synth.KITTI
    seeds = [s for m, s in BRIEF_TAP_SEEDS if m == mode]
    assert seeds
    imgs = [synth.image(s, cfg["rows"], cfg["cols"]) for s in seeds] + [synth.image(90, cfg["rows"], cfg["cols"])]
    ex = msorb_mod.ORBextractor(2000, 1.2, 8, 20, 7)
    ref = oracle.OracleExtractor(2000, 1.2, 8, 20, 7)
    try:
        base = [ex(im) for im in imgs]
        ex.set_semantics(brief_tap=mode)
        oracle.set_semantics(brief_tap=mode)
        want = [ref(im) for im in imgs]
        changed = 0
        for im, (m0, k0, d0), (rmono, rkps, rdesc) in zip(imgs, base, want):
            mono, kps, desc = ex(im)
            assert mono == rmono and np.array_equal(kps.view(np.uint8), rkps.view(np.uint8)) and np.array_equal(desc, rdesc)
            assert np.array_equal(kps.view(np.uint8), k0.view(np.uint8))            # keypoints do not depend on the convention
            changed += int((desc != d0).any(1).sum())
        assert changed >= len(seeds), "the convention changed no descriptor: not exercised"
        d_img = torch.from_numpy(np.stack((imgs * 16)[:16 + len(imgs)])).cuda()      # >= 16 images: the batch kernels
        counts, monos, d_kps, d_desc = ex.extract_batch(d_img)
        for i in range(d_img.shape[0]):
            rmono, rkps, rdesc = want[i % len(imgs)]
            assert np.array_equal(d_desc[i, :counts[i]].cpu().numpy(), rdesc)
        with pytest.raises(msorb_mod.MsorbError):
            ex.set_semantics(brief_tap=3)
        ex.set_semantics()
        oracle.set_semantics()
        for im, (m0, k0, d0) in zip(imgs, base):
            mono, kps, desc = ex(im)
            assert np.array_equal(desc, d0)
    finally:
        ex.close()
