"""Consumer of the OpenCV pin kit (tools/pin_opencv.py): compares oracle/ — the CPU restatement every GPU parity test
is checked against — with outputs of the REAL OpenCV primitives the reference calls
(ORBextractor.cc:102 fastAtan2, :826/:845 FAST, :1133 GaussianBlur, :1183 resize).

* `test_pins_against_real_opencv` runs when tests/golden/opencv_pins/ holds fixtures produced by one run of the kit on
  a machine with cv2 (this image has none: the test then SKIPS and parity stays "unpinned" for those four primitives).
* `test_pins_live_cv2` does the same in-process wherever cv2 happens to be importable.
* `test_pin_kit_plumbing` proves the kit and this consumer work end to end without cv2, by handing the kit a stand-in
  "cv" object backed by the oracle — so the first real run cannot fail for plumbing reasons.
"""
import hashlib
import importlib.util
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PINS = os.path.join(ROOT, "tests", "golden", "opencv_pins")


def _kit():
    spec = importlib.util.spec_from_file_location("pin_opencv", os.path.join(ROOT, "tools", "pin_opencv.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def compare(oracle, pins_dir):
    """-> list of human-readable mismatches between the oracle and the fixtures in pins_dir (empty = pinned)."""
    kit = _kit()
    meta = json.load(open(os.path.join(pins_dir, "meta.json")))
    bad = []
    # A.3 resize: 11-bit fixed-point bilinear, coefficient rounding, edge clamp
    z = np.load(os.path.join(pins_dir, "resize_linear.npz"))
    for i, (name, sr, sc, dr, dc) in enumerate(kit.RESIZE_CASES):
        src = kit.pin_image(100 + i, sr, sc)
        assert kit.sha(src) == meta["resize_inputs_sha256"][name], f"input of {name} differs from the generating run"
        got = oracle.resize_linear_u8(src, dr, dc)
        if not np.array_equal(got, z[name]):
            d = np.argwhere(got != z[name])
            bad.append(f"resize {name}: {len(d)} of {got.size} pixels differ, first at (y,x)={tuple(d[0])} "
                       f"oracle {got[tuple(d[0])]} vs OpenCV {z[name][tuple(d[0])]}")
    # A.5 blur: Q8.8 kernel [18,34,48,56,48,34,18], reflect-101, (x + 2^15) >> 16
    z = np.load(os.path.join(pins_dir, "gaussian7.npz"))
    for i, (name, r, c) in enumerate(kit.BLUR_CASES):
        src = kit.pin_image(200 + i, r, c)
        assert kit.sha(src) == meta["blur_inputs_sha256"][name]
        got = oracle.gaussian7(src)
        if not np.array_equal(got, z[name]):
            d = np.argwhere(got != z[name])
            bad.append(f"blur {name}: {len(d)} of {got.size} pixels differ, max |diff| "
                       f"{int(np.abs(got.astype(int) - z[name].astype(int)).max())}")
    # A.4 FAST: circle, strict 9-arc test, cornerScore, strict 3x3 NMS inside the call's ROI, scan order
    z = np.load(os.path.join(pins_dir, "fast9_nms.npz"))
    rois = kit.fast_inputs()
    assert kit.sha(np.concatenate([r.ravel() for r in rois])) == meta["fast_inputs_sha256"]
    for th in kit.FAST_THRESHOLDS:
        for i, roi in enumerate(rois):
            want = z[f"th{th}_roi{i}"].reshape(-1, 3)
            got = oracle.fast9_nms(roi, th).reshape(-1, 3)
            if got.shape != want.shape or not np.array_equal(got, want):
                bad.append(f"FAST th={th} roi {i}: oracle {len(got)} keypoints vs OpenCV {len(want)}"
                           + ("" if got.shape != want.shape else f", first differing row {int(np.argwhere((got != want).any(1))[0][0])}"))
    # A.6 fastAtan2: polynomial, octant fix-ups, float evaluation order
    z = np.load(os.path.join(pins_dir, "fast_atan2.npz"))
    y, x = kit.atan2_inputs()
    assert kit.sha(np.stack([y, x])) == meta["atan2_inputs_sha256"]
    got = np.array([oracle.fast_atan2(a, b) for a, b in zip(y, x)], np.float32).view(np.uint32)
    if not np.array_equal(got, z["bits"]):
        d = np.flatnonzero(got != z["bits"])
        bad.append(f"fastAtan2: {len(d)} of {len(got)} bit patterns differ, first (y,x)=({y[d[0]]},{x[d[0]]}) "
                   f"oracle {got[d[0]:d[0] + 1].view(np.float32)[0]!r} vs OpenCV {z['bits'][d[0]:d[0] + 1].view(np.float32)[0]!r}")
    return bad, meta


def select_semantics(oracle, pins_dir):
    """Which variant of the semantics table (oracle/cvprims.h `Semantics` == msorb_extractor_set_semantics) reproduces the
    fixtures?  Each of the three variable primitives is decided on its own: the Gaussian's Q8 taps (symmetric 7-tap kernels
    within +-2 of sigma 2's float taps, sum 255 .. 257), resize's vertical rounding (two-stage / single-stage), fastAtan2's
    contraction (separate multiply-add / FMA).  FAST has no variant.  -> dict(gauss_taps, resize_single_stage, atan2_fma,
    fast_ok, default) with None where NO variant matches (then parity is refuted, not merely unpinned)."""
    kit = _kit()
    meta = json.load(open(os.path.join(pins_dir, "meta.json")))
    out = {"cv2_version": meta.get("cv2_version")}
    try:
        # resize
        z = np.load(os.path.join(pins_dir, "resize_linear.npz"))
        out["resize_single_stage"] = None
        for single in (False, True):
            oracle.set_semantics(resize_single_stage=single)
            if all(np.array_equal(oracle.resize_linear_u8(kit.pin_image(100 + i, sr, sc), dr, dc), z[name])
                   for i, (name, sr, sc, dr, dc) in enumerate(kit.RESIZE_CASES)):
                out["resize_single_stage"] = single
                break
        # fastAtan2 (a sample decides, the full set confirms)
        z = np.load(os.path.join(pins_dir, "fast_atan2.npz"))
        y, x = kit.atan2_inputs()
        out["atan2_fma"] = None
        for fma in (False, True):
            oracle.set_semantics(atan2_fma=fma)
            sel = np.r_[0:len(y):53]
            if not np.array_equal(np.array([oracle.fast_atan2(a, b) for a, b in zip(y[sel], x[sel])], np.float32).view(np.uint32), z["bits"][sel]):
                continue
            if np.array_equal(np.array([oracle.fast_atan2(a, b) for a, b in zip(y, x)], np.float32).view(np.uint32), z["bits"]):
                out["atan2_fma"] = fma
                break
        # Gaussian taps: the small case first, every case to confirm
        z = np.load(os.path.join(pins_dir, "gaussian7.npz"))
        srcs = {name: kit.pin_image(200 + i, r, c) for i, (name, r, c) in enumerate(kit.BLUR_CASES)}
        small = min(kit.BLUR_CASES, key=lambda c: c[1] * c[2])[0]
        cands = [(18, 34, 48, 56)] + [(a, b, c, d) for a in range(16, 21) for b in range(32, 37) for c in range(46, 51) for d in range(53, 59)
                                      if 255 <= 2 * (a + b + c) + d <= 257 and (a, b, c, d) != (18, 34, 48, 56)]
        out["gauss_taps"] = None
        for a, b, c, d in cands:
            taps = [a, b, c, d, c, b, a]
            oracle.set_semantics(gauss_taps=taps)
            if not np.array_equal(oracle.gaussian7(srcs[small]), z[small]):
                continue
            if all(np.array_equal(oracle.gaussian7(srcs[n]), z[n]) for n in srcs):
                out["gauss_taps"] = taps
                break
    finally:
        oracle.set_semantics()
    z = np.load(os.path.join(pins_dir, "fast9_nms.npz"))
    rois = kit.fast_inputs()
    out["fast_ok"] = all(np.array_equal(oracle.fast9_nms(roi, th).reshape(-1, 3), z[f"th{th}_roi{i}"].reshape(-1, 3))
                         for th in kit.FAST_THRESHOLDS for i, roi in enumerate(rois))
    out["default"] = (out["gauss_taps"] == [18, 34, 48, 56, 48, 34, 18] and out["resize_single_stage"] is False and
                      out["atan2_fma"] is False and out["fast_ok"])
    return out


def test_pin_inputs_are_machine_independent():
    """pin_image / atan2_inputs are integer hashes: their bytes are constants of the kit (guards numpy drift)."""
    kit = _kit()
    assert hashlib.sha256(kit.pin_image(100, 376, 1241).tobytes()).hexdigest()[:16] == PIN_IMAGE_100_SHA16
    y, x = kit.atan2_inputs()
    assert kit.sha(np.stack([y, x]))[:16] == ATAN2_INPUTS_SHA16
    img = kit.pin_image(500, 57, 46)
    assert img.min() < 60 and img.max() > 170          # textured: FAST has work at both thresholds


def test_pins_against_real_opencv(oracle):
    if not os.path.exists(os.path.join(PINS, "meta.json")):
        pytest.skip("no OpenCV pins committed yet: run tools/pin_opencv.py on a machine with cv2 (README) — "
                    "until then the oracle is PARITY UNPINNED for resize / FAST / GaussianBlur / fastAtan2")
    bad, meta = compare(oracle, PINS)
    sel = select_semantics(oracle, PINS)
    with open(os.path.join(PINS, "selected_semantics.json"), "w") as f:   # what msorb_extractor_set_semantics must be given
        json.dump(sel, f, indent=1, sort_keys=True)
    print("semantics reproducing OpenCV", meta["cv2_version"], ":", sel)
    assert None not in sel.values() and sel["fast_ok"], f"NO variant of the semantics table reproduces OpenCV {meta['cv2_version']}: {sel}\n" + "\n".join(bad)
    assert not bad, (f"the DEFAULT semantics differ from OpenCV {meta['cv2_version']}; the variant {sel} reproduces it: make it the default "
                     "(oracle/cvprims.h Semantics, csrc/orb_device.h Semantics) or pass it to msorb_extractor_set_semantics\n" + "\n".join(bad))


def test_pins_live_cv2(oracle, tmp_path):
    cv2 = pytest.importorskip("cv2")
    kit = _kit()
    kit.generate(cv2, str(tmp_path))
    bad, meta = compare(oracle, str(tmp_path))
    assert not bad, f"oracle differs from OpenCV {meta['cv2_version']}:\n" + "\n".join(bad)


class _OracleAsCv:
    """Stand-in with the five cv2 entry points the kit calls, answered by the oracle (plumbing test only)."""
    __version__ = "oracle-stand-in"
    INTER_LINEAR = 1
    BORDER_REFLECT_101 = 4
    FAST_FEATURE_DETECTOR_TYPE_9_16 = 2

    def __init__(self, oracle):
        self.o = oracle

    def getBuildInformation(self):
        return "Version control: stand-in\n"

    def resize(self, src, dsize, interpolation):
        assert interpolation == self.INTER_LINEAR
        return self.o.resize_linear_u8(src, dsize[1], dsize[0])

    def GaussianBlur(self, src, ksize, sx, sy, borderType):
        assert ksize == (7, 7) and sx == 2 and sy == 2 and borderType == self.BORDER_REFLECT_101
        return self.o.gaussian7(src)

    def fastAtan2(self, y, x):
        return self.o.fast_atan2(y, x)

    def FastFeatureDetector_create(self, threshold, nonmaxSuppression, type):
        assert nonmaxSuppression and type == self.FAST_FEATURE_DETECTOR_TYPE_9_16
        o = self.o

        class KP:
            def __init__(self, r):
                self.pt, self.response = (float(r[0]), float(r[1])), float(r[2])

        class Det:
            def detect(self, roi, mask):
                return [KP(r) for r in o.fast9_nms(roi, threshold).reshape(-1, 3)]
        return Det()


def test_pin_kit_plumbing(oracle, tmp_path):
    kit = _kit()
    meta = kit.generate(_OracleAsCv(oracle), str(tmp_path))
    assert meta["cv2_version"] == "oracle-stand-in"
    for f in ("resize_linear.npz", "gaussian7.npz", "fast9_nms.npz", "fast_atan2.npz", "meta.json"):
        assert os.path.getsize(os.path.join(str(tmp_path), f)) > 0
    bad, _ = compare(oracle, str(tmp_path))
    assert bad == []
    # the comparison has teeth: a one-pixel / one-ulp / one-keypoint change in a fixture is reported
    z = dict(np.load(os.path.join(str(tmp_path), "gaussian7.npz")))
    z["tiny"] = z["tiny"].copy()
    z["tiny"][0, 0] ^= 1
    np.savez_compressed(os.path.join(str(tmp_path), "gaussian7.npz"), **z)
    z = dict(np.load(os.path.join(str(tmp_path), "fast_atan2.npz")))
    z["bits"] = z["bits"].copy()
    z["bits"][5] += 1
    np.savez_compressed(os.path.join(str(tmp_path), "fast_atan2.npz"), **z)
    bad, _ = compare(oracle, str(tmp_path))
    assert len(bad) == 2 and bad[0].startswith("blur tiny") and bad[1].startswith("fastAtan2")
    # FAST fixtures are not degenerate: keypoints at both thresholds, and the low-contrast cells only at th 7
    z = np.load(os.path.join(str(tmp_path), "fast9_nms.npz"))
    n20 = sum(len(z[f"th20_roi{i}"]) for i in range(kit.FAST_CELLS))
    n7 = sum(len(z[f"th7_roi{i}"]) for i in range(kit.FAST_CELLS))
    assert n20 > 200 and n7 > n20
    assert any(len(z[f"th20_roi{i}"]) == 0 and len(z[f"th7_roi{i}"]) > 0 for i in range(0, kit.FAST_CELLS, 6))


@pytest.mark.parametrize("variant", [dict(), dict(gauss_taps=[18, 34, 49, 55, 49, 34, 18], resize_single_stage=True, atan2_fma=True),
                                     dict(gauss_taps=[19, 34, 48, 54, 48, 34, 19], resize_single_stage=False, atan2_fma=True)])
def test_consumer_finds_the_semantics_variant_that_made_the_fixtures(oracle, tmp_path, variant):
    """The day real fixtures arrive, a disagreement with the defaults must come with its remedy: fixtures made by a stand-in
    "OpenCV" that runs a NON-default variant of the semantics table are recognised as exactly that variant (and the default
    comparison reports the primitives that differ); fixtures of the default variant select the defaults."""
    kit = _kit()
    try:
        oracle.set_semantics(**variant)
        kit.generate(_OracleAsCv(oracle), str(tmp_path))
    finally:
        oracle.set_semantics()
    sel = select_semantics(oracle, str(tmp_path))
    assert sel["gauss_taps"] == variant.get("gauss_taps", [18, 34, 48, 56, 48, 34, 18])
    assert sel["resize_single_stage"] is bool(variant.get("resize_single_stage", False))
    assert sel["atan2_fma"] is bool(variant.get("atan2_fma", False))
    assert sel["fast_ok"] and sel["default"] is (not variant)
    bad, _ = compare(oracle, str(tmp_path))
    assert (bad == []) is (not variant)
    if variant:
        assert any(b.startswith("blur") for b in bad) and any(b.startswith("fastAtan2") for b in bad)


PIN_IMAGE_100_SHA16 = "fd55b5ceb08b6542"
ATAN2_INPUTS_SHA16 = "df68cf345f42270c"


# ---- independent third-party cross-checks available in the build image (tools/pin_skimage.py) --------------------------------
SKPINS = os.path.join(ROOT, "tests", "golden", "skimage_pins")


def _skit():
    spec = importlib.util.spec_from_file_location("pin_skimage", os.path.join(ROOT, "tools", "pin_skimage.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.skipif(not os.path.exists(os.path.join(SKPINS, "meta.json")), reason="run /opt/conda/bin/python3.9 tools/pin_skimage.py")
def test_fast9_detection_and_score_against_scikit_image(oracle):
    """scikit-image's corner_fast(n=9) is an independent implementation of the FAST-9/16 segment test.  (a) The oracle's
    detection set must equal its mask at every threshold; (b) OpenCV's cornerScore is "the largest threshold at which the
    pixel is still a corner", so the score plane computed once at the lowest threshold must reproduce scikit-image's mask at
    EVERY threshold of the ladder: mask(t) == (score >= t)."""
    sk, kit = _skit(), _kit()
    meta = json.load(open(os.path.join(SKPINS, "meta.json")))
    z = np.load(os.path.join(SKPINS, "fast9_masks.npz"))
    rois = kit.fast_inputs()
    img, cells = rois[kit.FAST_CELLS], rois[:8]
    assert kit.sha(np.concatenate([img.ravel()] + [c.ravel() for c in cells])) == meta["fast_inputs_sha256"]

    def mask(name, shape):
        return np.unpackbits(z[name])[:shape[0] * shape[1]].reshape(shape).astype(bool)
    lowest = min(sk.FAST_LADDER)
    _, score_lo = oracle.fast9_planes(img, lowest)
    n_corners = []
    for t in sk.FAST_LADDER:
        want = mask(f"image_t{t}", img.shape)
        corner, score = oracle.fast9_planes(img, t)
        assert np.array_equal(corner.astype(bool), want), f"detection set at threshold {t}"
        assert np.array_equal(score_lo >= t, want), f"score semantics at threshold {t}"
        assert np.array_equal(score[want], score_lo[want])             # the score does not depend on the detection threshold
        n_corners.append(int(want.sum()))
    assert n_corners[0] > 2000 and n_corners[-1] < n_corners[0] // 10 and n_corners == sorted(n_corners, reverse=True)
    for i, c in enumerate(cells):
        for t in (7, 20):
            corner, _ = oracle.fast9_planes(c, t)
            assert np.array_equal(corner.astype(bool), mask(f"cell{i}_t{t}", c.shape)), (i, t)
    # every keypoint FAST + NMS returns is a detected corner with that score
    kp = oracle.fast9_nms(img, 20)
    corner, score = oracle.fast9_planes(img, 20)
    assert len(kp) > 100 and np.all(corner[kp[:, 1], kp[:, 0]] == 1) and np.array_equal(score[kp[:, 1], kp[:, 0]], kp[:, 2])


@pytest.mark.skipif(not os.path.exists(os.path.join(SKPINS, "meta.json")), reason="run /opt/conda/bin/python3.9 tools/pin_skimage.py")
def test_resize_and_blur_geometry_against_scikit_image_and_scipy(oracle):
    """Float references from scikit-image (bilinear resize, half-pixel centres, edge clamp) and scipy (7-tap Gaussian, sigma 2,
    reflect-101) against the oracle's fixed-point restatements: the sampling geometry, border rules and kernel shape must agree
    to within the fixed-point rounding.  (A wrong centre convention — x * scale instead of (x + 0.5) * scale - 0.5 — or
    reflect vs reflect-101 shows up as errors of tens of grey levels on this texture.)"""
    sk, kit = _skit(), _kit()
    z = np.load(os.path.join(SKPINS, "resize_float.npz"))
    for i, (name, sr, sc, dr, dc) in enumerate(sk.RESIZE_CASES):
        src = kit.pin_image(300 + i, sr, sc)
        got = oracle.resize_linear_u8(src, dr, dc).astype(np.float64)
        ref = z[name].astype(np.float64) / 8.0
        err = np.abs(got - ref)
        assert err.max() <= 1.0 and err.mean() < 0.30, (name, err.max(), err.mean())
    z = np.load(os.path.join(SKPINS, "gaussian_float.npz"))
    for i, (name, r, c) in enumerate(sk.BLUR_CASES):
        src = kit.pin_image(400 + i, r, c)
        got = oracle.gaussian7(src).astype(np.float64)
        ref = z[name].astype(np.float64) / 8.0
        err = np.abs(got - ref)
        assert err.max() <= 2.0 and err.mean() < 0.45, (name, err.max(), err.mean())


def test_fast_atan2_against_numpy(oracle):
    """cv::fastAtan2 is documented as accurate to about 0.3 degrees; the oracle's polynomial restatement must stay inside that
    band of numpy's arctan2 on the kit's inputs (dense integer moments, large moments, fractions, axes)."""
    kit = _kit()
    y, x = kit.atan2_inputs()
    sel = np.r_[0:len(y):7, len(y) - 15:len(y)]
    got = np.array([oracle.fast_atan2(a, b) for a, b in zip(y[sel], x[sel])], np.float64)
    ref = np.degrees(np.arctan2(y[sel].astype(np.float64), x[sel].astype(np.float64))) % 360.0
    nz = (np.abs(y[sel]) + np.abs(x[sel])) > 1e-6     # fastAtan2 adds DBL_EPSILON to the denominator: (1e-30, 1e-30) -> 0, not 45
    d = np.abs(got - ref)
    d = np.minimum(d, 360.0 - d)
    assert d[nz].max() < 0.3, d[nz].max()
    assert np.all(got[(y[sel] == 0) & (x[sel] == 0)] == 0.0)


# ---- the WHOLE extractor around the real primitives (tools/pin_opencv.py --extractor) --------------------------------------
def compare_extractor(extract, pins_dir, images=None):
    """extract(image, nfeatures) -> (keypoints KP_DTYPE [n], descriptors u8 [n, 32]) of ORBextractor(nfeatures, 1.2, 8, 20, 7) with
    vLappingArea (0, 0), compared with the fixtures extractor.npz / extractor_meta.json of pins_dir.  -> list of mismatches.
    images: {case name: image} for cases whose input is not one of the kit's pin images (hash checked all the same)."""
    kit = _kit()
    meta = json.load(open(os.path.join(pins_dir, "extractor_meta.json")))
    z = np.load(os.path.join(pins_dir, "extractor.npz"))
    bad = []
    for i, (name, rows, cols, nfeat) in enumerate(meta["cases"]):
        img = images[name] if images and name in images else kit.pin_image(700 + i, rows, cols)
        assert kit.sha(img) == meta["inputs_sha256"][name], f"input of {name} differs from the generating run"
        kps, desc = extract(img, nfeat)
        want_k = z[f"{name}_kps"].reshape(-1, 28).view(kit.KP_DTYPE).reshape(-1)
        want_d = z[f"{name}_desc"]
        if len(kps) != len(want_k):
            bad.append(f"extractor {name}: {len(kps)} keypoints vs {len(want_k)} in the fixtures")
            continue
        for f in kit.KP_DTYPE.names:
            d = np.flatnonzero(np.asarray(kps[f]).view(np.uint32) != want_k[f].view(np.uint32))
            if len(d):
                bad.append(f"extractor {name}: keypoint field {f} differs on {len(d)} of {len(kps)} keypoints, first {d[0]}: "
                           f"{kps[f][d[0]]!r} vs {want_k[f][d[0]]!r}")
        d = np.flatnonzero((np.asarray(desc) != want_d).any(1))
        if len(d):
            bad.append(f"extractor {name}: {len(d)} of {len(kps)} descriptors differ, first {d[0]} "
                       f"({int(np.unpackbits(np.asarray(desc)[d[0]] ^ want_d[d[0]]).sum())} bits)")
    return bad, meta


def _oracle_extract(oracle):
    def run(img, nfeat):
        _, kps, desc = oracle.OracleExtractor(nfeat, 1.2, 8, 20, 7)(img)
        return kps, desc
    return run


def test_kit_pattern_table_is_the_references(oracle):
    """the kit carries bit_pattern_31_ compressed inside its one file: it must be the table the product and the oracle compile in"""
    import re
    kit = _kit()
    txt = re.sub(r"//.*", "", open(os.path.join(ROOT, "ms-slam_amd", "csrc", "orb_pattern.inc")).read())
    want = np.array([int(x) for x in re.findall(r"-?\d+", txt)], np.int8).reshape(256, 4)
    assert np.array_equal(kit.orb_pattern(), want)


def test_extractor_kit_plumbing(oracle, tmp_path):
    """The kit's plain-Python restatement of everything AROUND the four primitives (level sizes, cell loop + threshold fallback,
    DistributeOctTree with libstdc++'s std::sort, IC_Angle, steered BRIEF), driven by a stand-in "cv2" backed by the oracle's
    primitives, must reproduce the C++ oracle's extractor bit for bit — two independent restatements of ORBextractor.cc, one in
    C++ (oracle/orb_extractor_oracle.cc), one in Python (tools/pin_opencv.py) — so that the first real run differs from the
    oracle only where real OpenCV differs from the restated primitives."""
    kit = _kit()
    meta = kit.generate_extractor(_OracleAsCv(oracle), str(tmp_path))
    assert [c[0] for c in meta["cases"]] == ["kitti", "euroc", "4seasons", "small_odd"] and "cosf" in meta["trig"]
    bad, _ = compare_extractor(_oracle_extract(oracle), str(tmp_path))
    assert bad == []
    z = np.load(os.path.join(str(tmp_path), "extractor.npz"))
    assert len(z["kitti_kps"]) > 2000 and len(z["euroc_kps"]) > 1000 and z["kitti_cand"].sum() > 10000
    # teeth: one flipped descriptor bit, one keypoint moved by one ulp
    arrs = dict(z)
    arrs["euroc_desc"] = arrs["euroc_desc"].copy(); arrs["euroc_desc"][7, 3] ^= 16
    k = arrs["kitti_kps"].copy().reshape(-1, 28); k[11, 12] ^= 1; arrs["kitti_kps"] = k     # byte 12 = lowest byte of `angle`
    np.savez_compressed(os.path.join(str(tmp_path), "extractor.npz"), **arrs)
    bad, _ = compare_extractor(_oracle_extract(oracle), str(tmp_path))
    assert len(bad) == 2 and "keypoint field angle" in bad[0] and "descriptors differ" in bad[1]


def test_extractor_kit_and_consumer_follow_brief_tap(oracle, tmp_path):
    """Fixtures made under tap contraction 1 on a frame whose descriptors depend on it (test_semantics_variants.BRIEF_TAP_SEEDS):
    the kit's Python extractor reproduces the oracle under that contraction, the meta data carries it, pinned_semantics hands it to
    the consumer, and select_brief_tap recovers it from the descriptors alone."""
    from msorb import synth
    from test_semantics_variants import BRIEF_TAP_SEEDS
    kit = _kit()
    mode, seed = BRIEF_TAP_SEEDS[0]
    img = synth.image(seed, synth.KITTI["rows"], synth.KITTI["cols"])
    kps, desc, _ = kit.extract_orb(_OracleAsCv(oracle), img, 2000, brief_tap=mode)
    np.savez_compressed(os.path.join(str(tmp_path), "extractor.npz"), frame_kps=kps.view(np.uint8).reshape(-1, 28), frame_desc=desc)
    meta = {"kit_version": kit.KIT_VERSION, "cv2_version": "oracle-stand-in", "trig": "libm", "brief_tap": mode, "brief_tap_source": "test",
            "inputs_sha256": {"frame": kit.sha(img)}, "cases": [["frame", img.shape[0], img.shape[1], 2000]]}
    json.dump(meta, open(os.path.join(str(tmp_path), "extractor_meta.json"), "w"))

    def extract_under(**sem):
        def run(image, nfeat):
            try:
                oracle.set_semantics(**sem)
                _, k, d = oracle.OracleExtractor(nfeat, 1.2, 8, 20, 7)(image)
                return k, d
            finally:
                oracle.set_semantics()
        return run
    images = {"frame": img}
    assert pinned_semantics(str(tmp_path)) == {"brief_tap": mode}
    assert select_brief_tap(extract_under, str(tmp_path), images=images) == [mode]
    bad, _ = compare_extractor(extract_under(), str(tmp_path), images)
    assert len(bad) == 1 and "descriptors differ" in bad[0]


def test_extractor_kit_follows_a_non_default_semantics_variant(oracle, tmp_path):
    """fixtures made with another Gaussian / resize / atan2 variant of the semantics table differ from the default oracle and are
    reproduced by the oracle once it is set to that variant: what the consumer of real fixtures would do after select_semantics()"""
    kit = _kit()
    variant = dict(gauss_taps=[18, 34, 49, 55, 49, 34, 18], resize_single_stage=True, atan2_fma=True)
    try:
        oracle.set_semantics(**variant)
        kit.generate_extractor(_OracleAsCv(oracle), str(tmp_path), cases=kit.EXTRACTOR_CASES[3:])
        bad, _ = compare_extractor(_oracle_extract(oracle), str(tmp_path))
        assert bad == []
    finally:
        oracle.set_semantics()
    bad, _ = compare_extractor(_oracle_extract(oracle), str(tmp_path))
    assert bad, "the default semantics reproduced fixtures of another variant"


def pinned_semantics(pins_dir):
    """The semantics table the fixtures of pins_dir call for, as keyword arguments of set_semantics (oracle and msorb alike): the
    variant the primitive fixtures selected (selected_semantics.json, written by test_pins_against_real_opencv) + the tap contraction
    the extractor fixtures were made with (extractor_meta.json "brief_tap": stated or probed by the generating run, kit version >= 2)."""
    kw = {}
    sel = os.path.join(pins_dir, "selected_semantics.json")
    if os.path.exists(sel):
        s = json.load(open(sel))
        if s.get("gauss_taps") and s.get("resize_single_stage") is not None and s.get("atan2_fma") is not None:
            kw.update(gauss_taps=s["gauss_taps"], resize_single_stage=s["resize_single_stage"], atan2_fma=s["atan2_fma"])
    meta = os.path.join(pins_dir, "extractor_meta.json")
    if os.path.exists(meta):
        kw["brief_tap"] = int(json.load(open(meta)).get("brief_tap", 0))
    return kw


def select_brief_tap(extract_under, pins_dir, base=None, images=None):
    """Which tap contractions reproduce the extractor fixtures?  extract_under(**semantics) -> extract(image, nfeatures).  For
    fixtures whose descriptors come from a compiled build of the reference (a dump of the real ORBextractor in the kit's format),
    whose contraction nobody stated: -> the list of brief_tap values with no mismatch (often all three — the conventions differ
    on ~1 descriptor in 40 frames — and [] when something else is wrong)."""
    base = dict(base or {})
    ok = []
    for m in (0, 1, 2):
        base["brief_tap"] = m
        bad, _ = compare_extractor(extract_under(**base), pins_dir, images)
        if not bad:
            ok.append(m)
    return ok


def test_extractor_pins_against_real_opencv(oracle):
    if not os.path.exists(os.path.join(PINS, "extractor_meta.json")):
        pytest.skip("no whole-extractor OpenCV pins committed yet: run `python tools/pin_opencv.py --extractor` on a machine with cv2 — "
                    "until then keypoints and descriptors are bit-exact against the oracle only (PARITY UNPINNED)")
    try:
        oracle.set_semantics(**pinned_semantics(PINS))
        bad, meta = compare_extractor(_oracle_extract(oracle), PINS)
    finally:
        oracle.set_semantics()
    assert not bad, f"the oracle's extractor differs from the kit's run on OpenCV {meta['cv2_version']} ({meta['trig']}):\n" + "\n".join(bad)
