"""GPU parity tests: HIP extractor (through the C ABI) vs the CPU oracle, stage by stage and end to end.
Bit-exact for every integer/byte product (pyramid, blur, candidates incl. FAST scores, descriptors) and for
the float keypoint fields (same IEEE operations in the same order)."""
import os
import sys

import numpy as np
import pytest

from msorb import synth

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))   # the pin-kit consumers live in test_oracle_pins.py

pytestmark = pytest.mark.gpu

CONFIGS = {
    "kitti": synth.KITTI,
    "euroc": synth.EUROC,
    "euroc_yaml": synth.EUROC_YAML,
    "fourseasons": synth.FOURSEASONS,
    "small": dict(rows=240, cols=320, nfeatures=500, scale=1.2, nlevels=8, ini_th=20, min_th=7),
}


def _pair(msorb_mod, oracle, cfg):
    ex = msorb_mod.ORBextractor(cfg["nfeatures"], cfg["scale"], cfg["nlevels"], cfg["ini_th"], cfg["min_th"])
    ref = oracle.OracleExtractor(cfg["nfeatures"], cfg["scale"], cfg["nlevels"], cfg["ini_th"], cfg["min_th"])
    return ex, ref


def _assert_same(kps, desc, rkps, rdesc):
    assert len(kps) == len(rkps)
    for f in ("octave", "x", "y", "response", "size", "angle", "class_id"):
        assert np.array_equal(kps[f].view(np.uint32), rkps[f].view(np.uint32)), f"keypoint field {f} differs"
    assert np.array_equal(desc, rdesc)


@pytest.mark.parametrize("name", list(CONFIGS))
def test_stagewise_and_end_to_end(msorb_mod, oracle, name):
    cfg = CONFIGS[name]
    ex, ref = _pair(msorb_mod, oracle, cfg)
    for seed in (0, 1):
        img = synth.image(100 + seed, cfg["rows"], cfg["cols"])
        mono, kps, desc = ex(img)
        rmono, rkps, rdesc = ref(img)
        for l in range(cfg["nlevels"]):
            assert np.array_equal(ex.debug_level(0, l), ref.level(l)), f"pyramid level {l}"
            assert np.array_equal(ex.pyramid_level(l), ref.level(l)), f"host pyramid view {l}"
            assert np.array_equal(ex.debug_candidates(0, l), ref.candidates(l)), f"FAST candidates level {l}"
            if len(ref.selected(l)):
                assert np.array_equal(ex.debug_level(0, l, blurred=True), ref.level(l, blurred=True)), f"blur {l}"
        assert mono == rmono
        _assert_same(kps, desc, rkps, rdesc)
        assert len(kps) > cfg["nfeatures"] // 2
    ex.close()


def test_lapping_area_split(msorb_mod, oracle):
    cfg = CONFIGS["small"]
    ex, ref = _pair(msorb_mod, oracle, cfg)
    img = synth.image(5, cfg["rows"], cfg["cols"])
    for lap in ((0, 0), (0, 1000), (100, 200), (150, 150)):
        mono, kps, desc = ex(img, lap)
        rmono, rkps, rdesc = ref(img, lap)
        assert mono == rmono
        _assert_same(kps, desc, rkps, rdesc)
    assert mono < len(kps) or lap == (0, 0)
    ex.close()


def test_flat_and_low_contrast_images(msorb_mod, oracle):
    """Threshold fallback (minThFAST) and the empty-output path (ORBextractor.cc:843-847, 1108-1114)."""
    cfg = CONFIGS["small"]
    ex, ref = _pair(msorb_mod, oracle, cfg)
    flat = np.full((cfg["rows"], cfg["cols"]), 90, np.uint8)
    mono, kps, desc = ex(flat)
    assert mono == 0 and len(kps) == 0 and desc.shape == (0, 32)
    rng = np.random.Generator(np.random.PCG64(3))
    low = (100 + 12 * rng.random((cfg["rows"], cfg["cols"]))).astype(np.uint8)  # only minTh fires
    low[::17, ::13] += 14
    mono, kps, desc = ex(low)
    rmono, rkps, rdesc = ref(low)
    assert mono == rmono
    _assert_same(kps, desc, rkps, rdesc)
    noise = rng.integers(0, 256, (cfg["rows"], cfg["cols"]), dtype=np.uint8)  # saturates every cell
    mono, kps, desc = ex(noise)
    rmono, rkps, rdesc = ref(noise)
    _assert_same(kps, desc, rkps, rdesc)
    ex.close()


def test_empty_and_too_small(msorb_mod):
    ex = msorb_mod.ORBextractor(500, 1.2, 8, 20, 7)
    mono, kps, desc = ex(np.zeros((0, 0), np.uint8))
    assert mono == -1 and len(kps) == 0          # ORBextractor.cc:1090-1091
    with pytest.raises(msorb_mod.MsorbError) as e:
        ex(np.zeros((100, 100), np.uint8))       # level 7 is 28 px: the reference divides by zero there
    assert e.value.code == msorb_mod.E_GEOMETRY
    ex.close()


def test_batch_matches_single(msorb_mod, oracle):
    import torch
    cfg = CONFIGS["kitti"]
    ex, ref = _pair(msorb_mod, oracle, cfg)
    batch = synth.stereo_batch(3, cfg["rows"], cfg["cols"], seed0=40)
    d = torch.from_numpy(batch).cuda()
    counts, mono, d_kps, d_desc = ex.extract_batch(d)
    kps_list = msorb_mod.keypoints_from_device(d_kps, counts)
    desc_all = d_desc.cpu().numpy()
    for i in range(batch.shape[0]):
        rmono, rkps, rdesc = ref(batch[i])
        assert counts[i] == len(rkps) and mono[i] == rmono
        _assert_same(kps_list[i], desc_all[i, :counts[i]], rkps, rdesc)
    # strided view: rows padded to 1280
    padded = torch.zeros((6, cfg["rows"], 1280), dtype=torch.uint8, device="cuda")
    padded[:, :, :cfg["cols"]] = d
    counts2, mono2, d_kps2, d_desc2 = ex.extract_batch(padded[:, :, :cfg["cols"]])
    assert np.array_equal(counts, counts2)
    assert torch.equal(d_desc2[0, :counts[0]], d_desc[0, :counts[0]])
    ex.close()


def test_rotated_images_cover_all_orientations(msorb_mod, oracle):
    """The angle enters the descriptor only through (cos,sin): rotated copies of one scene push keypoint
    orientations around the whole circle (and exercise portrait geometry, nIni == 1)."""
    cfg = CONFIGS["small"]
    ex, ref = _pair(msorb_mod, oracle, cfg)
    base = synth.image(9, cfg["rows"], cfg["cols"])
    for k in range(4):
        img = np.ascontiguousarray(np.rot90(base, k))
        mono, kps, desc = ex(img)
        rmono, rkps, rdesc = ref(img)
        assert mono == rmono and len(kps) > 100
        _assert_same(kps, desc, rkps, rdesc)
    ex.close()


def test_large_batch_uses_overlapped_sub_batches(msorb_mod, oracle):
    """>= 16 images take the multi-stream sub-batch pipeline (2 groups by default, also 3 and 1): every image must
    still equal the oracle, independent of the grouping."""
    import torch
    cfg = CONFIGS["small"]
    ex, ref = _pair(msorb_mod, oracle, cfg)
    n = 21
    batch = np.stack([synth.image(300 + i, cfg["rows"], cfg["cols"]) for i in range(n)])
    want = [ref(batch[i]) for i in range(n)]
    d = torch.from_numpy(batch).cuda()
    for groups, blur2 in ((2, True), (3, True), (1, False), (4, False)):
        ex.set_overlap(groups, blur2)
        counts, mono, d_kps, d_desc = ex.extract_batch(d)
        kps_list = msorb_mod.keypoints_from_device(d_kps, counts)
        desc_all = d_desc.cpu().numpy()
        for i in range(n):
            rmono, rkps, rdesc = want[i]
            assert counts[i] == len(rkps) and mono[i] == rmono, (groups, i)
            _assert_same(kps_list[i], desc_all[i, :counts[i]], rkps, rdesc)
    ex.close()


@pytest.mark.parametrize("name", ["kitti", "euroc", "fourseasons", "odd"])
def test_batch_kernels_full_size(msorb_mod, oracle, name):
    """Batches of >= 16 images switch to the row-streaming pyramid kernel (and the streaming blur): every dataset
    geometry plus an odd-sized one (width not a multiple of 4), one sub-batch, checked against the oracle on a sample
    of the images and against each other on all."""
    import torch
    cfg = CONFIGS[name] if name != "odd" else dict(rows=333, cols=517, nfeatures=700, scale=1.2, nlevels=8, ini_th=20, min_th=7)
    ex, ref = _pair(msorb_mod, oracle, cfg)
    n = 16
    batch = np.stack([synth.image(700 + (i % 4), cfg["rows"], cfg["cols"]) for i in range(n)])
    d = torch.from_numpy(batch).cuda()
    ex.set_overlap(1, False)
    counts, mono, d_kps, d_desc = ex.extract_batch(d)
    kps_list = msorb_mod.keypoints_from_device(d_kps, counts)
    desc_all = d_desc.cpu().numpy()
    for i in (0, 1, 2, 3):
        rmono, rkps, rdesc = ref(batch[i])
        assert counts[i] == len(rkps) and mono[i] == rmono
        _assert_same(kps_list[i], desc_all[i, :counts[i]], rkps, rdesc)
    for i in range(4, n):   # copies of the first four images
        assert counts[i] == counts[i % 4]
        assert np.array_equal(desc_all[i, :counts[i]], desc_all[i % 4, :counts[i]])
    ex2, ref2 = _pair(msorb_mod, oracle, cfg)
    ref2(batch[n - 1])
    for lvl in range(1, cfg["nlevels"]):   # pyramid / blurred planes of the last image vs the oracle
        assert np.array_equal(ex.debug_level(n - 1, lvl), ref2.level(lvl)), lvl
        assert np.array_equal(ex.debug_level(n - 1, lvl, blurred=True), ref2.level(lvl, blurred=True)), lvl
    ex2.close()
    ex.close()


@pytest.mark.parametrize("name", ["kitti", "euroc", "fourseasons", "odd"])
def test_batch_pyramid_kernel_on_padded_rows(msorb_mod, oracle, name):
    """The batch pyramid kernel (pyr_resize_bandreg_kernel: a band of 8 output rows per wave, source rows parked in registers)
    on a batch whose rows are padded to a multiple of 64 bytes, as bench.py lays its images out: every level of every distinct
    image is the oracle's cv::resize restatement, bit for bit."""
    import torch
    cfg = CONFIGS[name] if name != "odd" else dict(rows=333, cols=517, nfeatures=700, scale=1.2, nlevels=8, ini_th=20, min_th=7)
    ex, ref = _pair(msorb_mod, oracle, cfg)
    n = 16
    pitch = (cfg["cols"] + 63) // 64 * 64
    batch = np.stack([synth.image(900 + (i % 3), cfg["rows"], cfg["cols"]) for i in range(n)])
    store = torch.zeros((n, cfg["rows"], pitch), dtype=torch.uint8, device="cuda")
    view = store[:, :, :cfg["cols"]]
    view.copy_(torch.from_numpy(batch).cuda())
    ex.set_overlap(1, False)
    ex.pyramid_batch(view)
    torch.cuda.synchronize()
    for i in (0, 1, 2, n - 1):
        ref(batch[i])
        for lvl in range(1, cfg["nlevels"]):
            assert np.array_equal(ex.debug_level(i, lvl), ref.level(lvl)), (i, lvl)
    ex.close()


@pytest.mark.parametrize("seed", [11, 12])
def test_random_geometries_and_strides_batch(msorb_mod, oracle, seed):
    """Random image sizes, level counts, row pitches (tight, 4-, 16-, 64-byte aligned, odd padding) and buffer offsets
    through the batch kernels (which pick their aligned / LDS-DMA / staged variants from exactly these properties): keypoints,
    descriptors, and every pyramid / blurred level of the last image are the oracle's."""
    import torch
    rng = np.random.default_rng(seed)
    done = 0
    while done < 4:
        rows, cols = int(rng.integers(240, 480)), int(rng.integers(330, 1100))
        nfeat, nlev = int(rng.choice([500, 1000])), int(rng.choice([4, 6, 8]))
        pad = int(rng.choice([0, 3, 4, 16, 21, 64]))
        if rng.random() < 0.5:
            pad += (-(cols + pad)) % int(rng.choice([4, 16, 64]))
        off = int(rng.choice([0, 1, 4, 16]))
        n = int(rng.choice([16, 17, 64]))
        pitch = cols + pad
        ex = msorb_mod.ORBextractor(nfeat, 1.2, nlev, 20, 7)
        ref = oracle.OracleExtractor(nfeat, 1.2, nlev, 20, 7)
        try:
            imgs = np.stack([synth.image(6000 + 13 * done + (i % 3), rows, cols) for i in range(n)])
            flat = torch.zeros(n * rows * pitch + 64, dtype=torch.uint8, device="cuda")
            view = flat[off:off + n * rows * pitch].view(n, rows, pitch)[:, :, :cols]
            view.copy_(torch.from_numpy(imgs).cuda())
            try:
                counts, mono, d_kps, d_desc = ex.extract_batch(view)
            except msorb_mod.MsorbError as e:
                assert e.code == msorb_mod.E_GEOMETRY         # too small for the reference's cell grid at some level
                continue
            kps = msorb_mod.keypoints_from_device(d_kps, counts)
            desc = d_desc.cpu().numpy()
            for i in (0, 2, n - 1):
                rmono, rkps, rdesc = ref(imgs[i])
                assert counts[i] == len(rkps) and mono[i] == rmono, (rows, cols, pitch, off, n, nlev, i)
                _assert_same(kps[i], desc[i, :counts[i]], rkps, rdesc)
            for lvl in range(1, nlev):
                assert np.array_equal(ex.debug_level(n - 1, lvl), ref.level(lvl)), (rows, cols, pitch, off, lvl)
                assert np.array_equal(ex.debug_level(n - 1, lvl, blurred=True), ref.level(lvl, blurred=True)), (rows, cols, pitch, off, lvl)
            done += 1
        finally:
            ex.close()


def test_host_quadtree_mode_matches(msorb_mod, oracle, monkeypatch):
    """MSORB_QUADTREE=host keeps the selection on the host thread pool (orb_host.cc); same result."""
    monkeypatch.setenv("MSORB_QUADTREE", "host")
    cfg = CONFIGS["euroc"]
    ex, ref = _pair(msorb_mod, oracle, cfg)
    img = synth.image(55, cfg["rows"], cfg["cols"])
    mono, kps, desc = ex(img)
    rmono, rkps, rdesc = ref(img)
    assert mono == rmono
    _assert_same(kps, desc, rkps, rdesc)
    ex.close()


def test_saturated_single_cell(msorb_mod, oracle):
    """One 69x69-pixel cell of pure noise: more quick-test survivors than the FAST kernel's LDS work list holds,
    which takes the chunked / task-scan path of the kernel (nlevels = 1 keeps the tiny image legal)."""
    rng = np.random.Generator(np.random.PCG64(8))
    for rows, cols in ((101, 101), (101, 171), (90, 240)):
        img = rng.integers(0, 256, (rows, cols), dtype=np.uint8)
        ex = msorb_mod.ORBextractor(300, 1.2, 1, 20, 7)
        ref = oracle.OracleExtractor(300, 1.2, 1, 20, 7)
        mono, kps, desc = ex(img)
        rmono, rkps, rdesc = ref(img)
        assert np.array_equal(ex.debug_candidates(0, 0), ref.candidates(0))
        assert mono == rmono and len(kps) > 50
        _assert_same(kps, desc, rkps, rdesc)
        ex.close()


@pytest.mark.parametrize("nfeat,nlev,kind", [(10000, 8, "noise"), (10000, 8, "scene"), (12000, 8, "noise"), (6000, 2, "noise"), (11000, 1, "noise")])
def test_quotas_beyond_a_workgroups_lds_stay_on_the_device(msorb_mod, oracle, nfeat, nlev, kind):
    """Level quotas above ~1 800 keypoints (the monocular initialisation extractor of Tracking.cc:601 asks for 5 * nFeatures = 10 000):
    the selection's workspace does not fit 160 KB of LDS; the SAME selection code runs over a workspace in global memory
    (quadtree_global_kernels.hip) — on the device, in the full pipeline (per frame, batches, pairs), where rounds 1-5 handed these
    quotas to the host twin.  (11000 features on ONE level: 44 016 node slots — past the 14 bits of the LDS build's 16-bit labels; the
    global build's labels are 32 bits.)"""
    import torch
    cfg = synth.KITTI
    rng = np.random.Generator(np.random.PCG64(nfeat + nlev))
    imgs = [rng.integers(0, 256, (cfg["rows"], cfg["cols"]), dtype=np.uint8) if kind == "noise" else synth.image(190 + i, cfg["rows"], cfg["cols"])
            for i in range(2)]
    ex = msorb_mod.ORBextractor(nfeat, 1.2, nlev, 20, 7)
    ref = oracle.OracleExtractor(nfeat, 1.2, nlev, 20, 7)
    try:
        want = [ref(im) for im in imgs]
        for im, (rmono, rkps, rdesc) in zip(imgs, want):
            mono, kps, desc = ex(im)
            assert mono == rmono and len(kps) >= (nfeat * 9 // 10 if kind == "noise" else 2000)
            _assert_same(kps, desc, rkps, rdesc)
        d = torch.from_numpy(np.stack(imgs * 4)).cuda()       # 8 images: the batch path (device pipeline: it used to be refused here)
        counts, monos, d_kps, d_desc = ex.extract_batch(d)
        got = msorb_mod.keypoints_from_device(d_kps, counts)
        for i in (0, 1, 7):
            rmono, rkps, rdesc = want[i % 2]
            assert monos[i] == rmono
            _assert_same(got[i], d_desc[i, :counts[i]].cpu().numpy(), rkps, rdesc)
    finally:
        ex.close()


@pytest.mark.parametrize("name", ["kitti", "euroc", "small"])
def test_global_workspace_selection_at_ordinary_quotas(msorb_mod, oracle, monkeypatch, name):
    """MSORB_QUADTREE=global: the global-memory form of the selection at quotas the LDS form serves — one source, two address spaces,
    same keypoints (dense and low-texture levels, the careful sweep's sort included)."""
    monkeypatch.setenv("MSORB_QUADTREE", "global")
    cfg = CONFIGS[name]
    ex, ref = _pair(msorb_mod, oracle, cfg)
    try:
        for seed, kind in ((61, "scene"), (62, "low")):
            img = synth.image(seed, cfg["rows"], cfg["cols"])
            if kind == "low":
                img = (img.astype(np.float32) * 0.25 + 96).astype(np.uint8)      # low contrast: sparse levels, deep trees
            mono, kps, desc = ex(img)
            rmono, rkps, rdesc = ref(img)
            assert mono == rmono
            _assert_same(kps, desc, rkps, rdesc)
    finally:
        ex.close()


@pytest.mark.parametrize("nfeat,nlev,kind", [(3500, 8, "scene"), (3500, 8, "noise"), (8000, 8, "noise"), (3000, 2, "noise")])
def test_large_feature_quota_device_quadtree(msorb_mod, oracle, nfeat, nlev, kind):
    """Quotas whose quadtree workspace exceeds the 64 KB default LDS window (nfeatures > ~3300 at 1.2 / 8 levels): the
    select kernel raises its dynamic-LDS limit (160 KB per workgroup on gfx950) and the careful sweep sorts more nodes than
    the fixed 32-range lists of earlier rounds held (range lists are sized by the quota now).  Noise images saturate every
    cell, so the node lists reach the quota.  Per frame, batched and the stereo chain (device pipeline required)."""
    import torch
    cfg = synth.KITTI
    rng = np.random.Generator(np.random.PCG64(nfeat + nlev))
    imgs = [rng.integers(0, 256, (cfg["rows"], cfg["cols"]), dtype=np.uint8) if kind == "noise" else synth.image(90 + i, cfg["rows"], cfg["cols"])
            for i in range(2)]
    ex = msorb_mod.ORBextractor(nfeat, 1.2, nlev, 20, 7)
    ref = oracle.OracleExtractor(nfeat, 1.2, nlev, 20, 7)
    try:
        want = [ref(im) for im in imgs]
        for im, (rmono, rkps, rdesc) in zip(imgs, want):
            mono, kps, desc = ex(im)
            assert mono == rmono and len(kps) >= (nfeat * 9 // 10 if kind == "noise" else 1000)
            _assert_same(kps, desc, rkps, rdesc)
        d = torch.from_numpy(np.stack(imgs * 8)).cuda()       # 16 images: the batch kernels
        counts, monos, d_kps, d_desc = ex.extract_batch(d)
        got = msorb_mod.keypoints_from_device(d_kps, counts)
        for i in (0, 1, 15):
            rmono, rkps, rdesc = want[i % 2]
            assert monos[i] == rmono
            _assert_same(got[i], d_desc[i, :counts[i]].cpu().numpy(), rkps, rdesc)
    finally:
        ex.close()


@pytest.mark.parametrize("nfeat,nlev", [(40, 8), (12, 3), (100, 8)])
def test_tiny_feature_counts_exceed_their_quota_like_the_reference(msorb_mod, oracle, nfeat, nlev):
    """DistributeOctTree's first pass divides every initial column before any quota check (ORBextractor.cc:610-681): with a
    quota below 4 * nIni a level returns more keypoints than its quota + 3 — KITTI (nIni = 4 at the wide levels) with 40 features
    returns more than nfeatures + 3 * nlevels.  The capacity covers it (nfeatures + 19 * nlevels); per frame and batched."""
    import torch
    cfg = synth.KITTI
    imgs = [synth.image(520 + i, cfg["rows"], cfg["cols"]) for i in range(2)]
    ex = msorb_mod.ORBextractor(nfeat, 1.2, nlev, 20, 7)
    ref = oracle.OracleExtractor(nfeat, 1.2, nlev, 20, 7)
    try:
        assert ex.capacity == nfeat + 19 * nlev
        want = [ref(im) for im in imgs]
        for im, (rmono, rkps, rdesc) in zip(imgs, want):
            mono, kps, desc = ex(im)
            assert mono == rmono and len(rkps) > nfeat
            _assert_same(kps, desc, rkps, rdesc)
        d = torch.from_numpy(np.stack(imgs * 8)).cuda()
        counts, monos, d_kps, d_desc = ex.extract_batch(d)
        got = msorb_mod.keypoints_from_device(d_kps, counts)
        for i in (0, 1, 15):
            rmono, rkps, rdesc = want[i % 2]
            _assert_same(got[i], d_desc[i, :counts[i]].cpu().numpy(), rkps, rdesc)
    finally:
        ex.close()


def test_full_bench_size_properties(msorb_mod, oracle):
    """BASELINE.json configs[1] at bench.py's batch size (128 stereo pairs = 256 images, default 2 sub-batches): too
    big for the oracle image by image, so size-independent properties carry the check — copies of an image must give
    bit-identical outputs wherever they sit in the batch, the output contract holds for every image, a sample is
    compared with the oracle, and matching the descriptors against themselves must return the identity."""
    import torch
    cfg = CONFIGS["kitti"]
    ex, ref = _pair(msorb_mod, oracle, cfg)
    uniq, n = 8, 256
    base = synth.stereo_batch(uniq // 2, cfg["rows"], cfg["cols"], seed0=1000)          # 8 distinct images
    batch = np.concatenate([base] * (n // uniq))
    d = torch.from_numpy(batch).cuda()
    counts, mono, d_kps, d_desc = ex.extract_batch(d)
    kps_all = msorb_mod.keypoints_from_device(d_kps, counts)
    desc_all = d_desc.cpu().numpy()
    assert counts.min() > 1800 and counts.max() <= cfg["nfeatures"] + 2 * cfg["nlevels"]
    assert np.array_equal(mono, counts)                                                  # vLappingArea = {0, 0}
    for i in range(n):
        k = kps_all[i]
        assert np.all(np.diff(k["octave"]) >= 0) and np.all(k["class_id"] == -1)
        j = i % uniq
        if i >= uniq:
            assert counts[i] == counts[j]
            assert np.array_equal(k.view(np.uint8), kps_all[j].view(np.uint8))
            assert np.array_equal(desc_all[i, :counts[i]], desc_all[j, :counts[j]])
    for i in (3, 6):                                                                     # oracle on a sample
        rmono, rkps, rdesc = ref(batch[i])
        assert counts[i] == len(rkps)
        _assert_same(kps_all[i], desc_all[i, :counts[i]], rkps, rdesc)
    # descriptors against themselves: best = self at distance 0 unless an earlier row is identical
    cnt = torch.from_numpy(counts.astype(np.int32)).cuda()
    bi, bd, sd, _ = msorb_mod.hamming_dense_top2_batch(d_desc, d_desc, cnt, cnt)
    bi, bd = bi.cpu().numpy(), bd.cpu().numpy()
    for i in range(0, n, 37):
        c = counts[i]
        assert np.all(bd[i, :c] == 0)
        dup = bi[i, :c] != np.arange(c)
        assert np.all(bi[i, :c][dup] < np.arange(c)[dup])
        assert all(np.array_equal(desc_all[i, a], desc_all[i, b]) for a, b in zip(np.arange(c)[dup], bi[i, :c][dup]))
    ex.close()


def test_misaligned_batch_staged_and_in_place(msorb_mod, oracle):
    """Tightly packed rows that are not 4-byte aligned (EuRoC-sized images cut to 751 pixels): batches of 128 images and more
    have level 0 staged once into the handle's aligned planes and run the aligned kernels, smaller batches keep the rows in
    place and run the byte-granular variants.  Both must equal the oracle, and the caller's images stay untouched."""
    import torch
    cfg = dict(CONFIGS["euroc"], cols=751, rows=240)
    ex, ref = _pair(msorb_mod, oracle, cfg)
    try:
        imgs = [synth.image(800 + i, cfg["rows"], cfg["cols"]) for i in range(3)]
        want = [ref(im) for im in imgs]
        for n in (128, 16):          # staged / in place
            batch = np.stack([imgs[i % 3] for i in range(n)])
            d = torch.from_numpy(batch).cuda()
            assert d.stride(1) % 4 != 0
            counts, mono, d_kps, d_desc = ex.extract_batch(d)
            kps_list = msorb_mod.keypoints_from_device(d_kps, counts)
            desc_all = d_desc.cpu().numpy()
            for i in list(range(6)) + [n - 1]:
                rmono, rkps, rdesc = want[i % 3]
                assert counts[i] == len(rkps) and mono[i] == rmono, (n, i)
                _assert_same(kps_list[i], desc_all[i, :counts[i]], rkps, rdesc)
            assert torch.equal(d, torch.from_numpy(batch).cuda())
    finally:
        ex.close()


def test_batch_submit_wait_two_handles_in_flight(msorb_mod, oracle):
    """msorb_extract_batch_submit / _wait: two handles, four batches kept two deep in flight; every batch equals the synchronous
    call on the same images; a handle refuses a second submit (or any other entry) while a batch is pending."""
    import torch
    cfg = CONFIGS["euroc"]
    n = 16
    batches = [torch.from_numpy(np.stack([synth.image(1200 + 10 * b + (i % 4), cfg["rows"], cfg["cols"]) for i in range(n)])).cuda()
               for b in range(4)]
    ex = [msorb_mod.ORBextractor(cfg["nfeatures"], cfg["scale"], cfg["nlevels"], cfg["ini_th"], cfg["min_th"]) for _ in range(2)]
    ref = msorb_mod.ORBextractor(cfg["nfeatures"], cfg["scale"], cfg["nlevels"], cfg["ini_th"], cfg["min_th"])
    want = []
    for b in batches:
        c, m, k, d = ref.extract_batch(b)
        want.append((c.copy(), m.copy(), k.clone(), d.clone()))
    got = [None] * 4
    ex[0].extract_batch_submit(batches[0])
    with pytest.raises(msorb_mod.MsorbError) as e:
        ex[0].extract_batch_submit(batches[1])
    assert e.value.code == msorb_mod.E_INVALID
    with pytest.raises(msorb_mod.MsorbError):
        ex[0].pyramid_batch(batches[1])
    for b in range(1, 4):
        ex[b & 1].extract_batch_submit(batches[b])          # batch b goes in before batch b - 1 is waited for
        got[b - 1] = ex[(b - 1) & 1].extract_batch_wait()
    got[3] = ex[1].extract_batch_wait()
    with pytest.raises(msorb_mod.MsorbError):
        ex[1].extract_batch_wait()                           # nothing pending
    for b in range(4):
        c, m, k, d = got[b]
        wc, wm, wk, wd = want[b]
        assert np.array_equal(c, wc) and np.array_equal(m, wm)
        for i in range(n):
            assert torch.equal(k[i, :c[i]], wk[i, :c[i]]) and torch.equal(d[i, :c[i]], wd[i, :c[i]])
    for e_ in ex + [ref]:
        e_.close()


@pytest.mark.parametrize("n_images", [4, 32])
def test_texture_classes_at_full_kitti_size_through_the_batch_kernels(msorb_mod, oracle, n_images):
    """The input classes of bench.py's density sweep (msorb/synth.py TEXTURE) at 1241 x 376, 2000 features: on the LOW class a
    third of the reference's cells finds no corner at iniThFAST and takes the minThFAST retry (ORBextractor.cc:843-847 — the second
    pass of fast_cells_kernel, which the other tests of this file reach only at 320 x 240), half of those stay empty; the HIGH
    class saturates cells.  n_images 4 = the frame kernels' shapes (1024-thread quadtree), 32 = the batch shapes, two sub-batches.
    Candidates per level, keypoints and descriptors against the oracle."""
    import torch
    cfg = CONFIGS["kitti"]
    ex, ref = _pair(msorb_mod, oracle, cfg)
    try:
        imgs = np.stack([synth.stereo_pair(880 + i // 2, cfg["rows"], cfg["cols"], texture=("low", "high", "low", "default")[i % 4])[i % 2]
                         for i in range(n_images)])
        counts, mono, d_kps, d_desc = ex.extract_batch(torch.from_numpy(imgs).cuda())
        kps = msorb_mod.keypoints_from_device(d_kps, counts)
        desc = d_desc.cpu().numpy()
        retried = 0
        for i in ([0, 1, 2, 3] if n_images == 4 else [0, 1, 2, 6, 17, 30, 31]):
            rmono, rkps, rdesc = ref(imgs[i])
            assert counts[i] == len(rkps) and mono[i] == rmono, i
            _assert_same(kps[i], desc[i, :counts[i]], rkps, rdesc)
            st = ref.cell_stats()
            if i % 2 == 0:   # low class
                retried += sum(s[1] for s in st)
                assert sum(s[1] for s in st) > 0.2 * sum(s[0] for s in st), st
            if n_images == 4:   # single sub-batch: the candidate lists are inspectable
                for l in range(cfg["nlevels"]):
                    assert np.array_equal(ex.debug_candidates(i, l), ref.candidates(l)), (i, l)
        assert retried > 300
    finally:
        ex.close()


def _hip_extract(msorb_mod, **semantics):
    def run(img, nfeat):
        ex = msorb_mod.ORBextractor(nfeat, 1.2, 8, 20, 7)
        try:
            if semantics:
                ex.set_semantics(**semantics)     # msorb_extractor_set_semantics: what the fixtures call for (pinned_semantics)
            _, kps, desc = ex(img)
        finally:
            ex.close()
        return kps, desc
    return run


def test_hip_extractor_against_the_pin_kit_plumbing_fixtures(msorb_mod, oracle, tmp_path):
    """The consumer the whole-extractor OpenCV fixtures will meet on the GPU box, exercised today on fixtures made by the kit's
    Python restatement around the oracle's primitives (tools/pin_opencv.py --extractor, tests/test_oracle_pins.py): the HIP
    path must reproduce them — four geometries of the kit's integer-hash texture, which no other GPU test uses."""
    import test_oracle_pins as top
    kit = top._kit()
    kit.generate_extractor(top._OracleAsCv(oracle), str(tmp_path))
    bad, _ = top.compare_extractor(_hip_extract(msorb_mod), str(tmp_path))
    assert bad == []


def test_hip_extractor_against_real_opencv_extractor_pins(msorb_mod):
    import test_oracle_pins as top
    if not os.path.exists(os.path.join(top.PINS, "extractor_meta.json")):
        pytest.skip("no whole-extractor OpenCV pins committed yet (python tools/pin_opencv.py --extractor on a machine with cv2)")
    sem = top.pinned_semantics(top.PINS)     # the primitives' selected variant + the tap contraction the fixtures were made with
    bad, meta = top.compare_extractor(_hip_extract(msorb_mod, **sem), top.PINS)
    assert not bad, f"the HIP extractor under {sem} differs from the kit's run on OpenCV {meta['cv2_version']}:\n" + "\n".join(bad)


def test_extract_pair_staged_images_on_the_handles_own_staging_block(msorb_mod, oracle):
    """msorb_stage_image always stages into plane 0 of the handle's pinned block; msorb_extract_pair must not copy an un-staged image
    over a staged one it still has to upload (round-5 advice: staged = 2 on ONE handle returned image a's keypoints for image b).
    Every staged combination, with the host pyramid's level 0 pointing at the right image afterwards."""
    cfg = CONFIGS["small"]
    ex, ref = _pair(msorb_mod, oracle, cfg)
    other = msorb_mod.ORBextractor(cfg["nfeatures"], cfg["scale"], cfg["nlevels"], cfg["ini_th"], cfg["min_th"])
    ex.set_host_pyramid(True)
    L, R = synth.stereo_pair(77, cfg["rows"], cfg["cols"])
    want = [ref(L), ref(R)]
    try:
        for stagers in ((None, ex), (ex, None), (other, ex), (ex, other), (other, None), (None, other), (None, None)):
            (ma, ka, da), (mb, kb, db) = ex.extract_pair(L, R, stage=stagers)
            for which, img, (mono, kps, desc) in ((0, L, (ma, ka, da)), (1, R, (mb, kb, db))):
                rmono, rkps, rdesc = want[which]
                assert mono == rmono, stagers
                _assert_same(kps, desc, rkps, rdesc)
                assert np.array_equal(ex.pyramid_level_image(which, 0), img), f"level 0 of image {which} under staging {stagers}"
            assert np.array_equal(ex.pyramid_level(0), L)      # msorb_pyramid_level = image 0 of the pair
        # the same image staged once and passed for both eyes (staged = 3, one handle)
        (ma, ka, da), (mb, kb, db) = ex.extract_pair(R, R, stage=(ex, ex))
        _assert_same(ka, da, want[1][1], want[1][2])
        _assert_same(kb, db, want[1][1], want[1][2])
    finally:
        ex.close()
        other.close()


def oracle_level(ref, img, level):
    ref(img)
    return ref.level(level)


@pytest.mark.parametrize("name", ["kitti", "small", "fourseasons"])
def test_extract_pair_is_two_operator_calls(msorb_mod, oracle, name):
    """msorb_extract_pair (two images, one kernel chain, no stereo match) == two ORBextractor::operator() calls
    (Frame.cc:122-125's two threads): keypoints, descriptors, monoIndex with a lapping area, and both host pyramids; the
    staged form (msorb_stage_image) and a plain call on the same handle afterwards give the same."""
    cfg = CONFIGS[name]
    ex, ref = _pair(msorb_mod, oracle, cfg)
    ex.set_host_pyramid(True)
    L, R = synth.stereo_pair(321, cfg["rows"], cfg["cols"])
    for lap, stage in (((0, 0), False), ((40, cfg["cols"] // 2), False), ((0, 0), True)):
        (ma, ka, da), (mb, kb, db) = ex.extract_pair(L, R, lap, stage=stage)
        for img, which, mono, kps, desc in ((L, 0, ma, ka, da), (R, 1, mb, kb, db)):
            rmono, rkps, rdesc = ref(img, lap)
            assert mono == rmono
            _assert_same(kps, desc, rkps, rdesc)
            for l in range(cfg["nlevels"]):
                assert np.array_equal(ex.pyramid_level_image(which, l), ref.level(l)), f"image {which} host pyramid level {l}"
    ex.set_host_pyramid(False)                    # without it the pyramids are fetched on first request
    ex.extract_pair(L, R)
    assert np.array_equal(ex.pyramid_level_image(1, 2), oracle_level(ref, R, 2))
    assert np.array_equal(ex.pyramid_level_image(0, cfg["nlevels"] - 1), oracle_level(ref, L, cfg["nlevels"] - 1))
    mono, kps, desc = ex(R)                       # the handle goes back to single-image calls
    rmono, rkps, rdesc = ref(R)
    _assert_same(kps, desc, rkps, rdesc)
    assert np.array_equal(ex.pyramid_level(3), ref.level(3))
    with pytest.raises(msorb_mod.MsorbError):
        ex.pyramid_level_image(1, 0)              # image 1 exists only after a pair call
    ex.close()
