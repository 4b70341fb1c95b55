"""GPU parity tests: HIP extractor (through the C ABI) vs the CPU oracle, stage by stage and end to end.
Bit-exact for every integer/byte product (pyramid, blur, candidates incl. FAST scores, descriptors) and for
the float keypoint fields (same IEEE operations in the same order)."""
import numpy as np
import pytest

from msorb import synth

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
@pytest.mark.parametrize("kernel", ["dma", "band", "band_lds", "rows"])
def test_batch_pyramid_kernels_on_padded_rows(msorb_mod, oracle, monkeypatch, name, kernel):
    """The batch pyramid kernels (LDS-free band kernel with the parked rows in registers = default, the same with the rows
    parked in LDS, LDS-DMA band kernel on 16-byte aligned rows, row-streaming kernel) on a batch whose rows are padded to a
    multiple of 64 bytes, as bench.py lays its images out: every level of every distinct image is the oracle's cv::resize
    restatement, bit for bit."""
    import torch
    cfg = CONFIGS[name] if name != "odd" else dict(rows=333, cols=517, nfeatures=700, scale=1.2, nlevels=8, ini_th=20, min_th=7)
    if kernel == "dma":
        monkeypatch.setenv("MSORB_PYR_DMA", "1")
    elif kernel == "rows":
        monkeypatch.setenv("MSORB_PYR_ROWS", "1")
    elif kernel == "band_lds":
        monkeypatch.setenv("MSORB_PYR_BAND_LDS", "1")
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
            assert np.array_equal(ex.debug_level(i, lvl), ref.level(lvl)), (kernel, i, lvl)
    ex.close()


@pytest.mark.parametrize("name", ["kitti", "euroc", "odd"])
def test_batch_pyramid_fused_tail_levels(msorb_mod, oracle, monkeypatch, name):
    """MSORB_PYR_TAIL=1: batches of >= 64 images build their last three pyramid levels in one launch (pyr_resize_tail_kernel: a
    workgroup per image walks down the levels behind workgroup barriers; opt-in, see launch_pyramid): every level of a sample
    of the images is the oracle's, and the default launch-per-level path gives the same bytes."""
    import torch
    cfg = CONFIGS[name] if name != "odd" else dict(rows=333, cols=517, nfeatures=700, scale=1.2, nlevels=8, ini_th=20, min_th=7)
    ex, ref = _pair(msorb_mod, oracle, cfg)
    n = 64
    pitch = (cfg["cols"] + 63) // 64 * 64
    batch = np.stack([synth.image(950 + (i % 5), cfg["rows"], cfg["cols"]) for i in range(n)])
    store = torch.zeros((n, cfg["rows"], pitch), dtype=torch.uint8, device="cuda")
    view = store[:, :, :cfg["cols"]]
    view.copy_(torch.from_numpy(batch).cuda())
    monkeypatch.setenv("MSORB_PYR_TAIL", "1")
    ex.pyramid_batch(view)
    torch.cuda.synchronize()
    fused = {(i, l): ex.debug_level(i, l) for i in (0, 3, 4, 37, n - 1) for l in range(1, cfg["nlevels"])}
    for i in (0, 3, 4, 37, n - 1):
        ref(batch[i])
        for lvl in range(1, cfg["nlevels"]):
            assert np.array_equal(fused[(i, lvl)], ref.level(lvl)), (i, lvl)
    monkeypatch.delenv("MSORB_PYR_TAIL")
    ex.pyramid_batch(view)
    torch.cuda.synchronize()
    for (i, l), want in fused.items():
        assert np.array_equal(ex.debug_level(i, l), want), (i, l)
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


@pytest.mark.parametrize("name", ["kitti", "euroc", "fourseasons"])
def test_fast_strip_and_cell_forms_agree_with_the_oracle(msorb_mod, oracle, monkeypatch, name):
    """The FAST stage as strips of up to four cells per workgroup (fast_strip_kernel, batches) and as one workgroup per cell
    (fast_cells_kernel): same candidates in the same order as the oracle's cell loop, on scenes, on noise (every strip has
    more quick-test survivors than its work list holds: redone cell by cell), on low contrast (cells empty at iniThFAST are
    redone at minThFAST, per cell), on half-flat images (strips with empty AND full cells) — and the dataset geometries
    really take the strip form when it is asked for (MSORB_FAST_STRIP=1; the per-cell form is the default)."""
    import torch
    cfg = CONFIGS[name]
    rows, cols = cfg["rows"], cfg["cols"]
    rng = np.random.Generator(np.random.PCG64(41))
    scene = synth.image(311, rows, cols)
    noise = rng.integers(0, 256, (rows, cols), dtype=np.uint8)
    low = (scene.astype(np.int32) // 6 + 100).astype(np.uint8)
    half = scene.copy(); half[:, : cols // 2] = 77
    stripes = scene.copy(); stripes[:, ::97] = 255; stripes[::53, :] = 0
    imgs = [scene, noise, low, half, stripes]
    ex = msorb_mod.ORBextractor(cfg["nfeatures"], 1.2, 8, 20, 7)
    ref = oracle.OracleExtractor(cfg["nfeatures"], 1.2, 8, 20, 7)
    try:
        want = []
        for im in imgs:
            ref(im)
            want.append([ref.candidates(l) for l in range(8)])
        # rows padded to a multiple of 64 bytes, as bench.py lays its images out (the strip form stages 16-byte quads: it needs
        # 4-byte aligned rows; tightly packed 1241-byte rows take the per-cell kernel's byte path until a batch is big enough
        # to be re-staged)
        pitch = (cols + 63) // 64 * 64
        store = torch.zeros((16, rows, pitch), dtype=torch.uint8, device="cuda")
        d = store[:, :, :cols]
        d.copy_(torch.from_numpy(np.stack([imgs[i % len(imgs)] for i in range(16)])).cuda())
        ex.set_overlap(1, False)
        for form in ("1", "default", "0"):
            if form == "default":
                monkeypatch.delenv("MSORB_FAST_STRIP", raising=False)
            else:
                monkeypatch.setenv("MSORB_FAST_STRIP", form)
            monkeypatch.setenv("MSORB_GROUPS", "1")            # candidate inspection needs one sub-batch
            counts, monos, d_kps, d_desc = ex.extract_batch(d)
            assert ex.debug_fast_form() == (1 if form == "1" else 0), "the dataset geometries can take the strip form"
            for i in (0, 1, 2, 3, 4, 15):
                for l in range(8):
                    assert np.array_equal(ex.debug_candidates(i, l), want[i % len(imgs)][l]), (form, i, l)
        monkeypatch.delenv("MSORB_FAST_STRIP", raising=False)
        mono, kps, desc = ex(scene)                             # one frame: the per-cell form
        assert ex.debug_fast_form() == 0
    finally:
        ex.close()


@pytest.mark.parametrize("name", ["kitti", "euroc", "fourseasons", "odd", "small"])
def test_blur_on_the_matrix_cores_matches_the_oracle(msorb_mod, oracle, monkeypatch, name):
    """gauss7_mfma_kernel (banded matrix products: i8 MFMA for the rows, fp32 MFMA for the columns) against the oracle's
    GaussianBlur restatement, every level of a batch on 64-byte padded rows: scenes, noise, black / white / saturated images
    (the -128 bias and the 2^24 exactness bound), widths and heights that are not multiples of 32 (partial strips / blocks, both
    reflected borders inside one tile), default and alternative taps (sum 257: saturation); and the VALU form on the same batch."""
    import torch
    cfg = CONFIGS[name] if name != "odd" else dict(rows=333, cols=517, nfeatures=700, scale=1.2, nlevels=8, ini_th=20, min_th=7)
    rows, cols = cfg["rows"], cfg["cols"]
    rng = np.random.Generator(np.random.PCG64(5))
    imgs = [synth.image(411, rows, cols), rng.integers(0, 256, (rows, cols), dtype=np.uint8), np.zeros((rows, cols), np.uint8),
            np.full((rows, cols), 255, np.uint8), (rng.integers(0, 2, (rows, cols), dtype=np.uint8) * 255)]
    pitch = (cols + 63) // 64 * 64
    store = torch.zeros((16, rows, pitch), dtype=torch.uint8, device="cuda")
    d = store[:, :, :cols]
    d.copy_(torch.from_numpy(np.stack([imgs[i % len(imgs)] for i in range(16)])).cuda())
    ex, ref = _pair(msorb_mod, oracle, cfg)
    try:
        ex.set_overlap(1, False)
        for taps in (None, [18, 34, 49, 55, 49, 34, 18], [16, 32, 48, 64, 48, 32, 16]):
            ex.set_semantics(taps)
            oracle.set_semantics(taps)
            want = []
            for im in imgs:
                ref(im)   # (an image without keypoints never reaches the oracle's blur: blur its levels directly)
                want.append([oracle.gaussian7(ref.level(l)) for l in range(cfg["nlevels"])])
            for form in ("1", "0"):
                monkeypatch.setenv("MSORB_BLUR_MFMA", form)
                ex.extract_batch(d)
                # (taps summing to 257 need a 17-bit row sum: the matrix-core form leaves them to the VALU kernels)
                assert ex.debug_blur_form() == (int(form) if taps is None or sum(taps) <= 256 else 0)
                for i in (0, 1, 2, 3, 4, 15):
                    for l in range(cfg["nlevels"]):
                        got = ex.debug_level(i, l, blurred=True)
                        assert np.array_equal(got, want[i % len(imgs)][l]), (name, taps, form, i, l, np.argwhere(got != want[i % len(imgs)][l])[:4])
    finally:
        oracle.set_semantics()
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


def test_graph_replay_path_matches(tmp_path):
    """MSORB_GRAPH=1: msorb_extract replays the captured chain (third call onwards); results must equal the plain path and
    the oracle, also across a change of the lapping area (second cached graph) and of the geometry (graphs dropped)."""
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import os, sys
import numpy as np
sys.path[:0] = [os.path.join(ROOT, "ms-slam_amd"), os.path.join(ROOT, "oracle")]
import msorb, orb_oracle
from msorb import synth
ex = msorb.ORBextractor(1000, 1.2, 8, 20, 7)
ref = orb_oracle.OracleExtractor(1000, 1.2, 8, 20, 7)
def same(a, b):
    return a[0] == b[0] and np.array_equal(a[1].view(np.uint8), b[1].view(np.uint8)) and np.array_equal(a[2], b[2])
imgs = [synth.image(50 + i, 240, 320) for i in range(4)]
for rep in range(3):
    for i, im in enumerate(imgs):
        assert same(ex(im), ref(im)), (rep, i)
        assert same(ex(im, (100, 200)), ref(im, (100, 200))), (rep, i, "lap")
big = synth.image(9, 376, 1241)
for rep in range(3):
    assert same(ex(big), ref(big))
    assert same(ex(imgs[0]), ref(imgs[0]))
lvl = ex.debug_level(0, 3)
assert lvl.shape == ref.level(3).shape
print("graph-ok")
'''.replace("ROOT", repr(ROOT))
    env = dict(os.environ, MSORB_GRAPH="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert "graph-ok" in out.stdout, out.stdout + out.stderr


def test_misaligned_batch_staged_and_in_place(msorb_mod, oracle):
    """Tightly packed rows that are not 4-byte aligned (KITTI: 1241 pixels): by default level 0 is staged once into the
    handle's aligned planes and the aligned kernels run; MSORB_NO_STAGE0 keeps the byte-granular in-place variants.
    Both must equal the oracle."""
    import os
    import torch
    cfg = CONFIGS["kitti"]
    ex, ref = _pair(msorb_mod, oracle, cfg)
    n = 16                                             # MSORB_STAGE0_MIN=1 below: staging normally starts at 128 images
    batch = np.stack([synth.image(800 + (i % 3), cfg["rows"], cfg["cols"]) for i in range(n)])
    d = torch.from_numpy(batch).cuda()
    assert d.stride(1) % 4 != 0
    want = [ref(batch[i]) for i in range(3)]
    try:
        for mode in ("staged", "in_place"):
            os.environ["MSORB_STAGE0_MIN"] = "1"
            if mode == "in_place":
                os.environ["MSORB_NO_STAGE0"] = "1"
            else:
                os.environ.pop("MSORB_NO_STAGE0", None)
            counts, mono, d_kps, d_desc = ex.extract_batch(d)
            kps_list = msorb_mod.keypoints_from_device(d_kps, counts)
            desc_all = d_desc.cpu().numpy()
            for i in range(n):
                rmono, rkps, rdesc = want[i % 3]
                assert counts[i] == len(rkps) and mono[i] == rmono, (mode, i)
                _assert_same(kps_list[i], desc_all[i, :counts[i]], rkps, rdesc)
            # the staged copy must not touch the caller's images
            assert torch.equal(d, torch.from_numpy(batch).cuda())
    finally:
        os.environ.pop("MSORB_NO_STAGE0", None)
        os.environ.pop("MSORB_STAGE0_MIN", None)
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
