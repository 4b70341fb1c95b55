"""GPU parity tests of the matcher path: device kernels + host replay (through the C ABI) vs the CPU oracle.
Everything here is integer/index work: match indices, counts and the stereo uRight/depth floats must be
bit-identical."""
import os

import numpy as np
import pytest

from msorb import synth
import matcher_cases as mc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def stereo_frame(msorb_mod, oracle):
    cfg = synth.KITTI
    L, R = synth.stereo_pair(11, cfg["rows"], cfg["cols"])
    exl = msorb_mod.ORBextractor(2000, 1.2, 8, 20, 7)
    exr = msorb_mod.ORBextractor(2000, 1.2, 8, 20, 7)
    _, kl, dl = exl(L)
    _, kr, dr = exr(R)
    return dict(cfg=cfg, L=L, R=R, exl=exl, exr=exr, kl=kl, dl=dl, kr=kr, dr=dr, scale=exl.GetScaleFactors(),
                inv_scale=exl.GetInverseScaleFactors())


def test_stereo_matches_bit_exact(msorb_mod, oracle, stereo_frame):
    s = stereo_frame
    mbf, mb = mc.KITTI_BF, mc.KITTI_BF / mc.KITTI_FX
    ur, dp, oob = msorb_mod.stereo_matches(s["exl"], s["exr"], s["kl"], s["dl"], s["kr"], s["dr"], mb, mbf)
    pl = [s["exl"].pyramid_level(l) for l in range(8)]
    pr = [s["exr"].pyramid_level(l) for l in range(8)]
    rur, rdp, roob = oracle.compute_stereo_matches(s["kl"], s["dl"], s["kr"], s["dr"], pl, pr, s["scale"],
                                                   s["inv_scale"], mb, mbf)
    assert (rur > 0).sum() > 500
    assert np.array_equal(ur.view(np.uint32), rur.view(np.uint32))
    assert np.array_equal(dp.view(np.uint32), rdp.view(np.uint32))
    assert oob == roob
    # one extractor per GPU (MSORB_DEVICES=0,1): the right pyramid is pulled to the left device level by level, peer to peer;
    # forced here with both handles on one device (the switch is read when a handle is created)
    os.environ["MSORB_FORCE_PEER_PYRAMID"] = "1"
    try:
        exl2 = msorb_mod.ORBextractor(2000, 1.2, 8, 20, 7)
        exr2 = msorb_mod.ORBextractor(2000, 1.2, 8, 20, 7)
    finally:
        del os.environ["MSORB_FORCE_PEER_PYRAMID"]
    try:
        exl2(s["L"])
        exr2(s["R"])
        ur2, dp2, oob2 = msorb_mod.stereo_matches(exl2, exr2, s["kl"], s["dl"], s["kr"], s["dr"], mb, mbf)
    finally:
        exl2.close()
        exr2.close()
    assert np.array_equal(ur2.view(np.uint32), rur.view(np.uint32)) and np.array_equal(dp2.view(np.uint32), rdp.view(np.uint32)) and oob2 == roob
    # degenerate: no right keypoints / no left keypoints
    ur0, dp0, _ = msorb_mod.stereo_matches(s["exl"], s["exr"], s["kl"], s["dl"], s["kr"][:0], s["dr"][:0], mb, mbf)
    assert np.all(ur0 == -1) and np.all(dp0 == -1)
    ur1, _, _ = msorb_mod.stereo_matches(s["exl"], s["exr"], s["kl"][:0], s["dl"][:0], s["kr"], s["dr"], mb, mbf)
    assert len(ur1) == 0


def _frames(msorb_mod, oracle, s, ur):
    bounds = (0.0, float(s["cfg"]["cols"]), 0.0, float(s["cfg"]["rows"]))
    return (msorb_mod.Frame(s["kl"], s["dl"], ur, bounds, s["scale"]),
            oracle.OracleFrame(s["kl"], s["dl"], ur, bounds, s["scale"]))


def test_features_in_area_order(msorb_mod, oracle, stereo_frame):
    s = stereo_frame
    f, rf = _frames(msorb_mod, oracle, s, None)
    rng = np.random.Generator(np.random.PCG64(5))
    for _ in range(300):
        x, y = rng.uniform(-50, 1300), rng.uniform(-50, 420)
        r = rng.uniform(1, 120)
        lv = [(-1, -1), (0, 2), (3, -1), (2, 3), (0, 0)][rng.integers(0, 5)]
        assert np.array_equal(f.GetFeaturesInArea(x, y, r, *lv), rf.GetFeaturesInArea(x, y, r, *lv))


@pytest.mark.parametrize("seed,M,th,obs_zero,spars", [(1, 4096, 1.0, 0.15, 0.05), (2, 4096, 3.0, 0.0, 0.0),
                                                      (3, 6000, 5.0, 0.5, 0.3), (4, 1500, 15.0, 0.1, 0.02)])
def test_search_by_projection_map_points(msorb_mod, oracle, stereo_frame, seed, M, th, obs_zero, spars):
    s = stereo_frame
    rng = np.random.Generator(np.random.PCG64(seed))
    mbf, mb = mc.KITTI_BF, mc.KITTI_BF / mc.KITTI_FX
    ur, _, _ = msorb_mod.stereo_matches(s["exl"], s["exr"], s["kl"], s["dl"], s["kr"], s["dr"], mb, mbf)
    f, rf = _frames(msorb_mod, oracle, s, ur)
    mp = mc.map_point_table(rng, s["kl"], s["dl"], ur, s["scale"], M, obs_zero, spars)
    init = np.where(rng.random(len(s["kl"])) < 0.2, rng.integers(0, M, len(s["kl"])), -1).astype(np.int32)
    got, want = init.copy(), init.copy()
    n = f.SearchByProjection_mps(mp, got, th, bFarPoints=True, thFarPoints=60.0, nnratio=0.8)
    rn = rf.SearchByProjection_mps(mp, want, th, bFarPoints=True, thFarPoints=60.0, nnratio=0.8)
    assert rn > 100
    assert n == rn and np.array_equal(got, want)


@pytest.mark.parametrize("seed,NL,th,mode", [(1, 2000, 7.0, "none"), (2, 2000, 15.0, "fwd"), (3, 3000, 14.0, "bwd"),
                                             (4, 500, 30.0, "none")])
def test_search_by_projection_last_frame(msorb_mod, oracle, stereo_frame, seed, NL, th, mode):
    s = stereo_frame
    rng = np.random.Generator(np.random.PCG64(100 + seed))
    ur = np.where(rng.random(len(s["kl"])) < 0.6, s["kl"]["x"] - rng.uniform(1, 40, len(s["kl"])), -1).astype(np.float32)
    f, rf = _frames(msorb_mod, oracle, s, ur)
    last = mc.last_frame_table(rng, s["kl"], s["dl"], ur, s["scale"], NL)
    init = np.where(rng.random(len(s["kl"])) < 0.1, rng.integers(0, NL, len(s["kl"])), -1).astype(np.int32)
    for check in (True, False):
        got, want = init.copy(), init.copy()
        kw = dict(forward=mode == "fwd", backward=mode == "bwd", check_orientation=check)
        n = f.SearchByProjection_frames(last, got, th, **kw)
        rn = rf.SearchByProjection_frames(last, want, th, **kw)
        assert n == rn and np.array_equal(got, want)
    assert rn > 50


def test_hamming_top2_lists(msorb_mod, oracle):
    rng = np.random.Generator(np.random.PCG64(9))
    T, Q = 1500, 700
    t = rng.integers(0, 256, (T, 32), dtype=np.uint8)
    t[100:140] = t[100]                       # exact duplicates -> ties broken by list order
    q = t[rng.integers(0, T, Q)].copy()
    q[::3, 5] ^= 0x11
    lens = rng.integers(0, 90, Q)
    lens[:5] = [0, 1, 1, 2, 300]
    cb = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ci = rng.integers(0, T, cb[-1]).astype(np.int32)
    bi, bd, si, sd = msorb_mod.hamming_top2(q, t, cb, ci)
    for i in range(Q):
        best, bdist, sec, sdist = -1, 256, -1, 256     # ORBmatcher.cc:300-318 idiom
        for j in ci[cb[i]:cb[i + 1]]:
            d = oracle.descriptor_distance(q[i], t[j])
            if d < bdist:
                sec, sdist = best, bdist
                best, bdist = j, d
            elif d < sdist:
                sec, sdist = j, d
        assert (bi[i], bd[i], sd[i]) == (best, bdist, sdist)
        if sec >= 0:
            assert oracle.descriptor_distance(q[i], t[si[i]]) == sdist


def test_dense_top2_batch(msorb_mod, oracle):
    import torch
    rng = np.random.Generator(np.random.PCG64(21))
    F, cap = 3, 512
    nq = np.array([500, 37, 0], np.int32)
    nt = np.array([450, 512, 100], np.int32)
    q = rng.integers(0, 256, (F, cap, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (F, cap, 32), dtype=np.uint8)
    t[0, 100:110] = t[0, 100]                       # duplicates: lowest index must win
    q[0, :50] = t[0, rng.integers(0, 450, 50)]      # exact matches
    bi, bd, sd, ms = msorb_mod.hamming_dense_top2_batch(torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda(),
                                                       torch.from_numpy(nq).cuda(), torch.from_numpy(nt).cuda())
    bi, bd, sd = bi.cpu().numpy(), bd.cpu().numpy(), sd.cpu().numpy()
    for f in range(F):
        if nq[f] == 0:
            continue
        d = np.unpackbits(q[f, :nq[f], None, :] ^ t[f, None, :nt[f], :], axis=2).sum(2)     # [nq, nt]
        order = np.argsort(d, axis=1, kind="stable")
        assert np.array_equal(bi[f, :nq[f]], order[:, 0])
        assert np.array_equal(bd[f, :nq[f]], np.take_along_axis(d, order[:, :1], 1)[:, 0])
        assert np.array_equal(sd[f, :nq[f]], np.take_along_axis(d, order[:, 1:2], 1)[:, 0])
    assert oracle.descriptor_distance(q[0, 0], t[0, bi[0, 0]]) == bd[0, 0] == 0


def test_stereo_windows_at_the_image_borders(msorb_mod, oracle):
    """Hand-placed keypoints whose 11 x 11 / 11 x 21 SAD windows touch the first and last rows and columns of every pyramid
    level, on a tightly packed batch (row pitch = 1241 bytes, level 0 read in place, no row 4-byte aligned): the kernel stages
    the windows as whole dwords, so a window's last dword can reach past the end of its level plane — for the last image, past
    the caller's buffer — and must be fetched from inside and shifted.  The right image is the left one moved 6 pixels (+ noise),
    so every planted pair (disparity 6) has its SAD minimum in the middle of the search range and survives the parabola /
    median steps: uRight, depth and the out-of-bounds count must equal the oracle's byte-wise restatement of Frame.cc:829-897."""
    import torch
    cfg = synth.KITTI
    host = synth.stereo_batch(2, cfg["rows"], cfg["cols"], seed0=77)
    nz = np.random.Generator(np.random.PCG64(9)).integers(-3, 4, host[2].shape)
    host[3] = np.clip(np.roll(host[2], -6, axis=1).astype(np.int32) + nz, 0, 255).astype(np.uint8)
    mbf, mb = mc.KITTI_BF, mc.KITTI_BF / mc.KITTI_FX
    ex = msorb_mod.ORBextractor(2000, 1.2, 8, 20, 7)
    try:
        d_img = torch.from_numpy(host).cuda()
        assert d_img.stride(1) == cfg["cols"]
        counts, _, d_kps, d_desc = ex.extract_batch(d_img, (0, 0))
        scale = np.asarray(ex.GetScaleFactors(), np.float32)
        inv = np.asarray(ex.GetInverseScaleFactors(), np.float32)
        rng = np.random.Generator(np.random.PCG64(3))
        kl, kr = [], []
        for o in range(8):
            rows_o, cols_o = ex.debug_level(2, o).shape
            for dv in range(0, 3):
                for du in range(0, 4):
                    # bottom right: right strip columns su - 10 .. su + 10 with su + 11 < cols; the left window of du == 0
                    # ends on the level's last column, the windows of dv == 0 on its last row
                    su = cols_o - 12 - du
                    kr.append((np.float32(su) * scale[o], np.float32(rows_o - 6 - dv) * scale[o], o))
                    kl.append((np.float32(su + 6) * scale[o], np.float32(rows_o - 6 - dv) * scale[o], o))
                    # top left: the right strip starts on column du, the windows on row dv
                    kr.append((np.float32(10 + du) * scale[o], np.float32(5 + dv) * scale[o], o))
                    kl.append((np.float32(16 + du) * scale[o], np.float32(5 + dv) * scale[o], o))
        n = len(kl)
        K = oracle.KP_DTYPE
        kpl, kpr = np.zeros(n, K), np.zeros(n, K)
        for arr, lst in ((kpl, kl), (kpr, kr)):
            arr["x"] = [t[0] for t in lst]; arr["y"] = [t[1] for t in lst]; arr["octave"] = [t[2] for t in lst]
            arr["size"] = 31; arr["angle"] = 0; arr["response"] = 50; arr["class_id"] = -1
        desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)   # pair i shares one descriptor: distance 0
        kps_np = d_kps.cpu().numpy().copy()
        desc_np = d_desc.cpu().numpy().copy()
        for img, arr in ((2, kpl), (3, kpr)):
            kps_np[img, :n] = arr.view(np.uint8).reshape(n, 28)
            desc_np[img, :n] = desc
        cnt = counts.copy()
        cnt[2] = cnt[3] = n
        d_ur, d_dp, oob, _ = msorb_mod.stereo_matches_batch(ex, cnt, torch.from_numpy(kps_np).cuda(), torch.from_numpy(desc_np).cuda(), mb, mbf)
        pl = [ex.debug_level(2, l) for l in range(8)]
        pr = [ex.debug_level(3, l) for l in range(8)]
        rur, rdp, roob = oracle.compute_stereo_matches(kpl, desc, kpr, desc, pl, pr, scale, inv, mb, mbf)
        ur, dp = d_ur.cpu().numpy()[1, :n], d_dp.cpu().numpy()[1, :n]
        assert np.array_equal(ur.view(np.uint32), rur.view(np.uint32))
        assert np.array_equal(dp.view(np.uint32), rdp.view(np.uint32))
        assert oob[1] == roob and (rur > 0).sum() > n // 3
    finally:
        ex.close()


@pytest.mark.parametrize("seed,pad", [(3, 0), (9, 0), (5, 3), (6, 39)])
def test_stereo_random_keypoints_against_oracle(msorb_mod, oracle, seed, pad):
    """Random hand-placed keypoints at all octaves — anywhere in the image, 40 % of them on the rows / columns where the SAD
    windows start or stop fitting — with planted matches at random Hamming distances, on tight (1241-byte: no row is 4-byte
    aligned, a window's last dword can reach past the end of the caller's buffer) and padded rows: uRight, depth and the
    out-of-bounds count equal the oracle's.  (A run of tools/_fuzz_stereo.py found the case the fixed border test missed.)"""
    import torch
    cfg = synth.KITTI
    rng = np.random.default_rng(seed)
    mbf, mb = mc.KITTI_BF, mc.KITTI_BF / mc.KITTI_FX
    host = synth.stereo_batch(2, cfg["rows"], cfg["cols"], seed0=300 + seed)
    pitch = cfg["cols"] + pad
    ex = msorb_mod.ORBextractor(2000, 1.2, 8, 20, 7)
    try:
        flat = torch.zeros(4 * cfg["rows"] * pitch + 16, dtype=torch.uint8, device="cuda")
        view = flat[:4 * cfg["rows"] * pitch].view(4, cfg["rows"], pitch)[:, :, :cfg["cols"]]
        view.copy_(torch.from_numpy(host).cuda())
        counts, _, d_kps, d_desc = ex.extract_batch(view)
        scale = np.asarray(ex.GetScaleFactors(), np.float32)
        inv = np.asarray(ex.GetInverseScaleFactors(), np.float32)
        pl = [ex.debug_level(2, l) for l in range(8)]
        pr = [ex.debug_level(3, l) for l in range(8)]
        n = 1800
        K = oracle.KP_DTYPE
        kpl, kpr = np.zeros(n, K), np.zeros(n, K)
        for arr in (kpl, kpr):
            arr["size"] = 31; arr["angle"] = 0; arr["response"] = 50; arr["class_id"] = -1
        for i in range(n):
            o = int(rng.integers(0, 8))
            rows_o, cols_o = pl[o].shape
            edge = rng.random() < 0.4
            sv = int(rng.choice([0, 3, 5, 6, rows_o - 7, rows_o - 6, rows_o - 5, rows_o - 1])) if edge else int(rng.integers(0, rows_o))
            su = int(rng.choice([0, 4, 5, 9, 10, 11, cols_o - 12, cols_o - 11, cols_o - 6, cols_o - 5, cols_o - 1])) if edge else int(rng.integers(0, cols_o))
            sr = max(0, su - int(rng.integers(0, 40)))
            kpl[i]["x"], kpl[i]["y"], kpl[i]["octave"] = np.float32(su) * scale[o], np.float32(sv) * scale[o], o
            kpr[i]["x"], kpr[i]["y"] = np.float32(sr) * scale[o], np.float32(sv) * scale[o] + np.float32(rng.uniform(-1.5, 1.5))
            kpr[i]["octave"] = int(np.clip(o + rng.integers(-1, 2), 0, 7))
        desc_l = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        desc_r = mc.flip_bits(rng, desc_l, 90)
        kps_np, desc_np = d_kps.cpu().numpy().copy(), d_desc.cpu().numpy().copy()
        kps_np[2, :n] = kpl.view(np.uint8).reshape(n, 28); desc_np[2, :n] = desc_l
        kps_np[3, :n] = kpr.view(np.uint8).reshape(n, 28); desc_np[3, :n] = desc_r
        cnt = counts.copy()
        cnt[2] = cnt[3] = n
        d_ur, d_dp, oob, _ = msorb_mod.stereo_matches_batch(ex, cnt, torch.from_numpy(kps_np).cuda(), torch.from_numpy(desc_np).cuda(), mb, mbf)
        rur, rdp, roob = oracle.compute_stereo_matches(kpl, desc_l, kpr, desc_r, pl, pr, scale, inv, mb, mbf)
        ur, dp = d_ur.cpu().numpy()[1, :n], d_dp.cpu().numpy()[1, :n]
        assert np.array_equal(ur.view(np.uint32), rur.view(np.uint32))
        assert np.array_equal(dp.view(np.uint32), rdp.view(np.uint32))
        assert oob[1] == roob and roob > 100 and (rur > 0).sum() > 100
    finally:
        ex.close()


@pytest.mark.parametrize("kernel", ["mfma", "valu"])
def test_dense_top2_kernels_edge_cases(msorb_mod, oracle, kernel):
    """The matrix-core kernel (default: +-32 int8 encoding, the MFMA accumulator is the (distance << 11 | index) key) and the
    xor / popcount kernel on the shapes that stress tiling: train counts around the 32-train tile and the 2048 cap, query
    counts around the 512-query workgroup and 32-query fragment, one train, identical descriptors everywhere (every distance
    ties: lowest index wins, second = the same distance), all-zero vs all-one descriptors (distance 256)."""
    import torch
    form = msorb_mod.DENSE_POPCOUNT if kernel == "valu" else msorb_mod.DENSE_MATRIX_CORES
    rng = np.random.Generator(np.random.PCG64(77))
    cap = 2048                                       # msorb_hamming_dense_top2_batch: at most 2048 trains per frame
    shapes = [(1, 1), (31, 33), (33, 31), (64, 32), (513, 2047), (2000, 2048), (2048, 1999), (5, 0), (300, 1), (129, 65)]
    F = len(shapes) + 2
    nq = np.array([a for a, _ in shapes] + [200, 100], np.int32)
    nt = np.array([b for _, b in shapes] + [300, 64], np.int32)
    q = rng.integers(0, 256, (F, cap, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (F, cap, 32), dtype=np.uint8)
    q[F - 2] = 0x5A; t[F - 2] = 0x5A                 # every pair at distance 0
    q[F - 1] = 0x00; t[F - 1] = 0xFF                 # every pair at distance 256
    t[4, 1000:1040] = t[4, 1000]                     # a run of duplicates across a tile boundary
    q[4, :100] = t[4, rng.integers(990, 1050, 100)]
    bi, bd, sd, _ = msorb_mod.hamming_dense_top2_batch(torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda(),
                                                      torch.from_numpy(nq).cuda(), torch.from_numpy(nt).cuda(), formulation=form)
    bi, bd, sd = bi.cpu().numpy(), bd.cpu().numpy(), sd.cpu().numpy()
    for f in range(F):
        n_t = int(nt[f])
        if nq[f] == 0:
            continue
        if n_t == 0:
            assert np.all(bi[f, :nq[f]] == -1) and np.all(bd[f, :nq[f]] == 256) and np.all(sd[f, :nq[f]] == 256)
            continue
        wi, wd, ws = oracle.dense_top2(q[f, :nq[f]], t[f, :n_t])
        assert np.array_equal(bi[f, :nq[f]], wi), (kernel, f)
        assert np.array_equal(bd[f, :nq[f]], wd), (kernel, f)
        assert np.array_equal(sd[f, :nq[f]], ws), (kernel, f)
    assert np.all(bi[F - 2, :200] == 0) and np.all(bd[F - 2, :200] == 0) and np.all(sd[F - 2, :200] == 0)
    assert np.all(bd[F - 1, :100] == 256) and np.all(sd[F - 1, :100] == 256) and np.all(bi[F - 1, :100] == -1)   # strict '<' from 256


def test_dense_top2_kernels_on_extracted_descriptors_at_bench_size(msorb_mod, oracle):
    """bench.py's hamming_match leg as a test: 128 stereo pairs of KITTI-sized images through the extractor, the left-eye
    descriptors of every pair against the right-eye ones (~2000 x ~2000 rBRIEF descriptors per frame, duplicates and all) on
    BOTH dense kernels — matrix cores and xor / popcount — compared with each other on every valid row and with the CPU oracle
    on sampled frames (ORBmatcher::DescriptorDistance brute force, ORBmatcher.cc:2323-2339; the knnMatch(k = 2) shape of
    Frame.cc:1076).  Rows past a frame's query count are not results."""
    import torch
    cfg = synth.KITTI
    base = [img for s in range(8) for img in synth.stereo_pair(s, cfg["rows"], cfg["cols"])]
    host = np.stack([base[i % 16] for i in range(256)])
    ex = msorb_mod.ORBextractor(cfg["nfeatures"], cfg["scale"], cfg["nlevels"], cfg["ini_th"], cfg["min_th"])
    try:
        images = torch.from_numpy(host).cuda()
        counts, _, d_kps, d_desc = ex.extract_batch(images, (0, 0))
        dq, dt = d_desc[0::2].contiguous(), d_desc[1::2].contiguous()
        nq = torch.from_numpy(np.ascontiguousarray(counts[0::2])).cuda()
        nt = torch.from_numpy(np.ascontiguousarray(counts[1::2])).cuda()
        assert int(nq.min()) > 1500 and int(nt.min()) > 1500
        mfma = msorb_mod.hamming_dense_top2_batch(dq, dt, nq, nt, formulation=msorb_mod.DENSE_MATRIX_CORES)[:3]
        valu = msorb_mod.hamming_dense_top2_batch(dq, dt, nq, nt, formulation=msorb_mod.DENSE_POPCOUNT)[:3]
        live = torch.arange(dq.shape[1], device=dq.device)[None, :] < nq[:, None]
        for a, b, name in zip(mfma, valu, ("best_idx", "best_dist", "second_dist")):
            assert torch.equal(a[live], b[live]), name
            # the wrapper presets the never-written rows: no stale memory in the tensors either
            assert torch.equal(a, b), name + " (padding rows)"
        for f in (0, 5, 77, 127):
            n0, n1 = int(counts[2 * f]), int(counts[2 * f + 1])
            want = oracle.dense_top2(dq[f, :n0].cpu().numpy(), dt[f, :n1].cpu().numpy())
            for kern, got in (("mfma", mfma), ("popcount", valu)):
                for w, g in zip(want, got):
                    assert np.array_equal(w, g[f, :n0].cpu().numpy()), (kern, f)
    finally:
        ex.close()


def test_window_top4_against_features_in_area(msorb_mod, oracle, stereo_frame):
    """The raw window search: top-4 of DescriptorDistance over GetFeaturesInArea in scan order."""
    s = stereo_frame
    f, rf = _frames(msorb_mod, oracle, s, None)
    rng = np.random.Generator(np.random.PCG64(31))
    nq = 400
    src = rng.integers(0, len(s["kl"]), nq)
    x = s["kl"]["x"][src] + rng.normal(0, 5, nq).astype(np.float32)
    y = s["kl"]["y"][src] + rng.normal(0, 5, nq).astype(np.float32)
    r = rng.uniform(5, 60, nq).astype(np.float32)
    mn = rng.integers(-1, 4, nq).astype(np.int32)
    mx = np.where(rng.random(nq) < 0.5, -1, mn + rng.integers(0, 3, nq)).astype(np.int32)
    qd = mc.flip_bits(rng, s["dl"][src], 60)
    occ = (rng.random(len(s["kl"])) < 0.3).astype(np.uint8)
    skip = (rng.random(nq) < 0.5).astype(np.uint8)
    bi, bd = msorb_mod.window_top4(f, x, y, r, mn, mx, qd, skip_occupied=skip, occupied=occ)
    for i in range(nq):
        cand = [c for c in rf.GetFeaturesInArea(float(x[i]), float(y[i]), float(r[i]), int(mn[i]), int(mx[i]))
                if not (skip[i] and occ[c])]
        d = [oracle.descriptor_distance(qd[i], s["dl"][c]) for c in cand]
        order = sorted(range(len(cand)), key=lambda k: (d[k], k))[:4]
        want_i = [cand[k] for k in order] + [-1] * (4 - len(order))
        want_d = [d[k] for k in order] + [256] * (4 - len(order))
        assert bi[i].tolist() == want_i and bd[i].tolist() == want_d


def test_stereo_matches_batch_equals_per_frame_and_oracle(msorb_mod, oracle):
    """msorb_stereo_matches_batch (device-resident, median rejection on the device) on a batch of stereo pairs: every
    pair equals the per-frame msorb_stereo_matches on the same images, and pair 0 equals the oracle directly."""
    import torch
    cfg = synth.KITTI
    n_pairs = 5
    host = synth.stereo_batch(n_pairs, cfg["rows"], cfg["cols"], seed0=30)
    host[6] = 0                                            # pair 3: empty left image -> no left keypoints
    host[9] = 0                                            # pair 4: empty right image -> no candidates
    mbf, mb = mc.KITTI_BF, mc.KITTI_BF / mc.KITTI_FX
    ex = msorb_mod.ORBextractor(2000, 1.2, 8, 20, 7)
    exl = msorb_mod.ORBextractor(2000, 1.2, 8, 20, 7)
    exr = msorb_mod.ORBextractor(2000, 1.2, 8, 20, 7)
    try:
        d_img = torch.from_numpy(host).cuda()
        counts, _, d_kps, d_desc = ex.extract_batch(d_img, (0, 0))
        d_ur, d_dp, oob, ms = msorb_mod.stereo_matches_batch(ex, counts, d_kps, d_desc, mb, mbf)
        assert ms > 0
        kps = d_kps.cpu().numpy().view(oracle.KP_DTYPE)[..., 0]
        desc = d_desc.cpu().numpy()
        ur, dp = d_ur.cpu().numpy(), d_dp.cpu().numpy()
        assert counts[6] == 0 and counts[9] == 0
        for p in range(n_pairs):
            nl, nr = int(counts[2 * p]), int(counts[2 * p + 1])
            _, kl, dl = exl(host[2 * p])
            _, kr, dr = exr(host[2 * p + 1])
            assert np.array_equal(kl.view(np.uint8), kps[2 * p, :nl].view(np.uint8)) and len(kr) == nr
            if nl == 0:
                continue
            wur, wdp, woob = msorb_mod.stereo_matches(exl, exr, kl, dl, kr, dr, mb, mbf)
            assert np.array_equal(ur[p, :nl].view(np.uint32), wur.view(np.uint32)), p
            assert np.array_equal(dp[p, :nl].view(np.uint32), wdp.view(np.uint32)), p
            assert oob[p] == woob
            assert np.all(ur[p, nl:] == -1)
            if p == 0:
                pl = [exl.pyramid_level(l) for l in range(8)]
                pr = [exr.pyramid_level(l) for l in range(8)]
                rur, rdp, _ = oracle.compute_stereo_matches(kl, dl, kr, dr, pl, pr, exl.GetScaleFactors(),
                                                            exl.GetInverseScaleFactors(), mb, mbf)
                assert (rur > 0).sum() > 500
                assert np.array_equal(ur[0, :nl].view(np.uint32), rur.view(np.uint32))
                assert np.array_equal(dp[0, :nl].view(np.uint32), rdp.view(np.uint32))
        assert np.all(ur[4, :counts[8]] == -1)
    finally:
        ex.close(); exl.close(); exr.close()


@pytest.mark.parametrize("seed,th", [(0, 3.0), (1, 4.0), (2, 2.5)])
def test_fuse_search_matches_oracle(msorb_mod, oracle, stereo_frame, seed, th):
    """msorb_fuse_search (window search + reprojection-error gate of ORBmatcher::Fuse, ORBmatcher.cc:1499-1561) vs oracle."""
    s = stereo_frame
    rng = np.random.Generator(np.random.PCG64(seed))
    N = len(s["kl"])
    ur_kf = np.where(rng.random(N) < 0.6, s["kl"]["x"] - rng.uniform(1, 40, N), -1).astype(np.float32)
    if seed == 2:
        ur_kf[rng.random(N) < 0.1] = 0.0                   # mvuRight == 0 is stereo for Fuse (>= 0) but not for the trackers (> 0)
    F, R = _frames(msorb_mod, oracle, s, ur_kf)
    scale = np.asarray(s["scale"], np.float32)
    inv_sigma2 = (np.float32(1.0) / (scale * scale)).astype(np.float32)
    M = 3000
    src = rng.integers(0, N, M)
    sig = rng.choice([0.3, 1.0, 3.0], M)
    u = (s["kl"]["x"][src] + rng.normal(0, sig)).astype(np.float32)
    v = (s["kl"]["y"][src] + rng.normal(0, sig)).astype(np.float32)
    ur = np.where(ur_kf[src] >= 0, ur_kf[src] + rng.normal(0, sig), u - 20).astype(np.float32)
    level = np.clip(s["kl"]["octave"][src] + rng.integers(0, 2, M), 0, 7).astype(np.int32)
    radius = (np.float32(th) * scale[level]).astype(np.float32)
    valid = (rng.random(M) < 0.9).astype(np.uint8)
    u[rng.random(M) < 0.02] = -50.0                        # window entirely outside the grid
    desc = mc.flip_bits(rng, s["dl"][src], 40)
    try:
        bi, bd = F.FuseSearch(inv_sigma2, valid, u, v, ur, level, radius, desc)
        wi, wd = R.FuseSearch(inv_sigma2, valid, u, v, ur, level, radius, desc)
        assert np.array_equal(bi, wi) and np.array_equal(bd, wd)
        hit = wi >= 0
        assert 300 < hit.sum() < valid.sum()               # the gate and the level band reject a good share
        assert np.all(bi[valid == 0] == -1)
        # the gate matters: with mvInvLevelSigma2 = 0 it never rejects and more points find a keypoint
        gi, _ = F.FuseSearch(np.zeros(8, np.float32), valid, u, v, ur, level, radius, desc)
        assert (gi >= 0).sum() > hit.sum() + 50
    finally:
        F.close()


@pytest.mark.parametrize("seed,n,th,orb_dist", [(1, 2500, 10.0, 100), (2, 1500, 3.0, 64), (3, 4000, 15.0, 100)])
def test_search_by_projection_keyframe_relocalisation(msorb_mod, oracle, stereo_frame, seed, n, th, orb_dist):
    """msorb_search_by_projection_kf (Tracking::Relocalization's SearchByProjection(Frame, pKF, sFound, th, ORBdist),
    ORBmatcher.cc:2154-2275): any held keypoint is taken, no mvuRight test, bestDist <= ORBdist."""
    s = stereo_frame
    rng = np.random.Generator(np.random.PCG64(200 + seed))
    N = len(s["kl"])
    ur = np.where(rng.random(N) < 0.6, s["kl"]["x"] - rng.uniform(1, 40, N), -1).astype(np.float32)
    f, rf = _frames(msorb_mod, oracle, s, ur)
    t = mc.last_frame_table(rng, s["kl"], s["dl"], ur, s["scale"], n)
    pts = dict(valid=t["valid"], u=t["u"], v=t["v"], level=t["octave"], angle=t["angle"], desc=t["desc"], mp=t["mp"])
    init = np.where(rng.random(N) < 0.15, 100000 + rng.integers(0, 50, N), -1).astype(np.int32)   # found by the PnP stage before
    try:
        for check in (True, False):
            got, want = init.copy(), init.copy()
            nm = f.SearchByProjection_kf(pts, got, th, orb_dist, check)
            rn = rf.SearchByProjection_kf(pts, want, th, orb_dist, check)
            assert nm == rn and np.array_equal(got, want)
        assert rn > 50
        # differs from the last-frame form where that form would test mvuRight / look at Observations()
        last = dict(t)
        alt = init.copy()
        alt[alt >= 0] = -1
        rf.SearchByProjection_frames(last, alt, th, False, False, True)
        assert not np.array_equal(alt[init < 0], want[init < 0])
    finally:
        f.close()


@pytest.mark.parametrize("seed,n,th,ratio", [(1, 3000, 8.0, 1.5), (2, 1200, 4.0, 1.0), (3, 5000, 10.0, 0.73)])
def test_search_by_projection_sim3_forms(msorb_mod, oracle, stereo_frame, seed, n, th, ratio):
    """msorb_search_by_projection_sim3 (loop-closing window searches, ORBmatcher.cc:423-753): claims through vpMatched,
    level band predicted-1 .. predicted, (float)bestDist <= TH_LOW * ratioHamming."""
    s = stereo_frame
    rng = np.random.Generator(np.random.PCG64(300 + seed))
    N = len(s["kl"])
    f, rf = _frames(msorb_mod, oracle, s, None)
    t = mc.last_frame_table(rng, s["kl"], s["dl"], np.full(N, -1, np.float32), s["scale"], n)
    level = np.clip(t["octave"] + rng.integers(0, 2, n), 0, 7).astype(np.int32)
    pts = dict(valid=t["valid"], u=t["u"], v=t["v"], level=level, desc=t["desc"], mp=t["mp"])
    init = np.where(rng.random(N) < 0.2, 100000 + rng.integers(0, 50, N), -1).astype(np.int32)
    max_dist = float(np.float32(50) * np.float32(ratio))
    try:
        got, want = init.copy(), init.copy()
        nm = f.SearchByProjection_sim3(pts, got, th, max_dist)
        rn = rf.SearchByProjection_sim3(pts, want, th, max_dist)
        assert nm == rn and np.array_equal(got, want)
        assert rn > 50 and np.array_equal(got[init >= 0], init[init >= 0])
        # Fuse(pKF, Scw, ...) / SearchBySim3 window search = msorb_fuse_search with a zero gate table: same best as a
        # claim-free sim3 search restricted to one query
        bi, bd = f.FuseSearch(np.zeros(8, np.float32), t["valid"], t["u"], t["v"], t["u"], level,
                              (np.float32(th) * np.asarray(s["scale"], np.float32)[level]).astype(np.float32), t["desc"])
        for i in np.nonzero(t["valid"])[0][:60]:
            one = {k: a[i:i + 1] for k, a in pts.items()}
            m = np.full(N, -1, np.int32)
            rf.SearchByProjection_sim3(one, m, th, 255.0)     # (256 would accept the 'no candidate' state, like the reference)
            hit = np.nonzero(m >= 0)[0]
            assert (bi[i] == hit[0]) if len(hit) else bi[i] == -1
    finally:
        f.close()


def test_extract_stereo_equals_two_extractions_plus_stereo_matches(msorb_mod, oracle):
    """msorb_extract_stereo (both eyes through the batch pipeline + device-resident ComputeStereoMatches, one call) gives
    exactly what two msorb_extract calls followed by msorb_stereo_matches give (which is pinned to the oracle above)."""
    mbf, mb = mc.KITTI_BF, mc.KITTI_BF / mc.KITTI_FX
    # (nfeatures 501: an odd capacity — the output block's tail is not 16-byte aligned, so the median rule keeps its own launch
    # instead of riding the read-back: extractor.hip extract_stereo_sink, stereo_median_readback_kernel)
    for seed, (rows, cols), nfeat in ((40, (376, 1241), 2000), (41, (480, 752), 1000), (42, (240, 320), 500), (43, (240, 320), 501),
                                       (44, (376, 1241), 10000)):    # (a quota beyond a workgroup's LDS: the selection over global memory)
        L, R = synth.stereo_pair(seed, rows, cols)
        ex = msorb_mod.ORBextractor(nfeat, 1.2, 8, 20, 7)
        exl = msorb_mod.ORBextractor(nfeat, 1.2, 8, 20, 7)
        exr = msorb_mod.ORBextractor(nfeat, 1.2, 8, 20, 7)
        try:
            for rep in range(2):                 # second call: warmed handle, same answer
                kl, dl, kr, dr, ur, dp, oob = ex.extract_stereo(L, R, mb, mbf)
                _, wkl, wdl = exl(L)
                _, wkr, wdr = exr(R)
                assert np.array_equal(kl.view(np.uint8), wkl.view(np.uint8)) and np.array_equal(dl, wdl)
                assert np.array_equal(kr.view(np.uint8), wkr.view(np.uint8)) and np.array_equal(dr, wdr)
                wur, wdp, woob = msorb_mod.stereo_matches(exl, exr, wkl, wdl, wkr, wdr, mb, mbf)
                assert np.array_equal(ur.view(np.uint32), wur.view(np.uint32))
                assert np.array_equal(dp.view(np.uint32), wdp.view(np.uint32))
                assert oob == woob and (ur > 0).sum() > 50
            # a single-image call on the same handle afterwards still works (buffers are shared)
            _, k1, d1 = ex(L)
            assert np.array_equal(k1.view(np.uint8), wkl.view(np.uint8)) and np.array_equal(d1, wdl)
        finally:
            ex.close(); exl.close(); exr.close()


def test_extract_stereo_on_a_tall_image_takes_the_row_table(msorb_mod, oracle):
    """rows > 4095: the band records of a stereo frame hold 12-bit rows, the fused call builds the row table instead
    (extractor.hip extract_stereo_sink) — same contract as above."""
    mbf, mb = mc.KITTI_BF, mc.KITTI_BF / mc.KITTI_FX
    rows, cols, nfeat = 4200, 2200, 3000
    L, R = synth.stereo_pair(77, rows, cols)
    ex = msorb_mod.ORBextractor(nfeat, 1.2, 8, 20, 7)
    exl = msorb_mod.ORBextractor(nfeat, 1.2, 8, 20, 7)
    exr = msorb_mod.ORBextractor(nfeat, 1.2, 8, 20, 7)
    try:
        kl, dl, kr, dr, ur, dp, oob = ex.extract_stereo(L, R, mb, mbf)
        _, wkl, wdl = exl(L)
        _, wkr, wdr = exr(R)
        assert np.array_equal(kl.view(np.uint8), wkl.view(np.uint8)) and np.array_equal(dl, wdl)
        assert np.array_equal(kr.view(np.uint8), wkr.view(np.uint8)) and np.array_equal(dr, wdr)
        wur, wdp, woob = msorb_mod.stereo_matches(exl, exr, wkl, wdl, wkr, wdr, mb, mbf)
        assert np.array_equal(ur.view(np.uint32), wur.view(np.uint32))
        assert np.array_equal(dp.view(np.uint32), wdp.view(np.uint32))
        assert oob == woob and (ur > 0).sum() > 20
    finally:
        ex.close(); exl.close(); exr.close()
