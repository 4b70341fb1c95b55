"""GPU parity tests of the 2-GPU stereo split (BASELINE configs[3]) behind the C ABI: msorb_extract_stereo_split — the
left extractor object on device A, the right one on device B, gather of the right eye's features + pyramid onto A,
Frame::ComputeStereoMatches on A (Frame.cc:119-137, 743-913) — and the batch form used by `bench.py --gpus N`
(msorb_pyramid_batch + msorb_stereo_matches_split on gathered keypoints / descriptors).

On a 1-GPU box both handles live on device 0 (the identical code path: peer copies degenerate to device-to-device copies);
when msorb_device_count() >= 2 the same tests put the right eye on device 1 and the copies cross xGMI."""
import numpy as np
import pytest

from msorb import synth
import matcher_cases as mc

pytestmark = pytest.mark.gpu


def _devices(msorb_mod):
    n = msorb_mod.lib().msorb_device_count()
    return (0, 1) if n >= 2 else (0, 0)


@pytest.mark.parametrize("seed,shape,nfeat", [(40, (376, 1241), 2000), (41, (480, 752), 1000), (42, (240, 320), 500)])
def test_extract_stereo_split_bit_exact_vs_oracle(msorb_mod, oracle, seed, shape, nfeat):
    rows, cols = shape
    mbf, mb = mc.KITTI_BF, mc.KITTI_BF / mc.KITTI_FX
    dev_a, dev_b = _devices(msorb_mod)
    L, R = synth.stereo_pair(seed, rows, cols)
    exl = msorb_mod.ORBextractor(nfeat, 1.2, 8, 20, 7, device=dev_a)
    exr = msorb_mod.ORBextractor(nfeat, 1.2, 8, 20, 7, device=dev_b)
    orl = oracle.OracleExtractor(nfeat, 1.2, 8, 20, 7)
    orr = oracle.OracleExtractor(nfeat, 1.2, 8, 20, 7)
    try:
        _, okl, odl = orl(L)
        _, okr, odr = orr(R)
        pl = [orl.level(l) for l in range(8)]
        pr = [orr.level(l) for l in range(8)]
        tb = orl.tables()
        rur, rdp, roob = oracle.compute_stereo_matches(okl, odl, okr, odr, pl, pr, tb["scale"], tb["inv_scale"], mb, mbf)
        for rep in range(3):                 # cold handle, warm handle, and once more: same answer every time
            kl, dl, kr, dr, ur, dp, oob = exl.extract_stereo_split(exr, L, R, mb, mbf)
            assert np.array_equal(kl.view(np.uint8), okl.view(np.uint8)) and np.array_equal(dl, odl)
            assert np.array_equal(kr.view(np.uint8), okr.view(np.uint8)) and np.array_equal(dr, odr)
            assert np.array_equal(ur.view(np.uint32), rur.view(np.uint32))    # mvuRight bit patterns
            assert np.array_equal(dp.view(np.uint32), rdp.view(np.uint32))    # mvDepth bit patterns
            assert oob == roob and (ur > 0).sum() > 50
        # both handles stay usable on their own afterwards, and the fused single-handle call agrees
        _, k1, d1 = exr(R)
        assert np.array_equal(k1.view(np.uint8), okr.view(np.uint8)) and np.array_equal(d1, odr)
        f = exl.extract_stereo(L, R, mb, mbf)
        assert np.array_equal(f[4].view(np.uint32), rur.view(np.uint32))
        # swapped roles (right object drives): right image as "left" -> no positive disparities survive, still well-formed
        kl2, _, kr2, _, ur2, _, _ = exr.extract_stereo_split(exl, R, L, mb, mbf)
        assert np.array_equal(kl2.view(np.uint8), okr.view(np.uint8)) and np.array_equal(kr2.view(np.uint8), okl.view(np.uint8))
        assert len(ur2) == len(okr)
    finally:
        exl.close(); exr.close()


def test_extract_stereo_split_rejects_bad_arguments(msorb_mod):
    ex = msorb_mod.ORBextractor(500, 1.2, 8, 20, 7)
    ex2 = msorb_mod.ORBextractor(600, 1.2, 8, 20, 7)
    L, R = synth.stereo_pair(3, 240, 320)
    try:
        with pytest.raises(msorb_mod.MsorbError) as e:
            ex.extract_stereo_split(ex, L, R, 0.5, 380.0)     # one object cannot be both eyes
        assert e.value.code == msorb_mod.E_INVALID
        with pytest.raises(msorb_mod.MsorbError) as e:
            ex.extract_stereo_split(ex2, L, R, 0.5, 380.0)    # different nfeatures
        assert e.value.code == msorb_mod.E_INVALID
    finally:
        ex.close(); ex2.close()


def test_split_batch_equals_interleaved_batch(msorb_mod, oracle):
    """The `bench.py --gpus N` join: left eyes extracted by one handle, right eyes' keypoints / descriptors / counts arriving
    in separate arrays (gathered), the right pyramid rebuilt locally by msorb_pyramid_batch; msorb_stereo_matches_split must
    equal msorb_stereo_matches_batch on the interleaved batch (itself pinned to the oracle in test_matcher_gpu.py)."""
    import torch
    cfg = synth.KITTI
    n_pairs = 6
    host = synth.stereo_batch(n_pairs, cfg["rows"], cfg["cols"], seed0=70)
    host[6] = 0                                            # pair 3: empty left image
    host[9] = 0                                            # pair 4: empty right image
    mbf, mb = mc.KITTI_BF, mc.KITTI_BF / mc.KITTI_FX
    ex = msorb_mod.ORBextractor(2000, 1.2, 8, 20, 7)
    exl = msorb_mod.ORBextractor(2000, 1.2, 8, 20, 7)
    exr = msorb_mod.ORBextractor(2000, 1.2, 8, 20, 7)      # "the other device's" extractor
    exp = msorb_mod.ORBextractor(2000, 1.2, 8, 20, 7)      # pyramid-only handle beside exl
    try:
        d_img = torch.from_numpy(host).cuda()
        counts, _, d_kps, d_desc = ex.extract_batch(d_img, (0, 0))
        want_ur, want_dp, want_oob, _ = msorb_mod.stereo_matches_batch(ex, counts, d_kps, d_desc, mb, mbf)
        d_left, d_right = d_img[0::2].contiguous(), d_img[1::2].contiguous()
        cl, _, kl, dl = exl.extract_batch(d_left, (0, 0))
        cr, _, kr, dr = exr.extract_batch(d_right, (0, 0))
        assert np.array_equal(cl, counts[0::2]) and np.array_equal(cr, counts[1::2])
        kr2, dr2 = kr.clone(), dr.clone()                  # stand-ins for the received blocks
        exp.pyramid_batch(d_right)
        got_ur, got_dp, got_oob, ms = msorb_mod.stereo_matches_split(exl, exp, cl, kl, dl, cr, kr2, dr2, mb, mbf)
        assert ms > 0 and np.array_equal(got_oob, want_oob)
        w, g = want_ur.cpu().numpy(), got_ur.cpu().numpy()
        wd, gd = want_dp.cpu().numpy(), got_dp.cpu().numpy()
        for p in range(n_pairs):
            nl = int(cl[p])
            assert np.array_equal(w[p, :nl].view(np.uint32), g[p, :nl].view(np.uint32)), p
            assert np.array_equal(wd[p, :nl].view(np.uint32), gd[p, :nl].view(np.uint32)), p
        assert (g[0] > 0).sum() > 500 and cl[3] == 0 and np.all(g[4, :cl[4]] == -1)
        # the rebuilt pyramid equals the extractor's own
        for l in (1, 4, 7):
            assert np.array_equal(exp.debug_level(2, l), exr.debug_level(2, l))
    finally:
        ex.close(); exl.close(); exr.close(); exp.close()


def test_extract_stereo_split_without_peer_access(msorb_mod, oracle, monkeypatch):
    """The gather of msorb_extract_stereo_split when the two devices cannot reach each other over xGMI: staged explicitly through
    pinned host memory (device B -> pinned block -> device A, ordered by one event).  MSORB_SPLIT_NO_PEER=1 forces that path
    on any pair of handles — here both on one device, on a multi-GPU box across two — and the results must not change."""
    rows, cols = 376, 1241
    mbf, mb = mc.KITTI_BF, mc.KITTI_BF / mc.KITTI_FX
    dev_a, dev_b = _devices(msorb_mod)
    L, R = synth.stereo_pair(47, rows, cols)
    exl = msorb_mod.ORBextractor(2000, 1.2, 8, 20, 7, device=dev_a)
    exr = msorb_mod.ORBextractor(2000, 1.2, 8, 20, 7, device=dev_b)
    # the switch is read when a handle is created: a second pair of handles made under it takes the staged path
    monkeypatch.setenv("MSORB_SPLIT_NO_PEER", "1")
    sxl = msorb_mod.ORBextractor(2000, 1.2, 8, 20, 7, device=dev_a)
    sxr = msorb_mod.ORBextractor(2000, 1.2, 8, 20, 7, device=dev_b)
    monkeypatch.delenv("MSORB_SPLIT_NO_PEER")
    try:
        want = exl.extract_stereo_split(exr, L, R, mb, mbf)          # peer / device-local copies
        for _ in range(2):
            got = sxl.extract_stereo_split(sxr, L, R, mb, mbf)
            for a, b in zip(got[:6], want[:6]):
                assert np.array_equal(a.view(np.uint8) if a.dtype.fields else a.view(np.uint8), b.view(np.uint8))
            assert got[6] == want[6] and (got[4] > 0).sum() > 500
        again = exl.extract_stereo_split(exr, L, R, mb, mbf)
        assert np.array_equal(again[4].view(np.uint32), want[4].view(np.uint32))
    finally:
        exl.close(); exr.close(); sxl.close(); sxr.close()


@pytest.mark.parametrize("no_peer,force_peer", [(False, False), (True, False), (False, True), (True, True)])
def test_split_entries_under_both_test_hooks(msorb_mod, oracle, monkeypatch, no_peer, force_peer):
    """Everything of configs[3]'s product path one GPU can execute, through the C ABI, with the two test hooks both ways:
    msorb_extract_stereo_split (gather by peer copy / staged through pinned host memory: MSORB_SPLIT_NO_PEER), then the frame
    form msorb_stereo_matches on the same two handles (the right pyramid pulled over the peer-copy path: MSORB_FORCE_PEER_PYRAMID),
    then the batch form msorb_pyramid_batch + msorb_stereo_matches_split on 'gathered' blocks.  All against the oracle."""
    import torch
    cfg = synth.KITTI
    rows, cols = cfg["rows"], cfg["cols"]
    mbf, mb = mc.KITTI_BF, mc.KITTI_BF / mc.KITTI_FX
    dev_a, dev_b = _devices(msorb_mod)
    if no_peer:
        monkeypatch.setenv("MSORB_SPLIT_NO_PEER", "1")
    if force_peer:
        monkeypatch.setenv("MSORB_FORCE_PEER_PYRAMID", "1")
    exl = msorb_mod.ORBextractor(2000, 1.2, 8, 20, 7, device=dev_a)     # the switches are read at creation
    exr = msorb_mod.ORBextractor(2000, 1.2, 8, 20, 7, device=dev_b)
    exp = msorb_mod.ORBextractor(2000, 1.2, 8, 20, 7, device=dev_a)
    monkeypatch.delenv("MSORB_SPLIT_NO_PEER", raising=False)
    monkeypatch.delenv("MSORB_FORCE_PEER_PYRAMID", raising=False)
    L, R = synth.stereo_pair(52 + 2 * int(no_peer) + int(force_peer), rows, cols)
    orl, orr = oracle.OracleExtractor(2000, 1.2, 8, 20, 7), oracle.OracleExtractor(2000, 1.2, 8, 20, 7)
    try:
        _, okl, odl = orl(L)
        _, okr, odr = orr(R)
        tb = orl.tables()
        rur, rdp, roob = oracle.compute_stereo_matches(okl, odl, okr, odr, [orl.level(l) for l in range(8)],
                                                       [orr.level(l) for l in range(8)], tb["scale"], tb["inv_scale"], mb, mbf)
        kl, dl, kr, dr, ur, dp, oob = exl.extract_stereo_split(exr, L, R, mb, mbf)
        assert np.array_equal(kl.view(np.uint8), okl.view(np.uint8)) and np.array_equal(kr.view(np.uint8), okr.view(np.uint8))
        assert np.array_equal(dl, odl) and np.array_equal(dr, odr)
        assert np.array_equal(ur.view(np.uint32), rur.view(np.uint32)) and np.array_equal(dp.view(np.uint32), rdp.view(np.uint32)) and oob == roob
        # frame form on the two handles (their last calls hold L and R)
        exl(L)
        exr(R)
        ur2, dp2, oob2 = msorb_mod.stereo_matches(exl, exr, okl, odl, okr, odr, mb, mbf)
        assert np.array_equal(ur2.view(np.uint32), rur.view(np.uint32)) and np.array_equal(dp2.view(np.uint32), rdp.view(np.uint32)) and oob2 == roob
        # batch form: left eye extracted by exl, right eye's features arriving as separate blocks, its pyramid rebuilt by exp
        if dev_a == dev_b:
            d_l = torch.from_numpy(np.stack([L, L])).cuda(dev_a)
            d_r = torch.from_numpy(np.stack([R, R])).cuda(dev_a)
            cl, _, bkl, bdl = exl.extract_batch(d_l, (0, 0))
            cr, _, bkr, bdr = exr.extract_batch(d_r, (0, 0))
            exp.pyramid_batch(d_r)
            gur, gdp, goob, _ = msorb_mod.stereo_matches_split(exl, exp, cl, bkl, bdl, cr, bkr.clone(), bdr.clone(), mb, mbf)
            for p in range(2):
                n = int(cl[p])
                assert np.array_equal(gur[p, :n].cpu().numpy().view(np.uint32), rur.view(np.uint32)), p
                assert np.array_equal(gdp[p, :n].cpu().numpy().view(np.uint32), rdp.view(np.uint32)), p
            assert int(goob[0]) == roob
    finally:
        exl.close(); exr.close(); exp.close()
