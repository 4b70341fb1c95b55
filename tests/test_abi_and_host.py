"""CPU-only tests: the C-ABI library loads and exports every symbol include/msorb.h declares, the host
logic (parameter tables, geometry, quadtree) agrees with the oracle, and there is no CPU fallback."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from msorb import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(msorb_mod):
    hdr = open(os.path.join(ROOT, "include", "msorb.h")).read()
    declared = sorted(set(re.findall(r"\b(msorb_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 15
    lib = C.CDLL(msorb_mod.LIB_PATH)
    for sym in declared:
        assert hasattr(lib, sym), f"libmsorb.so does not export {sym}"
    assert set(declared) == set(msorb_mod.EXPORTS), set(declared) ^ set(msorb_mod.EXPORTS)


def test_no_cpu_fallback_without_gpu(msorb_mod):
    if msorb_mod.lib().msorb_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(msorb_mod.MsorbError) as e:
        msorb_mod.ORBextractor(1000, 1.2, 8, 20, 7)
    assert e.value.code == msorb_mod.E_NO_DEVICE


def test_invalid_parameters_rejected(msorb_mod):
    for args in ((0, 1.2, 8, 20, 7), (1000, 1.0, 8, 20, 7), (1000, 1.2, 0, 20, 7), (1000, 1.2, 17, 20, 7),
                 (1000, 1.2, 8, 7, 20)):
        with pytest.raises(msorb_mod.MsorbError) as e:
            msorb_mod.ORBextractor(*args)
        assert e.value.code == msorb_mod.E_INVALID


@pytest.mark.parametrize("seed,rows,cols,nfeat", [(0, 376, 1241, 2000), (1, 480, 752, 1000), (2, 400, 800, 2000),
                                                  (3, 240, 320, 500), (4, 320, 240, 500), (5, 376, 1241, 300)])
def test_host_quadtree_matches_oracle(msorb_mod, oracle, seed, rows, cols, nfeat):
    """Product host quadtree (orb_host.cc, index pool) vs oracle (std::list restatement of
    ORBextractor.cc:555-779) on real FAST candidates: same keypoints in the same order."""
    img = synth.image(seed, rows, cols)
    ex = oracle.OracleExtractor(nfeat, 1.2, 8, 20, 7)
    ex(img)
    quota = ex.tables()["per_level"]
    for l in range(8):
        c, sel = ex.candidates(l), ex.selected(l)
        h, w = ex.level(l).shape
        kept = msorb_mod.distribute_quadtree(c[:, 0], c[:, 1], c[:, 2], 16, w - 16, 16, h - 16, int(quota[l]))
        want = np.stack([sel["x"] - 16, sel["y"] - 16, sel["response"]], 1).astype(np.int32)
        assert np.array_equal(c[kept], want), f"level {l}"
        assert len(kept) <= quota[l] + 3


def test_host_quadtree_degenerate_inputs(msorb_mod, oracle):
    rng = np.random.Generator(np.random.PCG64(11))
    cases = []
    # all candidates on one pixel, ties everywhere, a single candidate, dense grid
    cases.append((np.full(50, 100), np.full(50, 40), np.full(50, 30)))
    cases.append((np.array([5]), np.array([7]), np.array([20])))
    gx, gy = np.meshgrid(np.arange(0, 600, 2), np.arange(0, 150, 2))
    cases.append((gx.ravel(), gy.ravel(), np.full(gx.size, 25)))
    cases.append((rng.integers(0, 1200, 5000), rng.integers(0, 340, 5000), rng.integers(7, 120, 5000)))
    for xs, ys, sc in cases:
        for N in (1, 5, 60, 434):
            kept = msorb_mod.distribute_quadtree(xs, ys, sc, 16, 1225, 16, 360, N)
            ref = oracle.distribute_quadtree(xs, ys, sc, 16, 1225, 16, 360, N)
            got = np.stack([xs[kept], ys[kept], sc[kept]], 1).astype(np.float32)
            assert np.array_equal(got, ref)


def test_device_quadtree_algorithm_on_host(tmp_path):
    """quadtree_device.h (the generation-synchronous DistributeOctTree the GPU runs, incl. the libstdc++
    std::sort restatement) executed with one host 'thread' must reproduce orb_host.cc's result on thousands of
    random / clustered / tie-heavy candidate sets."""
    import subprocess
    exe = tmp_path / "qt_host_check"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", os.path.join(ROOT, "tests", "qt_host_check.cc"),
                           os.path.join(ROOT, "ms-slam_amd", "csrc", "orb_host.cc"), "-o", str(exe)])
    out = subprocess.check_output([str(exe), "1500"]).decode()
    assert "bad=0" in out, out


def test_matcher_adapter_header_compiles(tmp_path):
    """ms-slam_amd/host/ORBmatcher_device.h against the stand-in Frame / MapPoint of tests/dropin_matcher_main.cc
    (syntax + template instantiation only: no GPU, no link)."""
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-Wall", f"-I{ROOT}/tests/cv_stub", f"-I{ROOT}/ms-slam_amd/host",
                           f"-I{ROOT}/include", os.path.join(ROOT, "tests", "dropin_matcher_main.cc")])
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-Wall", f"-I{ROOT}/ms-slam_amd/host", f"-I{ROOT}/include",
                           os.path.join(ROOT, "tests", "dropin_sparsify_main.cc")])
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-Wall", f"-I{ROOT}/tests/cv_stub", f"-I{ROOT}/ms-slam_amd/host",
                           f"-I{ROOT}/include", os.path.join(ROOT, "tests", "dropin_bow_main.cc")])
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-Wall", f"-I{ROOT}/tests/cv_stub", f"-I{ROOT}/ms-slam_amd/host",
                           f"-I{ROOT}/include", os.path.join(ROOT, "tests", "dropin_bowmatch_main.cc")])


def test_host_array_entries_validate_before_touching_a_device(msorb_mod):
    """The host-array matcher entries check their arguments on the host first: malformed input is MSORB_E_INVALID with or
    without a GPU, and well-formed input without a GPU is MSORB_E_NO_DEVICE (never a CPU answer)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bow_match_cases as bmc
    gpu = msorb_mod.lib().msorb_device_count() > 0
    p = bmc.make_pair(1, n1=60, n2=70, n_nodes=4)
    bad = dict(p)
    f = [a.copy() for a in bad["fv1"]]
    f[2][0] = 10 ** 6                                   # feature index out of range
    bad["fv1"] = tuple(f)
    with pytest.raises(msorb_mod.MsorbError) as e:
        msorb_mod.search_by_bow([bad])
    assert e.value.code == msorb_mod.E_INVALID
    t = bmc.make_triangulation_pair(1, n1=60, n2=70, n_nodes=4)
    badt = dict(t)
    badt["kp2"] = t["kp2"].copy()
    badt["kp2"]["octave"][0] = -1
    with pytest.raises(msorb_mod.MsorbError) as e:
        msorb_mod.search_for_triangulation([badt])
    assert e.value.code == msorb_mod.E_INVALID
    assert msorb_mod.search_by_bow([])[0] == [] and msorb_mod.search_for_triangulation([])[0] == []
    empty = bmc.make_pair(2, n1=0, n2=0)
    assert msorb_mod.search_by_bow([empty])[0][0][0] == 0      # nothing to match: answered without a device
    if not gpu:
        for call in (lambda: msorb_mod.search_by_bow([p]), lambda: msorb_mod.search_for_triangulation([t])):
            with pytest.raises(msorb_mod.MsorbError) as e:
                call()
            assert e.value.code == msorb_mod.E_NO_DEVICE
