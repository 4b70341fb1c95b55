"""The brute-force step of the two-camera (fisheye) stereo constructor: msorb_knn_match2 = cv::BFMatcher(NORM_HAMMING).knnMatch
(k = 2) (Frame.cc:1076) against the oracle's restatement of ORBmatcher::DescriptorDistance over all pairs, and
msorb_host::ComputeStereoFishEyeMatches (the reference's loop around it, Frame.cc:1057-1101) over a stand-in Frame."""
import os
import struct
import subprocess

import numpy as np
import pytest

import matcher_cases as mc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _descs(seed, nq, nt):
    rng = np.random.Generator(np.random.PCG64(seed))
    t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    q = np.where((rng.random(nq) < 0.7)[:, None], mc.flip_bits(rng, t[rng.integers(0, max(nt, 1), nq)] if nt else np.zeros((nq, 32), np.uint8), 60),
                 rng.integers(0, 256, (nq, 32), dtype=np.uint8)).astype(np.uint8)
    if nt >= 8 and nq >= 4:           # exact ties: duplicated train rows, a query equal to a train row
        t[5] = t[2]
        t[nt - 1] = t[2]
        q[0] = t[2]
        q[1] = t[5]
    return q, t


def _naive(oracle, q, t):
    d = np.array([[oracle.descriptor_distance(a, b) for b in t] for a in q], np.int32).reshape(len(q), len(t))
    bi = np.full(len(q), -1, np.int32); bd = np.full(len(q), 256, np.int32)
    si = np.full(len(q), -1, np.int32); sd = np.full(len(q), 256, np.int32)
    for i in range(len(q)):
        order = np.argsort(d[i], kind="stable")
        if len(order) >= 1:
            bi[i], bd[i] = order[0], d[i, order[0]]
        if len(order) >= 2:
            si[i], sd[i] = order[1], d[i, order[1]]
    return bi, bd, si, sd


@pytest.mark.parametrize("seed,nq,nt", [(1, 300, 400), (2, 50, 2048), (3, 120, 2500), (4, 64, 5000), (5, 10, 1), (6, 10, 0), (7, 0, 10)])
def test_knn_match2_vs_all_pairs(msorb_mod, oracle, seed, nq, nt):
    q, t = _descs(seed, nq, nt)
    bi, bd, si, sd = msorb_mod.knn_match2(q, t)
    rbi, rbd, rsi, rsd = _naive(oracle, q, t)
    assert np.array_equal(bi, rbi) and np.array_equal(bd, rbd) and np.array_equal(sd, rsd) and np.array_equal(si, rsi)


def test_cpp_compute_stereo_fisheye_matches(tmp_path, oracle):
    exe = tmp_path / "dropin_fisheye"
    subprocess.check_call(["g++", "-std=c++17", "-O2", f"-I{ROOT}/tests/cv_stub", f"-I{ROOT}/ms-slam_amd/host", f"-I{ROOT}/include",
                           f"{ROOT}/tests/dropin_fisheye_main.cc", f"-L{ROOT}/ms-slam_amd", "-lmsorb", f"-Wl,-rpath,{ROOT}/ms-slam_amd",
                           "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    rng = np.random.Generator(np.random.PCG64(9))
    NL, NR, mono_l, mono_r = 900, 850, 300, 260
    KP = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
    kl, kr = np.zeros(NL, KP), np.zeros(NR, KP)
    for k in (kl, kr):
        k["x"], k["y"], k["octave"] = rng.uniform(0, 700, len(k)), rng.uniform(0, 400, len(k)), rng.integers(0, 8, len(k))
    dr = rng.integers(0, 256, (NR, 32), dtype=np.uint8)
    dl = rng.integers(0, 256, (NL, 32), dtype=np.uint8)
    src = rng.integers(mono_r, NR, NL - mono_l)
    dl[mono_l:] = np.where((rng.random(NL - mono_l) < 0.75)[:, None], mc.flip_bits(rng, dr[src], 30), dl[mono_l:])
    blob, out = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(blob, "wb") as f:
        f.write(struct.pack("<4i", NL, NR, mono_l, mono_r))
        for a in (kl, kr, dl, dr):
            f.write(np.ascontiguousarray(a).tobytes())
    subprocess.check_call([str(exe), str(blob), str(out)])
    raw = out.read_bytes()
    n, close = struct.unpack_from("<2i", raw, 0)
    l2r = np.frombuffer(raw, np.int32, NL, 8)
    r2l = np.frombuffer(raw, np.int32, NR, 8 + 4 * NL)
    depth = np.frombuffer(raw, np.float32, NL, 8 + 4 * NL + 4 * NR)
    ur = np.frombuffer(raw, np.float32, NL, 8 + 8 * NL + 4 * NR)
    # the reference's loop (Frame.cc:1079-1099) on the all-pairs distances
    bi, bd, si, sd = _naive(oracle, dl[mono_l:], dr[mono_r:])
    want_l2r, want_r2l, want_depth, nm = np.full(NL, -1, np.int32), np.full(NR, -1, np.int32), np.full(NL, -1.0, np.float32), 0
    for qi in range(NL - mono_l):
        if not (np.float32(bd[qi]) < np.float32(sd[qi]) * 0.7):
            continue
        il, ir = qi + mono_l, int(bi[qi]) + mono_r
        d = np.float32(kl["x"][il]) - np.float32(kr["x"][ir])
        dep = np.float32(100.0) / d if d > 2.0 else np.float32(-1.0)
        if dep > 0.0001:
            want_l2r[il], want_r2l[ir], want_depth[il] = ir, il, dep
            nm += 1
    assert n == nm > 100 and close == 0
    assert np.array_equal(l2r, want_l2r) and np.array_equal(r2l, want_r2l)
    assert np.array_equal(depth.view(np.uint32), want_depth.view(np.uint32)) and np.all(ur == -1)
