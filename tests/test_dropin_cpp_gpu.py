"""Compiles the drop-in C++ class ORB_SLAM3::ORBextractor (ms-slam_amd/host, same declaration as the
reference's include/ORBextractor.h) against the test-only cv stub, links libmsorb.so, runs it like
Frame::ExtractORB (Frame.cc:418-425) and compares with the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

from msorb import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_cpp_dropin_matches_oracle(tmp_path, oracle, msorb_mod):
    exe = tmp_path / "dropin_extractor"
    subprocess.check_call(["g++", "-std=c++17", "-O2", f"-I{ROOT}/tests/cv_stub", f"-I{ROOT}/ms-slam_amd/host",
                           f"-I{ROOT}/include", f"{ROOT}/tests/dropin_extractor_main.cc",
                           f"{ROOT}/ms-slam_amd/host/ORBextractor.cc", f"-L{ROOT}/ms-slam_amd", "-lmsorb",
                           f"-Wl,-rpath,{ROOT}/ms-slam_amd", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    cfg = synth.EUROC
    img = synth.image(77, cfg["rows"], cfg["cols"])
    raw, out = tmp_path / "in.raw", tmp_path / "out.bin"
    img.tofile(raw)
    subprocess.check_call([str(exe), str(cfg["rows"]), str(cfg["cols"]), str(raw), str(out), "1000"])
    blob = out.read_bytes()
    mono, n = struct.unpack_from("<ii", blob, 0)
    kps = np.frombuffer(blob, oracle.KP_DTYPE, n, 8)
    desc = np.frombuffer(blob, np.uint8, n * 32, 8 + 28 * n).reshape(n, 32)
    l7r, l7c, lv = struct.unpack_from("<iii", blob, 8 + 60 * n)
    rmono, rkps, rdesc = oracle.OracleExtractor(1000, 1.2, 8, 20, 7)(img)
    assert (mono, n) == (rmono, len(rkps)) and (l7r, l7c, lv) == (134, 210, 8)
    assert np.array_equal(kps.view(np.uint8), rkps.view(np.uint8)) and np.array_equal(desc, rdesc)
