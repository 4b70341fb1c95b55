"""bench_legs.optional_leg: a parity self-check and a library / device failure always propagate, an environment failure becomes
the leg's "error" field and is listed in the line's "degraded_legs"."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_optional_leg_semantics(monkeypatch, capsys):
    import bench_legs as bl
    assert bl.optional_leg("x", lambda a, b=1: a + b, 2, b=3) == 5
    r = bl.optional_leg("boom", lambda: (_ for _ in ()).throw(FileNotFoundError("rocprofv3")))
    assert r == {"error": "FileNotFoundError: rocprofv3", "leg": "boom"}
    assert "optional leg 'boom' failed" in capsys.readouterr().err
    with pytest.raises(bl.SelfCheckError):
        bl.optional_leg("parity", lambda: bl.self_check(False, "gpu != cpu"))
    with pytest.raises(AssertionError):
        bl.optional_leg("assert", lambda: (_ for _ in ()).throw(AssertionError("x")))
    monkeypatch.setenv("MSORB_BENCH_FAIL_LEG", "forced")
    assert bl.optional_leg("forced", lambda: 1)["leg"] == "forced" and bl.optional_leg("other", lambda: 1) == 1
    assert bl.DEGRADED[-2:] == ["boom", "forced"]
    # a failure of libmsorb (an error code of a C-ABI entry) or of the device is a regression, not an environment problem: never degraded
    import msorb
    with pytest.raises(msorb.MsorbError):
        bl.optional_leg("lib", lambda: (_ for _ in ()).throw(msorb.MsorbError(-5, "msorb_extract_batch")))
    with pytest.raises(RuntimeError, match="HIP error"):
        bl.optional_leg("hip", lambda: (_ for _ in ()).throw(RuntimeError("HIP error: an illegal memory access was encountered")))
    monkeypatch.setenv("MSORB_BENCH_FAIL_LEG", "forced:library")
    with pytest.raises(RuntimeError, match="forced library failure"):
        bl.optional_leg("forced", lambda: 1)
    assert "lib" not in bl.DEGRADED and "hip" not in bl.DEGRADED


def test_every_leg_module_imports_without_a_gpu():
    import importlib
    for m in ("cpu", "density", "hamming", "host_fed", "per_frame", "pmc", "sparsification", "split", "stereo", "tracking", "unchanged"):
        importlib.import_module("bench_legs." + m)


def test_bench_py_still_names_the_contract_fields():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for key in ('"metric"', '"value"', '"unit"', '"n_gpus"', '"steps"', '"warmup"', '"ms_per_step"', '"higher_is_better"', '"scaling"',
                '"vs_baseline"', '"dtype"', '"data"', '"config"', '"roofline"', '"cpu_baseline"'):
        assert key in src, key
