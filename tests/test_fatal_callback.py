"""The edges of the product (VERDICT r4 #8a): the host classes call a registered fatal-error callback before their
default action, and the library carries an ABI version the host layer and the Python mirror check.  CPU tests: the failure is
provoked with a device ordinal that does not exist, which libmsorb answers with MSORB_E_NO_DEVICE with or without a GPU."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory, msorb_mod):
    out = tmp_path_factory.mktemp("fatal") / "fatal_callback"
    subprocess.check_call(["g++", "-std=c++17", "-O1", f"-I{ROOT}/tests/cv_stub", f"-I{ROOT}/ms-slam_amd/host", f"-I{ROOT}/include",
                           f"{ROOT}/tests/fatal_callback_main.cc", f"{ROOT}/ms-slam_amd/host/ORBextractor.cc", f"-L{ROOT}/ms-slam_amd",
                           "-lmsorb", f"-Wl,-rpath,{ROOT}/ms-slam_amd", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", str(out)])
    return str(out)


def _run(exe, mode, **env):
    e = dict(os.environ, MSORB_DEVICE="4096", **env)
    e.pop("MSORB_DEVICES", None)
    if "MSORB_THROW" not in env:
        e.pop("MSORB_THROW", None)
    p = subprocess.run([exe, mode], env=e, capture_output=True, text=True, timeout=120)
    return p.returncode, p.stdout, p.stderr


def test_handler_gets_control_before_the_default_action(exe):
    rc, out, err = _run(exe, "exit")
    assert rc == 42 and "HANDLER code=-2 user=atlas what=msorb_extractor_create" in out and "constructed" not in out


def test_handler_that_returns_is_followed_by_the_reference_style_exit(exe):
    rc, out, err = _run(exe, "return")
    assert rc == 255 and "HANDLER code=-2" in out and "msorb (GPU ORB extractor): msorb_extractor_create" in err


def test_throw_mode_still_calls_the_handler_first(exe):
    rc, out, err = _run(exe, "throw", MSORB_THROW="1")
    assert rc == 7 and out.index("HANDLER") < out.index("CAUGHT msorb_extractor_create")


def test_without_a_handler_unchanged_callers_behave_as_before(exe):
    rc, out, err = _run(exe, "none")
    assert rc == 255 and "HANDLER" not in out and "msorb (GPU ORB extractor)" in err


def test_abi_version_is_one_number_in_header_library_and_python(exe, msorb_mod):
    hdr = open(os.path.join(ROOT, "include", "msorb.h")).read()
    macro = int(re.search(r"#define MSORB_ABI_VERSION (\d+)", hdr).group(1))
    lib = msorb_mod.lib()
    assert macro == msorb_mod.ABI_VERSION == lib.msorb_abi_version()
    assert lib.msorb_abi_compatible(macro) == 1
    assert lib.msorb_abi_compatible(macro + 1) == 0          # a caller that needs a newer minor than the library has
    assert lib.msorb_abi_compatible(macro + 1000) == 0 and lib.msorb_abi_compatible(macro - 1000) == 0
    rc, out, _ = _run(exe, "abi")
    assert rc == 0 and out.strip() == f"lib={macro} header={macro} compatible=1 older_minor=0 next_major=0"


def test_callback_registry_through_ctypes(msorb_mod):
    lib = msorb_mod.lib()
    seen = []
    FN = C.CFUNCTYPE(None, C.c_int, C.c_char_p, C.c_void_p)
    cb = FN(lambda code, what, user: seen.append((code, what.decode())))
    lib.msorb_set_fatal_callback(C.cast(cb, C.c_void_p), None)
    lib.msorb_notify_fatal(-3, b"lost the GPU")
    lib.msorb_set_fatal_callback(None, None)
    lib.msorb_notify_fatal(-3, b"nobody listens")
    assert seen == [(-3, "lost the GPU")]


def test_dense_batch_keeps_its_round3_parameter_list():
    """ABI 5000 appends msorb_hamming_dense_top2_batch_ex(..., elapsed_ms, formulation); the older entry keeps its 15 parameters."""
    hdr = open(os.path.join(ROOT, "include", "msorb.h")).read()
    old = re.search(r"int msorb_hamming_dense_top2_batch\((.*?)\);", hdr, re.S).group(1)
    new = re.search(r"int msorb_hamming_dense_top2_batch_ex\((.*?)\);", hdr, re.S).group(1)
    assert old.count(",") == 14 and old.rstrip().endswith("float* elapsed_ms")
    assert new.count(",") == 15 and new.rstrip().endswith("int formulation")
