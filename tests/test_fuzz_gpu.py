"""A bounded re-draw of the round-4 fuzz campaigns inside the driver's suite: tools/fuzz_{params,geom,stereo,track,
visibility}.py — random ORB parameters and image statistics, random geometries / pitches / alignments, hand-placed stereo
keypoints incl. plane borders, the tracking chain (a13 / a14) over random cameras and poses, random sparsification windows — each
against the CPU oracle, a handful of cases per run.  The seed changes with the tree: it is derived from the bytes of the built
libmsorb.so (the GPU box has no .git), so every round's GPUTEST draws different cases; MSORB_FUZZ_SEED pins it, and a failure
prints the seed that reproduces it: `python tools/fuzz_<name>.py <args>` on the GPU box."""
import contextlib
import hashlib
import io
import os
import runpy
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _seed(msorb_mod):
    if os.environ.get("MSORB_FUZZ_SEED"):
        return int(os.environ["MSORB_FUZZ_SEED"])
    return int(hashlib.sha256(open(msorb_mod.LIB_PATH, "rb").read()).hexdigest()[:7], 16)


def _run(script, argv):
    buf = io.StringIO()
    old = sys.argv
    sys.argv = [script] + [str(a) for a in argv]
    code = 0
    try:
        with contextlib.redirect_stdout(buf):
            runpy.run_path(os.path.join(ROOT, "tools", script), run_name="__main__")
    except SystemExit as e:
        code = int(e.code or 0)
    finally:
        sys.argv = old
    return code, buf.getvalue()


CASES = [("fuzz_params.py", lambda s: [s, 10], "bad 0"),
         ("fuzz_geom.py", lambda s: [s, 8], "bad 0"),
         ("fuzz_stereo.py", lambda s: [s, 4, 6], "bad 0"),          # 6 pairs: the four-keypoints-per-wave kernel
         ("fuzz_track.py", lambda s: [8, s % 100000], "mismatches 0"),
         ("fuzz_visibility.py", lambda s: [s, 15], "bad 0")]


@pytest.mark.parametrize("script,args,verdict", CASES, ids=[c[0][5:-3] for c in CASES])
def test_bounded_fuzz(msorb_mod, oracle, script, args, verdict):
    seed = _seed(msorb_mod)
    argv = args(seed)
    code, out = _run(script, argv)
    assert code == 0 and verdict in out and "MISMATCH" not in out and "EXC" not in out, \
        f"reproduce with: python tools/{script} {' '.join(str(a) for a in argv)}\n{out[-3000:]}"


def test_bounded_fuzz_two_camera_arms(msorb_mod, oracle):
    """the two-camera matcher arms on drawn rigs (the draws of tests/_fuzz_matcher.py's rig entries, 6 per arm and tree): the first run of
    that driver over them found a claim-replay case no hand-written rig had (a point without observations freeing its own right partner)"""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_matcher_rig_gpu as tr
    seed = _seed(msorb_mod)

    def tolerant(fn, *a):
        try:
            fn(*a)
        except AssertionError as e:          # population asserts ("enough matches") of the regular tests may fail on a drawn size; parity asserts carry an `==` / array_equal
            msg = str(e)
            if "array_equal" in msg or "==" in msg:
                raise AssertionError(f"MSORB_FUZZ_SEED={seed} {fn.__name__}{a[2:]}: {msg[:1500]}")
    for i in range(6):
        r = np.random.Generator(np.random.PCG64(seed * 100 + i))
        case = dict(seed=int(r.integers(10, 10 ** 6)), n_left=int(r.integers(0, 2500)), n_right=int(r.integers(0, 2500)), M=int(r.integers(0, 7000)),
                    dense=bool(r.integers(0, 2)), th=float(r.uniform(0.5, 6)))
        tolerant(tr.test_search_by_projection_two_camera_frame, msorb_mod, oracle, case)
        tolerant(tr.test_search_by_projection_last_frame_two_camera_tables, msorb_mod, oracle, int(r.integers(10, 10 ** 6)), int(r.integers(0, 2500)),
                 int(r.integers(0, 2500)), int(r.integers(0, 5000)), float(r.uniform(1, 25)))
        n2 = int(r.integers(0, 4000))
        tolerant(tr.test_search_by_bow_two_camera_frame, msorb_mod, oracle, int(r.integers(10, 10 ** 6)), int(r.integers(0, 2500)), n2, int(r.integers(0, n2 + 1)),
                 bool(r.integers(0, 2)))
        tolerant(tr.test_fuse_search_right_camera_of_a_two_camera_keyframe, msorb_mod, oracle, int(r.integers(4, 10 ** 6)), int(r.integers(0, 2500)),
                 int(r.integers(1, 2500)), float(r.uniform(1, 8)))
        tolerant(tr.test_search_for_triangulation_with_the_callers_geometric_test, msorb_mod, oracle, int(r.integers(10, 10 ** 6)), int(r.integers(0, 3000)),
                 int(r.integers(0, 4000)), float(r.uniform(0, 1)), bool(r.integers(0, 2)))


def test_drawn_scenes_through_the_two_camera_class_paths(msorb_mod, oracle, tmp_path):
    """the class-level two-camera tests (Frame / KeyFrame stand-ins through ORB_SLAM3::ORBmatcher) on scenes drawn from this tree's seed"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_matcher_rig_gpu as tr
    seed = 1000 + _seed(msorb_mod) % 1000000

    def tolerant(fn, *a):
        try:
            fn(*a)
        except AssertionError as e:
            msg = str(e)
            if "array_equal" in msg or "==" in msg:
                raise AssertionError(f"{fn.__name__} with seed {a[3:]}: {msg[:1500]}")
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
    tolerant(tr.test_class_search_by_projection_on_two_camera_frames, msorb_mod, oracle, tmp_path / "a", seed, ["forward", "backward", "side"][seed % 3], bool(seed & 8))
    tolerant(tr.test_class_triangulation_and_fuse_on_two_camera_keyframes, msorb_mod, oracle, tmp_path / "b", seed + 1, bool(seed & 16))
