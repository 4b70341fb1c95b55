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
