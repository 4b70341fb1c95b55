"""The legs of bench.py, one module each (the JSON line they produce is assembled in bench.py and did not change when the
file was split in round 5).  Two kinds of failure:

  * a PARITY self-check (`self_check`: a `*_matches_cpu`, `identical_results`, `same_matches_*` that does not hold) raises
    SelfCheckError and aborts the run — a bench line is only printed when every cross-check it reports holds;
  * anything else that goes wrong inside an OPTIONAL leg (a missing rocprofv3, a compiler that is not there, a timeout of a
    child process, an out-of-memory in a side measurement) is caught by `optional_leg` and becomes {"error": "..."} in that
    leg's place: the headline `value` / `roofline` / `cpu_baseline` still reach the driver.
"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KITTI_MBF = 386.1448            # Camera.bf of Examples/Stereo/KITTI00-02.yaml
KITTI_MB = KITTI_MBF / 718.856  # mb = mbf / fx
HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec (MI355X_MICROARCH.md)

for _p in (os.path.join(ROOT, "ms-slam_amd"),):
    if _p not in sys.path:
        sys.path.insert(0, _p)


class SelfCheckError(AssertionError):
    """a cross-check of the bench line does not hold: never degraded to an "error" field"""


def self_check(ok, what):
    """Every cross-check the line reports (`*_matches_cpu`, `identical_results`, `same_matches_*`) is enforced: a bench line is
    only printed when all of them hold, so a `false` can never appear as a result."""
    if not ok:
        raise SelfCheckError("bench self-check failed: " + what)


def optional_leg(name, fn, *args, **kwargs):
    """Run an optional leg; a parity failure (SelfCheckError / AssertionError) propagates, any other exception becomes the leg's
    value: {"error": "<type>: <message>", "leg": name}.  MSORB_BENCH_FAIL_LEG=<name> makes the named leg fail on purpose (test
    hook of tests/test_bench_legs.py)."""
    try:
        if os.environ.get("MSORB_BENCH_FAIL_LEG") == name:
            raise RuntimeError("forced failure of the optional leg (MSORB_BENCH_FAIL_LEG)")
        return fn(*args, **kwargs)
    except AssertionError:
        raise
    except Exception as e:  # noqa: BLE001 — the point of the wrapper
        sys.stderr.write(f"bench.py: optional leg '{name}' failed and is reported as an error field:\n{traceback.format_exc()}\n")
        return {"error": f"{type(e).__name__}: {e}"[:400], "leg": name}


def oracle_module():
    """the CPU oracle (oracle/orb_oracle.py): imported only by the cpu_baseline / cross-check legs, never inside the timed region"""
    p = os.path.join(ROOT, "oracle")
    if p not in sys.path:
        sys.path.insert(0, p)
    import orb_oracle
    return orb_oracle


def tests_dir():
    p = os.path.join(ROOT, "tests")
    if p not in sys.path:
        sys.path.insert(0, p)
    return p
