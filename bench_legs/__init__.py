"""The legs of bench.py, one module each (the JSON line they produce is assembled in bench.py and did not change when the
file was split in round 5).  Two kinds of failure:

  * a PARITY self-check (`self_check`: a `*_matches_cpu`, `identical_results`, `same_matches_*` that does not hold) raises
    SelfCheckError and aborts the run — a bench line is only printed when every cross-check it reports holds;
  * a failure of the LIBRARY or the device inside any leg (msorb.MsorbError from a C-ABI entry, a HIP error surfacing through
    torch) propagates too: a regression in a kernel must fail the run, not hide in a field of a line that exits 0;
  * an ENVIRONMENT failure inside an OPTIONAL leg (a missing rocprofv3, a compiler that is not there, a timeout of a child
    process, an out-of-memory in a side measurement) is caught by `optional_leg` and becomes {"error": "..."} in that leg's
    place, and the leg's name is listed in the line's top-level "degraded_legs": the headline `value` / `roofline` /
    `cpu_baseline` still reach the driver.
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


DEGRADED = []   # names of the optional legs that ended as an "error" field in this process (bench.py prints them as "degraded_legs")


def is_library_failure(e):
    """A failure of libmsorb or of the device (never degraded): msorb.MsorbError — an error code of a C-ABI entry — or a HIP error
    torch reports (a kernel fault shows up at the next synchronising torch call)."""
    if any(c.__name__ == "MsorbError" for c in type(e).__mro__):
        return True
    msg = str(e)
    return isinstance(e, RuntimeError) and ("HIP error" in msg or "hipError" in msg or "CUDA error" in msg)


def optional_leg(name, fn, *args, **kwargs):
    """Run an optional leg; a parity failure (SelfCheckError / AssertionError) and a library / device failure (is_library_failure)
    propagate, any other exception — the environment's — becomes the leg's value: {"error": "<type>: <message>", "leg": name} and the
    name is appended to DEGRADED.  MSORB_BENCH_FAIL_LEG=<name> makes the named leg fail on purpose with an environment-type error,
    MSORB_BENCH_FAIL_LEG=<name>:library with a library-type one (test hooks of tests/test_bench_legs_cpu.py)."""
    try:
        hook = os.environ.get("MSORB_BENCH_FAIL_LEG", "")
        if hook == name:
            raise OSError("forced failure of the optional leg (MSORB_BENCH_FAIL_LEG)")
        if hook == name + ":library":
            raise type("MsorbError", (RuntimeError,), {})("forced library failure of the leg (MSORB_BENCH_FAIL_LEG)")
        return fn(*args, **kwargs)
    except AssertionError:
        raise
    except Exception as e:  # noqa: BLE001 — the point of the wrapper
        if is_library_failure(e):
            sys.stderr.write(f"bench.py: leg '{name}' hit a library / device failure — not degraded, the run fails:\n")
            raise
        sys.stderr.write(f"bench.py: optional leg '{name}' failed and is reported as an error field:\n{traceback.format_exc()}\n")
        DEGRADED.append(name)
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
