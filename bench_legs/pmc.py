"""roofline.traffic measured live: child rocprofv3 --pmc passes of bench.py."""
import os
import sys

from . import ROOT


def live_pmc(kernel_substr, counters=("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"), timeout_s=150):
    """roofline.traffic measured in THIS run: one rocprofv3 --pmc pass per counter (separate passes, kernel trace only — the
    guide's recipe) over a child `bench.py --steps 2 --warmup 1 --lean --isolated --no-pmc` (the same batch, every kernel alone
    on the GPU), per-launch average of the dominant kernel.  -> {counter: value} or None when rocprofv3 is missing, refuses the
    counter or does not finish (the line then falls back to the committed PMC summary and says so)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return None
    out = {}
    for ctr in counters:
        d = tempfile.mkdtemp(prefix="msorb_pmc_", dir="/tmp")
        try:
            env = dict(os.environ, TMPDIR="/tmp")
            subprocess.run([rp, "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
                            os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--lean", "--isolated", "--no-pmc", "--cpu-pairs", "0"],
                           cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=False)
            files = glob.glob(os.path.join(d, "**", "p_counter_collection.csv"), recursive=True)
            if not files:
                return None
            vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(files[0]))
                    if r["Counter_Name"] == ctr and kernel_substr in r["Kernel_Name"]]
            if not vals:
                return None
            out[ctr] = sum(vals) / len(vals)
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return out
