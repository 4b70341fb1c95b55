"""BASELINE configs[3] as far as ONE GPU can exercise it, inside `pytest -m gpu`: the whole `bench.py --gpus 2`
loop — two ranks launched by torch.distributed.run, one eye each, half-swap of the feature blocks, split association,
self-validation — with both ranks on cuda:0 and gloo carrying the exchange (MSORB_DIST_BACKEND=gloo); the RCCL launch itself
failing loudly and early on a box with fewer GPUs than ranks (on a box with >= 2 GPUs the same test runs the real RCCL path);
and the bench line surviving a failing optional leg."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _env(**kw):
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **kw)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    return e


def _last_json(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert lines, stdout[-2000:]
    return json.loads(lines[-1])


def _launch(n, port, extra, env, timeout=600):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + extra
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_two_ranks_share_one_gpu_over_gloo():
    port = 29000 + os.getpid() % 2000
    p = _launch(2, port, ["--pairs", "16", "--steps", "4", "--warmup", "2"], _env(MSORB_DIST_BACKEND="gloo"))
    assert p.returncode == 0, p.stderr[-3000:]
    out = _last_json(p.stdout)
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["value"] > 0 and out["scaling"] == "weak"
    v = out["split_validation"]
    assert v["backend"] == "gloo" and v["rccl"] is False and v["ranks_seen"] == 2
    assert sorted(r["rank"] for r in v["ranks"]) == [0, 1]
    for c in v["checks"]:
        assert c["gathered_features_equal_local"] is True and c["split_association_equals_local"] is True and c["matched"] > 1000, c
    assert out["stereo_join"]["joins"] >= 4 and out["stereo_join"]["pairs_per_join"] == 16
    # the same images on one GPU (32 pairs = the two ranks' 32 left + 32 right eyes): the same keypoints per step
    q = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--pairs", "32", "--steps", "4", "--warmup", "2", "--lean"],
                       cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=600)
    assert q.returncode == 0, q.stderr[-3000:]
    one = _last_json(q.stdout)
    per_rank = sorted(r["keypoints_per_step"] for r in v["ranks"])
    assert sum(per_rank) == out["keypoints_per_step"] == one["keypoints_per_step"], (per_rank, out["keypoints_per_step"], one["keypoints_per_step"])


def test_bench_rccl_launch_is_a_number_or_a_diagnosis():
    import torch
    port = 31000 + os.getpid() % 2000
    p = _launch(2, port, ["--pairs", "8", "--steps", "3", "--warmup", "1"], _env(MSORB_DIST_TIMEOUT_S="60"), timeout=400)
    if torch.cuda.device_count() >= 2:      # a multi-GPU box: this IS the RCCL run over xGMI
        assert p.returncode == 0, p.stderr[-3000:]
        out = _last_json(p.stdout)
        assert out["split_validation"]["rccl"] is True and out["split_validation"]["ranks_seen"] == 2
        for c in out["split_validation"]["checks"]:
            assert c["gathered_features_equal_local"] is True and c["split_association_equals_local"] is True
        return
    assert p.returncode != 0
    # a clear message from the ranks (the launcher may end a rank before it has printed its own), no JSON line, no hang
    assert any(f"bench.py rank {r}/2" in p.stderr for r in (0, 1)) and "device check FAILED" in p.stderr, p.stderr[-3000:]
    assert "needs 2 visible GPUs, found 1" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_a_failing_optional_leg_costs_the_leg_not_the_line():
    q = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--pairs", "16", "--steps", "4", "--warmup", "2", "--cpu-pairs", "0",
                        "--no-pmc"], cwd=ROOT, env=_env(MSORB_BENCH_FAIL_LEG="per_frame"), capture_output=True, text=True, timeout=900)
    assert q.returncode == 0, q.stderr[-3000:]
    out = _last_json(q.stdout)
    assert out["value"] > 0 and out["roofline"]["achieved"] > 0
    assert out["per_frame"]["leg"] == "per_frame" and "forced failure" in out["per_frame"]["error"]
    assert out["degraded_legs"] == ["per_frame"]
    assert out["tracking_loop"]["per_frame"]["ms_one_call"] > 0 and out["sparsification"]["ms_per_window"] > 0   # the other legs ran


def test_a_library_failure_inside_a_leg_fails_the_run():
    """round-5 advice: optional_leg used to turn an MsorbError of a core leg into an "error" field of a line that exits 0"""
    q = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--pairs", "16", "--steps", "4", "--warmup", "2", "--cpu-pairs", "0",
                        "--no-pmc"], cwd=ROOT, env=_env(MSORB_BENCH_FAIL_LEG="hamming_match:library"), capture_output=True, text=True, timeout=900)
    assert q.returncode != 0 and "forced library failure" in q.stderr
    assert not [l for l in q.stdout.splitlines() if l.startswith("{")]
