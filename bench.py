#!/usr/bin/env python3
"""bench.py — ORB front-end throughput on MI355X (BASELINE.json metric, configs[1] workload).

A "step" = one pass of the hot path (ORBextractor::operator(): pyramid -> FAST+NMS -> quadtree -> blur ->
IC-angle -> rBRIEF) over one batch of synthetic KITTI-00-like stereo pairs (1241x376, 2000 features/frame),
inputs already resident in HBM.  value = keypoints returned per second, whole job.

  python bench.py --gpus N --steps K --warmup W [--pairs B]

N=1: both eyes of every pair on the one GPU.  N>1 (launched by torch.distributed.run, one rank per GPU):
stereo left/right split — even ranks extract the left eyes, odd ranks the right eyes of their pair group,
then the odd rank sends counts/keypoints/descriptors to its even partner over RCCL (xGMI point-to-point);
per-GPU work is fixed (weak scaling).

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, HIP-event timed) and `cpu_baseline`
(the CPU oracle timed on a bounded sample of the same workload on this box's host cores).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ms-slam_amd")]

# The legs live in bench_legs/ (one module each); this file keeps the contract: the CLI, the timed loop, the JSON line.
from bench_legs import HBM_PEAK_GBS, KITTI_MB, KITTI_MBF, optional_leg, self_check  # noqa: E402
from bench_legs.cpu import cpu_baseline, level_bytes  # noqa: E402
from bench_legs.density import density_sweep_leg, fourseasons_leg  # noqa: E402
from bench_legs.hamming import hamming_cpu_leg, hamming_leg  # noqa: E402
from bench_legs.host_fed import host_fed_leg  # noqa: E402
from bench_legs.per_frame import per_frame_leg  # noqa: E402
from bench_legs.pmc import live_pmc  # noqa: E402
from bench_legs.sparsification import sparsification_leg  # noqa: E402
from bench_legs.split import split_self_validation  # noqa: E402
from bench_legs.stereo import stereo_leg  # noqa: E402
from bench_legs.tracking import reference_keyframe_leg, tracking_cpu_leg, tracking_loop_leg  # noqa: E402

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--pairs", type=int, default=128, help="stereo pairs per step per GPU-pair group")
    ap.add_argument("--cpu-pairs", type=int, default=160, help="stereo pairs in the CPU baseline sample (0 = skip)")
    ap.add_argument("--unique-pairs", type=int, default=8, help="distinct synthetic stereo pairs tiled to the batch (<= --pairs)")
    ap.add_argument("--lean", action="store_true",
                    help="profiling aid: only the extraction loop (no Hamming / stereo / tracking / per-frame / host-fed / sparsification "
                         "legs, no CPU baseline), so that a kernel trace or a counter pass holds nothing else")
    ap.add_argument("--no-pmc", action="store_true", help="do not collect roofline.traffic live with rocprofv3 (child passes of this script)")
    ap.add_argument("--isolated", action="store_true",
                    help="profiling aid: no sub-batch / blur overlap anywhere, so every kernel launch covers the whole "
                         "batch and runs alone (rocprofv3 per-kernel durations and PMC traffic are then per-launch clean)")
    args = ap.parse_args()
    if args.lean:
        args.cpu_pairs = 0

    import torch
    import msorb
    from msorb import synth

    cfg = synth.KITTI
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs WORLD_SIZE={args.gpus} (launch with torch.distributed.run)")
    # test aid: MSORB_DIST_BACKEND=gloo lets the N>1 path run with several ranks on ONE GPU (ranks share cuda:0, the
    # exchange goes through gloo) — how the stereo-split code path is exercised on the 1-GPU development box
    backend = os.environ.get("MSORB_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import datetime
        import torch.distributed as dist

        def die(stage, err):
            # one clear line per rank, then a non-zero exit: a SCALE record is either a number or this diagnosis
            sys.stderr.write(f"bench.py rank {rank}/{world} (cuda:{local}, backend {backend}): {stage} FAILED: {type(err).__name__}: {err}\n"
                             f"  visible GPUs: {torch.cuda.device_count()}; MASTER_ADDR={os.environ.get('MASTER_ADDR')} "
                             f"MASTER_PORT={os.environ.get('MASTER_PORT')} HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}\n"
                             "  (RCCL needs one visible GPU per rank and dmabuf IPC: HSA_ENABLE_IPC_MODE_LEGACY=0; use 127.0.0.1 for the rendezvous)\n")
            sys.stderr.flush()
            time.sleep(1.0)   # the launcher kills the other ranks as soon as one exits: give them the time to print their own line
            os._exit(3)
        if backend == "nccl" and torch.cuda.device_count() < world:
            die("device check", RuntimeError(f"--gpus {world} needs {world} visible GPUs, found {torch.cuda.device_count()} "
                                             "(MSORB_DIST_BACKEND=gloo runs the N > 1 code path with the ranks sharing one GPU)"))
        tmo = datetime.timedelta(seconds=int(os.environ.get("MSORB_DIST_TIMEOUT_S", "120")))
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev, timeout=tmo)
            else:
                dist.init_process_group(backend, timeout=tmo)
        except Exception as e:  # noqa: BLE001
            die("init_process_group", e)
        try:
            # first contact, before any buffer is allocated: an all_reduce (every rank present) and the pair's send / receive
            probe = torch.ones(1, dtype=torch.int32, device=dev)
            dist.all_reduce(probe)
            if int(probe.item()) != world:
                raise RuntimeError(f"all_reduce over {world} ranks returned {int(probe.item())}")
            peer = rank ^ 1
            if peer < world:
                token = torch.full((1,), rank, dtype=torch.int32, device=dev)
                got = torch.zeros_like(token)
                for w in dist.batch_isend_irecv([dist.P2POp(dist.irecv, got, peer), dist.P2POp(dist.isend, token, peer)]):
                    w.wait()
                torch.cuda.synchronize()
                if int(got.item()) != peer:
                    raise RuntimeError(f"point-to-point probe with rank {peer} returned {int(got.item())}")
        except Exception as e:  # noqa: BLE001
            die("first collective / point-to-point exchange", e)

    B = args.pairs
    # images of this rank: N=1 -> L,R interleaved (2B images); N>1 -> one eye of 2B pairs (2B images): fixed per-GPU work
    group, eye = (rank // 2, rank % 2) if world > 1 else (0, None)
    uniq = min(B, args.unique_pairs)  # distinct synthetic pairs, tiled to the batch size (content repeats, bytes do not alias)
    base = synth.stereo_batch(uniq, cfg["rows"], cfg["cols"], seed0=1000 * group)
    pitch = (cfg["cols"] + 63) // 64 * 64

    def resident(host_imgs):
        # device-resident input batch with a 64-byte row pitch (what hipMemcpy2D / a camera DMA would produce);
        # the API accepts any stride, aligned rows take the fast kernel variants
        st = torch.zeros((host_imgs.shape[0], cfg["rows"], pitch), dtype=torch.uint8, device=dev)
        v = st[:, :, :cfg["cols"]]
        v.copy_(torch.from_numpy(np.ascontiguousarray(host_imgs)).to(dev))
        return v

    other_images = None
    half = B   # N>1: images of a rank whose stereo pairs it associates itself (the first `half` of its 2B images)
    if world == 1:
        host = np.concatenate([base] * (B // uniq + 1))[:2 * B]
    else:
        # both ranks of a pair hold the same 2B stereo pairs, one eye each.  The association (Frame::ComputeStereoMatches) is
        # split evenly: the left-eye rank joins pairs [0, B), the right-eye rank pairs [B, 2B) — so the right-eye rank keeps its
        # images in the order [B, 2B) + [0, B): every rank joins the pairs of its FIRST B images and ships the features of its
        # LAST B images to its partner (msorb/stereo_split.swap_halves_async).  A rank also holds the other eye's images of
        # the pairs it joins (the capture DMA delivers both eyes): it rebuilds that pyramid for the stereo SAD locally
        # (msorb_pyramid_batch, 0.09 ms per 128 images) instead of pulling 1.5 MB per frame over xGMI; only the 60 B per
        # keypoint cross the link (BASELINE configs[3]: "gather of keypoints/descriptors")
        def eye_images(e):
            one = base[e::2]
            return np.concatenate([one] * (2 * B // uniq + 1))[:2 * B]
        own, oth = eye_images(eye), eye_images(1 - eye)
        if eye == 1:
            own = np.concatenate([own[B:], own[:B]])
            oth = np.concatenate([oth[B:], oth[:B]])
        host = own
        other_images = resident(oth[:half])
    n_img = host.shape[0]
    images = resident(host)

    def make_ex():
        e = msorb.ORBextractor(cfg["nfeatures"], cfg["scale"], cfg["nlevels"], cfg["ini_th"], cfg["min_th"], device=local)
        if args.isolated:
            e.set_overlap(1, False)
        return e

    ex = make_ex()
    cap = ex.capacity
    d_kps = torch.empty((n_img, cap, 28), dtype=torch.uint8, device=dev)
    d_desc = torch.empty((n_img, cap, 32), dtype=torch.uint8, device=dev)
    from msorb import stereo_split
    # N>1: two send/receive blocks used alternately, so that the exchange of step k (RCCL, torch's communication stream)
    # overlaps the extraction of step k+1 (the library's own streams) without the two touching the same memory.  The
    # left-eye rank alternates two extractor handles the same way: the stereo association of step k (Frame.cc:743-913) runs
    # when the right eye's features of step k have arrived — two steps later, just before block and handle are reused —
    # and needs the left pyramid of step k intact.
    mine = theirs = None
    pending = [[], []]
    exs = [ex, ex]
    ex_rp = None
    assoc = {"ms": 0.0, "n": 0, "matched": 0}
    if world > 1:
        mine = [stereo_split.FeatureBlock(n_img, cap, dev) for _ in range(2)]
        theirs = [stereo_split.FeatureBlock(half, cap, dev) for _ in range(2)]
        exs = [ex, make_ex()]
        if not (args.isolated or os.environ.get("MSORB_BENCH_SYNC")):
            for e in exs:
                e.set_overlap(1, True)   # two batches in flight, one sub-batch each (see the N = 1 loop below)
        ex_rp = make_ex()   # pyramid-only handle for the other eye's images of the pairs this rank joins
    step_no = [0]
    fence_kp = [0]
    filled = [False, False]
    last_ex = [ex]
    all_ex = [ex] if exs[1] is ex else [exs[0], exs[1]]

    def associate(b):
        """Frame::ComputeStereoMatches for the pairs of block b this rank joins (its exchange has completed): own features of
        the first `half` images + the partner's features of the same pairs."""
        if world == 1 or not filled[b]:
            return
        ex_rp.pyramid_batch(other_images)
        own = (mine[b].counts[:half], mine[b].kps[:half], mine[b].desc[:half])
        got = (theirs[b].counts, theirs[b].kps, theirs[b].desc)
        if eye == 0:
            d_ur, _, _, ms = msorb.stereo_matches_split(exs[b], ex_rp, *own, *got, KITTI_MB, KITTI_MBF)
        else:
            d_ur, _, _, ms = msorb.stereo_matches_split(ex_rp, exs[b], *got, *own, KITTI_MB, KITTI_MBF)
        assoc["ms"] += ms
        assoc["n"] += 1
        assoc["last"] = d_ur
        filled[b] = False

    # N=1: two batches in flight.  A synchronous msorb_extract_batch per step leaves the GPU under-used across step boundaries
    # (the last sub-batch's quadtree / descriptors and the next step's small pyramid levels run alone: ~25 % of a step in the
    # kernel timeline); with msorb_extract_batch_submit / _wait on two alternating handles, step k + 1 is enqueued before step k is
    # waited for — what a capture pipeline that always has the next batch ready does.  Every step still runs the full chain
    # on its own batch buffers; the timed region is bracketed by fence() on both sides.  (--isolated / the stage timings use
    # the synchronous call.)
    pipelined = world == 1 and not args.isolated and not os.environ.get("MSORB_BENCH_SYNC")
    mode = {"pipelined": pipelined, "sync_nccl": bool(args.isolated or os.environ.get("MSORB_BENCH_SYNC"))}
    submitted = [False, False]
    depth = int(os.environ.get("MSORB_BENCH_DEPTH", "2")) if pipelined else 1   # batches in flight
    exp = [ex] + [make_ex() for _ in range(depth - 1)]
    all_ex.extend(exp[1:])
    if pipelined:
        # one sub-batch per handle: the concurrency comes from the two batches (measured: 1.35 ms per step against 1.51 with two
        # sub-batches per handle — six streams on four hardware queues — and 1.44 for the one-batch-at-a-time loop)
        for e in exp:
            e.set_overlap(1, True)
    outs = [(d_kps, d_desc)] + [(torch.empty_like(d_kps), torch.empty_like(d_desc)) for _ in range(depth - 1)]
    inflight = []
    refill = [True]   # from a fence until `depth` batches are in flight again
    # Phase between the two batches in flight: started together they stay in lock-step (both finish, and are resubmitted,
    # together: FAST beside FAST, descriptors beside descriptors); half a step apart, one batch's pyramid / FAST (VALU issue)
    # runs beside the other's quadtree (latency) and descriptors (line fills): 1.147 instead of 1.184 ms per step (round 4; a
    # capture pipeline whose batches arrive evenly spaced is in this state by itself).  The offset is half of the sum of the
    # stages' own times (MSORB_BENCH_STAGGER_US overrides; 0 = lock-step), applied once after every fence, inside the timed region.
    stagger = [float(os.environ["MSORB_BENCH_STAGGER_US"]) * 1e-6 if "MSORB_BENCH_STAGGER_US" in os.environ else None]
    if not pipelined or args.steps < 8:   # (a handful of steps cannot pay for the half step the offset costs once)
        stagger[0] = 0.0

    def drain():
        total = 0
        refill[0] = True
        while inflight:
            counts, _, _, _ = inflight.pop(0).extract_batch_wait()
            total += int(counts.sum())
        return total

    def step():
        if world == 1:
            if not mode["pipelined"]:
                counts, mono, _, _ = ex.extract_batch(images, (0, 0), out=(d_kps, d_desc))
                last_ex[0] = ex
                return int(counts.sum())
            k = step_no[0] % depth
            step_no[0] += 1
            done = 0
            if len(inflight) == depth:  # handle k still holds the batch submitted `depth` steps ago
                counts, _, _, _ = inflight.pop(0).extract_batch_wait()
                done = int(counts.sum())
            exp[k].extract_batch_submit(images, (0, 0), out=outs[k])
            inflight.append(exp[k])
            if len(inflight) == depth:
                refill[0] = False
            last_ex[0] = exp[k]
            if 0 < len(inflight) < depth and refill[0] and stagger[0]:
                time.sleep(stagger[0])   # first batch after a fence: hold the second one back (see `stagger` above)
            return done
        b = step_no[0] & 1
        step_no[0] += 1
        stereo_split.finish(pending[b])          # the block's previous exchange (started one step ago) must be over
        pending[b] = []
        associate(b)                              # ... and its pairs are joined before block / handle are reused
        if mode["sync_nccl"]:                     # stage timings: one batch at a time
            last_ex[0] = exs[b]
            counts, mono, _, _ = exs[b].extract_batch(images, (0, 0), out=(mine[b].kps, mine[b].desc))
            ship(b, counts)
            return int(counts.sum())
        # two batches in flight, as at N = 1: this step's extraction is enqueued (the extractor writes straight into the send
        # block), then the previous step's is waited for and its feature block goes on the wire
        exs[b].extract_batch_submit(images, (0, 0), out=(mine[b].kps, mine[b].desc))
        submitted[b] = True
        return collect(b ^ 1)

    def ship(b, counts):
        """stereo split: swap the feature blocks of block b with the partner (replaces the join of Frame.cc:122-125)"""
        mine[b].counts.copy_(torch.from_numpy(counts))
        pending[b] = stereo_split.swap_halves_async(dist, rank, world, mine[b], theirs[b])
        filled[b] = bool(pending[b])

    def collect(b):
        if not submitted[b]:
            return 0
        counts, _, _, _ = exs[b].extract_batch_wait()
        submitted[b] = False
        last_ex[0] = exs[b]
        ship(b, counts)
        return int(counts.sum())

    def fence():
        if mode["pipelined"]:
            fence_kp[0] += drain()
        if world > 1:
            for b in range(2):
                fence_kp[0] += collect(b)
            for b in range(2):
                stereo_split.finish(pending[b])
                pending[b] = []
                associate(b)
            dist.barrier()
        torch.cuda.synchronize()

    # Order of the untimed work: the Hamming, stereo-association and per-kernel stage measurements run BEFORE the warm-up steps,
    # the CPU legs after the timed region.  A GPU that comes out of idle runs its first ~25 ms of work 5-15 % slower (clock
    # ramp: tools/ramp.py prints the per-step times), so the stage measurement discards its first 20 steps — and the timed
    # region that follows starts on a GPU at its working clocks even with `--steps 20 --warmup 5`.
    # second half of the metric: Gpairs/s of the brute-force Hamming match (left-eye descriptors of every pair
    # against the right-eye descriptors of the same pair, dense top-2), on the descriptors just extracted
    hamming = ham_ctx = None
    aux = world == 1 and not args.lean
    counts_h = None
    if aux:
        counts_h, _, _, _ = ex.extract_batch(images, (0, 0), out=(d_kps, d_desc))
        r = optional_leg("hamming_match", hamming_leg, msorb, torch, d_desc, counts_h, dev, local)
        hamming, ham_ctx = r if isinstance(r, tuple) else (r, None)
    # third: Frame::ComputeStereoMatches for the whole batch, device resident (pair p = images 2p / 2p+1), median
    # rejection included; the outputs of the extraction above are its inputs
    stereo = d_ur = None
    if aux:
        r = optional_leg("stereo_match", stereo_leg, msorb, ex, counts_h, d_kps, d_desc)
        stereo, d_ur = r if isinstance(r, tuple) else (r, None)

    # fourth: BASELINE configs[2] — the front-end of one tracking frame as a device-resident chain (csrc/track.hip): extraction of
    # both eyes + ComputeStereoMatches + AssignFeaturesToGrid + isInFrustum + the window search of SearchByProjection
    tracking = None
    if aux:
        tracking = optional_leg("tracking_loop", tracking_loop_leg, msorb, synth, torch, ex, cfg, base, counts_h, d_kps, d_desc, d_ur, dev,
                                local, args)

    per_frame = host_fed = None
    if aux:
        per_frame = optional_leg("per_frame", per_frame_leg, msorb, ex, base[0], base[1])

    # per-kernel roofline: the same step with every kernel alone on the GPU (1 sub-batch, blur on the main stream),
    # HIP events on the launching stream, 20 recorded steps (after 20 discarded ones) outside the timed region
    iso_steps, iso_discard = 20, 20
    for e in all_ex:
        e.set_profiling(True)
    saved_mode = dict(mode)
    stage_acc = {k: [] for k in msorb.STAGES}
    mode["pipelined"] = False   # the stage timings use the synchronous call on one handle
    mode["sync_nccl"] = True
    for e in all_ex:
        e.set_overlap(1, False)
    for _ in range(iso_discard):   # not recorded: lazy allocations, and the GPU's clocks ramp for ~25 ms after idle
        step()
    for _ in range(iso_steps):
        step()
        for k, v in last_ex[0].stage_ms().items():
            stage_acc[k].append(v)
    if not args.isolated:
        for e in all_ex:
            e.set_overlap(1 if pipelined else 2, True)
    fence()
    mode.update(saved_mode)



    if pipelined and stagger[0] is None:
        # half a step, taken from the stages' own times (HIP events, kernels alone on the GPU, measured above) BEFORE the warm-up
        # steps, so that they already run in the timed region's shape.  It used to be half of the warm-up steps' wall time, which
        # reads two to three times too long on a box whose host is slow to start, and an offset of a step and a half is lock-step
        # again: 1.08 instead of 1.05 ms per step (offsets of 0.53-0.80 ms measured equal, 0 / 0.25 / 1.5 ms 3 % slower); and a
        # warm-up in lock-step left a 20-step timed region 8 % slower than a staggered one did (tools/experiments/README.md)
        iso = sum(float(np.median(v)) for v in stage_acc.values() if len(v)) * 1e-3
        stagger[0] = min(max(iso / depth if iso > 0 else 1200e-6 / depth, 200e-6), 2e-3)
    # The stage measurement above runs one synchronous batch at a time: the GPU is half idle through it, and after that its
    # clocks need ~25-40 ms of continuous work to settle (a traced `--steps 20 --warmup 5` run: the first pair of steps of the
    # timed region took 2.37 ms, the tenth pair 2.09, same kernels).  A run with a short warm-up therefore gets the production
    # loop for `settle` steps first — untimed, like the stage measurement — so that `--warmup W` means W steps of a settled loop.
    import gc
    gc.collect()
    gc.disable()   # (one traced 20-step run in eight had a 4 ms hole in the host loop: no collector pass inside the timed region)
    settle = int(os.environ.get("MSORB_BENCH_SETTLE_STEPS", "80")) if (pipelined or (world > 1 and not mode["sync_nccl"])) else 0
    for _ in range(max(0, settle - args.warmup)):
        step()
    t_w = time.perf_counter()
    for _ in range(args.warmup):
        step()
    # timed region: the production shape (2 sub-batches in flight, blur on a second stream), stage events on
    for e in all_ex:
        e.set_profiling(True)
    overlapped_acc = {k: 0.0 for k in msorb.STAGES}
    fence()
    assoc.update(ms=0.0, n=0)
    fence_kp[0] = 0
    t0 = time.perf_counter()
    kp_total = 0
    trace = [] if os.environ.get("MSORB_BENCH_TRACE") else None   # diagnostic: wall time at the end of every step() call, to stderr
    for _ in range(args.steps):
        kp_total += step()
        if trace is not None:
            trace.append(time.perf_counter())
        if not pipelined and (world == 1 or mode["sync_nccl"]):
            for k, v in last_ex[0].stage_ms().items():
                overlapped_acc[k] += v
    fence()
    dt = time.perf_counter() - t0
    gc.enable()
    if trace:
        sys.stderr.write("step ends (us after t0): " + " ".join(f"{(t - t0) * 1e6:.0f}" for t in trace) + f" | fence {dt * 1e6:.0f}\n")
    kp_total += fence_kp[0]
    join = dict(assoc)   # association statistics of the timed region only

    dt_rank, kp_total_rank = dt, kp_total
    if world > 1:
        t = torch.tensor([dt, float(kp_total)], dtype=torch.float64, device=dev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt, kp_total = float(tmax[0]), int(t[1])

    if aux and not args.isolated:
        hf_ex = [ex, make_ex()]
        all_ex.append(hf_ex[1])
        host_fed = optional_leg("host_fed", host_fed_leg, msorb, torch, hf_ex, host, dev, cfg, pitch)

    # input-statistics sweep and configs[4]'s front-end geometry: after the timed region, on their own handles and batches
    density = fourseasons = None
    if aux and not args.isolated:
        density = optional_leg("density_sweep", density_sweep_leg, msorb, synth, torch, make_ex, cfg, dev, B, uniq, args.cpu_pairs > 0,
                               exp[:2] if world == 1 and pipelined and len(exp) >= 2 else None)
        same_params = all(synth.FOURSEASONS[k] == cfg[k] for k in ("nfeatures", "scale", "nlevels", "ini_th", "min_th"))
        fourseasons = optional_leg("fourseasons_frontend", fourseasons_leg, msorb, synth, torch, dev, B, uniq, local, args.cpu_pairs > 0,
                                   exp[:2] if world == 1 and pipelined and len(exp) >= 2 and same_params else None)

    validation = None
    if world > 1:
        validation = split_self_validation(msorb, torch, dist, stereo_split, rank, world, eye, half, exs[0], ex_rp, make_ex, images,
                                           other_images, mine[0], theirs[0], kp_total_rank / dt_rank if dt_rank > 0 else 0.0, dev,
                                           rank_keypoints_per_step=int(round(kp_total_rank / max(args.steps, 1))))

    if hamming is not None and ham_ctx is not None and args.cpu_pairs > 0 and rank == 0:
        err = optional_leg("hamming_match.cpu_baseline", hamming_cpu_leg, msorb, hamming, ham_ctx)
        if err is not None:
            hamming["cpu_baseline"] = err

    if rank == 0:
        steps = max(args.steps, 1)
        # median over the recorded steps (a mean is at the mercy of a few slow steps: clock ramps, a neighbour on the host)
        stages = {k: float(np.median(v)) if v else 0.0 for k, v in stage_acc.items()}
        stages_overlapped = {k: v / steps for k, v in overlapped_acc.items()}
        px = level_bytes(cfg)
        # dominant GPU kernel: FAST cells (one launch per step).  Algorithmic bytes per launch:
        # every level pixel read once + 8 B per emitted candidate (not counted: unknown a priori) per image.
        gpu_stages = {k: stages[k] for k in ("pyramid", "fast", "blur", "describe", "compact")}
        dom = max(gpu_stages, key=gpu_stages.get)
        alg_per_image = {
            "fast": sum(px),                                 # read every level once
            "pyramid": sum(px[:-1]) + sum(px[1:]),           # read L0..L6, write L1..L7 (L0 consumed in place)
            "blur": 2 * sum(px),                             # read + write every level
            "describe": cfg["nfeatures"] * (749 + 512 + 32 + 28),
            "compact": 0,
        }[dom]
        alg_bytes = alg_per_image * n_img
        achieved = alg_bytes / (stages[dom] * 1e-3) / 1e9 if stages[dom] > 0 else 0.0
        pf_bytes = (alg_per_image if dom in ("fast", "pyramid") else 0)
        # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process, so the value is the
        # committed rocprofv3 --pmc measurement of the same command (latest profiles/roundN_pmc_summary.json: separate
        # FETCH_SIZE / WRITE_SIZE passes, per launch, FETCH_SIZE scaled by the calibration kernels of the same run) — only
        # quoted when the batch shape matches
        traffic = None
        kname = {"fast": "fast_cells_kernel<true, GeoSmall>", "pyramid": "pyr_resize_bandreg_kernel<8, 12, 2>", "blur": "gauss7_stream_kernel<35>",
                 "describe": "describe_kernel", "compact": "cand_gather_kernel"}[dom]
        valu_frac = None
        pmc_file = None
        try:
            import glob
            pmc_file = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_pmc_summary.json")))[-1]
            summ = json.load(open(pmc_file))
            pm = summ["kernels"][kname]
            if B == 128 and world == 1:
                scale = summ.get("fetch_scale", 2.0)
                traffic = int((pm["FETCH_SIZE_KB"] * scale + pm["WRITE_SIZE_KB"]) * 1024) * (7 if dom == "pyramid" else 1)
                valu_frac = pm.get("valu_fraction_of_measured_peak")
        except Exception:
            traffic = None
        traffic_source = ("committed " + os.path.basename(pmc_file)) if (traffic is not None and pmc_file) else None
        if aux and not args.no_pmc and not args.isolated and B == 128 and dom == "fast":
            pm_live = optional_leg("roofline.live_pmc", live_pmc, "fast_cells_kernel")
            if pm_live and "error" not in pm_live:
                scale = 2.0    # FETCH_SIZE reports half of the bytes a coalesced stream reads on gfx950 (profiles/round6_fetch_calib.txt)
                traffic = int((pm_live["FETCH_SIZE"] * scale + pm_live["WRITE_SIZE"]) * 1024)
                valu_frac = round(pm_live["SQ_INSTS_VALU"] * 64 / (stages[dom] * 1e-3) / 51.5e12, 3)
                traffic_source = ("measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU, one pass each, over a child "
                                  "`bench.py --lean --isolated` of the same batch")
        out = {
            "metric": "Mkeypoints/s extract+describe, KITTI-00-like stereo 1241x376, 2000 feat/frame",
            "value": round(kp_total / dt / 1e6, 4),
            "unit": "Mkeypoints/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / steps * 1e3, 4),
            "keypoints_per_step": int(round(kp_total / steps)),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "metric_notes": {"inputs": "value assumes the images are already in HBM (bench contract); host_fed is the PCIe-inclusive batch "
                                       "rate, per_frame what one frame at a time from host memory costs",
                             "hamming_match": "gpairs_per_s is the north_star's xor + popcount kernel (no MFMA); the opt-in int8 matrix-core "
                                              "variant is reported under hamming_match.matrix_core_variant, same inputs, same run",
                             "two_gpus": "the product's stereo split (msorb_extract_stereo_split) gathers with hipMemcpyPeerAsync inside one "
                                         "process; RCCL (torch.distributed nccl) carries the exchange only in `bench.py --gpus N`",
                             "see": "BASELINE.md section 3"},
            "config": {"workload": "configs[1]: KITTI-00 stereo 1241x376, 2000 feat/frame, pyramid+FAST+rBRIEF",
                       "split_transport": None if world == 1 else "RCCL point-to-point (torch.distributed nccl backend), one rank per GPU",
                       "pairs_per_step_per_gpu": B if world == 1 else B, "images_per_step_per_gpu": n_img,
                       "stagger_us": round((stagger[0] or 0.0) * 1e6, 1),
                       "untimed_steps_before_warmup": max(0, settle - args.warmup),
                       "batches_in_flight": depth if world == 1 else (1 if args.isolated or os.environ.get("MSORB_BENCH_SYNC") else 2),
                       "parallelism": ("1 GPU, both eyes; msorb_extract_batch_submit / _wait on two alternating handles: step k+1 is enqueued "
                                       "before step k is waited for" if pipelined else "1 GPU, both eyes") if world == 1 else f"stereo L/R split over {world} GPUs: one eye per rank; partners swap the keypoints / "
                                                                             "descriptors of half of their images (RCCL send/recv over xGMI) and each joins half of the pairs (stereo association)"},
            "stage_ms_per_step": {k: round(v, 4) for k, v in stages.items()},
            "stage_ms_per_step_note": "each stage's kernels alone on the GPU (20 extra steps after 20 discarded ones, median, overlap off, HIP events on the "
                                      "launching stream); 'select' = device quadtree + output layout",
            "stage_ms_per_step_overlapped": None if (pipelined or (world > 1 and not args.isolated and not os.environ.get("MSORB_BENCH_SYNC"))) else {k: round(v, 4) for k, v in stages_overlapped.items()},
            "stage_ms_per_step_overlapped_note": "timed region: sum over the 2 concurrent sub-batches of each stage's event "
                                                 "interval (intervals overlap, so the sum exceeds ms_per_step); not recorded when "
                                                 "two batches are in flight (MSORB_BENCH_SYNC=1 for the one-batch-at-a-time loop)",
            "roofline": {"bound": "valu+lds" if dom == "fast" else "hbm",
                         "bound_note": "what the counters say limits this kernel (profiles/round6_fast_pmc.txt: VALU at 0.82 of its measured "
                                       "issue rate, LDS pipe busy 61 % of the launch, HBM traffic 1.05 x algorithmic); achieved / peak / frac "
                                       "are still the HBM yardstick the bench contract prescribes" if dom == "fast" else
                                       "streaming kernel: HBM bandwidth",
                         "kernel": {"fast": "fast_cells_kernel", "pyramid": "pyr_resize_bandreg_kernel (x7)",
                                                    "blur": "gauss7_kernel (x8)", "describe": "describe_kernel",
                                                    "compact": "cand_*"}[dom],
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": traffic_source,
                         "traffic_note": "HBM bytes per launch from rocprofv3 --pmc: FETCH_SIZE x 2 + WRITE_SIZE; FETCH_SIZE reports half of "
                                         "the bytes a coalesced stream reads on gfx950 (tools/fetch_calib.hip, profiles/round6_fetch_calib.txt), "
                                         "WRITE_SIZE is exact",
                         "algorithmic_bytes_per_launch": int(alg_bytes),
                         "valu_fraction_of_measured_peak": valu_frac,
                         "valu_note": "what actually bounds this kernel: 64 x SQ_INSTS_VALU / duration against the 51.5 T lane-ops/s "
                                      "the VALUs sustain (103 TFLOP/s v_fma_f32); same source as traffic"},
            "pyramid_fast_gbs": round((sum(px[:-1]) + sum(px[1:]) + sum(px)) * n_img /
                                      ((stages["pyramid"] + stages["fast"]) * 1e-3) / 1e9, 2),
        }
        out["hamming_match"] = hamming
        out["stereo_match"] = stereo
        if tracking is not None and "error" not in tracking:
            if args.cpu_pairs > 0:
                err = optional_leg("tracking_loop.cpu_baseline", tracking_cpu_leg, tracking, msorb)
                if err is not None:
                    tracking["cpu_baseline"] = err
            tracking.pop("_cpu", None)
            tracking.pop("_cpu_mm", None)
            tracking["reference_keyframe"] = optional_leg("tracking_loop.reference_keyframe", reference_keyframe_leg, msorb, args.cpu_pairs > 0)
        out["tracking_loop"] = tracking
        if aux:
            out["sparsification"] = optional_leg("sparsification", sparsification_leg, msorb, args.cpu_pairs > 0)
        if world > 1:
            out["stereo_join"] = {
                "what": "every rank joins half of its pair group's stereo pairs inside the timed region, once per step: the other "
                        "eye's pyramid of those pairs rebuilt from resident images (msorb_pyramid_batch) + "
                        "Frame::ComputeStereoMatches on its own features and the features gathered from its partner "
                        "(msorb_stereo_matches_split); figures of rank 0",
                "pairs_per_join": half,
                "joins": join["n"], "kernel_ms_per_join": round(join["ms"] / max(join["n"], 1), 4),
                "matched_last": int((join["last"] > 0).sum().item()) if "last" in join else None,
                "gathered_bytes_per_step_and_rank": theirs[0].nbytes() if theirs else None}
        out["per_frame"] = per_frame
        out["host_fed"] = host_fed
        out["density_sweep"] = density
        out["fourseasons_frontend"] = fourseasons
        if validation is not None:
            out["split_validation"] = validation
        if world == 1 and args.cpu_pairs > 0:
            out["cpu_baseline"] = optional_leg("cpu_baseline", cpu_baseline, cfg, args.cpu_pairs, 5000)
        else:
            out["cpu_baseline"] = None
        import bench_legs
        out["degraded_legs"] = list(bench_legs.DEGRADED)   # optional legs the ENVIRONMENT cost (each holds an "error" field); library failures never get here
        print(json.dumps(out), flush=True)
    for e in all_ex + ([ex_rp] if ex_rp is not None else []):
        e.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
