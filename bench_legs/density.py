"""density_sweep — is the headline an artefact of one synthetic recipe?  The same extraction loop on three input classes of
msorb/synth.py (low-texture: a third of the cells take the minThFAST retry of ORBextractor.cc:843-847; the default; cluttered),
outside the timed region; and the front-end half of configs[4] at its own geometry (4Seasons 800 x 400,
Examples/Stereo-Inertial/4season.yaml:62-63): batch rate + per-frame calls."""
import time

import numpy as np

from . import KITTI_MB, KITTI_MBF, self_check, oracle_module


def resident(torch, dev, host_imgs):
    """device-resident batch with a 64-byte row pitch (as bench.py's main loop holds its input)"""
    n, rows, cols = host_imgs.shape
    pitch = (cols + 63) // 64 * 64
    st = torch.zeros((n, rows, pitch), dtype=torch.uint8, device=dev)
    v = st[:, :, :cols]
    v.copy_(torch.from_numpy(np.ascontiguousarray(host_imgs)).to(dev))
    return v


def pipelined_rate(torch, exs, images, steps=200, warmup=20):
    """bench.py's N = 1 loop in small: two batches in flight on two handles (msorb_extract_batch_submit / _wait), the second
    one half a step behind the first.  -> (keypoints per second, ms per step, keypoints per image)."""
    outs = [None, None]
    for e in exs:
        e.set_overlap(1, True)
        e.set_profiling(False)

    def run(n, stagger):
        inflight, kp = [], 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(n):
            b = k & 1
            if len(inflight) == 2:
                counts, _, _, _ = inflight.pop(0).extract_batch_wait()
                kp += int(counts.sum())
            exs[b].extract_batch_submit(images, (0, 0), out=outs[b])
            outs[b] = exs[b]._pending[2]
            inflight.append(exs[b])
            if k == 0 and stagger:
                time.sleep(stagger)
        while inflight:
            counts, _, _, _ = inflight.pop(0).extract_batch_wait()
            kp += int(counts.sum())
        torch.cuda.synchronize()
        return kp, time.perf_counter() - t0

    _, dtw = run(warmup, 0.0)
    # (half a step; bounded to the band in which the offset was measured not to matter — a slow-starting host makes the warm-up's
    # wall time read two to three times too long, and an offset of more than a step is lock-step again: bench.py has the numbers)
    stagger = min(max(dtw / warmup / 2, 400e-6), 800e-6)
    kp, dt = run(steps, stagger)
    return kp / dt, dt / steps * 1e3, kp / steps / images.shape[0]


def stage_ms(ex, images, steps=12, discard=6):
    """per-stage kernel time of the synchronous call, every kernel alone on the GPU (as stage_ms_per_step of the main line)"""
    ex.set_overlap(1, False)
    ex.set_profiling(True)
    acc = {}
    for i in range(steps + discard):
        ex.extract_batch(images, (0, 0))
        if i >= discard:
            for k, v in ex.stage_ms().items():
                acc.setdefault(k, []).append(v)
    ex.set_profiling(False)
    return {k: float(np.median(v)) for k, v in acc.items()}


def input_statistics(cfg, left):
    """corner density of one level-0 image at iniThFAST, share of the reference's cells that take the minThFAST retry (all
    levels), FAST candidates handed to the quadtree — from the CPU oracle, once per class."""
    orb_oracle = oracle_module()
    ex = orb_oracle.OracleExtractor(cfg["nfeatures"], cfg["scale"], cfg["nlevels"], cfg["ini_th"], cfg["min_th"])
    _, kps, desc = ex(left)
    st = ex.cell_stats()
    corner, _ = orb_oracle.fast9_planes(left, cfg["ini_th"])
    total = sum(s[0] for s in st)
    return dict(corners_per_px=round(float(corner.mean()), 5), minth_cells_frac=round(sum(s[1] for s in st) / total, 4),
                empty_cells_frac=round(sum(s[2] for s in st) / total, 4),
                candidates_per_image=int(sum(len(ex.candidates(l)) for l in range(cfg["nlevels"])))), kps, desc


def density_sweep_leg(msorb, synth, torch, make_ex, cfg, dev, pairs, uniq, cpu, handles=None):
    """-> the `density_sweep` list of the bench line.
    handles: the two extractor handles of the main loop.  The sweep runs on THEM: a process that has created a dozen handles (every
    leg of bench.py makes its own) has more HIP streams than the runtime has hardware queues, the streams of two new handles share
    queues, and their two batches in flight serialise — the same loop on fresh handles read 411 Mkeypoints/s on the default class
    inside bench.py and 472-483 alone in a process (the main loop: 469)."""
    own = handles is None
    exs = [make_ex(), make_ex()] if own else list(handles)
    out = []
    try:
        for tex in ("low", "default", "high"):
            base = synth.stereo_batch(uniq, cfg["rows"], cfg["cols"], seed0=4000, texture=tex)
            host = np.concatenate([base] * (pairs // uniq + 1))[:2 * pairs]
            images = resident(torch, dev, host)
            rate, ms_step, kp_img = pipelined_rate(torch, exs, images)
            st = stage_ms(exs[0], images)
            row = {"class": tex, "value": round(rate / 1e6, 2), "unit": "Mkeypoints/s", "ms_per_step": round(ms_step, 4),
                   "keypoints_per_image": round(kp_img, 1), "fast_ms": round(st["fast"], 4), "quadtree_ms": round(st["select"], 4),
                   "pyramid_ms": round(st["pyramid"], 4), "blur_ms": round(st["blur"], 4), "describe_ms": round(st["describe"], 4)}
            if cpu:
                stats, okps, odesc = input_statistics(cfg, base[0])
                row.update(stats)
                # parity on the class: the first image through the batch kernels against the oracle
                counts, _, d_kps, d_desc = exs[0].extract_batch(images, (0, 0))
                n0 = int(counts[0])
                same = n0 == len(okps) and np.array_equal(d_kps[0, :n0].cpu().numpy(), okps.view(np.uint8).reshape(-1, 28)) and \
                    np.array_equal(d_desc[0, :n0].cpu().numpy(), odesc)
                self_check(same, f"density_sweep: the batch kernels differ from the CPU oracle on the '{tex}' class")
                row["gpu_matches_cpu"] = True
            out.append(row)
            del images
    finally:
        for e in exs:
            if own:
                e.close()
            else:
                e.set_profiling(False)
                e.set_overlap(1, True)
    return out


def fourseasons_leg(msorb, synth, torch, dev, pairs, uniq, local, cpu, handles=None):
    """configs[4]'s front-end half at the 4Seasons geometry: batch extraction rate (inputs in HBM) and the per-frame calls.
    handles: as density_sweep_leg (the parameters of 4season.yaml are KITTI's: a handle takes the new image size on its next call)."""
    from .per_frame import per_frame_leg
    cfg = synth.FOURSEASONS
    own = handles is None
    exs = [msorb.ORBextractor(cfg["nfeatures"], cfg["scale"], cfg["nlevels"], cfg["ini_th"], cfg["min_th"], device=local) for _ in range(2)] if own else list(handles)
    try:
        base = synth.stereo_batch(uniq, cfg["rows"], cfg["cols"], seed0=4400)
        host = np.concatenate([base] * (pairs // uniq + 1))[:2 * pairs]
        images = resident(torch, dev, host)
        rate, ms_step, kp_img = pipelined_rate(torch, exs, images)
        st = stage_ms(exs[0], images)
        pf = per_frame_leg(msorb, exs[0], base[0], base[1])
        res = {"what": "configs[4], front-end half: extract+describe at the 4Seasons geometry (800x400, 2000 features, "
                       "Examples/Stereo-Inertial/4season.yaml:62-63) — batch of %d stereo pairs resident in HBM, two batches in flight, and "
                       "one frame at a time through the C ABI; the window's constraint matrix is `sparsification`" % pairs,
               "value": round(rate / 1e6, 2), "unit": "Mkeypoints/s", "ms_per_step": round(ms_step, 4), "pairs_per_step": pairs,
               "keypoints_per_image": round(kp_img, 1),
               "stage_ms_per_step": {k: round(v, 4) for k, v in st.items()},
               "per_frame": {k: pf[k] for k in ("ms_one_image", "ms_stereo_frame_one_call", "keypoints_stereo_frame")}}
        if cpu:
            orb_oracle = oracle_module()
            orc = orb_oracle.OracleExtractor(cfg["nfeatures"], cfg["scale"], cfg["nlevels"], cfg["ini_th"], cfg["min_th"])
            t0 = time.perf_counter()
            _, okps, odesc = orc(base[0])
            dt = time.perf_counter() - t0
            counts, _, d_kps, d_desc = exs[0].extract_batch(images, (0, 0))
            n0 = int(counts[0])
            same = n0 == len(okps) and np.array_equal(d_kps[0, :n0].cpu().numpy(), okps.view(np.uint8).reshape(-1, 28)) and \
                np.array_equal(d_desc[0, :n0].cpu().numpy(), odesc)
            self_check(same, "fourseasons: the batch kernels differ from the CPU oracle at 800x400")
            res["cpu_baseline"] = {"ms_per_image": round(dt * 1e3, 2), "cores": 1, "kind": "port", "gpu_matches_cpu": True}
        return res
    finally:
        for e in exs:
            if own:
                e.close()
            else:
                e.set_profiling(False)
                e.set_overlap(1, True)
