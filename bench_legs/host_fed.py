"""The PCIe-inclusive batch rate."""
import os
import sys
import time

import numpy as np

from . import ROOT, KITTI_MB, KITTI_MBF, self_check, oracle_module, tests_dir


def host_fed_leg(msorb, torch, exs, host_images, dev, cfg, pitch, steps=12):
    """The same extraction fed from pinned HOST memory: the upload of batch k+1 (one hipMemcpy2D-shaped copy into the 64-byte
    pitch planes, on a copy stream) runs while batch k is extracted.  What a pipeline that receives frames in host memory gets,
    next to `value` (inputs already in HBM)."""
    n = host_images.shape[0]
    pinned = torch.from_numpy(np.ascontiguousarray(host_images)).pin_memory()
    bufs = [torch.zeros((n, cfg["rows"], pitch), dtype=torch.uint8, device=dev) for _ in range(2)]
    views = [b[:, :, :cfg["cols"]] for b in bufs]
    cs = torch.cuda.Stream(device=dev)
    n_up = int(os.environ.get("MSORB_BENCH_UPLOAD_STREAMS", "2"))   # the batch as two slices on two copy streams (two DMA engines:
    # 51.7 instead of 46.4 GB/s; four streams: no more)
    css = [cs] + [torch.cuda.Stream(device=dev) for _ in range(n_up - 1)]
    outs = [None, None]
    for e in exs:
        e.set_overlap(1, True)

    def upload(k):
        for i, c in enumerate(css):
            a, b = n * i // n_up, n * (i + 1) // n_up
            with torch.cuda.stream(c):
                views[k][a:b].copy_(pinned[a:b], non_blocking=True)
        for c in css[1:]:
            cs.wait_stream(c)

    upload(0)
    cs.synchronize()
    t_up = time.perf_counter()
    upload(1)
    cs.synchronize()
    t_up = time.perf_counter() - t_up
    kp, t0 = 0, None
    for k in range(steps + 2):
        b = k & 1
        exs[b].extract_batch_submit(views[b], (0, 0), out=outs[b])
        outs[b] = exs[b]._pending[2]
        if k >= 1:
            counts, _, _, _ = exs[b ^ 1].extract_batch_wait()      # batch k-1 done: its buffer is free ...
            if k >= 3:
                kp += int(counts.sum())
            upload(b ^ 1)                                          # ... for the upload of batch k+1, under batch k's kernels
        if k == 1:
            cs.synchronize()
            t0 = time.perf_counter()                               # steady state from here: every step = one upload + one batch
        else:
            cs.synchronize()
    counts, _, _, _ = exs[(steps + 1) & 1].extract_batch_wait()
    kp += int(counts.sum())
    dt = time.perf_counter() - t0
    nbytes = pinned.numel()
    return {"what": "extract+describe with every batch uploaded from pinned host memory (pitched copy, two slices on two copy streams) while the previous one is extracted",
            "mkeypoints_per_s": round(kp / dt / 1e6, 2), "ms_per_step": round(dt / (steps + 0) * 1e3, 4),
            "upload_ms_per_batch": round(t_up * 1e3, 4), "upload_gbs": round(nbytes / t_up / 1e9, 2), "bytes_per_batch": int(nbytes),
            "bound": "PCIe (the upload of a batch takes longer than its kernels)"}
