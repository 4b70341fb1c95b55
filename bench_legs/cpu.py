"""cpu_baseline: the CPU oracle timed on a bounded sample of the same workload (kind = port), and the algorithmic pixel counts."""
import threading
import os
import sys
import time

import numpy as np

from . import ROOT, KITTI_MB, KITTI_MBF, self_check, oracle_module, tests_dir


def level_bytes(cfg):
    """Algorithmic bytes per image: sum of level pixels etc."""
    import math
    sc = np.float32(1.0)
    px = []
    for l in range(cfg["nlevels"]):
        inv = np.float32(1.0) / sc
        w = int(np.rint(np.float32(cfg["cols"]) * inv))
        h = int(np.rint(np.float32(cfg["rows"]) * inv))
        px.append(w * h)
        sc = np.float32(np.float64(sc) * np.float64(np.float32(cfg["scale"])))
    return px


def cpu_baseline(cfg, sample_pairs, seed0):
    """Oracle (CPU restatement, kind=port) on `sample_pairs` stereo pairs, 2 threads = one per eye like
    the reference (Frame.cc:122-125)."""
    orb_oracle = oracle_module()
    from msorb import synth
    exs = [orb_oracle.OracleExtractor(cfg["nfeatures"], cfg["scale"], cfg["nlevels"], cfg["ini_th"], cfg["min_th"])
           for _ in range(2)]
    uniq = [synth.stereo_pair(seed0 + i, cfg["rows"], cfg["cols"]) for i in range(min(sample_pairs, 16))]
    pairs = [uniq[i % len(uniq)] for i in range(sample_pairs)]   # the oracle recomputes every image: repeats cost the same
    counts = [0, 0]

    def eye(e):
        for p in pairs:
            _, kps, _ = exs[e](p[e])
            counts[e] += len(kps)

    t0 = time.perf_counter()
    th = [threading.Thread(target=eye, args=(e,)) for e in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    out = dict(value=round(sum(counts) / dt / 1e6, 5), unit="Mkeypoints/s", cores=2, kind="port",
               sample=f"{sample_pairs} KITTI-like stereo pairs, oracle/ (scalar C++ restatement, not OpenCV SIMD), "
                      f"2 threads (one per eye), {dt:.1f} s",
               host_cpus=os.cpu_count())
    # the same port scaled over the host's cores by frame-level parallelism: one extractor object per
    # thread, 2 images each — what an offline CPU pipeline could reach on this box
    nthr = max(2, min(os.cpu_count() or 2, 128))
    exs2 = [orb_oracle.OracleExtractor(cfg["nfeatures"], cfg["scale"], cfg["nlevels"], cfg["ini_th"], cfg["min_th"])
            for _ in range(nthr)]
    imgs = [pairs[i % len(pairs)][i % 2] for i in range(4)]
    tot = [0] * nthr

    def worker(i):
        for k in range(2):
            _, kps, _ = exs2[i](imgs[(i + k) % 4])
            tot[i] += len(kps)

    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(i,)) for i in range(nthr)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt2 = time.perf_counter() - t0
    out["all_cores"] = dict(value=round(sum(tot) / dt2 / 1e6, 4), unit="Mkeypoints/s", cores=nthr,
                            sample=f"{2 * nthr} images over {nthr} threads, {dt2:.1f} s")
    # BASELINE.json configs[0]: "EuRoC MH_01 stereo, CPU ORBextractor at 1000 features/frame (reference path, no GPU)" — the
    # same port on EuRoC-like 752x480 pairs, 2 threads (one per eye)
    ec = synth.EUROC
    exs3 = [orb_oracle.OracleExtractor(ec["nfeatures"], ec["scale"], ec["nlevels"], ec["ini_th"], ec["min_th"]) for _ in range(2)]
    ne = max(2, min(12, sample_pairs // 8))
    epairs = [synth.stereo_pair(seed0 + 500 + i, ec["rows"], ec["cols"]) for i in range(min(ne, 4))]
    ecount = [0, 0]

    def eeye(e):
        for i in range(ne):
            _, kps, _ = exs3[e](epairs[i % len(epairs)][e])
            ecount[e] += len(kps)

    t0 = time.perf_counter()
    th = [threading.Thread(target=eeye, args=(e,)) for e in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt3 = time.perf_counter() - t0
    out["configs0_euroc_1000"] = dict(value=round(sum(ecount) / dt3 / 1e6, 5), unit="Mkeypoints/s", cores=2, kind="port",
                                      ms_per_stereo_frame=round(dt3 / ne * 1e3, 2),
                                      sample=f"{ne} EuRoC-like 752x480 stereo pairs at 1000 features, 2 threads, {dt3:.1f} s")
    return out
