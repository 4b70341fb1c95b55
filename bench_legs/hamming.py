"""Second half of BASELINE.json's metric: Gpairs/s of the brute-force Hamming match (left-eye descriptors of every pair against
the right-eye descriptors of the same pair, dense top-2) on the descriptors just extracted — both formulations — and its CPU leg."""
import os
import time

import numpy as np

from . import self_check, oracle_module


def hamming_leg(msorb, torch, d_desc, counts_h, dev, local, reps=60):
    """-> (the `hamming_match` object, a private context for hamming_cpu_leg)."""
    dq = d_desc[0::2].contiguous()
    dtr = d_desc[1::2].contiguous()
    nq = torch.from_numpy(np.ascontiguousarray(counts_h[0::2])).to(dev)
    nt = torch.from_numpy(np.ascontiguousarray(counts_h[1::2])).to(dev)
    pairs = int((counts_h[0::2].astype(np.int64) * counts_h[1::2].astype(np.int64)).sum())
    # The metric's kernel is the north_star's formulation: dense_top2_kernel<2, 4>, v_xor + accumulating v_bcnt per dword and a
    # per-lane top-2, no MFMA (MSORB_DENSE_POPCOUNT, the library default).  Ceiling: 8 x (v_xor + v_bcnt) per 64 pairs at 7.91
    # cycles@2.4GHz per instruction PAIR (tools/valu_ubench2.hip, round 4: in a stream that mixes the two classes a fast-class
    # instruction costs as much as a slow one, whether alternating or in runs of 16 — 2.5 cycles hold in pure fast-class streams
    # only), + 3 slow-class instructions of top-2 bookkeeping at 4.2.
    ceil_valu = 1024 * 2.4e9 * 64 / (8 * 7.91) / 1e9
    ceil_valu_top2 = 1024 * 2.4e9 * 64 / (8 * 7.91 + 3 * 4.2) / 1e9
    msorb.hamming_dense_top2_batch(dq, dtr, nq, nt, repeats=2, device=local, formulation=msorb.DENSE_POPCOUNT)   # warm-up
    bi_v, bd_v, sd_v, ms_v = msorb.hamming_dense_top2_batch(dq, dtr, nq, nt, repeats=reps, device=local,
                                                            formulation=msorb.DENSE_POPCOUNT)
    g_v = pairs * reps / (ms_v * 1e-3) / 1e9
    # The opt-in matrix-core variant (matcher.hip dense_top2_mfma_kernel: +-32 int8 encoding, 8 x v_mfma_i32_32x32x32_i8 per 32 x 32
    # pairs = 512 int8 operations per pair), reported beside it, never as the metric: north_star rules MFMA out for this path.
    # Its ceiling = the i8 MFMA rate this chip sustains with nothing else running (4.3 POPS: 37.5 cycles@2.4GHz per instruction
    # and SIMD, tools/mfma_rate.hip; docs/DESIGN_rounds1-3.md section 4).
    mfma_pops = 4.3e15
    ceil_mfma = mfma_pops / 512 / 1e9
    msorb.hamming_dense_top2_batch(dq, dtr, nq, nt, repeats=2, device=local, formulation=msorb.DENSE_MATRIX_CORES)
    bi_m, bd_m, sd_m, ms = msorb.hamming_dense_top2_batch(dq, dtr, nq, nt, repeats=reps, device=local, formulation=msorb.DENSE_MATRIX_CORES)
    g = pairs * reps / (ms * 1e-3) / 1e9
    # rows >= nq[f] of a frame are never written by either kernel: only the valid rows are results
    live = torch.arange(dq.shape[1], device=dev)[None, :] < nq[:, None]
    same_kernels = bool(torch.equal(bi_v[live], bi_m[live]) and torch.equal(bd_v[live], bd_m[live]) and
                        torch.equal(sd_v[live], sd_m[live]))
    self_check(same_kernels, "hamming_match: the popcount and the MFMA kernel disagree on a valid row")
    hamming = {"gpairs_per_s": round(g_v, 2), "pairs_per_launch": pairs, "ms_per_launch": round(ms_v / reps, 4),
               "kernel": "dense_top2_kernel<2, 4> (v_xor_b32 + accumulating v_bcnt_u32_b32 per dword, per-lane top-2; no MFMA)",
               "formulation": "MSORB_DENSE_POPCOUNT: the north_star's formulation and the library default",
               "ceiling_gpairs_per_s": round(ceil_valu_top2, 1), "frac": round(g_v / ceil_valu_top2, 3),
               "ceiling_note": "mixed-stream VALU issue rate measured on this chip (tools/valu_ubench2.hip: v_xor + v_bcnt = 7.91 "
                               "cycles@2.4GHz per pair of instructions) + 3 top-2 instructions per 64 pairs; PMC of this kernel in "
                               "profiles/round4_dense_popcount_pmc.txt",
               "ceiling_without_top2_gpairs_per_s": round(ceil_valu, 1),
               "bound": "integer VALU issue (xor + popcount + top-2); not HBM: (Q+T)*32 B per frame are reused Q*T times",
               "matrix_core_variant": {"what": "opt-in MSORB_DENSE_MATRIX_CORES: Hamming distance as an int8 dot product on the matrix "
                                               "cores (dense_top2_mfma_kernel, v_mfma_i32_32x32x32_i8) — exact, identical results, "
                                               "outside north_star's 'no MFMA' rule, hence not the metric",
                                       "gpairs_per_s": round(g, 2), "ms_per_launch": round(ms / reps, 4),
                                       "ceiling_gpairs_per_s": round(ceil_mfma, 1), "frac": round(g / ceil_mfma, 3),
                                       "ceiling_note": "i8 MFMA rate measured on this chip (4.3 POPS) / 512 operations per pair",
                                       "identical_results": same_kernels}}
    ctx = dict(dq=dq, dtr=dtr, nq=nq, nt=nt, pairs=pairs, popcount_out=(bi_v, bd_v, sd_v), mfma_out=(bi_m, bd_m, sd_m), counts_h=counts_h, local=local)
    return hamming, ctx


def hamming_cpu_leg(msorb, hamming, ctx):
    """CPU leg of the matcher on a bounded sample: ORBmatcher::DescriptorDistance brute force (oracle, 1 thread) on the first
    stereo pair's descriptors; its result also cross-checks the GPU's indices and distances of BOTH kernels."""
    orb_oracle = oracle_module()
    dq, dtr, nq, nt, counts_h, pairs = ctx["dq"], ctx["dtr"], ctx["nq"], ctx["nt"], ctx["counts_h"], ctx["pairs"]
    bi_g, bd_g, sd_g = ctx["mfma_out"]
    n0, n1 = int(counts_h[0]), int(counts_h[1])
    q0, t0_ = dq[0, :n0].cpu().numpy(), dtr[0, :n1].cpu().numpy()
    bi_c, bd_c, sd_c = orb_oracle.dense_top2(q0, t0_)
    # the same brute force over many frames: one core (16 frames) and every host core (all frames, one frame per task)
    from concurrent.futures import ThreadPoolExecutor
    dq_h, dt_h = dq.cpu().numpy(), dtr.cpu().numpy()
    frames16 = list(range(min(16, dq_h.shape[0])))
    tc1 = time.perf_counter()
    for f_ in frames16:
        orb_oracle.dense_top2(dq_h[f_, :int(counts_h[2 * f_])], dt_h[f_, :int(counts_h[2 * f_ + 1])])
    dt1 = time.perf_counter() - tc1
    pairs1 = sum(int(counts_h[2 * f_]) * int(counts_h[2 * f_ + 1]) for f_ in frames16)
    ncore = min(os.cpu_count() or 1, dq_h.shape[0])
    tca = time.perf_counter()
    with ThreadPoolExecutor(ncore) as pool:
        list(pool.map(lambda f_: orb_oracle.dense_top2(dq_h[f_, :int(counts_h[2 * f_])], dt_h[f_, :int(counts_h[2 * f_ + 1])]),
                      range(dq_h.shape[0])))
    dta = time.perf_counter() - tca
    same = (np.array_equal(bi_c, bi_g[0, :n0].cpu().numpy()) and np.array_equal(bd_c, bd_g[0, :n0].cpu().numpy()) and
            np.array_equal(sd_c, sd_g[0, :n0].cpu().numpy()))
    same_pop = all(np.array_equal(c, g[0, :n0].cpu().numpy()) for c, g in zip((bi_c, bd_c, sd_c), ctx["popcount_out"]))
    self_check(same, "hamming_match: the MFMA kernel differs from the CPU oracle")
    self_check(same_pop, "hamming_match: the popcount kernel differs from the CPU oracle")
    hamming["cpu_baseline"] = {"gpairs_per_s": round(pairs1 / dt1 / 1e9, 4), "cores": 1, "kind": "port",
                               "sample": f"{len(frames16)} stereo pairs of ~{n0} x {n1} descriptors, {dt1 * 1e3:.1f} ms "
                                         "(xor + __builtin_popcountll, -O3 x86-64-v3)",
                               "all_cores": {"gpairs_per_s": round(pairs / dta / 1e9, 3), "cores": ncore,
                                             "sample": f"all {dq_h.shape[0]} pairs, one frame per task, {dta * 1e3:.1f} ms"},
                               "gpu_matches_cpu": bool(same_pop), "matrix_core_variant_matches_cpu": bool(same)}
