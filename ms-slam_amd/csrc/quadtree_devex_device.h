// The device executor of the quadtree selection (quadtree_device.h's Ex interface on a workgroup): wave-level sorts, scans, the
// atomics of the generation passes.  A header because two translation units instantiate it: quadtree_kernels.hip with the workspace
// in LDS (QT_LDS = address space 3: ds_* instructions) and quadtree_global_kernels.hip with the workspace in GLOBAL memory (QT_LDS
// empty: quotas whose workspace no workgroup's LDS holds).  Include after quadtree_device.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "quadtree_device.h"

namespace msorb {

// inclusive prefix sum over the wave with DPP adds only (no LDS crossbar round trips)
__device__ __forceinline__ int wave_incl_scan_dpp(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112 /* row_shr:2 */, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114 /* row_shr:4 */, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118 /* row_shr:8 */, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142 /* row_bcast:15 */, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143 /* row_bcast:31 */, 0xc, 0xf, false);
    return v;
}

// std::sort of a range of at most 64 items ENTIRELY IN THE LANES of one wave: item i of the range lives in lane i, the whole
// recursion tree of libstdc++'s __introsort_loop is walked one LEVEL at a time — every open sub-range ("segment") of the level
// does its median-of-three and its __unguarded_partition at once, side by side in the same instructions — and
// __final_insertion_sort (a stable sort of every leaf, see finish_leaf) ends it.  No LDS round trip for the items, no hand-over
// of sub-ranges between waves, no barrier: a level is ~13 crossbar operations (ds_bpermute / ds_permute: they move registers
// between lanes through the LDS crossbar without touching LDS memory) instead of a median by lane 0 (0.4 us) + a partition
// through position lists in LDS (0.7 us) + a queue hand-over (0.5 us) per RANGE.  Same comparison outcomes, same swaps, same
// permutation as qt::lsort_acc (quadtree_device.h) — tests/test_quadtree_sort_gpu.py holds it to libstdc++'s std::sort.
// Wave collective: all 64 lanes call it in convergent code.  depth0 = the introsort depth budget left for this range.
__device__ __forceinline__ int lane_fetch(int v, int from_lane) { return __builtin_amdgcn_ds_bpermute(from_lane << 2, v); }
__device__ __forceinline__ int lane_send(int v, int to_lane) { return __builtin_amdgcn_ds_permute(to_lane << 2, v); }
__device__ __forceinline__ unsigned long long lanes_between(int a, int b) {   // bits a .. b-1, 0 <= a <= b <= 64
    const unsigned long long hi = b >= 64 ? ~0ull : ((1ull << b) - 1ull);
    return hi & ~((1ull << a) - 1ull);
}
__device__ inline void wave_introsort64(QT_LDS qt::SortItem* items, int first0, int last0, int depth0) {
    const int lane = threadIdx.x & 63;
    const int n = last0 - first0;
    if (n <= 1) return;
    const bool have = lane < n;
    uint32_t key = 0xFFFFFFFFu, node = 0;
    if (have) { const qt::SortItem it = items[first0 + lane]; key = it.key; node = it.node; }
    int sf = have ? 0 : lane, sl = have ? n : lane + 1;   // the lane's segment [sf, sl) in lane coordinates; lanes past the range: one of their own
    int depth = depth0;
    const unsigned long long below = (1ull << lane) - 1ull, above = lane == 63 ? 0ull : ~((2ull << lane) - 1ull);
    for (;;) {
        bool active = sl - sf > 16;                        // while (last - first > 16)
        if (__ballot(active) == 0) break;
        if (__ballot(active && depth == 0)) {
            // __partial_sort fallback of a segment whose depth budget is used up (never seen on these inputs; kept exact): through LDS,
            // serially, by the segment's first lane; its lanes then form finished one-item segments
            if (have) items[first0 + lane] = qt::SortItem{key, node};
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            if (active && depth == 0 && lane == sf) { qt::ArrayAcc a{items}; qt::heap_sort(a, first0 + sf, first0 + sl); }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            if (have) { const qt::SortItem it = items[first0 + lane]; key = it.key; node = it.node; }
            if (active && depth == 0) { sf = lane; sl = lane + 1; active = false; }
            if (__ballot(active) == 0) break;
        }
        --depth;
        // __move_median_to_first(first, first + 1, mid, last - 1)
        const int a = sf + 1, b = sf + ((sl - sf) >> 1), c = sl - 1;
        const uint32_t ka = (uint32_t)lane_fetch((int)key, a), kb = (uint32_t)lane_fetch((int)key, b), kc = (uint32_t)lane_fetch((int)key, c);
        int m;
        if (ka < kb) m = kb < kc ? b : (ka < kc ? c : a);
        else m = ka < kc ? a : (kb < kc ? c : b);
        const uint32_t pivot = m == a ? ka : (m == b ? kb : kc);
        int src = lane;
        if (active) src = lane == sf ? m : (lane == m ? sf : lane);
        key = (uint32_t)lane_fetch((int)key, src);
        node = (uint32_t)lane_fetch((int)node, src);
        // __unguarded_partition(first + 1, last, pivot = *first), as qt::partition_par counts it: G = positions with key >= pivot
        // ascending, L = positions with key <= pivot descending, swap (G[t], L[t]) while G[t] < L[t]
        const unsigned long long seg = lanes_between(sf, sl);
        const bool in = active && lane > sf;
        const bool is_g = in && key >= pivot, is_l = in && key <= pivot;
        const unsigned long long bg = __ballot(is_g) & seg, bl = __ballot(is_l) & seg;
        const int rg = __popcll(bg & below), NG = __popcll(bg), rl = __popcll(bl & above), NL = __popcll(bl);
        // slot t of a segment's lists lives in lane sf + 1 + t; lanes with nothing to send hit their segment's first lane (never read)
        const int G_t = lane_send(lane, is_g ? sf + 1 + rg : sf), L_t = lane_send(lane, is_l ? sf + 1 + rl : sf);
        const int t = lane - sf - 1, T = NG < NL ? NG : NL;
        const bool sw = in && t < T && G_t < L_t;
        const int k = __popcll(__ballot(sw) & seg);        // G[t] < L[t] is monotone in t: the count is the number of swaps
        const int g_k = lane_fetch(G_t, sf + 1 + k), l_k1 = lane_fetch(L_t, sf + k);
        int cut = sl;
        if (k < NG) cut = g_k;
        if (k > 0 && l_k1 < cut) cut = l_k1;
        const int part_g = lane_fetch(L_t, sf + 1 + rg), part_l = lane_fetch(G_t, sf + 1 + rl);
        src = lane;
        if (is_g && rg < k) src = part_g;
        else if (is_l && rl < k) src = part_l;
        key = (uint32_t)lane_fetch((int)key, src);
        node = (uint32_t)lane_fetch((int)node, src);
        if (active) { if (lane < cut) sl = cut; else sf = cut; }   // __introsort_loop(cut, last, depth); last = cut
    }
    // __final_insertion_sort: a stable sort of every segment (rank counting on the keys in lanes)
    int rank = 0;
    for (int j = 0; j < n; j++) {
        const uint32_t kj = (uint32_t)__builtin_amdgcn_readlane((int)key, j);
        rank += (j >= sf && j < sl) && ((kj < key) || (kj == key && j < lane));
    }
    const uint32_t key2 = (uint32_t)lane_send((int)key, sf + rank), node2 = (uint32_t)lane_send((int)node, sf + rank);
    if (have) items[first0 + lane] = qt::SortItem{key2, node2};
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
}

template <bool FRAME>
struct DevExT {
    // FRAME: a 1024-thread instance that has its CU to itself (single frames); otherwise a batch instance (256 / 512 threads, several
    // workgroups per CU, register budget of 128 with 24 resident candidates per thread: only the code it runs is compiled into it)
    static constexpr bool kSplitRank = FRAME;
    bool kLaneSort = true;   // ranges of <= 64 items finish in the lanes of one wave (wave_introsort64); false (MSORB_QT_LANE_SORT=0, an A/B switch): round 5's queue all the way down
    // std::sort restatement, data-parallel form (quadtree_device.h lsort_par), executed by wave 0 only: inside one
    // wave there is no s_barrier to pay and LDS operations complete in program order.
    struct WaveEx {
        __device__ int tid() const { return threadIdx.x & 63; }
        __device__ int nthreads() const { return 64; }
        __device__ void sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
        __device__ int excl_count(bool p, int* total) {
            const unsigned long long m = __ballot(p);
            *total = __popcll(m);
            return __popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull));
        }
        __device__ int excl_scan(int v, QT_LDS int*, int* total) {
            const int incl = wave_incl_scan_dpp(v);
            *total = __builtin_amdgcn_readlane(incl, 63);
            return incl - v;
        }
    };
    // Batches (256- and 512-thread instances, several workgroups per CU): the introsort loop as level-synchronous rounds —
    // waves without a range wait at the workgroup barrier, which costs the other workgroups of the CU nothing (the polling waves
    // of the barrier-free form below took issue slots from them: select stage 0.136 -> 0.186 ms per 256 images).
    // The introsort loop as level-synchronous rounds: the sub-ranges a partition leaves behind are independent, so every
    // round hands the current ranges (> 16 elements) to the workgroup's waves, one range per wave at a time; inside a wave
    // a partition is data-parallel (ballots, no s_barrier).  Which wave partitions which range, and in which order, cannot
    // change the result: ranges are disjoint and a partition only looks at its own range.  `stack` holds two range lists
    // of stack_ranges(m) entries (first, last, depth); ps.sc[0/1] their lengths.
    __device__ void sort_rounds(QT_LDS qt::SortItem* items, int n, QT_LDS int* stack, qt::ParScratch& ps) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = nt >> 6;
        int lg = 0;
        for (int t = n; t > 1; t >>= 1) lg++;
        if (threadIdx.x == 0) {
            stack[0] = 0; stack[1] = n; stack[2] = 2 * lg;
            ps.sc[0] = n > 16 ? 1 : 0;
            ps.sc[1] = 0;
        }
        __syncthreads();
        int which = 0;
        for (;;) {
            const int nr = ps.sc[which];
            if (nr == 0) break;
            QT_LDS int* cur = stack + which * ps.stack_half;
            QT_LDS int* nxt = stack + (which ^ 1) * ps.stack_half;
            WaveEx wex;
            for (int i = wave; i < nr; i += nwaves) {
                const int first = cur[3 * i], last = cur[3 * i + 1];
                int depth = cur[3 * i + 2];
                if (depth == 0) {  // __partial_sort fallback (:introsort depth limit)
                    if (lane == 0) { qt::ArrayAcc a{items}; qt::heap_sort(a, first, last); }
                    wex.sync();
                    continue;
                }
                --depth;
                if (lane == 0) {  // __move_median_to_first(first, first+1, mid, last-1)
                    qt::ArrayAcc acc{items};
                    const int a = first + 1, b = first + (last - first) / 2, c = last - 1;
                    const uint32_t ka = items[a].key, kb = items[b].key, kc = items[c].key;
                    if (ka < kb) {
                        if (kb < kc) qt::sort_swap(acc, first, b);
                        else if (ka < kc) qt::sort_swap(acc, first, c);
                        else qt::sort_swap(acc, first, a);
                    } else if (ka < kc) qt::sort_swap(acc, first, a);
                    else if (kb < kc) qt::sort_swap(acc, first, c);
                    else qt::sort_swap(acc, first, b);
                }
                wex.sync();
                qt::ParScratch pl = ps;  // this range's private stretch of the position lists
                pl.gpos = ps.gpos + first;
                pl.lpos = ps.lpos + first;
                const int cut = qt::partition_par(wex, items, first, last, pl);
                if (lane == 0) {
                    if (last - cut > 16) {
                        const int k = __hip_atomic_fetch_add(&ps.sc[which ^ 1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        nxt[3 * k] = cut; nxt[3 * k + 1] = last; nxt[3 * k + 2] = depth;
                    }
                    if (cut - first > 16) {
                        const int k = __hip_atomic_fetch_add(&ps.sc[which ^ 1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        nxt[3 * k] = first; nxt[3 * k + 1] = cut; nxt[3 * k + 2] = depth;
                    }
                }
            }
            __syncthreads();
            if (threadIdx.x == 0) ps.sc[which] = 0;
            which ^= 1;
            __syncthreads();
        }
        qt::final_stable_sort(*this, items, n, ps);  // rank counting: all threads
    }
    // Single frames (1024-thread instances, the workgroup has its CU to itself): the introsort loop without workgroup barriers.  The sub-ranges a partition leaves behind are independent (disjoint, and a
    // partition only looks at its own range), so which wave partitions which range, and when, cannot change the result.  A wave
    // that has partitioned a range keeps the left part and goes on with it (depth first); the right part goes into a ring of open
    // ranges in LDS that idle waves poll.  `pending` counts the chains that are still running or queued: the sort is over when it
    // reaches zero.  Rounds 2-4 ran the same partitions as level-synchronous rounds with two __syncthreads each: 22 us for the
    // ~150 nodes of a KITTI level-0 careful sweep, almost all of it barrier and hand-over latency; the critical path is now the
    // depth of the recursion (3-4 partitions).  Inside a wave a partition is data-parallel (ballots, no s_barrier).
    // `stack`: ring of 2 * stack_ranges(m) entries (first, last, depth) — at any time the open ranges are disjoint and longer
    // than 16 elements, i.e. fewer than the ring holds; ps.sc[0] = head (next to take), ps.sc[1] = tail (published entries),
    // ps.sc[2] = pending, ps.sc[3] = reserved entries (>= tail: an entry is written, then published in reservation order).
    __device__ void sort(QT_LDS qt::SortItem* items, int n, QT_LDS int* stack, qt::ParScratch& ps) {
        if constexpr (FRAME) sort_queue(items, n, stack, ps);
        else sort_rounds(items, n, stack, ps);
    }
    // __final_insertion_sort, leaf by leaf: the introsort loop leaves ranges of at most 16 elements unsorted, and every cut it made
    // separates keys <= pivot from keys >= pivot — the stable insertion sort that libstdc++ runs over the whole array afterwards never
    // moves an element across a cut, i.e. it is a stable sort of each leaf on its own.  The wave that ends up with a leaf sorts it at
    // once (rank counting on keys held in lanes: no LDS traffic, no pass over the array at the end, no workgroup barrier).
    __device__ void finish_leaf(QT_LDS qt::SortItem* items, int first, int last) {   // wave collective; last - first <= 64
        const int lane = threadIdx.x & 63, len = last - first;
        if (len <= 1) return;
        qt::SortItem it{0xFFFFFFFFu, 0u};
        if (lane < len) it = items[first + lane];
        int rank = 0;
        for (int j = 0; j < len; j++) {
            const uint32_t kj = (uint32_t)__builtin_amdgcn_readlane((int)it.key, j);
            rank += (kj < it.key) || (kj == it.key && j < lane);
        }
        if (lane < len) items[first + rank] = it;
        WaveEx wex;
        wex.sync();
    }
    __device__ void sort_queue(QT_LDS qt::SortItem* items, int n, QT_LDS int* stack, qt::ParScratch& ps) {
        mark(20);
        const int lane = threadIdx.x & 63;
        const int ring = 2 * ps.stack_half / 3;
        int lg = 0;
        for (int t = n; t > 1; t >>= 1) lg++;
        if (kLaneSort && n <= 64) {   // the whole sort in the lanes of wave 0: no queue, no ring, one barrier
            if (threadIdx.x < 64) wave_introsort64(items, 0, n, 2 * lg);
            __syncthreads();
            mark(21);
            mark(22);
            return;
        }
        if (threadIdx.x == 0) {
            stack[0] = 0; stack[1] = n; stack[2] = 2 * lg;
            ps.sc[0] = 0;
            ps.sc[1] = n > 16 ? 1 : 0;
            ps.sc[2] = n > 16 ? 1 : 0;
            ps.sc[3] = n > 16 ? 1 : 0;
        }
        __syncthreads();
        WaveEx wex;
        for (;;) {
            // Take an open range.  The polling loop is executed by the WHOLE wave (every lane reads the same LDS words, so the
            // loop's control flow is wave-uniform); only the claim itself is lane 0's.  (A first form ran the loop inside
            // `if (lane == 0)` with breaks out of it: hipcc then kept the code after the loop — readfirstlane, the partition with its
            // ballots and DPP scans — under lane 0's execution mask, and every partition saw one element.)
            int first = 0, last = 0, depth = -1;
            for (;;) {
                if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&ps.sc[2], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) == 0) break;   // nothing running, nothing queued
                const int hd = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&ps.sc[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
                const int tl = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&ps.sc[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (hd < tl) {
                    QT_LDS const int* e = stack + 3 * (hd % ring);
                    const int f = e[0], l = e[1], d = e[2];   // read before the claim: an unclaimed entry is never overwritten
                    int won = 0;
                    if (lane == 0) {
                        int expect = hd;
                        won = __hip_atomic_compare_exchange_strong(&ps.sc[0], &expect, hd + 1, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ? 1 : 0;
                    }
                    if (__builtin_amdgcn_readfirstlane(won)) {
                        first = __builtin_amdgcn_readfirstlane(f); last = __builtin_amdgcn_readfirstlane(l); depth = __builtin_amdgcn_readfirstlane(d);
                        break;
                    }
                } else {
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            if (depth < 0) break;
            mark(30);
            bool sorted = false;
            while (last - first > 16) {   // __introsort_loop on [first, last)
                if (kLaneSort && last - first <= 64) {   // the rest of this range's recursion, and its leaves, in the wave's lanes
                    wave_introsort64(items, first, last, depth);
                    sorted = true;
                    break;
                }
                if (depth == 0) {  // __partial_sort fallback (the introsort depth limit)
                    if (lane == 0) { qt::ArrayAcc a{items}; qt::heap_sort(a, first, last); }
                    wex.sync();
                    sorted = true;
                    break;
                }
                --depth;
                if (lane == 0) {  // __move_median_to_first(first, first+1, mid, last-1)
                    qt::ArrayAcc acc{items};
                    const int a = first + 1, b = first + (last - first) / 2, c = last - 1;
                    const uint32_t ka = items[a].key, kb = items[b].key, kc = items[c].key;
                    if (ka < kb) {
                        if (kb < kc) qt::sort_swap(acc, first, b);
                        else if (ka < kc) qt::sort_swap(acc, first, c);
                        else qt::sort_swap(acc, first, a);
                    } else if (ka < kc) qt::sort_swap(acc, first, a);
                    else if (kb < kc) qt::sort_swap(acc, first, c);
                    else qt::sort_swap(acc, first, b);
                }
                wex.sync();
                mark(31);
                qt::ParScratch pl = ps;  // this range's private stretch of the position lists
                pl.gpos = ps.gpos + first;
                pl.lpos = ps.lpos + first;
                const int cut = qt::partition_par(wex, items, first, last, pl);
                mark(32);
                if (last - cut > 16) {   // [cut, last) becomes an open range: the entry first, then the tail that publishes it
                    int slot = 0;
                    if (lane == 0) {
                        (void)__hip_atomic_fetch_add(&ps.sc[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        slot = __hip_atomic_fetch_add(&ps.sc[3], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // several waves may push at once: reserve a slot
                        QT_LDS int* e = stack + 3 * (slot % ring);
                        e[0] = cut; e[1] = last; e[2] = depth;
                    }
                    slot = __builtin_amdgcn_readfirstlane(slot);
                    // publish in reservation order: wait (whole wave, uniform loop) until every earlier reservation has been published
                    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&ps.sc[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) != slot) __builtin_amdgcn_s_sleep(0);
                    if (lane == 0) __hip_atomic_store(&ps.sc[1], slot + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else {
                    finish_leaf(items, cut, last);
                }
                last = cut;
                mark(33);
            }
            if (!sorted) finish_leaf(items, first, last);
            if (lane == 0) (void)__hip_atomic_fetch_sub(&ps.sc[2], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);   // this chain has ended
        }
        __syncthreads();
        mark(21);
        if (n <= 16) {   // no range was ever opened: the whole array is one leaf
            if (threadIdx.x < 64) finish_leaf(items, 0, n);
            __syncthreads();
        }
        mark(22);
    }
    int dbg = 0;
    int nt = 0;  // threads of this instance: blockDim.x, or fewer for the small levels of a mixed launch (the other waves have left)
#ifndef MSORB_QT_MARK_Y
#define MSORB_QT_MARK_Y 0   // the level whose instance is timed (marks build only)
#endif
#ifdef MSORB_QT_MARKS  // per-phase timestamps of instance (0, MSORB_QT_MARK_Y) (build with -DMSORB_QT_MARKS, run with MSORB_QT_DEBUG=3):
    int n_marks = 0;   // compiled out by default, the arrays would cost every wave 600 bytes of scratch
    long long t_mark[96];
    int id_mark[96];
    __device__ void mark(int id) {
        if (dbg == 3 && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == MSORB_QT_MARK_Y && n_marks < 96) {
            t_mark[n_marks] = wall_clock64(); id_mark[n_marks] = id; n_marks++;
        }
    }
    __device__ void dump() {
        if (dbg == 3 && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == MSORB_QT_MARK_Y)
            for (int i = 1; i < n_marks; i++) printf("mark %d dt_us=%.2f\n", id_mark[i], (double)(t_mark[i] - t_mark[i - 1]) * 0.01);
    }
#else
    __device__ void mark(int) {}
    __device__ void dump() {}
#endif
    __device__ int tid() const { return threadIdx.x; }
    __device__ int nthreads() const { return nt; }
    __device__ void sync() { __syncthreads(); }
    __device__ int atomic_add(QT_LDS int* p, int v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    // Wave collectives (every lane of the wave calls them, in convergent code).
    // add_runs: arr[idx] += 1 for every lane with idx >= 0, one atomic per RUN of equal indices in lane order: neighbouring lanes
    // hold neighbouring candidates, i.e. mostly the same node — 64 atomics on one LDS address are executed one after the other.
    __device__ void add_runs(QT_LDS int* arr, int idx) {
        const int lane = threadIdx.x & 63;
        const int prev = __builtin_amdgcn_update_dpp(idx, idx, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
        const bool head = lane == 0 || prev != idx;
        const uint64_t heads = __ballot(head);
        if (head && idx >= 0) {
            const uint64_t rest = (heads >> lane) >> 1;
            const int len = rest ? __builtin_ctzll(rest) + 1 : 64 - lane;
            (void)__hip_atomic_fetch_add(arr + idx, len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    // the path tables' 16-bit counters, two to a word (LDS has no 16-bit atomics): idx counts entries from the 4-byte aligned base
    __device__ void add16(QT_LDS uint16_t* arr, int idx) {
        (void)__hip_atomic_fetch_add((QT_LDS int*)arr + (idx >> 1), 1 << ((idx & 1) * 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // *p += sum of v over the wave (one atomic per wave that has anything to add)
    __device__ void wave_sum_add(QT_LDS int* p, int v, int) {
        const int incl = wave_incl_scan_dpp(v);
        if ((threadIdx.x & 63) == 63 && incl != 0) (void)__hip_atomic_fetch_add(p, incl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // claim: *ctr += (number of lanes with pred); returns a distinct value of the claimed range to every lane with pred
    __device__ int claim(QT_LDS int* ctr, bool pred) {
        const uint64_t m = __ballot(pred);
        if (m == 0) return 0;                      // wave-uniform
        const int lane = threadIdx.x & 63;
        const int leader = __builtin_ctzll(m);
        int base = 0;
        if (lane == leader) base = __hip_atomic_fetch_add(ctr, __popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        base = __builtin_amdgcn_readlane(base, leader);
        return base + __popcll(m & ((1ull << lane) - 1ull));
    }
    __device__ void atomic_max(QT_LDS int* p, int v) { (void)__hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    __device__ void atomic_or(QT_LDS int* p, int v) { (void)__hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    __device__ void atomic_min(QT_LDS int* p, int v) { (void)__hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    __device__ int excl_count(bool p, int* total) { int t = 0; const int r = excl_scan((int)p, nullptr, &t); *total = t; return r; }  // unused
    // block-wide exclusive prefix of v over threads (<= 16 waves); tmp = 16 ints of LDS
    __device__ int excl_scan(int v, QT_LDS int* tmp, int* total) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int incl = wave_incl_scan_dpp(v);
        __syncthreads();  // tmp may still be read by a previous scan
        if (lane == 63) tmp[wave] = incl;
        __syncthreads();
        int before = 0, tot = 0;
        const int nw = nt >> 6;
        for (int w = 0; w < nw; w++) {
            const int c = tmp[w];
            if (w < wave) before += c;
            tot += c;
        }
        *total = tot;
        return before + incl - v;
    }
};

}  // namespace msorb
