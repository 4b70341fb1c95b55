// Device-side keypoint selection: one workgroup per (pyramid level, image) runs the generation-synchronous
// DistributeOctTree of quadtree_device.h on the compacted FAST candidates, then one workgroup per image lays
// out the selected keypoints in ORBextractor::operator()'s output order (ORBextractor.cc:1122-1163).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <string>

#include "lds_limit.h"
#include "orb_device.h"
#include "quadtree_device.h"
#include "quadtree_paths_device.h"
#include "gauss7_stream_device.h"
#include "stereo_rowtable_device.h"

namespace msorb {

constexpr int kQtThreads = 256;  // 256 beats 512 (0.39 vs 0.47 ms per 256 images): fewer idle waves waiting at barriers
// register-resident candidates per thread: 0 for batches (with 16 the VGPR pressure halved the resident workgroups and
// measured slower); 8 for single frames, where a 1024-thread instance then holds a whole level (<= 8192 candidates) in
// registers and the point passes stop waiting on global memory
constexpr int kQtPointsPerThreadFrame = 8;
#ifndef QT_BATCH_PC
#define QT_BATCH_PC 24
#endif
constexpr int kQtPointsPerThreadBatch = QT_BATCH_PC;   // x 256 threads = 6144 candidates of a level in registers, the rest through global memory (select stage per 256 images: 12-20: 0.142-0.143, 24: 0.139-0.140, 28: 0.145, 32: 0.167 ms — spills)

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
__device__ void wave_introsort64(QT_LDS qt::SortItem* items, int first0, int last0, int depth0) {
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

#ifndef MSORB_QT_PATH_MIN_N_FACTOR
#define MSORB_QT_PATH_MIN_N_FACTOR 2
#endif
template <int PC, bool FRAME>
__device__ __forceinline__ void quadtree_select_body(const QtLevels& lv, const Cand16* __restrict__ compact,
                                                     const int* __restrict__ img_base,
                                                     const int* __restrict__ level_count, uint16_t* __restrict__ label,
                                                     int* __restrict__ sel_pt /* [img][sel_stride] candidate idx */,
                                                     int* __restrict__ sel_n /* [img][nlevels] */, int sel_stride,
                                                     int ws_N, int ws_nini, int debug, int big_levels, int small_nt, int path_cap) {
    extern __shared__ __attribute__((aligned(16))) char qt_mem[];
    // grid = (image, level): consecutive workgroups (dealt round-robin to the 8 XCDs) are different images of one
    // level, so the heavy level-0 instances are spread over all XCDs instead of piling up on one
    const int level = blockIdx.y, img = blockIdx.x;
    const int* lc = level_count + (size_t)img * lv.nlevels;
    int off = img_base[img];
    for (int l = 0; l < level; l++) off += lc[l];
    const int n = lc[level];
    // mixed launch: the levels with many candidates get every wave of the workgroup, the others only the first small_nt threads
    // (their remaining waves leave at once; s_barrier counts the waves that are still there)
    const int nt_eff = level < big_levels ? (int)blockDim.x : min((int)blockDim.x, small_nt);
    if ((int)threadIdx.x >= nt_eff) return;
    qt::Workspace w;
    qt::workspace_carve(w, qt_mem, ws_N, ws_nini);
    DevExT<FRAME> ex;
    ex.kLaneSort = !(debug & 0x100);
    debug &= 0xff;
    ex.dbg = debug;
    ex.nt = nt_eff;
    int* out = sel_pt + (size_t)img * sel_stride + lv.sel_off[level];
    int kept = -1;
    // selection by quadrant path (quadtree_paths_device.h) when the tree fits the tables the LDS behind the workspace holds
    // A sparse level (n <= 2 N) takes the general form at once: its tree grows until (almost) every candidate is alone — 8-10
    // generations for neighbours two pixels apart, below any table the LDS holds, or tables of six generations whose zeroing and
    // summing cost more than the general form's passes over a few hundred candidates (low-texture stereo frame: 0.206 ms with an
    // attempt on every level, 0.196 with attempts on N < n <= 2 N only, general form everywhere 0.188-0.193).
    const int path_min_n = MSORB_QT_PATH_MIN_N_FACTOR * lv.quota[level];
    if (path_cap > 0 && n > path_min_n && (debug == 0 || debug == 3)) {
        const int N = lv.quota[level], n_ini = lv.n_ini[level];
        qt::PathTables pt;
        qt::path_tables_carve(pt, qt_mem + qt::workspace_bytes(ws_N, ws_nini), n_ini, qt::path_gmax(N, n_ini, path_cap), lv.W[level], lv.H[level]);
        kept = qt::select_paths<PC>(ex, reinterpret_cast<const qt::Pt*>(compact + off), n, lv.W[level], lv.H[level], N, w, pt, out);
    }
    if (kept < 0) {
        {
            // (opaque to the optimiser: with the candidate loads and the workspace pointers of the two forms merged, their bodies
            // shared one register allocation — 113 VGPRs + 104 bytes of scratch per lane against 82 / 95 and none on their own)
            asm volatile("" : "+s"(off) :: "memory");
            qt::workspace_carve(w, qt_mem, ws_N, ws_nini);
        }
        kept = qt::select<PC>(ex, reinterpret_cast<const qt::Pt*>(compact + off), n, label + off, lv.W[level], lv.H[level],
                              lv.quota[level], w, out, debug);
    }
    if (threadIdx.x == 0) sel_n[(size_t)img * lv.nlevels + level] = kept;
    ex.dump();
}
template <int PC>
__global__ __launch_bounds__(1024) void quadtree_select_kernel(QtLevels lv, const Cand16* __restrict__ compact, const int* __restrict__ img_base,
                                                              const int* __restrict__ level_count, uint16_t* __restrict__ label,
                                                              int* __restrict__ sel_pt, int* __restrict__ sel_n, int sel_stride, int ws_N,
                                                              int ws_nini, int debug, int big_levels, int small_nt, int path_cap) {
    quadtree_select_body<PC, PC == kQtPointsPerThreadFrame>(lv, compact, img_base, level_count, label, sel_pt, sel_n, sel_stride, ws_N, ws_nini, debug, big_levels, small_nt, path_cap);
}
// The selection of a FRAME with the blur of its levels in the same launch: the selection keeps n_images x nlevels workgroups = 16
// of the chip's 256 CUs busy for 35-40 us (its level-0 instance is a chain of dependent steps), the blur — 9 us of bandwidth
// work that only the descriptor stage reads — runs on the others and leaves the frame's critical path (it sat beside FAST, which
// it lengthened by 3.5 us, and on a side stream before that: an event record, two stream waits, ~8 us of fork / join).  Blocks
// (x, y >= nlevels) are blur blocks: 1024 threads = four 256-thread blocks of gauss7_stream_kernel's numbering, waves independent.
template <int PC>
__global__ __launch_bounds__(1024) void quadtree_select_blur_kernel(QtLevels lv, const Cand16* __restrict__ compact, const int* __restrict__ img_base,
                                                                   const int* __restrict__ level_count, uint16_t* __restrict__ label,
                                                                   int* __restrict__ sel_pt, int* __restrict__ sel_n, int sel_stride, int ws_N,
                                                                   int ws_nini, int debug, int big_levels, int small_nt, int path_cap,
                                                                   FrameBlurJob blur) {
    if ((int)blockIdx.y >= lv.nlevels) {
        const int b = ((int)blockIdx.y - lv.nlevels) * (int)gridDim.x + (int)blockIdx.x;
        const int tile = 4 * b + (int)(threadIdx.x >> 8);
        if (tile < blur.blocks) gauss7_stream_body<kGaussRows>(blur.src, blur.dst, blur.plan, tile, (int)((threadIdx.x >> 6) & 3));
        return;
    }
    quadtree_select_body<PC, true>(lv, compact, img_base, level_count, label, sel_pt, sel_n, sel_stride, ws_N, ws_nini, debug, big_levels, small_nt, path_cap);
}
// Batch form: 256-thread instances whose first PC x 256 candidates stay in registers for the whole selection (a level-0 instance of
// the BASELINE geometries has ~6 800): the per-generation point passes then touch no global memory at all.
template <int PC>
__global__ __launch_bounds__(256, 4) void quadtree_select_batch_kernel(QtLevels lv, const Cand16* __restrict__ compact, const int* __restrict__ img_base,
                                                                   const int* __restrict__ level_count, uint16_t* __restrict__ label,
                                                                   int* __restrict__ sel_pt, int* __restrict__ sel_n, int sel_stride, int ws_N,
                                                                   int ws_nini, int debug, int big_levels, int small_nt, int path_cap) {
    quadtree_select_body<PC, false>(lv, compact, img_base, level_count, label, sel_pt, sel_n, sel_stride, ws_N, ws_nini, debug, big_levels, small_nt, path_cap);
}

// One workgroup per image: records in level-major / quadtree order; output row = mono index from the front for
// keypoints outside [lap0, lap1], stereo index from the back for those inside (ORBextractor.cc:1153-1162).
// A stereo frame's launch also writes the band record of every right keypoint (StereoRowJob, stereo_rowtable_device.h): what
// Frame::ComputeStereoMatches' vRowIndices would hold about it, from the same selection records.
__global__ __launch_bounds__(256) void quadtree_layout_kernel(QtLevels lv, const Cand16* __restrict__ compact,
                                                              const int* __restrict__ img_base,
                                                              const int* __restrict__ level_count,
                                                              const int* __restrict__ sel_pt, const int* __restrict__ sel_n,
                                                              int sel_stride, LevelScale scales, int lap0, int lap1,
                                                              int capacity, SelRec* __restrict__ sel,
                                                              int* __restrict__ sel_count, int* __restrict__ mono_out, int n_images,
                                                              StereoRowJob job) {
    __shared__ int lvl_begin[kMaxLevels + 1], cand_begin[kMaxLevels + 1];
    __shared__ int part[256];
    const int img = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        int a = 0, c = img_base[img];
        for (int l = 0; l < lv.nlevels; l++) {
            lvl_begin[l] = a; cand_begin[l] = c;
            a += sel_n[(size_t)img * lv.nlevels + l];
            c += level_count[(size_t)img * lv.nlevels + l];
        }
        lvl_begin[lv.nlevels] = a;
    }
    __syncthreads();
    const int n_all = lvl_begin[lv.nlevels];
    const int n = min(n_all, min(capacity, sel_stride));
    const int per = (n + 255) / 256;
    const int b = tid * per, e = min(b + per, n);
    SelRec* out = sel + (size_t)img * sel_stride;
    int2* const bands = job.band && img == job.right_img ? job.band : nullptr;
    if (bands && tid == 0 && job.n_oob) *job.n_oob = 0;
    if (lap1 < kMinBorder) {
        // no lapping area (mono / rectified stereo: every x is >= kMinBorder, so fx <= lap1 never holds): output row =
        // selection order, no second pass.  Items are taken 256 apart, four at a time, so that the two dependent loads
        // of an item (selected index -> candidate) are in flight for four items at once.
        for (int base = 0; base < n; base += 4 * 256) {
            int gg[4], ll[4], pt[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                gg[u] = base + u * 256 + tid;
                int l = 0;
                if (gg[u] < n) while (gg[u] >= lvl_begin[l + 1]) l++;
                ll[u] = l;
                pt[u] = gg[u] < n ? sel_pt[(size_t)img * sel_stride + lv.sel_off[l] + (gg[u] - lvl_begin[l])] : 0;
            }
            Cand16 cc[4];
#pragma unroll
            for (int u = 0; u < 4; u++) cc[u] = gg[u] < n ? compact[cand_begin[ll[u]] + pt[u]] : Cand16{};
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (gg[u] >= n) continue;
                SelRec r;
                r.x = (uint16_t)(cc[u].x + kMinBorder); r.y = (uint16_t)(cc[u].y + kMinBorder);
                r.score = cc[u].score; r.level = (uint8_t)ll[u]; r.pad = 0;
                r.dst = gg[u];
                out[gg[u]] = r;
                if (bands) {   // kp.pt = level coordinates times the level's scale factor, as the descriptor stage writes them (ORBextractor.cc:1149-1151)
                    const float fx = (float)r.x, fy = (float)r.y;
                    const int l = ll[u];
                    bands[gg[u]] = stereo_band_record(l ? __fmul_rn(fx, scales.scale[l]) : fx, l ? __fmul_rn(fy, scales.scale[l]) : fy, l, scales.scale, job.rows0);
                }
            }
        }
        if (tid == 0) {
            sel_count[img] = n_all > n ? -n_all : n;
            mono_out[img] = n;
        }
        return;
    }
    int lap_cnt = 0;
    for (int g = b; g < e; g++) {
        int l = 0;
        while (g >= lvl_begin[l + 1]) l++;
        const int pt = sel_pt[(size_t)img * sel_stride + lv.sel_off[l] + (g - lvl_begin[l])];
        const Cand16 c = compact[cand_begin[l] + pt];
        SelRec r;
        r.x = (uint16_t)(c.x + kMinBorder); r.y = (uint16_t)(c.y + kMinBorder);
        r.score = c.score; r.level = (uint8_t)l; r.pad = 0;
        const float fx = l ? __fmul_rn((float)r.x, scales.scale[l]) : (float)r.x;
        const bool lap = fx >= (float)lap0 && fx <= (float)lap1;
        r.dst = lap ? -1 : 0;  // provisional
        lap_cnt += lap;
        out[g] = r;
    }
    part[tid] = lap_cnt;
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int i = 0; i < 256; i++) { const int v = part[i]; part[i] = acc; acc += v; }
        sel_count[img] = n_all > n ? -n_all : n;   // negative = capacity exceeded (host turns it into an error)
        mono_out[img] = n - acc;
    }
    __syncthreads();
    int laps = part[tid];
    for (int g = b; g < e; g++) {
        if (out[g].dst < 0) { out[g].dst = n - 1 - laps; laps++; }
        else out[g].dst = g - laps;
    }
}

// Batches without a lapping area: the same records, but stored in the order the DESCRIPTOR stage should work through them —
// level-major as before, inside a level by tiles of 64 x 32 pixels, rows of tiles in serpentine order.  Output rows do not
// move (`dst` is the selection order, as in the plain form): only which keypoints the four waves of a descriptor workgroup,
// and the workgroups that share a CU, hold at the same time.  Keypoints in quadtree order are scattered over the level; in
// tile order their IC-angle patches and blurred neighbourhoods share cache lines, and describe_kernel — bound by the lines
// it pulls from L2 — is 8-9 % shorter (tools/experiments/README.md has the tile shapes that were measured).
// A counting sort in LDS: a pass that builds the records and takes a rank inside the record's tile (LDS atomic: the order
// inside a tile is whatever the atomics make it, and irrelevant), a scan over the tiles, a pass that stores.
struct TileOrder {
    int bin_begin[kMaxLevels + 1];   // first bin of each level; bins of a level = its tiles, row-major
    int ntx[kMaxLevels];             // tiles per row of tiles
    int cap;                         // records the LDS block holds (>= min(capacity, sel_stride))
};
// 256 threads: alone the launch takes 18 us where 1024 threads take 14.5 (every record's two dependent loads in flight at once),
// but a 16-wave workgroup has to wait for a CU with sixteen free wave slots beside the other batch's kernels: the pipelined step
// reads 1.048 ms with 256 threads, 1.079 with 1024 and 1.069 in selection order (five alternations on one box).
constexpr int kSortedLayoutThreads = 256;
__global__ __launch_bounds__(kSortedLayoutThreads) void quadtree_layout_sorted_kernel(QtLevels lv, const Cand16* __restrict__ compact,
                                                                     const int* __restrict__ img_base, const int* __restrict__ level_count,
                                                                     const int* __restrict__ sel_pt, const int* __restrict__ sel_n,
                                                                     int sel_stride, int capacity, SelRec* __restrict__ sel,
                                                                     int* __restrict__ sel_count, int* __restrict__ mono_out, TileOrder order) {
    extern __shared__ uint32_t tile_lds[];   // [bins] counters, then starts | [cap] records (2 dwords) | [cap] ranks (16 bit)
    __shared__ int lvl_begin[kMaxLevels + 1], cand_begin[kMaxLevels + 1], bin_begin[kMaxLevels + 1], ntx[kMaxLevels];
    __shared__ int part[256];
    constexpr int T = kSortedLayoutThreads;
    static_assert(T >= 256, "the scan over the tiles runs on the first 256 threads");
    const int img = blockIdx.x, tid = threadIdx.x;
    const int nb = order.bin_begin[lv.nlevels];
    uint32_t* const cnt = tile_lds;
    uint32_t* const rec = tile_lds + ((nb + 3) & ~3);
    uint16_t* const rank = reinterpret_cast<uint16_t*>(rec + 2 * (size_t)order.cap);
    if (tid == 0) {
        int a = 0, c = img_base[img];
        for (int l = 0; l < lv.nlevels; l++) {
            lvl_begin[l] = a; cand_begin[l] = c;
            a += sel_n[(size_t)img * lv.nlevels + l];
            c += level_count[(size_t)img * lv.nlevels + l];
        }
        lvl_begin[lv.nlevels] = a;
    }
    if (tid <= lv.nlevels) bin_begin[tid] = order.bin_begin[tid];
    if (tid < lv.nlevels) ntx[tid] = order.ntx[tid];
    for (int i = tid; i < nb; i += T) cnt[i] = 0;
    __syncthreads();
    const int n_all = lvl_begin[lv.nlevels];
    const int n = min(n_all, min(capacity, sel_stride));
    SelRec* out = sel + (size_t)img * sel_stride;
    auto bin_of = [&](uint32_t x, uint32_t y, int l) -> int {
        const int w = ntx[l], ty = (int)(y >> 5);
        int tx = min((int)(x >> 6), w - 1);
        if (ty & 1) tx = w - 1 - tx;
        return min(bin_begin[l] + ty * w + tx, bin_begin[l + 1] - 1);
    };
    // pass 1 (items T apart, two at a time: the two dependent loads of an item are in flight for all items at once)
    for (int base = 0; base < n; base += 2 * T) {
        int gg[2], ll[2], pt[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            gg[u] = base + u * T + tid;
            int l = 0;
            if (gg[u] < n) while (gg[u] >= lvl_begin[l + 1]) l++;
            ll[u] = l;
            pt[u] = gg[u] < n ? sel_pt[(size_t)img * sel_stride + lv.sel_off[l] + (gg[u] - lvl_begin[l])] : 0;
        }
        Cand16 cc[2];
#pragma unroll
        for (int u = 0; u < 2; u++) cc[u] = gg[u] < n ? compact[cand_begin[ll[u]] + pt[u]] : Cand16{};
#pragma unroll
        for (int u = 0; u < 2; u++) {
            if (gg[u] >= n) continue;
            const uint32_t x = (uint32_t)cc[u].x + kMinBorder, y = (uint32_t)cc[u].y + kMinBorder;
            rec[2 * gg[u] + 0] = x | (y << 16);
            rec[2 * gg[u] + 1] = (uint32_t)cc[u].score | ((uint32_t)ll[u] << 16);
            rank[gg[u]] = (uint16_t)atomicAdd(&cnt[bin_of(x, y, ll[u])], 1u);
        }
    }
    __syncthreads();
    // exclusive scan over the tiles: a run of consecutive bins per thread of the first four waves
    const int per = (nb + 255) / 256, b0 = min(tid * per, nb), b1 = tid < 256 ? min(b0 + per, nb) : b0;
    int mine = 0;
    for (int i = b0; i < b1; i++) mine += (int)cnt[i];
    if (tid < 256) part[tid] = mine;
    __syncthreads();
    if (tid < 256) {
        int run = 0;
        for (int i = 0; i < tid; i++) run += part[i];
        for (int i = b0; i < b1; i++) { const int c = (int)cnt[i]; cnt[i] = (uint32_t)run; run += c; }
    }
    __syncthreads();
    // pass 2
    for (int g = tid; g < n; g += T) {
        const uint32_t xy = rec[2 * g], sl = rec[2 * g + 1];
        SelRec r;
        r.x = (uint16_t)(xy & 0xffffu); r.y = (uint16_t)(xy >> 16);
        r.score = (uint16_t)(sl & 0xffffu); r.level = (uint8_t)(sl >> 16); r.pad = 0;
        r.dst = g;
        out[(int)cnt[bin_of(r.x, r.y, r.level)] + (int)rank[g]] = r;
    }
    if (tid == 0) {
        sel_count[img] = n_all > n ? -n_all : n;   // negative = capacity exceeded (host turns it into an error)
        mono_out[img] = n;
    }
}

// The layout of a frame or two without a lapping area: 1024 threads (two records each, both dependent load pairs in flight), the
// per-level offsets from uniform loads in every thread (no LDS, no barrier) — the 256-thread form above spent 11 us of a stereo
// frame here, most of it one thread's chain of sixteen dependent loads and four sequential rounds of two.
__global__ __launch_bounds__(1024) void quadtree_layout_frame_kernel(QtLevels lv, const Cand16* __restrict__ compact, const int* __restrict__ img_base,
                                                                    const int* __restrict__ level_count, const int* __restrict__ sel_pt,
                                                                    const int* __restrict__ sel_n, int sel_stride, LevelScale scales, int capacity,
                                                                    SelRec* __restrict__ sel, int* __restrict__ sel_count, int* __restrict__ mono_out,
                                                                    StereoRowJob job) {
    const int img = blockIdx.x, tid = threadIdx.x;
    int lb[kMaxLevels + 1], cb[kMaxLevels];
    {
        int sn[kMaxLevels], lc[kMaxLevels];
#pragma unroll
        for (int l = 0; l < kMaxLevels; l++) {
            sn[l] = l < lv.nlevels ? sel_n[(size_t)img * lv.nlevels + l] : 0;
            lc[l] = l < lv.nlevels ? level_count[(size_t)img * lv.nlevels + l] : 0;
        }
        int a = 0, c = img_base[img];
#pragma unroll
        for (int l = 0; l < kMaxLevels; l++) { lb[l] = a; cb[l] = c; a += sn[l]; c += lc[l]; }
        lb[kMaxLevels] = a;
    }
    const int n_all = lb[kMaxLevels];
    const int n = min(n_all, min(capacity, sel_stride));
    SelRec* out = sel + (size_t)img * sel_stride;
    int2* const bands = job.band && img == job.right_img ? job.band : nullptr;
    if (bands && tid == 0 && job.n_oob) *job.n_oob = 0;
    if (bands && job.level_begin && tid <= kMaxLevels) {
        int v = n;
#pragma unroll
        for (int k = 0; k <= kMaxLevels; k++) if (tid == k) v = min(lb[k], n);
        job.level_begin[tid] = v;
    }
    for (int base = 0; base < n; base += 2 * 1024) {
        int gg[2], ll[2], pt[2], cbeg[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            gg[u] = base + u * 1024 + tid;
            int l = 0, lbeg = 0, off = lv.sel_off[0];
            cbeg[u] = cb[0];
#pragma unroll
            for (int k = 1; k < kMaxLevels; k++)
                if (k < lv.nlevels && gg[u] >= lb[k]) { l = k; lbeg = lb[k]; off = lv.sel_off[k]; cbeg[u] = cb[k]; }
            ll[u] = l;
            pt[u] = gg[u] < n ? sel_pt[(size_t)img * sel_stride + off + (gg[u] - lbeg)] : 0;
        }
        Cand16 cc[2];
#pragma unroll
        for (int u = 0; u < 2; u++) cc[u] = gg[u] < n ? compact[cbeg[u] + pt[u]] : Cand16{};
#pragma unroll
        for (int u = 0; u < 2; u++) {
            if (gg[u] >= n) continue;
            SelRec r;
            r.x = (uint16_t)(cc[u].x + kMinBorder); r.y = (uint16_t)(cc[u].y + kMinBorder);
            r.score = cc[u].score; r.level = (uint8_t)ll[u]; r.pad = 0;
            r.dst = gg[u];
            out[gg[u]] = r;
            if (bands) {   // kp.pt = level coordinates times the level's scale factor, as the descriptor stage writes them (ORBextractor.cc:1149-1151)
                const float fx = (float)r.x, fy = (float)r.y;
                const int l = ll[u];
                bands[gg[u]] = stereo_band_record(l ? __fmul_rn(fx, scales.scale[l]) : fx, l ? __fmul_rn(fy, scales.scale[l]) : fy, l, scales.scale, job.rows0);
            }
        }
    }
    if (tid == 0) {
        sel_count[img] = n_all > n ? -n_all : n;   // negative = capacity exceeded (host turns it into an error)
        mono_out[img] = n;
    }
}

int launch_quadtree(const QtLevels& lv, const Cand16* compact, const int* img_base, const int* level_count,
                     uint16_t* label, int* sel_pt, int* sel_n, int sel_stride, const LevelScale& scales, int lap0,
                     int lap1, int capacity, SelRec* sel, int* sel_count, int* mono, int n_images, hipStream_t s,
                     const StereoRowJob* row_job, const FrameBlurJob* blur_job, bool* blur_carried) {
    if (blur_carried) *blur_carried = false;
    int maxN = 1, max_ini = 1;
    for (int l = 0; l < lv.nlevels; l++) { maxN = max(maxN, lv.quota[l]); max_ini = max(max_ini, lv.n_ini[l]); }
    const size_t lds = qt::workspace_bytes(maxN, max_ini);
#ifdef MSORB_QT_MARKS   // profiling build (tools/qt_marks.sh): MSORB_QT_DEBUG=3 prints the per-phase timestamps of instance (0, 0)
    static const int dbg0 = getenv("MSORB_QT_DEBUG") ? atoi(getenv("MSORB_QT_DEBUG")) : 0;
#else
    const int dbg0 = 0;
#endif
#ifdef MSORB_AB_BUILD    // in-process A/B builds (tools/ab_inprocess.py) read the switch per call; the library proper once
    const bool lane_sort_off = getenv("MSORB_QT_LANE_SORT") && atoi(getenv("MSORB_QT_LANE_SORT")) == 0;
#else
    static const bool lane_sort_off = getenv("MSORB_QT_LANE_SORT") && atoi(getenv("MSORB_QT_LANE_SORT")) == 0;
#endif
    const int dbg = dbg0 | (lane_sort_off ? 0x100 : 0);
    // Workgroup size by batch size: the generations are chains of dependent LDS round trips, hidden only by other
    // waves.  A big batch has other workgroups on the CU for that (256 threads: least barrier idling, best
    // throughput); a frame or two has nothing else, so the instance itself brings the waves (1024 threads).
    const int qt_threads = n_images <= 4 ? 1024 : n_images <= 16 ? 512 : kQtThreads;
    const int big_levels = kMaxLevels, small_nt = qt_threads;
    auto raise_lds = [&](const void* fn) -> bool {   // quotas beyond ~700 keypoints per level: past the default dynamic-LDS limit
        if (lds <= 64 * 1024) return true;
        if ((long long)lds <= dynamic_lds_room(fn)) return true;   // raised once per device to all the LDS there is, never lowered
        set_last_error("quadtree: the device refuses " + std::to_string(lds) + " bytes of LDS per workgroup");
        return false;
    };
    // path tables behind the workspace: the deepest generation (<= max_cap) whose tables the LDS still holds for every level
    static const bool paths_off = getenv("MSORB_QT_PATHS") && atoi(getenv("MSORB_QT_PATHS")) == 0;
    auto path_tables = [&](const void* fn, int max_cap, long long budget, int& path_cap, size_t& lds_paths) {
        path_cap = 0; lds_paths = 0;
        if (paths_off) return;
        const long long room = std::min<long long>(dynamic_lds_room(fn), budget);
        for (int cap = max_cap; cap >= 2 && !path_cap; cap--) {
            size_t need = 0;
            for (int l = 0; l < lv.nlevels; l++)
                need = max(need, qt::path_tables_bytes(lv.n_ini[l], qt::path_gmax(1 << 20, lv.n_ini[l], cap), lv.W[l], lv.H[l]));
            if ((long long)(lds + need) <= room) { path_cap = cap; lds_paths = need; }
        }
    };
    int path_cap = 0;
    size_t lds_paths = 0;
    if (qt_threads == 256) {
        const void* fn = reinterpret_cast<const void*>(quadtree_select_batch_kernel<kQtPointsPerThreadBatch>);
        // batches: tables of four generations keep three workgroups on a CU (workspace 39.7 KB + 11.7 KB at the KITTI quota; five
        // generations leave two: select stage alone 0.120 / 0.159 ms per 256 images against 0.139 without tables, pipelined step
        // 1.091 / 1.095 against 1.131 ms); a level whose tree grows deeper runs the general form
        static const int batch_cap = getenv("MSORB_QT_BATCH_PATHS") ? atoi(getenv("MSORB_QT_BATCH_PATHS")) : 4;
        if (batch_cap > 0) path_tables(fn, batch_cap, 80 * 1024, path_cap, lds_paths);
        if (!path_cap && !raise_lds(fn)) return MSORB_E_HIP;
        hipLaunchKernelGGL(quadtree_select_batch_kernel<kQtPointsPerThreadBatch>, dim3(n_images, lv.nlevels), dim3(qt_threads), lds + lds_paths, s, lv,
                           compact, img_base, level_count, label, sel_pt, sel_n, sel_stride, maxN, max_ini, dbg, big_levels, small_nt, path_cap);
    } else if (qt_threads == 1024) {
        if (blur_job && blur_job->blocks > 0) {   // the frame's blur rides this launch (quadtree_select_blur_kernel)
            const void* fn = reinterpret_cast<const void*>(quadtree_select_blur_kernel<kQtPointsPerThreadFrame>);
            path_tables(fn, 6, 1 << 30, path_cap, lds_paths);
            if (!path_cap && !raise_lds(fn)) return MSORB_E_HIP;
            const int wgs = (blur_job->blocks + 3) / 4, extra_rows = (wgs + n_images - 1) / n_images;
            hipLaunchKernelGGL(quadtree_select_blur_kernel<kQtPointsPerThreadFrame>, dim3(n_images, lv.nlevels + extra_rows), dim3(qt_threads),
                               lds + lds_paths, s, lv, compact, img_base, level_count, label, sel_pt, sel_n, sel_stride, maxN, max_ini, dbg,
                               big_levels, small_nt, path_cap, *blur_job);
            if (blur_carried) *blur_carried = true;
        } else {
        const void* fn = reinterpret_cast<const void*>(quadtree_select_kernel<kQtPointsPerThreadFrame>);
        path_tables(fn, 6, 1 << 30, path_cap, lds_paths);
        if (!path_cap && !raise_lds(fn)) return MSORB_E_HIP;
        hipLaunchKernelGGL(quadtree_select_kernel<kQtPointsPerThreadFrame>, dim3(n_images, lv.nlevels), dim3(qt_threads), lds + lds_paths, s, lv,
                           compact, img_base, level_count, label, sel_pt, sel_n, sel_stride, maxN, max_ini, dbg, big_levels, small_nt, path_cap);
        }
    } else {
        if (!raise_lds(reinterpret_cast<const void*>(quadtree_select_kernel<0>))) return MSORB_E_HIP;
        path_tables(reinterpret_cast<const void*>(quadtree_select_kernel<0>), 5, 80 * 1024, path_cap, lds_paths);
        hipLaunchKernelGGL(quadtree_select_kernel<0>, dim3(n_images, lv.nlevels), dim3(qt_threads), lds + lds_paths, s, lv, compact, img_base,
                           level_count, label, sel_pt, sel_n, sel_stride, maxN, max_ini, dbg, big_levels, small_nt, path_cap);
    }
    const StereoRowJob job = row_job && lap1 < kMinBorder && row_job->right_img < n_images ? *row_job : StereoRowJob{};
    if (n_images <= 4 && lap1 < kMinBorder)
        hipLaunchKernelGGL(quadtree_layout_frame_kernel, dim3(n_images), dim3(1024), 0, s, lv, compact, img_base, level_count, sel_pt, sel_n,
                           sel_stride, scales, capacity, sel, sel_count, mono, job);
    else {
        // batches without a lapping area: records in the descriptor stage's tile order (MSORB_DESC_ORDER=0: selection order)
        static const bool tile_order_off = getenv("MSORB_DESC_ORDER") && atoi(getenv("MSORB_DESC_ORDER")) == 0;
        TileOrder order{};
        size_t order_lds = 0;
        if (!tile_order_off && n_images > 4 && lap1 < kMinBorder && !job.band) {
            int nb = 0;
            for (int l = 0; l < lv.nlevels; l++) {
                order.bin_begin[l] = nb;
                order.ntx[l] = ((lv.W[l] + 2 * kMinBorder) >> 6) + 1;
                nb += order.ntx[l] * (((lv.H[l] + 2 * kMinBorder) >> 5) + 1);
            }
            order.bin_begin[lv.nlevels] = nb;
            order.cap = std::min(capacity, sel_stride);
            order_lds = ((size_t)((nb + 3) & ~3) + 2 * (size_t)order.cap) * 4 + 2 * (size_t)order.cap + 8;
            if (order.cap <= 0 || order.cap > 65535 || order_lds > 60 * 1024) order_lds = 0;   // (a geometry this form was not sized for: plain order)
        }
        if (order_lds)
            hipLaunchKernelGGL(quadtree_layout_sorted_kernel, dim3(n_images), dim3(kSortedLayoutThreads), order_lds, s, lv, compact, img_base, level_count, sel_pt,
                               sel_n, sel_stride, capacity, sel, sel_count, mono, order);
        else
            hipLaunchKernelGGL(quadtree_layout_kernel, dim3(n_images), dim3(256), 0, s, lv, compact, img_base, level_count, sel_pt,
                               sel_n, sel_stride, scales, lap0, lap1, capacity, sel, sel_count, mono, n_images, job);
    }
    return MSORB_OK;
}
// Test hook (msorb_debug_std_sort): the careful loop's std::sort restatement ALONE, as the selection kernels run it — FRAME: the
// 1024-thread queue form with wave_introsort64 below 65 items; otherwise the 256-thread level-synchronous form — on explicit keys;
// items = (key, position in the input).  The result must be the permutation libstdc++'s std::sort produces (tests/test_quadtree_sort_gpu.py).
template <bool FRAME>
__global__ __launch_bounds__(FRAME ? 1024 : 256) void debug_sort_kernel(const uint32_t* __restrict__ keys, int n, int m, uint32_t* __restrict__ out_key,
                                                                        uint32_t* __restrict__ out_node, int lane_sort, long long* __restrict__ ticks) {
    extern __shared__ __attribute__((aligned(16))) char mem[];
    char* p = mem;
    QT_LDS qt::SortItem* items = (QT_LDS qt::SortItem*)p; p += (size_t)(m + 4) * sizeof(qt::SortItem);
    qt::ParScratch ps;
    ps.tmp = (QT_LDS qt::SortItem*)p; p += (size_t)(m + 4) * sizeof(qt::SortItem);
    ps.gpos = (QT_LDS uint16_t*)p; p += (size_t)(m + 4) * sizeof(uint16_t);
    ps.lpos = (QT_LDS uint16_t*)p; p += (size_t)(m + 4) * sizeof(uint16_t);
    QT_LDS int* stack = (QT_LDS int*)p; p += 2 * 3 * (size_t)qt::stack_ranges(m) * sizeof(int);
    ps.stack_half = 3 * qt::stack_ranges(m);
    ps.scan_tmp = (QT_LDS int*)p; p += 16 * sizeof(int);
    ps.sc = (QT_LDS int*)p;
    for (int i = threadIdx.x; i < n; i += blockDim.x) items[i] = qt::SortItem{keys[i], (uint32_t)i};
    __syncthreads();
    DevExT<FRAME> ex;
    ex.nt = (int)blockDim.x;
    ex.kLaneSort = lane_sort != 0;
    const long long t0 = wall_clock64();
    ex.sort(items, n, stack, ps);
    __syncthreads();
    if (threadIdx.x == 0 && ticks) *ticks = wall_clock64() - t0;   // 100 MHz constant clock: 10 ns per tick
    for (int i = threadIdx.x; i < n; i += blockDim.x) { out_key[i] = items[i].key; out_node[i] = items[i].node; }
}
int launch_debug_sort(const uint32_t* h_keys, int n, int frame_form, uint32_t* h_nodes, uint32_t* h_keys_out, float* sort_us) {
    if (n < 0 || n > 4000) return MSORB_E_INVALID;
    if (n == 0) return MSORB_OK;
    const int m = ((n + 15) & ~1) | 0;   // (m + 4 items; gpos / lpos stay 4-byte aligned for an even m)
    const size_t lds = 2 * (size_t)(m + 4) * sizeof(qt::SortItem) + 2 * (size_t)(m + 4) * sizeof(uint16_t) +
                       2 * 3 * (size_t)qt::stack_ranges(m) * sizeof(int) + 20 * sizeof(int) + 64;
    uint32_t *d_in = nullptr, *d_k = nullptr, *d_n = nullptr;
    hipError_t e = hipMalloc((void**)&d_in, 3 * (size_t)n * sizeof(uint32_t) + 16);
    if (e != hipSuccess) return MSORB_E_HIP;
    d_k = d_in + n; d_n = d_k + n;
    long long* d_ticks = reinterpret_cast<long long*>(d_in + ((3 * (size_t)n + 1) & ~size_t(1)));
    const int lane_sort = !(frame_form & 2);
    frame_form &= 1;
    e = hipMemcpy(d_in, h_keys, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        if (frame_form) {
            if ((long long)lds > dynamic_lds_room(reinterpret_cast<const void*>(debug_sort_kernel<true>))) { (void)hipFree(d_in); return MSORB_E_INVALID; }
            hipLaunchKernelGGL(debug_sort_kernel<true>, dim3(1), dim3(1024), lds, 0, d_in, n, m, d_k, d_n, lane_sort, d_ticks);
        } else {
            if ((long long)lds > dynamic_lds_room(reinterpret_cast<const void*>(debug_sort_kernel<false>))) { (void)hipFree(d_in); return MSORB_E_INVALID; }
            hipLaunchKernelGGL(debug_sort_kernel<false>, dim3(1), dim3(256), lds, 0, d_in, n, m, d_k, d_n, lane_sort, d_ticks);
        }
        e = hipDeviceSynchronize();
    }
    if (e == hipSuccess) e = hipMemcpy(h_nodes, d_n, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess && h_keys_out) e = hipMemcpy(h_keys_out, d_k, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess && sort_us) { long long t = 0; e = hipMemcpy(&t, d_ticks, sizeof(t), hipMemcpyDeviceToHost); *sort_us = (float)t * 0.01f; }
    (void)hipFree(d_in);
    return e == hipSuccess ? MSORB_OK : MSORB_E_HIP;
}

size_t quadtree_lds_bytes(const QtLevels& lv) {
    int maxN = 1, max_ini = 1;
    for (int l = 0; l < lv.nlevels; l++) { maxN = max(maxN, lv.quota[l]); max_ini = max(max_ini, lv.n_ini[l]); }
    return qt::workspace_bytes(maxN, max_ini);
}

}  // namespace msorb
