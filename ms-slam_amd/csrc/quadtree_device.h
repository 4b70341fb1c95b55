// DistributeOctTree (ORBextractor.cc:555-779) as a generation-synchronous, label-based algorithm that one
// workgroup executes per (image, pyramid level) on the device.  Same observable result as the reference's
// std::list walk (and as orb_host.cc's index-pool version): which keypoints survive, in which order.
//
// How the list semantics map onto flat arrays
//  * Every node created after the initial columns is push_front'ed, so the final list order is "descending
//    creation time", followed by the surviving initial columns in ascending order.  Each node therefore only
//    needs a creation sequence number `seq`; the list itself is never materialised.
//  * One pass of the reference's main loop splits every multi-point node of the previous pass, walking the list
//    from the front = in reverse creation order.  A generation is an array of child slots indexed
//    4*(parent's processing rank) + quadrant, i.e. already in creation order; empty quadrants leave gaps,
//    which is harmless because only the relative order of seq matters.
//  * Points are never moved: a point carries the label of the node that currently owns it; a node's point list
//    "in insertion order" is the set of points with its label in candidate order, and the only order-sensitive
//    use (first strictly greater response wins, :757-776) is a max over the key (response, -candidate index).
//  * The "careful" phase (:689-753) sorts (count, UL.x) with std::sort — unstable, libstdc++ introsort — and
//    the tie order decides which nodes are split before the quota is hit.  lsort() below restates libstdc++'s
//    algorithm (introsort loop, median-of-three to first, unguarded partition, heapsort fallback, final
//    insertion sort with threshold 16) so the permutation is the same for the same comparison outcomes.
//
// The code is written once against a tiny execution interface (Ex: tid/nthreads/sync/atomics) and compiled
// for the device (256-thread workgroup, LDS workspace) and for the host (1 "thread"; tests/qt_host_check.cc
// compares it with orb_host.cc::distribute_quadtree on thousands of inputs).
#pragma once
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define QT_HD __host__ __device__ __forceinline__
#else
#define QT_HD inline
#endif
// The workspace lives in LDS on the device.  Its pointers carry the LDS address space in the device pass: through generic
// pointers every access becomes a FLAT instruction (and the LDS atomics flat atomics), several times slower than ds_*.
// (MSORB_QT_GLOBAL_WORKSPACE, defined by quadtree_global_kernels.hip before this header: the same code over a workspace in
// global memory — plain pointers — for quotas whose workspace exceeds a workgroup's LDS.)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MSORB_QT_GLOBAL_WORKSPACE)
#define QT_LDS __attribute__((address_space(3)))
#else
#define QT_LDS
#endif

namespace msorb {
namespace qt {

struct Pt {  // same layout as Cand16
    uint16_t x, y, score, pad;
};
struct SortItem {  // one element of vSizeAndPointerToNode: key = count << 16 | UL.x (compareNodes order), node
    uint32_t key, node;
};

struct Int4 { int x, y, z, w; };
QT_HD Int4 load_int4(QT_LDS const int* p) {   // p is 16-byte aligned
#if defined(__HIP_DEVICE_COMPILE__)
    typedef int v4i_t __attribute__((ext_vector_type(4)));
    const v4i_t v = *reinterpret_cast<QT_LDS const v4i_t*>(p);
    return Int4{v.x, v.y, v.z, v.w};
#else
    return Int4{p[0], p[1], p[2], p[3]};
#endif
}
// A candidate's label: the node it sits in — parity of the node's generation | the node's slot (or rank) — or "settled".  16 bits
// in the LDS build (14-bit slots: 4 N + 16 <= 16384, more than any quota a workgroup's LDS holds; two labels share a register with the
// response in the register-resident form); 32 bits in the global-workspace build (MSORB_QT_GLOBAL_WORKSPACE: quotas up to what the
// 16-bit rank tables allow, 4 N + 16 <= 65535).
#if defined(MSORB_QT_GLOBAL_WORKSPACE)
typedef uint32_t Label;
constexpr uint32_t kLabelSettled = 0xFFFFFFFFu;
constexpr int kParityShift = 30;
#else
typedef uint16_t Label;
constexpr uint32_t kLabelSettled = 0xFFFFu;
constexpr int kParityShift = 14;
#endif
constexpr uint32_t kParityBit = 1u << kParityShift;
constexpr uint32_t kSlotMask = kParityBit - 1u;

// ---- libstdc++ std::sort restated for SortItem with compareNodes (ORBextractor.cc:538-553) -------------
QT_HD bool sort_less(const SortItem& a, const SortItem& b) { return a.key < b.key; }
// Storage accessor: the same sort runs on a plain array (host, LDS) or on items spread over the lanes of one
// wave (device: element i lives in lane i & 63 of register i >> 6 and is reached with v_readlane / v_writelane,
// so the serial comparison chain costs a few cycles per step instead of an LDS round trip).
struct ArrayAcc {
    QT_LDS SortItem* v;
    QT_HD SortItem get(int i) const { return v[i]; }
    QT_HD uint32_t key(int i) const { return v[i].key; }
    QT_HD void set(int i, const SortItem& x) { v[i] = x; }
};

template <class A>
QT_HD void sort_swap(A& v, int i, int j) { const SortItem t = v.get(i); v.set(i, v.get(j)); v.set(j, t); }

template <class A>
QT_HD void adjust_heap(A& v, int first, int hole, int len, SortItem value) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (v.key(first + child) < v.key(first + child - 1)) child--;
        v.set(first + hole, v.get(first + child));
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        v.set(first + hole, v.get(first + child - 1));
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;  // __push_heap
    while (hole > top && v.key(first + parent) < value.key) {
        v.set(first + hole, v.get(first + parent));
        hole = parent;
        parent = (hole - 1) / 2;
    }
    v.set(first + hole, value);
}
template <class A>
QT_HD void heap_sort(A& v, int first, int last) {  // __partial_sort(first, last, last)
    const int len = last - first;
    if (len >= 2) {
        int parent = (len - 2) / 2;
        for (;;) {
            const SortItem value = v.get(first + parent);
            adjust_heap(v, first, parent, len, value);
            if (parent == 0) break;
            parent--;
        }
    }
    int l = last;
    while (l - first > 1) {
        --l;
        const SortItem value = v.get(l);
        v.set(l, v.get(first));
        adjust_heap(v, first, 0, l - first, value);
    }
}
template <class A>
QT_HD void unguarded_linear_insert(A& v, int last) {
    const SortItem val = v.get(last);
    int next = last - 1;
    for (;;) {  // while (val < v[next]) { v[last] = v[next]; last = next; --next; }  with two items read ahead
        const int j1 = next - 1 > 0 ? next - 1 : 0;
        const SortItem a0 = v.get(next), a1 = v.get(j1);
        if (!(val.key < a0.key)) break;
        v.set(last, a0);
        last = next;
        --next;
        if (!(val.key < a1.key)) break;
        v.set(last, a1);
        last = next;
        --next;
    }
    v.set(last, val);
}
template <class A>
QT_HD void insertion_sort(A& v, int first, int last) {
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        if (v.key(i) < v.key(first)) {
            const SortItem val = v.get(i);
            for (int k = i; k > first; --k) v.set(k, v.get(k - 1));
            v.set(first, val);
        } else {
            unguarded_linear_insert(v, i);
        }
    }
}
// std::sort(v, v + n, compareNodes).  The two sub-ranges produced by a partition are disjoint, so the order in
// which __introsort_loop's recursion visits them does not affect the result; only the depth budget each range
// inherits does.  `stack` holds (first, last, depth) records: 3 * 64 ints.
template <class A>
QT_HD void lsort_acc(A& v, int n, QT_LDS int* stack) {
    if (n <= 0) return;
    int lg = 0;
    for (int t = n; t > 1; t >>= 1) lg++;
    int sp = 0;
    stack[0] = 0; stack[1] = n; stack[2] = 2 * lg;
    sp = 1;
    while (sp > 0) {
        sp--;
        const int first = stack[3 * sp];
        int last = stack[3 * sp + 1], depth = stack[3 * sp + 2];
        while (last - first > 16) {  // __introsort_loop
            if (depth == 0) {
                heap_sort(v, first, last);
                break;
            }
            --depth;
            const int mid = first + (last - first) / 2;
            {  // __move_median_to_first(first, first+1, mid, last-1)
                const int a = first + 1, b = mid, c = last - 1;
                const uint32_t ka = v.key(a), kb = v.key(b), kc = v.key(c);
                if (ka < kb) {
                    if (kb < kc) sort_swap(v, first, b);
                    else if (ka < kc) sort_swap(v, first, c);
                    else sort_swap(v, first, a);
                } else if (ka < kc) sort_swap(v, first, a);
                else if (kb < kc) sort_swap(v, first, c);
                else sort_swap(v, first, b);
            }
            int lo = first + 1, hi = last;  // __unguarded_partition(first+1, last, pivot = *first)
            const uint32_t pivot = v.key(first);
            for (;;) {
                // scans read four keys ahead (independent LDS reads in flight); positions past the range are clamped
                // and only ever inspected after an in-range key has already stopped the scan
                for (;;) {
                    const int i1 = lo + 1 < n ? lo + 1 : n - 1, i2 = lo + 2 < n ? lo + 2 : n - 1, i3 = lo + 3 < n ? lo + 3 : n - 1;
                    const uint32_t k0 = v.key(lo), k1 = v.key(i1), k2 = v.key(i2), k3 = v.key(i3);
                    if (!(k0 < pivot)) break;
                    ++lo;
                    if (!(k1 < pivot)) break;
                    ++lo;
                    if (!(k2 < pivot)) break;
                    ++lo;
                    if (!(k3 < pivot)) break;
                    ++lo;
                }
                --hi;
                for (;;) {
                    const int i1 = hi - 1 > 0 ? hi - 1 : 0, i2 = hi - 2 > 0 ? hi - 2 : 0, i3 = hi - 3 > 0 ? hi - 3 : 0;
                    const uint32_t k0 = v.key(hi), k1 = v.key(i1), k2 = v.key(i2), k3 = v.key(i3);
                    if (!(pivot < k0)) break;
                    --hi;
                    if (!(pivot < k1)) break;
                    --hi;
                    if (!(pivot < k2)) break;
                    --hi;
                    if (!(pivot < k3)) break;
                    --hi;
                }
                if (!(lo < hi)) break;
                sort_swap(v, lo, hi);
                ++lo;
            }
            stack[3 * sp] = lo; stack[3 * sp + 1] = last; stack[3 * sp + 2] = depth;  // [cut, last) later
            sp++;
            last = lo;
        }
    }
    if (n > 16) {  // __final_insertion_sort
        insertion_sort(v, 0, 16);
        for (int i = 16; i < n; ++i) unguarded_linear_insert(v, i);
    } else {
        insertion_sort(v, 0, n);
    }
}
QT_HD void lsort(QT_LDS SortItem* v, int n, QT_LDS int* stack) {
    ArrayAcc a{v};
    lsort_acc(a, n, stack);
}

// ---- the same sort, data-parallel ------------------------------------------------------------------------
// std::sort's result is a deterministic function of the key sequence, so it can be reproduced without walking
// the comparison chain one element at a time:
//  * __unguarded_partition(first+1, last, pivot): let G = positions holding a key >= pivot in ascending order and
//    L = positions holding a key <= pivot in descending order.  The sequential scan swaps (G[t], L[t]) for
//    t = 0, 1, ... while G[t] < L[t]; with k such swaps the returned cut is min(G[k], L[k-1]).  Ranks in G / L are
//    prefix counts, so one scan + one scatter + one count replace the serial loop.
//  * __final_insertion_sort is a sequence of stable insertions, i.e. a stable sort of whatever the introsort
//    loop left behind: rank(i) = #{key < key_i} + #{j < i : key_j == key_i}.
//  * median-of-three and the heapsort fallback (never reached on these inputs in practice) stay serial.
// Ex: tid / nthreads / sync / excl_scan over the participating threads (one wave on the device).
struct ParScratch {
    QT_LDS uint16_t* gpos;   // capacity n
    QT_LDS uint16_t* lpos;   // capacity n
    QT_LDS SortItem* tmp;    // capacity n
    QT_LDS int* scan_tmp;    // 16 ints
    QT_LDS int* sc;          // 4 ints
    int stack_half = 96;     // ints per range list of the level-synchronous device sort (3 per range)
};

template <class Ex>
QT_HD int partition_par(Ex& ex, QT_LDS SortItem* v, int first, int last, ParScratch& ps) {
    const int tid = ex.tid(), nt = ex.nthreads();
    const uint32_t pivot = v[first].key;
    const int m = last - (first + 1);
    const int per = (m + nt - 1) / nt;
    const int cb = first + 1 + tid * per, ce = cb + per < last ? cb + per : last;
    int NG, NL, k = 0;
    if (per == 1) {
        // one element per thread: prefix counts are ballots (no shuffle chain)
        const bool in = cb < ce;
        const bool is_g = in && v[in ? cb : first].key >= pivot, is_l = in && v[in ? cb : first].key <= pivot;
        const int rg = ex.excl_count(is_g, &NG), ra = ex.excl_count(is_l, &NL);
        if (is_g) ps.gpos[rg] = (uint16_t)cb;
        if (is_l) ps.lpos[NL - 1 - ra] = (uint16_t)cb;
        ex.sync();
        const int T = NG < NL ? NG : NL;
        const bool sw = tid < T && ps.gpos[tid < T ? tid : 0] < ps.lpos[tid < T ? tid : 0];
        ex.excl_count(sw, &k);  // monotone predicate: the count is the number of swaps, and the swapping threads are 0..k-1
        if (sw) {
            const int a = ps.gpos[tid], b = ps.lpos[tid];
            const SortItem x = v[a]; v[a] = v[b]; v[b] = x;
        }
    } else {
        int ng = 0, nl = 0;
        for (int i = cb; i < ce; i++) { ng += v[i].key >= pivot; nl += v[i].key <= pivot; }
        int tot = 0;
        const int ex_packed = ex.excl_scan(ng | (nl << 16), ps.scan_tmp, &tot);
        NG = tot & 0xFFFF; NL = tot >> 16;
        int rg = ex_packed & 0xFFFF, ra = ex_packed >> 16;  // ascending exclusive ranks
        for (int i = cb; i < ce; i++) {
            if (v[i].key >= pivot) ps.gpos[rg++] = (uint16_t)i;
            if (v[i].key <= pivot) { ps.lpos[NL - 1 - ra] = (uint16_t)i; ra++; }  // descending rank
        }
        ex.sync();
        const int T = NG < NL ? NG : NL;
        int cnt = 0;
        for (int t = tid; t < T; t += nt) cnt += ps.gpos[t] < ps.lpos[t];
        ex.excl_scan(cnt, ps.scan_tmp, &k);   // G[t] < L[t] is monotone in t: the count is the number of swaps
        for (int t = tid; t < k; t += nt) {
            const int a = ps.gpos[t], b = ps.lpos[t];
            const SortItem x = v[a]; v[a] = v[b]; v[b] = x;
        }
    }
    ex.sync();
    int cut = last;
    if (k < NG) cut = ps.gpos[k];
    if (k > 0 && (int)ps.lpos[k - 1] < cut) cut = ps.lpos[k - 1];
    ex.sync();
    return cut;
}

template <class Ex>
QT_HD void lsort_par_partitions(Ex& ex, QT_LDS SortItem* v, int n, QT_LDS int* stack, ParScratch& ps) {
    if (n <= 0) return;
    const int tid = ex.tid();
    int lg = 0;
    for (int t = n; t > 1; t >>= 1) lg++;
    int sp = 1;
    if (tid == 0) { stack[0] = 0; stack[1] = n; stack[2] = 2 * lg; }
    ex.sync();
    while (sp > 0) {
        sp--;
        const int first = stack[3 * sp];
        int last = stack[3 * sp + 1], depth = stack[3 * sp + 2];
        ex.sync();
        while (last - first > 16) {
            if (depth == 0) {
                if (tid == 0) { ArrayAcc a{v}; heap_sort(a, first, last); }
                ex.sync();
                break;
            }
            --depth;
            if (tid == 0) {  // __move_median_to_first(first, first+1, mid, last-1)
                ArrayAcc acc{v};
                const int a = first + 1, b = first + (last - first) / 2, c = last - 1;
                const uint32_t ka = v[a].key, kb = v[b].key, kc = v[c].key;
                if (ka < kb) {
                    if (kb < kc) sort_swap(acc, first, b);
                    else if (ka < kc) sort_swap(acc, first, c);
                    else sort_swap(acc, first, a);
                } else if (ka < kc) sort_swap(acc, first, a);
                else if (kb < kc) sort_swap(acc, first, c);
                else sort_swap(acc, first, b);
            }
            ex.sync();
            const int cut = partition_par(ex, v, first, last, ps);
            if (tid == 0) { stack[3 * sp] = cut; stack[3 * sp + 1] = last; stack[3 * sp + 2] = depth; }
            sp++;
            ex.sync();
            last = cut;
        }
    }
}

// __final_insertion_sort == stable sort of the arrangement the introsort loop left behind (rank counting); any
// number of threads.  With more threads than items every item's comparisons are split over `parts` threads (each takes a
// stretch of j, the partial ranks meet in an LDS accumulator): the O(n^2 / threads) loop is the longest single step of a
// careful sweep otherwise (255 items on 255 of 1024 threads: 9.8 us; split four ways: see DESIGN.md section 5).
// The accumulators live in the partition position lists (gpos / lpos are contiguous, 4-byte aligned, 2 (m + 4) uint16 >= n ints),
// which nothing uses once the partitions are done.
template <class Ex>
QT_HD void final_stable_sort(Ex& ex, QT_LDS SortItem* v, int n, ParScratch& ps) {
    const int tid = ex.tid(), nt = ex.nthreads();
    if (n <= 0) return;
    int parts = 1;
    if constexpr (Ex::kSplitRank) {   // (an execution model without spare threads compiles the plain form only)
        parts = nt / n;
        parts = parts < 1 ? 1 : parts > 8 ? 8 : parts;
    }
    QT_LDS int* const acc = (QT_LDS int*)ps.gpos;
    if (parts > 1) {
        for (int i = tid; i < n; i += nt) acc[i] = 0;
        ex.sync();
    }
    for (int idx = tid; idx < n * parts; idx += nt) {
        const int i = idx % n, part = idx / n;
        const int j0 = (int)((long long)part * n / parts), j1 = (int)((long long)(part + 1) * n / parts);
        const uint32_t key = v[i].key;
        int rank = 0;
        int j = j0;
        for (; j + 4 <= j1; j += 4) {   // four independent LDS reads in flight (every lane of a wave reads the same j: broadcasts)
            const uint32_t k0 = v[j].key, k1 = v[j + 1].key, k2 = v[j + 2].key, k3 = v[j + 3].key;
            rank += (k0 < key) || (k0 == key && j < i);
            rank += (k1 < key) || (k1 == key && j + 1 < i);
            rank += (k2 < key) || (k2 == key && j + 2 < i);
            rank += (k3 < key) || (k3 == key && j + 3 < i);
        }
        for (; j < j1; j++) {
            const uint32_t kj = v[j].key;
            rank += (kj < key) || (kj == key && j < i);
        }
        if (parts > 1) ex.atomic_add(acc + i, rank);
        else ps.tmp[rank] = v[i];
    }
    ex.sync();
    if (parts > 1) {
        for (int i = tid; i < n; i += nt) ps.tmp[acc[i]] = v[i];
        ex.sync();
    }
    for (int i = tid; i < n; i += nt) v[i] = ps.tmp[i];
    ex.sync();
}

template <class Ex>
QT_HD void lsort_par(Ex& ex, QT_LDS SortItem* v, int n, QT_LDS int* stack, ParScratch& ps) {
    lsort_par_partitions(ex, v, n, stack, ps);
    final_stable_sort(ex, v, n, ps);
}

// ---- the selection itself ----------------------------------------------------------------------------
struct NodeB {  // a multi-point ("splittable") node of the current generation, stored by processing rank
    int16_t x0, x1, y0, y1;
    uint32_t mid;   // split point of DivideNode (:511-525): x0 + ceil(w / 2) | (y0 + ceil(h / 2)) << 16 — all a point pass reads of a node
    uint16_t slot;  // slot in its generation (creation order); the creation sequence number (final list order = descending
    uint16_t pad;   // sequence) is the generation's base + slot, -1 - slot for the initial columns
};
QT_HD uint32_t node_mid(int x0, int x1, int y0, int y1) {
    return (uint32_t)(x0 + ((x1 - x0 + 1) >> 1)) | ((uint32_t)(y0 + ((y1 - y0 + 1) >> 1)) << 16);
}
QT_HD int quadrant_of_mid(uint32_t xy, uint32_t mid) {  // xy = x | y << 16; DivideNode's assignment
    return ((xy & 0xFFFFu) < (mid & 0xFFFFu) ? 0 : 1) + ((xy >> 16) < (mid >> 16) ? 0 : 2);
}
struct Workspace {       // LDS on the device; `cap` = 4 * max(N, nIni) child slots per generation
    QT_LDS int* cnt[2];         // points per child slot (re-used as best-point keys at the end)
    QT_LDS uint16_t* rankof[2]; // child slot -> processing rank among multi-point nodes (0xFFFF = none)
    QT_LDS NodeB* nb[2];        // multi-point nodes by rank, capacity N + 4
    QT_LDS SortItem* items;     // capacity N + 4
    QT_LDS int* stack;          // 2 range lists of stack_ranges(m) entries x 3 ints (>= 3 * 64 ints for the serial sort)
    QT_LDS int* res_seq;        // capacity res_cap
    QT_LDS int* res_pt;
    QT_LDS int* sc;             // scalars: see enum below
    QT_LDS int* scan_tmp;       // 2 x 16 ints for the block-wide scans
    ParScratch ps;       // scratch of the data-parallel sort
    int cap, res_cap, m;
};
enum { kScSize = 0, kScS0, kScS1, kScNres, kScNToExpand, kScNsplit, kScFinish, kScCareful, kScGenBase, kScCount };

// ranges longer than 16 elements that can be open at once in the level-synchronous introsort rounds of m + 4 items
QT_HD int stack_ranges(int m) { const int r = ((m + 4) / 17 + 3) & ~1; return r > 32 ? r : 32; }   // (even: what follows stays 16-byte aligned)

QT_HD size_t workspace_bytes(int N, int n_ini) {
    const int m = N > n_ini ? N : n_ini;
    const size_t cap = 4 * (size_t)m + 16;
    size_t b = 2 * cap * sizeof(int) + 2 * cap * sizeof(uint16_t) + 2 * (size_t)(m + 4) * sizeof(NodeB) +
               2 * 3 * (size_t)stack_ranges(m) * sizeof(int) + 2 * (size_t)(m + 8 + 4 * n_ini) * sizeof(int) + (kScCount + 32 + 16 + 4) * sizeof(int);
    return (b + 15) & ~size_t(15);
}
QT_HD void workspace_carve(Workspace& w, void* mem, int N, int n_ini) {
    const int m = N > n_ini ? N : n_ini;
    w.cap = 4 * m + 16;
    w.res_cap = m + 8 + 4 * n_ini;
    w.m = m;
    char* p = (char*)mem;
    w.cnt[0] = (QT_LDS int*)p; p += w.cap * sizeof(int);
    w.cnt[1] = (QT_LDS int*)p; p += w.cap * sizeof(int);
    w.nb[0] = (QT_LDS NodeB*)p; p += (m + 4) * sizeof(NodeB);
    w.nb[1] = (QT_LDS NodeB*)p; p += (m + 4) * sizeof(NodeB);
    w.stack = (QT_LDS int*)p; p += 2 * 3 * stack_ranges(m) * sizeof(int);
    w.ps.stack_half = 3 * stack_ranges(m);
    w.res_seq = (QT_LDS int*)p; p += w.res_cap * sizeof(int);
    w.res_pt = (QT_LDS int*)p; p += w.res_cap * sizeof(int);
    w.sc = (QT_LDS int*)p; p += kScCount * sizeof(int);
    w.scan_tmp = (QT_LDS int*)p; p += 32 * sizeof(int);
    w.ps.scan_tmp = (QT_LDS int*)p; p += 16 * sizeof(int);
    w.ps.sc = (QT_LDS int*)p; p += 4 * sizeof(int);
    w.rankof[0] = (QT_LDS uint16_t*)p; p += w.cap * sizeof(uint16_t);
    w.rankof[1] = (QT_LDS uint16_t*)p;
    w.items = nullptr; w.ps.tmp = nullptr; w.ps.gpos = nullptr; w.ps.lpos = nullptr;  // aliased per sweep, see sort_scratch()
}
// The sort of a careful sweep runs before the other generation's arrays are (re)initialised: its items, the
// stable-finish buffer and the partition position lists live in cnt[np] / rankof[np].
QT_HD void sort_scratch(Workspace& w, QT_LDS int* cnt_np, QT_LDS uint16_t* rankof_np) {
    w.items = (QT_LDS SortItem*)cnt_np;
    w.ps.tmp = w.items + (w.m + 4);
    w.ps.gpos = rankof_np;
    w.ps.lpos = w.ps.gpos + (w.m + 4);
}

#ifndef QT_PASS_BATCH
#define QT_PASS_BATCH 4
#endif
constexpr int kPassBatch = QT_PASS_BATCH;  // points per thread whose LDS stages are issued together in the phased passes

QT_HD int quadrant_of(const Pt& p, const NodeB& b) {  // DivideNode's assignment (:511-525)
    const int mx = b.x0 + ((b.x1 - b.x0 + 1) >> 1), my = b.y0 + ((b.y1 - b.y0 + 1) >> 1);  // ceil(w/2), ceil(h/2)
    return ((int)p.x < mx ? 0 : 1) + ((int)p.y < my ? 0 : 2);
}

// Returns the number of kept candidates; out_pt[i] = candidate index of the i-th keypoint in the reference's
// result order.  `label` is an n-entry scratch array (global memory on the device).
template <int PC, class Ex>
QT_HD int select(Ex& ex, const Pt* pts, int n, Label* label, int W, int H, int N, Workspace& w, int* out_pt,
                 int debug = 0) {
    if (n <= 0) return 0;
    const int tid = ex.tid(), nt = ex.nthreads();
    const int n_ini = (int)roundf((float)W / (float)H);           // :559
    const float hX = (float)W / (float)n_ini;                      // :561
    QT_LDS int* const sc = w.sc;
    ex.mark(9);

    // Point ownership: thread t owns candidates t, t + nt, ...; the first PC of them live in registers for the whole
    // selection (coordinates + label), the rest (only when n > PC * nt) go through global memory.
    // (two packed dwords per point — x | y << 16 and label | score << 16 —: 28 points per thread of the batch form are 56 VGPRs)
    uint32_t cxy[PC > 0 ? PC : 1], cls[PC > 0 ? PC : 1];
#pragma unroll
    for (int k = 0; k < PC; k++) {
        const int p = tid + k * nt;
        const Pt t = pts[p < n ? p : 0];
        cxy[k] = (uint32_t)t.x | ((uint32_t)t.y << 16);
        cls[k] = (kLabelSettled & 0xFFFFu) | ((uint32_t)t.score << 16);
    }
    // Every thread of the workgroup makes the same sequence of body() calls (`valid` says whether the call carries a point), so
    // the bodies may use the wave-collective forms of the counters (Ex::add_runs / Ex::claim).
    auto for_points = [&](auto&& body) {
#pragma unroll
        for (int k = 0; k < PC; k++) {
            if (k * nt >= n) break;                            // workgroup-uniform
            const int p = tid + k * nt;
            const bool valid = p < n;
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "+v"(cxy[k]), "+v"(cls[k]));  // opaque: keeps the unpacked fields from being hoisted out of the generation loop (3 more VGPRs per point)
#endif
            Pt q;
            q.x = (uint16_t)(cxy[k] & 0xFFFFu); q.y = (uint16_t)(cxy[k] >> 16); q.score = (uint16_t)(cls[k] >> 16); q.pad = 0;
            Label lab = (Label)(cls[k] & 0xFFFFu);   // (the register-resident form is 16-bit labels only: PC = 0 in the global-workspace build)
            body(p, valid, q, lab);
            cls[k] = (cls[k] & 0xFFFF0000u) | lab;
        }
        // four candidates in flight per thread: the loads of a batch are issued together, so the pass pays the global
        // latency once per four points instead of once per point (the per-point bodies hold LDS atomics, which keep
        // the compiler from overlapping iterations on its own)
        constexpr int kBatch = kPassBatch;
        for (int b0 = PC * nt; b0 < n; b0 += kBatch * nt) {    // workgroup-uniform bounds
            Pt q[kBatch];
            Label l[kBatch], l0[kBatch];
#pragma unroll
            for (int u = 0; u < kBatch; u++) {
                const int p = b0 + u * nt + tid;
                const int pc = p < n ? p : b0;
                q[u] = pts[pc];
                l0[u] = l[u] = label[pc];
            }
#pragma unroll
            for (int u = 0; u < kBatch; u++) {
                if (b0 + u * nt >= n) break;                   // workgroup-uniform
                const int p = b0 + u * nt + tid;
                const bool valid = p < n;
                body(p, valid, q[u], l[u]);
                if (valid && l[u] != l0[u]) label[p] = l[u];
            }
        }
    };

    // ---- initial columns (:568-601) = generation of parity 0, processing order = ascending column ----
    for (int i = tid; i < n_ini; i += nt) w.cnt[0][i] = 0;
    if (tid == 0) { sc[kScNres] = 0; sc[kScFinish] = 0; sc[kScCareful] = 0; sc[kScGenBase] = 0; }
    ex.sync();
    for_points([&](int, bool valid, const Pt& q, Label& lab) {
        const int c = (int)((float)q.x / hX);                      // vpIniNodes[kp.pt.x/hX]
        if (valid) lab = (Label)c;
        ex.add_runs(w.cnt[0], valid ? c : -1);                     // (candidates arrive in cell order: a wave holds one or two columns)
    });
    ex.sync();
    if (tid == 0) {
        int size = 0, S = 0;
        for (int i = 0; i < n_ini; i++) {
            const int c = w.cnt[0][i];
            w.rankof[0][i] = 0xFFFF;
            if (c > 0) size++;
            if (c > 1) {
                NodeB b;
                b.x0 = (int16_t)(int)(hX * (float)i); b.x1 = (int16_t)(int)(hX * (float)(i + 1));
                b.y0 = 0; b.y1 = (int16_t)H;
                b.mid = node_mid(b.x0, b.x1, b.y0, b.y1); b.slot = (uint16_t)i; b.pad = 0;
                w.rankof[0][i] = (uint16_t)S;
                w.nb[0][S++] = b;
            }
        }
        sc[kScSize] = size; sc[kScS0] = S;
    }
    ex.sync();
    for_points([&](int p, bool valid, const Pt&, Label& lab) {  // single-point columns are final (bNoMore, :590-594)
        const int c = valid ? (int)lab : 0;
        const int t = w.rankof[0][c];          // 0xFFFF <=> exactly one point (the column holds this one)
        const bool settle = valid && t == 0xFFFF;
        const int r = ex.claim(&sc[kScNres], settle);
        if (settle) {
            w.res_seq[r] = -1 - c; w.res_pt[r] = p;
            lab = kLabelSettled;
        } else if (valid) {
            lab = (Label)t;                    // labels name a node by its RANK in its generation's processing order
        }
    });
    ex.sync();

    ex.mark(8);
    if (debug == 2) return 0;
    int par = 0;
    int iter = 0;
    constexpr int kColumnsBase = -0x40000000;
    int par_base = kColumnsBase;  // sequence base of the nodes of generation `par` (workgroup-uniform)
    for (;;) {
        if (debug >= 10 && iter++ >= debug - 10) return 0;  // one iteration = one pass of the main loop (:610-681) or one sweep of the careful loop (:689-753)
        const int np = par ^ 1;
        // this generation's / the next generation's arrays as plain pointers (indexing the pointer arrays with a runtime
        // parity would put them into scratch memory)
        QT_LDS int* const cnt_par = par ? w.cnt[1] : w.cnt[0];
        QT_LDS int* const cnt_np = par ? w.cnt[0] : w.cnt[1];
        QT_LDS uint16_t* const rk_par = par ? w.rankof[1] : w.rankof[0];
        QT_LDS uint16_t* const rk_np = par ? w.rankof[0] : w.rankof[1];
        QT_LDS NodeB* const nb_par = par ? w.nb[1] : w.nb[0];
        QT_LDS NodeB* const nb_np = par ? w.nb[0] : w.nb[1];
        const int S = sc[par ? kScS1 : kScS0];
        const int careful = sc[kScCareful];
        const int prev_size = sc[kScSize];
        if (careful) {
            sort_scratch(w, cnt_np, rk_np);
            // vPrevSizeAndPointerToNode in creation order = descending rank; sort; walk from the back
            for (int i = tid; i < S; i += nt) {
                const NodeB& b = nb_par[S - 1 - i];
                w.items[i].key = ((uint32_t)cnt_par[b.slot] << 16) | (uint32_t)(uint16_t)b.x0;
                w.items[i].node = (uint32_t)(S - 1 - i);
            }
            ex.sync();
            if (debug != 1) ex.sort(w.items, S, w.stack, w.ps);  // std::sort(vPrevSizeAndPointerToNode, compareNodes), :700
            ex.sync();
            // new processing order r: items[S-1-r]; permute nb[par] accordingly (via nb[np] as scratch)
            // (the points' labels still hold the ranks of before the sort: rk_par, free in a careful sweep, maps old -> new)
            for (int r = tid; r < S; r += nt) {
                const int old = (int)w.items[S - 1 - r].node;
                nb_np[r] = nb_par[old];
                rk_par[old] = (uint16_t)r;
            }
            ex.sync();
            for (int r = tid; r < S; r += nt) nb_par[r] = nb_np[r];
            ex.mark(23);
        }
        ex.mark(0);
        for (int i = tid; i < 4 * S; i += nt) cnt_np[i] = 0;
        if (tid == 0) sc[kScNsplit] = S;
        ex.sync();
        ex.mark(1);
        // pass A: children counts of every multi-point node (speculative for the careful sweep)
        if (PC == 0) {
            // phased form: the rank lookups of a batch are issued together, then the node boxes, then the atomics — a
            // per-point body is a chain of three dependent LDS round trips, and nothing else hides them in one instance
            for (int b0 = 0; b0 < n; b0 += kPassBatch * nt) {      // workgroup-uniform bounds: the counters are wave collectives
                const int p0 = b0 + tid;
                Pt q[kPassBatch];
                uint32_t lab[kPassBatch];
                int r[kPassBatch];
                bool act[kPassBatch];
#pragma unroll
                for (int u = 0; u < kPassBatch; u++) {
                    const int p = p0 + u * nt, pc = p < n ? p : b0;
                    q[u] = pts[pc];
                    lab[u] = label[pc];
                    act[u] = p < n && (int)(lab[u] >> kParityShift) == par;   // settled (all ones) has both top bits set
                }
                if (careful) {
#pragma unroll
                    for (int u = 0; u < kPassBatch; u++) r[u] = act[u] ? (int)rk_par[lab[u] & kSlotMask] : 0;
                } else {
#pragma unroll
                    for (int u = 0; u < kPassBatch; u++) r[u] = act[u] ? (lab[u] & kSlotMask) : 0;
                }
                uint32_t mid[kPassBatch];
#pragma unroll
                for (int u = 0; u < kPassBatch; u++) mid[u] = nb_par[r[u]].mid;
#pragma unroll
                for (int u = 0; u < kPassBatch; u++) {
                    if (b0 + u * nt >= n) break;                       // workgroup-uniform
                    const int slot = 4 * r[u] + quadrant_of_mid((uint32_t)q[u].x | ((uint32_t)q[u].y << 16), mid[u]);
                    // candidates arrive in cell order, so the lanes of a wave fall into a handful of nodes: one atomic per run
                    // of equal slots instead of 64 serialised ones on the same LDS address
                    ex.add_runs(cnt_np, act[u] ? slot : -1);
                    if (act[u]) label[p0 + u * nt] = (Label)((par ? kParityBit : 0u) | slot);  // pass B needs neither the point nor the node again
                }
            }
        } else {
        for_points([&](int, bool valid, const Pt& q, Label& lab_) {
            const int lab = lab_;
            const bool act = valid && (int)((uint32_t)lab >> kParityShift) == par;   // settled (all ones) has both top bits set
            const int r = !act ? 0 : careful ? (int)rk_par[lab & kSlotMask] : (lab & kSlotMask);
            const int slot = 4 * r + quadrant_of_mid((uint32_t)q.x | ((uint32_t)q.y << 16), nb_par[r].mid);
            ex.add_runs(cnt_np, act ? slot : -1);
            if (act) lab_ = (Label)((par ? kParityBit : 0u) | slot);
        });
        }
        ex.sync();
        ex.mark(2);
        if (careful) {  // :701-748: split from the largest until the quota is reached -> nsplit
            const int kp = (S + nt - 1) / nt;
            const int pb = tid * kp, pe = pb + kp < S ? pb + kp : S;
            int local = 0;
            for (int r = pb; r < pe; r++) {
                int nch = 0;
                for (int q = 0; q < 4; q++) nch += cnt_np[4 * r + q] > 0;
                local += nch - 1;
            }
            int tot = 0;
            int running = prev_size + ex.excl_scan(local, w.scan_tmp, &tot);
            for (int r = pb; r < pe; r++) {
                int nch = 0;
                for (int q = 0; q < 4; q++) nch += cnt_np[4 * r + q] > 0;
                running += nch - 1;
                if (running >= N) { ex.atomic_min(&sc[kScNsplit], r + 1); break; }
            }
            ex.sync();
        }
        {
            // next generation: children with more than one point, processing order = reverse creation order
            const int nsplit_ = sc[kScNsplit];
            const int kk = (4 * nsplit_ + nt - 1) / nt;
            const int sb = tid * kk, se = sb + kk < 4 * nsplit_ ? sb + kk : 4 * nsplit_;
            int nz = 0, nx = 0;
            for (int i = sb; i < se; i++) { nz += cnt_np[i] > 0; nx += cnt_np[i] > 1; }
            // one scan for both counts (non-empty children in the low half, multi-point children in the high half: <= 4 S < 2^16)
            int TOT = 0;
            int before = ex.excl_scan(nz | (nx << 16), w.scan_tmp, &TOT) >> 16;
            const int NZ = TOT & 0xFFFF, NX = TOT >> 16;
            for (int i = sb; i < se; i++) {
                rk_np[i] = 0xFFFF;
                if (cnt_np[i] > 1) {
                    const NodeB& pb_ = nb_par[i >> 2];
                    const int q = i & 3;
                    const int mx = pb_.x0 + ((pb_.x1 - pb_.x0 + 1) >> 1), my = pb_.y0 + ((pb_.y1 - pb_.y0 + 1) >> 1);
                    NodeB b;
                    b.x0 = (int16_t)((q & 1) ? mx : pb_.x0); b.x1 = (int16_t)((q & 1) ? pb_.x1 : mx);
                    b.y0 = (int16_t)((q & 2) ? my : pb_.y0); b.y1 = (int16_t)((q & 2) ? pb_.y1 : my);
                    b.mid = node_mid(b.x0, b.x1, b.y0, b.y1); b.slot = (uint16_t)i; b.pad = 0;
                    const int rank = NX - 1 - before;  // number of multi-point children created after this one
                    before++;
                    rk_np[i] = (uint16_t)rank;
                    nb_np[rank] = b;
                }
            }
            if (tid == 0) {
                const int size = prev_size - nsplit_ + NZ;
                sc[np ? kScS1 : kScS0] = NX;
                sc[kScNToExpand] = NX;
                sc[kScSize] = size;
                int finish = 0;
                if (size >= N || size == prev_size) finish = 1;                      // :685 / :751
                else if (!careful && size + 3 * NX > N) sc[kScCareful] = 1;          // :689
                sc[kScFinish] = finish;
            }
        }
        ex.sync();
        ex.mark(3);
        const int nsplit = sc[kScNsplit], genbase = sc[kScGenBase];
        // pass B: move the points of split nodes to their child; single-point children are final.  Pass A left the child's slot
        // (4 * rank + quadrant) in the label; rk_np[slot] is the child's rank in the next generation, 0xFFFF for a single point.
        if (PC == 0) {
            for (int b0 = 0; b0 < n; b0 += kPassBatch * nt) {  // phased like pass A
                const int p0 = b0 + tid;
                uint32_t lab[kPassBatch];
                int t[kPassBatch];
                bool act[kPassBatch];
#pragma unroll
                for (int u = 0; u < kPassBatch; u++) {
                    const int p = p0 + u * nt, pc = p < n ? p : b0;
                    lab[u] = label[pc];
                    act[u] = p < n && (int)(lab[u] >> kParityShift) == par;   // settled (all ones) has both top bits set
                }
#pragma unroll
                for (int u = 0; u < kPassBatch; u++) {
                    const int slot = lab[u] & kSlotMask;
                    t[u] = act[u] && (slot >> 2) < nsplit ? (int)rk_np[slot] : 0;
                }
#pragma unroll
                for (int u = 0; u < kPassBatch; u++) {
                    if (b0 + u * nt >= n) break;                       // workgroup-uniform
                    const int p = p0 + u * nt;
                    const int slot = lab[u] & kSlotMask;
                    const bool whole = act[u] && (slot >> 2) >= nsplit;  // careful sweep stopped before this node: it stays whole
                    const bool settle = act[u] && !whole && t[u] == 0xFFFF;
                    const int k = ex.claim(&sc[kScNres], settle);
                    if (settle) {
                        w.res_seq[k] = genbase + slot; w.res_pt[k] = p;
                        label[p] = kLabelSettled;
                    } else if (whole) {
                        label[p] = (Label)((par ? kParityBit : 0u) | (slot >> 2));
                    } else if (act[u]) {
                        label[p] = (Label)((np ? kParityBit : 0u) | t[u]);
                    }
                }
            }
        } else {
        for_points([&](int p, bool valid, const Pt&, Label& lab_) {
            const int lab = lab_;
            const bool act = valid && (int)((uint32_t)lab >> kParityShift) == par;   // settled (all ones) has both top bits set
            const int slot = lab & kSlotMask;
            const bool whole = act && (slot >> 2) >= nsplit;             // stays whole
            const int t = act && !whole ? (int)rk_np[slot] : 0;
            const bool settle = act && !whole && t == 0xFFFF;
            const int k = ex.claim(&sc[kScNres], settle);
            if (settle) {
                w.res_seq[k] = genbase + slot; w.res_pt[k] = p;
                lab_ = kLabelSettled;
            } else if (whole) {
                lab_ = (Label)((par ? kParityBit : 0u) | (slot >> 2));
            } else if (act) {
                lab_ = (Label)((np ? kParityBit : 0u) | t);
            }
        });
        }
        ex.sync();
        ex.mark(4);
        if (tid == 0) sc[kScGenBase] = genbase + 4 * S;
        const int finish = sc[kScFinish];
        ex.sync();
        if (finish) {
            // alive multi-point nodes: unsplit nodes of `par` (ranks nsplit..S) and the new children in `np`
            const int Sn = sc[np ? kScS1 : kScS0];
            for (int i = tid; i < S; i += nt) cnt_par[i] = 0;   // re-used as best-point keys, by rank
            for (int i = tid; i < Sn; i += nt) cnt_np[i] = 0;
            ex.sync();
            for_points([&](int p, bool valid, const Pt& q, Label& lab_) {  // first strictly greater response wins (:757-776)
                const int lab = lab_;
                if (!valid || lab == kLabelSettled) return;
                const int lp = (lab & kParityBit) ? 1 : 0;
                const int r = lab & kSlotMask;
                ex.atomic_max(&(lp == par ? cnt_par : cnt_np)[r], (int)(((uint32_t)q.score << 22) | (uint32_t)(0x3FFFFF - p)));
            });
            ex.sync();
            for (int i0 = 0; i0 < S - nsplit + Sn; i0 += nt) {
                const int i = i0 + tid;
                const bool in = i < S - nsplit + Sn;
                const int lp = i < S - nsplit ? par : np;
                const int r = !in ? 0 : i < S - nsplit ? nsplit + i : i - (S - nsplit);
                const int k = ex.claim(&sc[kScNres], in);
                if (in) {
                    w.res_seq[k] = lp == par ? (par_base == kColumnsBase ? -1 - (int)nb_par[r].slot : par_base + (int)nb_par[r].slot)
                                             : genbase + (int)nb_np[r].slot;
                    w.res_pt[k] = 0x3FFFFF - ((lp == par ? cnt_par : cnt_np)[r] & 0x3FFFFF);
                }
            }
            ex.sync();
            ex.mark(5);
            break;
        }
        par = np;
        par_base = genbase;
    }
    // result order = descending creation sequence: rank sort (all seq are distinct)
    // (splitting a result's comparisons over several threads with an LDS accumulator, as final_stable_sort does, was slower here:
    // 6.6 against 4.8 us for the 434 results of a KITTI level 0 — two parts do not pay for the extra barrier and the atomics)
    const int nres = sc[kScNres];
    // (read eight at a time: the list is padded to a multiple of eight with a value below every sequence number)
    for (int i = nres + tid; i < ((nres + 7) & ~7); i += nt) w.res_seq[i] = -0x7FFFFFFF - 1;
    ex.sync();
    for (int i = tid; i < nres; i += nt) {
        const int s = w.res_seq[i];
        int rank = 0;
        for (int j = 0; j < nres; j += 8) {
            const Int4 v = load_int4(w.res_seq + j), u = load_int4(w.res_seq + j + 4);
            rank += (v.x > s) + (v.y > s) + (v.z > s) + (v.w > s);
            rank += (u.x > s) + (u.y > s) + (u.z > s) + (u.w > s);
        }
        out_pt[rank] = w.res_pt[i];
    }
    ex.sync();
    ex.mark(6);
    return nres;
}

}  // namespace qt
}  // namespace msorb
