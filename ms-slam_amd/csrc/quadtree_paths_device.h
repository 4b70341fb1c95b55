// DistributeOctTree (ORBextractor.cc:555-779) without per-generation point passes: "selection by quadrant path".
//
// quadtree_device.h's select() walks the tree one generation at a time and touches every candidate twice per generation
// (count the children, move to the child): for a KITTI level 0 that is three generations + the careful sweep, 5 800 points
// each, and every pass is a chain of dependent LDS round trips between workgroup barriers.  But the node a candidate falls
// into at generation g is a function of its coordinates alone: DivideNode (:511-525) halves the box with integer arithmetic,
// so every candidate can walk ITS OWN path c, q1, q2, ... (initial column, then one quadrant digit per generation) without
// knowing anything about the other candidates.  With the path as an index,
//   * the number of points per node of EVERY generation is one histogram pass (tables cnt[g][c * 4^g + q1 q2 .. qg]);
//   * the size of the node list after generation g is the number of non-empty entries of table g (a single-point node is
//     never split again, :590-594 / :640-660, and its point stays alone in its prefix at every deeper generation), the
//     number of nodes still to expand is the number of entries > 1: the generation at which the main loop ends (:685) or
//     hands over to the careful loop (:689) follows from per-table counts, no point is touched;
//   * the order of the node list (children are push_front'ed, the list is walked from the front) makes the processing
//     order of generation g the table order with alternate digits reversed — digit j (j >= 1) descending iff g - j is even,
//     the column descending iff g is odd — so a node's processing rank, and with it the creation sequence number of its
//     children (4 * rank + quadrant), is a prefix count over the table in that order;
//   * the careful loop (:689-753) works on NODES (sort by (count, UL.x), split from the largest until the quota is reached):
//     it runs as in select(), except that a child's count is a table lookup instead of a point pass;
//   * what becomes of a candidate — alone in its node at some generation (a finished keypoint, sequence number from its
//     parent's rank) or member of a node that was not split (response maximum per node, :757-776) — is the same for every
//     candidate of one entry of the deepest generation reached, so it is worked out once per ENTRY; the candidates read it.
// A box halves its x and y extents independently (the x split of a node depends on its column and on the x bits of its
// path only), so a candidate's path is xtab[x] | ytab[y]: two small tables built once per instance, and the two point passes
// (histogram of the deepest generation — the shallower ones are sums of children —, final assignment) are a dozen
// instructions per candidate: at 6 000 candidates on one CU every instruction per candidate is 0.05 us.
// Same result as select(), element for element (tests/qt_host_check.cc runs both on every case; on the device the general
// form remains the fallback: select_paths() returns -1, before it has written anything, when the tree grows deeper than the
// tables the workgroup's LDS holds).
#pragma once
#include "quadtree_device.h"

namespace msorb {
namespace qt {

constexpr int kMaxPathGen = 7;

struct PathTables {
    QT_LDS uint16_t* cnt;   // generation g at entry offset n_ini * (4^g - 1) / 3, entry = natural path index c * 4^g + digits
    QT_LDS uint16_t* rank;  // same indexing: processing rank of a multi-point entry within its generation
    QT_LDS int* stat;       // [kMaxPathGen + 1]: non-empty entries | multi-point entries << 16
    QT_LDS int* nsplit;     // [kMaxPathGen + 2]: nodes of generation g that were split (careful sweeps)
    QT_LDS int* base;       // [kMaxPathGen + 2]: creation-sequence base of the nodes living in generation g
    QT_LDS uint32_t* xtab;  // [W + 2]: x -> column << 2 gmax | the x bits of the path's digits (bit 0 of each digit)
    QT_LDS uint32_t* ytab;  // [H + 2]: y -> the y bits (bit 1 of each digit)
    int gmax = -1;          // deepest generation with a table; < 1: the path form is not available
};
QT_HD int path_off(int n_ini, int g) { return n_ini * (((1 << (2 * g)) - 1) / 3); }
// deepest table generation for a quota of N nodes: with evenly spread points the careful loop starts where a generation
// holds more than N / 4 nodes and needs the two generations below it; n_ini * 4^G <= 65535 keeps a path in 16 bits
QT_HD int path_gmax(int N, int n_ini, int g_cap) {
    int G = 1;
    while (G < kMaxPathGen && G < g_cap && (n_ini << (2 * G)) < 4 * N) G++;
    while (G > 0 && (n_ini << (2 * G)) > 65535) G--;
    return G;
}
QT_HD size_t path_tables_bytes(int n_ini, int gmax, int W, int H) {
    if (gmax < 1) return 0;
    const size_t entries = ((size_t)path_off(n_ini, gmax + 1) + 7) & ~size_t(7);
    return 2 * entries * sizeof(uint16_t) + (size_t)(3 * (kMaxPathGen + 2) + 2 + W + 2 + H + 2) * sizeof(int);
}
QT_HD void path_tables_carve(PathTables& t, void* mem, int n_ini, int gmax, int W, int H) {
    t.gmax = gmax;
    if (gmax < 1) return;
    const size_t entries = ((size_t)path_off(n_ini, gmax + 1) + 7) & ~size_t(7);
    char* p = (char*)mem;
    t.cnt = (QT_LDS uint16_t*)p; p += entries * sizeof(uint16_t);
    t.rank = (QT_LDS uint16_t*)p; p += entries * sizeof(uint16_t);
    t.stat = (QT_LDS int*)p; p += (kMaxPathGen + 2) * sizeof(int);
    t.nsplit = (QT_LDS int*)p; p += (kMaxPathGen + 2) * sizeof(int);
    t.base = (QT_LDS int*)p; p += (kMaxPathGen + 2) * sizeof(int);
    t.xtab = (QT_LDS uint32_t*)p; p += (W + 2) * sizeof(int);
    t.ytab = (QT_LDS uint32_t*)p;
}

struct PathBox {   // a node's box while a candidate (or a node builder) walks down
    int x0, x1, y0, y1;
    QT_HD void column(int c, float hX, int H) {   // :568-580
        x0 = (int)(hX * (float)c); x1 = (int)(hX * (float)(c + 1)); y0 = 0; y1 = H;
    }
    QT_HD int step(int x, int y) {                // DivideNode's assignment (:511-525) + the child's box
        const int mx = x0 + ((x1 - x0 + 1) >> 1), my = y0 + ((y1 - y0 + 1) >> 1);
        const int qx = x < mx ? 0 : 1, qy = y < my ? 0 : 1;
        x0 = qx ? mx : x0; x1 = qx ? x1 : mx;
        y0 = qy ? my : y0; y1 = qy ? y1 : my;
        return qx + 2 * qy;
    }
    QT_HD void child(int q) {
        const int mx = x0 + ((x1 - x0 + 1) >> 1), my = y0 + ((y1 - y0 + 1) >> 1);
        x0 = (q & 1) ? mx : x0; x1 = (q & 1) ? x1 : mx;
        y0 = (q & 2) ? my : y0; y1 = (q & 2) ? y1 : my;
    }
};
QT_HD int popc32(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popc(v);
#else
    return __builtin_popcount(v);
#endif
}
// position in generation g's processing order -> natural path index (an involution)
QT_HD int path_of_position(int pos, int g, int n_ini) {
    const int low = (1 << (2 * g)) - 1;
    const int c = pos >> (2 * g);
    return (((g & 1) ? n_ini - 1 - c : c) << (2 * g)) | ((pos & low) ^ (0x33333333 & low));
}

struct PathPassFirst { static constexpr bool value = true; };
struct PathPassAgain { static constexpr bool value = false; };

// Returns the number of kept candidates (out_pt as select()), or -1 = not applicable (nothing written to out_pt).
template <int PC, class Ex>
QT_HD int select_paths(Ex& ex, const Pt* pts, int n, int W, int H, int N, Workspace& w, PathTables& t, int* out_pt) {
    if (n <= 0) return 0;
    const int G = t.gmax;
    if (G < 1 || n > 65535 || w.res_cap >= 0x8000) return -1;
    const int tid = ex.tid(), nt = ex.nthreads();
    const int n_ini = (int)roundf((float)W / (float)H);           // :559
    const float hX = (float)W / (float)n_ini;                      // :561
    QT_LDS int* const sc = w.sc;
    ex.mark(9);

    // candidates: thread t owns t, t + nt, ...; the first PC of them stay in registers (x | y << 16 until the path is known, then
    // the path; response)
    uint32_t cxy[PC > 0 ? PC : 1], csc[PC > 0 ? PC : 1];
#pragma unroll
    for (int k = 0; k < PC; k++) {
        const int p = tid + k * nt;
        const Pt q = pts[p < n ? p : 0];
        cxy[k] = (uint32_t)q.x | ((uint32_t)q.y << 16);
        csc[k] = q.score;
    }
    // every thread of the workgroup makes the same sequence of body() calls (`valid`: the call carries a candidate);
    // FIRST: the register copy still holds the coordinates and body() returns the path that replaces them
    // (a coordinate past the region — never produced by the cell loop — takes the last entry: every split point lies below it)
    auto path_of = [&](uint32_t x, uint32_t y) { return t.xtab[x < (uint32_t)W ? x : (uint32_t)W] | t.ytab[y < (uint32_t)H ? y : (uint32_t)H]; };
    auto for_points = [&](auto first, auto&& body) {
        constexpr bool kFirst = decltype(first)::value;
#pragma unroll
        for (int k = 0; k < PC; k++) {
            if (k * nt >= n) break;                            // workgroup-uniform
            const int p = tid + k * nt;
            if constexpr (kFirst) cxy[k] = body(p, p < n, path_of(cxy[k] & 0xFFFFu, cxy[k] >> 16), 0);
            else body(p, p < n, cxy[k], (int)csc[k]);
        }
        constexpr int kBatch = kPassBatch;
        for (int b0 = PC * nt; b0 < n; b0 += kBatch * nt) {    // workgroup-uniform bounds
            Pt q[kBatch];
#pragma unroll
            for (int u = 0; u < kBatch; u++) {
                const int p = b0 + u * nt + tid;
                q[u] = pts[p < n ? p : b0];
            }
#pragma unroll
            for (int u = 0; u < kBatch; u++) {
                if (b0 + u * nt >= n) break;                   // workgroup-uniform
                const int p = b0 + u * nt + tid;
                body(p, p < n, path_of(q[u].x, q[u].y), (int)q[u].score);
            }
        }
    };
    using First = PathPassFirst;
    using Again = PathPassAgain;

    // ---- 1. coordinate -> path tables; histogram of the deepest generation; the shallower ones by summing children ----
    {
        const int words = (path_off(n_ini, G + 1) + 1) >> 1;
        QT_LDS int* const z = (QT_LDS int*)t.cnt;
        for (int i = tid; i < words; i += nt) z[i] = 0;
        if (tid <= G) t.stat[tid] = 0;
        if (tid == 0) sc[kScNres] = 0;
        for (int x = tid; x <= W; x += nt) {                       // the x extent of the boxes a candidate at x falls into: column (:586), then halves (:511-525)
            int c = (int)((float)x / hX);                          // vpIniNodes[kp.pt.x/hX]
            c = c < n_ini ? c : n_ini - 1;
            int x0 = (int)(hX * (float)c), x1 = (int)(hX * (float)(c + 1));
            uint32_t v = (uint32_t)c << (2 * G);
            for (int g = 1; g <= G; g++) {
                const int mx = x0 + ((x1 - x0 + 1) >> 1);
                const int qx = x < mx ? 0 : 1;
                x0 = qx ? mx : x0; x1 = qx ? x1 : mx;
                v |= (uint32_t)qx << (2 * (G - g));
            }
            t.xtab[x] = v;
        }
        for (int y = tid; y <= H; y += nt) {
            int y0 = 0, y1 = H;
            uint32_t v = 0;
            for (int g = 1; g <= G; g++) {
                const int my = y0 + ((y1 - y0 + 1) >> 1);
                const int qy = y < my ? 0 : 1;
                y0 = qy ? my : y0; y1 = qy ? y1 : my;
                v |= (uint32_t)qy << (2 * (G - g) + 1);
            }
            t.ytab[y] = v;
        }
    }
    ex.sync();
    ex.mark(10);
    {
        const int off_G = path_off(n_ini, G);
        for_points(First{}, [&](int, bool valid, uint32_t path, int) {
            if (valid) ex.add16(t.cnt, off_G + (int)path);         // (entry index from the 4-byte aligned table base: two counts share a word)
            return path;
        });
    }
    ex.sync();
    for (int g = G - 1; g >= 0; g--) {   // a node's count = the sum of its children's; nodes / nodes to expand of generation g + 1 on the way
        QT_LDS const uint16_t* const dn = t.cnt + path_off(n_ini, g + 1);
        QT_LDS uint16_t* const up = t.cnt + path_off(n_ini, g);
        const int e = n_ini << (2 * g);
        int local = 0;
        for (int i = tid; i < e; i += nt) {
            const int a = dn[4 * i], b = dn[4 * i + 1], c = dn[4 * i + 2], d = dn[4 * i + 3];
            up[i] = (uint16_t)(a + b + c + d);
            local += (a > 0) + (b > 0) + (c > 0) + (d > 0) + (((a > 1) + (b > 1) + (c > 1) + (d > 1)) << 16);
        }
        ex.wave_sum_add(&t.stat[g + 1], local, e);
        ex.sync();
    }
    {
        int local = 0;
        for (int i = tid; i < n_ini; i += nt) { const int c = t.cnt[i]; local += (c > 0) + ((c > 1) << 16); }
        ex.wave_sum_add(&t.stat[0], local, n_ini);
    }
    ex.sync();
    ex.mark(8);
    // the main loop (:610-687) replayed on the per-generation counts, by every thread alike
    int F = -1, Gc = -1;       // generation of the final nodes when the main loop ends by itself / first generation of the careful loop
    {
        int prev_nz = t.stat[0] & 0xFFFF;
        for (int g = 0;; g++) {
            if (g + 1 > G) return -1;
            const int st = t.stat[g + 1];
            const int nz = st & 0xFFFF, nx = st >> 16;
            if (nz >= N || nz == prev_nz) { F = g + 1; break; }       // :685
            if (nz + 3 * nx > N) { Gc = g + 1; break; }               // :689
            prev_nz = nz;
        }
    }
    const int K = (Gc >= 0 ? Gc : F) + 1;   // generations 0 .. K-1 take their processing ranks from the table order

    // ---- 3. processing ranks of the multi-point nodes of generations 0 .. K-1: one prefix count over all of them ----
    {
        const int len = path_off(n_ini, K);
        const int per = (len + nt - 1) / nt;
        const int pb = tid * per, pe = pb + per < len ? pb + per : len;
        int local = 0;
        for (int i = pb; i < pe; i++) {
            int g = 0;
            while (i >= path_off(n_ini, g + 1)) g++;
            const int idx = path_of_position(i - path_off(n_ini, g), g, n_ini);
            local += t.cnt[path_off(n_ini, g) + idx] > 1;
        }
        int total = 0;
        int before = ex.excl_scan(local, w.scan_tmp, &total);
        for (int i = pb; i < pe; i++) {
            int g = 0, earlier = 0;
            while (i >= path_off(n_ini, g + 1)) { earlier += t.stat[g] >> 16; g++; }
            const int idx = path_of_position(i - path_off(n_ini, g), g, n_ini);
            if (t.cnt[path_off(n_ini, g) + idx] > 1) { t.rank[path_off(n_ini, g) + idx] = (uint16_t)(before - earlier); before++; }
        }
        if (tid == 0) {
            int b = 0;
            t.base[0] = 0; t.base[1] = 0;
            for (int g = 1; g < K; g++) { b += 4 * (t.stat[g - 1] >> 16); t.base[g + 1] = b; }   // children of generation g live in g + 1
        }
    }
    ex.sync();
    ex.mark(0);

    // ---- 4. the careful loop (:689-753) on nodes ----
    int L = -1;          // last generation a careful sweep processed
    int S_last = 0;      // its number of nodes, of which nsplit[L] were split
    int par = 0;
    if (Gc >= 0) {
        // nodes of generation Gc by processing rank
        {
            const int g = Gc, e = n_ini << (2 * g);
            QT_LDS const uint16_t* const tab = t.cnt + path_off(n_ini, g);
            QT_LDS const uint16_t* const rk = t.rank + path_off(n_ini, g);
            QT_LDS const uint16_t* const rk_up = t.rank + path_off(n_ini, g - 1);
            for (int i = tid; i < e; i += nt) {
                if (tab[i] <= 1) continue;
                PathBox bx;
                bx.column(i >> (2 * g), hX, H);
                for (int j = 1; j <= g; j++) bx.child((i >> (2 * (g - j))) & 3);
                NodeB b;
                b.x0 = (int16_t)bx.x0; b.x1 = (int16_t)bx.x1; b.y0 = (int16_t)bx.y0; b.y1 = (int16_t)bx.y1;
                b.mid = node_mid(bx.x0, bx.x1, bx.y0, bx.y1);
                b.slot = (uint16_t)(4 * (int)rk_up[i >> 2] + (i & 3));
                b.pad = (uint16_t)i;   // the node's path
                w.nb[0][rk[i]] = b;
            }
            if (tid == 0) { sc[kScS0] = t.stat[g] >> 16; sc[kScSize] = t.stat[g] & 0xFFFF; }
        }
        ex.sync();
        for (int g = Gc;; g++) {
            if (g + 1 > G) return -1;
            const int np = par ^ 1;
            QT_LDS int* const cnt_np = par ? w.cnt[0] : w.cnt[1];
            QT_LDS uint16_t* const rk_np = par ? w.rankof[0] : w.rankof[1];
            QT_LDS NodeB* const nb_par = par ? w.nb[1] : w.nb[0];
            QT_LDS NodeB* const nb_np = par ? w.nb[0] : w.nb[1];
            QT_LDS const uint16_t* const tab = t.cnt + path_off(n_ini, g);
            QT_LDS const uint16_t* const tab_dn = t.cnt + path_off(n_ini, g + 1);
            QT_LDS uint16_t* const rk = t.rank + path_off(n_ini, g);
            QT_LDS uint16_t* const rk_dn = t.rank + path_off(n_ini, g + 1);
            const int S = sc[par ? kScS1 : kScS0];
            const int prev_size = sc[kScSize];
            sort_scratch(w, cnt_np, rk_np);
            // vPrevSizeAndPointerToNode in creation order = descending rank; sort; walk from the back
            for (int i = tid; i < S; i += nt) {
                const NodeB& b = nb_par[S - 1 - i];
                w.items[i].key = ((uint32_t)tab[b.pad] << 16) | (uint32_t)(uint16_t)b.x0;
                w.items[i].node = (uint32_t)(S - 1 - i);
            }
            ex.sync();
            ex.sort(w.items, S, w.stack, w.ps);  // std::sort(vPrevSizeAndPointerToNode, compareNodes), :700
            ex.sync();
            for (int r = tid; r < S; r += nt) nb_np[r] = nb_par[(int)w.items[S - 1 - r].node];   // new processing order r: items[S-1-r]
            ex.sync();
            for (int r = tid; r < S; r += nt) { const NodeB b = nb_np[r]; nb_par[r] = b; rk[b.pad] = (uint16_t)r; }
            if (tid == 0) sc[kScNsplit] = S;
            ex.sync();
            ex.mark(23);
            // children counts of every node (speculative: the sweep may stop before the last one)
            for (int i = tid; i < 4 * S; i += nt) cnt_np[i] = tab_dn[4 * (int)nb_par[i >> 2].pad + (i & 3)];
            ex.sync();
            {   // :701-748: split from the largest until the quota is reached -> nsplit
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
            {   // next generation: children with more than one point, processing order = reverse creation order
                const int nsplit_ = sc[kScNsplit];
                const int kk = (4 * nsplit_ + nt - 1) / nt;
                const int sb = tid * kk, se = sb + kk < 4 * nsplit_ ? sb + kk : 4 * nsplit_;
                int nz = 0, nx = 0;
                for (int i = sb; i < se; i++) { nz += cnt_np[i] > 0; nx += cnt_np[i] > 1; }
                int TOT = 0;
                int before = ex.excl_scan(nz | (nx << 16), w.scan_tmp, &TOT) >> 16;
                const int NZ = TOT & 0xFFFF, NX = TOT >> 16;
                for (int i = sb; i < se; i++) {
                    if (cnt_np[i] > 1) {
                        const NodeB& pb_ = nb_par[i >> 2];
                        const int q = i & 3;
                        const int mx = pb_.x0 + ((pb_.x1 - pb_.x0 + 1) >> 1), my = pb_.y0 + ((pb_.y1 - pb_.y0 + 1) >> 1);
                        NodeB b;
                        b.x0 = (int16_t)((q & 1) ? mx : pb_.x0); b.x1 = (int16_t)((q & 1) ? pb_.x1 : mx);
                        b.y0 = (int16_t)((q & 2) ? my : pb_.y0); b.y1 = (int16_t)((q & 2) ? pb_.y1 : my);
                        b.mid = node_mid(b.x0, b.x1, b.y0, b.y1); b.slot = (uint16_t)i;
                        b.pad = (uint16_t)(4 * (int)pb_.pad + q);
                        const int rank = NX - 1 - before;  // number of multi-point children created after this one
                        before++;
                        rk_dn[b.pad] = (uint16_t)rank;
                        nb_np[rank] = b;
                    }
                }
                if (tid == 0) {
                    const int size = prev_size - nsplit_ + NZ;
                    sc[np ? kScS1 : kScS0] = NX;
                    sc[kScSize] = size;
                    sc[kScFinish] = (size >= N || size == prev_size) ? 1 : 0;             // :751
                    t.nsplit[g] = nsplit_;
                    t.base[g + 2] = t.base[g + 1] + 4 * S;
                }
            }
            ex.sync();
            if (sc[kScFinish]) { L = g; S_last = S; break; }
            par = np;
        }
    }
    ex.mark(3);

    // ---- 5. what becomes of the candidates of each entry of the deepest generation reached, and where in the output ----
    // The output order is descending creation sequence (the node list from its front).  Sequence numbers are distinct integers
    // below base[last_gen + 1]: every result sets its bit in a bitmap, a prefix population count over the words turns a sequence
    // number into its output position — no pairwise comparison of the results (434 x 434 for a KITTI level 0: 5 us).
    const int nsplit_L = L >= 0 ? t.nsplit[L] : 0;
    const int n_unsplit = L >= 0 ? S_last - nsplit_L : 0;
    const int last_gen = L >= 0 ? L + 1 : F;                       // generation of the deepest final nodes
    const int n_deep = L >= 0 ? sc[(par ^ 1) ? kScS1 : kScS0] : (t.stat[F] >> 16);
    const int n_nodes = n_unsplit + n_deep;
    const int R = t.base[last_gen + 1] + n_ini;                    // sequence numbers + n_ini lie in [0, R)
    const int words = (R + 31) >> 5;
    if (R >= 0x8000 || 2 * words > w.cap || n_nodes > w.res_cap) return -1;
    QT_LDS int* const best = par ? w.cnt[0] : w.cnt[1];           // (cnt_np of the last sweep: its counts are used up)
    QT_LDS int* const bits = par ? w.cnt[1] : w.cnt[0];           // [words] bitmap, [words] bits below each word
    QT_LDS int* const below = bits + words;
    QT_LDS int* const node_pos = w.res_seq;                        // a final node's sequence number, then its output position
    QT_LDS const uint16_t* const tab_last = t.cnt + path_off(n_ini, last_gen);
    QT_LDS uint16_t* const verdict = t.rank + path_off(n_ini, last_gen);
    const int e_last = n_ini << (2 * last_gen);
    for (int i = tid; i < n_nodes; i += nt) best[i] = 0;
    for (int i = tid; i < words; i += nt) bits[i] = 0;
    ex.sync();
    // Walk from the column down to the entry: alone at some generation -> a result with that node's sequence number; a node that
    // was not split -> its ordinal among the final nodes, | 0x8000.  The verdict replaces the entry's rank (only ranks of OTHER
    // generations are read on the way, its own one before it is overwritten).
    for (int i = tid; i < e_last; i += nt) {
        if (tab_last[i] == 0) continue;
        const int c = i >> (2 * last_gen);
        int v;
        if (t.cnt[c] == 1) {
            v = n_ini - 1 - c;                                     // a single-point column (:590-594): sequence -1 - c
            ex.atomic_or(&bits[v >> 5], 1 << (v & 31));
        } else {
            int seq = 0;                                           // of the node the walk stands on (generation >= 1)
            for (int g = 0;; g++) {
                const int a = i >> (2 * (last_gen - g));
                const int r = t.rank[path_off(n_ini, g) + a];
                const bool split = g < last_gen && (g != L || r < nsplit_L);
                if (!split) {   // (g >= 1: a column with several points is always split)
                    const int node = g == last_gen ? n_unsplit + r : r - nsplit_L;
                    node_pos[node] = seq;                          // (every entry below the node writes the same value)
                    ex.atomic_or(&bits[seq >> 5], 1 << (seq & 31));
                    v = 0x8000 | node;
                    break;
                }
                const int ch = i >> (2 * (last_gen - g - 1));
                seq = t.base[g + 1] + 4 * r + (ch & 3) + n_ini;
                if (t.cnt[path_off(n_ini, g + 1) + ch] == 1) { v = seq; ex.atomic_or(&bits[seq >> 5], 1 << (seq & 31)); break; }
            }
        }
        verdict[i] = (uint16_t)v;
    }
    ex.sync();
    ex.mark(4);
    int nres = 0;
    {
        const int per = (words + nt - 1) / nt;
        const int wb = tid * per, we = wb + per < words ? wb + per : words;
        int local = 0;
        for (int i = wb; i < we; i++) local += popc32((uint32_t)bits[i]);
        int before = ex.excl_scan(local, w.scan_tmp, &nres);
        for (int i = wb; i < we; i++) { below[i] = before; before += popc32((uint32_t)bits[i]); }
    }
    ex.sync();
    auto position = [&](int seq) {   // results with a larger sequence number come first
        return nres - 1 - (below[seq >> 5] + popc32((uint32_t)bits[seq >> 5] & ((1u << (seq & 31)) - 1u)));
    };
    for (int i = tid; i < e_last; i += nt) {
        if (tab_last[i] == 0) continue;
        const int v = verdict[i];
        if (!(v & 0x8000)) verdict[i] = (uint16_t)position(v);
    }
    for (int i = tid; i < n_nodes; i += nt) node_pos[i] = position(node_pos[i]);
    ex.sync();
    {
        const int sh = 2 * (G - last_gen);
        for_points(Again{}, [&](int p, bool valid, uint32_t path, int score) {
            if (!valid) return 0u;
            const int v = verdict[path >> sh];
            if (v & 0x8000)   // first strictly greater response wins (:757-776): max over (response, -candidate index)
                ex.atomic_max(&best[v & 0x7FFF], (int)(((uint32_t)score << 22) | (uint32_t)(0x3FFFFF - p)));
            else
                out_pt[v] = p;
            return 0u;
        });
    }
    ex.sync();
    ex.mark(5);
    for (int i = tid; i < n_nodes; i += nt) out_pt[node_pos[i]] = 0x3FFFFF - (best[i] & 0x3FFFFF);
    ex.sync();
    ex.mark(6);
    return nres;
}

}  // namespace qt
}  // namespace msorb
