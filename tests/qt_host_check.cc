// Host execution (1 "thread") of the device quadtree algorithm in ms-slam_amd/csrc/quadtree_device.h, compared
// with the product's host quadtree (orb_host.cc, itself parity-tested against the oracle's std::list
// restatement of ORBextractor.cc:555-779) on many random and structured candidate sets; also checks the
// libstdc++ std::sort restatement against std::sort.  Exit code 0 = all equal.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../ms-slam_amd/csrc/orb_host.h"
#include "../ms-slam_amd/csrc/quadtree_device.h"
#include "../ms-slam_amd/csrc/quadtree_paths_device.h"

using namespace msorb;

struct HostEx {
    static constexpr bool kSplitRank = true;
    int tid() const { return 0; }
    int nthreads() const { return 1; }
    void sync() {}
    void mark(int) {}
    int atomic_add(int* p, int v) { const int o = *p; *p = o + v; return o; }
    void add_runs(int* arr, int idx) { if (idx >= 0) arr[idx]++; }
    void add16(uint16_t* arr, int idx) { arr[idx]++; }
    void wave_sum_add(int* p, int v, int) { *p += v; }
    int claim(int* ctr, bool pred) { return pred ? (*ctr)++ : 0; }
    void atomic_max(int* p, int v) { if (v > *p) *p = v; }
    void atomic_or(int* p, int v) { *p |= v; }
    void atomic_min(int* p, int v) { if (v < *p) *p = v; }
    int excl_scan(int v, int*, int* total) { *total = v; return 0; }
    int excl_count(bool p, int* total) { *total = p; return 0; }
    void sort(qt::SortItem* v, int n, int* stack, qt::ParScratch&) { qt::lsort(v, n, stack); }
};

static int g_path_cap = qt::kMaxPathGen, g_path_fallbacks = 0, g_path_runs = 0;
static int run_case(const std::vector<Cand16>& c, int W, int H, int N, bool verbose) {
    std::vector<int> kept;
    distribute_quadtree(c.data(), (int)c.size(), 16, 16 + W, 16, 16 + H, N, kept);
    const int n_ini = (int)roundf((float)W / (float)H);
    std::vector<char> mem(qt::workspace_bytes(N, n_ini) + 64);
    qt::Workspace w;
    qt::workspace_carve(w, mem.data(), N, n_ini);
    std::vector<uint16_t> label(c.size() + 1);
    std::vector<int> out(c.size() + 8);
    HostEx ex;
    std::vector<int> out2(c.size() + 8);
    const int n = qt::select<0>(ex, reinterpret_cast<const qt::Pt*>(c.data()), (int)c.size(), label.data(), W, H, N, w, out.data());
    // register-cached variant (what the GPU instantiates): the first 16 points of the single host 'thread' are cached
    const int n2 = qt::select<16>(ex, reinterpret_cast<const qt::Pt*>(c.data()), (int)c.size(), label.data(), W, H, N, w, out2.data());
    if (n2 != n) return 1;
    for (int i = 0; i < n; i++) if (out[i] != out2[i]) return 1;
    // selection by quadrant path (what single frames run on the GPU; -1 = tree deeper than its tables: the general form runs)
    for (int pc : {0, 16}) {
        const int g_cap = g_path_cap;
        const int gmax = qt::path_gmax(N, n_ini, g_cap);
        std::vector<char> tmem(qt::path_tables_bytes(n_ini, gmax, W, H) + 64);
        qt::PathTables pt;
        qt::path_tables_carve(pt, tmem.data(), n_ini, gmax, W, H);
        std::vector<int> out3(c.size() + 8, -7);
        const int n3 = pc ? qt::select_paths<16>(ex, reinterpret_cast<const qt::Pt*>(c.data()), (int)c.size(), W, H, N, w, pt, out3.data())
                          : qt::select_paths<0>(ex, reinterpret_cast<const qt::Pt*>(c.data()), (int)c.size(), W, H, N, w, pt, out3.data());
        if (n3 < 0) { g_path_fallbacks++; continue; }
        g_path_runs++;
        bool same = n3 == n;
        for (int i = 0; same && i < n; i++) same = out[i] == out3[i];
        if (!same) {
            if (verbose) fprintf(stderr, "PATH MISMATCH n=%zu W=%d H=%d N=%d got=%d want=%d (pc %d)\n", c.size(), W, H, N, n3, n, pc);
            return 1;
        }
    }
    bool ok = n == (int)kept.size();
    for (int i = 0; ok && i < n; i++) ok = out[i] == kept[i];
    if (!ok && verbose) fprintf(stderr, "MISMATCH n=%zu W=%d H=%d N=%d got=%d want=%zu\n", c.size(), W, H, N, n, kept.size());
    return ok ? 0 : 1;
}

int main(int argc, char** argv) {
    const int trials = argc > 1 ? atoi(argv[1]) : 3000;
    if (argc > 3) g_path_cap = atoi(argv[3]);
    if (argc > 2 && argv[2][0]) {   // candidate sets from a file (records: int32 W, H, N, n; n x {u16 x, y, score, pad}) instead of the random ones
        FILE* f = fopen(argv[2], "rb");
        if (!f) { perror(argv[2]); return 2; }
        int hdr[4], bad = 0, total = 0;
        while (fread(hdr, sizeof(int), 4, f) == 4) {
            std::vector<Cand16> c(hdr[3]);
            if (hdr[3] && fread(c.data(), sizeof(Cand16), c.size(), f) != c.size()) return 2;
            bad += run_case(c, hdr[0], hdr[1], hdr[2], bad < 5);
            total++;
        }
        fclose(f);
        printf("cases=%d bad=%d path_runs=%d path_fallbacks=%d\n", total, bad, g_path_runs, g_path_fallbacks);
        return bad != 0;
    }
    std::mt19937 rng(12345);
    int bad = 0, total = 0;
    // 1. sort restatement
    for (int t = 0; t < 3000; t++) {
        const int n = 1 + rng() % (t % 40 == 0 ? 2500 : 400);
        std::vector<qt::SortItem> a(n);
        const int kc = 1 + rng() % 10, kx = 1 + rng() % 30;
        for (int i = 0; i < n; i++) a[i] = {(uint32_t)(((2 + rng() % kc) << 16) | ((rng() % kx) * 9)), (uint32_t)i};
        std::vector<qt::SortItem> b = a;
        std::sort(b.begin(), b.end(), [](const qt::SortItem& x, const qt::SortItem& y) { return x.key < y.key; });
        int stack[3 * 64];
        std::vector<qt::SortItem> a2 = a;
        qt::lsort(a.data(), n, stack);
        for (int i = 0; i < n; i++) if (a[i].node != b[i].node) { bad++; break; }
        total++;
        // data-parallel formulation (what the GPU runs), executed by one host "thread"
        std::vector<uint16_t> gp(n + 1), lp(n + 1);
        std::vector<qt::SortItem> tmp(n + 1);
        int scan_tmp[16], scs[4];
        qt::ParScratch ps{gp.data(), lp.data(), tmp.data(), scan_tmp, scs};
        HostEx hex;
        qt::lsort_par(hex, a2.data(), n, stack, ps);
        for (int i = 0; i < n; i++) if (a2[i].node != b[i].node) { bad++; if (bad < 4) fprintf(stderr, "lsort_par mismatch n=%d\n", n); break; }
        total++;
    }
    // 2. selection
    const int geoms[][2] = {{1209, 344}, {1002, 281}, {720, 448}, {314, 73}, {208, 288}, {768, 368}, {500, 500}};
    for (int t = 0; t < trials; t++) {
        const int* g = geoms[rng() % 7];
        const int W = g[0], H = g[1];
        const int mode = rng() % 6;
        int n = mode == 0 ? rng() % 30 : mode == 1 ? 5000 + rng() % 15000 : 200 + rng() % 6000;
        const int N = (rng() % 8 == 0) ? 1 + rng() % 40 : 60 + rng() % 420;
        std::vector<Cand16> c;
        if (mode == 2) {          // clustered
            const int k = 3 + rng() % 20;
            std::vector<int> cx(k), cy(k);
            for (int i = 0; i < k; i++) { cx[i] = rng() % W; cy[i] = rng() % H; }
            for (int i = 0; i < n; i++) {
                const int j = rng() % k;
                const int x = std::min(W - 4, std::max(3, cx[j] + (int)(rng() % 41) - 20)), y = std::min(H - 4, std::max(3, cy[j] + (int)(rng() % 41) - 20));
                c.push_back({(uint16_t)x, (uint16_t)y, (uint16_t)(7 + rng() % 100), 0});
            }
        } else if (mode == 3) {   // regular grid, constant score (ties everywhere)
            const int st = 2 + rng() % 6;
            for (int y = 3; y < H - 3; y += st) for (int x = 3; x < W - 3; x += st) c.push_back({(uint16_t)x, (uint16_t)y, 20, 0});
        } else if (mode == 4) {   // duplicates on a few pixels
            for (int i = 0; i < n; i++) c.push_back({(uint16_t)(3 + (rng() % 5) * 50), (uint16_t)(3 + (rng() % 4) * 30), (uint16_t)(7 + rng() % 3), 0});
        } else {
            for (int i = 0; i < n; i++) c.push_back({(uint16_t)(3 + rng() % (W - 6)), (uint16_t)(3 + rng() % (H - 6)), (uint16_t)(7 + rng() % 200), 0});
        }
        // reference order: cell-row-major then scan order is irrelevant for the algorithm's contract; keep as is
        bad += run_case(c, W, H, N, bad < 5);
        total++;
    }
    printf("cases=%d bad=%d path_runs=%d path_fallbacks=%d\n", total, bad, g_path_runs, g_path_fallbacks);
    return bad != 0;
}
