// msorb_visibility_csr through the C ABI on a synthetic sparsification window (BASELINE configs[4]: 30 window keyframes x 2000
// slots, half of them tracked, 6000 map points with ~8 observations, 100 keyframes outside the window): median wall time.
// Build: g++ -O2 -std=c++17 tools/visibility_bench.cc -Iinclude -Lms-slam_amd -lmsorb -o /tmp/visibility_bench
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

#include "msorb.h"

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    const int n_window = 30, n_outside = 100, n_points = 6000, slots = 2000, n_kf = n_window + n_outside;
    unsigned s = 4242;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
    std::vector<uint8_t> in_window(n_kf, 0);
    std::vector<int> window_ids;
    while ((int)window_ids.size() < n_window) { const int k = rnd() % n_kf; if (!in_window[k]) { in_window[k] = 1; window_ids.push_back(k); } }
    std::sort(window_ids.begin(), window_ids.end());
    std::vector<std::set<int>> obs(n_points);
    std::vector<int> kf_slot_begin{0}, slot_point, slot_cell;
    for (int k : window_ids) {
        std::vector<int> cells(slots);
        for (auto& c : cells) c = rnd() % (64 * 48);
        std::sort(cells.begin(), cells.end());
        for (int i = 0; i < slots; i++) {
            const bool tracked = rnd() % 2;
            const int p = rnd() % n_points;
            if (tracked && !obs[p].count(k)) { obs[p].insert(k); slot_point.push_back(p % 20 == 0 ? -1 : p); }
            else slot_point.push_back(-1);
            slot_cell.push_back(cells[i]);
        }
        kf_slot_begin.push_back((int)slot_point.size());
    }
    for (int p = 0; p < n_points; p++) {
        const int extra = rnd() % 16;
        for (int e = 0; e < extra; e++) { const int k = rnd() % n_kf; if (!in_window[k]) obs[p].insert(k); }
    }
    std::vector<int> obs_begin{0}, obs_kf, point_nobs(n_points), kf_num_mps(n_kf);
    for (int p = 0; p < n_points; p++) { for (int k : obs[p]) obs_kf.push_back(k); obs_begin.push_back((int)obs_kf.size()); point_nobs[p] = (int)obs[p].size(); }
    for (auto& v : kf_num_mps) v = 200 + rnd() % 1300;
    const int S = (int)slot_point.size();
    const int cap_cols = S + 1, cap_rows = S + n_kf + 1, cap_nnz = 2 * S + (int)obs_kf.size() + 1;
    std::vector<int> col_point(cap_cols), row_begin(cap_rows + 1), row_kind(cap_rows), row_owner(cap_rows), col_idx(cap_nnz);
    std::vector<float> row_rhs(cap_rows), obj(cap_cols);
    int n_cols = 0, n_rows = 0, nnz = 0, nmax = 0;
    std::vector<double> t;
    for (int i = 0; i < 5 + iters; i++) {
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = msorb_visibility_csr(0, n_window, kf_slot_begin.data(), slot_point.data(), slot_cell.data(), n_points, point_nobs.data(),
                                            obs_begin.data(), obs_kf.data(), n_kf, in_window.data(), kf_num_mps.data(), 100, 0, &n_cols,
                                            col_point.data(), cap_cols, &n_rows, row_begin.data(), row_kind.data(), row_owner.data(),
                                            row_rhs.data(), cap_rows, col_idx.data(), cap_nnz, &nnz, obj.data(), &nmax);
        if (rc) { printf("msorb_visibility_csr: %d %s\n", rc, msorb_last_error()); return 1; }
        if (i >= 5) t.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    std::sort(t.begin(), t.end());
    printf("{\"ms_visibility_csr_window30\": %.4f, \"p10\": %.4f, \"p90\": %.4f, \"slots\": %d, \"observations\": %zu, \"cols\": %d, \"rows\": %d, \"nnz\": %d}\n",
           t[t.size() / 2], t[t.size() / 10], t[t.size() * 9 / 10], S, obs_kf.size(), n_cols, n_rows, nnz);
    return 0;
}
