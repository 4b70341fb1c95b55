// Wall time of the BoW-node searches through the C ABI, per-call staging (msorb_search_by_bow / msorb_search_for_triangulation)
// against resident KeyFrames (msorb_kf_store + msorb_search_by_bow_kf / msorb_search_for_triangulation_kf): a relocalisation
// batch of 32 candidate KeyFrames against one frame and one CreateNewMapPoints pass against 16 neighbours, 2000 features a
// side, 100 vocabulary nodes.  build: g++ -O2 -std=c++17 tools/kf_store_bench.cc -Iinclude -Lms-slam_amd -lmsorb
//   -Wl,-rpath,$PWD/ms-slam_amd -o /tmp/kf_store_bench ; prints one JSON line (medians over 50 calls).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#include "msorb.h"

struct Side {
    int n = 2000;
    std::vector<uint8_t> desc, flag, stereo;
    std::vector<msorb_keypoint> kps;
    std::vector<int> node, begin, feat;
    std::vector<float> angle;
};
static unsigned rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return s >> 8; }
static Side make_side(unsigned seed, const Side* like, int nodes) {
    Side S;
    unsigned s = seed;
    S.desc.resize((size_t)S.n * 32); S.flag.assign(S.n, 1); S.stereo.assign(S.n, 0); S.kps.resize(S.n); S.angle.resize(S.n);
    std::vector<int> node_of(S.n);
    for (int i = 0; i < S.n; i++) {
        if (like) {
            memcpy(&S.desc[(size_t)i * 32], &like->desc[(size_t)i * 32], 32);
            for (int f = 0; f < (int)(rnd(s) % 20); f++) S.desc[(size_t)i * 32 + rnd(s) % 32] ^= (uint8_t)(1u << (rnd(s) % 8));
            node_of[i] = -1;
        } else {
            for (int b = 0; b < 32; b++) S.desc[(size_t)i * 32 + b] = (uint8_t)rnd(s);
        }
        S.kps[i] = msorb_keypoint{(float)(rnd(s) % 1200), (float)(rnd(s) % 370), 31.f, (float)(rnd(s) % 360), 50.f, (int)(rnd(s) % 8), -1};
        S.angle[i] = S.kps[i].angle;
        S.flag[i] = rnd(s) % 5 != 0;
        S.stereo[i] = rnd(s) % 2;
    }
    // node = hash of the first descriptor byte pair of the SOURCE feature, so that copies share their node
    std::vector<std::vector<int>> lists(nodes);
    for (int i = 0; i < S.n; i++) {
        const uint8_t* d = like ? &like->desc[(size_t)i * 32] : &S.desc[(size_t)i * 32];
        lists[(d[0] * 131 + d[1]) % nodes].push_back(i);
    }
    S.begin.push_back(0);
    for (int r = 0; r < nodes; r++) {
        if (lists[r].empty()) continue;
        S.node.push_back(3 * r + 7);
        for (int i : lists[r]) S.feat.push_back(i);
        S.begin.push_back((int)S.feat.size());
    }
    return S;
}
template <class F>
static double median_ms(int reps, F f) {
    std::vector<double> t;
    for (int i = 0; i < reps; i++) {
        const auto t0 = std::chrono::steady_clock::now();
        f();
        t.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main() {
    const int K = 32, NB = 16, nodes = 100;
    const Side frame = make_side(1, nullptr, nodes);
    std::vector<Side> kf;
    for (int k = 0; k < K; k++) kf.push_back(make_side(100 + k, &frame, nodes));
    float scale[8], sigma2[8];
    for (int l = 0; l < 8; l++) { scale[l] = l ? scale[l - 1] * 1.2f : 1.0f; sigma2[l] = scale[l] * scale[l]; }
    msorb_kf_store* st = nullptr;
    if (msorb_kf_store_create(0, &st)) { printf("store: %s\n", msorb_last_error()); return 1; }
    std::vector<int> ids(K);
    int fid = -1;
    for (int k = 0; k < K; k++)
        if (msorb_kf_store_add(st, kf[k].n, kf[k].kps.data(), kf[k].desc.data(), (int)kf[k].node.size(), kf[k].node.data(), kf[k].begin.data(),
                               kf[k].feat.data(), scale, sigma2, 8, &ids[k])) { printf("add: %s\n", msorb_last_error()); return 1; }
    msorb_kf_store_add(st, frame.n, frame.kps.data(), frame.desc.data(), (int)frame.node.size(), frame.node.data(), frame.begin.data(),
                       frame.feat.data(), scale, sigma2, 8, &fid);
    std::vector<std::vector<int>> m12(K, std::vector<int>(2000)), m21(K, std::vector<int>(2000));
    // ---- SearchByBoW: 32 candidate KeyFrames against one frame
    std::vector<msorb_bow_pair> pc(K);
    std::vector<msorb_bow_kf_pair> pr(K);
    for (int k = 0; k < K; k++) {
        msorb_bow_pair& P = pc[k];
        P = msorb_bow_pair{};
        P.n1 = kf[k].n; P.n2 = frame.n; P.desc1 = kf[k].desc.data(); P.desc2 = frame.desc.data(); P.valid1 = kf[k].flag.data(); P.avail2 = nullptr;
        P.fv1_nodes = (int)kf[k].node.size(); P.fv1_node = kf[k].node.data(); P.fv1_begin = kf[k].begin.data(); P.fv1_feat = kf[k].feat.data();
        P.fv2_nodes = (int)frame.node.size(); P.fv2_node = frame.node.data(); P.fv2_begin = frame.begin.data(); P.fv2_feat = frame.feat.data();
        P.angle1 = kf[k].angle.data(); P.angle2 = frame.angle.data(); P.match12 = m12[k].data(); P.match21 = m21[k].data();
        pr[k] = msorb_bow_kf_pair{ids[k], -1, kf[k].flag.data(), nullptr, m12[k].data(), m21[k].data(), 0};
    }
    const msorb_bow_frame bf{frame.n, frame.desc.data(), (int)frame.node.size(), frame.node.data(), frame.begin.data(), frame.feat.data(), frame.angle.data()};
    float kms = 0, kms_r = 0;
    msorb_search_by_bow(0, pc.data(), K, 50, 1, 0.7f, 1, &kms);
    const int nm_call = pc[0].nmatches;
    msorb_search_by_bow_kf(st, pr.data(), K, &bf, 50, 1, 0.7f, 1, &kms_r);
    const int nm_res = pr[0].nmatches;
    const double w_call = median_ms(50, [&] { msorb_search_by_bow(0, pc.data(), K, 50, 1, 0.7f, 1, &kms); });
    const double w_res = median_ms(50, [&] { msorb_search_by_bow_kf(st, pr.data(), K, &bf, 50, 1, 0.7f, 1, nullptr); });   // as production calls it: no timing events
    // ---- SearchForTriangulation: KeyFrame 0 against 16 neighbours
    std::vector<msorb_triangulation_pair> tc(NB);
    std::vector<msorb_triangulation_kf_pair> tr(NB);
    for (int k = 0; k < NB; k++) {
        msorb_triangulation_pair& P = tc[k];
        P = msorb_triangulation_pair{};
        const Side &A = kf[0], &B = kf[k + 1];
        P.n1 = A.n; P.n2 = B.n; P.desc1 = A.desc.data(); P.desc2 = B.desc.data(); P.valid1 = A.flag.data(); P.avail2 = B.flag.data();
        P.stereo1 = A.stereo.data(); P.stereo2 = B.stereo.data();
        P.fv1_nodes = (int)A.node.size(); P.fv1_node = A.node.data(); P.fv1_begin = A.begin.data(); P.fv1_feat = A.feat.data();
        P.fv2_nodes = (int)B.node.size(); P.fv2_node = B.node.data(); P.fv2_begin = B.begin.data(); P.fv2_feat = B.feat.data();
        P.kp1 = A.kps.data(); P.kp2 = B.kps.data(); P.scale_factors2 = scale; P.level_sigma2_2 = sigma2; P.n_levels2 = 8;
        const float F[9] = {0, -1e-5f, 2e-3f, 1e-5f, 0, -3e-3f, -2e-3f, 3e-3f, 0.1f};
        memcpy(P.F12, F, sizeof(F)); P.ep[0] = 600; P.ep[1] = 180; P.match12 = m12[k].data();
        msorb_triangulation_kf_pair& Q = tr[k];
        Q = msorb_triangulation_kf_pair{};
        Q.kf1 = ids[0]; Q.kf2 = ids[k + 1]; Q.valid1 = A.flag.data(); Q.avail2 = B.flag.data(); Q.stereo1 = A.stereo.data(); Q.stereo2 = B.stereo.data();
        memcpy(Q.F12, F, sizeof(F)); Q.ep[0] = 600; Q.ep[1] = 180; Q.match12 = m21[k].data();
    }
    float tms = 0, tms_r = 0;
    msorb_search_for_triangulation(0, tc.data(), NB, 1, 1, &tms);
    msorb_search_for_triangulation_kf(st, tr.data(), NB, 1, 1, &tms_r);
    int same = nm_call == nm_res;
    for (int k = 0; k < NB; k++) same = same && tc[k].nmatches == tr[k].nmatches && !memcmp(m12[k].data(), m21[k].data(), 2000 * 4);
    const double t_call = median_ms(50, [&] { msorb_search_for_triangulation(0, tc.data(), NB, 1, 1, &tms); });
    const double t_res = median_ms(50, [&] { msorb_search_for_triangulation_kf(st, tr.data(), NB, 1, 1, nullptr); });
    printf("{\"search_by_bow_batch32\": {\"per_call\": {\"wall_ms\": %.4f, \"kernel_ms\": %.4f}, \"resident\": {\"wall_ms\": %.4f, \"kernel_ms\": %.4f, "
           "\"wall_over_kernel\": %.2f}, \"matches_first\": %d}, \"search_for_triangulation_neighbours16\": {\"per_call\": {\"wall_ms\": %.4f, "
           "\"kernel_ms\": %.4f}, \"resident\": {\"wall_ms\": %.4f, \"kernel_ms\": %.4f, \"wall_over_kernel\": %.2f}, \"matches_first\": %d}, "
           "\"resident_equals_per_call\": %s}\n",
           w_call, kms, w_res, kms_r, w_res / kms_r, nm_res, t_call, tms, t_res, tms_r, t_res / tms_r, tr[0].nmatches, same ? "true" : "false");
    msorb_kf_store_destroy(st);
    return 0;
}
