// Per-frame latency of a stereo pair the way Frame.cc:122-125 drives the extractor: two host threads, one handle per
// eye, msorb_extract on host images.  Build: g++ -O2 -std=c++17 tools/latency_pair.cc -Iinclude -Lms-slam_amd -lmsorb
// -Wl,-rpath,$PWD/ms-slam_amd -lpthread -o /tmp/latency_pair ; prints median wall time per pair (both eyes done).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "msorb.h"

static void synth(std::vector<uint8_t>& img, int rows, int cols, unsigned seed) {
    img.resize((size_t)rows * cols);
    unsigned s = seed * 2654435761u + 12345u;
    for (auto& p : img) { s = s * 1664525u + 1013904223u; p = (uint8_t)(96 + ((s >> 24) & 63)); }
    for (int k = 0; k < 400; k++) {  // bright / dark rectangles: corners for FAST
        s = s * 1664525u + 1013904223u; const int x = (s >> 8) % (cols - 40);
        s = s * 1664525u + 1013904223u; const int y = (s >> 8) % (rows - 40);
        s = s * 1664525u + 1013904223u; const int w = 6 + (s >> 8) % 30, h = 6 + (s >> 16) % 30;
        const uint8_t v = (s & 1) ? 220 : 20;
        for (int yy = y; yy < y + h; yy++) for (int xx = x; xx < x + w; xx++) img[(size_t)yy * cols + xx] = v;
    }
}

int main(int argc, char** argv) {
    const int rows = 376, cols = 1241, iters = argc > 1 ? atoi(argv[1]) : 300;
    msorb_extractor* ex[2];
    for (auto& e : ex) if (msorb_extractor_create(2000, 1.2f, 8, 20, 7, 0, &e)) { printf("create: %s\n", msorb_last_error()); return 1; }
    if (getenv("LAT_SERIAL_BLUR")) for (auto& e : ex) msorb_extractor_set_overlap(e, 2, 0);  // experiment: blur on the main stream
    std::vector<uint8_t> img[2];
    synth(img[0], rows, cols, 1); synth(img[1], rows, cols, 2);
    const int cap = 2000 + 3 * 8 + 64;
    std::vector<msorb_keypoint> kps[2] = {std::vector<msorb_keypoint>(cap), std::vector<msorb_keypoint>(cap)};
    std::vector<uint8_t> desc[2] = {std::vector<uint8_t>((size_t)cap * 32), std::vector<uint8_t>((size_t)cap * 32)};
    int n[2] = {0, 0}, mono[2];
    auto eye = [&](int e) { msorb_extract(ex[e], img[e].data(), rows, cols, cols, 0, 0, kps[e].data(), desc[e].data(), cap, &n[e], &mono[e]); };
    for (int i = 0; i < 5; i++) { eye(0); eye(1); }
    std::vector<double> pair_ms, single_ms;
    for (int i = 0; i < iters; i++) {
        const auto t0 = std::chrono::steady_clock::now();
        std::thread a(eye, 0), b(eye, 1);  // fresh threads per frame, like the reference
        a.join(); b.join();
        pair_ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    for (int i = 0; i < iters; i++) {
        const auto t0 = std::chrono::steady_clock::now();
        eye(0);
        single_ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    std::sort(pair_ms.begin(), pair_ms.end()); std::sort(single_ms.begin(), single_ms.end());
    // the rest of a tracking frame (BASELINE.json configs[2]): stereo matches, frame upload (grid), SearchLocalPoints'
    // isInFrustum-style table of 4096 map points + SearchByProjection — all through the C ABI, host arrays in and out
    float scale[8];
    msorb_extractor_tables(ex[0], scale, nullptr, nullptr, nullptr, nullptr);
    const float mbf = 386.1448f, mb = mbf / 718.856f;
    std::vector<float> ur(cap), depth(cap);
    msorb_frame* fr = nullptr;
    msorb_frame_create(0, &fr);
    const int M = 4096;
    std::vector<uint8_t> inView(M, 1), bad(M, 0), spars(M, 0), mdesc((size_t)M * 32);
    std::vector<float> px(M), py(M), pxr(M), mdepth(M, 10.f), vcos(M, 0.999f);
    std::vector<int> level(M), obs(M, 3), frameMp(cap);
    std::vector<double> t_st, t_fs, t_sp, t_all, t_fr;
    unsigned s = 777;
    int nm = 0, oob = 0;
    for (int i = 0; i < iters / 3; i++) {
        const auto t0 = std::chrono::steady_clock::now();
        std::thread a(eye, 0), b(eye, 1);
        a.join(); b.join();
        const auto t1 = std::chrono::steady_clock::now();
        msorb_stereo_matches(ex[0], ex[1], kps[0].data(), n[0], desc[0].data(), kps[1].data(), n[1], desc[1].data(), mb, mbf,
                             ur.data(), depth.data(), &oob);
        const auto t2 = std::chrono::steady_clock::now();
        msorb_frame_set(fr, kps[0].data(), n[0], desc[0].data(), ur.data(), 0.f, (float)cols, 0.f, (float)rows, scale, 8);
        const auto t3 = std::chrono::steady_clock::now();
        for (int m = 0; m < M; m++) {  // map points = noisy copies of frame features (not timed)
            s = s * 1664525u + 1013904223u;
            const int src = (s >> 8) % n[0];
            memcpy(&mdesc[(size_t)m * 32], &desc[0][(size_t)src * 32], 32);
            s = s * 1664525u + 1013904223u;
            for (int f = 0; f < (int)((s >> 20) % 24); f++) { s = s * 1664525u + 1013904223u; mdesc[(size_t)m * 32 + ((s >> 8) & 31)] ^= (uint8_t)(1u << ((s >> 16) & 7)); }
            px[m] = kps[0][src].x + (float)((int)((s >> 4) % 7) - 3);
            py[m] = kps[0][src].y + (float)((int)((s >> 9) % 7) - 3);
            pxr[m] = ur[src] > 0 ? ur[src] : px[m] - 20.f;
            level[m] = kps[0][src].octave;
        }
        std::fill(frameMp.begin(), frameMp.end(), -1);
        // isInFrustum pre-pass over the same number of local map points (identity pose; results not used by the search
        // below, which keeps the table built above: this only times the call)
        {
            static std::vector<float> pw(3 * M), nr(3 * M), mxd(M, 40.f), mnd(M, 2.f), o_f(5 * M);
            static std::vector<uint8_t> o_v(M);
            static std::vector<int> o_l(M);
            for (int m = 0; m < M; m++) { pw[3 * m] = (px[m] - 607.f) * 10.f / 718.f; pw[3 * m + 1] = (py[m] - 185.f) * 10.f / 718.f; pw[3 * m + 2] = 10.f; nr[3 * m] = 0; nr[3 * m + 1] = 0; nr[3 * m + 2] = 1; }
            msorb_frustum F{};
            F.Rcw[0] = F.Rcw[4] = F.Rcw[8] = 1.f; F.fx = F.fy = 718.856f; F.cx = 607.19f; F.cy = 185.2f;
            F.min_x = 0; F.max_x = (float)cols; F.min_y = 0; F.max_y = (float)rows; F.mbf = mbf; F.log_scale_factor = 0.18232f; F.n_scale_levels = 8;
            const auto f0 = std::chrono::steady_clock::now();
            msorb_is_in_frustum(0, &F, 0.5f, M, pw.data(), nr.data(), mxd.data(), mnd.data(), o_v.data(), o_f.data(), o_f.data() + M,
                                o_f.data() + 2 * M, o_f.data() + 3 * M, o_l.data(), o_f.data() + 4 * M, nullptr);
            t_fr.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - f0).count());
        }
        const auto t4 = std::chrono::steady_clock::now();
        msorb_search_by_projection_mps(fr, M, inView.data(), bad.data(), spars.data(), px.data(), py.data(), pxr.data(), mdepth.data(),
                                       level.data(), vcos.data(), mdesc.data(), obs.data(), frameMp.data(), 3.0f, 0, 50.f, 0.8f, &nm);
        const auto t5 = std::chrono::steady_clock::now();
        auto ms = [](auto a_, auto b_) { return std::chrono::duration<double, std::milli>(b_ - a_).count(); };
        t_st.push_back(ms(t1, t2)); t_fs.push_back(ms(t2, t3)); t_sp.push_back(ms(t4, t5));
        t_all.push_back(ms(t0, t3) + ms(t4, t5) + t_fr.back());
    }
    msorb_frame_destroy(fr);
    auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    // the fused call: both eyes + stereo association, one synchronisation (msorb_extract_stereo)
    std::vector<double> t_fused;
    {
        msorb_extractor* fx = nullptr;
        if (msorb_extractor_create(2000, 1.2f, 8, 20, 7, 0, &fx)) { printf("create: %s\n", msorb_last_error()); return 1; }
        if (getenv("LAT_SERIAL_BLUR")) msorb_extractor_set_overlap(fx, 2, 0);
        int nl = 0, nr2 = 0, oob2 = 0;
        for (int i = 0; i < 5 + iters; i++) {
            const auto t0 = std::chrono::steady_clock::now();
            if (msorb_extract_stereo(fx, img[0].data(), img[1].data(), rows, cols, cols, cols, mb, mbf, kps[0].data(), desc[0].data(), &nl,
                                     kps[1].data(), desc[1].data(), &nr2, cap, ur.data(), depth.data(), &oob2)) {
                printf("extract_stereo: %s\n", msorb_last_error());
                return 1;
            }
            if (i >= 5) t_fused.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        }
        msorb_extractor_destroy(fx);
    }
    // the same tracking frame as ONE device-resident chain (csrc/track.hip): (a) msorb_extract_stereo_frame + msorb_search_local_points
    // (two calls, two synchronisations: the pose estimate of TrackWithMotionModel sits between them in the reference), (b)
    // msorb_track_frontend (one call, one synchronisation) — local map of M points on the viewing rays of the keypoints
    std::vector<double> t_esf, t_slp, t_one;
    int nm_chain = 0, nm_one = 0, rounds = 0;
    {
        msorb_extractor* fx = nullptr;
        msorb_frame* f2 = nullptr;
        if (msorb_extractor_create(2000, 1.2f, 8, 20, 7, 0, &fx) || msorb_frame_create(0, &f2)) { printf("create: %s\n", msorb_last_error()); return 1; }
        int nl = 0, nr2 = 0, oob2 = 0;
        msorb_extract_stereo(fx, img[0].data(), img[1].data(), rows, cols, cols, cols, mb, mbf, kps[0].data(), desc[0].data(), &nl,
                             kps[1].data(), desc[1].data(), &nr2, cap, ur.data(), depth.data(), &oob2);
        const float fxc = 718.856f, cxc = 607.19f, cyc = 185.2f;
        std::vector<float> pw(3 * M), nrm(3 * M), mxd(M), mnd(M);
        std::vector<uint8_t> visit(M, 1);
        for (int m = 0; m < M; m++) {
            s = s * 1664525u + 1013904223u;
            const int src = (s >> 8) % nl;
            memcpy(&mdesc[(size_t)m * 32], &desc[0][(size_t)src * 32], 32);
            s = s * 1664525u + 1013904223u;
            for (int f = 0; f < (int)((s >> 20) % 24); f++) { s = s * 1664525u + 1013904223u; mdesc[(size_t)m * 32 + ((s >> 8) & 31)] ^= (uint8_t)(1u << ((s >> 16) & 7)); }
            const float z = depth[src] > 0 ? depth[src] : 10.f;
            const float u = kps[0][src].x + (float)((int)((s >> 4) % 5) - 2), v = kps[0][src].y + (float)((int)((s >> 9) % 5) - 2);
            const float X = (u - cxc) * z / fxc, Y = (v - cyc) * z / fxc, d = std::sqrt(X * X + Y * Y + z * z);
            pw[3 * m] = X; pw[3 * m + 1] = Y; pw[3 * m + 2] = z;
            nrm[3 * m] = X / d; nrm[3 * m + 1] = Y / d; nrm[3 * m + 2] = z / d;
            mxd[m] = d * scale[kps[0][src].octave] * 0.97f; mnd[m] = mxd[m] / scale[7];
        }
        msorb_frustum F{};
        F.Rcw[0] = F.Rcw[4] = F.Rcw[8] = 1.f; F.fx = F.fy = fxc; F.cx = cxc; F.cy = cyc;
        F.min_x = 0; F.max_x = (float)cols; F.min_y = 0; F.max_y = (float)rows; F.mbf = mbf; F.log_scale_factor = std::log(1.2f); F.n_scale_levels = 8;
        std::vector<uint8_t> o_v(M);
        std::vector<float> o_f(5 * M);
        std::vector<int> o_l(M);
        for (int i = 0; i < 5 + iters; i++) {
            const auto t0 = std::chrono::steady_clock::now();
            if (msorb_extract_stereo_frame(fx, f2, img[0].data(), img[1].data(), rows, cols, cols, cols, mb, mbf, kps[0].data(), desc[0].data(),
                                           &nl, kps[1].data(), desc[1].data(), &nr2, cap, ur.data(), depth.data(), &oob2, 0.f, (float)cols, 0.f,
                                           (float)rows)) { printf("extract_stereo_frame: %s\n", msorb_last_error()); return 1; }
            const auto t1 = std::chrono::steady_clock::now();
            std::fill(frameMp.begin(), frameMp.end(), -1);
            if (msorb_search_local_points(f2, &F, 0.5f, M, pw.data(), nrm.data(), mxd.data(), mnd.data(), visit.data(), bad.data(), spars.data(),
                                          mdesc.data(), obs.data(), frameMp.data(), 1.0f, 0, 50.f, 0.8f, o_v.data(), o_f.data(), o_f.data() + M,
                                          o_f.data() + 2 * M, o_f.data() + 3 * M, o_l.data(), o_f.data() + 4 * M, &nm_chain)) { printf("search_local_points: %s\n", msorb_last_error()); return 1; }
            const auto t2 = std::chrono::steady_clock::now();
            if (i >= 5) { t_esf.push_back(std::chrono::duration<double, std::milli>(t1 - t0).count()); t_slp.push_back(std::chrono::duration<double, std::milli>(t2 - t1).count()); }
        }
        for (int i = 0; i < 5 + iters; i++) {
            const auto t0 = std::chrono::steady_clock::now();
            if (msorb_track_frontend(fx, f2, img[0].data(), img[1].data(), rows, cols, cols, cols, mb, mbf, kps[0].data(), desc[0].data(), &nl,
                                     kps[1].data(), desc[1].data(), &nr2, cap, ur.data(), depth.data(), &oob2, 0.f, (float)cols, 0.f, (float)rows,
                                     &F, 0.5f, M, pw.data(), nrm.data(), mxd.data(), mnd.data(), visit.data(), bad.data(), spars.data(), mdesc.data(),
                                     obs.data(), frameMp.data(), 1.0f, 0, 50.f, 0.8f, o_v.data(), o_f.data(), o_f.data() + M, o_f.data() + 2 * M,
                                     o_f.data() + 3 * M, o_l.data(), o_f.data() + 4 * M, &nm_one, &rounds)) { printf("track_frontend: %s\n", msorb_last_error()); return 1; }
            if (i >= 5) t_one.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        }
        msorb_frame_destroy(f2);
        msorb_extractor_destroy(fx);
    }
    char chain[512];
    snprintf(chain, sizeof chain, "{\"map_points\": %d, \"ms_extract_stereo_frame\": %.4f, \"ms_search_local_points\": %.4f, "
             "\"ms_two_calls\": %.4f, \"matches_two_calls\": %d, \"ms_track_frontend_one_call\": %.4f, \"matches_one_call\": %d, "
             "\"window_rounds\": %d}",
             M, med(t_esf), med(t_slp), med(t_esf) + med(t_slp), nm_chain, med(t_one), nm_one, rounds);
    printf("{\"tracking_chain\": %s, \"keypoints\": [%d, %d], \"ms_stereo_pair_two_threads_median\": %.4f, \"ms_single_image_median\": %.4f, "
           "\"ms_stereo_matches\": %.4f, \"ms_frame_grid_upload\": %.4f, \"ms_search_by_projection_4096\": %.4f, "
           "\"ms_is_in_frustum_4096\": %.4f, \"ms_tracking_frame_front_end\": %.4f, \"projection_matches\": %d, "
           "\"ms_extract_stereo_fused\": %.4f}\n",
           chain, n[0], n[1], pair_ms[iters / 2], single_ms[iters / 2], med(t_st), med(t_fs), med(t_sp), med(t_fr), med(t_all), nm, med(t_fused));
    for (auto& e : ex) msorb_extractor_destroy(e);
    return 0;
}
