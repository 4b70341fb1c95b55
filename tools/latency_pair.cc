// Per-frame latency of a stereo pair the way Frame.cc:122-125 drives the extractor: two host threads, one handle per
// eye, msorb_extract on host images.  Build: g++ -O2 -std=c++17 tools/latency_pair.cc -Iinclude -Lms-slam_amd -lmsorb
// -Wl,-rpath,$PWD/ms-slam_amd -lpthread -o /tmp/latency_pair ; prints median wall time per pair (both eyes done).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
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
    printf("{\"keypoints\": [%d, %d], \"ms_stereo_pair_two_threads_median\": %.4f, \"ms_single_image_median\": %.4f}\n", n[0], n[1],
           pair_ms[iters / 2], single_ms[iters / 2]);
    for (auto& e : ex) msorb_extractor_destroy(e);
    return 0;
}
