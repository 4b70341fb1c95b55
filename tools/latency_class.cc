// Per-frame latency THROUGH the drop-in class ORB_SLAM3::ORBextractor (ms-slam_amd/host), i.e. what an unchanged
// Frame.cc sees: two fresh host threads per frame, one extractor object per eye, operator() on cv::Mat images
// (Frame.cc:122-125, 418-425), with mvImagePyramid populated for the host-side ComputeStereoMatches (default) or not
// (MSORB_HOST_PYRAMID=0, device-side stereo matching); then the one-call forms msorb_host::ExtractStereo /
// ExtractStereoSplit.  Build (cv stand-in of the tests; inside MS-SLAM the real OpenCV):
//   g++ -O2 -std=c++17 -Itests/cv_stub -Ims-slam_amd/host -Iinclude tools/latency_class.cc ms-slam_amd/host/ORBextractor.cc \
//       -Lms-slam_amd -lmsorb -Wl,-rpath,$PWD/ms-slam_amd -lpthread -o /tmp/latency_class
// usage: latency_class [iters]   (run once per MSORB_HOST_PYRAMID setting; prints one JSON line)
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "ORBextractor.h"
#include "ORBmatcher_device.h"

namespace ORB_SLAM3 {
struct Frame {
    std::vector<cv::KeyPoint> mvKeys, mvKeysRight;
    cv::Mat mDescriptors, mDescriptorsRight;
    std::vector<float> mvuRight, mvDepth;
    float mb = 0, mbf = 0;
};
}  // namespace ORB_SLAM3

static void synth(std::vector<unsigned char>& img, int rows, int cols, unsigned seed) {
    img.resize((size_t)rows * cols);
    unsigned s = seed * 2654435761u + 12345u;
    for (auto& p : img) { s = s * 1664525u + 1013904223u; p = (unsigned char)(96 + ((s >> 24) & 63)); }
    for (int k = 0; k < 400; k++) {
        s = s * 1664525u + 1013904223u; const int x = (s >> 8) % (cols - 40);
        s = s * 1664525u + 1013904223u; const int y = (s >> 8) % (rows - 40);
        s = s * 1664525u + 1013904223u; const int w = 6 + (s >> 8) % 30, h = 6 + (s >> 16) % 30;
        const unsigned char v = (s & 1) ? 220 : 20;
        for (int yy = y; yy < y + h; yy++) for (int xx = x; xx < x + w; xx++) img[(size_t)yy * cols + xx] = v;
    }
}

int main(int argc, char** argv) {
    using namespace ORB_SLAM3;
    using clk = std::chrono::steady_clock;
    const int rows = 376, cols = 1241, iters = argc > 1 ? atoi(argv[1]) : 300;
    std::vector<unsigned char> bl, br;
    synth(bl, rows, cols, 1); synth(br, rows, cols, 2);
    cv::Mat imL(rows, cols, CV_8UC1, bl.data(), (size_t)cols), imR(rows, cols, CV_8UC1, br.data(), (size_t)cols);
    ORBextractor* exL = new ORBextractor(2000, 1.2f, 8, 20, 7);   // Tracking.cc:595-596
    ORBextractor* exR = new ORBextractor(2000, 1.2f, 8, 20, 7);
    Frame F;
    F.mbf = 386.1448f; F.mb = F.mbf / 718.856f;
    std::vector<int> lap = {0, 0};
    auto eye = [&](int e) {
        if (e == 0) (*exL)(imL, cv::Mat(), F.mvKeys, F.mDescriptors, lap);
        else (*exR)(imR, cv::Mat(), F.mvKeysRight, F.mDescriptorsRight, lap);
    };
    auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    for (int i = 0; i < 5; i++) { eye(0); eye(1); }
    std::vector<double> t_pair, t_single, t_match, t_fused, t_split;
    unsigned long long pyr_sum = 0;
    for (int i = 0; i < iters; i++) {
        const auto t0 = clk::now();
        std::thread a(eye, 0), b(eye, 1);   // Frame.cc:122-125
        a.join(); b.join();
        const auto t1 = clk::now();
        msorb_host::ComputeStereoMatches(F, *exL, *exR);   // device-side Frame::ComputeStereoMatches
        const auto t2 = clk::now();
        t_pair.push_back(ms(t0, t1)); t_match.push_back(ms(t1, t2));
        if (!exL->mvImagePyramid.empty() && exL->mvImagePyramid[7].rows) pyr_sum += exL->mvImagePyramid[7].ptr<unsigned char>(50)[100];
    }
    for (int i = 0; i < iters; i++) { const auto t0 = clk::now(); eye(0); t_single.push_back(ms(t0, clk::now())); }
    for (int i = 0; i < iters + 5; i++) {
        const auto t0 = clk::now();
        msorb_host::ExtractStereo(F, *exL, imL, imR);
        if (i >= 5) t_fused.push_back(ms(t0, clk::now()));
    }
    for (int i = 0; i < iters + 5; i++) {
        const auto t0 = clk::now();
        msorb_host::ExtractStereoSplit(F, *exL, *exR, imL, imR);
        if (i >= 5) t_split.push_back(ms(t0, clk::now()));
    }
    printf("{\"through\": \"ORB_SLAM3::ORBextractor (drop-in class)\", \"host_pyramid\": %s, \"devices\": [%d, %d], "
           "\"keypoints\": [%zu, %zu], \"ms_stereo_pair_two_threads\": %.4f, \"ms_single_image\": %.4f, "
           "\"ms_compute_stereo_matches_device\": %.4f, \"ms_extract_stereo_one_call\": %.4f, "
           "\"ms_extract_stereo_split_one_call\": %.4f, \"pyr_probe\": %llu}\n",
           exL->mvImagePyramid.empty() || !exL->mvImagePyramid[7].rows ? "false" : "true", exL->device(), exR->device(),
           F.mvKeys.size(), F.mvKeysRight.size(), med(t_pair), med(t_single), med(t_match), med(t_fused), med(t_split), pyr_sum);
    return 0;
}
