// A tracking frame's front-end through the UNCHANGED call sites of MS-SLAM — what `north_star` means by "drops into
// Tracking.cc / LocalMapping.cc unchanged" — timed call by call:
//
//   (1) Frame::Frame(imLeft, imRight, ...)   Frame.cc:119-137   two std::threads, one ORB_SLAM3::ORBextractor::operator() each
//                                                               (host pyramid ON: mvImagePyramid comes back, because an unchanged
//                                                               Frame::ComputeStereoMatches reads it on the host, Frame.cc:750,840-855)
//   (2) TrackWithMotionModel                  Tracking.cc:2833-2870  ORBmatcher(0.9, true).SearchByProjection(mCurrentFrame, mLastFrame,
//                                                               th, bMono) — the class method's default, HOST-projected path
//                                                               (host/ORBmatcher.cc -> msorb_search_by_projection_frames)
//   (3) SearchLocalPoints                     Tracking.cc:3343-3388  ORBmatcher(0.8).SearchByProjection(mCurrentFrame, mvpLocalMapPoints,
//                                                               th, bFarPoints, thFarPoints) after the host's own isInFrustum loop
//
// The classes are ms-slam_amd/host/ORBextractor.{h,cc} and ORBmatcher.{h,cc} compiled here against the stand-in Frame /
// MapPoint / cv types of tests/slam_stub + tests/cv_stub (the image has no OpenCV / Eigen); inside MS-SLAM they compile against
// the real headers.  What stays the reference's HOST code with unchanged callers is NOT timed here and is fed from the scene
// file: Frame::ComputeStereoMatches (mvuRight / mvDepth), Frame::isInFrustum (the MapPoint scratch), both PoseOptimizations.
//
//   g++ -O2 -std=c++17 -Itests/slam_stub -Itests/cv_stub -Ims-slam_amd/host -Iinclude tools/unchanged_callers.cc
//     ms-slam_amd/host/ORBextractor.cc ms-slam_amd/host/ORBmatcher.cc -Lms-slam_amd -lmsorb -lpthread -o tools/_unchanged_callers
// usage: _unchanged_callers <scene.bin> <out.bin> [iters]
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "ORBextractor.h"
#include "ORBmatcher.h"
#include "ORBmatcher_device.h"

using namespace ORB_SLAM3;
typedef std::shared_ptr<MapPoint> MP;

template <class T>
static std::vector<T> rd(FILE* f, size_t n) {
    std::vector<T> v(n);
    if (n && fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "unchanged_callers: short read\n"); exit(3); }
    return v;
}
template <class T>
static void wr(FILE* f, const std::vector<T>& v) { if (!v.empty()) fwrite(v.data(), sizeof(T), v.size(), f); }
static void wri(FILE* f, int v) { fwrite(&v, 4, 1, f); }
static void wrd(FILE* f, double v) { fwrite(&v, 8, 1, f); }

int main(int argc, char** argv) {
    using clk = std::chrono::steady_clock;
    if (argc < 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 3;
    const auto hi = rd<int>(f, 6);      // rows cols nfeatures nlevels NL M
    const int rows = hi[0], cols = hi[1], nfeat = hi[2], nlev = hi[3], NL = hi[4], M = hi[5];
    const auto hf = rd<float>(f, 13);   // scaleFactor fx fy cx cy mb mbf th_mm th_lp minx maxx miny maxy
    const int iters = argc > 3 ? atoi(argv[3]) : 60;
    auto imgL = rd<unsigned char>(f, (size_t)rows * cols), imgR = rd<unsigned char>(f, (size_t)rows * cols);
    const int N_expected = rd<int>(f, 1)[0];
    const auto ur_in = rd<float>(f, N_expected), depth_in = rd<float>(f, N_expected);
    const auto Rcw = rd<float>(f, 9), tcw = rd<float>(f, 3), Rlw = rd<float>(f, 9), tlw = rd<float>(f, 3);
    const auto lk = rd<cv::KeyPoint>(f, NL);
    const auto has = rd<unsigned char>(f, NL), outl = rd<unsigned char>(f, NL);
    const auto Xw = rd<float>(f, (size_t)3 * NL);
    const auto lobs = rd<int>(f, NL);
    const auto ldesc = rd<unsigned char>(f, (size_t)32 * NL);
    const auto in_view = rd<unsigned char>(f, M), bad = rd<unsigned char>(f, M), spars = rd<unsigned char>(f, M);
    const auto px = rd<float>(f, M), py = rd<float>(f, M), pxr = rd<float>(f, M), pdepth = rd<float>(f, M);
    const auto plevel = rd<int>(f, M);
    const auto pcos = rd<float>(f, M);
    const auto pdesc = rd<unsigned char>(f, (size_t)32 * M);
    const auto pobs = rd<int>(f, M);
    fclose(f);

    cv::Mat imL(rows, cols, CV_8UC1, imgL.data(), (size_t)cols), imR(rows, cols, CV_8UC1, imgR.data(), (size_t)cols);
    ORBextractor* exL = new ORBextractor(nfeat, hf[0], nlev, 20, 7);   // Tracking.cc:595-596
    ORBextractor* exR = new ORBextractor(nfeat, hf[0], nlev, 20, 7);
    GeometricCamera camera;
    camera.fx = hf[1]; camera.fy = hf[2]; camera.cx = hf[3]; camera.cy = hf[4];
    const std::vector<float> scale = exL->GetScaleFactors(), sigma2 = exL->GetScaleSigmaSquares();
    auto mat3 = [](const std::vector<float>& m) { Eigen::Matrix3f R; std::memcpy(R.m, m.data(), 36); return R; };
    auto vec3 = [](const float* v) { return Eigen::Vector3f{{v[0], v[1], v[2]}}; };

    // LastFrame: keypoints (octave, angle), map points (world position, descriptor, Observations()), outlier flags, pose
    Frame Last;
    Last.mnId = 1;
    {
        std::vector<unsigned char> zero((size_t)32 * NL, 0);
        Last.SetFeatures(lk, zero.data());
        Last.mvbOutlier.assign(NL, false);
        for (int i = 0; i < NL; i++) {
            Last.mvbOutlier[i] = outl[i] != 0;
            if (!has[i]) continue;
            auto p = std::make_shared<MapPoint>();
            p->mnId = 100000 + i;
            p->pos = vec3(&Xw[3 * (size_t)i]);
            p->nObs = lobs[i];
            std::memcpy(p->descriptor, &ldesc[(size_t)32 * i], 32);
            Last.mvpMapPoints[i] = p;
        }
        Last.mTcw = Sophus::SE3f(mat3(Rlw), vec3(tlw.data()));
    }
    // the local map of SearchLocalPoints; scratch = what the host's Frame::isInFrustum wrote (fed, not timed)
    std::vector<MP> local(M);
    for (int i = 0; i < M; i++) {
        auto p = std::make_shared<MapPoint>();
        p->mnId = 200000 + i;
        p->mbBad = bad[i] != 0; p->mbSparsified = spars[i] != 0;
        p->nObs = pobs[i];
        std::memcpy(p->descriptor, &pdesc[(size_t)32 * i], 32);
        local[i] = p;
    }
    auto set_scratch = [&]() {
        for (int i = 0; i < M; i++) {
            MapPoint& p = *local[i];
            p.mbTrackInView = in_view[i] != 0;
            p.mTrackProjX = px[i]; p.mTrackProjY = py[i]; p.mTrackProjXR = pxr[i]; p.mTrackDepth = pdepth[i];
            p.mnTrackScaleLevel = plevel[i]; p.mTrackViewCos = pcos[i];
        }
    };

    std::vector<cv::KeyPoint> kL, kR;
    cv::Mat dL, dR;
    std::vector<int> lap = {0, 0};
    auto eye = [&](int e) {
        if (e == 0) (*exL)(imL, cv::Mat(), kL, dL, lap);
        else (*exR)(imR, cv::Mat(), kR, dR, lap);
    };
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };

    std::vector<double> t_extract, t_a14, t_a13, t_total;
    std::vector<int> ids14, ids13;
    int n14 = 0, n13 = 0, N = 0, pyr_ok = 0;
    msorb_host::LastFrameProjection proj;
    const int warm = 5;
    for (int it = 0; it < iters + warm; it++) {
        const auto t0 = clk::now();
        std::thread a(eye, 0), b(eye, 1);                                  // Frame.cc:122-125
        a.join(); b.join();
        const auto t1 = clk::now();
        // the rest of Frame::Frame on the host (not timed): mvKeysUn = mvKeys (rectified input), ComputeStereoMatches' outputs
        Frame Cur;
        Cur.mnId = 1000 + it;                                              // a new frame every time: the class uploads it once
        N = (int)kL.size();
        if (N != N_expected) { fprintf(stderr, "unchanged_callers: %d keypoints, the scene file expects %d\n", N, N_expected); return 4; }
        std::vector<unsigned char> db((size_t)32 * N);
        for (int i = 0; i < N; i++) std::memcpy(&db[(size_t)32 * i], dL.ptr<unsigned char>(i), 32);
        Cur.SetFeatures(kL, db.data());
        Cur.mvuRight = ur_in; Cur.mvDepth = depth_in;
        Cur.mvScaleFactors = scale; Cur.mvLevelSigma2 = sigma2;
        Cur.mnScaleLevels = nlev; Cur.mfLogScaleFactor = std::log(hf[0]);
        Cur.mb = hf[5]; Cur.mbf = hf[6];
        Cur.mnMinX = hf[9]; Cur.mnMaxX = hf[10]; Cur.mnMinY = hf[11]; Cur.mnMaxY = hf[12];
        Cur.mpCamera = &camera;
        Cur.mTcw = Sophus::SE3f(mat3(Rcw), vec3(tcw.data()));             // mVelocity * mLastFrame.GetPose() (Tracking.cc:2854)
        Cur.mvbOutlier.assign(N, false);
        set_scratch();
        const auto t2 = clk::now();
        ORBmatcher m14(0.9f, true);                                         // Tracking.cc:2835
        n14 = m14.SearchByProjection(Cur, Last, hf[7], false);
        const auto t3 = clk::now();
        if (it == iters + warm - 1) {
            ids14.assign(N, -1);
            for (int j = 0; j < N; j++)
                if (Cur.mvpMapPoints[j]) ids14[j] = (int)(Cur.mvpMapPoints[j]->mnId - 100000);
            msorb_host::ProjectLastFrame(Cur, Last, false, proj);           // what the class projected (the oracle's input)
        }
        const auto t4 = clk::now();
        ORBmatcher m13(0.8f);                                               // Tracking.cc:3363
        n13 = m13.SearchByProjection(Cur, local, hf[8], false, 50.0f);
        const auto t5 = clk::now();
        if (it == iters + warm - 1) {
            ids13.assign(N, -1);
            for (int j = 0; j < N; j++)
                if (Cur.mvpMapPoints[j]) {
                    const long id = (long)Cur.mvpMapPoints[j]->mnId;
                    ids13[j] = id >= 200000 ? (int)(id - 200000) : M + (int)(id - 100000);
                }
            pyr_ok = (int)exL->mvImagePyramid.size() == nlev && exL->mvImagePyramid[nlev - 1].rows > 0;
        }
        if (it >= warm) {
            t_extract.push_back(ms(t0, t1)); t_a14.push_back(ms(t2, t3)); t_a13.push_back(ms(t4, t5));
            t_total.push_back(ms(t0, t1) + ms(t2, t3) + ms(t4, t5));
        }
    }
    FILE* o = fopen(argv[2], "wb");
    if (!o) return 5;
    wri(o, N); wri(o, (int)kR.size()); wri(o, n14); wri(o, n13); wri(o, proj.forward); wri(o, proj.backward); wri(o, pyr_ok); wri(o, iters);
    wrd(o, med(t_extract)); wrd(o, med(t_a14)); wrd(o, med(t_a13)); wrd(o, med(t_total));
    wr(o, ids14); wr(o, ids13);
    wr(o, proj.valid); wr(o, proj.u); wr(o, proj.v); wr(o, proj.ur);
    fwrite(kL.data(), sizeof(cv::KeyPoint), kL.size(), o);
    for (int i = 0; i < N; i++) fwrite(dL.ptr<unsigned char>(i), 1, 32, o);
    fclose(o);
    delete exL; delete exR;
    msorb_host::Shutdown();
    return 0;
}
