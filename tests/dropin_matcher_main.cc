// Compiles ms-slam_amd/host/ORBmatcher_device.h against minimal stand-ins of the reference's Frame / MapPoint (same
// member names as include/Frame.h / include/MapPoint.h; tests/cv_stub for cv::Mat / cv::KeyPoint) and runs the
// SearchLocalPoints matcher call the way Tracking.cc:3388 does.  Input: one binary blob written by the pytest;
// output: nmatches + the map-point index every keypoint holds afterwards.
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include <opencv2/opencv.hpp>

#include "ORBmatcher_device.h"

namespace ORB_SLAM3 {
struct MapPoint {  // the members ORBmatcher.cc:43-142 touches
    bool mbTrackInView = false, mbTrackInViewR = false, mbSparsified = false, mbBad = false;
    float mTrackProjX = 0, mTrackProjY = 0, mTrackProjXR = 0, mTrackDepth = 0, mTrackViewCos = 0;
    int mnTrackScaleLevel = 0, nObs = 0, id = -1;
    unsigned char descriptor[32];
    bool isBad() const { return mbBad; }
    int Observations() const { return nObs; }
    cv::Mat GetDescriptor() { return cv::Mat(1, 32, CV_8UC1, descriptor, 32); }
};
struct Frame {
    int N = 0, Nleft = -1;
    std::vector<cv::KeyPoint> mvKeysUn;
    cv::Mat mDescriptors;
    std::vector<float> mvuRight, mvScaleFactors;
    std::vector<std::shared_ptr<MapPoint>> mvpMapPoints;
    static float mnMinX, mnMaxX, mnMinY, mnMaxY;
};
float Frame::mnMinX, Frame::mnMaxX, Frame::mnMinY, Frame::mnMaxY;
}  // namespace ORB_SLAM3

template <class T>
static std::vector<T> rd(FILE* f, size_t n) {
    std::vector<T> v(n);
    if (n && fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(3); }
    return v;
}

int main(int argc, char** argv) {
    using namespace ORB_SLAM3;
    if (argc < 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 3;
    const auto hdr = rd<int>(f, 4);  // N, nlevels, M, bFarPoints
    const int N = hdr[0], nlev = hdr[1], M = hdr[2];
    const auto fl = rd<float>(f, 7);  // minX maxX minY maxY th thFar nnratio
    Frame F;
    F.N = N;
    F.mvKeysUn = rd<cv::KeyPoint>(f, N);
    std::vector<unsigned char> desc = rd<unsigned char>(f, (size_t)N * 32);
    F.mDescriptors = cv::Mat(N, 32, CV_8UC1, desc.data(), 32);
    F.mvuRight = rd<float>(f, N);
    F.mvScaleFactors = rd<float>(f, nlev);
    Frame::mnMinX = fl[0]; Frame::mnMaxX = fl[1]; Frame::mnMinY = fl[2]; Frame::mnMaxY = fl[3];
    const auto inView = rd<unsigned char>(f, M), bad = rd<unsigned char>(f, M), spars = rd<unsigned char>(f, M);
    const auto px = rd<float>(f, M), py = rd<float>(f, M), pxr = rd<float>(f, M), depth = rd<float>(f, M), vcos = rd<float>(f, M);
    const auto level = rd<int>(f, M), obs = rd<int>(f, M);
    const auto mdesc = rd<unsigned char>(f, (size_t)M * 32);
    const auto init = rd<int>(f, N);  // map point already held by keypoint i: index into the table, or -1
    fclose(f);
    std::vector<std::shared_ptr<MapPoint>> all(M);
    for (int i = 0; i < M; i++) {
        auto p = std::make_shared<MapPoint>();
        p->mbTrackInView = inView[i]; p->mbBad = bad[i]; p->mbSparsified = spars[i];
        p->mTrackProjX = px[i]; p->mTrackProjY = py[i]; p->mTrackProjXR = pxr[i]; p->mTrackDepth = depth[i];
        p->mTrackViewCos = vcos[i]; p->mnTrackScaleLevel = level[i]; p->nObs = obs[i]; p->id = i;
        memcpy(p->descriptor, &mdesc[(size_t)i * 32], 32);
        all[i] = p;
    }
    F.mvpMapPoints.assign(N, nullptr);
    for (int i = 0; i < N; i++) if (init[i] >= 0) F.mvpMapPoints[i] = all[init[i]];
    // the local map handed to the matcher: every second point of the table is NOT in it when it is only held by the frame
    // (exercises the "extra" entries of the adapter); points the frame does not hold are all in it
    std::vector<char> held(M, 0);
    for (int i = 0; i < N; i++) if (init[i] >= 0) held[init[i]] = 1;
    std::vector<std::shared_ptr<MapPoint>> local;
    for (int i = 0; i < M; i++) if (!(held[i] && (i & 1))) local.push_back(all[i]);
    msorb_host::DeviceFrame<Frame> dev;
    dev.Upload(F);
    const int nmatches = msorb_host::SearchByProjection(dev, F, local, fl[4], hdr[3] != 0, fl[5], fl[6]);  // Tracking.cc:3388
    FILE* o = fopen(argv[2], "wb");
    fwrite(&nmatches, 4, 1, o);
    for (int i = 0; i < N; i++) { const int id = F.mvpMapPoints[i] ? F.mvpMapPoints[i]->id : -1; fwrite(&id, 4, 1, o); }
    fclose(o);
    return 0;
}
