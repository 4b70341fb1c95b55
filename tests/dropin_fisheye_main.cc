// msorb_host::ComputeStereoFishEyeMatches (Frame.cc:1057-1101) over a stand-in Frame with the reference's member names; the
// triangulation is a deterministic stand-in functor (the camera model is the caller's).  Reads descriptors, writes the
// match tables.  usage: dropin_fisheye <in.bin> <out.bin>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <opencv2/opencv.hpp>
#include "ORBmatcher_device.h"

namespace ORB_SLAM3 {
struct Vec3 { float v[3] = {0, 0, 0}; };
struct Frame {
    std::vector<cv::KeyPoint> mvKeys, mvKeysRight;
    cv::Mat mDescriptors, mDescriptorsRight;
    int Nleft = 0, Nright = 0, monoLeft = 0, monoRight = 0, mnCloseMPs = 7;
    std::vector<int> mvLeftToRightMatch, mvRightToLeftMatch;
    std::vector<float> mvDepth, mvuRight, mvLevelSigma2;
    std::vector<Vec3> mvStereo3Dpoints;
};
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
    const auto hdr = rd<int>(f, 4);   // Nleft, Nright, monoLeft, monoRight
    Frame F;
    F.Nleft = hdr[0]; F.Nright = hdr[1]; F.monoLeft = hdr[2]; F.monoRight = hdr[3];
    F.mvKeys = rd<cv::KeyPoint>(f, F.Nleft);
    F.mvKeysRight = rd<cv::KeyPoint>(f, F.Nright);
    auto dl = rd<unsigned char>(f, (size_t)F.Nleft * 32), dr = rd<unsigned char>(f, (size_t)F.Nright * 32);
    fclose(f);
    F.mDescriptors = cv::Mat(F.Nleft, 32, CV_8UC1, dl.data(), 32);
    F.mDescriptorsRight = cv::Mat(F.Nright, 32, CV_8UC1, dr.data(), 32);
    F.mvLevelSigma2.resize(8);
    float s = 1.f;
    for (auto& v : F.mvLevelSigma2) { v = s * s; s *= 1.2f; }
    // stand-in for KannalaBrandt8::TriangulateMatches: depth from the keypoints' x difference, rejected when small
    auto tri = [](const cv::KeyPoint& a, const cv::KeyPoint& b, float s1, float s2, Vec3& p) {
        const float d = a.pt.x - b.pt.x;
        p.v[0] = a.pt.x; p.v[1] = a.pt.y; p.v[2] = d + s1 - s2;
        return d > 2.0f ? 100.0f / d : -1.0f;
    };
    const int n = msorb_host::ComputeStereoFishEyeMatches(F, tri);
    FILE* o = fopen(argv[2], "wb");
    fwrite(&n, 4, 1, o); fwrite(&F.mnCloseMPs, 4, 1, o);
    fwrite(F.mvLeftToRightMatch.data(), 4, F.Nleft, o);
    fwrite(F.mvRightToLeftMatch.data(), 4, F.Nright, o);
    fwrite(F.mvDepth.data(), 4, F.Nleft, o);
    fwrite(F.mvuRight.data(), 4, F.Nleft, o);
    for (int i = 0; i < F.Nleft; i++) fwrite(F.mvStereo3Dpoints[i].v, 4, 3, o);
    fclose(o);
    return 0;
}
