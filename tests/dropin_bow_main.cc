// Compiles ms-slam_amd/host/BoW_device.h against stand-ins of Frame (mDescriptors, mBowVec, mFeatVec with DBoW2's
// container types) and runs Frame::ComputeBoW + the distinctive-descriptor choice; dumps the results for the pytest.
// usage: dropin_bow <voc.txt> <in.bin> <out.bin>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#include <opencv2/opencv.hpp>

#include "BoW_device.h"

namespace DBoW2 {  // container types of Thirdparty/DBoW2/DBoW2/BowVector.h:59-60, FeatureVector.h
typedef unsigned int WordId;
typedef double WordValue;
typedef unsigned int NodeId;
class BowVector : public std::map<WordId, WordValue> {};
class FeatureVector : public std::map<NodeId, std::vector<unsigned int>> {};
}  // namespace DBoW2
namespace ORB_SLAM3 {
struct Frame {
    cv::Mat mDescriptors;
    DBoW2::BowVector mBowVec;
    DBoW2::FeatureVector mFeatVec;
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
    if (argc < 4) return 2;
    msorb_host::Vocabulary voc(argv[1]);
    FILE* f = fopen(argv[2], "rb");
    if (!f) return 3;
    const auto hdr = rd<int>(f, 2);  // n descriptors of the frame, P map points
    const int n = hdr[0], P = hdr[1];
    std::vector<unsigned char> desc = rd<unsigned char>(f, (size_t)n * 32);
    const auto begin = rd<int>(f, P + 1);
    std::vector<unsigned char> obs = rd<unsigned char>(f, (size_t)begin[P] * 32);
    fclose(f);
    Frame F;
    F.mDescriptors = cv::Mat(n, 32, CV_8UC1, desc.data(), 32);
    msorb_host::ComputeBoW(voc, F);
    const size_t nb0 = F.mBowVec.size();
    msorb_host::ComputeBoW(voc, F);  // second call: mBowVec not empty -> untouched (Frame.cc:672)
    std::vector<std::vector<cv::Mat>> per(P);
    for (int p = 0; p < P; p++)
        for (int k = begin[p]; k < begin[p + 1]; k++) per[p].push_back(cv::Mat(1, 32, CV_8UC1, &obs[(size_t)k * 32], 32));
    const std::vector<int> best = msorb_host::DistinctiveDescriptorIndices(per);
    FILE* o = fopen(argv[3], "wb");
    const int nb = (int)F.mBowVec.size(), nf = (int)F.mFeatVec.size(), same = nb0 == F.mBowVec.size();
    fwrite(&nb, 4, 1, o); fwrite(&nf, 4, 1, o); fwrite(&same, 4, 1, o);
    for (auto& e : F.mBowVec) { fwrite(&e.first, 4, 1, o); fwrite(&e.second, 8, 1, o); }
    for (auto& e : F.mFeatVec) {
        const int cnt = (int)e.second.size();
        fwrite(&e.first, 4, 1, o); fwrite(&cnt, 4, 1, o); fwrite(e.second.data(), 4, cnt, o);
    }
    fwrite(best.data(), 4, P, o);
    fclose(o);
    return 0;
}
