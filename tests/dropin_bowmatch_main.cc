// Compiles the SearchByBoW bodies of ms-slam_amd/host/ORBmatcher_device.h against stand-ins of KeyFrame / Frame /
// MapPoint (member names of include/KeyFrame.h, include/Frame.h) and runs (a) the relocalisation candidate loop
// (K KeyFrames against one Frame, Tracking.cc:3577-3600) and (b) SearchByBoW(pKF1, pKF2, ...) on the first two.
// usage: dropin_bowmatch <in.bin> <out.bin>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <vector>

#include <opencv2/opencv.hpp>

#include "ORBmatcher_device.h"

namespace DBoW2 {
typedef unsigned int NodeId;
class FeatureVector : public std::map<NodeId, std::vector<unsigned int>> {
public:
    void addFeature(NodeId id, unsigned int i_feature) {  // FeatureVector.cpp:30-45
        auto vit = this->lower_bound(id);
        if (vit != this->end() && vit->first == id) vit->second.push_back(i_feature);
        else { vit = this->insert(vit, value_type(id, std::vector<unsigned int>())); vit->second.push_back(i_feature); }
    }
};
}  // namespace DBoW2
namespace ORB_SLAM3 {
struct MapPoint {
    bool mbBad = false;
    int id = -1;
    bool isBad() const { return mbBad; }
};
struct Side {
    int N = 0;
    std::vector<unsigned char> bytes;
    cv::Mat mDescriptors;
    std::vector<cv::KeyPoint> mvKeys;
    DBoW2::FeatureVector mFeatVec;
    std::vector<std::shared_ptr<MapPoint>> mvpMapPoints;
};
struct KeyFrame : Side {
    std::vector<std::shared_ptr<MapPoint>> GetMapPointMatches() { return mvpMapPoints; }
    DBoW2::FeatureVector GetFeatureVector() { return mFeatVec; }
    std::vector<cv::KeyPoint> GetAllKeyUn() { return mvKeys; }
};
struct Frame : Side {};
}  // namespace ORB_SLAM3

template <class T>
static std::vector<T> rd(FILE* f, size_t n) {
    std::vector<T> v(n);
    if (n && fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(3); }
    return v;
}
static void read_side(FILE* f, ORB_SLAM3::Side& s, int& next_id) {
    using namespace ORB_SLAM3;
    const int n = rd<int>(f, 1)[0];
    s.N = n;
    s.bytes = rd<unsigned char>(f, (size_t)n * 32);
    s.mDescriptors = cv::Mat(n, 32, CV_8UC1, s.bytes.data(), 32);
    const auto angle = rd<float>(f, n);
    const auto node = rd<int>(f, n);
    const auto mp = rd<unsigned char>(f, n);  // 0 none, 1 good, 2 bad
    s.mvKeys.resize(n);
    s.mvpMapPoints.resize(n);
    for (int i = 0; i < n; i++) {
        s.mvKeys[i] = cv::KeyPoint{};
        s.mvKeys[i].angle = angle[i];
        if (node[i] >= 0) s.mFeatVec.addFeature((unsigned)node[i], (unsigned)i);
        if (mp[i]) {
            s.mvpMapPoints[i] = std::make_shared<MapPoint>();
            s.mvpMapPoints[i]->mbBad = mp[i] == 2;
            s.mvpMapPoints[i]->id = next_id;
        }
        next_id++;  // id = global feature number, so the pytest can map it back
    }
}

int main(int argc, char** argv) {
    using namespace ORB_SLAM3;
    if (argc < 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 3;
    const auto hdr = rd<int>(f, 2);
    const int K = hdr[0], ori = hdr[1];
    const float ratio = rd<float>(f, 1)[0];
    int next_id = 0;
    Frame F;
    read_side(f, F, next_id);
    std::vector<std::shared_ptr<KeyFrame>> kfs(K);
    std::vector<int> first_id(K);
    for (int k = 0; k < K; k++) {
        kfs[k] = std::make_shared<KeyFrame>();
        first_id[k] = next_id;
        read_side(f, *kfs[k], next_id);
    }
    fclose(f);
    std::vector<std::vector<std::shared_ptr<MapPoint>>> vvp;
    const std::vector<int> nm = msorb_host::SearchByBoWBatch(kfs, F, vvp, ratio, ori != 0);
    std::vector<std::shared_ptr<MapPoint>> single;
    const int nm_single = K ? msorb_host::SearchByBoW(kfs[0], F, single, ratio, ori != 0) : 0;
    FILE* o = fopen(argv[2], "wb");
    for (int k = 0; k < K; k++) {
        fwrite(&nm[k], 4, 1, o);
        for (int j = 0; j < F.N; j++) {
            const int v = vvp[k][j] ? vvp[k][j]->id - first_id[k] : -1;  // KF feature index of the matched map point
            fwrite(&v, 4, 1, o);
        }
    }
    int same = K == 0 || (nm_single == nm[0] && single.size() == vvp[0].size());
    for (size_t j = 0; same && K && j < single.size(); j++) same = single[j] == vvp[0][j];
    fwrite(&same, 4, 1, o);
    if (K >= 2) {
        std::vector<std::shared_ptr<MapPoint>> m12;
        const int n12 = msorb_host::SearchByBoWKeyFrames(kfs[0], kfs[1], m12, ratio, ori != 0);
        fwrite(&n12, 4, 1, o);
        for (size_t i = 0; i < m12.size(); i++) {
            const int v = m12[i] ? m12[i]->id - first_id[1] : -1;
            fwrite(&v, 4, 1, o);
        }
    }
    fclose(o);
    return 0;
}
