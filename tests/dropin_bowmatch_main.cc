// Compiles the SearchByBoW bodies of ms-slam_amd/host/ORBmatcher_device.h against stand-ins of KeyFrame / Frame /
// MapPoint (member names of include/KeyFrame.h, include/Frame.h) and runs (a) the relocalisation candidate loop
// (K KeyFrames against one Frame, Tracking.cc:3577-3600) and (b) SearchByBoW(pKF1, pKF2, ...) on the first two.
// usage: dropin_bowmatch <in.bin> <out.bin>
#include <cmath>
#include <cstdio>
#include <set>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
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
// Eigen / Sophus stand-ins (float, straightforward loops): just enough surface for TriangulationGeometry
struct Vec2 { float v[2]; float operator()(int i) const { return v[i]; } };
struct Vec3 {
    float v[3];
    float operator()(int i) const { return v[i]; }
    Vec3 eval() const { return *this; }
    Vec3 operator-(const Vec3& o) const { return Vec3{{v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]}}; }
    float dot(const Vec3& o) const { return v[0] * o.v[0] + v[1] * o.v[1] + v[2] * o.v[2]; }
    float norm() const { return std::sqrt(dot(*this)); }
};
struct Mat3 {
    float m[9];
    float operator()(int r, int c) const { return m[3 * r + c]; }
    Mat3 eval() const { return *this; }
    Mat3 transpose() const { Mat3 o; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) o.m[3 * r + c] = m[3 * c + r]; return o; }
    Mat3 inverse() const {
        const float a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
        const float det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g), id = 1.0f / det;
        return Mat3{{(e * i - f * h) * id, (c * h - b * i) * id, (b * f - c * e) * id, (f * g - d * i) * id, (a * i - c * g) * id,
                     (c * d - a * f) * id, (d * h - e * g) * id, (b * g - a * h) * id, (a * e - b * d) * id}};
    }
    Mat3 operator*(const Mat3& o) const {
        Mat3 r{};
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) r.m[3 * i + j] += m[3 * i + k] * o.m[3 * k + j];
        return r;
    }
    Vec3 operator*(const Vec3& x) const {
        Vec3 r{};
        for (int i = 0; i < 3; i++) r.v[i] = m[3 * i] * x.v[0] + m[3 * i + 1] * x.v[1] + m[3 * i + 2] * x.v[2];
        return r;
    }
};
struct SO3 { static Mat3 hat(const Vec3& t) { return Mat3{{0, -t.v[2], t.v[1], t.v[2], 0, -t.v[0], -t.v[1], t.v[0], 0}}; } };
struct SE3 {
    Mat3 R; Vec3 t;
    Mat3 rotationMatrix() const { return R; }
    Vec3 translation() const { return t; }
    SO3 so3() const { return SO3{}; }
    SE3 operator*(const SE3& o) const { SE3 r; r.R = R * o.R; const Vec3 x = R * o.t; r.t = Vec3{{x.v[0] + t.v[0], x.v[1] + t.v[1], x.v[2] + t.v[2]}}; return r; }
    Vec3 operator*(const Vec3& p) const { const Vec3 x = R * p; return Vec3{{x.v[0] + t.v[0], x.v[1] + t.v[1], x.v[2] + t.v[2]}}; }
    SE3 inverse() const { SE3 r; r.R = R.transpose(); const Vec3 x = r.R * t; r.t = Vec3{{-x.v[0], -x.v[1], -x.v[2]}}; return r; }
};
struct Camera {  // GeometricCamera / Pinhole
    float fx, fy, cx, cy;
    Vec2 project(const Vec3& p) { return Vec2{{fx * p.v[0] / p.v[2] + cx, fy * p.v[1] / p.v[2] + cy}}; }
    Mat3 toK_() { return Mat3{{fx, 0, cx, 0, fy, cy, 0, 0, 1}}; }
};
struct KeyFrame;
struct MapPoint {  // the members ORBmatcher::Fuse touches; Replace / AddObservation keep a small model of the map and a log
    bool mbBad = false;
    int id = -1, nObs = 0;
    std::set<const KeyFrame*> inKF;
    Vec3 pos{}, normal{};
    float mfMaxDistance = 0, mfMinDistance = 0;
    unsigned char descriptor[32] = {0};
    static std::vector<int>* log;  // (kind, a, b): 1 = a->Replace(b), 2 = a->AddObservation(kf, idx b)
    bool isBad() const { return mbBad; }
    int Observations() const { return nObs; }
    bool IsInKeyFrame(const std::shared_ptr<KeyFrame>& kf) const { return inKF.count(kf.get()) != 0; }
    Vec3 GetWorldPos() const { return pos; }
    Vec3 GetNormal() const { return normal; }
    float GetMaxDistanceInvariance() const { return 1.2f * mfMaxDistance; }
    float GetMinDistanceInvariance() const { return 0.8f * mfMinDistance; }
    cv::Mat GetDescriptor() { return cv::Mat(1, 32, CV_8UC1, descriptor, 32); }
    int PredictScale(const float& currentDist, const std::shared_ptr<KeyFrame>& kf);
    void Replace(const std::shared_ptr<MapPoint>& p) {
        if (log) { log->push_back(1); log->push_back(id); log->push_back(p->id); }
        mbBad = true;
        p->nObs += nObs;
        for (auto k : inKF) p->inKF.insert(k);
    }
    void AddObservation(const std::shared_ptr<KeyFrame>& kf, int idx) {
        if (log) { log->push_back(2); log->push_back(id); log->push_back(idx); }
        inKF.insert(kf.get());
        nObs += 2;
    }
};
std::vector<int>* MapPoint::log = nullptr;
struct Side {
    int N = 0;
    std::vector<unsigned char> bytes;
    cv::Mat mDescriptors;
    std::vector<cv::KeyPoint> mvKeys;
    DBoW2::FeatureVector mFeatVec;
    std::vector<std::shared_ptr<MapPoint>> mvpMapPoints;
};
struct KeyFrame : protected Side {  // the feature arrays are protected in MS-SLAM's KeyFrame: accessors only
    friend void read_kf(FILE*, KeyFrame&, int&, bool);
    std::vector<float> mvuRight_;
    SE3 Tcw;
    Camera* mpCamera = nullptr;
    Camera* mpCamera2 = nullptr;
    std::vector<float> mvScaleFactors, mvLevelSigma2;
    int GetN() { return N; }
    cv::Mat GetDescriptor(const int& idx) { return idx >= mDescriptors.rows ? cv::Mat() : mDescriptors.row(idx); }
    std::vector<std::shared_ptr<MapPoint>> GetMapPointMatches() { return mvpMapPoints; }
    DBoW2::FeatureVector GetFeatureVector() { return mFeatVec; }
    std::vector<cv::KeyPoint> GetAllKeyUn() { return mvKeys; }
    float GetuRight(size_t idx) { return mvuRight_[idx]; }
    int mnMinX = 0, mnMinY = 0, mnMaxX = 0, mnMaxY = 0, mnScaleLevels = 8;
    bool mbSparsified = false;  // KeyFrame.h:272 (set by EraseBadDescriptor, which also swaps mGrid away)
    float mbf = 0, mfLogScaleFactor = 0;
    std::vector<float> mvInvLevelSigma2;
    bool IsInImage(const float& x, const float& y) const { return x >= mnMinX && x < mnMaxX && y >= mnMinY && y < mnMaxY; }
    std::shared_ptr<MapPoint> GetMapPoint(const size_t& idx) { return mvpMapPoints[idx]; }
    void AddMapPoint(std::shared_ptr<MapPoint> p, const size_t& idx) { mvpMapPoints[idx] = p; }
    SE3 GetPose() { return Tcw; }
    SE3 GetPoseInverse() { return Tcw.inverse(); }
    Vec3 GetCameraCenter() { return Tcw.inverse().t; }
};
struct Frame : Side {};
inline int MapPoint::PredictScale(const float& currentDist, const std::shared_ptr<KeyFrame>& kf) {  // MapPoint.cc:540-555
    const float ratio = mfMaxDistance / currentDist;
    int nScale = (int)std::ceil(std::log(ratio) / kf->mfLogScaleFactor);
    if (nScale < 0) nScale = 0;
    else if (nScale >= kf->mnScaleLevels) nScale = kf->mnScaleLevels - 1;
    return nScale;
}
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

namespace ORB_SLAM3 {
void read_kf(FILE* f, KeyFrame& kf, int& next_id, bool full_kp) {
    if (!full_kp) { read_side(f, kf, next_id); return; }
    const int n = rd<int>(f, 1)[0];
    kf.N = n;
    kf.bytes = rd<unsigned char>(f, (size_t)n * 32);
    kf.mDescriptors = cv::Mat(n, 32, CV_8UC1, kf.bytes.data(), 32);
    kf.mvKeys = rd<cv::KeyPoint>(f, n);
    const auto node = rd<int>(f, n);
    const auto mp = rd<unsigned char>(f, n);
    kf.mvuRight_ = rd<float>(f, n);
    kf.mvpMapPoints.resize(n);
    for (int i = 0; i < n; i++) {
        if (node[i] >= 0) kf.mFeatVec.addFeature((unsigned)node[i], (unsigned)i);
        if (mp[i]) kf.mvpMapPoints[i] = std::make_shared<MapPoint>();
    }
    const auto pose = rd<float>(f, 12);
    for (int i = 0; i < 9; i++) kf.Tcw.R.m[i] = pose[i];
    for (int i = 0; i < 3; i++) kf.Tcw.t.v[i] = pose[9 + i];
    const int L = rd<int>(f, 1)[0];
    kf.mvScaleFactors = rd<float>(f, L);
    kf.mvLevelSigma2 = rd<float>(f, L);
}
}  // namespace ORB_SLAM3

// (pKF1, K neighbours) -> per neighbour F12, ep, nmatches, pairs
static int triangulation_main(const char* in, const char* out) {
    using namespace ORB_SLAM3;
    FILE* f = fopen(in, "rb");
    if (!f) return 3;
    const auto hdr = rd<int>(f, 4);
    const int K = hdr[0], only_stereo = hdr[1], coarse = hdr[2], ori = hdr[3];
    const auto cam = rd<float>(f, 4);
    Camera camera{cam[0], cam[1], cam[2], cam[3]};
    int dummy = 0;
    auto kf1 = std::make_shared<KeyFrame>();
    read_kf(f, *kf1, dummy, true);
    kf1->mpCamera = &camera;
    std::vector<std::shared_ptr<KeyFrame>> nb(K);
    for (int k = 0; k < K; k++) {
        nb[k] = std::make_shared<KeyFrame>();
        read_kf(f, *nb[k], dummy, true);
        nb[k]->mpCamera = &camera;
    }
    fclose(f);
    std::vector<std::vector<std::pair<size_t, size_t>>> vv;
    const std::vector<int> nm = msorb_host::SearchForTriangulationBatch(kf1, nb, vv, only_stereo != 0, coarse != 0, ori != 0);
    std::vector<std::pair<size_t, size_t>> single;
    const int nm0 = K ? msorb_host::SearchForTriangulation(kf1, nb[0], single, only_stereo != 0, coarse != 0, ori != 0) : 0;
    FILE* o = fopen(out, "wb");
    for (int k = 0; k < K; k++) {
        float F[9], ep[2];
        msorb_host::TriangulationGeometry(kf1, nb[k], F, ep);
        fwrite(F, 4, 9, o); fwrite(ep, 4, 2, o);
        const int cnt = (int)vv[k].size();
        fwrite(&nm[k], 4, 1, o); fwrite(&cnt, 4, 1, o);
        for (auto& pr : vv[k]) { const int a = (int)pr.first, b = (int)pr.second; fwrite(&a, 4, 1, o); fwrite(&b, 4, 1, o); }
    }
    const int same = K == 0 || (nm0 == nm[0] && single == vv[0]);
    fwrite(&same, 4, 1, o);
    fclose(o);
    return 0;
}

// ORBmatcher::Fuse through msorb_host::Fuse: one KeyFrame, M map points; dumps the geometry, nFused and the mutation log
static int fuse_main(const char* in, const char* out, bool sparsified) {
    using namespace ORB_SLAM3;
    FILE* f = fopen(in, "rb");
    if (!f) return 3;
    const auto hdr = rd<int>(f, 5);  // M, minX, maxX, minY, maxY
    const int M = hdr[0];
    const auto cam = rd<float>(f, 7);  // fx fy cx cy mbf logScale th
    Camera camera{cam[0], cam[1], cam[2], cam[3]};
    int dummy = 0;
    auto kf = std::make_shared<KeyFrame>();
    read_kf(f, *kf, dummy, true);
    kf->mpCamera = &camera;
    kf->mnMinX = hdr[1]; kf->mnMaxX = hdr[2]; kf->mnMinY = hdr[3]; kf->mnMaxY = hdr[4];
    kf->mbf = cam[4]; kf->mfLogScaleFactor = cam[5];
    kf->mnScaleLevels = (int)kf->mvScaleFactors.size();
    for (float s2 : kf->mvLevelSigma2) kf->mvInvLevelSigma2.push_back(1.0f / s2);
    const auto state = rd<unsigned char>(f, M);  // 0 null, 1 alive, 2 bad, 3 already in the KeyFrame
    const auto pos = rd<float>(f, (size_t)3 * M), nrm = rd<float>(f, (size_t)3 * M), maxd = rd<float>(f, M), mind = rd<float>(f, M);
    const auto obs = rd<int>(f, M);
    const auto desc = rd<unsigned char>(f, (size_t)M * 32);
    const auto kf_obs = rd<int>(f, kf->GetN());  // Observations() of the map point each KeyFrame feature holds (if it holds one)
    fclose(f);
    {
        auto held = kf->GetMapPointMatches();
        for (int j = 0; j < kf->GetN(); j++)
            if (held[j]) { held[j]->id = 100000 + j; held[j]->nObs = kf_obs[j]; held[j]->inKF.insert(kf.get()); }
    }
    std::vector<std::shared_ptr<MapPoint>> pts(M);
    for (int i = 0; i < M; i++) {
        if (!state[i]) continue;
        auto p = std::make_shared<MapPoint>();
        p->id = i; p->nObs = obs[i]; p->mbBad = state[i] == 2;
        if (state[i] == 3) p->inKF.insert(kf.get());
        memcpy(p->pos.v, &pos[3 * i], 12); memcpy(p->normal.v, &nrm[3 * i], 12);
        p->mfMaxDistance = maxd[i]; p->mfMinDistance = mind[i];
        memcpy(p->descriptor, &desc[(size_t)i * 32], 32);
        pts[i] = p;
    }
    msorb_host::FuseQueries Q;
    msorb_host::FuseGeometry(kf, pts, cam[6], Q);
    std::vector<int> log;
    MapPoint::log = &log;
    kf->mbSparsified = sparsified;  // KeyFrame::GetFeaturesInArea then returns nothing (KeyFrame.cc:800-801): Fuse must not touch the map
    msorb_host::DeviceFrame<Frame> dev;
    dev.UploadKeyFrame(kf);
    const int nFused = msorb_host::Fuse(dev, kf, pts, cam[6]);
    FILE* o = fopen(out, "wb");
    const int nlog = (int)log.size();
    fwrite(&nFused, 4, 1, o); fwrite(&nlog, 4, 1, o);
    fwrite(log.data(), 4, nlog, o);
    fwrite(Q.valid.data(), 1, M, o); fwrite(Q.u.data(), 4, M, o); fwrite(Q.v.data(), 4, M, o); fwrite(Q.ur.data(), 4, M, o);
    fwrite(Q.level.data(), 4, M, o); fwrite(Q.radius.data(), 4, M, o);
    fclose(o);
    return 0;
}

int main(int argc, char** argv) {
    using namespace ORB_SLAM3;
    if (argc < 3) return 2;
    if (argc > 3 && std::string(argv[3]) == "tri") return triangulation_main(argv[1], argv[2]);
    if (argc > 3 && std::string(argv[3]) == "fuse") return fuse_main(argv[1], argv[2], false);
    if (argc > 3 && std::string(argv[3]) == "fuse_sparsified") return fuse_main(argv[1], argv[2], true);
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
        read_kf(f, *kfs[k], next_id, false);
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
