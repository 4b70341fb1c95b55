// TEST-ONLY stand-ins for the MS-SLAM types that OUR host code (ms-slam_amd/host/ORBmatcher.{h,cc} and the *_device.h
// mirrors) touches: Eigen / Sophus value types, GeometricCamera, MapPoint, KeyFrame, Frame, DBoW2::FeatureVector — same
// member names and signatures as include/{MapPoint,KeyFrame,Frame}.h of the reference, minimal bodies.  They exist so that
// the drop-in ORBmatcher class can be compiled and exercised where OpenCV / Eigen / Sophus are absent (this image, the GPU
// box).  They are never used to build reference sources.
#pragma once
#include <algorithm>
#include <cmath>
#include <map>
#include <memory>
#include <set>
#include <tuple>
#include <vector>

#include <opencv2/opencv.hpp>

#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW

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

namespace Eigen {
struct Vector2f { float v[2]; float operator()(int i) const { return v[i]; } };
struct Vector3f {
    float v[3];
    float operator()(int i) const { return v[i]; }
    Vector3f eval() const { return *this; }
    Vector3f operator-(const Vector3f& o) const { return Vector3f{{v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]}}; }
    Vector3f operator+(const Vector3f& o) const { return Vector3f{{v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]}}; }
    Vector3f operator/(float s) const { return Vector3f{{v[0] / s, v[1] / s, v[2] / s}}; }
    Vector3f operator*(float s) const { return Vector3f{{v[0] * s, v[1] * s, v[2] * s}}; }
    float dot(const Vector3f& o) const { return v[0] * o.v[0] + v[1] * o.v[1] + v[2] * o.v[2]; }
    float norm() const { return std::sqrt(dot(*this)); }
};
struct Matrix3f {
    float m[9];
    float operator()(int r, int c) const { return m[3 * r + c]; }
    Matrix3f eval() const { return *this; }
    Matrix3f transpose() const { Matrix3f o; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) o.m[3 * r + c] = m[3 * c + r]; return o; }
    Matrix3f inverse() const {
        const float a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
        const float det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g), id = 1.0f / det;
        return Matrix3f{{(e * i - f * h) * id, (c * h - b * i) * id, (b * f - c * e) * id, (f * g - d * i) * id, (a * i - c * g) * id,
                         (c * d - a * f) * id, (d * h - e * g) * id, (b * g - a * h) * id, (a * e - b * d) * id}};
    }
    Matrix3f operator*(const Matrix3f& o) const {
        Matrix3f r{};
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) r.m[3 * i + j] += m[3 * i + k] * o.m[3 * k + j];
        return r;
    }
    Vector3f operator*(const Vector3f& x) const {
        Vector3f r{};
        for (int i = 0; i < 3; i++) r.v[i] = m[3 * i] * x.v[0] + m[3 * i + 1] * x.v[1] + m[3 * i + 2] * x.v[2];
        return r;
    }
};
}  // namespace Eigen

namespace Sophus {
struct SO3f { static Eigen::Matrix3f hat(const Eigen::Vector3f& t) { return Eigen::Matrix3f{{0, -t.v[2], t.v[1], t.v[2], 0, -t.v[0], -t.v[1], t.v[0], 0}}; } };
struct SE3f {
    Eigen::Matrix3f R{{1, 0, 0, 0, 1, 0, 0, 0, 1}};
    Eigen::Vector3f t{{0, 0, 0}};
    SE3f() {}
    SE3f(const Eigen::Matrix3f& r, const Eigen::Vector3f& tt) : R(r), t(tt) {}
    Eigen::Matrix3f rotationMatrix() const { return R; }
    Eigen::Vector3f translation() const { return t; }
    SO3f so3() const { return SO3f{}; }
    // Sophus holds the rotation as a unit quaternion; the stand-in derives it from R (w >= 0 branch is all the tests need)
    struct Quat { float qx, qy, qz, qw; float x() const { return qx; } float y() const { return qy; } float z() const { return qz; } float w() const { return qw; } };
    Quat unit_quaternion() const {
        const double w = std::sqrt(std::max(0.0, 1.0 + (double)R.m[0] + R.m[4] + R.m[8])) / 2.0;
        double q[4] = {((double)R.m[7] - R.m[5]) / (4 * w), ((double)R.m[2] - R.m[6]) / (4 * w), ((double)R.m[3] - R.m[1]) / (4 * w), w};
        const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        return Quat{(float)(q[0] / n), (float)(q[1] / n), (float)(q[2] / n), (float)(q[3] / n)};
    }
    SE3f operator*(const SE3f& o) const { return SE3f(R * o.R, R * o.t + t); }
    Eigen::Vector3f operator*(const Eigen::Vector3f& p) const { return R * p + t; }
    SE3f inverse() const { const Eigen::Matrix3f Rt = R.transpose(); const Eigen::Vector3f x = Rt * t; return SE3f(Rt, Eigen::Vector3f{{-x.v[0], -x.v[1], -x.v[2]}}); }
};
template <class T>
struct Sim3 {  // x -> s R x + t
    Eigen::Matrix3f R{{1, 0, 0, 0, 1, 0, 0, 0, 1}};
    Eigen::Vector3f t{{0, 0, 0}};
    float s = 1;
    Sim3() {}
    Sim3(float scale, const Eigen::Matrix3f& r, const Eigen::Vector3f& tt) : R(r), t(tt), s(scale) {}
    Eigen::Matrix3f rotationMatrix() const { return R; }
    Eigen::Vector3f translation() const { return t; }
    float scale() const { return s; }
    Eigen::Vector3f operator*(const Eigen::Vector3f& p) const { return (R * p) * s + t; }
    Sim3 inverse() const { const Eigen::Matrix3f Rt = R.transpose(); const Eigen::Vector3f x = (Rt * t) * (1.0f / s); return Sim3(1.0f / s, Rt, Eigen::Vector3f{{-x.v[0], -x.v[1], -x.v[2]}}); }
};
typedef Sim3<float> Sim3f;
}  // namespace Sophus

namespace ORB_SLAM3 {
using std::shared_ptr;
class KeyFrame;
class Frame;

class GeometricCamera {  // Pinhole (Pinhole.cpp:43-49)
public:
    float fx = 0, fy = 0, cx = 0, cy = 0;
    int id = 0;
    Eigen::Vector2f project(const Eigen::Vector3f& p) { return Eigen::Vector2f{{fx * p.v[0] / p.v[2] + cx, fy * p.v[1] / p.v[2] + cy}}; }
    float getParameter(int i) { return i == 0 ? fx : i == 1 ? fy : i == 2 ? cx : cy; }
    Eigen::Matrix3f toK_() { return Eigen::Matrix3f{{fx, 0, cx, 0, fy, cy, 0, 0, 1}}; }
    // GeometricCamera::epipolarConstrain (GeometricCamera.h; KannalaBrandt8.cpp:216-220 triangulates): a stand-in that is a pure
    // function of its arguments — the two keypoints, WHICH cameras, the signs of t12 — and logs every call (kp1.pt, kp2.pt, the two
    // camera ids, R12, t12, sigmaLevel, unc, answer: 21 floats) so that a test can check what the matcher handed it
    static inline std::vector<float>* epipolar_log = nullptr;
    bool epipolarConstrain(GeometricCamera* pCamera2, const cv::KeyPoint& kp1, const cv::KeyPoint& kp2, const Eigen::Matrix3f& R12,
                           const Eigen::Vector3f& t12, const float sigmaLevel, const float unc) {
        const long h = (long)kp1.pt.x * 7 + (long)kp1.pt.y * 13 + (long)kp2.pt.x * 29 + (long)kp2.pt.y * 31 + id * 5 + pCamera2->id * 11 +
                       (t12.v[0] > 0 ? 3 : 0) + (t12.v[1] > 0 ? 17 : 0);
        const bool ok = h % 3 != 0;
        if (epipolar_log) {
            const float row[21] = {kp1.pt.x, kp1.pt.y, kp2.pt.x, kp2.pt.y, (float)id, (float)pCamera2->id, R12.m[0], R12.m[1], R12.m[2], R12.m[3],
                                   R12.m[4], R12.m[5], R12.m[6], R12.m[7], R12.m[8], t12.v[0], t12.v[1], t12.v[2], sigmaLevel, unc, ok ? 1.0f : 0.0f};
            epipolar_log->insert(epipolar_log->end(), row, row + 21);
        }
        return ok;
    }
};

class MapPoint {  // the members ORBmatcher touches; Replace / AddObservation keep a small model of the map and a log
public:
    long unsigned int mnId = 0;
    bool mbBad = false;
    int nObs = 0;
    std::map<const KeyFrame*, int> obsIdx;  // KeyFrame -> index of this point in it (GetIndexInKeyFrame)
    Eigen::Vector3f pos{}, normal{};
    float mfMaxDistance = 0, mfMinDistance = 0;
    unsigned char descriptor[32] = {0};
    // scratch written by Frame::isInFrustum / read by SearchByProjection (MapPoint.h:132-141)
    float mTrackProjX = 0, mTrackProjY = 0, mTrackDepth = 0, mTrackProjXR = 0, mTrackViewCos = 0;
    float mTrackProjYR = 0, mTrackViewCosR = 0;   // the right camera's scratch of a two-camera frame (MapPoint.h:109-112)
    bool mbTrackInView = false, mbTrackInViewR = false, mbSparsified = false;
    int mnTrackScaleLevel = 0, mnTrackScaleLevelR = -1;
    long unsigned int mnLastFrameSeen = 0, mnLoopPointForKF = 0;
    int nVisible = 0;
    static inline std::vector<long>* log = nullptr;  // (kind, a, b): 1 = a->Replace(b), 2 = a->AddObservation(kf, idx b)
    bool isBad() const { return mbBad; }
    int Observations() const { return nObs; }
    bool IsInKeyFrame(const shared_ptr<KeyFrame>& kf) const { return obsIdx.count(kf.get()) != 0; }
    std::tuple<int, int> GetIndexInKeyFrame(const shared_ptr<KeyFrame>& kf) const {
        auto it = obsIdx.find(kf.get());
        return it == obsIdx.end() ? std::tuple<int, int>(-1, -1) : std::tuple<int, int>(it->second, -1);
    }
    Eigen::Vector3f GetWorldPos() const { return pos; }
    Eigen::Vector3f GetNormal() const { return normal; }
    float GetMaxDistanceInvariance() const { return 1.2f * mfMaxDistance; }
    float GetMinDistanceInvariance() const { return 0.8f * mfMinDistance; }
    float GetMaxDistance() const { return mfMaxDistance; }
    float GetMinDistance() const { return mfMinDistance; }
    cv::Mat GetDescriptor() { return cv::Mat(1, 32, CV_8UC1, descriptor, 32); }
    void IncreaseVisible(int n = 1) { nVisible += n; }
    int PredictScale(const float& currentDist, const shared_ptr<KeyFrame>& kf);
    int PredictScale(const float& currentDist, Frame* pF);
    void Replace(const shared_ptr<MapPoint>& p) {
        if (log) { log->push_back(1); log->push_back((long)mnId); log->push_back((long)p->mnId); }
        mbBad = true;
        p->nObs += nObs;
        for (auto& k : obsIdx) p->obsIdx.insert(k);
    }
    void AddObservation(const shared_ptr<KeyFrame>& kf, int idx) {
        if (log) { log->push_back(2); log->push_back((long)mnId); log->push_back(idx); }
        obsIdx[kf.get()] = idx;
        nObs += 2;
    }
};

struct FeatureSide {  // what Frame and KeyFrame share
    int N = 0;
    std::vector<unsigned char> bytes;
    cv::Mat mDescriptors;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
    std::vector<float> mvuRight, mvDepth;
    DBoW2::FeatureVector mFeatVec;
    std::vector<shared_ptr<MapPoint>> mvpMapPoints;
    std::vector<float> mvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
    int mnScaleLevels = 8;
    float mfLogScaleFactor = 0, mbf = 0, mb = 0;
    GeometricCamera* mpCamera = nullptr;
    Sophus::SE3f mTcw;
    void SetFeatures(const std::vector<cv::KeyPoint>& kps, const unsigned char* desc) {
        N = (int)kps.size();
        mvKeys = mvKeysUn = kps;
        bytes.assign(desc, desc + (size_t)N * 32);
        mDescriptors = cv::Mat(N, 32, CV_8UC1, bytes.data(), 32);
        mvuRight.assign(N, -1.0f); mvDepth.assign(N, -1.0f);
        mvpMapPoints.assign(N, shared_ptr<MapPoint>());
    }
};

class KeyFrame : protected FeatureSide {  // the feature arrays are protected in MS-SLAM's KeyFrame: accessors only
public:
    using FeatureSide::SetFeatures;
    using FeatureSide::mvScaleFactors; using FeatureSide::mvLevelSigma2; using FeatureSide::mvInvLevelSigma2;
    using FeatureSide::mnScaleLevels; using FeatureSide::mfLogScaleFactor; using FeatureSide::mbf; using FeatureSide::mb;
    using FeatureSide::mpCamera;
    long unsigned int mnId = 0;
    float fx = 0, fy = 0, cx = 0, cy = 0;
    int mnMinX = 0, mnMinY = 0, mnMaxX = 0, mnMaxY = 0;
    bool mbSparsified = false;
    GeometricCamera* mpCamera2 = nullptr;
    int GetN() { return N; }
    // a KeyFrame of a two-camera rig (KeyFrame.h:345-355, 377-410): mvKeys = the left camera's NLeft keypoints, mvKeysRight the right
    // camera's, N = NLeft + NRight descriptor rows / map points, mvuRight NLeft entries of -1 (Frame.cc:1069), mTrl / mTlr the rig
    int NLeft = -1;
    std::vector<cv::KeyPoint> mvKeysRight;
    Sophus::SE3f mTrl, mTlr;
    void SetRig(const std::vector<cv::KeyPoint>& kl, const std::vector<cv::KeyPoint>& kr, const unsigned char* desc, const Sophus::SE3f& Trl) {
        std::vector<cv::KeyPoint> all = kl;
        all.insert(all.end(), kr.begin(), kr.end());
        SetFeatures(all, desc);
        mvKeys = kl; mvKeysUn = kl; mvKeysRight = kr;
        NLeft = (int)kl.size();
        mvuRight.assign(kl.size(), -1.0f);
        mTrl = Trl; mTlr = Trl.inverse();
    }
    int GetNLeft() { return NLeft; }
    cv::KeyPoint GetKey(size_t idx) { return mvKeys[idx]; }
    cv::KeyPoint GetKeyRight(size_t idx) { return mvKeysRight[idx]; }
    bool FromRightImage(size_t idx) { return !(NLeft == -1 || (int)idx < NLeft); }
    Sophus::SE3f GetRightPose() { return mTrl * mTcw; }                             // KeyFrame.cc:952-956
    Sophus::SE3f GetRightPoseInverse() { return mTcw.inverse() * mTlr; }            // :958-962
    Eigen::Vector3f GetRightCameraCenter() { return (mTcw.inverse() * mTlr).translation(); }   // :964-968
    void SetPose(const Sophus::SE3f& T) { mTcw = T; }
    Sophus::SE3f GetPose() { return mTcw; }
    Sophus::SE3f GetPoseInverse() { return mTcw.inverse(); }
    Eigen::Vector3f GetCameraCenter() { return mTcw.inverse().translation(); }
    cv::Mat GetDescriptor(const int& idx) { return idx >= mDescriptors.rows ? cv::Mat() : mDescriptors.row(idx); }
    std::vector<shared_ptr<MapPoint>> GetMapPointMatches() { return mvpMapPoints; }
    std::set<shared_ptr<MapPoint>> GetMapPoints() {
        std::set<shared_ptr<MapPoint>> s;
        for (auto& p : mvpMapPoints) if (p && !p->isBad()) s.insert(p);
        return s;
    }
    shared_ptr<MapPoint> GetMapPoint(const size_t& idx) { return mvpMapPoints[idx]; }
    void AddMapPoint(shared_ptr<MapPoint> p, const size_t& idx) { mvpMapPoints[idx] = p; }
    DBoW2::FeatureVector GetFeatureVector() { return mFeatVec; }
    void SetFeatureVector(const DBoW2::FeatureVector& fv) { mFeatVec = fv; }
    std::vector<cv::KeyPoint> GetAllKeyUn() { return mvKeysUn; }
    cv::KeyPoint GetKeyUn(size_t idx) { return mvKeysUn[idx]; }
    cv::KeyPoint GetKeyPoint(size_t idx) {                                          // KeyFrame.h:377-385
        if (NLeft == -1) return mvKeysUn[idx];
        else if ((int)idx < NLeft) return mvKeys[idx];
        else return mvKeysRight[idx - NLeft];
    }
    float GetuRight(size_t idx) { return mvuRight[idx]; }
    void SetuRight(const std::vector<float>& ur) { mvuRight = ur; }
    bool IsInImage(const float& x, const float& y) const { return x >= mnMinX && x < mnMaxX && y >= mnMinY && y < mnMaxY; }
};

class Frame : public FeatureSide {
public:
    long unsigned int mnId = 0;
    // two-camera (KannalaBrandt8 stereo) frames: mvKeys = the left camera's Nleft keypoints, mvKeysRight the right camera's Nright,
    // mDescriptors / mvpMapPoints N = Nleft + Nright rows, left first (Frame.h:226, 324-332)
    int Nleft = -1, Nright = -1;
    std::vector<cv::KeyPoint> mvKeysRight;
    std::vector<int> mvLeftToRightMatch, mvRightToLeftMatch;
    GeometricCamera* mpCamera2 = nullptr;
    Sophus::SE3f mTrl;
    Sophus::SE3f GetRelativePoseTrl() { return mTrl; }
    float mnMinX = 0, mnMaxX = 0, mnMinY = 0, mnMaxY = 0;
    std::vector<bool> mvbOutlier;
    std::map<long unsigned int, cv::Point2f> mmProjectPoints;
    Sophus::SE3f GetPose() const { return mTcw; }
    Eigen::Vector3f GetCameraCenter() const { return mTcw.inverse().translation(); }
};

inline int predict_scale_(float maxDistance, float currentDist, float logScaleFactor, int nLevels) {  // MapPoint.cc:540-572
    const float ratio = maxDistance / currentDist;
    int nScale = (int)std::ceil(std::log(ratio) / logScaleFactor);
    if (nScale < 0) nScale = 0;
    else if (nScale >= nLevels) nScale = nLevels - 1;
    return nScale;
}
inline int MapPoint::PredictScale(const float& currentDist, const shared_ptr<KeyFrame>& kf) {
    return predict_scale_(mfMaxDistance, currentDist, kf->mfLogScaleFactor, kf->mnScaleLevels);
}
inline int MapPoint::PredictScale(const float& currentDist, Frame* pF) {
    return predict_scale_(mfMaxDistance, currentDist, pF->mfLogScaleFactor, pF->mnScaleLevels);
}
}  // namespace ORB_SLAM3
