// Device back-end for the per-frame matcher call sites of MS-SLAM's tracking thread, written against the reference's
// own types by name (templates: this header compiles inside MS-SLAM, where Frame / MapPoint / cv::Mat are the real
// classes, and in tests/dropin_matcher_main.cc, where they are minimal stand-ins with the same member names).
//
//   ORB_SLAM3::msorb_host::DeviceFrame<Frame>            the members of Frame the matcher reads, resident on the GPU
//   ORB_SLAM3::msorb_host::SearchByProjection(...)       body of ORBmatcher::SearchByProjection(Frame&, const
//                                                        vector<shared_ptr<MapPoint>>&, th, bFarPoints, thFarPoints)
//                                                        (src/ORBmatcher.cc:43-142, rectified / Nleft == -1 branch)
//   ORB_SLAM3::msorb_host::SearchByProjection(dev, Cur, Last, th, bMono, ...)
//                                                        body of ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono)
//                                                        (src/ORBmatcher.cc:1941-2152, Nleft == -1; TrackWithMotionModel)
//   ORB_SLAM3::msorb_host::SearchByProjection(dev, Cur, pKF, sAlreadyFound, th, ORBdist, ...)
//                                                        body of the relocalisation form (src/ORBmatcher.cc:2154-2275)
//   ORB_SLAM3::msorb_host::SearchLocalPointsPrepass(...) the isInFrustum loop of Tracking::SearchLocalPoints
//                                                        (src/Tracking.cc:3343-3361, src/Frame.cc:512-571)
//   ORB_SLAM3::msorb_host::SearchLocalPoints(dev, F, vpLocalMapPoints, th, ...)
//                                                        that loop AND the SearchByProjection call behind it (:3343-3388) as
//                                                        one device chain (msorb_search_local_points)
//   ORB_SLAM3::msorb_host::ComputeStereoMatches(...)     body of Frame::ComputeStereoMatches (src/Frame.cc:743-913)
//   ORB_SLAM3::msorb_host::ExtractStereo(F, left, imLeft, imRight)
//                                                        the two ExtractORB threads + ComputeStereoMatches of the stereo
//                                                        Frame constructor (src/Frame.cc:119-137) as ONE device call
//   ORB_SLAM3::msorb_host::ExtractStereoFrame(dev, F, left, imLeft, imRight)
//                                                        ExtractStereo + Frame::AssignFeaturesToGrid (src/Frame.cc:385-416) on the
//                                                        device: `dev` is ready for the searches without an upload
//   ORB_SLAM3::msorb_host::ComputeStereoFishEyeMatches(F, triangulate)
//                                                        body of Frame::ComputeStereoFishEyeMatches (src/Frame.cc:1057-1101): the
//                                                        BFMatcher knnMatch(k = 2) on the device, ratio test + triangulation as is
//   ORB_SLAM3::msorb_host::ExtractStereoSplit(F, left, right, imLeft, imRight)
//                                                        the same with one extractor object per GPU (left / right eye on
//                                                        devices A / B, gather over xGMI, association on A)
//   ORB_SLAM3::msorb_host::SearchByBoW(...)              bodies of ORBmatcher::SearchByBoW(pKF, F, vpMapPointMatches)
//                                                        (src/ORBmatcher.cc:223-421, Nleft == -1 branch) and
//                                                        SearchByBoW(pKF1, pKF2, vpMatches12) (:872-1016,
//                                                        SearchByBoWKeyFrames here); the batch form
//                                                        serves Relocalization's candidate loop (Tracking.cc:3577-3600)
//   ORB_SLAM3::msorb_host::Fuse(dev, pKF, vpMapPoints, th)
//                                                        body of ORBmatcher::Fuse(pKF, vpMapPoints, th, bRight = false)
//                                                        (:1404-1597): geometry and the map mutation stay the reference's
//                                                        code, the window search + error gates run on the device
//   ORB_SLAM3::msorb_host::SearchForTriangulation(...)   body of ORBmatcher::SearchForTriangulation (:1168-1402, no second
//                                                        camera); SearchForTriangulationBatch = all neighbours of one
//                                                        LocalMapping::CreateNewMapPoints pass (LocalMapping.cc:430-492)
//
// Use inside the reference: ORBmatcher::SearchByProjection keeps its signature and becomes
//     static thread_local msorb_host::DeviceFrame<Frame> dev;
//     dev.Upload(F);
//     return msorb_host::SearchByProjection(dev, F, vpMapPoints, th, bFarPoints, thFarPoints, mfNNratio);
// Same return value, same F.mvpMapPoints afterwards (same candidate sets, scan order, ratio test, sequential claims).
#ifndef MSORB_ORBMATCHER_DEVICE_H
#define MSORB_ORBMATCHER_DEVICE_H

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <utility>
#include <vector>

#include "msorb.h"

namespace ORB_SLAM3 {
namespace msorb_host {

// A failed call of the C ABI: the application's fatal-error callback first (msorb_set_fatal_callback, see ORBextractor.h),
// then std::runtime_error — which nothing in Tracking.cc / LocalMapping.cc / LoopClosing.cc catches, i.e. std::terminate for
// an unchanged caller, a catchable error for an embedding application.
inline void check(int rc, const char* what) {
    if (rc == MSORB_OK) return;
    const std::string msg = std::string(what) + ": " + msorb_last_error();
    msorb_notify_fatal(rc, msg.c_str());
    throw std::runtime_error(msg);
}
// the library that was loaded against the header this file was compiled with (MSORB_ABI_VERSION): checked once per process
inline void check_abi() {
    static const bool ok = msorb_abi_compatible(MSORB_ABI_VERSION) != 0;
    if (!ok) {
        const std::string msg = "libmsorb.so has ABI " + std::to_string(msorb_abi_version()) + ", the host layer was compiled against " +
                                std::to_string(MSORB_ABI_VERSION) + " (include/msorb.h)";
        msorb_notify_fatal(MSORB_E_INVALID, msg.c_str());
        throw std::runtime_error(msg);
    }
}

template <class FrameT>
class DeviceFrame {
public:
    explicit DeviceFrame(int device = 0) { check_abi(); check(msorb_frame_create(device, &h_), "msorb_frame_create"); }
    ~DeviceFrame() { msorb_frame_destroy(h_); }
    DeviceFrame(const DeviceFrame&) = delete;
    DeviceFrame& operator=(const DeviceFrame&) = delete;

    // mvKeysUn, mDescriptors, mvuRight, mnMinX..mnMaxY, mvScaleFactors -> device + 64x48 grid (Frame.cc:385-416)
    void Upload(const FrameT& F) {
        static_assert(sizeof(F.mvKeysUn[0]) == sizeof(msorb_keypoint), "cv::KeyPoint must be the 28-byte layout");
        const int n = (int)F.mvKeysUn.size();
        const uint8_t* desc = nullptr;
        if (n > 0 && F.mDescriptors.isContinuous()) {   // the extractor's output: one block of n x 32 bytes, handed over as it is
            desc = F.mDescriptors.template ptr<unsigned char>(0);
        } else {
            desc_.resize((size_t)n * 32);
            for (int i = 0; i < n; i++) std::memcpy(&desc_[(size_t)i * 32], F.mDescriptors.template ptr<unsigned char>(i), 32);
            desc = desc_.data();
        }
        check(msorb_frame_set(h_, reinterpret_cast<const msorb_keypoint*>(F.mvKeysUn.data()), n, desc,
                              F.mvuRight.empty() ? nullptr : F.mvuRight.data(), F.mnMinX, F.mnMaxX, F.mnMinY, F.mnMaxY,
                              F.mvScaleFactors.data(), (int)F.mvScaleFactors.size()),
              "msorb_frame_set");
    }
    // the same from a KeyFrame through its public accessors (the feature arrays are protected there).  KeyFrame keeps the
    // image bounds as ints (KeyFrame.h:251-254) next to Frame's float grid constants: identical for rectified input.
    // A sparsified KeyFrame (KeyFrame::EraseBadDescriptor, KeyFrame.cc:311-361) has given its grid away (:355), and
    // KeyFrame::GetFeaturesInArea returns nothing for it (:800-801): every window search on it — Fuse, the Sim3 /
    // relocalisation projections — finds no candidate in the reference.  It is uploaded without features for the same result.
    template <class KeyFramePtr>
    void UploadKeyFrame(const KeyFramePtr& pKF) {
        const auto keys = pKF->GetAllKeyUn();
        static_assert(sizeof(keys[0]) == sizeof(msorb_keypoint), "cv::KeyPoint must be the 28-byte layout");
        const int n = pKF->mbSparsified ? 0 : (int)keys.size();
        desc_.assign((size_t)n * 32, 0);
        std::vector<float> ur(n);
        for (int i = 0; i < n; i++) {
            const auto d = pKF->GetDescriptor(i);
            if (!d.empty()) std::memcpy(&desc_[(size_t)i * 32], d.template ptr<unsigned char>(0), 32);
            ur[i] = pKF->GetuRight(i);
        }
        check(msorb_frame_set(h_, reinterpret_cast<const msorb_keypoint*>(keys.data()), n, desc_.data(), ur.data(),
                              (float)pKF->mnMinX, (float)pKF->mnMaxX, (float)pKF->mnMinY, (float)pKF->mnMaxY,
                              pKF->mvScaleFactors.data(), (int)pKF->mvScaleFactors.size()),
              "msorb_frame_set");
    }
    msorb_frame* get() const { return h_; }

private:
    msorb_frame* h_ = nullptr;
    std::vector<uint8_t> desc_;
};

// pointer -> index table of one call (open addressing, power-of-two capacity, no allocation per entry: the std::unordered_map it
// replaces cost more per tracking frame than the device search it prepares)
class PointerIndex {
public:
    void reset(size_t n) {
        size_t cap = 64;
        while (cap < 2 * n + 2) cap <<= 1;
        if (key_.size() != cap) { key_.assign(cap, nullptr); val_.resize(cap); }
        else std::fill(key_.begin(), key_.end(), nullptr);
        mask_ = cap - 1;
    }
    // first insertion wins; returns the stored index
    int emplace(const void* k, int v) {
        size_t i = slot(k);
        while (key_[i] && key_[i] != k) i = (i + 1) & mask_;
        if (!key_[i]) { key_[i] = k; val_[i] = v; }
        return val_[i];
    }
    int find(const void* k) const {
        size_t i = slot(k);
        while (key_[i] && key_[i] != k) i = (i + 1) & mask_;
        return key_[i] ? val_[i] : -1;
    }
private:
    size_t slot(const void* k) const { return (size_t)((reinterpret_cast<uintptr_t>(k) >> 4) * 0x9E3779B97F4A7C15ull >> 20) & mask_; }
    std::vector<const void*> key_;
    std::vector<int> val_;
    size_t mask_ = 0;
};
// the marshalling arrays of SearchByProjection(F, vpMapPoints): kept per calling thread between frames (13 vectors of a few
// thousand entries: allocating and clearing them every frame was a third of the call)
struct LocalPointsScratch {
    PointerIndex index;
    std::vector<int> frameMp, before, extraObs, level, obs;
    std::vector<uint8_t> inView, bad, spars, desc;
    std::vector<float> px, py, pxr, depth, vcos;
};

// ORBmatcher::SearchByProjection(Frame &F, const vector<shared_ptr<MapPoint>> &vpMapPoints, th, bFarPoints, thFarPoints)
template <class FrameT, class MapPointPtr>
int SearchByProjection(DeviceFrame<FrameT>& dev, FrameT& F, const std::vector<MapPointPtr>& vpMapPoints, const float th,
                       const bool bFarPoints, const float thFarPoints, const float mfNNratio) {
    const int M = (int)vpMapPoints.size(), N = (int)F.mvpMapPoints.size();
    static thread_local LocalPointsScratch S;
    // table = the local map points in call order, then the map points the frame already holds that are not among them
    // (they are never queries — track_in_view 0 — but their Observations() decides whether a keypoint is taken, :89-91)
    S.index.reset((size_t)M + N);
    for (int i = 0; i < M; i++) S.index.emplace(vpMapPoints[i].get(), i);  // first occurrence wins, like the scan order
    S.frameMp.assign(N, -1);
    S.extraObs.clear();
    for (int i = 0; i < N; i++) {
        if (!F.mvpMapPoints[i]) continue;
        const int at = S.index.emplace(F.mvpMapPoints[i].get(), M + (int)S.extraObs.size());
        S.frameMp[i] = at;
        if (at == M + (int)S.extraObs.size()) S.extraObs.push_back(F.mvpMapPoints[i]->Observations());
    }
    const int T = M + (int)S.extraObs.size();
    S.inView.assign(T, 0); S.bad.assign(T, 0); S.spars.assign(T, 0); S.desc.resize((size_t)T * 32);
    S.px.resize(T); S.py.resize(T); S.pxr.resize(T); S.depth.resize(T); S.vcos.resize(T);
    S.level.resize(T); S.obs.resize(T);
    for (int i = 0; i < M; i++) {
        const auto& p = vpMapPoints[i];
        S.inView[i] = p->mbTrackInView; S.bad[i] = p->isBad(); S.spars[i] = p->mbSparsified;
        S.px[i] = p->mTrackProjX; S.py[i] = p->mTrackProjY; S.pxr[i] = p->mTrackProjXR; S.depth[i] = p->mTrackDepth;
        S.level[i] = p->mnTrackScaleLevel; S.vcos[i] = p->mTrackViewCos; S.obs[i] = p->Observations();
        // MapPoint::GetDescriptor clones under the point's mutex (MapPoint.cc:431-435): only for the points that are queries
        // (ORBmatcher.cc:53-59 skips the others before it would fetch theirs)
        if (S.inView[i] && !S.bad[i]) {
            const auto d = p->GetDescriptor();
            std::memcpy(&S.desc[(size_t)i * 32], d.template ptr<unsigned char>(0), 32);
        }
    }
    for (int k = 0; k < (int)S.extraObs.size(); k++) {
        S.obs[M + k] = S.extraObs[k];
        S.px[M + k] = S.py[M + k] = S.pxr[M + k] = S.depth[M + k] = S.vcos[M + k] = 0.f; S.level[M + k] = 0;
    }
    S.before = S.frameMp;
    int nmatches = 0;
    check(msorb_search_by_projection_mps(dev.get(), T, S.inView.data(), S.bad.data(), S.spars.data(), S.px.data(), S.py.data(),
                                         S.pxr.data(), S.depth.data(), S.level.data(), S.vcos.data(), S.desc.data(), S.obs.data(),
                                         S.frameMp.data(), th, bFarPoints ? 1 : 0, thFarPoints, mfNNratio, &nmatches),
          "msorb_search_by_projection_mps");
    for (int i = 0; i < N; i++)
        if (S.frameMp[i] != S.before[i] && S.frameMp[i] >= 0 && S.frameMp[i] < M) F.mvpMapPoints[i] = vpMapPoints[S.frameMp[i]];
    return nmatches;
}

// ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono)
// (:1941-2152, rectified / Nleft == -1): the per-keypoint projection of :1962-1990 stays the reference's own code (same
// expressions, compiled with the application's flags); window search, sequential claims and the rotation histogram
// run behind msorb_search_by_projection_frames.  `dev` holds CurrentFrame (dev.Upload(CurrentFrame) after extraction).
struct LastFrameProjection {  // what :1962-1990 computes per last-frame keypoint
    std::vector<uint8_t> valid, desc;
    std::vector<float> u, v, ur, angle;
    std::vector<int> octave, obs;
    bool forward = false, backward = false;
};
template <class FrameT>
void ProjectLastFrame(const FrameT& CurrentFrame, const FrameT& LastFrame, bool bMono, LastFrameProjection& P) {
    const auto Tcw = CurrentFrame.GetPose();
    const auto twc = Tcw.inverse().translation();
    const auto Tlw = LastFrame.GetPose();
    const auto tlc = Tlw * twc;
    P.forward = tlc(2) > CurrentFrame.mb && !bMono;                       // :1957
    P.backward = -tlc(2) > CurrentFrame.mb && !bMono;                     // :1958
    const int n = LastFrame.N;
    P.valid.assign(n, 0); P.desc.resize((size_t)n * 32);   // (descriptors of invalid entries are never read)
    P.u.assign(n, 0); P.v.assign(n, 0); P.ur.assign(n, 0); P.angle.assign(n, 0);
    P.octave.assign(n, 0); P.obs.assign(n, 0);
    for (int i = 0; i < n; i++) {
        const auto& pMP = LastFrame.mvpMapPoints[i];
        if (!pMP) continue;
        if (LastFrame.mvbOutlier[i]) continue;
        const auto x3Dw = pMP->GetWorldPos();
        const auto x3Dc = Tcw * x3Dw;
        const float invzc = 1.0 / x3Dc(2);                                // :1973
        if (invzc < 0) continue;
        const auto uv = CurrentFrame.mpCamera->project(x3Dc);
        if (uv(0) < CurrentFrame.mnMinX || uv(0) > CurrentFrame.mnMaxX) continue;
        if (uv(1) < CurrentFrame.mnMinY || uv(1) > CurrentFrame.mnMaxY) continue;
        P.valid[i] = 1;
        P.u[i] = uv(0);
        P.v[i] = uv(1);
        P.ur[i] = uv(0) - CurrentFrame.mbf * invzc;                       // :2019
        P.octave[i] = LastFrame.mvKeys[i].octave;                         // :1986
        P.angle[i] = LastFrame.mvKeysUn[i].angle;                         // :2044
        P.obs[i] = pMP->Observations();
        const auto d = pMP->GetDescriptor();
        std::memcpy(&P.desc[(size_t)i * 32], d.template ptr<unsigned char>(0), 32);
    }
}
template <class FrameT>
int SearchByProjection(DeviceFrame<FrameT>& dev, FrameT& CurrentFrame, const FrameT& LastFrame, const float th, const bool bMono,
                       const bool mbCheckOrientation) {
    // (the marshalling arrays live per calling thread between frames, like LocalPointsScratch)
    static thread_local LastFrameProjection P;
    static thread_local std::vector<int> lastMp, obs, curMp, before;
    ProjectLastFrame(CurrentFrame, LastFrame, bMono, P);
    const int nL = LastFrame.N, N = CurrentFrame.N;
    // ids: last-frame keypoint i -> i; map points the current frame already holds -> nL + k (only their Observations()
    // matter: a keypoint whose point has observations is never overwritten, :2011-2013)
    lastMp.resize(nL); obs.assign(P.obs.begin(), P.obs.end()); curMp.assign(N, -1);
    for (int i = 0; i < nL; i++) lastMp[i] = i;
    for (int j = 0; j < N; j++)
        if (CurrentFrame.mvpMapPoints[j]) {
            curMp[j] = (int)obs.size();
            obs.push_back(CurrentFrame.mvpMapPoints[j]->Observations());
        }
    before = curMp;
    int nmatches = 0;
    check(msorb_search_by_projection_frames(dev.get(), nL, P.valid.data(), P.u.data(), P.v.data(), P.ur.data(), P.octave.data(),
                                            P.angle.data(), P.desc.data(), lastMp.data(), obs.data(), (int)obs.size(), curMp.data(), th,
                                            P.forward, P.backward, mbCheckOrientation, &nmatches),
          "msorb_search_by_projection_frames");
    for (int j = 0; j < N; j++) {
        if (curMp[j] == before[j]) continue;
        if (curMp[j] < 0) CurrentFrame.mvpMapPoints[j] = nullptr;         // removed by the histogram filter (:2143)
        else CurrentFrame.mvpMapPoints[j] = LastFrame.mvpMapPoints[curMp[j]];   // :2037
    }
    return nmatches;
}

// The same method with the projection of :1962-1990 ON THE DEVICE (msorb_frame_set_last_points + msorb_search_last_frame):
// the host gathers what the loop reads of LastFrame — map point present and not an outlier, GetWorldPos(), GetDescriptor(),
// the keypoint's octave and angle — and the pose as Sophus holds it (unit quaternion + translation); Tcw * x3Dw,
// Pinhole::project, the bounds tests, the level band, the window search run on the device, the claims and the rotation
// histogram behind the C ABI.  Float convention of the device projection: include/msorb.h / DESIGN.md — the form above,
// which projects with the application's own build, stays the default of ORBmatcher::SearchByProjection; this one is for a
// Tracking that wants the table resident (the retry at 2 * th, Tracking.cc:2861-2868, is `resident = true`: no second upload).
// Pinhole camera only (mpCamera->getParameter(0..3) = fx, fy, cx, cy).
template <class FrameT>
void UploadLastFramePoints(DeviceFrame<FrameT>& dev, const FrameT& LastFrame, std::vector<int>& obs) {
    const int n = LastFrame.N;
    std::vector<uint8_t> has(n, 0), desc((size_t)n * 32, 0);
    std::vector<float> pos((size_t)n * 3, 0.0f), angle(n, 0.0f);
    std::vector<int> octave(n, 0);
    obs.assign(n, 0);
    for (int i = 0; i < n; i++) {
        const auto& pMP = LastFrame.mvpMapPoints[i];
        if (!pMP || LastFrame.mvbOutlier[i]) continue;                    // :1962-1965
        has[i] = 1;
        const auto x3Dw = pMP->GetWorldPos();
        pos[3 * i] = x3Dw(0); pos[3 * i + 1] = x3Dw(1); pos[3 * i + 2] = x3Dw(2);
        octave[i] = LastFrame.mvKeys[i].octave;                           // :1986
        angle[i] = LastFrame.mvKeysUn[i].angle;                           // :2044
        obs[i] = pMP->Observations();
        const auto d = pMP->GetDescriptor();
        std::memcpy(&desc[(size_t)i * 32], d.template ptr<unsigned char>(0), 32);
    }
    check(msorb_frame_set_last_points(dev.get(), n, has.data(), pos.data(), octave.data(), angle.data(), desc.data()),
          "msorb_frame_set_last_points");
}
template <class FrameT>
msorb_motion_model MotionModelOf(const FrameT& CurrentFrame, const FrameT& LastFrame, bool bMono) {
    msorb_motion_model m{};
    const auto Tcw = CurrentFrame.GetPose();
    const auto twc = Tcw.inverse().translation();
    const auto Tlw = LastFrame.GetPose();
    const auto tlc = Tlw * twc;
    m.forward = tlc(2) > CurrentFrame.mb && !bMono;                        // :1957
    m.backward = -tlc(2) > CurrentFrame.mb && !bMono;                      // :1958
    const auto q = Tcw.unit_quaternion();
    m.q[0] = q.x(); m.q[1] = q.y(); m.q[2] = q.z(); m.q[3] = q.w();
    const auto t = Tcw.translation();
    m.t[0] = t(0); m.t[1] = t(1); m.t[2] = t(2);
    m.fx = CurrentFrame.mpCamera->getParameter(0); m.fy = CurrentFrame.mpCamera->getParameter(1);
    m.cx = CurrentFrame.mpCamera->getParameter(2); m.cy = CurrentFrame.mpCamera->getParameter(3);
    m.mbf = CurrentFrame.mbf;
    return m;
}
template <class FrameT>
int SearchByProjectionDeviceProjected(DeviceFrame<FrameT>& dev, FrameT& CurrentFrame, const FrameT& LastFrame, const float th,
                                      const bool bMono, const bool mbCheckOrientation, std::vector<int>& lastObs, bool resident = false) {
    if (!resident) UploadLastFramePoints(dev, LastFrame, lastObs);
    const msorb_motion_model mm = MotionModelOf(CurrentFrame, LastFrame, bMono);
    const int nL = LastFrame.N, N = CurrentFrame.N;
    std::vector<int> obs(lastObs), curMp(N, -1);
    for (int j = 0; j < N; j++)
        if (CurrentFrame.mvpMapPoints[j]) {
            curMp[j] = (int)obs.size();
            obs.push_back(CurrentFrame.mvpMapPoints[j]->Observations());
        }
    const std::vector<int> before(curMp);
    int nmatches = 0;
    check(msorb_search_last_frame(dev.get(), &mm, obs.data(), (int)obs.size(), curMp.data(), th, mbCheckOrientation, &nmatches,
                                  nullptr, nullptr, nullptr, nullptr),
          "msorb_search_last_frame");
    for (int j = 0; j < N; j++) {
        if (curMp[j] == before[j]) continue;
        if (curMp[j] < 0) CurrentFrame.mvpMapPoints[j] = nullptr;         // removed by the histogram filter (:2143)
        else if (curMp[j] < nL) CurrentFrame.mvpMapPoints[j] = LastFrame.mvpMapPoints[curMp[j]];   // :2037
    }
    return nmatches;
}

// The projection loop of Tracking::SearchLocalPoints (src/Tracking.cc:3343-3361): mCurrentFrame.isInFrustum(pMP, 0.5)
// (src/Frame.cc:512-571, pinhole branch) for every local map point that is not already matched in this frame and not
// bad, in one device call.  Writes the same MapPoint scratch fields, calls IncreaseVisible() and fills
// F.mmProjectPoints exactly like the loop; returns nToMatch.  MapPoint needs two trivial accessors next to
// GetMaxDistanceInvariance(): `float GetMaxDistance()` / `float GetMinDistance()` returning mfMaxDistance / mfMinDistance
// (PredictScale divides the raw value, MapPoint.cc:562; the members are protected).
template <class FrameT, class MapPointPtr>
int SearchLocalPointsPrepass(FrameT& F, const std::vector<MapPointPtr>& vpLocalMapPoints, float viewingCosLimit = 0.5f,
                             int device = 0) {
    std::vector<int> which;
    which.reserve(vpLocalMapPoints.size());
    for (int i = 0; i < (int)vpLocalMapPoints.size(); i++) {
        const auto& p = vpLocalMapPoints[i];
        if (p->mnLastFrameSeen == F.mnId) continue;  // :3347-3348
        if (p->isBad()) continue;                    // :3349-3350
        which.push_back(i);
    }
    const int n = (int)which.size();
    if (n == 0) return 0;
    msorb_frustum fr{};
    const auto Tcw = F.GetPose();
    const auto R = Tcw.rotationMatrix();             // == mRcw (Frame::UpdatePoseMatrices, Frame.cc:472-479)
    const auto t = Tcw.translation();                // == mtcw
    const auto Ow = F.GetCameraCenter();             // == mOw
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) fr.Rcw[3 * r + c] = R(r, c);
        fr.tcw[r] = t(r);
        fr.Ow[r] = Ow(r);
    }
    fr.fx = F.mpCamera->getParameter(0); fr.fy = F.mpCamera->getParameter(1);
    fr.cx = F.mpCamera->getParameter(2); fr.cy = F.mpCamera->getParameter(3);
    fr.min_x = F.mnMinX; fr.max_x = F.mnMaxX; fr.min_y = F.mnMinY; fr.max_y = F.mnMaxY;
    fr.mbf = F.mbf; fr.log_scale_factor = F.mfLogScaleFactor; fr.n_scale_levels = F.mnScaleLevels;
    std::vector<float> pos((size_t)3 * n), nrm((size_t)3 * n), maxd(n), mind(n), px(n), py(n), pxr(n), depth(n), vcos(n);
    std::vector<int> level(n);
    std::vector<uint8_t> inView(n);
    for (int k = 0; k < n; k++) {
        const auto& p = vpLocalMapPoints[which[k]];
        const auto P = p->GetWorldPos();
        const auto N = p->GetNormal();
        for (int c = 0; c < 3; c++) { pos[3 * k + c] = P(c); nrm[3 * k + c] = N(c); }
        maxd[k] = p->GetMaxDistance();
        mind[k] = p->GetMinDistance();
    }
    check(msorb_is_in_frustum(device, &fr, viewingCosLimit, n, pos.data(), nrm.data(), maxd.data(), mind.data(), inView.data(),
                              px.data(), py.data(), pxr.data(), depth.data(), level.data(), vcos.data(), nullptr),
          "msorb_is_in_frustum");
    int nToMatch = 0;
    for (int k = 0; k < n; k++) {
        const auto& p = vpLocalMapPoints[which[k]];
        p->mbTrackInView = inView[k] != 0;           // Frame.cc:515-517, 563
        p->mTrackProjX = px[k];
        p->mTrackProjY = py[k];
        if (inView[k]) {                             // :563-571
            p->mTrackProjXR = pxr[k];
            p->mTrackDepth = depth[k];
            p->mnTrackScaleLevel = level[k];
            p->mTrackViewCos = vcos[k];
            p->IncreaseVisible();                    // Tracking.cc:3354-3355
            nToMatch++;
            F.mmProjectPoints[p->mnId] = {p->mTrackProjX, p->mTrackProjY};  // :3357-3360
        }
    }
    return nToMatch;
}

// Tracking::SearchLocalPoints from its second loop on (src/Tracking.cc:3343-3388) as ONE device chain: the isInFrustum loop
// above AND matcher.SearchByProjection(mCurrentFrame, mvpLocalMapPoints, th, bFarPoints, thFarPoints) behind
// msorb_search_local_points — one upload of the local map, frustum test + window queries + window search on the device, one
// read-back, the sequential claims replayed on the host.  Same MapPoint scratch fields, IncreaseVisible(), mmProjectPoints
// and F.mvpMapPoints as SearchLocalPointsPrepass followed by SearchByProjection; returns nmatches (*nToMatch = the loop's
// counter).  `dev` holds F (dev.Upload(F), or filled on the device by ExtractStereoFrame).
template <class FrameT, class MapPointPtr>
int SearchLocalPoints(DeviceFrame<FrameT>& dev, FrameT& F, const std::vector<MapPointPtr>& vpLocalMapPoints, const float th,
                      const bool bFarPoints, const float thFarPoints, const float mfNNratio, int* nToMatch = nullptr,
                      float viewingCosLimit = 0.5f) {
    const int M = (int)vpLocalMapPoints.size(), N = (int)F.mvpMapPoints.size();
    std::unordered_map<const void*, int> index;
    index.reserve((size_t)M * 2);
    for (int i = 0; i < M; i++) index.emplace(vpLocalMapPoints[i].get(), i);
    std::vector<int> frameMp(N, -1), extraObs;
    for (int i = 0; i < N; i++) {   // map points the frame holds that are not local: only their Observations() matter (:88-90)
        if (!F.mvpMapPoints[i]) continue;
        auto it = index.find(F.mvpMapPoints[i].get());
        if (it != index.end()) { frameMp[i] = it->second; continue; }
        frameMp[i] = M + (int)extraObs.size();
        index.emplace(F.mvpMapPoints[i].get(), frameMp[i]);
        extraObs.push_back(F.mvpMapPoints[i]->Observations());
    }
    const int T = M + (int)extraObs.size();
    msorb_frustum fr{};
    const auto Tcw = F.GetPose();
    const auto R = Tcw.rotationMatrix();
    const auto t = Tcw.translation();
    const auto Ow = F.GetCameraCenter();
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) fr.Rcw[3 * r + c] = R(r, c);
        fr.tcw[r] = t(r);
        fr.Ow[r] = Ow(r);
    }
    fr.fx = F.mpCamera->getParameter(0); fr.fy = F.mpCamera->getParameter(1);
    fr.cx = F.mpCamera->getParameter(2); fr.cy = F.mpCamera->getParameter(3);
    fr.min_x = F.mnMinX; fr.max_x = F.mnMaxX; fr.min_y = F.mnMinY; fr.max_y = F.mnMaxY;
    fr.mbf = F.mbf; fr.log_scale_factor = F.mfLogScaleFactor; fr.n_scale_levels = F.mnScaleLevels;
    std::vector<float> pos((size_t)3 * T, 0.f), nrm((size_t)3 * T, 0.f), maxd(T, 0.f), mind(T, 0.f), px(T), py(T), pxr(T), depth(T), vcos(T);
    std::vector<int> level(T), obs(T, 0);
    std::vector<uint8_t> visit(T, 0), bad(T, 0), spars(T, 0), desc((size_t)T * 32, 0), inView(T);
    for (int i = 0; i < M; i++) {
        const auto& p = vpLocalMapPoints[i];
        bad[i] = p->isBad();
        visit[i] = !(p->mnLastFrameSeen == F.mnId) && !bad[i];           // Tracking.cc:3347-3350
        spars[i] = p->mbSparsified;
        obs[i] = p->Observations();
        const auto P = p->GetWorldPos();
        const auto Nn = p->GetNormal();
        for (int c = 0; c < 3; c++) { pos[3 * i + c] = P(c); nrm[3 * i + c] = Nn(c); }
        maxd[i] = p->GetMaxDistance();
        mind[i] = p->GetMinDistance();
        const auto d = p->GetDescriptor();
        std::memcpy(&desc[(size_t)i * 32], d.template ptr<unsigned char>(0), 32);
    }
    for (int k = 0; k < (int)extraObs.size(); k++) obs[M + k] = extraObs[k];
    const std::vector<int> before = frameMp;
    int nmatches = 0;
    check(msorb_search_local_points(dev.get(), &fr, viewingCosLimit, T, pos.data(), nrm.data(), maxd.data(), mind.data(), visit.data(),
                                    bad.data(), spars.data(), desc.data(), obs.data(), frameMp.data(), th, bFarPoints ? 1 : 0,
                                    thFarPoints, mfNNratio, inView.data(), px.data(), py.data(), pxr.data(), depth.data(), level.data(),
                                    vcos.data(), &nmatches),
          "msorb_search_local_points");
    int nIn = 0;
    for (int i = 0; i < M; i++) {
        if (!visit[i]) continue;
        const auto& p = vpLocalMapPoints[i];
        p->mbTrackInView = inView[i] != 0;           // Frame.cc:515-517, 563
        p->mTrackProjX = px[i];
        p->mTrackProjY = py[i];
        if (inView[i]) {                             // :563-571
            p->mTrackProjXR = pxr[i];
            p->mTrackDepth = depth[i];
            p->mnTrackScaleLevel = level[i];
            p->mTrackViewCos = vcos[i];
            p->IncreaseVisible();                    // Tracking.cc:3354-3355
            nIn++;
            F.mmProjectPoints[p->mnId] = {p->mTrackProjX, p->mTrackProjY};  // :3357-3360
        }
    }
    if (nToMatch) *nToMatch = nIn;
    for (int i = 0; i < N; i++)
        if (frameMp[i] != before[i] && frameMp[i] >= 0 && frameMp[i] < M) F.mvpMapPoints[i] = vpLocalMapPoints[frameMp[i]];
    return nmatches;
}

// ---- SearchByBoW ----------------------------------------------------------------------------------------------
struct BowSide {  // one KeyFrame / Frame flattened for msorb_bow_pair
    std::vector<uint8_t> desc, flag;
    std::vector<int> node, begin, feat;
    std::vector<float> angle;
    // row(i) -> cv::Mat with the 32 descriptor bytes of feature i (Frame: mDescriptors.row(i); KeyFrame: GetDescriptor(i),
    // the public accessor — mDescriptors / mvKeysUn / mvuRight are protected members of MS-SLAM's KeyFrame)
    template <class RowFn, class FeatVecT, class KeysT>
    void Fill(int n, RowFn row, const FeatVecT& fv, const KeysT& keys) {
        desc.assign((size_t)n * 32, 0);
        for (int i = 0; i < n; i++) {
            const auto d = row(i);
            if (!d.empty()) std::memcpy(&desc[(size_t)i * 32], d.template ptr<unsigned char>(0), 32);
        }
        angle.resize(n);
        for (int i = 0; i < n; i++) angle[i] = keys[i].angle;
        node.clear(); feat.clear(); begin.assign(1, 0);
        for (const auto& e : fv) {  // std::map: ascending node id
            node.push_back((int)e.first);
            for (unsigned idx : e.second) feat.push_back((int)idx);
            begin.push_back((int)feat.size());
        }
    }
    template <class KeyFramePtr>
    void FillKeyFrame(const KeyFramePtr& pKF) {
        Fill(pKF->GetN(), [&](int i) { return pKF->GetDescriptor(i); }, pKF->GetFeatureVector(), pKF->GetAllKeyUn());
    }
    template <class MapPointPtr>
    void FlagGood(const std::vector<MapPointPtr>& mps) {  // pMP && !pMP->isBad()
        flag.assign(desc.size() / 32, 0);
        for (size_t i = 0; i < mps.size() && i < flag.size(); i++) flag[i] = mps[i] && !mps[i]->isBad();
    }
};
inline void BindBowPair(msorb_bow_pair& P, const BowSide& a, const BowSide& b, bool b_all_available, std::vector<int>& m12,
                 std::vector<int>& m21) {
    P = msorb_bow_pair{};
    P.n1 = (int)a.angle.size(); P.n2 = (int)b.angle.size();
    m12.assign(P.n1, -1); m21.assign(P.n2, -1);
    P.desc1 = a.desc.data(); P.desc2 = b.desc.data();
    P.valid1 = a.flag.data(); P.avail2 = b_all_available ? nullptr : b.flag.data();
    P.fv1_nodes = (int)a.node.size(); P.fv1_node = a.node.data(); P.fv1_begin = a.begin.data(); P.fv1_feat = a.feat.data();
    P.fv2_nodes = (int)b.node.size(); P.fv2_node = b.node.data(); P.fv2_begin = b.begin.data(); P.fv2_feat = b.feat.data();
    P.angle1 = a.angle.data(); P.angle2 = b.angle.data();
    P.match12 = m12.data(); P.match21 = m21.data();
}

// for(each candidate pKF) nmatches = matcher.SearchByBoW(pKF, F, vvpMapPointMatches[i]) in ONE device launch
// (Tracking::Relocalization, Tracking.cc:3577-3600; one candidate = ORBmatcher::SearchByBoW(pKF, F, ...), :223-421).
template <class KeyFramePtr, class FrameT, class MapPointPtr>
std::vector<int> SearchByBoWBatch(const std::vector<KeyFramePtr>& vpKFs, FrameT& F,
                                  std::vector<std::vector<MapPointPtr>>& vvpMapPointMatches, float mfNNratio,
                                  bool mbCheckOrientation, int device = 0) {
    const size_t K = vpKFs.size();
    BowSide frame;
    frame.Fill(F.N, [&](int i) { return F.mDescriptors.row(i); }, F.mFeatVec, F.mvKeys);
    std::vector<BowSide> kf(K);
    std::vector<std::vector<MapPointPtr>> mpsKF(K);
    std::vector<msorb_bow_pair> pairs(K);
    std::vector<std::vector<int>> m12(K), m21(K);
    for (size_t k = 0; k < K; k++) {
        mpsKF[k] = vpKFs[k]->GetMapPointMatches();                         // :225
        kf[k].FillKeyFrame(vpKFs[k]);
        kf[k].FlagGood(mpsKF[k]);                                          // :253-259
        BindBowPair(pairs[k], kf[k], frame, true, m12[k], m21[k]);
    }
    check(msorb_search_by_bow(device, pairs.data(), (int)K, 50 /* TH_LOW */, 1, mfNNratio, mbCheckOrientation, nullptr),
          "msorb_search_by_bow");
    std::vector<int> nmatches(K);
    vvpMapPointMatches.resize(K);
    for (size_t k = 0; k < K; k++) {
        vvpMapPointMatches[k].assign(F.N, MapPointPtr());                  // :227
        for (int j = 0; j < (int)m21[k].size() && j < F.N; j++)
            if (m21[k][j] >= 0) vvpMapPointMatches[k][j] = mpsKF[k][m21[k][j]];   // :336
        nmatches[k] = pairs[k].nmatches;
    }
    return nmatches;
}
template <class KeyFramePtr, class FrameT, class MapPointPtr>
int SearchByBoW(const KeyFramePtr& pKF, FrameT& F, std::vector<MapPointPtr>& vpMapPointMatches, float mfNNratio,
                bool mbCheckOrientation, int device = 0) {
    std::vector<std::vector<MapPointPtr>> out;
    const int n = SearchByBoWBatch(std::vector<KeyFramePtr>{pKF}, F, out, mfNNratio, mbCheckOrientation, device)[0];
    vpMapPointMatches = std::move(out[0]);
    return n;
}
// ORBmatcher::SearchByBoW(pKF1, pKF2, vpMatches12) (:872-1016, GetNLeft() == -1)
template <class KeyFramePtr, class MapPointPtr>
int SearchByBoWKeyFrames(const KeyFramePtr& pKF1, const KeyFramePtr& pKF2, std::vector<MapPointPtr>& vpMatches12, float mfNNratio,
                bool mbCheckOrientation, int device = 0) {
    const auto mps1 = pKF1->GetMapPointMatches();
    const auto mps2 = pKF2->GetMapPointMatches();
    BowSide a, b;
    a.FillKeyFrame(pKF1);
    b.FillKeyFrame(pKF2);
    a.FlagGood(mps1);
    b.FlagGood(mps2);                                                      // :934-944
    msorb_bow_pair P;
    std::vector<int> m12, m21;
    BindBowPair(P, a, b, false, m12, m21);
    check(msorb_search_by_bow(device, &P, 1, 50 /* TH_LOW */, 0 /* '<', :959 */, mfNNratio, mbCheckOrientation, nullptr),
          "msorb_search_by_bow");
    vpMatches12.assign(mps1.size(), MapPointPtr());                        // :884
    for (size_t i = 0; i < m12.size() && i < mps1.size(); i++)
        if (m12[i] >= 0) vpMatches12[i] = mps2[m12[i]];                    // :963
    return P.nmatches;
}

// ORBmatcher::SearchByProjection(Frame &CurrentFrame, shared_ptr<KeyFrame> pKF, const set<shared_ptr<MapPoint>>
// &sAlreadyFound, const float th, const int ORBdist) (:2154-2275; Tracking::Relocalization).  Projection, distance test
// and PredictScale are the reference's code; the search runs behind msorb_search_by_projection_kf.  `dev` holds
// CurrentFrame.
struct KeyFrameProjection {
    std::vector<uint8_t> valid, desc;
    std::vector<float> u, v, angle;
    std::vector<int> level;
};
template <class FrameT, class KeyFramePtr, class MapPointSet>
void ProjectKeyFramePoints(FrameT& CurrentFrame, const KeyFramePtr& pKF, const MapPointSet& sAlreadyFound, KeyFrameProjection& P) {
    const auto Tcw = CurrentFrame.GetPose();
    const auto Ow = Tcw.inverse().translation();
    const auto vpMPs = pKF->GetMapPointMatches();
    const int n = (int)vpMPs.size();
    P.valid.assign(n, 0); P.desc.resize((size_t)n * 32);   // (descriptors of invalid entries are never read)
    P.u.assign(n, 0); P.v.assign(n, 0); P.angle.assign(n, 0); P.level.assign(n, 0);
    for (int i = 0; i < n; i++) {
        const auto& pMP = vpMPs[i];
        if (!pMP) continue;
        if (pMP->isBad() || sAlreadyFound.count(pMP)) continue;            // :2175
        const auto x3Dw = pMP->GetWorldPos();
        const auto x3Dc = Tcw * x3Dw;
        const auto uv = CurrentFrame.mpCamera->project(x3Dc);
        if (uv(0) < CurrentFrame.mnMinX || uv(0) > CurrentFrame.mnMaxX) continue;
        if (uv(1) < CurrentFrame.mnMinY || uv(1) > CurrentFrame.mnMaxY) continue;
        const auto PO = (x3Dw - Ow).eval();                                // Eigen::Vector3f PO = x3Dw-Ow
        const float dist3D = PO.norm();
        const float maxDistance = pMP->GetMaxDistanceInvariance();
        const float minDistance = pMP->GetMinDistanceInvariance();
        if (dist3D < minDistance || dist3D > maxDistance) continue;        // :2195
        P.valid[i] = 1;
        P.u[i] = uv(0);
        P.v[i] = uv(1);
        P.level[i] = pMP->PredictScale(dist3D, &CurrentFrame);             // :2198
        P.angle[i] = pKF->GetKeyUn(i).angle;                               // :2237
        const auto d = pMP->GetDescriptor();
        std::memcpy(&P.desc[(size_t)i * 32], d.template ptr<unsigned char>(0), 32);
    }
}
template <class FrameT, class KeyFramePtr, class MapPointSet>
int SearchByProjection(DeviceFrame<FrameT>& dev, FrameT& CurrentFrame, const KeyFramePtr& pKF, const MapPointSet& sAlreadyFound,
                       const float th, const int ORBdist, const bool mbCheckOrientation) {
    KeyFrameProjection P;
    ProjectKeyFramePoints(CurrentFrame, pKF, sAlreadyFound, P);
    const auto vpMPs = pKF->GetMapPointMatches();
    const int n = (int)vpMPs.size(), N = CurrentFrame.N;
    std::vector<int> ids(n), curMp(N, -1);
    for (int i = 0; i < n; i++) ids[i] = i;
    for (int j = 0; j < N; j++)
        if (CurrentFrame.mvpMapPoints[j]) curMp[j] = n + j;               // any held keypoint is taken, :2214-2215
    const std::vector<int> before(curMp);
    int nmatches = 0;
    check(msorb_search_by_projection_kf(dev.get(), n, P.valid.data(), P.u.data(), P.v.data(), P.level.data(), P.angle.data(),
                                        P.desc.data(), ids.data(), curMp.data(), th, ORBdist, mbCheckOrientation, &nmatches),
          "msorb_search_by_projection_kf");
    for (int j = 0; j < N; j++) {
        if (curMp[j] == before[j]) continue;
        if (curMp[j] < 0) CurrentFrame.mvpMapPoints[j] = nullptr;         // :2266
        else CurrentFrame.mvpMapPoints[j] = vpMPs[curMp[j]];              // :2231
    }
    return nmatches;
}

// ---- Fuse ---------------------------------------------------------------------------------------------------
// ORBmatcher::Fuse(pKF, vpMapPoints, th, bRight = false), :1404-1597.  `dev` holds pKF (dev.UploadKeyFrame(pKF)).
// Pass 1 evaluates the reference's own geometric tests (:1448-1500) for every point that is alive now; the window
// search with the level band and the reprojection-error gates (:1502-1561) runs on the device for all of them at once
// (it reads only the KeyFrame's features, which the loop never changes); pass 2 is the reference's loop again with the
// search replaced by a lookup: the isBad() / IsInKeyFrame() tests are repeated there because earlier iterations
// mutate the map (Replace / AddObservation, :1563-1588) — points can only become bad or observed, never the reverse,
// so the points searched in pass 1 are a superset of the points the loop reaches.
struct FuseQueries {  // what :1448-1500 computes per map point
    std::vector<uint8_t> valid, desc;
    std::vector<float> u, v, ur, radius;
    std::vector<int> level;
};
// Tcw / Ow / pCamera: the camera the call searches (:1410-1421) — GetPose / GetCameraCenter / mpCamera, or the right camera's of a
// two-camera KeyFrame (ORBmatcher_rig_device.h)
template <class KeyFramePtr, class MapPointPtr, class PoseT, class CenterT, class CameraT>
void FuseGeometryOf(const KeyFramePtr& pKF, const PoseT& Tcw, const CenterT& Ow, CameraT* pCamera, const std::vector<MapPointPtr>& vpMapPoints,
                    const float th, FuseQueries& Q) {
    const float& bf = pKF->mbf;
    const int nMPs = (int)vpMapPoints.size();
    Q.valid.assign(nMPs, 0); Q.desc.assign((size_t)nMPs * 32, 0);
    Q.u.assign(nMPs, 0); Q.v.assign(nMPs, 0); Q.ur.assign(nMPs, 0); Q.radius.assign(nMPs, 0);
    Q.level.assign(nMPs, 0);
    for (int i = 0; i < nMPs; i++) {
        const MapPointPtr& pMP = vpMapPoints[i];
        if (!pMP) continue;
        if (pMP->isBad()) continue;
        else if (pMP->IsInKeyFrame(pKF)) continue;
        const auto p3Dw = pMP->GetWorldPos();
        const auto p3Dc = Tcw * p3Dw;
        if (p3Dc(2) < 0.0f) continue;                                      // :1451
        const float invz = 1 / p3Dc(2);
        const auto uv = pCamera->project(p3Dc);
        if (!pKF->IsInImage(uv(0), uv(1))) continue;                       // :1462
        const float maxDistance = pMP->GetMaxDistanceInvariance();
        const float minDistance = pMP->GetMinDistanceInvariance();
        const auto PO = (p3Dw - Ow).eval();                                // Eigen::Vector3f PO = p3Dw-Ow
        const float dist3D = PO.norm();
        if (dist3D < minDistance || dist3D > maxDistance) continue;        // :1476
        const auto Pn = pMP->GetNormal();
        if (PO.dot(Pn) < 0.5 * dist3D) continue;                           // :1484
        const int nPredictedLevel = pMP->PredictScale(dist3D, pKF);
        Q.valid[i] = 1;
        Q.u[i] = uv(0);
        Q.v[i] = uv(1);
        Q.ur[i] = uv(0) - bf * invz;                                       // :1468
        Q.level[i] = nPredictedLevel;
        Q.radius[i] = th * pKF->mvScaleFactors[nPredictedLevel];           // :1495
        const auto dMP = pMP->GetDescriptor();
        std::memcpy(&Q.desc[(size_t)i * 32], dMP.template ptr<unsigned char>(0), 32);
    }
}
// pass 2: the reference's loop with the search replaced by a lookup (:1563-1590).  idxOffset = NLeft for the right camera of a
// two-camera KeyFrame (`if(bRight) idx += pKF->GetNLeft()`, :1547), 0 otherwise.
template <class KeyFramePtr, class MapPointPtr>
int FuseCommit(const KeyFramePtr& pKF, const std::vector<MapPointPtr>& vpMapPoints, const std::vector<uint8_t>& valid,
               const std::vector<int>& bestIdx, const std::vector<int>& bestDist, const int idxOffset) {
    const int nMPs = (int)vpMapPoints.size();
    int nFused = 0;
    for (int i = 0; i < nMPs; i++) {
        MapPointPtr pMP = vpMapPoints[i];
        if (!pMP) continue;
        if (pMP->isBad()) continue;
        else if (pMP->IsInKeyFrame(pKF)) continue;
        if (!valid[i]) continue;
        if (bestDist[i] <= 50 /* TH_LOW */) {                              // :1563-1590
            const int bestIdxKF = bestIdx[i] + idxOffset;
            auto pMPinKF = pKF->GetMapPoint(bestIdxKF);
            if (pMPinKF) {
                if (!pMPinKF->isBad()) {
                    if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
                    else pMPinKF->Replace(pMP);
                }
            } else {
                pMP->AddObservation(pKF, bestIdxKF);
                pKF->AddMapPoint(pMP, bestIdxKF);
            }
            nFused++;
        }
    }
    return nFused;
}
template <class KeyFramePtr, class MapPointPtr>
void FuseGeometry(const KeyFramePtr& pKF, const std::vector<MapPointPtr>& vpMapPoints, const float th, FuseQueries& Q) {
    FuseGeometryOf(pKF, pKF->GetPose(), pKF->GetCameraCenter(), pKF->mpCamera, vpMapPoints, th, Q);
}
template <class FrameT, class KeyFramePtr, class MapPointPtr>
int Fuse(DeviceFrame<FrameT>& dev, const KeyFramePtr& pKF, const std::vector<MapPointPtr>& vpMapPoints, const float th) {
    FuseQueries Q;
    FuseGeometry(pKF, vpMapPoints, th, Q);
    const int nMPs = (int)vpMapPoints.size();
    std::vector<int> bestIdx(nMPs, -1), bestDist(nMPs, 256);
    const std::vector<uint8_t>& valid = Q.valid;
    check(msorb_fuse_search(dev.get(), pKF->mvInvLevelSigma2.data(), (int)pKF->mvInvLevelSigma2.size(), nMPs, Q.valid.data(),
                            Q.u.data(), Q.v.data(), Q.ur.data(), Q.level.data(), Q.radius.data(), Q.desc.data(),
                            bestIdx.data(), bestDist.data()),
          "msorb_fuse_search");
    return FuseCommit(pKF, vpMapPoints, valid, bestIdx, bestDist, 0);
}

// ---- SearchForTriangulation ---------------------------------------------------------------------------------
// The per-pair geometry of ORBmatcher.cc:1174-1194 and the fundamental matrix Pinhole::epipolarConstrain rebuilds for
// every candidate (Pinhole.cpp:109-112) — the reference's own Eigen / Sophus expressions, evaluated once per pair.
template <class KeyFramePtr>
void TriangulationGeometry(const KeyFramePtr& pKF1, const KeyFramePtr& pKF2, float F12[9], float ep[2]) {
    const auto T1w = pKF1->GetPose();
    const auto T2w = pKF2->GetPose();
    const auto Tw2 = pKF2->GetPoseInverse();
    const auto Cw = pKF1->GetCameraCenter();
    const auto C2 = T2w * Cw;
    const auto e = pKF2->mpCamera->project(C2);
    ep[0] = e(0); ep[1] = e(1);
    const auto T12 = T1w * Tw2;
    const auto R12 = T12.rotationMatrix();
    const auto t12 = T12.translation();
    using SO3 = typename std::decay<decltype(T12.so3())>::type;
    const auto t12x = SO3::hat(t12);
    const auto K1 = pKF1->mpCamera->toK_();
    const auto K2 = pKF2->mpCamera->toK_();
    // .eval(): the product is an expression template in Eigen (no coefficient access, operands held by reference); evaluating it
    // into a plain matrix is what the reference's `Eigen::Matrix3f F12 = ...` does
    const auto F = (K1.transpose().inverse() * t12x * R12 * K2.inverse()).eval();
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) F12[3 * r + c] = F(r, c);
}

struct TriSide {  // one KeyFrame flattened for msorb_triangulation_pair
    BowSide bow;
    std::vector<uint8_t> free_, stereo;
    std::vector<msorb_keypoint> kp;
    template <class KeyFramePtr>
    void Fill(const KeyFramePtr& pKF, bool bOnlyStereo) {
        static_assert(sizeof(pKF->GetAllKeyUn()[0]) == sizeof(msorb_keypoint), "cv::KeyPoint must be the 28-byte layout");
        bow.FillKeyFrame(pKF);
        const auto keys = pKF->GetAllKeyUn();                             // GetKeyPoint(idx) for NLeft == -1
        const auto mps = pKF->GetMapPointMatches();                       // GetMapPoint(idx) for every idx, one lock
        const int n = (int)keys.size();
        kp.resize(n);
        if (n) std::memcpy(kp.data(), keys.data(), (size_t)n * sizeof(msorb_keypoint));
        free_.assign(n, 0);
        stereo.assign(n, 0);
        for (int i = 0; i < n; i++) {
            stereo[i] = pKF->GetuRight(i) >= 0;                           // :1243 / :1267 (mpCamera2 == nullptr)
            free_[i] = !mps[i] && (!bOnlyStereo || stereo[i]);            // :1237-1247 / :1264-1271
        }
    }
};

// for(each neighbour pKF2) matcher.SearchForTriangulation(pKF1, pKF2, vMatchedIndices, bOnlyStereo, bCoarse) in ONE launch
template <class KeyFramePtr>
std::vector<int> SearchForTriangulationBatch(const KeyFramePtr& pKF1, const std::vector<KeyFramePtr>& vpKF2,
                                             std::vector<std::vector<std::pair<size_t, size_t>>>& vvMatchedPairs,
                                             bool bOnlyStereo, bool bCoarse, bool mbCheckOrientation, int device = 0) {
    const size_t K = vpKF2.size();
    TriSide a;
    a.Fill(pKF1, bOnlyStereo);
    std::vector<TriSide> b(K);
    std::vector<msorb_triangulation_pair> pairs(K);
    std::vector<std::vector<int>> m12(K);
    for (size_t k = 0; k < K; k++) {
        b[k].Fill(vpKF2[k], bOnlyStereo);
        msorb_triangulation_pair& P = pairs[k];
        P = msorb_triangulation_pair{};
        P.n1 = (int)a.kp.size(); P.n2 = (int)b[k].kp.size();
        m12[k].assign(P.n1, -1);
        P.desc1 = a.bow.desc.data(); P.desc2 = b[k].bow.desc.data();
        P.valid1 = a.free_.data(); P.avail2 = b[k].free_.data();
        P.stereo1 = a.stereo.data(); P.stereo2 = b[k].stereo.data();
        P.fv1_nodes = (int)a.bow.node.size(); P.fv1_node = a.bow.node.data(); P.fv1_begin = a.bow.begin.data();
        P.fv1_feat = a.bow.feat.data();
        P.fv2_nodes = (int)b[k].bow.node.size(); P.fv2_node = b[k].bow.node.data(); P.fv2_begin = b[k].bow.begin.data();
        P.fv2_feat = b[k].bow.feat.data();
        P.kp1 = a.kp.data(); P.kp2 = b[k].kp.data();
        P.scale_factors2 = vpKF2[k]->mvScaleFactors.data();
        P.level_sigma2_2 = vpKF2[k]->mvLevelSigma2.data();
        P.n_levels2 = (int)vpKF2[k]->mvScaleFactors.size();
        TriangulationGeometry(pKF1, vpKF2[k], P.F12, P.ep);
        P.match12 = m12[k].data();
    }
    check(msorb_search_for_triangulation(device, pairs.data(), (int)K, bCoarse, mbCheckOrientation, nullptr),
          "msorb_search_for_triangulation");
    std::vector<int> nmatches(K);
    vvMatchedPairs.assign(K, {});
    for (size_t k = 0; k < K; k++) {
        nmatches[k] = pairs[k].nmatches;
        vvMatchedPairs[k].reserve(nmatches[k]);                            // :1385-1393
        for (size_t i = 0; i < m12[k].size(); i++)
            if (m12[k][i] >= 0) vvMatchedPairs[k].push_back(std::make_pair(i, (size_t)m12[k][i]));
    }
    return nmatches;
}
template <class KeyFramePtr>
int SearchForTriangulation(const KeyFramePtr& pKF1, const KeyFramePtr& pKF2, std::vector<std::pair<size_t, size_t>>& vMatchedPairs,
                           bool bOnlyStereo, bool bCoarse, bool mbCheckOrientation, int device = 0) {
    std::vector<std::vector<std::pair<size_t, size_t>>> out;
    const int n = SearchForTriangulationBatch(pKF1, std::vector<KeyFramePtr>{pKF2}, out, bOnlyStereo, bCoarse,
                                              mbCheckOrientation, device)[0];
    vMatchedPairs = std::move(out[0]);
    return n;
}

// ---- resident KeyFrames ------------------------------------------------------------------------------------------
// msorb_kf_store behind the reference's types: a KeyFrame enters the store the first time it is searched (its features are
// fixed once ComputeBoW has run) and leaves it
//   * when the KeyFrame object dies: MS-SLAM holds its KeyFrames in std::shared_ptr everywhere (include/ORBmatcher.h:47-96 — the
//     fork's defining change), so a slot keeps a std::weak_ptr to its KeyFrame and is dropped once that has expired (checked
//     on every add and on Resident(); no edit of the reference's sources is needed for this);
//   * when the optional budget is exceeded (SetBudget: resident KeyFrames / estimated device bytes): least recently searched
//     first; an evicted KeyFrame that is searched again is simply uploaded again;
//   * on Forget() (a one-line hook in KeyFrame::SetBadFlag: frees the slot at the moment of culling instead of at the next
//     add), Reset() and Shutdown();
//   * when map sparsification compacts it (mbSparsified flips, N shrinks: re-added under a new id).
// The searches below then move one flag byte per feature per call instead of both KeyFrames: 0.12-0.15 ms instead of
// 0.65-1.1 ms for 16 neighbours / 32 candidates (profiles/).
namespace detail {
// the owner a slot watches: a weak reference for std::shared_ptr<KeyFrame>, nothing for other pointer types (raw KeyFrame*:
// upstream ORB-SLAM3 — there the hooks are the only way out)
template <class P> struct OwnerOf {
    static constexpr bool tracked = false;
    static std::weak_ptr<const void> get(const P&) { return {}; }
};
template <class T> struct OwnerOf<std::shared_ptr<T>> {
    static constexpr bool tracked = true;
    static std::weak_ptr<const void> get(const std::shared_ptr<T>& p) { return std::weak_ptr<const void>(std::shared_ptr<const void>(p)); }
};
inline bool same_owner(const std::weak_ptr<const void>& a, const std::weak_ptr<const void>& b) { return !a.owner_before(b) && !b.owner_before(a); }
}  // namespace detail

class KeyFrameStore {
public:
    // One resident KeyFrame.  A search holds a Lease (shared ownership) from Ensure() until its kernels have returned: the
    // device rows are removed when the LAST holder lets go — a re-add after map sparsification, an eviction, Forget() or Reset()
    // on another thread only drops the table's reference, it never pulls an id out from under a running search.
    struct Handle {  // the C handle; shared with the slots so that a lease that outlives Shutdown() never touches a freed store
        msorb_kf_store* h = nullptr;
        ~Handle() { if (h) msorb_kf_store_destroy(h); }
    };
    struct Slot {
        std::shared_ptr<Handle> store;
        int id;
        const void* identity;  // the KeyFrame object: mnId alone is not an identity (KeyFrame::nNextId restarts at 0 in Tracking::Reset)
        bool sparsified;
        int n;
        bool tracked = false;                // owner is meaningful (the caller holds its KeyFrames in std::shared_ptr)
        std::weak_ptr<const void> owner;     // the KeyFrame object's ownership group: expired = the KeyFrame is gone; and the exact
                                             // identity test (a new object at a recycled address has another control block)
        size_t bytes = 0;                    // estimate of the device memory behind the slot
        unsigned long long last_use = 0;     // tick of the last Ensure() that returned it (guarded by the store's mutex)
        ~Slot() { if (store && store->h && id >= 0) msorb_kf_store_remove(store->h, id); }
    };
    using Lease = std::shared_ptr<Slot>;

    explicit KeyFrameStore(int device = 0) : h_(std::make_shared<Handle>()) { check_abi(); check(msorb_kf_store_create(device, &h_->h), "msorb_kf_store_create"); }
    ~KeyFrameStore() { Shutdown(); }
    KeyFrameStore(const KeyFrameStore&) = delete;
    KeyFrameStore& operator=(const KeyFrameStore&) = delete;
    msorb_kf_store* get() const { return h_ ? h_->h : nullptr; }

    // The lock is held across lookup, upload and insert: two threads that miss on the same KeyFrame (TrackReferenceKeyFrame's
    // SearchByBoW and CreateNewMapPoints' SearchForTriangulation on the newest KeyFrame) produce ONE add; the second waits
    // and finds the first one's slot.
    template <class KeyFramePtr>
    Lease Ensure(const KeyFramePtr& pKF) {
        using Owner = detail::OwnerOf<KeyFramePtr>;
        const int n = pKF->GetN();
        const void* identity = static_cast<const void*>(&*pKF);
        const std::weak_ptr<const void> owner = Owner::get(pKF);
        std::lock_guard<std::mutex> lk(mu_);
        if (!h_) throw std::runtime_error("msorb KeyFrameStore used after Shutdown()");
        auto it = ids_.find(pKF->mnId);
        if (it != ids_.end() && it->second->identity == identity && it->second->sparsified == pKF->mbSparsified && it->second->n == n &&
            (!Owner::tracked || !it->second->tracked || detail::same_owner(it->second->owner, owner))) {
            it->second->last_use = ++tick_;
            return it->second;
        }
        sweep_expired_locked();   // an add is the moment the store grows: dead KeyFrames leave first
        stats_.adds++;
        BowSide side;
        side.FillKeyFrame(pKF);
        const auto keys = pKF->GetAllKeyUn();
        static_assert(sizeof(keys[0]) == sizeof(msorb_keypoint), "cv::KeyPoint must be the 28-byte layout");
        int id = -1;
        check(msorb_kf_store_add(h_->h, n, reinterpret_cast<const msorb_keypoint*>(keys.data()), side.desc.data(), (int)side.node.size(),
                                 side.node.data(), side.begin.data(), side.feat.data(), pKF->mvScaleFactors.data(),
                                 pKF->mvLevelSigma2.data(), (int)pKF->mvScaleFactors.size(), &id),
              "msorb_kf_store_add");
        Lease slot(new Slot{h_, id, identity, (bool)pKF->mbSparsified, n});
        slot->tracked = Owner::tracked;
        slot->owner = owner;
        slot->bytes = (size_t)n * (32 + sizeof(msorb_keypoint) + 8) + side.node.size() * 8 + side.feat.size() * 4;
        slot->last_use = ++tick_;
        auto old = ids_.find(pKF->mnId);
        if (old != ids_.end()) bytes_ -= old->second->bytes;   // the compacted / recycled predecessor loses the table's reference; running searches keep theirs
        ids_[pKF->mnId] = slot;
        bytes_ += slot->bytes;
        enforce_budget_locked(slot.get());
        return slot;
    }
    void Forget(unsigned long mnId) {  // KeyFrame::SetBadFlag (optional hook: frees the slot at once)
        std::lock_guard<std::mutex> lk(mu_);
        auto it = ids_.find(mnId);
        if (it != ids_.end()) { bytes_ -= it->second->bytes; ids_.erase(it); }
    }
    void Reset() {  // Tracking::Reset / ResetActiveMap: KeyFrame ids start again at 0
        std::lock_guard<std::mutex> lk(mu_);
        ids_.clear();
        bytes_ = 0;
    }
    // Upper bounds on what stays resident (0 = unlimited, the default): beyond them the least recently searched KeyFrames leave.
    void SetBudget(size_t max_keyframes, size_t max_bytes) {
        std::lock_guard<std::mutex> lk(mu_);
        max_kfs_ = max_keyframes; max_bytes_ = max_bytes;
        enforce_budget_locked(nullptr);
    }
    size_t Resident() {   // live entries: KeyFrames that have died since the last add are swept first
        std::lock_guard<std::mutex> lk(mu_);
        sweep_expired_locked();
        return ids_.size();
    }
    size_t ResidentBytes() {
        std::lock_guard<std::mutex> lk(mu_);
        sweep_expired_locked();
        return bytes_;
    }
    struct Counters { unsigned long long adds = 0, expired = 0, evicted = 0; };
    Counters Stats() {
        std::lock_guard<std::mutex> lk(mu_);
        return stats_;
    }
    // System::Shutdown: free the device side while the HIP runtime is still alive.  The C handle goes when its last holder
    // does — here, unless a lease is still out (there should be none: the worker threads have been joined).
    void Shutdown() {
        std::lock_guard<std::mutex> lk(mu_);
        ids_.clear();
        bytes_ = 0;
        h_.reset();
    }

private:
    void sweep_expired_locked() {
        for (auto it = ids_.begin(); it != ids_.end();) {
            if (it->second->tracked && it->second->owner.expired()) { bytes_ -= it->second->bytes; it = ids_.erase(it); stats_.expired++; }
            else ++it;
        }
    }
    void enforce_budget_locked(const Slot* keep) {
        while ((max_kfs_ && ids_.size() > max_kfs_) || (max_bytes_ && bytes_ > max_bytes_)) {
            auto victim = ids_.end();
            for (auto it = ids_.begin(); it != ids_.end(); ++it)
                if (it->second.get() != keep && (victim == ids_.end() || it->second->last_use < victim->second->last_use)) victim = it;
            if (victim == ids_.end()) break;   // only the entry just added is left
            bytes_ -= victim->second->bytes;
            ids_.erase(victim);
            stats_.evicted++;
        }
    }
    std::shared_ptr<Handle> h_;
    std::mutex mu_;
    std::unordered_map<unsigned long, Lease> ids_;
    size_t bytes_ = 0, max_kfs_ = 0, max_bytes_ = 0;
    unsigned long long tick_ = 0;
    Counters stats_;
};

// SearchForTriangulationBatch with the KeyFrames resident (same results)
template <class KeyFramePtr>
std::vector<int> SearchForTriangulationBatch(KeyFrameStore& store, const KeyFramePtr& pKF1, const std::vector<KeyFramePtr>& vpKF2,
                                             std::vector<std::vector<std::pair<size_t, size_t>>>& vvMatchedPairs, bool bOnlyStereo,
                                             bool bCoarse, bool mbCheckOrientation) {
    const size_t K = vpKF2.size();
    struct Flags { std::vector<uint8_t> free_, stereo; };
    auto flags_of = [&](const KeyFramePtr& pKF, Flags& f) {
        const auto mps = pKF->GetMapPointMatches();
        const int n = (int)mps.size();
        f.free_.assign(n, 0); f.stereo.assign(n, 0);
        for (int i = 0; i < n; i++) {
            f.stereo[i] = pKF->GetuRight(i) >= 0;                          // :1243 / :1267
            f.free_[i] = !mps[i] && (!bOnlyStereo || f.stereo[i]);         // :1237-1247 / :1264-1271
        }
    };
    Flags a;
    flags_of(pKF1, a);
    const KeyFrameStore::Lease l1 = store.Ensure(pKF1);
    std::vector<KeyFrameStore::Lease> l2(K);   // held until the search has returned
    std::vector<Flags> b(K);
    std::vector<msorb_triangulation_kf_pair> pairs(K);
    std::vector<std::vector<int>> m12(K);
    for (size_t k = 0; k < K; k++) {
        flags_of(vpKF2[k], b[k]);
        msorb_triangulation_kf_pair& P = pairs[k];
        P = msorb_triangulation_kf_pair{};
        l2[k] = store.Ensure(vpKF2[k]);
        P.kf1 = l1->id; P.kf2 = l2[k]->id;
        P.valid1 = a.free_.data(); P.avail2 = b[k].free_.data(); P.stereo1 = a.stereo.data(); P.stereo2 = b[k].stereo.data();
        TriangulationGeometry(pKF1, vpKF2[k], P.F12, P.ep);
        m12[k].assign(a.free_.size(), -1);
        P.match12 = m12[k].data();
    }
    check(msorb_search_for_triangulation_kf(store.get(), pairs.data(), (int)K, bCoarse, mbCheckOrientation, nullptr),
          "msorb_search_for_triangulation_kf");
    std::vector<int> nmatches(K);
    vvMatchedPairs.assign(K, {});
    for (size_t k = 0; k < K; k++) {
        nmatches[k] = pairs[k].nmatches;
        vvMatchedPairs[k].reserve(nmatches[k]);
        for (size_t i = 0; i < m12[k].size(); i++)
            if (m12[k][i] >= 0) vvMatchedPairs[k].push_back(std::make_pair(i, (size_t)m12[k][i]));
    }
    return nmatches;
}

// SearchByBoWBatch (candidate KeyFrames against one Frame) with the KeyFrames resident (same results)
template <class KeyFramePtr, class FrameT, class MapPointPtr>
std::vector<int> SearchByBoWBatch(KeyFrameStore& store, const std::vector<KeyFramePtr>& vpKFs, FrameT& F,
                                  std::vector<std::vector<MapPointPtr>>& vvpMapPointMatches, float mfNNratio, bool mbCheckOrientation) {
    const size_t K = vpKFs.size();
    BowSide frame;
    frame.Fill(F.N, [&](int i) { return F.mDescriptors.row(i); }, F.mFeatVec, F.mvKeys);
    const msorb_bow_frame bf{F.N, frame.desc.data(), (int)frame.node.size(), frame.node.data(), frame.begin.data(), frame.feat.data(),
                             frame.angle.data()};
    std::vector<std::vector<MapPointPtr>> mpsKF(K);
    std::vector<std::vector<uint8_t>> good(K);
    std::vector<std::vector<int>> m12(K), m21(K);
    std::vector<msorb_bow_kf_pair> pairs(K);
    std::vector<KeyFrameStore::Lease> lease(K);   // held until the search has returned
    for (size_t k = 0; k < K; k++) {
        mpsKF[k] = vpKFs[k]->GetMapPointMatches();
        good[k].assign(mpsKF[k].size(), 0);
        for (size_t i = 0; i < mpsKF[k].size(); i++) good[k][i] = mpsKF[k][i] && !mpsKF[k][i]->isBad();   // :253-259
        m12[k].assign(mpsKF[k].size(), -1);
        m21[k].assign(F.N, -1);
        lease[k] = store.Ensure(vpKFs[k]);
        pairs[k] = msorb_bow_kf_pair{lease[k]->id, -1, good[k].data(), nullptr, m12[k].data(), m21[k].data(), 0};
    }
    check(msorb_search_by_bow_kf(store.get(), pairs.data(), (int)K, &bf, 50 /* TH_LOW */, 1, mfNNratio, mbCheckOrientation, nullptr),
          "msorb_search_by_bow_kf");
    std::vector<int> nmatches(K);
    vvpMapPointMatches.resize(K);
    for (size_t k = 0; k < K; k++) {
        vvpMapPointMatches[k].assign(F.N, MapPointPtr());
        for (int j = 0; j < F.N; j++)
            if (m21[k][j] >= 0) vvpMapPointMatches[k][j] = mpsKF[k][m21[k][j]];
        nmatches[k] = pairs[k].nmatches;
    }
    return nmatches;
}

// Frame::ComputeStereoMatches(): fills F.mvuRight / F.mvDepth from the two extractors' device pyramids.
// ExtractorT = the drop-in ORB_SLAM3::ORBextractor of this directory (handle()).
template <class FrameT, class ExtractorT>
void ComputeStereoMatches(FrameT& F, const ExtractorT& left, const ExtractorT& right) {
    const int N = (int)F.mvKeys.size(), Nr = (int)F.mvKeysRight.size();
    F.mvuRight.assign(N, -1.0f);
    F.mvDepth.assign(N, -1.0f);
    std::vector<uint8_t> dl((size_t)N * 32), dr((size_t)Nr * 32);
    for (int i = 0; i < N; i++) std::memcpy(&dl[(size_t)i * 32], F.mDescriptors.template ptr<unsigned char>(i), 32);
    for (int i = 0; i < Nr; i++) std::memcpy(&dr[(size_t)i * 32], F.mDescriptorsRight.template ptr<unsigned char>(i), 32);
    int oob = 0;
    check(msorb_stereo_matches(left.handle(), right.handle(), reinterpret_cast<const msorb_keypoint*>(F.mvKeys.data()), N,
                               dl.data(), reinterpret_cast<const msorb_keypoint*>(F.mvKeysRight.data()), Nr, dr.data(), F.mb,
                               F.mbf, F.mvuRight.data(), F.mvDepth.data(), &oob),
          "msorb_stereo_matches");
}

// Frame::Frame(imLeft, imRight, ...), Frame.cc:119-137: `thread threadLeft(&Frame::ExtractORB, this, 0, imLeft, 0, 0);
// thread threadRight(&Frame::ExtractORB, this, 1, imRight, 0, 0); join; ... ComputeStereoMatches();` becomes one call that
// fills mvKeys / mDescriptors, mvKeysRight / mDescriptorsRight, mvuRight and mvDepth (the caller keeps N = mvKeys.size(),
// UndistortKeyPoints() etc. in between: they do not feed the stereo association for rectified input).  Identical results.
template <class FrameT, class ExtractorT, class MatT>
void ExtractStereo(FrameT& F, const ExtractorT& left, const MatT& imLeft, const MatT& imRight) {
    const int cap = msorb_extractor_capacity(left.handle());
    F.mvKeys.resize(cap);
    F.mvKeysRight.resize(cap);
    static_assert(sizeof(F.mvKeys[0]) == sizeof(msorb_keypoint), "cv::KeyPoint must be the 28-byte layout");
    std::vector<uint8_t> dl((size_t)cap * 32), dr((size_t)cap * 32);
    std::vector<float> ur(cap), depth(cap);
    int nl = 0, nr = 0, oob = 0;
    check(msorb_extract_stereo(left.handle(), imLeft.data, imRight.data, imLeft.rows, imLeft.cols, (size_t)imLeft.step,
                               (size_t)imRight.step, F.mb, F.mbf, reinterpret_cast<msorb_keypoint*>(F.mvKeys.data()), dl.data(),
                               &nl, reinterpret_cast<msorb_keypoint*>(F.mvKeysRight.data()), dr.data(), &nr, cap, ur.data(),
                               depth.data(), &oob),
          "msorb_extract_stereo");
    F.mvKeys.resize(nl);
    F.mvKeysRight.resize(nr);
    F.mDescriptors.create(nl, 32, 0 /* CV_8U */);
    F.mDescriptorsRight.create(nr, 32, 0 /* CV_8U */);
    for (int i = 0; i < nl; i++) std::memcpy(F.mDescriptors.template ptr<unsigned char>(i), &dl[(size_t)i * 32], 32);
    for (int i = 0; i < nr; i++) std::memcpy(F.mDescriptorsRight.template ptr<unsigned char>(i), &dr[(size_t)i * 32], 32);
    F.mvuRight.assign(ur.begin(), ur.begin() + nl);
    F.mvDepth.assign(depth.begin(), depth.begin() + nl);
}

// ExtractStereo that also leaves the frame on the DEVICE, ready for the searches: the stereo constructor up to and including
// AssignFeaturesToGrid (Frame.cc:119-137 + :385-416) — `dev` is filled from the device-resident left keypoints / descriptors /
// mvuRight (grid by frame_grid_kernel) inside the same call, so the dev.Upload(F) that SearchByProjection / SearchLocalPoints
// would need is gone.  F.mnMinX .. F.mnMaxY must be set (ComputeImageBounds runs before the extraction for the first frame,
// Frame.cc:139-160); for rectified input mvKeysUn == mvKeys (Frame.cc:681-685), which is what the device copy holds.
template <class FrameT, class ExtractorT, class MatT>
void ExtractStereoFrame(DeviceFrame<FrameT>& dev, FrameT& F, const ExtractorT& left, const MatT& imLeft, const MatT& imRight) {
    const int cap = msorb_extractor_capacity(left.handle());
    F.mvKeys.resize(cap);
    F.mvKeysRight.resize(cap);
    static_assert(sizeof(F.mvKeys[0]) == sizeof(msorb_keypoint), "cv::KeyPoint must be the 28-byte layout");
    std::vector<uint8_t> dl((size_t)cap * 32), dr((size_t)cap * 32);
    std::vector<float> ur(cap), depth(cap);
    int nl = 0, nr = 0, oob = 0;
    check(msorb_extract_stereo_frame(left.handle(), dev.get(), imLeft.data, imRight.data, imLeft.rows, imLeft.cols, (size_t)imLeft.step,
                                     (size_t)imRight.step, F.mb, F.mbf, reinterpret_cast<msorb_keypoint*>(F.mvKeys.data()), dl.data(),
                                     &nl, reinterpret_cast<msorb_keypoint*>(F.mvKeysRight.data()), dr.data(), &nr, cap, ur.data(),
                                     depth.data(), &oob, F.mnMinX, F.mnMaxX, F.mnMinY, F.mnMaxY),
          "msorb_extract_stereo_frame");
    F.mvKeys.resize(nl);
    F.mvKeysRight.resize(nr);
    F.mDescriptors.create(nl, 32, 0 /* CV_8U */);
    F.mDescriptorsRight.create(nr, 32, 0 /* CV_8U */);
    for (int i = 0; i < nl; i++) std::memcpy(F.mDescriptors.template ptr<unsigned char>(i), &dl[(size_t)i * 32], 32);
    for (int i = 0; i < nr; i++) std::memcpy(F.mDescriptorsRight.template ptr<unsigned char>(i), &dr[(size_t)i * 32], 32);
    F.mvuRight.assign(ur.begin(), ur.begin() + nl);
    F.mvDepth.assign(depth.begin(), depth.begin() + nl);
}

// Frame::ComputeStereoFishEyeMatches (src/Frame.cc:1057-1101; the two-camera fisheye constructor): the brute-force
// BFmatcher.knnMatch(stereoDescLeft, stereoDescRight, matches, 2) over the lapping-area descriptors runs on the device
// (msorb_knn_match2: dense top-2 Hamming kernel, ties to the lower train index like cv::BFMatcher); Lowe's ratio test and the
// triangulation stay the reference's code.  `triangulate(kpLeft, kpRight, sigma1, sigma2, p3D) -> depth` is the caller's
//     static_cast<KannalaBrandt8*>(mpCamera)->TriangulateMatches(mpCamera2, kpLeft, kpRight, mRlr, mtlr, sigma1, sigma2, p3D)
// (:1089; the camera classes are not part of this path).  Fills mvLeftToRightMatch, mvRightToLeftMatch, mvDepth, mvuRight,
// mvStereo3Dpoints exactly like the reference's loop; returns nMatches.
template <class FrameT, class TriangulateFn>
int ComputeStereoFishEyeMatches(FrameT& F, TriangulateFn triangulate, int device = 0) {
    const int nl = (int)F.mvKeys.size() - F.monoLeft, nr = (int)F.mvKeysRight.size() - F.monoRight;   // :1059-1060
    std::vector<uint8_t> dl((size_t)std::max(nl, 0) * 32), dr((size_t)std::max(nr, 0) * 32);
    for (int i = 0; i < nl; i++) std::memcpy(&dl[(size_t)i * 32], F.mDescriptors.template ptr<unsigned char>(F.monoLeft + i), 32);
    for (int i = 0; i < nr; i++) std::memcpy(&dr[(size_t)i * 32], F.mDescriptorsRight.template ptr<unsigned char>(F.monoRight + i), 32);
    F.mvLeftToRightMatch.assign(F.Nleft, -1);                                 // :1065-1069
    F.mvRightToLeftMatch.assign(F.Nright, -1);
    F.mvDepth.assign(F.Nleft, -1.0f);
    F.mvuRight.assign(F.Nleft, -1.0f);
    F.mvStereo3Dpoints.resize(F.Nleft);
    F.mnCloseMPs = 0;
    std::vector<int> bi(std::max(nl, 1)), bd(std::max(nl, 1)), sd(std::max(nl, 1));
    check(msorb_knn_match2(device, dl.data(), std::max(nl, 0), dr.data(), std::max(nr, 0), bi.data(), bd.data(), nullptr, sd.data()),
          "msorb_knn_match2");
    int nMatches = 0;
    for (int q = 0; q < nl; q++) {
        if (nr < 2 || bi[q] < 0) continue;                                    // (*it).size() >= 2
        if (!((float)bd[q] < (float)sd[q] * 0.7)) continue;                   // :1082 (DMatch::distance is a float)
        const int iL = q + F.monoLeft, iR = bi[q] + F.monoRight;
        const float sigma1 = F.mvLevelSigma2[F.mvKeys[iL].octave], sigma2 = F.mvLevelSigma2[F.mvKeysRight[iR].octave];
        auto p3D = F.mvStereo3Dpoints[iL];
        const float depth = triangulate(F.mvKeys[iL], F.mvKeysRight[iR], sigma1, sigma2, p3D);
        if (depth > 0.0001f) {                                                // :1090-1096
            F.mvLeftToRightMatch[iL] = iR;
            F.mvRightToLeftMatch[iR] = iL;
            F.mvStereo3Dpoints[iL] = p3D;
            F.mvDepth[iL] = depth;
            nMatches++;
        }
    }
    return nMatches;
}

// The same with one extractor object per GPU (BASELINE configs[3]): `left` lives on device A, `right` on device B
// (MSORB_DEVICES="0,1": Tracking.cc:595-596 constructs mpORBextractorLeft, then mpORBextractorRight).  Replaces, in the
// stereo Frame constructor (Frame.cc:119-137),
//     thread threadLeft(&Frame::ExtractORB,this,0,imLeft,0,0);  thread threadRight(&Frame::ExtractORB,this,1,imRight,0,0);
//     threadLeft.join(); threadRight.join();  ...  ComputeStereoMatches();
// by ONE call: each eye's kernel chain runs on its own device, the right eye's features and pyramid cross xGMI to device A
// (msorb_extract_stereo_split), the association runs there.  Falls back to nothing: both objects on one device take the
// same path.  Identical results.
template <class FrameT, class ExtractorT, class MatT>
void ExtractStereoSplit(FrameT& F, const ExtractorT& left, const ExtractorT& right, const MatT& imLeft, const MatT& imRight) {
    const int cap = msorb_extractor_capacity(left.handle());
    F.mvKeys.resize(cap);
    F.mvKeysRight.resize(cap);
    static_assert(sizeof(F.mvKeys[0]) == sizeof(msorb_keypoint), "cv::KeyPoint must be the 28-byte layout");
    std::vector<uint8_t> dl((size_t)cap * 32), dr((size_t)cap * 32);
    std::vector<float> ur(cap), depth(cap);
    int nl = 0, nr = 0, oob = 0;
    check(msorb_extract_stereo_split(left.handle(), right.handle(), imLeft.data, imRight.data, imLeft.rows, imLeft.cols,
                                     (size_t)imLeft.step, (size_t)imRight.step, F.mb, F.mbf,
                                     reinterpret_cast<msorb_keypoint*>(F.mvKeys.data()), dl.data(), &nl,
                                     reinterpret_cast<msorb_keypoint*>(F.mvKeysRight.data()), dr.data(), &nr, cap, ur.data(),
                                     depth.data(), &oob),
          "msorb_extract_stereo_split");
    F.mvKeys.resize(nl);
    F.mvKeysRight.resize(nr);
    F.mDescriptors.create(nl, 32, 0 /* CV_8U */);
    F.mDescriptorsRight.create(nr, 32, 0 /* CV_8U */);
    for (int i = 0; i < nl; i++) std::memcpy(F.mDescriptors.template ptr<unsigned char>(i), &dl[(size_t)i * 32], 32);
    for (int i = 0; i < nr; i++) std::memcpy(F.mDescriptorsRight.template ptr<unsigned char>(i), &dr[(size_t)i * 32], 32);
    F.mvuRight.assign(ur.begin(), ur.begin() + nl);
    F.mvDepth.assign(depth.begin(), depth.begin() + nl);
}

}  // namespace msorb_host
}  // namespace ORB_SLAM3

#endif
