// ORB_SLAM3::ORBmatcher on libmsorb.so (HIP kernels for gfx950 behind the C ABI of include/msorb.h).
//
// Build MS-SLAM with this directory ahead of its own include/ and with ORBmatcher.cc of this directory in place of
// src/ORBmatcher.cc: Tracking.cc, LocalMapping.cc and LoopClosing.cc compile unchanged, because every member below has the
// name, argument types, defaults and return type of the class the reference declares in include/ORBmatcher.h:36-112 (the
// parameter names and the order of the declarations are this file's own; neither is part of the interface).
// Served: rectified stereo and monocular pinhole rigs (Frame::Nleft == -1, KeyFrame::NLeft == -1 — every shipped example).
// Not served: the fisheye two-camera branches (reference ORBmatcher.cc:144-210, 2059-2124; Fuse with bRight = true).
#ifndef ORBMATCHER_H
#define ORBMATCHER_H

#include <cstddef>
#include <memory>
#include <set>
#include <utility>
#include <vector>

#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#include "sophus/sim3.hpp"

#include "MapPoint.h"
#include "KeyFrame.h"
#include "Frame.h"

namespace ORB_SLAM3 {

class ORBmatcher {
    // shorthands for the declarations below (the signatures are the reference's: these are the same types)
    using KF = std::shared_ptr<KeyFrame>;
    using MP = std::shared_ptr<MapPoint>;
    using KFs = std::vector<KF>;
    using MPs = std::vector<MP>;

public:
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW

    static const int TH_HIGH;       // 100
    static const int TH_LOW;        // 50
    static const int HISTO_LENGTH;  // 30 rotation bins

    ORBmatcher(float nnratio = 0.6, bool checkOri = true);

    // 256-bit Hamming distance of two descriptor rows (reference ORBmatcher.cc:2323-2339)
    static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b);

    // ---- Tracking thread --------------------------------------------------------------------------------------------------
    // SearchLocalPoints: local map points already projected by Frame::isInFrustum (:43-142)
    int SearchByProjection(Frame& frame, const MPs& localPoints, const float th = 3, const bool farPoints = false,
                           const float farThreshold = 50.0f);
    // TrackWithMotionModel: the last frame's points projected with the predicted pose (:1941-2152)
    int SearchByProjection(Frame& current, const Frame& last, const float th, const bool mono);
    // Relocalization: a candidate KeyFrame's points projected into the frame (:2154-2275)
    int SearchByProjection(Frame& current, KF keyframe, const std::set<MP>& alreadyFound, const float th, const int orbDist);
    // TrackReferenceKeyFrame / Relocalization: matches inside common vocabulary nodes (:223-421)
    int SearchByBoW(KF keyframe, Frame& frame, MPs& matches);
    // monocular map initialisation (:755-870)
    int SearchForInitialization(Frame& first, Frame& second, std::vector<cv::Point2f>& prevMatched, std::vector<int>& matches12,
                                int windowSize = 10);

    // ---- LocalMapping thread ----------------------------------------------------------------------------------------------
    // CreateNewMapPoints: unmatched features of two KeyFrames under the epipolar constraint (:1168-1402)
    int SearchForTriangulation(KF kf1, KF kf2, std::vector<std::pair<size_t, size_t>>& matchedPairs, const bool onlyStereo,
                               const bool coarse = false);
    // SearchInNeighbors: duplicated map points of a neighbour (:1404-1597)
    int Fuse(KF keyframe, const MPs& points, const float th = 3.0, const bool right = false);

    // ---- LoopClosing thread (place recognition, loop closing, map merging) -----------------------------------------------
    int SearchByBoW(KF kf1, KF kf2, MPs& matches12);                                                   // :872-1016
    int SearchByBoW(KF kf1, KF kf2, KFs& matchedCurrentKF, MPs& matchedCurrentMP, KFs& matchedLoopKF, MPs& matchedLoopMP,
                    long unsigned int& currentId);                                                     // :1018-1166
    int SearchByProjection(KF keyframe, Sophus::Sim3<float>& Scw, const MPs& points, MPs& matched, int th,
                           float ratioHamming = 1.0);                                                  // :423-530
    int SearchByProjection(KF keyframe, Sophus::Sim3<float>& Scw, const MPs& points, const KFs& pointKFs, MPs& matched,
                           KFs& matchedKF, int th, float ratioHamming = 1.0);                          // :639-753
    int SearchByProjectionLoop(KF keyframe, Sophus::Sim3<float>& Scw, const MPs& points, MPs& matched, KFs& matchedKF, int th,
                               float ratioHamming = 1.0);                                              // :532-637
    int SearchBySim3(KF kf1, KF kf2, MPs& matches12, const Sophus::Sim3f& S12, const float th);        // :1718-1939
    int Fuse(KF keyframe, Sophus::Sim3f& Scw, const MPs& points, float th, MPs& replaced);             // :1599-1716

protected:
    // kept for source compatibility with code that derives from the class; the searches above do this work on the device
    float RadiusByViewingCos(const float& viewCos);
    void ComputeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3);

    float mfNNratio;
    bool mbCheckOrientation;
};

}  // namespace ORB_SLAM3

// Lifetime of the device-side state behind the class (ORBmatcher.cc).  The matcher keeps KeyFrames resident on the GPU from
// their first BoW-node search on.  NO edit of the reference's sources is needed to keep that bounded: MS-SLAM holds every KeyFrame
// in a std::shared_ptr (include/ORBmatcher.h:47-96), the store watches each resident KeyFrame through a std::weak_ptr and drops
// the entry once the object is gone (KeyFrame::SetBadFlag -> the map erases it -> the last shared_ptr goes: KeyFrame.cc:311-361,
// LocalMapping.cc KeyFrameCulling), and SetKeyFrameBudget() caps what may stay (least recently searched first).  The hooks
// below are accelerators and housekeeping, each one line where it is used ("Lifetime"):
//   KeyFrame::SetBadFlag (KeyFrame.cc)                      ORB_SLAM3::msorb_host::ForgetKeyFrame(mnId);   // frees the slot at once instead of at the next add
//   Tracking::Reset / ResetActiveMap (Tracking.cc:3848)     ORB_SLAM3::msorb_host::ResetKeyFrames();       // (entries of dead objects go by themselves; this frees them at once)
//   System::Shutdown (System.cc), after the threads joined  ORB_SLAM3::msorb_host::Shutdown();             // frees while the HIP runtime is alive
//   a worker thread that stops using the matcher            ORB_SLAM3::msorb_host::ReleaseThread();        // its four device frames
namespace ORB_SLAM3 {
namespace msorb_host {
void ForgetKeyFrame(unsigned long mnId);
void ResetKeyFrames();
size_t ResidentKeyFrames();                                          // live entries (dead KeyFrames are swept before counting)
size_t ResidentKeyFrameBytes();                                      // estimate of the device memory behind them
void SetKeyFrameBudget(size_t max_keyframes, size_t max_bytes);     // 0 = unlimited (default): beyond it the least recently searched leave
void KeyFrameStoreStats(unsigned long long* uploads, unsigned long long* expired, unsigned long long* evicted);   // since the process started; NULL = skip
void ReleaseThread();
void Shutdown();
}  // namespace msorb_host
}  // namespace ORB_SLAM3

#endif  // ORBMATCHER_H
