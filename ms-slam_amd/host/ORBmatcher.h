// Drop-in replacement for MS-SLAM's include/ORBmatcher.h (/root/reference/include/ORBmatcher.h:36-112): the same class
// name, namespace, constructor, the 13 public search methods with their shared_ptr<KeyFrame / MapPoint> signatures,
// DescriptorDistance, TH_LOW / TH_HIGH / HISTO_LENGTH — implemented on libmsorb.so (HIP kernels for gfx950) through the C
// ABI of include/msorb.h.  Build MS-SLAM with this directory ahead of its own include/ and src/ORBmatcher.cc replaced by
// ORBmatcher.cc of this directory; Tracking.cc / LocalMapping.cc / LoopClosing.cc stay unchanged.
// Rectified-stereo configurations (Frame::Nleft == -1, KeyFrame::NLeft == -1: every shipped example) are served; the
// fisheye two-camera branches of the reference (ORBmatcher.cc:144-210, 2059-2124, bRight = true) are not.
#ifndef ORBMATCHER_H
#define ORBMATCHER_H

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
public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true);

    // Computes the Hamming distance between two ORB descriptors (ORBmatcher.cc:2323-2339)
    static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b);

    // Search matches between Frame keypoints and projected MapPoints. Returns number of matches (Tracking::SearchLocalPoints)
    int SearchByProjection(Frame& F, const std::vector<std::shared_ptr<MapPoint>>& vpMapPoints, const float th = 3,
                           const bool bFarPoints = false, const float thFarPoints = 50.0f);

    // Project MapPoints tracked in last frame into the current frame and search matches (Tracking::TrackWithMotionModel)
    int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono);

    // Project MapPoints seen in KeyFrame into the Frame and search matches (Tracking::Relocalization)
    int SearchByProjection(Frame& CurrentFrame, std::shared_ptr<KeyFrame> pKF, const std::set<std::shared_ptr<MapPoint>>& sAlreadyFound,
                           const float th, const int ORBdist);

    // Project MapPoints using a Similarity Transformation and search matches (Loop Closing)
    int SearchByProjection(std::shared_ptr<KeyFrame> pKF, Sophus::Sim3<float>& Scw, const std::vector<std::shared_ptr<MapPoint>>& vpPoints,
                           std::vector<std::shared_ptr<MapPoint>>& vpMatched, int th, float ratioHamming = 1.0);

    int SearchByProjectionLoop(std::shared_ptr<KeyFrame> pKF, Sophus::Sim3<float>& Scw, const std::vector<std::shared_ptr<MapPoint>>& vpPoints,
                               std::vector<std::shared_ptr<MapPoint>>& vpMatched, std::vector<std::shared_ptr<KeyFrame>>& vpMatchedKF, int th,
                               float ratioHamming = 1.0);

    // Project MapPoints using a Similarity Transformation and search matches (Place Recognition: Loop Closing and Merging)
    int SearchByProjection(std::shared_ptr<KeyFrame> pKF, Sophus::Sim3<float>& Scw, const std::vector<std::shared_ptr<MapPoint>>& vpPoints,
                           const std::vector<std::shared_ptr<KeyFrame>>& vpPointsKFs, std::vector<std::shared_ptr<MapPoint>>& vpMatched,
                           std::vector<std::shared_ptr<KeyFrame>>& vpMatchedKF, int th, float ratioHamming = 1.0);

    // Search matches between MapPoints in a KeyFrame and ORB in a Frame, constrained to the same vocabulary node
    int SearchByBoW(std::shared_ptr<KeyFrame> pKF, Frame& F, std::vector<std::shared_ptr<MapPoint>>& vpMapPointMatches);
    int SearchByBoW(std::shared_ptr<KeyFrame> pKF1, std::shared_ptr<KeyFrame> pKF2, std::vector<std::shared_ptr<MapPoint>>& vpMatches12);
    int SearchByBoW(std::shared_ptr<KeyFrame> pKF1, std::shared_ptr<KeyFrame> pKF2,
                    std::vector<std::shared_ptr<KeyFrame>>& vpMatchedCurrentKeyFrame,
                    std::vector<std::shared_ptr<MapPoint>>& vpMatchedCurrentMapPoint,
                    std::vector<std::shared_ptr<KeyFrame>>& vpMatchedLoopKeyFrame,
                    std::vector<std::shared_ptr<MapPoint>>& vpMatchedLoopMapPoint, long unsigned int& nCurrentId);

    // Matching for the Map Initialization (only used in the monocular case)
    int SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12,
                                int windowSize = 10);

    // Matching to triangulate new MapPoints. Check Epipolar Constraint.
    int SearchForTriangulation(std::shared_ptr<KeyFrame> pKF1, std::shared_ptr<KeyFrame> pKF2,
                               std::vector<std::pair<size_t, size_t>>& vMatchedPairs, const bool bOnlyStereo, const bool bCoarse = false);

    // Search matches between MapPoints seen in KF1 and KF2 transforming by a Sim3 [s12*R12|t12]
    int SearchBySim3(std::shared_ptr<KeyFrame> pKF1, std::shared_ptr<KeyFrame> pKF2, std::vector<std::shared_ptr<MapPoint>>& vpMatches12,
                     const Sophus::Sim3f& S12, const float th);

    // Project MapPoints into KeyFrame and search for duplicated MapPoints.
    int Fuse(std::shared_ptr<KeyFrame> pKF, const std::vector<std::shared_ptr<MapPoint>>& vpMapPoints, const float th = 3.0,
             const bool bRight = false);

    // Project MapPoints into KeyFrame using a given Sim3 and search for duplicated MapPoints.
    int Fuse(std::shared_ptr<KeyFrame> pKF, Sophus::Sim3f& Scw, const std::vector<std::shared_ptr<MapPoint>>& vpPoints, float th,
             std::vector<std::shared_ptr<MapPoint>>& vpReplacePoint);

public:
    static const int TH_LOW;
    static const int TH_HIGH;
    static const int HISTO_LENGTH;
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW

protected:
    float RadiusByViewingCos(const float& viewCos);
    void ComputeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3);

    float mfNNratio;
    bool mbCheckOrientation;
};

}  // namespace ORB_SLAM3

#endif  // ORBMATCHER_H
