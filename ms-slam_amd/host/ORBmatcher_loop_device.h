// Device back-end for the loop-closing / initialisation call sites of ORBmatcher, in the style of ORBmatcher_device.h:
// templates over the reference's own KeyFrame / MapPoint / Frame / Sophus types (they compile inside MS-SLAM with the
// real classes and in tests/ against stand-ins with the same member names).  The geometry of every function is the
// reference's own code (same expressions, compiled with the application's flags); the window searches run behind the C ABI.
//
//   SearchByProjection(dev, pKF, Scw, vpPoints, vpMatched, th, ratioHamming)                       ORBmatcher.cc:423-530
//   SearchByProjectionLoop(dev, pKF, Scw, vpPoints, vpMatched, vpMatchedKF, th, ratioHamming)       :532-637
//   SearchByProjection(dev, pKF, Scw, vpPoints, vpPointsKFs, vpMatched, vpMatchedKF, th, ratio)     :639-753
//   SearchBySim3(dev1, dev2, pKF1, pKF2, vpMatches12, S12, th)                                      :1718-1939
//   Fuse(dev, pKF, Scw, vpPoints, th, vpReplacePoint)                                               :1599-1716
//   SearchForInitialization(dev1, dev2, F1, F2, vbPrevMatched, vnMatches12, windowSize, ...)        :755-870
//   SearchByBoWLoop(pKF1, pKF2, vpMatchedCurrentKeyFrame, ..., nCurrentId, ...)                     :1018-1166
// `dev*` = msorb_host::DeviceFrame holding the KeyFrame / Frame the search runs in (UploadKeyFrame / Upload).
#ifndef MSORB_ORBMATCHER_LOOP_DEVICE_H
#define MSORB_ORBMATCHER_LOOP_DEVICE_H

#include <climits>
#include <set>
#include <tuple>

#include "ORBmatcher_device.h"

namespace ORB_SLAM3 {
namespace msorb_host {

struct Sim3Queries {  // per candidate point: what the projection part of a Sim3 search computes
    std::vector<uint8_t> valid, desc;
    std::vector<float> u, v;
    std::vector<int> level;
    void reset(int n) {
        valid.assign(n, 0); desc.assign((size_t)n * 32, 0);
        u.assign(n, 0); v.assign(n, 0); level.assign(n, 0);
    }
    template <class MapPointPtr>
    void accept(int i, float uu, float vv, int lvl, const MapPointPtr& pMP) {
        valid[i] = 1; u[i] = uu; v[i] = vv; level[i] = lvl;
        const auto d = pMP->GetDescriptor();
        std::memcpy(&desc[(size_t)i * 32], d.template ptr<unsigned char>(0), 32);
    }
};

// The geometric part shared by the three SearchByProjection(pKF, Scw, ...) forms and Fuse(pKF, Scw, ...): :433-480, :541-588,
// :648-695, :1608-1656.  `skip(iMP, pMP)` = the form's "discard" test; HAND_PROJECTION = the (pKF, Scw, vpPoints,
// vpPointsKFs, ...) form projects with fx * (X * invz) + cx (:671-676) instead of mpCamera->project (fx * X / Z + cx).
template <bool HAND_PROJECTION, class KeyFramePtr, class Sim3T, class MapPointPtr, class Skip>
void ProjectSim3(const KeyFramePtr& pKF, Sim3T& Scw, const std::vector<MapPointPtr>& vpPoints, Skip skip, Sim3Queries& Q) {
    const float& fx = pKF->fx;
    const float& fy = pKF->fy;
    const float& cx = pKF->cx;
    const float& cy = pKF->cy;
    using SE3 = typename std::decay<decltype(pKF->GetPose())>::type;
    SE3 Tcw = SE3(Scw.rotationMatrix(), Scw.translation() / Scw.scale());
    const auto Ow = Tcw.inverse().translation();
    const int nPoints = (int)vpPoints.size();
    Q.reset(nPoints);
    for (int iMP = 0; iMP < nPoints; iMP++) {
        const MapPointPtr& pMP = vpPoints[iMP];
        if (skip(iMP, pMP)) continue;
        const auto p3Dw = pMP->GetWorldPos();
        const auto p3Dc = Tcw * p3Dw;
        if (p3Dc(2) < 0.0) continue;                                      // depth must be positive
        float u, v;
        if (HAND_PROJECTION) {
            const float invz = 1 / p3Dc(2);
            const float x = p3Dc(0) * invz;
            const float y = p3Dc(1) * invz;
            u = fx * x + cx;
            v = fy * y + cy;
        } else {
            const auto uv = pKF->mpCamera->project(p3Dc);
            u = uv(0); v = uv(1);
        }
        if (!pKF->IsInImage(u, v)) continue;
        const float maxDistance = pMP->GetMaxDistanceInvariance();
        const float minDistance = pMP->GetMinDistanceInvariance();
        const auto PO = (p3Dw - Ow).eval();
        const float dist = PO.norm();
        if (dist < minDistance || dist > maxDistance) continue;
        const auto Pn = pMP->GetNormal();
        if (PO.dot(Pn) < 0.5 * dist) continue;                            // viewing angle below 60 deg
        const int nPredictedLevel = pMP->PredictScale(dist, pKF);
        Q.accept(iMP, u, v, nPredictedLevel, pMP);
    }
}

// shared tail of the two claiming forms: search + claims behind msorb_search_by_projection_sim3, then the pointer updates
template <class FrameT, class MapPointPtr, class OnMatch>
int RunClaimingSim3Search(DeviceFrame<FrameT>& dev, const Sim3Queries& Q, const std::vector<MapPointPtr>& vpPoints,
                          std::vector<MapPointPtr>& vpMatched, int th, float ratioHamming, OnMatch on_match) {
    const int n = (int)vpPoints.size(), N = (int)vpMatched.size();
    std::vector<int> ids(n), matched(N, -1);
    for (int i = 0; i < n; i++) ids[i] = i;
    for (int j = 0; j < N; j++)
        if (vpMatched[j]) matched[j] = n + j;                              // "if(vpMatched[idx]) continue"
    const std::vector<int> before(matched);
    int nmatches = 0;
    check(msorb_search_by_projection_sim3(dev.get(), n, Q.valid.data(), Q.u.data(), Q.v.data(), Q.level.data(), Q.desc.data(),
                                          ids.data(), matched.data(), (float)th, 50 /* TH_LOW */ * ratioHamming, &nmatches),
          "msorb_search_by_projection_sim3");
    for (int j = 0; j < N; j++)
        if (matched[j] != before[j] && matched[j] >= 0 && matched[j] < n) { vpMatched[j] = vpPoints[matched[j]]; on_match(j, matched[j]); }
    return nmatches;
}

// ORBmatcher::SearchByProjection(pKF, Scw, vpPoints, vpMatched, th, ratioHamming), :423-530
template <class FrameT, class KeyFramePtr, class Sim3T, class MapPointPtr>
int SearchByProjection(DeviceFrame<FrameT>& dev, const KeyFramePtr& pKF, Sim3T& Scw, const std::vector<MapPointPtr>& vpPoints,
                       std::vector<MapPointPtr>& vpMatched, int th, float ratioHamming) {
    std::set<MapPointPtr> spAlreadyFound(vpMatched.begin(), vpMatched.end());
    spAlreadyFound.erase(MapPointPtr());
    Sim3Queries Q;
    ProjectSim3<false>(pKF, Scw, vpPoints,
                       [&](int, const MapPointPtr& pMP) { return !pMP || pMP->isBad() || spAlreadyFound.count(pMP); }, Q);  // :448
    return RunClaimingSim3Search(dev, Q, vpPoints, vpMatched, th, ratioHamming, [](int, int) {});
}

// ORBmatcher::SearchByProjection(pKF, Scw, vpPoints, vpPointsKFs, vpMatched, vpMatchedKF, th, ratioHamming), :639-753
template <class FrameT, class KeyFramePtr, class Sim3T, class MapPointPtr>
int SearchByProjection(DeviceFrame<FrameT>& dev, const KeyFramePtr& pKF, Sim3T& Scw, const std::vector<MapPointPtr>& vpPoints,
                       const std::vector<KeyFramePtr>& vpPointsKFs, std::vector<MapPointPtr>& vpMatched,
                       std::vector<KeyFramePtr>& vpMatchedKF, int th, float ratioHamming) {
    std::set<MapPointPtr> spAlreadyFound(vpMatched.begin(), vpMatched.end());
    spAlreadyFound.erase(MapPointPtr());
    Sim3Queries Q;
    ProjectSim3<true>(pKF, Scw, vpPoints,
                      [&](int, const MapPointPtr& pMP) { return pMP->isBad() || spAlreadyFound.count(pMP); }, Q);            // :664
    return RunClaimingSim3Search(dev, Q, vpPoints, vpMatched, th, ratioHamming,
                                 [&](int idx, int iMP) { vpMatchedKF[idx] = vpPointsKFs[iMP]; });                            // :742-743
}

// ORBmatcher::SearchByProjectionLoop(pKF, Scw, vpPoints, vpMatched, vpMatchedKF, th, ratioHamming), :532-637
template <class FrameT, class KeyFramePtr, class Sim3T, class MapPointPtr>
int SearchByProjectionLoop(DeviceFrame<FrameT>& dev, const KeyFramePtr& pKF, Sim3T& Scw, const std::vector<MapPointPtr>& vpPoints,
                           std::vector<MapPointPtr>& vpMatched, std::vector<KeyFramePtr>& vpMatchedKF, int th, float ratioHamming) {
    const std::vector<MapPointPtr> vpMapPointsToMatch = pKF->GetMapPointMatches();
    Sim3Queries Q;
    ProjectSim3<false>(pKF, Scw, vpPoints, [&](int iMP, const MapPointPtr& pMP) { return pMP->isBad() || vpMatched[iMP]; }, Q);  // :555
    const int n = (int)vpPoints.size(), N = (int)vpMapPointsToMatch.size();
    std::vector<uint8_t> trainOk(N);
    for (int j = 0; j < N; j++) trainOk[j] = vpMapPointsToMatch[j] && !vpMapPointsToMatch[j]->isBad();                        // :609-610
    std::vector<int> bestIdx(n, -1);
    int nmatches = 0;
    check(msorb_search_by_projection_loop(dev.get(), n, Q.valid.data(), Q.u.data(), Q.v.data(), Q.level.data(), Q.desc.data(),
                                          trainOk.data(), (float)th, 50 /* TH_LOW */ * ratioHamming, bestIdx.data(), &nmatches),
          "msorb_search_by_projection_loop");
    for (int iMP = 0; iMP < n; iMP++)
        if (bestIdx[iMP] >= 0) {                                           // :626-631
            vpMatched[iMP] = vpMapPointsToMatch[bestIdx[iMP]];
            vpMatchedKF[iMP] = pKF;
        }
    return nmatches;
}

// The projection part of ORBmatcher::SearchBySim3 (:1720-1794, :1850-1885): Q1 = pKF1's map points seen from pKF2, Q2 = pKF2's
// seen from pKF1 (both with pKF1's intrinsics, like the reference).
template <class KeyFramePtr, class Sim3T, class MapPointPtr>
void Sim3PairGeometry(const KeyFramePtr& pKF1, const KeyFramePtr& pKF2, const std::vector<MapPointPtr>& vpMatches12, const Sim3T& S12,
                      Sim3Queries& Q1, Sim3Queries& Q2) {
    const float& fx = pKF1->fx;
    const float& fy = pKF1->fy;
    const float& cx = pKF1->cx;
    const float& cy = pKF1->cy;
    const auto T1w = pKF1->GetPose();
    const auto T2w = pKF2->GetPose();
    const auto S21 = S12.inverse();
    const std::vector<MapPointPtr> vpMapPoints1 = pKF1->GetMapPointMatches();
    const int N1 = (int)vpMapPoints1.size();
    const std::vector<MapPointPtr> vpMapPoints2 = pKF2->GetMapPointMatches();
    const int N2 = (int)vpMapPoints2.size();
    std::vector<bool> vbAlreadyMatched1(N1, false), vbAlreadyMatched2(N2, false);
    for (int i = 0; i < N1; i++) {
        const MapPointPtr pMP = vpMatches12[i];
        if (pMP) {
            vbAlreadyMatched1[i] = true;
            const int idx2 = std::get<0>(pMP->GetIndexInKeyFrame(pKF2));
            if (idx2 >= 0 && idx2 < N2) vbAlreadyMatched2[idx2] = true;
        }
    }
    Q1.reset(N1); Q2.reset(N2);
    for (int i1 = 0; i1 < N1; i1++) {                                      // :1758-1794
        const MapPointPtr& pMP = vpMapPoints1[i1];
        if (!pMP || vbAlreadyMatched1[i1]) continue;
        if (pMP->isBad()) continue;
        const auto p3Dw = pMP->GetWorldPos();
        const auto p3Dc1 = T1w * p3Dw;
        const auto p3Dc2 = S21 * p3Dc1;
        if (p3Dc2(2) < 0.0) continue;
        const float invz = 1.0 / p3Dc2(2);
        const float x = p3Dc2(0) * invz;
        const float y = p3Dc2(1) * invz;
        const float u = fx * x + cx;
        const float v = fy * y + cy;
        if (!pKF2->IsInImage(u, v)) continue;
        const float maxDistance = pMP->GetMaxDistanceInvariance();
        const float minDistance = pMP->GetMinDistanceInvariance();
        const float dist3D = p3Dc2.norm();
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        Q1.accept(i1, u, v, pMP->PredictScale(dist3D, pKF2), pMP);
    }
    for (int i2 = 0; i2 < N2; i2++) {                                      // :1850-1885
        const MapPointPtr& pMP = vpMapPoints2[i2];
        if (!pMP || vbAlreadyMatched2[i2]) continue;
        if (pMP->isBad()) continue;
        const auto p3Dw = pMP->GetWorldPos();
        const auto p3Dc2 = T2w * p3Dw;
        const auto p3Dc1 = S12 * p3Dc2;
        if (p3Dc1(2) < 0.0) continue;
        const float invz = 1.0 / p3Dc1(2);
        const float x = p3Dc1(0) * invz;
        const float y = p3Dc1(1) * invz;
        const float u = fx * x + cx;
        const float v = fy * y + cy;
        if (!pKF1->IsInImage(u, v)) continue;
        const float maxDistance = pMP->GetMaxDistanceInvariance();
        const float minDistance = pMP->GetMinDistanceInvariance();
        const float dist3D = p3Dc1.norm();
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        Q2.accept(i2, u, v, pMP->PredictScale(dist3D, pKF1), pMP);
    }
}

// ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, S12, th), :1718-1939.  dev1 / dev2 hold pKF1 / pKF2.
template <class FrameT, class KeyFramePtr, class Sim3T, class MapPointPtr>
int SearchBySim3(DeviceFrame<FrameT>& dev1, DeviceFrame<FrameT>& dev2, const KeyFramePtr& pKF1, const KeyFramePtr& pKF2,
                 std::vector<MapPointPtr>& vpMatches12, const Sim3T& S12, const float th) {
    Sim3Queries Q1, Q2;
    Sim3PairGeometry(pKF1, pKF2, vpMatches12, S12, Q1, Q2);
    const std::vector<MapPointPtr> vpMapPoints2 = pKF2->GetMapPointMatches();
    const int N1 = (int)Q1.valid.size(), N2 = (int)Q2.valid.size();
    std::vector<int> match12(N1, -1);
    int nFound = 0;
    check(msorb_search_by_sim3(dev1.get(), dev2.get(), N1, Q1.valid.data(), Q1.u.data(), Q1.v.data(), Q1.level.data(), Q1.desc.data(),
                               N2, Q2.valid.data(), Q2.u.data(), Q2.v.data(), Q2.level.data(), Q2.desc.data(), th, match12.data(),
                               &nFound),
          "msorb_search_by_sim3");
    for (int i1 = 0; i1 < N1; i1++)
        if (match12[i1] >= 0) vpMatches12[i1] = vpMapPoints2[match12[i1]];   // :1931-1934
    return nFound;
}

// ORBmatcher::Fuse(pKF, Scw, vpPoints, th, vpReplacePoint), :1599-1716.  The searches of all points run at once (they read
// only the KeyFrame's features); the loop that follows is the reference's, with the search replaced by a lookup — it reads
// pKF->GetMapPoint(bestIdx), which earlier iterations change through AddMapPoint.
template <class FrameT, class KeyFramePtr, class Sim3T, class MapPointPtr>
int Fuse(DeviceFrame<FrameT>& dev, const KeyFramePtr& pKF, Sim3T& Scw, const std::vector<MapPointPtr>& vpPoints, float th,
         std::vector<MapPointPtr>& vpReplacePoint) {
    const std::set<MapPointPtr> spAlreadyFound = pKF->GetMapPoints();
    Sim3Queries Q;
    ProjectSim3<false>(pKF, Scw, vpPoints, [&](int, const MapPointPtr& pMP) { return pMP->isBad() || spAlreadyFound.count(pMP); }, Q);  // :1624
    const int nPoints = (int)vpPoints.size();
    std::vector<int> bestIdx(nPoints, -1), bestDist(nPoints, INT_MAX);
    check(msorb_fuse_sim3_search(dev.get(), nPoints, Q.valid.data(), Q.u.data(), Q.v.data(), Q.level.data(), Q.desc.data(), th,
                                 bestIdx.data(), bestDist.data()),
          "msorb_fuse_sim3_search");
    int nFused = 0;
    for (int iMP = 0; iMP < nPoints; iMP++) {
        if (!Q.valid[iMP]) continue;
        if (bestDist[iMP] <= 50 /* TH_LOW */) {                            // :1698-1713
            MapPointPtr pMP = vpPoints[iMP];
            MapPointPtr pMPinKF = pKF->GetMapPoint(bestIdx[iMP]);
            if (pMPinKF) {
                if (!pMPinKF->isBad()) vpReplacePoint[iMP] = pMPinKF;
            } else {
                pMP->AddObservation(pKF, bestIdx[iMP]);
                pKF->AddMapPoint(pMP, bestIdx[iMP]);
            }
            nFused++;
        }
    }
    return nFused;
}

// ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize), :755-870.  dev1 / dev2 hold F1 / F2.
template <class FrameT, class Point2fT>
int SearchForInitialization(DeviceFrame<FrameT>& dev1, DeviceFrame<FrameT>& dev2, FrameT& F1, FrameT& F2,
                            std::vector<Point2fT>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize, float mfNNratio,
                            bool mbCheckOrientation) {
    const int N1 = (int)F1.mvKeysUn.size();
    vnMatches12 = std::vector<int>(N1, -1);
    std::vector<float> prev((size_t)2 * N1);
    for (int i = 0; i < N1; i++) { prev[2 * i] = vbPrevMatched[i].x; prev[2 * i + 1] = vbPrevMatched[i].y; }
    int nmatches = 0;
    check(msorb_search_for_initialization(dev1.get(), dev2.get(), prev.data(), windowSize, mfNNratio, mbCheckOrientation ? 1 : 0,
                                          vnMatches12.data(), &nmatches),
          "msorb_search_for_initialization");
    for (int i1 = 0; i1 < N1; i1++)                                        // :864-867
        if (vnMatches12[i1] >= 0) vbPrevMatched[i1] = F2.mvKeysUn[vnMatches12[i1]].pt;
    return nmatches;
}

// ORBmatcher::SearchByBoW(pKF1, pKF2, vpMatchedCurrentKeyFrame, vpMatchedCurrentMapPoint, vpMatchedLoopKeyFrame,
// vpMatchedLoopMapPoint, nCurrentId), :1018-1166: the KeyFrame-KeyFrame search with map points already used for this loop
// candidate (mnLoopPointForKF == nCurrentId) excluded on both sides; the rotation histogram is always applied and the
// survivors are appended in histogram-bin order (:1135-1161).
template <class KeyFramePtr, class MapPointPtr>
int SearchByBoWLoop(const KeyFramePtr& pKF1, const KeyFramePtr& pKF2, std::vector<KeyFramePtr>& vpMatchedCurrentKeyFrame,
                    std::vector<MapPointPtr>& vpMatchedCurrentMapPoint, std::vector<KeyFramePtr>& vpMatchedLoopKeyFrame,
                    std::vector<MapPointPtr>& vpMatchedLoopMapPoint, long unsigned int& nCurrentId, float mfNNratio, int device = 0) {
    const auto vKeysUn1 = pKF1->GetAllKeyUn();
    const auto vKeysUn2 = pKF2->GetAllKeyUn();
    const auto mps1 = pKF1->GetMapPointMatches();
    const auto mps2 = pKF2->GetMapPointMatches();
    BowSide a, b;
    a.FillKeyFrame(pKF1);
    b.FillKeyFrame(pKF2);
    a.flag.assign(mps1.size(), 0);
    b.flag.assign(mps2.size(), 0);
    for (size_t i = 0; i < mps1.size(); i++) a.flag[i] = mps1[i] && !mps1[i]->isBad() && mps1[i]->mnLoopPointForKF != nCurrentId;   // :1059-1066
    for (size_t i = 0; i < mps2.size(); i++) b.flag[i] = mps2[i] && !mps2[i]->isBad() && mps2[i]->mnLoopPointForKF != nCurrentId;   // :1086-1096
    msorb_bow_pair P;
    std::vector<int> m12, m21;
    BindBowPair(P, a, b, false, m12, m21);
    check(msorb_search_by_bow(device, &P, 1, 50 /* TH_LOW */, 0 /* '<', :1118 */, mfNNratio, 1 /* histogram always */, nullptr),
          "msorb_search_by_bow");
    // the survivors, pushed bin by bin in the order the merge walk found them (node ascending, list order inside a node)
    const int L = 30 /* HISTO_LENGTH */;
    const float factor = 1.0f / L;
    std::vector<std::vector<int>> rotHist(L);
    const auto fv1 = pKF1->GetFeatureVector();
    const auto fv2 = pKF2->GetFeatureVector();
    for (const auto& e : fv1) {
        if (!fv2.count(e.first)) continue;
        for (unsigned idx1 : e.second) {
            if ((size_t)idx1 >= m12.size() || m12[idx1] < 0) continue;
            float rot = vKeysUn1[idx1].angle - vKeysUn2[m12[idx1]].angle;
            if (rot < 0.0) rot += 360.0f;
            int bin = (int)std::round(rot * factor);
            if (bin == L) bin = 0;
            rotHist[bin].push_back((int)idx1);
        }
    }
    int nmatches = 0;
    for (int i = 0; i < L; i++)
        for (int idx1 : rotHist[i]) {
            const int idx2 = m12[idx1];
            mps1[idx1]->mnLoopPointForKF = nCurrentId;
            mps2[idx2]->mnLoopPointForKF = nCurrentId;
            vpMatchedCurrentKeyFrame.push_back(pKF1);
            vpMatchedCurrentMapPoint.push_back(mps1[idx1]);
            vpMatchedLoopKeyFrame.push_back(pKF2);
            vpMatchedLoopMapPoint.push_back(mps2[idx2]);
            nmatches++;
        }
    return nmatches;
}

}  // namespace msorb_host
}  // namespace ORB_SLAM3

#endif
