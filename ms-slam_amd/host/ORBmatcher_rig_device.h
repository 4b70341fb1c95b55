// Two-camera frames (Frame::Nleft != -1: the KannalaBrandt8 stereo rig of ORB-SLAM3) behind the drop-in ORBmatcher:
//   ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint>&, th, bFarPoints, thFarPoints)   ORBmatcher.cc:43-213 (right arm :144-210)
//   ORBmatcher::SearchByProjection(Frame& Current, const Frame& Last, th, bMono)                    ORBmatcher.cc:1941-2152 (right arm :2059-2124)
// The two cameras of such a frame are two device frames: the left one holds F.mvKeys[0, Nleft) with descriptor rows [0, Nleft), the
// right one F.mvKeysRight with rows [Nleft, N) — what Frame::GetFeaturesInArea(x, y, r, minLevel, maxLevel, bRight) walks
// (Frame.cc:589-655: the raw keypoints, not mvKeysUn, for such a frame) — both without mvuRight (the rectified-stereo test of
// :92 / :2015 is `Nleft == -1` only).  F.mvpMapPoints stays ONE array of N = Nleft + Nright entries.
//   ORBmatcher::SearchByBoW(pKF, F, vpMapPointMatches)                                              ORBmatcher.cc:223-421 (arms :276-309, :357-382)
//   ORBmatcher::Fuse(pKF, vpMapPoints, th, bRight) on a KeyFrame with NLeft != -1, both cameras         ORBmatcher.cc:1404-1597
//   ORBmatcher::SearchForTriangulation(pKF1, pKF2, ...) between two such KeyFrames                   ORBmatcher.cc:1168-1402 (arms :1195-1201, :1294-1330)
// No shipped configuration of MS-SLAM builds such frames (every YAML is Rectified / PinHole); the arms exist so that the class
// answers what the reference's class answers.
#ifndef MSORB_ORBMATCHER_RIG_DEVICE_H
#define MSORB_ORBMATCHER_RIG_DEVICE_H

#include <cmath>
#include <cstring>
#include <type_traits>
#include <utility>
#include <vector>

#include "ORBmatcher_device.h"

namespace ORB_SLAM3 {
namespace msorb_host {

// one camera of a two-camera frame -> device frame + 64 x 48 grid (Frame.cc:385-416 with Nleft != -1: mGrid / mGridRight)
template <class FrameT>
void UploadCamera(DeviceFrame<FrameT>& dev, const FrameT& F, bool right) {
    const std::vector<cv::KeyPoint>& keys = right ? F.mvKeysRight : F.mvKeys;
    static_assert(sizeof(keys[0]) == sizeof(msorb_keypoint), "cv::KeyPoint must be the 28-byte layout");
    const int n = right ? (int)F.mvKeysRight.size() : F.Nleft, row0 = right ? F.Nleft : 0;
    std::vector<uint8_t> desc((size_t)n * 32);
    for (int i = 0; i < n; i++) std::memcpy(&desc[(size_t)i * 32], F.mDescriptors.template ptr<unsigned char>(row0 + i), 32);
    check(msorb_frame_set(dev.get(), reinterpret_cast<const msorb_keypoint*>(keys.data()), n, desc.data(), nullptr, F.mnMinX, F.mnMaxX, F.mnMinY,
                          F.mnMaxY, F.mvScaleFactors.data(), (int)F.mvScaleFactors.size()),
          "msorb_frame_set");
}

// ORBmatcher::SearchByProjection(Frame &F, const vector<shared_ptr<MapPoint>> &vpMapPoints, th, bFarPoints, thFarPoints), F.Nleft != -1.
// devL / devR hold F's two cameras (UploadCamera).
template <class FrameT, class MapPointPtr>
int SearchByProjectionRig(DeviceFrame<FrameT>& devL, DeviceFrame<FrameT>& devR, FrameT& F, const std::vector<MapPointPtr>& vpMapPoints,
                          const float th, const bool bFarPoints, const float thFarPoints, const float mfNNratio) {
    const int M = (int)vpMapPoints.size(), N = (int)F.mvpMapPoints.size();
    // table = the local map points in call order, then the map points the frame already holds that are not among them (never
    // queries; their Observations() decides whether a keypoint is taken, :89-91 / :169-171)
    PointerIndex index;
    index.reset((size_t)M + N);
    for (int i = 0; i < M; i++) index.emplace(vpMapPoints[i].get(), i);
    std::vector<int> frameMp(N, -1), extraObs;
    for (int i = 0; i < N; i++) {
        if (!F.mvpMapPoints[i]) continue;
        const int at = index.emplace(F.mvpMapPoints[i].get(), M + (int)extraObs.size());
        frameMp[i] = at;
        if (at == M + (int)extraObs.size()) extraObs.push_back(F.mvpMapPoints[i]->Observations());
    }
    const int T = M + (int)extraObs.size();
    std::vector<uint8_t> inView(T, 0), inViewR(T, 0), bad(T, 0), spars(T, 0), desc((size_t)T * 32, 0);
    std::vector<float> px(T, 0.f), py(T, 0.f), pxr(T, 0.f), pyr(T, 0.f), depth(T, 0.f), vcos(T, 0.f), vcosR(T, 0.f);
    std::vector<int> level(T, 0), levelR(T, -1), obs(T, 0);
    for (int i = 0; i < M; i++) {
        const auto& p = vpMapPoints[i];
        inView[i] = p->mbTrackInView; inViewR[i] = p->mbTrackInViewR; bad[i] = p->isBad(); spars[i] = p->mbSparsified;
        px[i] = p->mTrackProjX; py[i] = p->mTrackProjY; pxr[i] = p->mTrackProjXR; pyr[i] = p->mTrackProjYR; depth[i] = p->mTrackDepth;
        level[i] = p->mnTrackScaleLevel; levelR[i] = p->mnTrackScaleLevelR; vcos[i] = p->mTrackViewCos; vcosR[i] = p->mTrackViewCosR;
        obs[i] = p->Observations();
        if ((inView[i] || inViewR[i]) && !bad[i]) {
            const auto d = p->GetDescriptor();
            std::memcpy(&desc[(size_t)i * 32], d.template ptr<unsigned char>(0), 32);
        }
    }
    for (int k = 0; k < (int)extraObs.size(); k++) obs[M + k] = extraObs[k];
    const std::vector<int> before = frameMp;
    int nmatches = 0;
    check(msorb_search_by_projection_mps_rig(devL.get(), devR.get(), T, inView.data(), inViewR.data(), bad.data(), spars.data(), px.data(), py.data(),
                                             pxr.data(), pyr.data(), depth.data(), level.data(), levelR.data(), vcos.data(), vcosR.data(), desc.data(),
                                             obs.data(), F.mvLeftToRightMatch.data(), F.mvRightToLeftMatch.data(), frameMp.data(), th,
                                             bFarPoints ? 1 : 0, thFarPoints, mfNNratio, &nmatches),
          "msorb_search_by_projection_mps_rig");
    for (int i = 0; i < N; i++)
        if (frameMp[i] != before[i] && frameMp[i] >= 0 && frameMp[i] < M) F.mvpMapPoints[i] = vpMapPoints[frameMp[i]];
    return nmatches;
}

// What :1962-1990 and :2060-2063 compute per last-frame keypoint for a two-camera CurrentFrame (the reference's own expressions,
// compiled with the application's flags)
struct LastFrameProjectionRig {
    std::vector<uint8_t> valid, desc;
    std::vector<float> u, v, ur, vr, angle;
    std::vector<int> octave, obs;
    bool forward = false, backward = false;
};
template <class FrameT>
void ProjectLastFrameRig(FrameT& CurrentFrame, const FrameT& LastFrame, bool bMono, LastFrameProjectionRig& P) {
    const auto Tcw = CurrentFrame.GetPose();
    const auto twc = Tcw.inverse().translation();
    const auto Tlw = LastFrame.GetPose();
    const auto tlc = Tlw * twc;
    P.forward = tlc(2) > CurrentFrame.mb && !bMono;                       // :1957
    P.backward = -tlc(2) > CurrentFrame.mb && !bMono;                     // :1958
    const auto Trl = CurrentFrame.GetRelativePoseTrl();
    const int n = LastFrame.N;
    P.valid.assign(n, 0); P.desc.assign((size_t)n * 32, 0);
    P.u.assign(n, 0); P.v.assign(n, 0); P.ur.assign(n, 0); P.vr.assign(n, 0); P.angle.assign(n, 0);
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
        const auto x3Dr = Trl * x3Dc;                                     // :2060
        const auto uvr = CurrentFrame.mpCamera->project(x3Dr);            // :2061 (mpCamera, as the reference has it)
        const bool lastLeft = LastFrame.Nleft == -1 || i < LastFrame.Nleft;
        const cv::KeyPoint& kpLast = lastLeft ? LastFrame.mvKeys[i] : LastFrame.mvKeysRight[i - LastFrame.Nleft];
        P.valid[i] = 1;
        P.u[i] = uv(0); P.v[i] = uv(1); P.ur[i] = uvr(0); P.vr[i] = uvr(1);
        P.octave[i] = kpLast.octave;                                      // :1986 / :2063
        P.angle[i] = LastFrame.Nleft == -1 ? LastFrame.mvKeysUn[i].angle : kpLast.angle;   // :2043-2045
        P.obs[i] = pMP->Observations();
        const auto d = pMP->GetDescriptor();
        std::memcpy(&P.desc[(size_t)i * 32], d.template ptr<unsigned char>(0), 32);
    }
}

// ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono), CurrentFrame.Nleft != -1: the projection
// above, then msorb_search_by_projection_frames_rig (the two arms' window searches and claims, the left-window-empty rule of
// :2003-2004, the one rotation histogram of :2129-2149).
template <class FrameT>
int SearchByProjectionRig(DeviceFrame<FrameT>& devL, DeviceFrame<FrameT>& devR, FrameT& CurrentFrame, const FrameT& LastFrame, const float th,
                          const bool bMono, const bool mbCheckOrientation) {
    LastFrameProjectionRig P;
    ProjectLastFrameRig(CurrentFrame, LastFrame, bMono, P);
    const int nL = LastFrame.N, N = CurrentFrame.N;
    std::vector<int> lastMp(nL), obs(P.obs.begin(), P.obs.end()), curMp(N, -1);
    for (int i = 0; i < nL; i++) lastMp[i] = i;
    for (int j = 0; j < N; j++)
        if (CurrentFrame.mvpMapPoints[j]) {
            curMp[j] = (int)obs.size();
            obs.push_back(CurrentFrame.mvpMapPoints[j]->Observations());
        }
    const std::vector<int> before = curMp;
    int nmatches = 0;
    check(msorb_search_by_projection_frames_rig(devL.get(), devR.get(), nL, P.valid.data(), P.u.data(), P.v.data(), P.ur.data(), P.vr.data(),
                                                P.octave.data(), P.angle.data(), P.desc.data(), lastMp.data(), obs.data(), (int)obs.size(),
                                                curMp.data(), th, P.forward, P.backward, mbCheckOrientation, &nmatches),
          "msorb_search_by_projection_frames_rig");
    for (int j = 0; j < N; j++) {
        if (curMp[j] == before[j]) continue;
        if (curMp[j] < 0) CurrentFrame.mvpMapPoints[j] = nullptr;         // removed by the histogram filter (:2143)
        else CurrentFrame.mvpMapPoints[j] = LastFrame.mvpMapPoints[curMp[j]];
    }
    return nmatches;
}

// ORBmatcher::SearchByBoW(pKF, F, vpMapPointMatches), F.Nleft != -1 (:223-421): msorb_search_by_bow_rig on the KeyFrame's and the
// frame's descriptors / FeatureVectors (per call; the resident-KeyFrame form serves one-camera frames).
template <class KeyFramePtr, class FrameT, class MapPointPtr>
int SearchByBoWRig(const KeyFramePtr& pKF, FrameT& F, std::vector<MapPointPtr>& vpMapPointMatches, float mfNNratio, bool mbCheckOrientation,
                   int device = 0) {
    const auto mpsKF = pKF->GetMapPointMatches();                          // :225
    BowSide kf, frame;
    kf.FillKeyFrame(pKF);
    kf.FlagGood(mpsKF);                                                    // :253-259
    frame.Fill(F.N, [&](int i) { return F.mDescriptors.row(i); }, F.mFeatVec, F.mvKeys);
    // the angle of a frame feature: mvKeys for the left camera's rows, mvKeysRight for the right camera's (:344-346, :365-367)
    frame.angle.resize(F.N);
    for (int i = 0; i < F.N; i++) frame.angle[i] = i < F.Nleft ? F.mvKeys[i].angle : F.mvKeysRight[i - F.Nleft].angle;
    msorb_bow_pair P;
    std::vector<int> m12, m21;
    BindBowPair(P, kf, frame, true, m12, m21);
    check(msorb_search_by_bow_rig(device, &P, F.Nleft, 50 /* TH_LOW */, mfNNratio, mbCheckOrientation), "msorb_search_by_bow_rig");
    vpMapPointMatches.assign(F.N, MapPointPtr());                          // :227
    for (int j = 0; j < F.N && j < (int)m21.size(); j++)
        if (m21[j] >= 0) vpMapPointMatches[j] = mpsKF[m21[j]];             // :336, :359
    return P.nmatches;
}

// ---- Fuse on a two-camera KeyFrame (pKF->GetNLeft() != -1): ORBmatcher::Fuse(pKF, vpMapPoints, th, bRight), ORBmatcher.cc:1404-1597,
// called once per camera by LocalMapping::SearchInNeighbors (LocalMapping.cc:793-795, 824-826).
// One camera of such a KeyFrame -> device frame + grid: what KeyFrame::GetFeaturesInArea(x, y, r, bRight) walks (KeyFrame.cc:826-836:
// mGrid / mvKeys for the left camera, mGridRight / mvKeysRight for the right one; descriptor rows [0, NLeft) / [NLeft, N)).  The left
// camera carries GetuRight (the stereo term of the error gate, :1515-1530); MS-SLAM's KeyFrame keeps its feature arrays private,
// hence the accessors (KeyFrame.h:357-405).
template <class FrameT, class KeyFramePtr>
void UploadKeyFrameCamera(DeviceFrame<FrameT>& dev, const KeyFramePtr& pKF, bool right) {
    const int NL = pKF->GetNLeft(), N = pKF->GetN();
    const int n = pKF->mbSparsified ? 0 : (right ? N - NL : NL), row0 = right ? NL : 0;
    std::vector<cv::KeyPoint> keys(n);
    static_assert(sizeof(keys[0]) == sizeof(msorb_keypoint), "cv::KeyPoint must be the 28-byte layout");
    std::vector<uint8_t> desc((size_t)n * 32, 0);
    std::vector<float> ur(n, -1.0f);
    for (int i = 0; i < n; i++) {
        keys[i] = right ? pKF->GetKeyRight(i) : pKF->GetKey(i);
        const auto d = pKF->GetDescriptor(row0 + i);
        if (!d.empty()) std::memcpy(&desc[(size_t)i * 32], d.template ptr<unsigned char>(0), 32);
        if (!right) ur[i] = pKF->GetuRight(i);
    }
    check(msorb_frame_set(dev.get(), reinterpret_cast<const msorb_keypoint*>(keys.data()), n, desc.data(), ur.data(), (float)pKF->mnMinX,
                          (float)pKF->mnMaxX, (float)pKF->mnMinY, (float)pKF->mnMaxY, pKF->mvScaleFactors.data(), (int)pKF->mvScaleFactors.size()),
          "msorb_frame_set");
}

// `devCam` holds the camera the call searches (UploadKeyFrameCamera(devCam, pKF, bRight)).  Geometry (:1410-1500, with the right
// camera's pose / centre / model for bRight) and the map mutation (:1563-1590) are the reference's statements (FuseGeometry /
// FuseCommit, ORBmatcher_device.h); the window search runs on the device.  bRight = false: the left camera's own keypoints gate
// (GetKeyPoint(idx) = mvKeys[idx] for idx < NLeft) — msorb_fuse_search.  bRight = true: the reference reads GetKeyPoint(idx) and
// GetuRight(idx) with the RIGHT-camera index idx before `idx += NLeft` (:1509, :1515, :1547), i.e. the LEFT keypoint of the same
// number while idx < NLeft and mvKeysRight[idx - NLeft] beyond (KeyFrame.h:377-385) — msorb_fuse_search_gated with exactly those
// values.  (GetuRight(idx) indexes mvuRight, which has NLeft entries for such a KeyFrame, Frame.cc:1069: for idx >= NLeft the
// reference reads past its end; here that reads as -1, "no stereo term", what every entry of that array holds.)
template <class KeyFramePtr, class MapPointPtr>
void FuseGeometryRig(const KeyFramePtr& pKF, const std::vector<MapPointPtr>& vpMapPoints, const float th, FuseQueries& Q, const bool bRight) {
    if (bRight) FuseGeometryOf(pKF, pKF->GetRightPose(), pKF->GetRightCameraCenter(), pKF->mpCamera2, vpMapPoints, th, Q);   // :1410-1414
    else FuseGeometry(pKF, vpMapPoints, th, Q);
}
template <class FrameT, class KeyFramePtr, class MapPointPtr>
int FuseRig(DeviceFrame<FrameT>& devCam, const KeyFramePtr& pKF, const std::vector<MapPointPtr>& vpMapPoints, const float th, const bool bRight) {
    FuseQueries Q;
    FuseGeometryRig(pKF, vpMapPoints, th, Q, bRight);
    const int nMPs = (int)vpMapPoints.size(), NL = pKF->GetNLeft();
    std::vector<int> bestIdx(nMPs, -1), bestDist(nMPs, 256);
    if (!bRight) {
        check(msorb_fuse_search(devCam.get(), pKF->mvInvLevelSigma2.data(), (int)pKF->mvInvLevelSigma2.size(), nMPs, Q.valid.data(), Q.u.data(),
                                Q.v.data(), Q.ur.data(), Q.level.data(), Q.radius.data(), Q.desc.data(), bestIdx.data(), bestDist.data()),
              "msorb_fuse_search");
        return FuseCommit(pKF, vpMapPoints, Q.valid, bestIdx, bestDist, 0);
    }
    const int nR = pKF->mbSparsified ? 0 : pKF->GetN() - NL;
    std::vector<cv::KeyPoint> gate(nR);
    std::vector<float> gateUr(nR, -1.0f);
    for (int j = 0; j < nR; j++) {
        gate[j] = pKF->GetKeyPoint(j);                                     // :1509, with the index GetFeaturesInArea(..., true) returned
        if (j < NL) gateUr[j] = pKF->GetuRight(j);                         // :1515
    }
    check(msorb_fuse_search_gated(devCam.get(), reinterpret_cast<const msorb_keypoint*>(gate.data()), gateUr.data(), pKF->mvInvLevelSigma2.data(),
                                  (int)pKF->mvInvLevelSigma2.size(), nMPs, Q.valid.data(), Q.u.data(), Q.v.data(), Q.ur.data(), Q.level.data(),
                                  Q.radius.data(), Q.desc.data(), bestIdx.data(), bestDist.data()),
          "msorb_fuse_search_gated");
    return FuseCommit(pKF, vpMapPoints, Q.valid, bestIdx, bestDist, NL);   // :1547
}

// ---- SearchForTriangulation between two KeyFrames of a two-camera rig (both have mpCamera2): ORBmatcher.cc:1168-1402 with the arms
// of :1195-1201 and :1294-1330.  Every feature of either camera takes part (the FeatureVector indexes all N rows; GetKeyPoint(idx)
// answers for both cameras).  For such KeyFrames bStereo1 / bStereo2 are false (`!pKF->mpCamera2 && ...`, :1243, :1267: with
// bOnlyStereo nothing is visited) and the epipole-distance test is off (`&& !pKF1->mpCamera2`, :1283).  The geometric test of :1332
// is the CAMERA MODEL's — pCamera1->epipolarConstrain(pCamera2, kp1, kp2, R12, t12, sigma1, sigma2), KannalaBrandt8 triangulates
// there — with the cameras and the relative pose picked per candidate from the side of the two features (:1294-1330): it stays the
// application's code, called back by msorb_search_for_triangulation_cb on the candidates the device found (descriptor distance
// <= TH_LOW inside a common BoW node), best distance first.  bCoarse short-circuits it as in the reference (`bCoarse || ...`).
template <class KeyFramePtr>
struct TriangulationRigTest {
    KeyFramePtr pKF1, pKF2;
    std::vector<cv::KeyPoint> kp1, kp2;
    bool bCoarse;
    // Tll, Tlr, Trl, Trr of :1196-1200 as rotation / translation (:1203-1204), indexed [bRight1][bRight2]
    typename std::decay<decltype(std::declval<KeyFramePtr>()->GetPose().rotationMatrix())>::type R[2][2];
    typename std::decay<decltype(std::declval<KeyFramePtr>()->GetPose().translation())>::type t[2][2];
    static int Accept(void* ctx, int idx1, int idx2) {
        TriangulationRigTest& T = *static_cast<TriangulationRigTest*>(ctx);
        if (T.bCoarse) return 1;
        const bool bRight1 = T.pKF1->FromRightImage(idx1), bRight2 = T.pKF2->FromRightImage(idx2);          // :1255, :1281
        auto* pCamera1 = bRight1 ? T.pKF1->mpCamera2 : T.pKF1->mpCamera;                                        // :1300-1327
        auto* pCamera2 = bRight2 ? T.pKF2->mpCamera2 : T.pKF2->mpCamera;
        const cv::KeyPoint &kp1 = T.kp1[idx1], &kp2 = T.kp2[idx2];
        return pCamera1->epipolarConstrain(pCamera2, kp1, kp2, T.R[bRight1][bRight2], T.t[bRight1][bRight2], T.pKF1->mvLevelSigma2[kp1.octave],
                                           T.pKF2->mvLevelSigma2[kp2.octave]) ? 1 : 0;                          // :1332
    }
};
template <class KeyFramePtr>
int SearchForTriangulationRig(const KeyFramePtr& pKF1, const KeyFramePtr& pKF2, std::vector<std::pair<size_t, size_t>>& vMatchedPairs,
                              const bool bOnlyStereo, const bool bCoarse, const bool mbCheckOrientation, int device = 0) {
    TriangulationRigTest<KeyFramePtr> T;
    T.pKF1 = pKF1; T.pKF2 = pKF2; T.bCoarse = bCoarse;
    {
        const auto T1w = pKF1->GetPose();                                                                       // :1175-1177, :1193-1200
        const auto Tw2 = pKF2->GetPoseInverse();
        const auto Tr1w = pKF1->GetRightPose();
        const auto Twr2 = pKF2->GetRightPoseInverse();
        const auto Tll = T1w * Tw2, Tlr = T1w * Twr2, Trl = Tr1w * Tw2, Trr = Tr1w * Twr2;
        T.R[0][0] = Tll.rotationMatrix(); T.t[0][0] = Tll.translation();
        T.R[0][1] = Tlr.rotationMatrix(); T.t[0][1] = Tlr.translation();
        T.R[1][0] = Trl.rotationMatrix(); T.t[1][0] = Trl.translation();
        T.R[1][1] = Trr.rotationMatrix(); T.t[1][1] = Trr.translation();
    }
    BowSide a, b;
    auto fill = [&](BowSide& side, const KeyFramePtr& pKF, std::vector<cv::KeyPoint>& kp) {
        const int n = pKF->GetN();
        kp.resize(n);
        for (int i = 0; i < n; i++) kp[i] = pKF->GetKeyPoint(i);                                                // :1254, :1280 (both cameras)
        side.Fill(n, [&](int i) { return pKF->GetDescriptor(i); }, pKF->GetFeatureVector(), kp);
        const auto mps = pKF->GetMapPointMatches();
        side.flag.assign(n, 0);
        for (int i = 0; i < n; i++) side.flag[i] = !mps[i] && !bOnlyStereo;                                     // :1237-1247 / :1259-1271, bStereo = false
    };
    fill(a, pKF1, T.kp1);
    fill(b, pKF2, T.kp2);
    msorb_bow_pair P;
    std::vector<int> m12, m21;
    BindBowPair(P, a, b, false, m12, m21);
    check(msorb_search_for_triangulation_cb(device, &P, 50 /* TH_LOW */, mbCheckOrientation, &TriangulationRigTest<KeyFramePtr>::Accept, &T),
          "msorb_search_for_triangulation_cb");
    vMatchedPairs.clear();                                                                                      // :1385-1393
    vMatchedPairs.reserve(P.nmatches);
    for (size_t i = 0; i < m12.size(); i++)
        if (m12[i] >= 0) vMatchedPairs.push_back(std::make_pair(i, (size_t)m12[i]));
    return P.nmatches;
}

}  // namespace msorb_host
}  // namespace ORB_SLAM3
#endif
