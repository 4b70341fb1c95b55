// Compiles OUR drop-in class ORB_SLAM3::ORBmatcher (ms-slam_amd/host/ORBmatcher.{h,cc}, the declaration of
// /root/reference/include/ORBmatcher.h:36-112) against the stand-ins of tests/slam_stub and drives its loop-closing /
// initialisation methods the way LoopClosing.cc / Tracking.cc do.  Reads one scene file, writes results plus the
// projections the host mirrors computed (the Python test feeds those to the oracle).
// usage: dropin_orbmatcher <in.bin> <out.bin>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ORBmatcher.h"
#include "ORBmatcher_loop_device.h"

using namespace ORB_SLAM3;
typedef std::shared_ptr<MapPoint> MP;
typedef std::shared_ptr<KeyFrame> KF;

template <class T>
static std::vector<T> rd(FILE* f, size_t n) {
    std::vector<T> v(n);
    if (n && fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(3); }
    return v;
}
template <class T>
static void wr(FILE* f, const std::vector<T>& v) { if (!v.empty()) fwrite(v.data(), sizeof(T), v.size(), f); }
static void wri(FILE* f, int v) { fwrite(&v, 4, 1, f); }

static long unsigned int g_next_id = 1;
static std::vector<MP> read_points(FILE* f, int n) {   // state (0 none, 1 good, 2 bad), pos, normal, maxd, mind, desc
    const auto state = rd<unsigned char>(f, n);
    const auto pos = rd<float>(f, (size_t)3 * n), nrm = rd<float>(f, (size_t)3 * n), maxd = rd<float>(f, n), mind = rd<float>(f, n);
    const auto desc = rd<unsigned char>(f, (size_t)32 * n);
    std::vector<MP> out(n);
    for (int i = 0; i < n; i++) {
        if (!state[i]) continue;
        auto p = std::make_shared<MapPoint>();
        p->mnId = g_next_id++;
        p->mbBad = state[i] == 2;
        memcpy(p->pos.v, &pos[3 * i], 12); memcpy(p->normal.v, &nrm[3 * i], 12);
        p->mfMaxDistance = maxd[i]; p->mfMinDistance = mind[i];
        memcpy(p->descriptor, &desc[(size_t)32 * i], 32);
        out[i] = p;
    }
    return out;
}
static KF read_kf(FILE* f, int n, const std::vector<float>& scale, const std::vector<float>& sigma2, const float* cam, float logs,
                  GeometricCamera* camera, int rows, int cols) {
    auto kf = std::make_shared<KeyFrame>();
    kf->mnId = g_next_id++;
    const auto kps = rd<cv::KeyPoint>(f, n);
    const auto desc = rd<unsigned char>(f, (size_t)32 * n);
    kf->SetFeatures(kps, desc.data());
    kf->mvScaleFactors = scale; kf->mvLevelSigma2 = sigma2;
    for (float s2 : sigma2) kf->mvInvLevelSigma2.push_back(1.0f / s2);
    kf->mnScaleLevels = (int)scale.size(); kf->mfLogScaleFactor = logs;
    kf->fx = cam[0]; kf->fy = cam[1]; kf->cx = cam[2]; kf->cy = cam[3];
    kf->mpCamera = camera;
    kf->mnMinX = 0; kf->mnMinY = 0; kf->mnMaxX = cols; kf->mnMaxY = rows;
    const auto mps = read_points(f, n);
    for (int i = 0; i < n; i++)
        if (mps[i]) { kf->AddMapPoint(mps[i], i); mps[i]->obsIdx[kf.get()] = i; mps[i]->nObs = 2; }
    const auto node = rd<int>(f, n);
    DBoW2::FeatureVector fv;
    for (int i = 0; i < n; i++) fv.addFeature((DBoW2::NodeId)node[i], (unsigned)i);
    kf->SetFeatureVector(fv);
    return kf;
}
static void dump_queries(FILE* o, const msorb_host::Sim3Queries& Q) {
    wr(o, Q.valid); wr(o, Q.u); wr(o, Q.v); wr(o, Q.level);
}
// pointer vector -> ids (index of the pointed-to object in `pool`, -1 null, -2 unknown)
static std::vector<int> ids_of(const std::vector<MP>& v, const std::vector<MP>& pool) {
    std::vector<int> out(v.size(), -1);
    for (size_t i = 0; i < v.size(); i++) {
        if (!v[i]) continue;
        out[i] = -2;
        for (size_t k = 0; k < pool.size(); k++)
            if (pool[k] == v[i]) { out[i] = (int)k; break; }
    }
    return out;
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 3;
    const auto hdr = rd<int>(f, 6);
    const int N1 = hdr[0], N2 = hdr[1], nl = hdr[2], P = hdr[3], rows = hdr[4], cols = hdr[5];
    const auto cam = rd<float>(f, 8);   // fx fy cx cy logs th_sim3 th_proj ratio
    const auto scale = rd<float>(f, nl), sigma2 = rd<float>(f, nl);
    GeometricCamera camera;
    camera.fx = cam[0]; camera.fy = cam[1]; camera.cx = cam[2]; camera.cy = cam[3];
    KF kf1 = read_kf(f, N1, scale, sigma2, cam.data(), cam[4], &camera, rows, cols);
    KF kf2 = read_kf(f, N2, scale, sigma2, cam.data(), cam[4], &camera, rows, cols);
    const auto t2 = rd<float>(f, 3);
    kf2->SetPose(Sophus::SE3f(Eigen::Matrix3f{{1, 0, 0, 0, 1, 0, 0, 0, 1}}, Eigen::Vector3f{{t2[0], t2[1], t2[2]}}));
    const auto already12 = rd<int>(f, N1);
    const std::vector<MP> cand = read_points(f, P);            // loop-closure candidate points (projected into kf2)
    const auto matched_init = rd<int>(f, N2);                  // initial vpMatched of the projection forms: candidate index or -1
    auto prev = rd<float>(f, (size_t)2 * N1);
    const auto loopflag1 = rd<unsigned char>(f, N1), loopflag2 = rd<unsigned char>(f, N2);
    fclose(f);
    const std::vector<MP> mps1 = kf1->GetMapPointMatches(), mps2 = kf2->GetMapPointMatches();
    FILE* o = fopen(argv[2], "wb");
    ORBmatcher matcher(0.9f, true);

    // ---- SearchBySim3 (LoopClosing: matcher.SearchBySim3(mpCurrentKF, pKF, vpMapPointMatches, gScm, 7.5))
    const Sophus::Sim3f S12(1.0f, Eigen::Matrix3f{{1, 0, 0, 0, 1, 0, 0, 0, 1}}, Eigen::Vector3f{{-t2[0], -t2[1], -t2[2]}});
    {
        std::vector<MP> vpMatches12(N1);
        for (int i = 0; i < N1; i++)
            if (already12[i] >= 0) vpMatches12[i] = mps2[already12[i]];
        msorb_host::Sim3Queries Q1, Q2;
        msorb_host::Sim3PairGeometry(kf1, kf2, vpMatches12, S12, Q1, Q2);
        const int nFound = matcher.SearchBySim3(kf1, kf2, vpMatches12, S12, cam[5]);
        wri(o, nFound);
        wr(o, ids_of(vpMatches12, mps2));
        dump_queries(o, Q1); dump_queries(o, Q2);
    }
    // ---- the three SearchByProjection(pKF, Scw, ...) forms on kf2 (Scw = kf2's pose as a Sim3 of scale 1.25)
    Sophus::Sim3f Scw(1.25f, kf2->GetPose().rotationMatrix(), kf2->GetPose().translation() * 1.25f);
    {
        msorb_host::Sim3Queries Q;
        std::vector<MP> vpMatched(N2);
        for (int j = 0; j < N2; j++)
            if (matched_init[j] >= 0) vpMatched[j] = cand[matched_init[j]];
        std::set<MP> found(vpMatched.begin(), vpMatched.end());
        found.erase(MP());
        msorb_host::ProjectSim3<false>(kf2, Scw, cand, [&](int, const MP& p) { return !p || p->isBad() || found.count(p); }, Q);
        const int n = matcher.SearchByProjection(kf2, Scw, cand, vpMatched, (int)cam[6], cam[7]);
        wri(o, n);
        wr(o, ids_of(vpMatched, cand));
        dump_queries(o, Q);
    }
    {   // (pKF, Scw, vpPoints, vpPointsKFs, vpMatched, vpMatchedKF, ...): bad / null candidates are not passed (the reference dereferences them)
        std::vector<MP> good;
        std::vector<int> good_idx;
        for (int i = 0; i < P; i++)
            if (cand[i]) { good.push_back(cand[i]); good_idx.push_back(i); }
        std::vector<KF> pointKFs(good.size());
        for (size_t i = 0; i < good.size(); i++) pointKFs[i] = (i & 1) ? kf1 : kf2;
        std::vector<MP> vpMatched(N2);
        std::vector<KF> vpMatchedKF(N2);
        for (int j = 0; j < N2; j++)
            if (matched_init[j] >= 0) vpMatched[j] = cand[matched_init[j]];
        std::set<MP> found(vpMatched.begin(), vpMatched.end());
        found.erase(MP());
        msorb_host::Sim3Queries Q;
        msorb_host::ProjectSim3<true>(kf2, Scw, good, [&](int, const MP& p) { return p->isBad() || found.count(p); }, Q);
        const int n = matcher.SearchByProjection(kf2, Scw, good, pointKFs, vpMatched, vpMatchedKF, (int)cam[6], cam[7]);
        wri(o, n);
        wri(o, (int)good.size());
        wr(o, good_idx);
        wr(o, ids_of(vpMatched, cand));
        std::vector<int> kfid(N2, -1);
        for (int j = 0; j < N2; j++) kfid[j] = !vpMatchedKF[j] ? -1 : (vpMatchedKF[j] == kf1 ? 1 : 2);
        wr(o, kfid);
        dump_queries(o, Q);
    }
    {   // SearchByProjectionLoop
        std::vector<MP> good;
        std::vector<int> good_idx;
        for (int i = 0; i < P; i++)
            if (cand[i]) { good.push_back(cand[i]); good_idx.push_back(i); }
        std::vector<MP> vpMatched(good.size());
        std::vector<KF> vpMatchedKF(good.size());
        for (size_t i = 0; i < good.size(); i += 7) vpMatched[i] = mps1[0] ? mps1[0] : good[0];   // some points are already matched
        msorb_host::Sim3Queries Q;
        const std::vector<MP> before(vpMatched);
        msorb_host::ProjectSim3<false>(kf2, Scw, good, [&](int i, const MP& p) { return p->isBad() || before[i]; }, Q);
        const int n = matcher.SearchByProjectionLoop(kf2, Scw, good, vpMatched, vpMatchedKF, (int)cam[6], cam[7]);
        wri(o, n);
        std::vector<int> res(good.size(), -1);
        for (size_t i = 0; i < good.size(); i++)
            if (vpMatched[i] && vpMatched[i] != before[i]) res[i] = ids_of({vpMatched[i]}, mps2)[0];
            else if (vpMatched[i]) res[i] = -3;   // was matched before the call
        wr(o, res);
        int okkf = 1;
        for (size_t i = 0; i < good.size(); i++)
            if (res[i] >= 0 && vpMatchedKF[i] != kf2) okkf = 0;
        wri(o, okkf);
        dump_queries(o, Q);
    }
    // ---- Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) on kf2 (LoopClosing::SearchAndFuse); mutates kf2 -> run after the forms above
    {
        std::vector<MP> good;
        std::vector<int> good_idx;
        for (int i = 0; i < P; i++)
            if (cand[i]) { good.push_back(cand[i]); good_idx.push_back(i); }
        const std::set<MP> inKF = kf2->GetMapPoints();
        msorb_host::Sim3Queries Q;
        msorb_host::ProjectSim3<false>(kf2, Scw, good, [&](int, const MP& p) { return p->isBad() || inKF.count(p); }, Q);
        std::vector<MP> vpReplacePoint(good.size());
        std::vector<long> log;
        MapPoint::log = &log;
        const int nFused = matcher.Fuse(kf2, Scw, good, 4.0f, vpReplacePoint);
        MapPoint::log = nullptr;
        wri(o, nFused);
        wr(o, ids_of(vpReplacePoint, mps2));
        wri(o, (int)log.size());
        std::vector<int> log32(log.begin(), log.end());
        wr(o, log32);
        std::vector<int> mp_ids(good.size());
        for (size_t i = 0; i < good.size(); i++) mp_ids[i] = (int)good[i]->mnId;
        wr(o, mp_ids);
        dump_queries(o, Q);
    }
    // ---- SearchForInitialization (Tracking::MonocularInitialization): frames with the keyframes' features
    {
        Frame F1, F2;
        F1.mnId = 1001; F2.mnId = 1002;
        const auto k1 = kf1->GetAllKeyUn(), k2 = kf2->GetAllKeyUn();
        std::vector<unsigned char> d1((size_t)32 * N1), d2((size_t)32 * N2);
        for (int i = 0; i < N1; i++) memcpy(&d1[(size_t)32 * i], kf1->GetDescriptor(i).ptr<unsigned char>(0), 32);
        for (int i = 0; i < N2; i++) memcpy(&d2[(size_t)32 * i], kf2->GetDescriptor(i).ptr<unsigned char>(0), 32);
        F1.SetFeatures(k1, d1.data()); F2.SetFeatures(k2, d2.data());
        for (Frame* F : {&F1, &F2}) { F->mvScaleFactors = scale; F->mnMinX = 0; F->mnMaxX = (float)cols; F->mnMinY = 0; F->mnMaxY = (float)rows; }
        std::vector<cv::Point2f> vbPrevMatched(N1);
        for (int i = 0; i < N1; i++) vbPrevMatched[i] = cv::Point2f{prev[2 * i], prev[2 * i + 1]};
        std::vector<int> vnMatches12;
        const int n = matcher.SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, 100);
        wri(o, n);
        wr(o, vnMatches12);
        std::vector<float> pv((size_t)2 * N1);
        for (int i = 0; i < N1; i++) { pv[2 * i] = vbPrevMatched[i].x; pv[2 * i + 1] = vbPrevMatched[i].y; }
        wr(o, pv);
    }
    // ---- SearchByBoW, the loop form (:1018-1166), next to the plain KeyFrame-KeyFrame form on the same pair
    {
        long unsigned int nCurrentId = 77;
        for (int i = 0; i < N1; i++) if (mps1[i] && loopflag1[i]) mps1[i]->mnLoopPointForKF = nCurrentId;
        for (int i = 0; i < N2; i++) if (mps2[i] && loopflag2[i]) mps2[i]->mnLoopPointForKF = nCurrentId;
        std::vector<KF> cKF, lKF;
        std::vector<MP> cMP, lMP;
        const int n = matcher.SearchByBoW(kf1, kf2, cKF, cMP, lKF, lMP, nCurrentId);
        wri(o, n);
        wr(o, ids_of(cMP, mps1));
        wr(o, ids_of(lMP, kf2->GetMapPointMatches()));
        int marked = 1;
        for (auto& p : cMP) if (p->mnLoopPointForKF != nCurrentId) marked = 0;
        for (auto& p : lMP) if (p->mnLoopPointForKF != nCurrentId) marked = 0;
        wri(o, marked && cKF.size() == cMP.size() && lKF.size() == lMP.size());
    }
    // ---- SearchForTriangulation / SearchByBoW(pKF, F) through the class (resident KeyFrame store) against the per-call templates
    {
        std::vector<std::pair<size_t, size_t>> viaClass, viaCall;
        const int a = matcher.SearchForTriangulation(kf1, kf2, viaClass, false, true);
        const int b = msorb_host::SearchForTriangulation(kf1, kf2, viaCall, false, true, true, 0);
        Frame F;
        F.mnId = 2001;
        const auto k2 = kf2->GetAllKeyUn();
        std::vector<unsigned char> d2((size_t)32 * N2);
        for (int i = 0; i < N2; i++) memcpy(&d2[(size_t)32 * i], kf2->GetDescriptor(i).ptr<unsigned char>(0), 32);
        F.SetFeatures(k2, d2.data());
        F.mFeatVec = kf2->GetFeatureVector();
        std::vector<MP> m1, m2v;
        const int c = matcher.SearchByBoW(kf1, F, m1);
        const int d = msorb_host::SearchByBoW(kf1, F, m2v, 0.9f, true, 0);
        wri(o, a); wri(o, c);
        wri(o, a == b && viaClass == viaCall && c == d && m1 == m2v);
    }
    // ---- DescriptorDistance
    {
        int acc = 0;
        for (int i = 0; i < std::min(N1, N2); i++) acc += ORBmatcher::DescriptorDistance(kf1->GetDescriptor(i), kf2->GetDescriptor(i));
        wri(o, acc);
        wri(o, ORBmatcher::TH_LOW * 10000 + ORBmatcher::TH_HIGH * 100 + ORBmatcher::HISTO_LENGTH);
    }
    fclose(o);
    return 0;
}
