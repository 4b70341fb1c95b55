// msorb_host::KeyFrameStore under the reference's threading: TrackReferenceKeyFrame's SearchByBoW on the newest
// KeyFrame (Tracking thread) coincides with CreateNewMapPoints' searches on it (LocalMapping thread) while KeyFrames die
// (SetBadFlag -> Forget), the map is reset (Tracking::Reset -> Reset) and KeyFrame ids are recycled.  Four threads run
// SearchByBoWBatch over the SAME KeyFrames — every one misses on every KeyFrame at the start —, interleaved with Forget / Reset /
// re-adds; every result must equal the single-threaded one, nothing may throw ("unknown KeyFrame id"), and at the end the
// store holds exactly one device entry per live table entry (no duplicate add survived).
// Self-contained data (LCG); compiled against tests/slam_stub.   exit code 0 = pass
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "Frame.h"
#include "KeyFrame.h"
#include "MapPoint.h"
#include "ORBmatcher_device.h"

using namespace ORB_SLAM3;

static unsigned g_s = 12345;
static unsigned rnd() { g_s = g_s * 1664525u + 1013904223u; return g_s >> 8; }

static void fill_side(FeatureSide& S, int n, const std::vector<unsigned char>& base, int flips, int n_nodes) {
    std::vector<cv::KeyPoint> kps(n);
    std::vector<unsigned char> d((size_t)n * 32);
    for (int i = 0; i < n; i++) {
        kps[i].pt.x = (float)(rnd() % 1200); kps[i].pt.y = (float)(rnd() % 370);
        kps[i].octave = (int)(rnd() % 8); kps[i].angle = (float)(rnd() % 360); kps[i].size = 31.f;
        memcpy(&d[(size_t)i * 32], &base[(size_t)(i % (base.size() / 32)) * 32], 32);
        for (int f = 0; f < flips; f++) d[(size_t)i * 32 + (rnd() & 31)] ^= (unsigned char)(1u << (rnd() & 7));
    }
    S.SetFeatures(kps, d.data());
    S.mFeatVec.clear();
    for (int i = 0; i < n; i++) S.mFeatVec[(unsigned)((i % (int)(base.size() / 32)) % n_nodes)].push_back((unsigned)i);
    S.mvScaleFactors.resize(8); S.mvLevelSigma2.resize(8);
    float s = 1.f;
    for (int l = 0; l < 8; l++) { S.mvScaleFactors[l] = s; S.mvLevelSigma2[l] = s * s; s *= 1.2f; }
}

struct KFAccess : KeyFrame {  // the stand-in keeps the feature arrays protected like MS-SLAM's KeyFrame
    void Fill(int n, const std::vector<unsigned char>& base, int n_nodes) {
        fill_side(*this, n, base, 6, n_nodes);
        for (int i = 0; i < n; i++)
            if (rnd() % 10 < 8) { auto p = std::make_shared<MapPoint>(); p->mnId = (unsigned long)i; mvpMapPoints[i] = p; }
    }
};

int main() {
    const int K = 6, N = 1200, NODES = 90, T = 4, ITERS = 60;
    std::vector<unsigned char> base((size_t)N * 32);
    for (auto& b : base) b = (unsigned char)rnd();
    Frame F;
    fill_side(F, N, base, 10, NODES);
    std::vector<std::shared_ptr<KeyFrame>> kfs(K);
    for (int k = 0; k < K; k++) {
        auto kf = std::make_shared<KFAccess>();
        kf->Fill(N - 20 * k, base, NODES);
        kf->mnId = (unsigned long)k;
        kfs[k] = kf;
    }
    msorb_host::KeyFrameStore store(0);
    std::vector<std::vector<std::shared_ptr<MapPoint>>> ref;
    const std::vector<int> nref = msorb_host::SearchByBoWBatch(store, kfs, F, ref, 0.7f, true);
    int total = 0;
    for (int n : nref) total += n;
    if (total < 100) { fprintf(stderr, "degenerate workload: %d matches\n", total); return 3; }
    store.Reset();
    std::atomic<int> bad{0}, thrown{0};
    auto worker = [&](int t) {
        unsigned s = 777u + 31u * (unsigned)t;
        auto r = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
        for (int it = 0; it < ITERS; it++) {
            try {
                if (r() % 4 == 0) store.Forget((unsigned long)(r() % K));      // KeyFrame::SetBadFlag on another thread
                if (t == 0 && it % 17 == 16) store.Reset();                     // Tracking::Reset
                std::vector<std::vector<std::shared_ptr<MapPoint>>> out;
                const std::vector<int> nm = msorb_host::SearchByBoWBatch(store, kfs, F, out, 0.7f, true);
                if (nm != nref) { bad++; continue; }
                for (int k = 0; k < K; k++)
                    for (size_t j = 0; j < out[k].size(); j++)
                        if (out[k][j] != ref[k][j]) { bad++; k = K; break; }
            } catch (const std::exception& e) {
                fprintf(stderr, "thread %d: %s\n", t, e.what());
                thrown++;
            }
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) th.emplace_back(worker, t);
    for (auto& x : th) x.join();
    // a recycled id (Tracking::Reset restarts KeyFrame::nNextId): a NEW KeyFrame object with mnId 0 and the same N must not be
    // answered with the old object's device copy
    auto fresh = std::make_shared<KFAccess>();
    fresh->Fill(N, base, NODES);
    fresh->mnId = 0;
    std::vector<std::vector<std::shared_ptr<MapPoint>>> out_old, out_new;
    msorb_host::SearchByBoWBatch(store, std::vector<std::shared_ptr<KeyFrame>>{kfs[0]}, F, out_old, 0.7f, true);
    const auto l_old = store.Ensure(kfs[0]);
    const auto l_new = store.Ensure(std::shared_ptr<KeyFrame>(fresh));
    const int recycled_ok = l_old->id != l_new->id;
    const size_t resident = store.Resident();
    const int dev_count = msorb_kf_store_count(store.get());
    printf("{\"mismatches\": %d, \"exceptions\": %d, \"resident\": %zu, \"device_entries\": %d, \"recycled_id_readded\": %d, \"matches\": %d}\n",
           bad.load(), thrown.load(), resident, dev_count, recycled_ok, total);
    // device entries = table entries + the one lease still held on the replaced KeyFrame 0 (l_old)
    const bool ok = bad == 0 && thrown == 0 && recycled_ok && dev_count == (int)resident + 1;
    store.Shutdown();
    return ok ? 0 : 1;
}
