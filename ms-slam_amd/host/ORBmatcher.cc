// ORB_SLAM3::ORBmatcher on libmsorb.so — see ORBmatcher.h.  Every method keeps the reference's signature and return
// value and forwards to the msorb_host templates (ORBmatcher_device.h, ORBmatcher_loop_device.h): the reference's own
// geometry code on the host, the Hamming searches on the GPU behind the C ABI.  The device copies of the frames /
// keyframes a thread searches in are kept per calling thread (the matcher is entered from the Tracking, LocalMapping and
// LoopClosing threads, each with stack-constructed ORBmatcher objects: Tracking.cc:2835, LocalMapping.cc:438,787,
// LoopClosing.cc:594) and re-uploaded only when the object changes.
#include "ORBmatcher.h"

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <cstdlib>
#include <memory>

#include "ORBmatcher_device.h"
#include "ORBmatcher_loop_device.h"
#include "ORBmatcher_rig_device.h"

namespace ORB_SLAM3 {

const int ORBmatcher::TH_HIGH = 100;
const int ORBmatcher::TH_LOW = 50;
const int ORBmatcher::HISTO_LENGTH = 30;

namespace {
int matcher_device() {  // MSORB_MATCHER_DEVICE, else MSORB_DEVICE, else 0
    static const int dev = [] {
        const char* e = std::getenv("MSORB_MATCHER_DEVICE");
        if (!e) e = std::getenv("MSORB_DEVICE");
        return e ? std::atoi(e) : 0;
    }();
    return dev;
}
// One device-resident Frame and two device-resident KeyFrames per calling thread, keyed by identity: consecutive searches
// in the same object (TrackWithMotionModel retries, SearchLocalPoints after TrackReferenceKeyFrame, the covisible loop of
// SearchInNeighbors) upload once.
struct ThreadCache {
    msorb_host::DeviceFrame<Frame> frame{matcher_device()}, frame2{matcher_device()}, kf[2] = {msorb_host::DeviceFrame<Frame>(matcher_device()),
                                                                                            msorb_host::DeviceFrame<Frame>(matcher_device())};
    const void* frame_key = nullptr;
    long unsigned int frame_id = ~0ul;
    const void* frame2_key = nullptr;
    long unsigned int frame2_id = ~0ul;
    // the two cameras of a two-camera frame (Frame::Nleft != -1): ORBmatcher_rig_device.h
    msorb_host::DeviceFrame<Frame> rig[2] = {msorb_host::DeviceFrame<Frame>(matcher_device()), msorb_host::DeviceFrame<Frame>(matcher_device())};
    const void* rig_key = nullptr;
    long unsigned int rig_id = ~0ul;
    // the two cameras of a two-camera KeyFrame (Fuse(pKF, ..., bRight), ORBmatcher_rig_device.h FuseRig)
    msorb_host::DeviceFrame<Frame> kfcam[2] = {msorb_host::DeviceFrame<Frame>(matcher_device()), msorb_host::DeviceFrame<Frame>(matcher_device())};
    const void* kfcam_key[2] = {nullptr, nullptr};
    long unsigned int kfcam_id[2] = {~0ul, ~0ul};
    bool kfcam_sparsified[2] = {false, false};
    int kfcam_n[2] = {-1, -1};
    const void* kf_key[2] = {nullptr, nullptr};
    long unsigned int kf_id[2] = {~0ul, ~0ul};
    bool kf_sparsified[2] = {false, false};
    int kf_n[2] = {-1, -1};
};
// KeyFrames resident on the device for the BoW-node searches, shared by the three threads that run the matcher.  Heap
// allocated and never destroyed by a static destructor (those may run after the HIP runtime has gone): msorb_host::Shutdown()
// releases it.
msorb_host::KeyFrameStore& keyframe_store() {
    static msorb_host::KeyFrameStore* store = new msorb_host::KeyFrameStore(matcher_device());
    return *store;
}
// the calling thread's device frames: released by msorb_host::ReleaseThread() / at thread exit.  The destructor of a
// DeviceFrame tolerates a dead runtime (msorb_frame_destroy skips the frees when hipSetDevice fails).
thread_local std::unique_ptr<ThreadCache> g_thread_cache;
ThreadCache& cache() {
    if (!g_thread_cache) g_thread_cache.reset(new ThreadCache());
    return *g_thread_cache;
}
void drop_thread_cache() { g_thread_cache.reset(); }
// the capability gap is reported like every other failure of the host layer (msorb_host::check throws std::runtime_error): the
// application decides what to do with a rig this build does not serve; it is never answered with left-camera associations
[[noreturn]] void unsupported_rig(const char* what) {
    throw std::runtime_error(std::string("msorb: ") + what + " — a call the reference itself cannot answer (it reads a camera / pose the KeyFrame "
                             "does not have); the two-camera branches this build serves are listed in ORBmatcher_rig_device.h");
}
msorb_host::DeviceFrame<Frame>& device_frame(const Frame& F, bool second = false) {
    ThreadCache& c = cache();
    const void*& key = second ? c.frame2_key : c.frame_key;
    long unsigned int& id = second ? c.frame2_id : c.frame_id;
    msorb_host::DeviceFrame<Frame>& d = second ? c.frame2 : c.frame;
    if (key != &F || id != F.mnId) { d.Upload(F); key = &F; id = F.mnId; }
    return d;
}
// both cameras of a two-camera frame, uploaded once per Frame object
msorb_host::DeviceFrame<Frame>* device_rig(const Frame& F) {
    ThreadCache& c = cache();
    if (c.rig_key != &F || c.rig_id != F.mnId) {
        msorb_host::UploadCamera(c.rig[0], F, false);
        msorb_host::UploadCamera(c.rig[1], F, true);
        c.rig_key = &F; c.rig_id = F.mnId;
    }
    return c.rig;
}
msorb_host::DeviceFrame<Frame>& device_keyframe(const std::shared_ptr<KeyFrame>& pKF, int slot = 0) {
    ThreadCache& c = cache();
    // a KeyFrame's features change once in its life: when map sparsification compacts them (KeyFrame::EraseBadDescriptor)
    if (c.kf_key[slot] != pKF.get() || c.kf_id[slot] != pKF->mnId || c.kf_sparsified[slot] != pKF->mbSparsified ||
        c.kf_n[slot] != pKF->GetN()) {
        c.kf[slot].UploadKeyFrame(pKF);
        c.kf_key[slot] = pKF.get(); c.kf_id[slot] = pKF->mnId; c.kf_sparsified[slot] = pKF->mbSparsified; c.kf_n[slot] = pKF->GetN();
    }
    return c.kf[slot];
}
// one camera of a two-camera KeyFrame, uploaded once per KeyFrame and camera (LocalMapping::SearchInNeighbors calls Fuse for the
// left, then for the right camera of every neighbour, LocalMapping.cc:793-795)
msorb_host::DeviceFrame<Frame>& device_keyframe_camera(const std::shared_ptr<KeyFrame>& pKF, bool right) {
    ThreadCache& c = cache();
    const int k = right ? 1 : 0;
    if (c.kfcam_key[k] != pKF.get() || c.kfcam_id[k] != pKF->mnId || c.kfcam_sparsified[k] != pKF->mbSparsified || c.kfcam_n[k] != pKF->GetN()) {
        msorb_host::UploadKeyFrameCamera(c.kfcam[k], pKF, right);
        c.kfcam_key[k] = pKF.get(); c.kfcam_id[k] = pKF->mnId; c.kfcam_sparsified[k] = pKF->mbSparsified; c.kfcam_n[k] = pKF->GetN();
    }
    return c.kfcam[k];
}
}  // namespace

// Lifetime hooks for the code around the matcher: declared in ORBmatcher.h.
namespace msorb_host {
void ForgetKeyFrame(unsigned long mnId) { keyframe_store().Forget(mnId); }
void ResetKeyFrames() { keyframe_store().Reset(); }
size_t ResidentKeyFrames() { return keyframe_store().Resident(); }
size_t ResidentKeyFrameBytes() { return keyframe_store().ResidentBytes(); }
void SetKeyFrameBudget(size_t max_keyframes, size_t max_bytes) { keyframe_store().SetBudget(max_keyframes, max_bytes); }
void KeyFrameStoreStats(unsigned long long* uploads, unsigned long long* expired, unsigned long long* evicted) {
    const auto c = keyframe_store().Stats();
    if (uploads) *uploads = c.adds;
    if (expired) *expired = c.expired;
    if (evicted) *evicted = c.evicted;
}
void ReleaseThread() { drop_thread_cache(); }
void Shutdown() {
    drop_thread_cache();
    keyframe_store().Shutdown();
}
}  // namespace msorb_host

ORBmatcher::ORBmatcher(float nnratio, bool checkOri) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

int ORBmatcher::SearchByProjection(Frame& F, const std::vector<std::shared_ptr<MapPoint>>& vpMapPoints, const float th,
                                   const bool bFarPoints, const float thFarPoints) {
    if (F.Nleft != -1) {   // two cameras: both arms of :43-213 (ORBmatcher_rig_device.h)
        msorb_host::DeviceFrame<Frame>* rig = device_rig(F);
        return msorb_host::SearchByProjectionRig(rig[0], rig[1], F, vpMapPoints, th, bFarPoints, thFarPoints, mfNNratio);
    }
    return msorb_host::SearchByProjection(device_frame(F), F, vpMapPoints, th, bFarPoints, thFarPoints, mfNNratio);
}

int ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono) {
    if (CurrentFrame.Nleft != -1) {   // two cameras: :1941-2152 with the right arm :2059-2124
        msorb_host::DeviceFrame<Frame>* rig = device_rig(CurrentFrame);
        return msorb_host::SearchByProjectionRig(rig[0], rig[1], CurrentFrame, LastFrame, th, bMono, mbCheckOrientation);
    }
    return msorb_host::SearchByProjection(device_frame(CurrentFrame), CurrentFrame, LastFrame, th, bMono, mbCheckOrientation);
}

int ORBmatcher::SearchByProjection(Frame& CurrentFrame, std::shared_ptr<KeyFrame> pKF,
                                   const std::set<std::shared_ptr<MapPoint>>& sAlreadyFound, const float th, const int ORBdist) {
    return msorb_host::SearchByProjection(device_frame(CurrentFrame), CurrentFrame, pKF, sAlreadyFound, th, ORBdist, mbCheckOrientation);
}

int ORBmatcher::SearchByProjection(std::shared_ptr<KeyFrame> pKF, Sophus::Sim3<float>& Scw,
                                   const std::vector<std::shared_ptr<MapPoint>>& vpPoints,
                                   std::vector<std::shared_ptr<MapPoint>>& vpMatched, int th, float ratioHamming) {
    return msorb_host::SearchByProjection(device_keyframe(pKF), pKF, Scw, vpPoints, vpMatched, th, ratioHamming);
}

int ORBmatcher::SearchByProjectionLoop(std::shared_ptr<KeyFrame> pKF, Sophus::Sim3<float>& Scw,
                                       const std::vector<std::shared_ptr<MapPoint>>& vpPoints,
                                       std::vector<std::shared_ptr<MapPoint>>& vpMatched,
                                       std::vector<std::shared_ptr<KeyFrame>>& vpMatchedKF, int th, float ratioHamming) {
    return msorb_host::SearchByProjectionLoop(device_keyframe(pKF), pKF, Scw, vpPoints, vpMatched, vpMatchedKF, th, ratioHamming);
}

int ORBmatcher::SearchByProjection(std::shared_ptr<KeyFrame> pKF, Sophus::Sim3<float>& Scw,
                                   const std::vector<std::shared_ptr<MapPoint>>& vpPoints,
                                   const std::vector<std::shared_ptr<KeyFrame>>& vpPointsKFs,
                                   std::vector<std::shared_ptr<MapPoint>>& vpMatched,
                                   std::vector<std::shared_ptr<KeyFrame>>& vpMatchedKF, int th, float ratioHamming) {
    return msorb_host::SearchByProjection(device_keyframe(pKF), pKF, Scw, vpPoints, vpPointsKFs, vpMatched, vpMatchedKF, th, ratioHamming);
}

int ORBmatcher::SearchByBoW(std::shared_ptr<KeyFrame> pKF, Frame& F, std::vector<std::shared_ptr<MapPoint>>& vpMapPointMatches) {
    // (:276-309: on a two-camera frame every KeyFrame feature keeps a best / second per camera inside its BoW node)
    if (F.Nleft != -1) return msorb_host::SearchByBoWRig(pKF, F, vpMapPointMatches, mfNNratio, mbCheckOrientation, matcher_device());
    std::vector<std::vector<std::shared_ptr<MapPoint>>> out;
    const int n = msorb_host::SearchByBoWBatch(keyframe_store(), std::vector<std::shared_ptr<KeyFrame>>{pKF}, F, out, mfNNratio,
                                               mbCheckOrientation)[0];
    vpMapPointMatches = std::move(out[0]);
    return n;
}

int ORBmatcher::SearchByBoW(std::shared_ptr<KeyFrame> pKF1, std::shared_ptr<KeyFrame> pKF2,
                            std::vector<std::shared_ptr<MapPoint>>& vpMatches12) {
    return msorb_host::SearchByBoWKeyFrames(pKF1, pKF2, vpMatches12, mfNNratio, mbCheckOrientation, matcher_device());
}

int ORBmatcher::SearchByBoW(std::shared_ptr<KeyFrame> pKF1, std::shared_ptr<KeyFrame> pKF2,
                            std::vector<std::shared_ptr<KeyFrame>>& vpMatchedCurrentKeyFrame,
                            std::vector<std::shared_ptr<MapPoint>>& vpMatchedCurrentMapPoint,
                            std::vector<std::shared_ptr<KeyFrame>>& vpMatchedLoopKeyFrame,
                            std::vector<std::shared_ptr<MapPoint>>& vpMatchedLoopMapPoint, long unsigned int& nCurrentId) {
    return msorb_host::SearchByBoWLoop(pKF1, pKF2, vpMatchedCurrentKeyFrame, vpMatchedCurrentMapPoint, vpMatchedLoopKeyFrame,
                                       vpMatchedLoopMapPoint, nCurrentId, mfNNratio, matcher_device());
}

int ORBmatcher::SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12,
                                        int windowSize) {
    return msorb_host::SearchForInitialization(device_frame(F1), device_frame(F2, true), F1, F2, vbPrevMatched, vnMatches12,
                                               windowSize, mfNNratio, mbCheckOrientation);
}

int ORBmatcher::SearchForTriangulation(std::shared_ptr<KeyFrame> pKF1, std::shared_ptr<KeyFrame> pKF2,
                                       std::vector<std::pair<size_t, size_t>>& vMatchedPairs, const bool bOnlyStereo, const bool bCoarse) {
    if (pKF1->mpCamera2 || pKF2->mpCamera2) {   // two-camera KeyFrames: the arms of :1195-1201 / :1294-1330 (ORBmatcher_rig_device.h)
        // one KeyFrame with a second camera and one without: the reference runs on with R12 / t12 never assigned (:1187-1201 vs :1294)
        if (!pKF1->mpCamera2 || !pKF2->mpCamera2) unsupported_rig("ORBmatcher::SearchForTriangulation between a two-camera KeyFrame and a one-camera KeyFrame");
        return msorb_host::SearchForTriangulationRig(pKF1, pKF2, vMatchedPairs, bOnlyStereo, bCoarse, mbCheckOrientation, matcher_device());
    }
    std::vector<std::vector<std::pair<size_t, size_t>>> out;
    const int n = msorb_host::SearchForTriangulationBatch(keyframe_store(), pKF1, std::vector<std::shared_ptr<KeyFrame>>{pKF2}, out,
                                                          bOnlyStereo, bCoarse, mbCheckOrientation)[0];
    vMatchedPairs = std::move(out[0]);
    return n;
}

int ORBmatcher::SearchBySim3(std::shared_ptr<KeyFrame> pKF1, std::shared_ptr<KeyFrame> pKF2,
                             std::vector<std::shared_ptr<MapPoint>>& vpMatches12, const Sophus::Sim3f& S12, const float th) {
    return msorb_host::SearchBySim3(device_keyframe(pKF1, 0), device_keyframe(pKF2, 1), pKF1, pKF2, vpMatches12, S12, th);
}

int ORBmatcher::Fuse(std::shared_ptr<KeyFrame> pKF, const std::vector<std::shared_ptr<MapPoint>>& vpMapPoints, const float th,
                     const bool bRight) {
    if (pKF->GetNLeft() != -1) return msorb_host::FuseRig(device_keyframe_camera(pKF, bRight), pKF, vpMapPoints, th, bRight);   // two cameras
    // the right-camera pass (:1410-1414) reads GetRightPose() / mpCamera2, which a one-camera KeyFrame does not have (the reference
    // dereferences a null camera there): never answer it with the left-camera search
    if (bRight) unsupported_rig("ORBmatcher::Fuse(pKF, vpMapPoints, th, bRight = true) on a KeyFrame without a second camera (NLeft == -1)");
    return msorb_host::Fuse(device_keyframe(pKF), pKF, vpMapPoints, th);
}

int ORBmatcher::Fuse(std::shared_ptr<KeyFrame> pKF, Sophus::Sim3f& Scw, const std::vector<std::shared_ptr<MapPoint>>& vpPoints, float th,
                     std::vector<std::shared_ptr<MapPoint>>& vpReplacePoint) {
    return msorb_host::Fuse(device_keyframe(pKF), pKF, Scw, vpPoints, th, vpReplacePoint);
}

float ORBmatcher::RadiusByViewingCos(const float& viewCos) {  // ORBmatcher.cc:215-221 (applied inside msorb_search_by_projection_mps)
    if (viewCos > 0.998) return 2.5;
    else return 4.0;
}

void ORBmatcher::ComputeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3) {  // :2277-2318
    std::vector<int> sizes(L);
    for (int i = 0; i < L; i++) sizes[i] = (int)histo[i].size();
    int ind[3] = {-1, -1, -1};
    msorb_three_maxima(sizes.data(), L, ind);
    ind1 = ind[0]; ind2 = ind[1]; ind3 = ind[2];
}

// Bit set count operation from http://graphics.stanford.edu/~seander/bithacks.html#CountBitsSetParallel in the reference
// (:2323-2339); a population count of the XOR, 8 x 32 bits.  Host code: MapPoint::ComputeDistinctiveDescriptors and
// Frame::ComputeStereoMatches call it per pair (MapPoint.cc:403, Frame.cc:818).
int ORBmatcher::DescriptorDistance(const cv::Mat& a, const cv::Mat& b) {
    const int32_t* pa = a.ptr<int32_t>(0);
    const int32_t* pb = b.ptr<int32_t>(0);
    int dist = 0;
    for (int i = 0; i < 8; i++, pa++, pb++) dist += __builtin_popcount((unsigned)(*pa ^ *pb));
    return dist;
}

}  // namespace ORB_SLAM3
