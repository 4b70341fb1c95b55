// ORB_SLAM3::ORBextractor on libmsorb.so — see ORBextractor.h.  Host glue only: argument marshalling between
// OpenCV containers and the flat C ABI; every computation happens in the HIP kernels.
#include "ORBextractor.h"

#include <atomic>
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <stdexcept>
#include <string>

#include "msorb.h"

namespace ORB_SLAM3 {

static_assert(sizeof(cv::KeyPoint) == sizeof(msorb_keypoint), "cv::KeyPoint must be the 28-byte POD layout");

namespace {
// The reference's constructor and operator() cannot fail, and Tracking.cc has no handler around them.  A GPU that is missing
// or a HIP error is fatal for a front-end without a CPU fallback.  Order of events: (1) the application's callback, if one is
// registered (msorb_set_fatal_callback / msorb_host::SetFatalErrorHandler in ORBextractor.h: save the map, log, exit its own
// way — it need not return); (2) MSORB_THROW=1: std::runtime_error; (3) the default, what the reference does on its own fatal
// conditions (message on cerr, exit(-1), e.g. System.cc:117-120) — so unchanged callers behave as before.
[[noreturn]] void fatal(int code, const std::string& what) {
    msorb_notify_fatal(code, what.c_str());
    if (std::getenv("MSORB_THROW")) throw std::runtime_error(what);
    std::cerr << "msorb (GPU ORB extractor): " << what << std::endl;
    std::exit(-1);
}
// HIP device of the n-th extractor object of the process.  MSORB_DEVICES="0,1" deals the objects round-robin onto the listed
// devices in construction order — Tracking.cc:595-596 builds mpORBextractorLeft first and mpORBextractorRight second, so
// the left eye lands on device 0 and the right eye on device 1 (BASELINE configs[3]); MSORB_DEVICE=k puts every object on
// device k; default 0.
int next_device() {
    static std::atomic<int> n_objects{0};
    const int k = n_objects.fetch_add(1);
    if (const char* list = std::getenv("MSORB_DEVICES")) {
        std::vector<int> devs;
        for (const char* p = list; *p;) {
            char* end = nullptr;
            const long v = std::strtol(p, &end, 10);
            if (end == p) break;
            devs.push_back((int)v);
            p = *end == ',' ? end + 1 : end;
        }
        if (!devs.empty()) return devs[k % devs.size()];
    }
    const char* dev = std::getenv("MSORB_DEVICE");
    return dev ? std::atoi(dev) : 0;
}
}  // namespace

namespace msorb_host {
void SetFatalErrorHandler(FatalErrorHandler handler, void* user) { msorb_set_fatal_callback(handler, user); }
}  // namespace msorb_host

ORBextractor::ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
    : mHandle(nullptr), mLevels(nlevels), mCapacity(0), mScaleFactor(scaleFactor) {
    if (!msorb_abi_compatible(MSORB_ABI_VERSION))
        fatal(MSORB_E_INVALID, "libmsorb.so has ABI " + std::to_string(msorb_abi_version()) + ", this host layer was compiled against " +
                                   std::to_string(MSORB_ABI_VERSION) + " (include/msorb.h): rebuild one of them");
    mDevice = next_device();
    const int rc = msorb_extractor_create(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, mDevice, &mHandle);
    if (rc != MSORB_OK) fatal(rc, std::string("msorb_extractor_create: ") + msorb_last_error());
    // mvImagePyramid is read on the host by an unchanged Frame::ComputeStereoMatches (Frame.cc:750,840-855): the levels
    // come back with one asynchronous copy that overlaps the extraction.  MSORB_HOST_PYRAMID=0 when the stereo
    // association runs on the device (msorb_host::ComputeStereoMatches / ExtractStereo): the vector then stays empty.
    const char* hp = std::getenv("MSORB_HOST_PYRAMID");
    mHostPyramid = !(hp && std::atoi(hp) == 0);
    msorb_extractor_set_host_pyramid(mHandle, mHostPyramid ? 1 : 0);
    mCapacity = msorb_extractor_capacity(mHandle);
    mvScaleFactor.resize(nlevels);
    mvInvScaleFactor.resize(nlevels);
    mvLevelSigma2.resize(nlevels);
    mvInvLevelSigma2.resize(nlevels);
    msorb_extractor_tables(mHandle, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(),
                           mvInvLevelSigma2.data(), nullptr);
    mvImagePyramid.resize(nlevels);
    mKpScratch.resize((size_t)mCapacity * (sizeof(msorb_keypoint) + MSORB_DESC_BYTES));
}

ORBextractor::~ORBextractor() { msorb_extractor_destroy(mHandle); }

int ORBextractor::operator()(cv::InputArray _image, cv::InputArray /*mask*/, std::vector<cv::KeyPoint>& _keypoints,
                             cv::OutputArray _descriptors, std::vector<int>& vLappingArea) {
    if (_image.empty()) return -1;
    cv::Mat image = _image.getMat();
    assert(image.type() == CV_8UC1);
    msorb_keypoint* kps = reinterpret_cast<msorb_keypoint*>(mKpScratch.data());
    unsigned char* desc = mKpScratch.data() + (size_t)mCapacity * sizeof(msorb_keypoint);
    int n = 0, mono = -1;
    const int rc = msorb_extract(mHandle, image.ptr<unsigned char>(0), image.rows, image.cols, (size_t)image.step,
                                 vLappingArea[0], vLappingArea[1], kps, desc, mCapacity, &n, &mono);
    if (rc == MSORB_E_EMPTY) return -1;
    if (rc != MSORB_OK) fatal(rc, std::string("msorb_extract: ") + msorb_last_error());
    if (n == 0) {
        _descriptors.release();
    } else {
        _descriptors.create(n, 32, CV_8U);
        cv::Mat d = _descriptors.getMat();
        for (int i = 0; i < n; i++) std::memcpy(d.ptr<unsigned char>(i), desc + (size_t)i * 32, 32);
    }
    _keypoints = std::vector<cv::KeyPoint>(n);
    if (n) std::memcpy(static_cast<void*>(_keypoints.data()), kps, (size_t)n * sizeof(msorb_keypoint));
    for (int l = 0; mHostPyramid && l < mLevels; l++) {  // mvImagePyramid stays populated for Frame::ComputeStereoMatches
        const unsigned char* p = nullptr;
        int rows = 0, cols = 0;
        size_t stride = 0;
        if (msorb_pyramid_level(mHandle, l, &p, &rows, &cols, &stride) == MSORB_OK)
            mvImagePyramid[l] = cv::Mat(rows, cols, CV_8UC1, const_cast<unsigned char*>(p), stride);
    }
    return mono;
}

}  // namespace ORB_SLAM3
