// ORB_SLAM3::ORBextractor on libmsorb.so — see ORBextractor.h.  Host glue only: argument marshalling between
// OpenCV containers and the flat C ABI; every computation happens in the HIP kernels.
#include "ORBextractor.h"

#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

#include "msorb.h"

namespace ORB_SLAM3 {

static_assert(sizeof(cv::KeyPoint) == sizeof(msorb_keypoint), "cv::KeyPoint must be the 28-byte POD layout");

ORBextractor::ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
    : mHandle(nullptr), mLevels(nlevels), mCapacity(0), mScaleFactor(scaleFactor) {
    const char* dev = std::getenv("MSORB_DEVICE");
    const int rc = msorb_extractor_create(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, dev ? std::atoi(dev) : 0,
                                          &mHandle);
    if (rc != MSORB_OK)  // the reference constructor cannot fail; without a GPU there is nothing to fall back to
        throw std::runtime_error(std::string("msorb_extractor_create: ") + msorb_last_error());
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
    if (rc != MSORB_OK) throw std::runtime_error(std::string("msorb_extract: ") + msorb_last_error());
    if (n == 0) {
        _descriptors.release();
    } else {
        _descriptors.create(n, 32, CV_8U);
        cv::Mat d = _descriptors.getMat();
        for (int i = 0; i < n; i++) std::memcpy(d.ptr<unsigned char>(i), desc + (size_t)i * 32, 32);
    }
    _keypoints = std::vector<cv::KeyPoint>(n);
    if (n) std::memcpy(static_cast<void*>(_keypoints.data()), kps, (size_t)n * sizeof(msorb_keypoint));
    for (int l = 0; l < mLevels; l++) {  // mvImagePyramid stays populated for Frame::ComputeStereoMatches
        const unsigned char* p = nullptr;
        int rows = 0, cols = 0;
        size_t stride = 0;
        if (msorb_pyramid_level(mHandle, l, &p, &rows, &cols, &stride) == MSORB_OK)
            mvImagePyramid[l] = cv::Mat(rows, cols, CV_8UC1, const_cast<unsigned char*>(p), stride);
    }
    return mono;
}

}  // namespace ORB_SLAM3
