// Drop-in replacement for MS-SLAM's include/ORBextractor.h: the same class name, namespace, constructor,
// call operator, getters and public pyramid member that Tracking.cc (:595-601, :1283-1289) and Frame.cc
// (:110-125, :418-425, :750, :840-855) use, implemented on libmsorb.so (HIP kernels for gfx950) through the
// C ABI of include/msorb.h.  Build MS-SLAM with this directory ahead of its own include/ and link -lmsorb;
// no other source file changes.
#ifndef ORBEXTRACTOR_H
#define ORBEXTRACTOR_H

#include <opencv2/opencv.hpp>
#include <vector>

struct msorb_extractor;

namespace ORB_SLAM3 {

namespace msorb_host {
// What happens when the GPU front-end cannot go on (no device, a HIP error, a library / header ABI mismatch).  The reference's
// extractor and matcher cannot fail and Tracking.cc has no handler around them, so the default stays "message on cerr +
// exit(-1)" (MSORB_THROW=1: std::runtime_error).  An application that wants to save its map first registers a handler once,
// e.g. in main() before System is constructed: it is called on the failing thread with the MSORB_E_* code and the message,
// BEFORE the default action, and may itself not return.  Process wide (kept inside libmsorb.so: msorb_set_fatal_callback), so
// it also covers ORBmatcher and the msorb_host:: templates.  nullptr unregisters.
using FatalErrorHandler = void (*)(int code, const char* what, void* user);
void SetFatalErrorHandler(FatalErrorHandler handler, void* user = nullptr);
}  // namespace msorb_host

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    // Same argument meaning as the reference constructor (ORBextractor.cc:409-412).  The HIP device comes from the
    // environment so that call sites stay unchanged: MSORB_DEVICES="0,1" deals the extractor objects of the process onto
    // the listed devices in construction order (left eye -> 0, right eye -> 1 with Tracking.cc:595-596), MSORB_DEVICE=k
    // puts all of them on device k (default 0).
    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
    ~ORBextractor();
    ORBextractor(const ORBextractor&) = delete;
    ORBextractor& operator=(const ORBextractor&) = delete;

    // ORB features of one 8-bit single-channel image; `mask` is ignored like in the reference.  Returns the
    // number of keypoints outside vLappingArea (monoIndex), or -1 for an empty image.
    int operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint>& keypoints,
                   cv::OutputArray descriptors, std::vector<int>& vLappingArea);

    int GetLevels() { return mLevels; }
    float GetScaleFactor() { return mScaleFactor; }
    std::vector<float> GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    // Host views of the device pyramid of the last call (interior pixels).  Valid until the next call.  Filled by one
    // asynchronous copy that overlaps the extraction; left empty with MSORB_HOST_PYRAMID=0 (device-side stereo matching).
    std::vector<cv::Mat> mvImagePyramid;

    // The underlying handle, for msorb_stereo_matches() / msorb_extract_stereo_split() (Frame::ComputeStereoMatches on the
    // device), and the HIP device it lives on.
    msorb_extractor* handle() const { return mHandle; }
    int device() const { return mDevice; }

private:
    msorb_extractor* mHandle;
    int mLevels, mCapacity, mDevice = 0;
    bool mHostPyramid = true;
    float mScaleFactor;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    std::vector<unsigned char> mKpScratch;
};

}  // namespace ORB_SLAM3

#endif
