// Compiles OUR drop-in ORB_SLAM3::ORBextractor (ms-slam_amd/host) against tests/cv_stub and runs it the way
// Frame::ExtractORB does (Frame.cc:418-425).  Reads a raw u8 image, writes keypoints + descriptors.
// usage: dropin_extractor <rows> <cols> <in.raw> <out.bin> <nfeatures>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "ORBextractor.h"
#include "ORBmatcher_device.h"

namespace ORB_SLAM3 {
struct Frame {  // the members the stereo constructor fills (include/Frame.h)
    std::vector<cv::KeyPoint> mvKeys, mvKeysRight;
    cv::Mat mDescriptors, mDescriptorsRight;
    std::vector<float> mvuRight, mvDepth;
    float mb = 0, mbf = 0;
    float mnMinX = 0, mnMaxX = 0, mnMinY = 0, mnMaxY = 0;
};
}  // namespace ORB_SLAM3

// mode "stereo": <rows> <cols> <left.raw> <out.bin> <nfeatures> stereo <right.raw> <mb> <mbf> — the stereo Frame constructor
// once through msorb_host::ExtractStereo and once the reference's way (two operator() calls + ComputeStereoMatches)
static int stereo_main(int rows, int cols, const char* lp, const char* rp, const char* out, int nf, float mb, float mbf) {
    using namespace ORB_SLAM3;
    std::vector<unsigned char> bl((size_t)rows * cols), br((size_t)rows * cols);
    FILE* f = fopen(lp, "rb");
    if (!f || fread(bl.data(), 1, bl.size(), f) != bl.size()) return 3;
    fclose(f);
    f = fopen(rp, "rb");
    if (!f || fread(br.data(), 1, br.size(), f) != br.size()) return 3;
    fclose(f);
    cv::Mat imL(rows, cols, CV_8UC1, bl.data(), (size_t)cols), imR(rows, cols, CV_8UC1, br.data(), (size_t)cols);
    ORBextractor exL(nf, 1.2f, 8, 20, 7), exR(nf, 1.2f, 8, 20, 7), exF(nf, 1.2f, 8, 20, 7);
    Frame A, B;
    A.mb = B.mb = mb; A.mbf = B.mbf = mbf;
    msorb_host::ExtractStereo(A, exF, imL, imR);
    std::vector<int> lap = {0, 0};
    exL(imL, cv::Mat(), B.mvKeys, B.mDescriptors, lap);
    exR(imR, cv::Mat(), B.mvKeysRight, B.mDescriptorsRight, lap);
    msorb_host::ComputeStereoMatches(B, exL, exR);
    // and with one extractor object per device (MSORB_DEVICES deals exS0 / exS1 onto the listed devices), twice
    ORBextractor exS0(nf, 1.2f, 8, 20, 7), exS1(nf, 1.2f, 8, 20, 7);
    Frame C, D;
    C.mb = D.mb = mb; C.mbf = D.mbf = mbf;
    msorb_host::ExtractStereoSplit(C, exS0, exS1, imL, imR);
    msorb_host::ExtractStereoSplit(D, exS0, exS1, imL, imR);
    // ... and the constructor up to AssignFeaturesToGrid with the frame left on the device (msorb_host::ExtractStereoFrame)
    Frame E;
    E.mb = mb; E.mbf = mbf;
    E.mnMinX = 0; E.mnMaxX = (float)cols; E.mnMinY = 0; E.mnMaxY = (float)rows;
    msorb_host::DeviceFrame<Frame> dev;
    msorb_host::ExtractStereoFrame(dev, E, exF, imL, imR);
    std::vector<int> cell_begin(64 * 48 + 1), cell_idx(E.mvKeys.size() + 1);
    int n_assigned = 0;
    if (msorb_frame_grid(dev.get(), cell_begin.data(), cell_idx.data(), (int)cell_idx.size(), &n_assigned)) return 4;
    FILE* o = fopen(out, "wb");
    const int devs[2] = {exS0.device(), exS1.device()};
    fwrite(devs, 4, 2, o);
    for (Frame* F : {&A, &B, &C, &D, &E}) {
        const int n = (int)F->mvKeys.size(), nr = (int)F->mvKeysRight.size();
        fwrite(&n, 4, 1, o); fwrite(&nr, 4, 1, o);
        fwrite(F->mvKeys.data(), sizeof(cv::KeyPoint), n, o);
        fwrite(F->mvKeysRight.data(), sizeof(cv::KeyPoint), nr, o);
        for (int i = 0; i < n; i++) fwrite(F->mDescriptors.ptr<unsigned char>(i), 1, 32, o);
        for (int i = 0; i < nr; i++) fwrite(F->mDescriptorsRight.ptr<unsigned char>(i), 1, 32, o);
        fwrite(F->mvuRight.data(), 4, n, o); fwrite(F->mvDepth.data(), 4, n, o);
    }
    fwrite(&n_assigned, 4, 1, o);
    fwrite(cell_begin.data(), 4, cell_begin.size(), o);
    fwrite(cell_idx.data(), 4, n_assigned, o);
    fclose(o);
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 6) return 2;
    const int rows = atoi(argv[1]), cols = atoi(argv[2]), nf = atoi(argv[5]);
    if (argc >= 10 && std::string(argv[6]) == "stereo")
        return stereo_main(rows, cols, argv[3], argv[7], argv[4], nf, (float)atof(argv[8]), (float)atof(argv[9]));
    std::vector<unsigned char> buf((size_t)rows * cols);
    FILE* f = fopen(argv[3], "rb");
    if (!f || fread(buf.data(), 1, buf.size(), f) != buf.size()) return 3;
    fclose(f);
    ORB_SLAM3::ORBextractor* ex = new ORB_SLAM3::ORBextractor(nf, 1.2f, 8, 20, 7);   // Tracking.cc:595
    cv::Mat im(rows, cols, CV_8UC1, buf.data(), (size_t)cols), desc;
    std::vector<cv::KeyPoint> keys;
    std::vector<int> lap = {0, 0};
    const int mono = (*ex)(im, cv::Mat(), keys, desc, lap);                            // Frame.cc:422
    FILE* o = fopen(argv[4], "wb");
    int n = (int)keys.size();
    fwrite(&mono, 4, 1, o);
    fwrite(&n, 4, 1, o);
    fwrite(keys.data(), sizeof(cv::KeyPoint), n, o);
    for (int i = 0; i < n; i++) fwrite(desc.ptr<unsigned char>(i), 1, 32, o);
    int l7r = ex->mvImagePyramid[7].rows, l7c = ex->mvImagePyramid[7].cols, lv = ex->GetLevels();
    fwrite(&l7r, 4, 1, o); fwrite(&l7c, 4, 1, o); fwrite(&lv, 4, 1, o);
    // mvImagePyramid as Frame::ComputeStereoMatches reads it on the host (Frame.cc:840-855): every level, row by row
    for (int l = 0; l < lv; l++) {
        const cv::Mat& m = ex->mvImagePyramid[l];
        fwrite(&m.rows, 4, 1, o); fwrite(&m.cols, 4, 1, o);
        for (int y = 0; y < m.rows; y++) fwrite(m.ptr<unsigned char>(y), 1, m.cols, o);
    }
    fclose(o);
    delete ex;
    return 0;
}
