// Compiles OUR drop-in ORB_SLAM3::ORBextractor (ms-slam_amd/host) against tests/cv_stub and runs it the way
// Frame::ExtractORB does (Frame.cc:418-425).  Reads a raw u8 image, writes keypoints + descriptors.
// usage: dropin_extractor <rows> <cols> <in.raw> <out.bin> <nfeatures>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ORBextractor.h"

int main(int argc, char** argv) {
    if (argc < 6) return 2;
    const int rows = atoi(argv[1]), cols = atoi(argv[2]), nf = atoi(argv[5]);
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
    fclose(o);
    delete ex;
    return 0;
}
