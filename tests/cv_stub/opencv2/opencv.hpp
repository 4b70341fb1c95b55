// TEST-ONLY minimal stand-in for the handful of OpenCV types that ms-slam_amd/host/ORBextractor.{h,cc} (OUR
// host glue, not the reference) touches, so the drop-in class can be compiled and exercised on machines
// without OpenCV (this image, the GPU box).  It is never used to build reference sources.
#pragma once
#include <cstddef>
#include <cstring>
#include <vector>
#define CV_8U 0
#define CV_8UC1 0
namespace cv {
struct Point2f { float x, y; };
struct KeyPoint { Point2f pt; float size, angle, response; int octave, class_id; };
class Mat {
public:
    int rows = 0, cols = 0;
    size_t step = 0;
    unsigned char* data = nullptr;
    std::vector<unsigned char> own;
    Mat() {}
    Mat(int r, int c, int, void* d, size_t s) : rows(r), cols(c), step(s), data((unsigned char*)d) {}
    void create(int r, int c, int) { rows = r; cols = c; step = (size_t)c; own.assign((size_t)r * c, 0); data = own.data(); }
    void release() { rows = cols = 0; step = 0; own.clear(); data = nullptr; }
    bool empty() const { return rows == 0 || cols == 0 || !data; }
    bool isContinuous() const { return step == (size_t)cols || rows <= 1; }
    int type() const { return CV_8UC1; }
    Mat row(int r) const { return Mat(1, cols, CV_8UC1, data + (size_t)r * step, step); }
    template <typename T> T* ptr(int r) { return (T*)(data + (size_t)r * step); }
    template <typename T> const T* ptr(int r) const { return (const T*)(data + (size_t)r * step); }
};
class _InputArray {
public:
    const Mat* m = nullptr;
    _InputArray() {}
    _InputArray(const Mat& mm) : m(&mm) {}
    bool empty() const { return !m || m->empty(); }
    Mat getMat() const { Mat r; r.rows = m->rows; r.cols = m->cols; r.step = m->step; r.data = m->data; return r; }
};
class _OutputArray {
public:
    Mat* m;
    _OutputArray(Mat& mm) : m(&mm) {}
    void create(int r, int c, int t) const { m->create(r, c, t); }
    void release() const { m->release(); }
    Mat getMat() const { Mat r; r.rows = m->rows; r.cols = m->cols; r.step = m->step; r.data = m->data; return r; }
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
}  // namespace cv
