// ORACLE — TEST INFRASTRUCTURE ONLY (see cvprims.h header).  PARITY UNPINNED (see cvprims.h).
//
// CPU restatement of ORB_SLAM3::ORBextractor (fishmarch/MS-SLAM, /root/reference/src/ORBextractor.cc).
// Scalar, single-threaded, written to follow the reference's arithmetic, container order and
// tie-breaks statement by statement; every function cites the lines it restates.
//
// Floating-point conventions that the reference leaves to its compiler (-O3 -march=native,
// /root/reference/CMakeLists.txt:10-13) are made explicit here and compiled with -ffp-contract=off:
//   * the rotated BRIEF tap  cvRound(x*b + y*a), cvRound(x*a - y*b)  (ORBextractor.cc:117-119) is
//     evaluated as fmaf(x, b, y*a) and fmaf(x, a, -(y*b)) — what g++ 11 -O3 -march=native emits on an
//     FMA-capable x86-64 (probed on this image's compiler: tools/probe_brief_tap.cc) — by default; the two other
//     contractions a build could have are Semantics::brief_tap 1 and 2 (cvprims.h; rotated_tap below);
//   * cos/sin are glibc's cosf/sinf (std::cos(float) via `using namespace std`, ORBextractor.cc:66,112).
#include <cstdio>
#include <list>
#include <utility>

#include "cvprims.h"

namespace orc {

static const signed char kPattern[256 * 4] = {
#include "../ms-slam_amd/csrc/orb_pattern.inc"
};

static const int PATCH_SIZE = 31;       // ORBextractor.cc:71
static const int HALF_PATCH_SIZE = 15;  // :72
static const int EDGE_THRESHOLD = 19;   // :73

struct KeyPoint {  // cv::KeyPoint layout, 28 bytes
    float x, y, size, angle, response;
    int octave, class_id;
};

struct Cand {  // one FAST keypoint handed to the quadtree: coordinates relative to (minBorderX,minBorderY)
    float x, y, response;
};

// ---------------------------------------------------------------------------------------------
// ExtractorNode / DistributeOctTree   (ORBextractor.cc:480-536, 538-553, 555-779)
// ---------------------------------------------------------------------------------------------
struct Node {
    std::vector<Cand> keys;
    int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
    std::list<Node>::iterator lit;
    bool no_more = false;

    void divide(Node& n1, Node& n2, Node& n3, Node& n4) const {  // :480-536
        const int halfX = (int)std::ceil(static_cast<float>(URx - ULx) / 2);
        const int halfY = (int)std::ceil(static_cast<float>(BRy - ULy) / 2);
        n1.ULx = ULx; n1.ULy = ULy;
        n1.URx = ULx + halfX; n1.URy = ULy;
        n1.BLx = ULx; n1.BLy = ULy + halfY;
        n1.BRx = ULx + halfX; n1.BRy = ULy + halfY;
        n2.ULx = n1.URx; n2.ULy = n1.URy;
        n2.URx = URx; n2.URy = URy;
        n2.BLx = n1.BRx; n2.BLy = n1.BRy;
        n2.BRx = URx; n2.BRy = ULy + halfY;
        n3.ULx = n1.BLx; n3.ULy = n1.BLy;
        n3.URx = n1.BRx; n3.URy = n1.BRy;
        n3.BLx = BLx; n3.BLy = BLy;
        n3.BRx = n1.BRx; n3.BRy = BLy;
        n4.ULx = n3.URx; n4.ULy = n3.URy;
        n4.URx = n2.BRx; n4.URy = n2.BRy;
        n4.BLx = n3.BRx; n4.BLy = n3.BRy;
        n4.BRx = BRx; n4.BRy = BRy;
        for (const Cand& kp : keys) {
            if (kp.x < n1.URx) {
                if (kp.y < n1.BRy) n1.keys.push_back(kp); else n3.keys.push_back(kp);
            } else if (kp.y < n1.BRy) n2.keys.push_back(kp);
            else n4.keys.push_back(kp);
        }
        if (n1.keys.size() == 1) n1.no_more = true;
        if (n2.keys.size() == 1) n2.no_more = true;
        if (n3.keys.size() == 1) n3.no_more = true;
        if (n4.keys.size() == 1) n4.no_more = true;
    }
};

typedef std::pair<int, Node*> SizedNode;
static bool compare_nodes(SizedNode& e1, SizedNode& e2) {  // :538-553
    if (e1.first < e2.first) return true;
    if (e1.first > e2.first) return false;
    return e1.second->ULx < e2.second->ULx;
}

static void push_children(std::list<Node>& nodes, Node* ch[4], std::vector<SizedNode>& sized, int* n_to_expand) {
    for (int c = 0; c < 4; c++) {  // :640-675 / :705-740: n1..n4 in order, each pushed to the FRONT
        if (ch[c]->keys.size() > 0) {
            nodes.push_front(*ch[c]);
            if (ch[c]->keys.size() > 1) {
                if (n_to_expand) (*n_to_expand)++;
                sized.push_back(std::make_pair((int)ch[c]->keys.size(), &nodes.front()));
                nodes.front().lit = nodes.begin();
            }
        }
    }
}

static std::vector<Cand> distribute_quadtree(const std::vector<Cand>& to_distribute, int minX, int maxX, int minY,
                                             int maxY, int N) {
    const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));  // :559
    const float hX = static_cast<float>(maxX - minX) / nIni;                            // :561
    std::list<Node> nodes;
    std::vector<Node*> ini(nIni);
    for (int i = 0; i < nIni; i++) {  // :568-579
        Node ni;
        ni.ULx = (int)(hX * static_cast<float>(i)); ni.ULy = 0;
        ni.URx = (int)(hX * static_cast<float>(i + 1)); ni.URy = 0;
        ni.BLx = ni.ULx; ni.BLy = maxY - minY;
        ni.BRx = ni.URx; ni.BRy = maxY - minY;
        nodes.push_back(ni);
        ini[i] = &nodes.back();
    }
    for (const Cand& kp : to_distribute) ini[(size_t)(kp.x / hX)]->keys.push_back(kp);  // :582-586

    for (auto lit = nodes.begin(); lit != nodes.end();) {  // :588-601
        if (lit->keys.size() == 1) { lit->no_more = true; ++lit; }
        else if (lit->keys.empty()) lit = nodes.erase(lit);
        else ++lit;
    }

    bool finish = false;
    std::vector<SizedNode> sized;
    while (!finish) {  // :610-755
        int prev_size = (int)nodes.size();
        int n_to_expand = 0;
        sized.clear();
        for (auto lit = nodes.begin(); lit != nodes.end();) {
            if (lit->no_more) { ++lit; continue; }
            Node n1, n2, n3, n4;
            lit->divide(n1, n2, n3, n4);
            Node* ch[4] = {&n1, &n2, &n3, &n4};
            push_children(nodes, ch, sized, &n_to_expand);
            lit = nodes.erase(lit);
        }
        if ((int)nodes.size() >= N || (int)nodes.size() == prev_size) {
            finish = true;
        } else if (((int)nodes.size() + n_to_expand * 3) > N) {
            while (!finish) {  // :689-753
                prev_size = (int)nodes.size();
                std::vector<SizedNode> prev = sized;
                sized.clear();
                std::sort(prev.begin(), prev.end(), compare_nodes);
                for (int j = (int)prev.size() - 1; j >= 0; j--) {
                    Node n1, n2, n3, n4;
                    prev[j].second->divide(n1, n2, n3, n4);
                    Node* ch[4] = {&n1, &n2, &n3, &n4};
                    push_children(nodes, ch, sized, nullptr);
                    nodes.erase(prev[j].second->lit);
                    if ((int)nodes.size() >= N) break;
                }
                if ((int)nodes.size() >= N || (int)nodes.size() == prev_size) finish = true;
            }
        }
    }

    std::vector<Cand> result;  // :757-776 — first strictly-greater response wins
    for (auto lit = nodes.begin(); lit != nodes.end(); ++lit) {
        const std::vector<Cand>& k = lit->keys;
        const Cand* best = &k[0];
        float max_response = best->response;
        for (size_t i = 1; i < k.size(); i++)
            if (k[i].response > max_response) { best = &k[i]; max_response = k[i].response; }
        result.push_back(*best);
    }
    return result;
}

// ---------------------------------------------------------------------------------------------
// IC_Angle (:76-103) and computeOrbDescriptor (:106-146)
// ---------------------------------------------------------------------------------------------
static float ic_angle(const Plane& image, float ptx, float pty, const std::vector<int>& u_max) {
    int m_01 = 0, m_10 = 0;
    const int step = image.cols;
    const uint8_t* center = image.row(cv_round(pty)) + cv_round(ptx);
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
        int v_sum = 0;
        const int d = u_max[v];
        for (int u = -d; u <= d; ++u) {
            const int val_plus = center[u + v * step], val_minus = center[u - v * step];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return fast_atan2((float)m_01, (float)m_10);
}

static const float kFactorPI = (float)(3.1415926535897932384626433832795 / 180.f);  // :106

// (row, column) offset of one rotated pattern point: cvRound(x*b + y*a), cvRound(x*a - y*b) (ORBextractor.cc:117-119) under the
// contraction Semantics::brief_tap names.  y * (-b) is -(y*b) bit for bit, so "fma(-y, b, .)" needs no form of its own.
static inline void rotated_tap(int mode, float x, float y, float a, float b, int* row, int* col) {
    float r, c;
    if (mode == 1) { r = fmaf(y, a, x * b); c = fmaf(-y, b, x * a); }
    else if (mode == 2) { r = x * b + y * a; c = x * a - y * b; }
    else { r = fmaf(x, b, y * a); c = fmaf(x, a, -(y * b)); }
    *row = cv_round(r);
    *col = cv_round(c);
}

static void orb_descriptor(const KeyPoint& kpt, const Plane& img, uint8_t* desc) {
    const float angle = (float)kpt.angle * kFactorPI;
    const float a = cosf(angle), b = sinf(angle);
    const uint8_t* center = img.row(cv_round(kpt.y)) + cv_round(kpt.x);
    const int step = img.cols;
    const signed char* pat = kPattern;
    const int mode = semantics().brief_tap;
    for (int i = 0; i < 32; ++i) {
        int val = 0;
        for (int k = 0; k < 8; ++k, pat += 4) {
            int r0, c0, r1, c1;
            rotated_tap(mode, (float)pat[0], (float)pat[1], a, b, &r0, &c0);
            rotated_tap(mode, (float)pat[2], (float)pat[3], a, b, &r1, &c1);
            const int t0 = center[r0 * step + c0];
            const int t1 = center[r1 * step + c1];
            val |= (t0 < t1) << k;
        }
        desc[i] = (uint8_t)val;
    }
}

// ---------------------------------------------------------------------------------------------
// ORBextractor  (ctor :409-469, ComputePyramid :1170-1195, ComputeKeyPointsOctTree :781-896,
//                operator() :1086-1168)
// ---------------------------------------------------------------------------------------------
struct Extractor {
    int nfeatures, nlevels, iniThFAST, minThFAST;
    double scaleFactor;  // the reference's member is a double (include/ORBextractor.h:94)
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    std::vector<int> mnFeaturesPerLevel, umax;
    std::vector<Plane> pyramid, blurred;
    std::vector<std::vector<Cand>> candidates;     // vToDistributeKeys per level (last call)
    std::vector<int> cells_total, cells_retried, cells_empty;   // per level (last call): cells visited, cells that took the minThFAST retry (:843-847), cells empty after it
    std::vector<std::vector<KeyPoint>> selected;   // allKeypoints per level (last call; level coords, angle set)

    Extractor(int nf, float sf, int nl, int ini, int mn)
        : nfeatures(nf), nlevels(nl), iniThFAST(ini), minThFAST(mn), scaleFactor(sf) {
        mvScaleFactor.resize(nlevels);
        mvLevelSigma2.resize(nlevels);
        mvScaleFactor[0] = 1.0f;
        mvLevelSigma2[0] = 1.0f;
        for (int i = 1; i < nlevels; i++) {
            mvScaleFactor[i] = mvScaleFactor[i - 1] * scaleFactor;  // float*double -> double -> float
            mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i];
        }
        mvInvScaleFactor.resize(nlevels);
        mvInvLevelSigma2.resize(nlevels);
        for (int i = 0; i < nlevels; i++) {
            mvInvScaleFactor[i] = 1.0f / mvScaleFactor[i];
            mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i];
        }
        mnFeaturesPerLevel.resize(nlevels);
        float factor = 1.0f / scaleFactor;  // double divide, rounded to float
        float nDesired = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
        int sum = 0;
        for (int level = 0; level < nlevels - 1; level++) {
            mnFeaturesPerLevel[level] = cv_round(nDesired);
            sum += mnFeaturesPerLevel[level];
            nDesired *= factor;
        }
        mnFeaturesPerLevel[nlevels - 1] = std::max(nfeatures - sum, 0);

        umax.resize(HALF_PATCH_SIZE + 1);  // :453-468
        int v, v0, vmax = cv_floor(HALF_PATCH_SIZE * std::sqrt(2.f) / 2 + 1);
        int vmin = cv_ceil(HALF_PATCH_SIZE * std::sqrt(2.f) / 2);
        const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
        for (v = 0; v <= vmax; ++v) umax[v] = cv_round(std::sqrt(hp2 - v * v));
        for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
            while (umax[v0] == umax[v0 + 1]) ++v0;
            umax[v] = v0;
            ++v0;
        }
    }

    void compute_pyramid(const uint8_t* img, int rows, int cols, int stride) {  // :1170-1195
        pyramid.assign(nlevels, Plane());
        for (int level = 0; level < nlevels; ++level) {
            const float scale = mvInvScaleFactor[level];
            const int w = cv_round((float)cols * scale), h = cv_round((float)rows * scale);
            pyramid[level] = Plane(h, w);
            if (level != 0) {
                resize_linear_u8(pyramid[level - 1], pyramid[level]);
            } else {
                for (int y = 0; y < rows; y++) memcpy(pyramid[0].row(y), img + (size_t)y * stride, cols);
            }
            // The 19-px BORDER_REFLECT_101 frame (:1185-1191) is never read by the extractor itself
            // (SURVEY.md A.2); the oracle keeps interior planes only.
        }
    }

    // Returns false when the level is too small for the reference's cell arithmetic (division by zero there).
    bool compute_keypoints_quadtree() {  // :781-896
        candidates.assign(nlevels, std::vector<Cand>());
        selected.assign(nlevels, std::vector<KeyPoint>());
        cells_total.assign(nlevels, 0);
        cells_retried.assign(nlevels, 0);
        cells_empty.assign(nlevels, 0);
        const float W = 35;
        std::vector<FastPt> cell;
        for (int level = 0; level < nlevels; ++level) {
            const Plane& im = pyramid[level];
            const int minBorderX = EDGE_THRESHOLD - 3;
            const int minBorderY = minBorderX;
            const int maxBorderX = im.cols - EDGE_THRESHOLD + 3;
            const int maxBorderY = im.rows - EDGE_THRESHOLD + 3;
            std::vector<Cand>& to_distribute = candidates[level];
            const float width = (maxBorderX - minBorderX);
            const float height = (maxBorderY - minBorderY);
            const int nCols = width / W;
            const int nRows = height / W;
            if (nCols < 1 || nRows < 1) return false;
            const int wCell = std::ceil(width / nCols);
            const int hCell = std::ceil(height / nRows);
            for (int i = 0; i < nRows; i++) {
                const float iniY = minBorderY + i * hCell;
                float maxY = iniY + hCell + 6;
                if (iniY >= maxBorderY - 3) continue;
                if (maxY > maxBorderY) maxY = maxBorderY;
                for (int j = 0; j < nCols; j++) {
                    const float iniX = minBorderX + j * wCell;
                    float maxX = iniX + wCell + 6;
                    if (iniX >= maxBorderX - 6) continue;
                    if (maxX > maxBorderX) maxX = maxBorderX;
                    const uint8_t* roi = im.row((int)iniY) + (int)iniX;
                    const int rrows = (int)maxY - (int)iniY, rcols = (int)maxX - (int)iniX;
                    fast9_nms(roi, im.cols, rrows, rcols, iniThFAST, cell);
                    cells_total[level]++;
                    if (cell.empty()) {   // the threshold fallback of :843-847
                        cells_retried[level]++;
                        fast9_nms(roi, im.cols, rrows, rcols, minThFAST, cell);
                        if (cell.empty()) cells_empty[level]++;
                    }
                    for (const FastPt& p : cell)
                        to_distribute.push_back({(float)p.x + j * wCell, (float)p.y + i * hCell, (float)p.score});
                }
            }
            if ((int)std::round(static_cast<float>(maxBorderX - minBorderX) / (maxBorderY - minBorderY)) < 1)
                return false;  // nIni == 0 divides by zero in the reference (:559-561)
            std::vector<Cand> kept;
            if (!to_distribute.empty())
                kept = distribute_quadtree(to_distribute, minBorderX, maxBorderX, minBorderY, maxBorderY,
                                           mnFeaturesPerLevel[level]);
            const int scaledPatchSize = PATCH_SIZE * mvScaleFactor[level];
            for (const Cand& c : kept) {
                KeyPoint kp;
                kp.x = c.x + minBorderX;
                kp.y = c.y + minBorderY;
                kp.size = scaledPatchSize;
                kp.angle = -1;
                kp.response = c.response;
                kp.octave = level;
                kp.class_id = -1;
                selected[level].push_back(kp);
            }
        }
        for (int level = 0; level < nlevels; ++level)  // :894-895
            for (KeyPoint& kp : selected[level]) kp.angle = ic_angle(pyramid[level], kp.x, kp.y, umax);
        return true;
    }

    // operator() :1086-1168.  Returns monoIndex, -1 on empty input, -2 on unsupported geometry.
    int extract(const uint8_t* img, int rows, int cols, int stride, int lap0, int lap1, KeyPoint* out_kps,
                uint8_t* out_desc, int cap, int* n_out) {
        *n_out = 0;
        if (!img || rows <= 0 || cols <= 0) return -1;
        compute_pyramid(img, rows, cols, stride);
        if (!compute_keypoints_quadtree()) return -2;
        int nkeypoints = 0;
        for (int level = 0; level < nlevels; ++level) nkeypoints += (int)selected[level].size();
        if (nkeypoints > cap) return -3;
        *n_out = nkeypoints;
        blurred.assign(nlevels, Plane());
        int monoIndex = 0, stereoIndex = nkeypoints - 1;
        for (int level = 0; level < nlevels; ++level) {
            std::vector<KeyPoint>& kps = selected[level];
            if (kps.empty()) continue;
            gaussian7_q88(pyramid[level], blurred[level]);
            const float scale = mvScaleFactor[level];
            for (const KeyPoint& k0 : kps) {
                uint8_t d[32];
                orb_descriptor(k0, blurred[level], d);
                KeyPoint kp = k0;
                if (level != 0) { kp.x *= scale; kp.y *= scale; }
                int dst;
                if (kp.x >= lap0 && kp.x <= lap1) dst = stereoIndex--;
                else dst = monoIndex++;
                out_kps[dst] = kp;
                memcpy(out_desc + (size_t)dst * 32, d, 32);
            }
        }
        return monoIndex;
    }
};

}  // namespace orc

// -------------------------------------------------------------------------------------------------
// flat C surface for ctypes (tests / bench cpu_baseline)
// -------------------------------------------------------------------------------------------------
using orc::Extractor;
extern "C" {

void* orc_extractor_create(int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh) {
    return new Extractor(nfeatures, scaleFactor, nlevels, iniTh, minTh);
}
void orc_extractor_destroy(void* h) { delete (Extractor*)h; }

int orc_extract(void* h, const uint8_t* img, int rows, int cols, int stride, int lap0, int lap1, void* kps,
                uint8_t* desc, int cap, int* n_out) {
    return ((Extractor*)h)->extract(img, rows, cols, stride, lap0, lap1, (orc::KeyPoint*)kps, desc, cap, n_out);
}
void orc_tables(void* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2, int* per_level, int* umax) {
    Extractor* e = (Extractor*)h;
    for (int i = 0; i < e->nlevels; i++) {
        scale[i] = e->mvScaleFactor[i]; inv_scale[i] = e->mvInvScaleFactor[i];
        sigma2[i] = e->mvLevelSigma2[i]; inv_sigma2[i] = e->mvInvLevelSigma2[i];
        per_level[i] = e->mnFeaturesPerLevel[i];
    }
    for (int i = 0; i < 16; i++) umax[i] = e->umax[i];
}
int orc_level_size(void* h, int level, int* rows, int* cols) {
    Extractor* e = (Extractor*)h;
    if (level < 0 || level >= (int)e->pyramid.size()) return -1;
    *rows = e->pyramid[level].rows; *cols = e->pyramid[level].cols;
    return 0;
}
int orc_level_copy(void* h, int level, int blurred, uint8_t* dst) {
    Extractor* e = (Extractor*)h;
    const std::vector<orc::Plane>& v = blurred ? e->blurred : e->pyramid;
    if (level < 0 || level >= (int)v.size() || v[level].px.empty()) return -1;
    memcpy(dst, v[level].px.data(), v[level].px.size());
    return 0;
}
// vToDistributeKeys of the last call, reference order; coordinates relative to (16,16).
int orc_candidates(void* h, int level, int* xs, int* ys, int* scores, int cap) {
    Extractor* e = (Extractor*)h;
    const auto& c = e->candidates[level];
    for (size_t i = 0; i < c.size() && (int)i < cap; i++) { xs[i] = (int)c[i].x; ys[i] = (int)c[i].y; scores[i] = (int)c[i].response; }
    return (int)c.size();
}
// Input statistics of the last call (bench.py's density sweep): per level the cells the loop visited, those whose first
// cv::FAST at iniThFAST found nothing (the minThFAST retry, :843-847) and those still empty after it.
int orc_cell_stats(void* h, int level, int* total, int* retried, int* empty) {
    Extractor* e = (Extractor*)h;
    if (level < 0 || level >= (int)e->cells_total.size()) return -1;
    *total = e->cells_total[level]; *retried = e->cells_retried[level]; *empty = e->cells_empty[level];
    return 0;
}
// allKeypoints[level] of the last call (level coordinates, angle set), quadtree order.
int orc_selected(void* h, int level, void* kps, int cap) {
    Extractor* e = (Extractor*)h;
    const auto& s = e->selected[level];
    for (size_t i = 0; i < s.size() && (int)i < cap; i++) ((orc::KeyPoint*)kps)[i] = s[i];
    return (int)s.size();
}

// stand-alone primitives
void orc_resize_linear_u8(const uint8_t* src, int srows, int scols, uint8_t* dst, int drows, int dcols) {
    orc::Plane s(srows, scols), d(drows, dcols);
    memcpy(s.px.data(), src, s.px.size());
    orc::resize_linear_u8(s, d);
    memcpy(dst, d.px.data(), d.px.size());
}
void orc_gaussian7(const uint8_t* src, int rows, int cols, uint8_t* dst) {
    orc::Plane s(rows, cols), d;
    memcpy(s.px.data(), src, s.px.size());
    orc::gaussian7_q88(s, d);
    memcpy(dst, d.px.data(), d.px.size());
}
int orc_fast9_nms(const uint8_t* img, int stride, int rows, int cols, int threshold, int* xs, int* ys, int* scores,
                  int cap) {
    std::vector<orc::FastPt> out;
    orc::fast9_nms(img, stride, rows, cols, threshold, out);
    for (size_t i = 0; i < out.size() && (int)i < cap; i++) { xs[i] = out[i].x; ys[i] = out[i].y; scores[i] = out[i].score; }
    return (int)out.size();
}
// The segment test and cornerScore of every pixel of the ROI's detection area, before non-maximum suppression: corner[y][x] =
// fast_is_corner at `threshold`, score[y][x] = cornerScore (0 where not a corner).  For the independent cross-checks of
// tests/test_oracle_pins.py (scikit-image's corner_fast has no NMS and another response definition).
void orc_fast9_planes(const uint8_t* img, int stride, int rows, int cols, int threshold, uint8_t* corner, int* score) {
    memset(corner, 0, (size_t)rows * cols);
    for (size_t i = 0; i < (size_t)rows * cols; i++) score[i] = 0;
    if (rows < 7 || cols < 7) return;
    threshold = std::min(std::max(threshold, 0), 255);
    for (int y = 3; y < rows - 3; y++)
        for (int x = 3; x < cols - 3; x++) {
            const uint8_t* p = img + (size_t)y * stride + x;
            if (orc::fast_is_corner(p, stride, threshold)) {
                corner[(size_t)y * cols + x] = 1;
                score[(size_t)y * cols + x] = orc::fast_corner_score(p, stride, threshold);
            }
        }
}
float orc_fast_atan2(float y, float x) { return orc::fast_atan2(y, x); }
void orc_fast_atan2_n(const float* y, const float* x, long long n, float* out) { for (long long i = 0; i < n; i++) out[i] = orc::fast_atan2(y[i], x[i]); }

// The semantics table of cvprims.h (process-wide in the oracle: test infrastructure).  taps == NULL restores the defaults.
// Returns 0, or -1 for taps the Q8.8 pipeline cannot hold (horizontal sums are 16 bit: 255 * sum(taps) must stay <= 65535).
int orc_set_semantics(const int* gauss_taps, int resize_single_stage, int atan2_fma, int brief_tap) {
    orc::Semantics s;
    if (gauss_taps) {
        int sum = 0;
        for (int i = 0; i < 7; i++) { if (gauss_taps[i] < 0 || gauss_taps[i] > 255) return -1; sum += gauss_taps[i]; s.gauss_taps[i] = gauss_taps[i]; }
        if (sum > 257 || sum < 1) return -1;
        if (brief_tap < 0 || brief_tap > 2) return -1;
        s.resize_single_stage = resize_single_stage != 0;
        s.atan2_fma = atan2_fma != 0;
        s.brief_tap = brief_tap;
    }
    orc::semantics() = s;
    return 0;
}
// One rotated pattern point under one contraction (Semantics::brief_tap numbering): tests/test_semantics_variants.py.
void orc_rotated_tap(int mode, int x, int y, float a, float b, int* row, int* col) { orc::rotated_tap(mode, (float)x, (float)y, a, b, row, col); }
// The exposure of the descriptor to that compiler choice: over n angles (degrees, as fastAtan2 returns them) and the 512 pattern
// points, how many (point, angle) pairs land on a different pixel under contraction m than under contraction 0 — flips[m] for
// m = 1, 2, flips[0] = pairs where 1 and 2 differ from each other; flipped_angles[m] = angles with at least one such point.
// first_* receive up to cap examples (angle bits, point index, mode) for the probe's discriminating set.
long long orc_brief_tap_sweep(const float* angles_deg, long long n, long long* flips, long long* flipped_angles, unsigned* ex_angle_bits,
                              int* ex_point, int* ex_mode, int cap, int* n_ex) {
    long long f[3] = {0, 0, 0}, fa[3] = {0, 0, 0};
    int ne = 0;
    for (long long i = 0; i < n; i++) {
        const float r = angles_deg[i] * orc::kFactorPI;
        const float a = cosf(r), b = sinf(r);
        bool any[3] = {false, false, false};
        for (int p = 0; p < 512; p++) {
            const float x = (float)orc::kPattern[2 * p], y = (float)orc::kPattern[2 * p + 1];
            int r0, c0, r1, c1, r2, c2;
            orc::rotated_tap(0, x, y, a, b, &r0, &c0);
            orc::rotated_tap(1, x, y, a, b, &r1, &c1);
            orc::rotated_tap(2, x, y, a, b, &r2, &c2);
            const bool d1 = r1 != r0 || c1 != c0, d2 = r2 != r0 || c2 != c0, d12 = r1 != r2 || c1 != c2;
            if (d1) { f[1]++; any[1] = true; }
            if (d2) { f[2]++; any[2] = true; }
            if (d12) { f[0]++; any[0] = true; }
            if ((d1 || d2) && ne < cap) {
                unsigned bits; memcpy(&bits, &angles_deg[i], 4);
                ex_angle_bits[ne] = bits; ex_point[ne] = p; ex_mode[ne] = (d1 ? 1 : 0) | (d2 ? 2 : 0); ne++;
            }
        }
        for (int m = 0; m < 3; m++) fa[m] += any[m];
    }
    for (int m = 0; m < 3; m++) { flips[m] = f[m]; flipped_angles[m] = fa[m]; }
    *n_ex = ne;
    return n * 512;
}
void orc_cos_sin(float angle_deg, float* a, float* b) {
    const float r = angle_deg * orc::kFactorPI;
    *a = cosf(r); *b = sinf(r);
}
// libstdc++'s std::sort itself on (key, input position) items compared by key alone — the shape of ORBextractor.cc:700's
// std::sort(vPrevSizeAndPointerToNode, compareNodes): unstable, the order of equal keys is the algorithm's.  order[i] = input
// position of the item that ends at position i.  (tests/test_quadtree_sort_gpu.py: the device's restatement against this.)
void orc_std_sort_order(const unsigned* keys, int n, unsigned* order) {
    std::vector<std::pair<unsigned, unsigned>> v(n);
    for (int i = 0; i < n; i++) v[i] = {keys[i], (unsigned)i};
    std::sort(v.begin(), v.end(), [](const std::pair<unsigned, unsigned>& a, const std::pair<unsigned, unsigned>& b) { return a.first < b.first; });
    for (int i = 0; i < n; i++) order[i] = v[i].second;
}
// quadtree alone: candidates (x,y,response) relative to min border -> kept candidates in list order
int orc_distribute_quadtree(const float* xs, const float* ys, const float* resp, int n, int minX, int maxX, int minY,
                            int maxY, int N, float* oxs, float* oys, float* oresp, int cap) {
    std::vector<orc::Cand> in(n);
    for (int i = 0; i < n; i++) in[i] = {xs[i], ys[i], resp[i]};
    std::vector<orc::Cand> out;
    if (n) out = orc::distribute_quadtree(in, minX, maxX, minY, maxY, N);
    for (size_t i = 0; i < out.size() && (int)i < cap; i++) { oxs[i] = out[i].x; oys[i] = out[i].y; oresp[i] = out[i].response; }
    return (int)out.size();
}
}  // extern "C"
