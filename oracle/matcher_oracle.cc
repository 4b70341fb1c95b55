// ORACLE — TEST INFRASTRUCTURE ONLY (see cvprims.h header).
//
// CPU restatement of the Hamming-matching path of fishmarch/MS-SLAM on flat stand-in arrays (the
// reference's Frame/KeyFrame/MapPoint need OpenCV+Eigen+Sophus and cannot be compiled here, SURVEY.md
// §8c).  Hamming distance is exact integer arithmetic, so the parity content is the candidate SET, the
// candidate ORDER (tie-breaks) and the accept rules; each function cites the lines it follows.
// Only the rectified-stereo branch (Nleft == -1) is restated: every BASELINE config takes it
// (SURVEY.md §3.1); the fisheye branches (ORBmatcher.cc:144-210, 2059-2124) are out of scope.
//
// PARITY UNPINNED for anything float that the reference leaves to Eigen/Sophus (projection of 3-D
// points): those steps stay on the caller's side of the boundary — the functions here start from the
// projected coordinates, as the C ABI does.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <utility>
#include <vector>

namespace orc {

static const int TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30;  // ORBmatcher.cc:35-37
static const int GRID_COLS = 64, GRID_ROWS = 48;                  // Frame.h:44-45

struct KeyPoint { float x, y, size, angle, response; int octave, class_id; };

// ORBmatcher::DescriptorDistance, ORBmatcher.cc:2323-2339 (SWAR popcount over 8 x 32 bit)
static int descriptor_distance(const uint8_t* a, const uint8_t* b) {
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t pa, pb;
        memcpy(&pa, a + 4 * i, 4);
        memcpy(&pb, b + 4 * i, 4);
        unsigned int v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

// Stand-in for the parts of Frame the matcher touches.
struct FrameSoA {
    int N = 0;
    std::vector<KeyPoint> kps;     // mvKeysUn (== mvKeys: rectified, k1 == 0, Frame.cc:681-685)
    std::vector<uint8_t> desc;     // mDescriptors, N x 32
    std::vector<float> uRight;     // mvuRight
    float minX, minY, maxX, maxY, gridWInv, gridHInv;  // mnMinX.. / mfGridElementWidthInv.. (Frame.cc:147-148)
    std::vector<float> scaleFactors;
    std::vector<int> grid[GRID_COLS][GRID_ROWS];

    bool pos_in_grid(const KeyPoint& kp, int& px, int& py) const {  // Frame.cc:657-667
        px = (int)std::round((kp.x - minX) * gridWInv);
        py = (int)std::round((kp.y - minY) * gridHInv);
        return !(px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS);
    }
    void assign_features_to_grid() {  // Frame.cc:385-416
        for (int i = 0; i < N; i++) {
            int gx, gy;
            if (pos_in_grid(kps[i], gx, gy)) grid[gx][gy].push_back(i);
        }
    }
    // Frame::GetFeaturesInArea, Frame.cc:589-655
    std::vector<size_t> features_in_area(float x, float y, float r, int minLevel, int maxLevel) const {
        std::vector<size_t> idx;
        const float factorX = r, factorY = r;
        const int nMinCellX = std::max(0, (int)std::floor((x - minX - factorX) * gridWInv));
        if (nMinCellX >= GRID_COLS) return idx;
        const int nMaxCellX = std::min(GRID_COLS - 1, (int)std::ceil((x - minX + factorX) * gridWInv));
        if (nMaxCellX < 0) return idx;
        const int nMinCellY = std::max(0, (int)std::floor((y - minY - factorY) * gridHInv));
        if (nMinCellY >= GRID_ROWS) return idx;
        const int nMaxCellY = std::min(GRID_ROWS - 1, (int)std::ceil((y - minY + factorY) * gridHInv));
        if (nMaxCellY < 0) return idx;
        const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
        for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
            for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
                const std::vector<int>& cell = grid[ix][iy];
                for (size_t j = 0; j < cell.size(); j++) {
                    const KeyPoint& kp = kps[cell[j]];
                    if (bCheckLevels) {
                        if (kp.octave < minLevel) continue;
                        if (maxLevel >= 0 && kp.octave > maxLevel) continue;
                    }
                    const float distx = kp.x - x, disty = kp.y - y;
                    if (std::fabs(distx) < factorX && std::fabs(disty) < factorY) idx.push_back(cell[j]);
                }
            }
        return idx;
    }
};

// ORBmatcher::ComputeThreeMaxima, ORBmatcher.cc:2277-2318
static void three_maxima(const std::vector<int>* histo, int L, int& ind1, int& ind2, int& ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = (int)histo[i].size();
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

}  // namespace orc

using orc::FrameSoA;
using orc::KeyPoint;

extern "C" {

int orc_descriptor_distance(const uint8_t* a, const uint8_t* b) { return orc::descriptor_distance(a, b); }

void* orc_frame_create(const void* kps, int N, const uint8_t* desc, const float* uRight, float minX, float maxX,
                       float minY, float maxY, const float* scaleFactors, int nlevels) {
    FrameSoA* f = new FrameSoA();
    f->N = N;
    f->kps.assign((const KeyPoint*)kps, (const KeyPoint*)kps + N);
    f->desc.assign(desc, desc + (size_t)N * 32);
    if (uRight) f->uRight.assign(uRight, uRight + N); else f->uRight.assign(N, -1.0f);
    f->minX = minX; f->maxX = maxX; f->minY = minY; f->maxY = maxY;
    f->gridWInv = static_cast<float>(orc::GRID_COLS) / (maxX - minX);
    f->gridHInv = static_cast<float>(orc::GRID_ROWS) / (maxY - minY);
    f->scaleFactors.assign(scaleFactors, scaleFactors + nlevels);
    f->assign_features_to_grid();
    return f;
}
void orc_frame_destroy(void* f) { delete (FrameSoA*)f; }

int orc_features_in_area(void* fp, float x, float y, float r, int minLevel, int maxLevel, int* out, int cap) {
    const std::vector<size_t> v = ((FrameSoA*)fp)->features_in_area(x, y, r, minLevel, maxLevel);
    for (size_t i = 0; i < v.size() && (int)i < cap; i++) out[i] = (int)v[i];
    return (int)v.size();
}

// Grid cell of every keypoint (-1 when outside), for the rank key of SURVEY.md B.1.
void orc_frame_cells(void* fp, int* cx, int* cy) {
    FrameSoA* f = (FrameSoA*)fp;
    for (int i = 0; i < f->N; i++) {
        int gx, gy;
        if (f->pos_in_grid(f->kps[i], gx, gy)) { cx[i] = gx; cy[i] = gy; } else { cx[i] = cy[i] = -1; }
    }
}

// mGrid after Frame::AssignFeaturesToGrid (Frame.cc:385-416) as CSR: cell = ix * FRAME_GRID_ROWS + iy = mGrid[ix][iy], the
// cell's vector in push_back order.  cell_begin has GRID_COLS * GRID_ROWS + 1 entries; returns the number of assigned keypoints.
int orc_frame_grid_csr(void* fp, int* cell_begin, int* cell_idx) {
    FrameSoA* f = (FrameSoA*)fp;
    int n = 0;
    for (int ix = 0; ix < orc::GRID_COLS; ix++)
        for (int iy = 0; iy < orc::GRID_ROWS; iy++) {
            cell_begin[ix * orc::GRID_ROWS + iy] = n;
            for (int i : f->grid[ix][iy]) cell_idx[n++] = i;
        }
    cell_begin[orc::GRID_COLS * orc::GRID_ROWS] = n;
    return n;
}

// ORBmatcher::SearchByProjection(Frame&, vector<MapPoint>, th, bFarPoints, thFarPoints), ORBmatcher.cc:43-142
// (left-eye / rectified part).  Map point table of M entries, visited in index order:
//   track_in_view  mbTrackInView          bad          isBad()              sparsified  mbSparsified
//   proj_x/y/xr    mTrackProjX/Y/XR       track_depth  mTrackDepth          level       mnTrackScaleLevel
//   view_cos       mTrackViewCos          mp_desc      GetDescriptor()      obs         Observations()
// frame_mp[N]: F.mvpMapPoints as map-point ids (-1 = empty); updated in place exactly like the reference
// (a later map point sees earlier assignments).  Returns nmatches.
int orc_search_by_projection_mps(void* fp, int M, const uint8_t* track_in_view, const uint8_t* bad,
                                 const uint8_t* sparsified, const float* proj_x, const float* proj_y,
                                 const float* proj_xr, const float* track_depth, const int* level,
                                 const float* view_cos, const uint8_t* mp_desc, const int* obs, int* frame_mp,
                                 float th, int bFarPoints, float thFarPoints, float nnratio) {
    FrameSoA& F = *(FrameSoA*)fp;
    int nmatches = 0;
    const bool bFactor = th != 1.0;
    for (int iMP = 0; iMP < M; iMP++) {
        if (!track_in_view[iMP]) continue;
        if (bFarPoints && track_depth[iMP] > thFarPoints) continue;
        if (bad[iMP]) continue;
        const int nPredictedLevel = level[iMP];
        float r = (view_cos[iMP] > 0.998) ? 2.5 : 4.0;  // RadiusByViewingCos, :215-221 (float vs double compare)
        if (bFactor) r *= th;
        const std::vector<size_t> vIndices = F.features_in_area(
            proj_x[iMP], proj_y[iMP], r * F.scaleFactors[nPredictedLevel], nPredictedLevel - 1, nPredictedLevel);
        if (vIndices.empty()) continue;
        const uint8_t* MPdescriptor = mp_desc + (size_t)iMP * 32;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (size_t k = 0; k < vIndices.size(); k++) {
            const size_t idx = vIndices[k];
            if (frame_mp[idx] >= 0 && !sparsified[iMP])
                if (obs[frame_mp[idx]] > 0) continue;
            if (F.uRight[idx] > 0) {
                const float er = std::fabs(proj_xr[iMP] - F.uRight[idx]);
                if (er > r * F.scaleFactors[nPredictedLevel]) continue;
            }
            const int dist = orc::descriptor_distance(MPdescriptor, &F.desc[idx * 32]);
            if (dist < bestDist) {
                bestDist2 = bestDist; bestDist = dist;
                bestLevel2 = bestLevel; bestLevel = F.kps[idx].octave;
                bestIdx = (int)idx;
            } else if (dist < bestDist2) {
                bestLevel2 = F.kps[idx].octave;
                bestDist2 = dist;
            }
        }
        if (bestDist <= orc::TH_HIGH) {
            if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
            if (bestLevel != bestLevel2 || bestDist <= nnratio * bestDist2) {
                frame_mp[bestIdx] = iMP;
                nmatches++;
            }
        }
    }
    return nmatches;
}

// The same function on a two-camera frame (F.Nleft != -1), ORBmatcher.cc:43-213 statement by statement with both arms.  fl / fr:
// the left camera's keypoints (F.mvKeys[0, Nleft), descriptor rows [0, Nleft)) and the right camera's (F.mvKeysRight, rows
// [Nleft, N)) as two FrameSoA — what Frame::GetFeaturesInArea(x, y, r, minLevel, maxLevel, bRight) walks (Frame.cc:589-655).
// frame_mp[NL + NR] = F.mvpMapPoints as table ids.  *_r: mbTrackInViewR, mTrackProjXR / YR, mnTrackScaleLevelR, mTrackViewCosR.
int orc_search_by_projection_mps_rig(void* fl, void* fr, int M, const uint8_t* track_in_view, const uint8_t* track_in_view_r,
                                     const uint8_t* bad, const uint8_t* sparsified, const float* proj_x, const float* proj_y,
                                     const float* proj_xr, const float* proj_yr, const float* track_depth, const int* level, const int* level_r,
                                     const float* view_cos, const float* view_cos_r, const uint8_t* mp_desc, const int* obs,
                                     const int* left_to_right, const int* right_to_left, int* frame_mp, float th, int bFarPoints,
                                     float thFarPoints, float nnratio) {
    FrameSoA& FL = *(FrameSoA*)fl;
    FrameSoA& FR = *(FrameSoA*)fr;
    const int Nleft = FL.N;
    int nmatches = 0;
    const bool bFactor = th != 1.0;
    for (int iMP = 0; iMP < M; iMP++) {
        if (!track_in_view[iMP] && !track_in_view_r[iMP]) continue;
        if (bFarPoints && track_depth[iMP] > thFarPoints) continue;
        if (bad[iMP]) continue;
        const uint8_t* MPdescriptor = mp_desc + (size_t)iMP * 32;
        if (track_in_view[iMP]) {
            const int nPredictedLevel = level[iMP];
            float r = (view_cos[iMP] > 0.998) ? 2.5 : 4.0;
            if (bFactor) r *= th;
            const std::vector<size_t> vIndices =
                FL.features_in_area(proj_x[iMP], proj_y[iMP], r * FL.scaleFactors[nPredictedLevel], nPredictedLevel - 1, nPredictedLevel);
            if (!vIndices.empty()) {
                int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
                for (size_t k = 0; k < vIndices.size(); k++) {
                    const size_t idx = vIndices[k];
                    if (frame_mp[idx] >= 0 && !sparsified[iMP])
                        if (obs[frame_mp[idx]] > 0) continue;
                    // (F.Nleft == -1 && F.mvuRight[idx] > 0: not on such a frame, :92)
                    const int dist = orc::descriptor_distance(MPdescriptor, &FL.desc[idx * 32]);
                    if (dist < bestDist) {
                        bestDist2 = bestDist; bestDist = dist;
                        bestLevel2 = bestLevel; bestLevel = FL.kps[idx].octave;   // F.mvKeys[idx].octave (idx < Nleft)
                        bestIdx = (int)idx;
                    } else if (dist < bestDist2) {
                        bestLevel2 = FL.kps[idx].octave;
                        bestDist2 = dist;
                    }
                }
                if (bestDist <= orc::TH_HIGH) {
                    if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;   // (leaves the map-point loop: no right pass)
                    if (bestLevel != bestLevel2 || bestDist <= nnratio * bestDist2) {
                        frame_mp[bestIdx] = iMP;
                        if (left_to_right[bestIdx] != -1) {   // also match with the stereo observation at the right camera
                            frame_mp[left_to_right[bestIdx] + Nleft] = iMP;
                            nmatches++;
                        }
                        nmatches++;
                    }
                }
            }
        }
        if (track_in_view_r[iMP]) {
            const int nPredictedLevel = level_r[iMP];
            if (nPredictedLevel != -1) {
                const float r = (view_cos_r[iMP] > 0.998) ? 2.5 : 4.0;
                const std::vector<size_t> vIndices =
                    FR.features_in_area(proj_xr[iMP], proj_yr[iMP], r * FR.scaleFactors[nPredictedLevel], nPredictedLevel - 1, nPredictedLevel);
                if (vIndices.empty()) continue;
                int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
                for (size_t k = 0; k < vIndices.size(); k++) {
                    const size_t idx = vIndices[k];
                    if (frame_mp[idx + Nleft] >= 0)
                        if (obs[frame_mp[idx + Nleft]] > 0) continue;
                    const int dist = orc::descriptor_distance(MPdescriptor, &FR.desc[idx * 32]);
                    if (dist < bestDist) {
                        bestDist2 = bestDist; bestDist = dist;
                        bestLevel2 = bestLevel; bestLevel = FR.kps[idx].octave;
                        bestIdx = (int)idx;
                    } else if (dist < bestDist2) {
                        bestLevel2 = FR.kps[idx].octave;
                        bestDist2 = dist;
                    }
                }
                if (bestDist <= orc::TH_HIGH) {
                    if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
                    if (right_to_left[bestIdx] != -1) {
                        frame_mp[right_to_left[bestIdx]] = iMP;
                        nmatches++;
                    }
                    frame_mp[bestIdx + Nleft] = iMP;
                    nmatches++;
                }
            }
        }
    }
    return nmatches;
}

// ORBmatcher::SearchByProjection(Frame& Cur, const Frame& Last, th, bMono), ORBmatcher.cc:1941-2057 and
// 2129-2152, from the projected coordinates onward.  One entry per last-frame keypoint i:
//   valid[i]   pMP exists, !mvbOutlier, invzc >= 0, uv inside the image bounds (:1961-1983)
//   u,v        uv = mpCamera->project(Tcw * x3Dw)        ur = uv(0) - mbf*invzc (:2018)
//   last_octave, last_angle   LastFrame.mvKeys(Un)[i]    mp_desc[i] = pMP->GetDescriptor()
//   last_mp[i] map point id written into cur_mp;  obs[id] = Observations() of map point id
// cur_mp[N]: CurrentFrame.mvpMapPoints as ids (-1 empty), updated in place.  Returns nmatches.
int orc_search_by_projection_frames(void* fp, int NL, const uint8_t* valid, const float* u, const float* v,
                                    const float* ur, const int* last_octave, const float* last_angle,
                                    const uint8_t* mp_desc, const int* last_mp, const int* obs, int* cur_mp,
                                    float th, int bForward, int bBackward, int check_orientation) {
    FrameSoA& C = *(FrameSoA*)fp;
    int nmatches = 0;
    std::vector<int> rotHist[orc::HISTO_LENGTH];
    const float factor = 1.0f / orc::HISTO_LENGTH;
    for (int i = 0; i < NL; i++) {
        if (!valid[i]) continue;
        const int nLastOctave = last_octave[i];
        const float radius = th * C.scaleFactors[nLastOctave];
        std::vector<size_t> vIndices2;
        if (bForward) vIndices2 = C.features_in_area(u[i], v[i], radius, nLastOctave, -1);
        else if (bBackward) vIndices2 = C.features_in_area(u[i], v[i], radius, 0, nLastOctave);
        else vIndices2 = C.features_in_area(u[i], v[i], radius, nLastOctave - 1, nLastOctave + 1);
        if (vIndices2.empty()) continue;
        const uint8_t* dMP = mp_desc + (size_t)i * 32;
        int bestDist = 256, bestIdx2 = -1;
        for (size_t k = 0; k < vIndices2.size(); k++) {
            const size_t i2 = vIndices2[k];
            if (cur_mp[i2] >= 0)
                if (obs[cur_mp[i2]] > 0) continue;
            if (C.uRight[i2] > 0) {
                const float er = std::fabs(ur[i] - C.uRight[i2]);
                if (er > radius) continue;
            }
            const int dist = orc::descriptor_distance(dMP, &C.desc[i2 * 32]);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = (int)i2; }
        }
        if (bestDist <= orc::TH_HIGH) {
            cur_mp[bestIdx2] = last_mp[i];
            nmatches++;
            if (check_orientation) {
                float rot = last_angle[i] - C.kps[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == orc::HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (check_orientation) {  // :2129-2149
        int ind1 = -1, ind2 = -1, ind3 = -1;
        orc::three_maxima(rotHist, orc::HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < orc::HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (size_t j = 0; j < rotHist[i].size(); j++) { cur_mp[rotHist[i][j]] = -1; nmatches--; }
    }
    return nmatches;
}

// ORBmatcher::SearchByProjection(Frame& Cur, const Frame& Last, th, bMono) on a two-camera CurrentFrame (Nleft != -1),
// ORBmatcher.cc:1941-2152 with the right-camera arm :2059-2124, from the projected coordinates on.  Per last-frame keypoint i:
// valid (:1962-1983: map point, not an outlier, invzc >= 0, the LEFT projection inside the image), u, v (left camera), u_r, v_r
// (CurrentFrame.mpCamera->project(GetRelativePoseTrl() * x3Dc), :2060-2061), last_octave, last_angle (the Nleft-aware keypoint of
// LastFrame, :1986 / :2043-2045), mp_desc, last_mp, obs.  fl / fr: the current frame's two cameras; cur_mp[NL + NR].
// What the arm inherits from the statement order: the right window is only searched when the left one held a candidate (:2003-2004
// `continue`s on an empty left window); both arms feed ONE rotation histogram (:2054, :2121: right entries as idx + Nleft).
int orc_search_by_projection_frames_rig(void* fl, void* fr, int NLast, const uint8_t* valid, const float* u, const float* v, const float* u_r,
                                        const float* v_r, const int* last_octave, const float* last_angle, const uint8_t* mp_desc,
                                        const int* last_mp, const int* obs, int* cur_mp, float th, int bForward, int bBackward,
                                        int check_orientation) {
    FrameSoA& CL = *(FrameSoA*)fl;
    FrameSoA& CR = *(FrameSoA*)fr;
    const int Nleft = CL.N;
    int nmatches = 0;
    std::vector<int> rotHist[orc::HISTO_LENGTH];
    const float factor = 1.0f / orc::HISTO_LENGTH;
    auto window = [&](FrameSoA& C, float x, float y, float radius, int oct) {
        if (bForward) return C.features_in_area(x, y, radius, oct, -1);
        if (bBackward) return C.features_in_area(x, y, radius, 0, oct);
        return C.features_in_area(x, y, radius, oct - 1, oct + 1);
    };
    for (int i = 0; i < NLast; i++) {
        if (!valid[i]) continue;
        const int nLastOctave = last_octave[i];
        const float radius = th * CL.scaleFactors[nLastOctave];
        const uint8_t* dMP = mp_desc + (size_t)i * 32;
        {
            const std::vector<size_t> vIndices2 = window(CL, u[i], v[i], radius, nLastOctave);
            if (vIndices2.empty()) continue;
            int bestDist = 256, bestIdx2 = -1;
            for (size_t k = 0; k < vIndices2.size(); k++) {
                const size_t i2 = vIndices2[k];
                if (cur_mp[i2] >= 0)
                    if (obs[cur_mp[i2]] > 0) continue;
                // (CurrentFrame.Nleft == -1 && mvuRight[i2] > 0: not on such a frame, :2015)
                const int dist = orc::descriptor_distance(dMP, &CL.desc[i2 * 32]);
                if (dist < bestDist) { bestDist = dist; bestIdx2 = (int)i2; }
            }
            if (bestDist <= orc::TH_HIGH) {
                cur_mp[bestIdx2] = last_mp[i];
                nmatches++;
                if (check_orientation) {
                    float rot = last_angle[i] - CL.kps[bestIdx2].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)std::round(rot * factor);
                    if (bin == orc::HISTO_LENGTH) bin = 0;
                    rotHist[bin].push_back(bestIdx2);
                }
            }
        }
        {   // :2059-2124
            const std::vector<size_t> vIndices2 = window(CR, u_r[i], v_r[i], radius, nLastOctave);
            int bestDist = 256, bestIdx2 = -1;
            for (size_t k = 0; k < vIndices2.size(); k++) {
                const size_t i2 = vIndices2[k];
                if (cur_mp[i2 + Nleft] >= 0)
                    if (obs[cur_mp[i2 + Nleft]] > 0) continue;
                const int dist = orc::descriptor_distance(dMP, &CR.desc[i2 * 32]);
                if (dist < bestDist) { bestDist = dist; bestIdx2 = (int)i2; }
            }
            if (bestDist <= orc::TH_HIGH) {
                cur_mp[bestIdx2 + Nleft] = last_mp[i];
                nmatches++;
                if (check_orientation) {
                    float rot = last_angle[i] - CR.kps[bestIdx2].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)std::round(rot * factor);
                    if (bin == orc::HISTO_LENGTH) bin = 0;
                    rotHist[bin].push_back(bestIdx2 + Nleft);
                }
            }
        }
    }
    if (check_orientation) {  // :2129-2149
        int ind1 = -1, ind2 = -1, ind3 = -1;
        orc::three_maxima(rotHist, orc::HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < orc::HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (size_t j = 0; j < rotHist[i].size(); j++) { cur_mp[rotHist[i][j]] = -1; nmatches--; }
    }
    return nmatches;
}

// The search of ORBmatcher::Fuse(pKF, vpMapPoints, th, false), ORBmatcher.cc:1499-1561, for map points that passed
// the geometric tests of :1436-1497.  fp = the KeyFrame's features (KeyFrame::GetFeaturesInArea, KeyFrame.cc:796-845,
// is Frame::GetFeaturesInArea without level arguments).  Float convention as elsewhere in this file: the products of a
// sum contracted like the reference's -O3 -march=native build (ex*ex+ey*ey+er*er -> fma(er,er,fma(ex,ex,ey*ey))).
void orc_fuse_search(void* fp, const float* inv_level_sigma2, int n, const uint8_t* valid, const float* u, const float* v,
                     const float* ur, const int* predicted_level, const float* radius, const uint8_t* mp_desc, int* best_idx,
                     int* best_dist) {
    const FrameSoA& K = *(const FrameSoA*)fp;
    for (int i = 0; i < n; i++) {
        best_idx[i] = -1;
        best_dist[i] = 256;
        if (!valid[i]) continue;
        const int nPredictedLevel = predicted_level[i];
        const std::vector<size_t> vIndices = K.features_in_area(u[i], v[i], radius[i], -1, -1);
        if (vIndices.empty()) continue;
        const uint8_t* dMP = mp_desc + (size_t)i * 32;
        int bestDist = 256, bestIdx = -1;
        for (size_t k = 0; k < vIndices.size(); k++) {
            const size_t idx = vIndices[k];
            const KeyPoint& kp = K.kps[idx];
            const int kpLevel = kp.octave;
            if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
            const float kpr = K.uRight[idx];
            const float ex = u[i] - kp.x, ey = v[i] - kp.y;
            if (kpr >= 0) {
                const float er = ur[i] - kpr;
                const float e2 = fmaf(er, er, fmaf(ex, ex, ey * ey));
                if (e2 * inv_level_sigma2[kpLevel] > 7.8) continue;
            } else {
                const float e2 = fmaf(ex, ex, ey * ey);
                if (e2 * inv_level_sigma2[kpLevel] > 5.99) continue;
            }
            const int dist = orc::descriptor_distance(dMP, &K.desc[idx * 32]);
            if (dist < bestDist) { bestDist = dist; bestIdx = (int)idx; }
        }
        best_idx[i] = bestIdx;
        best_dist[i] = bestDist;
    }
}

// The search of ORBmatcher::Fuse(pKF, vpMapPoints, th, bRight = true) on a two-camera KeyFrame, ORBmatcher.cc:1499-1561.  fp = the
// RIGHT camera's features (what KeyFrame::GetFeaturesInArea(u, v, r, true) walks, KeyFrame.cc:826-836: mGridRight / mvKeysRight).
// The loop then reads `pKF->GetKeyPoint(idx)` and `pKF->GetuRight(idx)` with the right-camera index as the grid returned it
// (:1509, :1515; `idx += pKF->GetNLeft()` only follows at :1547) — KeyFrame.h:377-385 answers mvKeys[idx] for idx < NLeft,
// mvKeysRight[idx - NLeft] beyond.  gate_kps / gate_uright hold those answers per right-camera index.  best_idx is returned in
// right-camera indices (the caller adds NLeft for GetMapPoint / AddObservation, :1565-1586).
void orc_fuse_search_gated(void* fp, const KeyPoint* gate_kps, const float* gate_uright, const float* inv_level_sigma2, int n,
                           const uint8_t* valid, const float* u, const float* v, const float* ur, const int* predicted_level,
                           const float* radius, const uint8_t* mp_desc, int* best_idx, int* best_dist) {
    const FrameSoA& K = *(const FrameSoA*)fp;
    for (int i = 0; i < n; i++) {
        best_idx[i] = -1;
        best_dist[i] = 256;
        if (!valid[i]) continue;
        const int nPredictedLevel = predicted_level[i];
        const std::vector<size_t> vIndices = K.features_in_area(u[i], v[i], radius[i], -1, -1);
        if (vIndices.empty()) continue;
        const uint8_t* dMP = mp_desc + (size_t)i * 32;
        int bestDist = 256, bestIdx = -1;
        for (size_t k = 0; k < vIndices.size(); k++) {
            const size_t idx = vIndices[k];
            const KeyPoint& kp = gate_kps[idx];                       // pKF->GetKeyPoint(idx)
            const int kpLevel = kp.octave;
            if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
            const float kpr = gate_uright ? gate_uright[idx] : -1.0f;  // pKF->GetuRight(idx)
            const float ex = u[i] - kp.x, ey = v[i] - kp.y;
            if (kpr >= 0) {
                const float er = ur[i] - kpr;
                const float e2 = fmaf(er, er, fmaf(ex, ex, ey * ey));
                if (e2 * inv_level_sigma2[kpLevel] > 7.8) continue;
            } else {
                const float e2 = fmaf(ex, ex, ey * ey);
                if (e2 * inv_level_sigma2[kpLevel] > 5.99) continue;
            }
            const int dist = orc::descriptor_distance(dMP, &K.desc[idx * 32]);   // row idx + NLeft of mDescriptors = row idx of the right camera
            if (dist < bestDist) { bestDist = dist; bestIdx = (int)idx; }
        }
        best_idx[i] = bestIdx;
        best_dist[i] = bestDist;
    }
}

// ORBmatcher::SearchByProjection(Frame& CurrentFrame, pKF, sAlreadyFound, th, ORBdist), ORBmatcher.cc:2154-2275, from the
// projected coordinates on.  One entry per KeyFrame map point that passed :2173-2196 (valid), with its projection u, v,
// predicted level (:2198), the KeyFrame keypoint's angle (:2237), descriptor and id.  cur_mp[N] in/out.
int orc_search_by_projection_kf(void* fp, int n, const uint8_t* valid, const float* u, const float* v,
                                const int* predicted_level, const float* kf_angle, const uint8_t* mp_desc, const int* mp_id,
                                int* cur_mp, float th, int ORBdist, int check_orientation) {
    FrameSoA& C = *(FrameSoA*)fp;
    int nmatches = 0;
    std::vector<int> rotHist[orc::HISTO_LENGTH];
    const float factor = 1.0f / orc::HISTO_LENGTH;
    for (int i = 0; i < n; i++) {
        if (!valid[i]) continue;
        const int nPredictedLevel = predicted_level[i];
        const float radius = th * C.scaleFactors[nPredictedLevel];
        const std::vector<size_t> vIndices2 = C.features_in_area(u[i], v[i], radius, nPredictedLevel - 1, nPredictedLevel + 1);
        if (vIndices2.empty()) continue;
        const uint8_t* dMP = mp_desc + (size_t)i * 32;
        int bestDist = 256, bestIdx2 = -1;
        for (size_t k = 0; k < vIndices2.size(); k++) {
            const size_t i2 = vIndices2[k];
            if (cur_mp[i2] >= 0) continue;  // :2214-2215
            const int dist = orc::descriptor_distance(dMP, &C.desc[i2 * 32]);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = (int)i2; }
        }
        if (bestDist <= ORBdist) {
            cur_mp[bestIdx2] = mp_id[i];
            nmatches++;
            if (check_orientation) {
                float rot = kf_angle[i] - C.kps[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == orc::HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        orc::three_maxima(rotHist, orc::HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < orc::HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (size_t j = 0; j < rotHist[i].size(); j++) { cur_mp[rotHist[i][j]] = -1; nmatches--; }
    }
    return nmatches;
}

// ORBmatcher::SearchByProjection(pKF, Scw, vpPoints, vpMatched, th, ratioHamming), ORBmatcher.cc:482-527 (the same
// search in the vpPointsKFs form :700-749; SearchByProjectionLoop differs, see orc_search_by_projection_loop), from the
// projected coordinates on.
int orc_search_by_projection_sim3(void* fp, int n, const uint8_t* valid, const float* u, const float* v,
                                  const int* predicted_level, const uint8_t* mp_desc, const int* mp_id, int* matched, float th,
                                  float max_dist) {
    FrameSoA& K = *(FrameSoA*)fp;
    int nmatches = 0;
    for (int i = 0; i < n; i++) {
        if (!valid[i]) continue;
        const int nPredictedLevel = predicted_level[i];
        const float radius = th * K.scaleFactors[nPredictedLevel];
        const std::vector<size_t> vIndices = K.features_in_area(u[i], v[i], radius, -1, -1);
        if (vIndices.empty()) continue;
        const uint8_t* dMP = mp_desc + (size_t)i * 32;
        int bestDist = 256, bestIdx = -1;
        for (size_t k = 0; k < vIndices.size(); k++) {
            const size_t idx = vIndices[k];
            if (matched[idx] >= 0) continue;
            const int kpLevel = K.kps[idx].octave;
            if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
            const int dist = orc::descriptor_distance(dMP, &K.desc[idx * 32]);
            if (dist < bestDist) { bestDist = dist; bestIdx = (int)idx; }
        }
        if ((float)bestDist <= max_dist) {
            matched[bestIdx] = mp_id[i];
            nmatches++;
        }
    }
    return nmatches;
}

// ORBmatcher::SearchByProjectionLoop(pKF, Scw, vpPoints, vpMatched, vpMatchedKF, th, ratioHamming), ORBmatcher.cc:532-637, from
// the projected coordinates on.  Unlike the other Sim3 forms: a candidate keypoint must HOLD a good map point
// (train_ok[idx] = vpMapPointsToMatch[idx] && !isBad(), :609-610), the level band is predicted-1 .. predicted+1 (:613), and
// the result is stored per POINT (vpMatched[iMP] = vpMapPointsToMatch[bestIdx], :626-631): no keypoint is claimed, queries
// are independent.  valid[i] includes "!vpMatched[iMP]" (:555).  best_idx[i] = accepted keypoint or -1; returns nmatches.
int orc_search_by_projection_loop(void* fp, int n, const uint8_t* valid, const float* u, const float* v,
                                  const int* predicted_level, const uint8_t* mp_desc, const uint8_t* train_ok, float th,
                                  float max_dist, int* best_idx) {
    const FrameSoA& K = *(const FrameSoA*)fp;
    int nmatches = 0;
    for (int i = 0; i < n; i++) {
        best_idx[i] = -1;
        if (!valid[i]) continue;
        const int nPredictedLevel = predicted_level[i];
        const float radius = th * K.scaleFactors[nPredictedLevel];
        const std::vector<size_t> vIndices = K.features_in_area(u[i], v[i], radius, -1, -1);
        if (vIndices.empty()) continue;
        const uint8_t* dMP = mp_desc + (size_t)i * 32;
        int bestDist = 256, bestIdx = -1;
        for (size_t k = 0; k < vIndices.size(); k++) {
            const size_t idx = vIndices[k];
            if (!train_ok[idx]) continue;
            const int kpLevel = K.kps[idx].octave;
            if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel + 1) continue;
            const int dist = orc::descriptor_distance(dMP, &K.desc[idx * 32]);
            if (dist < bestDist) { bestDist = dist; bestIdx = (int)idx; }
        }
        if ((float)bestDist <= max_dist) { best_idx[i] = bestIdx; nmatches++; }
    }
    return nmatches;
}

// ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, S12, th), ORBmatcher.cc:1718-1939, from the projected coordinates on.
// Pass 1 (:1758-1848): every map point i1 of pKF1 the loop reaches (valid1: present, not already matched, not bad, positive
// depth, inside pKF2's image, distance inside the scale pyramid) is searched in pKF2 — GetFeaturesInArea(u, v,
// th * mvScaleFactors[level]) (KeyFrame.cc:796-845: no level argument), level band predicted-1 .. predicted, first strict
// minimum, bestDist <= TH_HIGH -> vnMatch1[i1].  Pass 2 (:1850-1920) the same from pKF2 into pKF1.  Agreement (:1922-1937):
// a pair survives when both passes point at each other.  match12[i1] = idx2 or -1; returns nFound.
static void sim3_pass(const FrameSoA& K, int n, const uint8_t* valid, const float* u, const float* v, const int* level,
                      const uint8_t* desc, float th, std::vector<int>& vnMatch) {
    vnMatch.assign(n, -1);
    for (int i = 0; i < n; i++) {
        if (!valid[i]) continue;
        const int nPredictedLevel = level[i];
        const float radius = th * K.scaleFactors[nPredictedLevel];
        const std::vector<size_t> vIndices = K.features_in_area(u[i], v[i], radius, -1, -1);
        if (vIndices.empty()) continue;
        const uint8_t* dMP = desc + (size_t)i * 32;
        int bestDist = INT_MAX, bestIdx = -1;
        for (size_t k = 0; k < vIndices.size(); k++) {
            const size_t idx = vIndices[k];
            const KeyPoint& kp = K.kps[idx];
            if (kp.octave < nPredictedLevel - 1 || kp.octave > nPredictedLevel) continue;
            const int dist = orc::descriptor_distance(dMP, &K.desc[idx * 32]);
            if (dist < bestDist) { bestDist = dist; bestIdx = (int)idx; }
        }
        if (bestDist <= orc::TH_HIGH) vnMatch[i] = bestIdx;
    }
}
int orc_search_by_sim3(void* kf1, void* kf2, int n1, const uint8_t* valid1, const float* u1, const float* v1, const int* level1,
                       const uint8_t* desc1, int n2, const uint8_t* valid2, const float* u2, const float* v2, const int* level2,
                       const uint8_t* desc2, float th, int* match12) {
    const FrameSoA &K1 = *(const FrameSoA*)kf1, &K2 = *(const FrameSoA*)kf2;
    std::vector<int> vnMatch1, vnMatch2;
    sim3_pass(K2, n1, valid1, u1, v1, level1, desc1, th, vnMatch1);  // KF1's points searched in KF2
    sim3_pass(K1, n2, valid2, u2, v2, level2, desc2, th, vnMatch2);  // KF2's points searched in KF1
    int nFound = 0;
    for (int i1 = 0; i1 < n1; i1++) {
        match12[i1] = -1;
        const int idx2 = vnMatch1[i1];
        if (idx2 >= 0) {
            const int idx1 = idx2 < n2 ? vnMatch2[idx2] : -1;
            if (idx1 == i1) { match12[i1] = idx2; nFound++; }
        }
    }
    return nFound;
}

// The search of ORBmatcher::Fuse(pKF, Scw, vpPoints, th, vpReplacePoint), ORBmatcher.cc:1661-1696, for the points that passed
// :1622-1656 (not bad, not already in the KeyFrame, positive depth, inside the image, distance, viewing angle): no
// reprojection-error gate in this form; best_dist = INT_MAX when the window holds no keypoint of the level band.
void orc_fuse_sim3_search(void* fp, int n, const uint8_t* valid, const float* u, const float* v, const int* predicted_level,
                          const uint8_t* mp_desc, float th, int* best_idx, int* best_dist) {
    const FrameSoA& K = *(const FrameSoA*)fp;
    for (int i = 0; i < n; i++) {
        best_idx[i] = -1;
        best_dist[i] = INT_MAX;
        if (!valid[i]) continue;
        const int nPredictedLevel = predicted_level[i];
        const float radius = th * K.scaleFactors[nPredictedLevel];
        const std::vector<size_t> vIndices = K.features_in_area(u[i], v[i], radius, -1, -1);
        if (vIndices.empty()) continue;
        const uint8_t* dMP = mp_desc + (size_t)i * 32;
        int bestDist = INT_MAX, bestIdx = -1;
        for (size_t k = 0; k < vIndices.size(); k++) {
            const size_t idx = vIndices[k];
            const int kpLevel = K.kps[idx].octave;
            if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
            const int dist = orc::descriptor_distance(dMP, &K.desc[idx * 32]);
            if (dist < bestDist) { bestDist = dist; bestIdx = (int)idx; }
        }
        best_idx[i] = bestIdx;
        best_dist[i] = bestDist;
    }
}

// ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize), ORBmatcher.cc:755-870 (monocular
// initialisation).  prev = vbPrevMatched (x, y per F1 keypoint, updated in place like :864-867); returns nmatches.
int orc_search_for_initialization(void* f1, void* f2, float* prev_xy, int window_size, float nnratio, int check_orientation,
                                  int* vnMatches12) {
    const FrameSoA &F1 = *(const FrameSoA*)f1, &F2 = *(const FrameSoA*)f2;
    int nmatches = 0;
    for (int i = 0; i < F1.N; i++) vnMatches12[i] = -1;
    std::vector<int> rotHist[orc::HISTO_LENGTH];
    const float factor = 1.0f / orc::HISTO_LENGTH;
    std::vector<int> vMatchedDistance(F2.N, INT_MAX), vnMatches21(F2.N, -1);
    for (int i1 = 0; i1 < F1.N; i1++) {
        const KeyPoint kp1 = F1.kps[i1];
        const int level1 = kp1.octave;
        if (level1 > 0) continue;
        const std::vector<size_t> vIndices2 = F2.features_in_area(prev_xy[2 * i1], prev_xy[2 * i1 + 1], (float)window_size, level1, level1);
        if (vIndices2.empty()) continue;
        const uint8_t* d1 = &F1.desc[(size_t)i1 * 32];
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (size_t k = 0; k < vIndices2.size(); k++) {
            const size_t i2 = vIndices2[k];
            const int dist = orc::descriptor_distance(d1, &F2.desc[i2 * 32]);
            if (vMatchedDistance[i2] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = (int)i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= orc::TH_LOW) {
            if (bestDist < (float)bestDist2 * nnratio) {
                if (vnMatches21[bestIdx2] >= 0) { vnMatches12[vnMatches21[bestIdx2]] = -1; nmatches--; }
                vnMatches12[i1] = bestIdx2;
                vnMatches21[bestIdx2] = i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (check_orientation) {
                    float rot = F1.kps[i1].angle - F2.kps[bestIdx2].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)std::round(rot * factor);
                    if (bin == orc::HISTO_LENGTH) bin = 0;
                    rotHist[bin].push_back(i1);
                }
            }
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        orc::three_maxima(rotHist, orc::HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < orc::HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) {
                const int idx1 = rotHist[i][j];
                if (vnMatches12[idx1] >= 0) { vnMatches12[idx1] = -1; nmatches--; }
            }
        }
    }
    for (int i1 = 0; i1 < F1.N; i1++)
        if (vnMatches12[i1] >= 0) { prev_xy[2 * i1] = F2.kps[vnMatches12[i1]].x; prev_xy[2 * i1 + 1] = F2.kps[vnMatches12[i1]].y; }
    return nmatches;
}

void orc_three_maxima(const int* sizes, int L, int* ind) {
    std::vector<std::vector<int>> h(L);
    for (int i = 0; i < L; i++) h[i].resize(sizes[i]);
    ind[0] = ind[1] = ind[2] = -1;
    orc::three_maxima(h.data(), L, ind[0], ind[1], ind[2]);
}

// Frame::ComputeStereoMatches, Frame.cc:743-913.  left/right: keypoints + descriptors of the two eyes;
// pyrL/pyrR: the extractors' mvImagePyramid planes (interior pixels; plane l has rows[l] x cols[l], row
// stride strides[l]).  Outputs mvuRight / mvDepth (N entries, -1 = no match).
// The reference indexes cv::Mat ranges unchecked against the interior (an out-of-range window would trip
// a CV_Assert there); this restatement skips such a keypoint instead and reports the count in *n_oob.
void orc_compute_stereo_matches(const void* kpsL_, int N, const uint8_t* descL, const void* kpsR_, int Nr,
                                const uint8_t* descR, const uint8_t* const* pyrL, const uint8_t* const* pyrR,
                                const int* rows, const int* cols, const int* strides, const float* scaleFactors,
                                const float* invScaleFactors, float mb, float mbf, float* uRight, float* depth,
                                int* n_oob) {
    const KeyPoint* kpsL = (const KeyPoint*)kpsL_;
    const KeyPoint* kpsR = (const KeyPoint*)kpsR_;
    for (int i = 0; i < N; i++) { uRight[i] = -1.0f; depth[i] = -1.0f; }
    if (n_oob) *n_oob = 0;
    const int thOrbDist = (orc::TH_HIGH + orc::TH_LOW) / 2;
    const int nRows = rows[0];
    std::vector<std::vector<size_t>> vRowIndices(nRows);
    for (int iR = 0; iR < Nr; iR++) {  // :758-771
        const KeyPoint& kp = kpsR[iR];
        const float kpY = kp.y;
        const float r = 2.0f * scaleFactors[kp.octave];
        const int maxr = (int)std::ceil(kpY + r);
        const int minr = (int)std::floor(kpY - r);
        for (int yi = minr; yi <= maxr; yi++)
            if (yi >= 0 && yi < nRows) vRowIndices[yi].push_back(iR);  // guard: reference indexes unchecked
    }
    const float minZ = mb, minD = 0, maxD = mbf / minZ;
    std::vector<std::pair<int, int>> vDistIdx;
    for (int iL = 0; iL < N; iL++) {
        const KeyPoint& kpL = kpsL[iL];
        const int levelL = kpL.octave;
        const float vL = kpL.y, uL = kpL.x;
        const int row = (int)vL;  // vRowIndices[vL]: float -> size_t truncation
        if (row < 0 || row >= nRows) continue;
        const std::vector<size_t>& vCandidates = vRowIndices[row];
        if (vCandidates.empty()) continue;
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = orc::TH_HIGH;
        size_t bestIdxR = 0;
        const uint8_t* dL = descL + (size_t)iL * 32;
        for (size_t iC = 0; iC < vCandidates.size(); iC++) {
            const size_t iR = vCandidates[iC];
            const KeyPoint& kpR = kpsR[iR];
            if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
            const float uR = kpR.x;
            if (uR >= minU && uR <= maxU) {
                const int dist = orc::descriptor_distance(dL, descR + iR * 32);
                if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
            }
        }
        if (bestDist < thOrbDist) {  // :829-897
            const float uR0 = kpsR[bestIdxR].x;
            const float scaleFactor = invScaleFactors[kpL.octave];
            const float scaleduL = std::round(kpL.x * scaleFactor);
            const float scaledvL = std::round(kpL.y * scaleFactor);
            const float scaleduR0 = std::round(uR0 * scaleFactor);
            const int w = 5, L = 5;
            const int lv = kpL.octave;
            int bestDistS = INT_MAX, bestincR = 0;
            float vDists[2 * 5 + 1];
            const float iniu = scaleduR0 + L - w;
            const float endu = scaleduR0 + L + w + 1;
            if (iniu < 0 || endu >= cols[lv]) continue;
            const int y0 = (int)(scaledvL - w), xL0 = (int)(scaleduL - w);
            const int xR0 = (int)(scaleduR0 - L - w);
            if (y0 < 0 || y0 + 2 * w + 1 > rows[lv] || xL0 < 0 || xL0 + 2 * w + 1 > cols[lv] || xR0 < 0 ||
                (int)(scaleduR0 + L + w + 1) > cols[lv]) {
                if (n_oob) (*n_oob)++;
                continue;
            }
            for (int incR = -L; incR <= +L; incR++) {
                int sad = 0;  // cv::norm(IL, IR, NORM_L1) on 11x11 8-bit windows
                for (int yy = 0; yy < 2 * w + 1; yy++) {
                    const uint8_t* pl = pyrL[lv] + (size_t)(y0 + yy) * strides[lv] + xL0;
                    const uint8_t* pr = pyrR[lv] + (size_t)(y0 + yy) * strides[lv] + (int)(scaleduR0 + incR - w);
                    for (int xx = 0; xx < 2 * w + 1; xx++) sad += std::abs((int)pl[xx] - (int)pr[xx]);
                }
                const float dist = (float)(double)sad;
                if (dist < bestDistS) { bestDistS = dist; bestincR = incR; }
                vDists[L + incR] = dist;
            }
            if (bestincR == -L || bestincR == L) continue;
            const float dist1 = vDists[L + bestincR - 1], dist2 = vDists[L + bestincR], dist3 = vDists[L + bestincR + 1];
            const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
            if (deltaR < -1 || deltaR > 1) continue;
            float bestuR = scaleFactors[kpL.octave] * ((float)scaleduR0 + (float)bestincR + deltaR);
            float disparity = (uL - bestuR);
            if (disparity >= minD && disparity < maxD) {
                if (disparity <= 0) { disparity = 0.01; bestuR = uL - 0.01; }
                depth[iL] = mbf / disparity;
                uRight[iL] = bestuR;
                vDistIdx.push_back(std::pair<int, int>(bestDistS, iL));
            }
        }
    }
    if (vDistIdx.empty()) return;  // the reference reads vDistIdx[0] of an empty vector here (UB)
    std::sort(vDistIdx.begin(), vDistIdx.end());  // :899-912
    const float median = vDistIdx[vDistIdx.size() / 2].first;
    const float thDist = 1.5f * 1.4f * median;
    for (int i = (int)vDistIdx.size() - 1; i >= 0; i--) {
        if (vDistIdx[i].first < thDist) break;
        uRight[vDistIdx[i].second] = -1;
        depth[vDistIdx[i].second] = -1;
    }
}

// Dense brute-force top-2 (the BFMatcher::knnMatch(k=2) shape of Frame.cc:1076 on ORBmatcher::DescriptorDistance,
// ORBmatcher.cc:2323-2339): candidates scanned in index order, strict '<'.  best_idx -1 / distances 256 when absent.
void orc_dense_top2(const uint8_t* q, int nq, const uint8_t* t, int nt, int* best_idx, int* best_dist, int* second_dist) {
    for (int i = 0; i < nq; i++) {
        int b1 = 256, b2 = 256, bi = -1;
        for (int j = 0; j < nt; j++) {
            const int d = orc::descriptor_distance(q + (size_t)i * 32, t + (size_t)j * 32);
            if (d < b1) { b2 = b1; b1 = d; bi = j; }
            else if (d < b2) b2 = d;
        }
        best_idx[i] = bi; best_dist[i] = b1; second_dist[i] = b2;
    }
}

// ORBmatcher::SearchByBoW, the three forms (ORBmatcher.cc:223-421 KeyFrame->Frame with Nleft == -1, :872-1016 and
// :1018-1166 KeyFrame->KeyFrame).  Set 1 = the queries (pKF / pKF1), set 2 = the trains (F / pKF2).
// valid1[i]: the query is visited (:253-263 / :910-920 / :1066-1078: map point present, not bad, ..., descriptor not
// empty).  avail2[j]: the train may be chosen at all (:934-944 / :1090-1103; all ones for the Frame form, whose
// vpMapPointMatches starts empty, :227).  FeatureVectors as CSR with ascending node ids (std::map order).
// inclusive: bestDist1 <= TH_LOW (:332) vs bestDist1 < TH_LOW (:959, :1118).  The rotation histogram stores the pair
// (the Frame form pushes bestIdxF and resets that entry, the KeyFrame forms push idx1: the same pair either way).
// match12[n1] / match21[n2]: partner index or -1, after the histogram filter.  Returns nmatches.
int orc_search_by_bow(int n1, int n2, const uint8_t* desc1, const uint8_t* desc2, const uint8_t* valid1,
                      const uint8_t* avail2, int nn1, const int* node1, const int* begin1, const int* feat1, int nn2,
                      const int* node2, const int* begin2, const int* feat2, const float* angle1, const float* angle2,
                      int th_low, int inclusive, float nnratio, int check_orientation, int* match12, int* match21) {
    for (int i = 0; i < n1; i++) match12[i] = -1;
    std::vector<char> matched2(n2, 0);
    std::vector<int> rotHist[orc::HISTO_LENGTH];
    const float factor = 1.0f / orc::HISTO_LENGTH;
    int nmatches = 0;
    int it1 = 0, it2 = 0;
    while (it1 != nn1 && it2 != nn2) {
        if (node1[it1] == node2[it2]) {
            for (int k1 = begin1[it1]; k1 < begin1[it1 + 1]; k1++) {
                const int idx1 = feat1[k1];
                if (!valid1[idx1]) continue;
                const uint8_t* d1 = desc1 + (size_t)idx1 * 32;
                int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
                for (int k2 = begin2[it2]; k2 < begin2[it2 + 1]; k2++) {
                    const int idx2 = feat2[k2];
                    if (matched2[idx2] || (avail2 && !avail2[idx2])) continue;
                    const int dist = orc::descriptor_distance(d1, desc2 + (size_t)idx2 * 32);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = idx2; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (inclusive ? bestDist1 <= th_low : bestDist1 < th_low) {
                    if ((float)bestDist1 < nnratio * (float)bestDist2) {
                        match12[idx1] = bestIdx2;
                        matched2[bestIdx2] = 1;
                        if (check_orientation) {
                            float rot = angle1[idx1] - angle2[bestIdx2];
                            if (rot < 0.0) rot += 360.0f;
                            int bin = (int)std::round(rot * factor);
                            if (bin == orc::HISTO_LENGTH) bin = 0;
                            rotHist[bin].push_back(idx1);
                        }
                        nmatches++;
                    }
                }
            }
            it1++;
            it2++;
        } else if (node1[it1] < node2[it2]) {
            it1 = (int)(std::lower_bound(node1, node1 + nn1, node2[it2]) - node1);
        } else {
            it2 = (int)(std::lower_bound(node2, node2 + nn2, node1[it1]) - node2);
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        orc::three_maxima(rotHist, orc::HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < orc::HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) { match12[rotHist[i][j]] = -1; nmatches--; }
        }
    }
    if (match21) {
        for (int j = 0; j < n2; j++) match21[j] = -1;
        for (int i = 0; i < n1; i++) if (match12[i] >= 0) match21[match12[i]] = i;
    }
    return nmatches;
}

// ORBmatcher::SearchByBoW(pKF, F, vpMapPointMatches) on a two-camera frame (F.Nleft != -1), ORBmatcher.cc:223-421 statement by
// statement with the arms of :276-309 and :357-382.  Set 1 = the KeyFrame (valid1 = pMP && !pMP->isBad()), set 2 = the frame's
// n2 = N features (left camera's rows first), angle2 = mvKeys / mvKeysRight angles in row order.  match21[n2] = vpMapPointMatches as
// KeyFrame feature indices.  Returns nmatches.
int orc_search_by_bow_rig(int n1, int n2, int n_left, const uint8_t* desc1, const uint8_t* desc2, const uint8_t* valid1, int nn1,
                          const int* node1, const int* begin1, const int* feat1, int nn2, const int* node2, const int* begin2,
                          const int* feat2, const float* angle1, const float* angle2, int th_low, float nnratio, int check_orientation,
                          int* match21) {
    for (int j = 0; j < n2; j++) match21[j] = -1;
    std::vector<int> rotHist[orc::HISTO_LENGTH];
    const float factor = 1.0f / orc::HISTO_LENGTH;
    int nmatches = 0;
    int it1 = 0, it2 = 0;
    while (it1 != nn1 && it2 != nn2) {
        if (node1[it1] == node2[it2]) {
            for (int iKF = begin1[it1]; iKF < begin1[it1 + 1]; iKF++) {
                const int realIdxKF = feat1[iKF];
                if (!valid1[realIdxKF]) continue;
                const uint8_t* dKF = desc1 + (size_t)realIdxKF * 32;
                int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
                int bestDist1R = 256, bestIdxFR = -1, bestDist2R = 256;
                for (int iF = begin2[it2]; iF < begin2[it2 + 1]; iF++) {
                    const int realIdxF = feat2[iF];
                    if (match21[realIdxF] >= 0) continue;
                    const int dist = orc::descriptor_distance(dKF, desc2 + (size_t)realIdxF * 32);
                    if (realIdxF < n_left && dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
                    else if (realIdxF < n_left && dist < bestDist2) bestDist2 = dist;
                    if (realIdxF >= n_left && dist < bestDist1R) { bestDist2R = bestDist1R; bestDist1R = dist; bestIdxFR = realIdxF; }
                    else if (realIdxF >= n_left && dist < bestDist2R) bestDist2R = dist;
                }
                auto push = [&](int idxF) {
                    if (!check_orientation) return;
                    float rot = angle1[realIdxKF] - angle2[idxF];
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)std::round(rot * factor);
                    if (bin == orc::HISTO_LENGTH) bin = 0;
                    rotHist[bin].push_back(idxF);
                };
                if (bestDist1 <= th_low) {
                    if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
                        match21[bestIdxF] = realIdxKF;
                        push(bestIdxF);
                        nmatches++;
                    }
                    if (bestDist1R <= th_low) {   // (`|| true`: no ratio test)
                        match21[bestIdxFR] = realIdxKF;
                        push(bestIdxFR);
                        nmatches++;
                    }
                }
                (void)bestDist2R;
            }
            it1++; it2++;
        } else if (node1[it1] < node2[it2]) it1++;   // lower_bound on ascending ids
        else it2++;
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        orc::three_maxima(rotHist, orc::HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < orc::HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) { match21[rotHist[i][j]] = -1; nmatches--; }
        }
    }
    return nmatches;
}

// ORBmatcher::SearchForTriangulation (ORBmatcher.cc:1168-1402), pinhole cameras without a second camera
// (mpCamera2 == nullptr).  valid1[i]: the query is visited (no map point :1237-1241, stereo when bOnlyStereo :1245-1247,
// descriptor not empty); avail2[j]: the train is eligible apart from the running vbMatched2 (:1264, 1269-1271);
// stereo1/stereo2 = GetuRight(idx) >= 0 (:1243, :1267).  kp1/kp2 = GetKeyPoint (x, y per feature), ep_thresh2[j] =
// 100*pKF2->mvScaleFactors[kp2.octave] (:1287), unc2[j] = pKF2->mvLevelSigma2[kp2.octave] (:1332), F12 = the matrix
// Pinhole::epipolarConstrain builds from R12, t12 and the two K (Pinhole.cpp:109-112; the same for every candidate of
// a pair, so the caller evaluates that Eigen expression once), ep = the epipole (:1179-1181).
// Scalar float expressions: products contracted the way the reference's -O3 -march=native build does it (first
// product of a sum fused, the other one rounded: x*p + y*q + r -> fmaf(x, p, y*q) + r) — stated convention.
// Returns nmatches; match12[n1] after the orientation filter.
int orc_search_for_triangulation(int n1, int n2, const uint8_t* desc1, const uint8_t* desc2, const uint8_t* valid1,
                                 const uint8_t* avail2, const uint8_t* stereo1, const uint8_t* stereo2, int nn1,
                                 const int* node1, const int* begin1, const int* feat1, int nn2, const int* node2,
                                 const int* begin2, const int* feat2, const float* xy1, const float* xy2,
                                 const float* angle1, const float* angle2, const float* ep_thresh2, const float* unc2,
                                 const float* F12, const float* ep, int coarse, int check_orientation, int* match12) {
    for (int i = 0; i < n1; i++) match12[i] = -1;
    std::vector<char> matched2(n2, 0);
    std::vector<int> rotHist[orc::HISTO_LENGTH];
    const float factor = 1.0f / orc::HISTO_LENGTH;
    int nmatches = 0;
    int it1 = 0, it2 = 0;
    while (it1 != nn1 && it2 != nn2) {
        if (node1[it1] == node2[it2]) {
            for (int k1 = begin1[it1]; k1 < begin1[it1 + 1]; k1++) {
                const int idx1 = feat1[k1];
                if (!valid1[idx1]) continue;
                const bool bStereo1 = stereo1[idx1] != 0;
                const float x1 = xy1[2 * idx1], y1 = xy1[2 * idx1 + 1];
                const uint8_t* d1 = desc1 + (size_t)idx1 * 32;
                int bestDist = orc::TH_LOW, bestIdx2 = -1;
                for (int k2 = begin2[it2]; k2 < begin2[it2 + 1]; k2++) {
                    const int idx2 = feat2[k2];
                    if (matched2[idx2] || !avail2[idx2]) continue;
                    const bool bStereo2 = stereo2[idx2] != 0;
                    const int dist = orc::descriptor_distance(d1, desc2 + (size_t)idx2 * 32);
                    if (dist > orc::TH_LOW || dist > bestDist) continue;
                    const float x2 = xy2[2 * idx2], y2 = xy2[2 * idx2 + 1];
                    if (!bStereo1 && !bStereo2) {
                        const float distex = ep[0] - x2, distey = ep[1] - y2;
                        if (fmaf(distex, distex, distey * distey) < ep_thresh2[idx2]) continue;
                    }
                    bool ok = coarse != 0;
                    if (!ok) {  // Pinhole::epipolarConstrain, Pinhole.cpp:114-131
                        const float a = fmaf(x1, F12[0], y1 * F12[3]) + F12[6];
                        const float b = fmaf(x1, F12[1], y1 * F12[4]) + F12[7];
                        const float c = fmaf(x1, F12[2], y1 * F12[5]) + F12[8];
                        const float num = fmaf(a, x2, b * y2) + c;
                        const float den = fmaf(a, a, b * b);
                        if (den == 0) ok = false;
                        else {
                            const float dsqr = num * num / den;
                            ok = (double)dsqr < 3.84 * (double)unc2[idx2];
                        }
                    }
                    if (ok) { bestIdx2 = idx2; bestDist = dist; }
                }
                if (bestIdx2 >= 0) {
                    match12[idx1] = bestIdx2;
                    matched2[bestIdx2] = 1;
                    nmatches++;
                    if (check_orientation) {
                        float rot = angle1[idx1] - angle2[bestIdx2];
                        if (rot < 0.0) rot += 360.0f;
                        int bin = (int)std::round(rot * factor);
                        if (bin == orc::HISTO_LENGTH) bin = 0;
                        rotHist[bin].push_back(idx1);
                    }
                }
            }
            it1++;
            it2++;
        } else if (node1[it1] < node2[it2]) {
            it1 = (int)(std::lower_bound(node1, node1 + nn1, node2[it2]) - node1);
        } else {
            it2 = (int)(std::lower_bound(node2, node2 + nn2, node1[it1]) - node2);
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        orc::three_maxima(rotHist, orc::HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < orc::HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) { match12[rotHist[i][j]] = -1; nmatches--; }
        }
    }
    return nmatches;
}

// ORBmatcher::SearchForTriangulation (ORBmatcher.cc:1168-1402) for two KeyFrames of a two-camera rig (pKF1->mpCamera2 &&
// pKF2->mpCamera2): bStereo1 / bStereo2 are false (`!pKF->mpCamera2 && ...`, :1243, :1267 — valid1 / avail2 carry the bOnlyStereo
// consequence, nothing visited), the epipole-distance test is skipped (`&& !pKF1->mpCamera2`, :1283), and the geometric test of
// :1332 is pCamera1->epipolarConstrain(pCamera2, kp1, kp2, R12, t12, ...) with cameras and relative pose chosen per candidate from
// the side (left / right image) of the two features (:1294-1330).  accept(ctx, idx1, idx2) stands for that call — a camera model
// is not part of this oracle; the tests' accept reproduces what the harness's camera answers.  The scan, the running
// `dist > TH_LOW || dist > bestDist` rule (:1277), vbMatched2 and the rotation histogram are the reference's statements; accept is
// only asked where the reference asks (after :1277).
typedef int (*orc_pair_accept)(void* ctx, int idx1, int idx2);
int orc_search_for_triangulation_rig(int n1, int n2, const uint8_t* desc1, const uint8_t* desc2, const uint8_t* valid1, const uint8_t* avail2,
                                     int nn1, const int* node1, const int* begin1, const int* feat1, int nn2, const int* node2,
                                     const int* begin2, const int* feat2, const float* angle1, const float* angle2, int coarse,
                                     int check_orientation, orc_pair_accept accept, void* ctx, int* match12, long* n_accept_calls) {
    for (int i = 0; i < n1; i++) match12[i] = -1;
    std::vector<char> matched2(n2, 0);
    std::vector<int> rotHist[orc::HISTO_LENGTH];
    const float factor = 1.0f / orc::HISTO_LENGTH;
    int nmatches = 0;
    long calls = 0;
    int it1 = 0, it2 = 0;
    while (it1 != nn1 && it2 != nn2) {
        if (node1[it1] == node2[it2]) {
            for (int k1 = begin1[it1]; k1 < begin1[it1 + 1]; k1++) {
                const int idx1 = feat1[k1];
                if (!valid1[idx1]) continue;
                const uint8_t* d1 = desc1 + (size_t)idx1 * 32;
                int bestDist = orc::TH_LOW, bestIdx2 = -1;
                for (int k2 = begin2[it2]; k2 < begin2[it2 + 1]; k2++) {
                    const int idx2 = feat2[k2];
                    if (matched2[idx2] || (avail2 && !avail2[idx2])) continue;
                    const int dist = orc::descriptor_distance(d1, desc2 + (size_t)idx2 * 32);
                    if (dist > orc::TH_LOW || dist > bestDist) continue;
                    bool ok = coarse != 0;
                    if (!ok) { ok = accept(ctx, idx1, idx2) != 0; calls++; }
                    if (ok) { bestIdx2 = idx2; bestDist = dist; }
                }
                if (bestIdx2 >= 0) {
                    match12[idx1] = bestIdx2;
                    matched2[bestIdx2] = 1;
                    nmatches++;
                    if (check_orientation) {
                        float rot = angle1[idx1] - angle2[bestIdx2];
                        if (rot < 0.0) rot += 360.0f;
                        int bin = (int)std::round(rot * factor);
                        if (bin == orc::HISTO_LENGTH) bin = 0;
                        rotHist[bin].push_back(idx1);
                    }
                }
            }
            it1++;
            it2++;
        } else if (node1[it1] < node2[it2]) {
            it1 = (int)(std::lower_bound(node1, node1 + nn1, node2[it2]) - node1);
        } else {
            it2 = (int)(std::lower_bound(node2, node2 + nn2, node1[it1]) - node2);
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        orc::three_maxima(rotHist, orc::HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < orc::HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) { match12[rotHist[i][j]] = -1; nmatches--; }
        }
    }
    if (n_accept_calls) *n_accept_calls = calls;
    return nmatches;
}

}  // extern "C"
