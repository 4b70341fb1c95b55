// Compiles ms-slam_amd/host/MapSparsification_device.h against minimal stand-ins of the reference's KeyFrame / MapPoint
// (member names of include/KeyFrame.h / include/MapPoint.h), builds the object graph of a synthetic window from the flat
// arrays the pytest wrote, and dumps the constraint matrix in terms of the ORIGINAL point / keyframe ids.
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <tuple>
#include <vector>

#include "MapSparsification_device.h"

namespace ORB_SLAM3 {
struct KeyFrame;
struct MapPoint {
    int id = -1, nObs = 0;
    bool mbBad = false;
    long unsigned int mnMapSparsificationId = 0, mnIndexForSparsification = 0;
    std::map<std::shared_ptr<KeyFrame>, std::tuple<int, int>> mObservations;
    bool isBad() const { return mbBad; }
    int Observations() const { return nObs; }
    std::map<std::shared_ptr<KeyFrame>, std::tuple<int, int>> GetObservations() const { return mObservations; }
};
struct KeyFrame {
    int id = -1, numMps = 0;
    long unsigned int mnMapSaprsificationId = 0;
    std::vector<std::shared_ptr<MapPoint>> mvpMapPoints;
    std::vector<std::vector<std::vector<size_t>>> mGrid;  // [col][row] -> keypoint indices
    std::vector<std::vector<std::vector<size_t>>> GetFeatureGrids() const { return mGrid; }
    std::shared_ptr<MapPoint> GetMapPoint(size_t i) const { return mvpMapPoints[i]; }
    std::vector<std::shared_ptr<MapPoint>> GetMapPointMatches() const { return mvpMapPoints; }
    int GetNumberMPs() const { return numMps; }
};
}  // namespace ORB_SLAM3

template <class T>
static std::vector<T> rd(FILE* f, size_t n) {
    std::vector<T> v(n);
    if (n && fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(3); }
    return v;
}

int main(int argc, char** argv) {
    using namespace ORB_SLAM3;
    if (argc < 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 3;
    const auto hdr = rd<int>(f, 7);  // nWindow, nSlots, nPoints, nObs, nKf, N, gridRows
    const int nW = hdr[0], nS = hdr[1], nP = hdr[2], nO = hdr[3], nK = hdr[4], N = hdr[5], gridRows = hdr[6];
    const auto ksb = rd<int>(f, nW + 1), sp = rd<int>(f, nS), sc = rd<int>(f, nS), nobs = rd<int>(f, nP), ob = rd<int>(f, nP + 1),
               ok = rd<int>(f, nO), nmps = rd<int>(f, nK), window_ids = rd<int>(f, nW), slot_true = rd<int>(f, nS);
    const auto badp = rd<unsigned char>(f, nP);
    fclose(f);
    std::vector<std::shared_ptr<KeyFrame>> kfs(nK);
    for (int k = 0; k < nK; k++) { kfs[k] = std::make_shared<KeyFrame>(); kfs[k]->id = k; kfs[k]->numMps = nmps[k]; }
    std::vector<std::shared_ptr<MapPoint>> pts(nP);
    for (int p = 0; p < nP; p++) {
        pts[p] = std::make_shared<MapPoint>();
        pts[p]->id = p; pts[p]->nObs = nobs[p]; pts[p]->mbBad = badp[p];
        for (int o = ob[p]; o < ob[p + 1]; o++) pts[p]->mObservations[kfs[ok[o]]] = std::make_tuple(0, -1);
    }
    const int gridCols = 64;
    std::vector<std::shared_ptr<KeyFrame>> vpKFs;
    for (int w = 0; w < nW; w++) {
        auto& kf = kfs[window_ids[w]];
        kf->mGrid.assign(gridCols, std::vector<std::vector<size_t>>(gridRows));
        for (int s = ksb[w]; s < ksb[w + 1]; s++) {
            const size_t idx = kf->mvpMapPoints.size();
            // slot_true = the map point the keypoint really holds (-1 none); bad ones are held but isBad()
            kf->mvpMapPoints.push_back(slot_true[s] >= 0 ? pts[slot_true[s]] : nullptr);
            kf->mGrid[sc[s] / gridRows][sc[s] % gridRows].push_back(idx);
        }
        vpKFs.push_back(kf);
    }
    const auto cm = msorb_host::BuildConstraintMatrix(vpKFs, 7, N);
    FILE* o = fopen(argv[2], "wb");
    const int nnz = (int)cm.colIdx.size();
    fwrite(&cm.nCols, 4, 1, o); fwrite(&cm.nRows, 4, 1, o); fwrite(&nnz, 4, 1, o); fwrite(&cm.nMaxObservation, 4, 1, o);
    for (int c = 0; c < cm.nCols; c++) { const int id = cm.colPoint[c]->id; fwrite(&id, 4, 1, o); }
    fwrite(cm.objCoef.data(), 4, cm.nCols, o);
    fwrite(cm.rowBegin.data(), 4, cm.nRows + 1, o);
    fwrite(cm.rowKind.data(), 4, cm.nRows, o);
    for (int r = 0; r < cm.nRows; r++) { const int id = cm.rowKeyFrame[r]->id; fwrite(&id, 4, 1, o); }
    fwrite(cm.rowCell.data(), 4, cm.nRows, o);
    fwrite(cm.rowRhs.data(), 4, cm.nRows, o);
    fwrite(cm.colIdx.data(), 4, nnz, o);
    int side_ok = 1;  // the side effects the rest of Sparsifying reads
    for (int c = 0; c < cm.nCols; c++) side_ok &= cm.colPoint[c]->mnMapSparsificationId == 7 && (int)cm.colPoint[c]->mnIndexForSparsification == c;
    for (auto& kf : vpKFs) side_ok &= kf->mnMapSaprsificationId == 7;
    fwrite(&side_ok, 4, 1, o);
    fclose(o);
    return 0;
}
