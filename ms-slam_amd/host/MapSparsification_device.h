// Constraint-matrix assembly of MapSparsification::Sparsifying (src/MapSparsification.cc:58-151) on the device,
// written against the reference's own types by name (templates: compiles inside MS-SLAM with the real KeyFrame /
// MapPoint and, in tests/dropin_sparsify_main.cc, with minimal stand-ins that have the same members).
//
// Inside the reference the three loops :67-151 become
//     auto cm = msorb_host::BuildConstraintMatrix(vpKFs, mnId, mnMinNum);
// followed by GUROBI's C++ or C API on cm: one binary variable per column (objective
// cm.objCoef[c]), one slack per row (kind 0: binary, cost mfGridLambda; kinds 1, 2: integer 0..1000, cost mfLambda),
// constraint r:  sum_{c in row r} x_c + slack_r >= cm.rowRhs[r].  The side effects the rest of Sparsifying relies on are
// reproduced: KeyFrame::mnMapSaprsificationId, MapPoint::mnMapSparsificationId / mnIndexForSparsification, and
// cm.colPoint == vLocalMapPoints (same order).
#ifndef MSORB_MAPSPARSIFICATION_DEVICE_H
#define MSORB_MAPSPARSIFICATION_DEVICE_H

#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "msorb.h"

namespace ORB_SLAM3 {
namespace msorb_host {
#ifndef MSORB_HOST_FAIL_CALL
#define MSORB_HOST_FAIL_CALL
// a failed call of the C ABI: the application's fatal-error callback first (msorb_set_fatal_callback), then std::runtime_error
[[noreturn]] inline void fail_call(const char* what) {
    const std::string msg = std::string(what) + ": " + msorb_last_error();
    msorb_notify_fatal(MSORB_E_HIP, msg.c_str());
    throw std::runtime_error(msg);
}
#endif

template <class KeyFrameT, class MapPointT>
struct ConstraintMatrix {
    int nCols = 0, nRows = 0, nMaxObservation = 0;
    std::vector<std::shared_ptr<MapPointT>> colPoint;    // vLocalMapPoints
    std::vector<float> objCoef;                          // nMaxObsevation - Observations()
    std::vector<int> rowBegin, rowKind, colIdx;          // CSR; kind 0 = grid cell, 1 = window keyframe, 2 = outside keyframe
    std::vector<float> rowRhs;
    std::vector<std::shared_ptr<KeyFrameT>> rowKeyFrame; // owner of the row (kind 0: the cell's keyframe)
    std::vector<int> rowCell;                            // kind 0: grid column * rows + grid row, else -1
};

template <class KeyFrameT, class MapPointT = typename std::remove_reference<decltype(*std::declval<KeyFrameT>().GetMapPoint(0))>::type>
ConstraintMatrix<KeyFrameT, MapPointT> BuildConstraintMatrix(const std::vector<std::shared_ptr<KeyFrameT>>& vpKFs,
                                                             long unsigned int mnId, int mnMinNum, int device = 0) {
    using KFp = std::shared_ptr<KeyFrameT>;
    using MPp = std::shared_ptr<MapPointT>;
    std::unordered_map<const MapPointT*, int> pid;
    std::unordered_map<const KeyFrameT*, int> kid;
    std::vector<MPp> points;
    std::vector<KFp> kfs;
    auto kf_id = [&](const KFp& k) {
        auto it = kid.find(k.get());
        if (it != kid.end()) return it->second;
        kid.emplace(k.get(), (int)kfs.size());
        kfs.push_back(k);
        return (int)kfs.size() - 1;
    };
    // pass 1 (:66-76): max Observations() over every valid map point of the window keyframes (also those in no cell)
    int floorObs = 0;
    for (const KFp& kf : vpKFs) {
        kf_id(kf);
        for (const MPp& p : kf->GetMapPointMatches())
            if (p && !p->isBad() && p->Observations() > floorObs) floorObs = p->Observations();
    }
    // slots in the walk order of :80-84 (grid column, grid row, the cell's index list)
    std::vector<int> kfSlotBegin(1, 0), slotPoint, slotCell;
    for (const KFp& kf : vpKFs) {
        const auto grids = kf->GetFeatureGrids();
        kf->mnMapSaprsificationId = mnId;
        int cell = 0;
        for (const auto& gridi : grids)
            for (const auto& grid : gridi) {
                for (size_t i : grid) {
                    const MPp p = kf->GetMapPoint(i);
                    int id = -1;
                    if (p && !p->isBad()) {
                        auto it = pid.find(p.get());
                        if (it == pid.end()) { id = (int)points.size(); pid.emplace(p.get(), id); points.push_back(p); }
                        else id = it->second;
                    }
                    slotPoint.push_back(id);
                    slotCell.push_back(cell);
                }
                cell++;
            }
        kfSlotBegin.push_back((int)slotPoint.size());
    }
    // observations of the column points (:127-142) and the per-keyframe totals (:146)
    const int nP = (int)points.size();
    std::vector<int> pointNobs(nP), obsBegin(1, 0), obsKf;
    for (int p = 0; p < nP; p++) {
        pointNobs[p] = points[p]->Observations();
        for (const auto& ob : points[p]->GetObservations()) obsKf.push_back(kf_id(ob.first));
        obsBegin.push_back((int)obsKf.size());
    }
    const int nK = (int)kfs.size();
    std::vector<uint8_t> inWindow(nK, 0);
    std::vector<int> numMps(nK, 1);
    for (int k = 0; k < nK; k++) {
        inWindow[k] = kfs[k]->mnMapSaprsificationId == mnId;
        if (!inWindow[k]) numMps[k] = kfs[k]->GetNumberMPs();
    }
    const int S = (int)slotPoint.size();
    const int capCols = S + 1, capRows = S + (int)vpKFs.size() + nK + 1, capNnz = 2 * S + (int)obsKf.size() + 1;
    ConstraintMatrix<KeyFrameT, MapPointT> cm;
    std::vector<int> colPoint(capCols), rowOwner(capRows);
    cm.objCoef.resize(capCols); cm.rowBegin.resize(capRows + 1); cm.rowKind.resize(capRows); cm.rowRhs.resize(capRows);
    cm.colIdx.resize(capNnz);
    int nnz = 0;
    const int rc = msorb_visibility_csr(device, (int)vpKFs.size(), kfSlotBegin.data(), slotPoint.data(), slotCell.data(), nP,
                                        pointNobs.data(), obsBegin.data(), obsKf.data(), nK, inWindow.data(), numMps.data(),
                                        mnMinNum, floorObs, &cm.nCols, colPoint.data(), capCols, &cm.nRows, cm.rowBegin.data(),
                                        cm.rowKind.data(), rowOwner.data(), cm.rowRhs.data(), capRows, cm.colIdx.data(), capNnz,
                                        &nnz, cm.objCoef.data(), &cm.nMaxObservation);
    if (rc != MSORB_OK) fail_call("msorb_visibility_csr");
    cm.objCoef.resize(cm.nCols); cm.rowBegin.resize(cm.nRows + 1); cm.rowKind.resize(cm.nRows); cm.rowRhs.resize(cm.nRows);
    cm.colIdx.resize(nnz);
    cm.colPoint.resize(cm.nCols);
    for (int c = 0; c < cm.nCols; c++) {
        cm.colPoint[c] = points[colPoint[c]];
        cm.colPoint[c]->mnMapSparsificationId = mnId;        // :91-96
        cm.colPoint[c]->mnIndexForSparsification = c;
    }
    // row owners: a keyframe's cell rows (kind 0, owner = cell id) come right before its own row (kind 1, owner = index in vpKFs)
    cm.rowKeyFrame.resize(cm.nRows);
    cm.rowCell.assign(cm.nRows, -1);
    int firstCellRow = 0;
    for (int r = 0; r < cm.nRows; r++) {
        if (cm.rowKind[r] == 0) { cm.rowCell[r] = rowOwner[r]; continue; }
        if (cm.rowKind[r] == 1) {
            for (int q = firstCellRow; q <= r; q++) cm.rowKeyFrame[q] = vpKFs[rowOwner[r]];
            firstCellRow = r + 1;
        } else {
            cm.rowKeyFrame[r] = kfs[rowOwner[r]];
            firstCellRow = r + 1;
        }
    }
    return cm;
}

}  // namespace msorb_host
}  // namespace ORB_SLAM3

#endif
