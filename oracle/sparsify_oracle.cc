// ORACLE — TEST INFRASTRUCTURE ONLY (see cvprims.h header).
//
// CPU restatement of the constraint-matrix assembly inside MapSparsification::Sparsifying
// (/root/reference/src/MapSparsification.cc:58-151) on flat stand-in arrays.  The GUROBI solve (:153-166)
// is out of scope (proprietary, absent).  Inputs are what the reference reads through
// KeyFrame::GetFeatureGrids / GetMapPoint / MapPoint::isBad / Observations / GetObservations /
// KeyFrame::GetNumberMPs, flattened by the caller:
//   window KF k (vpKFs order) owns slots [kf_slot_begin[k], kf_slot_begin[k+1]) in the reference's grid walk
//   order (grid column, then grid row, then the cell's index list, :82-84); slot_point = map point id or -1
//   (null / bad), slot_cell = col*rows+row.
// Outputs: columns = map points in first-encounter order (:90-99); rows in the order the reference adds
// them: per window KF its valid cells (kind 0, rhs 1, :111-116) then the KF row (kind 1, rhs N, :119-122);
// then one row per outside KF that observes a column point (kind 2, rhs count/total*N, :125-151).  The
// reference iterates those through a std::map keyed by shared_ptr (pointer order, run dependent); they are
// emitted here in ascending KF id.  CSR entries are column indices in the order the terms are added; every
// row additionally owns one slack variable (th_grid / th) that is not listed.
#include <cstdint>
#include <cstddef>
#include <map>
#include <vector>

extern "C" int orc_visibility_csr(int n_window_kf, const int* kf_slot_begin, const int* slot_point, const int* slot_cell,
                                  int n_points, const int* point_nobs, const int* obs_begin, const int* obs_kf,
                                  int n_kf_total, const uint8_t* kf_in_window, const int* kf_num_mps, int N,
                                  int n_max_obs_floor, int* n_cols, int* col_point, int* n_rows, int* row_begin,
                                  int* row_kind, int* row_owner, float* row_rhs, int* col_idx, float* obj_coef,
                                  int* n_max_obs) {
    (void)n_kf_total;
    // pass 1: nMaxObsevation (:66-76)
    int nMax = n_max_obs_floor;
    for (int s = 0; s < kf_slot_begin[n_window_kf]; s++) {
        const int p = slot_point[s];
        if (p < 0) continue;
        if (point_nobs[p] > nMax) nMax = point_nobs[p];
    }
    *n_max_obs = nMax;
    std::vector<int> index_of(n_points, -1);  // mnIndexForSparsification (valid when mnMapSparsificationId == mnId)
    std::vector<int> local;                   // vLocalMapPoints
    int rows = 0, nnz = 0;
    row_begin[0] = 0;
    for (int k = 0; k < n_window_kf; k++) {  // :78-123
        std::vector<int> kf_terms;
        int s = kf_slot_begin[k];
        const int e = kf_slot_begin[k + 1];
        while (s < e) {
            const int cell = slot_cell[s];
            std::vector<int> cell_terms;
            bool valid_cell = false;
            for (; s < e && slot_cell[s] == cell; s++) {
                const int p = slot_point[s];
                if (p < 0) continue;
                if (index_of[p] < 0) {
                    index_of[p] = (int)local.size();
                    local.push_back(p);
                }
                kf_terms.push_back(index_of[p]);
                cell_terms.push_back(index_of[p]);
                valid_cell = true;
            }
            if (valid_cell) {
                for (int c : cell_terms) col_idx[nnz++] = c;
                row_kind[rows] = 0; row_owner[rows] = cell; row_rhs[rows] = 1.0f;
                row_begin[++rows] = nnz;
            }
        }
        for (int c : kf_terms) col_idx[nnz++] = c;
        row_kind[rows] = 1; row_owner[rows] = k; row_rhs[rows] = (float)N;
        row_begin[++rows] = nnz;
    }
    *n_cols = (int)local.size();
    for (size_t c = 0; c < local.size(); c++) {
        col_point[c] = local[c];
        obj_coef[c] = (float)(nMax - point_nobs[local[c]]);  // :95-96
    }
    std::map<int, std::vector<int>> extra;  // extraNum + extraConstrints keyed by KF (:125-142)
    for (size_t c = 0; c < local.size(); c++) {
        const int p = local[c];
        for (int o = obs_begin[p]; o < obs_begin[p + 1]; o++)
            if (!kf_in_window[obs_kf[o]]) extra[obs_kf[o]].push_back((int)c);
    }
    for (auto& it : extra) {  // :144-151
        const float nTotal = kf_num_mps[it.first];
        const float nMini = (float)it.second.size() / nTotal * N;
        for (int c : it.second) col_idx[nnz++] = c;
        row_kind[rows] = 2; row_owner[rows] = it.first; row_rhs[rows] = nMini;
        row_begin[++rows] = nnz;
    }
    *n_rows = rows;
    return nnz;
}
