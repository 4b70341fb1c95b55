"""configs[4]: the constraint matrix of one sparsification window."""
import os
import sys
import time

import numpy as np

from . import ROOT, KITTI_MB, KITTI_MBF, self_check, oracle_module, tests_dir


def sparsification_leg(msorb, cpu):
    """BASELINE configs[4]: the per-window constraint-matrix build of MapSparsification::Sparsifying (MapSparsification.cc:58-151)
    on a 4Seasons-like sliding window — 30 keyframes x 2000 slots, half of them tracked, 6000 map points, 100 keyframes outside
    the window —: msorb_visibility_csr through the C ABI (host arrays in, CSR out), every buffer prepared once."""
    import ctypes as C
    tests_dir()
    import sparsify_cases as sc
    w = sc.window(11)
    L = msorb.lib()
    arrs = {k: np.ascontiguousarray(w[k], np.uint8 if k == "kf_in_window" else np.int32) for k in
            ("kf_slot_begin", "slot_point", "slot_cell", "point_nobs", "obs_begin", "obs_kf", "kf_in_window", "kf_num_mps")}
    K, S, P, KT = len(arrs["kf_slot_begin"]) - 1, len(arrs["slot_point"]), len(arrs["point_nobs"]), len(arrs["kf_in_window"])
    cc, cr, cn = S + 1, S + K + KT + 1, 2 * S + len(arrs["obs_kf"]) + 1
    col_point, obj = np.zeros(cc, np.int32), np.zeros(cc, np.float32)
    row_begin, row_kind, row_owner, row_rhs = np.zeros(cr + 1, np.int32), np.zeros(cr, np.int32), np.zeros(cr, np.int32), np.zeros(cr, np.float32)
    col_idx = np.zeros(cn, np.int32)
    n_cols, n_rows, nnz, nmax = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    L.msorb_visibility_csr.argtypes = ([C.c_int, C.c_int] + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int] +
                                       [C.c_void_p] * 2 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int] +
                                       [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 3)
    call = (0, K, p(arrs["kf_slot_begin"]), p(arrs["slot_point"]), p(arrs["slot_cell"]), P, p(arrs["point_nobs"]), p(arrs["obs_begin"]),
            p(arrs["obs_kf"]), KT, p(arrs["kf_in_window"]), p(arrs["kf_num_mps"]), 100, 0, C.byref(n_cols), p(col_point), cc, C.byref(n_rows),
            p(row_begin), p(row_kind), p(row_owner), p(row_rhs), cr, p(col_idx), cn, C.byref(nnz), p(obj), C.byref(nmax))
    ts = []
    for i in range(45):
        t0 = time.perf_counter()
        rc = L.msorb_visibility_csr(*call)
        if i >= 5:
            ts.append(time.perf_counter() - t0)
        if rc:
            raise RuntimeError("msorb_visibility_csr: %d" % rc)
    out = {"what": "configs[4]: constraint matrix of one sparsification window (30 keyframes x 2000 slots, 64x48 grid, 100 outside "
                   "keyframes) as CSR, msorb_visibility_csr through the C ABI, host arrays in and out",
           "ms_per_window": round(float(np.median(ts)) * 1e3, 4), "slots": S, "observations": int(len(arrs["obs_kf"])),
           "cols": n_cols.value, "rows": n_rows.value, "nnz": nnz.value}
    if cpu:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import orb_oracle
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            want = orb_oracle.visibility_csr(N=100, **w)
        dt = (time.perf_counter() - t0) / reps
        nr = n_rows.value
        same = (want["n_cols"] == n_cols.value and want["n_rows"] == nr and np.array_equal(want["col_point"], col_point[:n_cols.value]) and
                np.array_equal(want["row_begin"], row_begin[:nr + 1]) and np.array_equal(want["col_idx"], col_idx[:nnz.value]))
        self_check(same, "sparsification: msorb_visibility_csr differs from the CPU oracle")
        out["cpu_baseline"] = {"ms_per_window": round(dt * 1e3, 4), "cores": 1, "kind": "port",
                               "sample": f"oracle/sparsify_oracle.cc on the same window, {reps} repetitions", "gpu_matches_cpu": bool(same)}
    return out
