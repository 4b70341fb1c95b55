"""ctypes loader for the CPU oracle (TEST INFRASTRUCTURE ONLY — see oracle/cvprims.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28

_u8p = C.POINTER(C.c_uint8)
_ip = C.POINTER(C.c_int)
_fp = C.POINTER(C.c_float)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liborb_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_extractor_create.restype = C.c_void_p
        L.orc_extractor_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        L.orc_extractor_destroy.argtypes = [C.c_void_p]
        L.orc_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                  C.c_void_p, C.c_int, _ip]
        L.orc_tables.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.orc_level_size.argtypes = [C.c_void_p, C.c_int, _ip, _ip]
        L.orc_level_copy.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_candidates.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_selected.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_cell_stats.argtypes = [C.c_void_p, C.c_int, _ip, _ip, _ip]
        L.orc_resize_linear_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.orc_gaussian7.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_fast9_nms.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_int]
        L.orc_fast_atan2.restype = C.c_float
        L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.orc_cos_sin.argtypes = [C.c_float, _fp, _fp]
        L.orc_distribute_quadtree.argtypes = [C.c_void_p] * 3 + [C.c_int] * 6 + [C.c_void_p] * 3 + [C.c_int]
        _LIB = L
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleExtractor:
    """Mirror of ORB_SLAM3::ORBextractor (include/ORBextractor.h:43-109) on the CPU oracle."""

    def __init__(self, nfeatures=2000, scale=1.2, nlevels=8, ini_th=20, min_th=7):
        self.L = lib()
        self.nfeatures, self.nlevels = nfeatures, nlevels
        self.h = self.L.orc_extractor_create(nfeatures, scale, nlevels, ini_th, min_th)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_extractor_destroy(self.h)
            self.h = None

    def tables(self):
        n = self.nlevels
        sc, isc, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        per = np.zeros(n, np.int32)
        umax = np.zeros(16, np.int32)
        self.L.orc_tables(self.h, _ptr(sc), _ptr(isc), _ptr(s2), _ptr(is2), _ptr(per), _ptr(umax))
        return dict(scale=sc, inv_scale=isc, sigma2=s2, inv_sigma2=is2, per_level=per, umax=umax)

    def __call__(self, img, lapping=(0, 0)):
        """-> (mono_index, keypoints[KP_DTYPE], descriptors[n,32] u8)"""
        img = np.ascontiguousarray(img, np.uint8)
        # a level's tree can end above its quota by up to 3 nodes per initial column: the library's own bound is nfeatures + 19 per level
        cap = self.nfeatures + 19 * self.nlevels + 64
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        rows, cols = (img.shape if img.size else (0, 0))
        mono = self.L.orc_extract(self.h, _ptr(img), rows, cols, cols, lapping[0], lapping[1], _ptr(kps), _ptr(desc),
                                  cap, C.byref(n))
        return mono, kps[:n.value].copy(), desc[:n.value].copy()

    def level(self, l, blurred=False):
        r, c = C.c_int(), C.c_int()
        assert self.L.orc_level_size(self.h, l, C.byref(r), C.byref(c)) == 0
        out = np.zeros((r.value, c.value), np.uint8)
        assert self.L.orc_level_copy(self.h, l, int(blurred), _ptr(out)) == 0
        return out

    def candidates(self, l):
        cap = 1 << 20
        xs, ys, sc = (np.zeros(cap, np.int32) for _ in range(3))
        n = self.L.orc_candidates(self.h, l, _ptr(xs), _ptr(ys), _ptr(sc), cap)
        return np.stack([xs[:n], ys[:n], sc[:n]], 1)

    def cell_stats(self):
        """per level of the last call: (cells visited, cells that took the minThFAST retry, cells empty after it)"""
        out = []
        for l in range(self.nlevels):
            t, r, e = C.c_int(), C.c_int(), C.c_int()
            assert self.L.orc_cell_stats(self.h, l, C.byref(t), C.byref(r), C.byref(e)) == 0
            out.append((t.value, r.value, e.value))
        return out

    def selected(self, l):
        cap = self.nfeatures + 64
        kps = np.zeros(cap, KP_DTYPE)
        n = self.L.orc_selected(self.h, l, _ptr(kps), cap)
        return kps[:n].copy()


def resize_linear_u8(src, drows, dcols):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((drows, dcols), np.uint8)
    lib().orc_resize_linear_u8(_ptr(src), src.shape[0], src.shape[1], _ptr(dst), drows, dcols)
    return dst


def gaussian7(src):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros_like(src)
    lib().orc_gaussian7(_ptr(src), src.shape[0], src.shape[1], _ptr(dst))
    return dst


def fast9_nms(roi, threshold):
    roi = np.ascontiguousarray(roi, np.uint8)
    cap = roi.size
    xs, ys, sc = (np.zeros(cap, np.int32) for _ in range(3))
    n = lib().orc_fast9_nms(_ptr(roi), roi.shape[1], roi.shape[0], roi.shape[1], threshold, _ptr(xs), _ptr(ys),
                            _ptr(sc), cap)
    return np.stack([xs[:n], ys[:n], sc[:n]], 1)


def fast9_planes(roi, threshold):
    """Segment-test mask and cornerScore plane of a ROI before NMS -> (corner u8 [rows, cols], score int32 [rows, cols])."""
    roi = np.ascontiguousarray(roi, np.uint8)
    corner = np.zeros(roi.shape, np.uint8)
    score = np.zeros(roi.shape, np.int32)
    L = lib()
    L.orc_fast9_planes.restype = None
    L.orc_fast9_planes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.orc_fast9_planes(_ptr(roi), roi.shape[1], roi.shape[0], roi.shape[1], int(threshold), _ptr(corner), _ptr(score))
    return corner, score


def set_semantics(gauss_taps=None, resize_single_stage=False, atan2_fma=False, brief_tap=0):
    """The [OpenCV-recall] variant table of oracle/cvprims.h (process-wide).  set_semantics() restores the defaults."""
    L = lib()
    L.orc_set_semantics.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    if gauss_taps is None and not resize_single_stage and not atan2_fma and not brief_tap:
        rc = L.orc_set_semantics(None, 0, 0, 0)
    else:
        t = np.ascontiguousarray(gauss_taps if gauss_taps is not None else [18, 34, 48, 56, 48, 34, 18], np.int32)
        assert len(t) == 7
        rc = L.orc_set_semantics(_ptr(t), int(resize_single_stage), int(atan2_fma), int(brief_tap))
    if rc:
        raise ValueError("taps outside the Q8.8 pipeline's range")


def fast_atan2(y, x):
    return float(lib().orc_fast_atan2(float(y), float(x)))


def fast_atan2_n(y, x):
    """fast_atan2 over arrays (float32 in, float32 degrees out)."""
    y, x = np.ascontiguousarray(y, np.float32).ravel(), np.ascontiguousarray(x, np.float32).ravel()
    out = np.empty(len(y), np.float32)
    L = lib()
    L.orc_fast_atan2_n.restype = None
    L.orc_fast_atan2_n.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]
    L.orc_fast_atan2_n(_ptr(y), _ptr(x), len(y), _ptr(out))
    return out


def rotated_tap(mode, x, y, a, b):
    """(row, col) = cvRound(x*b + y*a), cvRound(x*a - y*b) (ORBextractor.cc:117-119) under contraction `mode` (Semantics::brief_tap)."""
    L = lib()
    L.orc_rotated_tap.restype = None
    L.orc_rotated_tap.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    r, c = C.c_int(), C.c_int()
    L.orc_rotated_tap(int(mode), int(x), int(y), float(a), float(b), C.byref(r), C.byref(c))
    return r.value, c.value


def brief_tap_sweep(angles_deg, cap=4096):
    """Exposure of the descriptor to the tap contraction over `angles_deg` x the 512 pattern points: dict with the number of
    (point, angle) pairs evaluated, the pairs / angles that land on another pixel under contraction 1 or 2 than under 0
    (`vs0`), under 1 than under 2 (`between_1_2`), and up to `cap` examples (angle bits, point, bit mask of the modes that differ)."""
    ang = np.ascontiguousarray(angles_deg, np.float32)
    L = lib()
    L.orc_brief_tap_sweep.restype = C.c_longlong
    L.orc_brief_tap_sweep.argtypes = [C.c_void_p, C.c_longlong] + [C.c_void_p] * 5 + [C.c_int, C.c_void_p]
    flips, fang = np.zeros(3, np.int64), np.zeros(3, np.int64)
    ex_a, ex_p, ex_m = np.zeros(cap, np.uint32), np.zeros(cap, np.int32), np.zeros(cap, np.int32)
    n_ex = C.c_int()
    pairs = L.orc_brief_tap_sweep(_ptr(ang), len(ang), _ptr(flips), _ptr(fang), _ptr(ex_a), _ptr(ex_p), _ptr(ex_m), cap, C.byref(n_ex))
    k = n_ex.value
    return {"pairs": int(pairs), "angles": int(len(ang)),
            "flips_vs0": {1: int(flips[1]), 2: int(flips[2])}, "flips_between_1_2": int(flips[0]),
            "angles_vs0": {1: int(fang[1]), 2: int(fang[2])}, "angles_between_1_2": int(fang[0]),
            "examples": list(zip(ex_a[:k].tolist(), ex_p[:k].tolist(), ex_m[:k].tolist()))}


def cos_sin(angle_deg):
    a, b = C.c_float(), C.c_float()
    lib().orc_cos_sin(float(angle_deg), C.byref(a), C.byref(b))
    return a.value, b.value


_RIG_FIELDS = (("track_in_view", np.uint8), ("track_in_view_r", np.uint8), ("bad", np.uint8), ("sparsified", np.uint8), ("proj_x", np.float32),
               ("proj_y", np.float32), ("proj_xr", np.float32), ("proj_yr", np.float32), ("track_depth", np.float32), ("level", np.int32),
               ("level_r", np.int32), ("view_cos", np.float32), ("view_cos_r", np.float32), ("desc", np.uint8), ("obs", np.int32))


def search_by_projection_mps_rig(left, right, mp, left_to_right, right_to_left, frame_mp, th, bFarPoints=False, thFarPoints=50.0, nnratio=0.8):
    """orc_search_by_projection_mps_rig (ORBmatcher.cc:43-213, F.Nleft != -1); left / right: OracleFrame of the two cameras."""
    L = lib()
    arrs = [_c(mp[k], dt) for k, dt in _RIG_FIELDS] + [_c(left_to_right, np.int32), _c(right_to_left, np.int32)]
    L.orc_search_by_projection_mps_rig.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 18 + [C.c_float, C.c_int, C.c_float, C.c_float]
    return L.orc_search_by_projection_mps_rig(left.h, right.h, len(arrs[0]), *[_ptr(a) for a in arrs], _ptr(frame_mp), th, int(bFarPoints),
                                              thFarPoints, nnratio)


def search_by_projection_frames_rig(left, right, last, cur_mp, th, forward=False, backward=False, check_orientation=True):
    """orc_search_by_projection_frames_rig (ORBmatcher.cc:1941-2152 on a two-camera CurrentFrame); last: dict valid, u, v, u_r, v_r,
    octave, angle, desc, mp, obs; cur_mp int32 [n_left + n_right] in / out."""
    L = lib()
    arrs = [_c(last["valid"], np.uint8), _c(last["u"], np.float32), _c(last["v"], np.float32), _c(last["u_r"], np.float32),
            _c(last["v_r"], np.float32), _c(last["octave"], np.int32), _c(last["angle"], np.float32), _c(last["desc"], np.uint8),
            _c(last["mp"], np.int32), _c(last["obs"], np.int32)]
    L.orc_search_by_projection_frames_rig.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 11 + [C.c_float, C.c_int, C.c_int, C.c_int]
    return L.orc_search_by_projection_frames_rig(left.h, right.h, len(arrs[0]), *[_ptr(a) for a in arrs], _ptr(cur_mp), th, int(forward),
                                                 int(backward), int(check_orientation))


def std_sort_order(keys):
    """libstdc++ std::sort on (key, position) items compared by key only -> the input position of the item at each output position."""
    keys = np.ascontiguousarray(keys, np.uint32)
    order = np.zeros(len(keys), np.uint32)
    L = lib()
    L.orc_std_sort_order.restype = None
    L.orc_std_sort_order.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.orc_std_sort_order(_ptr(keys), len(keys), _ptr(order))
    return order


def distribute_quadtree(xs, ys, resp, minX, maxX, minY, maxY, N):
    xs, ys, resp = (np.ascontiguousarray(v, np.float32) for v in (xs, ys, resp))
    n = len(xs)
    cap = n + 8
    ox, oy, orr = (np.zeros(cap, np.float32) for _ in range(3))
    m = lib().orc_distribute_quadtree(_ptr(xs), _ptr(ys), _ptr(resp), n, minX, maxX, minY, maxY, N, _ptr(ox),
                                      _ptr(oy), _ptr(orr), cap)
    return np.stack([ox[:m], oy[:m], orr[:m]], 1)


# ------------------------------------------------------------------------------------------------
# matcher oracle (oracle/matcher_oracle.cc)
# ------------------------------------------------------------------------------------------------
_mready = False


def _mlib():
    global _mready
    L = lib()
    if not _mready:
        vp, ci, cf = C.c_void_p, C.c_int, C.c_float
        L.orc_descriptor_distance.argtypes = [vp, vp]
        L.orc_frame_create.restype = vp
        L.orc_frame_create.argtypes = [vp, ci, vp, vp, cf, cf, cf, cf, vp, ci]
        L.orc_frame_destroy.argtypes = [vp]
        L.orc_features_in_area.argtypes = [vp, cf, cf, cf, ci, ci, vp, ci]
        L.orc_search_by_projection_mps.argtypes = [vp, ci] + [vp] * 12 + [cf, ci, cf, cf]
        L.orc_search_by_projection_frames.argtypes = [vp, ci] + [vp] * 10 + [cf, ci, ci, ci]
        L.orc_three_maxima.argtypes = [vp, ci, vp]
        L.orc_compute_stereo_matches.argtypes = [vp, ci, vp, vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, cf, cf, vp, vp, vp]
        _mready = True
    return L


def _c(a, dt):
    return np.ascontiguousarray(a, dt)


def descriptor_distance(a, b):
    a, b = _c(a, np.uint8), _c(b, np.uint8)
    return _mlib().orc_descriptor_distance(_ptr(a), _ptr(b))


class OracleFrame:
    def __init__(self, keypoints, descriptors, u_right, bounds, scale_factors):
        self.L = _mlib()
        self.kps = _c(keypoints, KP_DTYPE)
        self.desc = _c(descriptors, np.uint8)
        self.n = len(self.kps)
        ur = None if u_right is None else _c(u_right, np.float32)
        self.u_right = np.full(self.n, -1.0, np.float32) if ur is None else ur
        sf = _c(scale_factors, np.float32)
        self.h = self.L.orc_frame_create(_ptr(self.kps), self.n, _ptr(self.desc), None if ur is None else _ptr(ur),
                                         bounds[0], bounds[1], bounds[2], bounds[3], _ptr(sf), len(sf))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_frame_destroy(self.h)
            self.h = None

    def grid_csr(self):
        """mGrid (Frame.cc:385-416) as CSR: (cell_begin[64*48+1], cell_idx), cell = ix*48 + iy, push_back order."""
        L = _mlib()
        cb = np.zeros(64 * 48 + 1, np.int32)
        ci = np.zeros(max(self.n, 1), np.int32)
        L.orc_frame_grid_csr.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        n = L.orc_frame_grid_csr(self.h, cb.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p))
        return cb, ci[:n].copy()

    def GetFeaturesInArea(self, x, y, r, minLevel=-1, maxLevel=-1):
        out = np.zeros(max(self.n, 1), np.int32)
        n = self.L.orc_features_in_area(self.h, x, y, r, minLevel, maxLevel, _ptr(out), len(out))
        return out[:n].copy()

    def SearchByProjection_mps(self, mp, frame_mp, th, bFarPoints=False, thFarPoints=50.0, nnratio=0.8):
        M = len(mp["proj_x"])
        arrs = [_c(mp["track_in_view"], np.uint8), _c(mp["bad"], np.uint8), _c(mp["sparsified"], np.uint8),
                _c(mp["proj_x"], np.float32), _c(mp["proj_y"], np.float32), _c(mp["proj_xr"], np.float32),
                _c(mp["track_depth"], np.float32), _c(mp["level"], np.int32), _c(mp["view_cos"], np.float32),
                _c(mp["desc"], np.uint8), _c(mp["obs"], np.int32)]
        return self.L.orc_search_by_projection_mps(self.h, M, *[_ptr(a) for a in arrs], _ptr(frame_mp), th,
                                                   int(bFarPoints), thFarPoints, nnratio)

    def SearchByProjection_frames(self, last, cur_mp, th, forward=False, backward=False, check_orientation=True):
        NL = len(last["u"])
        arrs = [_c(last["valid"], np.uint8), _c(last["u"], np.float32), _c(last["v"], np.float32),
                _c(last["ur"], np.float32), _c(last["octave"], np.int32), _c(last["angle"], np.float32),
                _c(last["desc"], np.uint8), _c(last["mp"], np.int32), _c(last["obs"], np.int32)]
        return self.L.orc_search_by_projection_frames(self.h, NL, *[_ptr(a) for a in arrs], _ptr(cur_mp), th,
                                                      int(forward), int(backward), int(check_orientation))

    def SearchByProjection_kf(self, pts, cur_mp, th, orb_dist, check_orientation=True):
        """orc_search_by_projection_kf (ORBmatcher.cc:2154-2275); pts: dict valid,u,v,level,angle,desc,mp"""
        arrs = [_c(pts["valid"], np.uint8), _c(pts["u"], np.float32), _c(pts["v"], np.float32), _c(pts["level"], np.int32),
                _c(pts["angle"], np.float32), _c(pts["desc"], np.uint8), _c(pts["mp"], np.int32)]
        self.L.orc_search_by_projection_kf.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 8 + [C.c_float, C.c_int, C.c_int]
        return self.L.orc_search_by_projection_kf(self.h, len(arrs[0]), *[_ptr(a) for a in arrs], _ptr(cur_mp), th,
                                                  int(orb_dist), int(check_orientation))

    def SearchByProjection_sim3(self, pts, matched, th, max_dist):
        """orc_search_by_projection_sim3 (ORBmatcher.cc:423-530); pts: dict valid,u,v,level,desc,mp"""
        arrs = [_c(pts["valid"], np.uint8), _c(pts["u"], np.float32), _c(pts["v"], np.float32), _c(pts["level"], np.int32),
                _c(pts["desc"], np.uint8), _c(pts["mp"], np.int32)]
        self.L.orc_search_by_projection_sim3.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 7 + [C.c_float, C.c_float]
        return self.L.orc_search_by_projection_sim3(self.h, len(arrs[0]), *[_ptr(a) for a in arrs], _ptr(matched), th, max_dist)

    def FuseSearch(self, inv_level_sigma2, valid, u, v, ur, predicted_level, radius, mp_desc):
        """orc_fuse_search (ORBmatcher.cc:1499-1561) -> (best_idx, best_dist)"""
        arrs = [_c(valid, np.uint8), _c(u, np.float32), _c(v, np.float32), _c(ur, np.float32), _c(predicted_level, np.int32),
                _c(radius, np.float32), _c(mp_desc, np.uint8)]
        n = len(arrs[0])
        inv = _c(inv_level_sigma2, np.float32)
        bi, bd = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32)
        self.L.orc_fuse_search.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 9
        self.L.orc_fuse_search.restype = None
        self.L.orc_fuse_search(self.h, _ptr(inv), n, *[_ptr(a) for a in arrs], _ptr(bi), _ptr(bd))
        return bi[:n], bd[:n]


    def FuseSearchGated(self, gate_kps, gate_uright, inv_level_sigma2, valid, u, v, ur, predicted_level, radius, mp_desc):
        """orc_fuse_search_gated (ORBmatcher.cc:1499-1561 with bRight = true: window on this — the right camera's — features, level band
        and error gate on gate_kps / gate_uright = pKF->GetKeyPoint(idx) / GetuRight(idx)) -> (best_idx, best_dist)"""
        arrs = [_c(valid, np.uint8), _c(u, np.float32), _c(v, np.float32), _c(ur, np.float32), _c(predicted_level, np.int32),
                _c(radius, np.float32), _c(mp_desc, np.uint8)]
        n = len(arrs[0])
        inv = _c(inv_level_sigma2, np.float32)
        gk = np.ascontiguousarray(gate_kps)
        gu = None if gate_uright is None else _c(gate_uright, np.float32)
        bi, bd = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32)
        self.L.orc_fuse_search_gated.argtypes = [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 9
        self.L.orc_fuse_search_gated.restype = None
        self.L.orc_fuse_search_gated(self.h, gk.ctypes.data_as(C.c_void_p), None if gu is None else _ptr(gu), _ptr(inv), n,
                                     *[_ptr(a) for a in arrs], _ptr(bi), _ptr(bd))
        return bi[:n], bd[:n]


def _sim3_side(p):
    return [_c(p["valid"], np.uint8), _c(p["u"], np.float32), _c(p["v"], np.float32), _c(p["level"], np.int32),
            _c(p["desc"], np.uint8)]


def search_by_sim3(kf1, kf2, p1, p2, th):
    """orc_search_by_sim3 (ORBmatcher.cc:1718-1939 from the projections on).  kf1 / kf2: OracleFrame; p1 / p2: dicts valid, u, v,
    level, desc of the map points of pKF1 / pKF2.  -> (match12[n1], nFound)"""
    L = _mlib()
    a, b = _sim3_side(p1), _sim3_side(p2)
    m12 = np.full(max(len(a[0]), 1), -1, np.int32)
    L.orc_search_by_sim3.argtypes = ([C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 5 +
                                     [C.c_float, C.c_void_p])
    n = L.orc_search_by_sim3(kf1.h, kf2.h, len(a[0]), *[_ptr(x) for x in a], len(b[0]), *[_ptr(x) for x in b], th, _ptr(m12))
    return m12[:len(a[0])], n


def search_by_projection_loop(kf, pts, train_ok, th, max_dist):
    """orc_search_by_projection_loop (ORBmatcher.cc:532-637).  -> (best_idx[n], nmatches)"""
    L = _mlib()
    a = _sim3_side(pts)
    n = len(a[0])
    ok = _c(train_ok, np.uint8)
    bi = np.full(max(n, 1), -1, np.int32)
    L.orc_search_by_projection_loop.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 6 + [C.c_float, C.c_float, C.c_void_p]
    nm = L.orc_search_by_projection_loop(kf.h, n, *[_ptr(x) for x in a], _ptr(ok), th, max_dist, _ptr(bi))
    return bi[:n], nm


def fuse_sim3_search(kf, pts, th):
    """orc_fuse_sim3_search (ORBmatcher.cc:1661-1696).  -> (best_idx, best_dist), INT_MAX = none"""
    L = _mlib()
    a = _sim3_side(pts)
    n = len(a[0])
    bi, bd = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32)
    L.orc_fuse_sim3_search.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_float, C.c_void_p, C.c_void_p]
    L.orc_fuse_sim3_search.restype = None
    L.orc_fuse_sim3_search(kf.h, n, *[_ptr(x) for x in a], th, _ptr(bi), _ptr(bd))
    return bi[:n], bd[:n]


def search_for_initialization(f1, f2, prev_xy, window_size=100, nnratio=0.9, check_orientation=True):
    """orc_search_for_initialization (ORBmatcher.cc:755-870).  prev_xy [N1, 2] float32 is updated in place.
    -> (vnMatches12, nmatches)"""
    L = _mlib()
    assert prev_xy.dtype == np.float32 and prev_xy.flags.c_contiguous and prev_xy.shape == (f1.n, 2)
    m12 = np.full(max(f1.n, 1), -1, np.int32)
    L.orc_search_for_initialization.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]
    n = L.orc_search_for_initialization(f1.h, f2.h, _ptr(prev_xy), int(window_size), nnratio, int(check_orientation), _ptr(m12))
    return m12[:f1.n], n


def three_maxima(sizes):
    s = _c(sizes, np.int32)
    ind = np.zeros(3, np.int32)
    _mlib().orc_three_maxima(_ptr(s), len(s), _ptr(ind))
    return ind


def compute_stereo_matches(kps_l, desc_l, kps_r, desc_r, pyr_l, pyr_r, scale, inv_scale, mb, mbf):
    """pyr_l / pyr_r: lists of contiguous uint8 planes (interior pixels). -> (mvuRight, mvDepth, n_oob)"""
    kl, kr = _c(kps_l, KP_DTYPE), _c(kps_r, KP_DTYPE)
    dl, dr = _c(desc_l, np.uint8), _c(desc_r, np.uint8)
    pl = [_c(p, np.uint8) for p in pyr_l]
    pr = [_c(p, np.uint8) for p in pyr_r]
    n = len(pl)
    PL = (C.c_void_p * n)(*[p.ctypes.data for p in pl])
    PR = (C.c_void_p * n)(*[p.ctypes.data for p in pr])
    rows = np.array([p.shape[0] for p in pl], np.int32)
    cols = np.array([p.shape[1] for p in pl], np.int32)
    strides = cols.copy()
    sc, isc = _c(scale, np.float32), _c(inv_scale, np.float32)
    ur = np.zeros(len(kl), np.float32)
    dp = np.zeros(len(kl), np.float32)
    oob = C.c_int()
    _mlib().orc_compute_stereo_matches(_ptr(kl), len(kl), _ptr(dl), _ptr(kr), len(kr), _ptr(dr), PL, PR, _ptr(rows),
                                       _ptr(cols), _ptr(strides), _ptr(sc), _ptr(isc), mb, mbf, _ptr(ur), _ptr(dp),
                                       C.byref(oob))
    return ur, dp, oob.value


def visibility_csr(kf_slot_begin, slot_point, slot_cell, point_nobs, obs_begin, obs_kf, kf_in_window, kf_num_mps, N,
                   n_max_obs_floor=0):
    """oracle/sparsify_oracle.cc — same outputs as msorb.visibility_csr."""
    L = lib()
    L.orc_visibility_csr.argtypes = ([C.c_int] + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int] +
                                     [C.c_void_p] * 2 + [C.c_int, C.c_int] + [C.c_void_p] * 10)
    ksb, sp, sc = _c(kf_slot_begin, np.int32), _c(slot_point, np.int32), _c(slot_cell, np.int32)
    pn, ob, ok = _c(point_nobs, np.int32), _c(obs_begin, np.int32), _c(obs_kf, np.int32)
    kw, km = _c(kf_in_window, np.uint8), _c(kf_num_mps, np.int32)
    K, S, KT = len(ksb) - 1, len(sp), len(kw)
    cap_rows, cap_nnz = S + K + KT + 1, 2 * S + len(ok) + 1
    col_point, obj = np.zeros(S + 1, np.int32), np.zeros(S + 1, np.float32)
    row_begin = np.zeros(cap_rows + 1, np.int32)
    row_kind, row_owner, row_rhs = np.zeros(cap_rows, np.int32), np.zeros(cap_rows, np.int32), np.zeros(cap_rows, np.float32)
    col_idx = np.zeros(cap_nnz, np.int32)
    n_cols, n_rows, nmax = C.c_int(), C.c_int(), C.c_int()
    nz = L.orc_visibility_csr(K, _ptr(ksb), _ptr(sp), _ptr(sc), len(pn), _ptr(pn), _ptr(ob), _ptr(ok), KT, _ptr(kw),
                              _ptr(km), N, n_max_obs_floor, C.byref(n_cols), _ptr(col_point), C.byref(n_rows),
                              _ptr(row_begin), _ptr(row_kind), _ptr(row_owner), _ptr(row_rhs), _ptr(col_idx), _ptr(obj),
                              C.byref(nmax))
    nc, nr = n_cols.value, n_rows.value
    return dict(n_cols=nc, col_point=col_point[:nc], obj_coef=obj[:nc], n_rows=nr, row_begin=row_begin[:nr + 1],
                row_kind=row_kind[:nr], row_owner=row_owner[:nr], row_rhs=row_rhs[:nr], col_idx=col_idx[:nz],
                n_max_obs=nmax.value)


class OracleVocabulary:
    """oracle/bow_oracle.cc — DBoW2 tree + transform() on the reference's own containers."""

    def __init__(self, k, L, scoring, weighting, parent, is_leaf, descriptors, weights):
        Lb = lib()
        Lb.orc_vocab_create.restype = C.c_void_p
        Lb.orc_vocab_create.argtypes = [C.c_int] * 5 + [C.c_void_p] * 4
        Lb.orc_vocab_destroy.argtypes = [C.c_void_p]
        Lb.orc_vocab_destroy.restype = None
        Lb.orc_bow_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 10
        par, lf = _c(parent, np.int32), _c(is_leaf, np.uint8)
        ds, ws = _c(descriptors, np.uint8), _c(weights, np.float64)
        self.h = Lb.orc_vocab_create(k, L, scoring, weighting, len(par), _ptr(par), _ptr(lf), _ptr(ds), _ptr(ws))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_vocab_destroy(self.h)
            self.h = None

    def transform(self, descriptors, levelsup=4):
        d = _c(descriptors, np.uint8).reshape(-1, 32)
        n = len(d)
        cap = max(n, 1)
        bw, bv = np.zeros(cap, np.int32), np.zeros(cap, np.float64)
        fn, fb, ff = np.zeros(cap, np.int32), np.zeros(cap + 1, np.int32), np.zeros(cap, np.int32)
        fw, fnode, fwt = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.float64)
        nb, nf = C.c_int(), C.c_int()
        lib().orc_bow_transform(self.h, _ptr(d), n, levelsup, _ptr(bw), _ptr(bv), C.addressof(nb), _ptr(fn), _ptr(fb),
                                _ptr(ff), C.addressof(nf), _ptr(fw), _ptr(fnode), _ptr(fwt))
        nb, nf = nb.value, nf.value
        return dict(bow_word=bw[:nb], bow_value=bv[:nb], fv_node=fn[:nf], fv_begin=fb[:nf + 1], fv_feat=ff[:fb[nf]],
                    feat_word=fw[:n], feat_node=fnode[:n], feat_weight=fwt[:n])


def distinctive_descriptors(descriptors, obs_begin):
    Lb = lib()
    Lb.orc_distinctive_descriptors.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    Lb.orc_distinctive_descriptors.restype = None
    d = _c(descriptors, np.uint8).reshape(-1, 32)
    ob = _c(obs_begin, np.int32)
    n = len(ob) - 1
    bi, bm = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32)
    Lb.orc_distinctive_descriptors(_ptr(d), _ptr(ob), n, _ptr(bi), _ptr(bm))
    return bi[:n], bm[:n]


def is_in_frustum(frustum, pos_w, normal, max_distance, min_distance, viewing_cos_limit=0.5):
    """oracle/frustum_oracle.cc; `frustum` is any ctypes struct with msorb_frustum's layout (msorb.Frustum)."""
    Lb = lib()
    Lb.orc_is_in_frustum.argtypes = [C.c_void_p, C.c_float, C.c_int] + [C.c_void_p] * 11
    Lb.orc_is_in_frustum.restype = None
    P, Nn = _c(pos_w, np.float32).reshape(-1, 3), _c(normal, np.float32).reshape(-1, 3)
    mx, mn = _c(max_distance, np.float32), _c(min_distance, np.float32)
    n = len(P)
    cap = max(n, 1)
    inv = np.zeros(cap, np.uint8)
    px, py, pxr, dep, vc = [np.zeros(cap, np.float32) for _ in range(5)]
    lvl = np.zeros(cap, np.int32)
    Lb.orc_is_in_frustum(C.addressof(frustum), viewing_cos_limit, n, _ptr(P), _ptr(Nn), _ptr(mx), _ptr(mn), _ptr(inv),
                         _ptr(px), _ptr(py), _ptr(pxr), _ptr(dep), _ptr(lvl), _ptr(vc))
    return dict(track_in_view=inv[:n], proj_x=px[:n], proj_y=py[:n], proj_xr=pxr[:n], track_depth=dep[:n], level=lvl[:n],
                view_cos=vc[:n])


class MotionModel(C.Structure):
    """layout of msorb_motion_model / orc_motion_model"""
    _fields_ = [("q", C.c_float * 4), ("t", C.c_float * 3), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float),
                ("cy", C.c_float), ("mbf", C.c_float), ("forward", C.c_int), ("backward", C.c_int)]


def project_last_frame(mm, bounds, has_point, pos_w):
    """oracle/frustum_oracle.cc orc_project_last_frame (ORBmatcher.cc:1951-1990, :2019) -> valid, u, v, ur.  `mm`: any ctypes
    struct with msorb_motion_model's layout; bounds = (mnMinX, mnMaxX, mnMinY, mnMaxY)."""
    Lb = lib()
    Lb.orc_project_last_frame.argtypes = [C.c_void_p] + [C.c_float] * 4 + [C.c_int] + [C.c_void_p] * 6
    Lb.orc_project_last_frame.restype = None
    hp, P = _c(has_point, np.uint8), _c(pos_w, np.float32).reshape(-1, 3)
    n = len(hp)
    cap = max(n, 1)
    valid = np.zeros(cap, np.uint8)
    u, v, ur = [np.zeros(cap, np.float32) for _ in range(3)]
    Lb.orc_project_last_frame(C.addressof(mm), *[float(b) for b in bounds], n, _ptr(hp), _ptr(P), _ptr(valid), _ptr(u), _ptr(v),
                              _ptr(ur))
    return valid[:n], u[:n], v[:n], ur[:n]


def dense_top2(q, t):
    """oracle/matcher_oracle.cc orc_dense_top2 -> (best_idx, best_dist, second_dist)"""
    Lb = _mlib()
    Lb.orc_dense_top2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    Lb.orc_dense_top2.restype = None
    qq, tt = _c(q, np.uint8).reshape(-1, 32), _c(t, np.uint8).reshape(-1, 32)
    n = len(qq)
    bi, bd, sd = (np.zeros(max(n, 1), np.int32) for _ in range(3))
    Lb.orc_dense_top2(_ptr(qq), n, _ptr(tt), len(tt), _ptr(bi), _ptr(bd), _ptr(sd))
    return bi[:n], bd[:n], sd[:n]


def search_by_bow(desc1, desc2, valid1, avail2, fv1, fv2, angle1, angle2, th_low=50, inclusive=True, nnratio=0.7,
                  check_orientation=True):
    """oracle/matcher_oracle.cc orc_search_by_bow (ORBmatcher.cc:223-421, 872-1166).  fv = (node, begin, feat) CSR.
    -> (nmatches, match12, match21)"""
    Lb = _mlib()
    Lb.orc_search_by_bow.argtypes = ([C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int] +
                                     [C.c_void_p] * 5 + [C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p])
    Lb.orc_search_by_bow.restype = C.c_int
    d1, d2 = _c(desc1, np.uint8).reshape(-1, 32), _c(desc2, np.uint8).reshape(-1, 32)
    n1, n2 = len(d1), len(d2)
    v1 = _c(valid1, np.uint8)
    a2 = None if avail2 is None else _c(avail2, np.uint8)
    f1 = [_c(a, np.int32) for a in fv1]
    f2 = [_c(a, np.int32) for a in fv2]
    g1, g2 = _c(angle1, np.float32), _c(angle2, np.float32)
    m12, m21 = np.zeros(max(n1, 1), np.int32), np.zeros(max(n2, 1), np.int32)
    nm = Lb.orc_search_by_bow(n1, n2, _ptr(d1), _ptr(d2), _ptr(v1), None if a2 is None else _ptr(a2), len(f1[0]),
                              _ptr(f1[0]), _ptr(f1[1]), _ptr(f1[2]), len(f2[0]), _ptr(f2[0]), _ptr(f2[1]), _ptr(f2[2]),
                              _ptr(g1), _ptr(g2), int(th_low), int(bool(inclusive)), float(nnratio),
                              int(bool(check_orientation)), _ptr(m12), _ptr(m21))
    return nm, m12[:n1], m21[:n2]


def search_by_bow_rig(p, n_left, th_low=50, nnratio=0.7, check_orientation=True):
    """orc_search_by_bow_rig (ORBmatcher.cc:223-421, F.Nleft != -1); p as for search_by_bow -> (nmatches, match21)"""
    Lb = _mlib()
    Lb.orc_search_by_bow_rig.argtypes = ([C.c_int] * 3 + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 5 +
                                         [C.c_int, C.c_float, C.c_int, C.c_void_p])
    Lb.orc_search_by_bow_rig.restype = C.c_int
    d1, d2 = _c(p["desc1"], np.uint8).reshape(-1, 32), _c(p["desc2"], np.uint8).reshape(-1, 32)
    v1 = _c(p["valid1"], np.uint8)
    f1 = [_c(a, np.int32) for a in p["fv1"]]
    f2 = [_c(a, np.int32) for a in p["fv2"]]
    g1, g2 = _c(p["angle1"], np.float32), _c(p["angle2"], np.float32)
    m21 = np.zeros(max(len(d2), 1), np.int32)
    nm = Lb.orc_search_by_bow_rig(len(d1), len(d2), int(n_left), _ptr(d1), _ptr(d2), _ptr(v1), len(f1[0]), _ptr(f1[0]), _ptr(f1[1]), _ptr(f1[2]),
                                  len(f2[0]), _ptr(f2[0]), _ptr(f2[1]), _ptr(f2[2]), _ptr(g1), _ptr(g2), int(th_low), float(nnratio),
                                  int(bool(check_orientation)), _ptr(m21))
    return nm, m21[:len(d2)]


PAIR_ACCEPT = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int)


def search_for_triangulation_rig(p, accept, coarse=False, check_orientation=True):
    """orc_search_for_triangulation_rig (ORBmatcher.cc:1168-1402 for KeyFrames of a two-camera rig; accept(idx1, idx2) stands for
    pCamera1->epipolarConstrain of :1332).  p: desc1/2, valid1, avail2 (or None), fv1/fv2, angle1/2.
    -> (nmatches, match12, number of accept calls)"""
    Lb = _mlib()
    Lb.orc_search_for_triangulation_rig.argtypes = ([C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int] +
                                                    [C.c_void_p] * 5 + [C.c_int, C.c_int, PAIR_ACCEPT, C.c_void_p, C.c_void_p, C.c_void_p])
    Lb.orc_search_for_triangulation_rig.restype = C.c_int
    d1, d2 = _c(p["desc1"], np.uint8).reshape(-1, 32), _c(p["desc2"], np.uint8).reshape(-1, 32)
    n1, n2 = len(d1), len(d2)
    v1 = _c(p["valid1"], np.uint8)
    a2 = None if p.get("avail2") is None else _c(p["avail2"], np.uint8)
    f1 = [_c(a, np.int32) for a in p["fv1"]]
    f2 = [_c(a, np.int32) for a in p["fv2"]]
    g1, g2 = _c(p["angle1"], np.float32), _c(p["angle2"], np.float32)
    m12 = np.zeros(max(n1, 1), np.int32)
    ncalls = C.c_long(0)
    fn = PAIR_ACCEPT(lambda _ctx, i1, i2: int(bool(accept(i1, i2))))
    nm = Lb.orc_search_for_triangulation_rig(n1, n2, _ptr(d1), _ptr(d2), _ptr(v1), None if a2 is None else _ptr(a2), len(f1[0]), _ptr(f1[0]),
                                             _ptr(f1[1]), _ptr(f1[2]), len(f2[0]), _ptr(f2[0]), _ptr(f2[1]), _ptr(f2[2]), _ptr(g1), _ptr(g2),
                                             int(bool(coarse)), int(bool(check_orientation)), fn, None, _ptr(m12), C.byref(ncalls))
    return nm, m12[:n1], ncalls.value


def search_for_triangulation(p, coarse=False, check_orientation=True):
    """oracle/matcher_oracle.cc orc_search_for_triangulation (ORBmatcher.cc:1168-1402).  p: dict as made by
    tests/bow_match_cases.make_triangulation_pair.  -> (nmatches, match12)"""
    Lb = _mlib()
    Lb.orc_search_for_triangulation.argtypes = ([C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.c_int] + [C.c_void_p] * 3 +
                                                [C.c_int] + [C.c_void_p] * 11 + [C.c_int, C.c_int, C.c_void_p])
    Lb.orc_search_for_triangulation.restype = C.c_int
    d1, d2 = _c(p["desc1"], np.uint8).reshape(-1, 32), _c(p["desc2"], np.uint8).reshape(-1, 32)
    n1, n2 = len(d1), len(d2)
    v1, a2, s1, s2 = (_c(p[k], np.uint8) for k in ("valid1", "avail2", "stereo1", "stereo2"))
    f1 = [_c(a, np.int32) for a in p["fv1"]]
    f2 = [_c(a, np.int32) for a in p["fv2"]]
    k1, k2 = p["kp1"], p["kp2"]
    xy1 = _c(np.stack([k1["x"], k1["y"]], 1), np.float32)
    xy2 = _c(np.stack([k2["x"], k2["y"]], 1), np.float32)
    g1, g2 = _c(k1["angle"], np.float32), _c(k2["angle"], np.float32)
    sc, sg = _c(p["scale_factors2"], np.float32), _c(p["level_sigma2_2"], np.float32)
    epth = _c(np.float32(100) * sc[k2["octave"]], np.float32)
    unc = _c(sg[k2["octave"]], np.float32)
    F, ep = _c(p["F12"], np.float32).reshape(9), _c(p["ep"], np.float32)
    m12 = np.zeros(max(n1, 1), np.int32)
    nm = Lb.orc_search_for_triangulation(n1, n2, _ptr(d1), _ptr(d2), _ptr(v1), _ptr(a2), _ptr(s1), _ptr(s2), len(f1[0]),
                                         _ptr(f1[0]), _ptr(f1[1]), _ptr(f1[2]), len(f2[0]), _ptr(f2[0]), _ptr(f2[1]),
                                         _ptr(f2[2]), _ptr(xy1), _ptr(xy2), _ptr(g1), _ptr(g2), _ptr(epth), _ptr(unc), _ptr(F),
                                         _ptr(ep), int(bool(coarse)), int(bool(check_orientation)), _ptr(m12))
    return nm, m12[:n1]
