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
        cap = self.nfeatures + 4 * self.nlevels + 64
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


def fast_atan2(y, x):
    return float(lib().orc_fast_atan2(float(y), float(x)))


def cos_sin(angle_deg):
    a, b = C.c_float(), C.c_float()
    lib().orc_cos_sin(float(angle_deg), C.byref(a), C.byref(b))
    return a.value, b.value


def distribute_quadtree(xs, ys, resp, minX, maxX, minY, maxY, N):
    xs, ys, resp = (np.ascontiguousarray(v, np.float32) for v in (xs, ys, resp))
    n = len(xs)
    cap = n + 8
    ox, oy, orr = (np.zeros(cap, np.float32) for _ in range(3))
    m = lib().orc_distribute_quadtree(_ptr(xs), _ptr(ys), _ptr(resp), n, minX, maxX, minY, maxY, N, _ptr(ox),
                                      _ptr(oy), _ptr(orr), cap)
    return np.stack([ox[:m], oy[:m], orr[:m]], 1)
