"""B = 1: one frame at a time through the C ABI from host images."""
import os
import sys
import time

import numpy as np

from . import ROOT, KITTI_MB, KITTI_MBF, self_check, oracle_module, tests_dir


def per_frame_leg(msorb, ex, left, right):
    """What an unchanged Frame.cc caller sees per frame (host cv::Mat in, host keypoints / descriptors out; the drop-in class adds
    the cv::Mat / std::vector conversions, ~0.01-0.04 ms, tools/latency_class.cc): msorb_extract on one image, and both eyes +
    ComputeStereoMatches in one call (msorb_extract_stereo), buffers prepared once."""
    import ctypes as C
    L = ex.L
    cap = ex.capacity
    rows, cols = left.shape
    left, right = np.ascontiguousarray(left), np.ascontiguousarray(right)
    kl, kr = np.zeros(cap, msorb.KP_DTYPE), np.zeros(cap, msorb.KP_DTYPE)
    dl, dr = np.zeros((cap, 32), np.uint8), np.zeros((cap, 32), np.uint8)
    ur, dp = np.zeros(cap, np.float32), np.zeros(cap, np.float32)
    n, mono, nl, nr, oob = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    one = (ex.h, p(left), rows, cols, cols, 0, 0, p(kl), p(dl), cap, C.byref(n), C.byref(mono))
    L.msorb_extract_stereo.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_float, C.c_float,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_void_p]
    st = (ex.h, p(left), p(right), rows, cols, cols, cols, KITTI_MB, KITTI_MBF, p(kl), p(dl), C.byref(nl), p(kr), p(dr), C.byref(nr), cap,
          p(ur), p(dp), C.byref(oob))
    vp, ci = C.c_void_p, C.c_int
    L.msorb_extract_pair.argtypes = [vp, vp, vp, ci, ci, C.c_size_t, C.c_size_t, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci]
    mono_r = C.c_int()
    pr = (ex.h, p(left), p(right), rows, cols, cols, cols, 0, 0, p(kl), p(dl), C.byref(nl), C.byref(mono), p(kr), p(dr), C.byref(nr), C.byref(mono_r), cap, 0)
    t1, t2, t3 = [], [], []
    for i in range(45):
        t0 = time.perf_counter(); L.msorb_extract(*one); t1.append(time.perf_counter() - t0)
    for i in range(45):
        t0 = time.perf_counter(); L.msorb_extract_pair(*pr); t3.append(time.perf_counter() - t0)
    for i in range(45):
        t0 = time.perf_counter(); L.msorb_extract_stereo(*st); t2.append(time.perf_counter() - t0)
    m1, m2 = float(np.median(t1[5:])), float(np.median(t2[5:]))
    return {"what": "one frame at a time through the C ABI from host images (B = 1): what the drop-in ORBextractor::operator() costs",
            "ms_one_image": round(m1 * 1e3, 4), "ms_two_images_one_call_no_match": round(float(np.median(t3[5:])) * 1e3, 4),
            "ms_stereo_frame_one_call": round(m2 * 1e3, 4),
            "keypoints_stereo_frame": int(nl.value + nr.value), "mkeypoints_per_s_stereo_frame": round((nl.value + nr.value) / m2 / 1e6, 2)}
