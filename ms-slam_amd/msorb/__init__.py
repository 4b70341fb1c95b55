"""Python (ctypes) binding of libmsorb.so — the C ABI in include/msorb.h.

Used by tests/ and bench.py.  PyTorch is only plumbing here (device buffers, streams, torch.distributed);
every computation happens in the hand-written HIP kernels behind the C ABI.  There is no CPU fallback:
if the library or a GPU is missing, the calls raise.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("MSORB_LIB") or os.path.join(_PKG, "libmsorb.so")   # MSORB_LIB: an experimental build (tools/)
_LIB = None

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28

STAGES = ("pyramid", "fast", "compact", "blur", "select", "describe")

OK, E_INVALID, E_NO_DEVICE, E_HIP, E_CAPACITY, E_GEOMETRY, E_EMPTY = 0, -1, -2, -3, -4, -5, -6

# every symbol include/msorb.h declares (tests check the library exports them all)
ABI_VERSION = 6001   # MSORB_ABI_VERSION of the include/msorb.h this mirror was written against (tests hold the two together)

EXPORTS = (
    "msorb_last_error", "msorb_device_count", "msorb_device_memory", "msorb_abi_version", "msorb_abi_compatible", "msorb_set_fatal_callback", "msorb_notify_fatal", "msorb_extractor_create", "msorb_extractor_destroy",
    "msorb_extractor_tables", "msorb_extractor_capacity", "msorb_extract", "msorb_pyramid_level",
    "msorb_extract_batch", "msorb_extractor_set_profiling", "msorb_extractor_set_overlap", "msorb_extractor_stage_ms", "msorb_debug_level_size",
    "msorb_debug_copy_level", "msorb_debug_candidates", "msorb_debug_patch_tables", "msorb_debug_std_sort", "msorb_distribute_quadtree", "msorb_extract_stereo",
    "msorb_extract_stereo_split", "msorb_pyramid_batch", "msorb_stereo_matches_split", "msorb_extractor_set_host_pyramid",
    "msorb_extractor_set_semantics", "msorb_extract_pair", "msorb_stage_image", "msorb_pyramid_level_image",
)


class MsorbError(RuntimeError):
    def __init__(self, code, what):
        super().__init__(f"{what}: error {code}: {lib().msorb_last_error().decode()}")
        self.code = code


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(_PKG, "csrc")])


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run __graft_entry__.build() (hipcc) first; "
                               "msorb has no CPU fallback")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7; if torch is going to be
        # used in this process (device buffers, torch.distributed) it must be loaded BEFORE libmsorb.so so that
        # libmsorb's NEEDED libamdhip64.so.7 resolves to the already-loaded copy instead of /opt/rocm's.
        if not os.environ.get("MSORB_NO_TORCH"):
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(LIB_PATH)
        vp, ci, cf = C.c_void_p, C.c_int, C.c_float
        if not hasattr(L, "msorb_abi_compatible") or not L.msorb_abi_compatible(ABI_VERSION):
            have = L.msorb_abi_version() if hasattr(L, "msorb_abi_version") else "< 5000 (no version symbol)"
            raise RuntimeError(f"{LIB_PATH} has ABI {have}, the Python mirror was written against {ABI_VERSION}: rebuild "
                               "(python __graft_entry__.py)")
        L.msorb_set_fatal_callback.argtypes = [vp, vp]
        L.msorb_set_fatal_callback.restype = None
        L.msorb_notify_fatal.argtypes = [ci, C.c_char_p]
        L.msorb_notify_fatal.restype = None
        L.msorb_last_error.restype = C.c_char_p
        L.msorb_extractor_create.argtypes = [ci, cf, ci, ci, ci, ci, C.POINTER(vp)]
        L.msorb_extractor_destroy.argtypes = [vp]
        L.msorb_extractor_destroy.restype = None
        L.msorb_extractor_tables.argtypes = [vp, vp, vp, vp, vp, vp]
        L.msorb_extractor_capacity.argtypes = [vp]
        L.msorb_extract.argtypes = [vp, vp, ci, ci, C.c_size_t, ci, ci, vp, vp, ci, C.POINTER(ci), C.POINTER(ci)]
        L.msorb_pyramid_level.argtypes = [vp, ci, C.POINTER(vp), C.POINTER(ci), C.POINTER(ci), C.POINTER(C.c_size_t)]
        L.msorb_extract_batch.argtypes = [vp, vp, ci, ci, ci, C.c_size_t, C.c_size_t, ci, ci, vp, vp, ci, vp, vp]
        L.msorb_extractor_set_profiling.argtypes = [vp, ci]
        L.msorb_extractor_stage_ms.argtypes = [vp, vp]
        L.msorb_extractor_set_overlap.argtypes = [vp, ci, ci]
        L.msorb_debug_level_size.argtypes = [vp, ci, C.POINTER(ci), C.POINTER(ci)]
        L.msorb_debug_copy_level.argtypes = [vp, ci, ci, ci, vp]
        L.msorb_debug_candidates.argtypes = [vp, ci, ci, vp, ci, C.POINTER(ci)]
        L.msorb_distribute_quadtree.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, vp, ci, C.POINTER(ci)]
        _LIB = L
    return _LIB


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _check(rc, what):
    if rc != OK:
        raise MsorbError(rc, what)


class ORBextractor:
    """Mirror of ORB_SLAM3::ORBextractor (reference include/ORBextractor.h:43-109): same constructor
    arguments, `__call__` = operator(), getters with the reference's names."""

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device=0):
        self.L = lib()
        self.nfeatures, self.nlevels = nfeatures, nlevels
        h = C.c_void_p()
        _check(self.L.msorb_extractor_create(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device,
                                             C.byref(h)), "msorb_extractor_create")
        self.h = h
        self.capacity = self.L.msorb_extractor_capacity(self.h)
        self._tables = None

    def close(self):
        if getattr(self, "h", None):
            self.L.msorb_extractor_destroy(self.h)
            self.h = None

    __del__ = close

    # --- getters (ORBextractor.h:61-81) ---
    def _tab(self):
        if self._tables is None:
            n = self.nlevels
            sc, isc, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
            per = np.zeros(n, np.int32)
            _check(self.L.msorb_extractor_tables(self.h, _np_ptr(sc), _np_ptr(isc), _np_ptr(s2), _np_ptr(is2),
                                                 _np_ptr(per)), "msorb_extractor_tables")
            self._tables = dict(scale=sc, inv_scale=isc, sigma2=s2, inv_sigma2=is2, per_level=per)
        return self._tables

    def GetLevels(self):
        return self.nlevels

    def GetScaleFactors(self):
        return self._tab()["scale"]

    def GetInverseScaleFactors(self):
        return self._tab()["inv_scale"]

    def GetScaleSigmaSquares(self):
        return self._tab()["sigma2"]

    def GetInverseScaleSigmaSquares(self):
        return self._tab()["inv_sigma2"]

    def features_per_level(self):
        return self._tab()["per_level"]

    # --- operator() on one host image (ORBextractor.cc:1086-1168) ---
    def __call__(self, image, lapping=(0, 0)):
        """-> (monoIndex, keypoints[KP_DTYPE], descriptors[n,32] u8); monoIndex == -1 for an empty image."""
        image = np.ascontiguousarray(image, np.uint8)
        rows, cols = image.shape if image.ndim == 2 and image.size else (0, 0)
        kps = np.zeros(self.capacity, KP_DTYPE)
        desc = np.zeros((self.capacity, 32), np.uint8)
        n, mono = C.c_int(0), C.c_int(0)
        rc = self.L.msorb_extract(self.h, _np_ptr(image) if image.size else None, rows, cols, cols, lapping[0],
                                  lapping[1], _np_ptr(kps), _np_ptr(desc), self.capacity, C.byref(n), C.byref(mono))
        if rc == E_EMPTY:
            return -1, kps[:0], desc[:0]
        _check(rc, "msorb_extract")
        return mono.value, kps[:n.value].copy(), desc[:n.value].copy()

    def extract_stereo(self, left, right, mb, mbf):
        """msorb_extract_stereo: both eyes + Frame::ComputeStereoMatches in one call.
        -> (kps_left, desc_left, kps_right, desc_right, mvuRight, mvDepth, n_oob)"""
        left, right = np.ascontiguousarray(left, np.uint8), np.ascontiguousarray(right, np.uint8)
        assert left.shape == right.shape and left.ndim == 2
        rows, cols = left.shape
        cap = self.capacity
        kl, kr = np.zeros(cap, KP_DTYPE), np.zeros(cap, KP_DTYPE)
        dl, dr = np.zeros((cap, 32), np.uint8), np.zeros((cap, 32), np.uint8)
        ur, dp = np.zeros(cap, np.float32), np.zeros(cap, np.float32)
        nl, nr, oob = C.c_int(0), C.c_int(0), C.c_int(0)
        self.L.msorb_extract_stereo.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_size_t,
                                                C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _check(self.L.msorb_extract_stereo(self.h, _np_ptr(left), _np_ptr(right), rows, cols, cols, cols, mb, mbf, _np_ptr(kl),
                                           _np_ptr(dl), C.byref(nl), _np_ptr(kr), _np_ptr(dr), C.byref(nr), cap, _np_ptr(ur),
                                           _np_ptr(dp), C.byref(oob)), "msorb_extract_stereo")
        a, b = nl.value, nr.value
        return kl[:a].copy(), dl[:a].copy(), kr[:b].copy(), dr[:b].copy(), ur[:a].copy(), dp[:a].copy(), oob.value

    def extract_pair(self, image_a, image_b, lapping=(0, 0), stage=False):
        """msorb_extract_pair: two same-sized images through one kernel chain, no stereo match.
        -> ((mono_a, kps_a, desc_a), (mono_b, kps_b, desc_b)); stage=True goes through msorb_stage_image for image_a; a pair
        (ex_a, ex_b) of extractors / None stages image_a on ex_a and image_b on ex_b (the `staged` bits; ex may be self)."""
        image_a, image_b = np.ascontiguousarray(image_a, np.uint8), np.ascontiguousarray(image_b, np.uint8)
        assert image_a.shape == image_b.shape and image_a.ndim == 2
        rows, cols = image_a.shape
        cap = self.capacity
        vp, ci, sz = C.c_void_p, C.c_int, C.c_size_t
        ka, kb = np.zeros(cap, KP_DTYPE), np.zeros(cap, KP_DTYPE)
        da, db = np.zeros((cap, 32), np.uint8), np.zeros((cap, 32), np.uint8)
        na, nb, ma, mb = ci(0), ci(0), ci(0), ci(0)
        ptr, stride, staged = [_np_ptr(image_a), _np_ptr(image_b)], [cols, cols], 0
        stagers = stage if isinstance(stage, (tuple, list)) else ((self if stage else None), None)
        self.L.msorb_stage_image.argtypes = [vp, vp, ci, ci, sz, vp, vp]
        for i, ex in enumerate(stagers):
            if ex is None:
                continue
            pin, pitch = vp(), sz()
            _check(self.L.msorb_stage_image(ex.h, ptr[i], rows, cols, cols, C.byref(pin), C.byref(pitch)), "msorb_stage_image")
            ptr[i], stride[i], staged = pin, pitch.value, staged | (1 << i)
        self.L.msorb_extract_pair.argtypes = [vp, vp, vp, ci, ci, sz, sz, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci]
        _check(self.L.msorb_extract_pair(self.h, ptr[0], ptr[1], rows, cols, stride[0], stride[1], lapping[0], lapping[1],
                                         _np_ptr(ka), _np_ptr(da), C.byref(na), C.byref(ma), _np_ptr(kb), _np_ptr(db),
                                         C.byref(nb), C.byref(mb), cap, staged), "msorb_extract_pair")
        a, b = na.value, nb.value
        return (ma.value, ka[:a].copy(), da[:a].copy()), (mb.value, kb[:b].copy(), db[:b].copy())

    def pyramid_level_image(self, image, level):
        """mvImagePyramid[level] of image 0 / 1 of the last extract_pair (needs set_host_pyramid for levels >= 1)."""
        p, r, c, s = C.c_void_p(), C.c_int(), C.c_int(), C.c_size_t()
        _check(self.L.msorb_pyramid_level_image(self.h, image, level, C.byref(p), C.byref(r), C.byref(c), C.byref(s)),
               "msorb_pyramid_level_image")
        buf = (C.c_uint8 * (s.value * r.value)).from_address(p.value)
        return np.frombuffer(buf, np.uint8).reshape(r.value, s.value)[:, :c.value].copy()

    def extract_stereo_split(self, right_ex, left, right, mb, mbf):
        """msorb_extract_stereo_split: self = the left extractor (device A), right_ex = the right extractor (device B, may
        equal A): each eye on its own device, gather onto A, Frame::ComputeStereoMatches on A.  Same return as extract_stereo."""
        left, right = np.ascontiguousarray(left, np.uint8), np.ascontiguousarray(right, np.uint8)
        assert left.shape == right.shape and left.ndim == 2
        rows, cols = left.shape
        cap = self.capacity
        kl, kr = np.zeros(cap, KP_DTYPE), np.zeros(cap, KP_DTYPE)
        dl, dr = np.zeros((cap, 32), np.uint8), np.zeros((cap, 32), np.uint8)
        ur, dp = np.zeros(cap, np.float32), np.zeros(cap, np.float32)
        nl, nr, oob = C.c_int(0), C.c_int(0), C.c_int(0)
        vp = C.c_void_p
        self.L.msorb_extract_stereo_split.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_float,
                                                      C.c_float, vp, vp, vp, vp, vp, vp, C.c_int, vp, vp, vp]
        _check(self.L.msorb_extract_stereo_split(self.h, right_ex.h, _np_ptr(left), _np_ptr(right), rows, cols, cols, cols, mb,
                                                 mbf, _np_ptr(kl), _np_ptr(dl), C.byref(nl), _np_ptr(kr), _np_ptr(dr),
                                                 C.byref(nr), cap, _np_ptr(ur), _np_ptr(dp), C.byref(oob)),
               "msorb_extract_stereo_split")
        a, b = nl.value, nr.value
        return kl[:a].copy(), dl[:a].copy(), kr[:b].copy(), dr[:b].copy(), ur[:a].copy(), dp[:a].copy(), oob.value

    def pyramid_batch(self, images):
        """msorb_pyramid_batch: ComputePyramid only, for a torch.uint8 CUDA tensor [n, rows, cols] (asynchronous)."""
        import torch
        assert images.is_cuda and images.dtype == torch.uint8 and images.dim() == 3 and images.stride(2) == 1
        n, rows, cols = images.shape
        self.L.msorb_pyramid_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t]
        _check(self.L.msorb_pyramid_batch(self.h, images.data_ptr(), n, rows, cols, images.stride(1), images.stride(0)),
               "msorb_pyramid_batch")

    def pyramid_level(self, level):
        """mvImagePyramid[level] of the last __call__ as a numpy array (copy)."""
        p, r, c, s = C.c_void_p(), C.c_int(), C.c_int(), C.c_size_t()
        _check(self.L.msorb_pyramid_level(self.h, level, C.byref(p), C.byref(r), C.byref(c), C.byref(s)),
               "msorb_pyramid_level")
        buf = (C.c_uint8 * (s.value * r.value)).from_address(p.value)
        return np.frombuffer(buf, np.uint8).reshape(r.value, s.value)[:, :c.value].copy()

    # --- batched operator() on device-resident images ---
    def extract_batch(self, images, lapping=(0, 0), out=None):
        """images: torch.uint8 CUDA tensor [n, rows, cols] (contiguous rows).  Returns
        (counts[n] np.int32, mono[n] np.int32, d_keypoints torch.uint8 [n, cap, 28], d_desc torch.uint8 [n, cap, 32]).
        Outputs stay on the device."""
        import torch
        assert images.is_cuda and images.dtype == torch.uint8 and images.dim() == 3 and images.stride(2) == 1
        n, rows, cols = images.shape
        if out is None:
            d_kps = torch.empty((n, self.capacity, 28), dtype=torch.uint8, device=images.device)
            d_desc = torch.empty((n, self.capacity, 32), dtype=torch.uint8, device=images.device)
        else:
            d_kps, d_desc = out
        counts = np.zeros(n, np.int32)
        mono = np.zeros(n, np.int32)
        _check(self.L.msorb_extract_batch(self.h, images.data_ptr(), n, rows, cols, images.stride(1), images.stride(0),
                                          lapping[0], lapping[1], d_kps.data_ptr(), d_desc.data_ptr(), self.capacity,
                                          _np_ptr(counts), _np_ptr(mono)), "msorb_extract_batch")
        return counts, mono, d_kps, d_desc

    def extract_batch_submit(self, images, lapping=(0, 0), out=None):
        """msorb_extract_batch_submit: enqueue only; images / out must stay alive and untouched until extract_batch_wait()."""
        import torch
        assert images.is_cuda and images.dtype == torch.uint8 and images.dim() == 3 and images.stride(2) == 1
        n, rows, cols = images.shape
        if out is None:
            out = (torch.empty((n, self.capacity, 28), dtype=torch.uint8, device=images.device),
                   torch.empty((n, self.capacity, 32), dtype=torch.uint8, device=images.device))
        L = self.L
        L.msorb_extract_batch_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_int,
                                                 C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        _check(L.msorb_extract_batch_submit(self.h, images.data_ptr(), n, rows, cols, images.stride(1), images.stride(0), lapping[0],
                                            lapping[1], out[0].data_ptr(), out[1].data_ptr(), self.capacity), "msorb_extract_batch_submit")
        self._pending = (n, images, out)

    def extract_batch_wait(self):
        """msorb_extract_batch_wait -> (counts, mono, d_keypoints, d_desc) of the submitted batch."""
        n, _, out = getattr(self, "_pending", None) or (0, None, (None, None))   # nothing pending: the library reports it
        counts, mono = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32)
        self.L.msorb_extract_batch_wait.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _check(self.L.msorb_extract_batch_wait(self.h, _np_ptr(counts), _np_ptr(mono)), "msorb_extract_batch_wait")
        self._pending = None
        return counts[:n], mono[:n], out[0], out[1]

    def set_semantics(self, gauss_taps=None, resize_single_stage=False, atan2_fma=False, brief_tap=0):
        """msorb_extractor_set_semantics: variants of the [OpenCV-recall] primitives; set_semantics() restores the defaults."""
        class Sem(C.Structure):
            _fields_ = [("gauss_taps", C.c_int * 7), ("resize_rounding", C.c_int), ("atan2_fma", C.c_int), ("brief_tap", C.c_int)]
        self.L.msorb_extractor_set_semantics.argtypes = [C.c_void_p, C.c_void_p]
        if gauss_taps is None and not resize_single_stage and not atan2_fma and not brief_tap:
            _check(self.L.msorb_extractor_set_semantics(self.h, None), "msorb_extractor_set_semantics")
            return
        sm = Sem()
        sm.gauss_taps[:] = [int(t) for t in (gauss_taps if gauss_taps is not None else (18, 34, 48, 56, 48, 34, 18))]
        sm.resize_rounding, sm.atan2_fma, sm.brief_tap = int(resize_single_stage), int(atan2_fma), int(brief_tap)
        _check(self.L.msorb_extractor_set_semantics(self.h, C.addressof(sm)), "msorb_extractor_set_semantics")

    def set_host_pyramid(self, on=True):
        self.L.msorb_extractor_set_host_pyramid.argtypes = [C.c_void_p, C.c_int]
        _check(self.L.msorb_extractor_set_host_pyramid(self.h, int(on)), "msorb_extractor_set_host_pyramid")

    def set_profiling(self, on=True):
        _check(self.L.msorb_extractor_set_profiling(self.h, int(on)), "set_profiling")

    def set_overlap(self, sub_batches=2, blur_on_second_stream=True):
        _check(self.L.msorb_extractor_set_overlap(self.h, sub_batches, int(blur_on_second_stream)), "set_overlap")

    def stage_ms(self):
        ms = np.zeros(len(STAGES), np.float32)
        _check(self.L.msorb_extractor_stage_ms(self.h, _np_ptr(ms)), "stage_ms")
        return dict(zip(STAGES, ms.tolist()))

    # --- inspection hooks ---
    def debug_level(self, image, level, blurred=False):
        r, c = C.c_int(), C.c_int()
        _check(self.L.msorb_debug_level_size(self.h, level, C.byref(r), C.byref(c)), "debug_level_size")
        out = np.zeros((r.value, c.value), np.uint8)
        _check(self.L.msorb_debug_copy_level(self.h, image, level, int(blurred), _np_ptr(out)), "debug_copy_level")
        return out

    def debug_patch_tables(self):
        """(pattern int8 [256, 4], umax int8 [16]) as they sit in the device's constant memory (msorb_debug_patch_tables)."""
        pat, um = np.zeros(1024, np.int8), np.zeros(16, np.int8)
        self.L.msorb_debug_patch_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _check(self.L.msorb_debug_patch_tables(self.h, _np_ptr(pat), _np_ptr(um)), "debug_patch_tables")
        return pat.reshape(256, 4), um

    def debug_level_size(self, level):
        r, c = C.c_int(), C.c_int()
        _check(self.L.msorb_debug_level_size(self.h, level, C.byref(r), C.byref(c)), "debug_level_size")
        return r.value, c.value

    def debug_candidates(self, image, level):
        cap = 1 << 18
        buf = np.zeros((cap, 3), np.int32)
        n = C.c_int()
        _check(self.L.msorb_debug_candidates(self.h, image, level, _np_ptr(buf), cap, C.byref(n)), "debug_candidates")
        return buf[:n.value].copy()


def debug_std_sort(keys, frame_form=True, device=0, lane_sort=True, timing=False):
    """msorb_debug_std_sort: the device's restatement of libstdc++ std::sort on uint32 keys -> (order, sorted_keys[, microseconds])."""
    keys = np.ascontiguousarray(keys, np.uint32)
    order, out = np.zeros(len(keys), np.uint32), np.zeros(len(keys), np.uint32)
    L = lib()
    L.msorb_debug_std_sort.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    us = C.c_float()
    _check(L.msorb_debug_std_sort(device, _np_ptr(keys), len(keys), int(bool(frame_form)) | (0 if lane_sort else 2), _np_ptr(order), _np_ptr(out),
                                  C.byref(us)), "msorb_debug_std_sort")
    return (order, out, us.value) if timing else (order, out)


def keypoints_from_device(d_kps, counts):
    """torch uint8 [n, cap, 28] -> list of numpy KP_DTYPE arrays."""
    host = d_kps.cpu().numpy()
    return [host[i, :counts[i]].copy().view(KP_DTYPE).reshape(-1) for i in range(len(counts))]


def distribute_quadtree(xs, ys, scores, min_x, max_x, min_y, max_y, n_features):
    """Host-only DistributeOctTree (ORBextractor.cc:555-779); returns kept candidate indices in result order."""
    xs, ys, scores = (np.ascontiguousarray(v, np.uint16) for v in (xs, ys, scores))
    n = len(xs)
    kept = np.zeros(n + 8, np.int32)
    nk = C.c_int()
    _check(lib().msorb_distribute_quadtree(_np_ptr(xs), _np_ptr(ys), _np_ptr(scores), n, min_x, max_x, min_y, max_y,
                                           n_features, _np_ptr(kept), len(kept), C.byref(nk)), "distribute_quadtree")
    return kept[:nk.value].copy()


# ------------------------------------------------------------------------------------------------
# Matcher (include/msorb.h "Matcher" section)
# ------------------------------------------------------------------------------------------------
EXPORTS = EXPORTS + (
    "msorb_frame_create", "msorb_frame_destroy", "msorb_frame_set", "msorb_frame_features_in_area",
    "msorb_search_by_projection_mps", "msorb_search_by_projection_frames", "msorb_hamming_top2",
    "msorb_stereo_matches", "msorb_three_maxima",
)
TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 50, 30   # ORBmatcher.cc:35-37


def _setup_matcher(L):
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    L.msorb_frame_create.argtypes = [ci, C.POINTER(vp)]
    L.msorb_frame_destroy.argtypes = [vp]
    L.msorb_frame_destroy.restype = None
    L.msorb_frame_set.argtypes = [vp, vp, ci, vp, vp, cf, cf, cf, cf, vp, ci]
    L.msorb_frame_features_in_area.argtypes = [vp, cf, cf, cf, ci, ci, vp, ci, C.POINTER(ci)]
    L.msorb_search_by_projection_mps.argtypes = [vp, ci] + [vp] * 12 + [cf, ci, cf, cf, C.POINTER(ci)]
    L.msorb_search_by_projection_frames.argtypes = [vp, ci] + [vp] * 9 + [ci, vp, cf, ci, ci, ci, C.POINTER(ci)]
    L.msorb_hamming_top2.argtypes = [ci, vp, ci, vp, ci, vp, vp, vp, vp, vp, vp]
    L.msorb_stereo_matches.argtypes = [vp, vp, vp, ci, vp, vp, ci, vp, cf, cf, vp, vp, C.POINTER(ci)]
    L.msorb_three_maxima.argtypes = [vp, ci, vp]


_matcher_ready = False


def _mlib():
    global _matcher_ready
    L = lib()
    if not _matcher_ready:
        _setup_matcher(L)
        _matcher_ready = True
    return L


def _c(a, dt):
    return np.ascontiguousarray(a, dt)


class Frame:
    """The members of ORB_SLAM3::Frame that ORBmatcher reads (mvKeysUn, mDescriptors, mvuRight, mGrid,
    image bounds, mvScaleFactors), resident on the device."""

    def __init__(self, keypoints, descriptors, u_right, bounds, scale_factors, device=0):
        self.L = _mlib()
        h = C.c_void_p()
        _check(self.L.msorb_frame_create(device, C.byref(h)), "msorb_frame_create")
        self.h = h
        self.kps = _c(keypoints, KP_DTYPE)
        self.desc = _c(descriptors, np.uint8)
        self.n = len(self.kps)
        ur = None if u_right is None else _c(u_right, np.float32)
        sf = _c(scale_factors, np.float32)
        min_x, max_x, min_y, max_y = bounds
        _check(self.L.msorb_frame_set(self.h, _np_ptr(self.kps), self.n, _np_ptr(self.desc),
                                      None if ur is None else _np_ptr(ur), min_x, max_x, min_y, max_y, _np_ptr(sf),
                                      len(sf)), "msorb_frame_set")

    def close(self):
        if getattr(self, "h", None):
            self.L.msorb_frame_destroy(self.h)
            self.h = None

    __del__ = close

    def GetFeaturesInArea(self, x, y, r, minLevel=-1, maxLevel=-1):
        out = np.zeros(max(self.n, 1), np.int32)
        n = C.c_int()
        _check(self.L.msorb_frame_features_in_area(self.h, x, y, r, minLevel, maxLevel, _np_ptr(out), len(out),
                                                   C.byref(n)), "features_in_area")
        return out[:n.value].copy()

    def SearchByProjection_mps(self, mp, frame_mp, th, bFarPoints=False, thFarPoints=50.0, nnratio=0.8):
        """ORBmatcher(nnratio).SearchByProjection(F, vpMapPoints, th, bFarPoints, thFarPoints); mp is a dict of
        per-map-point arrays (see include/msorb.h); frame_mp is updated in place. Returns nmatches."""
        M = len(mp["proj_x"])
        arrs = [_c(mp["track_in_view"], np.uint8), _c(mp["bad"], np.uint8), _c(mp["sparsified"], np.uint8),
                _c(mp["proj_x"], np.float32), _c(mp["proj_y"], np.float32), _c(mp["proj_xr"], np.float32),
                _c(mp["track_depth"], np.float32), _c(mp["level"], np.int32), _c(mp["view_cos"], np.float32),
                _c(mp["desc"], np.uint8), _c(mp["obs"], np.int32)]
        assert frame_mp.dtype == np.int32 and frame_mp.flags.c_contiguous
        nm = C.c_int()
        _check(self.L.msorb_search_by_projection_mps(self.h, M, *[_np_ptr(a) for a in arrs], _np_ptr(frame_mp), th,
                                                     int(bFarPoints), thFarPoints, nnratio, C.byref(nm)),
               "search_by_projection_mps")
        return nm.value

    def SearchByProjection_frames(self, last, cur_mp, th, forward=False, backward=False, check_orientation=True):
        NL = len(last["u"])
        arrs = [_c(last["valid"], np.uint8), _c(last["u"], np.float32), _c(last["v"], np.float32),
                _c(last["ur"], np.float32), _c(last["octave"], np.int32), _c(last["angle"], np.float32),
                _c(last["desc"], np.uint8), _c(last["mp"], np.int32), _c(last["obs"], np.int32)]
        assert cur_mp.dtype == np.int32 and cur_mp.flags.c_contiguous
        nm = C.c_int()
        _check(self.L.msorb_search_by_projection_frames(self.h, NL, *[_np_ptr(a) for a in arrs], len(arrs[-1]), _np_ptr(cur_mp), th,
                                                        int(forward), int(backward), int(check_orientation),
                                                        C.byref(nm)), "search_by_projection_frames")
        return nm.value

    def SearchByProjection_kf(self, pts, cur_mp, th, orb_dist, check_orientation=True):
        """msorb_search_by_projection_kf (relocalisation form, ORBmatcher.cc:2154-2275); pts: dict valid,u,v,level,angle,
        desc,mp; cur_mp updated in place.  Returns nmatches."""
        arrs = [_c(pts["valid"], np.uint8), _c(pts["u"], np.float32), _c(pts["v"], np.float32), _c(pts["level"], np.int32),
                _c(pts["angle"], np.float32), _c(pts["desc"], np.uint8), _c(pts["mp"], np.int32)]
        assert cur_mp.dtype == np.int32 and cur_mp.flags.c_contiguous
        nm = C.c_int()
        self.L.msorb_search_by_projection_kf.argtypes = ([C.c_void_p, C.c_int] + [C.c_void_p] * 8 +
                                                         [C.c_float, C.c_int, C.c_int, C.c_void_p])
        _check(self.L.msorb_search_by_projection_kf(self.h, len(arrs[0]), *[_np_ptr(a) for a in arrs], _np_ptr(cur_mp), th,
                                                    int(orb_dist), int(check_orientation), C.byref(nm)),
               "search_by_projection_kf")
        return nm.value

    def SearchByProjection_sim3(self, pts, matched, th, max_dist):
        """msorb_search_by_projection_sim3 (loop-closing forms, ORBmatcher.cc:423-753); matched updated in place."""
        arrs = [_c(pts["valid"], np.uint8), _c(pts["u"], np.float32), _c(pts["v"], np.float32), _c(pts["level"], np.int32),
                _c(pts["desc"], np.uint8), _c(pts["mp"], np.int32)]
        assert matched.dtype == np.int32 and matched.flags.c_contiguous
        nm = C.c_int()
        self.L.msorb_search_by_projection_sim3.argtypes = ([C.c_void_p, C.c_int] + [C.c_void_p] * 7 +
                                                           [C.c_float, C.c_float, C.c_void_p])
        _check(self.L.msorb_search_by_projection_sim3(self.h, len(arrs[0]), *[_np_ptr(a) for a in arrs], _np_ptr(matched), th,
                                                      max_dist, C.byref(nm)), "search_by_projection_sim3")
        return nm.value

    def FuseSearch(self, inv_level_sigma2, valid, u, v, ur, predicted_level, radius, mp_desc):
        """msorb_fuse_search: the window search of ORBmatcher::Fuse on this KeyFrame.  -> (best_idx, best_dist)"""
        arrs = [_c(valid, np.uint8), _c(u, np.float32), _c(v, np.float32), _c(ur, np.float32), _c(predicted_level, np.int32),
                _c(radius, np.float32), _c(mp_desc, np.uint8)]
        n = len(arrs[0])
        inv = _c(inv_level_sigma2, np.float32)
        bi, bd = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32)
        self.L.msorb_fuse_search.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 9
        _check(self.L.msorb_fuse_search(self.h, _np_ptr(inv), len(inv), n, *[_np_ptr(a) for a in arrs], _np_ptr(bi),
                                        _np_ptr(bd)), "fuse_search")
        return bi[:n], bd[:n]


    def FuseSearchGated(self, gate_kps, gate_uright, inv_level_sigma2, valid, u, v, ur, predicted_level, radius, mp_desc):
        """msorb_fuse_search_gated: ORBmatcher::Fuse(..., bRight = true) on a two-camera KeyFrame — this frame = the right camera, the
        level band and the error gate read gate_kps / gate_uright (pKF->GetKeyPoint(idx) / GetuRight(idx)).  -> (best_idx, best_dist)"""
        arrs = [_c(valid, np.uint8), _c(u, np.float32), _c(v, np.float32), _c(ur, np.float32), _c(predicted_level, np.int32),
                _c(radius, np.float32), _c(mp_desc, np.uint8)]
        n = len(arrs[0])
        inv = _c(inv_level_sigma2, np.float32)
        gk = np.ascontiguousarray(gate_kps)
        gu = None if gate_uright is None else _c(gate_uright, np.float32)
        bi, bd = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32)
        self.L.msorb_fuse_search_gated.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_int] + [C.c_void_p] * 9
        _check(self.L.msorb_fuse_search_gated(self.h, gk.ctypes.data_as(C.c_void_p), None if gu is None else _np_ptr(gu), _np_ptr(inv),
                                              len(inv), n, *[_np_ptr(a) for a in arrs], _np_ptr(bi), _np_ptr(bd)), "fuse_search_gated")
        return bi[:n], bd[:n]


def _sim3_side(p):
    return [_c(p["valid"], np.uint8), _c(p["u"], np.float32), _c(p["v"], np.float32), _c(p["level"], np.int32),
            _c(p["desc"], np.uint8)]


def search_by_sim3(kf1, kf2, p1, p2, th):
    """msorb_search_by_sim3: kf1 / kf2 = Frame handles of the two KeyFrames, p1 / p2 = dicts valid, u, v, level, desc of their
    map points (projected into the other KeyFrame).  -> (match12[n1], nFound)"""
    L = _mlib()
    a, b = _sim3_side(p1), _sim3_side(p2)
    m12 = np.full(max(len(a[0]), 1), -1, np.int32)
    nf = C.c_int()
    vp = C.c_void_p
    L.msorb_search_by_sim3.argtypes = [vp, vp, C.c_int] + [vp] * 5 + [C.c_int] + [vp] * 5 + [C.c_float, vp, vp]
    _check(L.msorb_search_by_sim3(kf1.h, kf2.h, len(a[0]), *[_np_ptr(x) for x in a], len(b[0]), *[_np_ptr(x) for x in b], th,
                                  _np_ptr(m12), C.byref(nf)), "msorb_search_by_sim3")
    return m12[:len(a[0])], nf.value


def fuse_sim3_search(kf, pts, th):
    """msorb_fuse_sim3_search: the search of Fuse(pKF, Scw, ...).  -> (best_idx, best_dist), INT_MAX = none"""
    L = _mlib()
    a = _sim3_side(pts)
    n = len(a[0])
    bi, bd = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32)
    vp = C.c_void_p
    L.msorb_fuse_sim3_search.argtypes = [vp, C.c_int] + [vp] * 5 + [C.c_float, vp, vp]
    _check(L.msorb_fuse_sim3_search(kf.h, n, *[_np_ptr(x) for x in a], th, _np_ptr(bi), _np_ptr(bd)), "msorb_fuse_sim3_search")
    return bi[:n], bd[:n]


def search_for_initialization(f1, f2, prev_xy, window_size=100, nnratio=0.9, check_orientation=True):
    """msorb_search_for_initialization; prev_xy [N1, 2] float32 updated in place.  -> (vnMatches12, nmatches)"""
    L = _mlib()
    assert prev_xy.dtype == np.float32 and prev_xy.flags.c_contiguous
    n1 = prev_xy.shape[0]
    m12 = np.full(max(n1, 1), -1, np.int32)
    nm = C.c_int()
    vp = C.c_void_p
    L.msorb_search_for_initialization.argtypes = [vp, vp, vp, C.c_int, C.c_float, C.c_int, vp, vp]
    _check(L.msorb_search_for_initialization(f1.h, f2.h, _np_ptr(prev_xy), int(window_size), nnratio, int(check_orientation),
                                             _np_ptr(m12), C.byref(nm)), "msorb_search_for_initialization")
    return m12[:n1], nm.value


def search_by_projection_loop(kf, pts, train_ok, th, max_dist):
    """msorb_search_by_projection_loop (SearchByProjectionLoop, ORBmatcher.cc:532-637).  -> (best_idx[n], nmatches)"""
    L = _mlib()
    a = _sim3_side(pts)
    n = len(a[0])
    ok = _c(train_ok, np.uint8)
    bi = np.full(max(n, 1), -1, np.int32)
    nm = C.c_int()
    vp = C.c_void_p
    L.msorb_search_by_projection_loop.argtypes = [vp, C.c_int] + [vp] * 6 + [C.c_float, C.c_float, vp, vp]
    _check(L.msorb_search_by_projection_loop(kf.h, n, *[_np_ptr(x) for x in a], _np_ptr(ok), th, max_dist, _np_ptr(bi),
                                             C.byref(nm)), "msorb_search_by_projection_loop")
    return bi[:n], nm.value


EXPORTS = EXPORTS + ("msorb_search_by_sim3", "msorb_fuse_sim3_search", "msorb_search_for_initialization",
                     "msorb_search_by_projection_loop")


def hamming_top2(query_desc, train_desc, cand_begin, cand_idx, device=0):
    q, t = _c(query_desc, np.uint8), _c(train_desc, np.uint8)
    cb, ci_ = _c(cand_begin, np.int32), _c(cand_idx, np.int32)
    nq = len(q)
    outs = [np.zeros(nq, np.int32) for _ in range(4)]
    _check(_mlib().msorb_hamming_top2(device, _np_ptr(q), nq, _np_ptr(t), len(t), _np_ptr(cb), _np_ptr(ci_),
                                      *[_np_ptr(o) for o in outs]), "hamming_top2")
    return outs


def stereo_matches(ex_left, ex_right, kps_l, desc_l, kps_r, desc_r, mb, mbf):
    """Frame::ComputeStereoMatches on the pyramids of the two extractors' last __call__.
    -> (mvuRight, mvDepth, n_oob)"""
    kl, kr = _c(kps_l, KP_DTYPE), _c(kps_r, KP_DTYPE)
    dl, dr = _c(desc_l, np.uint8), _c(desc_r, np.uint8)
    ur = np.zeros(len(kl), np.float32)
    dp = np.zeros(len(kl), np.float32)
    oob = C.c_int()
    _check(_mlib().msorb_stereo_matches(ex_left.h, ex_right.h, _np_ptr(kl), len(kl), _np_ptr(dl), _np_ptr(kr), len(kr),
                                        _np_ptr(dr), mb, mbf, _np_ptr(ur), _np_ptr(dp), C.byref(oob)),
           "stereo_matches")
    return ur, dp, oob.value


def three_maxima(sizes):
    s = _c(sizes, np.int32)
    ind = np.zeros(3, np.int32)
    _check(_mlib().msorb_three_maxima(_np_ptr(s), len(s), _np_ptr(ind)), "three_maxima")
    return ind


# ------------------------------------------------------------------------------------------------
# Map sparsification constraint matrix (include/msorb.h)
# ------------------------------------------------------------------------------------------------
EXPORTS = EXPORTS + ("msorb_visibility_csr",)


def visibility_csr(kf_slot_begin, slot_point, slot_cell, point_nobs, obs_begin, obs_kf, kf_in_window, kf_num_mps, N,
                   n_max_obs_floor=0, device=0):
    """MapSparsification::Sparsifying matrix assembly (MapSparsification.cc:58-151) on the device.
    -> dict(n_cols, col_point, obj_coef, n_rows, row_begin, row_kind, row_owner, row_rhs, col_idx, n_max_obs)"""
    L = lib()
    L.msorb_visibility_csr.argtypes = ([C.c_int, C.c_int] + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int] +
                                       [C.c_void_p] * 2 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int] +
                                       [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 3)
    ksb, sp, sc = _c(kf_slot_begin, np.int32), _c(slot_point, np.int32), _c(slot_cell, np.int32)
    pn, ob, ok = _c(point_nobs, np.int32), _c(obs_begin, np.int32), _c(obs_kf, np.int32)
    kw, km = _c(kf_in_window, np.uint8), _c(kf_num_mps, np.int32)
    K, S, P, KT = len(ksb) - 1, len(sp), len(pn), len(kw)
    cap_cols, cap_rows, cap_nnz = S + 1, S + K + KT + 1, 2 * S + len(ok) + 1
    col_point = np.zeros(cap_cols, np.int32)
    obj = np.zeros(cap_cols, np.float32)
    row_begin = np.zeros(cap_rows + 1, np.int32)
    row_kind, row_owner = np.zeros(cap_rows, np.int32), np.zeros(cap_rows, np.int32)
    row_rhs = np.zeros(cap_rows, np.float32)
    col_idx = np.zeros(cap_nnz, np.int32)
    n_cols, n_rows, nnz, nmax = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    _check(L.msorb_visibility_csr(device, K, _np_ptr(ksb), _np_ptr(sp), _np_ptr(sc), P, _np_ptr(pn), _np_ptr(ob),
                                  _np_ptr(ok), KT, _np_ptr(kw), _np_ptr(km), N, n_max_obs_floor, C.byref(n_cols),
                                  _np_ptr(col_point), cap_cols, C.byref(n_rows), _np_ptr(row_begin), _np_ptr(row_kind),
                                  _np_ptr(row_owner), _np_ptr(row_rhs), cap_rows, _np_ptr(col_idx), cap_nnz,
                                  C.byref(nnz), _np_ptr(obj), C.byref(nmax)), "msorb_visibility_csr")
    nc, nr, nz = n_cols.value, n_rows.value, nnz.value
    return dict(n_cols=nc, col_point=col_point[:nc], obj_coef=obj[:nc], n_rows=nr, row_begin=row_begin[:nr + 1],
                row_kind=row_kind[:nr], row_owner=row_owner[:nr], row_rhs=row_rhs[:nr], col_idx=col_idx[:nz],
                n_max_obs=nmax.value)


_RIG_FIELDS = (("track_in_view", np.uint8), ("track_in_view_r", np.uint8), ("bad", np.uint8), ("sparsified", np.uint8), ("proj_x", np.float32),
               ("proj_y", np.float32), ("proj_xr", np.float32), ("proj_yr", np.float32), ("track_depth", np.float32), ("level", np.int32),
               ("level_r", np.int32), ("view_cos", np.float32), ("view_cos_r", np.float32), ("desc", np.uint8), ("obs", np.int32))


def search_by_projection_mps_rig(left, right, mp, left_to_right, right_to_left, frame_mp, th, bFarPoints=False, thFarPoints=50.0, nnratio=0.8):
    """msorb_search_by_projection_mps_rig: ORBmatcher::SearchByProjection(F, vpMapPoints, ...) on a two-camera frame (F.Nleft != -1):
    left / right are Frame handles of the two cameras' keypoints, mp a dict of per-map-point arrays (_RIG_FIELDS), frame_mp
    (int32 [n_left + n_right]) is updated in place.  -> nmatches"""
    L = lib()
    arrs = [_c(mp[k], dt) for k, dt in _RIG_FIELDS] + [_c(left_to_right, np.int32), _c(right_to_left, np.int32)]
    assert frame_mp.dtype == np.int32 and frame_mp.flags.c_contiguous and len(frame_mp) == left.n + right.n
    nm = C.c_int()
    L.msorb_search_by_projection_mps_rig.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 18 + [C.c_float, C.c_int, C.c_float, C.c_float, C.c_void_p]
    _check(L.msorb_search_by_projection_mps_rig(left.h, right.h, len(arrs[0]), *[_np_ptr(a) for a in arrs], _np_ptr(frame_mp), th, int(bFarPoints),
                                                thFarPoints, nnratio, C.byref(nm)), "msorb_search_by_projection_mps_rig")
    return nm.value


def search_by_projection_frames_rig(left, right, last, cur_mp, th, forward=False, backward=False, check_orientation=True):
    """msorb_search_by_projection_frames_rig: SearchByProjection(Current, Last, th, bMono) on a two-camera CurrentFrame; last: dict valid,
    u, v, u_r, v_r, octave, angle, desc, mp, obs (obs indexed by map-point id); cur_mp int32 [n_left + n_right] in / out -> nmatches"""
    L = lib()
    arrs = [_c(last["valid"], np.uint8), _c(last["u"], np.float32), _c(last["v"], np.float32), _c(last["u_r"], np.float32),
            _c(last["v_r"], np.float32), _c(last["octave"], np.int32), _c(last["angle"], np.float32), _c(last["desc"], np.uint8),
            _c(last["mp"], np.int32), _c(last["obs"], np.int32)]
    assert cur_mp.dtype == np.int32 and cur_mp.flags.c_contiguous and len(cur_mp) == left.n + right.n
    nm = C.c_int()
    L.msorb_search_by_projection_frames_rig.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 10 + [C.c_int, C.c_void_p, C.c_float, C.c_int,
                                                                                                              C.c_int, C.c_int, C.c_void_p]
    _check(L.msorb_search_by_projection_frames_rig(left.h, right.h, len(arrs[0]), *[_np_ptr(a) for a in arrs], len(arrs[9]), _np_ptr(cur_mp), th,
                                                   int(forward), int(backward), int(check_orientation), C.byref(nm)),
           "msorb_search_by_projection_frames_rig")
    return nm.value


EXPORTS = EXPORTS + ("msorb_search_by_projection_mps_rig", "msorb_search_by_projection_frames_rig")
EXPORTS = EXPORTS + ("msorb_hamming_dense_top2_batch", "msorb_hamming_dense_top2_batch_ex", "msorb_knn_match2")


DENSE_MATRIX_CORES, DENSE_POPCOUNT = 0, 1


def hamming_dense_top2_batch(d_query, d_train, d_nq, d_nt, repeats=1, device=0, formulation=DENSE_POPCOUNT):
    """Dense brute-force top-2 on device tensors: d_query/d_train torch.uint8 [F, stride, 32], d_nq/d_nt torch.int32 [F].
    formulation: DENSE_POPCOUNT (xor + popcount, BASELINE north_star's form: the default) or DENSE_MATRIX_CORES (int8 MFMA, opt-in).
    -> (best_idx, best_dist, second_dist) torch.int32 [F, q_stride], elapsed_ms over `repeats` launches."""
    import torch
    L = lib()
    L.msorb_hamming_dense_top2_batch_ex.argtypes = [C.c_int] + [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_void_p] * 3 + [C.c_int, C.c_void_p, C.c_int]
    F, qs, _ = d_query.shape
    ts = d_train.shape[1]
    # rows >= d_nq[f] are not written by the kernels: best_idx -1, distances 256 there (a defined value, never stale memory)
    outs = [torch.full((F, qs), v, dtype=torch.int32, device=d_query.device) for v in (-1, 256, 256)]
    ms = C.c_float()
    _check(L.msorb_hamming_dense_top2_batch_ex(device, d_query.data_ptr(), d_train.data_ptr(), d_nq.data_ptr(), d_nt.data_ptr(),
                                               F, qs, ts, int(d_nq.max()), int(d_nt.max()), outs[0].data_ptr(),
                                               outs[1].data_ptr(), outs[2].data_ptr(), repeats, C.byref(ms), int(formulation)),
           "msorb_hamming_dense_top2_batch_ex")
    return outs[0], outs[1], outs[2], ms.value


EXPORTS = EXPORTS + ("msorb_window_top4",)


def window_top4(frame, x, y, r, min_level, max_level, query_desc, ur=None, skip_occupied=None, occupied=None):
    """msorb_window_top4: -> (idx[nq,4], dist[nq,4])"""
    L = _mlib()
    L.msorb_window_top4.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 11
    xs, ys, rs = _c(x, np.float32), _c(y, np.float32), _c(r, np.float32)
    mn, mx = _c(min_level, np.int32), _c(max_level, np.int32)
    qd = _c(query_desc, np.uint8)
    nq = len(xs)
    urs = None if ur is None else _c(ur, np.float32)
    sk = None if skip_occupied is None else _c(skip_occupied, np.uint8)
    oc = None if occupied is None else _c(occupied, np.uint8)
    bi, bd = np.zeros((nq, 4), np.int32), np.zeros((nq, 4), np.int32)
    _check(L.msorb_window_top4(frame.h, nq, _np_ptr(xs), _np_ptr(ys), _np_ptr(rs), None if urs is None else _np_ptr(urs),
                               _np_ptr(mn), _np_ptr(mx), None if sk is None else _np_ptr(sk), _np_ptr(qd),
                               None if oc is None else _np_ptr(oc), _np_ptr(bi), _np_ptr(bd)), "msorb_window_top4")
    return bi, bd


EXPORTS = EXPORTS + ("msorb_vocabulary_create", "msorb_vocabulary_load_text", "msorb_vocabulary_destroy",
                     "msorb_vocabulary_info", "msorb_bow_transform_batch", "msorb_bow_transform",
                     "msorb_distinctive_descriptors")


class Vocabulary:
    """DBoW2 ORB vocabulary tree on the device (msorb_vocabulary_*): transform() = TemplatedVocabulary::transform
    (features, BowVector, FeatureVector, levelsup) as Frame::ComputeBoW calls it (Frame.cc:670-677)."""

    def __init__(self, k=None, L=None, scoring=0, weighting=0, parent=None, is_leaf=None, descriptors=None, weights=None,
                 path=None, device=0):
        lb = lib()
        lb.msorb_vocabulary_create.argtypes = [C.c_int] * 6 + [C.c_void_p] * 4 + [C.POINTER(C.c_void_p)]
        lb.msorb_vocabulary_load_text.argtypes = [C.c_int, C.c_char_p, C.POINTER(C.c_void_p)]
        lb.msorb_vocabulary_destroy.argtypes = [C.c_void_p]
        lb.msorb_vocabulary_destroy.restype = None
        lb.msorb_vocabulary_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 4
        lb.msorb_bow_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 10
        lb.msorb_bow_transform_batch.argtypes = ([C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 4 +
                                                 [C.c_void_p] * 8)
        self.h = None
        h = C.c_void_p()
        if path is not None:
            _check(lb.msorb_vocabulary_load_text(device, str(path).encode(), C.byref(h)), "msorb_vocabulary_load_text")
        else:
            par, lf = _c(parent, np.int32), _c(is_leaf, np.uint8)
            ds, ws = _c(descriptors, np.uint8), _c(weights, np.float64)
            _check(lb.msorb_vocabulary_create(device, k, L, scoring, weighting, len(par), _np_ptr(par), _np_ptr(lf),
                                              _np_ptr(ds), _np_ptr(ws), C.byref(h)), "msorb_vocabulary_create")
        self.h = h
        self.device = device
        a = [C.c_int() for _ in range(4)]
        lb.msorb_vocabulary_info(h, *[C.byref(x) for x in a])
        self.k, self.L, self.n_nodes, self.n_words = [x.value for x in a]

    def close(self):
        if getattr(self, "h", None):
            lib().msorb_vocabulary_destroy(self.h)
            self.h = None

    __del__ = close

    def transform(self, descriptors, levelsup=4):
        """One frame, host arrays. -> dict(bow_word, bow_value, fv_node, fv_begin, fv_feat, feat_word, feat_node,
        feat_weight)"""
        d = _c(descriptors, np.uint8).reshape(-1, 32)
        n = len(d)
        cap = max(n, 1)
        bw, bv = np.zeros(cap, np.int32), np.zeros(cap, np.float64)
        fn, fb, ff = np.zeros(cap, np.int32), np.zeros(cap + 1, np.int32), np.zeros(cap, np.int32)
        fw, fnode, fwt = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.float64)
        nb, nf = C.c_int(), C.c_int()
        _check(lib().msorb_bow_transform(self.h, _np_ptr(d) if n else None, n, levelsup, _np_ptr(bw), _np_ptr(bv),
                                         C.addressof(nb), _np_ptr(fn), _np_ptr(fb), _np_ptr(ff), C.addressof(nf),
                                         _np_ptr(fw), _np_ptr(fnode), _np_ptr(fwt)), "msorb_bow_transform")
        nb, nf = nb.value, nf.value
        return dict(bow_word=bw[:nb], bow_value=bv[:nb], fv_node=fn[:nf], fv_begin=fb[:nf + 1], fv_feat=ff[:fb[nf]],
                    feat_word=fw[:n], feat_node=fnode[:n], feat_weight=fwt[:n])

    def transform_batch(self, d_desc, counts, levelsup=4):
        """Device-resident batch: d_desc torch.uint8 [F, stride, 32] (msorb_extract_batch's descriptors), counts
        host int array [F].  -> dict of torch tensors (bow_word [F,S], bow_value [F,S] f64, n_bow [F], fv_node [F,S],
        fv_begin [F,S+1], fv_feat [F,S], n_fv [F]) and elapsed_ms (kernels only)."""
        import torch
        F, S, _ = d_desc.shape
        cnt = _c(counts, np.int32)
        dev = d_desc.device
        i32 = dict(dtype=torch.int32, device=dev)
        out = dict(bow_word=torch.empty((F, S), **i32), bow_value=torch.empty((F, S), dtype=torch.float64, device=dev),
                   n_bow=torch.empty(F, **i32), fv_node=torch.empty((F, S), **i32), fv_begin=torch.empty((F, S + 1), **i32),
                   fv_feat=torch.empty((F, S), **i32), n_fv=torch.empty(F, **i32))
        ms = C.c_float()
        _check(lib().msorb_bow_transform_batch(self.h, d_desc.data_ptr(), _np_ptr(cnt), F, S, levelsup, S,
                                               out["bow_word"].data_ptr(), out["bow_value"].data_ptr(),
                                               out["n_bow"].data_ptr(), out["fv_node"].data_ptr(),
                                               out["fv_begin"].data_ptr(), out["fv_feat"].data_ptr(),
                                               out["n_fv"].data_ptr(), C.addressof(ms)), "msorb_bow_transform_batch")
        out["elapsed_ms"] = ms.value
        return out


def distinctive_descriptors(descriptors, obs_begin, device=0):
    """msorb_distinctive_descriptors -> (best_idx, best_median, kernel_ms)"""
    lb = lib()
    lb.msorb_distinctive_descriptors.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 3
    d = _c(descriptors, np.uint8).reshape(-1, 32)
    ob = _c(obs_begin, np.int32)
    n = len(ob) - 1
    bi, bm = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32)
    ms = C.c_float()
    _check(lb.msorb_distinctive_descriptors(device, _np_ptr(d) if len(d) else None, _np_ptr(ob), n, _np_ptr(bi),
                                            _np_ptr(bm), C.addressof(ms)), "msorb_distinctive_descriptors")
    return bi[:n], bm[:n], ms.value


EXPORTS = EXPORTS + ("msorb_is_in_frustum",)


class Frustum(C.Structure):
    """msorb_frustum (include/msorb.h): the Frame members isInFrustum reads."""
    _fields_ = [("Rcw", C.c_float * 9), ("tcw", C.c_float * 3), ("Ow", C.c_float * 3), ("fx", C.c_float), ("fy", C.c_float),
                ("cx", C.c_float), ("cy", C.c_float), ("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float),
                ("max_y", C.c_float), ("mbf", C.c_float), ("log_scale_factor", C.c_float), ("n_scale_levels", C.c_int)]

    @classmethod
    def make(cls, Rcw, tcw, Ow, fx, fy, cx, cy, bounds, mbf, log_scale_factor, n_scale_levels):
        f = cls()
        f.Rcw[:] = [float(x) for x in np.asarray(Rcw, np.float32).reshape(9)]
        f.tcw[:] = [float(x) for x in np.asarray(tcw, np.float32).reshape(3)]
        f.Ow[:] = [float(x) for x in np.asarray(Ow, np.float32).reshape(3)]
        f.fx, f.fy, f.cx, f.cy = fx, fy, cx, cy
        f.min_x, f.max_x, f.min_y, f.max_y = bounds
        f.mbf, f.log_scale_factor, f.n_scale_levels = mbf, log_scale_factor, n_scale_levels
        return f


def is_in_frustum(frustum, pos_w, normal, max_distance, min_distance, viewing_cos_limit=0.5, device=0):
    """msorb_is_in_frustum -> dict(track_in_view, proj_x, proj_y, proj_xr, track_depth, level, view_cos, kernel_ms)"""
    lb = lib()
    lb.msorb_is_in_frustum.argtypes = [C.c_int, C.c_void_p, C.c_float, C.c_int] + [C.c_void_p] * 12
    P, Nn = _c(pos_w, np.float32).reshape(-1, 3), _c(normal, np.float32).reshape(-1, 3)
    mx, mn = _c(max_distance, np.float32), _c(min_distance, np.float32)
    n = len(P)
    cap = max(n, 1)
    inv = np.zeros(cap, np.uint8)
    px, py, pxr, dep, vc = [np.zeros(cap, np.float32) for _ in range(5)]
    lvl = np.zeros(cap, np.int32)
    ms = C.c_float()
    _check(lb.msorb_is_in_frustum(device, C.addressof(frustum), viewing_cos_limit, n, _np_ptr(P), _np_ptr(Nn), _np_ptr(mx),
                                  _np_ptr(mn), _np_ptr(inv), _np_ptr(px), _np_ptr(py), _np_ptr(pxr), _np_ptr(dep),
                                  _np_ptr(lvl), _np_ptr(vc), C.addressof(ms)), "msorb_is_in_frustum")
    return dict(track_in_view=inv[:n], proj_x=px[:n], proj_y=py[:n], proj_xr=pxr[:n], track_depth=dep[:n], level=lvl[:n],
                view_cos=vc[:n], kernel_ms=ms.value)


EXPORTS = EXPORTS + ("msorb_search_by_bow",)


class BowPair(C.Structure):
    """msorb_bow_pair (include/msorb.h)."""
    _fields_ = [("n1", C.c_int), ("n2", C.c_int), ("desc1", C.c_void_p), ("desc2", C.c_void_p), ("valid1", C.c_void_p),
                ("avail2", C.c_void_p), ("fv1_nodes", C.c_int), ("fv1_node", C.c_void_p), ("fv1_begin", C.c_void_p),
                ("fv1_feat", C.c_void_p), ("fv2_nodes", C.c_int), ("fv2_node", C.c_void_p), ("fv2_begin", C.c_void_p),
                ("fv2_feat", C.c_void_p), ("angle1", C.c_void_p), ("angle2", C.c_void_p), ("match12", C.c_void_p),
                ("match21", C.c_void_p), ("nmatches", C.c_int)]


def search_by_bow(pairs, th_low=50, inclusive=True, nnratio=0.7, check_orientation=True, device=0):
    """msorb_search_by_bow over a batch.  pairs: list of dicts with desc1, desc2, valid1, avail2 (or None),
    fv1 / fv2 = (node, begin, feat), angle1, angle2.  -> (list of (nmatches, match12, match21), kernel_ms)"""
    lb = lib()
    lb.msorb_search_by_bow.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]
    arr = (BowPair * max(len(pairs), 1))()
    keep, outs = [], []
    for k, p in enumerate(pairs):
        d1, d2 = _c(p["desc1"], np.uint8).reshape(-1, 32), _c(p["desc2"], np.uint8).reshape(-1, 32)
        v1 = _c(p["valid1"], np.uint8)
        a2 = None if p.get("avail2") is None else _c(p["avail2"], np.uint8)
        f1 = [_c(a, np.int32) for a in p["fv1"]]
        f2 = [_c(a, np.int32) for a in p["fv2"]]
        g1, g2 = _c(p["angle1"], np.float32), _c(p["angle2"], np.float32)
        m12, m21 = np.zeros(max(len(d1), 1), np.int32), np.zeros(max(len(d2), 1), np.int32)
        keep.append((d1, d2, v1, a2, f1, f2, g1, g2))
        outs.append((m12, m21, len(d1), len(d2)))
        q = arr[k]
        q.n1, q.n2 = len(d1), len(d2)
        q.desc1, q.desc2, q.valid1 = _np_ptr(d1), _np_ptr(d2), _np_ptr(v1)
        q.avail2 = None if a2 is None else _np_ptr(a2)
        q.fv1_nodes, q.fv1_node, q.fv1_begin, q.fv1_feat = len(f1[0]), _np_ptr(f1[0]), _np_ptr(f1[1]), _np_ptr(f1[2])
        q.fv2_nodes, q.fv2_node, q.fv2_begin, q.fv2_feat = len(f2[0]), _np_ptr(f2[0]), _np_ptr(f2[1]), _np_ptr(f2[2])
        q.angle1, q.angle2, q.match12, q.match21 = _np_ptr(g1), _np_ptr(g2), _np_ptr(m12), _np_ptr(m21)
    ms = C.c_float()
    _check(lb.msorb_search_by_bow(device, C.addressof(arr), len(pairs), int(th_low), int(bool(inclusive)), float(nnratio),
                                  int(bool(check_orientation)), C.addressof(ms)), "msorb_search_by_bow")
    return [(arr[k].nmatches, o[0][:o[2]], o[1][:o[3]]) for k, o in enumerate(outs)], ms.value


def search_by_bow_rig(p, n_left, th_low=50, nnratio=0.7, check_orientation=True, device=0):
    """msorb_search_by_bow_rig: SearchByBoW(pKF, F) on a two-camera frame; p as in search_by_bow (set 2 = the frame's N features, the
    left camera's n_left rows first).  -> (nmatches, match21, match12 = the left partner of each KeyFrame feature)"""
    lb = lib()
    lb.msorb_search_by_bow_rig.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int]
    q = BowPair()
    d1, d2 = _c(p["desc1"], np.uint8).reshape(-1, 32), _c(p["desc2"], np.uint8).reshape(-1, 32)
    v1 = _c(p["valid1"], np.uint8)
    f1 = [_c(a, np.int32) for a in p["fv1"]]
    f2 = [_c(a, np.int32) for a in p["fv2"]]
    g1, g2 = _c(p["angle1"], np.float32), _c(p["angle2"], np.float32)
    m12, m21 = np.zeros(max(len(d1), 1), np.int32), np.zeros(max(len(d2), 1), np.int32)
    q.n1, q.n2 = len(d1), len(d2)
    q.desc1, q.desc2, q.valid1, q.avail2 = _np_ptr(d1), _np_ptr(d2), _np_ptr(v1), None
    q.fv1_nodes, q.fv1_node, q.fv1_begin, q.fv1_feat = len(f1[0]), _np_ptr(f1[0]), _np_ptr(f1[1]), _np_ptr(f1[2])
    q.fv2_nodes, q.fv2_node, q.fv2_begin, q.fv2_feat = len(f2[0]), _np_ptr(f2[0]), _np_ptr(f2[1]), _np_ptr(f2[2])
    q.angle1, q.angle2, q.match12, q.match21 = _np_ptr(g1), _np_ptr(g2), _np_ptr(m12), _np_ptr(m21)
    _check(lb.msorb_search_by_bow_rig(device, C.addressof(q), int(n_left), int(th_low), float(nnratio), int(bool(check_orientation))),
           "msorb_search_by_bow_rig")
    return q.nmatches, m21[:len(d2)], m12[:len(d1)]


PAIR_ACCEPT = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int)   # msorb_pair_accept


def search_for_triangulation_cb(p, accept, th_low=50, check_orientation=True, device=0):
    """msorb_search_for_triangulation_cb: SearchForTriangulation with the geometric test left to the caller (the two-camera arms,
    ORBmatcher.cc:1294-1332).  p: desc1/2, valid1, avail2 (or None), fv1/fv2, angle1/2; accept(idx1, idx2) -> bool, called for
    the candidates the library tries.  -> (nmatches, match12, calls = the (idx1, idx2) accept saw, in order)"""
    lb = lib()
    lb.msorb_search_for_triangulation_cb.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, PAIR_ACCEPT, C.c_void_p]
    q = BowPair()
    d1, d2 = _c(p["desc1"], np.uint8).reshape(-1, 32), _c(p["desc2"], np.uint8).reshape(-1, 32)
    v1 = _c(p["valid1"], np.uint8)
    a2 = None if p.get("avail2") is None else _c(p["avail2"], np.uint8)
    f1 = [_c(a, np.int32) for a in p["fv1"]]
    f2 = [_c(a, np.int32) for a in p["fv2"]]
    g1, g2 = _c(p["angle1"], np.float32), _c(p["angle2"], np.float32)
    m12, m21 = np.zeros(max(len(d1), 1), np.int32), np.zeros(max(len(d2), 1), np.int32)
    q.n1, q.n2 = len(d1), len(d2)
    q.desc1, q.desc2, q.valid1, q.avail2 = _np_ptr(d1), _np_ptr(d2), _np_ptr(v1), None if a2 is None else _np_ptr(a2)
    q.fv1_nodes, q.fv1_node, q.fv1_begin, q.fv1_feat = len(f1[0]), _np_ptr(f1[0]), _np_ptr(f1[1]), _np_ptr(f1[2])
    q.fv2_nodes, q.fv2_node, q.fv2_begin, q.fv2_feat = len(f2[0]), _np_ptr(f2[0]), _np_ptr(f2[1]), _np_ptr(f2[2])
    q.angle1, q.angle2, q.match12, q.match21 = _np_ptr(g1), _np_ptr(g2), _np_ptr(m12), _np_ptr(m21)
    calls = []

    def cb(_ctx, i1, i2):
        calls.append((i1, i2))
        return int(bool(accept(i1, i2)))
    fn = PAIR_ACCEPT(cb)
    _check(lb.msorb_search_for_triangulation_cb(device, C.addressof(q), int(th_low), int(bool(check_orientation)), fn, None),
           "msorb_search_for_triangulation_cb")
    return q.nmatches, m12[:len(d1)], calls


EXPORTS = EXPORTS + ("msorb_search_for_triangulation", "msorb_search_by_bow_rig", "msorb_search_for_triangulation_cb")


class TriangulationPair(C.Structure):
    """msorb_triangulation_pair (include/msorb.h)."""
    _fields_ = [("n1", C.c_int), ("n2", C.c_int), ("desc1", C.c_void_p), ("desc2", C.c_void_p), ("valid1", C.c_void_p),
                ("avail2", C.c_void_p), ("stereo1", C.c_void_p), ("stereo2", C.c_void_p), ("fv1_nodes", C.c_int),
                ("fv1_node", C.c_void_p), ("fv1_begin", C.c_void_p), ("fv1_feat", C.c_void_p), ("fv2_nodes", C.c_int),
                ("fv2_node", C.c_void_p), ("fv2_begin", C.c_void_p), ("fv2_feat", C.c_void_p), ("kp1", C.c_void_p),
                ("kp2", C.c_void_p), ("scale_factors2", C.c_void_p), ("level_sigma2_2", C.c_void_p), ("n_levels2", C.c_int),
                ("F12", C.c_float * 9), ("ep", C.c_float * 2), ("match12", C.c_void_p), ("nmatches", C.c_int)]


def search_for_triangulation(pairs, coarse=False, check_orientation=True, device=0):
    """msorb_search_for_triangulation over a batch.  pairs: dicts with desc1/2, valid1, avail2, stereo1/2, fv1/fv2,
    kp1/kp2 (KP_DTYPE records), scale_factors2, level_sigma2_2, F12 (9), ep (2).
    -> (list of (nmatches, match12), kernel_ms)"""
    lb = lib()
    lb.msorb_search_for_triangulation.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    arr = (TriangulationPair * max(len(pairs), 1))()
    keep, outs = [], []
    for k, p in enumerate(pairs):
        d1, d2 = _c(p["desc1"], np.uint8).reshape(-1, 32), _c(p["desc2"], np.uint8).reshape(-1, 32)
        fl = [_c(p[key], np.uint8) for key in ("valid1", "avail2", "stereo1", "stereo2")]
        f1 = [_c(a, np.int32) for a in p["fv1"]]
        f2 = [_c(a, np.int32) for a in p["fv2"]]
        k1, k2 = np.ascontiguousarray(p["kp1"], KP_DTYPE), np.ascontiguousarray(p["kp2"], KP_DTYPE)
        sc, sg = _c(p["scale_factors2"], np.float32), _c(p["level_sigma2_2"], np.float32)
        m12 = np.zeros(max(len(d1), 1), np.int32)
        keep.append((d1, d2, fl, f1, f2, k1, k2, sc, sg))
        outs.append((m12, len(d1)))
        q = arr[k]
        q.n1, q.n2 = len(d1), len(d2)
        q.desc1, q.desc2 = _np_ptr(d1), _np_ptr(d2)
        q.valid1, q.avail2, q.stereo1, q.stereo2 = (_np_ptr(a) for a in fl)
        q.fv1_nodes, q.fv1_node, q.fv1_begin, q.fv1_feat = len(f1[0]), _np_ptr(f1[0]), _np_ptr(f1[1]), _np_ptr(f1[2])
        q.fv2_nodes, q.fv2_node, q.fv2_begin, q.fv2_feat = len(f2[0]), _np_ptr(f2[0]), _np_ptr(f2[1]), _np_ptr(f2[2])
        q.kp1, q.kp2 = _np_ptr(k1), _np_ptr(k2)
        q.scale_factors2, q.level_sigma2_2, q.n_levels2 = _np_ptr(sc), _np_ptr(sg), len(sc)
        q.F12[:] = [float(x) for x in np.asarray(p["F12"], np.float32).reshape(9)]
        q.ep[:] = [float(x) for x in np.asarray(p["ep"], np.float32).reshape(2)]
        q.match12 = _np_ptr(m12)
    ms = C.c_float()
    _check(lb.msorb_search_for_triangulation(device, C.addressof(arr), len(pairs), int(bool(coarse)),
                                             int(bool(check_orientation)), C.addressof(ms)), "msorb_search_for_triangulation")
    return [(arr[k].nmatches, o[0][:o[1]]) for k, o in enumerate(outs)], ms.value


EXPORTS = EXPORTS + ("msorb_kf_store_create", "msorb_kf_store_destroy", "msorb_kf_store_count", "msorb_kf_store_add",
                     "msorb_kf_store_remove", "msorb_kf_store_rows", "msorb_search_by_bow_kf", "msorb_search_for_triangulation_kf")


class BowKfPair(C.Structure):
    _fields_ = [("kf1", C.c_int), ("kf2", C.c_int), ("valid1", C.c_void_p), ("avail2", C.c_void_p), ("match12", C.c_void_p),
                ("match21", C.c_void_p), ("nmatches", C.c_int)]


class BowFrame(C.Structure):
    _fields_ = [("n", C.c_int), ("desc", C.c_void_p), ("fv_nodes", C.c_int), ("fv_node", C.c_void_p), ("fv_begin", C.c_void_p),
                ("fv_feat", C.c_void_p), ("angle", C.c_void_p)]


class TriangulationKfPair(C.Structure):
    _fields_ = [("kf1", C.c_int), ("kf2", C.c_int), ("valid1", C.c_void_p), ("avail2", C.c_void_p), ("stereo1", C.c_void_p),
                ("stereo2", C.c_void_p), ("F12", C.c_float * 9), ("ep", C.c_float * 2), ("match12", C.c_void_p), ("nmatches", C.c_int)]


class KeyFrameStore:
    """msorb_kf_store: KeyFrame descriptors / keypoints / FeatureVectors resident on the device for the BoW-node searches."""

    def __init__(self, device=0):
        self.L = lib()
        h = C.c_void_p()
        self.L.msorb_kf_store_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        _check(self.L.msorb_kf_store_create(device, C.byref(h)), "msorb_kf_store_create")
        self.h = h
        self.n = {}

    def close(self):
        if getattr(self, "h", None):
            self.L.msorb_kf_store_destroy.argtypes = [C.c_void_p]
            self.L.msorb_kf_store_destroy.restype = None
            self.L.msorb_kf_store_destroy(self.h)
            self.h = None

    def count(self):
        self.L.msorb_kf_store_count.argtypes = [C.c_void_p]
        return self.L.msorb_kf_store_count(self.h)

    def add(self, kps, desc, fv, scale_factors, level_sigma2):
        k, d = np.ascontiguousarray(kps, KP_DTYPE), _c(desc, np.uint8).reshape(-1, 32)
        f = [_c(a, np.int32) for a in fv]
        sc, sg = _c(scale_factors, np.float32), _c(level_sigma2, np.float32)
        kid = C.c_int(-1)
        vp = C.c_void_p
        self.L.msorb_kf_store_add.argtypes = [vp, C.c_int, vp, vp, C.c_int, vp, vp, vp, vp, vp, C.c_int, vp]
        _check(self.L.msorb_kf_store_add(self.h, len(k), _np_ptr(k), _np_ptr(d), len(f[0]), _np_ptr(f[0]), _np_ptr(f[1]), _np_ptr(f[2]),
                                         _np_ptr(sc), _np_ptr(sg), len(sc), C.byref(kid)), "msorb_kf_store_add")
        self.n[kid.value] = len(k)
        return kid.value

    def remove(self, kf_id):
        self.L.msorb_kf_store_remove.argtypes = [C.c_void_p, C.c_int]
        _check(self.L.msorb_kf_store_remove(self.h, kf_id), "msorb_kf_store_remove")   # (the id and its rows go to the next add, which
        # overwrites self.n[id]; until then the stale size lets a search on the dead id reach the library, which refuses it)

    def rows(self):
        """(feature rows in use, rows reserved on the device)"""
        a, b = C.c_size_t(), C.c_size_t()
        self.L.msorb_kf_store_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _check(self.L.msorb_kf_store_rows(self.h, C.byref(a), C.byref(b)), "msorb_kf_store_rows")
        return a.value, b.value

    def search_by_bow(self, pairs, frame=None, th_low=50, inclusive=True, nnratio=0.7, check_orientation=True):
        """pairs: dicts kf1, kf2 (or -1 with `frame`), valid1, avail2 (or None).  frame: dict desc, fv, angle.
        -> (list of (nmatches, match12, match21), kernel_ms)"""
        arr = (BowKfPair * max(len(pairs), 1))()
        keep, outs = [], []
        fr = None
        if frame is not None:
            fd = _c(frame["desc"], np.uint8).reshape(-1, 32)
            ff = [_c(a, np.int32) for a in frame["fv"]]
            fa = _c(frame["angle"], np.float32)
            fr = BowFrame(len(fd), _np_ptr(fd), len(ff[0]), _np_ptr(ff[0]), _np_ptr(ff[1]), _np_ptr(ff[2]), _np_ptr(fa))
            keep.append((fd, ff, fa))
        for k, p in enumerate(pairs):
            n1 = self.n[p["kf1"]]
            n2 = len(frame["desc"]) if p["kf2"] < 0 else self.n[p["kf2"]]
            v1 = _c(p["valid1"], np.uint8)
            a2 = None if p.get("avail2") is None else _c(p["avail2"], np.uint8)
            m12, m21 = np.zeros(max(n1, 1), np.int32), np.zeros(max(n2, 1), np.int32)
            keep.append((v1, a2))
            outs.append((m12, m21, n1, n2))
            q = arr[k]
            q.kf1, q.kf2, q.valid1 = p["kf1"], p["kf2"], _np_ptr(v1)
            q.avail2 = None if a2 is None else _np_ptr(a2)
            q.match12, q.match21 = _np_ptr(m12), _np_ptr(m21)
        ms = C.c_float()
        vp = C.c_void_p
        self.L.msorb_search_by_bow_kf.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_float, C.c_int, vp]
        _check(self.L.msorb_search_by_bow_kf(self.h, C.addressof(arr), len(pairs), None if fr is None else C.addressof(fr), int(th_low),
                                             int(bool(inclusive)), float(nnratio), int(bool(check_orientation)), C.addressof(ms)),
               "msorb_search_by_bow_kf")
        return [(arr[k].nmatches, o[0][:o[2]], o[1][:o[3]]) for k, o in enumerate(outs)], ms.value

    def search_for_triangulation(self, pairs, coarse=False, check_orientation=True):
        """pairs: dicts kf1, kf2, valid1, avail2, stereo1, stereo2, F12, ep.  -> (list of (nmatches, match12), kernel_ms)"""
        arr = (TriangulationKfPair * max(len(pairs), 1))()
        keep, outs = [], []
        for k, p in enumerate(pairs):
            fl = [_c(p[key], np.uint8) for key in ("valid1", "avail2", "stereo1", "stereo2")]
            m12 = np.zeros(max(self.n[p["kf1"]], 1), np.int32)
            keep.append(fl)
            outs.append((m12, self.n[p["kf1"]]))
            q = arr[k]
            q.kf1, q.kf2 = p["kf1"], p["kf2"]
            q.valid1, q.avail2, q.stereo1, q.stereo2 = (_np_ptr(a) for a in fl)
            q.F12[:] = [float(x) for x in np.asarray(p["F12"], np.float32).reshape(9)]
            q.ep[:] = [float(x) for x in np.asarray(p["ep"], np.float32).reshape(2)]
            q.match12 = _np_ptr(m12)
        ms = C.c_float()
        vp = C.c_void_p
        self.L.msorb_search_for_triangulation_kf.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp]
        _check(self.L.msorb_search_for_triangulation_kf(self.h, C.addressof(arr), len(pairs), int(bool(coarse)),
                                                        int(bool(check_orientation)), C.addressof(ms)),
               "msorb_search_for_triangulation_kf")
        return [(arr[k].nmatches, o[0][:o[1]]) for k, o in enumerate(outs)], ms.value


EXPORTS = EXPORTS + ("msorb_stereo_matches_batch",)


def stereo_matches_batch(ex, counts, d_kps, d_desc, mb, mbf):
    """msorb_stereo_matches_batch on the outputs of ex.extract_batch (images 2p / 2p+1 = left / right of pair p).
    -> (d_u_right, d_depth torch.float32 [n_pairs, cap], n_oob np.int32 [n_pairs], kernel_ms); the outputs stay on the
    device, entries at and past counts[2p] are -1."""
    import torch
    lb = lib()
    lb.msorb_stereo_matches_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                              C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    counts = np.ascontiguousarray(counts, np.int32)
    n_pairs = len(counts) // 2
    cap = d_kps.shape[1]
    d_counts = torch.from_numpy(counts).to(d_kps.device)
    d_ur = torch.full((n_pairs, cap), -1.0, dtype=torch.float32, device=d_kps.device)
    d_dp = torch.full((n_pairs, cap), -1.0, dtype=torch.float32, device=d_kps.device)
    d_oob = torch.zeros(max(n_pairs, 1), dtype=torch.int32, device=d_kps.device)
    torch.cuda.synchronize()
    ms = C.c_float()
    max_left = int(counts[0::2][:n_pairs].max()) if n_pairs else 0
    _check(lb.msorb_stereo_matches_batch(ex.h, n_pairs, d_kps.data_ptr(), d_desc.data_ptr(), cap, d_counts.data_ptr(),
                                         max_left, mb, mbf, d_ur.data_ptr(), d_dp.data_ptr(), d_oob.data_ptr(),
                                         C.addressof(ms)), "msorb_stereo_matches_batch")
    return d_ur, d_dp, d_oob[:n_pairs].cpu().numpy(), ms.value


def stereo_matches_split(ex_left, ex_right, counts_left, d_kps_left, d_desc_left, counts_right, d_kps_right, d_desc_right,
                         mb, mbf):
    """msorb_stereo_matches_split: left images = the last batch of ex_left, right images = the last batch (extract_batch or
    pyramid_batch) of ex_right, both on one device; the right keypoints / descriptors / counts may come from elsewhere
    (gathered).  counts_* are numpy arrays or CUDA int32 tensors.  -> (d_u_right, d_depth [n_pairs, cap], n_oob, kernel_ms)."""
    import torch
    lb = lib()
    vp = C.c_void_p
    lb.msorb_stereo_matches_split.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_float, C.c_float,
                                              vp, vp, vp, vp]
    dev = d_kps_left.device

    def dcount(c):
        return c if torch.is_tensor(c) else torch.from_numpy(np.ascontiguousarray(c, np.int32)).to(dev)
    d_cl, d_cr = dcount(counts_left), dcount(counts_right)
    n_pairs = int(d_cl.numel())
    cap = d_kps_left.shape[1]
    d_ur = torch.full((n_pairs, cap), -1.0, dtype=torch.float32, device=dev)
    d_dp = torch.full((n_pairs, cap), -1.0, dtype=torch.float32, device=dev)
    d_oob = torch.zeros(max(n_pairs, 1), dtype=torch.int32, device=dev)
    max_left = int(d_cl.max().item()) if n_pairs else 0
    torch.cuda.synchronize(dev)
    ms = C.c_float()
    _check(lb.msorb_stereo_matches_split(ex_left.h, ex_right.h, n_pairs, d_kps_left.data_ptr(), d_desc_left.data_ptr(),
                                         d_cl.data_ptr(), d_kps_right.data_ptr(), d_desc_right.data_ptr(), d_cr.data_ptr(), cap,
                                         max_left, mb, mbf, d_ur.data_ptr(), d_dp.data_ptr(), d_oob.data_ptr(),
                                         C.addressof(ms)), "msorb_stereo_matches_split")
    return d_ur, d_dp, d_oob[:n_pairs].cpu().numpy(), ms.value


EXPORTS = EXPORTS + ("msorb_fuse_search", "msorb_fuse_search_gated", "msorb_search_by_projection_kf", "msorb_search_by_projection_sim3",
                     "msorb_extract_batch_submit", "msorb_extract_batch_wait")


# ------------------------------------------------------------------------------------------------
# The tracking front-end as one device-resident chain (include/msorb.h, last section; csrc/track.hip)
# ------------------------------------------------------------------------------------------------
EXPORTS = EXPORTS + ("msorb_frame_set_device", "msorb_frame_grid", "msorb_extract_stereo_frame", "msorb_search_local_points",
                     "msorb_track_frontend", "msorb_track_batch")
GRID_COLS, GRID_ROWS = 64, 48   # FRAME_GRID_COLS / FRAME_GRID_ROWS, Frame.h:44-45


def _empty_frame(device=0):
    f = Frame.__new__(Frame)
    f.L = _mlib()
    h = C.c_void_p()
    _check(f.L.msorb_frame_create(device, C.byref(h)), "msorb_frame_create")
    f.h = h
    f.n = 0
    f.kps = np.zeros(0, KP_DTYPE)
    f.desc = np.zeros((0, 32), np.uint8)
    return f


def frame_from_device(d_kps, n, d_desc, d_u_right, bounds, scale_factors, device=0):
    """msorb_frame_set_device: d_kps torch.uint8 [>= n, 28], d_desc torch.uint8 [>= n, 32], d_u_right torch.float32 or None."""
    f = _empty_frame(device)
    sf = _c(scale_factors, np.float32)
    f.L.msorb_frame_set_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float,
                                           C.c_float, C.c_void_p, C.c_int]
    _check(f.L.msorb_frame_set_device(f.h, d_kps.data_ptr(), n, d_desc.data_ptr(), None if d_u_right is None else d_u_right.data_ptr(),
                                      bounds[0], bounds[1], bounds[2], bounds[3], _np_ptr(sf), len(sf)), "msorb_frame_set_device")
    f.n = n
    f.kps = np.frombuffer(d_kps[:n].cpu().numpy().tobytes(), KP_DTYPE).copy()
    f.desc = d_desc[:n].cpu().numpy().copy()
    return f


def frame_grid(frame):
    """msorb_frame_grid -> (cell_begin[64*48+1], cell_idx): mGrid as CSR, cell = ix*48 + iy, insertion order inside a cell."""
    cb = np.zeros(GRID_COLS * GRID_ROWS + 1, np.int32)
    ci = np.zeros(max(frame.n, 1), np.int32)
    na = C.c_int()
    frame.L.msorb_frame_grid.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    _check(frame.L.msorb_frame_grid(frame.h, _np_ptr(cb), _np_ptr(ci), len(ci), C.byref(na)), "msorb_frame_grid")
    return cb, ci[:na.value].copy()


_LP_KEYS = ("pos_w", "normal", "max_distance", "min_distance", "visit", "bad", "sparsified", "desc", "obs")


def _lp_arrays(mp):
    m = len(mp["max_distance"])
    visit = mp.get("visit")
    arrs = [_c(mp["pos_w"], np.float32).reshape(-1), _c(mp["normal"], np.float32).reshape(-1), _c(mp["max_distance"], np.float32),
            _c(mp["min_distance"], np.float32), None if visit is None else _c(visit, np.uint8), _c(mp["bad"], np.uint8),
            _c(mp["sparsified"], np.uint8), _c(mp["desc"], np.uint8), _c(mp["obs"], np.int32)]
    return m, arrs


def _lp_outputs(m):
    cap = max(m, 1)
    return dict(track_in_view=np.zeros(cap, np.uint8), proj_x=np.zeros(cap, np.float32), proj_y=np.zeros(cap, np.float32),
                proj_xr=np.zeros(cap, np.float32), track_depth=np.zeros(cap, np.float32), level=np.zeros(cap, np.int32),
                view_cos=np.zeros(cap, np.float32))


_LP_OUT_ORDER = ("track_in_view", "proj_x", "proj_y", "proj_xr", "track_depth", "level", "view_cos")


def search_local_points(frame, frustum, mp, frame_mp, th, bFarPoints=False, thFarPoints=50.0, nnratio=0.8,
                        viewing_cos_limit=0.5):
    """msorb_search_local_points: Tracking::SearchLocalPoints' isInFrustum loop + SearchByProjection as one device chain.
    mp: dict(pos_w, normal, max_distance, min_distance, [visit], bad, sparsified, desc, obs).  frame_mp updated in place.
    -> (nmatches, dict of the isInFrustum scratch)."""
    L = frame.L
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    L.msorb_search_local_points.argtypes = [vp, vp, cf, ci] + [vp] * 10 + [cf, ci, cf, cf] + [vp] * 7 + [C.POINTER(ci)]
    m, arrs = _lp_arrays(mp)
    out = _lp_outputs(m)
    assert frame_mp.dtype == np.int32 and frame_mp.flags.c_contiguous
    nm = C.c_int()
    _check(L.msorb_search_local_points(frame.h, C.addressof(frustum), viewing_cos_limit, m,
                                       *[None if a is None else _np_ptr(a) for a in arrs], _np_ptr(frame_mp), th, int(bFarPoints),
                                       thFarPoints, nnratio, *[_np_ptr(out[k]) for k in _LP_OUT_ORDER], C.byref(nm)),
           "msorb_search_local_points")
    return nm.value, {k: v[:m] for k, v in out.items()}


def _stereo_frame_call(ex, fn_name, left, right, mb, mbf, bounds, extra_argtypes, extra_args, device, prepare=None):
    left, right = np.ascontiguousarray(left, np.uint8), np.ascontiguousarray(right, np.uint8)
    assert left.shape == right.shape and left.ndim == 2
    rows, cols = left.shape
    cap = ex.capacity
    kl, kr = np.zeros(cap, KP_DTYPE), np.zeros(cap, KP_DTYPE)
    dl, dr = np.zeros((cap, 32), np.uint8), np.zeros((cap, 32), np.uint8)
    ur, dp = np.zeros(cap, np.float32), np.zeros(cap, np.float32)
    nl, nr, oob = C.c_int(0), C.c_int(0), C.c_int(0)
    f = _empty_frame(device)
    if prepare is not None:
        prepare(f)
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    fn = getattr(ex.L, fn_name)
    fn.argtypes = [vp, vp, vp, vp, ci, ci, C.c_size_t, C.c_size_t, cf, cf, vp, vp, vp, vp, vp, vp, ci, vp, vp, vp, cf, cf, cf,
                   cf] + extra_argtypes
    _check(fn(ex.h, f.h, _np_ptr(left), _np_ptr(right), rows, cols, cols, cols, mb, mbf, _np_ptr(kl), _np_ptr(dl), C.byref(nl),
              _np_ptr(kr), _np_ptr(dr), C.byref(nr), cap, _np_ptr(ur), _np_ptr(dp), C.byref(oob), bounds[0], bounds[1], bounds[2],
              bounds[3], *extra_args), fn_name)
    a, b = nl.value, nr.value
    f.n = a
    f.kps = kl[:a].copy()
    f.desc = dl[:a].copy()
    return f, (kl[:a].copy(), dl[:a].copy(), kr[:b].copy(), dr[:b].copy(), ur[:a].copy(), dp[:a].copy(), oob.value)


def extract_stereo_frame(ex, left, right, mb, mbf, bounds=None, device=0):
    """msorb_extract_stereo_frame -> (Frame, (kps_left, desc_left, kps_right, desc_right, mvuRight, mvDepth, n_oob))."""
    if bounds is None:
        bounds = (0.0, float(left.shape[1]), 0.0, float(left.shape[0]))
    return _stereo_frame_call(ex, "msorb_extract_stereo_frame", left, right, mb, mbf, bounds, [], [], device)


def track_frontend(ex, left, right, mb, mbf, frustum, mp, th, bFarPoints=False, thFarPoints=50.0, nnratio=0.8,
                   viewing_cos_limit=0.5, bounds=None, device=0):
    """msorb_track_frontend -> (Frame, stereo outputs as extract_stereo_frame, frame_mp[n_left], nmatches, scratch dict, rounds)."""
    if bounds is None:
        bounds = (0.0, float(left.shape[1]), 0.0, float(left.shape[0]))
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    m, arrs = _lp_arrays(mp)
    out = _lp_outputs(m)
    frame_mp = np.full(ex.capacity, -1, np.int32)
    nm, rounds = C.c_int(), C.c_int()
    extra_t = [vp, cf, ci] + [vp] * 10 + [cf, ci, cf, cf] + [vp] * 7 + [C.POINTER(ci), C.POINTER(ci)]
    extra = [C.addressof(frustum), viewing_cos_limit, m] + [None if a is None else _np_ptr(a) for a in arrs] + \
        [_np_ptr(frame_mp), th, int(bFarPoints), thFarPoints, nnratio] + [_np_ptr(out[k]) for k in _LP_OUT_ORDER] + \
        [C.byref(nm), C.byref(rounds)]
    f, st = _stereo_frame_call(ex, "msorb_track_frontend", left, right, mb, mbf, bounds, extra_t, extra, device)
    return f, st, frame_mp[:f.n].copy(), nm.value, {k: v[:m] for k, v in out.items()}, rounds.value


def track_batch(d_kps, d_desc, d_u_right, counts, frame_step, bounds, scale_factors, frusta, d_mp, th, bFarPoints=False,
                thFarPoints=50.0, viewing_cos_limit=0.5, want_grid=False, count_pairs=False, device=0):
    """msorb_track_batch.  d_kps / d_desc: extract_batch outputs; counts: their host counts (all images); frame b = image
    b*frame_step; d_u_right torch.float32 [n_frames, cap] or None; frusta: list of Frustum; d_mp: dict of CUDA tensors pos_w
    [B, m, 3], normal [B, m, 3], max_distance [B, m], min_distance [B, m], flags uint8 [B, m], desc uint8 [B, m, 32].
    -> dict(topk_idx [B, m, 8], topk_dist [B, m, 8], in_view [B, m], ms (grid, frustum+queries, window search), n_pairs,
            [cell_begin, cell_idx])  (torch CUDA tensors)."""
    import torch
    lb = lib()
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    lb.msorb_track_batch.argtypes = [ci, ci, vp, vp, vp, vp, ci, ci, cf, cf, cf, cf, vp, ci, vp, cf, ci, vp, vp, vp, vp, vp, vp, cf,
                                     ci, cf, vp, vp, vp, vp, vp, vp]
    B = len(frusta)
    cap = d_kps.shape[1]
    dev = d_kps.device
    m = int(d_mp["max_distance"].shape[1]) if B else 0
    sf = _c(scale_factors, np.float32)
    d_counts = torch.from_numpy(np.ascontiguousarray(counts, np.int32)).to(dev)
    fr = (Frustum * max(B, 1))(*frusta)
    d_topk = torch.empty((B, m, 16), dtype=torch.int32, device=dev)
    d_inview = torch.empty((B, m), dtype=torch.uint8, device=dev)
    d_cb = torch.empty((B, GRID_COLS * GRID_ROWS + 1), dtype=torch.int32, device=dev) if want_grid else None
    d_ci = torch.full((B, cap), -1, dtype=torch.int32, device=dev) if want_grid else None
    ms = (C.c_float * 3)()
    npairs = C.c_ulonglong(0)
    for k in ("pos_w", "normal", "max_distance", "min_distance", "flags", "desc"):
        assert d_mp[k].is_contiguous()
    torch.cuda.synchronize(dev)
    _check(lb.msorb_track_batch(device, B, d_kps.data_ptr(), d_desc.data_ptr(), None if d_u_right is None else d_u_right.data_ptr(),
                                d_counts.data_ptr(), frame_step, cap, bounds[0], bounds[1], bounds[2], bounds[3], _np_ptr(sf), len(sf),
                                C.addressof(fr), viewing_cos_limit, m, d_mp["pos_w"].data_ptr(), d_mp["normal"].data_ptr(),
                                d_mp["max_distance"].data_ptr(), d_mp["min_distance"].data_ptr(), d_mp["flags"].data_ptr(),
                                d_mp["desc"].data_ptr(), th, int(bFarPoints), thFarPoints, d_topk.data_ptr(), d_inview.data_ptr(),
                                None if d_cb is None else d_cb.data_ptr(), None if d_ci is None else d_ci.data_ptr(), C.addressof(ms),
                                C.addressof(npairs) if count_pairs else None), "msorb_track_batch")
    r = dict(topk_idx=d_topk[:, :, :8], topk_dist=d_topk[:, :, 8:], in_view=d_inview, ms=tuple(ms), n_pairs=npairs.value)
    if want_grid:
        r["cell_begin"], r["cell_idx"] = d_cb, d_ci
    return r


class TrackFrontendRunner:
    """msorb_track_frontend / msorb_extract_stereo_frame + msorb_search_local_points with every buffer and ctypes argument
    prepared once — the per-frame cost that remains is the library call itself (what bench.py times at B = 1)."""

    def __init__(self, ex, left, right, mb, mbf, frustum, mp, th, bFarPoints=False, thFarPoints=50.0, nnratio=0.8,
                 viewing_cos_limit=0.5, device=0):
        vp, ci, cf = C.c_void_p, C.c_int, C.c_float
        self.ex, self.L = ex, ex.L
        self.left, self.right = np.ascontiguousarray(left, np.uint8), np.ascontiguousarray(right, np.uint8)
        rows, cols = self.left.shape
        cap = ex.capacity
        self.f = _empty_frame(device)
        self.kl, self.kr = np.zeros(cap, KP_DTYPE), np.zeros(cap, KP_DTYPE)
        self.dl, self.dr = np.zeros((cap, 32), np.uint8), np.zeros((cap, 32), np.uint8)
        self.ur, self.dp = np.zeros(cap, np.float32), np.zeros(cap, np.float32)
        self.nl, self.nr, self.oob, self.nm, self.rounds = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        self.m, self.arrs = _lp_arrays(mp)
        self.out = _lp_outputs(self.m)
        self.frame_mp = np.full(cap, -1, np.int32)
        self.frustum = frustum
        stereo_t = [vp, vp, vp, vp, ci, ci, C.c_size_t, C.c_size_t, cf, cf, vp, vp, vp, vp, vp, vp, ci, vp, vp, vp, cf, cf, cf, cf]
        lp_t = [vp, cf, ci] + [vp] * 10 + [cf, ci, cf, cf] + [vp] * 7 + [C.POINTER(ci)]
        self.L.msorb_track_frontend.argtypes = stereo_t + lp_t + [C.POINTER(ci)]
        self.L.msorb_extract_stereo_frame.argtypes = stereo_t
        self.L.msorb_search_local_points.argtypes = [vp] + lp_t
        self._stereo = (ex.h, self.f.h, _np_ptr(self.left), _np_ptr(self.right), rows, cols, cols, cols, mb, mbf, _np_ptr(self.kl),
                        _np_ptr(self.dl), C.byref(self.nl), _np_ptr(self.kr), _np_ptr(self.dr), C.byref(self.nr), cap, _np_ptr(self.ur),
                        _np_ptr(self.dp), C.byref(self.oob), 0.0, float(cols), 0.0, float(rows))
        self._lp = (C.addressof(frustum), viewing_cos_limit, self.m) + tuple(None if a is None else _np_ptr(a) for a in self.arrs) + \
            (_np_ptr(self.frame_mp), th, int(bFarPoints), thFarPoints, nnratio) + tuple(_np_ptr(self.out[k]) for k in _LP_OUT_ORDER) + \
            (C.byref(self.nm),)

    def one_call(self):
        _check(self.L.msorb_track_frontend(*self._stereo, *self._lp, C.byref(self.rounds)), "msorb_track_frontend")
        return self.nm.value

    def two_calls(self):
        _check(self.L.msorb_extract_stereo_frame(*self._stereo), "msorb_extract_stereo_frame")
        self.frame_mp[:] = -1
        _check(self.L.msorb_search_local_points(self.f.h, *self._lp), "msorb_search_local_points")
        return self.nm.value

    def close(self):
        self.f.close()


EXPORTS = EXPORTS + ("msorb_frame_set_last_points", "msorb_frame_last_points_count", "msorb_search_last_frame", "msorb_track_frontend_motion")


class MotionModel(C.Structure):
    """msorb_motion_model: Tcw (unit quaternion x, y, z, w + translation), pinhole parameters, mbf, bForward / bBackward"""
    _fields_ = [("q", C.c_float * 4), ("t", C.c_float * 3), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float),
                ("cy", C.c_float), ("mbf", C.c_float), ("forward", C.c_int), ("backward", C.c_int)]

    @classmethod
    def make(cls, q_xyzw, t, fx, fy, cx, cy, mbf, forward=False, backward=False):
        m = cls()
        m.q[:] = [float(np.float32(v)) for v in q_xyzw]
        m.t[:] = [float(np.float32(v)) for v in t]
        m.fx, m.fy, m.cx, m.cy, m.mbf = fx, fy, cx, cy, mbf
        m.forward, m.backward = int(forward), int(backward)
        return m


def _last_arrays(last):
    return [_c(last["has_point"], np.uint8), _c(last["pos_w"], np.float32).reshape(-1), _c(last["octave"], np.int32),
            _c(last["angle"], np.float32), _c(last["desc"], np.uint8)]


def frame_set_last_points(frame, last):
    """msorb_frame_set_last_points; last: dict(has_point, pos_w [n, 3], octave, angle, desc [n, 32]) — LastFrame's side of
    ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono)."""
    arrs = _last_arrays(last)
    frame.L.msorb_frame_set_last_points.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5
    _check(frame.L.msorb_frame_set_last_points(frame.h, len(arrs[0]), *[_np_ptr(a) for a in arrs]), "msorb_frame_set_last_points")


def search_last_frame(frame, mm, obs, cur_mp, th, check_orientation=True, want_projection=False):
    """msorb_search_last_frame on the resident table -> nmatches (cur_mp updated in place) [, dict(valid, u, v, ur)]."""
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    frame.L.msorb_search_last_frame.argtypes = [vp, vp, vp, ci, vp, cf, ci, C.POINTER(ci), vp, vp, vp, vp]
    ob = _c(obs, np.int32)
    assert cur_mp.dtype == np.int32 and cur_mp.flags.c_contiguous
    nm = C.c_int()
    n_obs = len(ob)
    frame.L.msorb_frame_last_points_count.argtypes = [vp]
    n = frame.L.msorb_frame_last_points_count(frame.h)      # the projection arrays have one entry per last-frame keypoint
    if n < 0:
        raise MsorbError(E_INVALID, "msorb_search_last_frame: no last-frame table on the handle (msorb_frame_set_last_points)")
    pv = np.zeros(max(n, 1), np.uint8)
    pu, pvv, pur = [np.zeros(max(n, 1), np.float32) for _ in range(3)]
    proj = [_np_ptr(a) for a in (pv, pu, pvv, pur)] if want_projection else [None] * 4
    _check(frame.L.msorb_search_last_frame(frame.h, C.addressof(mm), _np_ptr(ob), n_obs, _np_ptr(cur_mp), th, int(check_orientation),
                                           C.byref(nm), *proj), "msorb_search_last_frame")
    if want_projection:
        return nm.value, dict(valid=pv[:n], u=pu[:n], v=pvv[:n], ur=pur[:n])
    return nm.value


def track_frontend_motion(ex, left, right, mb, mbf, mm, last, obs, th, check_orientation=True, bounds=None, device=0):
    """msorb_frame_set_last_points + msorb_track_frontend_motion -> (Frame, stereo outputs, cur_mp[n_left], nmatches)."""
    if bounds is None:
        bounds = (0.0, float(left.shape[1]), 0.0, float(left.shape[0]))
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    ob = _c(obs, np.int32)
    cur_mp = np.full(ex.capacity, -1, np.int32)
    nm = C.c_int()
    fr = [None]

    def prepare(f):
        frame_set_last_points(f, last)
        fr[0] = f
    f, st = _stereo_frame_call(ex, "msorb_track_frontend_motion", left, right, mb, mbf, bounds, [vp, vp, vp, cf, ci, C.POINTER(ci)],
                               [C.addressof(mm), _np_ptr(ob), _np_ptr(cur_mp), th, int(check_orientation), C.byref(nm)], device,
                               prepare=prepare)
    return f, st, cur_mp[:f.n].copy(), nm.value


class MotionFrontendRunner:
    """configs[2], first half of a tracking frame with every buffer prepared once: msorb_track_frontend_motion (Frame::Frame +
    TrackWithMotionModel's SearchByProjection as one call) — or the same as three calls (table upload, extraction, search)."""

    def __init__(self, ex, left, right, mb, mbf, mm, last, obs, th, check_orientation=True, device=0):
        vp, ci, cf = C.c_void_p, C.c_int, C.c_float
        self.ex, self.L = ex, ex.L
        self.left, self.right = np.ascontiguousarray(left, np.uint8), np.ascontiguousarray(right, np.uint8)
        rows, cols = self.left.shape
        cap = ex.capacity
        self.f = _empty_frame(device)
        self.kl, self.kr = np.zeros(cap, KP_DTYPE), np.zeros(cap, KP_DTYPE)
        self.dl, self.dr = np.zeros((cap, 32), np.uint8), np.zeros((cap, 32), np.uint8)
        self.ur, self.dp = np.zeros(cap, np.float32), np.zeros(cap, np.float32)
        self.nl, self.nr, self.oob, self.nm = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        self.mm, self.obs = mm, _c(obs, np.int32)
        self.last = _last_arrays(last)
        self.cur_mp = np.full(cap, -1, np.int32)
        stereo_t = [vp, vp, vp, vp, ci, ci, C.c_size_t, C.c_size_t, cf, cf, vp, vp, vp, vp, vp, vp, ci, vp, vp, vp, cf, cf, cf, cf]
        self.L.msorb_track_frontend_motion.argtypes = stereo_t + [vp, vp, vp, cf, ci, C.POINTER(ci)]
        self.L.msorb_extract_stereo_frame.argtypes = stereo_t
        self.L.msorb_frame_set_last_points.argtypes = [vp, ci] + [vp] * 5
        self.L.msorb_search_last_frame.argtypes = [vp, vp, vp, ci, vp, cf, ci, C.POINTER(ci), vp, vp, vp, vp]
        self._stereo = (ex.h, self.f.h, _np_ptr(self.left), _np_ptr(self.right), rows, cols, cols, cols, mb, mbf, _np_ptr(self.kl),
                        _np_ptr(self.dl), C.byref(self.nl), _np_ptr(self.kr), _np_ptr(self.dr), C.byref(self.nr), cap, _np_ptr(self.ur),
                        _np_ptr(self.dp), C.byref(self.oob), 0.0, float(cols), 0.0, float(rows))
        self._set = (self.f.h, len(self.last[0])) + tuple(_np_ptr(a) for a in self.last)
        self._mm = (C.addressof(mm), _np_ptr(self.obs), _np_ptr(self.cur_mp), th, int(check_orientation), C.byref(self.nm))
        self._search = (self.f.h, C.addressof(mm), _np_ptr(self.obs), len(self.obs), _np_ptr(self.cur_mp), th, int(check_orientation),
                        C.byref(self.nm), None, None, None, None)

    def one_call(self):
        _check(self.L.msorb_frame_set_last_points(*self._set), "msorb_frame_set_last_points")
        _check(self.L.msorb_track_frontend_motion(*self._stereo, *self._mm), "msorb_track_frontend_motion")
        return self.nm.value

    def separate_calls(self):
        _check(self.L.msorb_frame_set_last_points(*self._set), "msorb_frame_set_last_points")
        _check(self.L.msorb_extract_stereo_frame(*self._stereo), "msorb_extract_stereo_frame")
        self.cur_mp[:] = -1
        _check(self.L.msorb_search_last_frame(*self._search), "msorb_search_last_frame")
        return self.nm.value

    def search_only(self):
        self.cur_mp[:] = -1
        _check(self.L.msorb_search_last_frame(*self._search), "msorb_search_last_frame")
        return self.nm.value

    def attach_local_points(self, frustum, mp, th, bFarPoints=False, thFarPoints=50.0, nnratio=0.8, viewing_cos_limit=0.5):
        """prepare the second call of the frame: msorb_search_local_points (TrackLocalMap's SearchLocalPoints) on the same handle.
        As in Tracking::Track the points TrackWithMotionModel matched stay in mvpMapPoints: their keypoints are occupied
        (ORBmatcher.cc:99-101) and the points themselves are part of the local map, already seen in this frame and skipped by
        SearchLocalPoints' loop (Tracking.cc:3316-3340: mnLastFrameSeen == mCurrentFrame.mnId).  The local-map table is a private
        copy whose first rows stand for those held points (visit = 0, Observations() >= 1); frame_total() points frame_mp at them."""
        vp, ci, cf = C.c_void_p, C.c_int, C.c_float
        mp = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in mp.items()}
        self.one_call()
        self.held_kp = np.flatnonzero(self.cur_mp[:self.nl.value] >= 0).astype(np.int64)[:len(mp["max_distance"])]
        rows = np.arange(len(self.held_kp))
        if mp.get("visit") is None:
            mp["visit"] = np.ones(len(mp["max_distance"]), np.uint8)
        mp["visit"][rows] = 0
        mp["bad"][rows] = 0
        mp["obs"][rows] = np.maximum(mp["obs"][rows], 1)
        self.m, self.arrs = _lp_arrays(mp)
        self.out = _lp_outputs(self.m)
        self.frame_mp = np.full(self.ex.capacity, -1, np.int32)
        self.frustum, self.nm_lp = frustum, C.c_int(0)
        self.L.msorb_search_local_points.argtypes = [vp, vp, cf, ci] + [vp] * 10 + [cf, ci, cf, cf] + [vp] * 7 + [C.POINTER(ci)]
        self._lp = (self.f.h, C.addressof(frustum), viewing_cos_limit, self.m) + \
            tuple(None if a is None else _np_ptr(a) for a in self.arrs) + \
            (_np_ptr(self.frame_mp), th, int(bFarPoints), thFarPoints, nnratio) + tuple(_np_ptr(self.out[k]) for k in _LP_OUT_ORDER) + \
            (C.byref(self.nm_lp),)

    def frame_total(self):
        """both device calls of one tracking frame: Frame::Frame + TrackWithMotionModel's search, then (the pose optimisation
        of the host sits here in the reference) SearchLocalPoints -> (motion-model matches, local-map matches)"""
        a = self.one_call()
        self.frame_mp[:] = -1
        held = np.flatnonzero(self.cur_mp[:self.nl.value] >= 0)[:self.m]
        self.frame_mp[held] = np.arange(len(held), dtype=np.int32)   # the motion-model matches stay in the frame: occupied keypoints
        _check(self.L.msorb_search_local_points(*self._lp), "msorb_search_local_points")
        return a, self.nm_lp.value

    def close(self):
        self.f.close()


def knn_match2(query, train, device=0):
    """msorb_knn_match2: BFMatcher(NORM_HAMMING).knnMatch(k=2) -> (best_idx, best_dist, second_idx, second_dist)."""
    lb = lib()
    vp, ci = C.c_void_p, C.c_int
    lb.msorb_knn_match2.argtypes = [ci, vp, ci, vp, ci, vp, vp, vp, vp]
    q, t = _c(query, np.uint8).reshape(-1, 32), _c(train, np.uint8).reshape(-1, 32)
    nq = len(q)
    out = [np.zeros(max(nq, 1), np.int32) for _ in range(4)]
    _check(lb.msorb_knn_match2(device, _np_ptr(q), nq, _np_ptr(t), len(t), *[_np_ptr(o) for o in out]), "msorb_knn_match2")
    return tuple(o[:nq] for o in out)


EXPORTS = EXPORTS + ("msorb_frame_search_rounds",)


def frame_search_rounds(frame):
    """msorb_frame_search_rounds -> (rounds of the last claim-replaying search, total rounds, total searches)."""
    tr, ts = C.c_longlong(), C.c_longlong()
    frame.L.msorb_frame_search_rounds.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    r = frame.L.msorb_frame_search_rounds(frame.h, C.addressof(tr), C.addressof(ts))
    return r, tr.value, ts.value
