"""ctypes binding of the C ABI in include/msfm_match.h (libmsfm_match.so, built in-tree).

There is no fallback: if the shared library is missing or no gfx950 GPU is usable, loading /
context creation raises.  Nothing here imports the CPU oracle.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmsfm_match.so")
if os.environ.get("MSFM_LIBRARY"):   # development hook: an alternative build of the same C ABI (tools/variant_bench.sh)
    LIB_PATH = os.path.abspath(os.environ["MSFM_LIBRARY"])

OK, E_INVALID, E_DEVICE, E_NOIMAGE, E_CAPACITY, E_STATE = range(6)
DTYPE_F32, DTYPE_U8 = 0, 1
ORDER_SSE4X4, ORDER_AVX2_FMA, ORDER_AVX512_FMA = 0, 1, 3
MAX_IMAGES = 10000
DIM = 128

EXPORTS = [
    "msfm_create", "msfm_destroy", "msfm_last_error", "msfm_device_info", "msfm_set_accum_order", "msfm_set_prefilter",
    "msfm_get_profile", "msfm_upload_image", "msfm_image_rows", "msfm_clear_images",
    "msfm_match_pair", "msfm_match_pairs", "msfm_fetch_matches", "msfm_knn2_pair",
    "msfm_topscale_select", "msfm_pair_id", "msfm_pair_from_id", "msfm_swap_image_pair",
    "msfm_version", "msfm_upload_keypoints", "msfm_match_pairs_verified", "msfm_subset_image", "msfm_view_matches", "msfm_set_limits", "msfm_fetch_matches_device",
    "msfm_fetch_order_certificate", "msfm_set_pipeline", "msfm_device_count", "msfm_finalize_store", "msfm_store_info",
    "msfm_match_pairs_begin", "msfm_match_pairs_next", "msfm_read_device", "msfm_memory_info", "msfm_match_pairs_end",
]


class MatchParams(C.Structure):
    _fields_ = [("ratio", C.c_float), ("cross_check", C.c_int), ("max_distance", C.c_double)]


class VerifyParams(C.Structure):
    _fields_ = [("threshold", C.c_double), ("confidence", C.c_double), ("max_iters", C.c_int), ("seed", C.c_ulonglong)]


class Profile(C.Structure):
    _fields_ = [("dist_kernel_ms", C.c_double), ("dist_kernel_launches", C.c_int),
                ("total_device_ms", C.c_double), ("descriptor_pairs", C.c_int64),
                ("dist_algo_bytes", C.c_int64), ("approx_kernel_ms", C.c_double),
                ("approx_kernel_launches", C.c_int), ("prefilter_pairs", C.c_int), ("fallback_pairs", C.c_int),
                ("candidates", C.c_int64), ("prefilter_descriptor_pairs", C.c_int64),
                ("exact_descriptor_pairs", C.c_int64), ("tie_rows", C.c_int64),
                ("sweep2_ms", C.c_double), ("sweep2_launches", C.c_int), ("compacted_pairs", C.c_int),
                ("sweep2_descriptor_pairs", C.c_int64), ("verify_ms", C.c_double),
                ("sub_batches", C.c_int), ("tie_queue_regrows", C.c_int), ("plan_regrows", C.c_int),
                ("sweep1_i8_launches", C.c_int), ("sweep1_q8_launches", C.c_int), ("sweep1b_launches", C.c_int),
                ("sweep1b_ms", C.c_double), ("sweep1b_descriptor_pairs", C.c_int64), ("order_sensitive_rows", C.c_int64),
                ("demoted_pairs", C.c_int), ("mixed_route_sub_batches", C.c_int), ("memory_shrinks", C.c_int)]


class Chunk(C.Structure):
    """msfm_chunk: one completed device sub-batch of a streaming series (msfm_match_pairs_begin / _next)."""
    _fields_ = [("first_pair", C.c_int), ("n_pairs", C.c_int), ("count", C.c_int64), ("offsets", C.POINTER(C.c_int64)),
                ("qt", C.POINTER(C.c_int32)), ("dist", C.POINTER(C.c_float)), ("d_qt", C.c_void_p), ("d_dist", C.c_void_p),
                ("sensitive_rows", C.POINTER(C.c_int32))]


class Memory(C.Structure):
    _fields_ = [(k, C.c_int64) for k in ("device_free", "device_total", "store", "inbox", "scratch", "results_device", "page_locked_host")]


class MsfmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("msfm error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load():
    """Load libmsfm_match.so; raises (loudly) if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (or `make -C monocularsfm_amd/csrc`). There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, ip, fp = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_float)
    L.msfm_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.msfm_destroy.argtypes = [vp]
    L.msfm_destroy.restype = None
    L.msfm_last_error.argtypes = [vp]
    L.msfm_last_error.restype = C.c_char_p
    L.msfm_device_info.argtypes = [vp, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.msfm_set_accum_order.argtypes = [vp, C.c_int]
    L.msfm_set_prefilter.argtypes = [vp, C.c_int]
    L.msfm_get_profile.argtypes = [vp, C.POINTER(Profile)]
    L.msfm_set_limits.argtypes = [vp, C.c_int, C.c_int64]
    L.msfm_set_pipeline.argtypes = [vp, C.c_int]
    L.msfm_upload_image.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int]
    L.msfm_image_rows.argtypes = [vp, C.c_int, C.POINTER(C.c_int)]
    L.msfm_subset_image.argtypes = [vp, C.c_int, C.c_int, ip, C.c_int]
    L.msfm_clear_images.argtypes = [vp]
    try:   # (the A/B tools also load older builds of the library: tools/ab.py)
        L.msfm_finalize_store.argtypes = [vp]
        L.msfm_store_info.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.msfm_match_pairs_begin.argtypes = [vp, ip, C.c_int, C.POINTER(MatchParams), C.c_int, C.POINTER(VerifyParams)]
        L.msfm_match_pairs_next.argtypes = [vp, C.POINTER(Chunk)]
        L.msfm_read_device.argtypes = [vp, vp, vp, C.c_int64]
        L.msfm_memory_info.argtypes = [vp, C.POINTER(Memory)]
        L.msfm_match_pairs_end.argtypes = [vp]
    except AttributeError:
        pass
    L.msfm_match_pair.argtypes = [vp, C.c_int, C.c_int, C.c_float, C.c_int, C.c_double, ip, fp, C.POINTER(C.c_int)]
    L.msfm_match_pairs.argtypes = [vp, ip, C.c_int, C.POINTER(MatchParams), C.POINTER(C.c_int64)]
    L.msfm_fetch_matches.argtypes = [vp, ip, fp]
    L.msfm_view_matches.argtypes = [vp, C.POINTER(ip), C.POINTER(fp), C.POINTER(C.c_int64)]
    L.msfm_fetch_matches_device.argtypes = [vp, vp, vp]
    L.msfm_fetch_order_certificate.argtypes = [vp, ip]
    L.msfm_upload_keypoints.argtypes = [vp, C.c_int, fp, C.c_int, C.c_int]
    L.msfm_match_pairs_verified.argtypes = [vp, ip, C.c_int, C.POINTER(MatchParams), C.POINTER(VerifyParams),
                                            C.POINTER(C.c_int64)]
    L.msfm_knn2_pair.argtypes = [vp, C.c_int, C.c_int, ip, fp, fp, ip, fp, fp]
    L.msfm_topscale_select.argtypes = [fp, C.c_int, C.c_int, ip, C.POINTER(C.c_int)]
    L.msfm_pair_id.argtypes = [C.c_int, C.c_int, ip]
    L.msfm_pair_from_id.argtypes = [C.c_int32, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.msfm_swap_image_pair.argtypes = [C.c_int, C.c_int]
    L.msfm_version.restype = C.c_char_p
    try:   # (the A/B tools also load older builds of the library: tools/ab.py)
        L.msfm_device_count.argtypes = []
        L.msfm_device_count.restype = C.c_int
    except AttributeError:
        pass
    for name in EXPORTS:
        getattr(L, name)  # raises AttributeError if the library lacks a declared symbol
    _lib = L
    return L


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Context:
    """One matcher context bound to one GPU (msfm_ctx)."""

    def __init__(self, device=0, order=ORDER_SSE4X4):
        # PyTorch-ROCm bundles a HIP runtime of its own (torch/lib/libamdhip64.so) next to the system one this library links: two
        # runtimes in one process.  Measured on the GPU box: torch's cannot find a device any more once the system runtime has been
        # initialised first ("No HIP GPUs are available"), the other order works.  So when torch is already imported and has not
        # touched the GPU yet, let it do so before msfm_create (bench.py / sharding.py set their device before creating a context).
        import sys
        torch = sys.modules.get("torch")
        if torch is not None:
            try:
                if torch.cuda.is_available() and not torch.cuda.is_initialized():
                    torch.cuda.init()
            except Exception:   # noqa: BLE001  (no GPU for torch: msfm_create below reports the real state)
                pass
        self._L = load()
        h = C.c_void_p()
        rc = self._L.msfm_create(int(device), C.byref(h))
        if rc != OK:
            raise MsfmError(rc, "msfm_create(device=%d) failed: no usable gfx950 GPU (there is no CPU fallback)" % device)
        self._h = h
        self.device = int(device)
        if order != ORDER_SSE4X4:
            self.set_accum_order(order)

    def close(self):
        if getattr(self, "_h", None):
            self._L.msfm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc):
        if rc != OK:
            raise MsfmError(rc, self._L.msfm_last_error(self._h).decode())

    def device_info(self):
        name = C.create_string_buffer(256)
        cu, mhz = C.c_int(), C.c_int()
        self._chk(self._L.msfm_device_info(self._h, name, 256, C.byref(cu), C.byref(mhz)))
        return {"name": name.value.decode(), "cu_count": cu.value, "clock_mhz": mhz.value}

    def set_accum_order(self, order):
        self._chk(self._L.msfm_set_accum_order(self._h, int(order)))

    def set_prefilter(self, enable):
        """True / 1 (default): MFMA prefilter + exact re-check (byte images on the integer matrix cores); 2: fp16 matrix
        cores only; False / 0: brute-force exact kernel only."""
        self._chk(self._L.msfm_set_prefilter(self._h, 2 if enable == 2 and enable is not True else int(bool(enable))))

    def set_limits(self, max_pairs_per_batch=0, scratch_bytes=0):
        """Sub-batch limits of match_pairs (<= 0: default).  Results do not depend on them."""
        self._chk(self._L.msfm_set_limits(self._h, int(max_pairs_per_batch), int(scratch_bytes)))

    def set_pipeline(self, min_sub_batches=0):
        """A large call is cut into at least this many (shrinking) sub-batches whose tails overlap the next ones' sweeps (<= 0:
        default 2; 1: off).  Results do not depend on it."""
        self._chk(self._L.msfm_set_pipeline(self._h, int(min_sub_batches)))

    def profile(self):
        p = Profile()
        self._chk(self._L.msfm_get_profile(self._h, C.byref(p)))
        return {k: getattr(p, k) for k, _ in Profile._fields_}

    def upload_image(self, image_id, desc):
        desc = np.asarray(desc)
        if desc.dtype == np.uint8:
            dtype = DTYPE_U8
        else:
            desc = desc.astype(np.float32, copy=False)
            dtype = DTYPE_F32
        desc = np.ascontiguousarray(desc)
        if desc.ndim != 2:
            raise ValueError("descriptors must be 2-D")
        n, dim = desc.shape
        if n == 0:
            dim = DIM
        self._chk(self._L.msfm_upload_image(self._h, int(image_id), desc.ctypes.data_as(C.c_void_p), n, dim, dtype))

    def subset_image(self, src_id, dst_id, rows):
        """dst = the given rows of the resident image src (device-side gather; no host descriptors needed)."""
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        self._chk(self._L.msfm_subset_image(self._h, int(src_id), int(dst_id), _ip(rows), len(rows)))

    def image_rows(self, image_id):
        n = C.c_int()
        self._chk(self._L.msfm_image_rows(self._h, int(image_id), C.byref(n)))
        return n.value

    def clear_images(self):
        self._chk(self._L.msfm_clear_images(self._h))

    def finalize_store(self):
        """Build everything uploaded so far now (otherwise the first matching call does it)."""
        if hasattr(self._L, "msfm_finalize_store"):
            self._chk(self._L.msfm_finalize_store(self._h))

    def store_info(self):
        """-> {device_bytes, rows, pending_images} of the descriptor store."""
        b, r, p = C.c_int64(), C.c_int64(), C.c_int64()
        self._chk(self._L.msfm_store_info(self._h, C.byref(b), C.byref(r), C.byref(p)))
        return {"device_bytes": b.value, "rows": r.value, "pending_images": p.value}

    def match_pair(self, id1, id2, ratio=0.8, cross_check=True, max_distance=0.7):
        n1 = max(self.image_rows(id1), 1)
        qt = np.empty((n1, 2), np.int32)
        d = np.empty(n1, np.float32)
        cnt = C.c_int()
        self._chk(self._L.msfm_match_pair(self._h, int(id1), int(id2), C.c_float(ratio), int(bool(cross_check)),
                                          float(max_distance), _ip(qt), _fp(d), C.byref(cnt)))
        m = cnt.value
        return qt[:m, 0].copy(), qt[:m, 1].copy(), d[:m].copy()

    def _view(self):
        """(qt[M,2], dist[M]) as NumPy views of the context's result buffers: no copy, valid until the next
        matching call on this context."""
        qp, dp, n = C.POINTER(C.c_int32)(), C.POINTER(C.c_float)(), C.c_int64()
        self._chk(self._L.msfm_view_matches(self._h, C.byref(qp), C.byref(dp), C.byref(n)))
        M = n.value
        if M == 0:
            return np.zeros((0, 2), np.int32), np.zeros(0, np.float32)
        return np.ctypeslib.as_array(qp, shape=(M, 2)), np.ctypeslib.as_array(dp, shape=(M,))

    def match_pairs(self, pairs, ratio=0.8, cross_check=True, max_distance=0.7, fetch=True):
        """pairs: P x 2 int array -> (offsets[P+1], qt[M,2], dist[M]); fetch=False skips the copy-out,
        fetch="view" returns views of the context's buffers (valid until the next matching call)."""
        pairs = np.ascontiguousarray(np.asarray(pairs, dtype=np.int32).reshape(-1, 2))
        P = pairs.shape[0]
        offs = np.zeros(P + 1, np.int64)
        prm = MatchParams(ratio, int(bool(cross_check)), max_distance)
        self._chk(self._L.msfm_match_pairs(self._h, _ip(pairs), P, C.byref(prm),
                                           offs.ctypes.data_as(C.POINTER(C.c_int64))))
        if not fetch:
            return offs, None, None
        if fetch == "view":
            return (offs,) + self._view()
        M = int(offs[-1])
        qt = np.empty((max(M, 1), 2), np.int32)
        d = np.empty(max(M, 1), np.float32)
        self._chk(self._L.msfm_fetch_matches(self._h, _ip(qt), _fp(d)))
        return offs, qt[:M], d[:M]

    def match_pairs_stream(self, pairs, ratio=0.8, cross_check=True, max_distance=0.7, verified=False, copy=True):
        """Generator over the device sub-batches of the job, in pair order (msfm_match_pairs_begin / _next): nothing accumulates in the
        library.  Yields dicts {first, n_pairs, offsets[n+1] (relative to the chunk), qt[m, 2], dist[m], sensitive[n], d_qt, d_dist
        (device pointers)}; with copy=False qt / dist / offsets are views, valid until the next item is requested."""
        pairs = np.ascontiguousarray(np.asarray(pairs, dtype=np.int32).reshape(-1, 2))
        prm = MatchParams(ratio, int(bool(cross_check)), max_distance)
        self._chk(self._L.msfm_match_pairs_begin(self._h, _ip(pairs), pairs.shape[0], C.byref(prm), int(bool(verified)), None))
        ch = Chunk()
        done = False
        try:
            while True:
                self._chk(self._L.msfm_match_pairs_next(self._h, C.byref(ch)))
                n = ch.n_pairs
                if n == 0:
                    done = True
                    return
                m = int(ch.count)
                offs = np.ctypeslib.as_array(ch.offsets, shape=(n + 1,))
                qt = np.ctypeslib.as_array(ch.qt, shape=(m, 2)) if m else np.zeros((0, 2), np.int32)
                d = np.ctypeslib.as_array(ch.dist, shape=(m,)) if m else np.zeros(0, np.float32)
                sens = np.ctypeslib.as_array(ch.sensitive_rows, shape=(n,))
                if copy:
                    offs, qt, d, sens = offs.copy(), qt.copy(), d.copy(), sens.copy()
                yield {"first": ch.first_pair, "n_pairs": n, "offsets": offs, "qt": qt, "dist": d, "sensitive": sens,
                       "d_qt": ch.d_qt, "d_dist": ch.d_dist}
        finally:
            # a consumer that breaks out of the loop (GeneratorExit) or whose body raises must not leave the series open: the store
            # would stay locked for uploads until the next matching call
            if not done and getattr(self, "_h", None):
                self._L.msfm_match_pairs_end(self._h)

    def memory_info(self):
        """Bytes the context holds (store, inbox, scratch, result lists, page-locked host) + the device's free / total memory."""
        m = Memory()
        self._chk(self._L.msfm_memory_info(self._h, C.byref(m)))
        return {k: getattr(m, k) for k, _ in Memory._fields_}

    def read_device(self, dev_ptr, shape, dtype):
        """A device buffer (raw pointer) as a new NumPy array, copied through the library's own runtime."""
        out = np.empty(shape, dtype)
        self._chk(self._L.msfm_read_device(self._h, out.ctypes.data_as(C.c_void_p), C.c_void_p(dev_ptr), out.nbytes))
        return out

    def fetch_matches_device(self, qt_ptr, dist_ptr=None):
        """Copy the last call's lists into caller-owned DEVICE memory on this context's GPU (raw pointers, e.g.
        torch.Tensor.data_ptr() of an int32 [M, 2] / float32 [M] tensor; None skips one)."""
        self._chk(self._L.msfm_fetch_matches_device(self._h, C.c_void_p(qt_ptr or 0), C.c_void_p(dist_ptr or 0)))

    def order_certificate(self, n_pairs):
        """Per pair of the last match_pairs call: rows / columns whose decisions are within the fp32 reassociation bound
        of flipping under another accumulation order (0 everywhere: the index lists are order-invariant)."""
        out = np.zeros(max(int(n_pairs), 1), np.int32)
        self._chk(self._L.msfm_fetch_order_certificate(self._h, _ip(out)))
        return out[:int(n_pairs)]

    def upload_keypoints(self, image_id, kpts):
        """kpts: n x k float32 (k >= 2), x and y in the first two columns."""
        kpts = np.ascontiguousarray(kpts, dtype=np.float32)
        if kpts.ndim != 2 or kpts.shape[1] < 2:
            raise ValueError("keypoints must be n x (>=2)")
        self._chk(self._L.msfm_upload_keypoints(self._h, int(image_id), _fp(kpts), kpts.shape[0], kpts.shape[1]))

    def match_pairs_verified(self, pairs, ratio=0.8, cross_check=True, max_distance=0.7, threshold=3.0,
                             confidence=0.99, max_iters=1000, seed=0x5eed5eed, fetch=True):
        """match_pairs + FeatureUtils::FilterMatches (F-matrix RANSAC) on the device."""
        pairs = np.ascontiguousarray(np.asarray(pairs, dtype=np.int32).reshape(-1, 2))
        P = pairs.shape[0]
        offs = np.zeros(P + 1, np.int64)
        prm = MatchParams(ratio, int(bool(cross_check)), max_distance)
        vprm = VerifyParams(threshold, confidence, int(max_iters), int(seed))
        self._chk(self._L.msfm_match_pairs_verified(self._h, _ip(pairs), P, C.byref(prm), C.byref(vprm),
                                                    offs.ctypes.data_as(C.POINTER(C.c_int64))))
        if not fetch:
            return offs, None, None
        if fetch == "view":
            return (offs,) + self._view()
        M = int(offs[-1])
        qt = np.empty((max(M, 1), 2), np.int32)
        d = np.empty(max(M, 1), np.float32)
        self._chk(self._L.msfm_fetch_matches(self._h, _ip(qt), _fp(d)))
        return offs, qt[:M], d[:M]

    def knn2_pair(self, id1, id2):
        n1, n2 = self.image_rows(id1), self.image_rows(id2)
        f_i = np.empty(max(n1, 1), np.int32)
        f_d0 = np.empty(max(n1, 1), np.float32)
        f_d1 = np.empty(max(n1, 1), np.float32)
        r_i = np.empty(max(n2, 1), np.int32)
        r_d0 = np.empty(max(n2, 1), np.float32)
        r_d1 = np.empty(max(n2, 1), np.float32)
        self._chk(self._L.msfm_knn2_pair(self._h, int(id1), int(id2), _ip(f_i), _fp(f_d0), _fp(f_d1),
                                         _ip(r_i), _fp(r_d0), _fp(r_d1)))
        return (f_i[:n1], f_d0[:n1], f_d1[:n1]), (r_i[:n2], r_d0[:n2], r_d1[:n2])


def device_count():
    """gfx950 devices msfm_create can open (0 without a GPU)."""
    return int(load().msfm_device_count())


def topscale_select(kpts, k):
    L = load()
    kpts = np.ascontiguousarray(kpts, dtype=np.float32).reshape(-1, 4)
    n = kpts.shape[0]
    out = np.empty(max(n, 1), np.int32)
    cnt = C.c_int()
    rc = L.msfm_topscale_select(_fp(kpts), n, int(k), _ip(out), C.byref(cnt))
    if rc != OK:
        raise MsfmError(rc, "msfm_topscale_select")
    return out[:cnt.value].copy()


def pair_id(id1, id2):
    L = load()
    out = C.c_int32()
    rc = L.msfm_pair_id(int(id1), int(id2), C.byref(out))
    if rc != OK:
        raise MsfmError(rc, "msfm_pair_id: ids must be in [0, %d)" % MAX_IMAGES)
    return out.value


def pair_from_id(pid):
    L = load()
    a, b = C.c_int(), C.c_int()
    rc = L.msfm_pair_from_id(int(pid), C.byref(a), C.byref(b))
    if rc != OK:
        raise MsfmError(rc, "msfm_pair_from_id")
    return a.value, b.value


def swap_image_pair(id1, id2):
    return bool(load().msfm_swap_image_pair(int(id1), int(id2)))
