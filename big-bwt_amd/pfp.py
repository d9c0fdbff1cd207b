"""Host-side mirror of the reference's stage interface over the C ABI of libpfpgpu.so.

The reference (alshai/Big-BWT) is three executables glued by files; this module exposes the
same three stages plus the whole chain as functions over numpy arrays that hold the reference's
file formats byte for byte:

    parse()     == newscanNT.x / pscan.x      (newscan.cpp:569-650)
    bwtparse()  == bwtparse                   (bwtparse.c:212-322)
    merge()     == pfbwtNT.x / pfbwt.x        (pfbwt.cpp:320-418)
    bigbwt()    == bigbwt -w W -p M [-S|-s|-e] (bigbwt:69-156)
    sacak_int() / sacak() / gsacak()  == gsa/gsacak.h:78-105

Everything computes in hand-written HIP kernels on the MI355X; there is no CPU fallback: the
import succeeds without a GPU (so that symbol checks can run), but creating a Context does not.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libpfpgpu.so")

FLAG_SA, FLAG_SSA, FLAG_ESA = 1, 2, 4

ERRORS = {0: "PFP_OK", -1: "PFP_EINVAL", -2: "PFP_ENODEV", -3: "PFP_EHIP", -4: "PFP_ECOLLISION",
          -5: "PFP_ELIMIT", -6: "PFP_EFORMAT", -7: "PFP_ENOMEM", -8: "PFP_ESHORT"}

# every symbol include/pfpgpu.h declares
SYMBOLS = ["pfp_device_count", "pfp_ctx_create", "pfp_ctx_destroy", "pfp_last_error", "pfp_strerror", "pfp_version", "pfp_ctx_stream",
           "pfp_free", "pfp_debug_check", "pfp_get_mem_stats", "pfp_get_pool_counters", "pfp_pool_trim", "pfp_scan", "pfp_parse", "pfp_parse_result_free", "pfp_sacak_int", "pfp_sacak", "pfp_gsacak", "pfp_sacak_int64", "pfp_sacak64", "pfp_gsacak64", "pfp_gsacak_lcp_da", "pfp_gsacak_lcp_da64",
           "pfp_bwtparse", "pfp_merge", "pfp_bwt_result_free", "pfp_bigbwt", "pfp_bigbwt_files", "pfp_bigbwt_dev", "pfp_bigbwt_formats_dev", "pfp_dev_free", "pfp_memcpy_d2h", "pfp_pack5_dev", "pfp_sample_runs_dev", "pfp_pwrite_dev", "pfp_get_stats",
           "pfp_set_profiling", "pfp_set_kernel_trace", "pfp_get_kernel_trace", "pfp_set_max_phrase", "pfp_set_window_hash", "pfp_set_parse_density", "pfp_debug_msd_sort", "pfp_dist_parse_plan", "pfp_dist_propose_triggers2", "pfp_dist_decide_density", "pfp_dist_local_parse2", "pfp_bigbwt_fd", "pfp_multi_rccl_selftest", "pfp_multi_rccl_selftest2", "pfp_set_index_bits", "pfp_stage_text_dev", "pfp_scan_staged", "pfp_scan_k1_enqueue",
           "pfp_dist_propose_triggers", "pfp_dist_local_parse", "pfp_dist_export_local", "pfp_dist_global", "pfp_dist_global_sort", "pfp_dist_global_finish", "pfp_dist_partition_words", "pfp_dist_export_partition",
           "pfp_dist_owner_dedup", "pfp_dist_export_owned", "pfp_dist_global_sort_distinct", "pfp_dist_merge", "pfp_dist_sample_runs", "pfp_dist_release", "pfp_bigbwt_files_multi", "pfp_dist_parse_sort", "pfp_dist_set_parse_sa"]


class PfpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{ERRORS.get(code, code)}: {msg}")
        self.code = code


class _ParseResult(C.Structure):
    _fields_ = [("n_used", C.c_uint64), ("dict", C.POINTER(C.c_uint8)), ("dict_size", C.c_uint64),
                ("occ", C.POINTER(C.c_uint32)), ("n_words", C.c_uint64),
                ("parse", C.POINTER(C.c_uint32)), ("n_phrases", C.c_uint64),
                ("last", C.POINTER(C.c_uint8)), ("sai", C.POINTER(C.c_uint8))]


class _BwtResult(C.Structure):
    _fields_ = [("bwt", C.POINTER(C.c_uint8)), ("bwt_size", C.c_uint64),
                ("sa", C.POINTER(C.c_uint8)), ("sa_bytes", C.c_uint64),
                ("ssa", C.POINTER(C.c_uint8)), ("ssa_bytes", C.c_uint64),
                ("esa", C.POINTER(C.c_uint8)), ("esa_bytes", C.c_uint64)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("launches", C.c_uint64), ("total_ms", C.c_double), ("algo_bytes", C.c_uint64)]


class Stats(C.Structure):
    _fields_ = [("n", C.c_uint64), ("n_phrases", C.c_uint64), ("n_words", C.c_uint64), ("dict_size", C.c_uint64),
                ("sa_rounds_dict", C.c_uint64), ("sa_rounds_parse", C.c_uint64),
                ("hard_groups", C.c_uint64), ("hard_chars", C.c_uint64),
                ("hard_big_groups", C.c_uint64), ("hard_max_chars", C.c_uint64), ("hard_max_members", C.c_uint64),
                ("hash_reseeds", C.c_uint64),
                ("extra_triggers", C.c_uint64), ("index_bits", C.c_uint64),
                ("hard_minor_groups", C.c_uint64), ("hard_minor_chars", C.c_uint64),
                ("ms_scan", C.c_double), ("ms_phrases", C.c_double), ("ms_sa_dict", C.c_double),
                ("ms_sa_parse", C.c_double), ("ms_merge", C.c_double), ("ms_total", C.c_double), ("parse_density", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_lib = None


def load_library():
    """dlopen libpfpgpu.so (built in-tree by __graft_entry__.build / csrc/Makefile)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: the HIP extension has not been built "
                              f"(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
        # PyTorch wheels bundle their own HIP runtime (torch/lib/libamdhip64.so, soname
        # libamdhip64.so.7).  Two HIP runtimes in one process cannot both own the GPU, so when
        # torch is installed it is imported FIRST: libpfpgpu.so's DT_NEEDED libamdhip64.so.7 then
        # binds to the runtime torch already loaded and both share devices, streams and memory.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        lib = C.CDLL(LIB_PATH)
        lib.pfp_last_error.restype = C.c_char_p
        lib.pfp_strerror.restype = C.c_char_p
        lib.pfp_version.restype = C.c_char_p
        lib.pfp_ctx_stream.restype = C.c_void_p
        lib.pfp_ctx_destroy.restype = None
        lib.pfp_free.restype = None
        lib.pfp_set_profiling.restype = None
        lib.pfp_set_max_phrase.restype = None
        lib.pfp_set_window_hash.restype = None
        lib.pfp_set_kernel_trace.restype = None
        lib.pfp_dist_release.restype = None
        lib.pfp_dev_free.restype = None
        lib.pfp_pool_trim.restype = None
        lib.pfp_parse_result_free.restype = None
        lib.pfp_bwt_result_free.restype = None
        _lib = lib
    return _lib


def _arr(a, dtype):
    if isinstance(a, (bytes, bytearray, memoryview)):
        a = np.frombuffer(a, dtype=np.uint8)
    a = np.ascontiguousarray(a)
    if a.dtype != dtype:
        a = a.astype(dtype) if dtype != np.uint8 or a.dtype.itemsize == 1 else a.view(np.uint8)
    return a


def _ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


def _take(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(int(n),)).astype(dtype, copy=True)


def _adopt(lib, ptr, n):
    """a malloc'ed result buffer of the library as a numpy array WITHOUT copying it (a copy of a 0.8 GB .bwt costs more
    than its trip across PCIe); the buffer is handed to pfp_free when the array is collected"""
    if n == 0 or not ptr:
        return np.zeros(0, dtype=np.uint8)
    import weakref
    addr = C.cast(ptr, C.c_void_p).value
    arr = np.ctypeslib.as_array(ptr, shape=(int(n),))
    weakref.finalize(arr, lib.pfp_free, C.c_void_p(addr))
    return arr


def unpack5(b):
    """5-byte little-endian ints (utils.c:112-129) -> u64 array"""
    b = np.frombuffer(bytes(b), dtype=np.uint8).reshape(-1, 5)
    out = np.zeros((len(b), 8), dtype=np.uint8)
    out[:, :5] = b
    return out.view(np.uint64).reshape(-1)


def pack5(v):
    v = np.ascontiguousarray(v, dtype=np.uint64).reshape(-1)
    return v.view(np.uint8).reshape(-1, 8)[:, :5].copy().reshape(-1)


class Context:
    """One HIP stream + memory pool on one GPU (pfp_ctx)."""

    def __init__(self, device=0):
        self.lib = load_library()
        self._h = C.c_void_p()
        rc = self.lib.pfp_ctx_create(C.byref(self._h), C.c_int(device))
        if rc:
            raise PfpError(rc, "pfp_ctx_create failed: " + self.lib.pfp_strerror(rc).decode() +
                           " (the HIP path is mandatory; no CPU fallback exists)")
        self.device = device

    def close(self):
        if self._h:
            self.lib.pfp_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise PfpError(rc, self.lib.pfp_last_error(self._h).decode(errors="replace"))

    @property
    def stream(self):
        return self.lib.pfp_ctx_stream(self._h)

    def debug_check(self):
        """PFP_POOL_DEBUG=1: raise if a canary band of any released device block was damaged"""
        self._check(self.lib.pfp_debug_check(self._h))

    def pool_trim(self):
        self.lib.pfp_pool_trim(self._h)

    def mem_stats(self):
        out = (C.c_uint64 * 4)()
        self._check(self.lib.pfp_get_mem_stats(self._h, out))
        return dict(held=int(out[0]), peak=int(out[1]), live=int(out[2]), debug_blocks=int(out[3]))

    def pool_counters(self):
        out = (C.c_uint64 * 2)()
        self._check(self.lib.pfp_get_pool_counters(self._h, out))
        return dict(driver_allocs=int(out[0]), trims=int(out[1]))

    def set_profiling(self, on=True):
        self.lib.pfp_set_profiling(self._h, C.c_int(1 if on else 0))

    def set_kernel_trace(self, on=True):
        """start (and clear) / stop per-kernel HIP-event timing on the ctx stream"""
        self.lib.pfp_set_kernel_trace(self._h, C.c_int(1 if on else 0))

    def kernel_trace(self):
        """rows {name, launches, total_ms, algo_bytes} accumulated since set_kernel_trace(True)"""
        n = self.lib.pfp_get_kernel_trace(self._h, None, C.c_int(0))
        if n <= 0:
            return []
        arr = (KernelStat * n)()
        self.lib.pfp_get_kernel_trace(self._h, arr, C.c_int(n))
        return [dict(name=r.name.decode(), launches=int(r.launches), total_ms=float(r.total_ms), algo_bytes=int(r.algo_bytes))
                for r in arr]

    def set_max_phrase(self, max_phrase):
        """fused chain: split phrases longer than this with extra trigger windows (0 = reference parse)"""
        self.lib.pfp_set_max_phrase(self._h, C.c_uint64(max_phrase))

    def set_window_hash(self, fast):
        """fused chain: cut the text by the cheap window hash (True, default) or by the reference's Karp-Rabin hash (False)"""
        self.lib.pfp_set_window_hash(self._h, C.c_int(1 if fast else 0))

    def debug_msd_sort(self, keys, vals, lo, hi):
        """radix.hip's sort on numpy arrays (uint64 keys, optional uint32 values), in place; stable on key bits [lo, hi)"""
        assert keys.dtype == np.uint64 and keys.flags.c_contiguous and (vals is None or (vals.dtype == np.uint32 and vals.flags.c_contiguous))
        self._check(self.lib.pfp_debug_msd_sort(self._h, keys.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                vals.ctypes.data_as(C.POINTER(C.c_uint32)) if vals is not None else None,
                                                C.c_uint64(len(keys)), C.c_int(lo), C.c_int(hi)))

    def set_parse_density(self, density):
        """fused chain, opt-in: the window hash cuts with probability density / p (outputs unchanged, see pfpgpu.h)"""
        self._check(self.lib.pfp_set_parse_density(self._h, C.c_double(density)))

    def set_index_bits(self, bits):
        """0: index width by size (32 bits below 4 GiB), 64: always the wide build (bigbwt:109-151)"""
        self._check(self.lib.pfp_set_index_bits(self._h, C.c_int(bits)))

    def stats(self):
        st = Stats()
        self._check(self.lib.pfp_get_stats(self._h, C.byref(st)))
        return st.as_dict()

    # -- stage 1a: newscan.cpp:168-202, 363-377
    def scan(self, text, w=10, p=100):
        t = _arr(text, np.uint8)
        ends = C.POINTER(C.c_uint64)()
        ne, used = C.c_uint64(), C.c_uint64()
        self._check(self.lib.pfp_scan(self._h, _ptr(t, C.c_uint8), C.c_uint64(len(t)), C.c_int(w), C.c_uint64(p),
                                      C.byref(ends), C.byref(ne), C.byref(used)))
        out = _take(ends, ne.value, np.uint64)
        self.lib.pfp_free(ends)
        return out, used.value

    # -- stage 1: newscan.cpp main
    def parse(self, text, w=10, p=100, want_sai=False):
        t = _arr(text, np.uint8)
        r = _ParseResult()
        self._check(self.lib.pfp_parse(self._h, _ptr(t, C.c_uint8), C.c_uint64(len(t)), C.c_int(w), C.c_uint64(p),
                                       C.c_int(1 if want_sai else 0), C.byref(r)))
        out = dict(n_used=int(r.n_used), dict=_take(r.dict, r.dict_size, np.uint8), occ=_take(r.occ, r.n_words, np.uint32),
                   parse=_take(r.parse, r.n_phrases, np.uint32), last=_take(r.last, r.n_phrases, np.uint8),
                   sai=_take(r.sai, 5 * r.n_phrases, np.uint8) if want_sai else None)
        self.lib.pfp_parse_result_free(C.byref(r))
        return out

    # -- gsa/gsacak.h
    def sacak_int(self, s, k=0):
        s = _arr(s, np.uint32)
        sa = np.zeros(len(s), dtype=np.uint32)
        self._check(self.lib.pfp_sacak_int(self._h, _ptr(s, C.c_uint32), _ptr(sa, C.c_uint32), C.c_uint64(len(s)), C.c_uint64(k)))
        return sa

    def sacak(self, s):
        s = _arr(s, np.uint8)
        sa = np.zeros(len(s), dtype=np.uint32)
        self._check(self.lib.pfp_sacak(self._h, _ptr(s, C.c_uint8), _ptr(sa, C.c_uint32), C.c_uint64(len(s))))
        return sa

    def gsacak(self, s):
        s = _arr(s, np.uint8)
        sa = np.zeros(len(s), dtype=np.uint32)
        self._check(self.lib.pfp_gsacak(self._h, _ptr(s, C.c_uint8), _ptr(sa, C.c_uint32), C.c_uint64(len(s))))
        return sa

    # -- the -DM64 build of gsa/gsacak.h: 64-bit SA entries
    def sacak_int64(self, s, k=0):
        s = _arr(s, np.uint32)
        sa = np.zeros(len(s), dtype=np.uint64)
        self._check(self.lib.pfp_sacak_int64(self._h, _ptr(s, C.c_uint32), _ptr(sa, C.c_uint64), C.c_uint64(len(s)), C.c_uint64(k)))
        return sa

    def sacak64(self, s):
        s = _arr(s, np.uint8)
        sa = np.zeros(len(s), dtype=np.uint64)
        self._check(self.lib.pfp_sacak64(self._h, _ptr(s, C.c_uint8), _ptr(sa, C.c_uint64), C.c_uint64(len(s))))
        return sa

    def gsacak64(self, s):
        s = _arr(s, np.uint8)
        sa = np.zeros(len(s), dtype=np.uint64)
        self._check(self.lib.pfp_gsacak64(self._h, _ptr(s, C.c_uint8), _ptr(sa, C.c_uint64), C.c_uint64(len(s))))
        return sa

    def gsacak_lcp_da(self, s, wide=False):
        """gsacak(s, SA, LCP, DA, n) with all three outputs (gsa/gsacak.h:96-105)"""
        s = _arr(s, np.uint8)
        it, ut, fn = (np.int64, np.uint64, self.lib.pfp_gsacak_lcp_da64) if wide else (np.int32, np.uint32, self.lib.pfp_gsacak_lcp_da)
        sa = np.zeros(len(s), dtype=ut); lcp = np.zeros(len(s), dtype=it); da = np.zeros(len(s), dtype=it)
        self._check(fn(self._h, _ptr(s, C.c_uint8), sa.ctypes.data_as(C.c_void_p), lcp.ctypes.data_as(C.c_void_p),
                       da.ctypes.data_as(C.c_void_p), C.c_uint64(len(s))))
        return sa, lcp, da

    # -- stage 2: bwtparse.c main
    def bwtparse(self, parse, last, occ, sai=None):
        parse = _arr(parse, np.uint32); last = _arr(last, np.uint8); occ = _arr(occ, np.uint32)
        P = len(parse)
        ilist = np.zeros(P + 1, dtype=np.uint32)
        bwlast = np.zeros(P + 1, dtype=np.uint8)
        bwsai = np.zeros(5 * (P + 1), dtype=np.uint8) if sai is not None else None
        sai_a = _arr(sai, np.uint8) if sai is not None else None
        self._check(self.lib.pfp_bwtparse(self._h, _ptr(parse, C.c_uint32), C.c_uint64(P), _ptr(last, C.c_uint8),
                                          _ptr(sai_a, C.c_uint8) if sai is not None else None,
                                          _ptr(occ, C.c_uint32), C.c_uint64(len(occ)), _ptr(ilist, C.c_uint32),
                                          _ptr(bwlast, C.c_uint8),
                                          _ptr(bwsai, C.c_uint8) if sai is not None else None))
        return ilist, bwlast, bwsai

    def _bwt_result(self, r):
        out = dict(bwt=_adopt(self.lib, r.bwt, r.bwt_size), sa=_adopt(self.lib, r.sa, r.sa_bytes),
                   ssa=_adopt(self.lib, r.ssa, r.ssa_bytes), esa=_adopt(self.lib, r.esa, r.esa_bytes))
        return out

    # -- stage 3: pfbwt.cpp main
    def merge(self, dict_, occ, ilist, bwlast, bwsai=None, w=10, flags=0):
        d = _arr(dict_, np.uint8); occ = _arr(occ, np.uint32); ilist = _arr(ilist, np.uint32); bwlast = _arr(bwlast, np.uint8)
        bs = _arr(bwsai, np.uint8) if bwsai is not None else None
        r = _BwtResult()
        self._check(self.lib.pfp_merge(self._h, _ptr(d, C.c_uint8), C.c_uint64(len(d)), _ptr(occ, C.c_uint32),
                                       C.c_uint64(len(occ)), _ptr(ilist, C.c_uint32), _ptr(bwlast, C.c_uint8),
                                       _ptr(bs, C.c_uint8) if bs is not None else None, C.c_uint64(len(ilist)),
                                       C.c_int(w), C.c_int(flags), C.byref(r)))
        return self._bwt_result(r)

    # -- bigbwt chain, host buffers
    def bigbwt(self, text, w=10, p=100, flags=0):
        t = _arr(text, np.uint8)
        r = _BwtResult()
        self._check(self.lib.pfp_bigbwt(self._h, _ptr(t, C.c_uint8), C.c_uint64(len(t)), C.c_int(w), C.c_uint64(p),
                                        C.c_int(flags), C.byref(r)))
        return self._bwt_result(r)

    def bigbwt_files(self, text, base, w=10, p=100, flags=0):
        """host text in, <base>.bwt/.sa/.ssa/.esa out (streamed from HBM); returns the sizes written"""
        t = _arr(text, np.uint8)
        sizes = (C.c_uint64 * 4)()
        self._check(self.lib.pfp_bigbwt_files(self._h, _ptr(t, C.c_uint8), C.c_uint64(len(t)), C.c_int(w), C.c_uint64(p),
                                              C.c_int(flags), C.c_char_p(os.fsencode(base)), sizes))
        return dict(bwt=int(sizes[0]), sa=int(sizes[1]), ssa=int(sizes[2]), esa=int(sizes[3]))

    # -- bigbwt chain, device-resident (raw device pointers, e.g. torch tensor.data_ptr())
    def bigbwt_dev(self, d_text_ptr, n, d_bwt_ptr, d_sa_ptr=None, w=10, p=100, flags=0):
        used = C.c_uint64()
        self._check(self.lib.pfp_bigbwt_dev(self._h, C.c_void_p(d_text_ptr), C.c_uint64(n), C.c_int(w), C.c_uint64(p),
                                            C.c_int(flags), C.c_void_p(d_bwt_ptr),
                                            C.c_void_p(d_sa_ptr) if d_sa_ptr else None, C.byref(used)))
        return used.value

    def bigbwt_formats_dev(self, d_text_ptr, n, d_bwt_ptr, w=10, p=100, flags=0):
        """device text in, .bwt into d_bwt, .sa/.ssa/.esa as library-owned device buffers: -> (n_used, {name: (ptr, bytes)});
        release every ptr with dev_free"""
        used = C.c_uint64()
        outs = (C.c_void_p * 3)()
        sizes = (C.c_uint64 * 3)()
        self._check(self.lib.pfp_bigbwt_formats_dev(self._h, C.c_void_p(d_text_ptr), C.c_uint64(n), C.c_int(w), C.c_uint64(p), C.c_int(flags),
                                                    C.c_void_p(d_bwt_ptr), outs, sizes, C.byref(used)))
        return used.value, {k: (outs[i], int(sizes[i])) for i, k in enumerate(("sa", "ssa", "esa")) if outs[i]}

    def fetch_dev(self, d_ptr, nbytes):
        """device bytes (a buffer the library handed out) -> numpy uint8 array"""
        out = np.empty(int(nbytes), dtype=np.uint8)
        self._check(self.lib.pfp_memcpy_d2h(self._h, out.ctypes.data_as(C.c_void_p), C.c_void_p(d_ptr), C.c_uint64(nbytes)))
        return out

    def dev_free(self, d_ptr):
        self.lib.pfp_dev_free(self._h, C.c_void_p(d_ptr))

    def pack5_dev(self, d_vals_ptr, count, d_out5_ptr):
        """count u64 device values -> 5-byte LE ints in device memory (utils.c:112-129)"""
        self._check(self.lib.pfp_pack5_dev(self._h, C.c_void_p(d_vals_ptr), C.c_uint64(count), C.c_void_p(d_out5_ptr)))

    def sample_runs_dev(self, d_bwt_ptr, d_sa_ptr, count, pos_base=0, left=-1, right=-1, run_end=False, d_out10_ptr=None, cap_pairs=0):
        """.ssa / .esa pairs of a BWT slice in device memory (pfbwt.cpp:605-676); returns the number of pairs"""
        k = C.c_uint64()
        rc = self.lib.pfp_sample_runs_dev(self._h, C.c_void_p(d_bwt_ptr), C.c_void_p(d_sa_ptr) if d_sa_ptr else None, C.c_uint64(count),
                                          C.c_uint64(pos_base), C.c_int(left), C.c_int(right), C.c_int(1 if run_end else 0),
                                          C.c_void_p(d_out10_ptr) if d_out10_ptr else None, C.c_uint64(cap_pairs), C.byref(k))
        self._check(rc)
        return k.value

    def pwrite_dev(self, path, file_offset, d_src_ptr, nbytes):
        """device bytes -> file at an offset (pfthreads.hpp:369-376 pattern), through pinned staging buffers"""
        self._check(self.lib.pfp_pwrite_dev(self._h, C.c_char_p(os.fsencode(path)), C.c_uint64(file_offset), C.c_void_p(d_src_ptr),
                                            C.c_uint64(nbytes)))

    # -- multi-GPU chain, one rank's share (device pointers; collectives are the caller's: dist.py)
    def dist_propose_triggers(self, d_text_ptr, n, w, p):
        hashes = (C.c_uint32 * 8)()
        cnt = C.c_uint32()
        self._check(self.lib.pfp_dist_propose_triggers(self._h, C.c_void_p(d_text_ptr), C.c_uint64(n), C.c_int(w),
                                                       C.c_uint64(p), hashes, C.byref(cnt)))
        return [int(hashes[i]) for i in range(cnt.value)]

    def dist_local_parse(self, d_text_ptr, n, halo_len, w, p, is_first, is_last, global_offset, want_sai, extra=()):
        sizes = (C.c_uint64 * 4)()
        ex = (C.c_uint32 * max(1, len(extra)))(*extra)
        self._check(self.lib.pfp_dist_local_parse(self._h, C.c_void_p(d_text_ptr), C.c_uint64(n), C.c_uint64(halo_len),
                                                  C.c_int(w), C.c_uint64(p), C.c_int(int(is_first)), C.c_int(int(is_last)),
                                                  C.c_uint64(global_offset), C.c_int(int(want_sai)), ex,
                                                  C.c_uint32(len(extra)), sizes))
        return dict(dict_bytes=int(sizes[0]), words=int(sizes[1]), phrases=int(sizes[2]), last_trigger=int(sizes[3]))

    def dist_parse_plan(self, first_bytes, w, p, ranks=1):
        """rank 0: (plan, first_hash) from the text's first bytes (pfp_dist_parse_plan); plan = four 64-bit integers"""
        fb = np.ascontiguousarray(np.frombuffer(bytes(first_bytes), dtype=np.uint8))
        plan = (C.c_uint64 * 4)()
        fh = C.c_uint64()
        self._check(self.lib.pfp_dist_parse_plan(self._h, fb.ctypes.data_as(C.POINTER(C.c_uint8)) if len(fb) else None, C.c_uint64(len(fb)),
                                                 C.c_int(w), C.c_uint64(p), C.c_uint32(ranks), plan, C.byref(fh)))
        return [int(x) for x in plan], int(fh.value)

    def dist_propose_triggers2(self, d_text_ptr, n, halo_len, w, p, plan, d_sample_ptr, sample_cap):
        hashes = (C.c_uint32 * 8)()
        cnt = C.c_uint32()
        ns = C.c_uint64()
        pl = (C.c_uint64 * 4)(*plan)
        self._check(self.lib.pfp_dist_propose_triggers2(self._h, C.c_void_p(d_text_ptr), C.c_uint64(n), C.c_uint64(halo_len), C.c_int(w),
                                                        C.c_uint64(p), pl, hashes, C.byref(cnt), C.c_void_p(d_sample_ptr) if sample_cap else None,
                                                        C.c_uint64(sample_cap), C.byref(ns)))
        return [int(hashes[i]) for i in range(cnt.value)], int(ns.value)

    def dist_decide_density(self, d_samples_ptr, count, p, plan):
        pl = (C.c_uint64 * 4)(*plan)
        self._check(self.lib.pfp_dist_decide_density(self._h, C.c_void_p(d_samples_ptr) if count else None, C.c_uint64(count), C.c_uint64(p), pl))
        return [int(x) for x in pl]

    def dist_local_parse2(self, d_text_ptr, n, halo_len, w, p, is_first, is_last, global_offset, want_sai, plan, extra=()):
        sizes = (C.c_uint64 * 4)()
        ex = (C.c_uint32 * max(1, len(extra)))(*extra)
        pl = (C.c_uint64 * 4)(*plan)
        self._check(self.lib.pfp_dist_local_parse2(self._h, C.c_void_p(d_text_ptr), C.c_uint64(n), C.c_uint64(halo_len),
                                                   C.c_int(w), C.c_uint64(p), C.c_int(int(is_first)), C.c_int(int(is_last)),
                                                   C.c_uint64(global_offset), C.c_int(int(want_sai)), pl, ex,
                                                   C.c_uint32(len(extra)), sizes))
        return dict(dict_bytes=int(sizes[0]), words=int(sizes[1]), phrases=int(sizes[2]), last_trigger=int(sizes[3]))

    def dist_export_local(self, d_dict=None, d_occ=None, d_last=None, d_sai=None):
        vp = lambda x: C.c_void_p(x) if x else None
        self._check(self.lib.pfp_dist_export_local(self._h, vp(d_dict), vp(d_occ), vp(d_last), vp(d_sai)))

    def dist_global(self, d_union, union_bytes, d_union_occ, n_union, my_word_base, d_sym_out):
        info = (C.c_uint64 * 3)()
        self._check(self.lib.pfp_dist_global(self._h, C.c_void_p(d_union), C.c_uint64(union_bytes), C.c_void_p(d_union_occ),
                                             C.c_uint64(n_union), C.c_uint64(my_word_base), C.c_void_p(d_sym_out), info))
        return dict(words=int(info[0]), dict_bytes=int(info[1]), rounds=int(info[2]))

    def dist_global_sort(self, d_union, union_bytes, d_union_occ, n_union, part, parts, d_wslot_out):
        info = (C.c_uint64 * 8)()
        self._check(self.lib.pfp_dist_global_sort(self._h, C.c_void_p(d_union), C.c_uint64(union_bytes), C.c_void_p(d_union_occ),
                                                  C.c_uint64(n_union), C.c_uint32(part), C.c_uint32(parts), C.c_void_p(d_wslot_out),
                                                  info))
        return dict(words=int(info[0]), dict_bytes=int(info[1]), rounds=int(info[2]), complete=bool(info[3]), slots=int(info[4]),
                    slot_base=int(info[5]), emits=int(info[6]), index_bits=int(info[7]))

    def dist_partition_words(self, parts):
        """-> [(words, bytes)] this rank sends to every owner (hash-partitioned dedup)"""
        cnt = (C.c_uint64 * (2 * parts))()
        self._check(self.lib.pfp_dist_partition_words(self._h, C.c_uint32(parts), cnt))
        return [(int(cnt[2 * o]), int(cnt[2 * o + 1])) for o in range(parts)]

    def dist_export_partition(self, d_bytes, d_occ):
        self._check(self.lib.pfp_dist_export_partition(self._h, C.c_void_p(d_bytes), C.c_void_p(d_occ)))

    def dist_owner_dedup(self, d_bytes, nbytes, d_occ, n_words, d_pid_out):
        out = (C.c_uint64 * 2)()
        vp = lambda x: C.c_void_p(x) if x else None
        self._check(self.lib.pfp_dist_owner_dedup(self._h, vp(d_bytes), C.c_uint64(nbytes), vp(d_occ), C.c_uint64(n_words), vp(d_pid_out), out))
        return int(out[0]), int(out[1])

    def dist_export_owned(self, d_bytes, d_occ):
        vp = lambda x: C.c_void_p(x) if x else None
        self._check(self.lib.pfp_dist_export_owned(self._h, vp(d_bytes), vp(d_occ)))

    def dist_global_sort_distinct(self, d_dict, dict_bytes, d_occ, n_words, d_gid_sent, part, parts, d_wslot_out):
        info = (C.c_uint64 * 8)()
        self._check(self.lib.pfp_dist_global_sort_distinct(self._h, C.c_void_p(d_dict), C.c_uint64(dict_bytes), C.c_void_p(d_occ),
                                                           C.c_uint64(n_words), C.c_void_p(d_gid_sent), C.c_uint32(part), C.c_uint32(parts),
                                                           C.c_void_p(d_wslot_out), info))
        return dict(words=int(info[0]), dict_bytes=int(info[1]), rounds=int(info[2]), complete=bool(info[3]), slots=int(info[4]),
                    slot_base=int(info[5]), emits=int(info[6]), index_bits=int(info[7]))

    def dist_global_finish(self, d_wslot_all, parts, my_word_base, d_sym_out):
        self._check(self.lib.pfp_dist_global_finish(self._h, C.c_void_p(d_wslot_all), C.c_uint32(parts), C.c_uint64(my_word_base),
                                                    C.c_void_p(d_sym_out)))

    def dist_parse_sort(self, d_sym, P, part, parts, d_sa_out):
        info = (C.c_uint64 * 4)()
        self._check(self.lib.pfp_dist_parse_sort(self._h, C.c_void_p(d_sym), C.c_uint64(P), C.c_uint32(part), C.c_uint32(parts),
                                                 C.c_void_p(d_sa_out), info))
        return dict(entries=int(info[0]), slot_base=int(info[1]), complete=bool(info[2]), rounds=int(info[3]))

    def dist_set_parse_sa(self, d_sa, count):
        self._check(self.lib.pfp_dist_set_parse_sa(self._h, C.c_void_p(d_sa) if count else None, C.c_uint64(count)))

    def dist_merge(self, d_sym, P, d_last, d_sai, flags, n_total, out_lo, out_hi, d_bwt_slice, d_sa_slice=None):
        self._check(self.lib.pfp_dist_merge(self._h, C.c_void_p(d_sym), C.c_uint64(P), C.c_void_p(d_last),
                                            C.c_void_p(d_sai) if d_sai else None, C.c_int(flags), C.c_uint64(n_total),
                                            C.c_uint64(out_lo), C.c_uint64(out_hi), C.c_void_p(d_bwt_slice),
                                            C.c_void_p(d_sa_slice) if d_sa_slice else None))

    def dist_sample_runs(self, run_end, drop_edge, d_out10_ptr=None, cap_pairs=0):
        """.ssa / .esa pairs of the slice the last dist_merge emitted (-s / -e without an SA slice), from its run maps;
        returns the number of pairs (count only without an output pointer)"""
        k = C.c_uint64()
        self._check(self.lib.pfp_dist_sample_runs(self._h, C.c_int(1 if run_end else 0), C.c_int(1 if drop_edge else 0),
                                                  C.c_void_p(d_out10_ptr) if d_out10_ptr else None, C.c_uint64(cap_pairs), C.byref(k)))
        return k.value

    def dist_release(self):
        self.lib.pfp_dist_release(self._h)

    def stage_text_dev(self, d_text_ptr, n, w=10):
        self._check(self.lib.pfp_stage_text_dev(self._h, C.c_void_p(d_text_ptr), C.c_uint64(n), C.c_int(w)))

    def scan_staged(self, p=100):
        ne = C.c_uint64()
        self._check(self.lib.pfp_scan_staged(self._h, C.c_uint64(p), C.byref(ne)))
        return ne.value

    def scan_k1_enqueue(self, p=100):
        self._check(self.lib.pfp_scan_k1_enqueue(self._h, C.c_uint64(p)))


def bigbwt_files_multi(text, base, devices, w=10, p=100, flags=0, halo=0):
    """pfp_bigbwt_files_multi: one BWT on len(devices) GPUs from this process (csrc/multi.hip); returns its statistics"""
    lib = load_library()

    class MultiStats(C.Structure):
        _fields_ = [(k, C.c_uint64) for k in ("n", "n_words", "n_phrases", "dict_size", "index_bits", "ranks", "sa_shares", "parse_shares")] + \
                   [("ms_chain", C.c_double), ("ms_total", C.c_double), ("parse_density", C.c_double)]
    text = _arr(text, np.uint8)
    st = MultiStats()
    err = C.create_string_buffer(1024)
    devs = (C.c_int * len(devices))(*devices)
    lib.pfp_bigbwt_files_multi.restype = C.c_int
    rc = lib.pfp_bigbwt_files_multi(C.c_int(len(devices)), devs, _ptr(text, C.c_uint8), C.c_uint64(len(text)), C.c_int(w), C.c_uint64(p),
                                    C.c_int(flags), C.c_uint64(halo), str(base).encode(), C.byref(st), err, C.c_uint64(len(err)))
    if rc != 0:
        raise PfpError(rc, err.value.decode(errors="replace"))
    return {k: getattr(st, k) for k, _ in MultiStats._fields_}


def multi_rccl_selftest(device=0, inject_failure=False):
    """pfp_multi_rccl_selftest2: the native multi-GPU host's RCCL transport over one device (raises PfpError with its message);
    inject_failure: also the error path - a failure inside an open group, the communicators aborted"""
    lib = load_library()
    err = C.create_string_buffer(1024)
    rc = lib.pfp_multi_rccl_selftest2(C.c_int(device), C.c_int(1 if inject_failure else 0), err, C.c_uint64(len(err)))
    if rc:
        raise PfpError(rc, err.value.decode(errors="replace"))
