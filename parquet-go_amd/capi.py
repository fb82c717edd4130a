"""ctypes binding of libpqdecode.so (the C ABI declared in include/pqdecode.h).

This is plumbing for tests and the benchmark: every decode goes through the
shared library's gfx950 kernels.  There is no Python or CPU decode path here; if
the library is missing or no HIP device is present the calls raise.
"""
import ctypes as C
import os
import weakref

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_DIR, "libpqdecode.so")

# ---- status codes (include/pqdecode.h) ----
OK, EOF, UNEXPECTED_EOF, EMPTY_BP_RUN, EMPTY_RLE_RUN, RLE_VALUE_TOO_LARGE = 0, 1, 2, 3, 4, 5
INT32_OUT_OF_RANGE, VARINT_OVERFLOW, INVALID_BLOCK_SIZE, INVALID_MINIBLOCK_COUNT = 6, 7, 8, 9
INVALID_MINIBLOCK_WIDTH, DICT_INVALID_WIDTH, DICT_INVALID_INDEX, NOT_INITIALIZED = 10, 11, 12, 13
INVALID_STREAM, PAGE_VALUES_FAILED = 14, 15
ERR_BAD_ARGUMENT, ERR_HIP, ERR_NO_DEVICE, ERR_OUT_OF_RANGE = 100, 101, 102, 103

KIND_HYBRID, KIND_DICT, KIND_LEVELS, KIND_DELTA32, KIND_DELTA64, KIND_PAGE_V1, KIND_PAGE_V2 = range(1, 8)
KIND_PLAIN, KIND_BOOL_PLAIN, KIND_BOOL_RLE, KIND_DELTA_LENGTH_BA, KIND_DELTA_BA = 8, 9, 10, 11, 12
BYTE_ARRAY_NEGATIVE_LEN, BYTE_ARRAY_COUNT_MISMATCH, BYTE_ARRAY_INVALID_PREFIX = 16, 17, 18
FLAG_LENGTH_PREFIXED, FLAG_LEVELS_CONST, FLAG_NO_INIT, FLAG_NO_LEVEL_OUTPUT = 1, 2, 4, 8
OPT_DELTA_WAVE_MIN_PAGES = 1
OPT_HYBRID_WAVE_MIN_PAGES = 2
OPT_SEGMENT_MAX_PAGES = 3
OPT_SEGMENT_BYTES = 4
OPT_CHUNK_SLOTS = 5
OPT_SEGMENT_SPIN_LIMIT = 6
OPT_CONCURRENT_GROUPS = 7
OPT_RUN_TABLES = 8

STAGE_NONE, STAGE_REP_INIT, STAGE_DEF_INIT, STAGE_VALUES_INIT, STAGE_REP, STAGE_DEF, STAGE_VALUES = range(7)

EXPORTS = [
    "pq_ctx_create", "pq_ctx_destroy", "pq_ctx_synchronize", "pq_ctx_set_option", "pq_status_string",
    "pq_last_error",
    "pq_version", "pq_bitunpack32_dev", "pq_bitunpack64_dev", "pq_bitunpack32", "pq_bitunpack64",
    "pq_plan_create", "pq_plan_run", "pq_plan_results", "pq_plan_segments", "pq_plan_destroy", "pq_decode_pages",
    "pq_plan_create_sets", "pq_plan_run_sets",
    "pq_decode_chunks", "pq_host_alloc", "pq_host_free",
    "pq_hybrid_decode", "pq_dict_decode", "pq_levels_decode", "pq_delta_decode32",
    "pq_delta_decode64", "pq_plain_decode", "pq_bool_plain_decode",
    "pq_levels_pack_dev", "pq_levels_pack", "pq_column_entries_dev", "pq_column_entries", "pq_column_entries_noalloc",
    "pq_delta_length_ba_decode", "pq_delta_ba_decode",
    "pq_walk_chunks_dev", "pq_walk_chunks", "pq_chunk_page_descs", "pq_chunks_page_descs",
]


class PageDesc(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("flags", C.c_uint32), ("src_off", C.c_uint64),
                ("src_len", C.c_uint64), ("dst_off", C.c_uint64), ("rep_off", C.c_uint64),
                ("def_off", C.c_uint64), ("dict_off", C.c_uint64), ("n_values", C.c_uint32),
                ("bit_width", C.c_uint32), ("max_rep", C.c_uint32), ("max_def", C.c_uint32),
                ("rep_len", C.c_uint32), ("def_len", C.c_uint32), ("value_kind", C.c_uint32),
                ("elem_size", C.c_uint32), ("dict_len", C.c_uint32), ("dst_cap", C.c_uint32)]


class PageResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("stage", C.c_int32), ("detail", C.c_int32),
                ("n_decoded", C.c_uint32), ("not_null", C.c_uint32), ("n_levels", C.c_uint32),
                ("consumed", C.c_uint64)]


class ChunkJob(C.Structure):
    _fields_ = [("descs", C.c_void_p), ("n_pages", C.c_size_t), ("src", C.c_void_p), ("src_bytes", C.c_size_t),
                ("values", C.c_void_p), ("values_bytes", C.c_size_t), ("levels", C.c_void_p),
                ("levels_bytes", C.c_size_t), ("dict", C.c_void_p), ("dict_bytes", C.c_size_t),
                ("results", C.c_void_p), ("status", C.c_int32), ("reserved", C.c_int32)]


CHUNKS_PINNED = 1


class BufferSet(C.Structure):
    _fields_ = [("d_src", C.c_void_p), ("src_bytes", C.c_size_t), ("d_values", C.c_void_p), ("values_bytes", C.c_size_t),
                ("d_levels", C.c_void_p), ("levels_bytes", C.c_size_t), ("d_dict", C.c_void_p), ("dict_bytes", C.c_size_t)]


class EntriesResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("n_entries", C.c_uint32), ("n_values", C.c_uint32),
                ("reserved", C.c_uint32)]


DESC_DTYPE = np.dtype([("kind", "<u4"), ("flags", "<u4"), ("src_off", "<u8"), ("src_len", "<u8"),
                       ("dst_off", "<u8"), ("rep_off", "<u8"), ("def_off", "<u8"),
                       ("dict_off", "<u8"), ("n_values", "<u4"), ("bit_width", "<u4"),
                       ("max_rep", "<u4"), ("max_def", "<u4"), ("rep_len", "<u4"),
                       ("def_len", "<u4"), ("value_kind", "<u4"), ("elem_size", "<u4"),
                       ("dict_len", "<u4"), ("dst_cap", "<u4")])
RESULT_DTYPE = np.dtype([("status", "<i4"), ("stage", "<i4"), ("detail", "<i4"),
                         ("n_decoded", "<u4"), ("not_null", "<u4"), ("n_levels", "<u4"),
                         ("consumed", "<u8")])
# page-header walk (pq_walk_chunks): one record per column chunk in, one per page out
WALK_HEADER_FAILED, WALK_SECOND_DICTIONARY, WALK_PAGE_TYPE, WALK_MISSING_HEADER = 19, 20, 21, 22
WALK_NEGATIVE_NUM_VALUES, WALK_NEGATIVE_LEVELS_LENGTH, WALK_DICT_ENCODING, WALK_LEVEL_ENCODING = 23, 24, 25, 26
WALK_VALUE_ENCODING, WALK_BLOCK_SIZE, WALK_UNCOMPRESSED_SIZE, WALK_SEEK = 27, 28, 29, 30
CHUNK_DTYPE = np.dtype([("src_off", "<u8"), ("src_len", "<u8"), ("total_compressed_size", "<i8"),
                        ("data_page_offset", "<i8"), ("dictionary_page_offset", "<i8"),
                        ("has_dictionary_page_offset", "<u4"), ("physical_type", "<u4"), ("type_length", "<i4"),
                        ("max_rep", "<u4"), ("max_def", "<u4"), ("codec", "<u4"), ("info_first", "<u4"),
                        ("info_cap", "<u4")])
PAGE_INFO_DTYPE = np.dtype([("type", "<i4"), ("encoding", "<i4"), ("num_values", "<i4"), ("compressed_size", "<i4"),
                            ("uncompressed_size", "<i4"), ("def_len", "<i4"), ("rep_len", "<i4"),
                            ("def_encoding", "<i4"), ("rep_encoding", "<i4"), ("is_compressed", "<u4"),
                            ("header_off", "<u8"), ("data_off", "<u8")])
CHUNK_RESULT_DTYPE = np.dtype([("status", "<i4"), ("n_pages", "<u4"), ("count", "<i8")])
CHUNK_LAYOUT_DTYPE = np.dtype([("values_off", "<u8"), ("levels_off", "<u8"), ("dict_off", "<u8"),
                               ("delta_ba_room", "<u4"), ("_pad", "<u4")])
assert CHUNK_DTYPE.itemsize == 72 and PAGE_INFO_DTYPE.itemsize == 56 and CHUNK_RESULT_DTYPE.itemsize == 16
assert DESC_DTYPE.itemsize == C.sizeof(PageDesc) == 96
assert RESULT_DTYPE.itemsize == C.sizeof(PageResult) == 32

_lib = None


def load():
    """dlopen the library (does not touch the GPU)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libpqdecode.so is not built: run `make -C parquet-go_amd/csrc` "
                "(or __graft_entry__.build()); there is no fallback decode path")
        # One HIP runtime per process: PyTorch ships its own libamdhip64 (soname
        # libamdhip64.so.7) and its extension libraries ask for it by file name, so if
        # the system copy were loaded first torch would bring in a second runtime that
        # finds no GPU.  Importing torch first makes this library bind to torch's copy.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        lib.pq_status_string.restype = C.c_char_p
        lib.pq_last_error.restype = C.c_char_p
        lib.pq_last_error.argtypes = [C.c_void_p]
        lib.pq_ctx_destroy.restype = None
        lib.pq_ctx_destroy.argtypes = [C.c_void_p]
        lib.pq_plan_destroy.restype = None
        lib.pq_plan_destroy.argtypes = [C.c_void_p]
        lib.pq_ctx_create.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
        lib.pq_ctx_synchronize.argtypes = [C.c_void_p]
        lib.pq_ctx_set_option.argtypes = [C.c_void_p, C.c_int, C.c_int64]
        lib.pq_plan_create.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        lib.pq_plan_run.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                    C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        lib.pq_plan_results.argtypes = [C.c_void_p, C.c_void_p]
        lib.pq_plan_create_sets.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                            C.POINTER(C.c_void_p)]
        lib.pq_plan_run_sets.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.pq_decode_chunks.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32]
        lib.pq_host_alloc.argtypes = [C.c_void_p, C.c_size_t]
        lib.pq_host_alloc.restype = C.c_void_p
        lib.pq_host_free.argtypes = [C.c_void_p, C.c_void_p]
        lib.pq_host_free.restype = None
        lib.pq_plan_segments.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        lib.pq_decode_pages.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                        C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                        C.c_void_p, C.c_size_t, C.c_void_p]
        for name in ("pq_bitunpack32_dev", "pq_bitunpack64_dev", "pq_bitunpack32", "pq_bitunpack64"):
            getattr(lib, name).argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t,
                                           C.c_void_p]
        lib.pq_hybrid_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int,
                                         C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32),
                                         C.POINTER(C.c_uint64)]
        lib.pq_dict_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint32,
                                       C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32),
                                       C.POINTER(C.c_uint64)]
        lib.pq_levels_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_int,
                                         C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32),
                                         C.POINTER(C.c_uint64)]
        for name in ("pq_delta_decode32", "pq_delta_decode64"):
            getattr(lib, name).argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32,
                                           C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
        lib.pq_plain_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32,
                                        C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
        for name in ("pq_levels_pack_dev", "pq_levels_pack"):
            getattr(lib, name).argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64]
        for name in ("pq_column_entries_dev", "pq_column_entries", "pq_column_entries_noalloc"):
            getattr(lib, name).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32,
                                           C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.pq_delta_length_ba_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p,
                                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.pq_delta_ba_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p,
                                           C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.pq_bool_plain_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p,
                                             C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
        for name in ("pq_walk_chunks_dev", "pq_walk_chunks"):
            getattr(lib, name).argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                           C.c_size_t, C.c_void_p]
        lib.pq_chunk_page_descs.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p]
        lib.pq_chunks_page_descs.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = lib
    return _lib


def status_string(code):
    return load().pq_status_string(int(code)).decode()


class PqError(RuntimeError):
    def __init__(self, code, detail=""):
        self.code = code
        super().__init__("pqdecode: %s (%d) %s" % (status_string(code), code, detail))


def _np_u8(buf):
    if isinstance(buf, np.ndarray):
        a = np.ascontiguousarray(buf).view(np.uint8).reshape(-1)
    else:
        a = np.frombuffer(bytes(buf), dtype=np.uint8)
    return a


def _ptr(a):
    return a.ctypes.data if (a is not None and a.size) else None


class Context:
    """A pq_ctx: one device + one HIP stream."""

    def __init__(self, device=0, stream=None):
        self._lib = load()
        h = C.c_void_p()
        st = self._lib.pq_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h))
        if st != OK:
            raise PqError(st, "(pq_ctx_create)")
        self._h = h
        self._plans = weakref.WeakSet()

    def close(self):
        if self._h:
            for p in list(self._plans):     # plans go before their context
                p.close()
            self._lib.pq_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st, what=""):
        if st >= ERR_BAD_ARGUMENT:
            raise PqError(st, "%s %s" % (what, self._lib.pq_last_error(self._h).decode()))
        return st

    def synchronize(self):
        self._check(self._lib.pq_ctx_synchronize(self._h))

    def set_option(self, option, value):
        self._check(self._lib.pq_ctx_set_option(self._h, int(option), int(value)), "set_option")

    # ---- host-memory entry points (numpy in / numpy out) ----
    def bitunpack(self, src, w, n_groups, bits=32):
        s = _np_u8(src)
        out = np.empty(8 * n_groups, dtype=np.int32 if bits == 32 else np.int64)
        fn = self._lib.pq_bitunpack32 if bits == 32 else self._lib.pq_bitunpack64
        self._check(fn(self._h, _ptr(s), s.size, int(w), int(n_groups), _ptr(out)), "bitunpack")
        return out

    def hybrid_decode(self, src, w, n, length_prefixed=False):
        s = _np_u8(src)
        out = np.zeros(max(n, 1), dtype=np.int32)
        nd, cons = C.c_uint32(0), C.c_uint64(0)
        st = self._check(self._lib.pq_hybrid_decode(self._h, _ptr(s), s.size, int(w),
                                                    int(bool(length_prefixed)), int(n), _ptr(out),
                                                    C.byref(nd), C.byref(cons)), "hybrid")
        return st, out[:nd.value].copy(), cons.value

    def dict_decode(self, src, dictionary, n):
        s = _np_u8(src)
        d = np.ascontiguousarray(dictionary)
        out = np.zeros(max(n, 1), dtype=d.dtype)
        nd, cons = C.c_uint32(0), C.c_uint64(0)
        st = self._check(self._lib.pq_dict_decode(self._h, _ptr(s), s.size, _ptr(d), d.size,
                                                  d.dtype.itemsize, int(n), _ptr(out), C.byref(nd),
                                                  C.byref(cons)), "dict")
        return st, out[:n], nd.value, cons.value

    def levels_decode(self, src, max_level, n, mode):
        s = _np_u8(src)
        out = np.zeros(max(n, 1), dtype=np.int32)
        nn, cons = C.c_uint32(0), C.c_uint64(0)
        st = self._check(self._lib.pq_levels_decode(self._h, _ptr(s), s.size, int(max_level),
                                                    int(mode), int(n), _ptr(out), C.byref(nn),
                                                    C.byref(cons)), "levels")
        return dict(status=st, levels=out[:n], not_null=nn.value, consumed=cons.value)

    def delta_decode(self, src, n, bits=64):
        s = _np_u8(src)
        out = np.zeros(max(n, 1), dtype=np.int32 if bits == 32 else np.int64)
        nd, cons = C.c_uint32(0), C.c_uint64(0)
        fn = self._lib.pq_delta_decode32 if bits == 32 else self._lib.pq_delta_decode64
        st = self._check(fn(self._h, _ptr(s), s.size, int(n), _ptr(out), C.byref(nd),
                            C.byref(cons)), "delta")
        return dict(status=st, values=out[:nd.value].copy(), n_decoded=nd.value,
                    consumed=cons.value)

    def plain_decode(self, src, n, dtype):
        """Int32/Int64/Float/DoublePlainDecoder.DecodeValues(dest[:n])."""
        s = _np_u8(src)
        dt = np.dtype(dtype)
        out = np.zeros(max(n, 1), dtype=dt)
        nd, cons = C.c_uint32(0), C.c_uint64(0)
        st = self._check(self._lib.pq_plain_decode(self._h, _ptr(s), s.size, dt.itemsize, int(n), _ptr(out),
                                                   C.byref(nd), C.byref(cons)), "plain")
        return dict(status=st, values=out[:nd.value].copy(), n_decoded=nd.value, consumed=cons.value)

    def bool_plain_decode(self, src, n):
        """BooleanPlainDecoder.DecodeValues(dest[:n]) of a fresh decoder; int32 0 / 1 per value."""
        s = _np_u8(src)
        out = np.zeros(max(n, 1), dtype=np.int32)
        nd, cons = C.c_uint32(0), C.c_uint64(0)
        st = self._check(self._lib.pq_bool_plain_decode(self._h, _ptr(s), s.size, int(n), _ptr(out),
                                                        C.byref(nd), C.byref(cons)), "bool")
        return dict(status=st, values=out[:nd.value].copy(), n_decoded=nd.value, consumed=cons.value)

    def decode_pages(self, descs, src, values_bytes=0, levels_bytes=0, dictionary=None):
        """descs: numpy array of DESC_DTYPE.  Returns (results, values_u8, levels_i32)."""
        descs = np.ascontiguousarray(descs, dtype=DESC_DTYPE)
        s = _np_u8(src)
        values = np.zeros(values_bytes, dtype=np.uint8)
        levels = np.zeros(levels_bytes // 4, dtype=np.int32)
        d = _np_u8(dictionary) if dictionary is not None else np.zeros(0, np.uint8)
        results = np.zeros(descs.size, dtype=RESULT_DTYPE)
        st = self._lib.pq_decode_pages(self._h, _ptr(descs), descs.size, _ptr(s), s.size,
                                       _ptr(values), values.size, _ptr(levels), levels.size * 4,
                                       _ptr(d), d.size, _ptr(results))
        if st != OK:
            raise PqError(st, self._lib.pq_last_error(self._h).decode())
        return results, values, levels

    # ---- device-resident entry points (raw device pointers) ----
    def decode_chunks(self, jobs, pinned=False):
        """pq_decode_chunks.  jobs: list of dicts with descs (DESC_DTYPE array), src, values, levels, dict
        (uint8 numpy arrays or None; outputs are written in place).  -> list of (status, results)"""
        n = len(jobs)
        arr = (ChunkJob * max(n, 1))()
        keep = []
        for i, j in enumerate(jobs):
            d = np.ascontiguousarray(j["descs"], dtype=DESC_DTYPE)
            res = np.zeros(max(d.size, 1), dtype=RESULT_DTYPE)
            keep.append((d, res))
            a = arr[i]
            a.descs, a.n_pages = d.ctypes.data, d.size
            for name in ("src", "values", "levels", "dict"):
                buf = j.get(name)
                if buf is not None and buf.size:
                    assert buf.dtype == np.uint8 and buf.flags["C_CONTIGUOUS"]
                    setattr(a, name, buf.ctypes.data)
                    setattr(a, name + "_bytes", buf.size)
            a.results = res.ctypes.data
        st = self._lib.pq_decode_chunks(self._h, arr, n, CHUNKS_PINNED if pinned else 0)
        if st != OK:
            raise PqError(st, "(pq_decode_chunks) " + self._lib.pq_last_error(self._h).decode())
        return [(int(arr[i].status), keep[i][1][: keep[i][0].size]) for i in range(n)]

    def host_alloc(self, nbytes):
        """pq_host_alloc as a uint8 numpy array (pinned); free with host_free(array)."""
        p = self._lib.pq_host_alloc(self._h, int(nbytes))
        if not p:
            raise PqError(ERR_HIP, "(pq_host_alloc) " + self._lib.pq_last_error(self._h).decode())
        a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(int(nbytes),))
        return a

    def host_free(self, array):
        self._lib.pq_host_free(self._h, C.c_void_p(array.ctypes.data))

    def delta_length_ba_decode(self, src, n):
        """ByteArrayDeltaLengthDecoder.Init + DecodeValues(dest[:n]) -> dict(status, init_failed, n_decoded,
        consumed, offsets[:n_decoded + 1], data[:offsets[-1]])."""
        s = _np_u8(src)
        offsets = np.zeros(n + 1, dtype=np.uint32)
        data = np.zeros(max(s.size, 1), dtype=np.uint8)
        nd, cons, bad = C.c_uint32(0), C.c_uint64(0), C.c_int(0)
        st = self._check(self._lib.pq_delta_length_ba_decode(self._h, _ptr(s), s.size, int(n), _ptr(offsets),
                                                             _ptr(data), C.byref(nd), C.byref(cons), C.byref(bad)),
                         "delta length byte array")
        return dict(status=st, init_failed=bad.value, n_decoded=nd.value, consumed=cons.value,
                    offsets=offsets[:nd.value + 1].copy(), data=data[:int(offsets[nd.value])].copy())

    def delta_ba_decode(self, src, n, data_cap):
        """ByteArrayDeltaDecoder.Init + DecodeValues(dest[:n]) with room for data_cap value bytes -> the same
        dictionary as delta_length_ba_decode."""
        s = _np_u8(src)
        offsets = np.zeros(n + 1, dtype=np.uint32)
        data = np.zeros(max(int(data_cap), 1), dtype=np.uint8)
        nd, cons, bad = C.c_uint32(0), C.c_uint64(0), C.c_int(0)
        st = self._check(self._lib.pq_delta_ba_decode(self._h, _ptr(s), s.size, int(n), _ptr(offsets), _ptr(data),
                                                      int(data_cap), C.byref(nd), C.byref(cons), C.byref(bad)),
                         "delta byte array")
        return dict(status=st, init_failed=bad.value, n_decoded=nd.value, consumed=cons.value,
                    offsets=offsets[:nd.value + 1].copy(), data=data[:int(offsets[nd.value])].copy())

    def levels_pack(self, levels, bit_width):
        """PackedArray bytes of `levels` (AppendSingle x n, Flush): ceil(n / 8) * bit_width bytes."""
        lv = np.ascontiguousarray(levels, dtype=np.int32)
        need = (lv.size + 7) // 8 * bit_width
        out = np.zeros(max(need, 1), dtype=np.uint8)
        self._check(self._lib.pq_levels_pack(self._h, _ptr(lv), lv.size, int(bit_width), _ptr(out), need), "pack")
        return out[:need]

    def levels_pack_dev(self, d_levels, n, bit_width, d_out, out_bytes):
        st = self._lib.pq_levels_pack_dev(self._h, C.c_void_p(d_levels), int(n), int(bit_width), C.c_void_p(d_out),
                                          int(out_bytes))
        if st != OK:
            raise PqError(st, self._lib.pq_last_error(self._h).decode())

    def column_entries(self, def_levels, rep_levels, max_def, max_rep, repeated):
        """ColumnStore.Get to the end of the store -> dict(n_entries, n_values, offsets, validity, entry_def)."""
        d = np.ascontiguousarray(def_levels, dtype=np.int32)
        r = np.ascontiguousarray(rep_levels, dtype=np.int32) if rep_levels is not None else np.zeros(0, np.int32)
        n = d.size
        offsets = np.zeros(n + 1, dtype=np.uint32)
        validity = np.zeros((n + 31) // 32 * 4 + 4, dtype=np.uint8)
        entry_def = np.zeros(max(n, 1), dtype=np.int32)
        res = EntriesResult()
        self._check(self._lib.pq_column_entries(self._h, _ptr(d) if n else None, _ptr(r) if r.size else None, n,
                                                int(max_def), int(max_rep), int(bool(repeated)), _ptr(offsets),
                                                _ptr(validity), _ptr(entry_def), C.byref(res)), "entries")
        ne = res.n_entries
        return dict(n_entries=ne, n_values=res.n_values, offsets=offsets[:ne + 1].copy(),
                    validity=validity[:(n + 31) // 32 * 4].copy(), entry_def=entry_def[:ne].copy())

    def column_entries_dev(self, d_def, d_rep, n, max_def, max_rep, repeated, d_offsets, d_validity, d_entry_def,
                           d_result, noalloc=False):
        """noalloc: through pq_column_entries_noalloc (the library's static device workspace)."""
        fn = self._lib.pq_column_entries_noalloc if noalloc else self._lib.pq_column_entries_dev
        st = fn(self._h, C.c_void_p(d_def), C.c_void_p(d_rep), int(n), int(max_def),
                int(max_rep), int(bool(repeated)), C.c_void_p(d_offsets),
                C.c_void_p(d_validity), C.c_void_p(d_entry_def), C.c_void_p(d_result))
        if st != OK:
            raise PqError(st, self._lib.pq_last_error(self._h).decode())

    def bitunpack_dev(self, d_src, src_len, w, n_groups, d_dst, bits=32):
        fn = self._lib.pq_bitunpack32_dev if bits == 32 else self._lib.pq_bitunpack64_dev
        st = fn(self._h, C.c_void_p(d_src), int(src_len), int(w), int(n_groups), C.c_void_p(d_dst))
        if st != OK:
            raise PqError(st, self._lib.pq_last_error(self._h).decode())

    # ---- page-header walk ----
    def walk_chunks(self, src, chunks, infos_cap):
        """ChunkReader.readPages over every chunk at once, host memory in and out -> (results, infos)."""
        s = _np_u8(src)
        ch = np.ascontiguousarray(chunks, dtype=CHUNK_DTYPE)
        infos = np.zeros(max(int(infos_cap), 1), dtype=PAGE_INFO_DTYPE)
        res = np.zeros(max(ch.size, 1), dtype=CHUNK_RESULT_DTYPE)
        st = self._lib.pq_walk_chunks(self._h, _ptr(s), s.size, _ptr(ch), ch.size, _ptr(infos), int(infos_cap),
                                      _ptr(res))
        if st != OK:
            raise PqError(st, "(pq_walk_chunks) " + self._lib.pq_last_error(self._h).decode())
        return res[:ch.size], infos[:int(infos_cap)]

    def walk_chunks_dev(self, d_src, src_bytes, chunks, d_infos, infos_cap):
        """Device-resident walk: d_src / d_infos are device pointers -> results (host)."""
        ch = np.ascontiguousarray(chunks, dtype=CHUNK_DTYPE)
        res = np.zeros(max(ch.size, 1), dtype=CHUNK_RESULT_DTYPE)
        st = self._lib.pq_walk_chunks_dev(self._h, C.c_void_p(d_src), int(src_bytes), _ptr(ch), ch.size,
                                          C.c_void_p(d_infos), int(infos_cap), _ptr(res))
        if st != OK:
            raise PqError(st, "(pq_walk_chunks_dev) " + self._lib.pq_last_error(self._h).decode())
        return res[:ch.size]

    def plan(self, descs, first_desc=None):
        """first_desc: None for a plan over one set of buffers (run()); else n_sets + 1 descriptor indices, set k
        owning descs[first_desc[k]:first_desc[k + 1]] (run_sets())."""
        p = Plan(self, descs, first_desc)
        self._plans.add(p)
        return p


def chunk_page_descs(chunk, infos, layout):
    """pq_chunk_page_descs: descriptors of an uncompressed chunk's data pages.  layout: a one-element
    CHUNK_LAYOUT_DTYPE array, advanced in place.  -> (descs, dict_src, dict_dst, dict_bytes)"""
    lib = load()
    ch = np.ascontiguousarray(chunk, dtype=CHUNK_DTYPE).reshape(1)
    inf = np.ascontiguousarray(infos, dtype=PAGE_INFO_DTYPE)
    descs = np.zeros(max(inf.size, 1), dtype=DESC_DTYPE)
    n = C.c_uint32(0)
    a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    st = lib.pq_chunk_page_descs(_ptr(ch), _ptr(inf), inf.size, _ptr(layout), _ptr(descs), C.byref(n), C.byref(a),
                                 C.byref(b), C.byref(c))
    if st != OK:
        raise PqError(st, "(pq_chunk_page_descs) " + status_string(st))
    return descs[:n.value], a.value, b.value, c.value


class Plan:
    """A pq_plan: uploaded descriptor table; run() only launches kernels."""

    def __init__(self, ctx, descs, first_desc=None):
        self._ctx = ctx
        self._lib = ctx._lib
        descs = np.ascontiguousarray(descs, dtype=DESC_DTYPE)
        self.n = descs.size
        h = C.c_void_p()
        if first_desc is None:
            st = self._lib.pq_plan_create(ctx._h, _ptr(descs), descs.size, C.byref(h))
        else:
            first = np.ascontiguousarray(first_desc, dtype=np.uint32)
            self.n_sets = first.size - 1
            st = self._lib.pq_plan_create_sets(ctx._h, _ptr(descs), descs.size, first.ctypes.data, self.n_sets, C.byref(h))
        if st != OK:
            raise PqError(st, "(pq_plan_create) " + self._lib.pq_last_error(ctx._h).decode())
        self._h = h

    def run_sets(self, sets):
        """sets: one (d_src, src_bytes, d_values, values_bytes, d_levels, levels_bytes, d_dict, dict_bytes) per
        buffer set, device pointers as integers (0 = none)."""
        arr = (BufferSet * max(len(sets), 1))()
        for a, t in zip(arr, sets):
            t = tuple(t) + (0,) * (8 - len(t))
            (a.d_src, a.src_bytes, a.d_values, a.values_bytes, a.d_levels, a.levels_bytes, a.d_dict, a.dict_bytes) = (
                t[0] or None, int(t[1]), t[2] or None, int(t[3]), t[4] or None, int(t[5]), t[6] or None, int(t[7]))
        st = self._lib.pq_plan_run_sets(self._h, arr, len(sets))
        if st != OK:
            raise PqError(st, "(pq_plan_run_sets) " + self._lib.pq_last_error(self._ctx._h).decode())

    def run(self, d_src, src_bytes, d_values=0, values_bytes=0, d_levels=0, levels_bytes=0,
            d_dict=0, dict_bytes=0):
        st = self._lib.pq_plan_run(self._h, C.c_void_p(d_src), int(src_bytes),
                                   C.c_void_p(d_values) if d_values else None, int(values_bytes),
                                   C.c_void_p(d_levels) if d_levels else None, int(levels_bytes),
                                   C.c_void_p(d_dict) if d_dict else None, int(dict_bytes))
        if st != OK:
            raise PqError(st, "(pq_plan_run) " + self._lib.pq_last_error(self._ctx._h).decode())

    def results(self):
        out = np.zeros(self.n, dtype=RESULT_DTYPE)
        st = self._lib.pq_plan_results(self._h, _ptr(out))
        if st != OK:
            raise PqError(st, "(pq_plan_results) " + self._lib.pq_last_error(self._ctx._h).decode())
        return out

    def segments(self):
        """pq_plan_segments: one SEGMENT_DTYPE record per segment of the delta pages cut into segments."""
        n = C.c_size_t(0)
        st = self._lib.pq_plan_segments(self._h, None, 0, C.byref(n))
        out = np.zeros(n.value, dtype=SEGMENT_DTYPE)
        if st == OK and n.value:
            st = self._lib.pq_plan_segments(self._h, _ptr(out), n.value, C.byref(n))
        if st != OK:
            raise PqError(st, "(pq_plan_segments) " + self._lib.pq_last_error(self._ctx._h).decode())
        return out

    def close(self):
        if self._h:
            self._lib.pq_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


SEGMENT_DTYPE = np.dtype([("key", "<u4"), ("exit", "<u4"), ("n_blocks", "<u4"), ("exit_key", "<u4"), ("sum", "<u8"),
                          ("exit_min_delta", "<u4"), ("decode", "<u4"), ("value", "<u8"), ("entry", "<u4"),
                          ("position", "<u4"), ("stop", "<u4"), ("reserved", "<u4", (3,))])
assert SEGMENT_DTYPE.itemsize == 64
DICT_COPY_DTYPE = np.dtype([("src", "<u8"), ("dst", "<u8"), ("bytes", "<u8")])


def chunks_page_descs(chunks, results, infos, layout, with_status=False, descs_cap=None):
    """pq_chunks_page_descs: descriptors for every walked chunk in one call.
    -> (descs, first_desc[n_chunks + 1], dictionary copies[, per-chunk status, return code])
    Without with_status any chunk the library cannot describe raises, as does a full table."""
    lib = load()
    ch = np.ascontiguousarray(chunks, dtype=CHUNK_DTYPE)
    res = np.ascontiguousarray(results, dtype=CHUNK_RESULT_DTYPE)
    inf = np.ascontiguousarray(infos, dtype=PAGE_INFO_DTYPE)
    cap = int(res["n_pages"].sum()) if descs_cap is None else int(descs_cap)
    descs = np.zeros(max(cap, 1), dtype=DESC_DTYPE)
    first = np.zeros(ch.size + 1, dtype=np.uint32)
    copies = np.zeros(max(ch.size, 1), dtype=DICT_COPY_DTYPE)
    status = np.zeros(max(ch.size, 1), dtype=np.int32)
    nc = C.c_uint32(0)
    st = lib.pq_chunks_page_descs(_ptr(ch), _ptr(res), ch.size, _ptr(inf), _ptr(layout), _ptr(descs), cap,
                                  _ptr(first), _ptr(copies), C.byref(nc), _ptr(status))
    status = status[:ch.size]
    if with_status:
        return descs[:int(first[-1])], first, copies[:nc.value], status, st
    if st != OK:
        raise PqError(st, "(pq_chunks_page_descs) " + status_string(st))
    bad = np.flatnonzero(status != OK)
    if bad.size:
        raise PqError(int(status[bad[0]]), "(pq_chunks_page_descs) chunk %d: %s" % (bad[0], status_string(status[bad[0]])))
    return descs[:int(first[-1])], first, copies[:nc.value]
