"""ctypes binding of the CPU oracle (oracle/libpqoracle.so).

Test infrastructure only: imported by tests/, by __graft_entry__.smoke() and by
the cpu_baseline leg of bench.py — never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# PQ_ORACLE_LIB: another build of the same library (oracle/libpqoracle_asan.so, run with the sanitizer
# runtime preloaded: LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0)
_SO = os.environ.get("PQ_ORACLE_LIB") or os.path.join(_ROOT, "oracle", "libpqoracle.so")

# status codes (oracle/pq_oracle.h, enum pqo_status)
OK, EOF, UNEXPECTED_EOF, EMPTY_BP_RUN, EMPTY_RLE_RUN, RLE_VALUE_TOO_LARGE = 0, 1, 2, 3, 4, 5
INT32_OUT_OF_RANGE, VARINT_OVERFLOW, INVALID_BLOCK_SIZE, INVALID_MINIBLOCK_COUNT = 6, 7, 8, 9
INVALID_MINIBLOCK_WIDTH, DICT_INVALID_WIDTH, DICT_INVALID_INDEX, NOT_INITIALIZED = 10, 11, 12, 13
INVALID_STREAM, PAGE_VALUES_FAILED = 14, 15

VALUES_DICT, VALUES_DELTA32, VALUES_DELTA64, VALUES_PLAIN, VALUES_BOOL_PLAIN, VALUES_BOOL_RLE = 1, 2, 3, 4, 5, 6
VALUES_DELTA_LENGTH_BA, VALUES_DELTA_BA = 7, 8
BYTE_ARRAY_NEGATIVE_LEN, BYTE_ARRAY_COUNT_MISMATCH, BYTE_ARRAY_INVALID_PREFIX, OUT_OF_RANGE_CAP = 16, 17, 18, 103


def build():
    """Compile the oracle if the shared object is missing or stale."""
    src = os.path.join(_ROOT, "oracle", "pq_oracle.c")
    hdr = os.path.join(_ROOT, "oracle", "pq_oracle.h")
    if (not os.path.exists(_SO)) or (
            os.path.exists(src) and os.path.getmtime(_SO) < max(os.path.getmtime(src),
                                                                os.path.getmtime(hdr))):
        subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle")])


class PageIn(C.Structure):
    _fields_ = [("src", C.c_void_p), ("src_len", C.c_uint64), ("version", C.c_uint32),
                ("num_values", C.c_uint32), ("max_rep", C.c_uint32), ("max_def", C.c_uint32),
                ("rep_len", C.c_uint32), ("def_len", C.c_uint32), ("value_kind", C.c_uint32),
                ("elem_size", C.c_uint32), ("dict", C.c_void_p), ("dict_len", C.c_uint32),
                ("data_cap", C.c_uint64)]


class PageOut(C.Structure):
    _fields_ = [("status", C.c_int32), ("stage", C.c_int32), ("detail", C.c_int32),
                ("n", C.c_uint32), ("not_null", C.c_uint32), ("values_decoded", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.pqo_page_read.restype = None
    return _lib


def _u8(buf):
    a = np.frombuffer(bytes(buf), dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.size == 0:
        a = np.zeros(1, dtype=np.uint8)[:0]
    return a


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def unpack8_32(data, w):
    d = _u8(list(data) + [0]) if len(data) == 0 else _u8(data)
    out = np.zeros(8, dtype=np.int32)
    lib().pqo_unpack8_32(_ptr(d), C.c_int(w), _ptr(out))
    return out


def unpack8_64(data, w):
    d = _u8(list(data) + [0]) if len(data) == 0 else _u8(data)
    out = np.zeros(8, dtype=np.int64)
    lib().pqo_unpack8_64(_ptr(d), C.c_int(w), _ptr(out))
    return out


def pack8_32(values, w):
    v = np.ascontiguousarray(values, dtype=np.int32)
    out = np.zeros(max(w, 1), dtype=np.uint8)
    lib().pqo_pack8_32(_ptr(v), C.c_int(w), _ptr(out))
    return out[:w]


def pack8_64(values, w):
    v = np.ascontiguousarray(values, dtype=np.int64)
    out = np.zeros(max(w, 1), dtype=np.uint8)
    lib().pqo_pack8_64(_ptr(v), C.c_int(w), _ptr(out))
    return out[:w]


def bitunpack32(src, w, n_groups):
    s = _u8(src)
    out = np.zeros(8 * n_groups, dtype=np.int32)
    lib().pqo_bitunpack32(_ptr(s), C.c_int(w), C.c_size_t(n_groups), _ptr(out))
    return out


def bitunpack64(src, w, n_groups):
    s = _u8(src)
    out = np.zeros(8 * n_groups, dtype=np.int64)
    lib().pqo_bitunpack64(_ptr(s), C.c_int(w), C.c_size_t(n_groups), _ptr(out))
    return out


def hybrid_decode(src, w, n, length_prefixed=False):
    """-> (status, values[:n_decoded], consumed)"""
    s = _u8(src)
    out = np.zeros(max(n, 1), dtype=np.int32)
    nd, cons = C.c_uint32(0), C.c_uint64(0)
    st = lib().pqo_hybrid_decode(_ptr(s), C.c_size_t(s.size), C.c_int(w),
                                 C.c_int(int(length_prefixed)), C.c_uint32(n), _ptr(out),
                                 C.byref(nd), C.byref(cons))
    return st, out[:nd.value].copy(), cons.value


def dict_decode(src, dictionary, n):
    """dictionary: 1-D numpy array (int32/int64/float32/float64). -> (status, values, n_decoded, consumed)"""
    s = _u8(src)
    d = np.ascontiguousarray(dictionary)
    out = np.zeros(max(n, 1), dtype=d.dtype)
    nd, cons = C.c_uint32(0), C.c_uint64(0)
    dptr = _ptr(d) if d.size else None
    st = lib().pqo_dict_decode(_ptr(s), C.c_size_t(s.size), dptr, C.c_uint32(d.size),
                               C.c_uint32(d.dtype.itemsize), C.c_uint32(n), _ptr(out),
                               C.byref(nd), C.byref(cons))
    return st, out[:n], nd.value, cons.value


def levels_decode(src, max_level, n, mode):
    """mode 0 const / 1 V1 length-prefixed / 2 V2 raw.
    -> dict(status, levels, packed, tail, not_null, consumed)"""
    s = _u8(src)
    bw = int(max_level).bit_length() if max_level < 65536 else 16
    levels = np.zeros(max(n, 1), dtype=np.int32)
    packed = np.zeros((n // 8 + 2) * max(bw, 1), dtype=np.uint8)
    plen, cons = C.c_uint64(0), C.c_uint64(0)
    tail = np.zeros(8, dtype=np.int32)
    tlen, nn = C.c_uint32(0), C.c_uint32(0)
    st = lib().pqo_levels_decode(_ptr(s), C.c_size_t(s.size), C.c_uint32(max_level), C.c_int(mode),
                                 C.c_uint32(n), _ptr(levels), _ptr(packed), C.byref(plen),
                                 _ptr(tail), C.byref(tlen), C.byref(nn), C.byref(cons))
    return dict(status=st, levels=levels[:n], packed=packed[:plen.value].copy(),
                tail=tail[:tlen.value].copy(), not_null=nn.value, consumed=cons.value)


def _delta(fn, dtype, src, n):
    s = _u8(src)
    out = np.zeros(max(n, 1), dtype=dtype)
    nd, cons, vc, ist = C.c_uint32(0), C.c_uint64(0), C.c_uint32(0), C.c_int(0)
    st = fn(_ptr(s), C.c_size_t(s.size), C.c_uint32(n), _ptr(out), C.byref(nd), C.byref(cons),
            C.byref(vc), C.byref(ist))
    return dict(status=st, values=out[:nd.value].copy(), n_decoded=nd.value,
                consumed=cons.value, values_count=vc.value, init_status=ist.value)


def delta_decode32(src, n):
    return _delta(lib().pqo_delta_decode32, np.int32, src, n)


def delta_decode64(src, n):
    return _delta(lib().pqo_delta_decode64, np.int64, src, n)


def plain_decode(src, n, dtype):
    """-> dict(status, values[:n_decoded], n_decoded, consumed)"""
    s = _u8(src)
    dt = np.dtype(dtype)
    out = np.zeros(max(n, 1), dtype=dt)
    nd, cons = C.c_uint32(0), C.c_uint64(0)
    st = lib().pqo_plain_decode(_ptr(s), C.c_size_t(s.size), C.c_uint32(dt.itemsize), C.c_uint32(n), _ptr(out),
                                C.byref(nd), C.byref(cons))
    return dict(status=st, values=out[:nd.value].copy(), n_decoded=nd.value, consumed=cons.value)


def bool_plain_decode(src, n):
    s = _u8(src)
    out = np.zeros(max(n, 1), dtype=np.int32)
    nd, cons = C.c_uint32(0), C.c_uint64(0)
    st = lib().pqo_bool_plain_decode(_ptr(s), C.c_size_t(s.size), C.c_uint32(n), _ptr(out), C.byref(nd),
                                     C.byref(cons))
    return dict(status=st, values=out[:nd.value].copy(), n_decoded=nd.value, consumed=cons.value)


def delta_length_ba_decode(src, n):
    """-> dict(status, init_status, n_decoded, consumed, values_count, offsets[:n_decoded + 1], data)"""
    s = _u8(src)
    offsets = np.zeros(n + 1, dtype=np.uint32)
    data = np.zeros(s.size + 1, dtype=np.uint8)
    nd, cons, vc, init = C.c_uint32(0), C.c_uint64(0), C.c_uint32(0), C.c_int(0)
    st = lib().pqo_delta_length_ba_decode(_ptr(s), C.c_size_t(s.size), C.c_uint32(n), _ptr(offsets), _ptr(data),
                                          C.byref(nd), C.byref(cons), C.byref(vc), C.byref(init))
    return dict(status=st, init_status=init.value, n_decoded=nd.value, consumed=cons.value, values_count=vc.value,
                offsets=offsets[:nd.value + 1].copy(), data=data[:int(offsets[nd.value])].copy())


def delta_ba_decode(src, n, data_cap):
    """-> the dictionary of delta_length_ba_decode, for ByteArrayDeltaDecoder with data_cap bytes of room"""
    s = _u8(src)
    offsets = np.zeros(n + 1, dtype=np.uint32)
    data = np.zeros(int(data_cap) + 1, dtype=np.uint8)
    nd, cons, vc, init = C.c_uint32(0), C.c_uint64(0), C.c_uint32(0), C.c_int(0)
    st = lib().pqo_delta_ba_decode(_ptr(s), C.c_size_t(s.size), C.c_uint32(n), _ptr(offsets), _ptr(data),
                                   C.c_uint64(int(data_cap)), C.byref(nd), C.byref(cons), C.byref(vc), C.byref(init))
    return dict(status=st, init_status=init.value, n_decoded=nd.value, consumed=cons.value, values_count=vc.value,
                offsets=offsets[:nd.value + 1].copy(), data=data[:int(offsets[nd.value])].copy())


class ChunkIn(C.Structure):
    _fields_ = [("src", C.c_void_p), ("src_len", C.c_uint64), ("total_compressed_size", C.c_int64),
                ("data_page_offset", C.c_int64), ("dictionary_page_offset", C.c_int64),
                ("has_dictionary_page_offset", C.c_uint32), ("physical_type", C.c_uint32),
                ("type_length", C.c_int32), ("max_rep", C.c_uint32), ("max_def", C.c_uint32), ("codec", C.c_uint32)]


WALK_INFO_DTYPE = np.dtype([("type", "<i4"), ("encoding", "<i4"), ("num_values", "<i4"), ("compressed_size", "<i4"),
                            ("uncompressed_size", "<i4"), ("def_len", "<i4"), ("rep_len", "<i4"),
                            ("def_encoding", "<i4"), ("rep_encoding", "<i4"), ("is_compressed", "<u4"),
                            ("header_off", "<u8"), ("data_off", "<u8")])


def walk_chunk(src, total_compressed_size, data_page_offset, dictionary_page_offset=None, physical_type=1,
               type_length=-1, max_rep=0, max_def=0, codec=0, cap=4096):
    """ChunkReader.readPages -> dict(status, infos[:n_pages], count)"""
    s = _u8(src)
    cin = ChunkIn()
    cin.src = s.ctypes.data if s.size else None
    cin.src_len = s.size
    cin.total_compressed_size = total_compressed_size
    cin.data_page_offset = data_page_offset
    cin.has_dictionary_page_offset = 0 if dictionary_page_offset is None else 1
    cin.dictionary_page_offset = dictionary_page_offset or 0
    cin.physical_type, cin.type_length = physical_type, type_length
    cin.max_rep, cin.max_def, cin.codec = max_rep, max_def, codec
    infos = np.zeros(max(cap, 1), dtype=WALK_INFO_DTYPE)
    n, count = C.c_uint32(0), C.c_int64(0)
    st = lib().pqo_walk_chunk(C.byref(cin), _ptr(infos), C.c_uint32(cap), C.byref(n), C.byref(count))
    return dict(status=st, infos=infos[:n.value].copy(), count=count.value)


def levels_pack(levels, w):
    """PackedArray bytes of AppendSingle x n and, for n > 0, Flush."""
    lv = np.ascontiguousarray(levels, dtype=np.int32)
    out = np.zeros((lv.size + 7) // 8 * w + 1, dtype=np.uint8)
    fn = lib().pqo_levels_pack
    fn.restype = C.c_uint64
    got = fn(_ptr(lv), C.c_uint64(lv.size), C.c_int(w), _ptr(out))
    return out[:got].copy()


def column_entries(def_levels, rep_levels, max_def, max_rep, repeated):
    """ColumnStore.Get until "out of range" -> dict(n_entries, n_values, offsets, validity, entry_def)."""
    d = np.ascontiguousarray(def_levels, dtype=np.int32)
    r = np.ascontiguousarray(rep_levels if rep_levels is not None else np.zeros(d.size), dtype=np.int32)
    n = d.size
    offsets = np.zeros(n + 1, dtype=np.uint32)
    validity = np.zeros((n + 31) // 32 * 4 + 4, dtype=np.uint8)
    entry_def = np.zeros(max(n, 1), dtype=np.int32)
    ne, nv = C.c_uint32(0), C.c_uint32(0)
    fn = lib().pqo_column_entries
    fn.restype = None
    fn(_ptr(d), _ptr(r), C.c_uint64(n), C.c_int32(max_def), C.c_int32(max_rep), C.c_int(int(bool(repeated))),
       _ptr(offsets), _ptr(validity), _ptr(entry_def), C.byref(ne), C.byref(nv))
    return dict(n_entries=ne.value, n_values=nv.value, offsets=offsets[:ne.value + 1].copy(),
                validity=validity[:(n + 31) // 32 * 4].copy(), entry_def=entry_def[:ne.value].copy())


def hybrid_encode(values, w):
    v = np.ascontiguousarray(values, dtype=np.int32)
    cap = v.size * 5 + v.size // 8 + 64
    out = np.zeros(cap, dtype=np.uint8)
    ol = C.c_size_t(0)
    vp = _ptr(v) if v.size else None
    rc = lib().pqo_hybrid_encode(vp, C.c_size_t(v.size), C.c_int(w), _ptr(out), C.c_size_t(cap),
                                 C.byref(ol))
    assert rc == 0, rc
    return out[:ol.value].copy()


def _delta_encode(fn, dtype, values, block, miniblocks):
    v = np.ascontiguousarray(values, dtype=dtype)
    cap = v.size * 10 + (v.size // 128 + 2) * (miniblocks + 16) + block * 9 + 128
    out = np.zeros(cap, dtype=np.uint8)
    ol = C.c_size_t(0)
    vp = _ptr(v) if v.size else None
    rc = fn(vp, C.c_size_t(v.size), C.c_int(block), C.c_int(miniblocks), _ptr(out),
            C.c_size_t(cap), C.byref(ol))
    return rc, out[:ol.value].copy()


def delta_encode32(values, block=128, miniblocks=4):
    return _delta_encode(lib().pqo_delta_encode32, np.int32, values, block, miniblocks)


def delta_encode64(values, block=128, miniblocks=4):
    return _delta_encode(lib().pqo_delta_encode64, np.int64, values, block, miniblocks)


def page_read(src, version, num_values, max_rep, max_def, value_kind, dictionary=None,
              rep_len=0, def_len=0, value_dtype=None, data_cap=0):
    """-> dict(status, stage, detail, n, not_null, values_decoded, rep, def_, values).  Byte-array value kinds:
    values = (num_values + 1) uint32 offsets followed by the value bytes (data_cap of them for DELTA_BA)."""
    s = _u8(src)
    pin = PageIn()
    pin.src = s.ctypes.data if s.size else None
    pin.src_len = s.size
    pin.version = version
    pin.num_values = num_values
    pin.max_rep, pin.max_def = max_rep, max_def
    pin.rep_len, pin.def_len = rep_len, def_len
    pin.value_kind = value_kind
    d = None
    if value_kind == VALUES_DICT:
        d = np.ascontiguousarray(dictionary)
        pin.dict = d.ctypes.data if d.size else None
        pin.dict_len = d.size
        pin.elem_size = d.dtype.itemsize
        vdt = d.dtype
    elif value_kind == VALUES_PLAIN:
        vdt = np.dtype(value_dtype)
        pin.elem_size = vdt.itemsize
    elif value_kind in (VALUES_BOOL_PLAIN, VALUES_BOOL_RLE):
        vdt = np.dtype(np.int32)
    elif value_kind in (VALUES_DELTA_LENGTH_BA, VALUES_DELTA_BA):
        vdt = np.dtype(np.uint8)
        pin.data_cap = int(data_cap)
    else:
        vdt = np.dtype(np.int32 if value_kind == VALUES_DELTA32 else np.int64)
    if value_dtype is not None:
        vdt = np.dtype(value_dtype)
    rep = np.zeros(max(num_values, 1), dtype=np.int32)
    deff = np.zeros(max(num_values, 1), dtype=np.int32)
    vals = np.zeros(max(num_values, 1), dtype=vdt)
    if value_kind in (VALUES_DELTA_LENGTH_BA, VALUES_DELTA_BA):   # (num_values + 1) uint32 offsets, then the bytes
        room = s.size if value_kind == VALUES_DELTA_LENGTH_BA else int(data_cap)
        vals = np.zeros(4 * (num_values + 1) + room + 8, dtype=np.uint8)
    po = PageOut()
    lib().pqo_page_read(C.byref(pin), _ptr(rep), _ptr(deff), _ptr(vals), C.byref(po))
    return dict(status=po.status, stage=po.stage, detail=po.detail, n=po.n, not_null=po.not_null,
                values_decoded=po.values_decoded, rep=rep[:num_values], def_=deff[:num_values],
                values=vals if value_kind in (VALUES_DELTA_LENGTH_BA, VALUES_DELTA_BA) else vals[:num_values])
