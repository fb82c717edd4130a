"""Synthetic encoded pages for tests and the benchmark (numpy, vectorised).

Writers for the three wire formats on the decode path, producing byte-for-byte
what the reference's own encoders emit for the same values (hybrid:
encoding/hybrid-encoder.go:88-237 — with each bit-packed run carrying only its
own groups; delta: encoding/delta-binary-packed-encoder.go:68-92,140-238,290-384),
plus the splitmix64 generator the workload definitions use.  tests/ checks these
writers against the CPU oracle's restated encoders on small inputs; nothing here
decodes anything.
"""
import numpy as np

_U64 = np.uint64


def splitmix64(seed, n):
    """n outputs of splitmix64 seeded with `seed` (counter form, vectorised)."""
    with np.errstate(over="ignore"):
        z = _U64(seed & 0xFFFFFFFFFFFFFFFF) + np.arange(1, n + 1, dtype=np.uint64) * _U64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> _U64(30))) * _U64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> _U64(27))) * _U64(0x94D049BB133111EB)
        return z ^ (z >> _U64(31))


def uleb128(x):
    out = []
    x = int(x)
    while x >= 0x80:
        out.append((x & 0x7F) | 0x80)
        x >>= 7
    out.append(x)
    return out


def zigzag64(v):
    v = int(v)
    return ((v << 1) ^ (v >> 63)) & 0xFFFFFFFFFFFFFFFF


def pack_bits(values, w):
    """LSB-first packing of len(values) (a multiple of 8) w-bit fields -> uint8 array.

    Consecutive 8-value groups of w bytes each concatenate into one bit stream in
    which value i occupies bits [i*w, (i+1)*w)."""
    v = np.ascontiguousarray(values).astype(np.uint64, copy=False).reshape(-1)
    n = v.size
    assert n % 8 == 0 and 0 <= w <= 64
    if w == 0 or n == 0:
        return np.zeros(0, dtype=np.uint8)
    if w == 64:
        return v.view(np.uint8).copy()
    pad = (-n) % 64
    if pad:
        v = np.concatenate([v, np.zeros(pad, dtype=np.uint64)])
    v = v & _U64((1 << w) - 1)
    rows = v.reshape(-1, 64)                    # 64 values = exactly w 64-bit words
    out = np.zeros((rows.shape[0], w + 1), dtype=np.uint64)
    for j in range(64):
        b = j * w
        k, s = b >> 6, b & 63
        out[:, k] |= rows[:, j] << _U64(s)
        if s + w > 64:
            out[:, k + 1] |= rows[:, j] >> _U64(64 - s)
    return np.ascontiguousarray(out[:, :w]).reshape(-1).view(np.uint8)[: n * w // 8].copy()


# ---------------------------------------------------------------------------
# RLE / bit-packed hybrid

def hybrid_encode(values, w):
    """The reference HybridEncoder's stream for `values` (each < 2**w).

    The encoder buffers 8 values at a time; when the 8 values following a flush
    point are all equal it switches to an RLE run that lasts while the value
    repeats; otherwise the 8 values become one bit-packed group.  Bit-packed
    groups are emitted in runs of at most 63 groups (one-byte header)."""
    v = np.ascontiguousarray(values, dtype=np.int64).reshape(-1)
    n = v.size
    if n == 0:
        return np.zeros(0, dtype=np.uint8)
    # forward run length at every index
    change = np.flatnonzero(v[1:] != v[:-1]) + 1
    starts = np.concatenate([[0], change])
    ends = np.concatenate([change, [n]])
    run_end = np.repeat(ends, ends - starts)
    rl = run_end - np.arange(n)
    eq8 = rl >= 8
    cand = [np.flatnonzero(eq8[ph::8]) * 8 + ph for ph in range(8)]

    parts = []       # (kind, a, b): 'bp' value range [a,b) ; 'rle' (start, length)
    p = 0
    while p < n:
        c = cand[p % 8]
        k = np.searchsorted(c, p)
        q = int(c[k]) if k < c.size else n
        if q > p:
            parts.append(("bp", p, q))
        if q < n:
            length = int(rl[q])
            parts.append(("rle", q, length))
            p = q + length
        else:
            p = n
    # gather all bit-packed values (each part padded to whole groups) and pack once
    bp_chunks, bp_groups = [], []
    for part in parts:
        if part[0] == "bp":
            a, b = part[1], part[2]
            g = (b - a + 7) // 8
            chunk = np.zeros(g * 8, dtype=np.int64)
            chunk[: b - a] = v[a:b]
            bp_chunks.append(chunk)
            bp_groups.append(g)
    packed = pack_bits(np.concatenate(bp_chunks), w) if bp_chunks else np.zeros(0, np.uint8)
    out = []
    off = 0
    gi = 0
    nb = (w + 7) // 8
    for part in parts:
        if part[0] == "bp":
            g = bp_groups[gi]
            gi += 1
            while g > 0:
                take = min(g, 63)
                out.append(np.array(uleb128((take << 1) | 1), dtype=np.uint8))
                out.append(packed[off: off + take * w])
                off += take * w
                g -= take
        else:
            q, length = part[1], part[2]
            out.append(np.array(uleb128(length << 1), dtype=np.uint8))
            val = int(v[q]) & 0xFFFFFFFF
            out.append(np.array([(val >> (8 * i)) & 0xFF for i in range(nb)], dtype=np.uint8))
    return np.concatenate(out) if out else np.zeros(0, np.uint8)


def hybrid_encode_bitpacked(values, w):
    """Fast path of hybrid_encode for inputs with no 8-long repeat at any group
    boundary (e.g. uniform random keys): bit-packed runs of 63 groups only."""
    v = np.ascontiguousarray(values, dtype=np.int64).reshape(-1)
    n = v.size
    g = (n + 7) // 8
    if g * 8 != n:
        v = np.concatenate([v, np.zeros(g * 8 - n, dtype=np.int64)])
    packed = pack_bits(v, w)
    full, rest = divmod(g, 63)
    run_bytes = 63 * w
    out = np.empty(full * (run_bytes + 1) + (rest * w + 1 if rest else 0), dtype=np.uint8)
    if full:
        body = out[: full * (run_bytes + 1)].reshape(full, run_bytes + 1)
        body[:, 0] = (63 << 1) | 1
        body[:, 1:] = packed[: full * run_bytes].reshape(full, run_bytes)
    if rest:
        tail = out[full * (run_bytes + 1):]
        tail[0] = (rest << 1) | 1
        tail[1:] = packed[full * run_bytes:]
    return out


def dict_page(keys, w=None, dict_len=None):
    """RLE_DICTIONARY data section: bit-width byte + hybrid stream (types/dict.go:34-57)."""
    if w is None:
        w = int(dict_len).bit_length()
    return np.concatenate([np.array([w], dtype=np.uint8), hybrid_encode(keys, w)])


def length_prefixed(stream):
    """V1 level section: u32 LE byte length + stream (encoding/hybrid-decoder.go:57-68)."""
    s = np.ascontiguousarray(stream, dtype=np.uint8)
    return np.concatenate([np.frombuffer(np.uint32(s.size).tobytes(), dtype=np.uint8), s])


# ---------------------------------------------------------------------------
# DELTA_BINARY_PACKED

def _bit_length_u64(x):
    """Element-wise bits.Len64 of a uint64 array."""
    x = x.astype(np.uint64)
    out = np.zeros(x.shape, dtype=np.int64)
    for s in (32, 16, 8, 4, 2, 1):
        m = (x >> _U64(s)) != 0
        out[m] += s
        x = np.where(m, x >> _U64(s), x)
    out[x != 0] += 1
    return out


def delta_encode(values, bits=64, block=128, miniblocks=4):
    """The reference DeltaBinaryPackEncoder32/64 stream for `values`."""
    assert block % 128 == 0 and block % miniblocks == 0 and (block // miniblocks) % 32 == 0
    mbv = block // miniblocks
    sdt, udt = (np.int64, np.uint64) if bits == 64 else (np.int32, np.uint32)
    v = np.ascontiguousarray(values, dtype=sdt).reshape(-1)
    n = v.size
    head = uleb128(block) + uleb128(miniblocks) + uleb128(n)
    head += uleb128(zigzag64(int(v[0]) if n else 0))
    head = np.array(head, dtype=np.uint8)
    if n <= 1:
        return head
    with np.errstate(over="ignore"):
        deltas = (v[1:].view(udt) - v[:-1].view(udt)).view(sdt)
    nd = deltas.size
    nblocks = (nd + block - 1) // block
    padn = nblocks * block - nd
    big = np.iinfo(sdt).max
    dpad = np.concatenate([deltas, np.full(padn, big, dtype=sdt)]).reshape(nblocks, block)
    # minDelta starts at math.MaxInt32 for both element sizes
    md = np.minimum(dpad.min(axis=1), sdt(2147483647)).astype(sdt)
    with np.errstate(over="ignore"):
        resid = (dpad.view(udt) - md.view(udt)[:, None]).astype(np.uint64)
    valid = (np.arange(nblocks * block) < nd).reshape(nblocks, block)
    resid[~valid] = 0                                   # padding packs as zeros
    rm = resid.reshape(nblocks, miniblocks, mbv)
    widths = _bit_length_u64(rm.max(axis=2))            # (nblocks, miniblocks)
    # a miniblock exists iff it holds at least one real delta
    mb_first = np.arange(nblocks)[:, None] * block + np.arange(miniblocks)[None, :] * mbv
    present = mb_first < nd
    widths = np.where(present, widths, 0)
    mb_bytes = np.where(present, widths * (mbv // 8), 0)

    # block headers: zig-zag varint of minDelta + `miniblocks` width bytes
    md64 = md.astype(np.int64)
    zz = ((md64 << 1) ^ (md64 >> 63)).view(np.uint64)
    vlen = np.maximum(1, (_bit_length_u64(zz) + 6) // 7)
    blk_bytes = vlen + miniblocks + mb_bytes.sum(axis=1)
    blk_off = np.concatenate([[0], np.cumsum(blk_bytes)[:-1]])
    total = int(blk_bytes.sum())
    out = np.zeros(total, dtype=np.uint8)
    for k in range(10):
        sel = vlen > k
        if not sel.any():
            break
        byte = (zz[sel] >> _U64(7 * k)) & _U64(0x7F)
        more = vlen[sel] > k + 1
        out[blk_off[sel] + k] = (byte | np.where(more, _U64(0x80), _U64(0))).astype(np.uint8)
    wpos = blk_off + vlen
    out[(wpos[:, None] + np.arange(miniblocks)[None, :]).reshape(-1)] = widths.reshape(-1).astype(np.uint8)
    data_off = wpos + miniblocks
    mb_off = data_off[:, None] + np.concatenate(
        [np.zeros((nblocks, 1), dtype=np.int64), np.cumsum(mb_bytes, axis=1)[:, :-1]], axis=1)
    flat_w = widths.reshape(-1)
    flat_off = mb_off.reshape(-1)
    flat_present = present.reshape(-1)
    flat_vals = rm.reshape(-1, mbv)
    for wv in np.unique(flat_w[flat_present]):
        wv = int(wv)
        if wv == 0:
            continue
        sel = np.flatnonzero(flat_present & (flat_w == wv))
        packed = pack_bits(flat_vals[sel].reshape(-1), wv).reshape(sel.size, mbv * wv // 8)
        idx = flat_off[sel][:, None] + np.arange(mbv * wv // 8)[None, :]
        out[idx.reshape(-1)] = packed.reshape(-1)
    return np.concatenate([head, out])


# ---------------------------------------------------------------------------
# DELTA_LENGTH_BYTE_ARRAY

def delta_length_ba_encode(values, block=128, miniblocks=4):
    """The reference ByteArrayDeltaLengthEncoder's stream (types/byte-array.go:120-164): the
    lengths as an INT32 DELTA_BINARY_PACKED stream, then the values back to back."""
    lens = np.array([len(v) for v in values], dtype=np.int32)
    body = np.frombuffer(b"".join(bytes(v) for v in values), dtype=np.uint8)
    return np.concatenate([delta_encode(lens, 32, block, miniblocks), body]).astype(np.uint8)


def common_prefix_len(a, b):
    """types/helpers.go:58-71: length of the longest common prefix."""
    n = min(len(a), len(b))
    if n == 0:
        return 0
    x = np.frombuffer(bytes(a[:n]), dtype=np.uint8) != np.frombuffer(bytes(b[:n]), dtype=np.uint8)
    return int(np.argmax(x)) if x.any() else n


def delta_ba_encode(values, block=128, miniblocks=4):
    """The reference ByteArrayDeltaEncoder's stream (types/byte-array.go:225-275): the prefix lengths
    against the value in front as an INT32 DELTA_BINARY_PACKED stream, then the suffixes as a
    DELTA_LENGTH_BYTE_ARRAY stream."""
    prev = b""
    plens, suffixes = [], []
    for v in values:
        v = bytes(v)
        k = common_prefix_len(prev, v)
        plens.append(k)
        suffixes.append(v[k:])
        prev = v
    return np.concatenate([delta_encode(np.array(plens, dtype=np.int32), 32, block, miniblocks),
                           delta_length_ba_encode(suffixes, block, miniblocks)]).astype(np.uint8)


# ---------------------------------------------------------------------------
# Thrift compact protocol: page headers (parquet/parquet.thrift PageHeader and its members)

def _zz(v):
    return (v << 1) ^ (v >> 63) if v >= 0 else ((-v) << 1) - 1


def thrift_value(kind, value):
    """Bytes of one value.  kind: "i32" / "i64" / "i16" / "byte" / "double" / "binary" / "struct" (value = field
    list) / ("list", elem kind) / ("map", key kind, value kind)."""
    if kind in ("i16", "i32", "i64"):
        return bytes(uleb128(_zz(int(value))))
    if kind == "byte":
        return bytes([value & 0xFF])
    if kind == "double":
        return np.array([value], dtype="<f8").tobytes()
    if kind == "binary":
        return bytes(uleb128(len(value))) + bytes(value)
    if kind == "struct":
        return thrift_struct(value)
    if kind == "bool":                      # inside a collection
        return bytes([1 if value else 2])
    if kind[0] in ("list", "set"):
        n, et = len(value), _CT[kind[1] if isinstance(kind[1], str) else kind[1][0]]
        head = bytes([(n << 4) | et]) if n < 15 else bytes([0xF0 | et]) + bytes(uleb128(n))
        return head + b"".join(thrift_value(kind[1], v) for v in value)
    if kind[0] == "map":
        if not value:
            return b"\x00"
        kt, vt = _CT[kind[1]], _CT[kind[2]]
        return bytes(uleb128(len(value))) + bytes([(kt << 4) | vt]) + b"".join(
            thrift_value(kind[1], k) + thrift_value(kind[2], v) for k, v in value)
    raise ValueError(kind)


_CT = {"bool": 1, "byte": 3, "i16": 4, "i32": 5, "i64": 6, "double": 7, "binary": 8, "list": 9, "set": 10, "map": 11,
       "struct": 12}


def thrift_struct(fields):
    """fields: (id, kind, value) in any order of ids; short-form headers where the id delta allows."""
    out, last = bytearray(), 0
    for fid, kind, value in fields:
        if kind == "bool":
            ct = 1 if value else 2
        else:
            ct = _CT[kind if isinstance(kind, str) else kind[0]]
        delta = fid - last
        if 0 < delta <= 15:
            out.append((delta << 4) | ct)
        else:
            out.append(ct)
            out += bytes(uleb128(_zz(fid)))
        last = fid
        if kind != "bool":
            out += thrift_value(kind, value)
    out.append(0)
    return bytes(out)


def statistics(rng=None):
    """A Statistics struct (binary min / max, counts), which the walk has to step over."""
    rng = rng or np.random.default_rng(0)
    return [(1, "binary", bytes(rng.integers(0, 256, size=int(rng.integers(0, 40)), dtype=np.uint8))),
            (2, "binary", b"ab"), (3, "i64", int(rng.integers(0, 1000))), (4, "i64", 77),
            (5, "binary", bytes(rng.integers(0, 256, size=int(rng.integers(0, 200)), dtype=np.uint8))),
            (6, "binary", b""), (7, "bool", True), (8, "bool", False)]


def page_header(page_type, uncompressed, compressed, data=None, dictionary=None, v2=None, crc=None, extra=()):
    """PageHeader bytes.  data = (num_values, encoding, def_encoding, rep_encoding[, statistics fields]);
    dictionary = (num_values, encoding[, is_sorted]); v2 = (num_values, num_nulls, num_rows, encoding, def_len,
    rep_len[, is_compressed[, statistics fields]]); extra: further (id, kind, value) fields."""
    f = [(1, "i32", page_type), (2, "i32", uncompressed), (3, "i32", compressed)]
    if crc is not None:
        f.append((4, "i32", crc))
    if data is not None:
        sub = [(1, "i32", data[0]), (2, "i32", data[1]), (3, "i32", data[2]), (4, "i32", data[3])]
        if len(data) > 4:
            sub.append((5, "struct", data[4]))
        f.append((5, "struct", sub))
    if dictionary is not None:
        sub = [(1, "i32", dictionary[0]), (2, "i32", dictionary[1])]
        if len(dictionary) > 2:
            sub.append((3, "bool", dictionary[2]))
        f.append((7, "struct", sub))
    if v2 is not None:
        sub = [(k + 1, "i32", v2[k]) for k in range(6)]
        if len(v2) > 6 and v2[6] is not None:
            sub.append((7, "bool", v2[6]))
        if len(v2) > 7:
            sub.append((8, "struct", v2[7]))
        f.append((8, "struct", sub))
    f += list(extra)
    return thrift_struct(f)
