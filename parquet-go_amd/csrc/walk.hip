// walk.hip — page-header walk of column chunks, one wave64 per chunk.
//
// Reference: ChunkReader.readPages (layout/chunk.go:129-217): read a Thrift compact PageHeader
// (readThrift, layout/helpers.go:16-23; PageHeader.Read and the nested readers of
// parquet/parquet.go:3994, :4334, :4586, :5886), apply the header checks of the page reader's
// read() (layout/page-v1.go:40-67, page-v2.go:42-108, page-dict.go:28-57), step over the page
// block (layout/block.go:17-47), repeat until TotalCompressedSize bytes are accounted for.
//
// The chain is sequential inside a chunk (a header says where the next one starts), so the
// parallelism is across the chunks of a row group.  A wave keeps a 1 KiB window of its chunk in
// LDS, filled with one coalesced 16-byte load per lane: a header costs one trip to memory, not
// one per byte.  Every lane parses the same bytes (the control flow is wave-uniform); lane 0
// writes the records.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pq_device.hpp"
#include "delta_wave.hpp"
#include "pq_launch.hpp"

namespace pq {

namespace {

constexpr int kWkWin = 1024;      // window bytes: 64 lanes x 16
constexpr int kWkWaves = 4;       // chunks per workgroup
constexpr int kWkDepth = 64;      // thrift.DEFAULT_RECURSION_DEPTH

// compact-protocol wire types
enum : uint32_t { CT_STOP = 0, CT_TRUE = 1, CT_FALSE = 2, CT_BYTE = 3, CT_I16 = 4, CT_I32 = 5, CT_I64 = 6,
                  CT_DOUBLE = 7, CT_BINARY = 8, CT_LIST = 9, CT_SET = 10, CT_MAP = 11, CT_STRUCT = 12 };

struct WalkReader {
    uint32_t *lds;      // kWkWin bytes
    uint32_t *frames;   // 2 * kWkDepth words: the skip stack
    const uint8_t *S;   // chunk byte 0
    int64_t len;
    int64_t c0;         // stream offset of the window (far below zero: nothing staged)
    int64_t next;       // stream offset of the byte behind the register buffer
    uint64_t buf;       // the next `have` bytes of the stream, lowest byte first
    uint32_t have;      // 0..8; the read position is next - have
    int lane;
};

constexpr int64_t kWkNone = INT64_MIN / 2;
constexpr int64_t kWkHeaderRoom = 512;  // the fast path takes a header from a window that holds this much behind its
                                        // first byte: one that ran out of window would go the careful way, byte by byte

__device__ __forceinline__ int64_t wk_pos(const WalkReader &r) { return r.next - (int64_t)r.have; }

// Move the read position (the register buffer is dropped).
__device__ __forceinline__ void wk_seek(WalkReader &r, int64_t pos)
{
    r.next = pos;
    r.have = 0;
}

__device__ __forceinline__ uint4 wk_load_piece(const WalkReader &r, int64_t c0)
{
    const int64_t o = c0 + 16 * (int64_t)r.lane;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (o + 16 > 0 && o < r.len) v = *reinterpret_cast<const uint4 *>(r.S + o);
    return v;
}

__device__ __forceinline__ void wk_stage(WalkReader &r, int64_t pos)
{
    r.c0 = chunk_base(r.S, pos);
    const uint4 v = wk_load_piece(r, r.c0);
    wave_sync();
    reinterpret_cast<uint4 *>(r.lds)[r.lane] = v;
    wave_sync();
}

// Up to eight stream bytes from the read position on into the register buffer: a header is
// parsed from registers, one trip to LDS per eight bytes instead of one per byte, and all the
// 64-bit position arithmetic happens here, not per byte.  false: the source is exhausted.
__device__ __forceinline__ bool wk_refill(WalkReader &r)
{
    const int64_t pos = r.next;  // have == 0
    if (pos >= r.len) return false;
    if (pos < r.c0 || pos + 8 > r.c0 + kWkWin) wk_stage(r, pos);  // a window starts at most 15 bytes below pos
    const uint32_t o = (uint32_t)(pos - r.c0);
    const uint32_t *w = r.lds + (o >> 2);
    const uint32_t sh = (o & 3u) * 8u;
    // every lane holds the same bytes: saying so keeps the whole parse on the scalar unit
    const uint32_t lo = uni(__builtin_amdgcn_alignbit(w[1], w[0], sh));
    const uint32_t hi = uni(__builtin_amdgcn_alignbit(w[2], w[1], sh));
    r.buf = ((uint64_t)hi << 32) | lo;
    const int64_t left = r.len - pos;
    r.have = left < 8 ? (uint32_t)left : 8u;
    r.next = pos + (int64_t)r.have;
    return true;
}

// ReadByte: false when the source is exhausted
__device__ __forceinline__ bool wk_byte(WalkReader &r, uint32_t *b)
{
    if (r.have == 0u && !wk_refill(r)) return false;
    *b = (uint32_t)r.buf & 255u;
    r.buf >>= 8;
    r.have -= 1u;
    return true;
}

// io.ReadFull of n bytes nobody looks at
__device__ __forceinline__ bool wk_advance(WalkReader &r, uint64_t n)
{
    if (n <= (uint64_t)r.have) {  // still inside the register buffer
        r.buf = n < 8u ? r.buf >> (8u * (uint32_t)n) : 0ull;
        r.have -= (uint32_t)n;
        return true;
    }
    const int64_t pos = wk_pos(r);
    if ((uint64_t)(r.len - pos) < n) {
        wk_seek(r, r.len);
        return false;
    }
    wk_seek(r, pos + (int64_t)n);
    return true;
}

// readVarint64: up to the first byte without the continuation bit, bits past 64 dropped
__device__ __forceinline__ bool wk_varint(WalkReader &r, uint64_t *out)
{
    if (r.have != 0u && ((uint32_t)r.buf & 0x80u) == 0u) {  // one byte, already in the register buffer: the usual case
        *out = (uint32_t)r.buf & 0x7fu;
        r.buf >>= 8;
        r.have -= 1u;
        return true;
    }
    uint64_t v = 0;
    uint32_t shift = 0, b;
    while (true) {
        if (!wk_byte(r, &b)) return false;
        if (shift < 64u) v |= (uint64_t)(b & 0x7fu) << shift;
        if ((b & 0x80u) == 0) break;
        shift += 7u;
    }
    *out = v;
    return true;
}

__device__ __forceinline__ bool wk_i32(WalkReader &r, int32_t *out)
{
    uint64_t u;
    if (!wk_varint(r, &u)) return false;
    const uint32_t x = (uint32_t)u;
    *out = (int32_t)((x >> 1) ^ (0u - (x & 1u)));
    return true;
}

// ReadFieldBegin.  type CT_STOP ends the struct.
__device__ __forceinline__ bool wk_field(WalkReader &r, uint32_t *type, int32_t *id, int32_t *last)
{
    uint32_t b;
    if (!wk_byte(r, &b)) return false;
    *type = b & 15u;
    if (*type == CT_STOP) return true;
    if (*type > CT_STRUCT) return false;
    const uint32_t delta = b >> 4;
    if (delta == 0) {
        int32_t v;
        if (!wk_i32(r, &v)) return false;
        *id = (int32_t)(int16_t)v;
    } else {
        *id = (int32_t)(int16_t)(*last + (int32_t)delta);
    }
    *last = *id;
    return true;
}

// A value without children.  in_collection: a boolean is a byte there, nothing in a field.
__device__ __forceinline__ bool wk_skip_scalar(WalkReader &r, uint32_t type, bool in_collection)
{
    uint64_t u;
    switch (type) {
    case CT_TRUE:
    case CT_FALSE: return in_collection ? wk_advance(r, 1) : true;
    case CT_BYTE: return wk_advance(r, 1);
    case CT_I16:
    case CT_I32:
    case CT_I64: return wk_varint(r, &u);
    case CT_DOUBLE: return wk_advance(r, 8);
    case CT_BINARY: {
        if (!wk_varint(r, &u)) return false;
        const int32_t n = (int32_t)(uint32_t)u;
        if (n < 0) return false;
        return wk_advance(r, (uint64_t)n);
    }
    default: return true;  // STOP
    }
}

// thrift.Skip without recursion: frames[2 k] = kind | key type << 8 | value type << 16,
// frames[2 k + 1] = elements left (collections) or the last field id (structs).
__device__ __forceinline__ bool wk_skip(WalkReader &r, uint32_t type)
{
    enum : uint32_t { F_STRUCT = 0, F_LIST = 1, F_MAP = 2 };
    uint32_t sp = 0;
    bool in_coll = false;
    while (true) {
        if (sp >= (uint32_t)kWkDepth) return false;  // "Depth limit exceeded"
        if (type == CT_STRUCT) {
            r.frames[2 * sp] = F_STRUCT;
            r.frames[2 * sp + 1] = 0;
            ++sp;
        } else if (type == CT_LIST || type == CT_SET) {
            uint32_t b;
            uint64_t u;
            if (!wk_byte(r, &b)) return false;
            int32_t n = (int32_t)(b >> 4);
            if (n == 15) {
                if (!wk_varint(r, &u)) return false;
                n = (int32_t)(uint32_t)u;
                if (n < 0) return false;
            }
            if ((b & 15u) > CT_STRUCT) return false;
            r.frames[2 * sp] = F_LIST | (b & 15u) << 8;
            r.frames[2 * sp + 1] = (uint32_t)n;
            ++sp;
        } else if (type == CT_MAP) {
            uint64_t u;
            uint32_t kv = 0;
            if (!wk_varint(r, &u)) return false;
            const int32_t n = (int32_t)(uint32_t)u;
            if (n < 0) return false;
            if (n != 0 && !wk_byte(r, &kv)) return false;
            if ((kv >> 4) > CT_STRUCT || (kv & 15u) > CT_STRUCT) return false;
            r.frames[2 * sp] = F_MAP | (kv >> 4) << 8 | (kv & 15u) << 16;
            r.frames[2 * sp + 1] = 2u * (uint32_t)n;  // n < 2^31: keys and values, counted together
            ++sp;
        } else if (!wk_skip_scalar(r, type, in_coll)) {
            return false;
        }
        // the next value to step over
        while (true) {
            if (sp == 0) return true;
            const uint32_t f = uni(r.frames[2 * (sp - 1)]);
            const uint32_t x = uni(r.frames[2 * (sp - 1) + 1]);
            if ((f & 255u) == F_STRUCT) {
                int32_t id = 0, last = (int32_t)x;
                if (!wk_field(r, &type, &id, &last)) return false;
                if (type == CT_STOP) {
                    --sp;
                    continue;
                }
                r.frames[2 * (sp - 1) + 1] = (uint32_t)last;
                in_coll = false;
                break;
            }
            if (x == 0) {
                --sp;
                continue;
            }
            r.frames[2 * (sp - 1) + 1] = x - 1u;
            type = (f & 255u) == F_LIST ? (f >> 8) & 255u : ((x & 1u) == 0 ? (f >> 8) & 255u : (f >> 16) & 255u);
            in_coll = true;
            break;
        }
    }
}

// DataPageHeader, IndexPageHeader, DictionaryPageHeader, DataPageHeaderV2
struct SubHeader {
    int32_t f[8];
    uint32_t isset;
    uint32_t present;
    uint32_t is_compressed;
};

__device__ __forceinline__ bool wk_sub(WalkReader &r, SubHeader &h, uint32_t i32_mask, uint32_t required,
                                       int32_t bool_id)
{
#pragma unroll
    for (int k = 0; k < 8; ++k) h.f[k] = 0;
    h.isset = 0;
    h.present = 1;
    h.is_compressed = 1;
    int32_t last = 0, id = 0;
    while (true) {
        uint32_t t;
        if (!wk_field(r, &t, &id, &last)) return false;
        if (t == CT_STOP) break;
        if (id >= 1 && id < 8 && ((i32_mask >> id) & 1u) && t == CT_I32) {
            int32_t v;
            if (!wk_i32(r, &v)) return false;
#pragma unroll
            for (int k = 1; k < 8; ++k)
                if (k == id) h.f[k] = v;
            h.isset |= 1u << id;
        } else if (id == bool_id && (t == CT_TRUE || t == CT_FALSE)) {
            h.is_compressed = t == CT_TRUE ? 1u : 0u;
        } else if (!wk_skip(r, t)) {
            return false;
        }
    }
    return (h.isset & required) == required;
}

struct PageHeader {
    int32_t type, uncompressed, compressed;
    SubHeader data, dict, v2;
};

__device__ __forceinline__ void wk_clear(SubHeader &h)
{
#pragma unroll
    for (int k = 0; k < 8; ++k) h.f[k] = 0;
    h.isset = 0;
    h.present = 0;
    h.is_compressed = 1;
}

__device__ __forceinline__ bool wk_page_header(WalkReader &r, PageHeader &h)
{
    h.type = h.uncompressed = h.compressed = 0;
    wk_clear(h.data);
    wk_clear(h.dict);
    wk_clear(h.v2);
    uint32_t isset = 0;
    int32_t last = 0, id = 0;
    while (true) {
        uint32_t t;
        if (!wk_field(r, &t, &id, &last)) return false;
        if (t == CT_STOP) break;
        bool ok;
        if (id >= 1 && id <= 4 && t == CT_I32) {
            int32_t v = 0;
            ok = wk_i32(r, &v);
            if (id == 1) h.type = v;
            if (id == 2) h.uncompressed = v;
            if (id == 3) h.compressed = v;
            isset |= 1u << id;
        } else if (id == 5 && t == CT_STRUCT) {
            ok = wk_sub(r, h.data, 0x1eu, 0x1eu, 0);
        } else if (id == 6 && t == CT_STRUCT) {
            SubHeader idx;
            ok = wk_sub(r, idx, 0u, 0u, 0);
        } else if (id == 7 && t == CT_STRUCT) {
            ok = wk_sub(r, h.dict, 0x06u, 0x06u, 0);
        } else if (id == 8 && t == CT_STRUCT) {
            ok = wk_sub(r, h.v2, 0x7eu, 0x7eu, 7);
        } else {
            ok = wk_skip(r, t);
        }
        if (!ok) return false;
    }
    return (isset & 0x0eu) == 0x0eu;
}

// ---- the header shapes every writer emits, as straight-line code ------------------------------
//
// A single wave pays for every taken branch, and the general reader above is nothing but
// branches.  Real headers are far more regular than the protocol allows: fields in id order with
// one-byte field headers, i32 values of at most four varint bytes, and at the end of the page-
// type header an optional Statistics struct of binaries, i64 counts and booleans.  The fast path takes
// exactly those (anything else — long-form field ids, unknown fields, collections, a crc that
// needs five varint bytes, a header that leaves the staged window — makes it step back, having
// changed nothing, and the general reader parses the header from its first byte).  What it
// accepts it decodes to the same PageHeader the general reader would.
//
// Bytes come from a VGPR that holds 256 bytes of the LDS window, one dword per lane: eight bytes
// at any offset are three v_readlane and two shifts, with no trip to LDS and nothing to wait for.
struct FastWin {
    uint32_t v;     // lane k: dword k of the window from byte offset `base` on
    uint32_t base;  // a multiple of 4
};

__device__ __forceinline__ void fw_load(FastWin &f, const WalkReader &r, uint32_t o)
{
    f.base = o & ~3u;
    f.v = r.lds[(f.base >> 2) + (uint32_t)r.lane];  // the window array has 64 words of slack behind it
}

// eight window bytes from offset o on (lowest first)
__device__ __forceinline__ uint64_t fw_get(FastWin &f, const WalkReader &r, uint32_t o)
{
    if (o < f.base || o + 12u > f.base + 256u) fw_load(f, r, o);
    const uint32_t i = (o - f.base) >> 2, sh = (o & 3u) * 8u;
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)f.v, (int)i);
    const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)f.v, (int)(i + 1u));
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)f.v, (int)(i + 2u));
    const uint32_t lo = (uint32_t)(((((uint64_t)b) << 32) | a) >> sh);
    const uint32_t hi = (uint32_t)(((((uint64_t)c) << 32) | b) >> sh);
    return ((uint64_t)hi << 32) | lo;
}

// A varint of 1..4 bytes in the low bytes of x: its value and length (0: it is longer).
__device__ __forceinline__ uint32_t fw_varint(uint32_t x, uint32_t *value)
{
    const uint32_t stop = ~x & 0x80808080u;
    if (stop == 0u) return 0u;
    const uint32_t vl = ((uint32_t)__builtin_ctz(stop) + 1u) >> 3;
    x &= 0xffffffffu >> (32u - 8u * vl);
    *value = (x & 0x7fu) | ((x >> 1) & 0x3f80u) | ((x >> 2) & 0x1fc000u) | ((x >> 3) & 0xfe00000u);
    return vl;
}

// The i32 fields of a header (field header byte 0x15, zig-zag varint) all at once: every lane decodes
// the field that WOULD start at its byte of the 64 window bytes from offset `o` on, and the scalar
// side follows the chain from field to field - a mask test and two v_readlane each instead of
// eight bytes out of the window register and a varint decode (~ 45 scalar instructions).
struct I32Span {
    uint32_t byte;   // the window byte at o + lane
    uint32_t value;  // the i32 behind it, were it a field header
    uint32_t behind; // the lane behind that field
    uint64_t good;   // lanes whose byte is 0x15 with a varint of at most four bytes behind it
    uint32_t p;      // the lane the scalar side stands at
};

__device__ __forceinline__ void span_load(I32Span &sp, const WalkReader &r, uint32_t o)
{
    const uint32_t q = o + (uint32_t)r.lane;  // the window array has 64 words of slack behind it
    const uint32_t *w = r.lds + (q >> 2);
    const uint32_t sh = (q & 3u) * 8u;
    const uint32_t lo = __builtin_amdgcn_alignbit(w[1], w[0], sh), hi = __builtin_amdgcn_alignbit(w[2], w[1], sh);
    uint32_t u = 0;
    const uint32_t vl = fw_varint((lo >> 8) | (hi << 24), &u);
    sp.byte = lo & 255u;
    sp.value = (u >> 1) ^ (0u - (u & 1u));
    sp.behind = (uint32_t)r.lane + 1u + vl;
    sp.good = __builtin_amdgcn_ballot_w64(sp.byte == 0x15u && vl != 0u);
    sp.p = 0;
}

// the byte the scalar side stands at (p < 64)
__device__ __forceinline__ uint32_t span_byte(const I32Span &sp)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)sp.byte, (int)sp.p);
}

// An i32 field at p: false when the bytes say otherwise (or the field might leave the 64 bytes).
__device__ __forceinline__ bool span_i32(I32Span &sp, int32_t *out)
{
    if (sp.p > 58u || ((sp.good >> sp.p) & 1ull) == 0ull) return false;
    *out = (int32_t)__builtin_amdgcn_readlane((int)sp.value, (int)sp.p);
    sp.p = (uint32_t)__builtin_amdgcn_readlane((int)sp.behind, (int)sp.p);
    return true;
}

// `count` i32 fields with ids 1, 2, 3 ... into h.f[1 ..]
__device__ __forceinline__ bool span_i32_run(I32Span &sp, SubHeader &h, uint32_t count)
{
    bool ok = true;
#pragma unroll
    for (uint32_t k = 1; k <= 6u; ++k) {
        if (k <= count) {
            int32_t v = 0;
            ok = ok && span_i32(sp, &v);
            h.f[k] = v;
        }
    }
    h.isset = (2u << count) - 2u;
    h.present = 1;
    return ok;
}

// An optional Statistics struct (field header `hdr`) and the stop byte of the struct around it.
// Binaries, i64 counts and booleans with one-byte field headers only; `lim` = window bytes inside the stream.
//
// Stepping over the fields one after the other on the scalar unit cost ~ 85 instructions a field
// (eight bytes out of the window register, a varint, the type rules).  Instead every lane looks
// at the field that WOULD start at its byte of the window - header byte, varint, type rules, and
// with them the lane the next field starts at - and the scalar side only follows that chain:
// two mask tests and one v_readlane a field.  A binary longer than the 64 bytes the lanes cover
// ends the span; the next one starts where it ends.
__device__ __forceinline__ bool fw_statistics_and_stop(FastWin &f, const WalkReader &r, uint32_t &o, uint32_t hdr,
                                                       uint32_t lim)
{
    uint64_t g = fw_get(f, r, o);
    if (((uint32_t)g & 255u) == hdr) {
        o += 1u;
        uint32_t fields = 0;  // Statistics has eight: four binaries, two counts, two booleans
        bool at_stop = false;
        while (!at_stop) {
            if (o + 12u > lim) return false;
            const uint32_t q = o + (uint32_t)r.lane;  // the window array has 64 words of slack behind it
            const uint32_t *w = r.lds + (q >> 2);
            const uint32_t sh = (q & 3u) * 8u;
            const uint32_t lo = __builtin_amdgcn_alignbit(w[1], w[0], sh), hi = __builtin_amdgcn_alignbit(w[2], w[1], sh);
            const uint32_t b = lo & 255u, t = b & 15u;
            uint32_t u = 0;
            const uint32_t vl = fw_varint((lo >> 8) | (hi << 24), &u);
            const bool flag = t == CT_TRUE || t == CT_FALSE;  // a boolean field is its header byte
            const bool good = (b >> 4) != 0u && (flag || (vl != 0u && (t == CT_BINARY || t == CT_I64)));
            const uint32_t behind = (uint32_t)r.lane + (flag ? 1u : 1u + vl + (t == CT_BINARY ? u : 0u));
            const uint64_t stops = __builtin_amdgcn_ballot_w64(b == 0u), goods = __builtin_amdgcn_ballot_w64(good);
            uint32_t p = 0;
            while (true) {
                if (o + p + 12u > lim) return false;
                if (((stops >> p) & 1ull) != 0ull) {
                    at_stop = true;
                    break;
                }
                if (((goods >> p) & 1ull) == 0ull || ++fields > 10u) return false;
                p = (uint32_t)__builtin_amdgcn_readlane((int)behind, (int)p);
                if (p >= (uint32_t)kWave) break;
            }
            o += p;
        }
        o += 1u;  // the stop byte of Statistics
        g = fw_get(f, r, o);
    }
    if (((uint32_t)g & 255u) != 0u) return false;
    o += 1u;
    return true;
}

// true: the header at the read position was one of the usual shapes; h is filled and the reader
// stands behind it.  false: nothing changed.
__device__ __forceinline__ bool wk_fast_header(WalkReader &r, PageHeader &h)
{
    const int64_t pos = wk_pos(r);
    if (pos >= r.len) return false;
    if (pos < r.c0 || pos + kWkHeaderRoom > r.c0 + kWkWin) wk_stage(r, pos);
    const int64_t left = r.len - r.c0;
    const uint32_t lim = left < (int64_t)kWkWin ? (uint32_t)left : (uint32_t)kWkWin;  // staged bytes inside the stream
    uint32_t o = (uint32_t)(pos - r.c0);
    if (o + 64u > lim) return false;  // a header this close to the end goes the careful way
    FastWin f;
    fw_load(f, r, o);
    I32Span sp;
    span_load(sp, r, o);  // o + 64 <= lim: the 64 bytes are the stream's
    const uint32_t o0 = o;
    h.type = h.uncompressed = h.compressed = 0;
    wk_clear(h.data);
    wk_clear(h.dict);
    wk_clear(h.v2);
    if (!span_i32(sp, &h.type) || !span_i32(sp, &h.uncompressed) || !span_i32(sp, &h.compressed)) return false;
    if (sp.p > 58u) return false;
    const uint32_t b = span_byte(sp);
    sp.p += 1u;
    bool ok;
    if (b == 0x2cu) {  // field 5: DataPageHeader
        ok = span_i32_run(sp, h.data, 4u);
        o = o0 + sp.p;
        ok = ok && fw_statistics_and_stop(f, r, o, 0x1cu, lim);
    } else if (b == 0x4cu) {  // field 7: DictionaryPageHeader, is_sorted optional
        ok = span_i32_run(sp, h.dict, 2u) && sp.p <= 58u;
        if (!ok) return false;
        const uint32_t s = span_byte(sp);
        if (s == 0x11u || s == 0x12u) sp.p += 1u;
        ok = span_byte(sp) == 0u;
        sp.p += 1u;
        o = o0 + sp.p;
    } else if (b == 0x5cu) {  // field 8: DataPageHeaderV2, is_compressed optional
        ok = span_i32_run(sp, h.v2, 6u) && sp.p <= 58u;
        if (!ok) return false;
        const uint32_t s = span_byte(sp);
        uint32_t stats_hdr = 0x2cu;  // Statistics is field 8: two past field 6 ...
        if (s == 0x11u || s == 0x12u) {
            h.v2.is_compressed = s == 0x11u ? 1u : 0u;
            sp.p += 1u;
            stats_hdr = 0x1cu;  // ... one past field 7
        }
        o = o0 + sp.p;
        ok = fw_statistics_and_stop(f, r, o, stats_hdr, lim);
    } else {
        return false;
    }
    if (!ok || o + 1u > lim || ((uint32_t)fw_get(f, r, o) & 255u) != 0u) return false;  // the PageHeader's stop
    o += 1u;
    wk_seek(r, r.c0 + (int64_t)o);
    return true;
}

// getValuesDecoder (layout/page-dict.go:143-345)
__device__ __forceinline__ bool wk_decoder_exists(uint32_t type, int32_t type_length, int32_t enc)
{
    if (enc == 2) enc = 8;
    const bool plain = enc == 0, dict = enc == 8;
    switch (type) {
    case 0: return plain || dict || enc == 3;
    case 1:
    case 2: return plain || dict || enc == 5;
    case 3:
    case 4:
    case 5: return plain || dict;
    case 6: return plain || dict || enc == 6 || enc == 7;
    case 7: return type_length >= 0 && (plain || dict || enc == 7);
    default: return false;
    }
}

// readBlockData (layout/block.go:17-47)
__device__ __forceinline__ int32_t wk_block(WalkReader &r, uint32_t codec, int32_t compressed, int32_t uncompressed)
{
    const int64_t pos = wk_pos(r);
    const uint64_t left = (uint64_t)(r.len - pos);
    const uint64_t want = compressed > 0 ? (uint64_t)compressed : 0ull;  // io.LimitReader(in, negative) reads nothing
    const uint64_t got = want < left ? want : left;
    wk_seek(r, pos + (int64_t)got);
    if ((int64_t)got != (int64_t)compressed) return PQ_WALK_BLOCK_SIZE;
    if (codec == 0 && compressed != uncompressed) return PQ_WALK_UNCOMPRESSED_SIZE;
    return PQ_OK;
}

__global__ __launch_bounds__(kWkWaves *kWave) void walk_chunks_kernel(const pq_chunk_desc *__restrict__ chunks,
                                                                     uint32_t n_chunks, const uint8_t *__restrict__ src,
                                                                     pq_page_info *__restrict__ infos,
                                                                     pq_chunk_result *__restrict__ results)
{
    // rows are written with 16-byte LDS stores: the array and every row start on 16-byte boundaries
    static_assert((kWkWin / 4 + 64) % 4 == 0, "a window row must be a whole number of 16-byte pieces");
    __shared__ __attribute__((aligned(16))) uint32_t s_win[kWkWaves][kWkWin / 4 + 64];  // slack: the fast path loads 64 dwords from any window offset
    __shared__ __attribute__((aligned(16))) uint32_t s_frames[kWkWaves][2 * kWkDepth];
    const int wave = (int)uni(threadIdx.x / kWave), lane = (int)(threadIdx.x % kWave);
    const uint32_t ci = blockIdx.x * kWkWaves + (uint32_t)wave;
    if (ci >= n_chunks) return;
    const pq_chunk_desc c = chunks[ci];

    WalkReader r;
    r.lds = s_win[wave];
    r.frames = s_frames[wave];
    r.S = src + c.src_off;
    r.len = (int64_t)c.src_len;
    r.c0 = kWkNone;
    r.next = 0;
    r.buf = 0;
    r.have = 0;
    r.lane = lane;

    const int64_t start = c.has_dictionary_page_offset ? c.dictionary_page_offset : c.data_page_offset;
    int64_t count = 0;
    uint32_t n = 0;
    bool have_dict = false;
    int32_t st = PQ_OK;
    pq_page_info *out = infos + c.info_first;
    while (true) {
        if (c.total_compressed_size - count < 1) break;  // chunk.go:136
        const int64_t h0 = wk_pos(r);
        PageHeader h;
        if (!wk_fast_header(r, h) && !wk_page_header(r, h)) {
            st = PQ_WALK_HEADER_FAILED;
            count += wk_pos(r) - h0;
            break;
        }
        if (n >= c.info_cap) {
            st = PQ_ERR_OUT_OF_RANGE;
            break;
        }
        pq_page_info pi;
        pi.type = h.type;
        pi.encoding = 0;
        pi.num_values = 0;
        pi.compressed_size = h.compressed;
        pi.uncompressed_size = h.uncompressed;
        pi.def_len = pi.rep_len = 0;
        pi.def_encoding = pi.rep_encoding = 0;
        pi.is_compressed = 1;
        pi.header_off = (uint64_t)h0;
        pi.data_off = (uint64_t)wk_pos(r);
        bool seek = false;
        if (h.type == 2) {  // DICTIONARY_PAGE (chunk.go:148-180, page-dict.go:28-57)
            if (have_dict)
                st = PQ_WALK_SECOND_DICTIONARY;
            else if (c.physical_type == 0 || c.physical_type > 7 || (c.physical_type == 7 && c.type_length < 0))
                st = PQ_WALK_VALUE_ENCODING;  // getDictValuesDecoder
            else if (!h.dict.present)
                st = PQ_WALK_MISSING_HEADER;
            else if (h.dict.f[1] < 0)
                st = PQ_WALK_NEGATIVE_NUM_VALUES;
            else if (h.dict.f[2] != 0 && h.dict.f[2] != 2)
                st = PQ_WALK_DICT_ENCODING;
            pi.num_values = h.dict.f[1];
            pi.encoding = h.dict.f[2];
            if (st == PQ_OK) st = wk_block(r, c.codec, h.compressed, h.uncompressed);
            have_dict = true;
            seek = st == PQ_OK && c.has_dictionary_page_offset && c.dictionary_page_offset != start + wk_pos(r);
        } else if (h.type == 0) {  // DATA_PAGE: init (page-v1.go:21-38), read (:40-67)
            if (!h.data.present)
                st = PQ_WALK_MISSING_HEADER;
            else if (c.max_rep != 0 && (h.data.f[3] != 3 || h.data.f[4] != 3))
                st = PQ_WALK_LEVEL_ENCODING;
            else if (h.data.f[1] < 0)
                st = PQ_WALK_NEGATIVE_NUM_VALUES;
            pi.num_values = h.data.f[1];
            pi.encoding = h.data.f[2];
            pi.def_encoding = h.data.f[3];
            pi.rep_encoding = h.data.f[4];
            if (st == PQ_OK) st = wk_block(r, c.codec, h.compressed, h.uncompressed);
            if (st == PQ_OK && !wk_decoder_exists(c.physical_type, c.type_length, h.data.f[2])) st = PQ_WALK_VALUE_ENCODING;
        } else if (h.type == 3) {  // DATA_PAGE_V2 (page-v2.go:42-108)
            if (!h.v2.present)
                st = PQ_WALK_MISSING_HEADER;
            else if (h.v2.f[1] < 0)
                st = PQ_WALK_NEGATIVE_NUM_VALUES;
            else if (h.v2.f[6] < 0 || h.v2.f[5] < 0)
                st = PQ_WALK_NEGATIVE_LEVELS_LENGTH;
            else if (!wk_decoder_exists(c.physical_type, c.type_length, h.v2.f[4]))
                st = PQ_WALK_VALUE_ENCODING;
            pi.num_values = h.v2.f[1];
            pi.encoding = h.v2.f[4];
            pi.def_len = h.v2.f[5];
            pi.rep_len = h.v2.f[6];
            pi.is_compressed = h.v2.present ? h.v2.is_compressed : 1u;
            if (st == PQ_OK) {
                const int32_t levels = (int32_t)((uint32_t)h.v2.f[5] + (uint32_t)h.v2.f[6]);
                if (levels > 0 && !wk_advance(r, (uint64_t)levels)) st = PQ_WALK_BLOCK_SIZE;  // io.ReadFull (:87)
                if (st == PQ_OK)
                    st = wk_block(r, c.codec, (int32_t)((uint32_t)h.compressed - (uint32_t)levels),
                                  (int32_t)((uint32_t)h.uncompressed - (uint32_t)levels));
            }
        } else {
            st = PQ_WALK_PAGE_TYPE;
        }
        count += wk_pos(r) - h0;
        if (st != PQ_OK) break;
        if (lane == 0) out[n] = pi;
        ++n;
        if (seek) {  // Seek(DataPageOffset): Count() moves by the distance (layout/helpers.go:52-60)
            const int64_t target = c.data_page_offset - start;
            if (target < 0) {
                st = PQ_WALK_SEEK;
                break;
            }
            count += target - wk_pos(r);
            wk_seek(r, target > r.len ? r.len : target);
        }
    }
    if (lane == 0) {
        pq_chunk_result res;
        res.status = st;
        res.n_pages = n;
        res.count = count;
        results[ci] = res;
    }
}

}  // namespace

hipError_t launch_walk_chunks(hipStream_t s, const pq_chunk_desc *d_chunks, uint32_t n_chunks, const void *d_src,
                              pq_page_info *d_infos, pq_chunk_result *d_results)
{
    if (n_chunks == 0) return hipSuccess;
    const uint32_t grid = (n_chunks + kWkWaves - 1) / kWkWaves;
    walk_chunks_kernel<<<grid, kWkWaves * kWave, 0, s>>>(d_chunks, n_chunks, static_cast<const uint8_t *>(d_src), d_infos,
                                                        d_results);
    return hipGetLastError();
}

}  // namespace pq
