// pq_device.hpp — device-side building blocks shared by the gfx950 decode kernels.
//
// Everything here assumes a 256-thread workgroup (4 wave64) and an LDS staging
// buffer of 32-bit words that mirrors a window of the encoded byte stream.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pqdecode.h"

namespace pq {

constexpr int kWG = 256;    // threads per workgroup
constexpr int kWave = 64;   // gfx950 wavefront

// ---- staging ---------------------------------------------------------------

// Zero the bytes of a 16-byte piece that fall outside [lo, hi) (byte positions
// relative to the piece, may be negative / beyond 16).
__device__ __forceinline__ uint4 mask_piece(uint4 v, int64_t lo, int64_t hi)
{
    uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t m = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            int64_t p = 4 * k + b;
            if (p >= lo && p < hi) m |= 0xffu << (8 * b);
        }
        d[k] &= m;
    }
    return make_uint4(d[0], d[1], d[2], d[3]);
}

// Copy stream bytes [c0, c0 + BYTES) into LDS with coalesced 16-byte loads.
// S is the global address of stream byte 0, len the stream length; S + c0 must
// be 16-byte aligned.  Bytes outside [0, len) read as zero and global memory
// outside the 16-byte lines that intersect the stream is never touched.
template <int BYTES>
__device__ __forceinline__ void stage_chunk(uint32_t *lds, const uint8_t *S, int64_t c0, int64_t len, int tid)
{
    uint4 *l4 = reinterpret_cast<uint4 *>(lds);
#pragma unroll
    for (int i = tid; i < BYTES / 16; i += kWG) {
        int64_t o = c0 + 16 * (int64_t)i;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (o + 16 > 0 && o < len) {
            v = *reinterpret_cast<const uint4 *>(S + o);
            if (o < 0 || o + 16 > len) v = mask_piece(v, -o, len - o);
        }
        l4[i] = v;
    }
}

// Stream offset (<= pos) at which a chunk must start so that S + c0 is 16-byte aligned.
__device__ __forceinline__ int64_t chunk_base(const uint8_t *S, int64_t pos)
{
    return pos - (int64_t)((reinterpret_cast<uintptr_t>(S) + (uint64_t)pos) & 15u);
}

// ---- streaming stores ---------------------------------------------------------
// Decoded values are written once and never read back by the kernel: non-temporal
// stores keep them from displacing the encoded input in L2 (measured on MI355X with
// the delta kernel's access pattern and no arithmetic: 6.7 TB/s of read + write
// traffic against 5.1 TB/s with default-policy stores).
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void store_nt16(void *p, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    u32x4_t v = {a, b, c, d};
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t *>(p));
}

__device__ __forceinline__ void store_nt8(void *p, uint32_t a, uint32_t b)
{
    u32x2_t v = {a, b};
    __builtin_nontemporal_store(v, reinterpret_cast<u32x2_t *>(p));
}

// ---- bit extraction from the LDS window --------------------------------------

// w-bit field (0 <= w <= 32) starting at bit `bit` of the little-endian LDS image.
__device__ __forceinline__ uint32_t extract32(const uint32_t *lds, uint32_t bit, uint32_t w)
{
    uint32_t i = bit >> 5;
    uint32_t lo = lds[i], hi = lds[i + 1];
    uint32_t v = __builtin_amdgcn_alignbit(hi, lo, bit & 31u);
    uint32_t m = (w >= 32u) ? 0xffffffffu : ((1u << w) - 1u);
    return v & m;
}

// w-bit field (0 <= w <= 64).
__device__ __forceinline__ uint64_t extract64(const uint32_t *lds, uint32_t bit, uint32_t w)
{
    uint32_t i = bit >> 5, s = bit & 31u;
    uint32_t d0 = lds[i], d1 = lds[i + 1], d2 = lds[i + 2];
    uint32_t lo = __builtin_amdgcn_alignbit(d1, d0, s);
    uint32_t hi = __builtin_amdgcn_alignbit(d2, d1, s);
    uint64_t v = ((uint64_t)hi << 32) | lo;
    uint64_t m = (w >= 64u) ? ~0ull : ((1ull << w) - 1ull);
    return v & m;
}

// ---- wave64 inclusive scans (DPP) ------------------------------------------
// Kogge-Stone inside each row of 16 lanes (row_shr 1,2,4,8), then row_bcast:15
// into rows 1 and 3 and row_bcast:31 into rows 2 and 3.

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp0(uint32_t x)
{
    // lanes with no valid source (or masked off) receive 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xf, false);
}

__device__ __forceinline__ uint32_t wave_scan_u32(uint32_t x)
{
    x += dpp0<0x111, 0xf>(x); // row_shr:1
    x += dpp0<0x112, 0xf>(x); // row_shr:2
    x += dpp0<0x114, 0xf>(x); // row_shr:4
    x += dpp0<0x118, 0xf>(x); // row_shr:8
    x += dpp0<0x142, 0xa>(x); // row_bcast:15 -> rows 1,3
    x += dpp0<0x143, 0xc>(x); // row_bcast:31 -> rows 2,3
    return x;
}

// Inclusive 64-bit prefix sum over the wave as two 32-bit ones.  The low words are summed
// modulo 2^32; adding lane j's low word wrapped exactly when the running sum came out below that
// word (one addition wraps at most once), so the carries into the high word up to lane j are the
// wraps of lanes 0..j: a ballot and a bit count instead of carrying through every DPP step.
__device__ __forceinline__ uint64_t wave_scan_u64(uint64_t x)
{
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    const uint32_t slo = wave_scan_u32(lo);
    const uint32_t shi = wave_scan_u32(hi);
    const bool wrapped = slo < lo;
    const uint64_t wraps = __builtin_amdgcn_ballot_w64(wrapped);
    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(wraps >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)wraps, 0u));
    return ((uint64_t)(shi + below + (wrapped ? 1u : 0u)) << 32) | slo;
}

__device__ __forceinline__ uint32_t wave_last_u32(uint32_t x) { return __builtin_amdgcn_readlane((int)x, 63); }

__device__ __forceinline__ uint64_t wave_last_u64(uint64_t x)
{
    uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)x, 63);
    uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), 63);
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t x) { return wave_last_u32(wave_scan_u32(x)); }

// ---- varints from the LDS byte image ------------------------------------------

// ULEB128 as Go's encoding/binary.ReadUvarint reads it: at most 10 bytes, the
// 10th at most 1.  B[i] is stream byte c0 + i.  Returns a PQ_* status; on
// success *val holds the value and *pos has advanced past it.
__device__ __forceinline__ int read_uvarint(const uint8_t *B, int64_t c0, int64_t len, int64_t *pos,
                                   uint64_t *val)
{
    uint64_t x = 0;
    int s = 0;
    int64_t p = *pos;
    for (int i = 0; i < 10; ++i) {
        if (p >= len) return PQ_EOF;
        uint32_t b = B[p - c0];
        ++p;
        if (b < 0x80u) {
            if (i == 9 && b > 1u) return PQ_VARINT_OVERFLOW;
            *val = x | ((uint64_t)b << s);
            *pos = p;
            return PQ_OK;
        }
        x |= (uint64_t)(b & 0x7fu) << s;
        s += 7;
    }
    return PQ_VARINT_OVERFLOW;
}

__device__ __forceinline__ int64_t unzigzag(uint64_t u)
{
    int64_t x = (int64_t)(u >> 1);
    return (u & 1) ? ~x : x;
}

__device__ __forceinline__ uint32_t bits_len16(uint32_t x) { return 32u - (uint32_t)__clz((int)(x & 0xffffu)); }

} // namespace pq
