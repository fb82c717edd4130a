// bytearray_wave.hpp — DELTA_LENGTH_BYTE_ARRAY by one wave.
//
// Reference: ByteArrayDeltaLengthDecoder (types/byte-array.go:168-219).  Init decodes ALL the
// lengths the stream declares (an Int32 DELTA_BINARY_PACKED stream; any error there fails Init)
// and leaves the reader at the first value byte; DecodeValues then cuts value after value off
// the bytes that follow.  Here: the lengths through delta_wave into the caller's offsets area,
// an exclusive scan of them in place, one contiguous copy of the value bytes.
#pragma once

#include "delta_stream.hpp"
#include "delta_wave.hpp"

namespace pq {

__device__ __forceinline__ uint64_t wave_readlane_u64(uint64_t x, uint32_t l)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, (int)l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), (int)l);
    return ((uint64_t)hi << 32) | lo;
}

struct ByteArrayResult {
    int32_t status;
    uint32_t n_decoded;
    uint64_t consumed;
    uint32_t init_failed;
};

// The value count a DELTA_BINARY_PACKED header declares, read straight from memory (only used in
// front of the one-lane decoder).  false: the header does not parse that far.
__device__ __forceinline__ bool delta_peek_total(const uint8_t *S, int64_t len, uint32_t *total)
{
    int64_t pos = 0;
    uint64_t v[3];
    for (int k = 0; k < 3; ++k) {
        uint64_t x = 0;
        int shift = 0;
        while (true) {
            if (pos >= len || shift > 63) return false;
            const uint8_t b = S[pos++];
            x |= (uint64_t)(b & 0x7fu) << shift;
            shift += 7;
            if ((b & 0x80u) == 0) break;
        }
        v[k] = x;
    }
    *total = v[2] > 0xffffffffull ? 0xffffffffu : (uint32_t)v[2];
    return true;
}

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));  // 16 bytes at a 4-byte boundary

// dst[0, n) = src[0, n): any alignment on either side.  16 bytes per lane and step, four steps
// in flight; the destination pieces are 16-byte aligned, the source words are funnel-shifted.
__device__ __forceinline__ void wave_copy_bytes(uint8_t *dst, const uint8_t *src, uint64_t n, int lane)
{
    if (n == 0) return;
    // bytes in front of the first 16-byte aligned destination address
    const uint32_t head = (uint32_t)min((uint64_t)((16u - (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 15u)) & 15u), n);
    if ((uint32_t)lane < head) dst[lane] = src[lane];
    dst += head;
    src += head;
    n -= head;
    const uint64_t pieces = n >> 4;
    const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(src) & 3u) * 8u;
    const uint32_t *s32 = reinterpret_cast<const uint32_t *>(reinterpret_cast<uintptr_t>(src) & ~(uintptr_t)3);
    constexpr int kInFlight = 4;
    for (uint64_t i0 = (uint64_t)lane; i0 < pieces; i0 += (uint64_t)kWave * kInFlight) {
        u32x4_a4 v[kInFlight];
        uint32_t e[kInFlight];
#pragma unroll
        for (int u = 0; u < kInFlight; ++u) {
            const uint64_t i = i0 + (uint64_t)kWave * u;
            if (i < pieces) {
                v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_a4 *>(s32 + 4 * i));
                // the word behind the piece holds at least one source byte when the source is
                // not word aligned (n >= 16 (i + 1)); it is not touched otherwise
                e[u] = sh ? __builtin_nontemporal_load(s32 + 4 * i + 4) : 0u;
            }
        }
#pragma unroll
        for (int u = 0; u < kInFlight; ++u) {
            const uint64_t i = i0 + (uint64_t)kWave * u;
            if (i < pieces) {
                u32x4_t o;
                o.x = __builtin_amdgcn_alignbit(v[u].y, v[u].x, sh);
                o.y = __builtin_amdgcn_alignbit(v[u].z, v[u].y, sh);
                o.z = __builtin_amdgcn_alignbit(v[u].w, v[u].z, sh);
                o.w = __builtin_amdgcn_alignbit(e[u], v[u].w, sh);
                __builtin_nontemporal_store(o, reinterpret_cast<u32x4_t *>(dst) + i);
            }
        }
    }
    const uint32_t tail = (uint32_t)(n & 15u);
    if ((uint32_t)lane < tail) dst[16 * pieces + lane] = src[16 * pieces + lane];
}

// Every length an INT32 DELTA_BINARY_PACKED stream declares, into area[0, total): what
// Int32DeltaBPDecoder.Init + decodeInt32(make([]int32, ValuesCount)) leave behind
// (types/byte-array.go:177-185, :286-295).  ok: Init succeeded; otherwise status is its error
// (kOverCap: the stream declares more than cap lengths and nothing was decoded).
struct ByteArrayLens {
    int32_t status;
    uint32_t total;
    uint64_t consumed;
    bool ok;
};

__device__ __forceinline__ ByteArrayLens ba_lengths_wave(uint32_t *window, const uint8_t *S, int64_t len, uint32_t cap,
                                                         uint32_t *area, int lane)
{
    WaveDeltaResult lr = delta_wave<uint32_t, false>(window, S, len, 0x7fffffffu, area, lane, cap);
    // More lengths declared than the area holds: decode as many as fit.  A stream that fails
    // before that many ends Init with the reference's own error; only one that really yields
    // them all is refused for want of room.
    const bool over = lr.status == kOverCap;
    if (over) lr = delta_wave<uint32_t, false>(window, S, len, cap, area, lane);
    if (lr.status == kNeedGeneric) {
        // a block layout other than 128 / 4: one lane decodes the lengths
        uint32_t total = 0;
        const bool parsed = delta_peek_total(S, len, &total);
        DeltaResult sr{PQ_OK, 0, 0, 0};
        if (lane == 0) sr = delta_serial<uint32_t>(S, len, parsed ? min(total, cap) : 0u, area);
        lr.status = (int32_t)uni((uint32_t)sr.status);
        lr.n_decoded = uni(sr.n_decoded);
        lr.consumed = ((uint64_t)uni((uint32_t)(sr.consumed >> 32)) << 32) | uni((uint32_t)sr.consumed);
        lr.init_failed = uni(sr.init_failed);
        lr.total = parsed ? total : 0u;
    }
    if (over && !lr.init_failed && lr.status == PQ_OK && lr.n_decoded == cap) lr.status = kOverCap;
    ByteArrayLens r;
    r.status = lr.status;
    r.total = lr.total;
    r.consumed = lr.consumed;
    // asking for more than there are ends in io.EOF right behind the last one, which is not an
    // error of Init
    r.ok = !over && !lr.init_failed && lr.n_decoded == lr.total && (lr.status == PQ_OK || lr.status == PQ_EOF);
    return r;
}

// Decode DecodeValues(dest[:n]) of a fresh decoder over [S, S + len).
//   area: cap + 1 uint32, 4-byte aligned; on return area[0..n_decoded] are the value offsets
//         (area[0] = 0) and area beyond them is scratch.
//   data: the value bytes, back to back (at most len of them).
// A stream declaring more than cap lengths is refused with PQ_ERR_OUT_OF_RANGE.
__device__ __forceinline__ ByteArrayResult delta_length_ba_wave(uint32_t *window, const uint8_t *S, int64_t len,
                                                                uint32_t n, uint32_t cap, uint32_t *area, uint8_t *data,
                                                                int lane)
{
    ByteArrayResult res{PQ_OK, 0, 0, 0};
    {
        // The usual page in one pass: the lengths leave the delta decoder as offsets (a second
        // scan inside the tile), the first value the bytes run out at is looked up afterwards.
        // Anything else — a failing stream, a negative or huge length, sums that might not fit
        // 32 bits — is decoded again below, lengths first.
        WaveOffsets wo{0u, 0u};
        const WaveDeltaResult fr = delta_wave<uint32_t, false, true>(window, S, len, 0x7fffffffu, area, lane, cap, &wo);
        const bool usual = !fr.init_failed && (fr.status == PQ_OK || fr.status == PQ_EOF) && fr.n_decoded == fr.total &&
                           uni(wo.big) == 0u;
        if (usual) {
            const uint32_t total = fr.total, m = min(n, total);
            const uint64_t avail = (uint64_t)len - fr.consumed;  // bytes behind the lengths
            if (lane == 0) area[total] = wo.sum;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            auto off = [&](uint32_t i) -> uint32_t {
                return uni(__hip_atomic_load(area + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            };
            uint32_t done = m;
            uint64_t bytes = off(m);
            res.status = n > total ? PQ_EOF : PQ_OK;  // next() past the last length (:205-207)
            if (bytes > avail) {
                // io.ReadFull of value `lo` fails (:212): the first i with off[i + 1] > avail
                uint32_t lo = 0, hi = m - 1u;
                while (lo < hi) {
                    const uint32_t mid = lo + (hi - lo) / 2u;
                    if ((uint64_t)off(mid + 1u) > avail)
                        hi = mid;
                    else
                        lo = mid + 1u;
                }
                done = lo;
                bytes = off(lo);
                res.status = bytes == avail ? PQ_EOF : PQ_UNEXPECTED_EOF;
                res.consumed = fr.consumed + avail;  // a short read takes what is left
            } else {
                res.consumed = fr.consumed + bytes;
            }
            wave_copy_bytes(data, S + fr.consumed, bytes, lane);
            res.n_decoded = done;
            return res;
        }
    }
    const ByteArrayLens lr = ba_lengths_wave(window, S, len, cap, area, lane);
    if (!lr.ok) {
        res.status = lr.status == kOverCap ? PQ_ERR_OUT_OF_RANGE : lr.status;
        res.init_failed = 1;
        res.consumed = lr.status == kOverCap ? 0 : lr.consumed;
        return res;
    }
    const uint32_t total = lr.total;
    // the lengths went out as plain and non-temporal stores: make them visible to the loads below
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    const uint64_t avail = (uint64_t)len - lr.consumed;  // bytes behind the lengths
    const uint32_t m = min(n, total);
    uint64_t carry = 0;
    uint32_t done = 0;
    int32_t status = PQ_OK;
    constexpr uint32_t kPer = 4;  // lengths per lane and step
    // offsets[i + 1] takes the place of length i + 1: a step's lengths are loaded before the
    // step in front of it stores
    auto load_lens = [&](uint32_t i0, int32_t *c) {
#pragma unroll
        for (uint32_t k = 0; k < kPer; ++k)
            c[k] = i0 + k < m ? (int32_t)__hip_atomic_load(area + i0 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0;
    };
    int32_t cur[kPer], next[kPer];
    load_lens(kPer * (uint32_t)lane, cur);
    for (uint32_t base = 0; base < m; base += kPer * kWave) {
        load_lens(base + kPer * (kWave + (uint32_t)lane), next);
        const uint32_t i0 = base + kPer * (uint32_t)lane;
        // this lane's lengths up to its first negative one
        uint64_t inc[kPer], sum = 0;
        uint32_t failk = kPer;  // first value of this lane that fails
        bool neg = false;
#pragma unroll
        for (uint32_t k = 0; k < kPer; ++k) {
            if (i0 + k < m && failk == kPer) {
                if (cur[k] < 0) {
                    failk = k;
                    neg = true;
                } else {
                    sum += (uint64_t)(uint32_t)cur[k];
                }
            }
            inc[k] = sum;
        }
        const uint64_t excl = carry + wave_scan_u64(sum) - sum;
        // io.ReadFull of a value fails when the bytes run out
#pragma unroll
        for (uint32_t k = 0; k < kPer; ++k) {
            if (i0 + k < m && k < failk && excl + inc[k] > avail) {
                failk = k;
                neg = false;
            }
        }
        const uint64_t fm = __builtin_amdgcn_ballot_w64(failk < kPer);
        const uint32_t ok_lanes = fm ? (uint32_t)__builtin_ctzll(fm) : (uint32_t)kWave;
        if ((uint32_t)lane <= ok_lanes) {
            const uint32_t lim = (uint32_t)lane < ok_lanes ? kPer : failk;
#pragma unroll
            for (uint32_t k = 0; k < kPer; ++k)
                if (i0 + k < m && k < lim) area[i0 + k + 1] = (uint32_t)(excl + inc[k]);
        }
        if (base == 0 && lane == 0) area[0] = 0;
        if (fm != 0ull) {
            // everything in front of the failing value is good
            const uint32_t fk = (uint32_t)__builtin_amdgcn_readlane((int)failk, (int)ok_lanes);
            uint64_t in_front = 0;  // this lane's good lengths
#pragma unroll
            for (uint32_t k = 0; k < kPer; ++k)
                if (k < failk) in_front = inc[k];
            const uint64_t mine = excl + in_front;
            const uint64_t before = wave_readlane_u64(mine, ok_lanes);
            const bool was_neg = ((__builtin_amdgcn_ballot_w64(neg) >> ok_lanes) & 1ull) != 0ull;
            done = base + kPer * ok_lanes + fk;
            status = was_neg ? PQ_BYTE_ARRAY_NEGATIVE_LEN : (before == avail ? PQ_EOF : PQ_UNEXPECTED_EOF);
            // a negative length panics before anything is read; a short read takes what is left
            res.consumed = lr.consumed + (was_neg ? before : avail);
            carry = before;
            break;
        }
        carry = wave_last_u64(excl + sum);
        done = min(base + kPer * (uint32_t)kWave, m);
#pragma unroll
        for (uint32_t k = 0; k < kPer; ++k) cur[k] = next[k];
    }
    if (m == 0 && lane == 0) area[0] = 0;
    if (status == PQ_OK) {
        res.consumed = lr.consumed + carry;
        if (n > total) status = PQ_EOF;  // next() past the last length (:205-207)
    }
    wave_copy_bytes(data, S + lr.consumed, carry, lane);
    res.status = status;
    res.n_decoded = done;
    return res;
}

// ---------------------------------------------------------------------------------------------
// DELTA_BYTE_ARRAY.  Reference: ByteArrayDeltaDecoder (types/byte-array.go:279-348).  Init decodes
// all prefix lengths, then the suffix decoder's Init all suffix lengths; the counts have to agree.
// Value i = the first prefixLen[i] bytes of value i - 1, then suffix i.
//
// Which byte of the suffix area ends up at position j of value i does not need the values in
// front of it: it is byte j - p[k] of suffix k for the nearest k <= i with p[k] <= j.  A batch of
// 64 values (one per lane) finds that k with one ballot per byte position; positions no value of
// the batch starts a suffix at come from the batch in front (its last value, already in `data`).

constexpr uint32_t kBaOutWin = 2048;           // bytes of the output a wave keeps in LDS (a power of two)
constexpr uint32_t kBaSufWin = kBaOutWin + 16;  // a batch's suffix bytes, from a 16-byte boundary
constexpr uint32_t kBaPrefixCols = 4;          // byte positions per step of the column loops: their LDS reads go
constexpr uint32_t kBaCols = 4;                // out together, but every position of a step is paid for
constexpr uint32_t kBaLdsBytes = kBaOutWin + kBaSufWin;

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = max(x, (uint32_t)__shfl_xor((int)x, o));
    return x;
}

__device__ __forceinline__ void workgroup_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

//   area: 2 * cap + 1 uint32, 4-byte aligned: area[0..n_decoded] receive the value offsets, the
//         rest is scratch (the prefix lengths live in its upper half).
//   data: data_cap bytes for the values, back to back.  Values that do not fit: PQ_ERR_OUT_OF_RANGE.
__device__ __forceinline__ ByteArrayResult delta_ba_wave(uint32_t *window, const uint8_t *S, int64_t len, uint32_t n,
                                                         uint32_t cap, uint32_t *area, uint8_t *data, uint64_t data_cap,
                                                         int lane)
{
    ByteArrayResult res{PQ_OK, 0, 0, 0};
    uint32_t *P = area + cap + 1;
    const ByteArrayLens lp = ba_lengths_wave(window, S, len, cap, P, lane);
    if (!lp.ok) {
        res.status = lp.status == kOverCap ? PQ_ERR_OUT_OF_RANGE : lp.status;
        res.init_failed = 1;
        res.consumed = lp.status == kOverCap ? 0 : lp.consumed;
        return res;
    }
    const ByteArrayLens ls = ba_lengths_wave(window, S + lp.consumed, len - (int64_t)lp.consumed, cap, area, lane);
    const uint64_t init_consumed = lp.consumed + ls.consumed;
    if (!ls.ok) {
        res.status = ls.status == kOverCap ? PQ_ERR_OUT_OF_RANGE : ls.status;
        res.init_failed = 1;
        res.consumed = ls.status == kOverCap ? 0 : init_consumed;
        return res;
    }
    if (lp.total != ls.total) {  // :301-308
        res.status = PQ_BYTE_ARRAY_COUNT_MISMATCH;
        res.init_failed = 1;
        res.consumed = init_consumed;
        return res;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    // ---- one pass over batches of values: lengths -> offsets, the first value DecodeValues fails at,
    // and the bytes ----
    // Batches of up to 64 values (one per lane).  A batch's lengths are read once: its offsets (which
    // take the place of the suffix lengths, one slot further on) and the checks of the reference's
    // loop come from wave scans of them.  The batch is then put together in LDS: its suffix bytes
    // are staged with 16-byte loads, every byte of a value finds its suffix there (or in the value
    // in front of the batch, which the output window still holds), and whole 16-byte pieces of the
    // output window go to memory.  The window is circular, indexed by the low bits of the output
    // ADDRESS, so a 16-byte piece of memory is a 16-byte piece of LDS and nothing is ever moved.
    const uint32_t total = ls.total;
    const uint64_t avail = (uint64_t)len - init_consumed;  // suffix bytes
    const uint32_t m = min(n, total);
    const uint8_t *SUF = S + init_consumed;
    uint8_t *obuf = reinterpret_cast<uint8_t *>(window);  // the lengths are done with the window
    uint8_t *sbuf = obuf + kBaOutWin;
    const uint32_t dlow = (uint32_t)reinterpret_cast<uintptr_t>(data);
    const uint64_t below = (2ull << lane) - 1ull;
    bool prev_in_lds = false;
    uint64_t sc = 0;        // suffix bytes in front of this batch
    uint64_t oc = 0;        // value bytes in front of this batch: its first value's offset
    uint32_t prev_len = 0;  // length of the value in front of this batch
    uint32_t o_prev = 0;    // its offset
    uint32_t flushed = 0;   // data[0, flushed) is in memory; [flushed, batch start) waits in the window
    uint32_t base = 0;
    int32_t status = PQ_OK;
    res.consumed = init_consumed;
    auto load_word = [&](const uint32_t *from, uint32_t i) -> int32_t {
        return i < m ? (int32_t)__hip_atomic_load(from + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0;
    };
    // The lengths of the 64 values from `b` on, when this batch took `took` of the 64 it looked at:
    // those it left come over from the lanes that hold them (the offset stored behind the last
    // value taken has overwritten the first of them in memory), the others are loaded.
    auto next_lens = [&](uint32_t b, uint32_t took, int32_t have_s, int32_t have_p, int32_t &ns, int32_t &np) {
        const uint32_t from = (uint32_t)lane + took;
        const int32_t moved_s = __shfl(have_s, (int)(from & 63u)), moved_p = __shfl(have_p, (int)(from & 63u));
        const bool fresh = from >= (uint32_t)kWave;
        const int32_t ld_s = fresh ? load_word(area, b + (uint32_t)lane) : 0;
        const int32_t ld_p = fresh ? load_word(P, b + (uint32_t)lane) : 0;
        ns = fresh ? ld_s : moved_s;
        np = fresh ? ld_p : moved_p;
    };
    int32_t cur_s = load_word(area, (uint32_t)lane), cur_p = load_word(P, (uint32_t)lane);
    if (lane == 0) area[0] = 0;
    bool stop = false;
    while (base < m && !stop) {
        const uint32_t i = base + (uint32_t)lane;
        const bool in_range = i < m;
        const int32_t sl = cur_s, pl = cur_p;
        // ---- the reference's loop, value by value (:318-343), for the whole batch at once ----
        const bool neg_s = in_range && sl < 0;  // make([]byte, size) panics (:210)
        // lengths below 2^24 (any real page): 64 of them add up in 32 bits, one scan instead of two and a carry count
        const bool small = __builtin_amdgcn_ballot_w64(in_range && (((uint32_t)sl | (uint32_t)pl) >> 24) != 0u) == 0ull;
        const uint64_t s_here = in_range && !neg_s ? (uint64_t)(uint32_t)sl : 0ull;
        const uint64_t sincl = sc + (small ? (uint64_t)wave_scan_u32((uint32_t)s_here) : wave_scan_u64(s_here));
        const bool rd_fail = in_range && !neg_s && sincl > avail;                  // io.ReadFull (:212)
        const bool neg_cap = in_range && !neg_s && (int64_t)pl + (int64_t)sl < 0;  // make(.., 0, prefixLen + len(suffix)) (:326)
        const uint32_t pz = pl > 0 ? (uint32_t)pl : 0u;
        const uint32_t vlen_all = in_range && !neg_s && !neg_cap ? pz + (uint32_t)sl : 0u;
        const uint32_t up = (uint32_t)__shfl_up((int)vlen_all, 1);
        const uint32_t before_len = lane == 0 ? prev_len : up;
        const bool bad_p = in_range && (int64_t)before_len < (int64_t)pl;  // :328-336
        const uint64_t fm = __builtin_amdgcn_ballot_w64(neg_s || rd_fail || neg_cap || bad_p);
        const uint32_t good = fm ? (uint32_t)__builtin_ctzll(fm) : (uint32_t)kWave;  // values of the batch in front of the first failure
        const uint64_t v_here = (uint32_t)lane < good ? (uint64_t)vlen_all : 0ull;
        const uint64_t oincl = oc + (small ? (uint64_t)wave_scan_u32((uint32_t)v_here) : wave_scan_u64(v_here));
        if (fm != 0ull) {
            const uint32_t f = good;
            const uint64_t s_before = f ? wave_readlane_u64(sincl, f - 1u) : sc;
            const uint64_t s_at = wave_readlane_u64(sincl, f);
            if ((__builtin_amdgcn_ballot_w64(neg_s) >> f) & 1ull) {
                status = PQ_BYTE_ARRAY_NEGATIVE_LEN;
                res.consumed = init_consumed + s_before;
            } else if ((__builtin_amdgcn_ballot_w64(rd_fail) >> f) & 1ull) {
                status = s_before == avail ? PQ_EOF : PQ_UNEXPECTED_EOF;
                res.consumed = init_consumed + avail;
            } else if ((__builtin_amdgcn_ballot_w64(neg_cap) >> f) & 1ull) {
                status = PQ_BYTE_ARRAY_NEGATIVE_LEN;
                res.consumed = init_consumed + s_at;
            } else {
                status = PQ_BYTE_ARRAY_INVALID_PREFIX;
                res.consumed = init_consumed + s_at;
            }
        }
        const uint32_t n_in = min(min((uint32_t)kWave, m - base), good);  // values this batch may take
        if (n_in == 0u) {
            // the batch's first value fails; what the batch in front left in the window goes out
            const uint32_t pend = (uint32_t)oc - flushed;  // < 16 bytes
            if ((uint32_t)lane < pend) data[flushed + lane] = obuf[(dlow + flushed + (uint32_t)lane) & (kBaOutWin - 1u)];
            break;
        }
        // values that do not fit the caller's area: nothing is returned
        const uint64_t o_last = wave_readlane_u64(oincl, n_in - 1u);
        if (o_last > data_cap) {
            res.status = PQ_ERR_OUT_OF_RANGE;
            res.n_decoded = 0;
            return res;
        }
        const uint32_t o1 = (uint32_t)oincl, o0 = o1 - ((uint32_t)lane < good ? vlen_all : 0u);
        const bool usable = (uint32_t)lane < n_in;
        const uint32_t o_start = (uint32_t)oc;
        const uint32_t ws = min(prev_in_lds ? o_prev : o_start, flushed);  // the window has to hold [ws, batch end)
        const uint64_t fits = __builtin_amdgcn_ballot_w64(usable && o1 - ws <= kBaOutWin);
        const uint32_t nb = ~fits ? (uint32_t)__builtin_ctzll(~fits) : (uint32_t)kWave;
        if (nb == 0u || (base != 0u && !prev_in_lds)) {
            // a value the window cannot hold next to the one in front of it, or whose predecessor
            // the window does not hold: straight through memory
            int32_t ns, np;
            next_lens(base + 1u, 1u, cur_s, cur_p, ns, np);
            const uint32_t pend = o_start - flushed;  // < 16 bytes of the batch in front
            if ((uint32_t)lane < pend) data[flushed + lane] = obuf[(dlow + flushed + (uint32_t)lane) & (kBaOutWin - 1u)];
            const uint32_t vl = uni(vlen_all);
            const uint32_t pfx = uni(pz);
            if (pfx > 0u) {
                workgroup_fence();  // the value in front has to be in memory
                wave_copy_bytes(data + o_start, data + o_prev, pfx, lane);
            }
            wave_copy_bytes(data + o_start + pfx, SUF + sc, vl - pfx, lane);
            workgroup_fence();
            prev_in_lds = vl <= kBaOutWin;
            if (prev_in_lds) {  // back into the window, for the batch behind it
                for (uint32_t j = (uint32_t)lane; j < vl; j += kWave)
                    obuf[(dlow + o_start + j) & (kBaOutWin - 1u)] = data[o_start + j];
                wave_sync();
            }
            if (lane == 0) area[base + 1u] = o_start + vl;
            o_prev = o_start;
            flushed = o_start + vl;
            prev_len = vl;
            sc += vl - pfx;
            oc += vl;
            base += 1u;
            res.consumed = fm != 0ull && good == 1u ? res.consumed : init_consumed + sc;
            stop = fm != 0ull && good == 1u;
            cur_s = ns;
            cur_p = np;
            continue;
        }
        const bool live = (uint32_t)lane < nb;
        const uint32_t p = live ? pz : 0x7fffffffu;
        const uint32_t vlen = live ? vlen_all : 0u;
        const uint32_t s = live ? vlen - pz : 0u;
        const uint32_t sincl32 = (uint32_t)(sincl - sc);  // the lanes that are live added their suffix lengths above
        const uint32_t stot = (uint32_t)__builtin_amdgcn_readlane((int)sincl32, (int)(nb - 1u));
        const uint32_t o_end = (uint32_t)__builtin_amdgcn_readlane((int)o1, (int)(nb - 1u));

        // the batch's suffix bytes: every 16-byte piece of the stream's buffer that holds one
        const uintptr_t sa = reinterpret_cast<uintptr_t>(SUF) + sc;
        const uint32_t shift = (uint32_t)(sa & 15u);
        const u32x4_t *s16 = reinterpret_cast<const u32x4_t *>(sa - shift);
        const uint32_t spieces = stot ? (shift + stot + 15u) >> 4 : 0u;  // <= (15 + 2048 + 15) / 16
        u32x4_t sv[3];
#pragma unroll
        for (uint32_t u = 0; u < 3u; ++u) {
            const uint32_t q = (uint32_t)lane + kWave * u;
            if (q < spieces) sv[u] = s16[q];
        }
        // the next batch's lengths are asked for while this one's suffix bytes are on their way
        int32_t ns, np;
        next_lens(base + nb, nb, cur_s, cur_p, ns, np);
        if (live) area[i + 1u] = o1;  // behind the loads above: slot i + 1 held the next value's suffix length
#pragma unroll
        for (uint32_t u = 0; u < 3u; ++u) {
            const uint32_t q = (uint32_t)lane + kWave * u;
            if (q < spieces) reinterpret_cast<u32x4_t *>(sbuf)[q] = sv[u];
        }
        wave_sync();

        const uint32_t srcb = shift + (sincl32 - s);     // sbuf index of this value's suffix (sbuf = obuf + kBaOutWin)
        const uint32_t srcp = srcb - p;                  // ... of its position 0, were the suffix the whole value
        const uint32_t ob = dlow + o0;                   // window index (mod its size) of the value's position 0
        const uint32_t pb = dlow + o_prev;               // ... of the value in front of the batch
        // positions in front of the suffix: the nearest value at or below this one that starts
        // its suffix at or before position j supplies the byte
        for (uint32_t j0 = 0; __builtin_amdgcn_ballot_w64(live && j0 < p) != 0ull; j0 += kBaPrefixCols) {
            uint8_t got[kBaPrefixCols];
#pragma unroll
            for (uint32_t u = 0; u < kBaPrefixCols; ++u) {
                const uint32_t j = j0 + u;
                const uint64_t mk = __builtin_amdgcn_ballot_w64(p <= j) & below;
                const bool has = mk != 0ull;
                const int k = has ? 63 - __builtin_clzll(mk) : 0;
                const uint32_t sb = (uint32_t)__shfl((int)srcp, k);
                // one LDS read either way: the staged suffix, or the value in front of the batch
                const uint32_t at = has ? kBaOutWin + sb + j : (pb + j) & (kBaOutWin - 1u);
                const uint8_t b = obuf[live && j < p ? at : 0u];
                got[u] = b;
            }
#pragma unroll
            for (uint32_t u = 0; u < kBaPrefixCols; ++u) {
                const uint32_t j = j0 + u;
                if (live && j < p) obuf[(ob + j) & (kBaOutWin - 1u)] = got[u];
            }
        }
        // the value's own suffix
        const uint32_t os = ob + p;
        for (uint32_t t0 = 0; __builtin_amdgcn_ballot_w64(t0 < s) != 0ull; t0 += kBaCols) {
            uint8_t got[kBaCols];
#pragma unroll
            for (uint32_t u = 0; u < kBaCols; ++u) got[u] = t0 + u < s ? sbuf[srcb + t0 + u] : (uint8_t)0;
#pragma unroll
            for (uint32_t u = 0; u < kBaCols; ++u)
                if (t0 + u < s) obuf[(os + t0 + u) & (kBaOutWin - 1u)] = got[u];
        }
        wave_sync();

        // whole 16-byte pieces of [flushed, o_end) go to memory; the rest waits for the next batch
        const bool failing_here = fm != 0ull && nb == good;  // the value behind this batch is the one that fails
        const bool last = base + nb >= m || failing_here;
        uint32_t f = flushed;
        const uint32_t head = min((16u - ((dlow + f) & 15u)) & 15u, o_end - f);
        if ((uint32_t)lane < head) data[f + lane] = obuf[(dlow + f + (uint32_t)lane) & (kBaOutWin - 1u)];
        f += head;
        const uint32_t opieces = (o_end - f) >> 4;  // <= 128
#pragma unroll
        for (uint32_t u = 0; u < 2u; ++u) {
            const uint32_t q = (uint32_t)lane + kWave * u;
            if (q < opieces) {
                const u32x4_t v = *reinterpret_cast<const u32x4_t *>(obuf + ((dlow + f + 16u * q) & (kBaOutWin - 1u)));
                reinterpret_cast<u32x4_t *>(data + f)[q] = v;
            }
        }
        f += 16u * opieces;
        if (last) {
            const uint32_t tail = o_end - f;
            if ((uint32_t)lane < tail) data[f + lane] = obuf[(dlow + f + (uint32_t)lane) & (kBaOutWin - 1u)];
            f = o_end;
        }
        flushed = f;
        o_prev = (uint32_t)__builtin_amdgcn_readlane((int)o0, (int)(nb - 1u));
        prev_len = (uint32_t)__builtin_amdgcn_readlane((int)vlen, (int)(nb - 1u));
        prev_in_lds = true;
        sc += stot;
        oc = (uint64_t)o_end;
        base += nb;
        if (!failing_here) res.consumed = init_consumed + sc;
        stop = failing_here;
        cur_s = ns;
        cur_p = np;
    }
    if (status == PQ_OK && n > total) status = PQ_EOF;  // next() past the last length (:205-207)
    res.status = status;
    res.n_decoded = status == PQ_BYTE_ARRAY_INVALID_PREFIX ? 0u : base;  // "return 0, err" (:330)
    return res;
}

// Either byte-array encoding, output areas taken from the page descriptor.
__device__ __forceinline__ ByteArrayResult byte_array_wave(uint32_t kind, uint32_t *window, const uint8_t *S, int64_t len,
                                                           uint32_t n, const pq_page_desc &d, uint8_t *values, int lane)
{
    uint32_t *area = reinterpret_cast<uint32_t *>(values + d.dict_off);
    if (kind == PQ_KIND_DELTA_BA)
        return delta_ba_wave(window, S, len, n, d.dict_len, area, values + d.dst_off, (uint64_t)d.dst_cap, lane);
    return delta_length_ba_wave(window, S, len, n, d.dict_len, area, values + d.dst_off, lane);
}

}  // namespace pq
