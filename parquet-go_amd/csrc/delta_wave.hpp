// delta_wave.hpp — one WAVE per page DELTA_BINARY_PACKED decoder for the layouts the
// mainstream writers emit: 128-value blocks of 4 miniblocks (32 values each; the reference's
// own encoders, parquet-mr, Arrow's INT32) and 256-value blocks of 4 miniblocks (64 values
// each; Arrow's INT64).  A tile is 128 values either way: a block, or half of one.
//
// Same value sequence as delta_stream.hpp (reference:
// encoding/delta-binary-packed-decoder.go:29-170,213-225,289-301), organised so that
// nothing ever waits on another wave: each wave owns a page, a private 3 KiB LDS
// window and its running value.  Per iteration a wave
//   1. walks up to 4 block headers from its LDS window on the SCALAR unit (zig-zag
//      minDelta, 4 width bytes, prefix of the widths by one multiply),
//   2. issues the coalesced 16-byte loads of the NEXT window into registers,
//   3. decodes the blocks: each lane unpacks 2 adjacent deltas with a funnel read
//      of 3 LDS words, a DPP inclusive scan over the lane sums gives the in-block
//      prefix (32-bit when widths and minDelta allow, else 64-bit), the running
//      value is added and 2 x int64 go out in one 16-byte store,
//   4. commits the prefetched registers to LDS.
// A block that is the page's last, is cut short or has an unusual header goes
// through wave_walk_slow, which applies every check of the reference's reader.
// Pages with any other layout are handed to the workgroup-per-page kernel.
#pragma once

#include "pq_device.hpp"

namespace pq {

constexpr int kDwChunk = 3072;    // bytes of the stream held in a wave's LDS window
constexpr int kDwBlockMax = 544;  // >= 10 + 4 + 512 (largest 128-value block), a multiple of 16
constexpr int kDwStride = kDwChunk - kDwBlockMax; // distance between consecutive windows
constexpr int kDwPieces = kDwChunk / 16 / kWave; // 16-byte pieces per lane
constexpr int32_t kNeedGeneric = -2;
constexpr int32_t kOverCap = -3;  // the stream declares more values than the caller's `cap`
constexpr int32_t kSegStopped = -4; // a segment decode reached the block header it was told to stop at

__device__ __forceinline__ void wave_sync()
{
    // orders this wave's LDS traffic for the compiler; DS operations of one wave
    // execute in order, so no hardware barrier is needed
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }

struct WavePrefetch {
    uint4 v[kDwPieces];
};

// Issue the loads of the window [c0, c0 + kDwChunk) and touch none of the results, so
// that they stay in flight until wave_commit.  Every 16-byte piece that intersects the
// stream is loaded whole: the few bytes of such a piece that lie before the stream or
// past its end belong to the same 16-byte line of the caller's buffer, and the decoder
// never lets them reach a stored value (availability is decided from `len` alone).
__device__ __forceinline__ void wave_fetch(WavePrefetch &pf, const uint8_t *S, int64_t c0, int64_t len, int lane)
{
#pragma unroll
    for (int k = 0; k < kDwPieces; ++k) {
        const int64_t o = c0 + 16 * (int64_t)(lane + kWave * k);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (o + 16 > 0 && o < len) v = *reinterpret_cast<const uint4 *>(S + o);
        pf.v[k] = v;
    }
}

__device__ __forceinline__ void wave_commit(uint32_t *lds, const WavePrefetch &pf, int lane)
{
    uint4 *l4 = reinterpret_cast<uint4 *>(lds);
#pragma unroll
    for (int k = 0; k < kDwPieces; ++k) l4[lane + kWave * k] = pf.v[k];
}

// What the walk of one block leaves for the decode (all wave-uniform).
struct WaveTile {
    uint32_t bitbase; // bit offset of the block's packed deltas in the LDS window
    uint32_t widths;  // 4 miniblock widths, one per byte
    uint32_t prefix;  // byte j = widths[0] + ... + widths[j-1]
    uint64_t min_delta;
    uint32_t p0;      // position of the block's first value
    uint32_t valid;   // 1 = decode this tile
    // fast walk only: minDelta is left as its raw varint (first 4 header bytes and the
    // varint's length, 1..4) for the vector unit to expand; md_raw == 0: min_delta holds it
    uint32_t md_raw, hdr_lo, vl;
};

template <typename UT>
struct WaveState {
    int64_t pos;      // stream offset of the next block header
    int64_t end_pos;  // just past the last group that was read
    uint32_t tiles_done;
    uint32_t need, total;
    uint32_t n_ok;
    int32_t status;
    uint32_t finished;
    uint32_t init_failed;
    uint32_t last_widths; // widths of the block that holds the last group read
    uint32_t last_group;  // index of that group inside its block (0..15, or 0..31 for 256-value blocks)
    uint32_t lg;          // log2 of the values per miniblock: 5 (blocks of 128) or 6 (blocks of 256)
};

// The block the walk is inside of (all wave-uniform).  A tile is always 128 values: a whole block
// of the 128 / 4 layout, half a block of the 256 / 4 layout (the int64 layout of Arrow's writer);
// the second half reuses the header its first half parsed.
struct WaveBlock {
    uint32_t half;    // 256 / 4 only: 1 = the next tile is the second half of this block
    uint32_t wd;      // the block's 4 miniblock widths
    uint32_t md_raw;  // 1: minDelta is still its raw varint (hdr_lo, vl), else md holds it
    uint32_t hdr_lo, vl;
    uint64_t md;
};

// minDelta from a raw varint of 1..4 bytes (scalar twin of wave_expand_min_delta).
__device__ __forceinline__ int64_t wave_raw_min_delta(uint32_t hdr_lo, uint32_t vl)
{
    const uint32_t v = hdr_lo & (0xffffffffu >> (32u - 8u * vl));
    const uint32_t x = (v & 0x7fu) | ((v >> 1) & 0x3f80u) | ((v >> 2) & 0x1fc000u) | ((v >> 3) & 0xfe00000u);
    return (int64_t)((int32_t)(x >> 1) ^ -(int32_t)(x & 1u));
}

__device__ __forceinline__ WaveTile wave_tile_none()
{
    WaveTile t;
    t.valid = 0;
    t.bitbase = 0;
    t.widths = 0;
    t.prefix = 0;
    t.min_delta = 0;
    t.p0 = 0;
    t.md_raw = 0;
    t.hdr_lo = 0;
    t.vl = 0;
    return t;
}

// Walk one block header at st.pos with every check of the reference's reader:
// taken for a page's last block, for blocks cut short by the end of the stream
// and for malformed headers.  B = LDS window bytes, c0 = its stream offset.
template <typename UT, uint32_t LG>
__device__ __forceinline__ WaveTile wave_walk_slow(WaveState<UT> &st, WaveBlock &blk, const uint8_t *B, int64_t c0,
                                                   int64_t len)
{
    WaveTile t = wave_tile_none();
    if (st.finished) return t;
    const uint32_t p0 = st.tiles_done * 128u;
    if (p0 >= st.need) {
        st.finished = 1;
        st.n_ok = st.need;
        return t;
    }
    constexpr uint32_t lg = LG;
    constexpr uint32_t G = 1u << (lg - 3u);    // groups per miniblock: 4 or 8
    constexpr uint32_t MT = 1u << (7u - lg);   // miniblocks per tile: 4 or 2
    const uint32_t half = lg == 5u ? 0u : blk.half;
    int64_t pos = st.pos;
    int64_t md = 0;
    uint32_t wd = 0;
    if (half == 0u) {
        // a block header: minDelta and the widths of all its miniblocks (:67-94)
        uint64_t u;
        int hs = read_uvarint(B, c0, len, &pos, &u);
        if (hs == PQ_OK) {
            md = unzigzag(u);
            if (sizeof(UT) == 4 && (md > 2147483647ll || md < -2147483648ll)) hs = PQ_INT32_OUT_OF_RANGE;
        }
        if (hs == PQ_OK) {
            const int64_t rem = len - pos;
            if (rem <= 0) hs = PQ_EOF;
            else if (rem < 4) hs = PQ_UNEXPECTED_EOF;
            else {
                const uint32_t o = (uint32_t)(pos - c0);
                wd = (uint32_t)B[o] | ((uint32_t)B[o + 1] << 8) | ((uint32_t)B[o + 2] << 16) | ((uint32_t)B[o + 3] << 24);
                pos += 4;
                // any byte > 32 ?  (byte + 95 carries into bit 7, or bit 7 already set)
                if ((((wd & 0x7f7f7f7fu) + 0x5f5f5f5fu) | wd) & 0x80808080u) hs = PQ_INVALID_MINIBLOCK_WIDTH;
            }
        }
        if (hs != PQ_OK) {
            st.status = hs;
            if (p0 == 0) st.init_failed = 1;
            st.n_ok = p0;
            st.finished = 1;
            return t;
        }
    } else {
        wd = blk.wd;
        md = blk.md_raw ? wave_raw_min_delta(blk.hdr_lo, blk.vl) : (int64_t)blk.md;
    }
    const uint32_t tw = lg == 5u ? wd : (half ? wd >> 16 : wd & 0xffffu);  // widths of this tile's miniblocks
    const uint32_t prefix = tw * 0x01010100u;           // bytes: 0, w0, w0+w1, w0+w1+w2
    const uint32_t total_w = (tw * 0x01010101u) >> 24;  // their sum, <= 128
    const uint32_t tile_vals = min(128u, st.need - p0);
    const uint32_t gn = (tile_vals + 7u) >> 3;          // groups that will be read
    const int64_t avail = len - pos;
    uint32_t ok_groups = gn;
    if ((int64_t)(G * total_w) > avail || gn < 16u) {
        // the tile is the stream's last or is cut short: io.ReadFull per group (:111-114)
        int64_t off = 0;
        for (uint32_t j = 0; j < MT; ++j) {
            const uint32_t w = (tw >> (8u * j)) & 0xffu;
            if (gn <= G * j) break;
            const uint32_t gj = min(G, gn - G * j);
            if (off + (int64_t)(gj * w) > avail) {
                const int64_t rem = avail > off ? avail - off : 0;
                const uint32_t gok = w ? (uint32_t)(rem / w) : gj;
                ok_groups = G * j + gok;
                st.status = (rem - (int64_t)gok * w) == 0 ? PQ_EOF : PQ_UNEXPECTED_EOF;
                break;
            }
            off += (int64_t)G * (int64_t)w;
        }
    }
    t.valid = 1;
    t.p0 = p0;
    t.bitbase = (uint32_t)(pos - c0) * 8u;
    t.widths = tw;
    t.prefix = prefix;
    t.min_delta = (uint64_t)md;
    if (ok_groups > 0) {
        const uint32_t gl = ok_groups - 1u; // last group read in this tile
        const uint32_t j = gl >> (lg - 3u);
        const uint32_t wj = (tw >> (8u * j)) & 0xffu;
        st.end_pos = pos + (int64_t)G * (int64_t)((prefix >> (8u * j)) & 0xffu) + (int64_t)((gl & (G - 1u)) + 1u) * wj;
        st.last_widths = wd;
        st.last_group = gl + 16u * half;
    }
    if (st.status != PQ_OK) {
        st.n_ok = p0 + 8u * ok_groups;
        st.finished = 1;
        st.end_pos = len;
    } else if (p0 + 128u >= st.need) {
        st.n_ok = st.need;
        st.finished = 1;
    }
    st.pos = pos + (int64_t)G * (int64_t)total_w;
    st.tiles_done += 1;
    if (lg != 5u) {
        blk.half = half ^ 1u;
        blk.wd = wd;
        blk.md_raw = 0;
        blk.md = (uint64_t)md;
    }
    return t;
}

// The common case, on the scalar unit and in 32-bit arithmetic relative to the LDS
// window: a whole block (all 16 groups present in the stream, more values needed
// after it) whose minDelta varint has at most 4 bytes.  `o` is the window offset of
// the block header, `lim` the window offset of the end of the stream, p0 the
// position of the block's first value.  Returns false, changing nothing, whenever
// any of that does not hold; the caller then takes wave_walk_slow.
template <uint32_t LG>
__device__ __forceinline__ bool wave_walk_fast(WaveTile &t, const uint32_t *lds, uint32_t &o, uint32_t lim,
                                               uint32_t &p0, uint32_t need, WaveBlock &blk)
{
    constexpr uint32_t lg = LG;
    if (p0 + 128u >= need) return false; // the page's last tile takes the checked path
    if (lg != 5u && blk.half != 0u) {
        // second half of a 256-value block: miniblocks 2 and 3 under the header already parsed
        const uint32_t tw = blk.wd >> 16;
        const uint32_t next = o + 8u * ((tw * 0x01010101u) >> 24);
        if (next > lim) return false;
        t.valid = 1;
        t.p0 = p0;
        t.bitbase = o * 8u;
        t.widths = tw;
        t.prefix = tw * 0x01010100u;
        t.md_raw = blk.md_raw;
        t.hdr_lo = blk.hdr_lo;
        t.vl = blk.vl;
        t.min_delta = blk.md;
        blk.half = 0;
        o = next;
        p0 += 128u;
        return true;
    }
    const uint32_t k = o >> 2, s = (o & 3u) * 8u;
    const uint32_t d0 = uni(lds[k]), d1 = uni(lds[k + 1]), d2 = uni(lds[k + 2]);
    const uint32_t lo = (uint32_t)(((((uint64_t)d1) << 32) | d0) >> s);  // header bytes 0..3
    const uint32_t mid = (uint32_t)(((((uint64_t)d2) << 32) | d1) >> s); // header bytes 4..7
    const uint32_t stop = ~lo & 0x80808080u;                              // first byte without bit 7
    if (stop == 0u) return false;
    const uint32_t vl = ((uint32_t)__builtin_ctz(stop) + 1u) >> 3;       // varint length 1..4
    const uint32_t wd = (uint32_t)(((((uint64_t)mid) << 32) | lo) >> (8u * vl));
    if ((((wd & 0x7f7f7f7fu) + 0x5f5f5f5fu) | wd) & 0x80808080u) return false; // a width > 32
    const uint32_t tw = lg == 5u ? wd : wd & 0xffffu;                     // this tile's miniblocks
    const uint32_t total_w = (tw * 0x01010101u) >> 24;
    const uint32_t data = o + vl + 4u;
    const uint32_t next = data + ((total_w << lg) >> 3);
    if (next > lim) return false;
    t.valid = 1;
    t.p0 = p0;
    t.bitbase = data * 8u;
    t.widths = tw;
    t.prefix = tw * 0x01010100u;
    t.md_raw = 1;
    t.hdr_lo = lo;
    t.vl = vl;
    if (lg != 5u) {  // the second half of the block will want its header
        blk.half = 1u;
        blk.wd = wd;
        blk.md_raw = 1;
        blk.hdr_lo = lo;
        blk.vl = vl;
    }
    o = next;
    p0 += 128u;
    return true;
}

// Bytes the reference's reader would have consumed (see delta_consumed in delta_stream.hpp).
template <typename UT>
__device__ __forceinline__ uint64_t wave_consumed(const WaveState<UT> &st, int64_t len)
{
    int64_t pos = min(st.end_pos, len);
    if (st.status != PQ_OK || st.n_ok == 0) return (uint64_t)pos;
    const uint32_t last_group_pos = ((st.n_ok - 1u) >> 3) << 3;
    if (last_group_pos + 8u < st.total) return (uint64_t)pos;
    const uint32_t G = 1u << (st.lg - 3u);  // groups per miniblock
    const uint32_t gl = st.last_group, j = gl >> (st.lg - 3u);
    const int64_t wj = (st.last_widths >> (8u * j)) & 0xffu;
    pos = min(len, pos + (int64_t)(G - 1u - (gl & (G - 1u))) * wj);
    if (j + 1u < 4u) {
        const int64_t bw = (st.last_widths >> (8u * (j + 1u))) & 0xffu;
        for (uint32_t i = j + 1u; i < 4u; ++i) pos = min(len, pos + (int64_t)G * bw);
    }
    return (uint64_t)pos;
}

// Per-lane constants of the decode.
struct WaveLane {
    uint32_t sh8; // 8 * index (inside the tile) of the miniblock of this lane's 2 values
    uint32_t k2;  // index of the first of them inside the miniblock
};

// minDelta of a fast-walked block from its raw varint, on the vector unit (the scalar
// unit is the scarcer resource in this kernel; the asm move keeps the compiler from
// pulling the uniform arithmetic back onto it).
__device__ __forceinline__ int32_t wave_expand_min_delta(uint32_t hdr_lo, uint32_t vl)
{
    uint32_t v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(hdr_lo));
    v &= 0xffffffffu >> (32u - 8u * vl);
    const uint32_t x = (v & 0x7fu) | ((v >> 1) & 0x3f80u) | ((v >> 2) & 0x1fc000u) | ((v >> 3) & 0xfe00000u);
    return (int32_t)(x >> 1) ^ -(int32_t)(x & 1u);
}

// OFFS (uint32 values only): what goes to dst is not value i but the sum of the values in front of
// it — the values are the lengths of a DELTA_LENGTH_BYTE_ARRAY page and dst its offsets.  `sum` is
// the running total; `big` says that a value of 2^24 or more (a negative one included) or a total
// of 2^31 or more was seen, i.e. that the 32-bit sums may have wrapped and the caller has to take
// the lengths one by one.
struct WaveOffsets {
    uint32_t sum;
    uint32_t big;
};

// CLIPPED = false: a whole block, stored unconditionally.
template <typename UT, bool CLIPPED, uint32_t LG, bool OFFS = false>
__device__ __forceinline__ void wave_decode_tile(const uint32_t *lds, const WaveTile &t, UT &carry, UT *dst,
                                                 uint32_t clip, bool vec_ok, int lane, WaveLane ln,
                                                 WaveOffsets *wo = nullptr)
{
    const uint32_t w = __builtin_amdgcn_ubfe(t.widths, ln.sh8, 8u);
    const uint32_t pre = __builtin_amdgcn_ubfe(t.prefix, ln.sh8, 8u);
    const uint32_t bit = t.bitbase + (pre << LG) + ln.k2 * w;
    const uint32_t i = bit >> 5, s = bit & 31u;
    const uint32_t d0 = lds[i], d1 = lds[i + 1], d2 = lds[i + 2];
    const uint32_t lo = __builtin_amdgcn_alignbit(d1, d0, s);
    const uint32_t hi = __builtin_amdgcn_alignbit(d2, d1, s);
    const uint32_t m = (uint32_t)(1ull << w) - 1u; // w == 32 -> 0xffffffff, w == 0 -> 0
    const uint32_t ua = lo & m;
    const uint32_t ub = (uint32_t)((((uint64_t)hi << 32) | lo) >> w) & m;
    const uint32_t p = t.p0 + 2u * (uint32_t)lane;
    // every width <= 23 ?
    const bool thin = (((t.widths & 0x7f7f7f7fu) + 0x68686868u) & 0x80808080u) == 0u;
    UT v0, v1;
    if constexpr (sizeof(UT) == 8) {
        // With every width <= 23 and |minDelta| < 2^22 each delta lies in (-2^22, 1.5 * 2^23),
        // so all 128 in-block partial sums fit a signed 32-bit integer.  A raw minDelta of
        // at most 3 varint bytes is below 2^20 in magnitude.
        const bool narrow = thin && (t.md_raw ? t.vl <= 3u
                                              : ((int64_t)t.min_delta > -(1ll << 22) && (int64_t)t.min_delta < (1ll << 22)));
        if (narrow) {
            const int32_t md = t.md_raw ? wave_expand_min_delta(t.hdr_lo, t.vl) : (int32_t)t.min_delta;
            const int32_t da = (int32_t)ua + md;
            const int32_t pair = da + (int32_t)ub + md;
            const int32_t incl = (int32_t)wave_scan_u32((uint32_t)pair);
            v0 = carry + (UT)(int64_t)(incl - pair);
            v1 = v0 + (UT)(int64_t)da;
            carry += (UT)(int64_t)(int32_t)wave_last_u32((uint32_t)incl);
        } else {
            const UT md = t.md_raw ? (UT)(int64_t)wave_expand_min_delta(t.hdr_lo, t.vl) : (UT)t.min_delta;
            const UT a = (UT)ua + md;
            const UT pair = a + (UT)ub + md;
            const UT incl = wave_scan_u64(pair);
            v0 = carry + (incl - pair);
            v1 = v0 + a;
            carry += wave_last_u64(incl);
        }
        if ((!CLIPPED || p + 1u < clip) && vec_ok) {
            store_nt16(dst + p, (uint32_t)v0, (uint32_t)(v0 >> 32), (uint32_t)v1, (uint32_t)(v1 >> 32));
            return;
        }
    } else {
        const UT md = t.md_raw ? (UT)wave_expand_min_delta(t.hdr_lo, t.vl) : (UT)t.min_delta;
        const UT a = ua + md;
        const UT pair = a + ub + md;
        const UT incl = wave_scan_u32(pair);
        v0 = carry + (incl - pair);
        v1 = v0 + a;
        carry += wave_last_u32(incl);
        if constexpr (OFFS) {
            const uint32_t l0 = (!CLIPPED || p < clip) ? v0 : 0u, l1 = (!CLIPPED || p + 1u < clip) ? v1 : 0u;
            const uint32_t both = l0 + l1;
            const uint32_t oincl = wave_scan_u32(both);
            wo->big |= __builtin_amdgcn_ballot_w64((l0 | l1) >= (1u << 24)) != 0ull ? 1u : 0u;
            v0 = wo->sum + (oincl - both);
            v1 = v0 + l0;
            wo->sum += wave_last_u32(oincl);
            wo->big |= wo->sum >> 31;
        }
        if ((!CLIPPED || p + 1u < clip) && vec_ok) {
            store_nt8(dst + p, v0, v1);
            return;
        }
    }
    if (!CLIPPED || p < clip) dst[p] = v0;
    if (!CLIPPED || p + 1u < clip) dst[p + 1u] = v1;
}

// Two blocks of 4-byte values in the 128 / 4 layout at a time, entirely on the vector unit: lanes
// 0..31 take the first block, lanes 32..63 the one behind it, four adjacent values each, so that
// one prefix sum and one 16-byte store per lane serve 256 values.  (Not for 8-byte values: four
// adjacent ones per lane make every store instruction write half of each 64-byte line, which
// halves the rate the stores go out at, and at one 16-byte store per two values that decode is
// within a few percent of what the memory system takes anyway.)  The first header is
// parsed by every lane as far as the place the second one starts at; then each half parses its
// own header in full (minDelta of 1..4 varint bytes, widths, where the block ends) and unpacks.
// Like the single-block loop behind it, this one only takes blocks in the middle of a page that
// lie whole inside the window, and leaves everything else where it is: it returns at the first
// pair that is not of that kind, having advanced o (window offset of the next header), p0
// (position of its first value), carry (that value) and *wo.  stop_rel: a header at or behind
// this window offset is not to be decoded (segments), else 0x7fffffff.
// tab / tab_n / ti / tab_md0 / c0: blocks whose headers a walk in front of the decode has parsed
// (see delta_wave_tiles); pairs of them are taken from there first.
template <typename UT, bool OFFS>
__device__ __forceinline__ void delta_pair_loop(const uint32_t *lds, UT *dst, uint32_t &o, uint32_t &p0, UT &carry,
                                                uint32_t need, uint32_t lim, uint32_t o_max, int32_t stop_rel,
                                                WaveOffsets *wo, int lane, const uint32_t *tab = nullptr,
                                                uint32_t tab_n = 0, uint32_t tab_p0 = 0, int32_t tab_md0 = 0,
                                                uint32_t c0 = 0)
{
    static_assert(sizeof(UT) == 4, "four adjacent values per lane go out in one 16-byte store");
    const bool upper = lane >= 32;
    const uint32_t j = (uint32_t)lane & 31u;
    const uint32_t sh8 = (j >> 3) * 8u;  // 8 * the miniblock of this lane's 4 values
    const uint32_t k4 = 4u * (j & 7u);   // index of the first of them inside the miniblock
    // the two blocks of a pair behind their headers: widths, window offset of the packed data and
    // minDelta per half of the wave
    auto emit_pair = [&](uint32_t widths, uint32_t data, int32_t md) {
        const uint32_t w = __builtin_amdgcn_ubfe(widths, sh8, 8u);
        const uint32_t sum2 = widths + (widths << 8);
        const uint32_t sums = sum2 + (sum2 << 16);
        const uint32_t pre = __builtin_amdgcn_ubfe(sums << 8, sh8, 8u);
        const uint32_t bit = data * 8u + (pre << 5) + __umul24(k4, w);
        const uint32_t m = (uint32_t)(1ull << w) - 1u;  // w == 32 -> 0xffffffff, w == 0 -> 0
        uint32_t u[4];
#pragma unroll
        for (uint32_t q = 0; q < 4u; ++q) {
            const uint32_t bq = bit + q * w;
            const uint32_t i = bq >> 5;
            u[q] = __builtin_amdgcn_alignbit(lds[i + 1], lds[i], bq) & m;
        }
        UT v[4];
        const UT s1 = u[0] + (UT)md, s2 = s1 + u[1] + (UT)md, s3 = s2 + u[2] + (UT)md, s4 = s3 + u[3] + (UT)md;
        const UT incl = wave_scan_u32(s4);
        v[0] = carry + (incl - s4);
        v[1] = v[0] + s1;
        v[2] = v[0] + s2;
        v[3] = v[0] + s3;
        carry += wave_last_u32(incl);
        if constexpr (OFFS) {
            // the values are lengths: what is stored is the sum of those in front.  256 lengths
            // below 2^23 on top of a total below 2^31 cannot wrap.
            const uint32_t t1 = v[0], t2 = t1 + v[1], t3 = t2 + v[2], t4 = t3 + v[3];
            const uint32_t oincl = wave_scan_u32(t4);
            wo->big |= __builtin_amdgcn_ballot_w64((v[0] | v[1] | v[2] | v[3]) >= (1u << 23)) != 0ull ? 1u : 0u;
            v[0] = wo->sum + (oincl - t4);
            v[1] = v[0] + t1;
            v[2] = v[0] + t2;
            v[3] = v[0] + t3;
            wo->sum += wave_last_u32(oincl);
            wo->big |= wo->sum >> 31;
        }
        store_nt16(dst + p0 + 4u * (uint32_t)lane, v[0], v[1], v[2], v[3]);
    };
    if (tab != nullptr) {
        // entry i is the block whose first value has position tab_p0 + 128 i: whichever loop took the
        // blocks in front, the table is read at the block that is next
        while (((p0 - tab_p0) >> 7) + 1u < tab_n && p0 + 256u < need) {
            const uint32_t ti = (p0 - tab_p0) >> 7;
            const uint32_t e = 4u * (ti + (upper ? 1u : 0u));
            const uint32_t data = tab[e] - c0, widths = tab[e + 1u];
            const int32_t md = (ti == 0u && !upper) ? tab_md0 : (int32_t)tab[e + 2u];
            const uint32_t next = data + 4u * __builtin_amdgcn_sad_u8(widths, 0u, 0u);
            const uint32_t next_b = (uint32_t)__builtin_amdgcn_readlane((int)next, 63);
            if (next_b > lim) break;  // the pair does not lie wholly in this window
            emit_pair(widths, data, md);
            o = next_b;
            p0 += 256u;
        }
    }
    while (o <= o_max && p0 + 256u < need && (int32_t)o < stop_rel) {
        // where the second header starts: the first one's varint length and the sum of its widths
        uint32_t next_a;
        {
            const uint32_t k = o >> 2;
            const uint32_t d0 = lds[k], d1 = lds[k + 1], d2 = lds[k + 2];
            const uint32_t lo = __builtin_amdgcn_alignbit(d1, d0, o * 8u);
            const uint32_t mid = __builtin_amdgcn_alignbit(d2, d1, o * 8u);
            const uint32_t end = ~lo & 0x80808080u;
            const uint32_t vl = ((uint32_t)__builtin_ctz(end | 0x80000000u) + 1u) >> 3;
            const uint32_t widths = (uint32_t)(((((uint64_t)mid) << 32) | lo) >> (8u * vl));
            next_a = o + vl + 4u + 4u * __builtin_amdgcn_sad_u8(widths, 0u, 0u);
        }
        // a first header that is not a plain one sends the second half anywhere: keep its reads inside the window
        const uint32_t ho = upper ? min(next_a, (uint32_t)kDwChunk) : o;
        const uint32_t k = ho >> 2;
        const uint32_t d0 = lds[k], d1 = lds[k + 1], d2 = lds[k + 2];
        const uint32_t hdr_lo = __builtin_amdgcn_alignbit(d1, d0, ho * 8u);
        const uint32_t mid = __builtin_amdgcn_alignbit(d2, d1, ho * 8u);
        const uint32_t end = ~hdr_lo & 0x80808080u;
        const uint32_t vl = ((uint32_t)__builtin_ctz(end | 0x80000000u) + 1u) >> 3;  // 1..4
        const uint32_t widths = (uint32_t)(((((uint64_t)mid) << 32) | hdr_lo) >> (8u * vl));
        const uint32_t data = ho + vl + 4u;
        const uint32_t next = data + 4u * __builtin_amdgcn_sad_u8(widths, 0u, 0u);
        // minDelta of five bytes and more, a width above 32, a block that is not all in the window, a
        // second block at or behind the stop
        const bool bad = end == 0u || ((((widths & 0x7f7f7f7fu) + 0x5f5f5f5fu) | widths) & 0x80808080u) != 0u || next > lim ||
                         (upper && (int32_t)next_a >= stop_rel);
        if (__builtin_amdgcn_ballot_w64(bad) != 0ull) break;

        // minDelta from its raw varint
        const uint32_t mv = hdr_lo & (0xffffffffu >> (32u - 8u * vl));
        const uint32_t mx = (mv & 0x7fu) | ((mv >> 1) & 0x3f80u) | ((mv >> 2) & 0x1fc000u) | ((mv >> 3) & 0xfe00000u);
        const int32_t md = (int32_t)(mx >> 1) ^ -(int32_t)(mx & 1u);
        emit_pair(widths, data, md);
        o = (uint32_t)__builtin_amdgcn_readlane((int)next, 63);
        p0 += 256u;
    }
}

// A block header whose minDelta varint has five to ten bytes (2^27 and more in size), put together
// on the scalar unit: hdr_lo / mid = bytes 0..3 / 4..7 of the header, found at byte shift sh of LDS
// word k, h2 = word k + 2.  false: no end within ten bytes, or more than 64 bits (errors the
// checked path reports).  widths = the four bytes behind the varint.
__device__ __forceinline__ bool wave_long_min_delta(const uint32_t *lds, uint32_t k, uint32_t sh, uint32_t hdr_lo, uint32_t mid,
                                                    uint32_t h2, uint32_t &vl, uint32_t &widths, int64_t &md)
{
    const uint32_t h3 = uni(lds[k + 3]), h4 = uni(lds[k + 4]);
    const uint32_t hi = (uint32_t)(((((uint64_t)h3) << 32) | h2) >> sh);    // bytes 8..11
    const uint32_t top = (uint32_t)(((((uint64_t)h4) << 32) | h3) >> sh);  // bytes 12..15
    const uint32_t end2 = ~mid & 0x80808080u, end3 = ~hi & 0x00008080u;
    vl = 4u;
    widths = 0u;
    md = 0;
    if (end2 == 0u && end3 == 0u) return false;
    vl = end2 != 0u ? 4u + (((uint32_t)__builtin_ctz(end2) + 1u) >> 3) : 8u + (((uint32_t)__builtin_ctz(end3) + 1u) >> 3);  // 5..10
    auto seven = [](uint32_t v) -> uint64_t {
        return (v & 0x7fu) | ((v >> 1) & 0x3f80u) | ((v >> 2) & 0x1fc000u) | ((v >> 3) & 0xfe00000u);
    };
    const uint32_t mid_v = vl >= 8u ? mid : mid & (0xffffffffu >> (64u - 8u * vl));
    const uint32_t hi_v = vl > 8u ? hi & (0xffffu >> (80u - 8u * vl)) : 0u;
    if (vl == 10u && (hi_v >> 8) > 1u) return false;
    const uint64_t zz = seven(hdr_lo) | (seven(mid_v) << 28) | (seven(hi_v) << 56);
    md = (int64_t)(zz >> 1) ^ -(int64_t)(zz & 1ull);
    // the width bytes behind the varint: bytes vl .. vl + 3 of the sixteen at hand
    const uint64_t lo64 = (((uint64_t)mid) << 32) | hdr_lo, hi64 = (((uint64_t)top) << 32) | hi;
    const uint32_t sv = 8u * vl;  // 40..80
    widths = sv < 64u ? (uint32_t)((lo64 >> sv) | (hi64 << (64u - sv))) : (uint32_t)(hi64 >> (sv - 64u));
    return true;
}

struct WaveDeltaResult {
    int32_t status; // PQ_*, or kNeedGeneric when the layout is not 128 / 4
    uint32_t n_decoded;
    uint64_t consumed;
    uint32_t init_failed;
    uint32_t total;  // the value count the stream header declares (0 when the header failed)
};

// The tiles of one page behind its stream header (pos = the first block header, first = the first
// value, the window [c0, c0 + kDwChunk) staged).  LG = log2 of the values per miniblock.
// SEG: the decode starts in the middle of the page, at the block header at `pos` whose first value
// has position seg_p0 (a multiple of the block size) and is `first`, and ends in front of the
// block header at stream offset seg_stop (status kSegStopped; < 0: runs to the end of the page
// like the whole-page decode, tail rules, status and consumed bytes included).  tab / tab_n: the
// segment's first tab_n blocks as the walk in front of the decode found them, four words each —
// stream offset of the block's packed data, its four width bytes, its minDelta (two words) —
// so that the decode does not parse their headers again; tab_md0 is the first block's minDelta
// (the walk may not know it).
template <typename UT, uint32_t LG, bool OFFS = false, bool SEG = false>
__device__ __forceinline__ WaveDeltaResult delta_wave_tiles(uint32_t *lds, const uint8_t *S, int64_t len, uint32_t n,
                                                            UT *dst, int lane, WavePrefetch &pf, int64_t c0, int64_t pos,
                                                            uint32_t total32, uint64_t first, bool vec_ok,
                                                            WaveOffsets *wo = nullptr, uint32_t seg_p0 = 0,
                                                            int64_t seg_stop = -1, const uint32_t *tab = nullptr,
                                                            uint32_t tab_n = 0, int64_t tab_md0 = 0)
{
    WaveDeltaResult res{PQ_OK, 0, 0, 0, total32};
    WaveState<UT> st;
    st.pos = pos;
    st.end_pos = pos;
    st.tiles_done = SEG ? seg_p0 >> 7 : 0u;
    st.total = total32;
    st.need = min(n, total32);
    st.n_ok = 0;
    st.status = PQ_OK;
    st.finished = 0;
    st.init_failed = 0;
    st.last_widths = 0;
    st.last_group = 0;
    st.lg = LG;
    WaveBlock blk{0, 0, 0, 0, 0, 0};
    const uint8_t *B = reinterpret_cast<const uint8_t *>(lds);

    if (st.need == 0) {
        // Init still reads the first block header eagerly (:198-204, :274-280); only
        // its header errors count, and the reader stops right behind the width bytes
        st.need = 1;
        (void)wave_walk_slow<UT, LG>(st, blk, B, c0, len);
        res.init_failed = uni(st.init_failed);
        res.status = res.init_failed ? (int)uni((uint32_t)st.status) : PQ_OK;
        int64_t p2 = pos;
        uint64_t z;
        if (!res.init_failed) {
            (void)read_uvarint(B, c0, len, &p2, &z);
            p2 += 4;
        }
        res.consumed = (uint64_t)uni((uint32_t)p2);
        return res;
    }

    const bool vec16 = (reinterpret_cast<uintptr_t>(dst) & 15u) == 0;  // 4 x int32 or 2 x int64 in one store
    WaveLane ln;
    ln.sh8 = ((2u * (uint32_t)lane) >> LG) * 8u;
    ln.k2 = (2u * (uint32_t)lane) & ((1u << LG) - 1u);
    UT carry = (UT)first;
    uint32_t o = (uint32_t)(pos - c0); // window offset of the next block header
    uint32_t p0 = SEG ? seg_p0 : 0u;   // position of that block's first value
    bool stopped = false;
    while (true) {
        // The window [c0, c0 + kDwChunk) is in LDS.  Windows advance by a fixed stride
        // (a block never needs more than kDwBlockMax bytes, so whatever block straddles
        // the end of this window starts inside the next one): the next window's loads
        // do not depend on the walk and are issued before it, a whole window ahead.
        const int64_t c1 = c0 + kDwStride;
        const bool tail = c0 + kDwChunk >= len; // this window reaches the end of the stream
        if (!tail) wave_fetch(pf, S, c1, len, lane);
        const int64_t left = len - c0;
        const uint32_t lim = left > (int64_t)kDwChunk ? (uint32_t)kDwChunk : (left < 0 ? 0u : (uint32_t)left);
        while (true) {
            {
                // The blocks in the middle of a page — whole, plain, more values wanted behind them —
                // in a loop of their own with one way out: header walk and decode back to back, no
                // state handed from one to the other.  Whatever ends it (the page's last block, a
                // block the window cuts, an unusual header, the segment's end) is left as it is for
                // the checked code below, which takes one tile and comes back here.  A block of 256
                // values is two tiles under one header.
                constexpr uint32_t kTiles = 1u << (LG - 5u), kBlockValues = 128u * kTiles;
                const uint32_t o_max = tail ? 0x7fffffffu : (uint32_t)(kDwChunk - kTiles * kDwBlockMax);
                int32_t stop_rel = 0x7fffffff;
                if constexpr (SEG) {
                    if (seg_stop >= 0 && seg_stop - c0 < (int64_t)0x7fffffff) stop_rel = (int32_t)(seg_stop - c0);
                }
                if constexpr (LG == 5u && sizeof(UT) == 4) {
                    // pairs of blocks first; what ends the pair loop is one of the single-block loop's cases
                    if (vec16)
                        delta_pair_loop<UT, OFFS>(lds, dst, o, p0, carry, st.need, lim,
                                                  tail ? 0x7fffffffu : (uint32_t)(kDwChunk - 2 * kDwBlockMax), stop_rel, wo, lane,
                                                  SEG ? tab : nullptr, tab_n, seg_p0, (int32_t)tab_md0, (uint32_t)c0);
                }
                // one block behind its header: widths (a byte per miniblock), window offset of its data, minDelta
                auto emit_block = [&](uint32_t widths, uint32_t data, int64_t md, uint32_t vl) {
                        // every width <= 23 and a minDelta of at most three bytes: a tile's partial sums
                        // fit 32 bits (see wave_decode_tile)
                        const bool narrow = (((widths & 0x7f7f7f7fu) + 0x68686868u) & 0x80808080u) == 0u && vl <= 3u;
#pragma unroll
                        for (uint32_t tile = 0; tile < kTiles; ++tile) {
                            // this tile's miniblocks: all four widths, or two of them
                            const uint32_t tw = LG == 5u ? widths : (tile == 0u ? widths & 0xffffu : widths >> 16);
                            const uint32_t bitbase = data * 8u + (tile == 0u ? 0u : 64u * (((widths & 0xffffu) * 0x0101u >> 8) & 0xffu));
                            const uint32_t w = __builtin_amdgcn_ubfe(tw, ln.sh8, 8u);
                            const uint32_t pre = __builtin_amdgcn_ubfe(tw * 0x01010100u, ln.sh8, 8u);
                            const uint32_t bit = bitbase + (pre << LG) + ln.k2 * w;
                            const uint32_t i = bit >> 5, sft = bit & 31u;
                            const uint32_t e0 = lds[i], e1 = lds[i + 1], e2 = lds[i + 2];
                            const uint32_t l32 = __builtin_amdgcn_alignbit(e1, e0, sft);
                            const uint32_t h32 = __builtin_amdgcn_alignbit(e2, e1, sft);
                            const uint32_t m = (uint32_t)(1ull << w) - 1u;
                            const uint32_t ua = l32 & m;
                            const uint32_t ub = (uint32_t)((((uint64_t)h32 << 32) | l32) >> w) & m;
                            UT *const q = dst + p0 + 128u * tile + 2u * (uint32_t)lane;
                            if constexpr (sizeof(UT) == 8) {
                                UT v0, v1;
                                if (narrow) {
                                    const int32_t da = (int32_t)ua + (int32_t)md;
                                    const int32_t pair = da + (int32_t)ub + (int32_t)md;
                                    const int32_t incl = (int32_t)wave_scan_u32((uint32_t)pair);
                                    v0 = carry + (UT)(int64_t)(incl - pair);
                                    v1 = v0 + (UT)(int64_t)da;
                                    carry += (UT)(int64_t)(int32_t)wave_last_u32((uint32_t)incl);
                                } else {
                                    const UT a = (UT)ua + (UT)md;
                                    const UT pair = a + (UT)ub + (UT)md;
                                    const UT incl = wave_scan_u64(pair);
                                    v0 = carry + (incl - pair);
                                    v1 = v0 + a;
                                    carry += wave_last_u64(incl);
                                }
                                if (vec_ok) {
                                    store_nt16(q, (uint32_t)v0, (uint32_t)(v0 >> 32), (uint32_t)v1, (uint32_t)(v1 >> 32));
                                } else {
                                    q[0] = v0;
                                    q[1] = v1;
                                }
                            } else {
                                const UT a = ua + (UT)md;
                                const UT pair = a + ub + (UT)md;
                                const UT incl = wave_scan_u32(pair);
                                UT v0 = carry + (incl - pair);
                                UT v1 = v0 + a;
                                carry += wave_last_u32(incl);
                                if constexpr (OFFS) {
                                    // the values are lengths: what is stored is the sum of those in front
                                    const uint32_t l0 = v0, both = v0 + v1;
                                    const uint32_t oincl = wave_scan_u32(both);
                                    wo->big |= __builtin_amdgcn_ballot_w64((v0 | v1) >= (1u << 24)) != 0ull ? 1u : 0u;
                                    v0 = wo->sum + (oincl - both);
                                    v1 = v0 + l0;
                                    wo->sum += wave_last_u32(oincl);
                                    wo->big |= wo->sum >> 31;
                                }
                                if (vec_ok) {
                                    store_nt8(q, v0, v1);
                                } else {
                                    q[0] = v0;
                                    q[1] = v1;
                                }
                            }
                        }
                };
                if constexpr (SEG && LG == 5u && sizeof(UT) == 8) {
                    // blocks whose headers the segment's walk has parsed already
                    // (entry i is the block whose first value has position seg_p0 + 128 i)
                    while (((p0 - seg_p0) >> 7) < tab_n && p0 + 128u < st.need) {
                        const uint32_t ti = (p0 - seg_p0) >> 7;
                        const uint32_t e0 = uni(tab[4u * ti]), e1 = uni(tab[4u * ti + 1u]);
                        const uint32_t data = e0 - (uint32_t)c0;
                        const uint32_t next = data + 4u * __builtin_amdgcn_sad_u8(e1, 0u, 0u);
                        if (next > lim) break;  // not wholly in this window
                        const int64_t md = ti == 0u ? tab_md0
                                                    : (int64_t)(((uint64_t)uni(tab[4u * ti + 3u]) << 32) | uni(tab[4u * ti + 2u]));
                        emit_block(e1, data, md, md >= -(1ll << 20) && md < (1ll << 20) ? 3u : 4u);
                        o = next;
                        p0 += 128u;
                    }
                }
                if constexpr (LG == 5u && sizeof(UT) == 8) {
                    // Two blocks an iteration: the lower half of the wave reads the first header, the upper
                    // half the second (which starts where the first says), all in vector registers; what
                    // the decode wants as scalars comes over with a handful of v_readlane.  The same
                    // walk on the scalar unit, one header at a time, was more than half of a block's
                    // instructions.
                    const bool upper = lane >= 32;
                    const uint32_t o_max2 = tail ? 0x7fffffffu : (uint32_t)(kDwChunk - 2 * kDwBlockMax);
                    while (o <= o_max2 && p0 + 256u < st.need && (int32_t)o < stop_rel) {
                        uint32_t next_a;
                        {
                            const uint32_t k = o >> 2;
                            const uint32_t d0 = lds[k], d1 = lds[k + 1], d2 = lds[k + 2];
                            const uint32_t lo = __builtin_amdgcn_alignbit(d1, d0, o * 8u);
                            const uint32_t mid = __builtin_amdgcn_alignbit(d2, d1, o * 8u);
                            const uint32_t end = ~lo & 0x80808080u;
                            const uint32_t vl = ((uint32_t)__builtin_ctz(end | 0x80000000u) + 1u) >> 3;
                            const uint32_t widths = (uint32_t)(((((uint64_t)mid) << 32) | lo) >> (8u * vl));
                            next_a = o + vl + 4u + 4u * __builtin_amdgcn_sad_u8(widths, 0u, 0u);
                        }
                        // a first header that is not a plain one sends the second half anywhere: keep its reads inside the window
                        const uint32_t ho = upper ? min(next_a, (uint32_t)kDwChunk) : o;
                        const uint32_t k = ho >> 2;
                        const uint32_t d0 = lds[k], d1 = lds[k + 1], d2 = lds[k + 2];
                        const uint32_t hdr_lo = __builtin_amdgcn_alignbit(d1, d0, ho * 8u);
                        const uint32_t mid = __builtin_amdgcn_alignbit(d2, d1, ho * 8u);
                        const uint32_t end = ~hdr_lo & 0x80808080u;
                        const uint32_t vl = ((uint32_t)__builtin_ctz(end | 0x80000000u) + 1u) >> 3;  // 1..4
                        const uint32_t widths = (uint32_t)(((((uint64_t)mid) << 32) | hdr_lo) >> (8u * vl));
                        const uint32_t data = ho + vl + 4u;
                        const uint32_t next = data + 4u * __builtin_amdgcn_sad_u8(widths, 0u, 0u);
                        const bool bad = end == 0u || ((((widths & 0x7f7f7f7fu) + 0x5f5f5f5fu) | widths) & 0x80808080u) != 0u ||
                                         next > lim || (upper && (int32_t)next_a >= stop_rel);
                        if (__builtin_amdgcn_ballot_w64(bad) != 0ull) break;
                        const int32_t md = wave_expand_min_delta(hdr_lo, vl);
                        emit_block((uint32_t)__builtin_amdgcn_readlane((int)widths, 0), (uint32_t)__builtin_amdgcn_readlane((int)data, 0),
                                   __builtin_amdgcn_readlane(md, 0), (uint32_t)__builtin_amdgcn_readlane((int)vl, 0));
                        p0 += 128u;
                        emit_block((uint32_t)__builtin_amdgcn_readlane((int)widths, 32), (uint32_t)__builtin_amdgcn_readlane((int)data, 32),
                                   __builtin_amdgcn_readlane(md, 32), (uint32_t)__builtin_amdgcn_readlane((int)vl, 32));
                        p0 += 128u;
                        o = (uint32_t)__builtin_amdgcn_readlane((int)next, 32);
                    }
                }
                while (o <= o_max && p0 + kBlockValues < st.need && (int32_t)o < stop_rel && (LG == 5u || blk.half == 0u)) {
                    const uint32_t k = o >> 2, sh = (o & 3u) * 8u;
                    const uint32_t h0 = uni(lds[k]), h1 = uni(lds[k + 1]), h2 = uni(lds[k + 2]);
                    const uint32_t hdr_lo = (uint32_t)(((((uint64_t)h1) << 32) | h0) >> sh);
                    const uint32_t mid = (uint32_t)(((((uint64_t)h2) << 32) | h1) >> sh);
                    const uint32_t end = ~hdr_lo & 0x80808080u;
                    if constexpr (sizeof(UT) == 8) {
                        if (end == 0u) {
                            // A minDelta of 2^27 and more (a column of nanosecond timestamps a second apart)
                            uint32_t vl, widths;
                            int64_t md;
                            if (!wave_long_min_delta(lds, k, sh, hdr_lo, mid, h2, vl, widths, md)) break;  // the checked path says what it is
                            const uint32_t data = o + vl + 4u;
                            const uint32_t next = data + ((((widths * 0x01010101u) >> 24) << LG) >> 3);
                            if (((((widths & 0x7f7f7f7fu) + 0x5f5f5f5fu) | widths) & 0x80808080u) != 0u || next > lim) break;
                            emit_block(widths, data, md, vl);
                            o = next;
                            p0 += kBlockValues;
                            continue;
                        }
                    }
                    const uint32_t vl = ((uint32_t)__builtin_ctz(end | 0x80000000u) + 1u) >> 3;  // 1..4
                    const uint32_t widths = (uint32_t)(((((uint64_t)mid) << 32) | hdr_lo) >> (8u * vl));
                    const uint32_t data = o + vl + 4u;
                    const uint32_t next = data + ((((widths * 0x01010101u) >> 24) << LG) >> 3);
                    // minDelta of five bytes and more, a width above 32, a block that is not all in the window
                    if (end == 0u || ((((widths & 0x7f7f7f7fu) + 0x5f5f5f5fu) | widths) & 0x80808080u) != 0u || next > lim)
                        break;
                    const int32_t md = wave_expand_min_delta(hdr_lo, vl);
                    emit_block(widths, data, md, vl);
                    o = next;
                    p0 += kBlockValues;
                }
            }
            if (!(!st.finished && (tail || o + (uint32_t)kDwBlockMax <= (uint32_t)kDwChunk))) break;
            if constexpr (SEG) {
                if (seg_stop >= 0 && (LG == 5u || blk.half == 0u) && c0 + (int64_t)o >= seg_stop) {
                    stopped = true;
                    break;
                }
            }
            WaveTile t = wave_tile_none();
            if (wave_walk_fast<LG>(t, lds, o, lim, p0, st.need, blk)) {
                wave_decode_tile<UT, false, LG, OFFS>(lds, t, carry, dst, 0xffffffffu, vec_ok, lane, ln, wo);
                continue;
            }
            st.pos = c0 + (int64_t)o;
            st.tiles_done = p0 >> 7;
            t = wave_walk_slow<UT, LG>(st, blk, B, c0, len);
            blk.half = uni(blk.half);
            blk.wd = uni(blk.wd);
            blk.md = ((uint64_t)uni((uint32_t)(blk.md >> 32)) << 32) | uni((uint32_t)blk.md);
            // values the checked path read back from LDS are uniform too
            o = uni((uint32_t)(st.pos - c0));
            p0 = uni(st.tiles_done) * 128u;
            st.finished = uni(st.finished);
            st.status = (int32_t)uni((uint32_t)st.status);
            st.n_ok = uni(st.n_ok);
            t.valid = uni(t.valid);
            t.bitbase = uni(t.bitbase);
            t.widths = uni(t.widths);
            t.prefix = uni(t.prefix);
            t.p0 = uni(t.p0);
            t.min_delta = ((uint64_t)uni((uint32_t)(t.min_delta >> 32)) << 32) | uni((uint32_t)t.min_delta);
            const uint32_t clip = st.finished ? st.n_ok : 0xffffffffu;
            if (t.valid) wave_decode_tile<UT, true, LG, OFFS>(lds, t, carry, dst, clip, vec_ok, lane, ln, wo);
        }
        if (st.finished || stopped) break;
        wave_sync();
        wave_commit(lds, pf, lane);
        wave_sync();
        c0 = c1;
        o -= (uint32_t)kDwStride;
    }

    if (SEG && stopped) {
        res.status = kSegStopped;
        return res;
    }
    res.status = st.status;
    res.n_decoded = st.n_ok;
    res.init_failed = st.init_failed;
    if (st.status == PQ_OK && n > st.total) res.status = PQ_EOF; // pulled past the declared count
    res.consumed = wave_consumed<UT>(st, len);
    return res;
}

// Decode one page with one wave.  `lds` is this wave's private window.  A stream that
// declares more than `cap` values is not decoded at all (status kOverCap).  WIDE = false: only
// the 128 / 4 layout is taken (for callers whose streams never use the other one).
template <typename UT, bool WIDE = true, bool OFFS = false>
__device__ __forceinline__ WaveDeltaResult delta_wave(uint32_t *lds, const uint8_t *S, int64_t len, uint32_t n,
                                                      UT *dst, int lane, uint32_t cap = 0xffffffffu,
                                                      WaveOffsets *wo = nullptr)
{
    static_assert(!OFFS || (sizeof(UT) == 4 && !WIDE), "offsets come from 32-bit lengths in the 128 / 4 layout");
    const uint8_t *B = reinterpret_cast<const uint8_t *>(lds);
    const bool vec_ok = (reinterpret_cast<uintptr_t>(dst) & (2 * sizeof(UT) - 1)) == 0;
    WaveDeltaResult res{PQ_OK, 0, 0, 0, 0};
    WavePrefetch pf;

    int64_t c0 = chunk_base(S, 0);
    wave_fetch(pf, S, c0, len, lane);
    wave_commit(lds, pf, lane);
    wave_sync();

    // ---- stream header (readBlockHeader, :29-65) ----
    int64_t pos = 0;
    uint64_t u;
    int hs = read_uvarint(B, c0, len, &pos, &u);
    if (hs == PQ_OK && u > 0x7fffffffull) hs = PQ_INT32_OUT_OF_RANGE;
    if (hs == PQ_OK && (u == 0 || (u % 128u) != 0)) hs = PQ_INVALID_BLOCK_SIZE;
    const uint64_t bs = u;
    uint64_t nm = 0;
    if (hs == PQ_OK) {
        hs = read_uvarint(B, c0, len, &pos, &nm);
        if (hs == PQ_OK && nm > 0x7fffffffull) hs = PQ_INT32_OUT_OF_RANGE;
        if (hs == PQ_OK && (nm == 0 || (bs % nm) != 0)) hs = PQ_INVALID_MINIBLOCK_COUNT;
    }
    uint64_t total = 0;
    if (hs == PQ_OK) {
        hs = read_uvarint(B, c0, len, &pos, &total);
        if (hs == PQ_OK && total > 0x7fffffffull) hs = PQ_INT32_OUT_OF_RANGE;
    }
    uint64_t first = 0;
    if (hs == PQ_OK && total != 0) {
        hs = read_uvarint(B, c0, len, &pos, &u);
        if (hs == PQ_OK) {
            const int64_t f = unzigzag(u);
            if (sizeof(UT) == 4 && (f > 2147483647ll || f < -2147483648ll)) hs = PQ_INT32_OUT_OF_RANGE;
            first = (uint64_t)f;
        }
    }
    // everything parsed from the window is the same in every lane: say so
    hs = (int)uni((uint32_t)hs);
    pos = (int64_t)uni((uint32_t)pos);
    if (hs != PQ_OK) {
        res.status = hs;
        res.init_failed = 1;
        res.consumed = (uint64_t)pos;
        return res;
    }
    const uint32_t total32 = uni((uint32_t)total);
    res.total = total32;
    if (total32 > cap) {
        res.status = kOverCap;
        return res;
    }
    if (total32 == 0) { // every pull is io.EOF (:97-100)
        res.status = n > 0 ? PQ_EOF : PQ_OK;
        res.consumed = (uint64_t)pos;
        return res;
    }
    // tiles of 128 values: blocks of 128 with 4 miniblocks of 32, or blocks of 256 with 4 of 64
    const uint32_t bs32 = uni((uint32_t)bs);
    if ((bs32 != 128u && !(WIDE && bs32 == 256u)) || uni((uint32_t)nm) != 4u) {
        res.status = kNeedGeneric;
        return res;
    }
    first = ((uint64_t)uni((uint32_t)(first >> 32)) << 32) | uni((uint32_t)first);
    if constexpr (WIDE) {
        if (bs32 == 256u) return delta_wave_tiles<UT, 6>(lds, S, len, n, dst, lane, pf, c0, pos, total32, first, vec_ok);
    }
    return delta_wave_tiles<UT, 5, OFFS>(lds, S, len, n, dst, lane, pf, c0, pos, total32, first, vec_ok, wo);
}

} // namespace pq
