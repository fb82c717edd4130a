// delta_wave.hpp — one WAVE per page DELTA_BINARY_PACKED decoder for the layout every
// mainstream writer emits: 128-value blocks of 4 miniblocks (32 values each).
//
// Same value sequence as delta_stream.hpp (reference:
// encoding/delta-binary-packed-decoder.go:29-170,213-225,289-301), organised so that
// nothing ever waits on another wave: each wave owns a page, a private 3 KiB LDS
// window and its running value.  Per iteration a wave
//   1. walks 4 block headers from its LDS window (zig-zag minDelta, 4 width bytes;
//      wave-uniform arithmetic: prefix of widths by one multiply),
//   2. issues the coalesced 16-byte loads of the NEXT window into registers,
//   3. decodes the 4 blocks: each lane unpacks 2 adjacent deltas with a 96-bit
//      funnel read, a DPP inclusive scan over the lane sums gives the in-block
//      prefix, the running value is added and 2 x int64 go out in one 16-byte store,
//   4. commits the prefetched registers to LDS.
// Pages with any other layout are handed to the workgroup-per-page kernel.
#pragma once

#include "pq_device.hpp"

namespace pq {

constexpr int kDwChunk = 3072;  // bytes staged per wave and iteration
constexpr int kDwTiles = 4;     // blocks decoded per iteration: 15 + 4 * (10 + 4 + 512) <= kDwChunk
constexpr int kDwPieces = kDwChunk / 16 / kWave; // 16-byte pieces per lane
constexpr int32_t kNeedGeneric = -2;

__device__ __forceinline__ void wave_sync()
{
    // orders this wave's LDS traffic for the compiler; DS operations of one wave
    // execute in order, so no hardware barrier is needed
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

struct WavePrefetch {
    uint4 v[kDwPieces];
};

__device__ __forceinline__ void wave_fetch(WavePrefetch &pf, const uint8_t *S, int64_t c0, int64_t len, int lane)
{
#pragma unroll
    for (int k = 0; k < kDwPieces; ++k) {
        const int64_t o = c0 + 16 * (int64_t)(lane + kWave * k);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (o + 16 > 0 && o < len) {
            v = *reinterpret_cast<const uint4 *>(S + o);
            if (o < 0 || o + 16 > len) v = mask_piece(v, -o, len - o);
        }
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
    uint32_t valid;   // 1 = decode this tile
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
    uint32_t last_group;  // index of that group inside its block (0..15)
};

// Walk one block header at st.pos.  B = LDS window bytes, c0 = its stream offset.
template <typename UT>
__device__ __forceinline__ WaveTile wave_walk(WaveState<UT> &st, const uint8_t *B, int64_t c0, int64_t len)
{
    WaveTile t;
    t.valid = 0;
    t.bitbase = 0;
    t.widths = 0;
    t.prefix = 0;
    t.min_delta = 0;
    if (st.finished) return t;
    const uint32_t p0 = st.tiles_done * 128u;
    if (p0 >= st.need) {
        st.finished = 1;
        st.n_ok = st.need;
        return t;
    }
    int64_t pos = st.pos;
    uint64_t u;
    int hs = read_uvarint(B, c0, len, &pos, &u);
    int64_t md = 0;
    if (hs == PQ_OK) {
        md = unzigzag(u);
        if (sizeof(UT) == 4 && (md > 2147483647ll || md < -2147483648ll)) hs = PQ_INT32_OUT_OF_RANGE;
    }
    uint32_t wd = 0;
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
    const uint32_t prefix = wd * 0x01010100u;           // bytes: 0, w0, w0+w1, w0+w1+w2
    const uint32_t total_w = (wd * 0x01010101u) >> 24;  // w0+w1+w2+w3 <= 128
    const uint32_t tile_vals = min(128u, st.need - p0);
    const uint32_t gn = (tile_vals + 7u) >> 3;          // groups that will be read
    const int64_t avail = len - pos;
    uint32_t ok_groups = gn;
    if ((int64_t)(4u * total_w) > avail || gn < 16u) {
        // the block is the stream's last or is cut short: io.ReadFull per group (:111-114)
        int64_t off = 0;
        for (uint32_t j = 0; j < 4u; ++j) {
            const uint32_t w = (wd >> (8u * j)) & 0xffu;
            if (gn <= 4u * j) break;
            const uint32_t gj = min(4u, gn - 4u * j);
            if (off + (int64_t)(gj * w) > avail) {
                const int64_t rem = avail > off ? avail - off : 0;
                const uint32_t gok = w ? (uint32_t)(rem / w) : gj;
                ok_groups = 4u * j + gok;
                st.status = (rem - (int64_t)gok * w) == 0 ? PQ_EOF : PQ_UNEXPECTED_EOF;
                break;
            }
            off += 4 * (int64_t)w;
        }
    }
    t.valid = 1;
    t.bitbase = (uint32_t)(pos - c0) * 8u;
    t.widths = wd;
    t.prefix = prefix;
    t.min_delta = (uint64_t)md;
    if (ok_groups > 0) {
        const uint32_t gl = ok_groups - 1u; // last group read in this block
        const uint32_t j = gl >> 2;
        const uint32_t wj = (wd >> (8u * j)) & 0xffu;
        st.end_pos = pos + 4 * (int64_t)((prefix >> (8u * j)) & 0xffu) + (int64_t)((gl & 3u) + 1u) * wj;
        st.last_widths = wd;
        st.last_group = gl;
    }
    if (st.status != PQ_OK) {
        st.n_ok = p0 + 8u * ok_groups;
        st.finished = 1;
        st.end_pos = len;
    } else if (p0 + 128u >= st.need) {
        st.n_ok = st.need;
        st.finished = 1;
    }
    st.pos = pos + 4 * (int64_t)total_w;
    st.tiles_done += 1;
    return t;
}

// Bytes the reference's reader would have consumed (see delta_consumed in delta_stream.hpp).
template <typename UT>
__device__ __forceinline__ uint64_t wave_consumed(const WaveState<UT> &st, int64_t len)
{
    int64_t pos = min(st.end_pos, len);
    if (st.status != PQ_OK || st.n_ok == 0) return (uint64_t)pos;
    const uint32_t last_group_pos = ((st.n_ok - 1u) >> 3) << 3;
    if (last_group_pos + 8u < st.total) return (uint64_t)pos;
    const uint32_t gl = st.last_group, j = gl >> 2;
    const int64_t wj = (st.last_widths >> (8u * j)) & 0xffu;
    pos = min(len, pos + (int64_t)(3u - (gl & 3u)) * wj);
    if (j + 1u < 4u) {
        const int64_t bw = (st.last_widths >> (8u * (j + 1u))) & 0xffu;
        for (uint32_t i = j + 1u; i < 4u; ++i) pos = min(len, pos + 4 * bw);
    }
    return (uint64_t)pos;
}

template <typename UT>
__device__ __forceinline__ void wave_decode_tile(const uint32_t *lds, const WaveTile &t, UT &carry, UT *dst,
                                                 uint32_t p0, uint32_t clip, bool vec_ok, int lane)
{
    const uint32_t sh8 = (uint32_t)(lane >> 4) * 8u;
    const uint32_t w = (t.widths >> sh8) & 0xffu;
    const uint32_t pre = (t.prefix >> sh8) & 0xffu;
    const uint32_t bit = t.bitbase + pre * 32u + (2u * (uint32_t)(lane & 15)) * w;
    const uint64_t x = extract64(lds, bit, 2u * w);
    const uint32_t m = (w >= 32u) ? 0xffffffffu : ((1u << w) - 1u);
    const UT md = (UT)t.min_delta;
    const UT a = (UT)((uint32_t)x & m) + md;
    const UT b = (UT)((uint32_t)(x >> w) & m) + md;
    const UT pair = a + b;
    UT incl;
    if constexpr (sizeof(UT) == 8) incl = wave_scan_u64(pair); else incl = wave_scan_u32(pair);
    const UT v0 = carry + (incl - pair);
    const UT v1 = v0 + a;
    UT tot;
    if constexpr (sizeof(UT) == 8) tot = wave_last_u64(incl); else tot = wave_last_u32(incl);
    carry += tot;
    const uint32_t p = p0 + 2u * (uint32_t)lane;
    if (p + 1u < clip && vec_ok) {
        if constexpr (sizeof(UT) == 8)
            *reinterpret_cast<uint4 *>(dst + p) = make_uint4((uint32_t)v0, (uint32_t)(v0 >> 32), (uint32_t)v1,
                                                             (uint32_t)(v1 >> 32));
        else
            *reinterpret_cast<uint2 *>(dst + p) = make_uint2(v0, v1);
    } else {
        if (p < clip) dst[p] = v0;
        if (p + 1u < clip) dst[p + 1u] = v1;
    }
}

struct WaveDeltaResult {
    int32_t status; // PQ_*, or kNeedGeneric when the layout is not 128 / 4
    uint32_t n_decoded;
    uint64_t consumed;
    uint32_t init_failed;
};

// Decode one page with one wave.  `lds` is this wave's private window.
template <typename UT>
__device__ __forceinline__ WaveDeltaResult delta_wave(uint32_t *lds, const uint8_t *S, int64_t len, uint32_t n,
                                                      UT *dst, int lane)
{
    const uint8_t *B = reinterpret_cast<const uint8_t *>(lds);
    const bool vec_ok = (reinterpret_cast<uintptr_t>(dst) & (2 * sizeof(UT) - 1)) == 0;
    WaveDeltaResult res{PQ_OK, 0, 0, 0};
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
    if (hs != PQ_OK) {
        res.status = hs;
        res.init_failed = 1;
        res.consumed = (uint64_t)pos;
        return res;
    }
    if (total == 0) { // every pull is io.EOF (:97-100)
        res.status = n > 0 ? PQ_EOF : PQ_OK;
        res.consumed = (uint64_t)pos;
        return res;
    }
    if (bs != 128u || nm != 4u) {
        res.status = kNeedGeneric;
        return res;
    }

    WaveState<UT> st;
    st.pos = pos;
    st.end_pos = pos;
    st.tiles_done = 0;
    st.total = (uint32_t)total;
    st.need = min(n, (uint32_t)total);
    st.n_ok = 0;
    st.status = PQ_OK;
    st.finished = 0;
    st.init_failed = 0;
    st.last_widths = 0;
    st.last_group = 0;

    if (st.need == 0) {
        // Init still reads the first block header eagerly (:198-204, :274-280)
        st.need = 1;
        (void)wave_walk<UT>(st, B, c0, len);
        res.status = st.init_failed ? st.status : PQ_OK;
        res.init_failed = st.init_failed;
        res.consumed = (uint64_t)(st.init_failed ? pos : st.pos - 0);
        if (!st.init_failed) {
            // position just past the header and widths: walk advanced past the data too
            int64_t p2 = pos;
            uint64_t z;
            (void)read_uvarint(B, c0, len, &p2, &z);
            res.consumed = (uint64_t)(p2 + 4);
        }
        return res;
    }

    UT carry = (UT)first;
    while (true) {
        const uint32_t base_tile = st.tiles_done;
        WaveTile tiles[kDwTiles];
#pragma unroll
        for (int k = 0; k < kDwTiles; ++k) tiles[k] = wave_walk<UT>(st, B, c0, len);
        const bool more = !st.finished;
        int64_t c1 = c0;
        if (more) {
            c1 = chunk_base(S, st.pos);
            wave_fetch(pf, S, c1, len, lane);
        }
        const uint32_t clip = st.finished ? st.n_ok : 0xffffffffu;
#pragma unroll
        for (int k = 0; k < kDwTiles; ++k)
            if (tiles[k].valid)
                wave_decode_tile<UT>(lds, tiles[k], carry, dst, (base_tile + (uint32_t)k) * 128u, clip, vec_ok, lane);
        if (!more) break;
        wave_sync();
        wave_commit(lds, pf, lane);
        wave_sync();
        c0 = c1;
    }

    res.status = st.status;
    res.n_decoded = st.n_ok;
    res.init_failed = st.init_failed;
    if (st.status == PQ_OK && n > st.total) res.status = PQ_EOF; // pulled past the declared count
    res.consumed = wave_consumed<UT>(st, len);
    return res;
}

} // namespace pq
