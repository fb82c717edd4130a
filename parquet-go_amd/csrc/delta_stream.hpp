// delta_stream.hpp — workgroup-wide DELTA_BINARY_PACKED decoder (INT32 / INT64).
//
// Value sequence of the reference's DeltaBinaryPackDecoder32/64
// (encoding/delta-binary-packed-decoder.go:29-170,213-225,289-301) for Init
// followed by n pulls:  out[0] = first, out[p+1] = out[p] + minDelta(block(p)) +
// unpacked[p]  with wrapping adds in the element type, where position p needs the
// 8-value group floor(p/8) to be readable (even if its delta is never used).
//
// Layout of the work: a TILE is 128 consecutive positions (never crosses a block
// because block sizes are multiples of 128).  Per pass, one lane walks the block
// headers (zig-zag minDelta, miniblock widths) that cover the next <= 16 tiles of
// the LDS-staged chunk and records, per tile, minDelta and up to 16 segments
// {bit offset in LDS, width}; then every wave decodes 4 contiguous tiles: each
// lane unpacks 2 adjacent deltas (one 16-byte store of 2 int64 later), a DPP
// inclusive scan over the 64 lane sums gives the in-tile prefix, the wave keeps a
// running sum across its 4 tiles, the 4 wave totals meet in LDS, and the carried
// base is added just before the coalesced store.
#pragma once

#include "pq_device.hpp"

namespace pq {

constexpr int kDlChunk = 8192;   // staged bytes per pass
constexpr int kDlTilesPerWave = 4;
constexpr int kDlTiles = 4 * kDlTilesPerWave; // tiles per pass
constexpr int kDlMaxMiniblocks = 128;         // widths kept per block header
constexpr int kDlTileBytesMax = 128 * 32 / 8; // 512: a tile's packed deltas at w = 32

struct DeltaShared {
    uint32_t chunk[kDlChunk / 4 + 4];
    uint32_t seg[kDlTiles][16];   // (bit offset in chunk << 8) | width, per tile segment
    uint64_t tile_min[kDlTiles];  // minDelta of the tile's block (element-type bits)
    uint64_t wave_tot[4];
    uint8_t widths[kDlMaxMiniblocks];
    // stream header
    uint32_t block_size, n_mini, mb_values, total;
    uint32_t span;        // values per segment: min(mb_values, 128)
    uint32_t span_log2;
    uint64_t first;
    // walker state
    int64_t pos;          // stream offset of the next unparsed byte
    int64_t end_pos;      // stream offset just past the last group actually read
    uint32_t tiles_done;  // tiles fully handled by earlier passes
    uint32_t ntiles;      // tiles described by this pass
    uint32_t n_ok;        // positions that can be returned (valid once `finished`)
    uint32_t need;        // min(n, total)
    uint32_t cur_mb;      // next miniblock index within the block (n_mini => header due)
    uint32_t mb_groups;   // groups already consumed from the current miniblock
    uint32_t cur_w;       // width of the current miniblock
    uint64_t cur_min;
    int32_t status;
    uint32_t finished;
    uint32_t supported;
    uint32_t init_failed; // the error came from Init (stream header or first block header)
};

struct DeltaResult {
    int32_t status;
    uint32_t n_decoded;
    uint64_t consumed;
    uint32_t init_failed;
};

template <typename UT>
__device__ __forceinline__ int delta_block_header(DeltaShared &sh, const uint8_t *B, int64_t c0, int64_t len,
                                         int64_t *pos);

// Stream header + eager first block header check (Init, :183-207 / :259-283).
template <typename UT>
__device__ __forceinline__ void delta_parse_header(DeltaShared &sh, int64_t c0, int64_t len, uint32_t n)
{
    const uint8_t *B = reinterpret_cast<const uint8_t *>(sh.chunk);
    int64_t pos = 0;
    uint64_t u;
    sh.status = PQ_OK;
    sh.finished = 0;
    sh.init_failed = 0;
    sh.supported = 1;
    sh.n_ok = 0;
    sh.tiles_done = 0;
    sh.ntiles = 0;
    sh.total = 0;
    sh.need = 0;
    sh.mb_groups = 0;
    sh.cur_w = 0;
    sh.end_pos = 0;
    int st = read_uvarint(B, c0, len, &pos, &u);
    if (st == PQ_OK && u > 0x7fffffffull) st = PQ_INT32_OUT_OF_RANGE;
    if (st == PQ_OK && (u == 0 || (u % 128u) != 0)) st = PQ_INVALID_BLOCK_SIZE;
    uint32_t bs = (uint32_t)u;
    uint32_t nm = 0;
    if (st == PQ_OK) {
        st = read_uvarint(B, c0, len, &pos, &u);
        if (st == PQ_OK && u > 0x7fffffffull) st = PQ_INT32_OUT_OF_RANGE;
        if (st == PQ_OK && (u == 0 || (bs % (uint32_t)u) != 0)) st = PQ_INVALID_MINIBLOCK_COUNT;
        nm = (uint32_t)u;
    }
    uint32_t total = 0;
    if (st == PQ_OK) {
        st = read_uvarint(B, c0, len, &pos, &u);
        if (st == PQ_OK && u > 0x7fffffffull) st = PQ_INT32_OUT_OF_RANGE;
        total = (uint32_t)u;
    }
    uint64_t first = 0;
    if (st == PQ_OK && total != 0) {
        st = read_uvarint(B, c0, len, &pos, &u);
        if (st == PQ_OK) {
            int64_t f = unzigzag(u);
            if (sizeof(UT) == 4 && (f > 2147483647ll || f < -2147483648ll)) st = PQ_INT32_OUT_OF_RANGE;
            first = (uint64_t)f;
        }
    }
    sh.pos = pos;
    sh.end_pos = pos;
    if (st != PQ_OK) {
        sh.status = st;
        sh.init_failed = 1;
        sh.finished = 1;
        return;
    }
    const uint32_t mbv = bs / nm;
    sh.block_size = bs;
    sh.n_mini = nm;
    sh.mb_values = mbv;
    sh.total = total;
    sh.first = first;
    sh.need = min(n, total);
    sh.cur_mb = nm; // forces the first block header
    // Tiled fast path: miniblocks are whole groups and line up with 128-value tiles.
    const bool ok = (mbv % 8u == 0) && ((mbv >= 128u) ? (mbv % 128u == 0) : (128u % mbv == 0)) &&
                    nm <= (uint32_t)kDlMaxMiniblocks;
    sh.supported = ok ? 1u : 0u;
    const uint32_t span = min(mbv, 128u);
    sh.span = span;
    sh.span_log2 = 31u - (uint32_t)__clz((int)max(span, 1u));
    if (total == 0) sh.finished = 1; // every pull is io.EOF (:97-100)
    if (total != 0 && sh.need == 0 && ok) {
        // No pull will happen, but Init still reads the first block header eagerly
        // (:198-204, :274-280) and its failure is the caller's error.
        int64_t p = pos;
        const int hst = delta_block_header<UT>(sh, B, c0, len, &p);
        sh.cur_mb = nm;
        if (hst != PQ_OK) { sh.status = hst; sh.init_failed = 1; }
        sh.end_pos = p;
        sh.finished = 1;
    }
}

// Block header at `pos` (readMiniBlockHeader, :67-94).  Returns PQ_OK or the error.
template <typename UT>
__device__ __forceinline__ int delta_block_header(DeltaShared &sh, const uint8_t *B, int64_t c0, int64_t len,
                                         int64_t *pos)
{
    uint64_t u;
    int st = read_uvarint(B, c0, len, pos, &u);
    if (st != PQ_OK) return st;
    const int64_t md = unzigzag(u);
    if (sizeof(UT) == 4 && (md > 2147483647ll || md < -2147483648ll)) return PQ_INT32_OUT_OF_RANGE;
    sh.cur_min = (uint64_t)md;
    const int64_t rem = len - *pos;
    const uint32_t nm = sh.n_mini;
    if (rem <= 0) return PQ_EOF; // io.ReadFull of the widths read nothing
    if (rem < (int64_t)nm) { *pos = len; return PQ_UNEXPECTED_EOF; }
    int bad = 0;
    for (uint32_t j = 0; j < nm; ++j) {
        const uint8_t wj = B[*pos + j - c0];
        sh.widths[j] = wj;
        if (wj > 32) bad = 1; // same cap for INT32 and INT64 (:78-88)
    }
    *pos += nm;
    if (bad) return PQ_INVALID_MINIBLOCK_WIDTH;
    sh.cur_mb = 0;
    return PQ_OK;
}

// One lane: describe the next tiles that are fully inside the staged chunk.
template <typename UT>
__device__ __forceinline__ void delta_walk(DeltaShared &sh, int64_t c0, int64_t len)
{
    const uint8_t *B = reinterpret_cast<const uint8_t *>(sh.chunk);
    const int64_t cend = c0 + kDlChunk;
    const bool tail = cend >= len;
    int64_t pos = sh.pos;
    const uint32_t need = sh.need;
    const uint32_t span = sh.span;
    const uint32_t seg_groups = span >> 3;
    const uint32_t segs = 128u / span;
    const uint32_t mb_groups_total = sh.mb_values >> 3;
    uint32_t t = 0;
    for (; t < (uint32_t)kDlTiles; ++t) {
        const uint32_t p0 = (sh.tiles_done + t) * 128u;
        if (p0 >= need) { sh.finished = 1; sh.n_ok = need; break; }
        // worst case for this tile: a block header (10 + n_mini) and 512 data bytes
        if (!tail && pos + 10 + (int64_t)sh.n_mini + kDlTileBytesMax > cend) break;
        const uint32_t tile_vals = min(128u, need - p0);
        const uint32_t groups_needed = (tile_vals + 7u) >> 3;
        bool stop = false;
        for (uint32_t e = 0; e < 16u; ++e) {
            const uint32_t g0 = e * seg_groups;
            if (e >= segs || g0 >= groups_needed || stop) { sh.seg[t][e] = 0; continue; }
            if (sh.mb_groups == 0) {
                // advanceBlock (:130-147)
                if (sh.cur_mb >= sh.n_mini) {
                    const int st = delta_block_header<UT>(sh, B, c0, len, &pos);
                    if (st != PQ_OK) {
                        sh.status = st;
                        if (p0 == 0 && g0 == 0) sh.init_failed = 1;
                        sh.n_ok = p0 + 8u * g0;
                        sh.finished = 1;
                        stop = true;
                        sh.seg[t][e] = 0;
                        continue;
                    }
                }
                sh.cur_w = sh.widths[sh.cur_mb];
                sh.cur_mb++;
            }
            const uint32_t w = sh.cur_w;
            if (e == 0) sh.tile_min[t] = sh.cur_min; // a tile never crosses a block
            const uint32_t gneed = min(seg_groups, groups_needed - g0);
            const int64_t avail = len - pos;
            sh.seg[t][e] = ((uint32_t)((pos - c0) * 8) << 8) | w;
            if ((int64_t)gneed * w > avail) {
                // io.ReadFull of a group fails (:111-114): nothing left -> EOF, else short
                const uint32_t gok = (uint32_t)(avail > 0 ? avail / w : 0);
                sh.status = (avail - (int64_t)gok * w) <= 0 ? PQ_EOF : PQ_UNEXPECTED_EOF;
                sh.n_ok = p0 + 8u * (g0 + gok);
                sh.finished = 1;
                stop = true;
                pos = len;
                sh.end_pos = len;
                continue;
            }
            sh.end_pos = pos + (int64_t)gneed * w;
            pos += (int64_t)seg_groups * w;
            sh.mb_groups += seg_groups;
            if (sh.mb_groups >= mb_groups_total) sh.mb_groups = 0;
        }
        if (stop) { ++t; break; }
        if (p0 + 128u >= need) { sh.finished = 1; sh.n_ok = need; ++t; break; }
    }
    sh.ntiles = t;
    sh.pos = pos;
}

// Bytes the reference's reader has consumed once the pulls are over: the last
// group read, plus readPadding (:149-170) when that group was the final one of
// the stream — the rest of its miniblock and then, for each later miniblock slot,
// the byte count of the NEXT miniblock's width (the :161 indexing), each read
// clamped to what is left.
__device__ __forceinline__ uint64_t delta_consumed(const DeltaShared &sh, int64_t len, uint32_t n_ok)
{
    int64_t pos = min(sh.end_pos, len);
    if (sh.status != PQ_OK || n_ok == 0) return (uint64_t)pos;
    const uint32_t last_group_pos = ((n_ok - 1u) >> 3) << 3;
    if (last_group_pos + 8u < sh.total) return (uint64_t)pos;
    const int64_t mbg = sh.mb_values >> 3;
    const uint32_t groups_in_mb = ((last_group_pos % sh.mb_values) >> 3) + 1u;
    int64_t l = (mbg - (int64_t)groups_in_mb) * (int64_t)sh.cur_w;
    pos = min(len, pos + l);
    // cur_mb already points at the next miniblock slot
    const uint32_t cm = sh.cur_mb;
    if (cm < sh.n_mini) {
        const int64_t bw = sh.widths[cm];
        for (uint32_t i = cm; i < sh.n_mini; ++i) pos = min(len, pos + mbg * bw);
    }
    return (uint64_t)pos;
}

// Serial path for layouts the tiled path does not cover (miniblocks that are not
// whole groups or do not line up with 128-value tiles, or more than
// kDlMaxMiniblocks miniblocks per block).  One lane follows the reference's
// control flow literally (Init :183-207, next :96-128, advanceBlock :130-147,
// readPadding :149-170), reading the stream straight from global memory; the
// miniblock widths are read in place, so any miniblock count works.
struct SerialReader {
    const uint8_t *S;
    int64_t len, pos;
    __device__ int byte(uint32_t *b)
    {
        if (pos >= len) return PQ_EOF;
        *b = S[pos++];
        return PQ_OK;
    }
    __device__ int uvarint(uint64_t *val)
    {
        uint64_t x = 0;
        int s = 0;
        for (int i = 0; i < 10; ++i) {
            uint32_t b;
            if (byte(&b) != PQ_OK) return PQ_EOF;
            if (b < 0x80u) {
                if (i == 9 && b > 1u) return PQ_VARINT_OVERFLOW;
                *val = x | ((uint64_t)b << s);
                return PQ_OK;
            }
            x |= (uint64_t)(b & 0x7fu) << s;
            s += 7;
        }
        return PQ_VARINT_OVERFLOW;
    }
    // io.ReadFull that discards the bytes
    __device__ int skip_full(int64_t n)
    {
        if (n <= 0) return PQ_OK;
        const int64_t avail = len - pos;
        if (avail <= 0) return PQ_EOF;
        if (avail < n) { pos = len; return PQ_UNEXPECTED_EOF; }
        pos += n;
        return PQ_OK;
    }
};

template <typename UT>
__device__ __forceinline__ DeltaResult delta_serial(const uint8_t *S, int64_t len, uint32_t n, UT *dst)
{
    SerialReader r{S, len, 0};
    DeltaResult res{PQ_OK, 0, 0, 0};
    uint64_t u;
    int st;
    auto fail = [&](int code) {
        res.status = code;
        res.consumed = (uint64_t)r.pos;
        res.init_failed = code != PQ_OK; // only used before the first pull
        return res;
    };
    // readBlockHeader (:29-65)
    if ((st = r.uvarint(&u)) != PQ_OK) return fail(st);
    if (u > 0x7fffffffull) return fail(PQ_INT32_OUT_OF_RANGE);
    if (u == 0 || u % 128u) return fail(PQ_INVALID_BLOCK_SIZE);
    const int64_t bs = (int64_t)u;
    if ((st = r.uvarint(&u)) != PQ_OK) return fail(st);
    if (u > 0x7fffffffull) return fail(PQ_INT32_OUT_OF_RANGE);
    if (u == 0 || bs % (int64_t)u) return fail(PQ_INVALID_MINIBLOCK_COUNT);
    const int64_t nm = (int64_t)u;
    const int64_t mbv = bs / nm;
    if ((st = r.uvarint(&u)) != PQ_OK) return fail(st);
    if (u > 0x7fffffffull) return fail(PQ_INT32_OUT_OF_RANGE);
    const int64_t total = (int64_t)u;
    if (total == 0) { // Init succeeds; every pull is io.EOF
        res.status = n > 0 ? PQ_EOF : PQ_OK;
        res.consumed = (uint64_t)r.pos;
        return res;
    }
    if ((st = r.uvarint(&u)) != PQ_OK) return fail(st);
    int64_t f = unzigzag(u);
    if (sizeof(UT) == 4 && (f > 2147483647ll || f < -2147483648ll)) return fail(PQ_INT32_OUT_OF_RANGE);
    UT prev = (UT)f;
    UT min_delta = 0;
    int64_t widths_at = 0; // stream offset of the current block's width bytes
    auto block_header = [&]() -> int {
        uint64_t z;
        int s2 = r.uvarint(&z);
        if (s2 != PQ_OK) return s2;
        int64_t md = unzigzag(z);
        if (sizeof(UT) == 4 && (md > 2147483647ll || md < -2147483648ll)) return PQ_INT32_OUT_OF_RANGE;
        min_delta = (UT)md;
        widths_at = r.pos;
        s2 = r.skip_full(nm);
        if (s2 != PQ_OK) return s2;
        for (int64_t j = 0; j < nm; ++j)
            if (S[widths_at + j] > 32) return PQ_INVALID_MINIBLOCK_WIDTH;
        return PQ_OK;
    };
    if ((st = block_header()) != PQ_OK) return fail(st);
    int64_t cur_mb = 0, mb_pos = 0, position = 0;
    uint32_t cur_w = 0;
    int64_t group_at = 0; // stream offset of the current 8-value group
    for (uint32_t i = 0; i < n; ++i) {
        if (position >= total) { res.status = PQ_EOF; break; }
        if (position % 8 == 0) {
            if (position % mbv == 0) {
                if (cur_mb >= nm) {
                    if ((st = block_header()) != PQ_OK) { res.status = st; break; }
                    cur_mb = 0;
                }
                cur_w = S[widths_at + cur_mb];
                mb_pos = 0;
                ++cur_mb;
            }
            group_at = r.pos;
            if ((st = r.skip_full(cur_w)) != PQ_OK) { res.status = st; break; }
            mb_pos += cur_w;
            if (position + 8 >= total) {
                const int64_t l = (mbv / 8) * (int64_t)cur_w - mb_pos;
                if (l < 0) { res.status = PQ_INVALID_STREAM; break; }
                (void)r.skip_full(l);
                for (int64_t k = cur_mb; k < nm; ++k) {
                    const int64_t bw = S[widths_at + cur_mb];
                    if (bw != 0) (void)r.skip_full((mbv / 8) * bw);
                }
            }
        }
        dst[i] = prev;
        // value k of the group: bits [k*w, (k+1)*w) of its w bytes
        const uint32_t k = (uint32_t)(position % 8);
        uint64_t d = 0;
        const uint32_t bit = k * cur_w;
        for (uint32_t b = 0; b < cur_w; ++b) {
            const uint32_t bb = bit + b;
            d |= (uint64_t)((S[group_at + (bb >> 3)] >> (bb & 7u)) & 1u) << b;
        }
        prev = prev + (UT)d + min_delta;
        ++position;
        res.n_decoded = i + 1;
    }
    res.consumed = (uint64_t)r.pos;
    return res;
}

template <typename UT>
__device__ __forceinline__ void delta_store2(UT *dst, uint32_t p, UT v0, UT v1, uint32_t n_ok, bool vec_ok)
{
    if (p + 1u < n_ok && vec_ok) {
        if constexpr (sizeof(UT) == 8)
            *reinterpret_cast<uint4 *>(dst + p) = make_uint4((uint32_t)v0, (uint32_t)(v0 >> 32),
                                                             (uint32_t)v1, (uint32_t)(v1 >> 32));
        else
            *reinterpret_cast<uint2 *>(dst + p) = make_uint2(v0, v1);
    } else {
        if (p < n_ok) dst[p] = v0;
        if (p + 1u < n_ok) dst[p + 1u] = v1;
    }
}

// Decode up to n values of the stream [S, S + len) into dst (UT = uint32_t or
// uint64_t, the element type's bits).  All 256 threads must call it.
template <typename UT>
__device__ __forceinline__ DeltaResult delta_stream(DeltaShared &sh, const uint8_t *S, int64_t len, uint32_t n,
                                           UT *dst)
{
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = tid >> 6;
    const bool vec_ok = (reinterpret_cast<uintptr_t>(dst) & (2 * sizeof(UT) - 1)) == 0;

    int64_t c0 = chunk_base(S, 0);
    stage_chunk<kDlChunk>(sh.chunk, S, c0, len, tid);
    __syncthreads();
    if (tid == 0) delta_parse_header<UT>(sh, c0, len, n);
    __syncthreads();

    DeltaResult res;
    if (!sh.supported && sh.status == PQ_OK && sh.total != 0) {
        res.status = -1; // the page is finished by the serial kernel
        res.n_decoded = 0;
        res.consumed = 0;
        res.init_failed = 0;
        return res;
    }

    UT carry = (UT)sh.first; // value at position tiles_done * 128
    const uint32_t span_log2 = sh.span_log2;
    const uint32_t span_mask = sh.span - 1u;
    bool first_pass = true;

    while (!sh.finished) {
        if (!first_pass) {
            c0 = chunk_base(S, sh.pos);
            stage_chunk<kDlChunk>(sh.chunk, S, c0, len, tid);
            __syncthreads();
        }
        first_pass = false;
        if (tid == 0) delta_walk<UT>(sh, c0, len);
        __syncthreads();

        const uint32_t ntiles = sh.ntiles;
        const uint32_t base_tile = sh.tiles_done;
        const bool fin = sh.finished != 0;
        // a pass that is not the last only stores whole tiles, so n_ok only clips the last
        const uint32_t clip = fin ? sh.n_ok : 0xffffffffu;

        UT v0[kDlTilesPerWave], v1[kDlTilesPerWave];
        UT run = 0;
#pragma unroll
        for (int k = 0; k < kDlTilesPerWave; ++k) {
            const uint32_t t = (uint32_t)(wave * kDlTilesPerWave + k);
            v0[k] = 0;
            v1[k] = 0;
            if (t < ntiles) {
                const uint32_t i2 = 2u * (uint32_t)lane;
                const uint32_t sg = sh.seg[t][i2 >> span_log2];
                const uint32_t w = sg & 0xffu;
                const uint32_t bit = (sg >> 8) + (i2 & span_mask) * w;
                const uint64_t x = extract64(sh.chunk, bit, 2u * w);
                const uint32_t m = (w >= 32u) ? 0xffffffffu : ((1u << w) - 1u);
                const UT md = (UT)sh.tile_min[t];
                const UT a = (UT)((uint32_t)x & m) + md;
                const UT b = (UT)((uint32_t)(x >> w) & m) + md;
                const UT pair = a + b;
                UT incl;
                if constexpr (sizeof(UT) == 8) incl = wave_scan_u64(pair); else incl = wave_scan_u32(pair);
                const UT excl = incl - pair;
                v0[k] = run + excl;
                v1[k] = run + excl + a;
                UT tot;
                if constexpr (sizeof(UT) == 8) tot = wave_last_u64(incl); else tot = wave_last_u32(incl);
                run += tot;
            }
        }
        if (lane == 0) sh.wave_tot[wave] = (uint64_t)run;
        __syncthreads();
        UT base = carry;
        UT all = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const UT tk = (UT)sh.wave_tot[k];
            if (k < wave) base += tk;
            all += tk;
        }
#pragma unroll
        for (int k = 0; k < kDlTilesPerWave; ++k) {
            const uint32_t t = (uint32_t)(wave * kDlTilesPerWave + k);
            if (t < ntiles) {
                const uint32_t p = (base_tile + t) * 128u + 2u * (uint32_t)lane;
                delta_store2<UT>(dst, p, base + v0[k], base + v1[k], clip, vec_ok);
            }
        }
        carry += all;
        __syncthreads(); // chunk, seg and wave_tot are free again
        if (tid == 0) sh.tiles_done = base_tile + ntiles;
        __syncthreads();
    }

    res.status = sh.status;
    res.n_decoded = sh.n_ok;
    res.init_failed = sh.init_failed;
    if (sh.status == PQ_OK && n > sh.total) {
        // pulling past the declared count: io.EOF after `total` values (:97-100)
        res.status = PQ_EOF;
    }
    res.consumed = delta_consumed(sh, len, sh.n_ok);
    __syncthreads();
    return res;
}

} // namespace pq
