// segdelta.hip — DELTA_BINARY_PACKED pages cut into segments, so that a few large pages fill the
// machine like many small ones (reference: encoding/delta-binary-packed-decoder.go:29-147; a page
// is one chain there: block b + 1 starts where block b's widths say, and its first value is the
// sum of everything in front).
//
// A page of the tiled layouts (blocks of 128 values / 4 miniblocks, or 256 / 4) is cut by BYTES
// into segments of seg_bytes.  Two launches:
//   1. seg_header_kernel  one lane per page: the stream header (block size, miniblock count,
//                         value count, first value).
//   2. seg_fused_kernel   one wave per segment, in three steps.
//      scan      finds the first block at or behind the segment's first byte — by trying every
//                byte offset in a block's length as a header (one candidate per lane) and taking
//                the first whose own fields and whose successor's are well formed — then walks
//                whole blocks from there to the first header in the next segment, adding up their
//                deltas.  The choice is a guess.  What is guessed is where the block's four width
//                bytes are, not where its header starts: a minDelta varint read from a byte too
//                early (a byte of packed data with its top bit set) or too late (inside the
//                varint) parses just as well, so the block's minDelta is left out of the sum and
//                comes from the segment in front, which reaches the header along the chain.
//      look-back the wave shows what it found (its summary) and looks at the segments in front
//                of it, 64 at a time: summaries whose guesses fit the walks in front of them, back
//                to the nearest segment that already knows its place in the chain, add up to the
//                number of blocks and the sum of deltas in front of this segment, and say where
//                its first block really starts.  Segment 0 starts at the true first header and
//                knows its place at once.  A wave whose guess was elsewhere walks its segment
//                again from the true start; the segments behind it wait for that, and only for
//                that.  Then it shows its own place (blocks and sum up to its end, the header its
//                walk ended on), which ends the look-back of the waves behind it.  Workgroups start
//                in the order of their index, so a wave only ever waits for waves that are running
//                or done; the wait is bounded all the same, and a wave that runs out of patience
//                leaves its page to the whole-page decoder (the sweep kernel behind this one; never
//                seen to happen: it is the exit every wait must have).
//      decode    the segment's blocks, with the page decoder's own tile loop, up to the next
//                segment's first header.  The segment whose walk ended on anything but a plain
//                block header, or which holds the last value the caller wants, runs to the end of
//                the page through the checked path, so the tail rules, the status and the consumed
//                bytes are those of the whole-page decode; the segments behind it do nothing.
// A page whose header is unusual in any way is decoded whole by its segment 0.
#include "delta_wave.hpp"
#include "pq_launch.hpp"

namespace pq {

constexpr uint32_t kSegNone = 0xffffffffu;
constexpr uint32_t kSegTab = 128;  // blocks of a segment whose headers the decode takes from the walk

// ---- 1. stream headers ----------------------------------------------------------------------

template <typename UT>
__global__ __launch_bounds__(kWave) void seg_header_kernel(const pq_page_desc *__restrict__ descs,
                                                           SegPage *__restrict__ pages, uint32_t n_pages,
                                                           PageBuffers buf, uint4 *__restrict__ looks, uint32_t n_look16)
{
    const uint32_t i = blockIdx.x * kWave + threadIdx.x;
    if (i < n_look16) looks[i] = make_uint4(0u, 0u, 0u, 0u);  // nothing shown yet (16 bytes a lane)
    if (i >= n_pages) return;
    SegPage pg = pages[i];
    const pq_page_desc d = descs[pg.desc];
    const uint8_t *S = buf.src + d.src_off;
    const int64_t len = (int64_t)d.src_len;
    // readBlockHeader (:29-65); anything but a plain success leaves the page to the whole-page decoder
    int64_t pos = 0;
    uint64_t bs = 0, nm = 0, total = 0, u = 0;
    int hs = read_uvarint(S, 0, len, &pos, &bs);
    if (hs == PQ_OK && (bs > 0x7fffffffull || bs == 0 || (bs % 128u) != 0)) hs = PQ_INVALID_BLOCK_SIZE;
    if (hs == PQ_OK) hs = read_uvarint(S, 0, len, &pos, &nm);
    if (hs == PQ_OK && (nm > 0x7fffffffull || nm == 0 || (bs % nm) != 0)) hs = PQ_INVALID_MINIBLOCK_COUNT;
    if (hs == PQ_OK) hs = read_uvarint(S, 0, len, &pos, &total);
    if (hs == PQ_OK && (total > 0x7fffffffull || total == 0)) hs = PQ_INT32_OUT_OF_RANGE;
    int64_t f = 0;
    if (hs == PQ_OK) hs = read_uvarint(S, 0, len, &pos, &u);
    if (hs == PQ_OK) {
        f = unzigzag(u);
        if (sizeof(UT) == 4 && (f > 2147483647ll || f < -2147483648ll)) hs = PQ_INT32_OUT_OF_RANGE;
    }
    if (hs == PQ_OK && !((bs == 128u || bs == 256u) && nm == 4u)) hs = PQ_INVALID_BLOCK_SIZE;
    const uint64_t need = total < (uint64_t)d.n_values ? total : (uint64_t)d.n_values;
    if (hs == PQ_OK && need < 2u * bs) hs = PQ_EOF;  // nothing to gain
    if (hs == PQ_OK) {
        // A sample: the first block.  The walks step over blocks whose widths are at most 32 (the
        // reference takes no others); a page that starts with another kind would have every wave
        // but the first scan its segment for nothing.
        int64_t p2 = pos;
        uint32_t vl = 0;
        while (vl < 10u && p2 < len && (S[p2] & 0x80u) != 0u) {
            ++p2;
            ++vl;
        }
        bool plain = vl < 10u && p2 + 5 <= len;
        for (int k = 1; plain && k <= 4; ++k) plain = S[p2 + k] <= 32u;
        if (!plain) hs = PQ_EOF;
    }
    pg.whole = hs == PQ_OK ? 0u : 1u;
    pg.lg = bs == 256u ? 6u : 5u;
    pg.total = (uint32_t)total;
    pg.pos0 = (uint32_t)pos;
    pg.first = (uint64_t)f;
    pg.giveup = 0;
    pages[i] = pg;
}

// ---- 2. guess and walk ----------------------------------------------------------------------

// A block header tried at window offset `wo` (bytes): zig-zag varint of at most 8 bytes, four
// width bytes of at most 32 each, all inside the stream (lim = window offset of its end).
struct SegCand {
    bool ok;
    uint32_t vl;    // bytes of the minDelta varint
    uint32_t next;  // window offset of the header behind this block
};

// What a scan finds (the same in every lane).
struct SegScan {
    uint32_t key;       // stream offset of the first block's width bytes (kSegNone: none found)
    uint32_t exit_off;  // header of the next segment's first block, reached by walking whole blocks (kSegNone: not reached)
    uint32_t nblk;      // blocks walked
    uint32_t exit_key;  // width bytes of the header at exit_off
    uint64_t sum;       // sum of the walked deltas, without the first block's minDelta (segment 0: with it)
    int64_t exit_md;    // minDelta of the header at exit_off
};

template <uint32_t LG>
__device__ __forceinline__ SegCand seg_parse(const uint32_t *lds, uint32_t wo, uint32_t lim)
{
    const uint32_t k = wo >> 2, sh = (wo & 3u) * 8u;
    const uint32_t d0 = lds[k], d1 = lds[k + 1], d2 = lds[k + 2], d3 = lds[k + 3];
    const uint32_t b0 = __builtin_amdgcn_alignbit(d1, d0, sh);
    const uint32_t b1 = __builtin_amdgcn_alignbit(d2, d1, sh);
    const uint32_t b2 = __builtin_amdgcn_alignbit(d3, d2, sh);
    const uint32_t m0 = ~b0 & 0x80808080u, m1 = ~b1 & 0x80808080u;
    uint32_t vl = 0;
    if (m0 != 0u)
        vl = ((uint32_t)__builtin_ctz(m0) + 1u) >> 3;
    else if (m1 != 0u)
        vl = 4u + (((uint32_t)__builtin_ctz(m1) + 1u) >> 3);
    const uint32_t q = 8u * vl;  // 8..64 when there is a varint
    const uint64_t lo64 = ((uint64_t)b1 << 32) | b0, hi64 = ((uint64_t)b2 << 32) | b1;
    const uint32_t wd = q < 32u ? (uint32_t)(lo64 >> q) : (uint32_t)(hi64 >> (q - 32u));
    const bool wide = ((((wd & 0x7f7f7f7fu) + 0x5f5f5f5fu) | wd) & 0x80808080u) != 0u;  // a width > 32
    const uint32_t tw = (wd * 0x01010101u) >> 24;
    SegCand c;
    c.ok = vl != 0u && !wide && wo + vl + 4u <= lim;
    c.vl = vl;
    c.next = wo + vl + 4u + ((tw << LG) >> 3);
    return c;
}

// forced != kSegNone: no guess, the first block's width bytes are known to be at `forced`.
template <uint32_t LG>
// tab (kSegTab entries of four words) / tab_n: the blocks walked, for the decode (delta_wave_tiles).
__device__ __forceinline__ SegScan seg_scan(uint32_t *lds, const uint8_t *S, int64_t len, uint32_t s, uint32_t seg_bytes,
                                            uint32_t pos0, int lane, uint32_t *tab, uint32_t &tab_n,
                                            uint32_t forced = kSegNone)
{
    constexpr uint32_t kBlockMax = LG == 5u ? 544u : 1040u;  // >= the longest block, a multiple of 16
    const int64_t lo = (int64_t)s * seg_bytes, hi = lo + seg_bytes;
    tab_n = 0;
    WavePrefetch pf;
    int64_t c0 = chunk_base(S, s == 0u ? 0 : (forced != kSegNone ? (int64_t)forced - 1 : lo - 1));
    wave_fetch(pf, S, c0, len, lane);
    wave_commit(lds, pf, lane);
    wave_sync();
    const int64_t left0 = len - c0;
    const uint32_t lim0 = left0 > (int64_t)kDwChunk ? (uint32_t)kDwChunk : (left0 < 0 ? 0u : (uint32_t)left0);

    // key = stream offset of the first block's width bytes (the byte in front of them ends the
    // minDelta varint, wherever that starts)
    uint32_t key = kSegNone;
    if (s == 0u) {
        key = pos0;  // not a guess, and not used: the walk starts at the header itself
    } else if (forced != kSegNone) {
        key = forced;
    } else {
        // A segment owns the blocks whose width bytes start in it.  The byte in front of those of
        // its first block (the end of the varint) is at lo - 1 or behind.
        const uint32_t base = (uint32_t)(lo - 1 - c0);  // 0..15
        for (uint32_t r = 0; r < kBlockMax; r += kWave) {
            const uint32_t wo = base + r + (uint32_t)lane;
            SegCand a = seg_parse<LG>(lds, wo, lim0);
            bool pass = a.ok;
            if (pass && a.next < lim0) {  // and the two headers behind it, as far as the window shows them
                const SegCand b = seg_parse<LG>(lds, a.next, lim0);
                pass = b.ok;
                if (pass && b.next + 16u <= (uint32_t)kDwChunk && b.next < lim0) pass = seg_parse<LG>(lds, b.next, lim0).ok;
            }
            const uint64_t m = __builtin_amdgcn_ballot_w64(pass);
            if (m != 0ull) {
                const uint32_t first = (uint32_t)__builtin_ctzll(m);
                key = (uint32_t)(lo - 1) + r + first + (uint32_t)__builtin_amdgcn_readlane((int)a.vl, (int)first);
                break;
            }
        }
    }
    key = uni(key);
    if (key == kSegNone) return SegScan{kSegNone, kSegNone, 0u, kSegNone, 0ull, 0};

    // whole blocks from there to the first block whose width bytes are at or behind `hi` (a
    // segment owns the blocks whose width bytes start in it).  The walk starts one byte in
    // front of the width bytes: that byte ends the varint, so it parses as a one-byte varint of
    // its own and the block's fields come out right, except minDelta.
    WaveLane ln;
    ln.sh8 = ((2u * (uint32_t)lane) >> LG) * 8u;
    ln.k2 = (2u * (uint32_t)lane) & ((1u << LG) - 1u);
    WaveBlock blk{0, 0, 0, 0, 0, 0};
    uint64_t acc = 0;      // per lane: the deltas it unpacked, minDelta included except in the first
                           // block's tiles (segment 0: there too)
    uint32_t nblk = 0, exit_off = kSegNone, exit_key = kSegNone;
    int64_t exit_md = 0;
    uint32_t o = (uint32_t)((s == 0u ? (int64_t)pos0 : (int64_t)key - 1) - c0), p0 = 0;  // segment 0 knows its header
    bool done = false;
    int32_t hi_rel = (int32_t)(hi - c0);  // `hi` as a window offset
    if constexpr (LG == 5u) {
        // Blocks of 128: the header walk of wave_walk_fast with nothing the sum does not need, and
        // one way out of the inner loop (what ended it is worked out behind it, once).
        uint32_t md_mask = s == 0u ? 0xffffffffu : 0u;  // the first block's minDelta is left out
        while (true) {
            const int64_t c1 = c0 + kDwStride;
            const bool tail = c0 + kDwChunk >= len;
            if (!tail) wave_fetch(pf, S, c1, len, lane);
            const int64_t left = len - c0;
            const uint32_t lim = left > (int64_t)kDwChunk ? (uint32_t)kDwChunk : (left < 0 ? 0u : (uint32_t)left);
            const uint32_t o_max = tail ? 0x7fffffffu : (uint32_t)(kDwChunk - kDwBlockMax);
            uint32_t hdr_lo = 0, vl = 0, bad = 0;
            bool lng = false;     // the header's minDelta varint has five bytes or more ...
            int64_t md_long = 0;  // ... and this value
            {
                // Two blocks an iteration, entirely on the vector unit, as the decode of 4-byte values
                // does it (delta_pair_loop): lanes 0..31 take the first block, lanes 32..63 the one
                // behind it, four values each; nothing is scanned or stored, a lane just adds up what
                // it unpacks.  Takes pairs of plain blocks that lie whole inside the window and the
                // segment; whatever ends it is the single-block loop's business.
                const bool upper = lane >= 32;
                const uint32_t j = (uint32_t)lane & 31u;
                const uint32_t sh8p = (j >> 3) * 8u;  // 8 * the miniblock of this lane's 4 values
                const uint32_t k4 = 4u * (j & 7u);    // index of the first of them inside the miniblock
                const uint32_t o_max2 = tail ? 0x7fffffffu : (uint32_t)(kDwChunk - 2 * kDwBlockMax);
                while (o <= o_max2) {
                    uint32_t next_a;
                    {
                        const uint32_t k = o >> 2;
                        const uint32_t d0 = lds[k], d1 = lds[k + 1], d2 = lds[k + 2];
                        const uint32_t lo = __builtin_amdgcn_alignbit(d1, d0, o * 8u);
                        const uint32_t mid = __builtin_amdgcn_alignbit(d2, d1, o * 8u);
                        const uint32_t end = ~lo & 0x80808080u;
                        const uint32_t vla = ((uint32_t)__builtin_ctz(end | 0x80000000u) + 1u) >> 3;
                        const uint32_t wa = (uint32_t)(((((uint64_t)mid) << 32) | lo) >> (8u * vla));
                        next_a = o + vla + 4u + 4u * __builtin_amdgcn_sad_u8(wa, 0u, 0u);
                    }
                    // a first header that is not a plain one sends the second half anywhere: keep its reads inside the window
                    const uint32_t ho = upper ? min(next_a, (uint32_t)kDwChunk) : o;
                    const uint32_t k = ho >> 2;
                    const uint32_t d0 = lds[k], d1 = lds[k + 1], d2 = lds[k + 2];
                    const uint32_t h_lo = __builtin_amdgcn_alignbit(d1, d0, ho * 8u);
                    const uint32_t mid = __builtin_amdgcn_alignbit(d2, d1, ho * 8u);
                    const uint32_t end = ~h_lo & 0x80808080u;
                    const uint32_t vlp = ((uint32_t)__builtin_ctz(end | 0x80000000u) + 1u) >> 3;  // 1..4
                    const uint32_t widths = (uint32_t)(((((uint64_t)mid) << 32) | h_lo) >> (8u * vlp));
                    const uint32_t data = ho + vlp + 4u;
                    const uint32_t next = data + 4u * __builtin_amdgcn_sad_u8(widths, 0u, 0u);
                    // minDelta of five bytes and more, a width above 32, a block that is not all in the
                    // window, a block that belongs to the next segment
                    const bool pbad = end == 0u || ((((widths & 0x7f7f7f7fu) + 0x5f5f5f5fu) | widths) & 0x80808080u) != 0u ||
                                      next > lim || (int32_t)(ho + vlp) >= hi_rel;
                    if (__builtin_amdgcn_ballot_w64(pbad) != 0ull) break;
                    const uint32_t mv = h_lo & (0xffffffffu >> (32u - 8u * vlp));
                    const uint32_t mx = (mv & 0x7fu) | ((mv >> 1) & 0x3f80u) | ((mv >> 2) & 0x1fc000u) | ((mv >> 3) & 0xfe00000u);
                    const int32_t md_raw = (int32_t)(mx >> 1) ^ -(int32_t)(mx & 1u);
                    if (j == 0u && nblk + 1u < kSegTab) {  // lanes 0 and 32: this pair's two entries
                        const uint32_t e = 4u * (nblk + (upper ? 1u : 0u));
                        tab[e] = (uint32_t)c0 + data;
                        tab[e + 1u] = widths;
                        tab[e + 2u] = (uint32_t)md_raw;
                        tab[e + 3u] = (uint32_t)(md_raw >> 31);
                    }
                    const int32_t mdp = md_raw & (int32_t)(upper ? 0xffffffffu : md_mask);
                    const uint32_t w = __builtin_amdgcn_ubfe(widths, sh8p, 8u);
                    const uint32_t sum2 = widths + (widths << 8);
                    const uint32_t sums = sum2 + (sum2 << 16);
                    const uint32_t pre = __builtin_amdgcn_ubfe(sums << 8, sh8p, 8u);
                    const uint32_t bit = data * 8u + (pre << 5) + __umul24(k4, w);
                    const uint32_t m = (uint32_t)(1ull << w) - 1u;  // w == 32 -> 0xffffffff, w == 0 -> 0
                    uint64_t four = (uint64_t)(4ll * (int64_t)mdp);
#pragma unroll
                    for (uint32_t q = 0; q < 4u; ++q) {
                        const uint32_t bq = bit + q * w;
                        const uint32_t i = bq >> 5;
                        four += (uint64_t)(__builtin_amdgcn_alignbit(lds[i + 1], lds[i], bq) & m);
                    }
                    acc += four;
                    md_mask = 0xffffffffu;
                    if (nblk + 1u < kSegTab) tab_n = nblk + 2u;
                    nblk += 2u;
                    o = (uint32_t)__builtin_amdgcn_readlane((int)next, 63);
                }
            }
            while (o <= o_max) {
                const uint32_t k = o >> 2, sh = (o & 3u) * 8u;
                const uint32_t d0 = uni(lds[k]), d1 = uni(lds[k + 1]), d2 = uni(lds[k + 2]);
                hdr_lo = (uint32_t)(((((uint64_t)d1) << 32) | d0) >> sh);
                const uint32_t mid = (uint32_t)(((((uint64_t)d2) << 32) | d1) >> sh);
                const uint32_t stop = ~hdr_lo & 0x80808080u;
                uint32_t widths;
                lng = false;
                if (stop != 0u) {
                    vl = ((uint32_t)__builtin_ctz(stop) + 1u) >> 3;  // 1..4
                    widths = (uint32_t)(((((uint64_t)mid) << 32) | hdr_lo) >> (8u * vl));
                } else {
                    lng = wave_long_min_delta(lds, k, sh, hdr_lo, mid, d2, vl, widths, md_long);  // five to ten bytes
                }
                const uint32_t data = o + vl + 4u;
                const uint32_t next = data + 4u * ((widths * 0x01010101u) >> 24);
                // a minDelta that is no varint of 64 bits, a width above 32, a block that is not all there
                bad = (stop == 0u && !lng ? 1u : 0u) | (((((widths & 0x7f7f7f7fu) + 0x5f5f5f5fu) | widths) & 0x80808080u) != 0u ? 1u : 0u) |
                      (next > lim ? 1u : 0u);
                if (bad != 0u || (int32_t)(o + vl) >= hi_rel) break;
                const uint32_t w = __builtin_amdgcn_ubfe(widths, ln.sh8, 8u);
                const uint32_t pre = __builtin_amdgcn_ubfe(widths * 0x01010100u, ln.sh8, 8u);
                const uint32_t bit = data * 8u + (pre << 5) + ln.k2 * w;
                const uint32_t i = bit >> 5, sft = bit & 31u;
                const uint32_t e0 = lds[i], e1 = lds[i + 1], e2 = lds[i + 2];
                const uint32_t l32 = __builtin_amdgcn_alignbit(e1, e0, sft);
                const uint32_t h32 = __builtin_amdgcn_alignbit(e2, e1, sft);
                const uint32_t m = (uint32_t)(1ull << w) - 1u;
                const uint32_t ua = l32 & m;
                const uint32_t ub = (uint32_t)((((uint64_t)h32 << 32) | l32) >> w) & m;
                // minDelta twice per lane (its two values), expanded on the vector unit like the decode does
                const int64_t md = md_mask == 0u ? 0ll : (lng ? md_long : (int64_t)wave_expand_min_delta(hdr_lo, vl));
                acc += (uint64_t)ua + (uint64_t)ub + 2ull * (uint64_t)md;
                if (nblk < kSegTab && tab_n == nblk) {  // the table has no gaps: it ends where one block was not recorded
                    const int64_t mr = lng ? md_long : wave_raw_min_delta(hdr_lo, vl);
                    if (lane == 0) {
                        tab[4u * nblk] = (uint32_t)c0 + data;
                        tab[4u * nblk + 1u] = widths;
                        tab[4u * nblk + 2u] = (uint32_t)mr;
                        tab[4u * nblk + 3u] = (uint32_t)((uint64_t)mr >> 32);
                    }
                    tab_n = nblk + 1u;
                }
                md_mask = 0xffffffffu;
                nblk += 1u;
                o = next;
            }
            if (o <= o_max) {  // left by the break
                if (bad == 0u) {
                    // the next segment's first block, reached along the chain: its header, where its
                    // width bytes are, and its minDelta
                    exit_off = (uint32_t)(c0 + (int64_t)o);
                    exit_key = exit_off + vl;
                    exit_md = lng ? md_long : wave_raw_min_delta(hdr_lo, vl);
                }
                break;  // else: not a whole, plain block; the walk (and the chain of proofs) ends here
            }
            if (tail) break;
            wave_sync();
            wave_commit(lds, pf, lane);
            wave_sync();
            c0 = c1;
            o -= (uint32_t)kDwStride;
            hi_rel -= kDwStride;
        }
    } else {
        while (true) {
            const int64_t c1 = c0 + kDwStride;
            const bool tail = c0 + kDwChunk >= len;
            if (!tail) wave_fetch(pf, S, c1, len, lane);
            const int64_t left = len - c0;
            const uint32_t lim = left > (int64_t)kDwChunk ? (uint32_t)kDwChunk : (left < 0 ? 0u : (uint32_t)left);
            while (tail || o + (uint32_t)kDwBlockMax <= (uint32_t)kDwChunk) {
                const bool header = blk.half == 0u;
                const int64_t at = c0 + (int64_t)o;
                WaveTile t = wave_tile_none();
                if (!wave_walk_fast<LG>(t, lds, o, lim, p0, 0xffffffffu, blk)) {
                    done = true;  // not a whole, plain block: the walk (and the chain of proofs) ends here
                    break;
                }
                if (header && at + (int64_t)t.vl >= hi) {
                    exit_off = (uint32_t)at;
                    exit_key = exit_off + t.vl;
                    exit_md = wave_raw_min_delta(t.hdr_lo, t.vl);
                    done = true;
                    break;
                }
                const uint32_t w = __builtin_amdgcn_ubfe(t.widths, ln.sh8, 8u);
                const uint32_t pre = __builtin_amdgcn_ubfe(t.prefix, ln.sh8, 8u);
                const uint32_t bit = t.bitbase + (pre << LG) + ln.k2 * w;
                const uint32_t i = bit >> 5, sft = bit & 31u;
                const uint32_t e0 = lds[i], e1 = lds[i + 1], e2 = lds[i + 2];
                const uint32_t l32 = __builtin_amdgcn_alignbit(e1, e0, sft);
                const uint32_t h32 = __builtin_amdgcn_alignbit(e2, e1, sft);
                const uint32_t m = (uint32_t)(1ull << w) - 1u;
                const uint32_t ua = l32 & m;
                const uint32_t ub = (uint32_t)((((uint64_t)h32 << 32) | l32) >> w) & m;
                const int64_t md2 = (nblk != 0u || s == 0u) ? 2ll * (int64_t)wave_expand_min_delta(t.hdr_lo, t.vl) : 0ll;
                acc += (uint64_t)ua + (uint64_t)ub + (uint64_t)md2;
                if (blk.half == 0u) nblk += 1u;
            }
            if (done || tail) break;
            wave_sync();
            wave_commit(lds, pf, lane);
            wave_sync();
            c0 = c1;
            o -= (uint32_t)kDwStride;
        }
    }
    const uint64_t total = wave_last_u64(wave_scan_u64(acc));
    return SegScan{key, uni(exit_off), uni(nblk), uni(exit_key), total,
                   (int64_t)(((uint64_t)uni((uint32_t)((uint64_t)exit_md >> 32)) << 32) | uni((uint32_t)exit_md))};
}

// ---- what segments show each other ------------------------------------------------------------
// Six 64-bit words per segment, all zero when the kernel starts (seg_header_kernel clears them).
// Bit 63 of a word says that it has been written; a record counts once all three of its words
// show that bit.  Words are stored and loaded with relaxed agent-scope atomics and nothing else:
// no fence, so no cache is written back or thrown away while the decode's output streams through.
//   summary  a0: key | blocks walked << 32 | bit 63 of the sum << 56
//            a1: header the walk ended on | bytes of its minDelta varint << 32 | bit 63 of that minDelta << 36
//            a2: bits 0..62 of the sum
//            a3: bits 0..62 of the minDelta
//   place    p0: as a1, for the walk that counts (0xffffffff: the page ends with this segment or one
//                in front; with varint bytes = 15: a wait ran out)
//            p1: blocks up to there | bit 63 of the carry << 32
//            p2: bits 0..62 of the carry (first value + every delta up to there)
//            p3: as a3
constexpr uint64_t kLookBit = 1ull << 63;
constexpr uint32_t kLookNone = 0, kLookSummary = 1, kLookPlace = 2, kLookGaveUp = 3;

__device__ __forceinline__ uint64_t look_ld(const uint64_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void look_st(uint64_t *p, uint64_t v)
{
    __hip_atomic_store(p, v | kLookBit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// exit word: header | bytes of its minDelta varint (1..10) << 32 | bit 63 of the minDelta << 36; the
// minDelta's other bits have a word of their own
__device__ __forceinline__ uint64_t look_exit_word(uint32_t exit_off, uint32_t exit_key, int64_t exit_md)
{
    if (exit_off == kSegNone) return (uint64_t)kSegNone;
    return (uint64_t)exit_off | ((uint64_t)((exit_key - exit_off) & 15u) << 32) | (((uint64_t)exit_md >> 63) << 36);
}
__device__ __forceinline__ void look_exit_fields(uint64_t w, uint64_t mdw, uint32_t &exit_off, uint32_t &exit_key, int64_t &exit_md)
{
    exit_off = (uint32_t)w;
    exit_key = exit_off == kSegNone ? kSegNone : exit_off + ((uint32_t)(w >> 32) & 15u);
    exit_md = (int64_t)((mdw & ~kLookBit) | (((w >> 36) & 1ull) << 63));
}

// Called by lane 0.
__device__ __forceinline__ void look_show_summary(SegLook *r, const SegScan &a)
{
    look_st(&r->a0, (uint64_t)a.key | ((uint64_t)a.nblk << 32) | ((a.sum >> 63) << 56));
    look_st(&r->a1, look_exit_word(a.exit_off, a.exit_key, a.exit_md));
    look_st(&r->a2, a.sum & ~kLookBit);
    look_st(&r->a3, (uint64_t)a.exit_md & ~kLookBit);
}
// exit_off == kSegNone: the segments behind this one have nothing to do.
__device__ __forceinline__ void look_show_place(SegLook *r, uint64_t blocks, uint64_t carry, uint32_t exit_off,
                                                uint32_t exit_key, int64_t exit_md)
{
    look_st(&r->p3, (uint64_t)exit_md & ~kLookBit);
    look_st(&r->p0, look_exit_word(exit_off, exit_key, exit_md));
    look_st(&r->p1, (blocks & 0xffffffffull) | ((carry >> 63) << 32));
    look_st(&r->p2, carry & ~kLookBit);
}
__device__ __forceinline__ void look_show_gave_up(SegLook *r)
{
    look_st(&r->p3, 0ull);
    look_st(&r->p0, (uint64_t)kSegNone | (15ull << 32));
    look_st(&r->p1, 0ull);
    look_st(&r->p2, 0ull);
}

__device__ __forceinline__ uint64_t seg_readlane_u64(uint64_t x, uint32_t l)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, (int)l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), (int)l);
    return ((uint64_t)hi << 32) | lo;
}

// What enters segment s (the same in every lane).
constexpr uint32_t kInOk = 0, kInDead = 1, kInGaveUp = 2;
struct SegIn {
    uint32_t status;    // kInDead: a segment in front decodes to the end of the page; kInGaveUp: a wait ran out
    uint32_t entry;     // stream offset of the header of the segment's first block
    uint32_t want_key;  // ... of that block's width bytes
    int64_t md;         // that block's minDelta
    uint64_t blocks;    // blocks in front of the segment
    uint64_t carry;     // first value + every delta in front of the segment
};

// Called by a whole wave, s > 0.  L: the page's records.  Lane i looks at segment j - i.  A window
// ends at the nearest segment that shows its place; the segments between it and s must show
// summaries whose first block is where the walk in front of them ended (a summary that does not
// fit is a wrong guess: its wave will walk again and show its place, which is waited for).  With
// no place among 64 segments, 63 summaries are added up and the window moves on by 63 (the last
// one is looked at again, next to the segment in front of it).
__device__ __forceinline__ SegIn seg_look_back(const SegLook *L, uint32_t s, int lane, uint64_t V, uint32_t spin_limit)
{
    SegIn in{kInGaveUp, 0u, 0u, 0, 0ull, 0ull};
    uint64_t acc_blocks = 0, acc_sum = 0;
    bool first_window = true;
    int64_t j = (int64_t)s - 1;
    uint32_t spins = 0;
    while (true) {
        const int64_t idx = j - lane;
        uint32_t st = kLookNone, key = kSegNone, nblk = 0, exit_off = kSegNone, exit_key = kSegNone, alive = 0;
        int64_t md = 0;
        uint64_t sum = 0, blocks = 0, carry = 0;
        // few lanes at first: every look is a memory transaction per lane that no cache serves
        uint32_t width = first_window ? 8u : (uint32_t)kWave;
        uint64_t known;
        uint32_t fp;
        while (true) {
            if (idx >= 0 && (uint32_t)lane < width && st < kLookPlace) {
                const SegLook *r = L + idx;
                const uint64_t p0 = look_ld(&r->p0);
                if ((p0 & kLookBit) != 0ull) {
                    const uint64_t p1 = look_ld(&r->p1), p2 = look_ld(&r->p2), p3 = look_ld(&r->p3);
                    if ((p1 & p2 & p3 & kLookBit) != 0ull) {
                        look_exit_fields(p0, p3, exit_off, exit_key, md);
                        blocks = p1 & 0xffffffffull;
                        carry = (p2 & ~kLookBit) | ((p1 >> 32) << 63);
                        alive = exit_off != kSegNone ? 1u : 0u;
                        st = exit_off == kSegNone && ((uint32_t)(p0 >> 32) & 15u) == 15u ? kLookGaveUp : kLookPlace;
                    }
                } else if (st == kLookNone) {
                    const uint64_t a0 = look_ld(&r->a0), a1 = look_ld(&r->a1), a2 = look_ld(&r->a2), a3 = look_ld(&r->a3);
                    if ((a0 & a1 & a2 & a3 & kLookBit) != 0ull) {
                        key = (uint32_t)a0;
                        nblk = (uint32_t)(a0 >> 32) & 0xffffffu;
                        look_exit_fields(a1, a3, exit_off, exit_key, md);
                        sum = (a2 & ~kLookBit) | (((a0 >> 56) & 1ull) << 63);
                        st = kLookSummary;
                    }
                }
            }
            known = __builtin_amdgcn_ballot_w64(st >= kLookPlace);
            fp = known != 0ull ? (uint32_t)__builtin_ctzll(known) : (uint32_t)kWave;
            if (known != 0ull) {
                if ((uint32_t)__builtin_amdgcn_readlane((int)st, (int)fp) == kLookGaveUp) return in;
                if (__builtin_amdgcn_readlane((int)alive, (int)fp) == 0) {
                    in.status = kInDead;
                    return in;
                }
            }
            // the segment in front of this lane's is the next lane's
            const uint32_t f_st = (uint32_t)__shfl_down((int)st, 1), f_exit_key = (uint32_t)__shfl_down((int)exit_key, 1);
            const bool fits = st == kLookSummary && key != kSegNone && exit_key != kSegNone && lane < kWave - 1 &&
                              (f_st == kLookSummary || f_st == kLookPlace) && f_exit_key == key;
            bool ok;
            if (known != 0ull)
                ok = (uint32_t)lane >= fp || fits;
            else
                ok = (uint32_t)lane >= width || ((uint32_t)lane == width - 1u ? st == kLookSummary : fits);
            const uint64_t waiting = __builtin_amdgcn_ballot_w64(!ok);
            if (waiting == 0ull) {
                if (known != 0ull || width == (uint32_t)kWave) break;
                width = kWave;  // the near ones all fit and none knows its place
                continue;
            }
            if (spins >= spin_limit) return in;
            spins += 1u;
            __builtin_amdgcn_s_sleep(2);
        }
        const uint32_t cnt = known != 0ull ? fp : (uint32_t)kWave - 1u;
        const bool mine = (uint32_t)lane < cnt;
        // the first block's minDelta: where the walk in front ended
        const uint64_t f_md = ((uint64_t)(uint32_t)__shfl_down((int)(uint32_t)((uint64_t)md >> 32), 1) << 32) |
                              (uint32_t)__shfl_down((int)(uint32_t)md, 1);
        const uint64_t my_sum = mine ? sum + (nblk != 0u ? V * f_md : 0ull) : 0ull;
        acc_blocks += wave_sum_u32(mine ? nblk : 0u);
        acc_sum += wave_last_u64(wave_scan_u64(my_sum));
        if (first_window) {
            in.entry = (uint32_t)__builtin_amdgcn_readlane((int)exit_off, 0);
            in.want_key = (uint32_t)__builtin_amdgcn_readlane((int)exit_key, 0);
            in.md = (int64_t)seg_readlane_u64((uint64_t)md, 0u);
            first_window = false;
        }
        if (known != 0ull) {
            in.blocks = seg_readlane_u64(blocks, fp) + acc_blocks;
            in.carry = seg_readlane_u64(carry, fp) + acc_sum;
            in.status = kInOk;
            return in;
        }
        j -= kWave - 1;
    }
}

// ---- 2. scan, look back, decode --------------------------------------------------------------------

template <typename UT>
__global__ __launch_bounds__(kWG) void seg_fused_kernel(const pq_page_desc *__restrict__ descs, SegPage *pages,
                                                        const uint32_t *__restrict__ owner, uint32_t n_segs,
                                                        SegRec *__restrict__ recs, SegLook *looks, PageBuffers buf,
                                                        pq_page_result *__restrict__ results, uint32_t *flag,
                                                        uint32_t epoch, uint32_t spin_limit)
{
    __shared__ __attribute__((aligned(16))) uint32_t window[kWG / kWave][kDwChunk / 4 + 4];
    __shared__ __attribute__((aligned(16))) uint32_t blocks[kWG / kWave][4 * kSegTab];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t g = blockIdx.x * (kWG / kWave) + (uint32_t)wave;
    if (g >= n_segs) return;
    uint32_t *tab = blocks[wave];
    uint32_t tab_n = 0;
    const uint32_t pi = owner[g];
    const SegPage pg = pages[pi];
    const uint32_t s = g - pg.seg_first;
    const pq_page_desc d = descs[pg.desc];
    const uint8_t *S = buf.src + d.src_off;
    const int64_t len = (int64_t)d.src_len;
    UT *dst = reinterpret_cast<UT *>(buf.values + d.dst_off);
    uint32_t *lds = window[wave];
    SegRec *rec = recs + g;
    SegRec out{};  // what pq_plan_segments shows of this segment
    out.key = out.exit = out.exit_key = kSegNone;
    out.flags = kSegSkip;
    out.stop = kSegNone;
    WaveDeltaResult r;
    bool tail = true;
    if (pg.whole) {
        if (s != 0u) {
            if (lane == 0) *rec = out;
            return;
        }
        out.flags = kSegWhole;
        r = delta_wave<UT>(lds, S, len, d.n_values, dst, lane);
    } else {
        SegLook *L = looks + pg.seg_first;
        const uint32_t need = min(d.n_values, pg.total);
        const uint64_t V = 128ull << (pg.lg - 5u);
        SegScan a = pg.lg == 5u ? seg_scan<5>(lds, S, len, s, pg.seg_bytes, pg.pos0, lane, tab, tab_n)
                                : seg_scan<6>(lds, S, len, s, pg.seg_bytes, pg.pos0, lane, tab, tab_n);
        uint64_t blocks = 0, carry = pg.first;
        uint32_t entry = pg.pos0;
        int64_t md0 = 0;  // the first block's minDelta: the walk of segment 0 has read it, the others are told
        if (s != 0u) {
            if (lane == 0) look_show_summary(L + s, a);
            SegIn in = seg_look_back(L, s, lane, V, spin_limit);
            // summaries say nothing of what the caller wants: a segment in front may hold the last value
            if (in.status == kInOk && V * in.blocks >= (uint64_t)need) in.status = kInDead;
            if (in.status != kInOk) {
                if (lane == 0) {
                    if (in.status == kInGaveUp) {
                        // The page goes to the sweep kernel behind this one, whole.  A segment behind this
                        // one may still find its place (it looks past this one's summary at segments that
                        // have shown theirs in the meantime), so the wave that reaches the page's end may
                        // write the page's result while this one writes its mark.  The mark has to be what
                        // stays: the page's word first, and only when it has arrived the status; the wave
                        // at the page's end writes its result first and looks at the page's word when that
                        // has arrived.  Either it sees the word, or its result was there before the word
                        // and so before the status written here.
                        look_show_gave_up(L + s);
                        __hip_atomic_store(&pages[pi].giveup, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __hip_atomic_store(&results[pg.desc].status, (int32_t)kNeedGeneric, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                        *flag = epoch;
                    } else {
                        look_show_place(L + s, 0ull, 0ull, kSegNone, kSegNone, 0u);
                    }
                    out.key = a.key;
                    *rec = out;
                }
                return;
            }
            if (a.key != in.want_key)  // the guess was elsewhere: once more, from where the chain says
                a = pg.lg == 5u ? seg_scan<5>(lds, S, len, s, pg.seg_bytes, pg.pos0, lane, tab, tab_n, in.want_key)
                                : seg_scan<6>(lds, S, len, s, pg.seg_bytes, pg.pos0, lane, tab, tab_n, in.want_key);
            blocks = in.blocks;
            carry = in.carry;
            entry = in.entry;
            md0 = in.md;
            a.sum += a.nblk != 0u ? V * (uint64_t)in.md : 0ull;
        }
        if (s == 0u && tab_n != 0u) md0 = (int64_t)(((uint64_t)uni(tab[3]) << 32) | uni(tab[2]));
        // the last one with a wave of its own: nothing to go on from, or nothing wanted behind it
        tail = a.exit_key == kSegNone || s + 1u == pg.n_segs || V * (blocks + a.nblk) >= (uint64_t)need;
        if (lane == 0) {
            look_show_place(L + s, blocks + a.nblk, carry + a.sum, tail ? kSegNone : a.exit_off, a.exit_key, a.exit_md);
            out.key = a.key;
            out.exit = a.exit_off;
            out.nblk = a.nblk;
            out.exit_key = a.exit_key;
            out.sum = a.sum;
            out.exit_md = (uint32_t)a.exit_md;  // its low half
            out.flags = tail ? kSegTail : kSegPart;
            out.value = carry;
            out.entry = entry;
            out.p0 = (uint32_t)(V * blocks);
            out.stop = tail ? kSegNone : a.exit_off;
            *rec = out;
        }
        const bool vec_ok = (reinterpret_cast<uintptr_t>(dst) & (2 * sizeof(UT) - 1)) == 0;
        const int64_t stop = tail ? -1 : (int64_t)a.exit_off;
        WavePrefetch pf;
        const int64_t c0 = chunk_base(S, (int64_t)entry);
        wave_sync();  // the scan's last reads of the window are done
        wave_fetch(pf, S, c0, len, lane);
        wave_commit(lds, pf, lane);
        wave_sync();
        if (pg.lg == 5u)
            r = delta_wave_tiles<UT, 5, false, true>(lds, S, len, d.n_values, dst, lane, pf, c0, (int64_t)entry, pg.total,
                                                     carry, vec_ok, nullptr, (uint32_t)(V * blocks), stop, tab, tab_n, md0);
        else
            r = delta_wave_tiles<UT, 6, false, true>(lds, S, len, d.n_values, dst, lane, pf, c0, (int64_t)entry, pg.total,
                                                     carry, vec_ok, nullptr, (uint32_t)(V * blocks), stop);
        if (!tail) return;  // the page's result comes from the wave that reaches its end
    }
    if (lane == 0) {
        if (pg.whole) *rec = out;
        pq_page_result res;
        res.status = r.status;
        res.stage = r.init_failed ? PQ_STAGE_VALUES_INIT : PQ_STAGE_NONE;  // Init failed, not DecodeValues
        res.detail = 0;
        res.n_decoded = r.n_decoded;
        res.not_null = 0;
        res.n_levels = 0;
        res.consumed = r.consumed;
        results[pg.desc] = res;
        if (r.status == kNeedGeneric) *flag = epoch;  // wakes the sweep kernel
        if (!pg.whole) {
            // a wave of this page that gave up (see above): its mark stays
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (__hip_atomic_load(&pages[pi].giveup, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                __hip_atomic_store(&results[pg.desc].status, (int32_t)kNeedGeneric, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                *flag = epoch;
            }
        }
    }
}

hipError_t launch_seg_delta(hipStream_t s, int bits, const pq_page_desc *d_descs, SegPage *d_pages, uint32_t n_pages,
                            const uint32_t *d_owner, uint32_t n_segs, SegRec *d_recs, SegLook *d_looks, PageBuffers buf,
                            pq_page_result *d_results, uint32_t *d_flag, uint32_t epoch, uint32_t spin_limit)
{
    if (n_pages == 0 || n_segs == 0) return hipSuccess;
    const uint32_t per = kWG / kWave;
    const uint32_t n_look16 = n_segs * (uint32_t)(sizeof(SegLook) / 16);
    const dim3 gs((n_segs + per - 1) / per), gh((max(n_pages, n_look16) + kWave - 1) / kWave);
    uint4 *looks16 = reinterpret_cast<uint4 *>(d_looks);
    if (bits == 64) {
        hipLaunchKernelGGL(seg_header_kernel<uint64_t>, gh, dim3(kWave), 0, s, d_descs, d_pages, n_pages, buf, looks16, n_look16);
        hipLaunchKernelGGL(seg_fused_kernel<uint64_t>, gs, dim3(kWG), 0, s, d_descs, d_pages, d_owner, n_segs, d_recs, d_looks,
                           buf, d_results, d_flag, epoch, spin_limit);
    } else {
        hipLaunchKernelGGL(seg_header_kernel<uint32_t>, gh, dim3(kWave), 0, s, d_descs, d_pages, n_pages, buf, looks16, n_look16);
        hipLaunchKernelGGL(seg_fused_kernel<uint32_t>, gs, dim3(kWG), 0, s, d_descs, d_pages, d_owner, n_segs, d_recs, d_looks,
                           buf, d_results, d_flag, epoch, spin_limit);
    }
    return hipGetLastError();
}

}  // namespace pq
