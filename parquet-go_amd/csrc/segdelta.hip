// segdelta.hip — DELTA_BINARY_PACKED pages cut into segments, so that a few large pages fill the
// machine like many small ones (reference: encoding/delta-binary-packed-decoder.go:29-147; a page
// is one chain there: block b + 1 starts where block b's widths say, and its first value is the
// sum of everything in front).
//
// A page of the tiled layouts (blocks of 128 values / 4 miniblocks, or 256 / 4) is cut by BYTES
// into segments of seg_bytes.  Four launches:
//   1. seg_header_kernel  one lane per page: the stream header (block size, miniblock count,
//                         value count, first value).
//   2. seg_scan_kernel    one wave per segment: finds the first block at or behind the segment's
//                         first byte — by trying every byte offset in a block's length as a header
//                         (one candidate per lane) and taking the first whose own fields and whose
//                         successor's are well formed — then walks whole blocks from there to the
//                         first header in the next segment, adding up their deltas.  The choice is
//                         a guess: nothing is decoded from it.  What is guessed is where the
//                         block's four width bytes are, not where its header starts: a minDelta
//                         varint read from a byte too early (a byte of packed data with its top
//                         bit set) or too late (inside the varint) parses just as well, so the
//                         block's minDelta is left out of the sum and comes from the segment in
//                         front, which reaches the header along the chain.
//   3. seg_link_kernel    one wave per page: segment 0 starts at the true first header; segment
//                         s + 1 is proven iff segment s is and its walk ended on a header whose
//                         width bytes are where segment s + 1 guessed them (and that block still
//                         has a value the caller wants).  Proven segments get their true header,
//                         their first value's position and the value itself (prefix sums of block
//                         counts and delta sums).  Where a guess was wrong the wave walks that
//                         segment again itself, from where the chain says it starts, and goes on
//                         (a few times per page; beyond that the rest of the page is left to the
//                         last proven segment).
//   4. seg_decode_kernel  one wave per proven segment decodes up to the next proven segment with
//                         the page decoder's own tile loop; the last proven one runs to the end of
//                         the page through the checked path, so the tail rules, the status and the
//                         consumed bytes are those of the whole-page decode.  Everything behind
//                         the first unproven segment is therefore decoded serially, as before.
// A page whose header is unusual in any way is decoded whole by its segment 0.
#include "delta_wave.hpp"
#include "pq_launch.hpp"

namespace pq {

constexpr uint32_t kSegNone = 0xffffffffu;

// ---- 1. stream headers ----------------------------------------------------------------------

template <typename UT>
__global__ __launch_bounds__(kWave) void seg_header_kernel(const pq_page_desc *__restrict__ descs,
                                                           SegPage *__restrict__ pages, uint32_t n_pages,
                                                           PageBuffers buf)
{
    const uint32_t i = blockIdx.x * kWave + threadIdx.x;
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
    pg.whole = hs == PQ_OK ? 0u : 1u;
    pg.lg = bs == 256u ? 6u : 5u;
    pg.total = (uint32_t)total;
    pg.pos0 = (uint32_t)pos;
    pg.first = (uint64_t)f;
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

// What a scan leaves in a segment's record (the link fields are written by seg_link_kernel).
__device__ __forceinline__ void seg_store(SegRec *rec, uint32_t key, uint32_t exit_off, uint32_t nblk, uint32_t exit_key,
                                          uint64_t sum, uint32_t exit_md)
{
    rec->key = key;
    rec->exit = exit_off;
    rec->nblk = nblk;
    rec->exit_key = exit_key;
    rec->sum = sum;
    rec->exit_md = exit_md;
}

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
__device__ __forceinline__ void seg_scan(uint32_t *lds, const uint8_t *S, int64_t len, uint32_t s, uint32_t seg_bytes,
                                         uint32_t pos0, SegRec *rec, int lane, uint32_t forced = kSegNone)
{
    constexpr uint32_t kBlockMax = LG == 5u ? 544u : 1040u;  // >= the longest block, a multiple of 16
    const int64_t lo = (int64_t)s * seg_bytes, hi = lo + seg_bytes;
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
    if (key == kSegNone) {
        if (lane == 0) seg_store(rec, kSegNone, kSegNone, 0u, kSegNone, 0ull, 0u);
        return;
    }

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
    uint32_t nblk = 0, exit_off = kSegNone, exit_key = kSegNone, exit_md = 0;
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
            while (o <= o_max) {
                const uint32_t k = o >> 2, sh = (o & 3u) * 8u;
                const uint32_t d0 = uni(lds[k]), d1 = uni(lds[k + 1]), d2 = uni(lds[k + 2]);
                hdr_lo = (uint32_t)(((((uint64_t)d1) << 32) | d0) >> sh);
                const uint32_t mid = (uint32_t)(((((uint64_t)d2) << 32) | d1) >> sh);
                const uint32_t stop = ~hdr_lo & 0x80808080u;
                vl = ((uint32_t)__builtin_ctz(stop | 0x80000000u) + 1u) >> 3;  // 1..4 (4 also when there is no end)
                const uint32_t widths = (uint32_t)(((((uint64_t)mid) << 32) | hdr_lo) >> (8u * vl));
                const uint32_t data = o + vl + 4u;
                const uint32_t next = data + 4u * ((widths * 0x01010101u) >> 24);
                // a minDelta of five bytes or more, a width above 32, a block that is not all there
                bad = (stop == 0u ? 1u : 0u) | (((((widths & 0x7f7f7f7fu) + 0x5f5f5f5fu) | widths) & 0x80808080u) != 0u ? 1u : 0u) |
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
                const int32_t md = wave_expand_min_delta(hdr_lo, vl) & (int32_t)md_mask;
                acc += (uint64_t)ua + (uint64_t)ub + (uint64_t)(2ll * (int64_t)md);
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
                    exit_md = (uint32_t)(int32_t)wave_raw_min_delta(hdr_lo, vl);
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
                    exit_md = (uint32_t)(int32_t)wave_raw_min_delta(t.hdr_lo, t.vl);
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
    if (lane == 0) seg_store(rec, key, exit_off, nblk, exit_key, total, exit_md);
}

__global__ __launch_bounds__(kWG) void seg_scan_kernel(const pq_page_desc *__restrict__ descs,
                                                       const SegPage *__restrict__ pages,
                                                       const uint32_t *__restrict__ owner, uint32_t n_segs,
                                                       SegRec *__restrict__ recs, PageBuffers buf)
{
    __shared__ __attribute__((aligned(16))) uint32_t window[kWG / kWave][kDwChunk / 4 + 4];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t g = blockIdx.x * (kWG / kWave) + (uint32_t)wave;
    if (g >= n_segs) return;
    const SegPage pg = pages[owner[g]];
    const uint32_t s = g - pg.seg_first;
    if (pg.whole) {
        if (lane == 0) seg_store(recs + g, kSegNone, kSegNone, 0u, kSegNone, 0ull, 0u);
        return;
    }
    const pq_page_desc d = descs[pg.desc];
    const uint8_t *S = buf.src + d.src_off;
    if (pg.lg == 5u)
        seg_scan<5>(window[wave], S, (int64_t)d.src_len, s, pg.seg_bytes, pg.pos0, recs + g, lane);
    else
        seg_scan<6>(window[wave], S, (int64_t)d.src_len, s, pg.seg_bytes, pg.pos0, recs + g, lane);
}

// ---- 3. proofs and prefix sums -----------------------------------------------------------------

constexpr uint32_t kSegRepairs = 8;  // wrong guesses per page the link pass walks again itself

__global__ __launch_bounds__(kWG) void seg_link_kernel(const pq_page_desc *__restrict__ descs,
                                                       const SegPage *__restrict__ pages, uint32_t n_pages,
                                                       SegRec *__restrict__ recs, PageBuffers buf)
{
    __shared__ __attribute__((aligned(16))) uint32_t window[kWG / kWave][kDwChunk / 4 + 4];
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = uni(threadIdx.x >> 6);
    const uint32_t pi = blockIdx.x * (kWG / kWave) + wave;
    if (pi >= n_pages) return;
    const SegPage pg = pages[pi];
    SegRec *R = recs + pg.seg_first;
    const uint32_t ns = pg.n_segs;
    if (pg.whole) {
        for (uint32_t s = lane; s < ns; s += kWave) R[s].flags = s == 0u ? kSegWhole : kSegSkip;
        return;
    }
    const pq_page_desc d = descs[pg.desc];
    const uint8_t *S = buf.src + d.src_off;
    const uint32_t need = min(d.n_values, pg.total);
    const uint64_t V = 128ull << (pg.lg - 5u);
    // K = number of proven segments (a prefix of the page's segments, at least segment 0)
    uint32_t K = ns, repairs = 0;
    uint64_t blocks_before = 0, carry = pg.first;
    uint32_t prev_exit = kSegNone, prev_key = kSegNone, prev_md = 0;
    uint32_t base = 0;
    while (base < ns) {
        const uint32_t s = base + lane;
        const bool valid = s < ns;
        uint32_t key = kSegNone, exit_off = kSegNone, nb = 0, exit_key = kSegNone, exit_md = 0;
        uint64_t rsum = 0;
        if (valid) {
            const SegRec &r = R[s];
            key = r.key;
            exit_off = r.exit;
            nb = r.nblk;
            exit_key = r.exit_key;
            exit_md = r.exit_md;
            rsum = r.sum;
        }
        uint32_t b_exit = (uint32_t)__shfl_up((int)exit_off, 1);
        uint32_t b_key = (uint32_t)__shfl_up((int)exit_key, 1);
        uint32_t b_md = (uint32_t)__shfl_up((int)exit_md, 1);
        if (lane == 0) {
            b_exit = prev_exit;
            b_key = prev_key;
            b_md = prev_md;
        }
        const bool link = s == 0u || (key != kSegNone && b_key != kSegNone && b_key == key);
        // the first block's minDelta, which the guess could not know
        const uint64_t sum = valid ? rsum + (s == 0u ? 0ull : V * (uint64_t)(int64_t)(int32_t)b_md) : 0ull;
        const uint32_t incl_b = wave_scan_u32(nb);
        const uint64_t incl_s = wave_scan_u64(sum);
        const uint64_t b_s = blocks_before + incl_b - nb;
        const uint64_t carry_s = carry + incl_s - sum;
        const bool wanted = V * b_s < (uint64_t)need;  // the segment's first value is one the caller pulls
        const uint64_t bad = __builtin_amdgcn_ballot_w64(valid && !(link && wanted));
        const uint32_t first_bad = bad != 0ull ? (uint32_t)__builtin_ctzll(bad) : (uint32_t)kWave;
        if (bad != 0ull) {
            // a wrong guess (the chain in front says where the segment starts, it still has values
            // the caller wants, and the guess was elsewhere): walk the segment again from there
            const uint32_t want_key = (uint32_t)__builtin_amdgcn_readlane((int)b_key, (int)first_bad);
            const bool still = __builtin_amdgcn_readlane((int)(wanted ? 1u : 0u), (int)first_bad) != 0;
            if (still && want_key != kSegNone && repairs < kSegRepairs) {
                repairs += 1u;
                const uint32_t sk = base + first_bad;
                if (pg.lg == 5u)
                    seg_scan<5>(window[wave], S, (int64_t)d.src_len, sk, pg.seg_bytes, pg.pos0, R + sk, (int)lane, want_key);
                else
                    seg_scan<6>(window[wave], S, (int64_t)d.src_len, sk, pg.seg_bytes, pg.pos0, R + sk, (int)lane, want_key);
                __threadfence();  // lane 0 wrote the record; every lane reads it next
                continue;         // the same 64 segments once more
            }
        }
        if (valid && lane < first_bad) {
            R[s].entry = s == 0u ? pg.pos0 : b_exit;  // the header itself, as the chain reached it
            R[s].p0 = (uint32_t)(V * b_s);
            R[s].value = carry_s;
        }
        if (bad != 0ull) {
            K = base + first_bad;
            break;
        }
        blocks_before += wave_last_u32(incl_b);
        carry += wave_last_u64(incl_s);
        prev_exit = (uint32_t)__builtin_amdgcn_readlane((int)exit_off, 63);
        prev_key = (uint32_t)__builtin_amdgcn_readlane((int)exit_key, 63);
        prev_md = (uint32_t)__builtin_amdgcn_readlane((int)exit_md, 63);
        base += kWave;
    }
    K = uni(K);
    for (uint32_t s = lane; s < ns; s += kWave) {
        uint32_t flags = kSegSkip, stop = kSegNone;
        if (s + 1u < K) {
            flags = kSegPart;
            stop = R[s].exit;  // the next proven segment's header
        } else if (s + 1u == K) {
            flags = kSegTail;
        }
        R[s].flags = flags;
        R[s].stop = stop;
    }
}

// ---- 4. decode ------------------------------------------------------------------------------------

template <typename UT>
__global__ __launch_bounds__(kWG) void seg_decode_kernel(const pq_page_desc *__restrict__ descs,
                                                         const SegPage *__restrict__ pages,
                                                         const uint32_t *__restrict__ owner, uint32_t n_segs,
                                                         const SegRec *__restrict__ recs, PageBuffers buf,
                                                         pq_page_result *__restrict__ results, uint32_t *flag,
                                                         uint32_t epoch)
{
    __shared__ __attribute__((aligned(16))) uint32_t window[kWG / kWave][kDwChunk / 4 + 4];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t g = blockIdx.x * (kWG / kWave) + (uint32_t)wave;
    if (g >= n_segs) return;
    const SegRec rec = recs[g];
    const uint32_t flags = uni(rec.flags);
    if (flags == kSegSkip) return;
    const SegPage pg = pages[owner[g]];
    const pq_page_desc d = descs[pg.desc];
    const uint8_t *S = buf.src + d.src_off;
    const int64_t len = (int64_t)d.src_len;
    UT *dst = reinterpret_cast<UT *>(buf.values + d.dst_off);
    uint32_t *lds = window[wave];
    WaveDeltaResult r;
    if (flags == kSegWhole) {
        r = delta_wave<UT>(lds, S, len, d.n_values, dst, lane);
    } else {
        const bool vec_ok = (reinterpret_cast<uintptr_t>(dst) & (2 * sizeof(UT) - 1)) == 0;
        const int64_t entry = (int64_t)uni(rec.entry);
        const int64_t stop = flags == kSegPart ? (int64_t)uni(rec.stop) : -1;
        const uint32_t p0 = uni(rec.p0);
        const uint64_t first = ((uint64_t)uni((uint32_t)(rec.value >> 32)) << 32) | uni((uint32_t)rec.value);
        WavePrefetch pf;
        const int64_t c0 = chunk_base(S, entry);
        wave_fetch(pf, S, c0, len, lane);
        wave_commit(lds, pf, lane);
        wave_sync();
        if (pg.lg == 5u)
            r = delta_wave_tiles<UT, 5, false, true>(lds, S, len, d.n_values, dst, lane, pf, c0, entry, pg.total, first,
                                                     vec_ok, nullptr, p0, stop);
        else
            r = delta_wave_tiles<UT, 6, false, true>(lds, S, len, d.n_values, dst, lane, pf, c0, entry, pg.total, first,
                                                     vec_ok, nullptr, p0, stop);
        if (flags == kSegPart) return;  // the page's result comes from the wave that reaches its end
    }
    if (lane == 0) {
        pq_page_result out;
        out.status = r.status;
        out.stage = r.init_failed ? PQ_STAGE_VALUES_INIT : PQ_STAGE_NONE;  // Init failed, not DecodeValues
        out.detail = 0;
        out.n_decoded = r.n_decoded;
        out.not_null = 0;
        out.n_levels = 0;
        out.consumed = r.consumed;
        results[pg.desc] = out;
        if (r.status == kNeedGeneric) *flag = epoch;  // wakes the sweep kernel
    }
}

hipError_t launch_seg_delta(hipStream_t s, int bits, const pq_page_desc *d_descs, SegPage *d_pages, uint32_t n_pages,
                            const uint32_t *d_owner, uint32_t n_segs, SegRec *d_recs, PageBuffers buf,
                            pq_page_result *d_results, uint32_t *d_flag, uint32_t epoch)
{
    if (n_pages == 0 || n_segs == 0) return hipSuccess;
    const uint32_t per = kWG / kWave;
    const dim3 gp((n_pages + kWave - 1) / kWave), gs((n_segs + per - 1) / per), gl((n_pages + per - 1) / per);
    if (bits == 64)
        hipLaunchKernelGGL(seg_header_kernel<uint64_t>, gp, dim3(kWave), 0, s, d_descs, d_pages, n_pages, buf);
    else
        hipLaunchKernelGGL(seg_header_kernel<uint32_t>, gp, dim3(kWave), 0, s, d_descs, d_pages, n_pages, buf);
    hipLaunchKernelGGL(seg_scan_kernel, gs, dim3(kWG), 0, s, d_descs, d_pages, d_owner, n_segs, d_recs, buf);
    hipLaunchKernelGGL(seg_link_kernel, gl, dim3(kWG), 0, s, d_descs, d_pages, n_pages, d_recs, buf);
    if (bits == 64)
        hipLaunchKernelGGL(seg_decode_kernel<uint64_t>, gs, dim3(kWG), 0, s, d_descs, d_pages, d_owner, n_segs, d_recs,
                           buf, d_results, d_flag, epoch);
    else
        hipLaunchKernelGGL(seg_decode_kernel<uint32_t>, gs, dim3(kWG), 0, s, d_descs, d_pages, d_owner, n_segs, d_recs,
                           buf, d_results, d_flag, epoch);
    return hipGetLastError();
}

}  // namespace pq
