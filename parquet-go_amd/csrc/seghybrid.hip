// seghybrid.hip — large RLE / bit-packed hybrid streams (dictionary-index pages, level streams,
// BOOLEAN RLE values, raw hybrid streams) cut into segments, so that one large page is decoded by
// many waves (reference: encoding/hybrid-decoder.go:70-136; a stream is one chain there: run k + 1
// starts where run k's header says, so a page is one wave's serial walk however long it is).
//
// Two shapes of stream are cut, told apart by the first run (hseg_header_kernel):
//
// Mode 1 — bit-packed runs that all carry the same header: the usual dictionary-index page with many
// distinct keys (the reference's encoder closes a bit-packed run after 63 groups,
// encoding/hybrid-encoder.go:14,158-161; other writers after 64).  Every run of such a stream
// starts a fixed stride behind the one before, which can be checked for all runs at once:
//   hseg_check_kernel   one lane per run: is the header at first + r * stride the first run's
//                       header again, and is the run all there?  (64 runs per wave.)
//   hseg_link_kernel    one wave per page: the runs in front of the first "no" form a proven chain;
//                       they are dealt out in segments of whole runs, what is behind them is the tail.
//
// Mode 2 — runs of any kind (level streams, run-heavy keys, booleans): where the headers are can only
// be found by following the chain, but chains that start at different bytes soon fall into step (a
// walk from a wrong byte lands on a real header after a few runs and stays on the chain from there).
//   lseg_scan_kernel    one wave per segment of seg_bytes: walks headers only, starting a few
//                       hundred bytes in front of the segment at an arbitrary byte, and notes where
//                       the chain enters the segment, where it leaves it and how many values the
//                       runs in between hold.  A guess: nothing is decoded from it.
//   lseg_fix_kernel     one wave per segment: where the segment in front left at another place than
//                       this one entered at, walks this one again from there.  Two rounds.
//   lseg_link_kernel    one wave per page: segment 0 starts at the true first header; segment s + 1
//                       is proven iff segment s is and left where s + 1 entered.  Prefix sums of the
//                       value counts give every proven segment its first value's index.  A link
//                       that still does not hold is walked again here (a few per page); behind the
//                       first one left broken everything belongs to the tail.
//
// Both: hseg_decode_kernel, one wave per proven segment, runs the page decoder's own hybrid_wave
// over the segment's bytes as a stream of its own (runs share no state).  The tail segment runs to
// the end of the page like the whole-page decode, so errors, counts and consumed bytes are that
// decode's by construction.  hseg_final_kernel merges the segments' results into the page's.
// A stream of any other shape (mode 0) is left to the one-wave-per-stream kernel.
#include "hybrid_wave.hpp"
#include "pq_launch.hpp"

namespace pq {

// Segments (waves) per workgroup of the decode kernel: 8 where the workgroup keeps a dictionary in LDS (one
// copy serves more waves), 4 where it does not (a launch of few thousand segments ends sooner when its
// workgroups are small: 64 level streams of 1 M slots 0.30 -> 0.27 ms).
constexpr int kHsWavesDict = 8, kHsWavesPlain = 4;
constexpr uint32_t kHsMinStride = 64;    // shorter runs are not worth a lane each
constexpr uint32_t kLsPreRoll = 512;     // bytes in front of a segment a scan starts at; half of it at width 1, where
                                         // it holds as many runs (measured: 256 bytes cost definition levels and
                                         // 12-bit keys a third in repairs, and gain one-bit streams 4 %)
constexpr uint32_t kLsGuard = 80;        // a span parse reads 64 + 12 bytes from its first byte on
constexpr uint32_t kLsRepairs = 1024;    // broken links per page the link pass walks again itself (a walk costs a
                                         // tenth of a decode: still far better than leaving the rest to the tail)
constexpr uint32_t kLsNone = 0xffffffffu;

extern __shared__ __attribute__((aligned(16))) uint8_t pq_hs_smem[];

constexpr uint32_t hs_align16(uint32_t x) { return (x + 15u) & ~15u; }

__device__ __forceinline__ uint32_t hs_le32(const uint8_t *p)
{
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

// ---- which shape ----------------------------------------------------------------------------

__global__ __launch_bounds__(kWave) void hseg_header_kernel(const pq_page_desc *__restrict__ descs,
                                                            HsegPage *__restrict__ pages, uint32_t n_pages,
                                                            PageBuffers buf)
{
    const uint32_t i = blockIdx.x * kWave + threadIdx.x;
    if (i >= n_pages) return;
    HsegPage pg = pages[i];
    const pq_page_desc d = descs[pg.desc];
    const uint8_t *S = buf.src + d.src_off;
    const int64_t len = (int64_t)d.src_len;
    // where the hybrid stream proper lies, as the one-wave-per-stream kernel works it out
    uint32_t w = d.bit_width, pos0 = 0, kind_mode = HY_RAW, fixed = kLsNone, max_level = 0;
    uint32_t flags = d.flags;
    int64_t end = len;
    bool ok = len > 16 && len < 0x7fffffffll;
    if (d.kind == PQ_KIND_DICT) {  // the bit-width byte first (types/dict.go:66-85)
        w = ok ? S[0] : 0u;
        pos0 = 1;
        kind_mode = HY_DICT;
        flags = 0;
    } else if (d.kind == PQ_KIND_BOOL_RLE) {  // NewHybridDecoder(1, false).InitSize (types/boolean.go:177-181)
        w = 1;
        flags = PQ_FLAG_LENGTH_PREFIXED;
    } else if (d.kind == PQ_KIND_LEVELS) {
        w = bits_len16(d.max_def);
        kind_mode = HY_LEVELS;
        max_level = d.max_def;
        fixed = (uint32_t)len;  // a buffered decoder reads its whole section at Init (layout/chunk.go:87,103)
        ok = ok && (flags & PQ_FLAG_LEVELS_CONST) == 0u;
    } else if (d.kind != PQ_KIND_HYBRID) {
        ok = false;
    }
    ok = ok && (flags & PQ_FLAG_NO_INIT) == 0u && w >= 1u && w <= 32u;
    if (ok && (flags & PQ_FLAG_LENGTH_PREFIXED) != 0u) {  // InitSize (encoding/hybrid-decoder.go:57-68)
        const int64_t size = (int64_t)hs_le32(S);
        pos0 = 4;
        end = 4 + (size < len - 4 ? size : len - 4);
        fixed = (uint32_t)end;
    }
    ok = ok && end - (int64_t)pos0 > 16;
    // the first run header
    uint32_t mode = 0, h = 0, hl = 0;
    if (ok) {
        int64_t pos = pos0;
        uint64_t u = 0;
        const bool parsed = read_uvarint(S, 0, end, &pos, &u) == PQ_OK && u <= (1u << 21) && (u >> 1) != 0u;
        h = (uint32_t)u;
        hl = (uint32_t)(pos - pos0);
        mode = 2;
        const uint64_t stride1 = (uint64_t)hl + (uint64_t)(h >> 1) * w;
        if (parsed && (u & 1u) != 0u && hl <= 4u && stride1 >= kHsMinStride && stride1 <= 0x7fffffffull &&
            kind_mode != HY_LEVELS)
            mode = 1;
    }
    const uint32_t groups = h >> 1;
    const uint64_t stride = (uint64_t)hl + (uint64_t)groups * w;
    uint32_t head = 0;
    if (mode == 1u)
        for (uint32_t b = 0; b < hl; ++b) head |= (uint32_t)S[pos0 + b] << (8u * b);
    pg.mode = mode;
    pg.w = w;
    pg.pos0 = pos0;
    pg.end = (uint32_t)end;
    pg.stride = mode == 1u ? (uint32_t)stride : 0u;
    pg.vpr = groups * 8u;
    pg.head = head;
    pg.hl = hl;
    const uint64_t per = pg.seg_bytes / (stride ? stride : 1u);
    pg.runs_per_seg = (uint32_t)(per ? per : 1u);
    pg.n_part = 0;
    pg.kind_mode = kind_mode;
    pg.max_level = max_level;
    pg.fixed_consumed = fixed;
    pages[i] = pg;
}

// ---- mode 1: equal runs ---------------------------------------------------------------------

// counts[j] = how many of the runs 64 j .. 64 j + 63 (from the first on) are the first run again.
__global__ __launch_bounds__(kWG) void hseg_check_kernel(const pq_page_desc *__restrict__ descs,
                                                         const HsegPage *__restrict__ pages,
                                                         const uint32_t *__restrict__ owner, uint32_t n_checks,
                                                         uint32_t *__restrict__ counts, PageBuffers buf)
{
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t g = blockIdx.x * (kWG / kWave) + (threadIdx.x >> 6);
    if (g >= n_checks) return;
    const HsegPage pg = pages[owner[g]];
    if (pg.mode != 1u) return;
    const pq_page_desc d = descs[pg.desc];
    const uint8_t *S = buf.src + d.src_off;
    const uint64_t len = pg.end;
    const uint64_t r = (uint64_t)(g - pg.check_first) * kWave + lane;
    const uint64_t off = pg.pos0 + r * pg.stride;
    bool same = off + pg.stride <= len;
    if (same) {
        uint32_t head = 0;
        for (uint32_t b = 0; b < pg.hl; ++b) head |= (uint32_t)S[off + b] << (8u * b);
        same = head == pg.head;
    }
    const uint64_t m = __builtin_amdgcn_ballot_w64(same);
    const uint32_t lead = ~m == 0ull ? (uint32_t)kWave : (uint32_t)__builtin_ctzll(~m);
    if (lane == 0) counts[g] = lead;
}

__device__ __forceinline__ HsegRec hseg_rec(uint64_t off, uint32_t len, uint32_t v0, uint32_t count, uint32_t flags,
                                            uint32_t runs = 0)
{
    HsegRec r;
    r.off = off;
    r.v0 = v0;
    r.count = count;
    r.flags = flags;
    r.status = PQ_OK;
    r.n_decoded = 0;
    r.count_max = 0;
    r.consumed = 0;
    r.len = len;
    r.runs = runs;
    return r;
}

__global__ __launch_bounds__(kWG) void hseg_link_kernel(const pq_page_desc *__restrict__ descs,
                                                        HsegPage *__restrict__ pages, uint32_t n_pages,
                                                        const uint32_t *__restrict__ counts,
                                                        HsegRec *__restrict__ recs)
{
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t pi = blockIdx.x * (kWG / kWave) + (threadIdx.x >> 6);
    if (pi >= n_pages) return;
    const HsegPage pg = pages[pi];
    if (pg.mode != 1u) return;
    HsegRec *R = recs + pg.seg_first;
    const pq_page_desc d = descs[pg.desc];
    // U = runs in front of the first one that is not the first run again
    uint64_t U = 0;
    for (uint32_t base = 0; base < pg.n_checks; base += kWave) {
        const uint32_t j = base + lane;
        const uint32_t c = j < pg.n_checks ? counts[pg.check_first + j] : 0u;
        const uint64_t full = __builtin_amdgcn_ballot_w64(c == (uint32_t)kWave);
        const uint32_t lead = ~full == 0ull ? (uint32_t)kWave : (uint32_t)__builtin_ctzll(~full);
        U += (uint64_t)lead * kWave;
        if (lead < (uint32_t)kWave) {
            U += (uint32_t)__builtin_amdgcn_readlane((int)c, (int)lead);
            break;
        }
    }
    // whole segments of proven runs all of whose values the caller pulls, and at least one value
    // left for the tail
    const uint64_t n = d.n_values;
    const uint64_t runs = n == 0 ? 0 : min(U, (n - 1u) / pg.vpr);
    uint64_t P = runs / pg.runs_per_seg;
    if (P + 1u > pg.n_segs) P = pg.n_segs - 1u;
    const uint64_t seg_stride = (uint64_t)pg.runs_per_seg * pg.stride, seg_vals = (uint64_t)pg.runs_per_seg * pg.vpr;
    for (uint32_t s = lane; s < pg.n_segs; s += kWave) {
        HsegRec r = hseg_rec(0, 0, 0, 0, kSegSkip);
        if (s <= P) {
            r.off = pg.pos0 + s * seg_stride;
            r.v0 = (uint32_t)(s * seg_vals);
            r.count = s < P ? (uint32_t)seg_vals : (uint32_t)(n - s * seg_vals);
            r.flags = s < P ? kSegPart : kSegTail;
            r.len = s < P ? (uint32_t)seg_stride : pg.end - (uint32_t)r.off;
        }
        R[s] = r;
    }
    if (lane == 0) pages[pi].n_part = (uint32_t)P;
}

// ---- mode 2: runs of any kind -----------------------------------------------------------------

// A wave's window on the stream and what a walk needs to know about it (all wave-uniform).
struct LsWalk {
    uint32_t *lds;     // kHwChunk bytes + slack
    const uint8_t *S;  // stream byte 0 is S[0]
    int64_t end;       // offset of the end of the stream
    int64_t c0;        // the window holds [c0, c0 + kHwChunk) once loaded
    bool loaded;
    HwConst hc;
};

__device__ __forceinline__ void ls_window(LsWalk &W, int64_t o, int lane)
{
    if (W.loaded && o >= W.c0 && o + (int64_t)kLsGuard <= W.c0 + (int64_t)kHwChunk) return;
    WavePrefetch pf;
    W.c0 = chunk_base(W.S, o);
    wave_sync();  // earlier reads of the window are done
    wave_fetch(pf, W.S, W.c0, W.end, lane);
    wave_commit_clean(W.lds, pf, W.c0, W.end, lane);
    wave_sync();
    W.loaded = true;
}

// Follows the chain of run headers from offset `from` to the first header at or behind `boundary`:
// at = its offset, or kLsNone when the chain meets something that is not a complete, well-formed
// run inside the stream (dead = the offset of that).  COUNT: acc (per lane, summed by the caller)
// gains the values of the runs passed.  Only headers are read: a bit-packed run's groups are
// stepped over.
struct LsEnd {
    uint32_t at;
    uint32_t dead;
};

// tab != nullptr (COUNT only): entry tn, tn + 1, ... of it receive the offsets, from `from` on, of the
// headers passed (the one at or behind the boundary is not among them); tn may come out above tcap,
// nothing is written beyond it.
template <bool COUNT>
__device__ __forceinline__ LsEnd ls_walk(LsWalk &W, uint32_t from, uint32_t boundary, uint64_t &acc, int lane,
                                         uint16_t *tab = nullptr, uint32_t *tn = nullptr, uint32_t tcap = 0)
{
    // stream offsets are below 2^31 (hseg_header_kernel), so 32-bit unsigned arithmetic holds them
    uint32_t o = from;
    const HwConst &hc = W.hc;
    const uint32_t end = (uint32_t)W.end;
    // every step moves on by at least one byte; the budget only makes that a property of the loop
    // itself (a walk that ran out of it counts as one that met a bad header)
    uint32_t budget = boundary > from ? boundary - from + 8u : 8u;
    while (o < boundary) {
        if (budget-- == 0u) return LsEnd{kLsNone, o};
        ls_window(W, (int64_t)o, lane);
        const uint32_t wo = (uint32_t)((int64_t)o - W.c0);
        const uint32_t p = wo + (uint32_t)lane;
        const uint32_t i = p >> 2, sh = (p & 3u) * 8u;
        const uint32_t d0 = W.lds[i], d1 = W.lds[i + 1], d2 = W.lds[i + 2];
        const uint32_t lo = __builtin_amdgcn_alignbit(d1, d0, sh);  // bytes p .. p+3
        const uint32_t hi = __builtin_amdgcn_alignbit(d2, d1, sh);  // bytes p+4 .. p+7
        const bool two = (lo & 0x80u) != 0u;
        const uint32_t hdr = two ? 2u : 1u;
        const uint32_t h = two ? (lo & 0x7fu) | ((lo >> 1) & 0x3f80u) : lo & 0xffu;
        const uint32_t cnt = h >> 1;  // at most 8191
        const bool isbp = (h & 1u) != 0u;
        const uint32_t val = __builtin_amdgcn_alignbit(hi, lo, 8u * hdr) & hc.vbmask;
        const uint32_t step = hdr + (isbp ? __umul24(cnt, hc.w) : hc.vb);  // bytes of the run: below 2^19
        const uint32_t nvals = isbp ? cnt * 8u : cnt;
        const uint32_t nextlane = (uint32_t)lane + step;
        // In front of a segment (COUNT == false) the walk is looking for the real chain from an arbitrary
        // byte: a bit-packed run of more than 64 groups is taken for what it almost always is there, a
        // byte of packed data read as a header (writers close bit-packed runs after 63 or 64 groups),
        // because following it would jump over the segment's first byte and find nothing.
        const bool plausible = COUNT || !isbp || cnt <= 64u;
        const bool ok = (lo & 0x8080u) != 0x8080u && cnt != 0u && o + nextlane <= end && (isbp || (val & ~hc.wmask) == 0u) &&
                        plausible;
        const uint64_t okmask = __builtin_amdgcn_ballot_w64(ok);
        // the chain of real headers from lane 0 on (see dn_span for the schedule of this loop)
        uint32_t cur;
        uint64_t chain;
        asm volatile("s_mov_b64 %[chain], 0\n\t"
                     "s_mov_b32 %[cur], 0\n"
                     "1:\n\t"
                     "s_bitcmp1_b64 %[ok], %[cur]\n\t"
                     "s_cbranch_scc0 2f\n\t"
                     "s_bitset1_b64 %[chain], %[cur]\n\t"
                     "v_readlane_b32 %[cur], %[next], %[cur]\n\t"
                     "s_cmp_lt_u32 %[cur], 64\n\t"
                     "s_cbranch_scc1 1b\n"
                     "2:"
                     : [chain] "=&s"(chain), [cur] "=&s"(cur)
                     : [ok] "s"(okmask), [next] "v"(nextlane)
                     : "scc");
        const uint32_t brel = boundary - o;  // >= 1
        const uint64_t below = brel < 64u ? ~(~0ull << brel) : ~0ull;
        const uint64_t over = chain & ~below;
        if (COUNT) acc += dn_select(chain & below, nvals);
        if (COUNT && tab != nullptr) {
            const uint64_t in = chain & below;
            const uint32_t slot = *tn + __builtin_amdgcn_mbcnt_hi((uint32_t)(in >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)in, 0u));
            if (((in >> lane) & 1ull) != 0ull && slot < tcap) tab[slot] = (uint16_t)(o + (uint32_t)lane - from);
            *tn += (uint32_t)__builtin_popcountll(in);
        }
        if (over != 0ull) return LsEnd{o + (uint32_t)__builtin_ctzll(over), 0u};
        if (cur >= 64u) {  // the last run of the span ends behind it (inside the stream: cur <= end - o)
            o += cur;
            continue;
        }
        // lane `cur` holds something the span parse does not take: behind the boundary it is the
        // next walk's business, in front of it one more try with headers of up to four bytes
        const uint32_t at = o + cur;
        if (cur >= brel) return LsEnd{at, 0u};
        HwState st;
        st.o = wo + cur;
        st.v = 0;
        st.pend = 0;
        st.pcount = 0;
        st.pskip = 0;
        st.pvalue = 0;
        st.status = PQ_OK;
        const int64_t lim64 = W.end - W.c0;
        const uint32_t lim = lim64 > (int64_t)(1 << 30) ? (1u << 30) : (uint32_t)lim64;
        if (!hw_header_fast(st, W.lds, lim, hc.w)) return LsEnd{kLsNone, at};
        if (!COUNT && st.pend == 2u && st.pcount > 64u) return LsEnd{kLsNone, at};  // see `plausible`
        // the run's end as a 64-bit number: a count of up to 2^27 groups times the width does not fit 32 bits
        uint64_t next = (uint64_t)o + (st.o - wo);
        uint64_t nv = st.pcount;
        if (st.pend == 2u) {
            next += (uint64_t)st.pcount * hc.w;
            nv *= 8u;
        }
        if (next > (uint64_t)end) return LsEnd{kLsNone, at};
        if (COUNT && lane == 0) acc += nv;
        if (COUNT && tab != nullptr) {
            if (lane == 0 && *tn < tcap) tab[*tn] = (uint16_t)(at - from);
            *tn += 1u;
        }
        o = (uint32_t)next;
    }
    return LsEnd{o, 0u};
}

__device__ __forceinline__ void ls_init(LsWalk &W, uint32_t *lds, const uint8_t *S, const HsegPage &pg, int lane)
{
    W.lds = lds;
    W.S = S;
    W.end = (int64_t)pg.end;
    W.c0 = 0;
    W.loaded = false;
    W.hc.w = pg.w;
    W.hc.vb = (pg.w + 7u) >> 3;
    W.hc.vbmask = 0xffffffffu >> (32u - 8u * W.hc.vb);
    W.hc.wmask = pg.w >= 32u ? 0xffffffffu : (1u << pg.w) - 1u;
    if (lane < kHwSlack) lds[kHwChunk / 4 + lane] = 0;
}

// The segment from `entry` on: where the chain leaves it and how many values it passes.
// tab: room for tcap header offsets of this segment's runs (nullptr: none kept); sc.runs says how
// many were written.
__device__ __forceinline__ HsegScan ls_segment(LsWalk &W, uint32_t entry, uint32_t hi, int lane, uint16_t *tab,
                                               uint32_t tcap)
{
    HsegScan sc;
    sc.entry = entry;
    sc.exit = entry;
    sc.nvals = 0;
    sc.runs = 0;
    if (entry == kLsNone || entry >= hi) return sc;  // a long run covers the whole segment: nothing starts here
    uint64_t acc = 0;
    uint32_t tn = 0;
    sc.exit = ls_walk<true>(W, entry, hi, acc, lane, tab, &tn, tcap).at;
    sc.runs = tab != nullptr && tn <= tcap ? tn : 0u;
    const uint64_t total = wave_last_u64(wave_scan_u64(acc));
    sc.nvals = total > 0x7fffffffull ? 0x7fffffffu : (uint32_t)total;
    return sc;
}

constexpr uint32_t kLsLdsWords = kHwChunk / 4 + kHwSlack;

// Segment s of a page keeps the offsets of its run headers in entries [s * seg_bytes / 2, ...) of the
// page's part of the table: a run is two bytes and more (the width is at least 1), so the k-th header
// of a segment lies 2 k bytes and more behind the segment's first byte, and the entries of one
// segment end in front of the next one's.
__device__ __forceinline__ uint16_t *ls_table(uint16_t *tabs, const HsegPage &pg, uint32_t s)
{
    if (tabs == nullptr || pg.tab_first == kLsNone || (pg.seg_bytes & 15u) != 0u || pg.seg_bytes > 65536u) return nullptr;
    return tabs + (size_t)pg.tab_first * 8u + (size_t)s * (pg.seg_bytes / 2u);
}

__global__ __launch_bounds__(kWG) void lseg_scan_kernel(const pq_page_desc *__restrict__ descs,
                                                        const HsegPage *__restrict__ pages,
                                                        const uint32_t *__restrict__ owner, uint32_t n_segs,
                                                        HsegScan *__restrict__ scans, PageBuffers buf,
                                                        uint16_t *__restrict__ tabs)
{
    __shared__ __attribute__((aligned(16))) uint32_t window[kWG / kWave][kLsLdsWords];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t g = blockIdx.x * (kWG / kWave) + (uint32_t)wave;
    if (g >= n_segs) return;
    const HsegPage pg = pages[owner[g]];
    if (pg.mode != 2u) return;
    const uint32_t s = g - pg.seg_first;
    const uint64_t lo64 = (uint64_t)s * pg.seg_bytes;
    HsegScan sc;
    sc.entry = kLsNone;
    sc.exit = kLsNone;
    sc.nvals = 0;
    sc.runs = 0;
    if (lo64 < (uint64_t)pg.end) {
        const pq_page_desc d = descs[pg.desc];
        LsWalk W;
        ls_init(W, window[wave], buf.src + d.src_off, pg, lane);
        const uint32_t lo = max((uint32_t)lo64, pg.pos0);
        const uint32_t hi = (uint32_t)min(lo64 + pg.seg_bytes, (uint64_t)pg.end);
        uint32_t entry = pg.pos0;
        if (s != 0u) {
            // from an arbitrary byte in front of the segment: by the time the walk reaches the
            // segment it has, as a rule, fallen into step with the real chain; whatever it runs
            // into that is no run header makes it start again one byte further on
            const uint32_t pre = pg.w == 1u ? kLsPreRoll / 2u : kLsPreRoll;
            uint32_t o = max(pg.pos0, lo > pre ? lo - pre : 0u);
            uint64_t unused = 0;
            entry = lo;
            for (uint32_t tries = 0; tries <= kLsPreRoll; ++tries) {  // each try starts at least a byte further on
                const LsEnd e = ls_walk<false>(W, o, lo, unused, lane);
                if (e.at != kLsNone) {
                    entry = e.at;
                    break;
                }
                o = max(e.dead, o) + 1u;
                if (o >= lo) break;
            }
        }
        sc = ls_segment(W, entry, hi, lane, ls_table(tabs, pg, s), pg.seg_bytes / 2u);
    }
    if (lane == 0) scans[g] = sc;
}

// in: the scans as first taken; out: the same, but a segment that the one in front leaves at
// another place than it was entered at is walked again from that place.
__global__ __launch_bounds__(kWG) void lseg_fix_kernel(const pq_page_desc *__restrict__ descs,
                                                       const HsegPage *__restrict__ pages,
                                                       const uint32_t *__restrict__ owner, uint32_t n_segs,
                                                       const HsegScan *__restrict__ in, HsegScan *__restrict__ out,
                                                       PageBuffers buf, uint16_t *__restrict__ tabs)
{
    __shared__ __attribute__((aligned(16))) uint32_t window[kWG / kWave][kLsLdsWords];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t g = blockIdx.x * (kWG / kWave) + (uint32_t)wave;
    if (g >= n_segs) return;
    const HsegPage pg = pages[owner[g]];
    if (pg.mode != 2u) return;
    const uint32_t s = g - pg.seg_first;
    HsegScan sc = in[g];
    const uint64_t lo64 = (uint64_t)s * pg.seg_bytes;
    if (s != 0u && lo64 < (uint64_t)pg.end) {
        const uint32_t px = uni(in[g - 1].exit);
        if (px != kLsNone && px != uni(sc.entry)) {
            const pq_page_desc d = descs[pg.desc];
            LsWalk W;
            ls_init(W, window[wave], buf.src + d.src_off, pg, lane);
            const uint32_t hi = (uint32_t)min(lo64 + pg.seg_bytes, (uint64_t)pg.end);
            sc = ls_segment(W, px, hi, lane, ls_table(tabs, pg, s), pg.seg_bytes / 2u);
        }
    }
    if (lane == 0) out[g] = sc;
}

__global__ __launch_bounds__(kWG) void lseg_link_kernel(const pq_page_desc *__restrict__ descs,
                                                        HsegPage *__restrict__ pages, uint32_t n_pages,
                                                        HsegScan *__restrict__ scans, HsegRec *__restrict__ recs,
                                                        PageBuffers buf, uint16_t *__restrict__ tabs)
{
    __shared__ __attribute__((aligned(16))) uint32_t window[kWG / kWave][kLsLdsWords];
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = uni(threadIdx.x >> 6);
    const uint32_t pi = blockIdx.x * (kWG / kWave) + wave;
    if (pi >= n_pages) return;
    const HsegPage pg = pages[pi];
    if (pg.mode != 2u) return;
    HsegRec *R = recs + pg.seg_first;
    HsegScan *SC = scans + pg.seg_first;
    const pq_page_desc d = descs[pg.desc];
    const uint64_t n = d.n_values;
    const uint32_t used = min(pg.n_segs, (uint32_t)(((uint64_t)pg.end + pg.seg_bytes - 1u) / pg.seg_bytes));
    LsWalk W;
    ls_init(W, window[wave], buf.src + d.src_off, pg, (int)lane);
    // K = number of proven segments (a prefix of the page's segments, at least segment 0)
    uint32_t K = used, repairs = 0, prev_exit = pg.pos0, base = 0;
    uint64_t before = 0;  // values in front of the batch
    uint32_t rounds = 2u * used + 64u;  // more than the batches and repairs a page can have
    while (base < used && rounds-- != 0u) {
        const uint32_t s = base + lane;
        const bool valid = s < used;
        HsegScan sc;
        sc.entry = kLsNone;
        sc.exit = kLsNone;
        sc.nvals = 0;
        sc.runs = 0;
        if (valid) {  // records another lane of this wave may have rewritten a moment ago: not through the L1
            sc.entry = __atomic_load_n(&SC[s].entry, __ATOMIC_RELAXED);
            sc.exit = __atomic_load_n(&SC[s].exit, __ATOMIC_RELAXED);
            sc.nvals = __atomic_load_n(&SC[s].nvals, __ATOMIC_RELAXED);
            sc.runs = __atomic_load_n(&SC[s].runs, __ATOMIC_RELAXED);
        }
        uint32_t px = (uint32_t)__shfl_up((int)sc.exit, 1);
        if (lane == 0) px = prev_exit;
        const bool linked = valid && px != kLsNone && sc.entry == px;
        const uint64_t nv = valid ? (uint64_t)sc.nvals : 0ull;
        const uint64_t incl = wave_scan_u64(nv);
        const uint64_t v0 = before + incl - nv;
        const bool wanted = v0 < n;  // the segment's first value is one the caller pulls
        const uint64_t bad = __builtin_amdgcn_ballot_w64(valid && !(linked && wanted));
        const uint32_t first_bad = bad != 0ull ? (uint32_t)__builtin_ctzll(bad) : (uint32_t)kWave;
        if (bad != 0ull) {
            // a link that does not hold, in front of values the caller still wants: the chain in
            // front says where the segment starts, walk it again from there
            const uint32_t want_entry = (uint32_t)__builtin_amdgcn_readlane((int)px, (int)first_bad);
            const bool still = __builtin_amdgcn_readlane((int)(wanted ? 1u : 0u), (int)first_bad) != 0;
            const bool was_linked = __builtin_amdgcn_readlane((int)(linked ? 1u : 0u), (int)first_bad) != 0;
            if (still && !was_linked && want_entry != kLsNone && repairs < kLsRepairs) {
                const uint32_t sk = base + first_bad;
                const uint64_t lo64 = (uint64_t)sk * pg.seg_bytes;
                const uint32_t hi = (uint32_t)min(lo64 + pg.seg_bytes, (uint64_t)pg.end);
                // stepping over a long run costs nothing and does not count
                if (want_entry < hi) repairs += 1u;
                const HsegScan fixed = ls_segment(W, want_entry, hi, (int)lane, ls_table(tabs, pg, sk), pg.seg_bytes / 2u);
                if (lane == 0) SC[sk] = fixed;
                __threadfence();  // lane 0 wrote the record; every lane reads it next
                continue;         // the same 64 segments once more
            }
        }
        if (valid && lane < first_bad)
            R[s] = hseg_rec(sc.entry, sc.exit != kLsNone ? sc.exit - sc.entry : 0u, (uint32_t)v0, sc.nvals, kSegPart, sc.runs);
        if (bad != 0ull) {
            K = base + first_bad;
            break;
        }
        before += wave_last_u64(incl);
        prev_exit = (uint32_t)__builtin_amdgcn_readlane((int)sc.exit, 63);
        base += kWave;
    }
    if (base < used && K == used) K = base;  // out of rounds: what is proven so far stands
    K = uni(K);
    if (K == 0u) {  // nothing is pulled at all: segment 0 is the tail of a decode of no values
        if (lane == 0) R[0] = hseg_rec(pg.pos0, 0, 0, 0, kSegPart);
        K = 1;
    }
    __threadfence();
    // the last proven segment runs to the end of the stream; everything behind it is its business
    for (uint32_t s = lane; s < pg.n_segs; s += kWave) {
        if (s + 1u < K) continue;
        if (s + 1u == K) {
            const HsegRec r = R[s];
            R[s] = hseg_rec(r.off, pg.end - (uint32_t)r.off, r.v0, (uint32_t)(n - (uint64_t)r.v0), kSegTail, r.runs);
        } else {
            R[s] = hseg_rec(0, 0, 0, 0, kSegSkip);
        }
    }
    if (lane == 0) pages[pi].n_part = K - 1u;
}

// ---- decode ---------------------------------------------------------------------------------------

// LDS of the decode kernel: a HybridWaveLds per wave, the dictionary behind
constexpr uint32_t hs_lds_fixed(int waves) { return hs_align16((uint32_t)sizeof(HybridWaveLds) * (uint32_t)waves); }

template <typename OutT>
__device__ __forceinline__ HybridResult hseg_values(HybridWaveLds &L, uint32_t kind_mode, const uint8_t *S, int64_t len,
                                                    uint32_t w, uint32_t n, uint8_t *dst, const uint8_t *gdict,
                                                    const uint32_t *ldict, uint32_t dict_len, uint32_t max_level,
                                                    int lane, const uint16_t *tab, uint32_t runs)
{
    if (kind_mode == HY_DICT) {
        const OutT *dict = ldict ? reinterpret_cast<const OutT *>(ldict) : reinterpret_cast<const OutT *>(gdict);
        return hybrid_wave<HY_DICT, OutT, kTabGiven>(L, S, len, w, n, reinterpret_cast<OutT *>(dst), dict, dict_len, 0, lane, 0, tab,
                                                     runs);
    }
    if (kind_mode == HY_LEVELS)
        return hybrid_wave<HY_LEVELS, uint32_t, kTabGiven>(L, S, len, w, n, reinterpret_cast<uint32_t *>(dst), nullptr, 0, max_level,
                                                           lane, 0, tab, runs);
    return hybrid_wave<HY_RAW, uint32_t, kTabGiven>(L, S, len, w, n, reinterpret_cast<uint32_t *>(dst), nullptr, 0, 0, lane, 0, tab,
                                                    runs);
}

// 128 vector registers: four waves a SIMD, as before the table's code came in
template <int kHsWaves>
__global__ __launch_bounds__(kHsWaves * kWave) __attribute__((amdgpu_waves_per_eu(4, 4))) void hseg_decode_kernel(
                                                            const pq_page_desc *__restrict__ descs,
                                                            const HsegPage *__restrict__ pages,
                                                            const uint32_t *__restrict__ seg_owner, uint32_t n_segs,
                                                            HsegRec *__restrict__ recs, PageBuffers buf,
                                                            uint32_t dict_lds_bytes, uint16_t *__restrict__ tabs)
{
    HybridWaveLds *LW = reinterpret_cast<HybridWaveLds *>(pq_hs_smem);
    // the workgroup's LDS copy of a dictionary: the one its first segment's page reads, if it fits
    uint32_t *ld = reinterpret_cast<uint32_t *>(pq_hs_smem + hs_lds_fixed(kHsWaves));
    uint64_t have_off = 0;
    uint32_t have_bytes = 0;
    {
        const pq_page_desc &d0 = descs[pages[seg_owner[blockIdx.x * kHsWaves]].desc];
        const uint32_t bytes = uni(d0.kind == PQ_KIND_DICT ? d0.dict_len * d0.elem_size : 0u);
        if (dict_lds_bytes != 0u && bytes != 0u && bytes <= dict_lds_bytes) {
            have_off = ((uint64_t)uni((uint32_t)(d0.dict_off >> 32)) << 32) | uni((uint32_t)d0.dict_off);
            have_bytes = bytes;
            const uint32_t *gd = reinterpret_cast<const uint32_t *>(buf.dict + have_off);
            for (uint32_t i = threadIdx.x; i < bytes / 4u; i += kHsWaves * kWave) ld[i] = gd[i];
            __syncthreads();
        }
    }
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t g = blockIdx.x * kHsWaves + (uint32_t)wave;
    if (g >= n_segs) return;
    const HsegPage pg = pages[seg_owner[g]];
    if (pg.mode == 0u) return;  // the one-wave-per-stream kernel decodes the page
    const HsegRec rec = recs[g];
    const uint32_t flags = uni(rec.flags);
    if (flags == kSegSkip) return;
    const pq_page_desc d = descs[pg.desc];
    HybridWaveLds &L = LW[wave];
    const uint8_t *S = buf.src + d.src_off;
    const uint32_t es = pg.kind_mode == HY_DICT ? d.elem_size : 4u;
    uint8_t *dst = pg.kind_mode == HY_LEVELS ? buf.levels + d.def_off : buf.values + d.dst_off;
    const uint8_t *gdict = buf.dict + d.dict_off;
    const uint32_t *ldict = d.kind == PQ_KIND_DICT && have_bytes != 0u && d.dict_off == have_off &&
                                    d.dict_len * d.elem_size == have_bytes
                                ? ld
                                : nullptr;
    const uint64_t off = ((uint64_t)uni((uint32_t)(rec.off >> 32)) << 32) | uni((uint32_t)rec.off);
    const uint32_t v0 = uni(rec.v0), count = uni(rec.count);
    // a proven segment is exactly its runs; the tail is everything from its first byte on
    const int64_t slen = (int64_t)uni(rec.len);
    uint8_t *sdst = dst + (size_t)v0 * es;
    // mode 2: where the segment's run headers are, as the walk that proved it found them
    const uint32_t runs = uni(rec.runs);
    const uint16_t *tab = pg.mode == 2u && runs != 0u ? ls_table(tabs, pg, g - pg.seg_first) : nullptr;
    HybridResult r{PQ_OK, 0, 0, 0};
    if (count != 0u || flags == kSegTail)
        r = es == 8u ? hseg_values<uint64_t>(L, pg.kind_mode, S + off, slen, pg.w, count, sdst, gdict, ldict, d.dict_len,
                                             pg.max_level, lane, tab, runs)
                     : hseg_values<uint32_t>(L, pg.kind_mode, S + off, slen, pg.w, count, sdst, gdict, ldict, d.dict_len,
                                             pg.max_level, lane, tab, runs);
    if (lane == 0) {
        recs[g].status = r.status;
        recs[g].n_decoded = r.n_decoded;
        recs[g].count_max = r.count_max;
        recs[g].consumed = r.consumed;
    }
}

// ---- the page's result from its segments' --------------------------------------------------------

__global__ __launch_bounds__(kWG) void hseg_final_kernel(const HsegPage *__restrict__ pages, uint32_t n_pages,
                                                         const HsegRec *__restrict__ recs,
                                                         pq_page_result *__restrict__ results)
{
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t pi = blockIdx.x * (kWG / kWave) + (threadIdx.x >> 6);
    if (pi >= n_pages) return;
    const HsegPage pg = pages[pi];
    if (pg.mode == 0u) return;  // decoded whole by the one-wave-per-stream kernel, which writes its result
    const HsegRec *R = recs + pg.seg_first;
    const uint32_t P = pg.n_part;
    // the first proven segment (in stream order) that did not end well, if any; the values equal to
    // the maximum level up to it (or up to the tail)
    uint32_t first_bad = 0xffffffffu, maxes = 0;
    for (uint32_t base = 0; base <= P && first_bad == 0xffffffffu; base += kWave) {
        const uint32_t s = base + lane;
        const bool part = s < P;
        const bool bad = part && (R[s].status != PQ_OK || R[s].n_decoded != R[s].count);
        const uint64_t m = __builtin_amdgcn_ballot_w64(bad);
        if (m != 0ull) first_bad = base + (uint32_t)__builtin_ctzll(m);
        const uint32_t upto = first_bad != 0xffffffffu ? first_bad : P;
        maxes += wave_sum_u32(s <= upto ? R[s].count_max : 0u);
    }
    if (lane != 0) return;
    const HsegRec t = R[first_bad != 0xffffffffu ? first_bad : P];
    pq_page_result out;
    out.status = t.status;
    out.stage = PQ_STAGE_NONE;
    out.detail = 0;
    // DictDecoder.DecodeValues reports count 0 with "invalid index" (types/dict.go:104-111)
    out.n_decoded = t.status == PQ_DICT_INVALID_INDEX ? 0u : t.v0 + t.n_decoded;
    out.not_null = maxes;
    out.n_levels = 0;
    out.consumed = pg.fixed_consumed != kLsNone ? (uint64_t)pg.fixed_consumed : t.off + t.consumed;
    if (pg.kind_mode == HY_LEVELS && t.status != PQ_OK) {
        // decodePackedArray returns (nil, 0, err) on any error (layout/helpers.go:78-80)
        out.n_decoded = 0;
        out.not_null = 0;
    }
    results[pg.desc] = out;
}

hipError_t launch_seg_hybrid(hipStream_t s, const pq_page_desc *d_descs, HsegPage *d_pages, uint32_t n_pages,
                             const uint32_t *d_check_owner, uint32_t n_checks, uint32_t *d_counts,
                             const uint32_t *d_seg_owner, uint32_t n_segs, HsegRec *d_recs, HsegScan *d_scans,
                             PageBuffers buf, pq_page_result *d_results, uint32_t dict_lds_bytes, uint16_t *d_tabs,
                             int fix_rounds)
{
    if (n_pages == 0 || n_segs == 0) return hipSuccess;
    const uint32_t per = kWG / kWave;
    const dim3 gp((n_pages + per - 1) / per), gs((n_segs + per - 1) / per);
    hipLaunchKernelGGL(hseg_header_kernel, dim3((n_pages + kWave - 1) / kWave), dim3(kWave), 0, s, d_descs, d_pages, n_pages,
                       buf);
    if (n_checks != 0)
        hipLaunchKernelGGL(hseg_check_kernel, dim3((n_checks + per - 1) / per), dim3(kWG), 0, s, d_descs, d_pages,
                           d_check_owner, n_checks, d_counts, buf);
    hipLaunchKernelGGL(hseg_link_kernel, gp, dim3(kWG), 0, s, d_descs, d_pages, n_pages, d_counts, d_recs);
    // d_scans: two arrays of n_segs records, the second the first after the fix pass
    hipLaunchKernelGGL(lseg_scan_kernel, gs, dim3(kWG), 0, s, d_descs, d_pages, d_seg_owner, n_segs, d_scans, buf, d_tabs);
    // A segment entered at the wrong place usually still leaves at the right one (its walk falls into
    // step inside it), so the first round sets most entries right; a second one sets right the few whose
    // predecessor's exit was itself wrong at first.  What is left the link pass repairs itself, a page's
    // repairs one after the other: with streams of narrow values (level streams, booleans) that is one
    // repair for a page at most and the second round is not worth its launch (64 definition-level streams
    // of 1 M: 0.26 -> 0.24 ms without it); with 12-bit keys it is (0.35 -> 0.40 ms without it).
    hipLaunchKernelGGL(lseg_fix_kernel, gs, dim3(kWG), 0, s, d_descs, d_pages, d_seg_owner, n_segs, d_scans, d_scans + n_segs,
                       buf, d_tabs);
    HsegScan *fixed = d_scans + n_segs;
    if (fix_rounds > 1) {
        hipLaunchKernelGGL(lseg_fix_kernel, gs, dim3(kWG), 0, s, d_descs, d_pages, d_seg_owner, n_segs, d_scans + n_segs,
                           d_scans, buf, d_tabs);
        fixed = d_scans;
    }
    hipLaunchKernelGGL(lseg_link_kernel, gp, dim3(kWG), 0, s, d_descs, d_pages, n_pages, fixed, d_recs, buf, d_tabs);
    if (dict_lds_bytes != 0u) {
        const size_t lds = hs_lds_fixed(kHsWavesDict) + dict_lds_bytes;
        if (lds > 65536) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(hseg_decode_kernel<kHsWavesDict>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(hseg_decode_kernel<kHsWavesDict>, dim3((n_segs + kHsWavesDict - 1) / kHsWavesDict),
                           dim3(kHsWavesDict * kWave), lds, s, d_descs, d_pages, d_seg_owner, n_segs, d_recs, buf,
                           dict_lds_bytes, d_tabs);
    } else {
        hipLaunchKernelGGL(hseg_decode_kernel<kHsWavesPlain>, dim3((n_segs + kHsWavesPlain - 1) / kHsWavesPlain),
                           dim3(kHsWavesPlain * kWave), hs_lds_fixed(kHsWavesPlain), s, d_descs, d_pages, d_seg_owner,
                           n_segs, d_recs, buf, 0u, d_tabs);
    }
    hipLaunchKernelGGL(hseg_final_kernel, gp, dim3(kWG), 0, s, d_pages, n_pages, d_recs, d_results);
    return hipGetLastError();
}

}  // namespace pq
