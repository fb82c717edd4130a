// hybrid_dense.hpp — the tiles in the middle of a run-dense hybrid stream.
//
// Included by hybrid_wave.hpp, whose window, state and sink it shares.  A stream of short runs
// (level streams, run-heavy dictionary keys, BOOLEAN RLE) is bound by the number of instructions a
// tile costs, so the tiles that need none of the reference's checks have a loop of their own:
//
//   * a run is ONE kind of piece, {bit address of (virtual) tile position 0, bits per position}: a
//     bit-packed run steps by the bit width, an RLE run steps by 0 and points at its value where it
//     lies in the window (the value of a run carried over from an earlier tile is parked in a word
//     behind the window) — the expansion has no case distinction per value;
//   * run headers are parsed 64 byte positions at a time (one- and two-byte headers; the chain of
//     real headers is followed on the scalar unit); whole runs that fit the tile are appended by
//     their own lanes, the first run that does not fit becomes the pending run;
//   * ONE place cuts runs: the pending run is cut at the end of the tile (in the middle of a group
//     if need be: HwState::pskip) and at the end of the window, so a tile is full whenever the
//     window can feed it, and every store of such a tile is a whole 16-byte piece;
//   * anything unusual — a header of three bytes and more, an empty run, an RLE value too large,
//     the last group of a stream, the end of the stream — ends the loop with the state untouched
//     at that point, and the checked tile of hybrid_wave reproduces what the reference does
//     (encoding/hybrid-decoder.go:70-170).
#pragma once

namespace pq {

enum DnReason : uint32_t {
    DN_DONE = 0,     // n values are out
    DN_WINDOW = 1,   // the window is used up
    DN_GENERIC = 2,  // the checked tile has to look at what comes next
    DN_BULK = 3,     // a fresh bit-packed run of 32 groups and more is pending: the bulk path takes it
    DN_NONE = 4,
};

constexpr uint32_t kDnOdd = 0xffffffffu;
// runs in a span of 64 bytes from which on the chain of headers is followed by pointer doubling
constexpr uint32_t kDnDenseRuns = 16;
// bit address of the first of the two words that hold RLE values of carried-over runs
constexpr uint32_t kDnConstBit = (uint32_t)(kHwChunk + 4 * kHwSlack) * 8u;
// highest bit address a position outside the tile is clamped to
constexpr uint32_t kDnMaxBit = kDnConstBit + 32u;

// lane's bit of `mask` set ? a : 0, in one instruction
__device__ __forceinline__ uint32_t dn_select(uint64_t mask, uint32_t a)
{
    uint32_t r;
    asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(r) : "v"(a), "s"(mask));
    return r;
}

struct DnNext {
    uint32_t kind;   // 0: no run pending; 1: RLE; 2: bit-packed
    uint32_t count;  // repeats | groups
    uint32_t value;  // RLE value
    uint32_t o;      // kind == 0: window offset of the next header; else: just behind the pending run's header (and value)
};

// The chain of run headers of a span from lane 0 on: chain = the lanes that hold real headers, cur = where
// it leaves the span (>= 64) or the lane of the first byte on it that is no header.  ok: this lane's byte
// parses as a simple header; nextlane: the lane its run ends in front of.
__device__ __forceinline__ void dn_chain(bool ok, uint32_t nextlane, uint64_t okmask, bool dense, int lane,
                                         uint64_t &chain, uint32_t &cur)
{
    if (dense) {
        // Many short runs (RLE-dense streams: a run every three bytes): the chain by pointer doubling
        // on the vector unit instead of a step per run.  Lane l keeps the set of byte positions the
        // chain from l visits in 2^k steps and where it stands after them; five rounds cover the 32
        // runs a span can hold at most (a run is two bytes or more).  ~60 instructions whatever the
        // number of runs, against six a run below.
        uint32_t J = ok ? min(nextlane, 64u) : 64u;  // 64: the chain from here has left the span or met no header
        uint32_t mlo = lane < 32 ? 1u << lane : 0u, mhi = lane < 32 ? 0u : 1u << (lane - 32);
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int at = (int)(J << 2);
            const uint32_t glo = (uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)mlo);
            const uint32_t ghi = (uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)mhi);
            const uint32_t gj = (uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)J);
            const bool on = J < 64u;
            mlo |= on ? glo : 0u;
            mhi |= on ? ghi : 0u;
            J = on ? gj : 64u;
        }
        const uint64_t visited = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)mhi) << 32) |
                                 (uint32_t)__builtin_amdgcn_readfirstlane((int)mlo);
        // the walk ends on a byte that is no header (it is visited, not on the chain), or behind the span
        const uint32_t last = 63u - (uint32_t)__builtin_clzll(visited);
        chain = visited & okmask;
        cur = ((okmask >> last) & 1ull) != 0ull ? (uint32_t)__builtin_amdgcn_readlane((int)nextlane, (int)last) : last;
    } else {
        // five scalar instructions between the v_readlane that writes `cur` and the one that uses it as
        // its lane select cover the wait states that dependency needs
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
    }
}

// Parse candidate headers at window offsets o .. o + 63, follow the chain of real ones from lane 0,
// append the leading runs that lie wholly in the window and fit `room` values to the tile (each by
// its own lane), describe what follows in nx.  Returns the number of runs appended, kDnOdd when the
// header at o itself is not a simple one.
// dense: the span in front of this one held many runs (in/out; see kDnDenseRuns).
__device__ __forceinline__ uint32_t dn_span(HybridWaveLds &L, uint32_t o, uint32_t wl, const HwConst &hc, uint32_t room,
                                            uint32_t np, uint32_t fill, uint32_t phase, int lane, uint32_t &values,
                                            DnNext &nx, bool &dense)
{
    const uint32_t p = o + (uint32_t)lane;
    const uint32_t i = p >> 2, sh = (p & 3u) * 8u;
    const uint32_t d0 = L.window[i], d1 = L.window[i + 1], d2 = L.window[i + 2];
    const uint32_t lo = __builtin_amdgcn_alignbit(d1, d0, sh);  // bytes p .. p+3
    const uint32_t hi = __builtin_amdgcn_alignbit(d2, d1, sh);  // bytes p+4 .. p+7
    const bool two = (lo & 0x80u) != 0u;
    const uint32_t hdr = two ? 2u : 1u;
    const uint32_t h = two ? (lo & 0x7fu) | ((lo >> 1) & 0x3f80u) : lo & 0xffu;
    const uint32_t cnt = h >> 1;
    const bool isbp = (h & 1u) != 0u;
    const uint32_t val = __builtin_amdgcn_alignbit(hi, lo, 8u * hdr) & hc.vbmask;
    const uint32_t body = isbp ? __umul24(cnt, hc.w) : hc.vb;
    const uint32_t hend = p + hdr + (isbp ? 0u : hc.vb);  // end of the header and, for RLE, its value
    const uint32_t nvals = isbp ? cnt * 8u : cnt;
    const bool ok = (lo & 0x8080u) != 0x8080u && cnt != 0u && hend <= wl && (isbp || (val & ~hc.wmask) == 0u);
    const bool whole = p + hdr + body <= wl;
    const uint32_t nextlane = (uint32_t)lane + hdr + body;
    const uint64_t okmask = __builtin_amdgcn_ballot_w64(ok);

    uint32_t cur;
    uint64_t chain;
    dn_chain(ok, nextlane, okmask, dense, lane, chain, cur);
    dense = hc.w != 0u && (uint32_t)__builtin_popcountll(chain) >= kDnDenseRuns;
    values = 0;
    nx.kind = 0;
    nx.count = 0;
    nx.value = 0;
    nx.o = o;
    if (chain == 0ull) return kDnOdd;

    const uint32_t incl = wave_scan_u32(dn_select(chain, nvals));
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(chain >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)chain, 0u));
    const bool fit = dn_select(chain, 1u) != 0u && whole && incl <= room && np + rank < (uint32_t)kWave;
    const uint64_t tmask = __builtin_amdgcn_ballot_w64(fit);
    if (tmask != 0ull) {
        const uint32_t last = 63u - (uint32_t)__builtin_clzll(tmask);
        values = (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)last);
    }
    const uint64_t untaken = chain & ~tmask;
    if (untaken != 0ull) {
        const uint32_t u = (uint32_t)__builtin_ctzll(untaken);
        if (values < room) {
            const uint32_t hu = (uint32_t)__builtin_amdgcn_readlane((int)h, (int)u);
            const uint32_t eu = (uint32_t)__builtin_amdgcn_readlane((int)hend, (int)u);
            nx.kind = (hu & 1u) != 0u ? 2u : 1u;
            nx.count = hu >> 1;
            nx.value = (uint32_t)__builtin_amdgcn_readlane((int)val, (int)u);
            nx.o = eu;
        } else {
            // the runs taken fill the room to the last value: this header is not read yet (the reference
            // reads a header when it needs the run's first value; a pull may end right here)
            nx.o = o + u;
        }
    } else {
        nx.o = o + cur;
    }
    if (fit) {
        const uint32_t start = fill + incl - nvals;
        const uint32_t at = (p + hdr) * 8u;
        L.pieces[np + rank] = make_uint2(isbp ? at - __umul24(start, hc.w) : at, isbp ? hc.w : 0u);
        reinterpret_cast<uint8_t *>(L.marks)[phase + start] = 1;
    }
    return (uint32_t)__builtin_popcountll(tmask);
}

// The run headers of a stream as a walk in front of the decode found them (seghybrid.hip: the scan
// of a segment notes the offset of every header it passes, 16 bits each, from the segment's first
// header on).  The decode still reads every header itself; what the table saves is finding them.
// Two sources: a table in memory, written by the scan of a segment in front of this decode (offsets from
// the stream's first byte; kDnTabCache entries of it at a time in LDS), or the wave's own walk over the
// window it is about to decode (hw_window_scan: window offsets, all of them in LDS, tab == nullptr).
struct DnTable {
    const uint16_t *tab;  // 16-byte aligned; nullptr: the entries in LDS are all there are
    uint32_t tn;          // entries; 0: none (or given up: the stream is not where the table says)
    uint32_t ti;          // entry of the next header, if the walk and the table are in step
    uint16_t *cache;      // kDnTabCache entries in LDS: [cbase, cbase + kDnTabCache) of the table
    uint32_t cbase;       // a multiple of 8
};
constexpr uint32_t kDnNoTable = 0xfffffffeu;
enum DnTableMode : int { kTabNone = 0, kTabGiven = 1, kTabScan = 2 };

__device__ __forceinline__ void dn_table_fill(DnTable &T, uint32_t base, int lane)
{
    wave_sync();  // earlier reads of the cache are done
    const u32x4_t *src = reinterpret_cast<const u32x4_t *>(T.tab + base);
    constexpr uint32_t kLoads = kDnTabCache / (8u * (uint32_t)kWave);  // 16-byte loads a lane
    u32x4_t v[kLoads];
#pragma unroll
    for (uint32_t k = 0; k < kLoads; ++k) {
        const uint32_t q = (uint32_t)lane + (uint32_t)kWave * k;
        v[k] = u32x4_t{0u, 0u, 0u, 0u};
        if (base + 8u * q < T.tn) v[k] = src[q];  // the table's room is whole 16-byte pieces
    }
#pragma unroll
    for (uint32_t k = 0; k < kLoads; ++k) reinterpret_cast<u32x4_t *>(T.cache)[(uint32_t)lane + (uint32_t)kWave * k] = v[k];
    wave_sync();
    T.cbase = base;
}

// entry ti + lane, 0xffffffff behind the table's end
__device__ __forceinline__ uint32_t dn_table_load(DnTable &T, uint32_t ti, int lane)
{
    if (T.tab != nullptr && (ti < T.cbase || ti + (uint32_t)kWave > T.cbase + kDnTabCache)) dn_table_fill(T, ti & ~7u, lane);
    const uint32_t idx = ti + (uint32_t)lane;
    return idx < T.tn ? (uint32_t)T.cache[idx - T.cbase] : 0xffffffffu;
}

// dn_span with the headers' places taken from the table: lane l parses run T.ti + l, so a step
// covers up to 64 runs whatever bytes they take, and no chain is followed.  c0 = stream offset of
// window byte 0.  Returns kDnNoTable (nothing changed, T.tn = 0) when the header at o is not in the
// table: the caller goes on finding headers itself.
__device__ __forceinline__ uint32_t dn_span_table(HybridWaveLds &L, uint32_t o, uint32_t wl, const HwConst &hc,
                                                  uint32_t room, uint32_t np, uint32_t fill, uint32_t phase, int lane,
                                                  uint32_t &values, DnNext &nx, DnTable &T, int32_t c0)
{
    const uint32_t here = o + (uint32_t)c0;  // stream offset of the header at o
    uint32_t rel = dn_table_load(T, T.ti, lane);
    if ((uint32_t)__builtin_amdgcn_readfirstlane((int)rel) != here) {
        // something else took runs since the last step (the bulk path, a checked tile): the entries
        // are in stream order, look for this header further on
        uint64_t m = 0;
        for (uint32_t tries = 0; tries < 8u; ++tries) {
            m = __builtin_amdgcn_ballot_w64(rel >= here);  // lanes behind the table's end hold 0xffffffff
            if (m != 0ull) break;
            T.ti += (uint32_t)kWave;
            rel = dn_table_load(T, T.ti, lane);
        }
        const uint32_t k = m != 0ull ? (uint32_t)__builtin_ctzll(m) : 0u;
        if (m == 0ull || (uint32_t)__builtin_amdgcn_readlane((int)rel, (int)k) != here) {
            T.tn = 0;
            return kDnNoTable;
        }
        T.ti += k;
        if (k != 0u) rel = dn_table_load(T, T.ti, lane);
    }
    const uint32_t pr = rel - (uint32_t)c0;  // window offset of this lane's header
    const bool inwin = rel != 0xffffffffu && pr < wl;
    const uint32_t p = inwin ? pr : o;
    const uint32_t i = p >> 2, sh = (p & 3u) * 8u;
    const uint32_t d0 = L.window[i], d1 = L.window[i + 1], d2 = L.window[i + 2];
    const uint32_t lo = __builtin_amdgcn_alignbit(d1, d0, sh);  // bytes p .. p+3
    const uint32_t hi = __builtin_amdgcn_alignbit(d2, d1, sh);  // bytes p+4 .. p+7
    const bool two = (lo & 0x80u) != 0u;
    const uint32_t hdr = two ? 2u : 1u;
    const uint32_t h = two ? (lo & 0x7fu) | ((lo >> 1) & 0x3f80u) : lo & 0xffu;
    const uint32_t cnt = h >> 1;
    const bool isbp = (h & 1u) != 0u;
    const uint32_t val = __builtin_amdgcn_alignbit(hi, lo, 8u * hdr) & hc.vbmask;
    const uint32_t body = isbp ? __umul24(cnt, hc.w) : hc.vb;
    const uint32_t hend = p + hdr + (isbp ? 0u : hc.vb);  // end of the header and, for RLE, its value
    const uint32_t rend = p + hdr + body;                 // end of the run
    const uint32_t nvals = isbp ? cnt * 8u : cnt;
    const bool ok = inwin && (lo & 0x8080u) != 0x8080u && cnt != 0u && hend <= wl && (isbp || (val & ~hc.wmask) == 0u);
    const uint64_t okmask = __builtin_amdgcn_ballot_w64(ok);
    const uint64_t wholemask = __builtin_amdgcn_ballot_w64(ok && rend <= wl);
    values = 0;
    nx.kind = 0;
    nx.count = 0;
    nx.value = 0;
    nx.o = o;
    if ((okmask & 1ull) == 0ull) return kDnOdd;  // T stays: a checked tile takes this run, the next step finds its place again
    // the leading runs that lie wholly in the window
    const uint32_t lead = ~wholemask != 0ull ? (uint32_t)__builtin_ctzll(~wholemask) : (uint32_t)kWave;
    const uint64_t leadmask = lead >= 64u ? ~0ull : (1ull << lead) - 1ull;
    const uint32_t incl = wave_scan_u32(dn_select(leadmask, nvals));
    const bool fit = (uint32_t)lane < lead && incl <= room && np + (uint32_t)lane < (uint32_t)kWave;
    const uint64_t tmask = __builtin_amdgcn_ballot_w64(fit);  // lanes 0 .. taken - 1
    const uint32_t taken = (uint32_t)__builtin_popcountll(tmask);
    if (taken != 0u) {
        values = (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)(taken - 1u));
        nx.o = (uint32_t)__builtin_amdgcn_readlane((int)rend, (int)(taken - 1u));
    }
    uint32_t adv = taken;
    if (taken < (uint32_t)kWave && ((okmask >> taken) & 1ull) != 0ull && values < room) {
        // a well-formed run that does not fit (the tile, the window, the piece table): the caller cuts it.
        // Not when the runs taken fill the room to the last value: the reference reads a header when it
        // needs the run's first value, and a pull that ends here has not read this one.
        const uint32_t hu = (uint32_t)__builtin_amdgcn_readlane((int)h, (int)taken);
        nx.kind = (hu & 1u) != 0u ? 2u : 1u;
        nx.count = hu >> 1;
        nx.value = (uint32_t)__builtin_amdgcn_readlane((int)val, (int)taken);
        nx.o = (uint32_t)__builtin_amdgcn_readlane((int)hend, (int)taken);
        adv += 1u;
    }
    if (fit) {
        const uint32_t start = fill + incl - nvals;
        const uint32_t at = (p + hdr) * 8u;
        L.pieces[np + (uint32_t)lane] = make_uint2(isbp ? at - __umul24(start, hc.w) : at, isbp ? hc.w : 0u);
        reinterpret_cast<uint8_t *>(L.marks)[phase + start] = 1;
    }
    T.ti += adv;
    return taken;
}

// 32 bits of the window from bit address `bit` on
__device__ __forceinline__ uint32_t dn_read(const uint32_t *lds, uint32_t bit)
{
    const uint32_t i = bit >> 5;
    return __builtin_amdgcn_alignbit(lds[i + 1], lds[i], bit & 31u);
}

// Expand the tile: positions [0, clip) go to dst[v ... v + clip).  posv[q][k] is the tile position of
// this lane's value k of quad q (256 q + 4 lane + k - phase: quads are 4-element aligned in memory).
// RAGGED = false: phase == 0 and clip == 256 Q, every quad is whole.  SOLO: the tile is the one
// piece {y0, s0} (a bit-packed one).
template <int MODE, typename OutT, int Q, bool RAGGED, bool SOLO>
__device__ __forceinline__ void dn_expand(HybridWaveLds &L, const HwSink<OutT> &sink, uint32_t wmask,
                                          const uint32_t (&posv)[Q][4], uint32_t y0, uint32_t s0, uint32_t clip,
                                          uint32_t phase, uint32_t v, int lane, uint32_t &local_count, uint32_t &scount,
                                          bool &bad_index)
{
    uint32_t key[Q][4];
    if (SOLO) {
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t bit = __umul24(posv[q][k], s0) + y0;
                if (RAGGED) bit = min(bit, kDnMaxBit);
                key[q][k] = dn_read(L.window, bit) & wmask;
            }
    } else {
        uint32_t x[Q], packed = 0;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            uint32_t m = L.marks[kWave * q + lane];
            m += m << 8;
            asm volatile("" : "+v"(m));  // two shift-adds, not a multiplication by 0x01010101 (quarter rate)
            m += m << 16;  // byte k = pieces started at or before position k of this quad
            x[q] = m;
            packed |= (m >> 24) << (8 * q);
        }
        const uint32_t incl = wave_scan_u32(packed);
        const uint32_t tot = wave_last_u32(incl);
        uint32_t before = 0xffffffffu;  // pieces started in the quads of lower q, less one
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const uint32_t excl = before + ((incl >> (8 * q)) & 0xffu) - (x[q] >> 24);
            before += (tot >> (8 * q)) & 0xffu;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t id = excl + ((x[q] >> (8 * k)) & 0xffu);
                if (RAGGED) id &= (uint32_t)(kWave - 1);
                const uint2 rec = L.pieces[id];
                uint32_t bit = __umul24(posv[q][k], rec.y) + rec.x;
                if (RAGGED) bit = min(bit, kDnMaxBit);
                key[q][k] = dn_read(L.window, bit) & wmask;
            }
        }
    }
    OutT *const qp0 = sink.dst + ((int64_t)v - (int64_t)phase + 4 * (int64_t)lane);  // quad q: + 256 q
    const uint32_t dmax = sink.dict_len - 1u;  // no_dict: wraps, every comparison below still holds
    if (!RAGGED) {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            OutT out[4];
            if (MODE == HY_DICT) {
                const uint32_t top = max(max(key[q][0], key[q][1]), max(key[q][2], key[q][3]));
                bad_index |= top > dmax || sink.no_dict;
#pragma unroll
                for (int k = 0; k < 4; ++k) out[k] = sink.dict[sink.no_dict ? 0u : min(key[q][k], dmax)];
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) out[k] = (OutT)key[q][k];
                if (MODE == HY_LEVELS) {
                    // counted on the scalar unit (a compare, a bit count, an addition): the vector unit is
                    // the busy one here
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        scount += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(key[q][k] == sink.max_level));
                }
            }
            hw_store4<OutT>(qp0 + 256 * q, out, 0xfu, true);
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        OutT out[4];
        uint32_t valid = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool ok = posv[q][k] < clip;  // also false "below zero"
            valid |= ok ? 1u << k : 0u;
            if (MODE == HY_DICT) {
                const bool inside = key[q][k] <= dmax && !sink.no_dict;
                bad_index |= ok && !inside;
                out[k] = sink.dict[inside ? key[q][k] : 0u];
            } else {
                out[k] = (OutT)key[q][k];
                if (MODE == HY_LEVELS) local_count += (ok && key[q][k] == sink.max_level) ? 1u : 0u;
            }
        }
        if (valid != 0u) hw_store4<OutT>(qp0 + 256 * q, out, valid, true);
    }
}

// A tile that is one RLE piece: a constant.
template <int MODE, typename OutT, int Q>
__device__ __forceinline__ void dn_expand_const(const HwSink<OutT> &sink, uint32_t key, const uint32_t (&posv)[Q][4],
                                                uint32_t clip, uint32_t phase, uint32_t v, int lane, uint32_t &scount,
                                                bool &bad_index)
{
    OutT o;
    if (MODE == HY_DICT) {
        const bool inside = key < sink.dict_len && !sink.no_dict;
        if (!inside) bad_index = true;
        o = sink.dict[inside ? key : 0u];
    } else {
        o = (OutT)key;
        if (MODE == HY_LEVELS && key == sink.max_level) scount += clip;
    }
    OutT out[4] = {o, o, o, o};
    OutT *const qp0 = sink.dst + ((int64_t)v - (int64_t)phase + 4 * (int64_t)lane);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        uint32_t valid = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) valid |= posv[q][k] < clip ? 1u << k : 0u;
        if (valid != 0u) hw_store4<OutT>(qp0 + 256 * q, out, valid, true);
    }
}

// The wave's own table: the chain of run headers of the window from window offset `from` on, as far as
// it is made of simple headers (one or two bytes, the header and an RLE value inside the valid bytes),
// up to the first header at or behind `stop` and at most cap entries.  Window offsets, ascending.
// What the walk costs is a span parse per 64 bytes and six scalar instructions a run; what it saves is
// every other piece of bookkeeping of the tiles that find their runs as they go.
__device__ __forceinline__ uint32_t hw_window_scan(HybridWaveLds &L, uint32_t from, uint32_t stop, uint32_t wl,
                                                   const HwConst &hc, uint16_t *tab, uint32_t cap, int lane)
{
    uint32_t o = from, tn = 0;
    bool dense = false;  // the span in front held many runs: the next one's chain by pointer doubling
    while (o < stop && tn + (uint32_t)kWave <= cap) {
        const uint32_t p = o + (uint32_t)lane;
        const uint32_t i = p >> 2, sh = (p & 3u) * 8u;
        const uint32_t d0 = L.window[i], d1 = L.window[i + 1], d2 = L.window[i + 2];
        const uint32_t lo = __builtin_amdgcn_alignbit(d1, d0, sh);  // bytes p .. p+3
        const uint32_t hi = __builtin_amdgcn_alignbit(d2, d1, sh);  // bytes p+4 .. p+7
        const bool two = (lo & 0x80u) != 0u;
        const uint32_t hdr = two ? 2u : 1u;
        const uint32_t h = two ? (lo & 0x7fu) | ((lo >> 1) & 0x3f80u) : lo & 0xffu;
        const uint32_t cnt = h >> 1;
        const bool isbp = (h & 1u) != 0u;
        const uint32_t val = __builtin_amdgcn_alignbit(hi, lo, 8u * hdr) & hc.vbmask;
        const uint32_t hend = p + hdr + (isbp ? 0u : hc.vb);
        const bool ok = (lo & 0x8080u) != 0x8080u && cnt != 0u && hend <= wl && (isbp || (val & ~hc.wmask) == 0u);
        const uint32_t nextlane = (uint32_t)lane + hdr + (isbp ? __umul24(cnt, hc.w) : hc.vb);
        const uint64_t okmask = __builtin_amdgcn_ballot_w64(ok);
        uint32_t cur;
        uint64_t chain;
        dn_chain(ok, nextlane, okmask, dense, lane, chain, cur);
        dense = (uint32_t)__builtin_popcountll(chain) >= kDnDenseRuns;
        const uint32_t brel = stop - o;  // >= 1
        const uint64_t in = brel < 64u ? chain & ~(~0ull << brel) : chain;
        if (in == 0ull) break;
        const uint32_t slot = tn + __builtin_amdgcn_mbcnt_hi((uint32_t)(in >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)in, 0u));
        if (((in >> lane) & 1ull) != 0ull) tab[slot] = (uint16_t)p;
        tn += (uint32_t)__builtin_popcountll(in);
        if (in != chain || cur < 64u) break;  // the header at `stop`, or something the span parse does not take
        o += cur;
    }
    return tn;
}

// Whole tiles straight from the table of run headers, for as long as the table, the window and the
// tile agree: lane l parses run ti + l, a prefix sum of the runs' lengths says which of them reach
// into the tile [v, v + 256 Q), each of those writes its own piece (a run the tile's first value lies
// inside is a piece whose position 0 is in front of the tile; a run that goes on behind the tile is
// simply met again by the next one), and the tile is expanded.  No run is ever "pending" here: where
// a tile ends inside a run, the next tile starts from the same table entry.  Whatever does not fit —
// a header of three bytes and more, more than 64 runs in a tile, a tile the window does not hold
// whole, the stream's last values — ends the loop with st describing exactly what is out, in the
// terms of the loops below (which take the tile and come back).
// posv: the positions of phase 0 (256 q + 4 lane + k).
template <int MODE, typename OutT, int Q>
__device__ __forceinline__ void hw_table_tiles(HybridWaveLds &L, HwState &st, const HwSink<OutT> &sink,
                                               const HwConst &hc, uint32_t n, uint32_t wl, int lane,
                                               uint32_t &local_count, uint32_t &scount, bool &bad_index, DnTable &T,
                                               int32_t c0, const uint32_t (&posv)[Q][4])
{
    constexpr uint32_t TILE = 256u * Q;
    // ---- the run the walk stands at or in: the last header in front of `probe` ----
    const uint32_t here = st.o + (uint32_t)c0;  // pend == 0: the next header; else a byte behind the pending run's header
    const uint32_t probe = st.pend == 0u ? here + 1u : here;
    uint32_t base = T.ti != 0u ? T.ti - 1u : 0u;
    uint32_t found = 0xffffffffu;
    for (uint32_t tries = 0; tries < 8u; ++tries) {
        const uint32_t rel = dn_table_load(T, base, lane);
        const uint32_t c = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(rel < probe));  // entries are in order
        if (c == 0u) break;  // in front of where the search began: not this table's business
        if (c < (uint32_t)kWave) {
            found = base + c - 1u;
            break;
        }
        base += (uint32_t)kWave - 1u;  // all 64 lie in front: the last of them may be the one
    }
    if (found == 0xffffffffu) return;  // e.g. a pending run whose header an earlier window held: the loops below finish it
    uint32_t ti = found;        // the table entry of the run that holds value v (or starts with it)
    uint32_t v = st.v;
    uint32_t a = st.v;          // index of that run's first value
    uint32_t o_cur = 0;         // window offset of its header
    bool first = true;
    for (;;) {
        const uint32_t rel = dn_table_load(T, ti, lane);
        const uint32_t pr = rel - (uint32_t)c0;
        const bool inwin = rel != 0xffffffffu && pr < wl;
        const uint32_t p = inwin ? pr : 0u;
        const uint32_t i = p >> 2, sh = (p & 3u) * 8u;
        const uint32_t d0 = L.window[i], d1 = L.window[i + 1], d2 = L.window[i + 2];
        const uint32_t lo = __builtin_amdgcn_alignbit(d1, d0, sh);  // bytes p .. p+3
        const uint32_t hi = __builtin_amdgcn_alignbit(d2, d1, sh);  // bytes p+4 .. p+7
        const bool two = (lo & 0x80u) != 0u;
        const uint32_t hdr = two ? 2u : 1u;
        const uint32_t h = two ? (lo & 0x7fu) | ((lo >> 1) & 0x3f80u) : lo & 0xffu;
        const uint32_t cnt = h >> 1;
        const bool isbp = (h & 1u) != 0u;
        const uint32_t val = __builtin_amdgcn_alignbit(hi, lo, 8u * hdr) & hc.vbmask;
        const uint32_t hend = p + hdr + (isbp ? 0u : hc.vb);
        const uint32_t rend = p + hdr + (isbp ? __umul24(cnt, hc.w) : hc.vb);
        const uint32_t nvals = isbp ? cnt * 8u : cnt;
        const bool ok = inwin && (lo & 0x8080u) != 0x8080u && cnt != 0u && hend <= wl && (isbp || (val & ~hc.wmask) == 0u);
        const uint64_t okmask = __builtin_amdgcn_ballot_w64(ok);
        const uint32_t lead = ~okmask != 0ull ? (uint32_t)__builtin_ctzll(~okmask) : (uint32_t)kWave;
        // the run itself is not one for this loop (a long header, or — a pending run — a header the window
        // no longer holds): the loops below take it
        if (lead == 0u) break;
        const uint32_t p0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)p);
        if (first) {
            // where in its run the walk stands
            const uint32_t cnt0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt);
            const uint32_t hdr0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)hdr);
            const bool bp0 = __builtin_amdgcn_readfirstlane((int)(isbp ? 1u : 0u)) != 0;
            bool fits;
            if (st.pend == 0u) {
                fits = p0 == st.o;
            } else if (st.pend == 1u) {
                fits = !bp0 && st.pcount <= cnt0 && st.o == p0 + hdr0 + hc.vb;
                a = st.v - (cnt0 - st.pcount);
            } else {
                fits = bp0 && st.pcount <= cnt0 && st.o == p0 + hdr0 + (cnt0 - st.pcount) * hc.w && st.pskip < 8u;
                a = st.v - (8u * (cnt0 - st.pcount) + st.pskip);
            }
            if (!fits || a > st.v) {
                T.tn = 0;
                return;  // st untouched
            }
            first = false;
        } else if (p0 != o_cur) {
            T.tn = 0;  // the table and the chain of runs part ways (never, for a table the scan of this stream wrote)
            break;
        }
        o_cur = p0;
        if (v + TILE > n) break;
        const uint32_t tile_end = v + TILE;
        const uint64_t leadmask = lead >= 64u ? ~0ull : (1ull << lead) - 1ull;
        const uint32_t end_abs = a + wave_scan_u32(dn_select(leadmask, nvals));  // index behind the run's last value
        const uint32_t start = end_abs - nvals;
        const bool led = (uint32_t)lane < lead;
        const bool intile = led && start < tile_end;
        if (__builtin_amdgcn_ballot_w64(led && end_abs >= tile_end) == 0ull) break;  // 64 runs do not fill the tile
        // the bytes of the run the tile reads
        const uint32_t upto = min(end_abs, tile_end) - start;
        const uint32_t need = isbp ? p + hdr + ((__umul24(upto, hc.w) + 7u) >> 3) : hend;
        if (__builtin_amdgcn_ballot_w64(intile && need > wl) != 0ull) break;  // the window ends inside the tile
        if (intile) {
            const int32_t rs = (int32_t)(start - v);  // < 0: the run began in front of the tile
            const uint32_t at = (p + hdr) * 8u;
            L.pieces[lane] = make_uint2(isbp ? at - (uint32_t)__mul24(rs, (int32_t)hc.w) : at, isbp ? hc.w : 0u);
            reinterpret_cast<uint8_t *>(L.marks)[rs > 0 ? (uint32_t)rs : 0u] = 1;
        }
        wave_sync();
        dn_expand<MODE, OutT, Q, false, false>(L, sink, hc.wmask, posv, 0, 0, TILE, 0, v, lane, local_count, scount, bad_index);
        wave_sync();  // the marks were read; the next tile sets its own
#pragma unroll
        for (int q = 0; q < Q; ++q) L.marks[kWave * q + lane] = 0;
        wave_sync();
        // the runs that are out whole
        const uint32_t u = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(led && end_abs <= tile_end));
        if (u != 0u) {
            a = (uint32_t)__builtin_amdgcn_readlane((int)end_abs, (int)(u - 1u));
            o_cur = (uint32_t)__builtin_amdgcn_readlane((int)rend, (int)(u - 1u));
        }
        ti += u;
        v = tile_end;
    }
    if (first) return;  // nothing done, st untouched
    // ---- back to the terms of the other loops ----
    T.ti = ti;
    st.o = o_cur;
    st.pend = 0;
    st.pcount = 0;
    st.pskip = 0;
    if (a != v) {
        // inside run ti: its header once more, on the scalar unit, then what is out of it
        (void)hw_header_fast(st, L.window, wl, hc.w);  // a header this loop took: one or two bytes, all in the window
        const uint32_t out = v - a;
        if (st.pend == 1u) {
            st.pcount -= out;
        } else {
            st.pskip = out & 7u;
            st.pcount -= out >> 3;
            st.o += (out >> 3) * hc.w;
        }
    }
    st.v = v;
}

// Tiles of 256 Q values for as long as nothing needs the reference's checks.  On return the state
// describes exactly what is out: st.v values, the next byte (or the pending run) at st.o.
// allow_bulk = false: a fresh long bit-packed run is cut here like any other (the bulk path has just
// declined it).
// TABLE (DnTableMode): the caller may have a table of run headers (T; c0 = stream offset of window byte 0
// where the table counts from the stream's first byte, 0 where it counts from the window's); kTabNone:
// none of that code exists.
template <int MODE, typename OutT, int Q, int TABLE>
__device__ __forceinline__ uint32_t hw_dense_tiles(HybridWaveLds &L, HwState &st, const HwSink<OutT> &sink,
                                                   const HwConst &hc, uint32_t n, uint32_t mis, uint32_t wl,
                                                   uint32_t o_end, bool tail, uint32_t rw, bool allow_bulk, int lane,
                                                   uint32_t &local_count, uint32_t &scount, bool &bad_index,
                                                   DnTable &T, int32_t c0)
{
    constexpr uint32_t TILE = 256u * Q;
    const uint32_t w = hc.w;
    uint32_t cur_phase = 4;  // none yet
    uint32_t posv[Q][4];
    uint32_t cslot = 0;
    bool dense = false;  // what the last span said of the stream
    for (;;) {
        if (st.v >= n) return DN_DONE;
        const uint32_t phase = (st.v + mis) & 3u;
        if (phase != cur_phase) {
            cur_phase = phase;
#pragma unroll
            for (int q = 0; q < Q; ++q)
#pragma unroll
                for (int k = 0; k < 4; ++k) posv[q][k] = 4u * (uint32_t)lane + (uint32_t)(256 * q + k) - phase;
        }
        if (TABLE != kTabNone && T.tn != 0u && phase == 0u && st.v + TILE <= n && st.o < o_end && st.o < wl) {
            hw_table_tiles<MODE, OutT, Q>(L, st, sink, hc, n, wl, lane, local_count, scount, bad_index, T, c0, posv);
            if (st.v >= n) return DN_DONE;
            if (st.o >= o_end) return DN_WINDOW;
        }
        const uint32_t cap = TILE - phase;
        const uint32_t want = min(cap, n - st.v);
        uint32_t fill = 0, np = 0, reason = DN_NONE;
        while (fill < want && np < (uint32_t)kWave) {
            if (st.pend == 1u) {
                const uint32_t take = min(st.pcount, want - fill);
                cslot ^= 1u;
                if (lane == 0) {
                    L.window[kHwChunk / 4 + kHwSlack + cslot] = st.pvalue;
                    L.pieces[np] = make_uint2(kDnConstBit + 32u * cslot, 0u);
                    reinterpret_cast<uint8_t *>(L.marks)[phase + fill] = 1;
                }
                ++np;
                fill += take;
                st.pcount -= take;
                if (st.pcount == 0u) st.pend = 0;
            } else if (st.pend == 2u) {
                const bool fresh_long = allow_bulk && st.pskip == 0u && st.pcount >= 32u;
                if (fresh_long && (fill == 0u || 2u * fill >= cap)) {
                    reason = DN_BULK;
                    break;
                }
                // groups that lie wholly in the window
                const uint32_t bytes = st.o < wl ? wl - st.o : 0u;
                uint32_t gfit = (bytes * rw) >> 16;  // floor(bytes / w), or one more
                if (gfit * w > bytes) --gfit;
                const uint32_t gav = min(st.pcount, gfit);
                if (8u * gav <= st.pskip) {
                    reason = tail ? DN_GENERIC : DN_WINDOW;
                    break;
                }
                const uint32_t avail = 8u * gav - st.pskip;
                uint32_t take = min(avail, want - fill);
                uint32_t t = st.pskip + take;
                // what is left of a long run goes back to the bulk path: cut it between two groups
                // (the tile closes a few values short)
                bool shortened = false;
                if (allow_bulk && (t & 7u) != 0u && take > (t & 7u) && st.pcount - (t >> 3) >= 32u) {
                    take -= t & 7u;
                    t = st.pskip + take;
                    shortened = true;
                }
                if (lane == 0) {
                    L.pieces[np] = make_uint2(st.o * 8u + st.pskip * w - fill * w, w);
                    reinterpret_cast<uint8_t *>(L.marks)[phase + fill] = 1;
                }
                ++np;
                fill += take;
                const uint32_t adv = t >> 3;
                st.o += adv * w;
                st.pcount -= adv;
                st.pskip = t & 7u;
                if (st.pcount == 0u) st.pend = 0;
                if (shortened) break;
            } else {
                if (st.o >= o_end) { reason = DN_WINDOW; break; }
                if (st.o >= wl) { reason = DN_GENERIC; break; }
                uint32_t values;
                DnNext nx;
                uint32_t taken = kDnNoTable;
                if (TABLE != kTabNone && T.tn != 0u) taken = dn_span_table(L, st.o, wl, hc, want - fill, np, fill, phase, lane, values, nx, T, c0);
                if (taken == kDnNoTable)
                    taken = dn_span(L, st.o, wl, hc, want - fill, np, fill, phase, lane, values, nx, dense);
                if (taken == kDnOdd) { reason = DN_GENERIC; break; }
                np += taken;
                fill += values;
                st.o = nx.o;
                if (nx.kind != 0u) {
                    st.pend = nx.kind;
                    st.pcount = nx.count;
                    st.pvalue = nx.value;
                    st.pskip = 0;
                }
            }
        }
        const uint32_t clip = fill;
        if (clip != 0u) {
            wave_sync();
            const bool full = phase == 0u && clip == TILE;
            if (np == 1u) {
                const uint2 r0 = L.pieces[0];
                const uint32_t y0 = uni(r0.x), s0 = uni(r0.y);
                if (s0 == 0u) {
                    const uint32_t key = uni(dn_read(L.window, y0)) & hc.wmask;
                    dn_expand_const<MODE, OutT, Q>(sink, key, posv, clip, phase, st.v, lane, scount, bad_index);
                } else if (full) {
                    dn_expand<MODE, OutT, Q, false, true>(L, sink, hc.wmask, posv, y0, s0, clip, phase, st.v, lane,
                                                          local_count, scount, bad_index);
                } else {
                    dn_expand<MODE, OutT, Q, true, true>(L, sink, hc.wmask, posv, y0, s0, clip, phase, st.v, lane,
                                                         local_count, scount, bad_index);
                }
            } else if (full) {
                dn_expand<MODE, OutT, Q, false, false>(L, sink, hc.wmask, posv, 0, 0, clip, phase, st.v, lane, local_count,
                                                       scount, bad_index);
            } else {
                dn_expand<MODE, OutT, Q, true, false>(L, sink, hc.wmask, posv, 0, 0, clip, phase, st.v, lane, local_count,
                                                      scount, bad_index);
            }
            wave_sync();  // the marks were read; the next tile sets its own
#pragma unroll
            for (int q = 0; q < Q; ++q) L.marks[kWave * q + lane] = 0;
            wave_sync();
            st.v += clip;
        }
        if (reason != DN_NONE) return reason;
    }
}

}  // namespace pq
