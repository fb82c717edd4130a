// hybrid_wave.hpp — one WAVE per stream decoder for the RLE / bit-packed hybrid.
//
// Same value sequence as hybrid_stream.hpp (reference: HybridDecoder.Next and its
// helpers, encoding/hybrid-decoder.go:70-170), organised so that a wave never waits on
// another one: it owns a stream, a private LDS window that advances by a fixed stride
// (the next window's loads are issued a whole window ahead) and a 64-entry piece table.
// Per output tile of up to 512 values the wave
//   1. walks run headers on the SCALAR unit (1..4-byte ULEB128 headers and the RLE value
//      are taken from three broadcast LDS words with 32-bit arithmetic; anything unusual
//      goes through a checked byte-wise parse) and records each run, cut at the tile and
//      window boundaries, as a piece {first position, RLE flag, value | bit address},
//   2. maps positions to pieces: a piece marks its first position in a 512-byte LDS
//      array, a DPP prefix sum over the marks numbers the pieces,
//   3. expands: every lane produces 2 x 4 adjacent values (funnel read of two LDS words for
//      bit-packed pieces), optionally gathers them through a dictionary or counts the
//      ones equal to the maximum level, and writes 16 or 32 bytes with non-temporal stores.
#pragma once

#include "delta_wave.hpp"
#include "hybrid_stream.hpp"

namespace pq {

constexpr int kHwChunk = kDwChunk;              // bytes of the stream in the LDS window
constexpr int kHwGuard = 64;                    // >= header (10) + RLE value (4) and >= one group (32)
constexpr int kHwStride = kHwChunk - kHwGuard;  // distance between consecutive windows
constexpr int kHwQuads = 2;                     // 4-value quads per lane and tile
constexpr int kHwTile = 256 * kHwQuads;         // values per tile
constexpr int kHwSlack = 12;                    // zeroed words behind the window (a last group may hang over)
constexpr int kHwConstWords = 4;                // behind them: RLE values of runs carried from tile to tile
constexpr int kDnQuads = 4;                     // quads per lane and tile of the run-dense tiles: dictionary keys
constexpr int kDnQuadsPlain = 4;                // ... values that are stored as they are (levels, booleans)
constexpr uint32_t kDnTabCache = 1024;  // entries of a table of run headers a wave keeps in LDS: 16 per lane
constexpr uint32_t kHwMaxBit = (uint32_t)(kHwChunk + 4 * kHwSlack - 8) * 8u;  // highest bit address a funnel read may use

struct HybridWaveLds {
    uint32_t window[kHwChunk / 4 + kHwSlack + kHwConstWords];
    uint32_t marks[kWave * kDnQuadsPlain];  // one byte per tile position: 1 = a piece starts here
    uint2 pieces[kWave];    // RLE piece: x = ~0, y = the value; bit-packed: x = 0, y = bit address of
                            // (virtual) tile position 0, so that position j sits at y + j * w
    uint16_t tab[kDnTabCache];  // offsets of run headers: the window's, or a piece of a segment's table (hybrid_dense.hpp)
};

// wave_commit that also zeroes what lies past the end of the stream inside the last
// 16-byte piece: a short final bit-packed group reads its missing bytes as zero
// (Read, not ReadFull: encoding/hybrid-decoder.go:138-149).
__device__ __forceinline__ void wave_commit_clean(uint32_t *lds, const WavePrefetch &pf, int64_t c0, int64_t len,
                                                  int lane)
{
    uint4 *l4 = reinterpret_cast<uint4 *>(lds);
#pragma unroll
    for (int k = 0; k < kDwPieces; ++k) {
        const int64_t o = c0 + 16 * (int64_t)(lane + kWave * k);
        uint4 v = pf.v[k];
        if (o < len && o + 16 > len) v = mask_piece(v, 0, len - o);
        l4[lane + kWave * k] = v;
    }
}

// Piece r of the tile under construction, written by the walk (all arguments wave-uniform).
__device__ __forceinline__ void put_piece(HybridWaveLds &L, uint32_t r, uint32_t sk, uint32_t pl, uint32_t mark_at,
                                          int lane)
{
    if (lane == 0) {
        L.pieces[r] = make_uint2(sk, pl);
        reinterpret_cast<uint8_t *>(L.marks)[mark_at] = 1;
    }
}

// Wave-uniform state of the walk.
struct HwState {
    uint32_t o;       // window offset of the next unparsed byte
    uint32_t v;       // values produced so far
    uint32_t pend;    // unfinished run: 0 none, 1 RLE, 2 bit-packed
    uint32_t pcount;  // RLE: repeats left; bit-packed: groups left
    uint32_t pskip;   // bit-packed: values of the first of those groups that are out already (0..7)
    uint32_t pvalue;  // RLE value
    int32_t status;
};

// Checked parse of the run header at window offset st.o: readRunHeader and
// readRLERunValue (encoding/hybrid-decoder.go:111-136,151-170) with every error of the
// reference.  B = window bytes, lim = window offset of the end of the stream.
__device__ __forceinline__ void hw_header_slow(HwState &st, const uint8_t *B, uint32_t lim, uint32_t w)
{
    int64_t pos = (int64_t)st.o;
    uint32_t pend = 0, pc = 0, pv = 0;
    int status = PQ_OK;
    if (st.o >= lim) {
        status = PQ_EOF;
    } else {
        uint64_t h64 = 0;
        status = read_uvarint(B, 0, (int64_t)lim, &pos, &h64);
        if (status == PQ_OK && h64 > 0x7fffffffull) status = PQ_INT32_OUT_OF_RANGE;
        if (status == PQ_OK) {
            const uint32_t h = (uint32_t)h64;
            pc = h >> 1;
            if (h & 1u) {
                if (pc == 0) status = PQ_EMPTY_BP_RUN; else pend = 2;
            } else if (pc == 0) {
                status = PQ_EMPTY_RLE_RUN;
            } else {
                const uint32_t vb = (w + 7u) >> 3;
                const int64_t rem = (int64_t)lim - pos;
                if (rem <= 0) {
                    status = PQ_EOF;
                } else if (rem < (int64_t)vb) {
                    status = PQ_UNEXPECTED_EOF;
                    pos = (int64_t)lim;
                } else {
                    for (uint32_t k = 0; k < vb; ++k) pv |= (uint32_t)B[pos + k] << (8u * k);
                    pos += vb;
                    if (w < 32u && (pv >> w) != 0u) status = PQ_RLE_VALUE_TOO_LARGE; else pend = 1;
                }
            }
        }
    }
    // everything read back from LDS is the same in all lanes: say so
    st.o = uni((uint32_t)pos);
    st.pend = uni(pend);
    st.pcount = uni(pend ? pc : 0u);
    st.pvalue = uni(pv);
    st.status = (int32_t)uni((uint32_t)status);
}

// The common headers on the scalar unit: a 1..4-byte varint, a non-empty run, and for RLE a
// value that fits w bits, all inside the stream.  Returns false, changing nothing, otherwise.
__device__ __forceinline__ bool hw_header_fast(HwState &st, const uint32_t *lds, uint32_t lim, uint32_t w)
{
    const uint32_t o = st.o;
    const uint32_t k = o >> 2, s = (o & 3u) * 8u;
    const uint32_t d0 = uni(lds[k]), d1 = uni(lds[k + 1]), d2 = uni(lds[k + 2]);
    const uint32_t lo = (uint32_t)(((((uint64_t)d1) << 32) | d0) >> s);   // bytes 0..3 at o
    const uint32_t mid = (uint32_t)(((((uint64_t)d2) << 32) | d1) >> s);  // bytes 4..7
    const uint32_t stop = ~lo & 0x80808080u;
    if (stop == 0u) return false;
    const uint32_t vl = ((uint32_t)__builtin_ctz(stop) + 1u) >> 3;  // varint length 1..4
    const uint32_t x = lo & (0xffffffffu >> (32u - 8u * vl));
    const uint32_t h = (x & 0x7fu) | ((x >> 1) & 0x3f80u) | ((x >> 2) & 0x1fc000u) | ((x >> 3) & 0xfe00000u);
    const uint32_t pc = h >> 1;
    if (pc == 0u) return false;
    if (h & 1u) {
        if (o + vl > lim) return false;
        st.pend = 2;
        st.pcount = pc;
        st.o = o + vl;
        return true;
    }
    const uint32_t vb = (w + 7u) >> 3;  // 1..4 (w == 0 never gets here)
    if (o + vl + vb > lim) return false;
    const uint32_t raw = (uint32_t)(((((uint64_t)mid) << 32) | lo) >> (8u * vl));
    const uint32_t val = raw & (0xffffffffu >> (32u - 8u * vb));
    if (w < 32u && (val >> w) != 0u) return false;
    st.pend = 1;
    st.pcount = pc;
    st.pvalue = val;
    st.o = o + vl + vb;
    return true;
}

// Per-stream constants of the span parse.
struct HwConst {
    uint32_t w;
    uint32_t vb;      // bytes of an RLE value
    uint32_t vbmask;  // their mask
    uint32_t wmask;   // 2^w - 1
};

// Span parse: all 64 lanes parse a candidate run header at window offsets o .. o + 63 at once;
// the chain of real headers is then followed from lane 0 with one v_readlane per run.  Only
// complete, well-formed runs with a one- or two-byte header that lie wholly inside [0, wl) qualify;
// the chain stops at the first one that does not.  Of the chain, the leading runs that fit
// `room` values (and the piece table) are appended to the tile under construction: each of
// their lanes writes its own piece record and start mark.  Returns the number of runs taken.
// nx describes what follows them: the offset of the next header and, when that header belongs
// to the chain (a well-formed run that merely did not fit), its parsed fields, so that the
// caller can cut it without parsing it again.
struct HwNext {
    uint32_t o;       // window offset of the first header not taken
    uint32_t values;  // values of the runs taken
    uint32_t kind;    // 0: unknown (parse it); 1: RLE; 2: bit-packed
    uint32_t count;   // repeats | groups
    uint32_t value;   // RLE value
    uint32_t hdr;     // bytes of its header (1 or 2)
};

__device__ __forceinline__ uint32_t hw_span(HybridWaveLds &L, uint32_t o, uint32_t wl, const HwConst &hc, uint32_t room,
                                            uint32_t np, uint32_t fill, uint32_t phase, int lane, HwNext &nx)
{
    const uint32_t p = o + (uint32_t)lane;
    const uint32_t i = p >> 2, sh = (p & 3u) * 8u;
    const uint32_t d0 = L.window[i], d1 = L.window[i + 1], d2 = L.window[i + 2];
    const uint32_t lo = __builtin_amdgcn_alignbit(d1, d0, sh);  // bytes p .. p+3
    const uint32_t hi = __builtin_amdgcn_alignbit(d2, d1, sh);  // bytes p+4 .. p+7
    const bool two = (lo & 0x80u) != 0u;  // a second header byte follows
    const uint32_t hdr = two ? 2u : 1u;
    const uint32_t val = __builtin_amdgcn_alignbit(hi, lo, 8u * hdr) & hc.vbmask;
    const uint32_t h = two ? (lo & 0x7fu) | ((lo >> 1) & 0x3f80u) : lo & 0xffu;  // the ULEB128 header
    const uint32_t cnt = h >> 1;
    const bool isbp = (h & 1u) != 0u;
    const uint32_t nbytes = hdr + (isbp ? __umul24(cnt, hc.w) : hc.vb);
    const uint32_t nvals = isbp ? cnt * 8u : cnt;
    const bool ok = (lo & 0x8080u) != 0x8080u && cnt != 0u && p + nbytes <= wl && (isbp || (val & ~hc.wmask) == 0u);
    const uint32_t nextlane = (uint32_t)lane + nbytes;
    const uint64_t okmask = __builtin_amdgcn_ballot_w64(ok);
    nx.o = o;
    nx.values = 0;
    nx.kind = 0;
    nx.count = 0;
    nx.value = 0;
    nx.hdr = 1;

    // follow the chain from lane 0: chain = lanes that hold real headers, cur = where it leaves
    // the span (>= 64) or stops at a header that is not simple.  Hand-scheduled: five scalar
    // instructions between the v_readlane that writes `cur` and the one that uses it as its lane
    // select cover the four wait states that dependency needs.
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
    if (chain == 0ull) return 0;

    const bool on = ((chain >> lane) & 1ull) != 0ull;
    const uint32_t incl = wave_scan_u32(on ? nvals : 0u);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(chain >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)chain, 0u));
    const bool take = on && incl <= room && np + rank < (uint32_t)kWave;
    const uint64_t tmask = __builtin_amdgcn_ballot_w64(take);
    if (tmask != 0ull) {
        const uint32_t last = 63u - (uint32_t)__builtin_clzll(tmask);
        nx.values = (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)last);
    }
    const uint64_t untaken = chain & ~tmask;
    if (untaken != 0ull) {
        const uint32_t u = (uint32_t)__builtin_ctzll(untaken);
        const uint32_t hu = (uint32_t)__builtin_amdgcn_readlane((int)h, (int)u);
        nx.o = o + u;
        nx.kind = (hu & 1u) != 0u ? 2u : 1u;
        nx.count = hu >> 1;
        nx.value = (uint32_t)__builtin_amdgcn_readlane((int)val, (int)u);
        nx.hdr = (uint32_t)__builtin_amdgcn_readlane((int)hdr, (int)u);
    } else {
        nx.o = o + cur;
    }
    if (take) {
        const uint32_t start = fill + incl - nvals;
        L.pieces[np + rank] = make_uint2(isbp ? 0u : 0xffffffffu, isbp ? (p + hdr) * 8u - __umul24(start, hc.w) : val);
        reinterpret_cast<uint8_t *>(L.marks)[phase + start] = 1;
    }
    return (uint32_t)__builtin_popcountll(tmask);
}

template <typename OutT>
__device__ __forceinline__ void hw_store4(OutT *p, const OutT *out, uint32_t valid, bool vec_ok)
{
    if (valid == 0xfu && vec_ok) {
        if constexpr (sizeof(OutT) == 4) {
            store_nt16(p, out[0], out[1], out[2], out[3]);
        } else {
            store_nt16(p, (uint32_t)out[0], (uint32_t)(out[0] >> 32), (uint32_t)out[1], (uint32_t)(out[1] >> 32));
            store_nt16(p + 2, (uint32_t)out[2], (uint32_t)(out[2] >> 32), (uint32_t)out[3], (uint32_t)(out[3] >> 32));
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (valid & (1u << k)) p[k] = out[k];
}

// What a tile's expansion needs besides the tile itself.
template <typename OutT>
struct HwSink {
    OutT *dst;
    const OutT *dict;
    uint32_t dict_len;   // > 0 (callers hand an empty dictionary over as "every key is bad")
    uint32_t max_level;
    uint32_t w;
    bool vec_ok;
    bool no_dict;        // DICT with an empty dictionary: every key is out of range
};

// The tile of the bulk path in its usual shape: bit-packed values only, clip a multiple of 4
// values, dst + v aligned to 4 elements, so every quad is whole or empty and nothing is tracked
// per value.  Positions below `split` (a multiple of 8) belong to a piece whose position 0 sits
// at bit address pl_a, the others to a second piece with (virtual) position 0 at pl_b: the tail
// of one run and the head of the next share a tile.
template <int MODE, typename OutT>
__device__ __forceinline__ void hw_expand_bulk(HybridWaveLds &L, const HwSink<OutT> &sink, uint32_t pl_a,
                                               uint32_t split, uint32_t pl_b, uint32_t clip, uint32_t v, int lane,
                                               uint32_t &local_count, bool &bad_index)
{
    const uint32_t w = sink.w;
    const uint32_t jw = __umul24(4u * (uint32_t)lane, w);
    OutT *const qp0 = sink.dst + ((size_t)v + 4u * (uint32_t)lane);
    const uint32_t dmax = sink.dict_len - 1u;
#pragma unroll
    for (int q = 0; q < kHwQuads; ++q) {
        uint32_t key[4];
        // lanes past the end of the tile read a clamped address and store nothing
        uint32_t bit0 = jw + (4u * (uint32_t)lane + (uint32_t)(256 * q) < split ? pl_a : pl_b);
        asm volatile("" : "+v"(bit0));  // keep the per-position offsets below on the scalar unit
#pragma unroll
        for (int k = 0; k < 4; ++k)
            key[k] = extract32(L.window, min(bit0 + (uint32_t)(256 * q + k) * w, kHwMaxBit), w);
        const bool whole = 4u * (uint32_t)lane + (uint32_t)(256 * q) < clip;
        OutT out[4];
        if (MODE == HY_DICT) {
            const uint32_t top = max(max(key[0], key[1]), max(key[2], key[3]));
            bad_index |= whole && (top > dmax || sink.no_dict);
#pragma unroll
            for (int k = 0; k < 4; ++k) out[k] = sink.dict[sink.no_dict ? 0u : min(key[k], dmax)];
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) out[k] = (OutT)key[k];
        }
        if (whole) {
            if (MODE == HY_LEVELS) {
#pragma unroll
                for (int k = 0; k < 4; ++k) local_count += key[k] == sink.max_level ? 1u : 0u;
            }
            hw_store4<OutT>(qp0 + 256 * q, out, 0xfu, true);
        }
    }
}

// Expand one tile: positions [0, clip) go to dst[v ... v + clip).  np pieces; solo: the single
// piece {sk, pl} never went through LDS.  Lane l owns the quads that start at tile positions
// 256 q + 4 l - phase, which are 4-element aligned in memory.
template <int MODE, typename OutT>
__device__ __forceinline__ void hw_expand(HybridWaveLds &L, const HwSink<OutT> &sink, uint32_t np, bool solo,
                                          uint32_t sk0, uint32_t pl0, uint32_t clip, uint32_t phase, uint32_t v,
                                          int lane, uint32_t &local_count, bool &bad_index)
{
    const uint32_t w = sink.w;
    if (!solo) wave_sync();
    const uint32_t jbase = 4u * (uint32_t)lane - phase;  // tile position of this lane's first value
    uint32_t key[kHwQuads][4];
    if (np == 1) {
        uint32_t skv = sk0, plv = pl0;
        if (!solo) {
            skv = L.pieces[0].x;
            plv = L.pieces[0].y;
        }
        const uint32_t sk = uni(skv), pl = uni(plv);
        if (sk != 0u) {
#pragma unroll
            for (int q = 0; q < kHwQuads; ++q)
#pragma unroll
                for (int k = 0; k < 4; ++k) key[q][k] = pl;
        } else {
            // positions outside [0, clip) may point anywhere: their bit address is clamped into
            // the window and their value is never stored
            uint32_t bit0 = pl + (uint32_t)__mul24((int)jbase, (int)w);
            asm volatile("" : "+v"(bit0));  // keep the per-position offsets below on the scalar unit
#pragma unroll
            for (int q = 0; q < kHwQuads; ++q)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    key[q][k] = extract32(L.window, min(bit0 + (uint32_t)(256 * q + k) * w, kHwMaxBit), w);
        }
    } else {
        uint32_t x[kHwQuads], packed = 0;
#pragma unroll
        for (int q = 0; q < kHwQuads; ++q) {
            uint32_t m = L.marks[kWave * q + lane];
            m += m << 8;
            asm volatile("" : "+v"(m));  // two shift-adds, not a multiplication by 0x01010101 (quarter rate)
            m += m << 16;  // byte k = pieces started at or before position k of this quad
            x[q] = m;
            packed |= (m >> 24) << (16 * q);
        }
        const uint32_t incl = wave_scan_u32(packed);
        uint32_t jw = (uint32_t)__mul24((int)jbase, (int)w);
        asm volatile("" : "+v"(jw));  // keep the per-position offsets below on the scalar unit
        uint32_t before = 0;  // pieces started in the quads of lower q
#pragma unroll
        for (int q = 0; q < kHwQuads; ++q) {
            const uint32_t excl = before + ((incl >> (16 * q)) & 0xffffu) - (x[q] >> 24);
            before += (wave_last_u32(incl) >> (16 * q)) & 0xffffu;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t id = (excl + ((x[q] >> (8 * k)) & 0xffu) - 1u) & (uint32_t)(kWave - 1);
                const uint2 rec = L.pieces[id];
                const uint32_t bit = min(rec.y + jw + (uint32_t)(256 * q + k) * w, kHwMaxBit) & ~rec.x;
                key[q][k] = (extract32(L.window, bit, w) & ~rec.x) | (rec.y & rec.x);
            }
        }
    }
    // a quad is whole, empty or (at the edges of the tile) partly inside [0, clip)
    bool whole[kHwQuads];
    bool ragged = !sink.vec_ok;
#pragma unroll
    for (int q = 0; q < kHwQuads; ++q) {
        const int32_t j0 = (int32_t)jbase + 256 * q;  // >= -3
        whole[q] = j0 >= 0 && j0 + 3 < (int32_t)clip;
        ragged |= j0 < (int32_t)clip && !whole[q];
    }
    OutT *const qp0 = sink.dst + ((int64_t)v - (int64_t)phase + 4 * (int64_t)lane);  // quad q: + 256 q
    const uint32_t dmax = sink.dict_len - 1u;  // no_dict: wraps, every comparison below still holds
    if (__builtin_amdgcn_ballot_w64(ragged) == 0ull) {
        // the usual tile: whole quads only, no per-value bookkeeping
#pragma unroll
        for (int q = 0; q < kHwQuads; ++q) {
            OutT out[4];
            if (MODE == HY_DICT) {
                const uint32_t top = max(max(key[q][0], key[q][1]), max(key[q][2], key[q][3]));
                bad_index |= whole[q] && (top > dmax || sink.no_dict);
#pragma unroll
                for (int k = 0; k < 4; ++k) out[k] = sink.dict[sink.no_dict ? 0u : min(key[q][k], dmax)];
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) out[k] = (OutT)key[q][k];
            }
            if (whole[q]) {
                if (MODE == HY_LEVELS) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) local_count += key[q][k] == sink.max_level ? 1u : 0u;
                }
                hw_store4<OutT>(qp0 + 256 * q, out, 0xfu, true);
            }
        }
        return;
    }
    // a tile with ragged edges (first or last of a stream, or after a short tile)
#pragma unroll
    for (int q = 0; q < kHwQuads; ++q) {
        OutT out[4];
        uint32_t valid = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool ok = jbase + (uint32_t)(256 * q + k) < clip;  // also false "below zero"
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
        hw_store4<OutT>(qp0 + 256 * q, out, valid, sink.vec_ok);
    }
}

}  // namespace pq

#include "hybrid_dense.hpp"

namespace pq {

// Decode n values of the stream [S, S + len) at bit width w into dst with ONE wave.
// Modes as in hybrid_stream.  `dict` may point to LDS or global memory.  raw_groups != 0: the
// stream has no run headers at all, it is that many bit-packed groups (PLAIN fixed-width values
// are groups of width 32, BOOLEAN PLAIN bytes groups of width 1).
// TABLE (DnTableMode): kTabGiven: tab[0 .. tab_n) holds the offsets of the stream's run headers, from the
// stream's first byte, as a walk in front of this decode found them; kTabScan: the wave walks the headers of
// every window it decodes run-dense tiles from itself.
template <int MODE, typename OutT, int TABLE = 0>
__device__ __forceinline__ HybridResult hybrid_wave(HybridWaveLds &L, const uint8_t *S, int64_t len, uint32_t w,
                                                    uint32_t n, OutT *dst, const OutT *dict, uint32_t dict_len,
                                                    uint32_t max_level, int lane, uint32_t raw_groups = 0,
                                                    const uint16_t *tab = nullptr, uint32_t tab_n = 0)
{
    HybridResult res{PQ_OK, 0, 0, 0};
    const uint32_t mis = (uint32_t)((reinterpret_cast<uintptr_t>(dst) / sizeof(OutT)) & 3u);
    const bool vec_ok = (reinterpret_cast<uintptr_t>(dst) & (sizeof(OutT) - 1)) == 0;

    if (w == 0) {
        // bitWidth == 0: Next() yields 0 forever and never touches the reader (hybrid-decoder.go:74-76)
        OutT zv = 0;
        bool bad = false;
        if (MODE == HY_DICT) {
            if (dict_len == 0) bad = true; else zv = dict[0];
        }
        if (!bad)
            for (uint32_t i = (uint32_t)lane; i < n; i += kWave) dst[i] = zv;
        res.status = bad && n > 0 ? PQ_DICT_INVALID_INDEX : PQ_OK;
        res.n_decoded = bad ? 0 : n;
        res.count_max = (MODE == HY_LEVELS && max_level == 0) ? n : 0;
        return res;
    }

    const uint8_t *B = reinterpret_cast<const uint8_t *>(L.window);
    WavePrefetch pf;
    int64_t c0 = chunk_base(S, 0);
    wave_fetch(pf, S, c0, len, lane);
    if (lane < kHwSlack + kHwConstWords) L.window[kHwChunk / 4 + lane] = 0;
#pragma unroll
    for (int q = 0; q < kDnQuadsPlain; ++q) L.marks[kWave * q + lane] = 0;
    wave_commit_clean(L.window, pf, c0, len, lane);
    wave_sync();

    HwState st;
    st.o = (uint32_t)(0 - c0);
    st.v = 0;
    st.pend = raw_groups != 0u ? 2u : 0u;
    st.pcount = raw_groups;
    st.pskip = 0;
    st.pvalue = 0;
    st.status = PQ_OK;

    HwConst hc;
    hc.w = w;
    hc.vb = (w + 7u) >> 3;
    hc.vbmask = 0xffffffffu >> (32u - 8u * hc.vb);
    hc.wmask = w >= 32u ? 0xffffffffu : (1u << w) - 1u;
    const uint32_t rw = 65536u / w + 1u;  // x * rw >> 16 == x / w (or one more) for x < 4096

    HwSink<OutT> sink;
    sink.dst = dst;
    sink.no_dict = MODE == HY_DICT && dict_len == 0u;
    sink.dict = sink.no_dict ? reinterpret_cast<const OutT *>(L.window) : dict;  // something loadable
    sink.dict_len = dict_len;
    sink.max_level = max_level;
    sink.w = w;
    sink.vec_ok = vec_ok;

    uint32_t local_count = 0;  // LEVELS: values equal to max_level seen by this lane
    uint32_t scount = 0;       // LEVELS: the same, counted for whole tiles of one RLE run
    uint32_t stuck_v = 0xffffffffu;  // st.v when the run-dense tiles last handed a run to the bulk path
    bool bad_index = false;    // DICT: a key outside the dictionary seen by this lane
    DnTable T;                 // where a walk in front of this decode found the run headers, if one was made
    T.tab = TABLE == kTabGiven ? tab : nullptr;
    T.tn = TABLE == kTabGiven && tab != nullptr ? tab_n : 0u;
    T.ti = 0;
    T.cache = L.tab;
    T.cbase = 0;
    if (TABLE == kTabGiven && T.tn != 0u) dn_table_fill(T, 0, lane);
    bool scanned = false;      // kTabScan: this window's headers are in T

    while (true) {
        const int64_t c1 = c0 + kHwStride;
        const bool tail = c0 + kHwChunk >= len;  // this window reaches the end of the stream
        if (!tail) wave_fetch(pf, S, c1, len, lane);
        const int64_t left = len - c0;
        const uint32_t lim = left > (int64_t)(1 << 30) ? (1u << 30) : (left < 0 ? 0u : (uint32_t)left);
        const uint32_t wl = min(lim, (uint32_t)kHwChunk);  // valid stream bytes in the window
        // a run header or a group may start at offsets below this one
        const uint32_t o_end = tail ? 0xffffffffu : (uint32_t)(kHwChunk - kHwGuard) + 1u;
        bool done = false;
        if (TABLE == kTabScan) {
            T.tn = 0;
            scanned = false;
        }

        while (true) {
        // ---- long bit-packed runs: whole tiles straight from the window, nothing through LDS ----
        while (st.v < n && st.o < o_end) {
            uint32_t groups = st.pcount, o = st.o;
            if (st.pend == 0) {
                if (o >= wl) break;
                const uint32_t b0 = (uni(L.window[o >> 2]) >> ((o & 3u) * 8u)) & 0xffu;
                if (b0 >= 0x80u || (b0 & 1u) == 0u || b0 < 65u) break;  // not a one-byte header of >= 32 groups
                groups = b0 >> 1;
                o += 1u;
            } else if (st.pend != 2 || st.pskip != 0u) {
                break;
            }
            const uint32_t phase = (st.v + mis) & 3u;
            const uint32_t cap = (uint32_t)kHwTile - phase;
            const uint32_t room = min(cap >> 3, (n - st.v + 7u) >> 3);  // groups the tile may hold
            uint32_t g = min(groups, room);
            bool cut = false;
            if (o + g * w > wl) {
                // the window ends inside the run: take the groups that are wholly in it (the
                // end of the stream is left to the checked path below)
                if (tail) break;
                uint32_t fit = (((uint32_t)kHwChunk - o) * rw) >> 16;  // floor((chunk - o) / w), or one more
                if (fit * w > (uint32_t)kHwChunk - o) --fit;
                g = min(g, fit);
                cut = true;
            }
            // the run ends inside the tile: the head of the next long run fills the rest
            uint32_t g2 = 0, groups2 = 0, o2 = o + g * w;
            if (!cut && g < room && phase == 0u && vec_ok && o2 < wl && o2 < o_end) {
                const uint32_t b0 = (uni(L.window[o2 >> 2]) >> ((o2 & 3u) * 8u)) & 0xffu;
                if (b0 < 0x80u && (b0 & 1u) != 0u && b0 >= 65u) {
                    groups2 = b0 >> 1;
                    o2 += 1u;
                    g2 = min(groups2, room - g);
                    if (o2 + g2 * w > wl) {
                        uint32_t fit = 0;
                        if (!tail && o2 < (uint32_t)kHwChunk) {
                            fit = (((uint32_t)kHwChunk - o2) * rw) >> 16;
                            if (fit * w > (uint32_t)kHwChunk - o2) --fit;
                        }
                        g2 = min(g2, fit);
                    }
                    // the two-piece tile needs whole quads: not where the stream's last values
                    // end in the middle of one
                    if (8u * (g + g2) > n - st.v && ((n - st.v) & 3u) != 0u) g2 = 0;
                }
            }
            // short leftovers of a run with nothing to join are packed with what follows by the
            // generic path below; the few groups in front of a window end go out right here
            if (g2 == 0u && (cut ? g == 0u : 16u * g < cap)) break;
            const uint32_t clip = min(8u * (g + g2), n - st.v);
            if (phase == 0u && (clip & 3u) == 0u && vec_ok)
                hw_expand_bulk<MODE, OutT>(L, sink, o * 8u, 8u * g, o2 * 8u - 8u * g * w, clip, st.v, lane, local_count,
                                           bad_index);
            else  // g2 == 0 here
                hw_expand<MODE, OutT>(L, sink, 1, true, 0, o * 8u, clip, phase, st.v, lane, local_count, bad_index);
            st.v += clip;
            if (g2 != 0u) {
                st.o = o2 + g2 * w;
                st.pcount = groups2 - g2;
            } else {
                st.o = o + g * w;
                st.pcount = groups - g;
            }
            st.pend = st.pcount != 0u ? 2u : 0u;
        }
        if (st.v >= n) { done = true; break; }

        // ---- run-dense tiles: everything that needs none of the checks below ----
        if (vec_ok) {
            const bool allow_bulk = st.v != stuck_v;
            // (not for streams of ten values and more a byte — long RLE runs —, whose tiles find a whole tile's
            // runs in one or two spans as it is: BOOLEAN RLE at a mean run of 24 comes out 7 % slower with the walk)
            if (TABLE == kTabScan && !scanned && (uint64_t)n <= 10ull * (uint64_t)len) {
                // the headers of this window from the next one on, once, when the first run-dense tile is due
                scanned = true;
                const uint64_t from = st.pend == 2u ? (uint64_t)st.o + (uint64_t)st.pcount * w : (uint64_t)st.o;
                const uint32_t stop = min(o_end, wl);
                T.ti = 0;
                T.tn = 0;
                if (from < (uint64_t)stop) {
                    // not where the next run is a bit-packed one of 32 groups and more: such streams are the
                    // bulk path's, and what comes here of them are the odd ends of their runs
                    const uint32_t f = (uint32_t)from;
                    const uint32_t b0 = (uni(L.window[f >> 2]) >> ((f & 3u) * 8u)) & 0xffu;
                    if (!(b0 < 0x80u && (b0 & 1u) != 0u && b0 >= 65u))
                        T.tn = hw_window_scan(L, f, stop, wl, hc, L.tab, kDnTabCache, lane);
                }
                wave_sync();
            }
            const uint32_t why = hw_dense_tiles<MODE, OutT, MODE == HY_DICT ? kDnQuads : kDnQuadsPlain, TABLE>(
                L, st, sink, hc, n, mis, wl, o_end, tail, rw, allow_bulk, lane, local_count, scount, bad_index, T,
                TABLE == kTabGiven ? (int32_t)c0 : 0);
            if (why == DN_DONE) { done = true; break; }
            if (why == DN_WINDOW) break;
            if (why == DN_BULK) { stuck_v = st.v; continue; }
        }

        // ---- one tile of anything else ----
        {
            const uint32_t phase = (st.v + mis) & 3u;
            const uint32_t cap = (uint32_t)kHwTile - phase;
            const uint32_t want = min(cap, n - st.v);  // values this tile may hold
            uint32_t fill = 0, np = 0;
            uint32_t sk0 = 0, pl0 = 0;
            bool solo = false;         // the tile is one piece that never went through LDS
            bool want_window = false;
            while (fill < want && np < (uint32_t)kWave) {
                if (st.o >= o_end) { want_window = true; break; }
                if (st.pend == 0) {
                    // peek at the header byte: short runs with one-byte headers go through the
                    // span parse, 64 candidate positions at a time
                    const uint32_t b0 = (uni(L.window[st.o >> 2]) >> ((st.o & 3u) * 8u)) & 0xffu;
                    const uint32_t room = want - fill;
                    if (st.o < wl) {
                        if (b0 < 0x80u && (b0 & 1u) != 0u && b0 >= 65u) {
                            // a bit-packed run of 32 groups and more: no span parse, and left to
                            // the bulk path when the tile is half full already
                            if (8u * (b0 >> 1) > room && 2u * fill >= cap) break;
                            st.pend = 2;
                            st.pcount = b0 >> 1;
                            st.o += 1u;
                        } else {
                            HwNext nx;
                            const uint32_t taken = hw_span(L, st.o, wl, hc, room, np, fill, phase, lane, nx);
                            np += taken;
                            fill += nx.values;
                            st.o = nx.o;
                            if (nx.kind != 0u && np < (uint32_t)kWave && fill < want) {
                                // a well-formed run that does not fit whole: its head fills the tile,
                                // the rest stays pending (the piece code below cuts it)
                                st.pend = nx.kind;
                                st.pcount = nx.count;
                                st.pvalue = nx.value;
                                st.o = nx.o + nx.hdr + (nx.kind == 1u ? hc.vb : 0u);
                            } else if (taken != 0u) {
                                continue;
                            }
                            // else: the first run is malformed, cut short or unusual: parsed below
                        }
                    }
                    if (st.pend == 0 && !hw_header_fast(st, L.window, lim, w)) {
                        hw_header_slow(st, B, lim, w);
                        if (st.status != PQ_OK) break;
                    }
                }
                if (st.pend == 1) {
                    const uint32_t take = min(st.pcount, want - fill);
                    if (np == 0 && take == want) {
                        solo = true;
                        sk0 = 0xffffffffu;
                        pl0 = st.pvalue;
                    } else {
                        put_piece(L, np, 0xffffffffu, st.pvalue, fill + phase, lane);
                    }
                    ++np;
                    fill += take;
                    st.pcount -= take;
                    if (st.pcount == 0) st.pend = 0;
                } else {
                    // readBitPackedRun (hybrid-decoder.go:138-149): a group is readable while at
                    // least one byte is left; pieces are whole groups that lie inside the window
                    // (the first group may have lost st.pskip values to a run-dense tile)
                    const uint32_t sk = st.pskip;
                    uint32_t g = min(st.pcount, min((cap - fill + sk) >> 3, (want - fill + sk + 7u) >> 3));
                    if (g == 0) break;  // fewer than 8 positions left in the tile
                    if (st.o + g * w > wl) {
                        if (st.o >= lim) { st.status = PQ_EOF; break; }
                        const uint32_t feed = (lim - st.o + w - 1u) / w;  // groups the stream can feed
                        const uint32_t fit = tail ? feed : ((uint32_t)kHwChunk - st.o) / w;
                        g = min(g, min(feed, fit));
                        if (g == 0) { want_window = true; break; }
                    }
                    // a tile left with fewer than 16 free positions is closed
                    const bool closes = fill + 8u * g - sk + 16u > want;
                    if (np == 0 && closes) {
                        solo = true;
                        sk0 = 0;
                        pl0 = st.o * 8u + sk * w;
                    } else {
                        put_piece(L, np, 0u, st.o * 8u + sk * w - fill * w, fill + phase, lane);
                    }
                    ++np;
                    st.pskip = 0;
                    fill += 8u * g - sk;
                    st.o += g * w;
                    st.pcount -= g;
                    if (st.pcount == 0) st.pend = 0;
                    if (closes) break;
                }
            }

            // ---- expand the tile ----
            const uint32_t clip = min(fill, want);  // a last group may carry values beyond n
            if (clip != 0)
                hw_expand<MODE, OutT>(L, sink, np, solo, sk0, pl0, clip, phase, st.v, lane, local_count, bad_index);
            if (np != 0 && !solo) {
                wave_sync();  // the marks were read; the next walk sets its own
#pragma unroll
                for (int q = 0; q < kHwQuads; ++q) L.marks[kWave * q + lane] = 0;
                wave_sync();
            }
            st.v += clip;
            if (st.status != PQ_OK || st.v >= n) { done = true; break; }
            if (want_window) break;
        }
        }
        if (done) break;
        wave_sync();
        wave_commit_clean(L.window, pf, c1, len, lane);
        wave_sync();
        c0 = c1;
        st.o -= (uint32_t)kHwStride;
    }

    if (MODE == HY_LEVELS) res.count_max = wave_sum_u32(local_count) + scount;
    res.status = st.status;
    res.n_decoded = st.v;
    // a group some of whose values are out has been read whole (readBitPackedRun)
    const int64_t pos = c0 + (int64_t)st.o + (st.pend == 2u && st.pskip != 0u ? (int64_t)w : 0);
    res.consumed = (uint64_t)min(pos, len);
    if (MODE == HY_DICT && __builtin_amdgcn_ballot_w64(bad_index) != 0ull) {
        // DictDecoder.DecodeValues returns (0, "invalid index") (types/dict.go:104-111)
        res.status = PQ_DICT_INVALID_INDEX;
        res.n_decoded = 0;
    }
    return res;
}

}  // namespace pq
