// hybrid_stream.hpp — workgroup-wide decoder for one RLE / bit-packed hybrid stream.
//
// Follows the value sequence of the reference's HybridDecoder
// (encoding/hybrid-decoder.go:70-170) pulled exactly n times from a fresh decoder,
// but push-based: the stream is staged through LDS in 16-byte-aligned chunks; one
// lane walks the run headers of the chunk (ULEB128 header, RLE payload) into a run
// table of {first value index, RLE value | bit offset of the packed groups}; all
// four waves then expand the covered value range, each lane producing 16 bytes of
// consecutive outputs (binary search of the run table once per lane, then a
// forward walk), optionally gathering through a dictionary or counting levels
// equal to the maximum.
#pragma once

#include "pq_device.hpp"

namespace pq {

constexpr int kHyChunk = 8192; // staged bytes per pass
constexpr int kHyRuns = 512;   // run-table entries per pass

enum HybridMode { HY_RAW = 0, HY_DICT = 1, HY_LEVELS = 2 };

struct HybridShared {
    uint32_t chunk[kHyChunk / 4 + 4];
    uint32_t run_start[kHyRuns + 1]; // value index | 0x80000000 for RLE runs
    uint32_t run_data[kHyRuns];      // RLE: the value; bit-packed: bit offset in `chunk`
    int64_t pos;                     // stream offset of the next unparsed byte
    uint32_t vals_done;              // values already written
    uint32_t v_end;                  // values covered once this pass is expanded
    uint32_t nruns;
    uint32_t pend_rle;               // 1 = the unfinished run is RLE
    uint32_t pend_count;             // RLE: repeats left; bit-packed: groups left
    uint32_t pend_value;
    int32_t status;                  // terminal status of the walker (PQ_OK while running)
    uint32_t bad_index;              // DICT: some key was outside the dictionary
    uint32_t count_max;              // LEVELS: #{value == max_level}
};

struct HybridResult {
    int32_t status;
    uint32_t n_decoded;
    uint32_t count_max;
    uint64_t consumed;
};

__device__ __forceinline__ uint4 pack16(const uint32_t *o) { return make_uint4(o[0], o[1], o[2], o[3]); }

__device__ __forceinline__ uint4 pack16(const uint64_t *o)
{
    return make_uint4((uint32_t)o[0], (uint32_t)(o[0] >> 32), (uint32_t)o[1], (uint32_t)(o[1] >> 32));
}

// One lane: extend the run table over the staged chunk [c0, c0 + kHyChunk).
__device__ __forceinline__ void hybrid_walk(HybridShared &sh, int64_t c0, int64_t len, uint32_t w, uint32_t n)
{
    const uint8_t *B = reinterpret_cast<const uint8_t *>(sh.chunk);
    int64_t pos = sh.pos;
    uint32_t v = sh.vals_done;
    uint32_t prle = sh.pend_rle, pc = sh.pend_count, pv = sh.pend_value;
    int status = PQ_OK;
    uint32_t nr = 0;
    const int64_t cend = c0 + kHyChunk;
    const bool tail = cend >= len; // the chunk reaches the end of the stream
    const uint32_t vb = (w + 7u) >> 3;

    while (nr < (uint32_t)kHyRuns && v < n) {
        if (pc == 0) {
            // readRunHeader (hybrid-decoder.go:111-136)
            if (pos >= len) { status = PQ_EOF; break; }
            if (!tail && pos + 16 > cend) break; // header may straddle the chunk: restage
            uint64_t h64;
            int st = read_uvarint(B, c0, len, &pos, &h64);
            if (st != PQ_OK) { status = st; break; }
            if (h64 > 0x7fffffffull) { status = PQ_INT32_OUT_OF_RANGE; break; }
            const uint32_t h = (uint32_t)h64;
            pc = h >> 1;
            if (h & 1u) {
                if (pc == 0) { status = PQ_EMPTY_BP_RUN; break; }
                prle = 0;
            } else {
                if (pc == 0) { status = PQ_EMPTY_RLE_RUN; break; }
                // readRLERunValue (hybrid-decoder.go:151-170)
                const int64_t rem = len - pos;
                if (rem <= 0) { status = PQ_EOF; pc = 0; break; }
                if (rem < (int64_t)vb) { status = PQ_UNEXPECTED_EOF; pos = len; pc = 0; break; }
                uint32_t val = 0;
                for (uint32_t k = 0; k < vb; ++k) val |= (uint32_t)B[pos + k - c0] << (8u * k);
                pos += vb;
                if (w < 32u && (val >> w) != 0u) { status = PQ_RLE_VALUE_TOO_LARGE; pc = 0; break; }
                prle = 1;
                pv = val;
            }
        }
        if (prle) {
            const uint32_t take = min(pc, n - v);
            sh.run_start[nr] = v | 0x80000000u;
            sh.run_data[nr] = pv;
            ++nr;
            v += take;
            pc -= take;
        } else {
            // readBitPackedRun (hybrid-decoder.go:138-149): Read, not ReadFull — a
            // group is readable while at least one byte is left, missing bytes are 0.
            const int64_t avail = len - pos;
            if (avail <= 0) { status = PQ_EOF; break; }
            const int64_t fit = (cend - pos) / (int64_t)w;        // groups inside the staged window
            const int64_t feed = (avail + (int64_t)w - 1) / (int64_t)w; // groups the stream can feed
            const int64_t need = (int64_t)((n - v + 7u) >> 3);
            int64_t take = min(min((int64_t)pc, fit), min(feed, need));
            if (take == 0) break; // window exhausted: restage from pos
            sh.run_start[nr] = v;
            sh.run_data[nr] = (uint32_t)(pos - c0) * 8u;
            ++nr;
            v += (uint32_t)take * 8u;
            pos += take * (int64_t)w;
            pc -= (uint32_t)take;
        }
    }
    sh.run_start[nr] = v; // sentinel
    sh.nruns = nr;
    sh.v_end = min(v, n);
    sh.pos = pos;
    sh.pend_rle = prle;
    sh.pend_count = pc;
    sh.pend_value = pv;
    sh.status = status;
}

// Decode n values of the stream [S, S + len) at bit width w into dst.
//   MODE HY_RAW    : dst[i] = value                       (OutT = uint32_t)
//   MODE HY_DICT   : dst[i] = dict[value], value < dict_len else bad_index
//   MODE HY_LEVELS : dst[i] = value, count values == max_level
// All 256 threads must call it; the result is valid in every thread.
template <int MODE, typename OutT>
__device__ __forceinline__ HybridResult hybrid_stream(HybridShared &sh, const uint8_t *S, int64_t len,
                                             uint32_t w, uint32_t n, OutT *dst,
                                             const OutT *dict, uint32_t dict_len,
                                             uint32_t max_level)
{
    constexpr int QUAD = 16 / sizeof(OutT);
    const int tid = threadIdx.x;
    HybridResult res;

    if (tid == 0) {
        sh.pos = 0;
        sh.vals_done = 0;
        sh.pend_count = 0;
        sh.pend_rle = 0;
        sh.pend_value = 0;
        sh.status = PQ_OK;
        sh.bad_index = 0;
        sh.count_max = 0;
    }
    __syncthreads();

    uint32_t local_count = 0;
    const uint32_t mis = (uint32_t)((reinterpret_cast<uintptr_t>(dst) / sizeof(OutT)) & (QUAD - 1));

    if (w == 0) {
        // bitWidth == 0: Next() yields 0 forever and never touches the reader
        // (hybrid-decoder.go:74-76).
        OutT zv = 0;
        bool bad = false;
        if (MODE == HY_DICT) {
            if (dict_len == 0) bad = true; else zv = dict[0];
        }
        if (!bad)
            for (uint32_t i = tid; i < n; i += kWG) dst[i] = zv;
        res.status = bad && n > 0 ? PQ_DICT_INVALID_INDEX : PQ_OK;
        res.n_decoded = bad ? 0 : n;
        res.count_max = (MODE == HY_LEVELS && max_level == 0) ? n : 0;
        res.consumed = 0;
        __syncthreads();
        return res;
    }

    while (true) {
        const uint32_t done = sh.vals_done;
        if (done >= n || sh.status != PQ_OK) break;
        const int64_t c0 = chunk_base(S, sh.pos);
        stage_chunk<kHyChunk>(sh.chunk, S, c0, len, tid);
        __syncthreads();
        if (tid == 0) hybrid_walk(sh, c0, len, w, n);
        __syncthreads();

        const uint32_t lo = done, hi = sh.v_end;
        const uint32_t nr = sh.nruns;
        if (hi > lo) {
            const uint32_t q_lo = (lo + mis) / QUAD;
            const uint32_t q_hi = (hi + mis + QUAD - 1) / QUAD;
            for (uint32_t q = q_lo + tid; q < q_hi; q += kWG) {
                const int64_t e0 = (int64_t)q * QUAD - (int64_t)mis;
                const uint32_t a = (uint32_t)max(e0, (int64_t)lo);
                const uint32_t b = (uint32_t)min(e0 + QUAD, (int64_t)hi);
                // last run whose start <= a
                uint32_t r = 0, rhi = nr;
                while (rhi - r > 1) {
                    const uint32_t mid = (r + rhi) >> 1;
                    if ((sh.run_start[mid] & 0x7fffffffu) <= a) r = mid; else rhi = mid;
                }
                OutT out[QUAD];
#pragma unroll
                for (int j = 0; j < QUAD; ++j) {
                    const int64_t e = e0 + j;
                    out[j] = 0;
                    if (e >= (int64_t)a && e < (int64_t)b) {
                        while ((uint32_t)e >= (sh.run_start[r + 1] & 0x7fffffffu)) ++r;
                        const uint32_t rs = sh.run_start[r];
                        uint32_t key;
                        if (rs & 0x80000000u)
                            key = sh.run_data[r];
                        else
                            key = extract32(sh.chunk, sh.run_data[r] + ((uint32_t)e - rs) * w, w);
                        if (MODE == HY_DICT) {
                            if (key < dict_len) out[j] = dict[key]; else sh.bad_index = 1;
                        } else {
                            out[j] = (OutT)key;
                            if (MODE == HY_LEVELS && key == max_level) ++local_count;
                        }
                    }
                }
                if (a == (uint32_t)e0 && b == (uint32_t)(e0 + QUAD) && e0 >= 0) {
                    *reinterpret_cast<uint4 *>(dst + e0) = pack16(out);
                } else {
#pragma unroll
                    for (int j = 0; j < QUAD; ++j) {
                        const int64_t e = e0 + j;
                        if (e >= (int64_t)a && e < (int64_t)b) dst[e] = out[j];
                    }
                }
            }
        }
        __syncthreads(); // run table and chunk are free again
        if (tid == 0) sh.vals_done = hi;
        __syncthreads();
    }

    if (MODE == HY_LEVELS) {
        const uint32_t wsum = wave_sum_u32(local_count);
        if ((tid & (kWave - 1)) == 0 && wsum) atomicAdd(&sh.count_max, wsum);
        __syncthreads();
    }
    res.status = sh.status;
    res.n_decoded = sh.vals_done;
    res.count_max = sh.count_max;
    res.consumed = (uint64_t)min(sh.pos, len);
    if (MODE == HY_DICT && sh.bad_index) {
        // DictDecoder.DecodeValues returns (0, "invalid index") (types/dict.go:104-111)
        res.status = PQ_DICT_INVALID_INDEX;
        res.n_decoded = 0;
    }
    __syncthreads(); // callers may reuse `sh` right away
    return res;
}

} // namespace pq
