// seghybrid.hip — large RLE / bit-packed hybrid streams (dictionary-index pages, raw hybrid
// streams) cut into segments, so that one large page is decoded by many waves (reference:
// encoding/hybrid-decoder.go:70-136; a stream is one chain there: run k + 1 starts where run k's
// header says, so a page is one wave's serial walk however long it is).
//
// What is cut is the common shape of a dictionary-index page with many distinct keys: bit-packed
// runs that all carry the same header (the reference's encoder closes a bit-packed run after 63
// groups, encoding/hybrid-encoder.go:14,158-161; other writers after 64).  Every run of such a
// stream starts a fixed stride behind the one before, which can be checked for all runs at once:
//   1. hseg_header_kernel  one lane per page: bit width, the first run's header, the stride.
//   2. hseg_check_kernel   one lane per run: is the header at first + r * stride the first run's
//                          header again, and is the run all there?  (64 runs per wave.)
//   3. hseg_link_kernel    one wave per page: U = the runs in front of the first "no" form a
//                          proven chain.  They are dealt out in segments of G runs; what is behind
//                          them (other runs, the stream's end, errors) is the tail segment.
//   4. hseg_decode_kernel  one wave per segment: the page decoder's own hybrid_wave over the
//                          segment's bytes as a stream of its own (runs share no state).  The
//                          tail runs to the end of the page like the whole-page decode.
//   5. hseg_final_kernel   one wave per page: the page's result from its segments' — the tail's
//                          status, count and consumed bytes moved by where the tail starts; a key
//                          outside the dictionary anywhere makes the page "invalid index" with
//                          count 0 (types/dict.go:104-111).
// A stream of any other shape is decoded whole by its first segment's wave.
#include "hybrid_wave.hpp"
#include "pq_launch.hpp"

namespace pq {

constexpr int kHsWG = 512;               // threads per workgroup of the decode kernel
constexpr int kHsWaves = kHsWG / kWave;  // segments per workgroup
constexpr uint32_t kHsMinStride = 64;    // shorter runs are not worth a lane each

extern __shared__ __attribute__((aligned(16))) uint8_t pq_hs_smem[];

constexpr uint32_t hs_align16(uint32_t x) { return (x + 15u) & ~15u; }

// ---- 1 ---------------------------------------------------------------------------------------

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
    uint32_t w = d.bit_width, pos0 = 0;
    bool ok = len > 8;
    if (d.kind == PQ_KIND_DICT) {  // the bit-width byte first (types/dict.go:66-85)
        w = ok ? S[0] : 0u;
        pos0 = 1;
    }
    ok = ok && w >= 1u && w <= 32u;
    // the first run header: a bit-packed run (odd), of at most 2^20 groups
    uint32_t h = 0, hl = 0;
    if (ok) {
        int64_t pos = pos0;
        uint64_t u = 0;
        ok = read_uvarint(S, 0, len, &pos, &u) == PQ_OK && u <= (1u << 21) && (u & 1u) != 0u && (u >> 1) != 0u;
        h = (uint32_t)u;
        hl = (uint32_t)(pos - pos0);
    }
    const uint32_t groups = h >> 1;
    const uint64_t stride = (uint64_t)hl + (uint64_t)groups * w;
    ok = ok && hl <= 4u && stride >= kHsMinStride && stride <= 0x7fffffffull;
    uint32_t head = 0;
    if (ok)
        for (uint32_t b = 0; b < hl; ++b) head |= (uint32_t)S[pos0 + b] << (8u * b);
    pg.whole = ok ? 0u : 1u;
    pg.w = w;
    pg.pos0 = pos0;
    pg.stride = (uint32_t)stride;
    pg.vpr = groups * 8u;
    pg.head = head;
    pg.hl = hl;
    const uint64_t per = pg.seg_bytes / (stride ? stride : 1u);
    pg.runs_per_seg = (uint32_t)(per ? per : 1u);
    pages[i] = pg;
}

// ---- 2 ---------------------------------------------------------------------------------------

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
    if (pg.whole) return;
    const pq_page_desc d = descs[pg.desc];
    const uint8_t *S = buf.src + d.src_off;
    const uint64_t len = d.src_len;
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

// ---- 3 ---------------------------------------------------------------------------------------

__global__ __launch_bounds__(kWG) void hseg_link_kernel(const pq_page_desc *__restrict__ descs,
                                                        HsegPage *__restrict__ pages, uint32_t n_pages,
                                                        const uint32_t *__restrict__ counts,
                                                        HsegRec *__restrict__ recs)
{
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t pi = blockIdx.x * (kWG / kWave) + (threadIdx.x >> 6);
    if (pi >= n_pages) return;
    const HsegPage pg = pages[pi];
    HsegRec *R = recs + pg.seg_first;
    const pq_page_desc d = descs[pg.desc];
    if (pg.whole) {
        for (uint32_t s = lane; s < pg.n_segs; s += kWave) {
            HsegRec r{0ull, 0u, d.n_values, s == 0u ? kSegWhole : kSegSkip, PQ_OK, 0u, 0u, 0ull};
            R[s] = r;
        }
        return;
    }
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
        HsegRec r{0ull, 0u, 0u, kSegSkip, PQ_OK, 0u, 0u, 0ull};
        if (s <= P) {
            r.off = pg.pos0 + s * seg_stride;
            r.v0 = (uint32_t)(s * seg_vals);
            r.count = s < P ? (uint32_t)seg_vals : (uint32_t)(n - s * seg_vals);
            r.flags = s < P ? kSegPart : kSegTail;
        }
        R[s] = r;
    }
    if (lane == 0) pages[pi].n_part = (uint32_t)P;
}

// ---- 4 ---------------------------------------------------------------------------------------

template <typename OutT>
__device__ __forceinline__ HybridResult hseg_values(HybridWaveLds &L, uint32_t kind, const uint8_t *S, int64_t len,
                                                    uint32_t w, uint32_t n, uint8_t *dst, const uint8_t *gdict,
                                                    const uint32_t *ldict, uint32_t dict_len, int lane)
{
    if (kind == PQ_KIND_DICT) {
        const OutT *dict = ldict ? reinterpret_cast<const OutT *>(ldict) : reinterpret_cast<const OutT *>(gdict);
        return hybrid_wave<HY_DICT, OutT>(L, S, len, w, n, reinterpret_cast<OutT *>(dst), dict, dict_len, 0, lane);
    }
    return hybrid_wave<HY_RAW, uint32_t>(L, S, len, w, n, reinterpret_cast<uint32_t *>(dst), nullptr, 0, 0, lane);
}

__global__ __launch_bounds__(kHsWG) void hseg_decode_kernel(const pq_page_desc *__restrict__ descs,
                                                            const HsegPage *__restrict__ pages,
                                                            const uint32_t *__restrict__ seg_owner, uint32_t n_segs,
                                                            HsegRec *__restrict__ recs, PageBuffers buf,
                                                            pq_page_result *__restrict__ results,
                                                            uint32_t dict_lds_bytes)
{
    HybridWaveLds *LW = reinterpret_cast<HybridWaveLds *>(pq_hs_smem);
    // the workgroup's LDS copy of a dictionary: the one its first segment's page reads, if it fits
    uint32_t *ld = reinterpret_cast<uint32_t *>(pq_hs_smem + hs_align16((uint32_t)sizeof(HybridWaveLds) * kHsWaves));
    uint64_t have_off = 0;
    uint32_t have_bytes = 0;
    {
        const pq_page_desc &d0 = descs[pages[seg_owner[blockIdx.x * kHsWaves]].desc];
        const uint32_t bytes = uni(d0.kind == PQ_KIND_DICT ? d0.dict_len * d0.elem_size : 0u);
        if (dict_lds_bytes != 0u && bytes != 0u && bytes <= dict_lds_bytes) {
            have_off = ((uint64_t)uni((uint32_t)(d0.dict_off >> 32)) << 32) | uni((uint32_t)d0.dict_off);
            have_bytes = bytes;
            const uint32_t *gd = reinterpret_cast<const uint32_t *>(buf.dict + have_off);
            for (uint32_t i = threadIdx.x; i < bytes / 4u; i += kHsWG) ld[i] = gd[i];
            __syncthreads();
        }
    }
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t g = blockIdx.x * kHsWaves + (uint32_t)wave;
    if (g >= n_segs) return;
    const HsegRec rec = recs[g];
    const uint32_t flags = uni(rec.flags);
    if (flags == kSegSkip) return;
    const HsegPage pg = pages[seg_owner[g]];
    const pq_page_desc d = descs[pg.desc];
    HybridWaveLds &L = LW[wave];
    const uint8_t *S = buf.src + d.src_off;
    const int64_t len = (int64_t)d.src_len;
    const uint32_t es = d.kind == PQ_KIND_DICT ? d.elem_size : 4u;
    uint8_t *dst = buf.values + d.dst_off;
    const uint8_t *gdict = buf.dict + d.dict_off;
    const uint32_t *ldict = d.kind == PQ_KIND_DICT && have_bytes != 0u && d.dict_off == have_off &&
                                    d.dict_len * d.elem_size == have_bytes
                                ? ld
                                : nullptr;
    HybridResult r{PQ_OK, 0, 0, 0};
    if (flags == kSegWhole) {
        // the whole stream as stream_wave_kernel decodes it
        int32_t stage = PQ_STAGE_NONE;
        if (d.kind == PQ_KIND_DICT) {
            if (len <= 0) {
                r.status = PQ_EOF;
                stage = PQ_STAGE_VALUES_INIT;  // DictDecoder.Init failed
            } else {
                const uint32_t w = uni((uint32_t)S[0]);
                if (w > 32u) {
                    r.status = PQ_DICT_INVALID_WIDTH;
                    r.consumed = 1;
                    stage = PQ_STAGE_VALUES_INIT;
                } else {
                    r = es == 8u ? hseg_values<uint64_t>(L, d.kind, S + 1, len - 1, w, d.n_values, dst, gdict, ldict, d.dict_len, lane)
                                 : hseg_values<uint32_t>(L, d.kind, S + 1, len - 1, w, d.n_values, dst, gdict, ldict, d.dict_len, lane);
                    r.consumed += 1;
                }
            }
        } else {
            r = hseg_values<uint32_t>(L, d.kind, S, len, d.bit_width, d.n_values, dst, gdict, ldict, 0, lane);
        }
        if (lane == 0) {
            pq_page_result out;
            out.status = r.status;
            out.stage = stage;
            out.detail = 0;
            out.n_decoded = r.n_decoded;
            out.not_null = 0;
            out.n_levels = 0;
            out.consumed = r.consumed;
            results[pg.desc] = out;
        }
        return;
    }
    const uint64_t off = ((uint64_t)uni((uint32_t)(rec.off >> 32)) << 32) | uni((uint32_t)rec.off);
    const uint32_t v0 = uni(rec.v0), count = uni(rec.count);
    // a proven segment is exactly its runs; the tail is everything from its first byte on
    const int64_t slen = flags == kSegPart ? (int64_t)pg.runs_per_seg * pg.stride : len - (int64_t)off;
    uint8_t *sdst = dst + (size_t)v0 * es;
    r = es == 8u ? hseg_values<uint64_t>(L, d.kind, S + off, slen, pg.w, count, sdst, gdict, ldict, d.dict_len, lane)
                 : hseg_values<uint32_t>(L, d.kind, S + off, slen, pg.w, count, sdst, gdict, ldict, d.dict_len, lane);
    if (lane == 0) {
        recs[g].status = r.status;
        recs[g].n_decoded = r.n_decoded;
        recs[g].consumed = r.consumed;
    }
}

// ---- 5 ---------------------------------------------------------------------------------------

__global__ __launch_bounds__(kWG) void hseg_final_kernel(const HsegPage *__restrict__ pages, uint32_t n_pages,
                                                         const HsegRec *__restrict__ recs,
                                                         pq_page_result *__restrict__ results)
{
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t pi = blockIdx.x * (kWG / kWave) + (threadIdx.x >> 6);
    if (pi >= n_pages) return;
    const HsegPage pg = pages[pi];
    if (pg.whole) return;  // its first segment wrote the page's result
    const HsegRec *R = recs + pg.seg_first;
    const uint32_t P = pg.n_part;
    // the first proven segment (in stream order) that did not end well, if any
    uint32_t first_bad = 0xffffffffu;
    for (uint32_t base = 0; base < P && first_bad == 0xffffffffu; base += kWave) {
        const uint32_t s = base + lane;
        const bool bad = s < P && (R[s].status != PQ_OK || R[s].n_decoded != R[s].count);
        const uint64_t m = __builtin_amdgcn_ballot_w64(bad);
        if (m != 0ull) first_bad = base + (uint32_t)__builtin_ctzll(m);
    }
    if (lane != 0) return;
    const HsegRec t = R[first_bad != 0xffffffffu ? first_bad : P];
    pq_page_result out;
    out.status = t.status;
    out.stage = PQ_STAGE_NONE;
    out.detail = 0;
    // DictDecoder.DecodeValues reports count 0 with "invalid index" (types/dict.go:104-111)
    out.n_decoded = t.status == PQ_DICT_INVALID_INDEX ? 0u : t.v0 + t.n_decoded;
    out.not_null = 0;
    out.n_levels = 0;
    out.consumed = t.off + t.consumed;
    results[pg.desc] = out;
}

hipError_t launch_seg_hybrid(hipStream_t s, const pq_page_desc *d_descs, HsegPage *d_pages, uint32_t n_pages,
                             const uint32_t *d_check_owner, uint32_t n_checks, uint32_t *d_counts,
                             const uint32_t *d_seg_owner, uint32_t n_segs, HsegRec *d_recs, PageBuffers buf,
                             pq_page_result *d_results, uint32_t dict_lds_bytes)
{
    if (n_pages == 0 || n_segs == 0) return hipSuccess;
    const uint32_t per = kWG / kWave;
    hipLaunchKernelGGL(hseg_header_kernel, dim3((n_pages + kWave - 1) / kWave), dim3(kWave), 0, s, d_descs, d_pages, n_pages,
                       buf);
    if (n_checks != 0)
        hipLaunchKernelGGL(hseg_check_kernel, dim3((n_checks + per - 1) / per), dim3(kWG), 0, s, d_descs, d_pages,
                           d_check_owner, n_checks, d_counts, buf);
    hipLaunchKernelGGL(hseg_link_kernel, dim3((n_pages + per - 1) / per), dim3(kWG), 0, s, d_descs, d_pages, n_pages, d_counts,
                       d_recs);
    const size_t lds = hs_align16((uint32_t)sizeof(HybridWaveLds) * kHsWaves) + dict_lds_bytes;
    if (lds > 65536) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(hseg_decode_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(hseg_decode_kernel, dim3((n_segs + kHsWaves - 1) / kHsWaves), dim3(kHsWG), lds, s, d_descs, d_pages,
                       d_seg_owner, n_segs, d_recs, buf, d_results, dict_lds_bytes);
    hipLaunchKernelGGL(hseg_final_kernel, dim3((n_pages + per - 1) / per), dim3(kWG), 0, s, d_pages, n_pages, d_recs,
                       d_results);
    return hipGetLastError();
}

}  // namespace pq
