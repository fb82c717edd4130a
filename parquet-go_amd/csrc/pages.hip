// pages.hip — one workgroup per page descriptor.
//
// Three kernels share the stream decoders of hybrid_stream.hpp / delta_stream.hpp:
//   stream_pages_kernel : PQ_KIND_HYBRID, PQ_KIND_DICT, PQ_KIND_LEVELS
//   delta_pages_kernel  : PQ_KIND_DELTA32 / PQ_KIND_DELTA64
//   data_pages_kernel   : PQ_KIND_PAGE_V1 / PQ_KIND_PAGE_V2 (rep levels, def levels,
//                         then exactly not_null values), the order and the error
//                         precedence of layout/page-v1.go:40-116 and
//                         layout/page-v2.go:42-149.
// Pages are independent (each data page re-creates its decoders,
// layout/chunk.go:196-209), so there is no inter-workgroup traffic of any kind.
#include "delta_stream.hpp"
#include "delta_wave.hpp"
#include "hybrid_stream.hpp"
#include "hybrid_wave.hpp"
#include "bytearray_wave.hpp"
#include "page_finish.hpp"
#include "pq_launch.hpp"

namespace pq {

constexpr uint32_t align16(uint32_t x) { return (x + 15u) & ~15u; }

union PageShared {
    HybridShared hy;
    DeltaShared dl;
};

extern __shared__ __attribute__((aligned(16))) uint8_t pq_smem[];

__device__ __forceinline__ uint32_t load_le32(const uint8_t *p)
{
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

// ---- one hybrid section with the Init / InitSize / never-initialised variants ----

struct Section {
    const uint8_t *S; // first byte of the hybrid stream proper
    int64_t len;      // bytes the decoder may read
    int32_t init_status;
    uint32_t consumed_by_init; // bytes of the enclosing reader taken by InitSize
    bool no_reader;
};

// InitSize on a reader positioned at S with `avail` bytes left
// (encoding/hybrid-decoder.go:57-68): nothing is read when w == 0; otherwise a u32
// LE byte count, then the decoder owns the next min(count, remaining) bytes.
__device__ __forceinline__ Section section_init_size(const uint8_t *S, int64_t avail, uint32_t w)
{
    Section s{S, 0, PQ_OK, 0, false};
    if (w == 0) return s;
    if (avail < 4) {
        s.init_status = avail <= 0 ? PQ_EOF : PQ_UNEXPECTED_EOF;
        s.consumed_by_init = avail <= 0 ? 0u : (uint32_t)avail; // io.ReadFull took what there was
        return s;
    }
    const uint32_t size = load_le32(S);
    s.S = S + 4;
    s.len = min((int64_t)size, avail - 4);
    s.consumed_by_init = 4u + (uint32_t)s.len;
    return s;
}

template <int MODE, typename OutT>
__device__ __forceinline__ HybridResult run_section(HybridShared &sh, const Section &sec, uint32_t w, uint32_t n,
                                           OutT *dst, const OutT *dict, uint32_t dict_len,
                                           uint32_t max_level)
{
    if (sec.no_reader && w != 0) {
        // "reader is not initialized" on the first pull (hybrid-decoder.go:78-80)
        HybridResult r{n > 0 ? PQ_NOT_INITIALIZED : PQ_OK, 0, 0, 0};
        return r;
    }
    return hybrid_stream<MODE, OutT>(sh, sec.S, sec.len, w, n, dst, dict, dict_len, max_level);
}

// Levels from ConstDecoder(0) (encoding/const.go:7-19): n zeros, nothing read.
__device__ __forceinline__ HybridResult const_levels(uint32_t *dst, uint32_t n, uint32_t max_level, bool write = true)
{
    if (write)
        for (uint32_t i = threadIdx.x; i < n; i += kWG) dst[i] = 0;
    HybridResult r{PQ_OK, n, max_level == 0 ? n : 0u, 0};
    return r;
}

// Copy a dictionary into LDS when it fits the budget; returns true if cached.
__device__ __forceinline__ bool cache_dict(uint32_t *ldict, const uint8_t *gdict, uint32_t bytes, uint32_t budget)
{
    if (bytes == 0 || bytes > budget) return false;
    const uint32_t *g = reinterpret_cast<const uint32_t *>(gdict);
    for (uint32_t i = threadIdx.x; i < bytes / 4u; i += kWG) ldict[i] = g[i];
    __syncthreads();
    return true;
}

// PLAIN fixed-width values (types/int32.go:56-72, types/int64.go:56-72, types/float.go:67-79,
// types/double.go:68-80): n values of es little-endian bytes; what is missing is io.EOF, a
// partial value io.ErrUnexpectedEOF.  Result fields only; the copy is done by the callers.
__device__ __forceinline__ HybridResult plain_result(int64_t len, uint32_t n, uint32_t es, uint32_t *n_ok)
{
    const uint64_t have = len > 0 ? (uint64_t)len / es : 0u;
    *n_ok = have < (uint64_t)n ? (uint32_t)have : n;
    HybridResult r{PQ_OK, *n_ok, 0, (uint64_t)*n_ok * es};
    if (*n_ok < n) {
        const bool partial = len > 0 && ((uint64_t)len % es) != 0u;
        r.status = partial ? PQ_UNEXPECTED_EOF : PQ_EOF;
        if (partial) r.consumed = (uint64_t)len;
    }
    return r;
}

// BOOLEAN PLAIN on a fresh decoder (types/boolean.go:59-104): one byte per 8 values; a missing
// byte is io.EOF and the count is the index of the group that could not be read.
__device__ __forceinline__ HybridResult bool_result(int64_t len, uint32_t n, uint32_t *n_ok)
{
    const uint64_t have = len > 0 ? 8u * (uint64_t)len : 0u;
    *n_ok = have < (uint64_t)n ? (uint32_t)have : n;
    HybridResult r{*n_ok < n ? PQ_EOF : PQ_OK, *n_ok, 0, (uint64_t)((*n_ok + 7u) >> 3)};
    return r;
}

// The same two by a whole workgroup, element by element (the fallback kernels).
__device__ __forceinline__ HybridResult plain_values_wg(const uint8_t *S, int64_t len, uint32_t n, uint32_t es,
                                                        uint32_t *dst)
{
    uint32_t n_ok;
    const HybridResult r = plain_result(len, n, es, &n_ok);
    const uint32_t words = n_ok * (es / 4u);
    for (uint32_t i = threadIdx.x; i < words; i += kWG) dst[i] = load_le32(S + 4u * (size_t)i);
    return r;
}

__device__ __forceinline__ HybridResult bool_values_wg(const uint8_t *S, int64_t len, uint32_t n, uint32_t *dst)
{
    uint32_t n_ok;
    const HybridResult r = bool_result(len, n, &n_ok);
    for (uint32_t i = threadIdx.x; i < n_ok; i += kWG) dst[i] = (S[i >> 3] >> (i & 7u)) & 1u;
    return r;
}

// DictDecoder over [S, S+len): bit-width byte, then an unprefixed hybrid stream
// (types/dict.go:66-117).  init_status reports Init alone.
template <typename OutT>
__device__ __forceinline__ HybridResult dict_values(HybridShared &sh, const uint8_t *S, int64_t len, uint32_t n,
                                           OutT *dst, const OutT *gdict, const OutT *ldict,
                                           bool in_lds, uint32_t dict_len, int32_t *init_status)
{
    HybridResult r{PQ_OK, 0, 0, 0};
    if (len <= 0) {
        *init_status = PQ_EOF;
        r.status = PQ_EOF;
        return r;
    }
    const uint32_t w = S[0];
    if (w > 32u) {
        *init_status = PQ_DICT_INVALID_WIDTH;
        r.status = PQ_DICT_INVALID_WIDTH;
        r.consumed = 1;
        return r;
    }
    *init_status = PQ_OK;
    if (in_lds)
        r = hybrid_stream<HY_DICT, OutT>(sh, S + 1, len - 1, w, n, dst, ldict, dict_len, 0);
    else
        r = hybrid_stream<HY_DICT, OutT>(sh, S + 1, len - 1, w, n, dst, gdict, dict_len, 0);
    r.consumed += 1;
    return r;
}

// DELTA_LENGTH_BYTE_ARRAY / DELTA_BYTE_ARRAY in the workgroup-per-stream kernels: the first wave
// does it alone, in the staging window of the hybrid decoder.
static_assert(sizeof(HybridWaveLds) >= kBaLdsBytes && sizeof(HybridShared::chunk) >= kBaLdsBytes,
              "the byte-array decoder stages a batch in the LDS of the decoder it replaces");
static_assert(offsetof(HybridWaveLds, window) == 0, "the byte-array decoder takes the whole struct as its window");

__device__ __forceinline__ ByteArrayResult byte_array_wg(uint32_t kind, HybridShared &sh, const uint8_t *S, int64_t len,
                                                         uint32_t n, const pq_page_desc &d, uint8_t *values)
{
    static_assert(sizeof(sh.chunk) >= (size_t)kDwChunk + 16, "window too small for delta_wave");
    __shared__ ByteArrayResult slot;
    __syncthreads();
    if (threadIdx.x < (unsigned)kWave) {
        const ByteArrayResult br = byte_array_wave(kind, sh.chunk, S, len, n, d, values, (int)threadIdx.x);
        if (threadIdx.x == 0) slot = br;
    }
    __syncthreads();
    return slot;
}

// ---------------------------------------------------------------------------

__global__ __launch_bounds__(kWG) void stream_pages_kernel(const pq_page_desc *__restrict__ descs,
                                                           const uint32_t *__restrict__ ids,
                                                           PageBuffers buf,
                                                           pq_page_result *__restrict__ results,
                                                           uint32_t dict_lds_bytes,
                                                           const HsegPage *__restrict__ hseg_pages, uint32_t n_plain)
{
    HybridShared &sh = *reinterpret_cast<HybridShared *>(pq_smem);
    uint32_t *ldict = reinterpret_cast<uint32_t *>(pq_smem + align16(sizeof(HybridShared)));
    // a stream the segmented decode has taken care of
    if (blockIdx.x >= n_plain && hseg_pages[blockIdx.x - n_plain].mode != 0u) return;
    const uint32_t page = ids[blockIdx.x];
    const pq_page_desc d = descs[page];
    const uint8_t *S = buf.src + d.src_off;
    const int64_t len = (int64_t)d.src_len;
    const uint32_t n = d.n_values;

    HybridResult r{PQ_OK, 0, 0, 0};
    uint32_t extra = 0;
    if (d.kind == PQ_KIND_HYBRID || d.kind == PQ_KIND_LEVELS || d.kind == PQ_KIND_BOOL_RLE) {
        // BOOL_RLE: NewHybridDecoder(1, false).InitSize (types/boolean.go:177-181)
        const bool levels = d.kind == PQ_KIND_LEVELS, brle = d.kind == PQ_KIND_BOOL_RLE;
        const uint32_t w = levels ? bits_len16(d.max_def) : brle ? 1u : d.bit_width;
        const uint32_t flags = brle ? (uint32_t)PQ_FLAG_LENGTH_PREFIXED : d.flags;
        uint32_t *dst = reinterpret_cast<uint32_t *>(levels ? buf.levels + d.def_off : buf.values + d.dst_off);
        if (levels && (flags & PQ_FLAG_LEVELS_CONST)) {
            r = const_levels(dst, n, d.max_def);
        } else {
            Section sec{S, len, PQ_OK, 0, (flags & PQ_FLAG_NO_INIT) != 0};
            if (flags & PQ_FLAG_LENGTH_PREFIXED) sec = section_init_size(S, len, w);
            if (sec.init_status != PQ_OK) {
                r.status = sec.init_status;
                if (brle) extra = PQ_STAGE_VALUES_INIT;  // BooleanRLEDecoder.Init failed, not DecodeValues
            } else if (levels) {
                r = run_section<HY_LEVELS, uint32_t>(sh, sec, w, n, dst, nullptr, 0, d.max_def);
                // decodePackedArray returns (nil, 0, err) on any error (layout/helpers.go:78-80)
                if (r.status != PQ_OK) { r.n_decoded = 0; r.count_max = 0; }
            } else {
                r = run_section<HY_RAW, uint32_t>(sh, sec, w, n, dst, nullptr, 0, 0);
            }
            if (flags & PQ_FLAG_LENGTH_PREFIXED) {
                // the buffered decoder slurped its whole section at InitSize
                r.consumed = sec.consumed_by_init;
            } else if (levels && !sec.no_reader) {
                // level decoders are buffered (layout/chunk.go:87,103): Init reads the
                // whole section it is given, whatever the width
                r.consumed = (uint64_t)len;
            }
        }
    } else if (d.kind == PQ_KIND_DELTA_LENGTH_BA || d.kind == PQ_KIND_DELTA_BA) {
        const ByteArrayResult br = byte_array_wg(d.kind, sh, S, len, n, d, buf.values);
        r.status = br.status;
        r.n_decoded = br.n_decoded;
        r.consumed = br.consumed;
        extra = br.init_failed ? PQ_STAGE_VALUES_INIT : PQ_STAGE_NONE;
    } else if (d.kind == PQ_KIND_PLAIN) {
        r = plain_values_wg(S, len, n, d.elem_size, reinterpret_cast<uint32_t *>(buf.values + d.dst_off));
    } else if (d.kind == PQ_KIND_BOOL_PLAIN) {
        r = bool_values_wg(S, len, n, reinterpret_cast<uint32_t *>(buf.values + d.dst_off));
    } else { // PQ_KIND_DICT
        const uint8_t *gd = buf.dict + d.dict_off;
        const uint32_t dbytes = d.dict_len * d.elem_size;
        const bool in_lds = cache_dict(ldict, gd, dbytes, dict_lds_bytes);
        int32_t init_status;
        if (d.elem_size == 8)
            r = dict_values<uint64_t>(sh, S, len, n, reinterpret_cast<uint64_t *>(buf.values + d.dst_off),
                                      reinterpret_cast<const uint64_t *>(gd),
                                      reinterpret_cast<const uint64_t *>(ldict), in_lds, d.dict_len,
                                      &init_status);
        else
            r = dict_values<uint32_t>(sh, S, len, n, reinterpret_cast<uint32_t *>(buf.values + d.dst_off),
                                      reinterpret_cast<const uint32_t *>(gd),
                                      reinterpret_cast<const uint32_t *>(ldict), in_lds, d.dict_len,
                                      &init_status);
        if (init_status != PQ_OK) extra = PQ_STAGE_VALUES_INIT;  // DictDecoder.Init failed
    }
    if (threadIdx.x == 0) {
        pq_page_result out;
        out.status = r.status;
        out.stage = (int32_t)extra;
        out.detail = 0;
        out.n_decoded = r.n_decoded;
        out.not_null = r.count_max;
        out.n_levels = 0;
        out.consumed = r.consumed;
        results[page] = out;
    }
}

// ---------------------------------------------------------------------------

// delta_stream reports status kNeedSerial for layouts outside its tiled path; the
// page is then finished by delta_serial_kernel, launched right behind.
constexpr int32_t kNeedSerial = -1;

// SWEEP = false: every descriptor in ids[] is decoded; layouts outside the tiled path
//                 are left marked kNeedSerial for delta_serial_kernel.
// SWEEP = true : runs behind delta_wave_kernel and only looks at the pages that kernel
//                 declined (marked kNeedGeneric); `flag` holds the current `epoch` iff
//                 there is at least one, so the usual launch ends after one load.  A
//                 page the tiled path cannot take either is finished here by one lane.
template <typename UT, bool SWEEP>
__global__ __launch_bounds__(kWG) void delta_pages_kernel(const pq_page_desc *__restrict__ descs,
                                                          const uint32_t *__restrict__ ids, uint32_t count,
                                                          PageBuffers buf, pq_page_result *results,
                                                          const uint32_t *flag, uint32_t epoch)
{
    __shared__ DeltaShared sh;
    if (SWEEP && *flag != epoch) return;
    for (uint32_t i = blockIdx.x; i < count; i += gridDim.x) {
        const uint32_t page = ids[i];
        if (SWEEP && results[page].status != kNeedGeneric) continue;
        const pq_page_desc d = descs[page];
        const uint8_t *S = buf.src + d.src_off;
        UT *dst = reinterpret_cast<UT *>(buf.values + d.dst_off);
        DeltaResult r = delta_stream<UT>(sh, S, (int64_t)d.src_len, d.n_values, dst);
        if (threadIdx.x == 0) {
            if (SWEEP && r.status == kNeedSerial) r = delta_serial<UT>(S, (int64_t)d.src_len, d.n_values, dst);
            pq_page_result out;
            out.status = r.status;
            out.stage = r.init_failed ? PQ_STAGE_VALUES_INIT : PQ_STAGE_NONE;  // Init failed, not DecodeValues
            out.detail = 0;
            out.n_decoded = r.n_decoded;
            out.not_null = 0;
            out.n_levels = 0;
            out.consumed = r.consumed;
            results[page] = out;
        }
        __syncthreads();
    }
}

// One wave per page (4 pages per workgroup, no workgroup barrier anywhere).
template <typename UT>
__global__ __launch_bounds__(kWG) void delta_wave_kernel(const pq_page_desc *__restrict__ descs,
                                                         const uint32_t *__restrict__ ids, uint32_t count,
                                                         PageBuffers buf,
                                                         pq_page_result *__restrict__ results,
                                                         uint32_t *flag, uint32_t epoch)
{
    __shared__ __attribute__((aligned(16))) uint32_t window[kWG / kWave][kDwChunk / 4 + 4];
    const int lane = threadIdx.x & (kWave - 1);
    // the wave index is the same in all 64 lanes; telling the compiler so keeps the
    // page descriptor, the stream walk and all loop control on the scalar unit
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t slot = blockIdx.x * (kWG / kWave) + (uint32_t)wave;
    if (slot >= count) return;
    const uint32_t page = ids[slot];
    const pq_page_desc d = descs[page];
    const WaveDeltaResult r = delta_wave<UT>(window[wave], buf.src + d.src_off, (int64_t)d.src_len, d.n_values,
                                             reinterpret_cast<UT *>(buf.values + d.dst_off), lane);
    if (lane == 0) {
        pq_page_result out;
        out.status = r.status;
        out.stage = r.init_failed ? PQ_STAGE_VALUES_INIT : PQ_STAGE_NONE;
        out.detail = 0;
        out.n_decoded = r.n_decoded;
        out.not_null = 0;
        out.n_levels = 0;
        out.consumed = r.consumed;
        results[page] = out;
        if (r.status == kNeedGeneric) *flag = epoch; // wakes the sweep kernel behind this one
    }
}

// ---------------------------------------------------------------------------

// One data page by one workgroup.
__device__ __forceinline__ void data_page_wg(PageShared &sh, uint32_t *ldict, const pq_page_desc &d, uint32_t page,
                                             PageBuffers buf, pq_page_result *results, uint32_t dict_lds_bytes)
{
    const uint8_t *S = buf.src + d.src_off;
    const int64_t len = (int64_t)d.src_len;
    const uint32_t size = d.n_values;
    uint32_t *rep_dst = reinterpret_cast<uint32_t *>(buf.levels + d.rep_off);
    uint32_t *def_dst = reinterpret_cast<uint32_t *>(buf.levels + d.def_off);

    pq_page_result out;
    out.status = PQ_OK;
    out.stage = PQ_STAGE_NONE;
    out.detail = 0;
    out.n_decoded = 0;
    out.not_null = 0;
    out.n_levels = 0;
    out.consumed = 0;

    // ---- read(): where each decoder's bytes are ----
    const bool konst = d.max_rep == 0; // layout/chunk.go:111-124: both level decoders become ConstDecoder(0)
    const uint32_t wr = bits_len16(d.max_rep), wd = bits_len16(d.max_def);
    Section rep{S, 0, PQ_OK, 0, false}, def{S, 0, PQ_OK, 0, false};
    int64_t vpos = 0; // where the values decoder starts
    if (!konst) {
        if (d.kind == PQ_KIND_PAGE_V1) {
            rep = section_init_size(S, len, wr);
            if (rep.init_status != PQ_OK) {
                out.status = rep.init_status;
                out.stage = PQ_STAGE_REP_INIT;
            } else {
                vpos = rep.consumed_by_init;
                def = section_init_size(S + vpos, len - vpos, wd);
                if (def.init_status != PQ_OK) {
                    out.status = def.init_status;
                    out.stage = PQ_STAGE_DEF_INIT;
                } else {
                    vpos += def.consumed_by_init;
                }
            }
        } else {
            const int64_t lv = (int64_t)d.rep_len + (int64_t)d.def_len;
            if (lv > 0) {
                if (lv > len) { // io.ReadFull of the level bytes (page-v2.go:87-90)
                    out.status = len <= 0 ? PQ_EOF : PQ_UNEXPECTED_EOF;
                    out.stage = PQ_STAGE_REP_INIT;
                }
                rep.S = S;
                rep.len = d.rep_len;
                rep.no_reader = d.rep_len == 0;
                def.S = S + d.rep_len;
                def.len = d.def_len;
                def.no_reader = d.def_len == 0;
                vpos = lv;
            } else {
                rep.no_reader = true;
                def.no_reader = true;
            }
        }
    } else if (d.kind == PQ_KIND_PAGE_V2) {
        // ConstDecoder.Init ignores its bytes, but the level bytes are still skipped
        const int64_t lv = (int64_t)d.rep_len + (int64_t)d.def_len;
        if (lv > len) {
            out.status = len <= 0 ? PQ_EOF : PQ_UNEXPECTED_EOF;
            out.stage = PQ_STAGE_REP_INIT;
        }
        vpos = lv;
    }

    if (out.status != PQ_OK) { // failed before the values decoder was initialised
        if (threadIdx.x == 0) results[page] = out;
        return;
    }

    // ---- ReadValues: levels ----
    int32_t lvl_status = PQ_OK, lvl_stage = PQ_STAGE_NONE;
    uint32_t not_null = 0;
    if (size != 0) {
        const bool zeros_out = (d.flags & PQ_FLAG_NO_LEVEL_OUTPUT) == 0u;
        HybridResult rr = konst ? const_levels(rep_dst, size, d.max_rep, zeros_out)
                                : run_section<HY_LEVELS, uint32_t>(sh.hy, rep, wr, size, rep_dst, nullptr, 0, d.max_rep);
        if (rr.status != PQ_OK) {
            lvl_status = rr.status;
            lvl_stage = PQ_STAGE_REP;
        } else {
            HybridResult dr = konst ? const_levels(def_dst, size, d.max_def, zeros_out)
                                    : run_section<HY_LEVELS, uint32_t>(sh.hy, def, wd, size, def_dst, nullptr, 0, d.max_def);
            if (dr.status != PQ_OK) {
                lvl_status = dr.status;
                lvl_stage = PQ_STAGE_DEF;
            } else {
                not_null = dr.count_max;
            }
        }
    }
    const uint32_t nvals = (lvl_status == PQ_OK) ? not_null : 0;

    // ---- values: Init always runs (read()), DecodeValues only for not_null != 0 ----
    const uint8_t *VS = S + vpos;
    const int64_t vlen = len - vpos;
    int32_t v_init = PQ_OK, v_status = PQ_OK;
    uint32_t v_count = 0;
    if (d.value_kind == PQ_KIND_DICT) {
        const uint8_t *gd = buf.dict + d.dict_off;
        const bool in_lds = cache_dict(ldict, gd, d.dict_len * d.elem_size, dict_lds_bytes);
        HybridResult vr;
        if (d.elem_size == 8)
            vr = dict_values<uint64_t>(sh.hy, VS, vlen, nvals, reinterpret_cast<uint64_t *>(buf.values + d.dst_off),
                                       reinterpret_cast<const uint64_t *>(gd),
                                       reinterpret_cast<const uint64_t *>(ldict), in_lds, d.dict_len, &v_init);
        else
            vr = dict_values<uint32_t>(sh.hy, VS, vlen, nvals, reinterpret_cast<uint32_t *>(buf.values + d.dst_off),
                                       reinterpret_cast<const uint32_t *>(gd),
                                       reinterpret_cast<const uint32_t *>(ldict), in_lds, d.dict_len, &v_init);
        v_status = vr.status;
        v_count = vr.n_decoded;
    } else if (d.value_kind == PQ_KIND_DELTA_LENGTH_BA || d.value_kind == PQ_KIND_DELTA_BA) {
        const ByteArrayResult br = byte_array_wg(d.value_kind, sh.hy, VS, vlen, nvals, d, buf.values);
        if (br.init_failed) v_init = br.status;
        v_status = br.status;
        v_count = br.n_decoded;
    } else if (d.value_kind == PQ_KIND_BOOL_RLE) {
        // BooleanRLEDecoder: InitSize at read(), then n x Next (types/boolean.go:177-196)
        const Section sec = section_init_size(VS, vlen, 1u);
        v_init = sec.init_status;
        if (v_init == PQ_OK) {
            const HybridResult vr = run_section<HY_RAW, uint32_t>(
                sh.hy, sec, 1u, nvals, reinterpret_cast<uint32_t *>(buf.values + d.dst_off), nullptr, 0, 0);
            v_status = vr.status;
            v_count = vr.n_decoded;
        }
    } else if (d.value_kind == PQ_KIND_PLAIN || d.value_kind == PQ_KIND_BOOL_PLAIN) {
        // Init only keeps the reader (types/int32.go:50-53)
        uint32_t *vdst = reinterpret_cast<uint32_t *>(buf.values + d.dst_off);
        const HybridResult vr = d.value_kind == PQ_KIND_PLAIN ? plain_values_wg(VS, vlen, nvals, d.elem_size, vdst)
                                                              : bool_values_wg(VS, vlen, nvals, vdst);
        v_status = vr.status;
        v_count = vr.n_decoded;
    } else {
        DeltaResult vr;
        if (d.value_kind == PQ_KIND_DELTA64)
            vr = delta_stream<uint64_t>(sh.dl, VS, vlen, nvals, reinterpret_cast<uint64_t *>(buf.values + d.dst_off));
        else
            vr = delta_stream<uint32_t>(sh.dl, VS, vlen, nvals, reinterpret_cast<uint32_t *>(buf.values + d.dst_off));
        if (vr.status == kNeedSerial) {
            // park what ReadValues knows so far; delta_serial_kernel completes the page
            if (threadIdx.x == 0) {
                out.status = kNeedSerial;
                out.stage = lvl_stage;
                out.detail = lvl_status;
                out.n_decoded = nvals;
                out.not_null = not_null;
                out.n_levels = size;
                out.consumed = (uint64_t)vpos;
                results[page] = out;
            }
            return;
        }
        if (vr.init_failed) v_init = vr.status;
        v_status = vr.status;
        v_count = vr.n_decoded;
    }

    finish_page(out, v_init, v_status, v_count, lvl_status, lvl_stage, not_null, nvals, size);
    if (threadIdx.x == 0) results[page] = out;
}

// SWEEP = false: one workgroup per descriptor in ids[].
// SWEEP = true : runs behind data_wave_kernel and redoes only the pages that kernel declined
//                 (marked kNeedGeneric); `flag` holds the current `epoch` iff there is one.
template <bool SWEEP>
__global__ __launch_bounds__(kWG) void data_pages_kernel(const pq_page_desc *__restrict__ descs,
                                                         const uint32_t *__restrict__ ids, uint32_t count,
                                                         PageBuffers buf, pq_page_result *results,
                                                         uint32_t dict_lds_bytes, const uint32_t *flag,
                                                         uint32_t epoch)
{
    PageShared &sh = *reinterpret_cast<PageShared *>(pq_smem);
    uint32_t *ldict = reinterpret_cast<uint32_t *>(pq_smem + align16(sizeof(PageShared)));
    if (SWEEP && *flag != epoch) return;
    for (uint32_t i = blockIdx.x; i < count; i += gridDim.x) {
        const uint32_t page = ids[i];
        if (SWEEP && results[page].status != kNeedGeneric) continue;
        const pq_page_desc d = descs[page];
        data_page_wg(sh, ldict, d, page, buf, results, dict_lds_bytes);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// One WAVE per descriptor: kHwWaves descriptors per workgroup, no barrier after the optional
// dictionary load.  A workgroup keeps ONE dictionary in LDS: the one its first descriptor reads,
// when it fits the launch's budget.  Its other waves use that copy when their descriptor names the
// same dictionary (same offset, same size: the pages of one column chunk, which a plan keeps side
// by side) and gather from global memory otherwise.

constexpr int kHwWG = 512;               // threads per workgroup of the hybrid wave kernels
constexpr int kHwWaves = kHwWG / kWave;   // descriptors per workgroup

struct WaveDict {
    const uint8_t *g;  // the page's dictionary in global memory
    const uint32_t *l; // the workgroup's LDS copy, or nullptr
};

// The dictionary a descriptor reads: bytes == 0 when it reads none.
struct DictRef {
    uint64_t off;
    uint32_t bytes;
};

__device__ __forceinline__ DictRef dict_ref(const pq_page_desc &d)
{
    const bool page = d.kind == PQ_KIND_PAGE_V1 || d.kind == PQ_KIND_PAGE_V2;
    const bool user = d.kind == PQ_KIND_DICT || (page && d.value_kind == PQ_KIND_DICT);
    DictRef r{d.dict_off, user ? d.dict_len * d.elem_size : 0u};
    return r;
}

// Loads the dictionary of descriptor `first` (the workgroup's first) into LDS when it fits
// `budget`; every thread of the workgroup calls this with the same arguments.  Returns what is
// in LDS now (bytes == 0: nothing).
template <int WG>
__device__ __forceinline__ DictRef load_wg_dict(uint32_t *ldict, const pq_page_desc &first, PageBuffers buf, uint32_t budget)
{
    DictRef r = dict_ref(first);
    r.off = ((uint64_t)uni((uint32_t)(r.off >> 32)) << 32) | uni((uint32_t)r.off);
    r.bytes = uni(r.bytes);
    if (budget == 0u || r.bytes == 0u || r.bytes > budget) {
        r.bytes = 0u;
        return r;
    }
    const uint32_t *g = reinterpret_cast<const uint32_t *>(buf.dict + r.off);
    for (uint32_t i = threadIdx.x; i < r.bytes / 4u; i += WG) ldict[i] = g[i];
    __syncthreads();
    return r;
}

// The LDS copy if it is this descriptor's dictionary, else nullptr.
__device__ __forceinline__ const uint32_t *wg_dict_for(const uint32_t *ldict, DictRef have, const pq_page_desc &d)
{
    const DictRef want = dict_ref(d);
    return have.bytes != 0u && want.bytes == have.bytes && want.off == have.off ? ldict : nullptr;
}

// TAB (DnTableMode): whether the wave walks a window's run headers in front of its run-dense tiles.  The page
// kernel does (nested pages are where such tiles are most of the work); the stream kernel does not: its
// code for streams of long bit-packed runs comes out slower when the rest needs more registers.
template <int MODE, typename OutT, int TAB = kTabNone>
__device__ __forceinline__ HybridResult run_section_wave(HybridWaveLds &L, const Section &sec, uint32_t w, uint32_t n,
                                                         OutT *dst, const OutT *dict, uint32_t dict_len,
                                                         uint32_t max_level, int lane)
{
    if (sec.no_reader && w != 0) {
        HybridResult r{n > 0 ? PQ_NOT_INITIALIZED : PQ_OK, 0, 0, 0};
        return r;
    }
    return hybrid_wave<MODE, OutT, TAB>(L, sec.S, sec.len, w, n, dst, dict, dict_len, max_level, lane);
}

__device__ __forceinline__ HybridResult const_levels_wave(uint32_t *dst, uint32_t n, uint32_t max_level, int lane,
                                                          bool write = true)
{
    if (write)
        for (uint32_t i = (uint32_t)lane; i < n; i += kWave) dst[i] = 0;
    HybridResult r{PQ_OK, n, max_level == 0 ? n : 0u, 0};
    return r;
}

template <typename OutT, int TAB = kTabNone>
__device__ __forceinline__ HybridResult dict_values_wave(HybridWaveLds &L, const uint8_t *S, int64_t len, uint32_t n,
                                                         OutT *dst, WaveDict dict, uint32_t dict_len,
                                                         int32_t *init_status, int lane)
{
    HybridResult r{PQ_OK, 0, 0, 0};
    if (len <= 0) {
        *init_status = PQ_EOF;
        r.status = PQ_EOF;
        return r;
    }
    const uint32_t w = uni((uint32_t)S[0]);
    if (w > 32u) {
        *init_status = PQ_DICT_INVALID_WIDTH;
        r.status = PQ_DICT_INVALID_WIDTH;
        r.consumed = 1;
        return r;
    }
    *init_status = PQ_OK;
    if (dict.l)
        r = hybrid_wave<HY_DICT, OutT, TAB>(L, S + 1, len - 1, w, n, dst, reinterpret_cast<const OutT *>(dict.l), dict_len,
                                       0, lane);
    else
        r = hybrid_wave<HY_DICT, OutT, TAB>(L, S + 1, len - 1, w, n, dst, reinterpret_cast<const OutT *>(dict.g), dict_len,
                                       0, lane);
    r.consumed += 1;
    return r;
}

// PLAIN values are copied as they are; BOOLEAN PLAIN bytes are headerless groups of width 1 and ride
// the bulk path of hybrid_wave.
__device__ __forceinline__ HybridResult plain_values_wave(HybridWaveLds &L, const uint8_t *S, int64_t len, uint32_t n,
                                                          uint32_t es, uint32_t *dst, int lane)
{
    uint32_t n_ok;
    const HybridResult r = plain_result(len, n, es, &n_ok);
    // little-endian values are their own bytes: a copy, 16 bytes per lane and step
    (void)L;
    wave_copy_bytes(reinterpret_cast<uint8_t *>(dst), S, (uint64_t)n_ok * es, lane);
    return r;
}

__device__ __forceinline__ HybridResult bool_values_wave(HybridWaveLds &L, const uint8_t *S, int64_t len, uint32_t n,
                                                         uint32_t *dst, int lane)
{
    uint32_t n_ok;
    const HybridResult r = bool_result(len, n, &n_ok);
    if (n_ok == 0u) return r;
    if ((reinterpret_cast<uintptr_t>(dst) & 15u) != 0u) {
        (void)hybrid_wave<HY_RAW, uint32_t>(L, S, len, 1u, n_ok, dst, nullptr, 0, 0, lane, (n_ok + 7u) >> 3);
        return r;
    }
    // One bit a value, least significant first (types/boolean.go:86-101): a lane takes the four bits of
    // values 4 lane .. 4 lane + 3 of a block of 256 (half a byte of the block's 32) and stores 16 bytes;
    // four blocks a step, so that four loads are in flight.
    const uint32_t blocks = n_ok >> 8;
    const uint32_t half = (uint32_t)lane >> 1, sh = ((uint32_t)lane & 1u) * 4u;
    for (uint32_t b0 = 0; b0 < blocks; b0 += 4u) {
        uint32_t byte[4];
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u)
            byte[u] = b0 + u < blocks ? (uint32_t)S[32u * (size_t)(b0 + u) + half] : 0u;
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) {
            if (b0 + u < blocks) {
                const uint32_t nib = byte[u] >> sh;
                store_nt16(dst + 256u * (size_t)(b0 + u) + 4u * (uint32_t)lane, nib & 1u, (nib >> 1) & 1u, (nib >> 2) & 1u,
                           (nib >> 3) & 1u);
            }
        }
    }
    for (uint32_t i = 256u * blocks + (uint32_t)lane; i < n_ok; i += kWave) dst[i] = ((uint32_t)S[i >> 3] >> (i & 7u)) & 1u;
    return r;
}

__global__ __launch_bounds__(kHwWG) void stream_wave_kernel(const pq_page_desc *__restrict__ descs,
                                                          const uint32_t *__restrict__ ids, uint32_t count,
                                                          PageBuffers buf, pq_page_result *__restrict__ results,
                                                          uint32_t dict_lds_bytes,
                                                          const HsegPage *__restrict__ hseg_pages, uint32_t n_plain)
{
    HybridWaveLds *LW = reinterpret_cast<HybridWaveLds *>(pq_smem);
    uint32_t *lds_dict = reinterpret_cast<uint32_t *>(pq_smem + align16((uint32_t)sizeof(HybridWaveLds) * (kHwWaves)));
    const DictRef have = load_wg_dict<kHwWG>(lds_dict, descs[ids[blockIdx.x * kHwWaves]], buf, dict_lds_bytes);
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t slot = blockIdx.x * (kHwWaves) + (uint32_t)wave;
    if (slot >= count) return;
    // a stream the segmented decode has taken care of
    if (slot >= n_plain && uni(hseg_pages[slot - n_plain].mode) != 0u) return;
    HybridWaveLds &L = LW[wave];
    const uint32_t page = ids[slot];
    const pq_page_desc d = descs[page];
    const uint32_t *ldict = wg_dict_for(lds_dict, have, d);
    const uint8_t *S = buf.src + d.src_off;
    const int64_t len = (int64_t)d.src_len;
    const uint32_t n = d.n_values;

    HybridResult r{PQ_OK, 0, 0, 0};
    int32_t stage = PQ_STAGE_NONE;  // DELTA_LENGTH_BA: tells a failed Init from a failed DecodeValues
    if (d.kind == PQ_KIND_HYBRID || d.kind == PQ_KIND_LEVELS || d.kind == PQ_KIND_BOOL_RLE) {
        // BOOL_RLE: NewHybridDecoder(1, false).InitSize (types/boolean.go:177-181)
        const bool levels = d.kind == PQ_KIND_LEVELS, brle = d.kind == PQ_KIND_BOOL_RLE;
        const uint32_t w = levels ? bits_len16(d.max_def) : brle ? 1u : d.bit_width;
        const uint32_t flags = brle ? (uint32_t)PQ_FLAG_LENGTH_PREFIXED : d.flags;
        uint32_t *dst = reinterpret_cast<uint32_t *>(levels ? buf.levels + d.def_off : buf.values + d.dst_off);
        if (levels && (flags & PQ_FLAG_LEVELS_CONST)) {
            r = const_levels_wave(dst, n, d.max_def, lane);
        } else {
            Section sec{S, len, PQ_OK, 0, (flags & PQ_FLAG_NO_INIT) != 0};
            if (flags & PQ_FLAG_LENGTH_PREFIXED) sec = section_init_size(S, len, w);
            if (sec.init_status != PQ_OK) {
                r.status = sec.init_status;
                if (brle) stage = PQ_STAGE_VALUES_INIT;  // BooleanRLEDecoder.Init failed, not DecodeValues
            } else if (levels) {
                r = run_section_wave<HY_LEVELS, uint32_t>(L, sec, w, n, dst, nullptr, 0, d.max_def, lane);
                if (r.status != PQ_OK) { r.n_decoded = 0; r.count_max = 0; }
            } else {
                r = run_section_wave<HY_RAW, uint32_t>(L, sec, w, n, dst, nullptr, 0, 0, lane);
            }
            if (flags & PQ_FLAG_LENGTH_PREFIXED)
                r.consumed = sec.consumed_by_init;
            else if (levels && !sec.no_reader)
                r.consumed = (uint64_t)len;
        }
    } else if (d.kind == PQ_KIND_DELTA_LENGTH_BA || d.kind == PQ_KIND_DELTA_BA) {
        const ByteArrayResult br = byte_array_wave(d.kind, reinterpret_cast<uint32_t *>(&L), S, len, n, d, buf.values, lane);
        r.status = br.status;
        r.n_decoded = br.n_decoded;
        r.consumed = br.consumed;
        stage = br.init_failed ? PQ_STAGE_VALUES_INIT : PQ_STAGE_NONE;
    } else if (d.kind == PQ_KIND_PLAIN) {
        r = plain_values_wave(L, S, len, n, d.elem_size, reinterpret_cast<uint32_t *>(buf.values + d.dst_off), lane);
    } else if (d.kind == PQ_KIND_BOOL_PLAIN) {
        r = bool_values_wave(L, S, len, n, reinterpret_cast<uint32_t *>(buf.values + d.dst_off), lane);
    } else { // PQ_KIND_DICT
        WaveDict wd{buf.dict + d.dict_off, ldict};
        int32_t init_status;
        if (d.elem_size == 8)
            r = dict_values_wave<uint64_t>(L, S, len, n, reinterpret_cast<uint64_t *>(buf.values + d.dst_off), wd,
                                           d.dict_len, &init_status, lane);
        else
            r = dict_values_wave<uint32_t>(L, S, len, n, reinterpret_cast<uint32_t *>(buf.values + d.dst_off), wd,
                                           d.dict_len, &init_status, lane);
        if (init_status != PQ_OK) stage = PQ_STAGE_VALUES_INIT;  // DictDecoder.Init failed
    }
    if (lane == 0) {
        pq_page_result out;
        out.status = r.status;
        out.stage = stage;
        out.detail = 0;
        out.n_decoded = r.n_decoded;
        out.not_null = r.count_max;
        out.n_levels = 0;
        out.consumed = r.consumed;
        results[page] = out;
    }
}

// A whole V1 / V2 data page by one wave: same order and error precedence as data_page_wg.
// 128 vector registers: four waves a SIMD, as before the table's code came in
__global__ __launch_bounds__(kHwWG) __attribute__((amdgpu_waves_per_eu(4, 4))) void data_wave_kernel(const pq_page_desc *__restrict__ descs,
                                                        const uint32_t *__restrict__ ids, uint32_t count,
                                                        PageBuffers buf, pq_page_result *__restrict__ results,
                                                        uint32_t dict_lds_bytes, uint32_t *flag, uint32_t epoch)
{
    HybridWaveLds *LW = reinterpret_cast<HybridWaveLds *>(pq_smem);
    uint32_t *lds_dict = reinterpret_cast<uint32_t *>(pq_smem + align16((uint32_t)sizeof(HybridWaveLds) * (kHwWaves)));
    const DictRef have = load_wg_dict<kHwWG>(lds_dict, descs[ids[blockIdx.x * kHwWaves]], buf, dict_lds_bytes);
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t slot = blockIdx.x * (kHwWaves) + (uint32_t)wave;
    if (slot >= count) return;
    HybridWaveLds &L = LW[wave];
    const uint32_t page = ids[slot];
    const pq_page_desc d = descs[page];
    const uint32_t *ldict = wg_dict_for(lds_dict, have, d);
    const uint8_t *S = buf.src + d.src_off;
    const int64_t len = (int64_t)d.src_len;
    const uint32_t size = d.n_values;
    uint32_t *rep_dst = reinterpret_cast<uint32_t *>(buf.levels + d.rep_off);
    uint32_t *def_dst = reinterpret_cast<uint32_t *>(buf.levels + d.def_off);

    pq_page_result out;
    out.status = PQ_OK;
    out.stage = PQ_STAGE_NONE;
    out.detail = 0;
    out.n_decoded = 0;
    out.not_null = 0;
    out.n_levels = 0;
    out.consumed = 0;

    // ---- read(): where each decoder's bytes are (see data_page_wg) ----
    const bool konst = d.max_rep == 0;
    const uint32_t wr = bits_len16(d.max_rep), wdw = bits_len16(d.max_def);
    Section rep{S, 0, PQ_OK, 0, false}, def{S, 0, PQ_OK, 0, false};
    int64_t vpos = 0;
    if (!konst) {
        if (d.kind == PQ_KIND_PAGE_V1) {
            rep = section_init_size(S, len, wr);
            if (rep.init_status != PQ_OK) {
                out.status = rep.init_status;
                out.stage = PQ_STAGE_REP_INIT;
            } else {
                vpos = rep.consumed_by_init;
                def = section_init_size(S + vpos, len - vpos, wdw);
                if (def.init_status != PQ_OK) {
                    out.status = def.init_status;
                    out.stage = PQ_STAGE_DEF_INIT;
                } else {
                    vpos += def.consumed_by_init;
                }
            }
        } else {
            const int64_t lv = (int64_t)d.rep_len + (int64_t)d.def_len;
            if (lv > 0) {
                if (lv > len) {
                    out.status = len <= 0 ? PQ_EOF : PQ_UNEXPECTED_EOF;
                    out.stage = PQ_STAGE_REP_INIT;
                }
                rep.S = S;
                rep.len = d.rep_len;
                rep.no_reader = d.rep_len == 0;
                def.S = S + d.rep_len;
                def.len = d.def_len;
                def.no_reader = d.def_len == 0;
                vpos = lv;
            } else {
                rep.no_reader = true;
                def.no_reader = true;
            }
        }
    } else if (d.kind == PQ_KIND_PAGE_V2) {
        const int64_t lv = (int64_t)d.rep_len + (int64_t)d.def_len;
        if (lv > len) {
            out.status = len <= 0 ? PQ_EOF : PQ_UNEXPECTED_EOF;
            out.stage = PQ_STAGE_REP_INIT;
        }
        vpos = lv;
    }
    if (out.status != PQ_OK) {
        if (lane == 0) results[page] = out;
        return;
    }

    // ---- ReadValues: levels ----
    int32_t lvl_status = PQ_OK, lvl_stage = PQ_STAGE_NONE;
    uint32_t not_null = 0;
    if (size != 0) {
        const bool zeros_out = (d.flags & PQ_FLAG_NO_LEVEL_OUTPUT) == 0u;
        // (the count of repetition levels equal to the maximum is never used: layout/page-v1.go:88-101)
        HybridResult rr = konst ? const_levels_wave(rep_dst, size, d.max_rep, lane, zeros_out)
                                : run_section_wave<HY_RAW, uint32_t, kTabScan>(L, rep, wr, size, rep_dst, nullptr, 0, 0, lane);
        if (rr.status != PQ_OK) {
            lvl_status = rr.status;
            lvl_stage = PQ_STAGE_REP;
        } else {
            HybridResult dr = konst ? const_levels_wave(def_dst, size, d.max_def, lane, zeros_out)
                                    : run_section_wave<HY_LEVELS, uint32_t, kTabScan>(L, def, wdw, size, def_dst, nullptr, 0,
                                                                             d.max_def, lane);
            if (dr.status != PQ_OK) {
                lvl_status = dr.status;
                lvl_stage = PQ_STAGE_DEF;
            } else {
                not_null = dr.count_max;
            }
        }
    }
    const uint32_t nvals = (lvl_status == PQ_OK) ? not_null : 0;

    // ---- values ----
    const uint8_t *VS = S + vpos;
    const int64_t vlen = len - vpos;
    int32_t v_init = PQ_OK, v_status = PQ_OK;
    uint32_t v_count = 0;
    if (d.value_kind == PQ_KIND_DICT) {
        WaveDict wd{buf.dict + d.dict_off, ldict};
        HybridResult vr;
        if (d.elem_size == 8)
            vr = dict_values_wave<uint64_t, kTabScan>(L, VS, vlen, nvals, reinterpret_cast<uint64_t *>(buf.values + d.dst_off), wd,
                                            d.dict_len, &v_init, lane);
        else
            vr = dict_values_wave<uint32_t, kTabScan>(L, VS, vlen, nvals, reinterpret_cast<uint32_t *>(buf.values + d.dst_off), wd,
                                            d.dict_len, &v_init, lane);
        v_status = vr.status;
        v_count = vr.n_decoded;
    } else if (d.value_kind == PQ_KIND_DELTA_LENGTH_BA || d.value_kind == PQ_KIND_DELTA_BA) {
        const ByteArrayResult br = byte_array_wave(d.value_kind, reinterpret_cast<uint32_t *>(&L), VS, vlen, nvals, d, buf.values, lane);
        if (br.init_failed) v_init = br.status;
        v_status = br.status;
        v_count = br.n_decoded;
    } else if (d.value_kind == PQ_KIND_BOOL_RLE) {
        const Section sec = section_init_size(VS, vlen, 1u);
        v_init = sec.init_status;
        if (v_init == PQ_OK) {
            const HybridResult vr = run_section_wave<HY_RAW, uint32_t, kTabScan>(
                L, sec, 1u, nvals, reinterpret_cast<uint32_t *>(buf.values + d.dst_off), nullptr, 0, 0, lane);
            v_status = vr.status;
            v_count = vr.n_decoded;
        }
    } else if (d.value_kind == PQ_KIND_PLAIN || d.value_kind == PQ_KIND_BOOL_PLAIN) {
        uint32_t *vdst = reinterpret_cast<uint32_t *>(buf.values + d.dst_off);
        const HybridResult vr = d.value_kind == PQ_KIND_PLAIN
                                    ? plain_values_wave(L, VS, vlen, nvals, d.elem_size, vdst, lane)
                                    : bool_values_wave(L, VS, vlen, nvals, vdst, lane);
        v_status = vr.status;
        v_count = vr.n_decoded;
    } else {
        WaveDeltaResult vr;
        if (d.value_kind == PQ_KIND_DELTA64)
            vr = delta_wave<uint64_t>(L.window, VS, vlen, nvals, reinterpret_cast<uint64_t *>(buf.values + d.dst_off), lane);
        else
            vr = delta_wave<uint32_t>(L.window, VS, vlen, nvals, reinterpret_cast<uint32_t *>(buf.values + d.dst_off), lane);
        if (vr.status == kNeedGeneric) {
            // a layout other than 128 / 4: the workgroup-per-page sweep redoes the page
            if (lane == 0) {
                out.status = kNeedGeneric;
                results[page] = out;
                *flag = epoch;
            }
            return;
        }
        if (vr.init_failed) v_init = vr.status;
        v_status = vr.status;
        v_count = vr.n_decoded;
    }
    finish_page(out, v_init, v_status, v_count, lvl_status, lvl_stage, not_null, nvals, size);
    if (lane == 0) results[page] = out;
}

// Completes the delta pages whose layout the tiled path declined: one lane per page.
__global__ __launch_bounds__(kWave) void delta_serial_kernel(const pq_page_desc *__restrict__ descs,
                                                             const uint32_t *__restrict__ ids, uint32_t count,
                                                             PageBuffers buf,
                                                             pq_page_result *__restrict__ results)
{
    const uint32_t i = blockIdx.x * kWave + threadIdx.x;
    if (i >= count) return;
    const uint32_t page = ids[i];
    pq_page_result out = results[page];
    if (out.status != kNeedSerial) return;
    const pq_page_desc d = descs[page];
    const bool is_page = d.kind == PQ_KIND_PAGE_V1 || d.kind == PQ_KIND_PAGE_V2;
    const uint32_t vk = is_page ? d.value_kind : d.kind;
    const int64_t vpos = is_page ? (int64_t)out.consumed : 0;
    const uint32_t n = is_page ? out.n_decoded : d.n_values;
    const uint8_t *S = buf.src + d.src_off + vpos;
    const int64_t len = (int64_t)d.src_len - vpos;
    DeltaResult r;
    if (vk == PQ_KIND_DELTA64)
        r = delta_serial<uint64_t>(S, len, n, reinterpret_cast<uint64_t *>(buf.values + d.dst_off));
    else
        r = delta_serial<uint32_t>(S, len, n, reinterpret_cast<uint32_t *>(buf.values + d.dst_off));
    if (!is_page) {
        out.status = r.status;
        out.stage = r.init_failed ? PQ_STAGE_VALUES_INIT : PQ_STAGE_NONE;
        out.detail = 0;
        out.n_decoded = r.n_decoded;
        out.not_null = 0;
        out.n_levels = 0;
        out.consumed = r.consumed;
    } else {
        const int32_t lvl_stage = out.stage, lvl_status = out.detail;
        const uint32_t not_null = out.not_null, size = out.n_levels;
        out.status = PQ_OK;
        out.stage = PQ_STAGE_NONE;
        out.detail = 0;
        out.n_decoded = 0;
        out.not_null = 0;
        out.n_levels = 0;
        out.consumed = 0;
        finish_page(out, r.init_failed ? r.status : PQ_OK, r.status, r.n_decoded, lvl_status, lvl_stage,
                    not_null, n, size);
    }
    results[page] = out;
}

// Data pages of flat columns (max_rep == 0, max_def == 0) that the host turned into the value streams
// they are: results[page] holds what the stream kernels left (status, count, stage == VALUES_INIT
// when the decoder's Init failed); ReadValues' outcome is made of it here (layout/page-v1.go:80-116:
// both level arrays are n constant zeros, every slot is a value).
__global__ __launch_bounds__(kWave) void lowered_finish_kernel(const pq_page_desc *__restrict__ descs,
                                                               const uint32_t *__restrict__ ids, uint32_t count,
                                                               pq_page_result *__restrict__ results)
{
    const uint32_t i = blockIdx.x * kWave + threadIdx.x;
    if (i >= count) return;
    const uint32_t page = ids[i];
    const pq_page_result r = results[page];
    const uint32_t n = descs[page].n_values;
    pq_page_result out;
    out.status = PQ_OK;
    out.stage = PQ_STAGE_NONE;
    out.detail = 0;
    out.n_decoded = 0;
    out.not_null = 0;
    out.n_levels = 0;
    out.consumed = 0;
    finish_page(out, r.stage == PQ_STAGE_VALUES_INIT ? r.status : PQ_OK, r.status, r.n_decoded, PQ_OK, PQ_STAGE_NONE, n, n,
                n);
    results[page] = out;
}

// Their constant levels, for the callers that want them written: blockIdx.x = page, the slots dealt
// out over blockIdx.y.
__global__ __launch_bounds__(kWG) void lowered_levels_kernel(const pq_page_desc *__restrict__ descs,
                                                             const uint32_t *__restrict__ ids, PageBuffers buf)
{
    const pq_page_desc &d = descs[ids[blockIdx.x]];
    if (d.flags & PQ_FLAG_NO_LEVEL_OUTPUT) return;
    uint32_t *rep = reinterpret_cast<uint32_t *>(buf.levels + d.rep_off);
    uint32_t *def = reinterpret_cast<uint32_t *>(buf.levels + d.def_off);
    const uint32_t n = d.n_values;
    for (uint32_t i = blockIdx.y * kWG + threadIdx.x; i < n; i += gridDim.y * kWG) {
        rep[i] = 0;
        def[i] = 0;
    }
}

hipError_t launch_lowered_finish(hipStream_t s, const pq_page_desc *d_descs, const uint32_t *d_ids, uint32_t count,
                                 PageBuffers buf, pq_page_result *d_results, bool write_levels)
{
    if (count == 0) return hipSuccess;
    if (write_levels)
        hipLaunchKernelGGL(lowered_levels_kernel, dim3(count, count >= 1024u ? 4u : 64u), dim3(kWG), 0, s, d_descs, d_ids,
                           buf);
    hipLaunchKernelGGL(lowered_finish_kernel, dim3((count + kWave - 1) / kWave), dim3(kWave), 0, s, d_descs, d_ids, count,
                       d_results);
    return hipGetLastError();
}

// Plans over several buffer sets: out[i] = in[i] with its offsets moved by how far the buffers of
// its set lie behind the first set's (deltas: 4 words per set: src, values, levels, dict; 64-bit
// wrap-around arithmetic), so that every kernel can go on adding offsets to one base pointer.
__global__ __launch_bounds__(256) void rebase_kernel(const pq_page_desc *__restrict__ in,
                                                     const uint32_t *__restrict__ set_of,
                                                     const uint64_t *__restrict__ deltas, uint32_t n,
                                                     pq_page_desc *__restrict__ out)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    pq_page_desc d = in[i];
    if (deltas == nullptr) {
        out[i] = d;
        return;
    }
    const uint64_t *w = deltas + 4u * (size_t)set_of[i];
    const bool page = d.kind == PQ_KIND_PAGE_V1 || d.kind == PQ_KIND_PAGE_V2;
    const uint32_t vk = page ? d.value_kind : d.kind;
    d.src_off += w[0];
    d.dst_off += w[1];
    d.rep_off += w[2];
    d.def_off += w[2];
    // byte arrays keep their offsets area in the values buffer
    d.dict_off += (vk == PQ_KIND_DELTA_LENGTH_BA || vk == PQ_KIND_DELTA_BA) ? w[1] : w[3];
    out[i] = d;
}

hipError_t launch_rebase(hipStream_t s, const pq_page_desc *d_in, const uint32_t *d_set_of, const uint64_t *d_deltas,
                         uint32_t n, pq_page_desc *d_out)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(rebase_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s, d_in, d_set_of, d_deltas, n, d_out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------

static size_t wave_lds_bytes(uint32_t dict_lds_bytes)
{
    return align16((uint32_t)sizeof(HybridWaveLds) * kHwWaves) + dict_lds_bytes;
}

// More than 64 KiB of dynamic LDS (a large shared dictionary) has to be asked for.
template <typename K>
static hipError_t allow_lds(K kernel, size_t bytes)
{
    if (bytes <= 65536) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)bytes);
}

hipError_t launch_stream_pages(hipStream_t s, const pq_page_desc *d_descs, const uint32_t *d_ids,
                               uint32_t count, PageBuffers buf, pq_page_result *d_results,
                               uint32_t dict_lds_bytes, const WaveLaunch &wl, const HsegPage *d_hseg_pages,
                               uint32_t n_hseg)
{
    if (count == 0) return hipSuccess;
    const uint32_t n_plain = count - n_hseg;
    if (count >= wl.min_pages) {
        const uint32_t grid = (count + kHwWaves - 1) / kHwWaves;
        const hipError_t e = allow_lds(stream_wave_kernel, wave_lds_bytes(dict_lds_bytes));
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(stream_wave_kernel, dim3(grid), dim3(kHwWG), wave_lds_bytes(dict_lds_bytes), s, d_descs,
                           d_ids, count, buf, d_results, dict_lds_bytes, d_hseg_pages, n_plain);
        return hipGetLastError();
    }
    const size_t shmem = align16(sizeof(HybridShared)) + dict_lds_bytes;
    hipLaunchKernelGGL(stream_pages_kernel, dim3(count), dim3(kWG), shmem, s, d_descs, d_ids, buf, d_results,
                       dict_lds_bytes, d_hseg_pages, n_plain);
    return hipGetLastError();
}

hipError_t launch_delta_pages(hipStream_t s, int bits, const pq_page_desc *d_descs, const uint32_t *d_ids,
                              uint32_t count, uint32_t sweep_count, PageBuffers buf, pq_page_result *d_results,
                              uint32_t wave_min_pages, uint32_t *d_flag, uint32_t epoch, bool *needs_serial)
{
    *needs_serial = false;
    if (sweep_count == 0) return hipSuccess;
    // Enough pages to fill the chip with one wave each (256 CUs x 8 waves and up): the
    // barrier-free wave-per-page kernel first, then a sweep by the workgroup-per-page
    // kernel over the pages whose layout is not 128 / 4 (normally none: it ends after
    // reading one flag word).  Few, large pages go to the workgroup-per-page kernel directly.
    if (count >= wave_min_pages || sweep_count > count) {
        const uint32_t grid = (count + kWG / kWave - 1) / (kWG / kWave);
        if (count != 0) {
            if (bits == 64)
                hipLaunchKernelGGL(delta_wave_kernel<uint64_t>, dim3(grid), dim3(kWG), 0, s, d_descs, d_ids, count,
                                   buf, d_results, d_flag, epoch);
            else
                hipLaunchKernelGGL(delta_wave_kernel<uint32_t>, dim3(grid), dim3(kWG), 0, s, d_descs, d_ids, count,
                                   buf, d_results, d_flag, epoch);
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        const uint32_t sweep = min(sweep_count, 256u);
        if (bits == 64)
            hipLaunchKernelGGL((delta_pages_kernel<uint64_t, true>), dim3(sweep), dim3(kWG), 0, s, d_descs, d_ids,
                               sweep_count, buf, d_results, d_flag, epoch);
        else
            hipLaunchKernelGGL((delta_pages_kernel<uint32_t, true>), dim3(sweep), dim3(kWG), 0, s, d_descs, d_ids,
                               sweep_count, buf, d_results, d_flag, epoch);
        return hipGetLastError();
    }
    *needs_serial = true;
    if (bits == 64)
        hipLaunchKernelGGL((delta_pages_kernel<uint64_t, false>), dim3(count), dim3(kWG), 0, s, d_descs, d_ids,
                           count, buf, d_results, d_flag, epoch);
    else
        hipLaunchKernelGGL((delta_pages_kernel<uint32_t, false>), dim3(count), dim3(kWG), 0, s, d_descs, d_ids,
                           count, buf, d_results, d_flag, epoch);
    return hipGetLastError();
}

hipError_t launch_delta_serial(hipStream_t s, const pq_page_desc *d_descs, const uint32_t *d_ids,
                               uint32_t count, PageBuffers buf, pq_page_result *d_results)
{
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(delta_serial_kernel, dim3((count + kWave - 1) / kWave), dim3(kWave), 0, s, d_descs, d_ids,
                       count, buf, d_results);
    return hipGetLastError();
}

hipError_t launch_data_pages(hipStream_t s, const pq_page_desc *d_descs, const uint32_t *d_ids,
                             uint32_t count, PageBuffers buf, pq_page_result *d_results,
                             uint32_t dict_lds_bytes, const WaveLaunch &wl, uint32_t *d_flag, uint32_t epoch)
{
    if (count == 0) return hipSuccess;
    const size_t shmem = align16(sizeof(PageShared)) + dict_lds_bytes;
    if (count >= wl.min_pages) {
        const uint32_t grid = (count + kHwWaves - 1) / kHwWaves;
        const hipError_t ea = allow_lds(data_wave_kernel, wave_lds_bytes(dict_lds_bytes));
        if (ea != hipSuccess) return ea;
        hipLaunchKernelGGL(data_wave_kernel, dim3(grid), dim3(kHwWG), wave_lds_bytes(dict_lds_bytes), s, d_descs,
                           d_ids, count, buf, d_results, dict_lds_bytes, d_flag, epoch);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(data_pages_kernel<true>, dim3(min(count, 256u)), dim3(kWG), shmem, s, d_descs, d_ids,
                           count, buf, d_results, dict_lds_bytes, d_flag, epoch);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(data_pages_kernel<false>, dim3(count), dim3(kWG), shmem, s, d_descs, d_ids, count, buf,
                       d_results, dict_lds_bytes, d_flag, epoch);
    return hipGetLastError();
}

} // namespace pq
