// bytearray.hip — DELTA_LENGTH_BYTE_ARRAY and DELTA_BYTE_ARRAY in kernels of their own.
//
// One wave per descriptor, four to a workgroup, no barrier.  A descriptor is either a bare value
// stream (PQ_KIND_DELTA_LENGTH_BA / PQ_KIND_DELTA_BA) or a data page of a flat column whose
// levels are constant and kept to itself (max_rep == 0, PQ_FLAG_NO_LEVEL_OUTPUT) with one of
// those as its value encoding.  The decoders are the ones of bytearray_wave.hpp, which every
// other page kernel uses too; what a kernel of their own buys is registers: without the hybrid
// and dictionary paths beside them the length streams run at twice the waves per SIMD, and the
// serial chain of a length stream (header, block, header, ...) is hidden by the other waves.
// One kernel per encoding: the front-coded one needs more registers than the plain lengths do.
//
// Reference: types/byte-array.go:166-231 (ByteArrayDeltaLengthDecoder), :279-348
// (ByteArrayDeltaDecoder); the page frame around them layout/page-v1.go:40-116,
// layout/page-v2.go:42-149.
#include "delta_wave.hpp"
#include "bytearray_wave.hpp"
#include "page_finish.hpp"
#include "pq_launch.hpp"

namespace pq {

// A wave's LDS: the delta decoder's stream window or the front-coded values' output and suffix
// windows, whichever is larger.
constexpr uint32_t kBaWaveLds = ((kBaLdsBytes > (uint32_t)kDwChunk + 16u ? kBaLdsBytes : (uint32_t)kDwChunk + 16u) + 15u) & ~15u;

template <uint32_t KIND>
__global__ __launch_bounds__(kWG) void byte_array_kernel(const pq_page_desc *__restrict__ descs,
                                                         const uint32_t *__restrict__ ids, uint32_t count,
                                                         PageBuffers buf, pq_page_result *__restrict__ results)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[kWG / kWave][kBaWaveLds / 4];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t slot = blockIdx.x * (kWG / kWave) + (uint32_t)wave;
    if (slot >= count) return;
    const uint32_t page = ids[slot];
    const pq_page_desc d = descs[page];
    const uint8_t *S = buf.src + d.src_off;
    const int64_t len = (int64_t)d.src_len;
    const bool is_page = d.kind == PQ_KIND_PAGE_V1 || d.kind == PQ_KIND_PAGE_V2;

    pq_page_result out;
    out.status = PQ_OK;
    out.stage = PQ_STAGE_NONE;
    out.detail = 0;
    out.n_decoded = 0;
    out.not_null = 0;
    out.n_levels = 0;
    out.consumed = 0;

    // a flat page: the value bytes start behind the V2 level bytes (V1 level bytes are not
    // skipped when max_rep == 0), every value is there when max_def == 0 and none otherwise
    int64_t vpos = 0;
    uint32_t nvals = d.n_values;
    if (is_page) {
        if (d.kind == PQ_KIND_PAGE_V2) {
            vpos = (int64_t)d.rep_len + (int64_t)d.def_len;
            if (vpos > len) {
                out.status = len <= 0 ? PQ_EOF : PQ_UNEXPECTED_EOF;
                out.stage = PQ_STAGE_REP_INIT;
                if (lane == 0) results[page] = out;
                return;
            }
        }
        nvals = d.max_def == 0 ? d.n_values : 0u;
    }

    uint32_t *area = reinterpret_cast<uint32_t *>(buf.values + d.dict_off);
    ByteArrayResult br;
    if constexpr (KIND == PQ_KIND_DELTA_BA)
        br = delta_ba_wave(lds[wave], S + vpos, len - vpos, nvals, d.dict_len, area, buf.values + d.dst_off,
                           (uint64_t)d.dst_cap, lane);
    else
        br = delta_length_ba_wave(lds[wave], S + vpos, len - vpos, nvals, d.dict_len, area, buf.values + d.dst_off,
                                  lane);
    if (is_page) {
        finish_page(out, br.init_failed ? br.status : PQ_OK, br.status, br.n_decoded, PQ_OK, PQ_STAGE_NONE, nvals, nvals,
                    d.n_values);
    } else {
        out.status = br.status;
        out.stage = br.init_failed ? PQ_STAGE_VALUES_INIT : PQ_STAGE_NONE;
        out.n_decoded = br.n_decoded;
        out.consumed = br.consumed;
    }
    if (lane == 0) results[page] = out;
}

hipError_t launch_byte_arrays(hipStream_t s, uint32_t kind, const pq_page_desc *d_descs, const uint32_t *d_ids,
                              uint32_t count, PageBuffers buf, pq_page_result *d_results)
{
    if (count == 0) return hipSuccess;
    const uint32_t grid = (count + kWG / kWave - 1) / (kWG / kWave);
    if (kind == PQ_KIND_DELTA_BA)
        hipLaunchKernelGGL(byte_array_kernel<PQ_KIND_DELTA_BA>, dim3(grid), dim3(kWG), 0, s, d_descs, d_ids, count, buf,
                           d_results);
    else
        hipLaunchKernelGGL(byte_array_kernel<PQ_KIND_DELTA_LENGTH_BA>, dim3(grid), dim3(kWG), 0, s, d_descs, d_ids, count,
                           buf, d_results);
    return hipGetLastError();
}

}  // namespace pq
