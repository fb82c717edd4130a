// nested.hip — large data pages of nested columns as three units of work.
//
// A V1 / V2 data page of a column with max_rep > 0 is its repetition levels, its definition levels
// and its values, in that order, read by one reader (dataPageReaderV1.read + ReadValues,
// layout/page-v1.go:40-116; dataPageReaderV2, layout/page-v2.go:42-149).  Where the three sections
// start is known before anything is decoded — the V1 length prefixes, the V2 header's byte lengths —
// and the only thing one section needs from another is the number of values: the count of
// definition levels equal to the maximum (layout/helpers.go:84-86).  So a plan with few, large
// nested pages runs each page as
//   1. nested_split_kernel   one lane per page: the sections' places, written into the descriptors
//                            of the page's three units (a level stream each for rep and def, and
//                            the page's own slot turned into a stream of its value encoding);
//                            errors of read() itself (a length prefix that is not there, V2 level
//                            bytes beyond the page) end the page here;
//   2. the level units through the hybrid-stream kernels, cut into segments like any large stream;
//   3. nested_values_kernel  one lane per page: how the levels ended, and not_null as the value
//                            unit's count;
//   4. the value units among the plan's other descriptors;
//   5. nested_finish_kernel  one lane per page: the page's result from its units', with the error
//                            precedence of the page reader (page_finish.hpp).
// One difference from the one-wave-per-page decode, in pages that fail: the definition levels are
// decoded even when the repetition levels failed, so the level areas of a failed page may hold more
// than the reference would have produced; results, counts and the outputs of good pages are the same.
#include "page_finish.hpp"
#include "pq_launch.hpp"

namespace pq {

__device__ __forceinline__ uint32_t np_le32(const uint8_t *p)
{
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

// InitSize on `avail` bytes at S (encoding/hybrid-decoder.go:57-68): bytes it takes from the page's
// reader, or a status.  w >= 1 here (max_rep > 0; a definition level width of 0 reads nothing).
__device__ __forceinline__ int32_t np_init_size(const uint8_t *S, int64_t avail, uint32_t w, uint32_t *taken)
{
    *taken = 0;
    if (w == 0) return PQ_OK;
    if (avail < 4) return avail <= 0 ? PQ_EOF : PQ_UNEXPECTED_EOF;
    const int64_t size = (int64_t)np_le32(S);
    *taken = 4u + (uint32_t)(size < avail - 4 ? size : avail - 4);
    return PQ_OK;
}

__global__ __launch_bounds__(kWave) void nested_split_kernel(pq_page_desc *__restrict__ descs,
                                                             NestedPage *__restrict__ pages, uint32_t n_pages,
                                                             PageBuffers buf)
{
    const uint32_t i = blockIdx.x * kWave + threadIdx.x;
    if (i >= n_pages) return;
    NestedPage np = pages[i];
    pq_page_desc v = descs[np.desc];      // the value unit, still holding the page's own place
    pq_page_desc r = descs[np.rep_unit];
    pq_page_desc d = descs[np.def_unit];
    const uint8_t *S = buf.src + v.src_off;
    const int64_t len = (int64_t)v.src_len;
    const uint32_t wr = bits_len16(np.max_rep), wd = bits_len16(np.max_def);
    int32_t early = PQ_OK, stage = PQ_STAGE_NONE;
    int64_t vpos = 0;
    if (np.version == 1u) {
        uint32_t rt = 0, dt = 0;
        early = np_init_size(S, len, wr, &rt);
        if (early != PQ_OK) {
            stage = PQ_STAGE_REP_INIT;
        } else {
            early = np_init_size(S + rt, len - rt, wd, &dt);
            if (early != PQ_OK) stage = PQ_STAGE_DEF_INIT;
        }
        // each level unit does its own InitSize on what is left of the page at its place
        r.src_off = v.src_off;
        r.src_len = (uint64_t)len;
        r.flags = PQ_FLAG_LENGTH_PREFIXED;
        d.src_off = v.src_off + rt;
        d.src_len = (uint64_t)(len - rt);
        d.flags = wd != 0u ? PQ_FLAG_LENGTH_PREFIXED : 0u;  // width 0: nothing is read, every level is 0
        vpos = (int64_t)rt + dt;
    } else {
        const int64_t lv = (int64_t)np.rep_len + (int64_t)np.def_len;
        if (lv > len) {  // io.ReadFull of the level bytes (page-v2.go:87-90)
            early = len <= 0 ? PQ_EOF : PQ_UNEXPECTED_EOF;
            stage = PQ_STAGE_REP_INIT;
        }
        // a zero-length section leaves its decoder without a reader (page-v2.go:92-102)
        r.src_off = v.src_off;
        r.src_len = np.rep_len;
        r.flags = np.rep_len == 0u ? PQ_FLAG_NO_INIT : 0u;
        d.src_off = v.src_off + np.rep_len;
        d.src_len = np.def_len;
        d.flags = np.def_len == 0u ? PQ_FLAG_NO_INIT : 0u;
        vpos = lv;
    }
    r.n_values = np.n_values;
    d.n_values = np.n_values;
    if (early != PQ_OK) {
        // read() failed: nothing of the page is decoded.  The units stay in the launches and do nothing.
        r.n_values = 0;
        r.flags = PQ_FLAG_LEVELS_CONST;
        d.n_values = 0;
        d.flags = PQ_FLAG_LEVELS_CONST;
        v.src_len = 0;
        vpos = 0;
    } else {
        v.src_off += (uint64_t)vpos;
        v.src_len = (uint64_t)(len - vpos);
    }
    v.n_values = 0;  // until the definition levels are counted
    descs[np.desc] = v;
    descs[np.rep_unit] = r;
    descs[np.def_unit] = d;
    np.early_status = early;
    np.early_stage = stage;
    np.lvl_status = PQ_OK;
    np.lvl_stage = PQ_STAGE_NONE;
    np.not_null = 0;
    pages[i] = np;
}

__global__ __launch_bounds__(kWave) void nested_values_kernel(pq_page_desc *__restrict__ descs,
                                                              NestedPage *__restrict__ pages, uint32_t n_pages,
                                                              const pq_page_result *__restrict__ results)
{
    const uint32_t i = blockIdx.x * kWave + threadIdx.x;
    if (i >= n_pages) return;
    NestedPage np = pages[i];
    if (np.early_status != PQ_OK) return;
    // ReadValues: repetition levels, then definition levels; the first failure ends it (page-v1.go:88-101)
    const pq_page_result rr = results[np.rep_unit], dr = results[np.def_unit];
    if (np.n_values != 0u) {
        if (rr.status != PQ_OK) {
            np.lvl_status = rr.status;
            np.lvl_stage = PQ_STAGE_REP;
        } else if (dr.status != PQ_OK) {
            np.lvl_status = dr.status;
            np.lvl_stage = PQ_STAGE_DEF;
        } else {
            np.not_null = dr.not_null;
        }
    }
    descs[np.desc].n_values = np.lvl_status == PQ_OK ? np.not_null : 0u;
    pages[i] = np;
}

__global__ __launch_bounds__(kWave) void nested_finish_kernel(const pq_page_desc *__restrict__ descs,
                                                              const NestedPage *__restrict__ pages, uint32_t n_pages,
                                                              pq_page_result *__restrict__ results)
{
    const uint32_t i = blockIdx.x * kWave + threadIdx.x;
    if (i >= n_pages) return;
    const NestedPage np = pages[i];
    pq_page_result out;
    out.status = PQ_OK;
    out.stage = PQ_STAGE_NONE;
    out.detail = 0;
    out.n_decoded = 0;
    out.not_null = 0;
    out.n_levels = 0;
    out.consumed = 0;
    if (np.early_status != PQ_OK) {
        out.status = np.early_status;
        out.stage = np.early_stage;
    } else {
        const pq_page_result vr = results[np.desc];
        const uint32_t nvals = descs[np.desc].n_values;
        finish_page(out, vr.stage == PQ_STAGE_VALUES_INIT ? vr.status : PQ_OK, vr.status, vr.n_decoded, np.lvl_status,
                    np.lvl_stage, np.not_null, nvals, np.n_values);
    }
    results[np.desc] = out;
}

hipError_t launch_nested_split(hipStream_t s, pq_page_desc *d_descs, NestedPage *d_pages, uint32_t n_pages, PageBuffers buf)
{
    if (n_pages == 0) return hipSuccess;
    hipLaunchKernelGGL(nested_split_kernel, dim3((n_pages + kWave - 1) / kWave), dim3(kWave), 0, s, d_descs, d_pages, n_pages,
                       buf);
    return hipGetLastError();
}

hipError_t launch_nested_values(hipStream_t s, pq_page_desc *d_descs, NestedPage *d_pages, uint32_t n_pages,
                                const pq_page_result *d_results)
{
    if (n_pages == 0) return hipSuccess;
    hipLaunchKernelGGL(nested_values_kernel, dim3((n_pages + kWave - 1) / kWave), dim3(kWave), 0, s, d_descs, d_pages,
                       n_pages, d_results);
    return hipGetLastError();
}

hipError_t launch_nested_finish(hipStream_t s, const pq_page_desc *d_descs, const NestedPage *d_pages, uint32_t n_pages,
                                pq_page_result *d_results)
{
    if (n_pages == 0) return hipSuccess;
    hipLaunchKernelGGL(nested_finish_kernel, dim3((n_pages + kWave - 1) / kWave), dim3(kWave), 0, s, d_descs, d_pages,
                       n_pages, d_results);
    return hipGetLastError();
}

}  // namespace pq
