// page_finish.hpp — how a data page's pieces (value decoder set-up, levels, values) become its
// one result record.  Shared by every kernel that decodes whole pages.
#pragma once
#include "pq_device.hpp"

namespace pq {

// Error precedence of read() followed by ReadValues (layout/page-v1.go:40-116).
__device__ __forceinline__ void finish_page(pq_page_result &out, int32_t v_init, int32_t v_status, uint32_t v_count,
                                   int32_t lvl_status, int32_t lvl_stage, uint32_t not_null, uint32_t nvals,
                                   uint32_t size)
{
    if (v_init != PQ_OK) { // read() failed: ReadValues never ran
        out.status = v_init;
        out.stage = PQ_STAGE_VALUES_INIT;
    } else if (lvl_status != PQ_OK) {
        out.status = lvl_status;
        out.stage = lvl_stage;
    } else {
        out.not_null = not_null;
        out.n_decoded = v_count;
        if (nvals != 0 && v_status != PQ_OK) { // any error, io.EOF included (page-v1.go:103-110)
            out.status = PQ_PAGE_VALUES_FAILED;
            out.detail = v_status;
            out.stage = PQ_STAGE_VALUES;
        } else {
            out.n_levels = size;
        }
    }
}

}  // namespace pq
