// capi.cpp — the C ABI of include/pqdecode.h: contexts, plans, staging, launches.
//
// Host-side only; all decoding happens in the gfx950 kernels of bitunpack.hip and
// pages.hip.  There is no CPU decode path in this library: without a HIP device
// every entry point fails with PQ_ERR_NO_DEVICE / PQ_ERR_HIP.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "pq_launch.hpp"
#include "pqdecode.h"

struct pq_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string last_error;
    // grow-only device staging used by the host-pointer entry points
    // measured on MI355X: the wave kernels are at least as fast from a single page on
    uint32_t delta_wave_min_pages = 1;
    uint32_t hybrid_wave_min_pages = 1;
    // delta pages are cut into segments when a plan has fewer than seg_max_pages of one element
    // size; seg_bytes != 0 fixes the segment length (else it follows the plan's size)
    uint32_t seg_max_pages = 1024;
    uint32_t seg_bytes = 0;
    void *stage[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t stage_cap[4] = {0, 0, 0, 0};
    // grow-only scratch of the device-pointer entry points that need one
    void *work = nullptr;
    size_t work_cap = 0;
    // plans: the device block of the last plan destroyed (the next one takes it if it is large
    // enough) and a pinned host image the plan tables are put together in
    void *spare_block = nullptr;
    size_t spare_bytes = 0;
    // two images, used in turn: a plan's copy may still be in flight when the next plan is made
    void *host_img[2] = {nullptr, nullptr};
    size_t host_img_cap[2] = {0, 0};
    hipEvent_t host_img_done[2] = {nullptr, nullptr};  // recorded behind the copy out of the image
    unsigned host_img_next = 0;
    // plans made from this context and not yet destroyed: pq_ctx_destroy detaches them
    std::vector<pq_plan *> live_plans;
    // pq_decode_chunks: pipeline slots, each a context of its own (stream, device staging) plus
    // pinned host staging for callers whose memory is pageable
    struct ChunkSlot *slots = nullptr;
    int n_slots = 3;  // slots pq_decode_chunks uses (PQ_OPT_CHUNK_SLOTS)
    uint32_t seg_spin_limit = pq::kSegSpinLimit;  // PQ_OPT_SEGMENT_SPIN_LIMIT
    // all copies to the device go through one stream and all copies back through another, so that
    // each direction's copies queue up behind one another without gaps
    hipStream_t chunk_in = nullptr, chunk_out = nullptr;
    // plans whose groups are large enough run them side by side: up to three streams next to the
    // context's own, forked from it and joined to it with events (PQ_OPT_CONCURRENT_GROUPS)
    bool concurrent_groups = true;
    bool run_tables = true;  // PQ_OPT_RUN_TABLES
    hipStream_t lane_stream[3] = {nullptr, nullptr, nullptr};
    hipEvent_t lane_fork = nullptr, lane_join[3] = {nullptr, nullptr, nullptr};
};

constexpr int kChunkSlotsMax = 8;
constexpr int kCopyThreads = 4;

struct ChunkSlot {
    pq_ctx *sub = nullptr;
    void *pin_in = nullptr, *pin_out = nullptr;   // src | dict, values | levels | results
    size_t pin_in_cap = 0, pin_out_cap = 0;
    hipEvent_t copied_in = nullptr, decoded = nullptr, done = nullptr;
    pq_plan *plan = nullptr;
    pq_chunk_job *job = nullptr;  // in flight
    size_t off_levels = 0, off_results = 0;
};

constexpr int kGroups = 7;  // kernel groups of a plan (the last: level units of split nested pages)

struct SetNeed {
    uint64_t src = 0, values = 0, levels = 0, dict = 0;
};

struct pq_plan {
    pq_ctx *ctx = nullptr;
    size_t n = 0;
    pq_page_desc *d_descs = nullptr;
    size_t block_bytes = 0;  // of the device block d_descs starts
    uint32_t *d_ids = nullptr;
    pq_page_result *d_results = nullptr;
    uint32_t *d_flags = nullptr; // [0], [32], [64]: sweep wake-up words (delta32, delta64, data pages)
    uint32_t epoch = 0;          // bumped every run; a flag word equal to it is "raised"
    // descriptor ids grouped by kernel:
    // [stream | delta32 | delta64 | data pages | DELTA_LENGTH_BYTE_ARRAY | DELTA_BYTE_ARRAY | level units]
    uint32_t count[kGroups] = {0, 0, 0, 0, 0, 0, 0};
    uint32_t first[kGroups] = {0, 0, 0, 0, 0, 0, 0};
    uint32_t dict_lds_bytes = 0;  // the largest dictionary that fits LDS: what a workgroup sets aside for its copy
    bool has_delta_page = false; // some PAGE_V1/V2 descriptor carries delta values
    // bytes in and out of the four lanes a run may work side by side: streams | delta pages | data
    // pages | byte arrays; `fork`: at least two of them are worth a stream of their own
    uint64_t lane_work[4] = {0, 0, 0, 0};
    bool fork = false;
    // delta32 / delta64 descriptors that are decoded in segments: the last n_seg_pages[k] ids of
    // group 1 + k, their page records, the segment -> page map and the segment records
    uint32_t n_seg_pages[2] = {0, 0};
    uint32_t n_segs[2] = {0, 0};
    pq::SegPage *d_seg_pages[2] = {nullptr, nullptr};
    uint32_t *d_seg_owner[2] = {nullptr, nullptr};
    pq::SegRec *d_seg_recs[2] = {nullptr, nullptr};
    pq::SegLook *d_seg_looks[2] = {nullptr, nullptr};
    // hybrid streams that may be decoded in segments: [0] the last hs[0].n_pages ids of group 0, [1] the
    // level units of split nested pages (group 6, all of them candidates)
    struct HsegDev {
        uint32_t n_pages = 0, n_segs = 0, n_checks = 0;
        pq::HsegPage *d_pages = nullptr;
        uint32_t *d_seg_owner = nullptr, *d_check_owner = nullptr, *d_counts = nullptr;
        pq::HsegRec *d_recs = nullptr;
        pq::HsegScan *d_scans = nullptr;  // two records per segment
        uint16_t *d_tabs = nullptr;       // run-header offsets the scans leave for the decode (HsegPage::tab_first)
        int fix_rounds = 2;               // 1: every stream is one of narrow values (levels, booleans)
    } hs[2];
    // large pages of nested columns run as three units each (nested.hip): the internal table holds
    // n_total = n + 2 * n_split descriptors and results, the caller sees the first n
    size_t n_total = 0;
    uint32_t n_split = 0;
    pq::NestedPage *d_npages = nullptr;
    // data pages of flat columns that are decoded as the value streams they hold (see lower_flat_page):
    // their ids, and whether any of them wants its constant levels written out
    uint32_t n_lowered = 0;
    uint32_t *d_lowered = nullptr;
    bool lowered_levels = false;
    // Plans over several buffer sets (pq_plan_create_sets): descriptor i belongs to set set_of[i]; its
    // offsets are relative to that set's buffers.  d_descs0 keeps the table as created; every run
    // writes d_descs from it with the offsets moved to the first set's buffers (d_deltas: per set,
    // how far its src / values / levels / dict buffers lie behind the first set's).
    size_t n_sets = 1;
    std::vector<SetNeed> need;         // per set: bytes of each buffer the descriptors reach
    pq_page_desc *d_descs0 = nullptr;
    uint32_t *d_set_of = nullptr;
    uint64_t *d_deltas = nullptr;
    uint64_t *h_deltas = nullptr;      // pinned, 4 words per set
    hipEvent_t deltas_sent = nullptr;  // recorded behind the copy out of h_deltas
};

namespace {

bool hip_ok(pq_ctx *ctx, hipError_t e, const char *what)
{
    if (e == hipSuccess) return true;
    if (ctx) ctx->last_error = std::string(what) + ": " + hipGetErrorString(e);
    return false;
}

#define PQ_HIP(ctx, call)                                  \
    do {                                                   \
        if (!hip_ok((ctx), (call), #call)) return PQ_ERR_HIP; \
    } while (0)

int kind_group(uint32_t kind)
{
    switch (kind) {
    case PQ_KIND_HYBRID:
    case PQ_KIND_DICT:
    case PQ_KIND_LEVELS:
    case PQ_KIND_PLAIN:
    case PQ_KIND_BOOL_PLAIN:
    case PQ_KIND_BOOL_RLE: return 0;
    case PQ_KIND_DELTA_LENGTH_BA: return 4;
    case PQ_KIND_DELTA_BA: return 5;
    case PQ_KIND_DELTA32: return 1;
    case PQ_KIND_DELTA64: return 2;
    case PQ_KIND_PAGE_V1:
    case PQ_KIND_PAGE_V2: return 3;
    default: return -1;
    }
}

// PQ_KIND_DELTA_LENGTH_BA / PQ_KIND_DELTA_BA when the descriptor (or its page's values) is one, else 0
uint32_t byte_array_kind(const pq_page_desc &d)
{
    const bool page = d.kind == PQ_KIND_PAGE_V1 || d.kind == PQ_KIND_PAGE_V2;
    const uint32_t k = page ? d.value_kind : d.kind;
    return k == PQ_KIND_DELTA_LENGTH_BA || k == PQ_KIND_DELTA_BA ? k : 0u;
}

// The kernel group of a descriptor.  Byte arrays have kernels of their own (bytearray.hip), also
// as the values of a flat column's page whose constant levels are not written out.
int group_of(const pq_page_desc &d)
{
    const int g = kind_group(d.kind);
    if (g == 3 && d.max_rep == 0 && (d.flags & PQ_FLAG_NO_LEVEL_OUTPUT) != 0)
        if (const uint32_t ba = byte_array_kind(d)) return kind_group(ba);
    return g;
}

// Bytes per output value of a descriptor (0 = it writes no fixed-width value output).
uint32_t value_size(const pq_page_desc &d)
{
    uint32_t k = d.kind;
    if (k == PQ_KIND_PAGE_V1 || k == PQ_KIND_PAGE_V2) k = d.value_kind;
    switch (k) {
    case PQ_KIND_HYBRID: return 4;
    case PQ_KIND_DICT: return d.elem_size;
    case PQ_KIND_DELTA32: return 4;
    case PQ_KIND_DELTA64: return 8;
    case PQ_KIND_PLAIN: return d.elem_size;
    case PQ_KIND_BOOL_PLAIN:
    case PQ_KIND_BOOL_RLE: return 4;
    default: return 0;
    }
}

// A data page of a flat column is its value stream: with max_rep == 0 both level decoders are
// ConstDecoder(0) and read nothing (layout/chunk.go:111-124), the V1 block starts with the values
// (page-v1.go:69-77 initialises the constant decoders on it without consuming a byte), the V2
// values start behind the level bytes the header announces (page-v2.go:81-110), and with
// max_def == 0 every slot is a value (layout/helpers.go:84-86).  Such a page is decoded by the
// kernels of its value encoding — which cut large pages into segments — and a last small kernel
// turns the stream's result into the page's (page-v1.go:80-116).  Pages that decode no value
// (max_def != 0: the constant level 0 never equals it), empty pages and V2 pages whose level bytes
// are not all there stay with the page kernels.
constexpr uint64_t kSegValues = 8192;        // values a segment of a run-dense hybrid stream aims at
constexpr uint64_t kHsegTableMax = 2ull << 30;  // bytes of run-header offsets a plan keeps for its cut hybrid streams at most
constexpr uint32_t kNestedSplitMin = 65536;  // level slots from which a nested page is worth three units
constexpr uint64_t kForkMinBytes = 16ull << 20;  // a lane of a plan with less to move than this is not worth a fork
constexpr uint32_t kFlagLowered = 0x80000000u;  // internal: never set by callers (validate refuses it)

bool lower_flat_page(pq_page_desc &d)
{
    if (d.kind != PQ_KIND_PAGE_V1 && d.kind != PQ_KIND_PAGE_V2) return false;
    if (d.max_rep != 0 || d.max_def != 0 || d.n_values == 0) return false;
    switch (d.value_kind) {
    case PQ_KIND_DICT:
    case PQ_KIND_DELTA32:
    case PQ_KIND_DELTA64:
    case PQ_KIND_PLAIN:
    case PQ_KIND_BOOL_PLAIN:
    case PQ_KIND_BOOL_RLE: break;
    default: return false;  // byte arrays have page kernels of their own
    }
    const uint64_t vpos = d.kind == PQ_KIND_PAGE_V2 ? (uint64_t)d.rep_len + d.def_len : 0u;
    if (vpos > d.src_len) return false;
    d.kind = d.value_kind;
    d.src_off += vpos;
    d.src_len -= vpos;
    d.flags = (d.flags & PQ_FLAG_NO_LEVEL_OUTPUT) | kFlagLowered;
    return true;
}

int validate(const pq_page_desc &d)
{
    if (d.flags & ~(uint32_t)(PQ_FLAG_LENGTH_PREFIXED | PQ_FLAG_LEVELS_CONST | PQ_FLAG_NO_INIT | PQ_FLAG_NO_LEVEL_OUTPUT))
        return PQ_ERR_BAD_ARGUMENT;
    if (kind_group(d.kind) < 0) return PQ_ERR_BAD_ARGUMENT;
    if (d.n_values > 0x7fffffffu) return PQ_ERR_BAD_ARGUMENT; // counts are int32 in the reference
    if (d.src_len > 0x7fffffffffffull) return PQ_ERR_BAD_ARGUMENT;
    const bool page = d.kind == PQ_KIND_PAGE_V1 || d.kind == PQ_KIND_PAGE_V2;
    if (page && d.value_kind != PQ_KIND_DICT && d.value_kind != PQ_KIND_DELTA32 &&
        d.value_kind != PQ_KIND_DELTA64 && d.value_kind != PQ_KIND_PLAIN && d.value_kind != PQ_KIND_BOOL_PLAIN &&
        d.value_kind != PQ_KIND_BOOL_RLE && d.value_kind != PQ_KIND_DELTA_LENGTH_BA &&
        d.value_kind != PQ_KIND_DELTA_BA)
        return PQ_ERR_BAD_ARGUMENT;
    if (byte_array_kind(d) != 0) {
        if ((d.dict_off & 3u) || d.dict_len < d.n_values || d.dict_len > 0x3fffffffu || d.src_len > 0xffffffffull)
            return PQ_ERR_BAD_ARGUMENT;
    }
    const bool plain = d.kind == PQ_KIND_PLAIN || (page && d.value_kind == PQ_KIND_PLAIN);
    if (plain && d.elem_size != 4 && d.elem_size != 8) return PQ_ERR_BAD_ARGUMENT;
    const bool dict = d.kind == PQ_KIND_DICT || (page && d.value_kind == PQ_KIND_DICT);
    if (dict) {
        if (d.elem_size != 4 && d.elem_size != 8) return PQ_ERR_BAD_ARGUMENT;
        if (d.dict_off % d.elem_size) return PQ_ERR_BAD_ARGUMENT;
        if (d.dict_len > 0x7fffffffu) return PQ_ERR_BAD_ARGUMENT;
    }
    // NewHybridDecoder indexes a 33-entry table with the width (hybrid-decoder.go:34)
    if (d.kind == PQ_KIND_HYBRID && d.bit_width > 32) return PQ_ERR_BAD_ARGUMENT;
    if (d.max_rep > 0xffffu || d.max_def > 0xffffu) return PQ_ERR_BAD_ARGUMENT; // levels are uint16
    const uint32_t vs = value_size(d);
    if (vs && d.dst_off % vs) return PQ_ERR_BAD_ARGUMENT;
    if ((page || d.kind == PQ_KIND_LEVELS) && ((d.rep_off | d.def_off) & 3u)) return PQ_ERR_BAD_ARGUMENT;
    return PQ_OK;
}

int ensure_stage(pq_ctx *ctx, int slot, size_t bytes)
{
    if (bytes <= ctx->stage_cap[slot]) return PQ_OK;
    if (ctx->stage[slot]) {
        PQ_HIP(ctx, hipFree(ctx->stage[slot]));
        ctx->stage[slot] = nullptr;
        ctx->stage_cap[slot] = 0;
    }
    const size_t cap = std::max<size_t>(bytes + bytes / 4, 1u << 20);
    PQ_HIP(ctx, hipMalloc(&ctx->stage[slot], cap));
    ctx->stage_cap[slot] = cap;
    return PQ_OK;
}

} // namespace

extern "C" {

int pq_version(void) { return 101; }

const char *pq_status_string(int s)
{
    switch (s) {
    case PQ_OK: return "ok";
    case PQ_EOF: return "EOF";
    case PQ_UNEXPECTED_EOF: return "unexpected EOF";
    case PQ_EMPTY_BP_RUN: return "rle: empty bit-packed run";
    case PQ_EMPTY_RLE_RUN: return "rle: empty RLE run";
    case PQ_RLE_VALUE_TOO_LARGE: return "rle: RLE run value is too large";
    case PQ_INT32_OUT_OF_RANGE: return "int32 out of range";
    case PQ_VARINT_OVERFLOW: return "binary: varint overflows a 64-bit integer";
    case PQ_INVALID_BLOCK_SIZE: return "invalid block size";
    case PQ_INVALID_MINIBLOCK_COUNT: return "invalid miniblock count";
    case PQ_INVALID_MINIBLOCK_WIDTH: return "invalid miniblock bit-width";
    case PQ_DICT_INVALID_WIDTH: return "invalid bit-width";
    case PQ_DICT_INVALID_INDEX: return "invalid index";
    case PQ_NOT_INITIALIZED: return "reader is not initialized";
    case PQ_INVALID_STREAM: return "invalid stream";
    case PQ_PAGE_VALUES_FAILED: return "read values from page failed";
    case PQ_BYTE_ARRAY_NEGATIVE_LEN: return "negative byte array length (the reference panics)";
    case PQ_BYTE_ARRAY_COUNT_MISMATCH: return "bytearray/delta: different number of suffixes and prefixes";
    case PQ_BYTE_ARRAY_INVALID_PREFIX: return "invalid prefix len in the stream";
    case PQ_WALK_HEADER_FAILED: return "failed to read page header";
    case PQ_WALK_SECOND_DICTIONARY: return "there should be only one dictionary";
    case PQ_WALK_PAGE_TYPE: return "page type not supported";
    case PQ_WALK_MISSING_HEADER: return "missing data page header";
    case PQ_WALK_NEGATIVE_NUM_VALUES: return "negative NumValues";
    case PQ_WALK_NEGATIVE_LEVELS_LENGTH: return "invalid levels byte length";
    case PQ_WALK_DICT_ENCODING: return "only Encoding_PLAIN and Encoding_PLAIN_DICTIONARY are supported for dict values encoder";
    case PQ_WALK_LEVEL_ENCODING: return "encoding not supported for definition and repetition level";
    case PQ_WALK_VALUE_ENCODING: return "encoding not supported for the column type";
    case PQ_WALK_BLOCK_SIZE: return "invalid size for compressed data";
    case PQ_WALK_UNCOMPRESSED_SIZE: return "invalid size for decompressed data";
    case PQ_WALK_SEEK: return "failed to set the read index to the start of the next page";
    case PQ_ERR_BAD_ARGUMENT: return "bad argument";
    case PQ_ERR_HIP: return "HIP runtime error";
    case PQ_ERR_NO_DEVICE: return "no HIP device";
    case PQ_ERR_OUT_OF_RANGE: return "descriptor outside the given buffers";
    default: return "unknown status";
    }
}

const char *pq_last_error(pq_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int pq_ctx_create(int device, void *hip_stream, pq_ctx **out)
{
    if (!out) return PQ_ERR_BAD_ARGUMENT;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return PQ_ERR_NO_DEVICE;
    if (device < 0 || device >= count) return PQ_ERR_BAD_ARGUMENT;
    pq_ctx *ctx = new (std::nothrow) pq_ctx;
    if (!ctx) return PQ_ERR_BAD_ARGUMENT;
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess) {
        delete ctx;
        return PQ_ERR_HIP;
    }
    if (hip_stream) {
        ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
            delete ctx;
            return PQ_ERR_HIP;
        }
        ctx->own_stream = true;
    }
    {
        // the static device workspace of the allocation-free entry point: its address on this
        // device is asked for here, once, so that the entry point itself only launches
        static std::mutex m;
        std::lock_guard<std::mutex> lock(m);
        hip_ok(ctx, pq::static_scratch_init(device), "hipGetSymbolAddress(scratch)");  // failure surfaces at the entry point
    }
    *out = ctx;
    return PQ_OK;
}

void pq_ctx_destroy(pq_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < 4; ++i)
        if (ctx->stage[i]) (void)hipFree(ctx->stage[i]);
    if (ctx->work) (void)hipFree(ctx->work);
    // plans that outlive their context lose their device block here; what is left of them is a
    // host record that pq_plan_destroy deletes and every other call refuses
    for (pq_plan *p : ctx->live_plans) {
        if (p->d_descs) (void)hipFree(p->d_descs);
        p->d_descs = nullptr;
        p->ctx = nullptr;
    }
    if (ctx->slots) {
        for (int k = 0; k < kChunkSlotsMax; ++k) {
            ChunkSlot &sl = ctx->slots[k];
            if (sl.plan) pq_plan_destroy(sl.plan);
            if (sl.sub) pq_ctx_destroy(sl.sub);
            if (sl.pin_in) (void)hipHostFree(sl.pin_in);
            if (sl.pin_out) (void)hipHostFree(sl.pin_out);
            if (sl.done) (void)hipEventDestroy(sl.done);
            if (sl.copied_in) (void)hipEventDestroy(sl.copied_in);
            if (sl.decoded) (void)hipEventDestroy(sl.decoded);
        }
        delete[] ctx->slots;
        if (ctx->chunk_in) (void)hipStreamDestroy(ctx->chunk_in);
        if (ctx->chunk_out) (void)hipStreamDestroy(ctx->chunk_out);
        (void)hipSetDevice(ctx->device);
    }
    if (ctx->spare_block) (void)hipFree(ctx->spare_block);
    for (int i = 0; i < 3; ++i) {
        if (ctx->lane_stream[i]) {
            (void)hipStreamSynchronize(ctx->lane_stream[i]);
            (void)hipStreamDestroy(ctx->lane_stream[i]);
        }
        if (ctx->lane_join[i]) (void)hipEventDestroy(ctx->lane_join[i]);
    }
    if (ctx->lane_fork) (void)hipEventDestroy(ctx->lane_fork);
    for (int i = 0; i < 2; ++i) {
        if (ctx->host_img[i]) (void)hipHostFree(ctx->host_img[i]);
        if (ctx->host_img_done[i]) (void)hipEventDestroy(ctx->host_img_done[i]);
    }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int pq_ctx_synchronize(pq_ctx *ctx)
{
    if (!ctx) return PQ_ERR_BAD_ARGUMENT;
    PQ_HIP(ctx, hipSetDevice(ctx->device));
    PQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PQ_OK;
}

int pq_ctx_set_option(pq_ctx *ctx, int option, int64_t value)
{
    if (!ctx) return PQ_ERR_BAD_ARGUMENT;
    switch (option) {
    case PQ_OPT_DELTA_WAVE_MIN_PAGES:
        if (value < 0) return PQ_ERR_BAD_ARGUMENT;
        ctx->delta_wave_min_pages = value > 0xffffffffll ? 0xffffffffu : (uint32_t)value;
        return PQ_OK;
    case PQ_OPT_HYBRID_WAVE_MIN_PAGES:
        if (value < 0) return PQ_ERR_BAD_ARGUMENT;
        ctx->hybrid_wave_min_pages = value > 0xffffffffll ? 0xffffffffu : (uint32_t)value;
        return PQ_OK;
    case PQ_OPT_SEGMENT_MAX_PAGES:
        if (value < 0) return PQ_ERR_BAD_ARGUMENT;
        ctx->seg_max_pages = value > 0xffffffffll ? 0xffffffffu : (uint32_t)value;
        return PQ_OK;
    case PQ_OPT_CHUNK_SLOTS:
        if (value < 1 || value > kChunkSlotsMax) return PQ_ERR_BAD_ARGUMENT;
        ctx->n_slots = (int)value;
        return PQ_OK;
    case PQ_OPT_SEGMENT_BYTES:
        // a segment holds at least one block header (a 256-value block is at most 1038 bytes)
        if (value != 0 && (value < 2048 || value > (1ll << 24) || (value & 15) != 0)) return PQ_ERR_BAD_ARGUMENT;
        ctx->seg_bytes = (uint32_t)value;
        return PQ_OK;
    case PQ_OPT_CONCURRENT_GROUPS:
        if (value != 0 && value != 1) return PQ_ERR_BAD_ARGUMENT;
        ctx->concurrent_groups = value != 0;
        return PQ_OK;
    case PQ_OPT_RUN_TABLES:
        if (value != 0 && value != 1) return PQ_ERR_BAD_ARGUMENT;
        ctx->run_tables = value != 0;
        return PQ_OK;
    case PQ_OPT_SEGMENT_SPIN_LIMIT:
        if (value < 0) return PQ_ERR_BAD_ARGUMENT;
        ctx->seg_spin_limit = value > 0xffffffffll ? 0xffffffffu : (uint32_t)value;
        return PQ_OK;
    default: return PQ_ERR_BAD_ARGUMENT;
    }
}

// ---- bit-unpack -----------------------------------------------------------

static int bitunpack_dev(pq_ctx *ctx, const void *d_src, size_t src_len, int w, size_t n_groups,
                         void *d_dst, int bits)
{
    if (!ctx || w < 0 || w > bits) return PQ_ERR_BAD_ARGUMENT;
    if (n_groups == 0) return PQ_OK;
    if (!d_dst || (w > 0 && !d_src)) return PQ_ERR_BAD_ARGUMENT;
    if ((uint64_t)n_groups * (uint64_t)w > (uint64_t)src_len) return PQ_ERR_OUT_OF_RANGE;
    if (reinterpret_cast<uintptr_t>(d_dst) % (bits / 8)) return PQ_ERR_BAD_ARGUMENT;
    PQ_HIP(ctx, hipSetDevice(ctx->device));
    if (bits == 32)
        PQ_HIP(ctx, pq::launch_bitunpack32(ctx->stream, d_src, src_len, w, n_groups, (int32_t *)d_dst));
    else
        PQ_HIP(ctx, pq::launch_bitunpack64(ctx->stream, d_src, src_len, w, n_groups, (int64_t *)d_dst));
    return PQ_OK;
}

int pq_bitunpack32_dev(pq_ctx *ctx, const void *d_src, size_t src_len, int w, size_t n_groups,
                       int32_t *d_dst)
{
    return bitunpack_dev(ctx, d_src, src_len, w, n_groups, d_dst, 32);
}

int pq_bitunpack64_dev(pq_ctx *ctx, const void *d_src, size_t src_len, int w, size_t n_groups,
                       int64_t *d_dst)
{
    return bitunpack_dev(ctx, d_src, src_len, w, n_groups, d_dst, 64);
}

static int bitunpack_host(pq_ctx *ctx, const uint8_t *src, size_t src_len, int w, size_t n_groups,
                          void *dst, int bits)
{
    if (!ctx || w < 0 || w > bits) return PQ_ERR_BAD_ARGUMENT;
    if (n_groups == 0) return PQ_OK;
    if (!dst || (w > 0 && !src)) return PQ_ERR_BAD_ARGUMENT;
    const size_t in_bytes = n_groups * (size_t)w;
    if (in_bytes > src_len) return PQ_ERR_OUT_OF_RANGE;
    const size_t out_bytes = n_groups * 8u * (size_t)(bits / 8);
    PQ_HIP(ctx, hipSetDevice(ctx->device));
    int st = ensure_stage(ctx, 0, in_bytes + 16);
    if (st == PQ_OK) st = ensure_stage(ctx, 1, out_bytes);
    if (st != PQ_OK) return st;
    if (in_bytes)
        PQ_HIP(ctx, hipMemcpyAsync(ctx->stage[0], src, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    st = bitunpack_dev(ctx, ctx->stage[0], in_bytes, w, n_groups, ctx->stage[1], bits);
    if (st != PQ_OK) return st;
    PQ_HIP(ctx, hipMemcpyAsync(dst, ctx->stage[1], out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    PQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PQ_OK;
}

int pq_bitunpack32(pq_ctx *ctx, const uint8_t *src, size_t src_len, int w, size_t n_groups, int32_t *dst)
{
    return bitunpack_host(ctx, src, src_len, w, n_groups, dst, 32);
}

int pq_bitunpack64(pq_ctx *ctx, const uint8_t *src, size_t src_len, int w, size_t n_groups, int64_t *dst)
{
    return bitunpack_host(ctx, src, src_len, w, n_groups, dst, 64);
}

// ---- plans ------------------------------------------------------------------

static int plan_create(pq_ctx *ctx, const pq_page_desc *user_descs, size_t n_pages, const uint32_t *first_desc,
                       size_t n_sets, pq_plan **out)
{
    if (!ctx || !out || (n_pages && !user_descs) || n_pages > 0x7fffffffu || n_sets == 0 || n_sets > 0xffffffu)
        return PQ_ERR_BAD_ARGUMENT;
    *out = nullptr;
    if (first_desc) {  // set k owns descriptors [first_desc[k], first_desc[k + 1])
        if (first_desc[0] != 0 || first_desc[n_sets] != n_pages) return PQ_ERR_BAD_ARGUMENT;
        for (size_t k = 0; k < n_sets; ++k)
            if (first_desc[k] > first_desc[k + 1]) return PQ_ERR_BAD_ARGUMENT;
    } else if (n_sets != 1) {
        return PQ_ERR_BAD_ARGUMENT;
    }
    std::vector<uint32_t> ids[kGroups];
    pq_plan *plan = new (std::nothrow) pq_plan;
    if (!plan) return PQ_ERR_BAD_ARGUMENT;
    plan->ctx = ctx;
    plan->n = n_pages;
    plan->n_sets = n_sets;
    plan->need.resize(n_sets);
    uint64_t dict_bytes_max = 0;
    // the table the kernels see: the caller's, with the pages of flat columns turned into the value
    // streams they are
    std::vector<pq_page_desc> descs(user_descs, user_descs + n_pages);
    std::vector<uint32_t> lowered, set_of(n_sets > 1 ? n_pages : 0);
    // Turning a flat page into its value stream pays where the stream's group is small enough to
    // have its large pages cut into segments; in a plan of thousands of pages everything a page
    // kernel can take stays in that one launch, where streams that are bound by memory and streams
    // that are bound by instruction issue run side by side.
    size_t in_group[kGroups] = {0};
    for (size_t i = 0; i < n_pages; ++i) {
        pq_page_desc t = user_descs[i];
        if (kind_group(t.kind) < 0) continue;  // refused below
        (void)lower_flat_page(t);
        in_group[group_of(t)] += 1;
    }
    size_t set = 0;
    for (size_t i = 0; i < n_pages; ++i) {
        const pq_page_desc &d = user_descs[i];
        const int st = validate(d);
        if (st != PQ_OK) {
            delete plan;
            return st;
        }
        while (first_desc && i >= first_desc[set + 1]) ++set;
        if (n_sets > 1) set_of[i] = (uint32_t)set;
        SetNeed &need = plan->need[set];
        need.src = std::max(need.src, d.src_off + d.src_len);
        const uint32_t vs = value_size(d);
        if (vs) need.values = std::max(need.values, d.dst_off + (uint64_t)d.n_values * vs);
        const bool page = d.kind == PQ_KIND_PAGE_V1 || d.kind == PQ_KIND_PAGE_V2;
        if (const uint32_t ba = byte_array_kind(d)) {
            const bool front = ba == PQ_KIND_DELTA_BA;  // values rebuilt from prefixes: caller-sized
            need.values = std::max(need.values, d.dst_off + (front ? (uint64_t)d.dst_cap : d.src_len));
            need.values = std::max(need.values, d.dict_off + (uint64_t)4 * ((uint64_t)(front ? 2 : 1) * d.dict_len + 1));
        }
        // a page of a flat column that keeps its constant-zero levels to itself needs no level room
        const bool no_levels = page && d.max_rep == 0 && (d.flags & PQ_FLAG_NO_LEVEL_OUTPUT) != 0;
        if ((page && !no_levels) || d.kind == PQ_KIND_LEVELS)
            need.levels = std::max(need.levels, d.def_off + (uint64_t)4 * d.n_values);
        if (page && !no_levels) need.levels = std::max(need.levels, d.rep_off + (uint64_t)4 * d.n_values);
        if (d.kind == PQ_KIND_DICT || (page && d.value_kind == PQ_KIND_DICT)) {
            const uint64_t db = (uint64_t)d.dict_len * d.elem_size;
            need.dict = std::max(need.dict, d.dict_off + db);
            if (db <= pq::kDictLdsMax) dict_bytes_max = std::max(dict_bytes_max, db);
        }
        {
            pq_page_desc t = descs[i];
            if (lower_flat_page(t) && in_group[group_of(t)] < ctx->seg_max_pages) {
                descs[i] = t;
                lowered.push_back((uint32_t)i);
                if (!no_levels) plan->lowered_levels = true;
            }
        }
        const pq_page_desc &t = descs[i];
        ids[group_of(t)].push_back((uint32_t)i);
        {
            static const int lane_of[kGroups] = {0, 1, 1, 2, 3, 3, 0};
            const uint32_t out = vs ? vs : 4u;
            plan->lane_work[lane_of[group_of(t)]] += t.src_len + (uint64_t)t.n_values * out;
        }
        if ((t.kind == PQ_KIND_PAGE_V1 || t.kind == PQ_KIND_PAGE_V2) &&
            (t.value_kind == PQ_KIND_DELTA32 || t.value_kind == PQ_KIND_DELTA64))
            plan->has_delta_page = true;
    }
    plan->n_lowered = (uint32_t)lowered.size();
    {
        int big = 0;
        for (uint64_t w : plan->lane_work) big += w >= kForkMinBytes ? 1 : 0;
        plan->fork = big >= 2;
    }
    plan->dict_lds_bytes = (uint32_t)((dict_bytes_max + 15u) & ~15ull);

    // Few large pages of nested columns: each becomes three units of work — its repetition levels,
    // its definition levels (two level streams appended to the table) and its values (the page's own
    // slot, turned into a stream of the value encoding) — whose places and counts are filled in on
    // the device at every run (nested.hip).
    std::vector<pq::NestedPage> npages;
    std::vector<uint64_t> level_bound;  // per appended unit: an upper estimate of its stream's bytes
    if (!ids[3].empty() && ids[3].size() < ctx->seg_max_pages) {
        std::vector<uint32_t> keep;
        for (uint32_t i : ids[3]) {
            const pq_page_desc page = descs[i];
            if (page.max_rep == 0 || page.n_values < kNestedSplitMin || page.src_len >= 0x7fffffffull) {
                keep.push_back(i);
                continue;
            }
            pq::NestedPage np;
            std::memset(&np, 0, sizeof np);
            np.desc = i;
            np.rep_unit = (uint32_t)descs.size();
            np.def_unit = np.rep_unit + 1u;
            np.version = page.kind == PQ_KIND_PAGE_V1 ? 1u : 2u;
            np.rep_len = page.rep_len;
            np.def_len = page.def_len;
            np.max_rep = page.max_rep;
            np.max_def = page.max_def;
            np.n_values = page.n_values;
            npages.push_back(np);
            for (int k = 0; k < 2; ++k) {
                pq_page_desc u;
                std::memset(&u, 0, sizeof u);
                u.kind = PQ_KIND_LEVELS;
                u.src_off = page.src_off;
                u.src_len = page.src_len;
                u.n_values = page.n_values;
                u.max_def = k == 0 ? page.max_rep : page.max_def;
                u.def_off = k == 0 ? page.rep_off : page.def_off;
                uint32_t w = 0;  // bits.Len16 of the maximum level
                for (uint32_t m = u.max_def & 0xffffu; m != 0u; m >>= 1) ++w;
                // V2 headers say how long the level sections are; for V1 one and a half times what plain
                // bit-packing would take (a stream that turns out longer is finished by its tail segment)
                uint64_t bound = np.version == 2u ? (k == 0 ? page.rep_len : page.def_len)
                                                  : (uint64_t)page.n_values * w / 8u * 3u / 2u + 1024u;
                level_bound.push_back(std::min<uint64_t>(bound, page.src_len));
                ids[6].push_back((uint32_t)descs.size());
                if (n_sets > 1) set_of.push_back(set_of[i]);
                descs.push_back(u);
            }
            pq_page_desc &v = descs[i];  // the value unit: place and count are set on the device
            v.kind = page.value_kind;
            v.flags = 0;
            ids[group_of(v)].push_back(i);
        }
        ids[3].swap(keep);
    }
    plan->n_split = (uint32_t)npages.size();
    const size_t n_total = descs.size();
    plan->n_total = n_total;

    if (hipSetDevice(ctx->device) != hipSuccess) {
        delete plan;
        return PQ_ERR_HIP;
    }
    // Longest first: a launch ends with its slowest wave, so descriptors that take long (large
    // pages, byte arrays) are dispatched before the short ones that can fill the tail.  Work is
    // bucketed by half powers of two, and the order inside a bucket stays the caller's, so a
    // plan of like pages keeps its order (and its locality).
    auto work_bucket = [&](uint32_t i) -> uint32_t {
        const pq_page_desc &d = descs[i];
        uint64_t w = d.src_len + (uint64_t)d.n_values * std::max<uint32_t>(value_size(d), 4u) + 1;
        if (byte_array_kind(d)) w *= 4;
        const uint32_t b = 63u - (uint32_t)__builtin_clzll(w);
        return 2u * b + (b ? (uint32_t)((w >> (b - 1u)) & 1u) : 0u);
    };
    std::vector<uint8_t> bucket(n_total);
    for (size_t i = 0; i < n_total; ++i) bucket[i] = (uint8_t)work_bucket((uint32_t)i);
    for (int g = 0; g < kGroups; ++g) {
        bool mixed = false;
        for (uint32_t i : ids[g]) mixed = mixed || bucket[i] != bucket[ids[g][0]];
        if (mixed)
            std::stable_sort(ids[g].begin(), ids[g].end(), [&](uint32_t a, uint32_t b) { return bucket[a] > bucket[b]; });
    }
    // Few delta pages of one element size cannot fill the machine one wave a page: the large
    // ones among them are cut into segments by bytes (segdelta.hip) and go behind the others.
    std::vector<pq::SegPage> seg_pages[2];
    std::vector<uint32_t> seg_owner[2];
    for (int k = 0; k < 2; ++k) {
        std::vector<uint32_t> &g = ids[1 + k];
        if (g.empty() || g.size() >= ctx->seg_max_pages) continue;
        uint64_t bytes = 0;
        for (uint32_t i : g) bytes += descs[i].src_len;
        uint64_t L = ctx->seg_bytes;
        if (L == 0) {
            L = 8192;  // up to 16384 segments (two waves to each of the machine's wave slots), 8 to 64 KiB each
            while (L < 65536 && L * 16384 < bytes) L *= 2;
        }
        auto cut = [&](uint32_t i) {
            const pq_page_desc &d = descs[i];
            return d.src_len >= 3 * L && d.src_len < 0x7fffffffull && d.n_values >= 1024u;
        };
        std::stable_partition(g.begin(), g.end(), [&](uint32_t i) { return !cut(i); });
        for (uint32_t i : g) {
            if (!cut(i)) continue;
            pq::SegPage pg;
            std::memset(&pg, 0, sizeof pg);
            pg.desc = i;
            pg.seg_first = (uint32_t)seg_owner[k].size();
            pg.n_segs = (uint32_t)((descs[i].src_len + L - 1) / L);
            pg.seg_bytes = (uint32_t)L;
            seg_owner[k].insert(seg_owner[k].end(), pg.n_segs, (uint32_t)seg_pages[k].size());
            seg_pages[k].push_back(pg);
        }
        plan->n_seg_pages[k] = (uint32_t)seg_pages[k].size();
        plan->n_segs[k] = (uint32_t)seg_owner[k].size();
    }
    // The same for hybrid streams (seghybrid.hip): dictionary-index pages, level streams, BOOLEAN RLE
    // values and raw streams of group 0, and the level units of split nested pages.
    struct HsegHost {
        std::vector<pq::HsegPage> pages;
        std::vector<uint32_t> seg_owner, check_owner;
        uint64_t tab_entries = 0;  // 16-bit entries of the run-header table, a multiple of 8 per page
        bool narrow = true;        // every cut stream holds values of at most 3 bits (as far as the descriptor tells)
    } hsh[2];
    auto build_hseg = [&](std::vector<uint32_t> &g, HsegHost &out, bool units) {
        if (g.empty() || (!units && g.size() >= ctx->seg_max_pages)) return;
        uint64_t bytes = 0;
        for (uint32_t i : g) bytes += descs[i].src_len;
        uint64_t L = ctx->seg_bytes;
        if (L == 0) {
            L = 8192;
            while (L < 65536 && L * 16384 < bytes) L *= 2;
        }
        // what the stream is expected to hold: the descriptor's bytes, for a level unit its estimate
        auto stream_bytes = [&](uint32_t i) -> uint64_t {
            return units ? level_bound[i - n_pages] : descs[i].src_len;
        };
        // A run-dense stream (levels, run-heavy keys) holds many values per byte, and the work of a
        // segment goes by its values: such a page gets shorter segments, about kSegValues values each.
        auto seg_len = [&](uint32_t i) -> uint64_t {
            const pq_page_desc &d = descs[i];
            if (ctx->seg_bytes != 0 || d.n_values == 0) return L;
            const uint64_t by_values = (kSegValues * stream_bytes(i) / d.n_values + 15u) & ~15ull;
            return std::min<uint64_t>(L, std::max<uint64_t>(by_values, 1024u));
        };
        auto cut = [&](uint32_t i) {
            if (units) return true;  // all of them: their places are only known on the device
            const pq_page_desc &d = descs[i];
            // every kind that is a hybrid stream with a reader behind it (hseg_header_kernel looks closer)
            const bool hybrid = d.kind == PQ_KIND_DICT || d.kind == PQ_KIND_BOOL_RLE ||
                                ((d.kind == PQ_KIND_HYBRID || d.kind == PQ_KIND_LEVELS) &&
                                 (d.flags & (PQ_FLAG_NO_INIT | PQ_FLAG_LEVELS_CONST)) == 0u);
            return hybrid && d.src_len >= 3 * seg_len(i) && d.src_len < 0x7fffffffull && d.n_values >= 4096u;
        };
        std::stable_partition(g.begin(), g.end(), [&](uint32_t i) { return !cut(i); });
        for (uint32_t i : g) {
            if (!cut(i)) continue;
            const uint64_t Lp = seg_len(i), sb = stream_bytes(i);
            pq::HsegPage pg;
            std::memset(&pg, 0, sizeof pg);
            pg.desc = i;
            pg.seg_first = (uint32_t)out.seg_owner.size();
            pg.n_segs = (uint32_t)(2 * ((sb + Lp - 1) / Lp) + 2);
            pg.seg_bytes = (uint32_t)Lp;
            pg.check_first = (uint32_t)out.check_owner.size();
            pg.n_checks = units ? 0u : (uint32_t)((sb + 4095) / 4096);  // level streams are never chains of equal runs
            // the offsets of a segment's run headers, for the decode behind the scan: two bytes of table
            // per two bytes of stream the segments can hold
            pg.tab_first = (uint32_t)(out.tab_entries / 8u);
            out.tab_entries += ((uint64_t)pg.n_segs * (pg.seg_bytes / 2u) + 7u) & ~7ull;
            {
                const pq_page_desc &d = descs[i];
                const bool few_bits = d.kind == PQ_KIND_BOOL_RLE || (d.kind == PQ_KIND_LEVELS && d.max_def < 8u) ||
                                      (d.kind == PQ_KIND_HYBRID && d.bit_width <= 3u);
                out.narrow = out.narrow && few_bits;
            }
            out.seg_owner.insert(out.seg_owner.end(), pg.n_segs, (uint32_t)out.pages.size());
            out.check_owner.insert(out.check_owner.end(), pg.n_checks, (uint32_t)out.pages.size());
            out.pages.push_back(pg);
        }
    };
    build_hseg(ids[0], hsh[0], false);
    build_hseg(ids[6], hsh[1], true);
    for (HsegHost &h : hsh)
        if (!ctx->run_tables || h.tab_entries * sizeof(uint16_t) > kHsegTableMax) {  // none wanted, or too much to keep: the decode finds the headers itself
            h.tab_entries = 0;
            for (pq::HsegPage &pg : h.pages) pg.tab_first = 0xffffffffu;
        }
    for (int k = 0; k < 2; ++k) {
        plan->hs[k].n_pages = (uint32_t)hsh[k].pages.size();
        plan->hs[k].n_segs = (uint32_t)hsh[k].seg_owner.size();
        plan->hs[k].n_checks = (uint32_t)hsh[k].check_owner.size();
        plan->hs[k].fix_rounds = hsh[k].narrow ? 1 : 2;
    }
    std::vector<uint32_t> all;
    all.reserve(n_total);
    for (int g = 0; g < kGroups; ++g) {
        plan->first[g] = (uint32_t)all.size();
        plan->count[g] = (uint32_t)ids[g].size();
        all.insert(all.end(), ids[g].begin(), ids[g].end());
    }
    if (n_pages) {
        // One device block.  In front what the host fills — descriptors | ids | wake-up words
        // (uint32 [0], [32], [64], each on a 128-byte line of its own) | segment page records and
        // segment -> page maps | ids of the lowered pages — put together in a pinned host image
        // and sent in one copy; behind it what the device writes: results | segment records.
        struct Region {
            size_t at, bytes;
        };
        size_t img = 0;
        auto up = [](size_t x) { return (x + 255) / 256 * 256; };
        auto take = [&](size_t bytes) {
            Region r{img, bytes};
            img += up(bytes);
            return r;
        };
        constexpr size_t kFlagBytes = 512;
        const Region r_descs = take(n_total * sizeof(pq_page_desc)), r_ids = take(n_total * sizeof(uint32_t));
        const Region r_flags = take(kFlagBytes);
        Region r_sp[2], r_so[2];
        for (int k = 0; k < 2; ++k) {
            r_sp[k] = take(seg_pages[k].size() * sizeof(pq::SegPage));
            r_so[k] = take(seg_owner[k].size() * sizeof(uint32_t));
        }
        Region r_hp[2], r_hs[2], r_hc[2];
        for (int k = 0; k < 2; ++k) {
            r_hp[k] = take(hsh[k].pages.size() * sizeof(pq::HsegPage));
            r_hs[k] = take(hsh[k].seg_owner.size() * sizeof(uint32_t));
            r_hc[k] = take(hsh[k].check_owner.size() * sizeof(uint32_t));
        }
        const Region r_low = take(lowered.size() * sizeof(uint32_t));
        const Region r_set = take(set_of.size() * sizeof(uint32_t));
        const Region r_np = take(npages.size() * sizeof(pq::NestedPage));
        // several buffer sets, or descriptors the device rewrites: the table as created stays here,
        // every run derives r_descs from it
        const bool keep_table = n_sets > 1 || !npages.empty();
        const Region r_descs0 = take(keep_table ? n_total * sizeof(pq_page_desc) : 0);
        const size_t img_bytes = img;
        const Region r_deltas = take(n_sets > 1 ? n_sets * 4 * sizeof(uint64_t) : 0);
        const Region r_res = take(n_total * sizeof(pq_page_result));
        Region r_sr[2];
        for (int k = 0; k < 2; ++k) r_sr[k] = take(seg_owner[k].size() * sizeof(pq::SegRec));
        Region r_sl[2];
        for (int k = 0; k < 2; ++k) r_sl[k] = take(seg_owner[k].size() * sizeof(pq::SegLook));
        Region r_hr[2], r_hn[2], r_hx[2], r_ht[2];
        for (int k = 0; k < 2; ++k) {
            r_hr[k] = take(hsh[k].seg_owner.size() * sizeof(pq::HsegRec));
            r_hn[k] = take(hsh[k].check_owner.size() * sizeof(uint32_t));
            r_hx[k] = take(2 * hsh[k].seg_owner.size() * sizeof(pq::HsegScan));
            r_ht[k] = take(hsh[k].tab_entries * sizeof(uint16_t));
        }
        const size_t total = img;
        uint8_t *block = nullptr;
        hipError_t e = hipSuccess;
        if (ctx->spare_block && ctx->spare_bytes >= total) {
            block = static_cast<uint8_t *>(ctx->spare_block);
            plan->block_bytes = ctx->spare_bytes;
            ctx->spare_block = nullptr;
            ctx->spare_bytes = 0;
        } else {
            e = hipMalloc((void **)&block, total);
            plan->block_bytes = total;
        }
        const unsigned slot = ctx->host_img_next++ & 1u;
        if (e == hipSuccess && !ctx->host_img_done[slot])
            e = hipEventCreateWithFlags(&ctx->host_img_done[slot], hipEventDisableTiming);
        // the copy of the plan before last came out of this image: it has to be over
        if (e == hipSuccess) e = hipEventSynchronize(ctx->host_img_done[slot]);
        if (e == hipSuccess && ctx->host_img_cap[slot] < img_bytes) {
            if (ctx->host_img[slot]) (void)hipHostFree(ctx->host_img[slot]);
            ctx->host_img[slot] = nullptr;
            ctx->host_img_cap[slot] = 0;
            const size_t cap = img_bytes + img_bytes / 2;
            e = hipHostMalloc(&ctx->host_img[slot], cap, hipHostMallocDefault);
            if (e == hipSuccess) ctx->host_img_cap[slot] = cap;
        }
        if (e == hipSuccess) {
            uint8_t *h = static_cast<uint8_t *>(ctx->host_img[slot]);
            auto put = [&](const Region &r, const void *p) {
                if (r.bytes) std::memcpy(h + r.at, p, r.bytes);
            };
            plan->d_descs = reinterpret_cast<pq_page_desc *>(block + r_descs.at);
            plan->d_ids = reinterpret_cast<uint32_t *>(block + r_ids.at);
            plan->d_flags = reinterpret_cast<uint32_t *>(block + r_flags.at);
            plan->d_results = reinterpret_cast<pq_page_result *>(block + r_res.at);
            put(r_descs, descs.data());
            put(r_ids, all.data());
            std::memset(h + r_flags.at, 0, kFlagBytes);
            for (int k = 0; k < 2; ++k) {
                plan->d_seg_pages[k] = reinterpret_cast<pq::SegPage *>(block + r_sp[k].at);
                plan->d_seg_owner[k] = reinterpret_cast<uint32_t *>(block + r_so[k].at);
                plan->d_seg_recs[k] = reinterpret_cast<pq::SegRec *>(block + r_sr[k].at);
                plan->d_seg_looks[k] = reinterpret_cast<pq::SegLook *>(block + r_sl[k].at);
                put(r_sp[k], seg_pages[k].data());
                put(r_so[k], seg_owner[k].data());
            }
            for (int k = 0; k < 2; ++k) {
                pq_plan::HsegDev &hd = plan->hs[k];
                hd.d_pages = reinterpret_cast<pq::HsegPage *>(block + r_hp[k].at);
                hd.d_seg_owner = reinterpret_cast<uint32_t *>(block + r_hs[k].at);
                hd.d_check_owner = reinterpret_cast<uint32_t *>(block + r_hc[k].at);
                hd.d_recs = reinterpret_cast<pq::HsegRec *>(block + r_hr[k].at);
                hd.d_counts = reinterpret_cast<uint32_t *>(block + r_hn[k].at);
                hd.d_scans = reinterpret_cast<pq::HsegScan *>(block + r_hx[k].at);
                hd.d_tabs = r_ht[k].bytes ? reinterpret_cast<uint16_t *>(block + r_ht[k].at) : nullptr;
                put(r_hp[k], hsh[k].pages.data());
                put(r_hs[k], hsh[k].seg_owner.data());
                put(r_hc[k], hsh[k].check_owner.data());
            }
            plan->d_lowered = reinterpret_cast<uint32_t *>(block + r_low.at);
            put(r_low, lowered.data());
            plan->d_npages = reinterpret_cast<pq::NestedPage *>(block + r_np.at);
            put(r_np, npages.data());
            if (keep_table) {
                plan->d_descs0 = reinterpret_cast<pq_page_desc *>(block + r_descs0.at);
                put(r_descs0, descs.data());
            }
            if (n_sets > 1) {
                plan->d_set_of = reinterpret_cast<uint32_t *>(block + r_set.at);
                plan->d_deltas = reinterpret_cast<uint64_t *>(block + r_deltas.at);
                put(r_set, set_of.data());
            }
            e = hipMemcpyAsync(block, h, img_bytes, hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) e = hipEventRecord(ctx->host_img_done[slot], ctx->stream);
        } else if (block) {
            plan->d_descs = reinterpret_cast<pq_page_desc *>(block);  // for pq_plan_destroy
        }
        // nothing waits here: the tables are on their way on the stream the plan runs on
        if (e == hipSuccess) e = hipMemsetAsync(plan->d_results, 0xff, n_total * sizeof(pq_page_result), ctx->stream);
        if (e == hipSuccess && n_sets > 1) {
            e = hipHostMalloc((void **)&plan->h_deltas, n_sets * 4 * sizeof(uint64_t), hipHostMallocDefault);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&plan->deltas_sent, hipEventDisableTiming);
        }
        if (e != hipSuccess) {
            hip_ok(ctx, e, "pq_plan_create");
            pq_plan_destroy(plan);
            return PQ_ERR_HIP;
        }
    }
    ctx->live_plans.push_back(plan);
    *out = plan;
    return PQ_OK;
}

int pq_plan_create(pq_ctx *ctx, const pq_page_desc *descs, size_t n_pages, pq_plan **out)
{
    return plan_create(ctx, descs, n_pages, nullptr, 1, out);
}

int pq_plan_create_sets(pq_ctx *ctx, const pq_page_desc *descs, size_t n_pages, const uint32_t *first_desc,
                        size_t n_sets, pq_plan **out)
{
    if (!first_desc) return PQ_ERR_BAD_ARGUMENT;
    return plan_create(ctx, descs, n_pages, first_desc, n_sets, out);
}

void pq_plan_destroy(pq_plan *plan)
{
    if (!plan) return;
    if (plan->h_deltas) (void)hipHostFree(plan->h_deltas);
    if (plan->deltas_sent) (void)hipEventDestroy(plan->deltas_sent);
    plan->h_deltas = nullptr;
    plan->deltas_sent = nullptr;
    if (!plan->ctx) {  // its context went first and took the device block with it
        delete plan;
        return;
    }
    {
        std::vector<pq_plan *> &lp = plan->ctx->live_plans;
        lp.erase(std::remove(lp.begin(), lp.end(), plan), lp.end());
    }
    (void)hipSetDevice(plan->ctx->device);
    (void)hipStreamSynchronize(plan->ctx->stream);
    if (plan->d_descs) {  // one block holds ids, results and flags too
        pq_ctx *ctx = plan->ctx;
        if (plan->block_bytes > ctx->spare_bytes) {  // kept for the next plan; the smaller one goes
            if (ctx->spare_block) (void)hipFree(ctx->spare_block);
            ctx->spare_block = plan->d_descs;
            ctx->spare_bytes = plan->block_bytes;
        } else {
            (void)hipFree(plan->d_descs);
        }
    }
    delete plan;
}

// Every kernel of a run, on the context's stream: buf holds the buffers the table's offsets refer to.
static int plan_launch(pq_plan *plan, pq::PageBuffers buf)
{
    pq_ctx *ctx = plan->ctx;
    const pq::WaveLaunch wl{ctx->hybrid_wave_min_pages};
    auto seg_hybrid = [&](const pq_plan::HsegDev &hd) {
        return pq::launch_seg_hybrid(ctx->stream, plan->d_descs, hd.d_pages, hd.n_pages, hd.d_check_owner, hd.n_checks,
                                     hd.d_counts, hd.d_seg_owner, hd.n_segs, hd.d_recs, hd.d_scans, buf, plan->d_results,
                                     &hd == &plan->hs[1] ? 0u : plan->dict_lds_bytes,  // level units read no dictionary
                                     hd.d_tabs, hd.fix_rounds);
    };
    if (plan->n_split != 0) {
        // large nested pages: where their three sections lie, then the level units; the values run
        // below among everything else, once their counts are known
        PQ_HIP(ctx, pq::launch_nested_split(ctx->stream, plan->d_descs, plan->d_npages, plan->n_split, buf));
        PQ_HIP(ctx, seg_hybrid(plan->hs[1]));
        PQ_HIP(ctx, pq::launch_stream_pages(ctx->stream, plan->d_descs, plan->d_ids + plan->first[6], plan->count[6], buf,
                                            plan->d_results, 0, wl, plan->hs[1].d_pages, plan->hs[1].n_pages));
        PQ_HIP(ctx, pq::launch_nested_values(ctx->stream, plan->d_descs, plan->d_npages, plan->n_split, plan->d_results));
    }
    // The four lanes — streams, delta pages, data pages, byte arrays — share nothing but the
    // descriptor table they read.  A plan with two lanes and more that are worth it forks them off
    // the context's stream and joins them again: kernels bound by memory and kernels bound by
    // instruction issue then run side by side instead of each waiting for the other's last wave.
    // Not under stream capture (what a capture makes of side streams is the caller's design).
    bool fork = plan->fork && ctx->concurrent_groups;
    if (fork) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        PQ_HIP(ctx, hipStreamIsCapturing(ctx->stream, &cs));
        fork = cs == hipStreamCaptureStatusNone;
    }
    hipStream_t lane[4] = {ctx->stream, ctx->stream, ctx->stream, ctx->stream};
    if (fork) {
        if (!ctx->lane_fork) PQ_HIP(ctx, hipEventCreateWithFlags(&ctx->lane_fork, hipEventDisableTiming));
        PQ_HIP(ctx, hipEventRecord(ctx->lane_fork, ctx->stream));
        int used = 0;
        for (int k = 0; k < 4; ++k) {
            if (plan->lane_work[k] == 0) continue;
            if (used++ == 0) continue;  // the first lane with work stays on the context's stream
            const int i = used - 2;
            if (!ctx->lane_stream[i]) {
                PQ_HIP(ctx, hipStreamCreateWithFlags(&ctx->lane_stream[i], hipStreamNonBlocking));
                PQ_HIP(ctx, hipEventCreateWithFlags(&ctx->lane_join[i], hipEventDisableTiming));
            }
            lane[k] = ctx->lane_stream[i];
            PQ_HIP(ctx, hipStreamWaitEvent(lane[k], ctx->lane_fork, 0));
        }
    }
    auto seg_hybrid0 = [&](hipStream_t st) {
        const pq_plan::HsegDev &hd = plan->hs[0];
        return pq::launch_seg_hybrid(st, plan->d_descs, hd.d_pages, hd.n_pages, hd.d_check_owner, hd.n_checks, hd.d_counts,
                                     hd.d_seg_owner, hd.n_segs, hd.d_recs, hd.d_scans, buf, plan->d_results,
                                     &hd == &plan->hs[1] ? 0u : plan->dict_lds_bytes,  // level units read no dictionary
                                     hd.d_tabs, hd.fix_rounds);
    };
    PQ_HIP(ctx, seg_hybrid0(lane[0]));
    PQ_HIP(ctx, pq::launch_stream_pages(lane[0], plan->d_descs, plan->d_ids + plan->first[0], plan->count[0], buf,
                                        plan->d_results, plan->dict_lds_bytes, wl, plan->hs[0].d_pages,
                                        plan->hs[0].n_pages));
    plan->epoch += 1;
    if (plan->epoch == 0) plan->epoch = 1; // 0 is what the flag words start as
    bool serial32 = false, serial64 = false;
    for (int k = 0; k < 2; ++k)  // the pages cut into segments first: they are the long ones
        PQ_HIP(ctx, pq::launch_seg_delta(lane[1], k ? 64 : 32, plan->d_descs, plan->d_seg_pages[k],
                                         plan->n_seg_pages[k], plan->d_seg_owner[k], plan->n_segs[k],
                                         plan->d_seg_recs[k], plan->d_seg_looks[k], buf, plan->d_results,
                                         plan->d_flags + 32 * k, plan->epoch, ctx->seg_spin_limit));
    PQ_HIP(ctx, pq::launch_delta_pages(lane[1], 32, plan->d_descs, plan->d_ids + plan->first[1],
                                       plan->count[1] - plan->n_seg_pages[0], plan->count[1], buf, plan->d_results,
                                       ctx->delta_wave_min_pages, plan->d_flags, plan->epoch, &serial32));
    PQ_HIP(ctx, pq::launch_delta_pages(lane[1], 64, plan->d_descs, plan->d_ids + plan->first[2],
                                       plan->count[2] - plan->n_seg_pages[1], plan->count[2], buf, plan->d_results,
                                       ctx->delta_wave_min_pages, plan->d_flags + 32, plan->epoch, &serial64));
    PQ_HIP(ctx, pq::launch_data_pages(lane[2], plan->d_descs, plan->d_ids + plan->first[3],
                                      plan->count[3], buf, plan->d_results, plan->dict_lds_bytes, wl,
                                      plan->d_flags + 64, plan->epoch));
    PQ_HIP(ctx, pq::launch_byte_arrays(lane[3], PQ_KIND_DELTA_LENGTH_BA, plan->d_descs, plan->d_ids + plan->first[4],
                                       plan->count[4], buf, plan->d_results));
    PQ_HIP(ctx, pq::launch_byte_arrays(lane[3], PQ_KIND_DELTA_BA, plan->d_descs, plan->d_ids + plan->first[5],
                                       plan->count[5], buf, plan->d_results));
    if (fork)
        for (int i = 0; i < 3; ++i) {
            bool in_use = false;
            for (int k = 0; k < 4; ++k) in_use = in_use || lane[k] == ctx->lane_stream[i];
            if (!in_use || !ctx->lane_stream[i]) continue;
            PQ_HIP(ctx, hipEventRecord(ctx->lane_join[i], ctx->lane_stream[i]));
            PQ_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->lane_join[i], 0));
        }
    // groups 1..3 are contiguous in d_ids; pages whose delta layout the tiled kernels
    // declined (and did not finish themselves) are completed one lane per page
    if (serial32 || serial64 || plan->has_delta_page)
        PQ_HIP(ctx, pq::launch_delta_serial(ctx->stream, plan->d_descs, plan->d_ids + plan->first[1],
                                            plan->count[1] + plan->count[2] + plan->count[3], buf,
                                            plan->d_results));
    // pages of flat columns that ran as their value streams: the stream's result becomes the page's
    PQ_HIP(ctx, pq::launch_lowered_finish(ctx->stream, plan->d_descs, plan->d_lowered, plan->n_lowered, buf,
                                          plan->d_results, plan->lowered_levels));
    PQ_HIP(ctx, pq::launch_nested_finish(ctx->stream, plan->d_descs, plan->d_npages, plan->n_split, plan->d_results));
    return PQ_OK;
}

// What a run checks of one buffer set before anything is launched.
static int check_set(const SetNeed &need, const void *d_src, size_t src_bytes, const void *d_values, size_t values_bytes,
                     const void *d_levels, size_t levels_bytes, const void *d_dict, size_t dict_bytes)
{
    if (need.src > src_bytes || need.values > values_bytes || need.levels > levels_bytes || need.dict > dict_bytes)
        return PQ_ERR_OUT_OF_RANGE;
    if ((need.src && !d_src) || (need.values && !d_values) || (need.levels && !d_levels) || (need.dict && !d_dict))
        return PQ_ERR_BAD_ARGUMENT;
    // value / level outputs are written with naturally aligned element stores
    if ((reinterpret_cast<uintptr_t>(d_values) & 7u) || (reinterpret_cast<uintptr_t>(d_levels) & 3u) ||
        (reinterpret_cast<uintptr_t>(d_dict) & 7u))
        return PQ_ERR_BAD_ARGUMENT;
    return PQ_OK;
}

int pq_plan_run(pq_plan *plan, const void *d_src, size_t src_bytes, void *d_values, size_t values_bytes,
                void *d_levels, size_t levels_bytes, const void *d_dict, size_t dict_bytes)
{
    if (!plan || !plan->ctx || plan->n_sets != 1) return PQ_ERR_BAD_ARGUMENT;  // no context: it was destroyed before the plan
    pq_ctx *ctx = plan->ctx;
    if (plan->n == 0) return PQ_OK;
    const int st = check_set(plan->need[0], d_src, src_bytes, d_values, values_bytes, d_levels, levels_bytes, d_dict,
                             dict_bytes);
    if (st != PQ_OK) return st;
    PQ_HIP(ctx, hipSetDevice(ctx->device));
    pq::PageBuffers buf{(const uint8_t *)d_src, (uint8_t *)d_values, (uint8_t *)d_levels,
                        (const uint8_t *)d_dict};
    if (plan->d_descs0)  // the last run rewrote descriptors: start from the table as created
        PQ_HIP(ctx, pq::launch_rebase(ctx->stream, plan->d_descs0, nullptr, nullptr, (uint32_t)plan->n_total, plan->d_descs));
    return plan_launch(plan, buf);
}

int pq_plan_run_sets(pq_plan *plan, const pq_buffer_set *sets, size_t n_sets)
{
    if (!plan || !plan->ctx || !sets || n_sets != plan->n_sets) return PQ_ERR_BAD_ARGUMENT;
    pq_ctx *ctx = plan->ctx;
    if (plan->n == 0) return PQ_OK;
    for (size_t k = 0; k < n_sets; ++k) {
        const pq_buffer_set &b = sets[k];
        const int st = check_set(plan->need[k], b.d_src, b.src_bytes, b.d_values, b.values_bytes, b.d_levels,
                                 b.levels_bytes, b.d_dict, b.dict_bytes);
        if (st != PQ_OK) return st;
    }
    PQ_HIP(ctx, hipSetDevice(ctx->device));
    pq::PageBuffers buf{(const uint8_t *)sets[0].d_src, (uint8_t *)sets[0].d_values, (uint8_t *)sets[0].d_levels,
                        (const uint8_t *)sets[0].d_dict};
    if (n_sets == 1) {
        if (plan->d_descs0)
            PQ_HIP(ctx, pq::launch_rebase(ctx->stream, plan->d_descs0, nullptr, nullptr, (uint32_t)plan->n_total, plan->d_descs));
        return plan_launch(plan, buf);
    }
    // Offsets are added to the first set's pointers in 64-bit arithmetic, so a set that lies
    // anywhere else is reached by moving its descriptors' offsets by the distance between the two
    // buffers (modulo 2^64).  The run before may still be reading the staging words.
    PQ_HIP(ctx, hipEventSynchronize(plan->deltas_sent));
    for (size_t k = 0; k < n_sets; ++k) {
        const pq_buffer_set &b = sets[k];
        uint64_t *w = plan->h_deltas + 4 * k;
        w[0] = (uint64_t)(reinterpret_cast<uintptr_t>(b.d_src) - reinterpret_cast<uintptr_t>(sets[0].d_src));
        w[1] = (uint64_t)(reinterpret_cast<uintptr_t>(b.d_values) - reinterpret_cast<uintptr_t>(sets[0].d_values));
        w[2] = (uint64_t)(reinterpret_cast<uintptr_t>(b.d_levels) - reinterpret_cast<uintptr_t>(sets[0].d_levels));
        w[3] = (uint64_t)(reinterpret_cast<uintptr_t>(b.d_dict) - reinterpret_cast<uintptr_t>(sets[0].d_dict));
    }
    PQ_HIP(ctx, hipMemcpyAsync(plan->d_deltas, plan->h_deltas, n_sets * 4 * sizeof(uint64_t), hipMemcpyHostToDevice,
                               ctx->stream));
    PQ_HIP(ctx, hipEventRecord(plan->deltas_sent, ctx->stream));
    PQ_HIP(ctx, pq::launch_rebase(ctx->stream, plan->d_descs0, plan->d_set_of, plan->d_deltas, (uint32_t)plan->n_total,
                                  plan->d_descs));
    return plan_launch(plan, buf);
}

int pq_plan_results(pq_plan *plan, pq_page_result *results)
{
    if (!plan || !plan->ctx || (plan->n && !results)) return PQ_ERR_BAD_ARGUMENT;
    pq_ctx *ctx = plan->ctx;
    PQ_HIP(ctx, hipSetDevice(ctx->device));
    PQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (plan->n)
        PQ_HIP(ctx, hipMemcpy(results, plan->d_results, plan->n * sizeof(pq_page_result),
                              hipMemcpyDeviceToHost));
    return PQ_OK;
}

static_assert(sizeof(pq_segment_info) == sizeof(pq::SegRec), "pq_segment_info is the device record");

int pq_plan_segments(pq_plan *plan, pq_segment_info *segs, size_t cap, size_t *n_segments)
{
    if (!plan || !plan->ctx || !n_segments || (cap && !segs)) return PQ_ERR_BAD_ARGUMENT;
    pq_ctx *ctx = plan->ctx;
    PQ_HIP(ctx, hipSetDevice(ctx->device));
    PQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_segments = (size_t)plan->n_segs[0] + plan->n_segs[1];
    size_t done = 0;
    for (int k = 0; k < 2 && done < cap; ++k) {
        const size_t take = std::min<size_t>(plan->n_segs[k], cap - done);
        if (take)
            PQ_HIP(ctx, hipMemcpy(segs + done, plan->d_seg_recs[k], take * sizeof(pq_segment_info), hipMemcpyDeviceToHost));
        done += take;
    }
    return PQ_OK;
}

// ---- host-memory batch --------------------------------------------------------

int pq_decode_pages(pq_ctx *ctx, const pq_page_desc *descs, size_t n_pages, const void *src,
                    size_t src_bytes, void *values, size_t values_bytes, void *levels,
                    size_t levels_bytes, const void *dict, size_t dict_bytes, pq_page_result *results)
{
    if (!ctx || (n_pages && (!descs || !results))) return PQ_ERR_BAD_ARGUMENT;
    if (n_pages == 0) return PQ_OK;
    if ((src_bytes && !src) || (values_bytes && !values) || (levels_bytes && !levels) ||
        (dict_bytes && !dict))
        return PQ_ERR_BAD_ARGUMENT;
    pq_plan *plan = nullptr;
    int st = pq_plan_create(ctx, descs, n_pages, &plan);
    if (st != PQ_OK) return st;
    auto done = [&](int code) {
        pq_plan_destroy(plan);
        return code;
    };
    const size_t sizes[4] = {src_bytes, values_bytes, levels_bytes, dict_bytes};
    for (int i = 0; i < 4; ++i)
        if ((st = ensure_stage(ctx, i, sizes[i] + 16)) != PQ_OK) return done(st);
    if (src_bytes && !hip_ok(ctx, hipMemcpyAsync(ctx->stage[0], src, src_bytes, hipMemcpyHostToDevice, ctx->stream), "H2D src"))
        return done(PQ_ERR_HIP);
    if (dict_bytes && !hip_ok(ctx, hipMemcpyAsync(ctx->stage[3], dict, dict_bytes, hipMemcpyHostToDevice, ctx->stream), "H2D dict"))
        return done(PQ_ERR_HIP);
    st = pq_plan_run(plan, ctx->stage[0], src_bytes, ctx->stage[1], values_bytes, ctx->stage[2], levels_bytes,
                     ctx->stage[3], dict_bytes);
    if (st != PQ_OK) return done(st);
    if (values_bytes && !hip_ok(ctx, hipMemcpyAsync(values, ctx->stage[1], values_bytes, hipMemcpyDeviceToHost, ctx->stream), "D2H values"))
        return done(PQ_ERR_HIP);
    if (levels_bytes && !hip_ok(ctx, hipMemcpyAsync(levels, ctx->stage[2], levels_bytes, hipMemcpyDeviceToHost, ctx->stream), "D2H levels"))
        return done(PQ_ERR_HIP);
    st = pq_plan_results(plan, results);
    return done(st);
}

// ---- column chunks from host memory, pipelined ---------------------------------------------

namespace {

// dst <- src with a few threads: one host core copies at about half the rate of a PCIe link
void par_copy(void *dst, const void *src, size_t n)
{
    if (n < (8u << 20)) {
        std::memcpy(dst, src, n);
        return;
    }
    std::thread th[kCopyThreads - 1];
    const size_t piece = (n / kCopyThreads + 63) & ~(size_t)63;
    for (int t = 0; t < kCopyThreads - 1; ++t) {
        const size_t lo = std::min(n, (size_t)(t + 1) * piece), hi = std::min(n, (size_t)(t + 2) * piece);
        th[t] = std::thread([=] { if (hi > lo) std::memcpy((char *)dst + lo, (const char *)src + lo, hi - lo); });
    }
    std::memcpy(dst, src, std::min(n, piece));
    for (auto &t : th) t.join();
}

int ensure_pinned(pq_ctx *ctx, void **p, size_t *cap, size_t bytes)
{
    if (bytes <= *cap) return PQ_OK;
    if (*p) {
        PQ_HIP(ctx, hipHostFree(*p));
        *p = nullptr;
        *cap = 0;
    }
    const size_t want = std::max<size_t>(bytes + bytes / 4, 1u << 20);
    PQ_HIP(ctx, hipHostMalloc(p, want, hipHostMallocDefault));
    *cap = want;
    return PQ_OK;
}

size_t up256(size_t x) { return (x + 255) / 256 * 256; }

// Waits for the slot's job and hands its outputs to the caller.
int finish_slot(pq_ctx *ctx, ChunkSlot &sl, bool pinned)
{
    if (!sl.job) return PQ_OK;
    pq_chunk_job &j = *sl.job;
    sl.job = nullptr;
    if (!hip_ok(ctx, hipEventSynchronize(sl.done), "hipEventSynchronize")) {
        // the job's outputs are not to be trusted; its plan goes all the same
        j.status = PQ_ERR_HIP;
        pq_plan_destroy(sl.plan);
        sl.plan = nullptr;
        return PQ_ERR_HIP;
    }
    if (!pinned) {
        const char *out = static_cast<const char *>(sl.pin_out);
        if (j.values_bytes) par_copy(j.values, out, j.values_bytes);
        if (j.levels_bytes) par_copy(j.levels, out + sl.off_levels, j.levels_bytes);
    }
    std::memcpy(j.results, static_cast<const char *>(sl.pin_out) + sl.off_results, j.n_pages * sizeof(pq_page_result));
    pq_plan_destroy(sl.plan);
    sl.plan = nullptr;
    return PQ_OK;
}

// The queued part of a job: copies in on the one stream, the decode on the slot's, copies out on the
// other.  The slot's device buffers are free: the job that used them last has been waited for.
// Returns a HIP failure; a job the run refuses (a descriptor points outside its buffers) has its
// status set and PQ_OK returned: the other jobs go on.
int queue_job(pq_ctx *ctx, ChunkSlot &sl, pq_chunk_job &j, bool pinned, size_t off_dict)
{
    pq_ctx *sub = sl.sub;
    const void *h_src = j.src, *h_dict = j.dict;
    if (!pinned) {
        char *in = static_cast<char *>(sl.pin_in);
        if (j.src_bytes) par_copy(in, j.src, j.src_bytes);
        if (j.dict_bytes) std::memcpy(in + off_dict, j.dict, j.dict_bytes);
        h_src = in;
        h_dict = in + off_dict;
    }
    if (j.src_bytes) PQ_HIP(ctx, hipMemcpyAsync(sub->stage[0], h_src, j.src_bytes, hipMemcpyHostToDevice, ctx->chunk_in));
    if (j.dict_bytes) PQ_HIP(ctx, hipMemcpyAsync(sub->stage[3], h_dict, j.dict_bytes, hipMemcpyHostToDevice, ctx->chunk_in));
    PQ_HIP(ctx, hipEventRecord(sl.copied_in, ctx->chunk_in));
    hipStream_t s = sub->stream;
    PQ_HIP(ctx, hipStreamWaitEvent(s, sl.copied_in, 0));
    const int st = pq_plan_run(sl.plan, sub->stage[0], j.src_bytes, sub->stage[1], j.values_bytes, sub->stage[2], j.levels_bytes,
                     sub->stage[3], j.dict_bytes);
    if (st != PQ_OK) {
        j.status = st;
        if (st == PQ_ERR_HIP) ctx->last_error = sub->last_error;
        return st == PQ_ERR_HIP ? st : PQ_OK;
    }
    PQ_HIP(ctx, hipEventRecord(sl.decoded, s));
    PQ_HIP(ctx, hipStreamWaitEvent(ctx->chunk_out, sl.decoded, 0));
    char *out = static_cast<char *>(sl.pin_out);
    void *h_values = pinned ? j.values : out, *h_levels = pinned ? j.levels : out + sl.off_levels;
    if (j.values_bytes)
        PQ_HIP(ctx, hipMemcpyAsync(h_values, sub->stage[1], j.values_bytes, hipMemcpyDeviceToHost, ctx->chunk_out));
    if (j.levels_bytes)
        PQ_HIP(ctx, hipMemcpyAsync(h_levels, sub->stage[2], j.levels_bytes, hipMemcpyDeviceToHost, ctx->chunk_out));
    PQ_HIP(ctx, hipMemcpyAsync(out + sl.off_results, sl.plan->d_results, j.n_pages * sizeof(pq_page_result),
                               hipMemcpyDeviceToHost, ctx->chunk_out));
    PQ_HIP(ctx, hipEventRecord(sl.done, ctx->chunk_out));
    return PQ_OK;
}

// Queues one job on the slot's stream: copies in, decode, copies out, an event behind them.
int start_slot(pq_ctx *ctx, ChunkSlot &sl, pq_chunk_job &j, bool pinned)
{
    pq_ctx *sub = sl.sub;
    j.status = PQ_OK;
    if (j.n_pages == 0) return PQ_OK;
    if (!j.descs || !j.results || (j.src_bytes && !j.src) || (j.values_bytes && !j.values) ||
        (j.levels_bytes && !j.levels) || (j.dict_bytes && !j.dict)) {
        j.status = PQ_ERR_BAD_ARGUMENT;
        return PQ_OK;
    }
    const size_t sizes[4] = {j.src_bytes, j.values_bytes, j.levels_bytes, j.dict_bytes};
    for (int i = 0; i < 4; ++i) {
        const int st = ensure_stage(sub, i, sizes[i] + 16);
        if (st != PQ_OK) {
            ctx->last_error = sub->last_error;
            j.status = st;
            return st;
        }
    }
    sl.off_levels = up256(j.values_bytes);
    sl.off_results = pinned ? 0 : sl.off_levels + up256(j.levels_bytes);
    int st = ensure_pinned(ctx, &sl.pin_out, &sl.pin_out_cap, sl.off_results + j.n_pages * sizeof(pq_page_result));
    const size_t off_dict = up256(j.src_bytes);
    if (st == PQ_OK && !pinned) st = ensure_pinned(ctx, &sl.pin_in, &sl.pin_in_cap, off_dict + j.dict_bytes);
    if (st != PQ_OK) {
        j.status = st;
        return st;
    }
    st = pq_plan_create(sub, j.descs, j.n_pages, &sl.plan);  // before anything is queued: it may refuse the job
    if (st != PQ_OK) {
        j.status = st;
        sl.plan = nullptr;
        if (st == PQ_ERR_HIP) ctx->last_error = sub->last_error;
        return st == PQ_ERR_HIP ? st : PQ_OK;
    }
    // from here on the slot owns a plan: whatever fails, it is destroyed before the slot is used again
    const int qs = queue_job(ctx, sl, j, pinned, off_dict);
    if (qs != PQ_OK || j.status != PQ_OK) {
        (void)hipStreamSynchronize(ctx->chunk_in);  // copies in may still read the slot's staging
        (void)hipStreamSynchronize(sub->stream);
        (void)hipStreamSynchronize(ctx->chunk_out);
        pq_plan_destroy(sl.plan);
        sl.plan = nullptr;
        if (qs != PQ_OK) j.status = qs;
        return qs;
    }
    sl.job = &j;
    return PQ_OK;
}

} // namespace

int pq_decode_chunks(pq_ctx *ctx, pq_chunk_job *jobs, size_t n_jobs, uint32_t flags)
{
    if (!ctx || (n_jobs && !jobs) || (flags & ~(uint32_t)PQ_CHUNKS_PINNED)) return PQ_ERR_BAD_ARGUMENT;
    if (n_jobs == 0) return PQ_OK;
    const bool pinned = (flags & PQ_CHUNKS_PINNED) != 0;
    PQ_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->slots) {
        ctx->slots = new (std::nothrow) ChunkSlot[kChunkSlotsMax];
        if (!ctx->slots) return PQ_ERR_BAD_ARGUMENT;
    }
    const size_t n_slots = (size_t)ctx->n_slots;
    for (size_t k = 0; k < n_slots; ++k) {
        ChunkSlot &sl = ctx->slots[k];
        if (!sl.sub) {
            const int st = pq_ctx_create(ctx->device, nullptr, &sl.sub);
            if (st != PQ_OK) return st;
        }
        sl.sub->delta_wave_min_pages = ctx->delta_wave_min_pages;
        sl.sub->hybrid_wave_min_pages = ctx->hybrid_wave_min_pages;
        sl.sub->seg_max_pages = ctx->seg_max_pages;
        sl.sub->seg_bytes = ctx->seg_bytes;
        sl.sub->seg_spin_limit = ctx->seg_spin_limit;
        sl.sub->run_tables = ctx->run_tables;
        if (!sl.done) PQ_HIP(ctx, hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
        if (!sl.copied_in) PQ_HIP(ctx, hipEventCreateWithFlags(&sl.copied_in, hipEventDisableTiming));
        if (!sl.decoded) PQ_HIP(ctx, hipEventCreateWithFlags(&sl.decoded, hipEventDisableTiming));
    }
    if (!ctx->chunk_in) PQ_HIP(ctx, hipStreamCreateWithFlags(&ctx->chunk_in, hipStreamNonBlocking));
    if (!ctx->chunk_out) PQ_HIP(ctx, hipStreamCreateWithFlags(&ctx->chunk_out, hipStreamNonBlocking));
    // a job that is never started (a HIP failure ends the loop) says so
    for (size_t i = 0; i < n_jobs; ++i) jobs[i].status = PQ_ERR_HIP;
    int rc = PQ_OK;
    for (size_t i = 0; i < n_jobs && rc == PQ_OK; ++i) {
        ChunkSlot &sl = ctx->slots[i % n_slots];
        rc = finish_slot(ctx, sl, pinned);        // the job that had the slot, n_slots jobs ago
        if (rc == PQ_OK) rc = start_slot(ctx, sl, jobs[i], pinned);
    }
    // drain, oldest first
    for (size_t d = 0; d < n_slots; ++d) {
        ChunkSlot &sl = ctx->slots[(n_jobs + d) % n_slots];
        const int st = finish_slot(ctx, sl, pinned);
        if (rc == PQ_OK) rc = st;
    }
    return rc;
}

void *pq_host_alloc(pq_ctx *ctx, size_t bytes)
{
    if (!ctx || bytes == 0) return nullptr;
    void *p = nullptr;
    if (hipSetDevice(ctx->device) != hipSuccess) return nullptr;
    if (!hip_ok(ctx, hipHostMalloc(&p, bytes, hipHostMallocDefault), "hipHostMalloc")) return nullptr;
    return p;
}

void pq_host_free(pq_ctx *ctx, void *p)
{
    if (!ctx || !p) return;
    (void)hipSetDevice(ctx->device);
    (void)hipHostFree(p);
}

// ---- single-stream conveniences -------------------------------------------------

static int one_page(pq_ctx *ctx, pq_page_desc &d, const void *src, size_t src_len, void *values,
                    size_t values_bytes, void *levels, size_t levels_bytes, const void *dict,
                    size_t dict_bytes, pq_page_result *r)
{
    d.src_off = 0;
    d.src_len = src_len;
    return pq_decode_pages(ctx, &d, 1, src, src_len, values, values_bytes, levels, levels_bytes, dict,
                           dict_bytes, r);
}

int pq_hybrid_decode(pq_ctx *ctx, const uint8_t *src, size_t src_len, int w, int length_prefixed,
                     uint32_t n, int32_t *dst, uint32_t *n_decoded, uint64_t *consumed)
{
    if (w < 0 || w > 32) return PQ_ERR_BAD_ARGUMENT;
    pq_page_desc d;
    std::memset(&d, 0, sizeof d);
    d.kind = PQ_KIND_HYBRID;
    d.flags = length_prefixed ? PQ_FLAG_LENGTH_PREFIXED : 0;
    d.bit_width = (uint32_t)w;
    d.n_values = n;
    pq_page_result r;
    const int st = one_page(ctx, d, src, src_len, dst, (size_t)n * 4, nullptr, 0, nullptr, 0, &r);
    if (st != PQ_OK) return st;
    if (n_decoded) *n_decoded = r.n_decoded;
    if (consumed) *consumed = r.consumed;
    return r.status;
}

int pq_dict_decode(pq_ctx *ctx, const uint8_t *src, size_t src_len, const void *dict, uint32_t dict_len,
                   uint32_t elem_size, uint32_t n, void *dst, uint32_t *n_decoded, uint64_t *consumed)
{
    pq_page_desc d;
    std::memset(&d, 0, sizeof d);
    d.kind = PQ_KIND_DICT;
    d.n_values = n;
    d.elem_size = elem_size;
    d.dict_len = dict_len;
    pq_page_result r;
    const int st = one_page(ctx, d, src, src_len, dst, (size_t)n * elem_size, nullptr, 0, dict,
                            (size_t)dict_len * elem_size, &r);
    if (st != PQ_OK) return st;
    if (n_decoded) *n_decoded = r.n_decoded;
    if (consumed) *consumed = r.consumed;
    return r.status;
}

int pq_levels_decode(pq_ctx *ctx, const uint8_t *src, size_t src_len, uint32_t max_level, int mode,
                     uint32_t n, int32_t *levels, uint32_t *not_null, uint64_t *consumed)
{
    if (mode < 0 || mode > 2) return PQ_ERR_BAD_ARGUMENT;
    pq_page_desc d;
    std::memset(&d, 0, sizeof d);
    d.kind = PQ_KIND_LEVELS;
    d.max_def = max_level;
    d.n_values = n;
    if (mode == 0) d.flags = PQ_FLAG_LEVELS_CONST;
    if (mode == 1) d.flags = PQ_FLAG_LENGTH_PREFIXED;
    if (mode == 2 && src_len == 0) d.flags = PQ_FLAG_NO_INIT;
    pq_page_result r;
    const int st = one_page(ctx, d, src, src_len, nullptr, 0, levels, (size_t)n * 4, nullptr, 0, &r);
    if (st != PQ_OK) return st;
    if (not_null) *not_null = r.not_null;
    if (consumed) *consumed = r.consumed;
    return r.status;
}

static int delta_host(pq_ctx *ctx, uint32_t kind, const uint8_t *src, size_t src_len, uint32_t n, void *dst,
                      size_t elem, uint32_t *n_decoded, uint64_t *consumed)
{
    pq_page_desc d;
    std::memset(&d, 0, sizeof d);
    d.kind = kind;
    d.n_values = n;
    pq_page_result r;
    const int st = one_page(ctx, d, src, src_len, dst, (size_t)n * elem, nullptr, 0, nullptr, 0, &r);
    if (st != PQ_OK) return st;
    if (n_decoded) *n_decoded = r.n_decoded;
    if (consumed) *consumed = r.consumed;
    return r.status;
}

int pq_plain_decode(pq_ctx *ctx, const uint8_t *src, size_t src_len, uint32_t elem_size, uint32_t n, void *dst,
                    uint32_t *n_decoded, uint64_t *consumed)
{
    pq_page_desc d;
    std::memset(&d, 0, sizeof d);
    d.kind = PQ_KIND_PLAIN;
    d.n_values = n;
    d.elem_size = elem_size;
    pq_page_result r;
    const int st = one_page(ctx, d, src, src_len, dst, (size_t)n * elem_size, nullptr, 0, nullptr, 0, &r);
    if (st != PQ_OK) return st;
    if (n_decoded) *n_decoded = r.n_decoded;
    if (consumed) *consumed = r.consumed;
    return r.status;
}

int pq_bool_plain_decode(pq_ctx *ctx, const uint8_t *src, size_t src_len, uint32_t n, int32_t *dst,
                         uint32_t *n_decoded, uint64_t *consumed)
{
    return delta_host(ctx, PQ_KIND_BOOL_PLAIN, src, src_len, n, dst, 4, n_decoded, consumed);
}

int pq_delta_decode32(pq_ctx *ctx, const uint8_t *src, size_t src_len, uint32_t n, int32_t *dst,
                      uint32_t *n_decoded, uint64_t *consumed)
{
    return delta_host(ctx, PQ_KIND_DELTA32, src, src_len, n, dst, 4, n_decoded, consumed);
}

int pq_delta_decode64(pq_ctx *ctx, const uint8_t *src, size_t src_len, uint32_t n, int64_t *dst,
                      uint32_t *n_decoded, uint64_t *consumed)
{
    return delta_host(ctx, PQ_KIND_DELTA64, src, src_len, n, dst, 8, n_decoded, consumed);
}

// The value count an INT32 DELTA_BINARY_PACKED header at src declares, clamped to what a stream of
// that size can possibly hold; *end = the offset behind the three header varints.
static uint64_t declared_lengths(const uint8_t *src, size_t src_len, size_t *end)
{
    uint64_t v[3] = {0, 0, 0};
    size_t pos = 0;
    for (int k = 0; k < 3; ++k) {
        int shift = 0;
        while (pos < src_len && shift < 64) {
            const uint8_t b = src[pos++];
            v[k] |= (uint64_t)(b & 0x7f) << shift;
            shift += 7;
            if (!(b & 0x80)) break;
        }
    }
    if (end) *end = pos;
    return std::min<uint64_t>(v[2], 32ull * src_len + 256);
}

static int byte_array_host(pq_ctx *ctx, uint32_t kind, const uint8_t *src, size_t src_len, uint32_t n,
                           uint32_t *offsets, uint8_t *data, size_t data_cap, uint32_t *n_decoded,
                           uint64_t *consumed, int *init_failed)
{
    if (!ctx || !offsets || (src_len && !src) || (data_cap && !data) || n > 0x3fffffffu ||
        src_len > 0xffffffffull || data_cap > 0xffffffffull)
        return PQ_ERR_BAD_ARGUMENT;
    // room for every length the stream declares.  The suffix lengths of a DELTA_BYTE_ARRAY stream
    // start somewhere behind the prefix lengths: the two counts have to agree for Init to succeed,
    // so the first count sizes both.
    uint32_t cap = (uint32_t)std::min<uint64_t>(0x3fffffffull, std::max<uint64_t>(n, declared_lengths(src, src_len, nullptr)));
    const bool front = kind == PQ_KIND_DELTA_BA;
    pq_page_desc d;
    pq_page_result r;
    std::vector<uint8_t> tmp;
    while (true) {
        std::memset(&d, 0, sizeof d);
        d.kind = kind;
        d.n_values = n;
        d.dst_off = 0;
        d.dst_cap = front ? (uint32_t)data_cap : 0;
        d.dict_off = (data_cap + 15) / 16 * 16;
        d.dict_len = cap;
        const size_t bytes = (size_t)d.dict_off + 4 * ((front ? 2 : 1) * (size_t)cap + 1);
        tmp.resize(bytes);
        const int st = one_page(ctx, d, src, src_len, tmp.data(), bytes, nullptr, 0, nullptr, 0, &r);
        if (st != PQ_OK) return st;
        // the suffix lengths may be more than the prefix lengths that sized the area: once more
        // with room for as many as a stream of this size can hold, so that Init ends the way the
        // reference's does (different number of suffixes and prefixes, or whatever comes first)
        const uint32_t most = (uint32_t)std::min<uint64_t>(0x3fffffffull, 32ull * src_len + 256);
        if (front && r.status == PQ_ERR_OUT_OF_RANGE && r.stage == PQ_STAGE_VALUES_INIT && cap < most) {
            cap = most;
            continue;
        }
        break;
    }
    const bool init_bad = r.stage == PQ_STAGE_VALUES_INIT;
    const uint32_t nd = init_bad ? 0 : r.n_decoded;
    offsets[0] = 0;
    if (!init_bad && r.status != PQ_ERR_OUT_OF_RANGE) {
        std::memcpy(offsets, tmp.data() + d.dict_off, 4 * ((size_t)nd + 1));
        if (offsets[nd]) std::memcpy(data, tmp.data(), offsets[nd]);
    }
    if (n_decoded) *n_decoded = nd;
    if (consumed) *consumed = r.consumed;
    if (init_failed) *init_failed = init_bad ? 1 : 0;
    return r.status;
}

int pq_delta_length_ba_decode(pq_ctx *ctx, const uint8_t *src, size_t src_len, uint32_t n,
                              uint32_t *offsets, uint8_t *data, uint32_t *n_decoded,
                              uint64_t *consumed, int *init_failed)
{
    return byte_array_host(ctx, PQ_KIND_DELTA_LENGTH_BA, src, src_len, n, offsets, data, src_len, n_decoded, consumed,
                           init_failed);
}

int pq_delta_ba_decode(pq_ctx *ctx, const uint8_t *src, size_t src_len, uint32_t n, uint32_t *offsets,
                       uint8_t *data, size_t data_cap, uint32_t *n_decoded, uint64_t *consumed,
                       int *init_failed)
{
    return byte_array_host(ctx, PQ_KIND_DELTA_BA, src, src_len, n, offsets, data, data_cap, n_decoded, consumed,
                           init_failed);
}

// ---- page-header walk -----------------------------------------------------------

static int ensure_work(pq_ctx *ctx, size_t bytes);

int pq_walk_chunks_dev(pq_ctx *ctx, const void *d_src, size_t src_bytes, const pq_chunk_desc *chunks,
                       size_t n_chunks, pq_page_info *d_infos, size_t infos_cap, pq_chunk_result *results)
{
    if (!ctx || (n_chunks && (!chunks || !results)) || n_chunks > 0x7fffffffu) return PQ_ERR_BAD_ARGUMENT;
    if (n_chunks == 0) return PQ_OK;
    for (size_t i = 0; i < n_chunks; ++i) {
        const pq_chunk_desc &c = chunks[i];
        // the kernel reads whole 16-byte lines around [src_off, src_off + src_len)
        if (c.src_off > src_bytes || c.src_len > src_bytes - c.src_off || c.src_len > 0x7fffffffffffull)
            return PQ_ERR_BAD_ARGUMENT;
        if ((uint64_t)c.info_first + c.info_cap > infos_cap) return PQ_ERR_BAD_ARGUMENT;
        if (c.src_len && !d_src) return PQ_ERR_BAD_ARGUMENT;
        if (c.info_cap && !d_infos) return PQ_ERR_BAD_ARGUMENT;
    }
    PQ_HIP(ctx, hipSetDevice(ctx->device));
    const size_t cbytes = n_chunks * sizeof(pq_chunk_desc), rbytes = n_chunks * sizeof(pq_chunk_result);
    const size_t roff = (cbytes + 15) / 16 * 16;
    const int st = ensure_work(ctx, roff + rbytes);
    if (st != PQ_OK) return st;
    uint8_t *w = static_cast<uint8_t *>(ctx->work);
    PQ_HIP(ctx, hipMemcpyAsync(w, chunks, cbytes, hipMemcpyHostToDevice, ctx->stream));
    PQ_HIP(ctx, pq::launch_walk_chunks(ctx->stream, reinterpret_cast<const pq_chunk_desc *>(w), (uint32_t)n_chunks, d_src,
                                       d_infos, reinterpret_cast<pq_chunk_result *>(w + roff)));
    PQ_HIP(ctx, hipMemcpyAsync(results, w + roff, rbytes, hipMemcpyDeviceToHost, ctx->stream));
    PQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PQ_OK;
}

int pq_walk_chunks(pq_ctx *ctx, const void *src, size_t src_bytes, const pq_chunk_desc *chunks, size_t n_chunks,
                   pq_page_info *infos, size_t infos_cap, pq_chunk_result *results)
{
    if (!ctx || (src_bytes && !src) || (infos_cap && !infos)) return PQ_ERR_BAD_ARGUMENT;
    PQ_HIP(ctx, hipSetDevice(ctx->device));
    int st = ensure_stage(ctx, 0, src_bytes + 32);
    if (st == PQ_OK) st = ensure_stage(ctx, 1, infos_cap * sizeof(pq_page_info) + 16);
    if (st != PQ_OK) return st;
    if (src_bytes) PQ_HIP(ctx, hipMemcpyAsync(ctx->stage[0], src, src_bytes, hipMemcpyHostToDevice, ctx->stream));
    st = pq_walk_chunks_dev(ctx, ctx->stage[0], src_bytes, chunks, n_chunks, static_cast<pq_page_info *>(ctx->stage[1]),
                            infos_cap, results);
    if (st != PQ_OK) return st;
    if (infos_cap)
        PQ_HIP(ctx, hipMemcpyAsync(infos, ctx->stage[1], infos_cap * sizeof(pq_page_info), hipMemcpyDeviceToHost,
                                   ctx->stream));
    PQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PQ_OK;
}

int pq_chunk_page_descs(const pq_chunk_desc *chunk, const pq_page_info *infos, uint32_t n_pages,
                        pq_chunk_layout *layout, pq_page_desc *descs, uint32_t *n_descs, uint64_t *dict_src,
                        uint64_t *dict_dst, uint64_t *dict_bytes)
{
    if (!chunk || !layout || !n_descs || (n_pages && (!infos || !descs))) return PQ_ERR_BAD_ARGUMENT;
    if (chunk->codec != 0) return PQ_ERR_BAD_ARGUMENT;
    uint32_t es = 0;  // bytes per fixed-width value
    switch (chunk->physical_type) {
    case 0: break;             // BOOLEAN
    case 1: case 4: es = 4; break;  // INT32, FLOAT
    case 2: case 5: es = 8; break;  // INT64, DOUBLE
    case 6: break;             // BYTE_ARRAY
    default: return PQ_ERR_BAD_ARGUMENT;
    }
    uint32_t dict_len = 0;
    uint64_t d_src = 0, d_dst = 0, d_bytes = 0;
    uint32_t out = 0;
    for (uint32_t i = 0; i < n_pages; ++i) {
        const pq_page_info &p = infos[i];
        if (p.type == 2) {
            // a PLAIN dictionary of fixed-width values is its page body, little endian
            // (dictPageReader.read, layout/page-dict.go:59-77; types/int32.go:56-72)
            if (es == 0 || p.num_values < 0) return PQ_ERR_BAD_ARGUMENT;
            if ((uint64_t)p.num_values * es > (uint64_t)(uint32_t)p.uncompressed_size)
                return PQ_UNEXPECTED_EOF;  // "unexpected number of values" (:70-77)
            dict_len = (uint32_t)p.num_values;
            d_src = chunk->src_off + p.data_off;
            d_dst = (layout->dict_off + 7) / 8 * 8;
            d_bytes = (uint64_t)dict_len * es;
            layout->dict_off = d_dst + d_bytes;
            continue;
        }
        if (p.type != 0 && p.type != 3) return PQ_ERR_BAD_ARGUMENT;
        pq_page_desc d;
        std::memset(&d, 0, sizeof d);
        d.kind = p.type == 0 ? PQ_KIND_PAGE_V1 : PQ_KIND_PAGE_V2;
        d.src_off = chunk->src_off + p.data_off;
        d.src_len = (uint64_t)(uint32_t)p.compressed_size;
        d.n_values = (uint32_t)p.num_values;
        d.max_rep = chunk->max_rep;
        d.max_def = chunk->max_def;
        d.rep_len = (uint32_t)p.rep_len;
        d.def_len = (uint32_t)p.def_len;
        const int32_t enc = p.encoding == 2 ? 8 : p.encoding;  // PLAIN_DICTIONARY (layout/page-dict.go:145-147)
        const uint32_t t = chunk->physical_type;
        uint64_t value_bytes = 0;
        if (enc == 8) {
            if (es == 0) return PQ_ERR_BAD_ARGUMENT;
            d.value_kind = PQ_KIND_DICT;
            d.elem_size = es;
            d.dict_off = d_dst;
            d.dict_len = dict_len;
            value_bytes = (uint64_t)d.n_values * es;
        } else if (enc == 0 && es != 0) {
            d.value_kind = PQ_KIND_PLAIN;
            d.elem_size = es;
            value_bytes = (uint64_t)d.n_values * es;
        } else if (enc == 0 && t == 0) {
            d.value_kind = PQ_KIND_BOOL_PLAIN;
            value_bytes = (uint64_t)d.n_values * 4;
        } else if (enc == 3 && t == 0) {
            d.value_kind = PQ_KIND_BOOL_RLE;
            value_bytes = (uint64_t)d.n_values * 4;
        } else if (enc == 5 && (t == 1 || t == 2)) {
            d.value_kind = t == 1 ? PQ_KIND_DELTA32 : PQ_KIND_DELTA64;
            value_bytes = (uint64_t)d.n_values * es;
        } else if ((enc == 6 || enc == 7) && t == 6) {
            d.value_kind = enc == 6 ? PQ_KIND_DELTA_LENGTH_BA : PQ_KIND_DELTA_BA;
        } else {
            return PQ_ERR_BAD_ARGUMENT;
        }
        if (d.value_kind == PQ_KIND_DELTA_LENGTH_BA || d.value_kind == PQ_KIND_DELTA_BA) {
            const bool front = d.value_kind == PQ_KIND_DELTA_BA;
            const uint64_t room = front ? layout->delta_ba_room : d.src_len;
            // 16-byte boundaries: the value bytes go out in 16-byte pieces, the offsets two a store
            d.dst_off = (layout->values_off + 15) / 16 * 16;
            d.dst_cap = front ? layout->delta_ba_room : 0;
            d.dict_off = (d.dst_off + room + 15) / 16 * 16;
            d.dict_len = d.n_values;
            layout->values_off = d.dict_off + 4 * ((front ? 2ull : 1ull) * d.n_values + 1);
        } else {
            const uint32_t vs = value_size(d);
            d.dst_off = (layout->values_off + vs - 1) / vs * vs;
            layout->values_off = d.dst_off + value_bytes;
        }
        if (chunk->max_rep == 0) {
            // both level decoders are ConstDecoder(0) (layout/chunk.go:111-124): nothing worth writing
            d.flags |= PQ_FLAG_NO_LEVEL_OUTPUT;
        } else {
            layout->levels_off = (layout->levels_off + 3) / 4 * 4;
            d.rep_off = layout->levels_off;
            d.def_off = d.rep_off + 4ull * d.n_values;
            layout->levels_off = d.def_off + 4ull * d.n_values;
        }
        descs[out++] = d;
    }
    *n_descs = out;
    if (dict_src) *dict_src = d_src;
    if (dict_dst) *dict_dst = d_dst;
    if (dict_bytes) *dict_bytes = d_bytes;
    return PQ_OK;
}

int pq_chunks_page_descs(const pq_chunk_desc *chunks, const pq_chunk_result *results, size_t n_chunks,
                         const pq_page_info *infos, pq_chunk_layout *layout, pq_page_desc *descs, size_t descs_cap,
                         uint32_t *first_desc, pq_dict_copy *copies, uint32_t *n_copies, int32_t *chunk_status)
{
    if (!layout || !first_desc || !n_copies || (n_chunks && (!chunks || !results || !infos || !copies)))
        return PQ_ERR_BAD_ARGUMENT;
    size_t out = 0;
    uint32_t nc = 0;
    int rc = PQ_OK;
    for (size_t i = 0; i < n_chunks; ++i) {
        first_desc[i] = (uint32_t)out;
        if (rc != PQ_OK) {  // the table is full: the chunks behind get no descriptors
            if (chunk_status) chunk_status[i] = PQ_ERR_OUT_OF_RANGE;
            continue;
        }
        if (results[i].status != PQ_OK) {
            if (chunk_status) chunk_status[i] = results[i].status;
            continue;
        }
        const uint32_t np = results[i].n_pages;
        if (out + np > descs_cap || (np && !descs)) {
            rc = PQ_ERR_OUT_OF_RANGE;
            if (chunk_status) chunk_status[i] = PQ_ERR_OUT_OF_RANGE;
            continue;
        }
        uint32_t made = 0;
        pq_dict_copy c{0, 0, 0};
        const pq_chunk_layout before = *layout;
        const int st = pq_chunk_page_descs(&chunks[i], infos + chunks[i].info_first, np, layout, descs + out, &made,
                                           &c.src, &c.dst, &c.bytes);
        if (chunk_status) chunk_status[i] = st;
        if (st != PQ_OK) {
            // a chunk this library does not decode (or a damaged dictionary page): no descriptors,
            // no room taken, the other chunks of the row group go on
            *layout = before;
            continue;
        }
        out += made;
        if (c.bytes) copies[nc++] = c;
    }
    first_desc[n_chunks] = (uint32_t)out;
    *n_copies = nc;
    return rc;
}

// ---- level post-processing ----------------------------------------------------

static int ensure_work(pq_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->work_cap) return PQ_OK;
    if (ctx->work) {
        // work queued on the stream may still use the old block
        PQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
        PQ_HIP(ctx, hipFree(ctx->work));
        ctx->work = nullptr;
        ctx->work_cap = 0;
    }
    const size_t cap = std::max<size_t>(bytes + bytes / 4, 1u << 20);
    PQ_HIP(ctx, hipMalloc(&ctx->work, cap));
    ctx->work_cap = cap;
    return PQ_OK;
}

int pq_levels_pack_dev(pq_ctx *ctx, const int32_t *d_levels, uint64_t n, int bit_width, void *d_out,
                       uint64_t out_bytes)
{
    if (!ctx || bit_width < 0 || bit_width > 16 || n > 0x7fffffffull) return PQ_ERR_BAD_ARGUMENT;
    const uint64_t need = (n + 7) / 8 * (uint64_t)bit_width;
    if (need == 0) return PQ_OK;
    if (!d_levels || !d_out || (reinterpret_cast<uintptr_t>(d_out) & 3u) ||
        (reinterpret_cast<uintptr_t>(d_levels) & 3u))
        return PQ_ERR_BAD_ARGUMENT;
    if (out_bytes < need) return PQ_ERR_OUT_OF_RANGE;
    PQ_HIP(ctx, hipSetDevice(ctx->device));
    PQ_HIP(ctx, pq::launch_levels_pack(ctx->stream, d_levels, n, bit_width, static_cast<uint8_t *>(d_out)));
    return PQ_OK;
}

int pq_levels_pack(pq_ctx *ctx, const int32_t *levels, uint64_t n, int bit_width, uint8_t *out,
                   uint64_t out_bytes)
{
    if (!ctx || bit_width < 0 || bit_width > 16 || n > 0x7fffffffull) return PQ_ERR_BAD_ARGUMENT;
    const uint64_t need = (n + 7) / 8 * (uint64_t)bit_width;
    if (need == 0) return PQ_OK;
    if (!levels || !out) return PQ_ERR_BAD_ARGUMENT;
    if (out_bytes < need) return PQ_ERR_OUT_OF_RANGE;
    PQ_HIP(ctx, hipSetDevice(ctx->device));
    int st = ensure_stage(ctx, 0, n * 4);
    if (st == PQ_OK) st = ensure_stage(ctx, 1, need);
    if (st != PQ_OK) return st;
    PQ_HIP(ctx, hipMemcpyAsync(ctx->stage[0], levels, n * 4, hipMemcpyHostToDevice, ctx->stream));
    st = pq_levels_pack_dev(ctx, static_cast<const int32_t *>(ctx->stage[0]), n, bit_width, ctx->stage[1], need);
    if (st != PQ_OK) return st;
    PQ_HIP(ctx, hipMemcpyAsync(out, ctx->stage[1], need, hipMemcpyDeviceToHost, ctx->stream));
    PQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PQ_OK;
}

int pq_column_entries_dev(pq_ctx *ctx, const int32_t *d_def, const int32_t *d_rep, uint64_t n,
                          uint32_t max_def, uint32_t max_rep, int repeated, uint32_t *d_offsets,
                          void *d_validity, int32_t *d_entry_def, pq_entries_result *d_result)
{
    if (!ctx || !d_offsets || !d_result || n > 0x7fffffffull) return PQ_ERR_BAD_ARGUMENT;
    if (max_def > 0xffffu || max_rep > 0xffffu) return PQ_ERR_BAD_ARGUMENT; // levels are uint16
    if (n && (!d_def || !d_validity || !d_entry_def || (repeated && !d_rep))) return PQ_ERR_BAD_ARGUMENT;
    if (reinterpret_cast<uintptr_t>(d_validity) & 3u) return PQ_ERR_BAD_ARGUMENT;
    PQ_HIP(ctx, hipSetDevice(ctx->device));
    const int st = ensure_work(ctx, pq::entries_workspace_bytes(n) + 64);
    if (st != PQ_OK) return st;
    PQ_HIP(ctx, pq::launch_column_entries(ctx->stream, d_def, d_rep, n, (int32_t)max_def, (int32_t)max_rep,
                                          repeated, d_offsets, static_cast<uint32_t *>(d_validity), d_entry_def,
                                          d_result, ctx->work));
    return PQ_OK;
}

int pq_column_entries_noalloc(pq_ctx *ctx, const int32_t *d_def, const int32_t *d_rep, uint64_t n,
                              uint32_t max_def, uint32_t max_rep, int repeated, uint32_t *d_offsets,
                              void *d_validity, int32_t *d_entry_def, pq_entries_result *d_result)
{
    if (!ctx || !d_offsets || !d_result || n > 0x7fffffffull) return PQ_ERR_BAD_ARGUMENT;
    if (max_def > 0xffffu || max_rep > 0xffffu) return PQ_ERR_BAD_ARGUMENT;
    if (n && (!d_def || !d_validity || !d_entry_def || (repeated && !d_rep))) return PQ_ERR_BAD_ARGUMENT;
    if (reinterpret_cast<uintptr_t>(d_validity) & 3u) return PQ_ERR_BAD_ARGUMENT;
    if (n > PQ_NOALLOC_MAX_SLOTS || pq::entries_workspace_bytes(n) > pq::static_scratch_bytes()) {
        ctx->last_error = "pq_column_entries_noalloc: more level slots than the static workspace serves";
        return PQ_ERR_OUT_OF_RANGE;
    }
    PQ_HIP(ctx, hipSetDevice(ctx->device));
    PQ_HIP(ctx, pq::launch_column_entries_static(ctx->device, ctx->stream, d_def, d_rep, n, (int32_t)max_def,
                                                 (int32_t)max_rep, repeated, d_offsets,
                                                 static_cast<uint32_t *>(d_validity), d_entry_def, d_result));
    return PQ_OK;
}

int pq_column_entries(pq_ctx *ctx, const int32_t *def, const int32_t *rep, uint64_t n, uint32_t max_def,
                      uint32_t max_rep, int repeated, uint32_t *offsets, uint8_t *validity,
                      int32_t *entry_def, pq_entries_result *result)
{
    if (!ctx || !offsets || !result || n > 0x7fffffffull) return PQ_ERR_BAD_ARGUMENT;
    if (n && (!def || !validity || !entry_def || (repeated && !rep))) return PQ_ERR_BAD_ARGUMENT;
    PQ_HIP(ctx, hipSetDevice(ctx->device));
    // stage 0: def | rep, stage 1: offsets | entry_def, stage 2: validity | result
    const size_t lv = (size_t)n * 4, vbytes = (size_t)((n + 31) / 32 * 4);
    int st = ensure_stage(ctx, 0, 2 * lv + 16);
    if (st == PQ_OK) st = ensure_stage(ctx, 1, 2 * lv + 16);
    if (st == PQ_OK) st = ensure_stage(ctx, 2, vbytes + 64);
    if (st != PQ_OK) return st;
    uint8_t *s0 = static_cast<uint8_t *>(ctx->stage[0]), *s1 = static_cast<uint8_t *>(ctx->stage[1]),
            *s2 = static_cast<uint8_t *>(ctx->stage[2]);
    if (n) PQ_HIP(ctx, hipMemcpyAsync(s0, def, lv, hipMemcpyHostToDevice, ctx->stream));
    if (n && repeated) PQ_HIP(ctx, hipMemcpyAsync(s0 + lv, rep, lv, hipMemcpyHostToDevice, ctx->stream));
    pq_entries_result *d_res = reinterpret_cast<pq_entries_result *>(s2 + vbytes + ((16 - vbytes % 16) % 16));
    st = pq_column_entries_dev(ctx, reinterpret_cast<const int32_t *>(s0), reinterpret_cast<const int32_t *>(s0 + lv),
                               n, max_def, max_rep, repeated, reinterpret_cast<uint32_t *>(s1), s2,
                               reinterpret_cast<int32_t *>(s1 + lv + 4), d_res);
    if (st != PQ_OK) return st;
    PQ_HIP(ctx, hipMemcpyAsync(result, d_res, sizeof *result, hipMemcpyDeviceToHost, ctx->stream));
    PQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const size_t ne = result->n_entries;
    PQ_HIP(ctx, hipMemcpyAsync(offsets, s1, (ne + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (ne) PQ_HIP(ctx, hipMemcpyAsync(entry_def, s1 + lv + 4, ne * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (vbytes) PQ_HIP(ctx, hipMemcpyAsync(validity, s2, vbytes, hipMemcpyDeviceToHost, ctx->stream));
    PQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PQ_OK;
}

} // extern "C"
