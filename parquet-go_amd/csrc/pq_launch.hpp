// pq_launch.hpp — host-side launchers of the gfx950 kernels (internal to the library).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "pqdecode.h"

namespace pq {

hipError_t launch_bitunpack32(hipStream_t s, const void *d_src, size_t src_len, int w, size_t n_groups,
                              int32_t *d_dst);
hipError_t launch_bitunpack64(hipStream_t s, const void *d_src, size_t src_len, int w, size_t n_groups,
                              int64_t *d_dst);

// Buffers every page kernel works on (device pointers).
struct PageBuffers {
    const uint8_t *src;
    uint8_t *values;
    uint8_t *levels;
    const uint8_t *dict;
};

// How the one-wave-per-descriptor kernels are used: from min_pages descriptors on.
struct WaveLaunch {
    uint32_t min_pages;
};

// ids: device array of `count` descriptor indices handled by this launch.
// dict_lds_bytes: LDS set aside per workgroup for a copy of a dictionary (0 = gather from global):
// the largest dictionary of the plan that fits kDictLdsMax.
// The last n_hseg of the ids are the streams launch_seg_hybrid looked at before, in the order of
// d_hseg_pages: those it cut (mode != 0) are done, the others are decoded here.
struct HsegPage;
hipError_t launch_stream_pages(hipStream_t s, const pq_page_desc *d_descs, const uint32_t *d_ids,
                               uint32_t count, PageBuffers buf, pq_page_result *d_results,
                               uint32_t dict_lds_bytes, const WaveLaunch &wl, const HsegPage *d_hseg_pages,
                               uint32_t n_hseg);
// wave_min_pages: from this many pages on, the one-wave-per-page kernel runs first and a
// sweep kernel (woken through *d_flag == epoch) finishes the pages it declined.
// *needs_serial: launch_delta_serial must follow for these descriptors.
// The sweep looks at ids[0 .. sweep_count): the pages of this launch and, behind them, the pages
// launch_seg_delta decoded.
hipError_t launch_delta_pages(hipStream_t s, int bits, const pq_page_desc *d_descs,
                              const uint32_t *d_ids, uint32_t count, uint32_t sweep_count, PageBuffers buf,
                              pq_page_result *d_results, uint32_t wave_min_pages, uint32_t *d_flag,
                              uint32_t epoch, bool *needs_serial);
hipError_t launch_data_pages(hipStream_t s, const pq_page_desc *d_descs, const uint32_t *d_ids,
                             uint32_t count, PageBuffers buf, pq_page_result *d_results,
                             uint32_t dict_lds_bytes, const WaveLaunch &wl, uint32_t *d_flag, uint32_t epoch);

// Finishes, one lane per page, the delta descriptors among ids[0..count) that the
// tiled kernels marked as needing the serial path; a no-op for all others.
// bytearray.hip: kind = PQ_KIND_DELTA_LENGTH_BA or PQ_KIND_DELTA_BA; the ids name bare streams of that
// kind or flat-column pages (max_rep == 0, PQ_FLAG_NO_LEVEL_OUTPUT) whose values are of that kind
hipError_t launch_byte_arrays(hipStream_t s, uint32_t kind, const pq_page_desc *d_descs, const uint32_t *d_ids,
                              uint32_t count, PageBuffers buf, pq_page_result *d_results);
hipError_t launch_delta_serial(hipStream_t s, const pq_page_desc *d_descs, const uint32_t *d_ids,
                               uint32_t count, PageBuffers buf, pq_page_result *d_results);

// Behind every other kernel of a run: ids[0..count) name descriptors that were data pages of a flat
// column and ran as their value streams; turns each stream result into the page's, and writes the
// pages' constant levels where asked for (write_levels: some descriptor lacks PQ_FLAG_NO_LEVEL_OUTPUT).
hipError_t launch_lowered_finish(hipStream_t s, const pq_page_desc *d_descs, const uint32_t *d_ids, uint32_t count,
                                 PageBuffers buf, pq_page_result *d_results, bool write_levels);

// d_out[i] = d_in[i]; plans over several buffer sets: with its offsets moved by d_deltas[4 * d_set_of[i] ..]
// (src, values, levels, dict).  d_set_of / d_deltas null: a plain copy (plans that rewrite descriptors
// on the device start every run from the table as created).
hipError_t launch_rebase(hipStream_t s, const pq_page_desc *d_in, const uint32_t *d_set_of, const uint64_t *d_deltas,
                         uint32_t n, pq_page_desc *d_out);

constexpr uint32_t kDictLdsMax = 48u * 1024u;

// ---- segmented delta pages (segdelta.hip) -------------------------------------------------------
// One record per page that is cut into segments.  The host fills desc, seg_first, n_segs and
// seg_bytes; the rest is written on the device every run.
struct SegPage {
    uint32_t desc;       // index of the page's descriptor
    uint32_t seg_first;  // index of its first segment record
    uint32_t n_segs;
    uint32_t seg_bytes;  // bytes of the stream per segment
    uint32_t whole;      // 1: unusual stream header, segment 0 decodes the whole page
    uint32_t lg;         // log2 of the values per miniblock: 5 (128 / 4) or 6 (256 / 4)
    uint32_t total;      // value count the stream declares
    uint32_t pos0;       // stream offset of the first block header
    uint64_t first;      // first value
    uint32_t giveup;     // a wave's wait for the segments in front of it ran out: the page is decoded whole
    uint32_t pad;
};
// One record per segment, written on the device: what pq_plan_segments shows.
constexpr uint32_t kSegSkip = 0, kSegPart = 1, kSegTail = 2, kSegWhole = 3;
struct SegRec {
    // what the segment's walk found (from the guess, or from the true start when the guess was elsewhere)
    uint32_t key;      // stream offset of the first block's width bytes (0xffffffff: none found)
    uint32_t exit;     // header of the first block of the next segment, reached by walking whole blocks
    uint32_t nblk;     // blocks walked
    uint32_t exit_key; // stream offset of the width bytes of the header at `exit`
    uint64_t sum;      // sum of the walked blocks' deltas, without the first block's minDelta (segment 0: with it)
    uint32_t exit_md;  // minDelta of the header at `exit`
    // its place in the page
    uint32_t flags;    // kSeg*: what the decode does with the segment
    uint64_t value;    // the value at position p0
    uint32_t entry;    // stream offset of the first block's header
    uint32_t p0;       // position of the segment's first value
    uint32_t stop;     // kSegPart: stream offset to stop in front of
    uint32_t pad[3];
};
static_assert(sizeof(SegRec) == 64, "segment records are 64 bytes");
// One per segment: what the waves of one page show each other while the kernel runs (segdelta.hip
// has the layout of the words).  A segment first shows a summary of its own walk (a0..a2), later
// its place in the page (p0..p2: everything up to its end).
struct SegLook {
    uint64_t a0, a1, a2, a3;
    uint64_t p0, p1, p2, p3;
};
static_assert(sizeof(SegLook) == 64, "look-back records are 64 bytes");
constexpr uint32_t kSegSpinLimit = 1u << 20;  // looks a wave takes at the segments in front of it before it gives up
// d_pages / d_owner (segment -> index into d_pages) / d_recs / d_looks: n_pages, n_segs, n_segs, n_segs entries.
hipError_t launch_seg_delta(hipStream_t s, int bits, const pq_page_desc *d_descs, SegPage *d_pages, uint32_t n_pages,
                            const uint32_t *d_owner, uint32_t n_segs, SegRec *d_recs, SegLook *d_looks, PageBuffers buf,
                            pq_page_result *d_results, uint32_t *d_flag, uint32_t epoch, uint32_t spin_limit);

// ---- segmented hybrid streams (seghybrid.hip) ---------------------------------------------------
// One record per hybrid stream (dictionary-index page, level stream, BOOLEAN RLE values, raw stream)
// that may be cut into segments.  The host fills the first six fields; the rest is written on the
// device every run.
struct HsegPage {
    uint32_t desc;         // index of the page's descriptor
    uint32_t seg_first;    // index of its first segment record
    uint32_t n_segs;       // segment records it owns
    uint32_t seg_bytes;    // bytes of the stream a segment aims at
    uint32_t check_first;  // index of its first run-check counter
    uint32_t n_checks;     // counters it owns (64 runs each)
    uint32_t mode;         // 0: not cut, the one-wave-per-stream kernel decodes it; 1: a chain of equal
                           // bit-packed runs; 2: runs of any kind
    uint32_t w;            // bit width
    uint32_t pos0;         // offset of the first run header from the descriptor's first byte
    uint32_t end;          // offset of the end of the stream
    uint32_t stride;       // mode 1: bytes from one run header to the next
    uint32_t vpr;          // mode 1: values per run
    uint32_t head;         // mode 1: the run header's bytes
    uint32_t hl;           // and how many they are
    uint32_t runs_per_seg;
    uint32_t n_part;       // segments decoded apart from the tail; the tail is segment n_part
    uint32_t kind_mode;    // HY_RAW, HY_DICT or HY_LEVELS: what the decode does with a value
    uint32_t max_level;    // HY_LEVELS
    uint32_t fixed_consumed;  // != 0xffffffff: the consumed bytes the result reports (InitSize, buffered Init)
    uint32_t tab_first;    // mode 2: where the page's run-header offsets start in the plan's table, in units of
                           // 8 entries (0xffffffff: the plan keeps none); n_segs * seg_bytes / 2 entries
};
struct HsegRec {
    uint64_t off;        // stream offset the segment starts at
    uint32_t v0;         // index of its first value
    uint32_t count;      // values it decodes
    uint32_t flags;      // kSeg*
    int32_t status;      // what its decode returned
    uint32_t n_decoded;
    uint32_t count_max;  // HY_LEVELS: values equal to the maximum level
    uint64_t consumed;
    uint32_t len;        // bytes of the stream it may read
    uint32_t runs;       // mode 2: run-header offsets the walk left in the segment's part of the table
};
// Mode 2: what a walk over the headers of one segment found.
struct HsegScan {
    uint32_t entry;  // offset of the first run header at or behind the segment's first byte
    uint32_t exit;   // offset of the first run header at or behind its end (0xffffffff: the walk ran into
                     // something that is not a complete, well-formed run)
    uint32_t nvals;  // values of the runs from entry to exit (saturating at 2^31 - 1)
    uint32_t runs;   // header offsets (from entry on, 16 bits each) the walk wrote to the segment's part of the
                     // table: all of its runs, or 0
};
// d_check_owner / d_seg_owner: counter -> page and segment -> page maps.  d_scans: 2 * n_segs records.
// d_tabs: the run-header table HsegPage::tab_first points into, or nullptr.  fix_rounds: 1 or 2 rounds of
// re-walking segments entered at the wrong place in front of the pass that links them.
hipError_t launch_seg_hybrid(hipStream_t s, const pq_page_desc *d_descs, HsegPage *d_pages, uint32_t n_pages,
                             const uint32_t *d_check_owner, uint32_t n_checks, uint32_t *d_counts,
                             const uint32_t *d_seg_owner, uint32_t n_segs, HsegRec *d_recs, HsegScan *d_scans,
                             PageBuffers buf, pq_page_result *d_results, uint32_t dict_lds_bytes, uint16_t *d_tabs,
                             int fix_rounds);

// ---- large nested pages as three units (nested.hip) ---------------------------------------------
// The host fills the first nine fields; the rest is written on the device every run.
struct NestedPage {
    uint32_t desc;      // the page's own descriptor slot: its value unit
    uint32_t rep_unit;  // descriptor slots of the two level units
    uint32_t def_unit;
    uint32_t version;   // 1 or 2
    uint32_t rep_len, def_len;  // V2: the header's level byte lengths
    uint32_t max_rep, max_def;
    uint32_t n_values;  // the header's NumValues: level slots
    int32_t early_status, early_stage;  // read() itself failed
    int32_t lvl_status, lvl_stage;      // how the level decode of ReadValues ended
    uint32_t not_null;
    uint32_t pad[2];
};
static_assert(sizeof(NestedPage) == 64, "nested page records are 64 bytes");
hipError_t launch_nested_split(hipStream_t s, pq_page_desc *d_descs, NestedPage *d_pages, uint32_t n_pages, PageBuffers buf);
hipError_t launch_nested_values(hipStream_t s, pq_page_desc *d_descs, NestedPage *d_pages, uint32_t n_pages,
                                const pq_page_result *d_results);
hipError_t launch_nested_finish(hipStream_t s, const pq_page_desc *d_descs, const NestedPage *d_pages, uint32_t n_pages,
                                pq_page_result *d_results);

// Level post-processing (assemble.hip).  d_out: ceil(n / 8) * w bytes, 4-byte aligned.
hipError_t launch_levels_pack(hipStream_t s, const int32_t *d_levels, uint64_t n, int w, uint8_t *d_out);
// d_work: entries_workspace_bytes(n) bytes, 8-byte aligned; d_validity: 4 * ceil(n / 32) bytes.
size_t entries_workspace_bytes(uint64_t n);
hipError_t launch_column_entries(hipStream_t s, const int32_t *d_def, const int32_t *d_rep, uint64_t n,
                                 int32_t max_def, int32_t max_rep, int repeated, uint32_t *d_offsets,
                                 uint32_t *d_validity, int32_t *d_entry_def, pq_entries_result *d_result,
                                 void *d_work);

// The same over the library's static device workspace (static_scratch_bytes() per GPU): no
// allocation.  static_scratch_init(device) once per device, with that device current, first.
size_t static_scratch_bytes();
hipError_t static_scratch_init(int device);
hipError_t launch_column_entries_static(int device, hipStream_t s, const int32_t *d_def, const int32_t *d_rep,
                                        uint64_t n, int32_t max_def, int32_t max_rep, int repeated,
                                        uint32_t *d_offsets, uint32_t *d_validity, int32_t *d_entry_def,
                                        pq_entries_result *d_result);

// Page-header walk (walk.hip): one wave per chunk; d_infos / d_results as in pq_walk_chunks_dev.
hipError_t launch_walk_chunks(hipStream_t s, const pq_chunk_desc *d_chunks, uint32_t n_chunks, const void *d_src,
                              pq_page_info *d_infos, pq_chunk_result *d_results);

} // namespace pq
