/*
 * pqdecode.h — C ABI of the MI355X (gfx950) Parquet page-value decoder.
 *
 * Drop-in boundary for the decode hot path of hexbee-net/parquet-go.  The
 * reference has no FFI of its own; each entry point below names the Go
 * interface or function whose body a cgo shim replaces with a call to it
 * (paths are relative to the reference root).  Everything is plain pointers
 * and sizes; the caller owns every buffer; the library keeps no caller pointer
 * after a call returns unless stated.  No function throws or aborts: every
 * failure is an integer status.
 *
 * Two flavours of each operation:
 *   *_dev / pq_plan_*  device pointers in and out, asynchronous on the
 *                      context's HIP stream (the device-resident path);
 *   plain names        host pointers; the call stages H2D, decodes, copies
 *                      D2H and synchronises before returning.
 *
 * Source buffers: the kernels read encoded bytes in whole, 16-byte aligned lines.  A unit that
 * occupies [p, p + len) of a source buffer makes them read every aligned 16-byte line that
 * intersects that range, i.e. up to 15 bytes in front of p and up to 15 bytes behind p + len (never
 * a line that holds none of its bytes; what is read outside the unit never reaches an output).
 * Device source buffers (d_src of pq_plan_run, pq_bitunpack*_dev, pq_walk_chunks_dev) must therefore
 * be readable from the 16-byte boundary at or below their first byte to the one at or above their
 * last: any hipMalloc'ed buffer is (its start is aligned and its size is rounded up), a pointer
 * into the middle of one is as long as the allocation covers those lines.  The host-pointer entry
 * points stage into buffers of their own and ask nothing of the caller's memory.
 *
 * Threading: a pq_ctx (and the plans made from it) may be used by one thread at
 * a time; create one context per calling thread / goroutine-locked OS thread.
 */
#ifndef PQDECODE_H
#define PQDECODE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ---------------------------------------------------------
 * 1..30 are the reference's error classes; >= 100 are the library's own. */
#define PQ_OK 0
#define PQ_EOF 1                     /* io.EOF */
#define PQ_UNEXPECTED_EOF 2          /* io.ErrUnexpectedEOF */
#define PQ_EMPTY_BP_RUN 3            /* encoding/hybrid-decoder.go:123 */
#define PQ_EMPTY_RLE_RUN 4           /* encoding/hybrid-decoder.go:130 */
#define PQ_RLE_VALUE_TOO_LARGE 5     /* encoding/hybrid-decoder.go:166 */
#define PQ_INT32_OUT_OF_RANGE 6      /* encoding/helpers.go:36,54 */
#define PQ_VARINT_OVERFLOW 7         /* Go encoding/binary varint overflow */
#define PQ_INVALID_BLOCK_SIZE 8      /* encoding/delta-binary-packed-decoder.go:34 */
#define PQ_INVALID_MINIBLOCK_COUNT 9 /* encoding/delta-binary-packed-decoder.go:46 */
#define PQ_INVALID_MINIBLOCK_WIDTH 10/* encoding/delta-binary-packed-decoder.go:80 */
#define PQ_DICT_INVALID_WIDTH 11     /* types/dict.go:75 */
#define PQ_DICT_INVALID_INDEX 12     /* types/dict.go:104 (and the negative-key panic) */
#define PQ_NOT_INITIALIZED 13        /* encoding/hybrid-decoder.go:79 */
#define PQ_INVALID_STREAM 14         /* encoding/delta-binary-packed-decoder.go:154 */
#define PQ_PAGE_VALUES_FAILED 15     /* layout/page-v1.go:104, layout/page-v2.go:137 */
#define PQ_BYTE_ARRAY_NEGATIVE_LEN 16/* types/byte-array.go:210,326: make([]byte, ..) panics for a size < 0 */
#define PQ_BYTE_ARRAY_COUNT_MISMATCH 17 /* types/byte-array.go:303: different number of suffixes and prefixes */
#define PQ_BYTE_ARRAY_INVALID_PREFIX 18 /* types/byte-array.go:331: invalid prefix len in the stream */
/* page-header walk (pq_walk_chunks) */
#define PQ_WALK_HEADER_FAILED 19          /* layout/chunk.go:142: failed to read page header */
#define PQ_WALK_SECOND_DICTIONARY 20      /* layout/chunk.go:150: there should be only one dictionary */
#define PQ_WALK_PAGE_TYPE 21              /* layout/chunk.go:189: page type not supported */
#define PQ_WALK_MISSING_HEADER 22         /* layout/page-v1.go:42 (in fact the nil dereference of :22),
                                             page-v2.go:46, page-dict.go:30 */
#define PQ_WALK_NEGATIVE_NUM_VALUES 23    /* layout/page-v1.go:45, page-v2.go:49, page-dict.go:33 */
#define PQ_WALK_NEGATIVE_LEVELS_LENGTH 24 /* layout/page-v2.go:57-71 */
#define PQ_WALK_DICT_ENCODING 25          /* layout/page-dict.go:42 */
#define PQ_WALK_LEVEL_ENCODING 26         /* layout/chunk.go:79,95 */
#define PQ_WALK_VALUE_ENCODING 27         /* layout/page-dict.go:84-345: no decoder for this type / encoding */
#define PQ_WALK_BLOCK_SIZE 28             /* layout/block.go:23, page-v2.go:87-90: the page's bytes are not there */
#define PQ_WALK_UNCOMPRESSED_SIZE 29      /* layout/block.go:37 */
#define PQ_WALK_SEEK 30                   /* layout/chunk.go:174 */

#define PQ_ERR_BAD_ARGUMENT 100
#define PQ_ERR_HIP 101               /* a HIP runtime call failed; see pq_last_error */
#define PQ_ERR_NO_DEVICE 102
#define PQ_ERR_OUT_OF_RANGE 103      /* a descriptor points outside the buffers given */

/* stage of a data-page read at which `status` surfaced (pq_page_result.stage) */
#define PQ_STAGE_NONE 0
#define PQ_STAGE_REP_INIT 1
#define PQ_STAGE_DEF_INIT 2
#define PQ_STAGE_VALUES_INIT 3
#define PQ_STAGE_REP 4
#define PQ_STAGE_DEF 5
#define PQ_STAGE_VALUES 6

/* ---- context ------------------------------------------------------------- */
typedef struct pq_ctx pq_ctx;

/* hip_stream: a hipStream_t to launch on (e.g. the caller's stream), or NULL to
 * let the library create its own non-blocking stream on `device`. */
int pq_ctx_create(int device, void *hip_stream, pq_ctx **out);
void pq_ctx_destroy(pq_ctx *ctx);
int pq_ctx_synchronize(pq_ctx *ctx);
/* Tuning knobs (no counterpart in the reference).
 * PQ_OPT_DELTA_WAVE_MIN_PAGES: a batch with at least this many delta descriptors of one
 * element size is decoded one wave per page; smaller batches use one workgroup per page.
 * Default 1: on MI355X the wave kernel is faster from a single page on (80 us against
 * 500 us for one 20 000-value page); the knob remains for experiments and tests. */
#define PQ_OPT_DELTA_WAVE_MIN_PAGES 1
/* PQ_OPT_HYBRID_WAVE_MIN_PAGES: the same choice for hybrid / dictionary / level streams and
 * whole data pages.  Default 1 (equal up to about 256 pages, the wave kernels ahead beyond). */
#define PQ_OPT_HYBRID_WAVE_MIN_PAGES 2
/* PQ_OPT_SEGMENT_MAX_PAGES: a plan with fewer than this many DELTA32 (or DELTA64) descriptors has
 * its large ones decoded in segments, many waves to a page (a page is one serial chain of blocks,
 * encoding/delta-binary-packed-decoder.go:130-147; a few large pages would otherwise leave most of
 * the machine idle; measured on MI355X the cut pays below about 1000 pages, whatever their size).
 * Only blocks whose miniblock widths are at most 32 can be stepped over without decoding them: a
 * page is cut up to its first block that is wider, and a page whose first block is wider is not cut
 * (the reference rejects such a block in INT64 columns too,
 * encoding/delta-binary-packed-decoder.go:78-88).  A minDelta of any size is stepped over (five
 * varint bytes and more: INT64 columns that grow by 2^27 and more a value, at about half the rate
 * of shorter ones); in INT32 columns and in blocks of 256 values only up to four bytes.  Data pages of flat columns (PQ_KIND_PAGE_V1/_V2 with
 * max_rep == 0, max_def == 0) are cut like the bare streams of their value encoding.
 * Hybrid streams (encoding/hybrid-decoder.go:70-170) are cut under the same option: dictionary-index
 * pages (PQ_KIND_DICT), raw streams (PQ_KIND_HYBRID), level streams (PQ_KIND_LEVELS) and BOOLEAN RLE
 * values, whether their runs are bit-packed runs of one repeated length or runs of any kind (then
 * in segments of about 8192 values).  Data pages of nested columns (max_rep != 0) with 65 536
 * slots and more run as three such units, repetition levels, definition levels and values, which
 * the V1 length prefixes and V2 header lengths tell apart before anything is decoded
 * (layout/page-v1.go:69-77, layout/page-v2.go:110-126).  A plan that cuts streams of runs of any kind
 * keeps, in its device block, two bytes of table per two bytes of such a stream's segments (about twice
 * the streams' size; the walk that proves a segment leaves the places of its run headers there for the
 * decode); plans for which that would be more than 2 GiB keep none and decode a little slower.
 * Default 1024; 0 = never.  Results are the same either way.
 * PQ_OPT_SEGMENT_BYTES: bytes of the stream per segment, a multiple of 16 from 2048 on; 0 (the
 * default) = chosen from the plan's size, 8 to 64 KiB.  Pages shorter than three segments are not
 * cut.  Both options apply to plans created after the call. */
#define PQ_OPT_SEGMENT_MAX_PAGES 3
#define PQ_OPT_SEGMENT_BYTES 4
/* PQ_OPT_CHUNK_SLOTS: pipeline slots of pq_decode_chunks, 1..8 (default 3). */
#define PQ_OPT_CHUNK_SLOTS 5
/* PQ_OPT_SEGMENT_SPIN_LIMIT (for tests): the waves that decode the segments of one delta page learn
 * where their segment stands in the page from the waves in front of them, and wait for that while
 * the same kernel runs.  The wait is bounded: after this many looks a wave gives the page up, and the
 * page is decoded whole by the kernel behind.  Default 2^20 (never reached on MI355X, where
 * workgroups start in the order of their index); 0 = a wave that would have to wait gives up at once,
 * which is how the tests reach that path.  Results are the same either way. */
#define PQ_OPT_SEGMENT_SPIN_LIMIT 6
/* PQ_OPT_CONCURRENT_GROUPS: a plan decodes its descriptors in groups, one kernel each (streams, delta
 * pages, whole data pages, byte arrays).  With 1 (the default) a plan in which at least two of them
 * have 16 MiB and more to move runs them side by side on streams of the library's own, forked from
 * the context's stream and joined to it again before pq_plan_run returns control of that stream to
 * the work behind it: nothing changes for the caller but the time.  0: one after the other on the
 * context's stream.  Never used while the context's stream is being captured into a graph. */
#define PQ_OPT_CONCURRENT_GROUPS 7
/* PQ_OPT_RUN_TABLES: 1 (the default): a plan that cuts hybrid streams of runs of any kind into segments keeps
 * the table of run-header offsets described under PQ_OPT_SEGMENT_MAX_PAGES; 0: plans created from now on keep
 * none (their device block is smaller by about twice those streams' bytes, their segments' decode finds the
 * run headers itself, about a third slower).  Results are the same either way. */
#define PQ_OPT_RUN_TABLES 8
int pq_ctx_set_option(pq_ctx *ctx, int option, int64_t value);
const char *pq_status_string(int status);
const char *pq_last_error(pq_ctx *ctx); /* text of the last HIP failure on ctx */
int pq_version(void);

/* ---- fixed-width bit-unpack ----------------------------------------------
 * Replaces the unpack8int32_w / unpack8int64_w table calls
 * (encoding/bitpacking32.go:10-44, encoding/bitpacking64.go:10-76) over
 * n_groups consecutive groups of 8 values: group g is the w bytes at
 * src + g*w, value k of it = bits [k*w,(k+1)*w) of that little-endian group.
 * 0 <= w <= 32 (int32) / 64 (int64); src_len >= n_groups*w. */
int pq_bitunpack32_dev(pq_ctx *ctx, const void *d_src, size_t src_len, int w, size_t n_groups,
                       int32_t *d_dst);
int pq_bitunpack64_dev(pq_ctx *ctx, const void *d_src, size_t src_len, int w, size_t n_groups,
                       int64_t *d_dst);
int pq_bitunpack32(pq_ctx *ctx, const uint8_t *src, size_t src_len, int w, size_t n_groups,
                   int32_t *dst);
int pq_bitunpack64(pq_ctx *ctx, const uint8_t *src, size_t src_len, int w, size_t n_groups,
                   int64_t *dst);

/* ---- page descriptors ----------------------------------------------------
 * One descriptor = one unit of independent work (one encoded stream or one
 * whole data page).  Offsets are BYTES into the buffers handed to
 * pq_plan_run / pq_decode_pages. */
#define PQ_KIND_HYBRID 1   /* encoding.HybridDecoder: Init/InitSize + n x Next()
                              (encoding/hybrid-decoder.go:42-109) -> int32 */
#define PQ_KIND_DICT 2     /* types.DictDecoder.Init + DecodeValues (types/dict.go:66-117):
                              src starts at the bit-width byte -> elem_size-byte values */
#define PQ_KIND_LEVELS 3   /* layout.decodePackedArray over a level decoder
                              (layout/helpers.go:68-90, layout/chunk.go:78-124) -> int32
                              levels + not_null */
#define PQ_KIND_DELTA32 4  /* types.Int32DeltaBPDecoder Init + DecodeValues
                              (types/int32.go:103-123, encoding/delta-binary-packed-decoder.go:174-246) */
#define PQ_KIND_DELTA64 5  /* types.Int64DeltaBPDecoder (types/int64.go:103-123,
                              encoding/delta-binary-packed-decoder.go:250-322) */
#define PQ_KIND_PAGE_V1 6  /* dataPageReaderV1.read + ReadValues (layout/page-v1.go:40-116) */
#define PQ_KIND_PAGE_V2 7  /* dataPageReaderV2.read + ReadValues (layout/page-v2.go:42-149) */
#define PQ_KIND_PLAIN 8    /* Int32/Int64/Float/DoublePlainDecoder.DecodeValues (types/int32.go:56-72,
                              types/int64.go:56-72, types/float.go:67-79, types/double.go:68-80):
                              elem_size (4 or 8) little-endian bytes per value */
#define PQ_KIND_BOOL_PLAIN 9 /* BooleanPlainDecoder.DecodeValues on a fresh decoder
                              (types/boolean.go:59-104): bit j of byte i is value 8 i + j,
                              written as the int32 0 / 1 that unpackByte yields */
#define PQ_KIND_BOOL_RLE 10  /* BooleanRLEDecoder.Init + DecodeValues (types/boolean.go:173-196): a u32 LE
                              length, then a hybrid stream of width 1; one int32 0 / 1 per value.
                              Alone it is PQ_KIND_HYBRID with bit_width 1 and
                              PQ_FLAG_LENGTH_PREFIXED; as a value_kind it is the values section of
                              a BOOLEAN page in RLE encoding */
#define PQ_KIND_DELTA_LENGTH_BA 11 /* ByteArrayDeltaLengthDecoder.Init + DecodeValues(dest[:n_values])
                              (types/byte-array.go:168-219).  Output, both in the values buffer:
                              the value bytes back to back at dst_off (at most src_len of them),
                              and at dict_off (4-byte aligned; 16-byte boundaries for both areas
                              are faster, and what pq_chunk_page_descs lays out) n_decoded + 1
                              uint32 offsets into them, offsets[0] = 0.  dict_len = the number of lengths the area at
                              dict_off has room for (it holds dict_len + 1 words and serves as
                              scratch): at least n_values, and at least the count the stream
                              itself declares, else the page ends with PQ_ERR_OUT_OF_RANGE.
                              src_len < 2^32.  Also a value_kind of data pages. */
#define PQ_KIND_DELTA_BA 12 /* ByteArrayDeltaDecoder.Init + DecodeValues(dest[:n_values])
                              (types/byte-array.go:279-348).  Output as for DELTA_LENGTH_BA, except
                              that the values can be longer than the stream: dst_cap is the room
                              at dst_off in bytes (PQ_ERR_OUT_OF_RANGE when they do not fit), and
                              the area at dict_off holds 2 * dict_len + 1 words.  "invalid prefix
                              len" reports n_decoded 0 like the reference (:330).  Also a
                              value_kind of data pages. */

#define PQ_FLAG_LENGTH_PREFIXED 1u /* HYBRID/LEVELS: InitSize (u32 LE length first) */
#define PQ_FLAG_LEVELS_CONST 2u    /* LEVELS: ConstDecoder(0) (layout/chunk.go:111-124) */
#define PQ_FLAG_NO_INIT 4u         /* HYBRID/LEVELS: decoder never initialised
                                      (zero-length V2 level section, layout/page-v2.go:92-102) */
#define PQ_FLAG_NO_LEVEL_OUTPUT 8u /* pages of a column with max_rep == 0: both level decoders are
                                      ConstDecoder(0) (layout/chunk.go:111-124) and every level is 0;
                                      do not write the zeros (rep_off / def_off are not used) */

typedef struct pq_page_desc {
    uint32_t kind;        /* PQ_KIND_* */
    uint32_t flags;       /* PQ_FLAG_* */
    uint64_t src_off;     /* first byte of this unit in the source buffer */
    uint64_t src_len;     /* bytes available to it */
    uint64_t dst_off;     /* byte offset of its value output in the values buffer */
    uint64_t rep_off;     /* byte offset of its int32 rep-level output (levels buffer) */
    uint64_t def_off;     /* byte offset of its int32 def-level output (levels buffer);
                             PQ_KIND_LEVELS writes its single stream here */
    uint64_t dict_off;    /* byte offset of the dictionary in the dictionary buffer;
                             DELTA_LENGTH_BA, DELTA_BA: of the offsets area in the values buffer */
    uint32_t n_values;    /* values to pull; for pages: header NumValues (level slots) */
    uint32_t bit_width;   /* HYBRID: w (0..32) */
    uint32_t max_rep;     /* pages: column max repetition level */
    uint32_t max_def;     /* pages: max definition level; LEVELS: the stream's max level */
    uint32_t rep_len;     /* PAGE_V2: RepetitionLevelsByteLength */
    uint32_t def_len;     /* PAGE_V2: DefinitionLevelsByteLength */
    uint32_t value_kind;  /* pages: PQ_KIND_DICT, PQ_KIND_DELTA32, PQ_KIND_DELTA64, PQ_KIND_PLAIN,
                             PQ_KIND_BOOL_PLAIN, PQ_KIND_BOOL_RLE, PQ_KIND_DELTA_LENGTH_BA or
                             PQ_KIND_DELTA_BA */
    uint32_t elem_size;   /* DICT: bytes per dictionary entry (4 or 8); PLAIN: bytes per value */
    uint32_t dict_len;    /* DICT: number of dictionary entries; DELTA_LENGTH_BA, DELTA_BA: lengths
                             the offsets area has room for */
    uint32_t dst_cap;     /* DELTA_BA: bytes of room for the values at dst_off; else 0 */
} pq_page_desc;

typedef struct pq_page_result {
    int32_t status;      /* PQ_OK or the reference's error class */
    int32_t stage;       /* PQ_STAGE_*: pages.  Bare DICT, DELTA32/64, BOOL_RLE and byte-array
                            descriptors: PQ_STAGE_VALUES_INIT when the error is the decoder's Init's
                            (types/dict.go:66-89, types/int32.go:103-106, types/boolean.go:177-181),
                            PQ_STAGE_NONE when it is DecodeValues' */
    int32_t detail;      /* underlying status when status == PQ_PAGE_VALUES_FAILED */
    uint32_t n_decoded;  /* values written (what DecodeValues returns); levels: slots */
    uint32_t not_null;   /* LEVELS and pages: #{def == max_def} */
    uint32_t n_levels;   /* pages: ReadValues' n (0 on any error) */
    uint64_t consumed;   /* bytes of src the reference's reader would have consumed
                            (defined when status == PQ_OK) */
} pq_page_result;

/* ---- batched, device-resident decode --------------------------------------
 * A plan is an uploaded descriptor table plus a device-side result table.
 * pq_plan_run launches the decode kernels for every descriptor, asynchronously
 * on the context's stream, and touches no host memory: this is the path the
 * reference's readPageData loop (file-reader.go:268-302) maps to when all pages
 * of a column chunk are handed over at once.  Buffers are device pointers;
 * *_bytes are their sizes (used only for range checks).
 * pq_plan_create copies descs (the caller's array is free again on return) and
 * sends the tables on the context's stream without waiting for them; it waits
 * only when the plan before last is still being sent.  A plan belongs to its context: destroy it
 * before the context.  pq_ctx_destroy detaches the plans still alive (frees their device tables):
 * such a plan can only be handed to pq_plan_destroy any more, pq_plan_run and pq_plan_results
 * return PQ_ERR_BAD_ARGUMENT for it.  The context keeps the device block of the largest plan
 * destroyed so far for the next plan to use, and frees it in pq_ctx_destroy. */
typedef struct pq_plan pq_plan;

int pq_plan_create(pq_ctx *ctx, const pq_page_desc *descs, size_t n_pages, pq_plan **out);
int pq_plan_run(pq_plan *plan, const void *d_src, size_t src_bytes, void *d_values,
                size_t values_bytes, void *d_levels, size_t levels_bytes, const void *d_dict,
                size_t dict_bytes);
/* Waits for the stream and copies the per-page results to host memory. */
int pq_plan_results(pq_plan *plan, pq_page_result *results);

/* One plan over the pages of MANY column chunks that lie in buffers of their own: the loop over a
 * row group's columns (readRowGroup, file-reader.go:156-191; one readPageData per chunk,
 * file-reader.go:268-302) as a single set of launches, so that chunks of a few hundred pages each
 * fill the machine together instead of one after the other.  A buffer set is what pq_plan_run takes
 * for one chunk; descriptors first_desc[k] .. first_desc[k + 1] - 1 belong to set k and their
 * offsets are relative to that set's buffers (first_desc: n_sets + 1 ascending entries from 0 to
 * n_pages — the table pq_chunks_page_descs fills).  pq_plan_run_sets takes the sets in the same
 * order, checks each against what its descriptors reach, and launches once; results are in
 * descriptor order as for any plan.  pq_plan_run refuses such a plan and pq_plan_run_sets one made
 * by pq_plan_create with n_sets != 1 (PQ_ERR_BAD_ARGUMENT). */
typedef struct pq_buffer_set {
    const void *d_src;
    size_t src_bytes;
    void *d_values;
    size_t values_bytes;
    void *d_levels;
    size_t levels_bytes;
    const void *d_dict;
    size_t dict_bytes;
} pq_buffer_set;
int pq_plan_create_sets(pq_ctx *ctx, const pq_page_desc *descs, size_t n_pages, const uint32_t *first_desc,
                        size_t n_sets, pq_plan **out);
int pq_plan_run_sets(pq_plan *plan, const pq_buffer_set *sets, size_t n_sets);
/* Diagnostics (no counterpart in the reference): how the last pq_plan_run treated the delta pages
 * it cut into segments (PQ_OPT_SEGMENT_MAX_PAGES).  One record per segment, pages in plan order,
 * a page's segments in stream order; *n_segments = how many there are, at most `cap` are copied
 * (segs may be NULL with cap 0).  Waits for the stream. */
typedef struct pq_segment_info {
    uint32_t key;      /* where the segment's scan took its first block's four width bytes to be (stream
                          offset; 0xffffffff: found none) */
    uint32_t exit;     /* header of the next segment's first block as its walk of whole blocks reached it;
                          0xffffffff: it did not */
    uint32_t n_blocks; /* blocks walked */
    uint32_t exit_key; /* width bytes of the header at `exit` */
    uint64_t sum;      /* sum of the walked deltas (without the first block's minDelta, but in segment 0) */
    uint32_t exit_min_delta;
    uint32_t decode;   /* 0: decoded by a segment in front of it, serially; 1: by a wave of its own, up to
                          the next such segment; 2: by a wave of its own, to the end of the page;
                          3: the whole page by one wave (its segment 0) */
    uint64_t value;    /* decode 1, 2: the value at position `position` */
    uint32_t entry;    /* decode 1, 2: stream offset of the block header the segment starts at */
    uint32_t position; /* decode 1, 2: index in the page of the segment's first value */
    uint32_t stop;     /* decode 1: stream offset of the block header it stops in front of */
    uint32_t reserved[3];
} pq_segment_info;
int pq_plan_segments(pq_plan *plan, pq_segment_info *segs, size_t cap, size_t *n_segments);
void pq_plan_destroy(pq_plan *plan);

/* ---- batched decode from host memory ---------------------------------------
 * Same as create + run + results with host buffers: copies src and dict to the
 * device, decodes, copies values/levels back.  Any of values/levels/dict may be
 * NULL when no descriptor uses it. */
int pq_decode_pages(pq_ctx *ctx, const pq_page_desc *descs, size_t n_pages, const void *src,
                    size_t src_bytes, void *values, size_t values_bytes, void *levels,
                    size_t levels_bytes, const void *dict, size_t dict_bytes,
                    pq_page_result *results);

/* ---- column chunks from host memory, pipelined ------------------------------------------------
 * The loop over a row group's column chunks (readRowGroup, file-reader.go:156-191) with each
 * chunk's pages handed over at once (readPageData, file-reader.go:268-302): one job = one chunk's
 * descriptor table and its host buffers, offsets relative to those buffers, as for
 * pq_decode_pages.  The jobs run through three pipeline slots of the context (a stream, device
 * staging and pinned host staging each), so that the copy to the device of one chunk, the decode of
 * another and the copy back of a third overlap; the call returns when every job is done and its
 * values, levels and results are in the caller's memory.
 * The caller's buffers may be ordinary pageable memory (what cgo hands over): the library copies
 * through pinned staging of its own with a few host threads.  PQ_CHUNKS_PINNED promises that src,
 * values, levels and dict of every job came from pq_host_alloc: they are then copied to and from
 * directly.  job.status: PQ_OK, or why this job's descriptors were refused (PQ_ERR_BAD_ARGUMENT,
 * PQ_ERR_OUT_OF_RANGE) — the other jobs still run; the per-page outcome is in job.results.  Returns
 * PQ_OK or PQ_ERR_HIP / PQ_ERR_BAD_ARGUMENT for the call as a whole.  Contexts with a stream of the
 * caller's: the slots use streams of their own, the caller's stream is not involved. */
typedef struct pq_chunk_job {
    const pq_page_desc *descs;
    size_t n_pages;
    const void *src;
    size_t src_bytes;
    void *values;
    size_t values_bytes;
    void *levels;
    size_t levels_bytes;
    const void *dict;
    size_t dict_bytes;
    pq_page_result *results; /* n_pages */
    int32_t status;          /* out */
    int32_t reserved;
} pq_chunk_job;
#define PQ_CHUNKS_PINNED 1u
int pq_decode_chunks(pq_ctx *ctx, pq_chunk_job *jobs, size_t n_jobs, uint32_t flags);
/* Pinned host memory for PQ_CHUNKS_PINNED (hipHostMalloc / hipHostFree). */
void *pq_host_alloc(pq_ctx *ctx, size_t bytes);
void pq_host_free(pq_ctx *ctx, void *p);

/* ---- single-stream conveniences (host memory), one call per reference call --
 * hybrid:  HybridDecoder.Init/InitSize + n x Next()   (encoding/hybrid-decoder.go:42-109)
 * dict:    DictDecoder.Init + DecodeValues(dest[:n])  (types/dict.go:66-117)
 * levels:  decodePackedArray(d, n)                    (layout/helpers.go:68-90);
 *          mode 0 = ConstDecoder, 1 = V1 length-prefixed, 2 = V2 raw
 * delta:   Int32/Int64DeltaBPDecoder.Init + DecodeValues(dest[:n])
 * Each returns the status; n_decoded / not_null / consumed may be NULL. */
int pq_hybrid_decode(pq_ctx *ctx, const uint8_t *src, size_t src_len, int w, int length_prefixed,
                     uint32_t n, int32_t *dst, uint32_t *n_decoded, uint64_t *consumed);
int pq_dict_decode(pq_ctx *ctx, const uint8_t *src, size_t src_len, const void *dict,
                   uint32_t dict_len, uint32_t elem_size, uint32_t n, void *dst,
                   uint32_t *n_decoded, uint64_t *consumed);
int pq_levels_decode(pq_ctx *ctx, const uint8_t *src, size_t src_len, uint32_t max_level, int mode,
                     uint32_t n, int32_t *levels, uint32_t *not_null, uint64_t *consumed);
int pq_delta_decode32(pq_ctx *ctx, const uint8_t *src, size_t src_len, uint32_t n, int32_t *dst,
                      uint32_t *n_decoded, uint64_t *consumed);
int pq_delta_decode64(pq_ctx *ctx, const uint8_t *src, size_t src_len, uint32_t n, int64_t *dst,
                      uint32_t *n_decoded, uint64_t *consumed);
/* plain:  Int32/Int64/Float/DoublePlainDecoder.DecodeValues(dest[:n]), elem_size 4 or 8
 * bool:   BooleanPlainDecoder.DecodeValues(dest[:n]) of a fresh decoder, int32 0 / 1 per value */
int pq_plain_decode(pq_ctx *ctx, const uint8_t *src, size_t src_len, uint32_t elem_size, uint32_t n,
                    void *dst, uint32_t *n_decoded, uint64_t *consumed);
int pq_bool_plain_decode(pq_ctx *ctx, const uint8_t *src, size_t src_len, uint32_t n, int32_t *dst,
                         uint32_t *n_decoded, uint64_t *consumed);
/* ByteArrayDeltaLengthDecoder.Init(reader) + DecodeValues(dest[:n]) (types/byte-array.go:168-219).
 * offsets: n + 1 entries, value i is data[offsets[i] .. offsets[i + 1]); data: src_len bytes.
 * *init_failed (may be NULL) tells an error of Init from one of DecodeValues. */
int pq_delta_length_ba_decode(pq_ctx *ctx, const uint8_t *src, size_t src_len, uint32_t n,
                              uint32_t *offsets, uint8_t *data, uint32_t *n_decoded,
                              uint64_t *consumed, int *init_failed);
/* ByteArrayDeltaDecoder.Init(reader) + DecodeValues(dest[:n]) (types/byte-array.go:279-348).
 * offsets: n + 1 entries; data: data_cap bytes (< 2^32).  Returns PQ_ERR_OUT_OF_RANGE when the
 * values need more than data_cap bytes (prefix expansion has no bound from the stream's size):
 * the caller retries with a larger area. */
int pq_delta_ba_decode(pq_ctx *ctx, const uint8_t *src, size_t src_len, uint32_t n,
                       uint32_t *offsets, uint8_t *data, size_t data_cap, uint32_t *n_decoded,
                       uint64_t *consumed, int *init_failed);

/* ---- Page-header walk: ChunkReader.readPages (layout/chunk.go:129-217) -------------------
 *
 * Replaces, for every column chunk of a row group at once, the loop that reads one Thrift
 * compact PageHeader after the other (readThrift, layout/helpers.go:16-23; PageHeader.Read,
 * parquet/parquet.go:5886), applies the header checks of the page readers' read()
 * (layout/page-v1.go:40-67, page-v2.go:42-108, page-dict.go:28-57) and steps over the page block
 * (readBlockData, layout/block.go:17-47).  The chunk bytes are in device memory; the result is
 * one pq_page_info per page, from which pq_chunk_page_descs builds the descriptors of the data
 * pages.  Decompression stays with the caller's codec: for codec != 0 the walk reports where
 * each block lies and how large it is on either side. */
typedef struct pq_chunk_desc {
    uint64_t src_off;              /* offset in the source buffer of the chunk's first page header: the
                                      byte at file offset dictionary_page_offset when present, else
                                      data_page_offset (layout/chunk.go:58-61) */
    uint64_t src_len;              /* bytes the source holds from there on (to the end of the file) */
    int64_t total_compressed_size; /* ColumnMetaData.TotalCompressedSize */
    int64_t data_page_offset;      /* ColumnMetaData.DataPageOffset */
    int64_t dictionary_page_offset;
    uint32_t has_dictionary_page_offset;
    uint32_t physical_type;        /* parquet.Type: 0 BOOLEAN 1 INT32 2 INT64 3 INT96 4 FLOAT 5 DOUBLE
                                      6 BYTE_ARRAY 7 FIXED_LEN_BYTE_ARRAY */
    int32_t type_length;           /* SchemaElement.TypeLength, < 0 when nil */
    uint32_t max_rep;
    uint32_t max_def;
    uint32_t codec;                /* parquet.CompressionCodec; 0 = UNCOMPRESSED */
    uint32_t info_first;           /* index of this chunk's first record in the info table */
    uint32_t info_cap;             /* records it may write */
} pq_chunk_desc;

typedef struct pq_page_info {
    int32_t type;              /* parquet.PageType: 0 DATA_PAGE, 2 DICTIONARY_PAGE, 3 DATA_PAGE_V2 */
    int32_t encoding;          /* parquet.Encoding of the values */
    int32_t num_values;
    int32_t compressed_size;   /* PageHeader.CompressedPageSize */
    int32_t uncompressed_size; /* PageHeader.UncompressedPageSize */
    int32_t def_len;           /* V2: DefinitionLevelsByteLength */
    int32_t rep_len;           /* V2: RepetitionLevelsByteLength */
    int32_t def_encoding;      /* V1 */
    int32_t rep_encoding;      /* V1 */
    uint32_t is_compressed;    /* V2 (true when absent) */
    uint64_t header_off;       /* offset from the chunk's src_off of the header's first byte */
    uint64_t data_off;         /* ... of the first byte behind the header */
} pq_page_info;

typedef struct pq_chunk_result {
    int32_t status;   /* PQ_OK, a PQ_WALK_* class, or PQ_ERR_OUT_OF_RANGE (more pages than info_cap) */
    uint32_t n_pages; /* records written (the pages in front of the failing one) */
    int64_t count;    /* offsetReader.Count() when the loop ended (layout/helpers.go:38-64) */
} pq_chunk_result;

/* d_src: device memory.  d_infos: device memory for the info table.  results: host memory, one
 * per chunk; the call waits for the stream. */
int pq_walk_chunks_dev(pq_ctx *ctx, const void *d_src, size_t src_bytes, const pq_chunk_desc *chunks,
                       size_t n_chunks, pq_page_info *d_infos, size_t infos_cap,
                       pq_chunk_result *results);
/* Same from host memory; infos: host memory for infos_cap records. */
int pq_walk_chunks(pq_ctx *ctx, const void *src, size_t src_bytes, const pq_chunk_desc *chunks,
                   size_t n_chunks, pq_page_info *infos, size_t infos_cap, pq_chunk_result *results);

/* Where the outputs of a walked chunk go (all offsets in bytes, all advanced by the call). */
typedef struct pq_chunk_layout {
    uint64_t values_off;  /* values buffer */
    uint64_t levels_off;  /* levels buffer */
    uint64_t dict_off;    /* dictionary buffer: the chunk's dictionary is expected at the 8-byte aligned
                             offset the call reports in *dict_dst, copied from src[*dict_src ..) */
    uint32_t delta_ba_room; /* DELTA_BYTE_ARRAY pages: bytes of room per page for the rebuilt values */
} pq_chunk_layout;

/* Host-side bookkeeping, no device work: descriptors (PQ_KIND_PAGE_V1 / _V2) for the data pages
 * of an UNCOMPRESSED chunk, in page order, as getValuesDecoder pairs decoders with encodings
 * (layout/page-dict.go:143-345).  descs: room for n_pages.  *dict_src / *dict_dst / *dict_bytes:
 * the PLAIN fixed-width dictionary body the caller copies (device to device) before running the
 * plan; *dict_bytes == 0 when the chunk has none.  Returns PQ_ERR_BAD_ARGUMENT for a compressed
 * chunk and for types whose values this library does not decode (INT96, FIXED_LEN_BYTE_ARRAY,
 * PLAIN byte arrays, byte-array and boolean dictionaries). */
int pq_chunk_page_descs(const pq_chunk_desc *chunk, const pq_page_info *infos, uint32_t n_pages,
                        pq_chunk_layout *layout, pq_page_desc *descs, uint32_t *n_descs,
                        uint64_t *dict_src, uint64_t *dict_dst, uint64_t *dict_bytes);

/* The same for every walked chunk of a row group in one call.  infos: the table pq_walk_chunks
 * filled (chunk i's records at chunks[i].info_first, results[i].n_pages of them).  descs: room for
 * descs_cap descriptors; first_desc: n_chunks + 1 entries, chunk i's descriptors are
 * descs[first_desc[i] .. first_desc[i + 1]); copies: room for n_chunks dictionary copies (source
 * offset in src, destination offset in the dictionary buffer, bytes).
 * Chunks are judged one by one: chunk_status[i] (n_chunks entries, may be NULL) is PQ_OK, the
 * status of a failed walk (results[i].status), or what pq_chunk_page_descs returned for the chunk
 * (PQ_ERR_BAD_ARGUMENT for compressed chunks and the types this library does not decode, such as
 * a PLAIN BYTE_ARRAY column; PQ_UNEXPECTED_EOF for a dictionary page with too few bytes).  A chunk
 * whose status is not PQ_OK gets no descriptors (first_desc[i + 1] == first_desc[i]), no dictionary
 * copy and takes no room in `layout`; the other chunks are laid out as if it were not there, and the
 * caller reads that column through the reference's own path.  first_desc[n_chunks] and *n_copies
 * are always written.  Returns PQ_OK, PQ_ERR_BAD_ARGUMENT for missing arguments, or
 * PQ_ERR_OUT_OF_RANGE when the descriptors do not fit descs_cap (the chunk that did not fit and
 * all behind it carry that status and have no descriptors). */
typedef struct pq_dict_copy {
    uint64_t src;
    uint64_t dst;
    uint64_t bytes;
} pq_dict_copy;

int pq_chunks_page_descs(const pq_chunk_desc *chunks, const pq_chunk_result *results, size_t n_chunks,
                         const pq_page_info *infos, pq_chunk_layout *layout, pq_page_desc *descs,
                         size_t descs_cap, uint32_t *first_desc, pq_dict_copy *copies, uint32_t *n_copies,
                         int32_t *chunk_status);

/* ---- Level post-processing: what the reference does with decoded levels, value by value ----
 *
 * pq_levels_pack*: the bytes of an encoding.PackedArray (encoding/packed-array.go:47-55,118-125,
 * pack8int32_w of encoding/bitpacking32.go:45-79) holding levels[0..n): AppendSingle x n and, for
 * n > 0, Flush — ceil(n / 8) groups of bit_width bytes, value i of a group in bits
 * [i w, (i + 1) w) of it.  This is the container readPageData appends page levels to
 * (file-reader.go:285-296).  bit_width = bits.Len16(max level), 0..16; levels must be below
 * 2^bit_width (what decodePackedArray yields).  out_bytes >= ceil(n / 8) * bit_width; d_out 4-byte
 * aligned.
 *
 * pq_column_entries*: ColumnStore.Get (datastore/column.go:127-177, reached from Column.GetData,
 * schema/column.go:132-162) called until the store is exhausted, over a column whose level arrays
 * are def[0..n) / rep[0..n).  Entry k is the k-th call:
 *   validity bit k (LSB-first)  the call returned a value (REPEATED: the array of values collected
 *                               while the repetition level stays >= max_rep), not nil
 *   entry_def[k]                the definition level it returned (max_def for a value)
 *   offsets[k]                  values handed out by GetNext before it; offsets[n_entries] = all
 * so entry k owns values[offsets[k] .. offsets[k + 1]) of the column's dense value array.  rep is
 * read only when repeated != 0.  Capacities: offsets n + 1, validity 4 * ceil(n / 32) bytes
 * (written whole), entry_def n.  n <= 2^31 - 1.
 * The _dev forms take device pointers (result included) and run asynchronously on the context's
 * stream; their scratch memory belongs to the context and grows on demand. */
typedef struct pq_entries_result {
    int32_t status;     /* PQ_OK */
    uint32_t n_entries; /* calls that returned without "out of range" */
    uint32_t n_values;  /* values they consumed */
    uint32_t reserved;
} pq_entries_result;

int pq_levels_pack_dev(pq_ctx *ctx, const int32_t *d_levels, uint64_t n, int bit_width, void *d_out,
                       uint64_t out_bytes);
int pq_levels_pack(pq_ctx *ctx, const int32_t *levels, uint64_t n, int bit_width, uint8_t *out,
                   uint64_t out_bytes);
int pq_column_entries_dev(pq_ctx *ctx, const int32_t *d_def, const int32_t *d_rep, uint64_t n,
                          uint32_t max_def, uint32_t max_rep, int repeated, uint32_t *d_offsets,
                          void *d_validity, int32_t *d_entry_def, pq_entries_result *d_result);
/* pq_column_entries_dev for callers that cannot allocate (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same
 * arguments, same outputs; the workspace is a static device array inside the library (8 MiB on
 * every GPU the library runs on, held until the process ends), so the call allocates nothing and
 * frees nothing.  n <= PQ_NOALLOC_MAX_SLOTS, else PQ_ERR_OUT_OF_RANGE.  All calls on one GPU share
 * that array, whatever their context or stream, so the library puts them in order: a call's
 * kernels start when those of the call before it on that GPU are done (an event wait on the
 * caller's stream; a call never blocks the host).  A stream that is being captured into a HIP
 * graph is exempt (events recorded outside a capture cannot be waited for inside it): the recorded
 * launches are ordered by the graph, and a replay must not overlap another call on the same GPU —
 * overlapping calls do not merely race, the earlier one's workgroups can wait for ever on records
 * the later one cleared.  The launches hold no memset nodes, only kernels. */
#define PQ_NOALLOC_MAX_SLOTS 30000000ull
int pq_column_entries_noalloc(pq_ctx *ctx, const int32_t *d_def, const int32_t *d_rep, uint64_t n,
                              uint32_t max_def, uint32_t max_rep, int repeated, uint32_t *d_offsets,
                              void *d_validity, int32_t *d_entry_def, pq_entries_result *d_result);
int pq_column_entries(pq_ctx *ctx, const int32_t *def, const int32_t *rep, uint64_t n, uint32_t max_def,
                      uint32_t max_rep, int repeated, uint32_t *offsets, uint8_t *validity,
                      int32_t *entry_def, pq_entries_result *result);

#ifdef __cplusplus
}
#endif
#endif /* PQDECODE_H */
