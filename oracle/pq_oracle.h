/*
 * pq_oracle.h — CPU oracle for the Parquet page-value decode path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a scalar, value-at-a-time C restatement of
 * the decoding algorithm of hexbee-net/parquet-go (Go), used as the checker in
 * tests/, in the smoke test and as the timed "port" CPU baseline of bench.py.
 * Nothing in the shipped library links, loads or calls it.
 *
 * Parity pinning: every function below is checked against the reference's own
 * known-answer vectors (the JSON files under tests/golden, transcribed from
 * encoding/bitpacking32_test.go, encoding/bitpacking64_test.go,
 * encoding/hybrid-decoder_test.go, encoding/hybrid-encoder_test.go,
 * encoding/delta-binary-packed-decoder_test.go,
 * encoding/delta-binary-packed-encoder_test.go, encoding/packed-array_test.go,
 * types/float_test.go, types/double_test.go).
 * The reference has no tests for types.DictDecoder, layout.decodePackedArray,
 * the V1/V2 page section order, multi-block delta streams, multi-run hybrid
 * decoding, the int32 / int64 PLAIN decoders, the BOOLEAN decoders,
 * ColumnStore.Get, the two byte-array decoders of types/byte-array.go or the
 * page-header loop of layout/chunk.go: for those the oracle follows the
 * reference source line by line and its parity is UNPINNED by any reference
 * fixture.  The Go toolchain is not available, so the reference itself was
 * never executed.  The Thrift compact reader behind the page-header loop is a
 * dependency the reference does not vendor; it is restated from the published
 * encoding (see pqo_walk_chunk).  What the tests add besides: Parquet files
 * written by pyarrow, read back through the oracle and compared with what
 * pyarrow reads (tests/test_walk_oracle.py) - an independent check of the wire
 * formats, not of the reference's quirks.
 */
#ifndef PQ_ORACLE_H
#define PQ_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Error classes of the reference (numbering is shared with include/pqdecode.h
 * on purpose so tests can compare codes directly). */
enum pqo_status {
    PQO_OK = 0,
    PQO_EOF = 1,                     /* io.EOF */
    PQO_UNEXPECTED_EOF = 2,          /* io.ErrUnexpectedEOF */
    PQO_EMPTY_BP_RUN = 3,            /* "rle: empty bit-packed run"     hybrid-decoder.go:123 */
    PQO_EMPTY_RLE_RUN = 4,           /* "rle: empty RLE run"            hybrid-decoder.go:130 */
    PQO_RLE_VALUE_TOO_LARGE = 5,     /* "rle: RLE run value is too large" hybrid-decoder.go:166 */
    PQO_INT32_OUT_OF_RANGE = 6,      /* "int32 out of range"            helpers.go:36,54 */
    PQO_VARINT_OVERFLOW = 7,         /* encoding/binary: varint overflows a 64-bit integer */
    PQO_INVALID_BLOCK_SIZE = 8,      /* delta-binary-packed-decoder.go:34 */
    PQO_INVALID_MINIBLOCK_COUNT = 9, /* delta-binary-packed-decoder.go:46 */
    PQO_INVALID_MINIBLOCK_WIDTH = 10,/* delta-binary-packed-decoder.go:80 */
    PQO_DICT_INVALID_WIDTH = 11,     /* types/dict.go:75 */
    PQO_DICT_INVALID_INDEX = 12,     /* types/dict.go:104 (also the negative-key panic) */
    PQO_NOT_INITIALIZED = 13,        /* "reader is not initialized"     hybrid-decoder.go:79 */
    PQO_INVALID_STREAM = 14,         /* "invalid stream"                delta-binary-packed-decoder.go:154 */
    PQO_PAGE_VALUES_FAILED = 15,     /* "read values from page failed"  page-v1.go:104, page-v2.go:137 */
    PQO_BYTE_ARRAY_NEGATIVE_LEN = 16,/* make([]byte, size) with size < 0 panics, types/byte-array.go:210,326 */
    PQO_BYTE_ARRAY_COUNT_MISMATCH = 17, /* "bytearray/delta: different number of suffixes and prefixes" byte-array.go:303 */
    PQO_BYTE_ARRAY_INVALID_PREFIX = 18, /* "invalid prefix len in the stream"  byte-array.go:331 */
    PQO_OUT_OF_RANGE_CAP = 103       /* not the reference's: the caller's output area is too small */
};

/* Stage of a data-page read at which an error surfaced (page-v1.go, page-v2.go). */
enum pqo_stage {
    PQO_STAGE_NONE = 0,
    PQO_STAGE_REP_INIT = 1,
    PQO_STAGE_DEF_INIT = 2,
    PQO_STAGE_VALUES_INIT = 3,
    PQO_STAGE_REP = 4,
    PQO_STAGE_DEF = 5,
    PQO_STAGE_VALUES = 6
};

enum pqo_value_kind {
    PQO_VALUES_DICT = 1,    /* types.DictDecoder, elem_size bytes per dictionary entry */
    PQO_VALUES_DELTA32 = 2, /* types.Int32DeltaBPDecoder */
    PQO_VALUES_DELTA64 = 3, /* types.Int64DeltaBPDecoder */
    PQO_VALUES_PLAIN = 4,   /* Int32/Int64/Float/DoublePlainDecoder, elem_size bytes per value */
    PQO_VALUES_BOOL_PLAIN = 5, /* BooleanPlainDecoder, one int32 0/1 per value */
    PQO_VALUES_BOOL_RLE = 6,   /* BooleanRLEDecoder (types/boolean.go:173-196), one int32 0/1 per value */
    PQO_VALUES_DELTA_LENGTH_BA = 7,/* ByteArrayDeltaLengthDecoder (types/byte-array.go:168-219); values =
                                      (num_values + 1) uint32 offsets followed by the value bytes */
    PQO_VALUES_DELTA_BA = 8        /* ByteArrayDeltaDecoder (types/byte-array.go:279-348); same layout,
                                      data_cap bytes of room behind the offsets */
};

/* ---- fixed-width bit packing (encoding/bitpacking32.go, bitpacking64.go) ---- */
void pqo_unpack8_32(const uint8_t *data, int w, int32_t out[8]);
void pqo_pack8_32(const int32_t in[8], int w, uint8_t *out);
void pqo_unpack8_64(const uint8_t *data, int w, int64_t out[8]);
void pqo_pack8_64(const int64_t in[8], int w, uint8_t *out);
/* n_groups consecutive groups, value-at-a-time group calls like the Go table dispatch */
void pqo_bitunpack32(const uint8_t *src, int w, size_t n_groups, int32_t *dst);
void pqo_bitunpack64(const uint8_t *src, int w, size_t n_groups, int64_t *dst);

/* ---- hybrid RLE / bit-packed (encoding/hybrid-decoder.go) ----
 * length_prefixed: 1 = InitSize (u32 LE byte length first; no-op when w == 0),
 *                  0 = Init on the whole slice.
 * Pulls exactly n values with Next(); stops at the first error.
 * consumed = bytes of src taken by the decoder's reader when it stopped
 * (for the length-prefixed form: 4 + min(len, remaining) as InitSize slurps it). */
int pqo_hybrid_decode(const uint8_t *src, size_t src_len, int w, int length_prefixed,
                      uint32_t n, int32_t *dst, uint32_t *n_decoded, uint64_t *consumed);

/* ---- dictionary-index pages (types/dict.go:61-117) ----
 * src starts at the bit-width byte.  dict holds dict_len entries of elem_size bytes. */
int pqo_dict_decode(const uint8_t *src, size_t src_len, const void *dict, uint32_t dict_len,
                    uint32_t elem_size, uint32_t n, void *dst, uint32_t *n_decoded,
                    uint64_t *consumed);

/* ---- levels (layout/helpers.go:68-90, layout/chunk.go:78-124, encoding/packed-array.go) ----
 * mode: 0 = ConstDecoder(0) (reads nothing), 1 = V1 (length-prefixed hybrid),
 *       2 = V2 (raw hybrid over exactly src_len bytes; src_len == 0 leaves the
 *           decoder uninitialised as page-v2.go:92-102 does).
 * levels: n int32 (what Next() yields).  packed: PackedArray.data bytes
 * (capacity >= (n/8+1)*bw), packed_len its length, tail[8]/tail_len the
 * unflushed PackedArray.buf / bufPos. */
int pqo_levels_decode(const uint8_t *src, size_t src_len, uint32_t max_level, int mode,
                      uint32_t n, int32_t *levels, uint8_t *packed, uint64_t *packed_len,
                      int32_t tail[8], uint32_t *tail_len, uint32_t *not_null,
                      uint64_t *consumed);

/* ---- DELTA_BINARY_PACKED (encoding/delta-binary-packed-decoder.go) ----
 * Init + n pulls.  values_count = header's total value count (0 if the header failed).
 * init_status = result of Init alone (DecodeValues is attempted only if it is OK). */
int pqo_delta_decode32(const uint8_t *src, size_t src_len, uint32_t n, int32_t *dst,
                       uint32_t *n_decoded, uint64_t *consumed, uint32_t *values_count,
                       int *init_status);
int pqo_delta_decode64(const uint8_t *src, size_t src_len, uint32_t n, int64_t *dst,
                       uint32_t *n_decoded, uint64_t *consumed, uint32_t *values_count,
                       int *init_status);

/* ---- PLAIN fixed width (types/int32.go:56-72, types/int64.go:56-72, types/float.go:67-79,
 *      types/double.go:68-80) and BOOLEAN PLAIN (types/boolean.go:59-104; values as the
 *      int32 0/1 that unpackByte yields).  Pinned by the decoder vectors of
 *      types/float_test.go and types/double_test.go; the boolean decoder has no reference test. ---- */
int pqo_plain_decode(const uint8_t *src, size_t src_len, uint32_t elem_size, uint32_t n, void *dst,
                     uint32_t *n_decoded, uint64_t *consumed);
int pqo_bool_plain_decode(const uint8_t *src, size_t src_len, uint32_t n, int32_t *dst,
                          uint32_t *n_decoded, uint64_t *consumed);

/* ---- data pages (layout/page-v1.go:40-116, layout/page-v2.go:42-149) ---- */
typedef struct pqo_page_in {
    const uint8_t *src;      /* V1: decompressed block; V2: rep bytes, def bytes, values */
    uint64_t src_len;
    uint32_t version;        /* 1 or 2 */
    uint32_t num_values;     /* header NumValues == len(values) handed to ReadValues */
    uint32_t max_rep, max_def;
    uint32_t rep_len, def_len; /* V2 only */
    uint32_t value_kind;     /* enum pqo_value_kind */
    uint32_t elem_size;      /* dictionary entry bytes (DICT) */
    const void *dict;
    uint32_t dict_len;
    uint64_t data_cap;       /* DELTA_BA: bytes of room for the values behind the offsets */
} pqo_page_in;

typedef struct pqo_page_out {
    int32_t status;          /* enum pqo_status */
    int32_t stage;           /* enum pqo_stage */
    int32_t detail;          /* underlying status when status == PQO_PAGE_VALUES_FAILED */
    uint32_t n;              /* ReadValues' n */
    uint32_t not_null;
    uint32_t values_decoded; /* count returned by DecodeValues */
} pqo_page_out;

/* rep/def: num_values int32 each; values: num_values entries of the value type. */
void pqo_page_read(const pqo_page_in *in, int32_t *rep, int32_t *def, void *values,
                   pqo_page_out *out);

/* ---- DELTA_LENGTH_BYTE_ARRAY (types/byte-array.go:168-219): Init + DecodeValues(dest[:n]).
 * offsets: n + 1; data: src_len bytes.  *init_status != 0: Init failed (n_decoded 0). */
int pqo_delta_length_ba_decode(const uint8_t *src, size_t src_len, uint32_t n, uint32_t *offsets,
                               uint8_t *data, uint32_t *n_decoded, uint64_t *consumed,
                               uint32_t *values_count, int *init_status);

/* ---- DELTA_BYTE_ARRAY (types/byte-array.go:279-348): Init + DecodeValues(dest[:n]).
 * offsets: n + 1; data: data_cap bytes (PQO_OUT_OF_RANGE_CAP when the values do not fit).
 * "invalid prefix len" reports n_decoded 0 as the reference does (:330). */
int pqo_delta_ba_decode(const uint8_t *src, size_t src_len, uint32_t n, uint32_t *offsets,
                        uint8_t *data, uint64_t data_cap, uint32_t *n_decoded, uint64_t *consumed,
                        uint32_t *values_count, int *init_status);

/* ---- level post-processing ----
 * pqo_levels_pack: PackedArray bytes (encoding/packed-array.go:47-55,118-125) of n levels at
 * width w: AppendSingle x n, then Flush when n > 0.  out: ceil(n / 8) * w bytes.  Returns them. */
uint64_t pqo_levels_pack(const int32_t *levels, uint64_t n, int w, uint8_t *out);
/* pqo_column_entries: ColumnStore.Get (datastore/column.go:127-177) repeated to the end of the
 * store.  offsets: n + 1, validity: 4 * ceil(n / 32) bytes (LSB-first), entry_def: n.
 * rep is only read when repeated != 0. */
void pqo_column_entries(const int32_t *def, const int32_t *rep, uint64_t n, int32_t max_def,
                        int32_t max_rep, int repeated, uint32_t *offsets, uint8_t *validity,
                        int32_t *entry_def, uint32_t *n_entries, uint32_t *n_values);

/* ---- page-header walk of a column chunk: ChunkReader.readPages (layout/chunk.go:129-217) with the
 *      header checks of the page readers' read() (layout/page-v1.go:40-67, layout/page-v2.go:42-108,
 *      layout/page-dict.go:28-57) and layout/block.go:17-47 for the page block.  The Thrift compact
 *      protocol reader is github.com/apache/thrift v0.0.0-20181112125854-24918abba929 (go.mod:10), a
 *      dependency that is not vendored: restated from the published compact-protocol encoding
 *      (field header = id delta << 4 | type, zig-zag varints, binary = varint length + bytes,
 *      list/set header = size << 4 | type with 15 = varint size, map = varint size + type byte) and
 *      from the generated readers in parquet/parquet.go (:3994, :4334, :4586, :5886).  How the library
 *      treats malformed input beyond "an error" is unpinned; every header failure is one class here.
 *      Nothing in the reference tests this loop: parity unpinned. ---- */
enum pqo_walk_status {
    PQO_WALK_HEADER_FAILED = 19,          /* "failed to read page header"                 chunk.go:142 */
    PQO_WALK_SECOND_DICTIONARY = 20,      /* "there should be only one dictionary"        chunk.go:150 */
    PQO_WALK_PAGE_TYPE = 21,              /* "page type not supported"                    chunk.go:189 */
    PQO_WALK_MISSING_HEADER = 22,         /* "missing data page header" / dictionary      page-v1.go:42 (a nil
                                             dereference in init, :22, comes first), page-v2.go:46, page-dict.go:30 */
    PQO_WALK_NEGATIVE_NUM_VALUES = 23,    /* page-v1.go:45, page-v2.go:49, page-dict.go:33 */
    PQO_WALK_NEGATIVE_LEVELS_LENGTH = 24, /* page-v2.go:57-71 */
    PQO_WALK_DICT_ENCODING = 25,          /* page-dict.go:42 */
    PQO_WALK_LEVEL_ENCODING = 26,         /* "encoding not supported for definition and repetition level" chunk.go:79,95 */
    PQO_WALK_VALUE_ENCODING = 27,         /* getValuesDecoder / getDictValuesDecoder       page-dict.go:84-345 */
    PQO_WALK_BLOCK_SIZE = 28,             /* "invalid size for compressed data" block.go:23; page-v2.go:87-90 */
    PQO_WALK_UNCOMPRESSED_SIZE = 29,      /* "invalid size for decompressed data"         block.go:37 */
    PQO_WALK_SEEK = 30                    /* "failed to set the read index ..."            chunk.go:174 */
};

typedef struct pqo_chunk_in {
    const uint8_t *src;            /* the byte at file offset (dictionary_page_offset if present, else
                                      data_page_offset) and everything the source holds behind it */
    uint64_t src_len;
    int64_t total_compressed_size; /* ColumnMetaData.TotalCompressedSize */
    int64_t data_page_offset;
    int64_t dictionary_page_offset;
    uint32_t has_dictionary_page_offset;
    uint32_t physical_type;        /* parquet.Type: 0 BOOLEAN 1 INT32 2 INT64 3 INT96 4 FLOAT 5 DOUBLE
                                      6 BYTE_ARRAY 7 FIXED_LEN_BYTE_ARRAY */
    int32_t type_length;           /* SchemaElement.TypeLength, < 0 = nil */
    uint32_t max_rep, max_def;
    uint32_t codec;                /* 0 = UNCOMPRESSED: the block is the page; else sizes are not checked */
} pqo_chunk_in;

typedef struct pqo_page_info {
    int32_t type;              /* parquet.PageType: 0 DATA_PAGE, 2 DICTIONARY_PAGE, 3 DATA_PAGE_V2 */
    int32_t encoding;
    int32_t num_values;
    int32_t compressed_size;   /* PageHeader.CompressedPageSize */
    int32_t uncompressed_size;
    int32_t def_len, rep_len;  /* V2 */
    int32_t def_encoding, rep_encoding; /* V1 */
    uint32_t is_compressed;    /* V2 (default true) */
    uint64_t header_off;       /* offset in src of the header's first byte */
    uint64_t data_off;         /* offset in src of the first byte behind the header */
} pqo_page_info;

/* infos: room for cap records.  Returns PQO_OK or the first error (the pages in front of it are
 * reported).  *count = offsetReader.Count() when the loop ended.  More than cap pages:
 * PQO_OUT_OF_RANGE_CAP. */
int pqo_walk_chunk(const pqo_chunk_in *in, pqo_page_info *infos, uint32_t cap, uint32_t *n_pages,
                   int64_t *count);

/* ---- encoders, used only to generate inputs the reference itself would produce
 *      (encoding/hybrid-encoder.go, encoding/delta-binary-packed-encoder.go) ---- */
int pqo_hybrid_encode(const int32_t *values, size_t n, int w, uint8_t *out, size_t cap,
                      size_t *out_len);
int pqo_delta_encode32(const int32_t *values, size_t n, int block_size, int miniblocks,
                       uint8_t *out, size_t cap, size_t *out_len);
int pqo_delta_encode64(const int64_t *values, size_t n, int block_size, int miniblocks,
                       uint8_t *out, size_t cap, size_t *out_len);

#ifdef __cplusplus
}
#endif
#endif /* PQ_ORACLE_H */
