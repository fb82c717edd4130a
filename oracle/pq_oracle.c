/*
 * pq_oracle.c — CPU oracle (TEST INFRASTRUCTURE ONLY; see pq_oracle.h).
 *
 * A scalar, value-at-a-time restatement of the page-value decoders of
 * hexbee-net/parquet-go.  Structure deliberately follows the reference's
 * pull-based Next() design so that (a) every quirk falls out of the same
 * control flow and (b) timing it is an honest stand-in for the Go code.
 * Each function cites the reference file:line it follows (paths relative to
 * the reference root).
 */
#include "pq_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* In-memory reader: models bytes.Reader (layout/block.go:46 hands one to every
 * decoder) including the Read / io.ReadFull / ReadByte return conventions.    */

typedef struct {
    const uint8_t *p;
    size_t len;
    size_t pos;
} rd_t;

/* bytes.Reader.Read: at end -> (0, EOF) even for an empty buffer; otherwise
 * copies what is there and reports no error. */
static int rd_read(rd_t *r, uint8_t *b, size_t n, size_t *got)
{
    if (r->pos >= r->len) {
        *got = 0;
        return PQO_EOF;
    }
    size_t k = r->len - r->pos;
    if (k > n) k = n;
    memcpy(b, r->p + r->pos, k);
    r->pos += k;
    *got = k;
    return PQO_OK;
}

/* io.ReadFull: n == 0 -> ok; nothing read -> EOF; short -> ErrUnexpectedEOF. */
static int rd_read_full(rd_t *r, uint8_t *b, size_t n)
{
    if (n == 0) return PQO_OK;
    size_t avail = r->len - r->pos;
    if (avail == 0) return PQO_EOF;
    if (avail < n) {
        if (b) memcpy(b, r->p + r->pos, avail);
        r->pos = r->len;
        return PQO_UNEXPECTED_EOF;
    }
    if (b) memcpy(b, r->p + r->pos, n);
    r->pos += n;
    return PQO_OK;
}

static int rd_byte(rd_t *r, uint8_t *b)
{
    if (r->pos >= r->len) return PQO_EOF;
    *b = r->p[r->pos++];
    return PQO_OK;
}

/* encoding/binary.ReadUvarint as used by encoding/helpers.go:30,48,66,80:
 * 7 bits per byte, at most 10 bytes, the 10th may only be 0 or 1.  End of input
 * anywhere inside the varint is io.EOF (the behaviour the comment at
 * encoding/hybrid-decoder.go:114-115 relies on).
 * Which Go: the standard library's ReadUvarint from Go 1.15 on, which stops after
 * MaxVarintLen64 = 10 bytes and reports overflow there.  The reference's go.mod names Go 1.14,
 * whose ReadUvarint goes on reading continuation bytes and reports overflow only at the byte
 * that ends the varint; the two differ in status and in bytes consumed for inputs of 11 and
 * more bytes with the top bit set (for example eleven 0x80 and then the end of input: io.EOF
 * after 11 bytes there, overflow after 10 here).  Any toolchain the reference builds with today
 * has the newer routine, so that is the one restated here and in the device code. */
static int read_uvarint(rd_t *r, uint64_t *out)
{
    uint64_t x = 0;
    unsigned s = 0;
    for (int i = 0; i < 10; i++) {
        uint8_t b;
        if (rd_byte(r, &b) != PQO_OK) return PQO_EOF;
        if (b < 0x80) {
            if (i == 9 && b > 1) return PQO_VARINT_OVERFLOW;
            *out = x | ((uint64_t)b << s);
            return PQO_OK;
        }
        x |= (uint64_t)(b & 0x7f) << s;
        s += 7;
    }
    return PQO_VARINT_OVERFLOW;
}

/* encoding/helpers.go:24-40 */
static int read_uvarint32(rd_t *r, int32_t *out)
{
    uint64_t u;
    int st = read_uvarint(r, &u);
    if (st != PQO_OK) return st;
    if (u > 0x7fffffffULL) return PQO_INT32_OUT_OF_RANGE;
    *out = (int32_t)u;
    return PQO_OK;
}

/* encoding/binary.ReadVarint: zig-zag on top of ReadUvarint (helpers.go:74-81) */
static int read_varint64(rd_t *r, int64_t *out)
{
    uint64_t u;
    int st = read_uvarint(r, &u);
    if (st != PQO_OK) return st;
    int64_t x = (int64_t)(u >> 1);
    if (u & 1) x = ~x;
    *out = x;
    return PQO_OK;
}

/* encoding/helpers.go:42-58 */
static int read_varint32(rd_t *r, int32_t *out)
{
    int64_t x;
    int st = read_varint64(r, &x);
    if (st != PQO_OK) return st;
    if (x > 2147483647LL || x < -2147483648LL) return PQO_INT32_OUT_OF_RANGE;
    *out = (int32_t)x;
    return PQO_OK;
}

/* ------------------------------------------------------------------------- */
/* Fixed-width bit packing.  encoding/bitpacking32.go:81-1154 and
 * encoding/bitpacking64.go:145-3314 are per-width unrolled byte expressions of
 * one closed form: with S the little-endian integer made of the group's w
 * bytes, value k is bits [k*w, (k+1)*w) of S, zero-extended.                  */

static inline uint64_t load_le64(const uint8_t *p)
{
    uint64_t v;
    memcpy(&v, p, 8);
    return v; /* host is little-endian (x86-64) */
}

void pqo_unpack8_32(const uint8_t *data, int w, int32_t out[8])
{
    uint8_t g[48];
    memset(g, 0, sizeof g);
    if (w > 0) memcpy(g, data, (size_t)w);
    uint64_t mask = (w >= 32) ? 0xffffffffULL : ((1ULL << w) - 1);
    for (int k = 0; k < 8; k++) {
        unsigned bit = (unsigned)(k * w);
        uint64_t v = load_le64(g + (bit >> 3)) >> (bit & 7);
        out[k] = (int32_t)(uint32_t)(v & mask);
    }
}

void pqo_unpack8_64(const uint8_t *data, int w, int64_t out[8])
{
    uint8_t g[88];
    memset(g, 0, sizeof g);
    if (w > 0) memcpy(g, data, (size_t)w);
    uint64_t mask = (w >= 64) ? ~0ULL : ((1ULL << w) - 1);
    for (int k = 0; k < 8; k++) {
        unsigned bit = (unsigned)(k * w);
        unsigned sh = bit & 7;
        const uint8_t *p = g + (bit >> 3);
        uint64_t v = load_le64(p) >> sh;
        if (sh) v |= (uint64_t)p[8] << (64 - sh);
        out[k] = (int64_t)(v & mask);
    }
}

/* pack8: byte j is the OR of every value whose field overlaps it, shifted into
 * place and truncated to 8 bits; the values are NOT masked first, so bits above
 * w leak into the higher bits of the same byte exactly as in the generated Go
 * (e.g. encoding/bitpacking32.go:141-147). */
void pqo_pack8_32(const int32_t in[8], int w, uint8_t *out)
{
    for (int j = 0; j < w; j++) {
        unsigned acc = 0;
        int lo = 8 * j, hi = 8 * j + 8;
        for (int k = 0; k < 8; k++) {
            int fs = k * w, fe = fs + w;
            if (fe <= lo || fs >= hi) continue;
            uint32_t v = (uint32_t)in[k];
            acc |= (fs >= lo) ? (v << (fs - lo)) : (v >> (lo - fs));
        }
        out[j] = (uint8_t)acc;
    }
}

void pqo_pack8_64(const int64_t in[8], int w, uint8_t *out)
{
    for (int j = 0; j < w; j++) {
        uint64_t acc = 0;
        int lo = 8 * j, hi = 8 * j + 8;
        for (int k = 0; k < 8; k++) {
            int fs = k * w, fe = fs + w;
            if (fe <= lo || fs >= hi) continue;
            uint64_t v = (uint64_t)in[k];
            acc |= (fs >= lo) ? (v << (fs - lo)) : (v >> (lo - fs));
        }
        out[j] = (uint8_t)acc;
    }
}

void pqo_bitunpack32(const uint8_t *src, int w, size_t n_groups, int32_t *dst)
{
    for (size_t g = 0; g < n_groups; g++) pqo_unpack8_32(src + g * (size_t)w, w, dst + 8 * g);
}

void pqo_bitunpack64(const uint8_t *src, int w, size_t n_groups, int64_t *dst)
{
    for (size_t g = 0; g < n_groups; g++) pqo_unpack8_64(src + g * (size_t)w, w, dst + 8 * g);
}

/* ------------------------------------------------------------------------- */
/* Hybrid RLE / bit-packed decoder: encoding/hybrid-decoder.go:13-187          */

typedef struct {
    rd_t own;  /* buffered copy (Init with buffered=true, hybrid-decoder.go:43-49) */
    rd_t *r;   /* NULL until Init ran (hybrid-decoder.go:78-80) */
    int w;
    int rle_size;
    int32_t bp[8];
    uint32_t rle_count;
    int32_t rle_value;
    uint32_t bp_count;
    uint8_t bp_pos;
} hybrid_t;

static void hybrid_new(hybrid_t *d, int w)
{
    memset(d, 0, sizeof *d);
    d->w = w;
    d->rle_size = (w + 7) / 8; /* hybrid-decoder.go:36 */
}

/* Init, unbuffered: the decoder shares the caller's reader (hybrid-decoder.go:50-52) */
static void hybrid_init_shared(hybrid_t *d, rd_t *src) { d->r = src; }

/* Init, buffered: ioutil.ReadAll(reader) then a private bytes.Reader (:43-49) */
static void hybrid_init_buffered(hybrid_t *d, rd_t *src, size_t limit)
{
    size_t avail = src->len - src->pos;
    if (limit > avail) limit = avail;
    d->own.p = src->p + src->pos;
    d->own.len = limit;
    d->own.pos = 0;
    src->pos += limit;
    d->r = &d->own;
}

/* InitSize (hybrid-decoder.go:57-68): no-op for w == 0; else u32 LE length, then
 * Init over io.LimitReader(reader, size).  buffered decides who owns the bytes. */
static int hybrid_init_size(hybrid_t *d, rd_t *src, int buffered)
{
    if (d->w == 0) return PQO_OK;
    uint8_t b[4];
    int st = rd_read_full(src, b, 4); /* binary.Read -> io.ReadFull */
    if (st != PQO_OK) return st;
    uint32_t size = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) |
                    ((uint32_t)b[3] << 24);
    (void)buffered; /* both forms see exactly min(size, remaining) bytes */
    hybrid_init_buffered(d, src, size);
    return PQO_OK;
}

/* decodeRLEValue, hybrid-decoder.go:172-187 */
static int32_t decode_rle_value(const uint8_t *v, int n)
{
    uint32_t x = 0;
    for (int i = 0; i < n; i++) x += (uint32_t)v[i] << (8 * i);
    return (int32_t)x;
}

static int leading_zeros32(uint32_t x) { return x ? __builtin_clz(x) : 32; }

/* readRLERunValue, hybrid-decoder.go:151-170 */
static int hybrid_read_rle_value(hybrid_t *d)
{
    uint8_t v[4] = {0, 0, 0, 0};
    size_t n;
    int st = rd_read(d->r, v, (size_t)d->rle_size, &n);
    if (st != PQO_OK) return st;
    if ((int)n != d->rle_size) return PQO_UNEXPECTED_EOF;
    d->rle_value = decode_rle_value(v, d->rle_size);
    if (leading_zeros32((uint32_t)d->rle_value) < 32 - d->w) return PQO_RLE_VALUE_TOO_LARGE;
    return PQO_OK;
}

/* readRunHeader, hybrid-decoder.go:111-136 */
static int hybrid_read_run_header(hybrid_t *d)
{
    int32_t h;
    int st = read_uvarint32(d->r, &h);
    if (st != PQO_OK) return st;
    if (h & 1) {
        d->bp_count = (uint32_t)(h >> 1);
        if (d->bp_count == 0) return PQO_EMPTY_BP_RUN;
        d->bp_pos = 0;
    } else {
        d->rle_count = (uint32_t)(h >> 1);
        if (d->rle_count == 0) return PQO_EMPTY_RLE_RUN;
        return hybrid_read_rle_value(d);
    }
    return PQO_OK;
}

/* readBitPackedRun, hybrid-decoder.go:138-149: Read, not ReadFull — a short
 * final group is zero-filled (pinned by hybrid-decoder_test.go:11-35). */
static int hybrid_read_bp_group(hybrid_t *d)
{
    uint8_t data[32];
    memset(data, 0, sizeof data);
    size_t n;
    int st = rd_read(d->r, data, (size_t)d->w, &n);
    if (st != PQO_OK) return st;
    pqo_unpack8_32(data, d->w, d->bp);
    return PQO_OK;
}

/* Next, hybrid-decoder.go:70-109 */
static int hybrid_next(hybrid_t *d, int32_t *out)
{
    if (d->w == 0) {
        *out = 0;
        return PQO_OK;
    }
    if (d->r == NULL) return PQO_NOT_INITIALIZED;
    if (d->rle_count == 0 && d->bp_count == 0 && d->bp_pos == 0) {
        int st = hybrid_read_run_header(d);
        if (st != PQO_OK) return st;
    }
    if (d->rle_count > 0) {
        *out = d->rle_value;
        d->rle_count--;
    } else if (d->bp_count > 0 || d->bp_pos > 0) {
        if (d->bp_pos == 0) {
            int st = hybrid_read_bp_group(d);
            if (st != PQO_OK) return st;
            d->bp_count--;
        }
        *out = d->bp[d->bp_pos];
        d->bp_pos = (uint8_t)((d->bp_pos + 1) % 8);
    } else {
        return PQO_EOF;
    }
    return PQO_OK;
}

int pqo_hybrid_decode(const uint8_t *src, size_t src_len, int w, int length_prefixed,
                      uint32_t n, int32_t *dst, uint32_t *n_decoded, uint64_t *consumed)
{
    rd_t r = {src, src_len, 0};
    hybrid_t d;
    hybrid_new(&d, w);
    int st = PQO_OK;
    uint32_t i = 0;
    if (length_prefixed)
        st = hybrid_init_size(&d, &r, 1);
    else
        hybrid_init_shared(&d, &r);
    if (st == PQO_OK) {
        for (; i < n; i++) {
            int32_t v;
            st = hybrid_next(&d, &v);
            if (st != PQO_OK) break;
            dst[i] = v;
        }
    }
    if (n_decoded) *n_decoded = i;
    if (consumed) *consumed = r.pos;
    return st;
}

/* ------------------------------------------------------------------------- */
/* types.DictDecoder: types/dict.go:61-117                                     */

typedef struct {
    hybrid_t keys;
    int have_keys;
} dict_dec_t;

/* Init, types/dict.go:66-89 */
static int dict_init(dict_dec_t *d, rd_t *src)
{
    d->have_keys = 0;
    uint8_t b;
    int st = rd_read_full(src, &b, 1);
    if (st != PQO_OK) return st;
    int w = b;
    if (w > 32) return PQO_DICT_INVALID_WIDTH;
    hybrid_new(&d->keys, w);
    hybrid_init_shared(&d->keys, src); /* NewHybridDecoder(w, false).Init(reader) */
    d->have_keys = 1;
    return PQO_OK;
}

/* DecodeValues, types/dict.go:91-117.  key >= size -> (0, "invalid index").
 * A negative key (w == 32, top bit set) passes that test in Go and then panics
 * on the slice index; it is reported here as the same invalid-index error. */
static int dict_decode_values(dict_dec_t *d, const uint8_t *dict, uint32_t dict_len,
                              uint32_t elem, uint32_t n, uint8_t *dst, uint32_t *count)
{
    *count = 0;
    if (!d->have_keys) return PQO_NOT_INITIALIZED;
    int32_t size = (int32_t)dict_len;
    for (uint32_t i = 0; i < n; i++) {
        int32_t key;
        int st = hybrid_next(&d->keys, &key);
        if (st != PQO_OK) {
            *count = i;
            return st;
        }
        if (key >= size || key < 0) {
            *count = 0;
            return PQO_DICT_INVALID_INDEX;
        }
        memcpy(dst + (size_t)i * elem, dict + (size_t)key * elem, elem);
    }
    *count = n;
    return PQO_OK;
}

int pqo_dict_decode(const uint8_t *src, size_t src_len, const void *dict, uint32_t dict_len,
                    uint32_t elem_size, uint32_t n, void *dst, uint32_t *n_decoded,
                    uint64_t *consumed)
{
    rd_t r = {src, src_len, 0};
    dict_dec_t d;
    uint32_t cnt = 0;
    int st = dict_init(&d, &r);
    if (st == PQO_OK)
        st = dict_decode_values(&d, (const uint8_t *)dict, dict_len, elem_size, n,
                                (uint8_t *)dst, &cnt);
    if (n_decoded) *n_decoded = cnt;
    if (consumed) *consumed = r.pos;
    return st;
}

/* ------------------------------------------------------------------------- */
/* PackedArray (encoding/packed-array.go:16-129) and decodePackedArray
 * (layout/helpers.go:68-90) with the level decoder factories of
 * layout/chunk.go:78-124.                                                     */

static int bits_len16(uint32_t x) { return x ? 32 - __builtin_clz(x & 0xffffu) : 0; }

typedef struct {
    int bw;
    uint32_t count;
    uint8_t *data; /* caller storage */
    uint64_t data_len;
    int32_t buf[8];
    uint32_t buf_pos;
} packed_array_t;

static void packed_reset(packed_array_t *a, int bw, uint8_t *storage)
{
    memset(a, 0, sizeof *a);
    a->bw = bw;
    a->data = storage;
}

/* Flush, packed-array.go:118-125 */
static void packed_flush(packed_array_t *a)
{
    for (uint32_t i = a->buf_pos; i < 8; i++) a->buf[i] = 0;
    if (a->data) pqo_pack8_32(a->buf, a->bw, a->data + a->data_len);
    a->data_len += (uint64_t)a->bw;
    a->buf_pos = 0;
}

/* AppendSingle, packed-array.go:47-55: flushes the PREVIOUS full group first. */
static void packed_append(packed_array_t *a, int32_t v)
{
    if (a->buf_pos == 8) packed_flush(a);
    a->buf[a->buf_pos++] = v;
    a->count++;
}

/* A level decoder: HybridDecoder(bits.Len16(max), buffered=true) or
 * ConstDecoder(0) (encoding/const.go:7-19), wrapped with its max
 * (layout/level-decoder.go:5-18). */
typedef struct {
    int is_const;
    hybrid_t h;
    uint32_t max;
} level_dec_t;

static int level_next(level_dec_t *d, int32_t *v)
{
    if (d->is_const) {
        *v = 0;
        return PQO_OK;
    }
    return hybrid_next(&d->h, v);
}

/* decodePackedArray, layout/helpers.go:68-90 */
static int decode_packed_array(level_dec_t *d, uint32_t count, int32_t *levels,
                               packed_array_t *arr, uint8_t *storage, uint32_t *not_null)
{
    uint32_t nn = 0;
    packed_reset(arr, bits_len16(d->max), storage);
    for (uint32_t i = 0; i < count; i++) {
        int32_t u;
        int st = level_next(d, &u);
        if (st != PQO_OK) {
            *not_null = 0;
            return st;
        }
        packed_append(arr, u);
        if (levels) levels[i] = u;
        if (u == (int32_t)d->max) nn++;
    }
    *not_null = nn;
    return PQO_OK;
}

int pqo_levels_decode(const uint8_t *src, size_t src_len, uint32_t max_level, int mode,
                      uint32_t n, int32_t *levels, uint8_t *packed, uint64_t *packed_len,
                      int32_t tail[8], uint32_t *tail_len, uint32_t *not_null,
                      uint64_t *consumed)
{
    rd_t r = {src, src_len, 0};
    level_dec_t d;
    memset(&d, 0, sizeof d);
    d.max = max_level;
    int st = PQO_OK;
    if (mode == 0) {
        d.is_const = 1;
    } else {
        hybrid_new(&d.h, bits_len16(max_level));
        if (mode == 1) {
            st = hybrid_init_size(&d.h, &r, 1);
        } else if (src_len > 0) {
            hybrid_init_buffered(&d.h, &r, src_len);
        }
    }
    packed_array_t arr;
    uint32_t nn = 0;
    packed_reset(&arr, bits_len16(max_level), packed);
    if (st == PQO_OK) st = decode_packed_array(&d, n, levels, &arr, packed, &nn);
    if (packed_len) *packed_len = arr.data_len;
    if (tail) memcpy(tail, arr.buf, sizeof arr.buf);
    if (tail_len) *tail_len = arr.buf_pos;
    if (not_null) *not_null = nn;
    if (consumed) *consumed = r.pos;
    return st;
}

/* The bytes of a PackedArray after AppendSingle x n and, for n > 0, Flush
 * (packed-array.go:47-55,118-125): ceil(n / 8) groups of w bytes. */
uint64_t pqo_levels_pack(const int32_t *levels, uint64_t n, int w, uint8_t *out)
{
    packed_array_t a;
    packed_reset(&a, w, out);
    for (uint64_t i = 0; i < n; i++) packed_append(&a, levels[i]);
    if (n > 0) packed_flush(&a);
    return a.data_len;
}

/* ColumnStore.Get (datastore/column.go:127-177) called until it reports "out of range",
 * over a column store whose level arrays hold def[0..n) and rep[0..n) (what readPageData
 * appended, file-reader.go:285-296).  Entry k is the k-th call: validity bit k says it
 * returned a value (a scalar, or for a REPEATED column the array collected while the
 * repetition level stays >= maxR), entry_def[k] the definition level it returned,
 * offsets[k] how many values GetNext had handed out before it; offsets[n_entries] is the
 * total.  Every position of a value entry takes a value, whatever its own definition
 * level (column.go:161-176 calls GetNext without looking at it). */
void pqo_column_entries(const int32_t *def, const int32_t *rep, uint64_t n, int32_t max_def,
                        int32_t max_rep, int repeated, uint32_t *offsets, uint8_t *validity,
                        int32_t *entry_def, uint32_t *n_entries, uint32_t *n_values)
{
    uint64_t read_pos = 0;
    uint32_t k = 0, vals = 0;
    memset(validity, 0, (size_t)((n + 31) / 32 * 4));
    while (read_pos < n) {
        int32_t dl = def[read_pos];
        offsets[k] = vals;
        if (dl < max_def) { /* a null: advance the levels only (column.go:139-142) */
            entry_def[k++] = dl;
            read_pos++;
            continue;
        }
        vals++; /* GetNext, column.go:144 */
        validity[k >> 3] |= (uint8_t)(1u << (k & 7));
        entry_def[k++] = max_def;
        if (!repeated) { /* column.go:150-153 */
            read_pos++;
            continue;
        }
        for (;;) { /* column.go:161-176 */
            read_pos++;
            if (read_pos >= n || rep[read_pos] < max_rep) break;
            vals++;
        }
    }
    offsets[k] = vals;
    *n_entries = k;
    *n_values = vals;
}

/* ------------------------------------------------------------------------- */
/* DELTA_BINARY_PACKED: encoding/delta-binary-packed-decoder.go:11-322         */

typedef struct {
    rd_t *r;
    int is64;
    int32_t block_size, miniblock_count, values_count, mb_value_count;
    uint8_t *widths; /* miniblock_count entries */
    int32_t cur_mb;
    uint8_t cur_width;
    int32_t mb_pos;   /* miniBlockPosition */
    int32_t position;
    int64_t prev, min_delta;
    int64_t mini[8];
} delta_t;

static void delta_free(delta_t *d)
{
    free(d->widths);
    d->widths = NULL;
}

/* readBlockHeader, :29-65 */
static int delta_read_block_header(delta_t *d)
{
    int st = read_uvarint32(d->r, &d->block_size);
    if (st != PQO_OK) return st;
    if (d->block_size <= 0 || d->block_size % 128 != 0) return PQO_INVALID_BLOCK_SIZE;
    st = read_uvarint32(d->r, &d->miniblock_count);
    if (st != PQO_OK) return st;
    if (d->miniblock_count <= 0 || d->block_size % d->miniblock_count != 0)
        return PQO_INVALID_MINIBLOCK_COUNT;
    d->mb_value_count = d->block_size / d->miniblock_count;
    st = read_uvarint32(d->r, &d->values_count);
    if (st != PQO_OK) return st;
    if (d->values_count == 0) return PQO_OK;
    if (d->is64) { /* setPreviousValue64, :308-314 */
        return read_varint64(d->r, &d->prev);
    } else { /* setPreviousValue32, :232-238 */
        int32_t v = 0;
        st = read_varint32(d->r, &v);
        d->prev = v;
        return st;
    }
}

/* readMiniBlockHeader, :67-94 */
static int delta_read_miniblock_header(delta_t *d)
{
    int st;
    if (d->is64) {
        st = read_varint64(d->r, &d->min_delta);
    } else {
        int32_t v = 0;
        st = read_varint32(d->r, &v);
        d->min_delta = v;
    }
    if (st != PQO_OK) return st;
    if (!d->widths) {
        d->widths = (uint8_t *)malloc((size_t)d->miniblock_count);
        if (!d->widths) abort();
    }
    st = rd_read_full(d->r, d->widths, (size_t)d->miniblock_count);
    if (st != PQO_OK) return st;
    for (int32_t i = 0; i < d->miniblock_count; i++)
        if (d->widths[i] > 32) return PQO_INVALID_MINIBLOCK_WIDTH; /* both element sizes */
    d->cur_mb = 0;
    return PQO_OK;
}

/* Init, :183-207 and :259-283 */
static int delta_init(delta_t *d, rd_t *src, int is64)
{
    memset(d, 0, sizeof *d);
    d->r = src;
    d->is64 = is64;
    int st = delta_read_block_header(d);
    if (st != PQO_OK) return st;
    if (d->values_count == 0) return PQO_OK;
    return delta_read_miniblock_header(d);
}

/* advanceBlock, :130-147 */
static int delta_advance_block(delta_t *d)
{
    if (d->position % d->mb_value_count == 0) {
        if (d->cur_mb >= d->miniblock_count) {
            int st = delta_read_miniblock_header(d);
            if (st != PQO_OK) return st;
        }
        d->cur_width = d->widths[d->cur_mb];
        d->mb_pos = 0;
        d->cur_mb++;
    }
    return PQO_OK;
}

/* readPadding, :149-170 (including the widths[currentMiniBlock] indexing of :161) */
static int delta_read_padding(delta_t *d, int32_t w)
{
    if (d->position + 8 >= d->values_count) {
        int32_t l = (d->mb_value_count / 8) * w - d->mb_pos;
        if (l < 0) return PQO_INVALID_STREAM;
        (void)rd_read_full(d->r, NULL, (size_t)l);
        for (int32_t i = d->cur_mb; i < d->miniblock_count; i++) {
            int32_t bw = d->widths[d->cur_mb];
            if (bw != 0) (void)rd_read_full(d->r, NULL, (size_t)((d->mb_value_count / 8) * bw));
        }
    }
    return PQO_OK;
}

/* next, :96-128 */
static int delta_next_group(delta_t *d)
{
    if (d->position >= d->values_count) return PQO_EOF;
    if (d->position % 8 == 0) {
        int st = delta_advance_block(d);
        if (st != PQO_OK) return st;
        int32_t bw = d->cur_width;
        uint8_t buf[32];
        st = rd_read_full(d->r, buf, (size_t)bw);
        if (st != PQO_OK) return st;
        if (d->is64) {
            pqo_unpack8_64(buf, bw, d->mini); /* unpackMiniBlock64, :303-306 */
        } else {
            int32_t m[8];
            pqo_unpack8_32(buf, bw, m);       /* unpackMiniBlock32, :227-230 */
            for (int k = 0; k < 8; k++) d->mini[k] = m[k];
        }
        d->mb_pos += bw;
        st = delta_read_padding(d, bw);
        if (st != PQO_OK) return st;
    }
    return PQO_OK;
}

/* Next, :213-225 (int32, wrapping) and :289-301 (int64, wrapping) */
static int delta_next(delta_t *d, int64_t *out)
{
    int st = delta_next_group(d);
    if (st != PQO_OK) return st;
    *out = d->prev;
    if (d->is64) {
        d->prev = (int64_t)((uint64_t)d->prev + (uint64_t)d->mini[d->position % 8] +
                            (uint64_t)d->min_delta);
    } else {
        d->prev = (int32_t)((uint32_t)d->prev + (uint32_t)d->mini[d->position % 8] +
                            (uint32_t)d->min_delta);
    }
    d->position++;
    return PQO_OK;
}

static int delta_decode(const uint8_t *src, size_t src_len, int is64, uint32_t n, void *dst,
                        uint32_t *n_decoded, uint64_t *consumed, uint32_t *values_count,
                        int *init_status)
{
    rd_t r = {src, src_len, 0};
    delta_t d;
    uint32_t i = 0;
    int st = delta_init(&d, &r, is64);
    if (init_status) *init_status = st;
    if (values_count) *values_count = (st == PQO_OK) ? (uint32_t)d.values_count : 0;
    if (st == PQO_OK) {
        /* types/int32.go:108-123, types/int64.go:108-123: loop Next into dest */
        for (; i < n; i++) {
            int64_t v;
            st = delta_next(&d, &v);
            if (st != PQO_OK) break;
            if (is64)
                ((int64_t *)dst)[i] = v;
            else
                ((int32_t *)dst)[i] = (int32_t)v;
        }
    }
    if (n_decoded) *n_decoded = i;
    if (consumed) *consumed = r.pos;
    delta_free(&d);
    return st;
}

int pqo_delta_decode32(const uint8_t *src, size_t src_len, uint32_t n, int32_t *dst,
                       uint32_t *n_decoded, uint64_t *consumed, uint32_t *values_count,
                       int *init_status)
{
    return delta_decode(src, src_len, 0, n, dst, n_decoded, consumed, values_count, init_status);
}

int pqo_delta_decode64(const uint8_t *src, size_t src_len, uint32_t n, int64_t *dst,
                       uint32_t *n_decoded, uint64_t *consumed, uint32_t *values_count,
                       int *init_status)
{
    return delta_decode(src, src_len, 1, n, dst, n_decoded, consumed, values_count, init_status);
}

/* ------------------------------------------------------------------------- */
/* DELTA_LENGTH_BYTE_ARRAY: types/byte-array.go:168-219                          */

/* ByteArrayDeltaLengthDecoder.Init (:174-186) then DecodeValues(dest[:n]) (:188-219).
 * Init decodes ALL ValuesCount lengths (decodeInt32, types/helpers.go:22-33); any error
 * there fails Init.  next(): past the last length io.EOF; a negative length makes the
 * reference panic in make([]byte, size) (PQO_BYTE_ARRAY_NEGATIVE_LEN here); then
 * io.ReadFull of `size` bytes ("there is no byte left" wraps io.EOF / io.ErrUnexpectedEOF,
 * reported as that class).  offsets[0] = 0, offsets[i + 1] = offsets[i] + len(value i);
 * the value bytes are copied to data back to back. */
/* Int32DeltaBPDecoder.Init + decodeInt32 over make([]int32, ValuesCount) (types/byte-array.go:177-185,
 * :286-295; types/helpers.go:22-33): every declared length has to decode.  *lens is malloc'ed. */
static int ba_read_lens(rd_t *r, int32_t **lens, uint32_t *count)
{
    delta_t d;
    int st = delta_init(&d, r, 0);
    uint32_t vc = st == PQO_OK ? (uint32_t)d.values_count : 0;
    /* a stream cannot yield more lengths than this before it runs dry */
    size_t cap = (size_t)vc < 32 * r->len + 256 ? (size_t)vc : 32 * r->len + 256;
    *lens = (int32_t *)malloc((cap + 1) * sizeof **lens);
    if (!*lens) abort();
    for (uint32_t i = 0; st == PQO_OK && i < vc; i++) {
        int64_t v;
        st = delta_next(&d, &v);
        if (st == PQO_OK && i < cap) (*lens)[i] = (int32_t)v;
    }
    delta_free(&d);
    *count = vc;
    return st;
}

int pqo_delta_length_ba_decode(const uint8_t *src, size_t src_len, uint32_t n, uint32_t *offsets,
                               uint8_t *data, uint32_t *n_decoded, uint64_t *consumed,
                               uint32_t *values_count, int *init_status)
{
    rd_t r = {src, src_len, 0};
    int32_t *lens = NULL;
    uint32_t vc = 0;
    int st = ba_read_lens(&r, &lens, &vc);
    if (values_count) *values_count = vc;
    if (init_status) *init_status = st;
    offsets[0] = 0;
    uint32_t i = 0;
    if (st == PQO_OK) {
        for (; i < n; i++) {
            if (i >= vc) { st = PQO_EOF; break; }
            int32_t size = lens[i];
            if (size < 0) { st = PQO_BYTE_ARRAY_NEGATIVE_LEN; break; }
            st = rd_read_full(&r, data + offsets[i], (size_t)size);
            if (st != PQO_OK) break;
            offsets[i + 1] = offsets[i] + (uint32_t)size;
        }
    }
    free(lens);
    if (n_decoded) *n_decoded = i;
    if (consumed) *consumed = r.pos;
    return st;
}

/* ------------------------------------------------------------------------- */
/* DELTA_BYTE_ARRAY: types/byte-array.go:279-348                                 */

/* ByteArrayDeltaDecoder.Init (:285-313) then DecodeValues(dest[:n]) (:315-348).
 * Init: the prefix lengths (all of them), then the suffix decoder's Init (all its lengths), then
 * the two counts have to agree.  DecodeValues, per value: suffixDecoder.next() (EOF past the last,
 * negative length panics, io.ReadFull); make([]byte, 0, prefixLen + len(suffix)) panics when that
 * sum is negative; a prefix longer than the previous value is "invalid prefix len" and RETURNS
 * COUNT 0 (:330); a negative prefix length that survives make() copies nothing (:338).
 * data_cap: room at data; a value that does not fit ends the call with PQO_OUT_OF_RANGE_CAP
 * (the library's limit, not the reference's).  offsets[i + 1] - offsets[i] = len(value i). */
int pqo_delta_ba_decode(const uint8_t *src, size_t src_len, uint32_t n, uint32_t *offsets,
                        uint8_t *data, uint64_t data_cap, uint32_t *n_decoded, uint64_t *consumed,
                        uint32_t *values_count, int *init_status)
{
    rd_t r = {src, src_len, 0};
    int32_t *plens = NULL, *slens = NULL;
    uint32_t pc = 0, sc = 0;
    int st = ba_read_lens(&r, &plens, &pc);
    if (st == PQO_OK) {
        /* the suffix decoder reads on from where the prefix lengths ended */
        rd_t r2 = {src + r.pos, src_len - r.pos, 0};
        st = ba_read_lens(&r2, &slens, &sc);
        r.pos += r2.pos;
        if (st == PQO_OK && pc != sc) st = PQO_BYTE_ARRAY_COUNT_MISMATCH;
    }
    if (values_count) *values_count = st == PQO_OK ? pc : 0;
    if (init_status) *init_status = st;
    offsets[0] = 0;
    uint32_t i = 0;
    if (st == PQO_OK) {
        uint64_t prev_off = 0, prev_len = 0;
        for (; i < n; i++) {
            if (i >= sc) { st = PQO_EOF; break; }
            int32_t size = slens[i];
            if (size < 0) { st = PQO_BYTE_ARRAY_NEGATIVE_LEN; break; }
            const uint8_t *suffix = r.p + r.pos;
            st = rd_read_full(&r, NULL, (size_t)size);
            if (st != PQO_OK) break;
            int64_t plen = plens[i];
            if (plen + size < 0) { st = PQO_BYTE_ARRAY_NEGATIVE_LEN; break; }
            if ((int64_t)prev_len < plen) { st = PQO_BYTE_ARRAY_INVALID_PREFIX; i = 0; break; }
            uint64_t keep = plen > 0 ? (uint64_t)plen : 0;
            uint64_t off = offsets[i];
            if (off + keep + (uint64_t)size > data_cap) { st = PQO_OUT_OF_RANGE_CAP; i = 0; break; }
            memmove(data + off, data + prev_off, keep);
            memcpy(data + off + keep, suffix, (size_t)size);
            prev_off = off;
            prev_len = keep + (uint64_t)size;
            offsets[i + 1] = (uint32_t)(off + prev_len);
        }
    }
    free(plens);
    free(slens);
    if (n_decoded) *n_decoded = i;
    if (consumed) *consumed = r.pos;
    return st;
}

/* ------------------------------------------------------------------------- */
/* PLAIN fixed width and BOOLEAN PLAIN                                           */

/* Int32/Int64/Float/DoublePlainDecoder.DecodeValues (types/int32.go:56-72,
 * types/int64.go:56-72, types/float.go:67-79, types/double.go:68-80): one
 * binary.Read of elem_size little-endian bytes per value; nothing left is io.EOF,
 * a partial value io.ErrUnexpectedEOF. */
static int plain_next(rd_t *r, uint32_t elem_size, uint8_t *out)
{
    size_t rem = r->len - r->pos;
    if (rem == 0) return PQO_EOF;
    if (rem < elem_size) {
        r->pos = r->len;
        return PQO_UNEXPECTED_EOF;
    }
    memcpy(out, r->p + r->pos, elem_size);
    r->pos += elem_size;
    return PQO_OK;
}

int pqo_plain_decode(const uint8_t *src, size_t src_len, uint32_t elem_size, uint32_t n, void *dst,
                     uint32_t *n_decoded, uint64_t *consumed)
{
    rd_t r = {src, src_len, 0};
    int st = PQO_OK;
    uint32_t i = 0;
    for (; i < n; i++) {
        st = plain_next(&r, elem_size, (uint8_t *)dst + (size_t)i * elem_size);
        if (st != PQO_OK) break;
    }
    if (n_decoded) *n_decoded = i;
    if (consumed) *consumed = r.pos;
    return st;
}

/* BooleanPlainDecoder.DecodeValues on a fresh decoder (types/boolean.go:59-104): one
 * byte per 8 values, bit j of it is value j (unpackByte); a missing byte is io.EOF and
 * the count returned is the index of the group that could not be read. */
int pqo_bool_plain_decode(const uint8_t *src, size_t src_len, uint32_t n, int32_t *dst,
                          uint32_t *n_decoded, uint64_t *consumed)
{
    rd_t r = {src, src_len, 0};
    int st = PQO_OK;
    uint32_t i = 0;
    for (; i < n; i += 8) {
        if (r.pos >= r.len) { st = PQO_EOF; break; }
        uint8_t b = r.p[r.pos++];
        for (uint32_t j = 0; j < 8 && i + j < n; j++) dst[i + j] = (b >> j) & 1;
    }
    if (n_decoded) *n_decoded = i < n ? i : n;
    if (consumed) *consumed = r.pos;
    return st;
}

/* ------------------------------------------------------------------------- */
/* Data pages: layout/page-v1.go:40-116, layout/page-v2.go:42-149              */

static void make_level_decoders(uint32_t max_rep, uint32_t max_def, level_dec_t *rep,
                                level_dec_t *def)
{
    memset(rep, 0, sizeof *rep);
    memset(def, 0, sizeof *def);
    rep->max = max_rep;
    def->max = max_def;
    if (max_rep == 0) { /* layout/chunk.go:111-124: BOTH become ConstDecoder(0) */
        rep->is_const = 1;
        def->is_const = 1;
    } else {
        hybrid_new(&rep->h, bits_len16(max_rep));
        hybrid_new(&def->h, bits_len16(max_def));
    }
}

void pqo_page_read(const pqo_page_in *in, int32_t *rep, int32_t *def, void *values,
                   pqo_page_out *out)
{
    memset(out, 0, sizeof *out);
    rd_t r = {in->src, (size_t)in->src_len, 0};
    level_dec_t rd, dd;
    make_level_decoders(in->max_rep, in->max_def, &rd, &dd);

    dict_dec_t dict;
    delta_t delta;
    hybrid_t brle;
    memset(&delta, 0, sizeof delta);
    hybrid_new(&brle, 1);
    int st = PQO_OK;

    /* ---- read(): decoder initialisation ---- */
    if (in->version == 1) {
        /* page-v1.go:69-77: rep.InitSize, def.InitSize, values.Init on one reader */
        if (!rd.is_const) {
            st = hybrid_init_size(&rd.h, &r, 1);
            if (st != PQO_OK) { out->status = st; out->stage = PQO_STAGE_REP_INIT; return; }
        }
        if (!dd.is_const) {
            st = hybrid_init_size(&dd.h, &r, 1);
            if (st != PQO_OK) { out->status = st; out->stage = PQO_STAGE_DEF_INIT; return; }
        }
    } else {
        /* page-v2.go:81-103: raw level bytes, lengths from the header; a zero
         * length leaves that decoder uninitialised */
        uint64_t lv = (uint64_t)in->rep_len + in->def_len;
        if (lv > 0) {
            if (lv > in->src_len) { /* io.ReadFull(reader, data) fails, page-v2.go:87-90 */
                out->status = in->src_len == 0 ? PQO_EOF : PQO_UNEXPECTED_EOF;
                out->stage = PQO_STAGE_REP_INIT;
                return;
            }
            rd_t rr = {in->src, in->rep_len, 0};
            rd_t dr = {in->src + in->rep_len, in->def_len, 0};
            if (in->rep_len > 0 && !rd.is_const) hybrid_init_buffered(&rd.h, &rr, in->rep_len);
            if (in->def_len > 0 && !dd.is_const) hybrid_init_buffered(&dd.h, &dr, in->def_len);
            r.pos = (size_t)lv;
        }
    }
    switch (in->value_kind) {
    case PQO_VALUES_DICT: st = dict_init(&dict, &r); break;
    case PQO_VALUES_DELTA32: st = delta_init(&delta, &r, 0); break;
    case PQO_VALUES_DELTA64: st = delta_init(&delta, &r, 1); break;
    case PQO_VALUES_PLAIN:   /* Init only keeps the reader (types/int32.go:50-53) */
    case PQO_VALUES_BOOL_PLAIN: st = PQO_OK; break;
    case PQO_VALUES_BOOL_RLE: /* NewHybridDecoder(1, false).InitSize(reader), types/boolean.go:177-181 */
        st = hybrid_init_size(&brle, &r, 0);
        break;
    case PQO_VALUES_DELTA_LENGTH_BA: { /* Init decodes every length (types/byte-array.go:174-186) */
        uint32_t off0[1];
        int init = PQO_OK;
        (void)pqo_delta_length_ba_decode(r.p + r.pos, r.len - r.pos, 0, off0, NULL, NULL, NULL, NULL, &init);
        st = init;
        break;
    }
    case PQO_VALUES_DELTA_BA: { /* Init decodes both length streams (types/byte-array.go:285-313) */
        uint32_t off0[1];
        int init = PQO_OK;
        (void)pqo_delta_ba_decode(r.p + r.pos, r.len - r.pos, 0, off0, NULL, 0, NULL, NULL, NULL, &init);
        st = init;
        break;
    }
    default: st = PQO_NOT_INITIALIZED; break;
    }
    if (st != PQO_OK) {
        out->status = st;
        out->stage = PQO_STAGE_VALUES_INIT;
        delta_free(&delta);
        return;
    }

    /* ---- ReadValues(values[:num_values]), page-v1.go:80-116 / page-v2.go:113-149 ---- */
    uint32_t size = in->num_values;
    if (size == 0) { delta_free(&delta); return; }

    packed_array_t ra, da;
    uint32_t nn = 0, dummy = 0;
    st = decode_packed_array(&rd, size, rep, &ra, NULL, &dummy);
    if (st != PQO_OK) { out->status = st; out->stage = PQO_STAGE_REP; delta_free(&delta); return; }
    st = decode_packed_array(&dd, size, def, &da, NULL, &nn);
    if (st != PQO_OK) { out->status = st; out->stage = PQO_STAGE_DEF; delta_free(&delta); return; }
    out->not_null = nn;

    if (nn != 0) {
        uint32_t cnt = 0;
        if (in->value_kind == PQO_VALUES_DICT) {
            st = dict_decode_values(&dict, (const uint8_t *)in->dict, in->dict_len, in->elem_size,
                                    nn, (uint8_t *)values, &cnt);
        } else if (in->value_kind == PQO_VALUES_PLAIN) {
            st = pqo_plain_decode(r.p + r.pos, r.len - r.pos, in->elem_size, nn, values, &cnt, NULL);
        } else if (in->value_kind == PQO_VALUES_BOOL_PLAIN) {
            st = pqo_bool_plain_decode(r.p + r.pos, r.len - r.pos, nn, (int32_t *)values, &cnt, NULL);
        } else if (in->value_kind == PQO_VALUES_DELTA_LENGTH_BA) {
            /* values: (num_values + 1) uint32 offsets, then the value bytes */
            uint32_t *offs = (uint32_t *)values;
            st = pqo_delta_length_ba_decode(r.p + r.pos, r.len - r.pos, nn, offs,
                                            (uint8_t *)(offs + in->num_values + 1), &cnt, NULL, NULL, NULL);
        } else if (in->value_kind == PQO_VALUES_DELTA_BA) {
            uint32_t *offs = (uint32_t *)values;
            st = pqo_delta_ba_decode(r.p + r.pos, r.len - r.pos, nn, offs, (uint8_t *)(offs + in->num_values + 1),
                                     in->data_cap, &cnt, NULL, NULL, NULL);
        } else if (in->value_kind == PQO_VALUES_BOOL_RLE) {
            for (; cnt < nn; cnt++) { /* dest[i] = n == 1, types/boolean.go:186-193 */
                int32_t v;
                st = hybrid_next(&brle, &v);
                if (st != PQO_OK) break;
                ((int32_t *)values)[cnt] = v == 1;
            }
        } else {
            int is64 = in->value_kind == PQO_VALUES_DELTA64;
            for (; cnt < nn; cnt++) {
                int64_t v;
                st = delta_next(&delta, &v);
                if (st != PQO_OK) break;
                if (is64) ((int64_t *)values)[cnt] = v; else ((int32_t *)values)[cnt] = (int32_t)v;
            }
        }
        out->values_decoded = cnt;
        if (st != PQO_OK) { /* any error, EOF included: page-v1.go:103-110 */
            out->status = PQO_PAGE_VALUES_FAILED;
            out->detail = st;
            out->stage = PQO_STAGE_VALUES;
            delta_free(&delta);
            return;
        }
    }
    out->n = size;
    delta_free(&delta);
}

/* ------------------------------------------------------------------------- */
/* Thrift compact protocol, as much of it as PageHeader.Read needs               */

enum { T_STOP = 0, T_BOOL_TRUE = 1, T_BOOL_FALSE = 2, T_BYTE = 3, T_I16 = 4, T_I32 = 5, T_I64 = 6,
       T_DOUBLE = 7, T_BINARY = 8, T_LIST = 9, T_SET = 10, T_MAP = 11, T_STRUCT = 12 };

/* readVarint64: bytes until one without the continuation bit; bits beyond 64 fall off. */
static int th_varint(rd_t *r, uint64_t *out)
{
    uint64_t v = 0;
    unsigned shift = 0;
    for (;;) {
        uint8_t b;
        if (rd_byte(r, &b) != PQO_OK) return PQO_EOF;
        if (shift < 64) v |= (uint64_t)(b & 0x7f) << shift;
        if (!(b & 0x80)) break;
        shift += 7;
    }
    *out = v;
    return PQO_OK;
}

static int th_i32(rd_t *r, int32_t *out) /* ReadI32: int32(varint), zig-zag */
{
    uint64_t u;
    if (th_varint(r, &u) != PQO_OK) return PQO_EOF;
    uint32_t x = (uint32_t)u;
    *out = (int32_t)((x >> 1) ^ (uint32_t)-(int32_t)(x & 1));
    return PQO_OK;
}

/* ReadFieldBegin: type 0 in the low nibble is STOP whatever the high nibble says. */
static int th_field(rd_t *r, int *type, int32_t *id, int32_t *last_id)
{
    uint8_t b;
    if (rd_byte(r, &b) != PQO_OK) return PQO_EOF;
    *type = b & 0x0f;
    if (*type == T_STOP) return PQO_OK;
    if (*type > T_STRUCT) return PQO_INVALID_STREAM;
    int delta = b >> 4;
    if (delta == 0) {
        int32_t v;
        if (th_i32(r, &v) != PQO_OK) return PQO_EOF;
        *id = (int16_t)v;
    } else {
        *id = (int16_t)(*last_id + delta);
    }
    *last_id = *id;
    return PQO_OK;
}

static int th_skip(rd_t *r, int type, int depth, int in_collection)
{
    uint64_t u;
    if (depth <= 0) return PQO_INVALID_STREAM; /* "Depth limit exceeded" */
    switch (type) {
    case T_BOOL_TRUE:
    case T_BOOL_FALSE: /* a field's boolean lives in its type nibble; in a collection it is a byte */
        return in_collection ? rd_read_full(r, NULL, 1) : PQO_OK;
    case T_BYTE: return rd_read_full(r, NULL, 1);
    case T_I16:
    case T_I32:
    case T_I64: return th_varint(r, &u);
    case T_DOUBLE: return rd_read_full(r, NULL, 8) == PQO_OK ? PQO_OK : PQO_EOF;
    case T_BINARY: {
        if (th_varint(r, &u) != PQO_OK) return PQO_EOF;
        int32_t n = (int32_t)(uint32_t)u;
        if (n < 0) return PQO_INVALID_STREAM;
        if (n == 0) return PQO_OK;
        return rd_read_full(r, NULL, (size_t)n) == PQO_OK ? PQO_OK : PQO_EOF;
    }
    case T_STRUCT: {
        int32_t last = 0, id = 0;
        for (;;) {
            int t;
            int st = th_field(r, &t, &id, &last);
            if (st != PQO_OK) return st;
            if (t == T_STOP) return PQO_OK;
            st = th_skip(r, t, depth - 1, 0);
            if (st != PQO_OK) return st;
        }
    }
    case T_LIST:
    case T_SET: {
        uint8_t b;
        if (rd_byte(r, &b) != PQO_OK) return PQO_EOF;
        int32_t n = b >> 4;
        int et = b & 0x0f;
        if (n == 15) {
            if (th_varint(r, &u) != PQO_OK) return PQO_EOF;
            n = (int32_t)(uint32_t)u;
            if (n < 0) return PQO_INVALID_STREAM;
        }
        if (et > T_STRUCT) return PQO_INVALID_STREAM;
        for (int32_t i = 0; i < n; i++) {
            int st = th_skip(r, et, depth - 1, 1);
            if (st != PQO_OK) return st;
        }
        return PQO_OK;
    }
    case T_MAP: {
        if (th_varint(r, &u) != PQO_OK) return PQO_EOF;
        int32_t n = (int32_t)(uint32_t)u;
        if (n < 0) return PQO_INVALID_STREAM;
        uint8_t kv = 0;
        if (n != 0 && rd_byte(r, &kv) != PQO_OK) return PQO_EOF;
        int kt = kv >> 4, vt = kv & 0x0f;
        if (kt > T_STRUCT || vt > T_STRUCT) return PQO_INVALID_STREAM;
        for (int32_t i = 0; i < n; i++) {
            int st = th_skip(r, kt, depth - 1, 1);
            if (st == PQO_OK) st = th_skip(r, vt, depth - 1, 1);
            if (st != PQO_OK) return st;
        }
        return PQO_OK;
    }
    default: return PQO_OK; /* STOP: nothing to skip */
    }
}

typedef struct {
    int present;
    int32_t f[8];     /* captured i32 fields, by field id */
    uint32_t isset;   /* bit id: field seen with the expected type */
    uint32_t is_compressed;
} th_sub_t;

/* One of DataPageHeader / IndexPageHeader / DictionaryPageHeader / DataPageHeaderV2
 * (parquet/parquet.go:3994, :4252, :4334, :4586): i32_mask = ids read as i32, required = ids that must
 * be there, bool_id = the boolean field (0: none). */
static int th_sub(rd_t *r, th_sub_t *h, uint32_t i32_mask, uint32_t required, int bool_id)
{
    memset(h, 0, sizeof *h);
    h->present = 1;
    h->is_compressed = 1; /* NewDataPageHeaderV2 */
    int32_t last = 0, id = 0;
    for (;;) {
        int t;
        int st = th_field(r, &t, &id, &last);
        if (st != PQO_OK) return st;
        if (t == T_STOP) break;
        if (id >= 1 && id < 8 && ((i32_mask >> id) & 1) && t == T_I32) {
            if (th_i32(r, &h->f[id]) != PQO_OK) return PQO_EOF;
            h->isset |= 1u << id;
        } else if (id == bool_id && (t == T_BOOL_TRUE || t == T_BOOL_FALSE)) {
            h->is_compressed = t == T_BOOL_TRUE;
        } else {
            st = th_skip(r, t, 64, 0);
            if (st != PQO_OK) return st;
        }
    }
    return (h->isset & required) == required ? PQO_OK : PQO_INVALID_STREAM;
}

typedef struct {
    int32_t type, uncompressed, compressed;
    th_sub_t data, dict, v2;
} th_page_header_t;

/* PageHeader.Read (parquet/parquet.go:5886-6009) */
static int th_page_header(rd_t *r, th_page_header_t *h)
{
    memset(h, 0, sizeof *h);
    uint32_t isset = 0;
    int32_t last = 0, id = 0;
    for (;;) {
        int t;
        int st = th_field(r, &t, &id, &last);
        if (st != PQO_OK) return st;
        if (t == T_STOP) break;
        if (id >= 1 && id <= 4 && t == T_I32) {
            int32_t v;
            if (th_i32(r, &v) != PQO_OK) return PQO_EOF;
            if (id == 1) h->type = v;
            if (id == 2) h->uncompressed = v;
            if (id == 3) h->compressed = v;
            isset |= 1u << id;
        } else if (id == 5 && t == T_STRUCT) {
            st = th_sub(r, &h->data, 0x1e, 0x1e, 0);
        } else if (id == 6 && t == T_STRUCT) {
            th_sub_t idx;
            st = th_sub(r, &idx, 0, 0, 0);
        } else if (id == 7 && t == T_STRUCT) {
            st = th_sub(r, &h->dict, 0x06, 0x06, 0);
        } else if (id == 8 && t == T_STRUCT) {
            st = th_sub(r, &h->v2, 0x7e, 0x7e, 7);
        } else {
            st = th_skip(r, t, 64, 0);
        }
        if (st != PQO_OK) return st;
    }
    return (isset & 0x0e) == 0x0e ? PQO_OK : PQO_INVALID_STREAM;
}

/* getValuesDecoder (layout/page-dict.go:143-345): does the reference have a decoder for this
 * encoding of this type? */
static int values_decoder_exists(uint32_t type, int32_t type_length, int32_t enc)
{
    if (enc == 2) enc = 8; /* PLAIN_DICTIONARY -> RLE_DICTIONARY */
    switch (type) {
    case 0: return enc == 0 || enc == 3 || enc == 8;
    case 1:
    case 2: return enc == 0 || enc == 5 || enc == 8;
    case 3:
    case 4:
    case 5: return enc == 0 || enc == 8;
    case 6: return enc == 0 || enc == 6 || enc == 7 || enc == 8;
    case 7: return type_length >= 0 && (enc == 0 || enc == 7 || enc == 8);
    default: return 0;
    }
}

/* readBlockData (layout/block.go:17-47) for `compressed` bytes at the reader. */
static int walk_block(rd_t *r, uint32_t codec, int32_t compressed, int32_t uncompressed)
{
    size_t left = r->len - r->pos;
    size_t want = compressed > 0 ? (size_t)compressed : 0; /* io.LimitReader of a negative size reads nothing */
    size_t got = want < left ? want : left;
    r->pos += got;
    if ((int64_t)got != (int64_t)compressed) return PQO_WALK_BLOCK_SIZE;
    if (codec == 0 && compressed != uncompressed) return PQO_WALK_UNCOMPRESSED_SIZE;
    return PQO_OK;
}

int pqo_walk_chunk(const pqo_chunk_in *in, pqo_page_info *infos, uint32_t cap, uint32_t *n_pages,
                   int64_t *count_out)
{
    rd_t r = {in->src, in->src_len, 0};
    const int64_t start = in->has_dictionary_page_offset ? in->dictionary_page_offset : in->data_page_offset;
    int64_t count = 0; /* offsetReader.count: bytes read plus seek distances */
    uint32_t n = 0;
    int have_dict = 0, st = PQO_OK;
    for (;;) {
        if (in->total_compressed_size - count < 1) break; /* chunk.go:136 */
        size_t h0 = r.pos;
        th_page_header_t h;
        if (th_page_header(&r, &h) != PQO_OK) { st = PQO_WALK_HEADER_FAILED; count += (int64_t)(r.pos - h0); break; }
        if (n >= cap) { st = PQO_OUT_OF_RANGE_CAP; break; }
        pqo_page_info *pi = &infos[n];
        memset(pi, 0, sizeof *pi);
        pi->type = h.type;
        pi->compressed_size = h.compressed;
        pi->uncompressed_size = h.uncompressed;
        pi->header_off = h0;
        pi->data_off = r.pos;
        pi->is_compressed = 1;
        if (h.type == 2) { /* DICTIONARY_PAGE, chunk.go:148-180 + page-dict.go:28-57 */
            if (have_dict) st = PQO_WALK_SECOND_DICTIONARY;
            else if (in->physical_type == 0 || in->physical_type > 7 ||
                     (in->physical_type == 7 && in->type_length < 0)) st = PQO_WALK_VALUE_ENCODING;
            else if (!h.dict.present) st = PQO_WALK_MISSING_HEADER;
            else if (h.dict.f[1] < 0) st = PQO_WALK_NEGATIVE_NUM_VALUES;
            else if (h.dict.f[2] != 0 && h.dict.f[2] != 2) st = PQO_WALK_DICT_ENCODING;
            if (h.dict.present) { pi->num_values = h.dict.f[1]; pi->encoding = h.dict.f[2]; }
            if (st == PQO_OK) st = walk_block(&r, in->codec, h.compressed, h.uncompressed);
            count += (int64_t)(r.pos - h0);
            if (st != PQO_OK) break;
            have_dict = 1;
            n++;
            if (in->has_dictionary_page_offset && in->dictionary_page_offset != start + (int64_t)r.pos) {
                /* Seek(DataPageOffset): count moves by the distance (layout/helpers.go:52-60) */
                int64_t target = in->data_page_offset - start;
                if (target < 0) { st = PQO_WALK_SEEK; break; }
                count += target - (int64_t)r.pos;
                r.pos = (uint64_t)target > r.len ? r.len : (size_t)target;
            }
            continue;
        }
        if (h.type == 0) { /* DATA_PAGE: init (page-v1.go:21-38) then read (:40-67) */
            if (!h.data.present) st = PQO_WALK_MISSING_HEADER;
            else if (in->max_rep != 0 && (h.data.f[3] != 3 || h.data.f[4] != 3)) st = PQO_WALK_LEVEL_ENCODING;
            else if (h.data.f[1] < 0) st = PQO_WALK_NEGATIVE_NUM_VALUES;
            if (h.data.present) {
                pi->num_values = h.data.f[1];
                pi->encoding = h.data.f[2];
                pi->def_encoding = h.data.f[3];
                pi->rep_encoding = h.data.f[4];
            }
            if (st == PQO_OK) st = walk_block(&r, in->codec, h.compressed, h.uncompressed);
            if (st == PQO_OK && !values_decoder_exists(in->physical_type, in->type_length, h.data.f[2]))
                st = PQO_WALK_VALUE_ENCODING;
        } else if (h.type == 3) { /* DATA_PAGE_V2: page-v2.go:42-108 */
            if (!h.v2.present) st = PQO_WALK_MISSING_HEADER;
            else if (h.v2.f[1] < 0) st = PQO_WALK_NEGATIVE_NUM_VALUES;
            else if (h.v2.f[6] < 0 || h.v2.f[5] < 0) st = PQO_WALK_NEGATIVE_LEVELS_LENGTH;
            else if (!values_decoder_exists(in->physical_type, in->type_length, h.v2.f[4])) st = PQO_WALK_VALUE_ENCODING;
            if (h.v2.present) {
                pi->num_values = h.v2.f[1];
                pi->encoding = h.v2.f[4];
                pi->def_len = h.v2.f[5];
                pi->rep_len = h.v2.f[6];
                pi->is_compressed = h.v2.is_compressed;
            }
            if (st == PQO_OK) {
                int32_t levels = (int32_t)((uint32_t)h.v2.f[5] + (uint32_t)h.v2.f[6]);
                if (levels > 0) { /* io.ReadFull of both level sections */
                    size_t left = r.len - r.pos;
                    if (left < (size_t)levels) { r.pos = r.len; st = PQO_WALK_BLOCK_SIZE; }
                    else r.pos += (size_t)levels;
                }
                if (st == PQO_OK)
                    st = walk_block(&r, in->codec, (int32_t)((uint32_t)h.compressed - (uint32_t)levels),
                                    (int32_t)((uint32_t)h.uncompressed - (uint32_t)levels));
            }
        } else {
            st = PQO_WALK_PAGE_TYPE;
        }
        count += (int64_t)(r.pos - h0);
        if (st != PQO_OK) break;
        n++;
    }
    if (n_pages) *n_pages = n;
    if (count_out) *count_out = count;
    return st;
}

/* ------------------------------------------------------------------------- */
/* Encoders — input generators only.                                           */

typedef struct {
    uint8_t *p;
    size_t len, cap;
    int overflow;
} wr_t;

static void wr_bytes(wr_t *w, const uint8_t *b, size_t n)
{
    if (w->len + n > w->cap) { w->overflow = 1; return; }
    memcpy(w->p + w->len, b, n);
    w->len += n;
}

static void wr_uvarint(wr_t *w, uint64_t v) /* binary.PutUvarint */
{
    uint8_t b[10];
    int n = 0;
    while (v >= 0x80) { b[n++] = (uint8_t)(v | 0x80); v >>= 7; }
    b[n++] = (uint8_t)v;
    wr_bytes(w, b, (size_t)n);
}

static void wr_varint(wr_t *w, int64_t v) /* binary.PutVarint (zig-zag) */
{
    uint64_t u = (uint64_t)v << 1;
    if (v < 0) u = ~u;
    wr_uvarint(w, u);
}

/* HybridEncoder: encoding/hybrid-encoder.go:17-237 */
typedef struct {
    int w;
    int32_t prev;
    int repeat;
    int bp_groups;
    uint8_t *pack; /* packBuffer */
    size_t pack_len, pack_cap;
    int32_t vbuf[8];
    int vpos;
    int bp_run;
    wr_t *out;
} henc_t;

static void henc_end_bp_run(henc_t *e) /* endPreviousBitPackedRun, :177-194 */
{
    if (!e->bp_run) return;
    wr_uvarint(e->out, (uint64_t)((e->bp_groups << 1) | 1));
    /* the reference never clears packBuffer between runs of one encoder
     * (:165,188): every later bit-packed run re-emits all earlier groups.  That
     * is an encoder defect, irrelevant to decoding; as an input generator this
     * restatement emits each run's own groups only, which is what the decoder
     * side of the reference expects to read. */
    wr_bytes(e->out, e->pack, e->pack_len);
    e->pack_len = 0;
    e->bp_run = 0;
    e->bp_groups = 0;
}

static void henc_write_or_append_bp(henc_t *e) /* writeOrAppendBitPackedRun, :157-175 */
{
    if (e->bp_groups >= 63) henc_end_bp_run(e);
    e->bp_run = 1;
    if (e->pack_len + 32 > e->pack_cap) {
        e->pack_cap = e->pack_cap ? e->pack_cap * 2 : 4096;
        e->pack = (uint8_t *)realloc(e->pack, e->pack_cap);
        if (!e->pack) abort();
    }
    pqo_pack8_32(e->vbuf, e->w, e->pack + e->pack_len);
    e->pack_len += (size_t)e->w;
    e->vpos = 0;
    e->repeat = 0;
    e->bp_groups++;
}

static void henc_write_rle(henc_t *e) /* writeRLERun, :196-237 */
{
    henc_end_bp_run(e);
    wr_uvarint(e->out, (uint64_t)((uint64_t)e->repeat << 1));
    int nb = (e->w + 7) / 8;
    uint8_t b[4];
    for (int i = 0; i < nb; i++) b[i] = (uint8_t)((int64_t)e->prev >> (8 * i));
    if (nb == 0) return; /* :208-209 returns before resetting the counters */
    wr_bytes(e->out, b, (size_t)nb);
    e->repeat = 0;
    e->vpos = 0;
}

static void henc_append(henc_t *e, int32_t v) /* AppendSingle, :88-124 */
{
    if (v == e->prev) {
        e->repeat++;
        if (e->repeat >= 8) return;
    } else {
        if (e->repeat >= 8) henc_write_rle(e);
        e->repeat = 1;
        e->prev = v;
    }
    e->vbuf[e->vpos++] = v;
    if (e->vpos == 8) henc_write_or_append_bp(e);
}

int pqo_hybrid_encode(const int32_t *values, size_t n, int w, uint8_t *out, size_t cap,
                      size_t *out_len)
{
    if (w < 0 || w > 32) return -1;
    wr_t wr = {out, 0, cap, 0};
    henc_t e;
    memset(&e, 0, sizeof e);
    e.w = w;
    e.out = &wr;
    for (size_t i = 0; i < n; i++) henc_append(&e, values[i]);
    /* Write, :136-155 */
    if (e.repeat >= 8) {
        henc_write_rle(&e);
    } else if (e.vpos > 0) {
        for (int i = e.vpos; i < 8; i++) e.vbuf[i] = 0;
        henc_write_or_append_bp(&e);
        henc_end_bp_run(&e);
    } else {
        henc_end_bp_run(&e);
    }
    free(e.pack);
    *out_len = wr.len;
    return wr.overflow ? -2 : 0;
}

/* DeltaBinaryPackEncoder32/64: encoding/delta-binary-packed-encoder.go:20-384 */
static int bits_len64(uint64_t x) { return x ? 64 - __builtin_clzll(x) : 0; }

typedef struct {
    int is64;
    int block_size, mb_count, mb_values;
    int64_t *deltas;
    int n_deltas;
    int64_t min_delta, prev, first;
    size_t count;
    wr_t body;
} denc_t;

static void denc_flush(denc_t *e) /* flush32 :172-238, flush64 :318-384 */
{
    for (int i = 0; i < e->n_deltas; i++) {
        if (e->is64)
            e->deltas[i] = (int64_t)((uint64_t)e->deltas[i] - (uint64_t)e->min_delta);
        else
            e->deltas[i] = (int32_t)((uint32_t)e->deltas[i] - (uint32_t)e->min_delta);
    }
    wr_varint(&e->body, e->min_delta);
    uint8_t *widths = (uint8_t *)calloc((size_t)e->mb_count, 1);
    int nmb = 0;
    size_t data_cap = (size_t)e->block_size * 8 + 64;
    uint8_t *data = (uint8_t *)malloc(data_cap);
    size_t data_len = 0;
    if (!widths || !data) abort();
    for (int i = 0; i < e->n_deltas; i += e->mb_values) {
        int end = i + e->mb_values;
        if (end >= e->n_deltas) end = e->n_deltas;
        uint64_t mx = e->is64 ? (uint64_t)e->deltas[i] : (uint64_t)(uint32_t)e->deltas[i];
        for (int j = i; j < end; j++) {
            uint64_t u = e->is64 ? (uint64_t)e->deltas[j] : (uint64_t)(uint32_t)e->deltas[j];
            if (mx < u) mx = u;
        }
        int bw = bits_len64(mx);
        widths[nmb++] = (uint8_t)bw;
        for (int g = 0; g < e->mb_values / 8; g++) {
            if (e->is64) {
                int64_t grp[8];
                for (int k = 0; k < 8; k++) {
                    int j = i + g * 8 + k;
                    grp[k] = j < end ? e->deltas[j] : 0;
                }
                pqo_pack8_64(grp, bw, data + data_len);
            } else {
                int32_t grp[8];
                for (int k = 0; k < 8; k++) {
                    int j = i + g * 8 + k;
                    grp[k] = j < end ? (int32_t)e->deltas[j] : 0;
                }
                pqo_pack8_32(grp, bw, data + data_len);
            }
            data_len += (size_t)bw;
        }
    }
    wr_bytes(&e->body, widths, (size_t)e->mb_count);
    wr_bytes(&e->body, data, data_len);
    free(widths);
    free(data);
    e->min_delta = 2147483647LL; /* math.MaxInt32 for BOTH element sizes (:234, :380) */
    e->n_deltas = 0;
}

static void denc_add(denc_t *e, int64_t n) /* AddInt32 :140-162, AddInt64 :290-312 */
{
    e->count++;
    if (e->count == 1) {
        e->first = n;
        e->prev = n;
        return;
    }
    int64_t delta = e->is64 ? (int64_t)((uint64_t)n - (uint64_t)e->prev)
                            : (int64_t)(int32_t)((uint32_t)n - (uint32_t)e->prev);
    e->prev = n;
    e->deltas[e->n_deltas++] = delta;
    if (delta < e->min_delta) e->min_delta = delta;
    if (e->n_deltas == e->block_size) denc_flush(e);
}

static int delta_encode(const void *values, size_t n, int is64, int block_size, int miniblocks,
                        uint8_t *out, size_t cap, size_t *out_len)
{
    /* checkBlockParameters, :39-66 */
    if (block_size <= 0 || block_size % 128 != 0) return PQO_INVALID_BLOCK_SIZE;
    if (miniblocks <= 0 || block_size % miniblocks != 0) return PQO_INVALID_MINIBLOCK_COUNT;
    if ((block_size / miniblocks) % 32 != 0) return PQO_INVALID_MINIBLOCK_COUNT;
    denc_t e;
    memset(&e, 0, sizeof e);
    e.is64 = is64;
    e.block_size = block_size;
    e.mb_count = miniblocks;
    e.mb_values = block_size / miniblocks;
    e.deltas = (int64_t *)malloc(sizeof(int64_t) * (size_t)block_size);
    e.min_delta = 2147483647LL;
    e.body.cap = n * 9 + (n / (size_t)block_size + 2) * (size_t)(miniblocks + 16) +
                 (size_t)block_size * 9 + 64;
    e.body.p = (uint8_t *)malloc(e.body.cap);
    if (!e.deltas || !e.body.p) abort();
    for (size_t i = 0; i < n; i++)
        denc_add(&e, is64 ? ((const int64_t *)values)[i] : (int64_t)((const int32_t *)values)[i]);
    /* write, :68-92 */
    if (e.n_deltas > 0) denc_flush(&e);
    wr_t w = {out, 0, cap, 0};
    wr_uvarint(&w, (uint64_t)block_size);
    wr_uvarint(&w, (uint64_t)miniblocks);
    wr_uvarint(&w, (uint64_t)e.count);
    wr_varint(&w, e.first);
    wr_bytes(&w, e.body.p, e.body.len);
    int bad = w.overflow || e.body.overflow;
    free(e.deltas);
    free(e.body.p);
    *out_len = w.len;
    return bad ? -2 : 0;
}

int pqo_delta_encode32(const int32_t *values, size_t n, int block_size, int miniblocks,
                       uint8_t *out, size_t cap, size_t *out_len)
{
    return delta_encode(values, n, 0, block_size, miniblocks, out, cap, out_len);
}

int pqo_delta_encode64(const int64_t *values, size_t n, int block_size, int miniblocks,
                       uint8_t *out, size_t cap, size_t *out_len)
{
    return delta_encode(values, n, 1, block_size, miniblocks, out, cap, out_len);
}
