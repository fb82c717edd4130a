// pq_host.hpp — host-side C++ mirror of the reference's decoder interfaces over the C ABI.
//
// The reference (hexbee-net/parquet-go) is Go; its decode seams are Go interfaces:
//   encoding.Decoder      Init / InitSize / Next                (encoding/encoding.go:18-23)
//   types.ValuesDecoder   Init / DecodeValues                   (types/types.go:22-27)
//   layout.levelDecoder + decodePackedArray                     (layout/level-decoder.go:5-18, layout/helpers.go:68-90)
//   layout.PageReader     ReadValues / NumValues                (layout/page.go:14-21)
//   readPageData                                                 (file-reader.go:268-302)
// The classes below keep those names, argument meanings and error behaviour, with two
// mechanical differences: a Go `error` is a pqhost::Error value (falsy == nil), and a
// `[]interface{}` destination is a typed pointer + count.  Every decode goes through
// libpqdecode (include/pqdecode.h) to the GPU; nothing here decodes on the CPU.  What is
// host-side is only what is host-side bookkeeping in the reference too: reader positions,
// the one-byte dictionary width check and the PackedArray container.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "pqdecode.h"

namespace pqhost {

// ---- error value -------------------------------------------------------------------------
struct Error {
    int status = PQ_OK;  // PQ_* code; PQ_OK == Go's nil
    int stage = PQ_STAGE_NONE;
    int detail = 0;
    Error() = default;
    explicit Error(int s, int st = PQ_STAGE_NONE, int d = 0) : status(s), stage(st), detail(d) {}
    explicit operator bool() const { return status != PQ_OK; }
    bool is(int s) const { return status == s; }
    std::string message() const
    {
        static const char *stages[] = {"", "failed to initialize repetition decoder: ",
                                       "failed to initialize definition decoder: ", "",
                                       "read repetition levels failed: ",
                                       "read definition levels failed: ", ""};
        std::string m = (stage >= 0 && stage <= 6) ? stages[stage] : "";
        m += pq_status_string(status);
        if (status == PQ_PAGE_VALUES_FAILED && detail) m += std::string(" (") + pq_status_string(detail) + ")";
        return m;
    }
};

// ---- bytes.Reader ----------------------------------------------------------------------------
// Every reader that reaches a decoder in the reference is an in-memory bytes.Reader
// (layout/block.go:46, layout/page-v2.go:93,99), so the mirror is a pointer, a length and a cursor.
class Reader {
public:
    Reader() = default;
    Reader(const uint8_t *p, size_t n) : base_(p), len_(n) {}
    explicit Reader(const std::vector<uint8_t> &v) : base_(v.data()), len_(v.size()) {}
    size_t Len() const { return len_ - pos_; }  // unread bytes, as bytes.Reader.Len()
    const uint8_t *cursor() const { return base_ + pos_; }
    void advance(size_t n) { pos_ += std::min(n, Len()); }

private:
    const uint8_t *base_ = nullptr;
    size_t len_ = 0, pos_ = 0;
};

// ---- device handle -----------------------------------------------------------------------
// One per calling thread (a pq_ctx is single-threaded).  Throws when there is no usable GPU:
// there is no CPU path to fall back to.
class Device {
public:
    explicit Device(int device = 0, void *hip_stream = nullptr)
    {
        int st = pq_ctx_create(device, hip_stream, &ctx_);
        if (st != PQ_OK) throw std::runtime_error(std::string("pq_ctx_create: ") + pq_status_string(st));
    }
    ~Device() { pq_ctx_destroy(ctx_); }
    Device(const Device &) = delete;
    Device &operator=(const Device &) = delete;
    pq_ctx *ctx() const { return ctx_; }

private:
    pq_ctx *ctx_ = nullptr;
};

inline void check_library(int st, const char *what)
{
    if (st >= PQ_ERR_BAD_ARGUMENT) throw std::runtime_error(std::string(what) + ": " + pq_status_string(st));
}

// ---- encoding.PackedArray (encoding/packed-array.go:16-129) -------------------------------
// Container semantics kept exactly: AppendSingle flushes a full buffer only BEFORE inserting,
// so the last 1..8 values stay in buf and At() falls back to it.
class PackedArray {
public:
    Error Reset(int bitWidth)
    {
        if (bitWidth < 0 || bitWidth > 32) return Error(PQ_ERR_BAD_ARGUMENT);
        bw_ = bitWidth;
        count_ = 0;
        bufPos_ = 0;
        data_.clear();
        return Error();
    }
    void AppendSingle(int32_t v)
    {
        if (bufPos_ == 8) Flush();
        buf_[bufPos_++] = v;
        ++count_;
    }
    // n x AppendSingle
    void AppendLevels(const int32_t *v, size_t n)
    {
        data_.reserve(data_.size() + (n / 8 + 1) * (size_t)bw_);
        for (size_t i = 0; i < n; ++i) AppendSingle(v[i]);
    }
    Error AppendArray(const PackedArray *other)
    {
        if (other == nullptr || bw_ != other->bw_) return Error(PQ_ERR_BAD_ARGUMENT);
        for (int i = 0; i < other->count_; ++i) {
            int32_t v = 0;
            if (Error e = other->At(i, &v)) return e;
            AppendSingle(v);
        }
        return Error();
    }
    int Count() const { return count_; }
    int BitWidth() const { return bw_; }
    Error At(int pos, int32_t *out) const
    {
        *out = 0;
        if (pos < 0 || pos >= count_) return Error(PQ_ERR_OUT_OF_RANGE);
        if (bw_ == 0) return Error();
        size_t block = (size_t)(pos / 8) * (size_t)bw_;
        int idx = pos % 8;
        if (block >= data_.size()) {
            *out = buf_[idx];
            return Error();
        }
        // value idx of the group: bits [idx*bw, (idx+1)*bw) of the little-endian group
        uint64_t bit = (uint64_t)idx * (uint64_t)bw_;
        uint64_t acc = 0;
        size_t first = block + bit / 8;
        for (int b = 0; b < 5 && first + b < block + (size_t)bw_; ++b) acc |= (uint64_t)data_[first + b] << (8 * b);
        acc >>= bit % 8;
        *out = (int32_t)(uint32_t)(bw_ == 32 ? acc : (acc & ((1ull << bw_) - 1)));
        return Error();
    }
    void Flush()
    {
        for (int i = bufPos_; i < 8; ++i) buf_[i] = 0;
        // pack8int32_w (encoding/bitpacking32.go:45-79): a value contributes to exactly the
        // bytes its bw-bit field overlaps, unmasked, so a value wider than bw bleeds into the
        // upper bits of its last byte as in the reference
        size_t at = data_.size();
        data_.resize(at + (size_t)bw_, 0);
        for (int k = 0; k < 8 && bw_ > 0; ++k) {
            int first_bit = k * bw_, last_byte = (first_bit + bw_ - 1) / 8;
            uint64_t v = (uint64_t)(uint32_t)buf_[k] << (first_bit % 8);
            for (int b = first_bit / 8; b <= last_byte; ++b, v >>= 8) data_[at + b] |= (uint8_t)v;
        }
        bufPos_ = 0;
    }
    const std::vector<uint8_t> &Data() const { return data_; }  // what Write() emits
    // The array that AppendSingle x count would have built, from the bytes pq_levels_pack made of
    // those values (ceil(count / 8) groups): the last 1..8 values go back into buf, the groups in
    // front of them are the flushed data.
    Error Adopt(int bitWidth, int count, const uint8_t *packed, size_t packedLen)
    {
        if (Error e = Reset(bitWidth)) return e;
        if (count < 0 || packedLen < (size_t)((count + 7) / 8) * (size_t)bitWidth) return Error(PQ_ERR_BAD_ARGUMENT);
        if (count == 0) return Error();
        const size_t flushed = (size_t)((count - 1) / 8);
        data_.assign(packed, packed + ((size_t)((count + 7) / 8)) * (size_t)bitWidth);
        count_ = count;
        for (int i = 0; i < 8; ++i) {
            int32_t v = 0;
            int pos = (int)flushed * 8 + i;
            if (pos < count) At(pos, &v);  // still whole in data_ at this point
            buf_[i] = v;
        }
        bufPos_ = count - (int)flushed * 8;
        data_.resize(flushed * (size_t)bitWidth);
        return Error();
    }

private:
    int count_ = 0, bw_ = 0, bufPos_ = 0;
    int32_t buf_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<uint8_t> data_;
};

// ---- encoding.Decoder ---------------------------------------------------------------------
class Decoder {
public:
    virtual ~Decoder() = default;
    virtual Error Init(Reader &r) = 0;
    virtual Error InitSize(Reader &r) = 0;
    virtual Error Next(int32_t *v) = 0;
    // n x Next() in one device call: dst[0..*count) are the values pulled before `error`.
    virtual Error DecodeBatch(int32_t *dst, uint32_t n, uint32_t *count) = 0;
};

// encoding.ConstDecoder (encoding/const.go:7-19)
class ConstDecoder final : public Decoder {
public:
    explicit ConstDecoder(int32_t v) : v_(v) {}
    Error Init(Reader &) override { return Error(); }
    Error InitSize(Reader &) override { return Error(); }
    Error Next(int32_t *v) override
    {
        *v = v_;
        return Error();
    }
    Error DecodeBatch(int32_t *dst, uint32_t n, uint32_t *count) override
    {
        std::fill(dst, dst + n, v_);
        *count = n;
        return Error();
    }

private:
    int32_t v_;
};

// encoding.HybridDecoder (encoding/hybrid-decoder.go:13-187)
class HybridDecoder final : public Decoder {
public:
    HybridDecoder(Device &dev, int bitWidth, bool buffered) : dev_(dev), w_(bitWidth), buffered_(buffered)
    {
        // the reference indexes a 33-entry table here and panics beyond it (hybrid-decoder.go:34)
        if (bitWidth < 0 || bitWidth > 32) throw std::out_of_range("HybridDecoder: bit width out of range");
    }

    Error Init(Reader &r) override
    {
        attach(r, r.Len());
        if (buffered_) r.advance(len_);  // ioutil.ReadAll
        return Error();
    }

    Error InitSize(Reader &r) override
    {
        if (w_ == 0) return Error();  // reads nothing (hybrid-decoder.go:58-60)
        if (r.Len() == 0) return Error(PQ_EOF);
        if (r.Len() < 4) {
            r.advance(r.Len());
            return Error(PQ_UNEXPECTED_EOF);
        }
        const uint8_t *p = r.cursor();
        uint32_t size = (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24;
        r.advance(4);
        attach(r, std::min<size_t>(size, r.Len()));  // io.LimitReader
        if (buffered_) r.advance(len_);
        return Error();
    }

    Error DecodeBatch(int32_t *dst, uint32_t n, uint32_t *count) override
    {
        *count = 0;
        if (w_ != 0 && !init_) return Error(PQ_NOT_INITIALIZED);
        uint64_t consumed = 0;
        int st = pq_hybrid_decode(dev_.ctx(), src_, len_, w_, 0, n, dst, count, &consumed);
        check_library(st, "pq_hybrid_decode");
        if (st == PQ_EOF) consumed = len_;  // the header read that hit EOF had drained the reader
        if (!buffered_ && reader_ && (st == PQ_OK || st == PQ_EOF) && consumed > fed_) {
            // an unbuffered decoder reads through the caller's reader
            reader_->advance((size_t)consumed - fed_);
            fed_ = (size_t)consumed;
        }
        return Error(st);
    }

    // Value-at-a-time pulls are served from a decoded prefix of the stream that grows
    // geometrically; reader positions are exact only for DecodeBatch.
    Error Next(int32_t *v) override
    {
        *v = 0;
        if (w_ == 0) return Error();
        if (!init_) return Error(PQ_NOT_INITIALIZED);
        if (pos_ == cache_.size()) {
            if (tail_) return tail_;
            uint32_t want = (uint32_t)std::max<size_t>(4096, 2 * cache_.size()), got = 0;
            std::vector<int32_t> buf(want);
            Error e = DecodeBatch(buf.data(), want, &got);
            buf.resize(got);
            cache_.swap(buf);
            if (e) tail_ = e;
            if (pos_ == cache_.size()) return tail_ ? tail_ : Error(PQ_EOF);
        }
        *v = cache_[pos_++];
        return Error();
    }

private:
    void attach(Reader &r, size_t n)
    {
        reader_ = &r;
        src_ = r.cursor();
        len_ = n;
        init_ = true;
        fed_ = 0;
        pos_ = 0;
        cache_.clear();
        tail_ = Error();
    }
    Device &dev_;
    int w_;
    bool buffered_, init_ = false;
    Reader *reader_ = nullptr;
    const uint8_t *src_ = nullptr;
    size_t len_ = 0, fed_ = 0, pos_ = 0;
    std::vector<int32_t> cache_;
    Error tail_;
};

// ---- types.DictDecoder (types/dict.go:61-117) ----------------------------------------------
template <typename T>
class DictDecoder {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "fixed-width dictionary entries");

public:
    explicit DictDecoder(Device &dev) : dev_(dev) {}
    std::vector<T> Values;  // the PLAIN-decoded dictionary page

    Error Init(Reader &r)
    {
        if (r.Len() == 0) return Error(PQ_EOF);
        if (*r.cursor() > 32) {
            r.advance(1);
            return Error(PQ_DICT_INVALID_WIDTH);
        }
        reader_ = &r;
        src_ = r.cursor();  // the width byte stays part of what the device call parses
        len_ = r.Len();
        r.advance(1);
        fed_ = 1;
        return Error();
    }

    // dest[0..n): returns the reference's (count, err): count == 0 on an invalid index,
    // the number of values pulled on EOF, n on success.
    std::pair<int, Error> DecodeValues(T *dest, uint32_t n)
    {
        if (src_ == nullptr) return {0, Error(PQ_NOT_INITIALIZED)};  // "no value is inside dictionary"
        uint32_t count = 0;
        uint64_t consumed = 0;
        int st = pq_dict_decode(dev_.ctx(), src_, len_, Values.data(), (uint32_t)Values.size(), sizeof(T), n, dest,
                                &count, &consumed);
        check_library(st, "pq_dict_decode");
        if (st == PQ_OK && consumed > fed_) {
            reader_->advance((size_t)consumed - fed_);
            fed_ = (size_t)consumed;
        }
        return {(int)count, Error(st)};
    }

private:
    Device &dev_;
    Reader *reader_ = nullptr;
    const uint8_t *src_ = nullptr;
    size_t len_ = 0, fed_ = 0;
};

// ---- types.Int32DeltaBPDecoder / Int64DeltaBPDecoder (types/int32.go:103-123, types/int64.go:103-123) ----
// Init runs the header parse the reference does eagerly (delta-binary-packed-decoder.go:174-205):
// a call that pulls zero values reports exactly Init's error.
template <typename T>
class DeltaBPDecoder {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "int32 or int64");

public:
    explicit DeltaBPDecoder(Device &dev) : dev_(dev) {}

    Error Init(Reader &r)
    {
        reader_ = &r;
        src_ = r.cursor();
        len_ = r.Len();
        uint32_t count = 0;
        T none;
        return Error(run(0, &none, &count, nullptr));
    }

    std::pair<int, Error> DecodeValues(T *dest, uint32_t n)
    {
        uint32_t count = 0;
        uint64_t consumed = 0;
        int st = run(n, dest, &count, &consumed);
        if (st == PQ_OK && reader_) reader_->advance((size_t)consumed);
        return {(int)count, Error(st)};
    }

private:
    int run(uint32_t n, T *dest, uint32_t *count, uint64_t *consumed)
    {
        int st;
        if (sizeof(T) == 4)
            st = pq_delta_decode32(dev_.ctx(), src_, len_, n, reinterpret_cast<int32_t *>(dest), count, consumed);
        else
            st = pq_delta_decode64(dev_.ctx(), src_, len_, n, reinterpret_cast<int64_t *>(dest), count, consumed);
        check_library(st, "pq_delta_decode");
        return st;
    }
    Device &dev_;
    Reader *reader_ = nullptr;
    const uint8_t *src_ = nullptr;
    size_t len_ = 0;
};
using Int32DeltaBPDecoder = DeltaBPDecoder<int32_t>;
using Int64DeltaBPDecoder = DeltaBPDecoder<int64_t>;

// ---- types.Int32/Int64/Float/DoublePlainDecoder (types/int32.go:45-72, types/int64.go:45-72,
//      types/float.go:56-79, types/double.go:57-80) and BooleanPlainDecoder (types/boolean.go:46-104) ----
template <typename T>
class PlainDecoder {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "int32, int64, float or double");

public:
    explicit PlainDecoder(Device &dev) : dev_(dev) {}
    Error Init(Reader &r)
    {
        reader_ = &r;
        return Error();
    }
    // (count, err): io.EOF when values are missing, io.ErrUnexpectedEOF for a partial one
    std::pair<int, Error> DecodeValues(T *dest, uint32_t n)
    {
        if (!reader_) return {0, Error(PQ_NOT_INITIALIZED)};
        uint32_t count = 0;
        uint64_t consumed = 0;
        // through a staging vector: dest[count..n) stays untouched, as in the reference's loop
        std::vector<T> tmp(n ? n : 1);
        int st = pq_plain_decode(dev_.ctx(), reader_->cursor(), reader_->Len(), sizeof(T), n, tmp.data(), &count,
                                 &consumed);
        check_library(st, "pq_plain_decode");
        std::memcpy(dest, tmp.data(), (size_t)count * sizeof(T));
        reader_->advance((size_t)consumed);
        return {(int)count, Error(st)};
    }

private:
    Device &dev_;
    Reader *reader_ = nullptr;
};
using Int32PlainDecoder = PlainDecoder<int32_t>;
using Int64PlainDecoder = PlainDecoder<int64_t>;
using FloatPlainDecoder = PlainDecoder<float>;
using DoublePlainDecoder = PlainDecoder<double>;

// One DecodeValues call on a fresh decoder: the bits left over in the last byte are dropped
// here, where the reference keeps them for its next call (types/boolean.go:62-70).
class BooleanPlainDecoder {
public:
    explicit BooleanPlainDecoder(Device &dev) : dev_(dev) {}
    Error Init(Reader &r)
    {
        reader_ = &r;
        return Error();
    }
    std::pair<int, Error> DecodeValues(bool *dest, uint32_t n)
    {
        if (!reader_) return {0, Error(PQ_NOT_INITIALIZED)};
        std::vector<int32_t> tmp(n ? n : 1);
        uint32_t count = 0;
        uint64_t consumed = 0;
        int st = pq_bool_plain_decode(dev_.ctx(), reader_->cursor(), reader_->Len(), n, tmp.data(), &count, &consumed);
        check_library(st, "pq_bool_plain_decode");
        for (uint32_t i = 0; i < count; ++i) dest[i] = tmp[i] == 1;
        reader_->advance((size_t)consumed);
        return {(int)count, Error(st)};
    }

private:
    Device &dev_;
    Reader *reader_ = nullptr;
};

// ---- types.BooleanRLEDecoder (types/boolean.go:173-196) -----------------------------------------
class BooleanRLEDecoder {
public:
    explicit BooleanRLEDecoder(Device &dev) : decoder_(dev, 1, false) {}
    Error Init(Reader &r) { return decoder_.InitSize(r); }
    std::pair<int, Error> DecodeValues(bool *dest, uint32_t n)
    {
        std::vector<int32_t> tmp(n ? n : 1);
        uint32_t count = 0;
        Error e = decoder_.DecodeBatch(tmp.data(), n, &count);
        for (uint32_t i = 0; i < count; ++i) dest[i] = tmp[i] == 1;
        return {(int)count, e};
    }

private:
    HybridDecoder decoder_;
};

// ---- types.ByteArrayDeltaLengthDecoder / ByteArrayDeltaDecoder (types/byte-array.go:168-219, 279-348) ----
// One DecodeValues call on a fresh decoder, like the reference's page readers make.  A value is a
// []byte: here a std::vector<uint8_t>.  Init reports what the reference's Init reports (all
// lengths are decoded there); a DELTA_BYTE_ARRAY page whose values outgrow the room first guessed
// is decoded again with more.
using ByteArray = std::vector<uint8_t>;

class ByteArrayDecoderBase {
public:
    Error Init(Reader &r)
    {
        reader_ = &r;
        std::vector<ByteArray> none;
        return DecodeValues(none.data(), 0).second;
    }

    std::pair<int, Error> DecodeValues(ByteArray *dest, uint32_t n)
    {
        if (!reader_) return {0, Error(PQ_NOT_INITIALIZED)};
        const uint8_t *src = reader_->cursor();
        const size_t len = reader_->Len();
        std::vector<uint32_t> offsets((size_t)n + 1);
        size_t room = front_ ? 4 * len + 4096 : len;
        while (true) {
            std::vector<uint8_t> data(room ? room : 1);
            uint32_t count = 0;
            uint64_t consumed = 0;
            int init_failed = 0;
            const int st = front_ ? pq_delta_ba_decode(dev_.ctx(), src, len, n, offsets.data(), data.data(), room, &count,
                                                       &consumed, &init_failed)
                                  : pq_delta_length_ba_decode(dev_.ctx(), src, len, n, offsets.data(), data.data(), &count,
                                                              &consumed, &init_failed);
            if (st == PQ_ERR_OUT_OF_RANGE && front_ && !init_failed && room < 0xffffffffull) {
                room = std::min<size_t>(0xffffffffull, room * 4);
                continue;
            }
            check_library(st, "pq_delta_(length_)ba_decode");
            for (uint32_t i = 0; i < count; ++i) dest[i].assign(data.begin() + offsets[i], data.begin() + offsets[i + 1]);
            if (!init_failed && n) reader_->advance((size_t)consumed);
            return {(int)count, Error(st)};
        }
    }

protected:
    ByteArrayDecoderBase(Device &dev, bool front) : dev_(dev), front_(front) {}

private:
    Device &dev_;
    bool front_;
    Reader *reader_ = nullptr;
};

class ByteArrayDeltaLengthDecoder final : public ByteArrayDecoderBase {
public:
    explicit ByteArrayDeltaLengthDecoder(Device &dev) : ByteArrayDecoderBase(dev, false) {}
};

class ByteArrayDeltaDecoder final : public ByteArrayDecoderBase {
public:
    explicit ByteArrayDeltaDecoder(Device &dev) : ByteArrayDecoderBase(dev, true) {}
};

// ---- layout.decodePackedArray (layout/helpers.go:68-90) ------------------------------------
// levelDecoder = a Decoder that knows its maximum level (layout/level-decoder.go:5-18).
struct LevelDecoder {
    Decoder *decoder;
    uint16_t max;
    uint16_t maxLevel() const { return max; }
};

inline int bitsLen16(uint16_t x)
{
    int n = 0;
    while (x) {
        ++n;
        x >>= 1;
    }
    return n;
}

struct PackedLevels {
    PackedArray array;
    int notNull = 0;
    Error err;
};

inline PackedLevels decodePackedArray(LevelDecoder d, int count)
{
    PackedLevels out;
    out.array.Reset(bitsLen16(d.maxLevel()));
    std::vector<int32_t> lv((size_t)count);
    uint32_t got = 0;
    out.err = d.decoder->DecodeBatch(lv.data(), (uint32_t)count, &got);
    if (out.err) return out;  // the reference returns (nil, 0, err)
    for (uint32_t i = 0; i < got; ++i) out.notNull += lv[i] == (int32_t)d.maxLevel();
    out.array.AppendLevels(lv.data(), got);
    return out;
}

// ---- data pages (layout/page-v1.go:40-116, layout/page-v2.go:42-149) --------------------------
struct ColumnSpec {
    uint16_t maxR = 0, maxD = 0;
    uint32_t valueKind = PQ_KIND_DICT;  // PQ_KIND_DICT (RLE_DICTIONARY), PQ_KIND_DELTA32, PQ_KIND_DELTA64,
                                        // PQ_KIND_PLAIN, PQ_KIND_BOOL_PLAIN, PQ_KIND_BOOL_RLE (int32 0 / 1)
    uint32_t elemSize = 4;              // bytes per value
    const void *dict = nullptr;         // PLAIN-decoded dictionary page (DICT only)
    uint32_t dictLen = 0;
};

struct PageValues {
    int n = 0;                   // level slots read (0 on any error)
    int notNull = 0;             // values written to the destination
    PackedArray dLevel, rLevel;  // valid when n > 0
    Error err;
};

// One decompressed data page.  version 1: block = [u32 len][rep][u32 len][def][values];
// version 2: block = [rep][def][values] with the two lengths from the page header.
struct DataPage {
    int version = 1;
    const uint8_t *block = nullptr;
    size_t blockLen = 0;
    int32_t numValues = 0;
    int32_t repLen = 0, defLen = 0;  // V2 only
};

namespace detail {
inline pq_page_desc page_desc(const DataPage &p, const ColumnSpec &c, uint64_t src_off, uint64_t dst_off,
                              uint64_t lvl_off)
{
    pq_page_desc d;
    std::memset(&d, 0, sizeof d);
    d.kind = p.version == 2 ? PQ_KIND_PAGE_V2 : PQ_KIND_PAGE_V1;
    d.src_off = src_off;
    d.src_len = p.blockLen;
    d.dst_off = dst_off;
    d.rep_off = lvl_off;
    d.def_off = lvl_off + 4ull * (uint64_t)p.numValues;
    d.n_values = (uint32_t)p.numValues;
    d.max_rep = c.maxR;
    d.max_def = c.maxD;
    d.rep_len = (uint32_t)p.repLen;
    d.def_len = (uint32_t)p.defLen;
    d.value_kind = c.valueKind;
    d.elem_size = c.elemSize;
    d.dict_len = c.dictLen;
    return d;
}
}  // namespace detail

// PageReader for one page: read() then ReadValues() (layout/page.go:14-21).  The whole page is
// decoded by one device call when ReadValues asks for all of it, which is what readPageData does.
class DataPageReader {
public:
    DataPageReader(Device &dev, const ColumnSpec &col) : dev_(dev), col_(col) {}

    Error read(const DataPage &page)
    {
        if (page.numValues < 0 || page.repLen < 0 || page.defLen < 0) return Error(PQ_ERR_BAD_ARGUMENT);
        page_ = page;
        position_ = 0;
        return Error();
    }
    int32_t NumValues() const { return page_.numValues; }

    // values: room for NumValues() entries of col.elemSize bytes
    PageValues ReadValues(void *values, int size)
    {
        PageValues out;
        size = std::min(size, (int)page_.numValues - position_);
        if (size <= 0) return out;
        if (position_ != 0 || size != page_.numValues) {
            out.err = Error(PQ_ERR_BAD_ARGUMENT);  // partial reads of a page are not batched
            return out;
        }
        std::vector<int32_t> lv(2 * (size_t)size);
        pq_page_desc d = detail::page_desc(page_, col_, 0, 0, 0);
        pq_page_result r;
        int st = pq_decode_pages(dev_.ctx(), &d, 1, page_.block, page_.blockLen, values,
                                 (size_t)size * col_.elemSize, lv.data(), lv.size() * 4, col_.dict,
                                 (size_t)col_.dictLen * col_.elemSize, &r);
        check_library(st, "pq_decode_pages");
        out.err = Error(r.status, r.stage, r.detail);
        if (out.err) return out;
        out.n = (int)r.n_levels;
        out.notNull = (int)r.not_null;
        out.rLevel.Reset(bitsLen16(col_.maxR));
        out.dLevel.Reset(bitsLen16(col_.maxD));
        out.rLevel.AppendLevels(lv.data(), (size_t)size);
        out.dLevel.AppendLevels(lv.data() + size, (size_t)size);
        position_ += size;
        return out;
    }

private:
    Device &dev_;
    ColumnSpec col_;
    DataPage page_;
    int position_ = 0;
};

// ---- readPageData for a whole column chunk (file-reader.go:268-302) --------------------------
// The reference loops page by page, appending values and levels; here every page of the chunk
// goes to the device in one call and the appends happen afterwards in page order.  Stops at the
// first failing page like the reference (the pages before it are appended, the error returned).
struct ChunkData {
    std::vector<uint8_t> values;  // dense non-null values, elemSize bytes each, pages concatenated
    size_t numValues = 0;         // entries in `values`
    PackedArray dLevels, rLevels;
    Error err;
    int failedPage = -1;
};

inline ChunkData readPageData(Device &dev, const ColumnSpec &col, const std::vector<DataPage> &pages)
{
    ChunkData out;
    out.rLevels.Reset(bitsLen16(col.maxR));
    out.dLevels.Reset(bitsLen16(col.maxD));
    if (pages.empty()) return out;
    std::vector<pq_page_desc> descs;
    std::vector<uint8_t> src;
    uint64_t slots = 0;
    for (const DataPage &p : pages) {
        size_t at = (src.size() + 15) / 16 * 16;  // page starts 16-byte aligned in the staging copy
        src.resize(at + p.blockLen);
        if (p.blockLen) std::memcpy(src.data() + at, p.block, p.blockLen);
        descs.push_back(detail::page_desc(p, col, at, slots * col.elemSize, slots * 8));
        slots += (uint64_t)p.numValues;
    }
    std::vector<uint8_t> vals(std::max<size_t>(16, slots * col.elemSize));
    std::vector<int32_t> lv(std::max<size_t>(4, 2 * slots));
    std::vector<pq_page_result> res(pages.size());
    int st = pq_decode_pages(dev.ctx(), descs.data(), descs.size(), src.data(), src.size(), vals.data(), vals.size(),
                             lv.data(), lv.size() * 4, col.dict, (size_t)col.dictLen * col.elemSize, res.data());
    check_library(st, "pq_decode_pages");
    out.values.reserve(slots * col.elemSize);
    for (size_t i = 0; i < pages.size(); ++i) {
        if (res[i].status != PQ_OK) {
            out.err = Error(res[i].status, res[i].stage, res[i].detail);
            out.failedPage = (int)i;
            break;
        }
        size_t n = (size_t)pages[i].numValues;
        const int32_t *rep = lv.data() + descs[i].rep_off / 4;
        out.rLevels.AppendLevels(rep, n);
        out.dLevels.AppendLevels(rep + n, n);
        const uint8_t *v = vals.data() + descs[i].dst_off;
        out.values.insert(out.values.end(), v, v + (size_t)res[i].not_null * col.elemSize);
        out.numValues += res[i].not_null;
    }
    return out;
}

// ---- layout.ChunkReader.readPages (layout/chunk.go:129-217) ---------------------------------------
// The page headers of column chunks, all chunks of a row group in one device call.  ColumnChunkMeta
// carries what the loop takes from parquet.ColumnMetaData and schema.Column.
struct ColumnChunkMeta {
    int64_t TotalCompressedSize = 0;
    int64_t DataPageOffset = 0;
    bool hasDictionaryPageOffset = false;
    int64_t DictionaryPageOffset = 0;
    uint32_t Type = 1;          // parquet.Type
    int32_t TypeLength = -1;    // < 0: nil
    uint16_t MaxRepetitionLevel = 0, MaxDefinitionLevel = 0;
    uint32_t Codec = 0;         // parquet.CompressionCodec
};

struct ChunkPages {
    std::vector<pq_page_info> pages;  // in file order, the dictionary page included
    int64_t count = 0;                // offsetReader.Count()
    Error err;                        // the first error; pages in front of it are kept
};

// file: the bytes of the source from offset fileOffset on (a whole file: fileOffset 0).  Each
// chunk is read from its DictionaryPageOffset when it has one, else from DataPageOffset (:58-61).
inline std::vector<ChunkPages> readPages(Device &dev, const uint8_t *file, size_t fileLen, int64_t fileOffset,
                                         const std::vector<ColumnChunkMeta> &chunks, uint32_t maxPagesPerChunk = 4096)
{
    std::vector<pq_chunk_desc> descs(chunks.size());
    for (size_t i = 0; i < chunks.size(); ++i) {
        const ColumnChunkMeta &m = chunks[i];
        pq_chunk_desc &d = descs[i];
        std::memset(&d, 0, sizeof d);
        const int64_t start = (m.hasDictionaryPageOffset ? m.DictionaryPageOffset : m.DataPageOffset) - fileOffset;
        if (start < 0 || (uint64_t)start > fileLen) throw std::runtime_error("readPages: chunk outside the source");
        d.src_off = (uint64_t)start;
        d.src_len = fileLen - (uint64_t)start;
        d.total_compressed_size = m.TotalCompressedSize;
        d.data_page_offset = m.DataPageOffset;
        d.dictionary_page_offset = m.DictionaryPageOffset;
        d.has_dictionary_page_offset = m.hasDictionaryPageOffset ? 1u : 0u;
        d.physical_type = m.Type;
        d.type_length = m.TypeLength;
        d.max_rep = m.MaxRepetitionLevel;
        d.max_def = m.MaxDefinitionLevel;
        d.codec = m.Codec;
        d.info_first = (uint32_t)(i * maxPagesPerChunk);
        d.info_cap = maxPagesPerChunk;
    }
    std::vector<pq_page_info> infos(chunks.size() * (size_t)maxPagesPerChunk);
    std::vector<pq_chunk_result> res(chunks.size());
    check_library(pq_walk_chunks(dev.ctx(), file, fileLen, descs.data(), descs.size(), infos.data(), infos.size(),
                                 res.data()),
                  "pq_walk_chunks");
    std::vector<ChunkPages> out(chunks.size());
    for (size_t i = 0; i < chunks.size(); ++i) {
        if (res[i].status >= PQ_ERR_BAD_ARGUMENT) check_library(res[i].status, "readPages: more pages than maxPagesPerChunk");
        out[i].pages.assign(infos.begin() + i * maxPagesPerChunk, infos.begin() + i * maxPagesPerChunk + res[i].n_pages);
        out[i].count = res[i].count;
        out[i].err = Error(res[i].status);
    }
    return out;
}

// ---- level post-processing ----------------------------------------------------------------
// packLevels: the PackedArray the reference would hold after appending these levels one by one
// (file-reader.go:285-296), packed on the device.
inline Error packLevels(Device &dev, const int32_t *levels, size_t n, int bitWidth, PackedArray *out)
{
    std::vector<uint8_t> bytes(std::max<size_t>(1, (n + 7) / 8 * (size_t)bitWidth));
    int st = pq_levels_pack(dev.ctx(), levels, n, bitWidth, bytes.data(), bytes.size());
    check_library(st, "pq_levels_pack");
    if (st != PQ_OK) return Error(st);
    return out->Adopt(bitWidth, (int)n, bytes.data(), bytes.size());
}

// columnEntries: ColumnStore.Get (datastore/column.go:127-177) called to the end of the store, as
// arrays.  Entry k is the k-th call: IsNull(k) == the call returned nil, entryDef[k] the definition
// level it returned, and its values are values[offsets[k] .. offsets[k + 1]) of the column's dense
// value array (one value for a column that is not REPEATED).
struct ColumnEntries {
    uint32_t numEntries = 0, numValues = 0;
    std::vector<uint32_t> offsets;   // numEntries + 1
    std::vector<uint8_t> validity;   // bit k, LSB first
    std::vector<int32_t> entryDef;   // numEntries
    bool IsNull(uint32_t k) const { return ((validity[k >> 3] >> (k & 7)) & 1) == 0; }
};

inline ColumnEntries columnEntries(Device &dev, const int32_t *dLevels, const int32_t *rLevels, size_t n,
                                   uint16_t maxD, uint16_t maxR, bool repeated)
{
    ColumnEntries e;
    e.offsets.resize(n + 1);
    e.validity.resize((n + 31) / 32 * 4 + 4);
    e.entryDef.resize(std::max<size_t>(n, 1));
    pq_entries_result r;
    int st = pq_column_entries(dev.ctx(), dLevels, rLevels, n, maxD, maxR, repeated ? 1 : 0, e.offsets.data(),
                               e.validity.data(), e.entryDef.data(), &r);
    check_library(st, "pq_column_entries");
    e.numEntries = r.n_entries;
    e.numValues = r.n_values;
    e.offsets.resize((size_t)r.n_entries + 1);
    e.entryDef.resize(r.n_entries);
    return e;
}

}  // namespace pqhost
