// Tests of the host-side C++ mirror (parquet-go_amd/host/pq_host.hpp) on a real GPU.
//
// The cases follow the reference's own tests where it has them
// (encoding/hybrid-decoder_test.go, encoding/hybrid-encoder_test.go,
// encoding/delta-binary-packed-decoder_test.go, encoding/packed-array_test.go) and check the
// interfaces the reference does not test (DictDecoder, decodePackedArray, page readers,
// readPageData) against the CPU oracle on seeded random inputs.
//
// Build: make -C tests/host     Run: tests/host/host_mirror_test   (exit code 0 = all passed)
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <numeric>

#include "pq_host.hpp"
#include "pq_oracle.h"

using namespace pqhost;

static int g_failed = 0, g_checks = 0;
#define CHECK(cond)                                                              \
    do {                                                                         \
        ++g_checks;                                                              \
        if (!(cond)) {                                                           \
            ++g_failed;                                                          \
            std::printf("    FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);    \
        }                                                                        \
    } while (0)
#define CHECK_EQ(a, b)                                                                               \
    do {                                                                                             \
        ++g_checks;                                                                                  \
        long long _a = (long long)(a), _b = (long long)(b);                                          \
        if (_a != _b) {                                                                              \
            ++g_failed;                                                                              \
            std::printf("    FAILED %s:%d: %s == %s (%lld vs %lld)\n", __FILE__, __LINE__, #a, #b, _a, _b); \
        }                                                                                            \
    } while (0)

static uint64_t g_rng = 0x9E3779B97F4A7C15ull;
static uint64_t rnd()
{
    uint64_t z = (g_rng += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static std::vector<uint8_t> hybrid_stream(const std::vector<int32_t> &v, int w)
{
    std::vector<uint8_t> out(v.size() * 5 + 64);
    size_t n = 0;
    int st = pqo_hybrid_encode(v.data(), v.size(), w, out.data(), out.size(), &n);
    if (st != 0) std::abort();
    out.resize(n);
    return out;
}

static std::vector<uint8_t> prefixed(const std::vector<uint8_t> &s)
{
    std::vector<uint8_t> out = {(uint8_t)s.size(), (uint8_t)(s.size() >> 8), (uint8_t)(s.size() >> 16),
                                (uint8_t)(s.size() >> 24)};
    out.insert(out.end(), s.begin(), s.end());
    return out;
}

// ---- encoding/hybrid-decoder_test.go:11-35 ------------------------------------------------
static void TestHybridDecoder(Device &dev)
{
    std::vector<uint8_t> data = {0x03, 1 | 2 << 2 | 3 << 4};  // one bit-packed group, second byte missing
    Reader r(data);
    HybridDecoder d(dev, 2, false);
    CHECK(!d.Init(r));
    for (int32_t want : {1, 2, 3}) {
        int32_t v = -1;
        CHECK(!d.Next(&v));
        CHECK_EQ(v, want);
    }
    CHECK_EQ(r.Len(), 0);  // the whole reader was consumed
    // the rest of the zero-filled group, then EOF from the next header read
    for (int i = 0; i < 5; ++i) {
        int32_t v = -1;
        CHECK(!d.Next(&v));
        CHECK_EQ(v, 0);
    }
    int32_t v;
    CHECK(d.Next(&v).is(PQ_EOF));
}

// ---- streams the reference's encoder tests expect (encoding/hybrid-encoder_test.go) ------------
static void TestHybridDecoder_EncoderStreams(Device &dev)
{
    {  // RLE only, w=3: 100 x 4 then 100 x 5  (:15-54)
        std::vector<uint8_t> data = {0xc8, 0x01, 0x04, 0xc8, 0x01, 0x05};
        Reader r(data);
        HybridDecoder d(dev, 3, false);
        CHECK(!d.Init(r));
        std::vector<int32_t> out(201);
        uint32_t n = 0;
        Error e = d.DecodeBatch(out.data(), 201, &n);
        CHECK(e.is(PQ_EOF));
        CHECK_EQ(n, 200);
        for (int i = 0; i < 200; ++i) CHECK_EQ(out[i], i < 100 ? 4 : 5);
    }
    {  // w == 0: zeros forever, nothing is read, never EOF  (:80-101, hybrid-decoder.go:74-76)
        std::vector<uint8_t> data = {0x14};
        Reader r(data);
        HybridDecoder d(dev, 0, false);
        CHECK(!d.InitSize(r));
        CHECK_EQ(r.Len(), 1);
        std::vector<int32_t> out(1000, -1);
        uint32_t n = 0;
        CHECK(!d.DecodeBatch(out.data(), 1000, &n));
        CHECK_EQ(n, 1000);
        CHECK(std::all_of(out.begin(), out.end(), [](int32_t x) { return x == 0; }));
    }
    {  // never initialised with w > 0  (hybrid-decoder.go:78-80)
        HybridDecoder d(dev, 1, true);
        int32_t v;
        CHECK(d.Next(&v).is(PQ_NOT_INITIALIZED));
    }
    {  // InitSize: prefix, limit, and the reader position of a buffered decoder
        std::vector<int32_t> vals(1000);
        for (auto &x : vals) x = (int32_t)(rnd() % 3 == 0 ? rnd() % 8 : 5);
        std::vector<uint8_t> s = prefixed(hybrid_stream(vals, 3));
        size_t body = s.size() - 4;
        s.push_back(0xEE);  // bytes after the limited stream stay unread
        s.push_back(0xEE);
        Reader r(s);
        HybridDecoder d(dev, 3, true);
        CHECK(!d.InitSize(r));
        CHECK_EQ(r.Len(), 2);
        std::vector<int32_t> out(1000);
        uint32_t n = 0;
        CHECK(!d.DecodeBatch(out.data(), 1000, &n));
        CHECK(out == vals);
        (void)body;
        std::vector<uint8_t> shortbuf = {1, 0, 0};
        Reader r2(shortbuf);
        CHECK(d.InitSize(r2).is(PQ_UNEXPECTED_EOF));
        Reader r3(nullptr, 0);
        CHECK(d.InitSize(r3).is(PQ_EOF));
    }
}

// ---- encoding/delta-binary-packed-decoder_test.go ---------------------------------------------
template <typename T>
static void TestDeltaBinaryPackDecoder(Device &dev)
{
    {  // Init (:29-38) and Next_NoValues (:141-153)
        std::vector<uint8_t> data = {128, 1, 4, 0, 0};
        Reader r(data);
        DeltaBPDecoder<T> d(dev);
        CHECK(!d.Init(r));
        T v = 0;
        auto res = d.DecodeValues(&v, 1);
        CHECK(res.second.is(PQ_EOF));
        CHECK_EQ(res.first, 0);
    }
    {  // Init_EmptyBuffer (:50-57)
        Reader r(nullptr, 0);
        DeltaBPDecoder<T> d(dev);
        CHECK(d.Init(r).is(PQ_EOF));
    }
    for (const std::vector<uint8_t> &input :
         std::vector<std::vector<uint8_t>>{{0, 1, 4, 0, 0}, {127, 1, 4, 0, 0}, {129, 1, 4, 0, 0}}) {
        Reader r(input);  // Init_InvalidBlockSize (:59-76)
        DeltaBPDecoder<T> d(dev);
        CHECK(d.Init(r).is(PQ_INVALID_BLOCK_SIZE));
    }
    for (const std::vector<uint8_t> &input :
         std::vector<std::vector<uint8_t>>{{128, 1, 0, 0, 0}, {128, 1, 3, 0, 0}, {128, 1, 128, 2, 0}}) {
        Reader r(input);  // Init_InvalidMiniblockCount (:78-95)
        DeltaBPDecoder<T> d(dev);
        CHECK(d.Init(r).is(PQ_INVALID_MINIBLOCK_COUNT));
    }
    {  // Next (:120-139)
        std::vector<uint8_t> data = {128, 1, 4, 8, 14, 3, 2, 0, 0, 0, 192, 63, 0, 0, 0, 0, 0, 0};
        Reader r(data);
        DeltaBPDecoder<T> d(dev);
        CHECK(!d.Init(r));
        std::vector<T> out(8);
        auto res = d.DecodeValues(out.data(), 8);
        CHECK(!res.second);
        CHECK_EQ(res.first, 8);
        const T want[8] = {7, 5, 3, 1, 2, 3, 4, 5};
        for (int i = 0; i < 8; ++i) CHECK_EQ(out[i], want[i]);
    }
    {  // multi-block stream written by the restated reference encoder, against the oracle
        const uint32_t n = 5000;
        std::vector<T> vals(n);
        T x = (T)rnd();
        for (auto &v : vals) v = (x += (T)(rnd() % 4096) - 100);
        std::vector<uint8_t> s(n * 10 + 256);
        size_t len = 0;
        int st = sizeof(T) == 4
                     ? pqo_delta_encode32((const int32_t *)vals.data(), n, 128, 4, s.data(), s.size(), &len)
                     : pqo_delta_encode64((const int64_t *)vals.data(), n, 128, 4, s.data(), s.size(), &len);
        CHECK_EQ(st, 0);
        s.resize(len);
        s.push_back(0x77);  // a byte the decoder must leave in the reader
        Reader r(s);
        DeltaBPDecoder<T> d(dev);
        CHECK(!d.Init(r));
        std::vector<T> out(n);
        auto res = d.DecodeValues(out.data(), n);
        CHECK(!res.second);
        CHECK_EQ(res.first, n);
        CHECK(out == vals);
        std::vector<T> ref(n);
        uint32_t nd = 0, vc = 0;
        uint64_t consumed = 0;
        int ist = 0;
        int ost = sizeof(T) == 4 ? pqo_delta_decode32(s.data(), s.size(), n, (int32_t *)ref.data(), &nd, &consumed, &vc, &ist)
                                 : pqo_delta_decode64(s.data(), s.size(), n, (int64_t *)ref.data(), &nd, &consumed, &vc, &ist);
        CHECK_EQ(ost, 0);
        CHECK(ref == out);
        CHECK_EQ(s.size() - r.Len(), consumed);
    }
}

// ---- encoding/packed-array_test.go --------------------------------------------------------------
static void TestPackedArray()
{
    PackedArray a;
    CHECK(a.Reset(33).status != PQ_OK);
    CHECK(a.Reset(-1).status != PQ_OK);
    CHECK(!a.Reset(3));
    for (int i = 0; i < 8; ++i) a.AppendSingle(i);
    CHECK_EQ(a.Count(), 8);
    CHECK_EQ(a.Data().size(), 0);  // a full buffer is flushed only by the next append
    a.Flush();
    const uint8_t want[3] = {0x88, 0xc6, 0xfa};  // :211-229
    CHECK_EQ(a.Data().size(), 3);
    for (int i = 0; i < 3; ++i) CHECK_EQ(a.Data()[i], want[i]);
    for (int i = 0; i < 8; ++i) {
        int32_t v = -1;
        CHECK(!a.At(i, &v));
        CHECK_EQ(v, i);
    }
    int32_t v;
    CHECK(a.At(8, &v).status != PQ_OK);
    CHECK(a.At(-1, &v).status != PQ_OK);

    // every width: packed bytes and tail against the oracle's pack8, reads through At()
    for (int bw = 0; bw <= 32; ++bw) {
        PackedArray p;
        CHECK(!p.Reset(bw));
        const int n = 8 * 5 + 3;
        std::vector<int32_t> vals(n);
        for (auto &x : vals) x = (int32_t)(bw == 32 ? rnd() : rnd() & ((1ull << bw) - 1));
        p.AppendLevels(vals.data(), n);
        CHECK_EQ(p.Count(), n);
        CHECK_EQ(p.Data().size(), (size_t)5 * bw);
        std::vector<uint8_t> ref((size_t)5 * bw + 1);
        for (int g = 0; g < 5; ++g) pqo_pack8_32(vals.data() + 8 * g, bw, ref.data() + (size_t)g * bw);
        ref.resize((size_t)5 * bw);
        CHECK(ref == p.Data());
        for (int i = 0; i < n; ++i) {
            int32_t got = -1;
            CHECK(!p.At(i, &got));
            CHECK_EQ(got, vals[i]);
        }
        PackedArray q;
        q.Reset(bw);
        q.AppendSingle(bw ? 1 : 0);
        CHECK(!q.AppendArray(&p));
        CHECK_EQ(q.Count(), n + 1);
        int32_t last = -1;
        CHECK(!q.At(n, &last));
        CHECK_EQ(last, vals[n - 1]);
        PackedArray other;
        other.Reset(bw == 0 ? 1 : bw - 1);
        CHECK(q.AppendArray(&other).status != PQ_OK);
        CHECK(q.AppendArray(nullptr).status != PQ_OK);
    }
}

// ---- level post-processing: PackedArray built on the device, ColumnStore.Get as arrays --------
static void TestLevelPostProcessing(Device &dev)
{
    for (int w : {1, 2, 3, 7, 16}) {
        for (size_t n : {size_t(0), size_t(1), size_t(8), size_t(9), size_t(16), size_t(1000), size_t(70001)}) {
            std::vector<int32_t> lv(n);
            for (auto &v : lv) v = (int32_t)(rnd() & ((1u << w) - 1));
            PackedArray want, got;
            CHECK(!want.Reset(w));
            want.AppendLevels(lv.data(), n);
            CHECK(!packLevels(dev, lv.data(), n, w, &got));
            CHECK_EQ(got.Count(), want.Count());
            CHECK(got.Data() == want.Data());
            bool same = true;
            for (size_t i = 0; i < n; ++i) {
                int32_t a = -1, b = -2;
                same = same && !got.At((int)i, &a) && !want.At((int)i, &b) && a == b;
            }
            CHECK(same);
            got.AppendSingle(1);  // the adopted array keeps growing like the original
            want.AppendSingle(1);
            got.Flush();
            want.Flush();
            CHECK(got.Data() == want.Data());
        }
    }
    // [a b c] null [d e] of a REPEATED leaf
    {
        const int32_t def[] = {1, 1, 1, 0, 1, 1}, rep[] = {0, 1, 1, 0, 0, 1};
        ColumnEntries e = columnEntries(dev, def, rep, 6, 1, 1, true);
        CHECK_EQ(e.numEntries, 3);
        CHECK_EQ(e.numValues, 5);
        CHECK((e.offsets == std::vector<uint32_t>{0, 3, 3, 5}));
        CHECK((e.entryDef == std::vector<int32_t>{1, 0, 1}));
        CHECK(!e.IsNull(0) && e.IsNull(1) && !e.IsNull(2));
    }
    // against the oracle
    for (int rp = 0; rp < 2; ++rp) {
        const size_t n = 50021;
        std::vector<int32_t> def(n), rep(n);
        for (size_t i = 0; i < n; ++i) {
            rep[i] = i ? (int32_t)(rnd() % 3) : 0;
            def[i] = (rnd() % 10) < 7 ? 4 : (int32_t)(rnd() % 5);
        }
        ColumnEntries e = columnEntries(dev, def.data(), rep.data(), n, 4, 2, rp != 0);
        std::vector<uint32_t> off(n + 1);
        std::vector<uint8_t> val((n + 31) / 32 * 4);
        std::vector<int32_t> ed(n);
        uint32_t ne = 0, nv = 0;
        pqo_column_entries(def.data(), rep.data(), n, 4, 2, rp, off.data(), val.data(), ed.data(), &ne, &nv);
        CHECK_EQ(e.numEntries, ne);
        CHECK_EQ(e.numValues, nv);
        off.resize(ne + 1);
        ed.resize(ne);
        CHECK(e.offsets == off);
        CHECK(e.entryDef == ed);
        CHECK(std::equal(val.begin(), val.end(), e.validity.begin()));
    }
}

// ---- types/float_test.go:120-171, types/double_test.go:120-171 -----------------------------------
static void TestPlainDecoders(Device &dev)
{
    {  // FloatPlainDecoder_DecodeValues and _NotEnoughValues
        std::vector<uint8_t> data = {0x00, 0x00, 0x80, 0x3f, 0x00, 0x00, 0x00, 0x40, 0x00, 0x00, 0x40, 0x40};
        Reader r(data);
        FloatPlainDecoder d(dev);
        CHECK(!d.Init(r));
        float dest[3] = {0, 0, 0};
        auto res = d.DecodeValues(dest, 3);
        CHECK(!res.second);
        CHECK_EQ(res.first, 3);
        CHECK(dest[0] == 1.f && dest[1] == 2.f && dest[2] == 3.f);
        data.resize(8);
        Reader r2(data);
        CHECK(!d.Init(r2));
        float d2[3] = {0, 0, 0};
        res = d.DecodeValues(d2, 3);
        CHECK(res.second.is(PQ_EOF));
        CHECK_EQ(res.first, 2);
        CHECK(d2[0] == 1.f && d2[1] == 2.f && d2[2] == 0.f);
    }
    {  // DoublePlainDecoder_DecodeValues and _NotEnoughValues
        std::vector<uint8_t> data = {0, 0, 0, 0, 0, 0, 0xF0, 0x3F, 0, 0, 0, 0, 0, 0, 0, 0x40, 0, 0, 0, 0, 0, 0, 8, 0x40};
        Reader r(data);
        DoublePlainDecoder d(dev);
        CHECK(!d.Init(r));
        double dest[3] = {0, 0, 0};
        auto res = d.DecodeValues(dest, 3);
        CHECK(!res.second);
        CHECK_EQ(res.first, 3);
        CHECK(dest[0] == 1. && dest[1] == 2. && dest[2] == 3.);
        CHECK_EQ(r.Len(), 0);
        data.resize(16);
        Reader r2(data);
        CHECK(!d.Init(r2));
        res = d.DecodeValues(dest, 3);
        CHECK(res.second.is(PQ_EOF));
        CHECK_EQ(res.first, 2);
    }
    {  // int32 / int64 against the oracle, with a partial last value
        const uint32_t n = 70001;
        std::vector<uint8_t> s((size_t)n * 8 - 3);
        for (auto &b : s) b = (uint8_t)rnd();
        Reader r(s);
        Int64PlainDecoder d(dev);
        CHECK(!d.Init(r));
        std::vector<int64_t> out(n), ref(n);
        auto res = d.DecodeValues(out.data(), n);
        uint32_t nd = 0;
        uint64_t consumed = 0;
        int ost = pqo_plain_decode(s.data(), s.size(), 8, n, ref.data(), &nd, &consumed);
        CHECK_EQ(res.second.status, ost);
        CHECK(res.second.is(PQ_UNEXPECTED_EOF));
        CHECK_EQ(res.first, nd);
        CHECK(std::equal(out.begin(), out.begin() + nd, ref.begin()));
        Reader r32(s);
        Int32PlainDecoder d32(dev);
        CHECK(!d32.Init(r32));
        std::vector<int32_t> o32(1000), r32v(1000);
        auto res32 = d32.DecodeValues(o32.data(), 1000);
        CHECK(!res32.second);
        CHECK_EQ(pqo_plain_decode(s.data(), s.size(), 4, 1000, r32v.data(), &nd, &consumed), 0);
        CHECK(o32 == r32v);
        CHECK_EQ(s.size() - r32.Len(), 4000);
    }
    {  // BooleanPlainDecoder: bit j of byte i is value 8 i + j
        std::vector<uint8_t> s = {0xA5, 0x01};
        Reader r(s);
        BooleanPlainDecoder d(dev);
        CHECK(!d.Init(r));
        bool dest[12];
        auto res = d.DecodeValues(dest, 12);
        CHECK(!res.second);
        CHECK_EQ(res.first, 12);
        const bool want[12] = {1, 0, 1, 0, 0, 1, 0, 1, 1, 0, 0, 0};
        for (int i = 0; i < 12; ++i) CHECK_EQ(dest[i], want[i]);
        Reader r2(s);
        CHECK(!d.Init(r2));
        bool big[40];
        res = d.DecodeValues(big, 40);
        CHECK(res.second.is(PQ_EOF));
        CHECK_EQ(res.first, 16);
    }
    {  // BooleanRLEDecoder: u32 length, hybrid width 1; checked against the oracle
        std::vector<int32_t> bits(5000);
        for (size_t i = 0; i < bits.size(); ++i) bits[i] = (i / 37) & 1 ? 1 : (int32_t)(rnd() & 1);
        std::vector<uint8_t> enc(bits.size() + 64);
        size_t len = 0;
        CHECK_EQ(pqo_hybrid_encode(bits.data(), bits.size(), 1, enc.data(), enc.size(), &len), 0);
        std::vector<uint8_t> s = {(uint8_t)len, (uint8_t)(len >> 8), (uint8_t)(len >> 16), (uint8_t)(len >> 24)};
        s.insert(s.end(), enc.begin(), enc.begin() + len);
        Reader r(s);
        BooleanRLEDecoder d(dev);
        CHECK(!d.Init(r));
        std::vector<char> got(bits.size() + 100);
        auto res = d.DecodeValues(reinterpret_cast<bool *>(got.data()), (uint32_t)bits.size());
        CHECK(!res.second);
        CHECK_EQ(res.first, bits.size());
        bool same = true;
        for (size_t i = 0; i < bits.size(); ++i) same = same && (got[i] != 0) == (bits[i] == 1);
        CHECK(same);
        std::vector<uint8_t> shortS(s.begin(), s.begin() + 3);
        Reader r2(shortS);
        CHECK(d.Init(r2).is(PQ_UNEXPECTED_EOF));
    }
}

// ---- types.DictDecoder (no reference test; checked against the oracle) ---------------------------
template <typename T>
static void TestDictDecoder(Device &dev)
{
    const uint32_t n = 30000, dict_len = 1000;
    DictDecoder<T> d(dev);
    d.Values.resize(dict_len);
    for (auto &x : d.Values) x = (T)rnd();
    std::vector<int32_t> keys(n);
    for (uint32_t i = 0; i < n; ++i) keys[i] = (int32_t)((rnd() % 7 == 0 || i == 0) ? rnd() % dict_len : keys[i - 1]);
    std::vector<uint8_t> s = {10};
    std::vector<uint8_t> body = hybrid_stream(keys, 10);
    s.insert(s.end(), body.begin(), body.end());
    {
        Reader r(s);
        CHECK(!d.Init(r));
        std::vector<T> out(n), ref(n);
        auto res = d.DecodeValues(out.data(), n);
        CHECK(!res.second);
        CHECK_EQ(res.first, n);
        uint32_t nd = 0;
        uint64_t consumed = 0;
        CHECK_EQ(pqo_dict_decode(s.data(), s.size(), d.Values.data(), dict_len, sizeof(T), n, ref.data(), &nd, &consumed), 0);
        CHECK(out == ref);
        CHECK_EQ(s.size() - r.Len(), consumed);
        for (uint32_t i = 0; i < n; i += 997) CHECK(out[i] == d.Values[keys[i]]);
    }
    {  // asking for more than the stream holds: (count, EOF)
        Reader r(s);
        CHECK(!d.Init(r));
        std::vector<T> out(n + 50);
        auto res = d.DecodeValues(out.data(), n + 50);
        CHECK(res.second.is(PQ_EOF));
        CHECK(res.first >= (int)n && res.first < (int)n + 8);
    }
    {  // an index beyond the dictionary: (0, invalid index)  (types/dict.go:104-111)
        d.Values.resize(500);
        Reader r(s);
        CHECK(!d.Init(r));
        std::vector<T> out(n);
        auto res = d.DecodeValues(out.data(), n);
        CHECK(res.second.is(PQ_DICT_INVALID_INDEX));
        CHECK_EQ(res.first, 0);
    }
    {  // width byte checks (types/dict.go:66-81)
        std::vector<uint8_t> bad = {33, 0};
        Reader r(bad);
        CHECK(d.Init(r).is(PQ_DICT_INVALID_WIDTH));
        Reader e(nullptr, 0);
        CHECK(d.Init(e).is(PQ_EOF));
        DictDecoder<T> fresh(dev);
        T v;
        CHECK(fresh.DecodeValues(&v, 1).second.status != PQ_OK);
    }
}

// ---- decodePackedArray (no reference test; against the oracle) -----------------------------------
static void TestDecodePackedArray(Device &dev)
{
    for (uint16_t max : {1, 3, 7, 200}) {
        const int n = 20003;
        int bw = bitsLen16(max);
        std::vector<int32_t> lv(n);
        for (int i = 0; i < n; ++i) lv[i] = (int32_t)(rnd() % 4 ? max : rnd() % (max + 1u));
        std::vector<uint8_t> s = prefixed(hybrid_stream(lv, bw));
        Reader r(s);
        HybridDecoder hd(dev, bw, true);
        CHECK(!hd.InitSize(r));
        PackedLevels got = decodePackedArray(LevelDecoder{&hd, max}, n);
        CHECK(!got.err);

        std::vector<int32_t> ref(n);
        std::vector<uint8_t> packed((size_t)(n / 8 + 1) * bw + 8);
        uint64_t packed_len = 0, consumed = 0;
        int32_t tail[8];
        uint32_t tail_len = 0, not_null = 0;
        CHECK_EQ(pqo_levels_decode(s.data(), s.size(), max, 1, n, ref.data(), packed.data(), &packed_len, tail, &tail_len,
                                   &not_null, &consumed), 0);
        CHECK_EQ(got.notNull, not_null);
        CHECK_EQ(got.array.Count(), n);
        CHECK_EQ(got.array.Data().size(), packed_len);
        packed.resize(packed_len);
        CHECK(packed == got.array.Data());
        for (int i = 0; i < n; i += 13) {
            int32_t v = -1;
            CHECK(!got.array.At(i, &v));
            CHECK_EQ(v, lv[i]);
        }
        for (uint32_t i = 0; i < tail_len; ++i) {  // the unflushed tail
            int32_t v = -1;
            CHECK(!got.array.At(n - (int)tail_len + (int)i, &v));
            CHECK_EQ(v, tail[i]);
        }
        // a stream that ends early: (nil, 0, err)
        Reader r2(s);
        HybridDecoder hd2(dev, bw, true);
        CHECK(!hd2.InitSize(r2));
        PackedLevels bad = decodePackedArray(LevelDecoder{&hd2, max}, n + 64);
        CHECK(bad.err.is(PQ_EOF));
        CHECK_EQ(bad.array.Count(), 0);
    }
    ConstDecoder c(0);  // flat columns: layout/chunk.go:111-124
    PackedLevels z = decodePackedArray(LevelDecoder{&c, 0}, 100);
    CHECK(!z.err);
    CHECK_EQ(z.notNull, 100);
    CHECK_EQ(z.array.Count(), 100);
}

// ---- data pages and readPageData (no reference test; against the oracle) --------------------------
struct BuiltPage {
    std::vector<uint8_t> block;
    DataPage page;
};

static BuiltPage make_page(int version, int n, const ColumnSpec &col, bool corrupt_values)
{
    BuiltPage b;
    std::vector<int32_t> rep(n), def(n);
    int not_null = 0;
    for (int i = 0; i < n; ++i) {
        rep[i] = (int32_t)(rnd() % 4 ? col.maxR : 0);
        def[i] = (int32_t)(rnd() % 5 ? col.maxD : rnd() % (col.maxD + 1u));
        not_null += def[i] == col.maxD;
    }
    std::vector<uint8_t> rs = hybrid_stream(rep, bitsLen16(col.maxR)), ds = hybrid_stream(def, bitsLen16(col.maxD));
    std::vector<uint8_t> vs;
    if (col.valueKind == PQ_KIND_DICT) {
        std::vector<int32_t> keys(not_null);
        for (auto &k : keys) k = (int32_t)(rnd() % col.dictLen);
        vs = {12};
        std::vector<uint8_t> body = hybrid_stream(keys, 12);
        vs.insert(vs.end(), body.begin(), body.end());
    } else {
        std::vector<int64_t> v(std::max(not_null, 1));
        int64_t x = (int64_t)(rnd() >> 20);
        for (auto &e : v) e = (x += (int64_t)(rnd() % 100000) - 3000);
        vs.resize(v.size() * 10 + 256);
        size_t len = 0;
        if (pqo_delta_encode64(v.data(), v.size(), 128, 4, vs.data(), vs.size(), &len) != 0) std::abort();
        vs.resize(len);
    }
    if (corrupt_values) vs.resize(vs.size() / 2);
    if (version == 1) {
        b.block = prefixed(rs);
        std::vector<uint8_t> pd = prefixed(ds);
        b.block.insert(b.block.end(), pd.begin(), pd.end());
    } else {
        b.block = rs;
        b.block.insert(b.block.end(), ds.begin(), ds.end());
        b.page.repLen = (int32_t)rs.size();
        b.page.defLen = (int32_t)ds.size();
    }
    b.block.insert(b.block.end(), vs.begin(), vs.end());
    b.page.version = version;
    b.page.numValues = n;
    return b;
}

static pqo_page_out oracle_page(const BuiltPage &b, const ColumnSpec &col, std::vector<int32_t> &rep,
                                std::vector<int32_t> &def, std::vector<uint8_t> &values)
{
    pqo_page_in in;
    std::memset(&in, 0, sizeof in);
    in.src = b.block.data();
    in.src_len = b.block.size();
    in.version = (uint32_t)b.page.version;
    in.num_values = (uint32_t)b.page.numValues;
    in.max_rep = col.maxR;
    in.max_def = col.maxD;
    in.rep_len = (uint32_t)b.page.repLen;
    in.def_len = (uint32_t)b.page.defLen;
    in.value_kind = col.valueKind == PQ_KIND_DICT ? PQO_VALUES_DICT : PQO_VALUES_DELTA64;
    in.elem_size = col.elemSize;
    in.dict = col.dict;
    in.dict_len = col.dictLen;
    rep.assign(in.num_values, 0);
    def.assign(in.num_values, 0);
    values.assign((size_t)in.num_values * col.elemSize + 8, 0);
    pqo_page_out out;
    pqo_page_read(&in, rep.data(), def.data(), values.data(), &out);
    return out;
}

static void TestDataPages(Device &dev)
{
    std::vector<int32_t> dict(4096);
    for (auto &x : dict) x = (int32_t)rnd();
    ColumnSpec dict_col;
    dict_col.maxR = 1;
    dict_col.maxD = 3;
    dict_col.valueKind = PQ_KIND_DICT;
    dict_col.elemSize = 4;
    dict_col.dict = dict.data();
    dict_col.dictLen = 4096;
    ColumnSpec delta_col;
    delta_col.maxR = 2;
    delta_col.maxD = 2;
    delta_col.valueKind = PQ_KIND_DELTA64;
    delta_col.elemSize = 8;

    for (const ColumnSpec &col : {dict_col, delta_col}) {
        for (int version : {1, 2}) {
            // single pages through the PageReader shape
            for (bool corrupt : {false, true}) {
                BuiltPage b = make_page(version, 5000 + 17 * version, col, corrupt);
                b.page.block = b.block.data();
                b.page.blockLen = b.block.size();
                DataPageReader pr(dev, col);
                CHECK(!pr.read(b.page));
                CHECK_EQ(pr.NumValues(), b.page.numValues);
                std::vector<uint8_t> values((size_t)b.page.numValues * col.elemSize);
                PageValues got = pr.ReadValues(values.data(), b.page.numValues);
                std::vector<int32_t> rep, def;
                std::vector<uint8_t> ref;
                pqo_page_out want = oracle_page(b, col, rep, def, ref);
                CHECK_EQ(got.err.status, want.status);
                CHECK_EQ(got.err.stage, want.stage);
                CHECK_EQ(got.err.detail, want.detail);
                CHECK_EQ(got.n, want.n);
                if (corrupt) CHECK(got.err.is(PQ_PAGE_VALUES_FAILED));
                if (!got.err) {
                    CHECK_EQ(got.notNull, want.not_null);
                    CHECK(std::memcmp(values.data(), ref.data(), (size_t)want.not_null * col.elemSize) == 0);
                    for (int i = 0; i < got.n; i += 7) {
                        int32_t r = -1, d = -1;
                        CHECK(!got.rLevel.At(i, &r));
                        CHECK(!got.dLevel.At(i, &d));
                        CHECK_EQ(r, rep[i]);
                        CHECK_EQ(d, def[i]);
                    }
                    PageValues again = pr.ReadValues(values.data(), 10);  // page exhausted: (0, nil, nil, nil)
                    CHECK(!again.err);
                    CHECK_EQ(again.n, 0);
                }
            }
            // a column chunk of pages in one call
            std::vector<BuiltPage> built;
            for (int p = 0; p < 12; ++p) built.push_back(make_page(version, 1000 + 333 * p, col, false));
            std::vector<DataPage> pages;
            for (auto &b : built) {
                b.page.block = b.block.data();
                b.page.blockLen = b.block.size();
                pages.push_back(b.page);
            }
            ChunkData chunk = readPageData(dev, col, pages);
            CHECK(!chunk.err);
            std::vector<uint8_t> all_values;
            std::vector<int32_t> all_rep, all_def;
            for (auto &b : built) {
                std::vector<int32_t> rep, def;
                std::vector<uint8_t> ref;
                pqo_page_out want = oracle_page(b, col, rep, def, ref);
                CHECK_EQ(want.status, 0);
                all_values.insert(all_values.end(), ref.begin(), ref.begin() + (size_t)want.not_null * col.elemSize);
                all_rep.insert(all_rep.end(), rep.begin(), rep.end());
                all_def.insert(all_def.end(), def.begin(), def.end());
            }
            CHECK(chunk.values == all_values);
            CHECK_EQ(chunk.numValues * col.elemSize, all_values.size());
            CHECK_EQ(chunk.rLevels.Count(), all_rep.size());
            CHECK_EQ(chunk.dLevels.Count(), all_def.size());
            for (size_t i = 0; i < all_rep.size(); i += 11) {
                int32_t r = -1, d = -1;
                CHECK(!chunk.rLevels.At((int)i, &r));
                CHECK(!chunk.dLevels.At((int)i, &d));
                CHECK_EQ(r, all_rep[i]);
                CHECK_EQ(d, all_def[i]);
            }
            // a failing page in the middle stops the chunk there
            BuiltPage bad = make_page(version, 2000, col, true);
            bad.page.block = bad.block.data();
            bad.page.blockLen = bad.block.size();
            pages.insert(pages.begin() + 5, bad.page);
            ChunkData broken = readPageData(dev, col, pages);
            CHECK(broken.err.is(PQ_PAGE_VALUES_FAILED));
            CHECK_EQ(broken.failedPage, 5);
            size_t slots5 = 0;
            for (int p = 0; p < 5; ++p) slots5 += (size_t)built[p].page.numValues;
            CHECK_EQ(broken.rLevels.Count(), slots5);
        }
    }
}

// ---- types.ByteArrayDeltaLengthDecoder / ByteArrayDeltaDecoder (no reference test; streams as the
//      reference's encoders lay them out, types/byte-array.go:148-158,264-275; checked against the
//      values fed in and against the oracle) -------------------------------------------------------
static std::vector<uint8_t> delta32_stream(const std::vector<int32_t> &v)
{
    std::vector<uint8_t> s(v.size() * 5 + 64);
    size_t len = 0;
    if (pqo_delta_encode32(v.data(), v.size(), 128, 4, s.data(), s.size(), &len) != 0) std::abort();
    s.resize(len);
    return s;
}

static void TestByteArrayDecoders(Device &dev)
{
    const uint32_t n = 5000;
    std::vector<ByteArray> vals(n);
    for (uint32_t i = 0; i < n; ++i) {
        if (i && rnd() % 3 == 0) {  // extend or cut the value in front
            vals[i] = vals[i - 1];
            vals[i].resize(rnd() % (vals[i].size() + 1));
        }
        for (uint32_t k = rnd() % 9; k > 0; --k) vals[i].push_back((uint8_t)rnd());
    }
    std::vector<int32_t> lens(n), plens(n), slens(n);
    std::vector<uint8_t> body, sbody;
    for (uint32_t i = 0; i < n; ++i) {
        lens[i] = (int32_t)vals[i].size();
        body.insert(body.end(), vals[i].begin(), vals[i].end());
        size_t p = 0;
        if (i)
            while (p < vals[i].size() && p < vals[i - 1].size() && vals[i][p] == vals[i - 1][p]) ++p;
        plens[i] = (int32_t)p;
        slens[i] = (int32_t)(vals[i].size() - p);
        sbody.insert(sbody.end(), vals[i].begin() + p, vals[i].end());
    }
    std::vector<uint8_t> ls = delta32_stream(lens);
    ls.insert(ls.end(), body.begin(), body.end());
    std::vector<uint8_t> fs = delta32_stream(plens), tail = delta32_stream(slens);
    fs.insert(fs.end(), tail.begin(), tail.end());
    fs.insert(fs.end(), sbody.begin(), sbody.end());
    {
        std::printf("    delta length: all values\n");
        Reader r(ls);
        ByteArrayDeltaLengthDecoder d(dev);
        CHECK(!d.Init(r));
        std::vector<ByteArray> got(n + 5);
        auto res = d.DecodeValues(got.data(), n + 5);  // io.EOF with fewer values is acceptable
        CHECK(res.second.is(PQ_EOF));
        CHECK_EQ(res.first, n);
        got.resize(n);
        CHECK(got == vals);
        CHECK_EQ(r.Len(), 0);
    }
    {
        std::printf("    delta: all values\n");
        Reader r(fs);
        ByteArrayDeltaDecoder d(dev);
        CHECK(!d.Init(r));
        std::vector<ByteArray> got(n);
        auto res = d.DecodeValues(got.data(), n);
        CHECK(!res.second);
        CHECK_EQ(res.first, n);
        CHECK(got == vals);
        CHECK_EQ(r.Len(), 0);
        // same call through the oracle
        std::vector<uint32_t> off(n + 1);
        std::vector<uint8_t> data(body.size() + 1);
        uint32_t cnt = 0;
        CHECK_EQ(pqo_delta_ba_decode(fs.data(), fs.size(), n, off.data(), data.data(), body.size(), &cnt, nullptr, nullptr,
                                     nullptr), 0);
        CHECK_EQ(cnt, n);
        data.resize(body.size());
        CHECK(data == body);
    }
    {  // a prefix longer than the value in front: "invalid prefix len", count 0 (:328-336)
        std::printf("    delta: invalid prefix\n");
        std::vector<int32_t> bad = plens;
        bad[100] = lens[99] + 1;
        std::vector<uint8_t> s = delta32_stream(bad);
        s.insert(s.end(), tail.begin(), tail.end());
        s.insert(s.end(), sbody.begin(), sbody.end());
        Reader r(s);
        ByteArrayDeltaDecoder d(dev);
        CHECK(!d.Init(r));
        std::vector<ByteArray> got(n);
        auto res = d.DecodeValues(got.data(), n);
        CHECK(res.second.is(PQ_BYTE_ARRAY_INVALID_PREFIX));
        CHECK_EQ(res.first, 0);
    }
    {  // unequal counts fail Init (:301-308); truncated lengths fail Init too
        std::printf("    delta: unequal counts\n");
        std::vector<int32_t> fewer(plens.begin(), plens.begin() + 4000);
        std::vector<uint8_t> s = delta32_stream(fewer);
        s.insert(s.end(), tail.begin(), tail.end());
        Reader r(s);
        ByteArrayDeltaDecoder d(dev);
        CHECK(d.Init(r).is(PQ_BYTE_ARRAY_COUNT_MISMATCH));
        std::vector<uint8_t> cut(ls.begin(), ls.begin() + 40);
        Reader r2(cut);
        ByteArrayDeltaLengthDecoder d2(dev);
        CHECK(d2.Init(r2).is(PQ_UNEXPECTED_EOF));
    }
}

// ---- ChunkReader.readPages (no reference test; headers written by hand in the compact protocol,
//      checked field by field and against the oracle) -------------------------------------------------
static void put_varint(std::vector<uint8_t> &o, uint64_t v)
{
    while (v >= 0x80) {
        o.push_back((uint8_t)(v | 0x80));
        v >>= 7;
    }
    o.push_back((uint8_t)v);
}
static void put_i32(std::vector<uint8_t> &o, int delta, int32_t v)  // field header (short form) + zig-zag value
{
    o.push_back((uint8_t)(delta << 4 | 5));
    put_varint(o, (uint32_t)((v << 1) ^ (v >> 31)));
}
// PageHeader{type, uncompressed, compressed, data_page_header{num_values, encoding, RLE, RLE}} or a dictionary header
static std::vector<uint8_t> page_header_bytes(int32_t type, int32_t size, int32_t num_values, int32_t encoding)
{
    std::vector<uint8_t> o;
    put_i32(o, 1, type);
    put_i32(o, 1, size);
    put_i32(o, 1, size);
    if (type == 0) {
        o.push_back(2 << 4 | 12);  // field 5: struct
        put_i32(o, 1, num_values);
        put_i32(o, 1, encoding);
        put_i32(o, 1, 3);
        put_i32(o, 1, 3);
        o.push_back(0);
    } else {
        o.push_back(4 << 4 | 12);  // field 7: struct
        put_i32(o, 1, num_values);
        put_i32(o, 1, encoding);
        o.push_back(0);
    }
    o.push_back(0);
    return o;
}

static void TestReadPages(Device &dev)
{
    // a file with two chunks: [dictionary, data, data] and [data], some bytes in front and between
    std::vector<uint8_t> file(37, 0xee);
    std::vector<ColumnChunkMeta> metas(2);
    auto add_page = [&](int32_t type, int32_t size, int32_t nv, int32_t enc) {
        std::vector<uint8_t> h = page_header_bytes(type, size, nv, enc);
        file.insert(file.end(), h.begin(), h.end());
        file.insert(file.end(), (size_t)size, (uint8_t)0x11);
    };
    metas[0].hasDictionaryPageOffset = true;
    metas[0].DictionaryPageOffset = (int64_t)file.size();
    add_page(2, 40, 10, 0);
    metas[0].DataPageOffset = (int64_t)file.size();
    add_page(0, 300, 200, 8);
    add_page(0, 123, 77, 2);
    metas[0].TotalCompressedSize = (int64_t)file.size() - metas[0].DictionaryPageOffset;
    file.insert(file.end(), 5, 0xee);
    metas[1].DataPageOffset = (int64_t)file.size();
    metas[1].Type = 2;
    add_page(0, 8000, 1000, 5);
    metas[1].TotalCompressedSize = (int64_t)file.size() - metas[1].DataPageOffset;

    std::vector<ChunkPages> got = readPages(dev, file.data(), file.size(), 0, metas);
    CHECK_EQ(got.size(), 2);
    CHECK(!got[0].err);
    CHECK(!got[1].err);
    CHECK_EQ(got[0].pages.size(), 3);
    CHECK_EQ(got[1].pages.size(), 1);
    CHECK_EQ(got[0].count, metas[0].TotalCompressedSize);
    CHECK_EQ(got[1].count, metas[1].TotalCompressedSize);
    CHECK_EQ(got[0].pages[0].type, 2);
    CHECK_EQ(got[0].pages[0].num_values, 10);
    CHECK_EQ(got[0].pages[1].num_values, 200);
    CHECK_EQ(got[0].pages[1].encoding, 8);
    CHECK_EQ(got[0].pages[2].compressed_size, 123);
    CHECK_EQ(got[1].pages[0].uncompressed_size, 8000);
    CHECK_EQ(got[1].pages[0].def_encoding, 3);
    // the same chunks through the oracle
    for (size_t c = 0; c < 2; ++c) {
        pqo_chunk_in in;
        std::memset(&in, 0, sizeof in);
        const int64_t start = metas[c].hasDictionaryPageOffset ? metas[c].DictionaryPageOffset : metas[c].DataPageOffset;
        in.src = file.data() + start;
        in.src_len = file.size() - (size_t)start;
        in.total_compressed_size = metas[c].TotalCompressedSize;
        in.data_page_offset = metas[c].DataPageOffset;
        in.dictionary_page_offset = metas[c].DictionaryPageOffset;
        in.has_dictionary_page_offset = metas[c].hasDictionaryPageOffset;
        in.physical_type = metas[c].Type;
        in.type_length = -1;
        std::vector<pqo_page_info> want(16);
        uint32_t n = 0;
        int64_t count = 0;
        CHECK_EQ(pqo_walk_chunk(&in, want.data(), 16, &n, &count), 0);
        CHECK_EQ(n, got[c].pages.size());
        CHECK_EQ(count, got[c].count);
        static_assert(sizeof(pqo_page_info) == sizeof(pq_page_info), "same record layout");
        CHECK(std::memcmp(want.data(), got[c].pages.data(), n * sizeof(pq_page_info)) == 0);
    }
    // a second dictionary page, and a chunk cut in the middle of a block
    std::vector<ColumnChunkMeta> bad(2);
    std::vector<uint8_t> f2;
    auto add2 = [&](int32_t type, int32_t size, int32_t nv, int32_t enc, size_t body) {
        std::vector<uint8_t> h = page_header_bytes(type, size, nv, enc);
        f2.insert(f2.end(), h.begin(), h.end());
        f2.insert(f2.end(), body, (uint8_t)0);
    };
    bad[0].DataPageOffset = 0;
    add2(2, 16, 4, 0, 16);
    add2(2, 16, 4, 0, 16);
    bad[0].TotalCompressedSize = (int64_t)f2.size();
    bad[1].DataPageOffset = (int64_t)f2.size();
    add2(0, 100, 5, 0, 60);
    bad[1].TotalCompressedSize = 200;
    std::vector<ChunkPages> r2 = readPages(dev, f2.data(), f2.size(), 0, bad);
    CHECK(r2[0].err.is(PQ_WALK_SECOND_DICTIONARY));
    CHECK_EQ(r2[0].pages.size(), 1);
    CHECK(r2[1].err.is(PQ_WALK_BLOCK_SIZE));
    CHECK_EQ(r2[1].pages.size(), 0);
}

int main()
{
    Device dev(0);
    struct {
        const char *name;
        std::function<void()> fn;
    } tests[] = {
        {"TestHybridDecoder", [&] { TestHybridDecoder(dev); }},
        {"TestHybridDecoder_EncoderStreams", [&] { TestHybridDecoder_EncoderStreams(dev); }},
        {"TestDeltaBinaryPackDecoder32", [&] { TestDeltaBinaryPackDecoder<int32_t>(dev); }},
        {"TestDeltaBinaryPackDecoder64", [&] { TestDeltaBinaryPackDecoder<int64_t>(dev); }},
        {"TestPackedArray", [&] { TestPackedArray(); }},
        {"TestPlainDecoders", [&] { TestPlainDecoders(dev); }},
        {"TestLevelPostProcessing", [&] { TestLevelPostProcessing(dev); }},
        {"TestDictDecoder_int32", [&] { TestDictDecoder<int32_t>(dev); }},
        {"TestDictDecoder_int64", [&] { TestDictDecoder<int64_t>(dev); }},
        {"TestDecodePackedArray", [&] { TestDecodePackedArray(dev); }},
        {"TestDataPages", [&] { TestDataPages(dev); }},
        {"TestByteArrayDecoders", [&] { TestByteArrayDecoders(dev); }},
        {"TestReadPages", [&] { TestReadPages(dev); }},
    };
    for (auto &t : tests) {
        int before = g_failed;
        std::printf("=== RUN   %s\n", t.name);
        std::fflush(stdout);
        try {
            t.fn();
        } catch (const std::exception &e) {  // the mirror throws only for library-level failures
            std::printf("    exception: %s\n", e.what());
            ++g_failed;
        }
        std::printf("--- %s: %s\n", g_failed == before ? "PASS" : "FAIL", t.name);
    }
    std::printf("%d checks, %d failed\n", g_checks, g_failed);
    return g_failed ? 1 : 0;
}
