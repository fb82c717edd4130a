"""GPU parity: the HIP path (through the C ABI of libpqdecode.so) against the CPU
oracle on identical inputs — bit-exact values, identical status / counts — and
against the reference's golden vectors.  Needs a real MI355X: `pytest -m gpu`."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O
from gpu_harness import Case, guard_ok, run_cases
from parquet_go_amd import capi, synth

pytestmark = pytest.mark.gpu

K = capi


@pytest.fixture(scope="module", params=["wave_per_page", "workgroup_per_page"])
def ctx(request):
    """Every test runs under both kernel selections for hybrid / dictionary / level streams and
    data pages: one wave per descriptor (forced from the first page on) and one workgroup per
    descriptor.  The delta tests switch their own option as well."""
    c = capi.Context(0)
    c.set_option(K.OPT_HYBRID_WAVE_MIN_PAGES, 0 if request.param == "wave_per_page" else 1 << 30)
    yield c
    c.close()


def _load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


def _rng(seed):
    return np.random.default_rng(seed)


# ---------------------------------------------------------------------------
# a1 / a2: fixed-width bit-unpack

def test_bitunpack_golden_vectors(ctx, golden_dir):
    for bits, name in ((32, "bitpack32.json"), (64, "bitpack64.json")):
        for v in _load(golden_dir, name)["vectors"]:
            got = ctx.bitunpack(np.array(v["data"], dtype=np.uint8), v["w"], 1, bits)
            assert got.tolist() == v["values"], (bits, v)


@pytest.mark.parametrize("bits", [32, 64])
def test_bitunpack_all_widths_vs_oracle(ctx, bits):
    rng = _rng(bits)
    for w in range(bits + 1):
        for n_groups in (1, 7, 511, 512, 513, 2049):
            src = rng.integers(0, 256, size=n_groups * w + 3, dtype=np.uint8)
            got = ctx.bitunpack(src, w, n_groups, bits)
            want = (O.bitunpack32 if bits == 32 else O.bitunpack64)(src, w, n_groups)
            assert np.array_equal(got, want), (bits, w, n_groups)


def test_bitunpack_device_resident_unaligned(ctx):
    import torch
    rng = _rng(3)
    dev = torch.device("cuda:0")
    for bits, w, n_groups, soff, doff in ((32, 12, 5000, 5, 1), (32, 32, 1024, 1, 0), (64, 33, 3000, 7, 1),
                                          (64, 64, 700, 3, 0), (32, 1, 100000, 9, 2), (32, 0, 100, 0, 1)):
        src = rng.integers(0, 256, size=n_groups * w + soff + 4, dtype=np.uint8)
        t_src = torch.from_numpy(src).to(dev)
        es = bits // 8
        t_dst = torch.full(((8 * n_groups + doff + 2) * es,), 0xA5, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        ctx.bitunpack_dev(t_src.data_ptr() + soff, n_groups * w, w, n_groups, t_dst.data_ptr() + doff * es, bits)
        ctx.synchronize()
        out = t_dst.cpu().numpy().view(np.int32 if bits == 32 else np.int64)
        want = (O.bitunpack32 if bits == 32 else O.bitunpack64)(src[soff:], w, n_groups)
        assert np.array_equal(out[doff: doff + 8 * n_groups], want), (bits, w)
        assert guard_ok(out[:doff]) and guard_ok(out[doff + 8 * n_groups:])


# ---------------------------------------------------------------------------
# a3 / a4: hybrid

def _check_hybrid(c, w, prefixed=False):
    st, want, cons = O.hybrid_decode(c.src[: c.result_src_len], w, c.n, length_prefixed=prefixed)
    r = c.result
    assert r["status"] == st, (w, c.n, r, st)
    assert r["n_decoded"] == want.size
    assert np.array_equal(c.values[: want.size], want)
    assert guard_ok(c.guard)
    if st == O.OK:
        assert r["consumed"] == cons


def _runny(rng, n, w, mean_run):
    if mean_run == 1:
        return rng.integers(0, 1 << w, size=n) if w else np.zeros(n, np.int64)
    out = np.empty(n, dtype=np.int64)
    i = 0
    while i < n:
        ln = int(rng.geometric(1.0 / mean_run))
        out[i:i + ln] = rng.integers(0, 1 << w) if w else 0
        i += ln
    return out


def test_hybrid_golden_vectors(ctx, golden_dir):
    cases = []
    for v in _load(golden_dir, "hybrid.json")["vectors"]:
        stream = v["stream"] if v["stream"] is not None else O.hybrid_encode(v["values"], v["w"]).tolist()
        n = len(v.get("padded_values", v["values"]))
        c = Case(K.KIND_HYBRID, np.array(stream, dtype=np.uint8), n, bit_width=v["w"])
        c.want = v.get("padded_values", v["values"])
        cases.append(c)
    # tail of the group-boundary vector: 1,2,3 then zero fill, then EOF on the 9th pull
    cases.append(Case(K.KIND_HYBRID, np.array([0x03, 0x39], dtype=np.uint8), 9, bit_width=2))
    run_cases(ctx, cases)
    for c in cases[:-1]:
        assert c.result["status"] == K.OK
        assert c.values.tolist() == c.want
        assert guard_ok(c.guard)
    t = cases[-1]
    assert t.result["status"] == K.EOF and t.result["n_decoded"] == 8
    assert t.values[:8].tolist() == [1, 2, 3, 0, 0, 0, 0, 0]


def test_hybrid_random_streams_vs_oracle(ctx):
    rng = _rng(7)
    cases = []
    for w, mean_run, n in [(1, 3, 5000), (1, 40, 30000), (2, 6, 20000), (3, 1, 1000), (3, 12, 4003),
                           (8, 2, 100000), (9, 32, 20000), (12, 1, 1000000), (12, 40, 7777),
                           (16, 1, 70000), (17, 9, 3001), (24, 1, 20000), (31, 5, 5000),
                           (32, 20, 2048), (32, 1, 20000), (0, 1, 100), (5, 1, 9), (5, 100, 1),
                           (4, 8, 8), (4, 8, 7), (4, 5000, 300000), (7, 1, 0)]:
        vals = _runny(rng, n, w, mean_run)
        stream = synth.hybrid_encode(vals, w)
        for extra in (0, 5):       # pull exactly n, and more than the stream holds
            c = Case(K.KIND_HYBRID, stream, n + extra, bit_width=w)
            c.w = w
            c.result_src_len = stream.size
            cases.append(c)
        if stream.size > 3:        # truncated streams -> EOF / unexpected EOF / zero fill
            for cut in sorted({1, 2, stream.size // 3, stream.size - 1}):
                c = Case(K.KIND_HYBRID, stream[:cut].copy(), n, bit_width=w)
                c.w = w
                c.result_src_len = cut
                cases.append(c)
    run_cases(ctx, cases, seed=1)
    for c in cases:
        _check_hybrid(c, c.w)


def test_hybrid_length_prefixed_and_errors(ctx):
    rng = _rng(8)
    cases = []
    body = synth.hybrid_encode(_runny(rng, 5000, 5, 6), 5)
    # InitSize: prefix shorter than, equal to and longer than what follows; short prefix
    for size in (0, 1, body.size // 2, body.size, body.size + 100):
        s = np.concatenate([np.frombuffer(np.uint32(size).tobytes(), np.uint8), body, body[:40]])
        c = Case(K.KIND_HYBRID, s, 5000, bit_width=5, flags=K.FLAG_LENGTH_PREFIXED)
        c.w, c.prefixed, c.result_src_len = 5, True, s.size
        cases.append(c)
    for k in (0, 1, 3):
        c = Case(K.KIND_HYBRID, np.array([9, 0, 0, 0][:k], dtype=np.uint8), 10, bit_width=3,
                 flags=K.FLAG_LENGTH_PREFIXED)
        c.w, c.prefixed, c.result_src_len = 3, True, k
        cases.append(c)
    # w == 0 with a prefix flag reads nothing at all
    c = Case(K.KIND_HYBRID, np.array([1, 2, 3], dtype=np.uint8), 50, bit_width=0, flags=K.FLAG_LENGTH_PREFIXED)
    c.w, c.prefixed, c.result_src_len = 0, True, 3
    cases.append(c)
    # malformed streams
    bad = [
        ([0x01], 3),                          # bit-packed run of 0 groups
        ([0x00, 0x05], 3),                    # RLE run of 0 values
        ([0x10, 0x08], 3),                    # RLE value 8 needs 4 bits > w = 3
        ([0x10], 3),                          # RLE value missing -> EOF
        ([0x10, 0x01], 9),                    # 2-byte RLE value, 1 present -> unexpected EOF
        ([0x80] * 10 + [0x01], 4),            # varint longer than 10 bytes
        ([0xff] * 9 + [0x02], 4),             # 10th byte > 1
        ([0xff, 0xff, 0xff, 0xff, 0x0f], 4),  # 2^32-1 does not fit int32
        ([0xff, 0xff, 0xff, 0xff, 0x07, 1, 2], 4),  # 2^31-1: huge bit-packed run, data ends
        ([0x80], 4),                          # EOF inside the header varint
        ([0x03, 0x39, 0x04, 0x02], 2),        # group zero-filled... but here 2 bytes: full group then RLE
    ]
    for stream, w in bad:
        c = Case(K.KIND_HYBRID, np.array(stream, dtype=np.uint8), 40, bit_width=w)
        c.w, c.prefixed, c.result_src_len = w, False, len(stream)
        cases.append(c)
    run_cases(ctx, cases, seed=2)
    for c in cases:
        _check_hybrid(c, c.w, prefixed=c.prefixed)


def test_hybrid_fuzz_random_bytes(ctx):
    """Arbitrary bytes: the GPU path and the oracle must agree on status, count and values."""
    rng = _rng(9)
    cases = []
    for i in range(300):
        w = int(rng.integers(1, 33))
        ln = int(rng.integers(1, 400))
        s = rng.integers(0, 256, size=ln, dtype=np.uint8)
        # keep run headers small so streams are not all "count too large"
        s[rng.integers(0, ln, size=ln // 3)] &= 0x3F
        c = Case(K.KIND_HYBRID, s, int(rng.integers(1, 5000)), bit_width=w)
        c.w, c.result_src_len = w, ln
        cases.append(c)
    run_cases(ctx, cases, seed=3)
    for c in cases:
        _check_hybrid(c, c.w)


def _random_run_stream(rng, w, target_bytes):
    """A well-formed stream built run by run (not through the encoder): RLE runs of 1..700
    repeats and bit-packed runs of 1..70 groups, i.e. one- and two-byte headers, in random
    order, long enough to cross several staging windows."""
    out = bytearray()
    vb = (w + 7) // 8
    while len(out) < target_bytes:
        kind = rng.integers(0, 24) % 6 if len(out) > 200 else rng.integers(0, 4)
        if kind >= 4:
            # a header in more bytes than it needs (binary.ReadUvarint takes it), or a
            # three-byte one (an RLE run of 8192 repeats and more)
            bp = bool(rng.integers(0, 2)) and kind == 4
            cnt = int(rng.integers(1, 40)) if kind == 4 else int(rng.integers(8192, 20000))
            h = (cnt << 1) | (1 if bp else 0)
            out += bytes([0x80 | (h & 0x7F), (h >> 7) & 0x7F] if kind == 4 else synth.uleb128(h))
            if bp:
                out += rng.integers(0, 256, size=cnt * w, dtype=np.uint8).tobytes()
            else:
                out += int(rng.integers(0, 1 << w)).to_bytes(8, "little")[:vb]
            continue
        if kind == 0:                                   # short RLE
            cnt, bp = int(rng.integers(1, 20)), False
        elif kind == 1:                                 # long RLE (two-byte header from 64 on)
            cnt, bp = int(rng.integers(20, 700)), False
        elif kind == 2:                                 # short bit-packed
            cnt, bp = int(rng.integers(1, 6)), True
        else:                                           # long bit-packed (two-byte header from 64 on)
            cnt, bp = int(rng.integers(6, 71)), True
        out += bytes(synth.uleb128((cnt << 1) | (1 if bp else 0)))
        if bp:
            out += rng.integers(0, 256, size=cnt * w, dtype=np.uint8).tobytes()
        else:
            v = int(rng.integers(0, 1 << w)) if w < 63 else 0
            out += v.to_bytes(8, "little")[:vb]
    return np.frombuffer(bytes(out), dtype=np.uint8).copy()


def test_hybrid_structured_runs_stress(ctx):
    """Run lists of every shape, cut at random bytes and pulled to random counts: windows,
    tiles, the span parse and the checked header path all meet their boundaries here."""
    rng = _rng(21)
    cases = []
    for i in range(260):
        w = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 12, 15, 16, 17, 20, 24, 25, 31, 32]))
        s = _random_run_stream(rng, w, int(rng.integers(10, 16000)))
        if i % 3 == 0:
            s = s[: int(rng.integers(1, s.size + 1))].copy()      # cut anywhere
        total_cap = 60000
        n = int(rng.integers(1, total_cap))
        c = Case(K.KIND_HYBRID, s, n, bit_width=w)
        c.w, c.result_src_len = w, s.size
        cases.append(c)
    run_cases(ctx, cases, seed=5)
    for c in cases:
        _check_hybrid(c, c.w)


def test_hybrid_pull_ends_inside_a_group(ctx):
    """The pull stops at every position of a bit-packed group, behind RLE runs of every length modulo
    8: run-dense tiles cut a run in the middle of a group, and the bytes reported as consumed are those
    of the whole group (readBitPackedRun reads it whole, encoding/hybrid-decoder.go:138-149)."""
    rng = _rng(33)
    cases = []
    for w in (1, 2, 3, 7, 8, 12, 17, 32):
        vb = (w + 7) // 8
        for lead in range(1, 10):
            out = bytearray()
            total = 0
            # enough short runs to fill several tiles of 512 values and cross a 3 KiB window at w >= 12
            for r in range(160):
                cnt = lead + (r % 5)
                out += bytes(synth.uleb128(cnt << 1)) + int(rng.integers(0, 1 << min(w, 31))).to_bytes(8, "little")[:vb]
                total += cnt
                g = 1 + (r % 4)
                out += bytes(synth.uleb128((g << 1) | 1)) + rng.integers(0, 256, size=g * w, dtype=np.uint8).tobytes()
                total += 8 * g
            s = np.frombuffer(bytes(out), dtype=np.uint8).copy()
            for cut in range(0, 9):
                c = Case(K.KIND_HYBRID, s, total - cut, bit_width=w)
                c.w, c.result_src_len = w, s.size
                cases.append(c)
    run_cases(ctx, cases, seed=8)
    for c in cases:
        _check_hybrid(c, c.w)
        assert c.result["status"] == K.OK


# ---------------------------------------------------------------------------
# a5: dictionary pages

def _check_dict(c):
    st, want, nd, cons = O.dict_decode(c.src, c.dictionary, c.n)
    r = c.result
    assert r["status"] == st, (r, st)
    assert r["n_decoded"] == nd
    assert np.array_equal(c.values[:nd].view(np.uint8), want[:nd].view(np.uint8))
    assert guard_ok(c.guard)


def test_dict_pages_vs_oracle(ctx):
    rng = _rng(10)
    cases = []
    for dt in (np.int32, np.int64, np.float32, np.float64):
        for dlen, n, mean_run in ((4096, 100000, 1), (4096, 20000, 32), (7, 5000, 3), (1, 100, 1),
                                  (70000, 30000, 1), (300, 1, 1)):
            if np.issubdtype(dt, np.integer):
                d = rng.integers(np.iinfo(dt).min, np.iinfo(dt).max, size=dlen, dtype=dt)
            else:
                d = rng.standard_normal(dlen).astype(dt)
            w = int(dlen).bit_length()
            keys = np.minimum(_runny(rng, n, w, mean_run), dlen - 1)
            page = synth.dict_page(keys, w=w)
            cases.append(Case(K.KIND_DICT, page, n, dictionary=d))
            cases.append(Case(K.KIND_DICT, page, n + 3, dictionary=d))            # runs out -> EOF
            cases.append(Case(K.KIND_DICT, page[: max(1, page.size // 2)].copy(), n, dictionary=d))
    d = np.arange(10, dtype=np.int32) * 3
    keys = np.array([1, 2, 3, 9, 10, 1, 1, 1], dtype=np.int64)                     # 10 is out of range
    cases.append(Case(K.KIND_DICT, synth.dict_page(keys, w=4), 8, dictionary=d))
    cases.append(Case(K.KIND_DICT, synth.dict_page(keys, w=4), 4, dictionary=d))  # stops before it
    cases.append(Case(K.KIND_DICT, np.array([0], dtype=np.uint8), 20, dictionary=d))   # w = 0 -> dict[0]
    cases.append(Case(K.KIND_DICT, np.array([33, 1, 2], dtype=np.uint8), 5, dictionary=d))  # w > 32
    cases.append(Case(K.KIND_DICT, np.zeros(0, dtype=np.uint8), 5, dictionary=d))      # no width byte
    neg = np.concatenate([[32], [0x02], np.frombuffer(np.uint32(0x80000001).tobytes(), np.uint8)])
    cases.append(Case(K.KIND_DICT, neg.astype(np.uint8), 1, dictionary=d))              # negative key
    run_cases(ctx, cases, seed=4)
    for c in cases:
        _check_dict(c)


# ---------------------------------------------------------------------------
# a10 / a11: levels

def test_levels_vs_oracle(ctx):
    rng = _rng(11)
    cases = []
    for max_level, n, mean_run in ((1, 20000, 4), (3, 20000, 5), (3, 100000, 1), (7, 999, 2), (255, 5000, 3),
                                   (65535, 3000, 2), (0, 500, 1), (2, 0, 1),
                                   # long runs: whole tiles of one RLE run are counted without looking at their values
                                   (3, 50000, 3000), (1, 30000, 300), (2, 70000, 20000)):
        bw = int(max_level).bit_length()
        vals = np.minimum(_runny(rng, n, bw, mean_run), max_level)
        stream = synth.hybrid_encode(vals, bw)
        for mode in (0, 1, 2):
            src = synth.length_prefixed(stream) if mode == 1 else stream
            flags = {0: K.FLAG_LEVELS_CONST, 1: K.FLAG_LENGTH_PREFIXED, 2: 0}[mode]
            if mode == 2 and src.size == 0:
                flags = K.FLAG_NO_INIT
            c = Case(K.KIND_LEVELS, src, n, max_def=max_level, flags=flags)
            c.mode = mode
            cases.append(c)
        if n:   # not enough levels in the stream
            c = Case(K.KIND_LEVELS, stream, n + 9, max_def=max_level)
            c.mode = 2
            cases.append(c)
    run_cases(ctx, cases, seed=5)
    for c in cases:
        want = O.levels_decode(c.src, c.max_def, c.n, c.mode)
        r = c.result
        assert r["status"] == want["status"], (c.max_def, c.n, c.mode, r)
        if want["status"] == O.OK:
            assert r["not_null"] == want["not_null"]
            assert np.array_equal(c.def_levels, want["levels"])
            assert r["consumed"] == want["consumed"]
        else:
            assert r["n_decoded"] == 0 and r["not_null"] == 0
        assert guard_ok(c.def_guard)


# ---------------------------------------------------------------------------
# a6 / a7: DELTA_BINARY_PACKED

def _check_delta(c, bits):
    want = (O.delta_decode32 if bits == 32 else O.delta_decode64)(c.src[: c.result_src_len], c.n)
    r = c.result
    assert r["status"] == want["status"], (bits, c.n, c.result_src_len, r, want["status"], want["n_decoded"])
    assert r["n_decoded"] == want["n_decoded"], (bits, c.n, r, want["n_decoded"])
    assert np.array_equal(c.values[: want["n_decoded"]], want["values"])
    assert guard_ok(c.guard)
    if want["status"] == O.OK:
        assert r["consumed"] == want["consumed"], (bits, c.n, c.result_src_len)


def _run_delta_both_ways(ctx, cases, bits, seed):
    """Same cases through the one-wave-per-page kernel (which passes other layouts on)
    and through the workgroup-per-page kernel alone."""
    try:
        for min_pages in (0, 1 << 30):
            ctx.set_option(K.OPT_DELTA_WAVE_MIN_PAGES, min_pages)
            run_cases(ctx, cases, seed=seed)
            for c in cases:
                _check_delta(c, bits)
    finally:
        ctx.set_option(K.OPT_DELTA_WAVE_MIN_PAGES, 1)


def test_delta_golden_vectors(ctx, golden_dir):
    d = _load(golden_dir, "delta.json")
    for bits, kind in ((32, K.KIND_DELTA32), (64, K.KIND_DELTA64)):
        cases = []
        for v in d["decode"]:
            c = Case(kind, np.array(v["stream"], dtype=np.uint8), max(len(v.get("values", [])), 1))
            c.result_src_len = len(v["stream"])
            c.want = v.get("values", [])
            cases.append(c)
        for s in d["invalid_block_size"] + d["invalid_miniblock_count"]:
            c = Case(kind, np.array(s, dtype=np.uint8), 1)
            c.result_src_len = len(s)
            c.want = None
            cases.append(c)
        run_cases(ctx, cases, seed=bits)
        assert cases[0].values.tolist() == [7, 5, 3, 1, 2, 3, 4, 5] and cases[0].result["status"] == K.OK
        assert cases[1].result["status"] == K.EOF and cases[1].result["n_decoded"] == 0
        assert cases[2].result["status"] == K.EOF
        for c in cases[3:6]:
            assert c.result["status"] == K.INVALID_BLOCK_SIZE
        for c in cases[6:9]:
            assert c.result["status"] == K.INVALID_MINIBLOCK_COUNT
        for c in cases:
            _check_delta(c, bits)


@pytest.mark.parametrize("bits", [32, 64])
def test_delta_random_vs_oracle(ctx, bits):
    rng = _rng(20 + bits)
    kind = K.KIND_DELTA32 if bits == 32 else K.KIND_DELTA64
    dt = np.int32 if bits == 32 else np.int64
    cases = []
    sizes = [1, 2, 8, 9, 10, 127, 128, 129, 130, 136, 137, 255, 256, 257, 1000, 2048, 2049, 20000, 100001]
    for n in sizes:
        for wmax in (0, 3, 16, 31):
            steps = rng.integers(0, 1 << wmax, size=n) if wmax else np.zeros(n, np.int64)
            base = int(rng.integers(-1000, 1000))
            with np.errstate(over="ignore"):
                vals = (np.cumsum(steps + base) + int(rng.integers(-2**40, 2**40))).astype(dt)
            for block, mb in ((128, 4), (256, 4), (256, 8), (512, 4), (1024, 4), (128, 1)):
                if n > 3000 and (block, mb) not in ((128, 4), (256, 4)):
                    continue
                stream = synth.delta_encode(vals, bits, block, mb)
                for pull in sorted({n, max(n - 1, 0), n + 2, n // 2}):
                    c = Case(kind, stream, pull)
                    c.result_src_len = stream.size
                    cases.append(c)
    _run_delta_both_ways(ctx, cases, bits, seed=6)


@pytest.mark.parametrize("bits", [32, 64])
def test_delta_truncated_at_every_byte(ctx, bits):
    """Error classes and counts when the stream ends anywhere (the group fetched at position N - 1 and
    io.ReadFull of a cut group: encoding/delta-binary-packed-decoder.go:103,112)."""
    rng = _rng(30 + bits)
    kind = K.KIND_DELTA32 if bits == 32 else K.KIND_DELTA64
    dt = np.int32 if bits == 32 else np.int64
    cases = []
    for block, count in ((128, 300), (256, 300), (256, 600), (256, 385)):   # blocks of 128 and of 256 values
        vals = np.cumsum(rng.integers(-50, 1 << 11, size=count)).astype(dt)
        stream = synth.delta_encode(vals, bits, block, 4)
        for cut in range(stream.size + 1):
            c = Case(kind, stream[:cut].copy(), count)
            c.result_src_len = cut
            cases.append(c)
    _run_delta_both_ways(ctx, cases, bits, seed=7)


@pytest.mark.parametrize("bits", [32, 64])
def test_delta_fuzz_headers_and_bodies(ctx, bits):
    """Random bodies behind plausible headers, including layouts only the serial
    path handles (miniblocks that are not whole groups or do not tile 128)."""
    rng = _rng(40 + bits)
    kind = K.KIND_DELTA32 if bits == 32 else K.KIND_DELTA64
    layouts = [(128, 4), (256, 4), (256, 4), (128, 2), (128, 8), (128, 16), (256, 1), (256, 2), (384, 3), (384, 16), (128, 128),
               (128, 64), (128, 32), (640, 5), (1024, 2), (2048, 1), (128, 3), (0, 4), (100, 4), (256, 256)]
    cases = []
    for i in range(400):
        block, mb = layouts[i % len(layouts)]
        total = int(rng.integers(0, 700))
        first = int(rng.integers(-2**31, 2**31))
        head = synth.uleb128(block) + synth.uleb128(mb) + synth.uleb128(total) + synth.uleb128(synth.zigzag64(first))
        body = rng.integers(0, 256, size=int(rng.integers(0, 1500)), dtype=np.uint8)
        # widths mostly legal so that decoding gets somewhere
        body[rng.integers(0, max(body.size, 1), size=body.size // 2)] &= 0x1F
        s = np.concatenate([np.array(head, dtype=np.uint8), body])
        c = Case(kind, s, int(rng.integers(0, 800)))
        c.result_src_len = s.size
        cases.append(c)
    _run_delta_both_ways(ctx, cases, bits, seed=8)


# ---------------------------------------------------------------------------
# a12: whole data pages

def _make_page(rng, version, n, max_rep, max_def, value_kind, dictionary=None, p_max=0.8, plain_dtype=None):
    rep = rng.integers(0, max_rep + 1, size=n) if max_rep else np.zeros(n, np.int64)
    deff = np.where(rng.random(n) < p_max, max_def, rng.integers(0, max_def + 1, size=n)) if max_def else np.zeros(n, np.int64)
    not_null = int((deff == max_def).sum()) if max_rep else (n if max_def == 0 else 0)
    rs = synth.hybrid_encode(rep, int(max_rep).bit_length())
    ds = synth.hybrid_encode(deff, int(max_def).bit_length())
    if value_kind == K.KIND_DICT:
        keys = rng.integers(0, dictionary.size, size=not_null)
        vs = synth.dict_page(keys, dict_len=dictionary.size)
    elif value_kind == K.KIND_PLAIN:
        vs = rng.integers(0, 256, size=not_null * np.dtype(plain_dtype).itemsize, dtype=np.uint8)
    elif value_kind == K.KIND_BOOL_PLAIN:
        vs = rng.integers(0, 256, size=(not_null + 7) // 8, dtype=np.uint8)
    elif value_kind == K.KIND_BOOL_RLE:
        vs = synth.length_prefixed(synth.hybrid_encode(_runny(rng, not_null, 1, 40), 1))
    else:
        bits = 32 if value_kind == K.KIND_DELTA32 else 64
        v = np.cumsum(rng.integers(-9, 5000, size=not_null)).astype(np.int32 if bits == 32 else np.int64)
        vs = synth.delta_encode(v, bits)
    if version == 1:
        parts = []
        if max_rep:
            if int(max_rep).bit_length():
                parts.append(synth.length_prefixed(rs))
            if int(max_def).bit_length():
                parts.append(synth.length_prefixed(ds))
        parts.append(vs)
        return np.concatenate(parts), 0, 0
    return np.concatenate([rs, ds, vs]), rs.size, ds.size


def _check_page(c):
    ok = {K.KIND_DICT: O.VALUES_DICT, K.KIND_DELTA32: O.VALUES_DELTA32, K.KIND_DELTA64: O.VALUES_DELTA64,
          K.KIND_PLAIN: O.VALUES_PLAIN, K.KIND_BOOL_PLAIN: O.VALUES_BOOL_PLAIN,
          K.KIND_BOOL_RLE: O.VALUES_BOOL_RLE}
    want = O.page_read(c.src, 1 if c.kind == K.KIND_PAGE_V1 else 2, c.n, c.max_rep, c.max_def,
                       ok[c.value_kind], dictionary=c.dictionary, rep_len=c.rep_len, def_len=c.def_len,
                       value_dtype=c.plain_dtype)
    r = c.result
    assert r["status"] == want["status"], (r, want)
    assert r["stage"] == want["stage"], (r, want)
    assert r["detail"] == want["detail"]
    assert r["n_levels"] == want["n"]
    if want["status"] == O.OK and c.n:
        assert r["not_null"] == want["not_null"]
        assert np.array_equal(c.rep_levels, want["rep"])
        assert np.array_equal(c.def_levels, want["def_"])
        nn = want["not_null"]
        assert np.array_equal(c.values[:nn].view(np.uint8), want["values"][:nn].view(np.uint8))
    assert guard_ok(c.guard) and guard_ok(c.def_guard) and guard_ok(c.rep_guard)


def test_data_pages_vs_oracle(ctx):
    rng = _rng(50)
    d32 = rng.integers(-2**31, 2**31, size=4096, dtype=np.int64).astype(np.int32)
    d64 = rng.integers(-2**62, 2**62, size=500, dtype=np.int64)
    cases = []
    for version in (1, 2):
        kind = K.KIND_PAGE_V1 if version == 1 else K.KIND_PAGE_V2
        for n in (0, 1, 9, 1000, 20000):
            for max_rep, max_def in ((1, 3), (0, 0), (0, 1), (2, 2), (1, 1), (3, 7)):
                for vk, dic in ((K.KIND_DICT, d32), (K.KIND_DICT, d64), (K.KIND_DELTA32, None), (K.KIND_DELTA64, None)):
                    src, rl, dl = _make_page(rng, version, n, max_rep, max_def, vk, dic)
                    cases.append(Case(kind, src, n, max_rep=max_rep, max_def=max_def, value_kind=vk,
                                      dictionary=dic, rep_len=rl, def_len=dl))
                    if n == 1000:   # damaged pages: every stage's error path
                        for cut in (0, 3, src.size // 4, src.size // 2, src.size - 1):
                            cases.append(Case(kind, src[:cut].copy(), n, max_rep=max_rep, max_def=max_def,
                                              value_kind=vk, dictionary=dic, rep_len=rl, def_len=dl))
                        if version == 2:   # zero-length level sections leave decoders uninitialised
                            cases.append(Case(kind, src[rl:].copy(), n, max_rep=max_rep, max_def=max_def,
                                              value_kind=vk, dictionary=dic, rep_len=0, def_len=dl))
    run_cases(ctx, cases, seed=9)
    for c in cases:
        _check_page(c)


def test_flat_pages_without_level_output(ctx):
    """PQ_FLAG_NO_LEVEL_OUTPUT: pages of a column with max_rep == 0 (both level decoders are ConstDecoder(0),
    layout/chunk.go:111-124) decode to the same result without writing their all-zero levels."""
    rng = _rng(55)
    d32 = rng.integers(-2**31, 2**31, size=300, dtype=np.int64).astype(np.int32)
    cases = []
    for version in (1, 2):
        kind = K.KIND_PAGE_V1 if version == 1 else K.KIND_PAGE_V2
        for n in (0, 7, 5000):
            for max_def in (0, 1):
                for vk, dic in ((K.KIND_DICT, d32), (K.KIND_DELTA64, None), (K.KIND_PLAIN, None)):
                    src, rl, dl = _make_page(rng, version, n, 0, max_def, vk, dic, plain_dtype=np.int64)
                    cases.append(Case(kind, src, n, max_rep=0, max_def=max_def, value_kind=vk, dictionary=dic,
                                      rep_len=rl, def_len=dl, flags=K.FLAG_NO_LEVEL_OUTPUT,
                                      **({"plain_dtype": np.int64} if vk == K.KIND_PLAIN else {})))
    run_cases(ctx, cases, seed=10)
    for c in cases:
        want = O.page_read(c.src, 1 if c.kind == K.KIND_PAGE_V1 else 2, c.n, 0, c.max_def,
                           {K.KIND_DICT: O.VALUES_DICT, K.KIND_DELTA64: O.VALUES_DELTA64, K.KIND_PLAIN: O.VALUES_PLAIN}[c.value_kind],
                           dictionary=c.dictionary, rep_len=c.rep_len, def_len=c.def_len, value_dtype=c.plain_dtype)
        r = c.result
        assert (r["status"], r["stage"], r["detail"], r["n_levels"]) == (want["status"], want["stage"], want["detail"], want["n"])
        if want["status"] == O.OK and c.n:
            assert r["not_null"] == want["not_null"]
            nn = want["not_null"]
            assert np.array_equal(c.values[:nn].view(np.uint8), want["values"][:nn].view(np.uint8))
        assert not guard_ok(np.zeros(1, np.int32))
        assert guard_ok(c.def_levels) and guard_ok(c.rep_levels)      # untouched
        assert guard_ok(c.guard) and guard_ok(c.def_guard) and guard_ok(c.rep_guard)


def test_plain_and_boolean_vs_oracle(ctx, golden_dir):
    """PLAIN fixed-width values and BOOLEAN PLAIN bits, as streams and as the values of data pages."""
    rng = _rng(70)
    cases = []
    for v in _load(golden_dir, "plain.json")["vectors"]:          # the reference's decoder vectors
        c = Case(K.KIND_PLAIN, np.array(v["data"], dtype=np.uint8), v["n"], plain_dtype=v["dtype"])
        c.golden = v
        cases.append(c)
    for dt in (np.int32, np.int64, np.float32, np.float64):
        es = np.dtype(dt).itemsize
        for n in (0, 1, 7, 8, 9, 511, 512, 513, 20000, 100003):
            src = rng.integers(0, 256, size=n * es, dtype=np.uint8)
            cases.append(Case(K.KIND_PLAIN, src, n, plain_dtype=dt))
            cases.append(Case(K.KIND_PLAIN, src, n + 3, plain_dtype=dt))                    # io.EOF
            if n:
                cases.append(Case(K.KIND_PLAIN, src[:-1].copy(), n, plain_dtype=dt))        # unexpected EOF
    for n in (0, 1, 7, 8, 9, 63, 64, 65, 4096, 4097, 20000, 100003):
        src = rng.integers(0, 256, size=(n + 7) // 8, dtype=np.uint8)
        cases.append(Case(K.KIND_BOOL_PLAIN, src, n))
        cases.append(Case(K.KIND_BOOL_PLAIN, src, n + 20))                                  # io.EOF at a group
        cases.append(Case(K.KIND_BOOL_PLAIN, src[: src.size // 2].copy(), n))
    for n in (0, 1, 9, 4097, 100003):                      # BooleanRLEDecoder: InitSize, then n x Next
        bits = _runny(rng, n, 1, 30)
        src = synth.length_prefixed(synth.hybrid_encode(bits, 1))
        cases.append(Case(K.KIND_BOOL_RLE, src, n))
        cases.append(Case(K.KIND_BOOL_RLE, src, n + 50))
        cases.append(Case(K.KIND_BOOL_RLE, src[: src.size // 2].copy(), n))
        cases.append(Case(K.KIND_BOOL_RLE, src[:3].copy(), n))
    run_cases(ctx, cases, seed=11)
    for c in cases:
        if c.kind == K.KIND_PLAIN:
            want = O.plain_decode(c.src, c.n, c.plain_dtype)
        elif c.kind == K.KIND_BOOL_RLE:
            st, vals, cons = O.hybrid_decode(c.src, 1, c.n, length_prefixed=True)
            want = {"status": st, "n_decoded": vals.size, "consumed": cons, "values": vals}
        else:
            want = O.bool_plain_decode(c.src, c.n)
        r = c.result
        assert r["status"] == want["status"], (c.kind, c.n, r, want["status"])
        assert r["n_decoded"] == want["n_decoded"]
        assert r["consumed"] == want["consumed"]
        nd = want["n_decoded"]
        assert np.array_equal(c.values[:nd].view(np.uint8), want["values"].view(np.uint8))
        assert guard_ok(c.guard)
        if hasattr(c, "golden"):
            assert c.values[:nd].tolist() == c.golden["values"] and nd == c.golden["count"]
    # as page values
    pages = []
    for version in (1, 2):
        kind = K.KIND_PAGE_V1 if version == 1 else K.KIND_PAGE_V2
        for n in (0, 1, 9, 1000, 20000):
            for max_rep, max_def in ((1, 3), (0, 0), (2, 2)):
                for vk, dt in ((K.KIND_PLAIN, np.int32), (K.KIND_PLAIN, np.float64), (K.KIND_BOOL_PLAIN, None),
                               (K.KIND_BOOL_RLE, None)):
                    src, rl, dl = _make_page(rng, version, n, max_rep, max_def, vk, plain_dtype=dt)
                    kw = dict(max_rep=max_rep, max_def=max_def, value_kind=vk, rep_len=rl, def_len=dl)
                    if dt is not None:
                        kw["plain_dtype"] = dt
                    pages.append(Case(kind, src, n, **kw))
                    if n == 1000:
                        for cut in (src.size // 2, src.size - 1, src.size - 60):
                            pages.append(Case(kind, src[:cut].copy(), n, **kw))
    run_cases(ctx, pages, seed=12)
    for c in pages:
        _check_page(c)


# ---------------------------------------------------------------------------
# single-call conveniences (host pointers, one reference call each)

def test_single_call_entry_points(ctx):
    rng = _rng(60)
    vals = _runny(rng, 3000, 7, 5)
    stream = synth.hybrid_encode(vals, 7)
    st, got, cons = ctx.hybrid_decode(stream, 7, 3000)
    assert st == K.OK and np.array_equal(got, vals) and cons == stream.size
    st, got, cons = ctx.hybrid_decode(synth.length_prefixed(stream), 7, 3000, length_prefixed=True)
    assert st == K.OK and np.array_equal(got, vals) and cons == stream.size + 4
    d = rng.standard_normal(128)
    st, got, nd, _ = ctx.dict_decode(synth.dict_page(np.minimum(vals, 127), w=7), d, 3000)
    assert st == K.OK and nd == 3000 and np.array_equal(got, d[np.minimum(vals, 127)])
    lv = ctx.levels_decode(synth.length_prefixed(stream), 127, 3000, 1)
    assert lv["status"] == K.OK and np.array_equal(lv["levels"], vals) and lv["not_null"] == int((vals == 127).sum())
    v64 = np.cumsum(rng.integers(0, 1 << 30, size=5000)).astype(np.int64)
    r = ctx.delta_decode(synth.delta_encode(v64, 64), 5000, 64)
    assert r["status"] == K.OK and np.array_equal(r["values"], v64)
    v32 = np.cumsum(rng.integers(-100, 1 << 12, size=5000)).astype(np.int32)
    r = ctx.delta_decode(synth.delta_encode(v32, 32), 5000, 32)
    assert r["status"] == K.OK and np.array_equal(r["values"], v32)


def test_plans_side_by_side_and_one_after_another(ctx):
    """Plans of one context: two alive at once, their runs interleaved; a destroyed plan's device
    block taken over by a smaller plan, outgrown by a larger one."""
    import torch

    rng = np.random.default_rng(77)
    dev = torch.device("cuda:0")

    def make(pages, n):
        vals = [np.cumsum(rng.integers(-50, 1000, size=n)).astype(np.int64) + p for p in range(pages)]
        streams = [synth.delta_encode(v, 64) for v in vals]
        offs = np.concatenate([[0], np.cumsum([s.size for s in streams])])
        descs = np.zeros(pages, dtype=capi.DESC_DTYPE)
        descs["kind"] = capi.KIND_DELTA64
        descs["src_off"], descs["src_len"], descs["n_values"] = offs[:-1], [s.size for s in streams], n
        descs["dst_off"] = np.arange(pages, dtype=np.uint64) * np.uint64(8 * n)
        src = torch.from_numpy(np.concatenate(streams + [np.zeros(16, np.uint8)])).to(dev)
        out = torch.zeros(pages * n * 8, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()                 # the plans run on the context's own stream
        return descs, src, out, np.concatenate(vals)

    def check(plan, src, out, want, pages):
        res = plan.results()
        assert res.size == pages and (res["status"] == capi.OK).all()
        assert np.array_equal(out.cpu().numpy().view(np.int64), want)

    a, b = make(300, 258), make(7, 1000)
    pa, pb = ctx.plan(a[0]), ctx.plan(b[0])
    for _ in range(2):
        pa.run(a[1].data_ptr(), a[1].numel(), a[2].data_ptr(), a[2].numel())
        pb.run(b[1].data_ptr(), b[1].numel(), b[2].data_ptr(), b[2].numel())
    check(pa, a[1], a[2], a[3], 300)
    check(pb, b[1], b[2], b[3], 7)
    pa.close()                                  # the context keeps this block ...
    c = make(40, 130)
    pc = ctx.plan(c[0])                         # ... a smaller plan takes it ...
    pc.run(c[1].data_ptr(), c[1].numel(), c[2].data_ptr(), c[2].numel())
    check(pc, c[1], c[2], c[3], 40)
    d = make(900, 64)
    pd = ctx.plan(d[0])                         # ... a larger one gets a block of its own
    pd.run(d[1].data_ptr(), d[1].numel(), d[2].data_ptr(), d[2].numel())
    pb.run(b[1].data_ptr(), b[1].numel(), b[2].data_ptr(), b[2].numel())
    check(pd, d[1], d[2], d[3], 900)
    check(pb, b[1], b[2], b[3], 7)
    for p in (pb, pc, pd):
        p.close()
    e = ctx.plan(a[0])
    e.run(a[1].data_ptr(), a[1].numel(), a[2].data_ptr(), a[2].numel())
    check(e, a[1], a[2], a[3], 300)
    e.close()


# ---------------------------------------------------------------------------
# The CPU reference configuration as worded: INT32 RLE_DICTIONARY, bit width 12, 1 M values

def test_c1_dictionary_column_width_12(ctx):
    """4096-entry int32 dictionary (low 32 bits of splitmix64, seed 0xD1C7), 1 M keys uniform in
    [0, 4096) (seed 1) written by the reference encoder's algorithm at width 12: as one 1 M-value
    PQ_KIND_DICT page and as 50 pages of 20 000 values, every value against the oracle."""
    d = synth.splitmix64(0xD1C7, 4096).astype(np.uint32).view(np.int32)
    keys = (synth.splitmix64(1, 1_000_000) % np.uint64(4096)).astype(np.int64)
    pages = [keys] + [keys[p * 20000:(p + 1) * 20000] for p in range(50)]
    cases = [Case(K.KIND_DICT, synth.dict_page(k, w=12), k.size, dictionary=d) for k in pages]
    assert all(c.src[0] == 12 for c in cases)
    run_cases(ctx, cases, seed=12)
    for c, k in zip(cases, pages):
        st, want, nd, consumed = O.dict_decode(c.src, d, k.size)
        assert st == O.OK and nd == k.size
        assert c.result["status"] == K.OK and c.result["n_decoded"] == k.size
        assert c.result["consumed"] == consumed
        assert np.array_equal(c.values, want) and np.array_equal(want, d[k])
        assert guard_ok(c.guard)
