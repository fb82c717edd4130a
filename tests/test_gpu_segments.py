"""GPU parity of the segmented DELTA_BINARY_PACKED decode (many waves to one page) against the CPU
oracle: values, status, counts and consumed bytes must be those of the whole-page decode, whatever
the segment length, wherever the stream is cut, and also when the guessed segment starts are wrong.
Reference: encoding/delta-binary-packed-decoder.go:29-170 (a page is one chain of blocks)."""
import numpy as np
import pytest

import oracle_lib as O
from gpu_harness import Case, guard_ok, run_cases
from parquet_go_amd import capi, synth

pytestmark = pytest.mark.gpu
K = capi


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def _check(c, bits):
    want = (O.delta_decode32 if bits == 32 else O.delta_decode64)(c.src[: c.result_src_len], c.n)
    r = c.result
    assert r["status"] == want["status"], (bits, c.n, c.result_src_len, r, want["status"], want["n_decoded"])
    assert r["n_decoded"] == want["n_decoded"], (bits, c.n, c.result_src_len, r, want["n_decoded"])
    got = c.values[: want["n_decoded"]]
    if not np.array_equal(got, want["values"]):
        bad = int(np.flatnonzero(got != want["values"])[0])
        raise AssertionError((bits, c.n, c.result_src_len, "first difference at", bad, got[bad], want["values"][bad]))
    assert guard_ok(c.guard)
    if want["status"] == O.OK:
        assert r["consumed"] == want["consumed"], (bits, c.n, c.result_src_len)


def _run(ctx, cases, bits, seg_bytes, seed=0):
    """The same cases cut into segments of seg_bytes and not cut at all."""
    try:
        for max_pages, sb in ((1 << 30, seg_bytes), (0, 0)):
            ctx.set_option(K.OPT_SEGMENT_MAX_PAGES, max_pages)
            ctx.set_option(K.OPT_SEGMENT_BYTES, sb)
            run_cases(ctx, cases, seed=seed)
            for c in cases:
                _check(c, bits)
    finally:
        ctx.set_option(K.OPT_SEGMENT_MAX_PAGES, 1024)
        ctx.set_option(K.OPT_SEGMENT_BYTES, 0)


def _values(rng, n, wmax, dt, base=None):
    steps = rng.integers(0, 1 << wmax, size=n) if wmax else np.zeros(n, np.int64)
    base = int(rng.integers(-1000, 1000)) if base is None else base
    with np.errstate(over="ignore"):
        return (np.cumsum(steps + base) + int(rng.integers(-2**40, 2**40))).astype(dt)


def _case(bits, stream, n):
    c = Case(K.KIND_DELTA32 if bits == 32 else K.KIND_DELTA64, stream, n)
    c.result_src_len = stream.size
    return c


@pytest.mark.parametrize("bits", [32, 64])
@pytest.mark.parametrize("seg_bytes", [2048, 4096, 16384])
def test_segmented_pages_vs_oracle(ctx, bits, seg_bytes):
    rng = np.random.default_rng(100 + bits + seg_bytes)
    dt = np.int32 if bits == 32 else np.int64
    cases = []
    for n in (5000, 20000, 20001, 16385, 16513, 100_000):
        for wmax in (0, 1, 5, 12, 16, 24, 31):
            for block in (128, 256):
                vals = _values(rng, n, wmax, dt)
                stream = synth.delta_encode(vals, bits, block, 4)
                for pull in sorted({n, n - 1, n + 3, n // 2, 129, 1}):
                    cases.append(_case(bits, stream, pull))
    _run(ctx, cases, bits, seg_bytes, seed=1)


@pytest.mark.parametrize("bits", [32, 64])
def test_one_million_value_pages(ctx, bits):
    """The large-page shape (1 M values a page) under the segment length the library picks."""
    rng = np.random.default_rng(7 + bits)
    dt = np.int32 if bits == 32 else np.int64
    cases = []
    for wmax, block in ((16, 128), (9, 256), (30, 128)):
        vals = _values(rng, 1_000_000, wmax, dt)
        stream = synth.delta_encode(vals, bits, block, 4)
        cases.append(_case(bits, stream, vals.size))
        # cut inside a late segment: the first failing byte still decides count and status
        for cut in (stream.size - 1, stream.size - 700, stream.size * 7 // 8 + 3, stream.size // 2):
            c = Case(cases[-1].kind, stream[:cut].copy(), vals.size)
            c.result_src_len = cut
            cases.append(c)
    try:
        ctx.set_option(K.OPT_SEGMENT_MAX_PAGES, 1024)
        ctx.set_option(K.OPT_SEGMENT_BYTES, 0)
        run_cases(ctx, cases, seed=2)
        for c in cases:
            _check(c, bits)
    finally:
        ctx.set_option(K.OPT_SEGMENT_BYTES, 0)


@pytest.mark.parametrize("bits", [32, 64])
def test_truncated_inside_every_segment(ctx, bits):
    rng = np.random.default_rng(300 + bits)
    dt = np.int32 if bits == 32 else np.int64
    cases = []
    for block in (128, 256):
        vals = _values(rng, 9000, 11, dt)
        stream = synth.delta_encode(vals, bits, block, 4)
        cuts = sorted(set(rng.integers(0, stream.size, size=250).tolist()) |
                      set(range(stream.size - 40, stream.size + 1)) |
                      {k * 2048 + d for k in range(1, stream.size // 2048 + 1) for d in (-2, -1, 0, 1, 2, 7)
                       if k * 2048 + d <= stream.size})
        for cut in cuts:
            c = Case(K.KIND_DELTA32 if bits == 32 else K.KIND_DELTA64, stream[:cut].copy(), vals.size)
            c.result_src_len = cut
            cases.append(c)
    _run(ctx, cases, bits, 2048, seed=3)


@pytest.mark.parametrize("bits", [32, 64])
def test_wrong_guesses_fall_back_to_the_chain(ctx, bits):
    """Streams in which byte offsets that are not block headers look like block headers (constant
    strides: blocks of a few small bytes), streams whose block headers the quick walk refuses
    (minDelta of five varint bytes and more), damaged blocks and layouts that are not tiled."""
    rng = np.random.default_rng(500 + bits)
    dt = np.int32 if bits == 32 else np.int64
    cases = []
    n = 60000
    # all deltas equal: every block is minDelta + four zero widths
    for stride in (0, 5, -3, 1000):
        vals = (np.arange(n, dtype=np.int64) * stride + 17).astype(dt)
        for block in (128, 256):
            cases.append(_case(bits, synth.delta_encode(vals, bits, block, 4), n))
    # tiny widths: packed bytes are small numbers, many offsets pass as headers
    for wmax in (1, 2, 3):
        vals = _values(rng, 300_000, wmax, dt, base=0)
        cases.append(_case(bits, synth.delta_encode(vals, bits, 128, 4), vals.size))
    # minDelta needs 5 varint bytes (int64 only: |minDelta| >= 2^27 with 32-bit residuals)
    big = 10**9 if bits == 64 else 2**28
    vals = _values(rng, 50000, 10, dt, base=big if bits == 64 else 0)
    if bits == 32:
        with np.errstate(over="ignore"):
            vals = (np.arange(50000, dtype=np.int64) * big + rng.integers(0, 1000, size=50000)).astype(np.int32)
    cases.append(_case(bits, synth.delta_encode(vals, bits, 128, 4), vals.size))
    # a width byte above 32 in the middle of the page, and garbage behind a good start
    vals = _values(rng, 40000, 13, dt)
    stream = synth.delta_encode(vals, bits, 128, 4)
    for frac in (0.3, 0.6, 0.95):
        s = stream.copy()
        at = int(s.size * frac)
        s[at: at + 600] = 0xC7
        cases.append(_case(bits, s, vals.size))
    s = stream.copy()
    s[s.size // 2:] = rng.integers(0, 256, size=s.size - s.size // 2, dtype=np.uint8)
    cases.append(_case(bits, s, vals.size))
    # layouts the tiled decoders decline (segment 0 hands the page on)
    for block, mb in ((128, 2), (512, 4), (256, 8)):
        cases.append(_case(bits, synth.delta_encode(vals, bits, block, mb), vals.size))
    _run(ctx, cases, bits, 2048, seed=4)
    _run(ctx, cases, bits, 8192, seed=5)


@pytest.mark.parametrize("bits", [32, 64])
def test_waves_that_run_out_of_patience_leave_the_page_to_the_whole_page_decoder(ctx, bits):
    """A segment's wave learns its place in the page from the waves in front of it, and its wait for them is
    bounded.  With the bound at 0 looks (PQ_OPT_SEGMENT_SPIN_LIMIT) every wave that would have to wait gives up at
    once: such a page is marked and decoded whole by the kernel behind.  Values, counts, status and consumed bytes
    are the oracle's all the same, for good pages, cut pages and pages whose guesses are wrong."""
    rng = np.random.default_rng(900 + bits)
    dt = np.int32 if bits == 32 else np.int64
    cases = []
    for n, wmax, block in ((300_000, 16, 128), (120_000, 7, 256), (60_000, 2, 128), (200_000, 24, 128)):
        vals = _values(rng, n, wmax, dt)
        stream = synth.delta_encode(vals, bits, block, 4)
        cases.append(_case(bits, stream, n))
        cases.append(_case(bits, stream, n // 3))
        cut = int(stream.size * 0.77)
        c = Case(cases[-1].kind, stream[:cut].copy(), n)
        c.result_src_len = cut
        cases.append(c)
    stride = (np.arange(60000, dtype=np.int64) * 5 + 17).astype(dt)    # every offset looks like a header
    cases.append(_case(bits, synth.delta_encode(stride, bits, 128, 4), stride.size))
    try:
        for limit in (0, 3):
            ctx.set_option(K.OPT_SEGMENT_SPIN_LIMIT, limit)
            _run(ctx, cases, bits, 2048, seed=21 + limit)
            _run(ctx, cases, bits, 16384, seed=23 + limit)
    finally:
        ctx.set_option(K.OPT_SEGMENT_SPIN_LIMIT, 1 << 20)


def _mixed_width_values(rng, n, dt):
    """Deltas whose miniblocks (32 values) differ in width, 0 and 32 bits included, under block
    minimums whose zig-zag varints take one to five bytes."""
    n -= 1    # the first value stands in the stream header; block k holds deltas 128 k ..
    deltas = np.zeros(n, dtype=np.int64)
    for b0 in range(0, n, 128):
        md = int(rng.choice([0, -3, 60, -5000, 70000, -(1 << 22), 1 << 26, -(1 << 27), (1 << 30) - 7, -(1 << 30)]))
        for m0 in range(b0, min(b0 + 128, n), 32):
            w = int(rng.choice([0, 1, 5, 12, 17, 22, 23, 24, 31, 32]))
            k = min(32, n - m0)
            noise = rng.integers(0, 1 << w, size=k, dtype=np.int64) if w else np.zeros(k, np.int64)
            if w and m0 == b0:
                noise[0] = 0    # the block's minimum is md itself
            deltas[m0:m0 + k] = md + noise
    with np.errstate(over="ignore"):
        v = np.cumsum(np.concatenate([[int(rng.integers(-2**31, 2**31))], deltas]))
    return v.astype(dt)    # int64 sums wrap the way the decoder's do; the low 32 bits are the int32 sums


@pytest.mark.parametrize("bits", [32, 64])
def test_blocks_of_mixed_widths_and_long_minimums(ctx, bits):
    """What the two-blocks-at-a-time loop of the 4-byte decode parses itself: per-miniblock widths
    from 0 to 32 inside one block, minDelta varints of 1..4 bytes (and of 5, which it hands on),
    pages that end after an odd or even number of blocks, cut anywhere in the last two pairs."""
    rng = np.random.default_rng(170 + bits)
    dt = np.int32 if bits == 32 else np.int64
    cases = []
    for n in (258, 386, 514, 1000, 128 * 41 + 7, 128 * 64 + 2, 40000):
        vals = _mixed_width_values(rng, n, dt)
        stream = synth.delta_encode(vals, bits)
        want = (O.delta_decode32 if bits == 32 else O.delta_decode64)(stream, n)
        assert want["status"] == O.OK and np.array_equal(want["values"], vals)
        for pull in sorted({n, n - 1, n - 127, n - 129, n // 2, n + 3}):
            cases.append(_case(bits, stream, pull))
        for cut in rng.integers(max(stream.size - 1500, 0), stream.size, size=12):
            cases.append(_case(bits, stream[: int(cut)].copy(), n))
    _run(ctx, cases, bits, 2048, seed=17)
    _run(ctx, cases, bits, 4096, seed=18)


def test_block_minimums_of_every_varint_length(ctx):
    """INT64 blocks whose minDelta zig-zag varint takes one to ten bytes, either sign, from block to block inside
    one page (a column of nanosecond timestamps is five bytes a block): decoded by the block loops themselves, and
    stepped over when the page is cut into segments.  Cut streams too: the failing byte may lie in a long varint."""
    rng = np.random.default_rng(4242)
    mags = [0, 5, 1 << 6, 1 << 13, 1 << 20, (1 << 27) - 1, 1 << 27, 10**9, 1 << 34, 1 << 41, 1 << 48, 1 << 55, 1 << 61,
            (1 << 62) + 12345]
    cases = []
    for n, block in ((3000, 128), (40_000, 128), (300_000, 128), (3000, 256), (300_000, 256)):
        nd = n - 1
        deltas = np.zeros(nd, dtype=np.int64)
        for b0 in range(0, nd, block):
            md = int(rng.choice(mags)) * int(rng.choice([1, -1]))
            if md > 2**31 - 1:
                md = -md    # the writers start a block's minimum at MaxInt32: larger ones only come negative
            k = min(block, nd - b0)
            noise = rng.integers(0, 1 << int(rng.choice([0, 3, 11, 16, 25, 32])), size=k, dtype=np.int64)
            noise[0] = 0
            with np.errstate(over="ignore"):
                deltas[b0:b0 + k] = np.int64(md) + noise
        with np.errstate(over="ignore"):
            vals = np.cumsum(np.concatenate([[int(rng.integers(-2**40, 2**40))], deltas])).astype(np.int64)
        if True:
            stream = synth.delta_encode(vals, 64, block, 4)
            want = O.delta_decode64(stream, n)
            assert want["status"] == O.OK and np.array_equal(want["values"], vals)
            cases.append(_case(64, stream, n))
            cases.append(_case(64, stream, n // 2 + 1))
            for cut in rng.integers(stream.size // 3, stream.size, size=6):
                c = Case(K.KIND_DELTA64, stream[: int(cut)].copy(), n)
                c.result_src_len = int(cut)
                cases.append(c)
    _run(ctx, cases, 64, 2048, seed=31)
    _run(ctx, cases, 64, 16384, seed=32)


def test_large_and_small_pages_in_one_plan(ctx):
    rng = np.random.default_rng(9)
    cases = []
    for i in range(40):
        n = int(rng.choice([10, 300, 5000, 40000, 250_000]))
        bits = 32 if i % 3 == 0 else 64
        dt = np.int32 if bits == 32 else np.int64
        vals = _values(rng, n, int(rng.integers(0, 25)), dt)
        c = _case(bits, synth.delta_encode(vals, bits, 128 if i % 2 else 256, 4), n)
        c.bits = bits
        cases.append(c)
    run_cases(ctx, cases, seed=6)
    for c in cases:
        _check(c, c.bits)


def test_a_healthy_page_is_decoded_by_all_its_segments(ctx):
    """On ordinary pages every guess is right: each segment is decoded by a wave of its own (all but
    the few behind the page's last block).  A regression here would only show as lost speed."""
    import torch
    rng = np.random.default_rng(77)
    dev = torch.device("cuda:0")
    for bits, wmax, block in ((64, 16, 128), (64, 9, 256), (32, 12, 128), (64, 3, 128)):
        dt = np.int32 if bits == 32 else np.int64
        vals = _values(rng, 1_000_000, wmax, dt)
        stream = synth.delta_encode(vals, bits, block, 4)
        descs = np.zeros(1, dtype=K.DESC_DTYPE)
        descs["kind"] = K.KIND_DELTA32 if bits == 32 else K.KIND_DELTA64
        descs["src_len"], descs["n_values"] = stream.size, vals.size
        t_src = torch.from_numpy(np.concatenate([stream, np.zeros(32, np.uint8)])).to(dev)
        t_out = torch.zeros(vals.size, dtype=torch.int32 if bits == 32 else torch.int64, device=dev)
        plan = ctx.plan(descs)
        plan.run(t_src.data_ptr(), stream.size, t_out.data_ptr(), vals.size * (bits // 8))
        res = plan.results()
        segs = plan.segments()
        plan.close()
        assert res[0]["status"] == K.OK and res[0]["n_decoded"] == vals.size
        assert np.array_equal(t_out.cpu().numpy(), vals)
        assert segs.size == (stream.size + 8191) // 8192
        apart = int((segs["decode"] != 0).sum())
        assert apart >= segs.size - 2, (bits, wmax, block, apart, segs.size)
        assert int((segs["decode"] == 2).sum()) == 1


# ---------------------------------------------------------------------------
# dictionary-index pages and raw hybrid streams in segments (runs of one repeated header)

def _bp_stream(keys, w, groups_per_run):
    """Bit-packed runs of groups_per_run groups each (the last one shorter), any header length."""
    keys = np.asarray(keys, dtype=np.int64)
    g = (keys.size + 7) // 8
    padded = np.concatenate([keys, np.zeros(g * 8 - keys.size, np.int64)])
    packed = synth.pack_bits(padded, w).reshape(g, w)
    out = []
    for lo in range(0, g, groups_per_run):
        hi = min(g, lo + groups_per_run)
        out.append(np.array(synth.uleb128(((hi - lo) << 1) | 1), dtype=np.uint8))
        out.append(packed[lo:hi].reshape(-1))
    return np.concatenate(out)


def _check_hybrid(c):
    if c.kind == K.KIND_DICT:
        st, want, nd, consumed = O.dict_decode(c.src[: c.result_src_len], c.dictionary, c.n)
    else:
        st, want, consumed = O.hybrid_decode(c.src[: c.result_src_len], c.bit_width, c.n, False)
        nd = want.size
    r = c.result
    assert r["status"] == st, (c.tag, r, st, nd)
    assert r["n_decoded"] == nd, (c.tag, r, nd)
    if st != K.DICT_INVALID_INDEX:
        assert np.array_equal(c.values[:nd], want[:nd]), c.tag
    assert guard_ok(c.guard), c.tag
    if st == O.OK:
        assert r["consumed"] == consumed, (c.tag, r, consumed)


def _hybrid_cases(rng):
    cases = []

    def add(tag, kind, stream, n, **kw):
        c = Case(kind, stream, n, **kw)
        c.result_src_len = stream.size
        c.tag = tag
        cases.append(c)
        return c
    d32 = rng.integers(-2**31, 2**31, size=4096, dtype=np.int64).astype(np.int32)
    d64 = rng.integers(-2**62, 2**62, size=700, dtype=np.int64)
    n = 300_000
    keys = rng.integers(0, 4096, size=n)
    page = np.concatenate([[12], synth.hybrid_encode_bitpacked(keys, 12)]).astype(np.uint8)
    for pull in (n, n - 1, n + 5, n // 2, 504 * 40, 504 * 40 + 1, 4096):
        add("dict w=12 pull %d" % pull, K.KIND_DICT, page, pull, dictionary=d32)
    for cut in (page.size - 1, page.size - 400, page.size // 2, page.size // 2 + 1, 30000, 757 * 50 + 1, 757 * 50 + 2):
        c = add("dict w=12 cut %d" % cut, K.KIND_DICT, page[:cut].copy(), n, dictionary=d32)
    # 8-byte dictionary entries, 10-bit keys
    k10 = rng.integers(0, 700, size=200_000)
    add("dict64 w=10", K.KIND_DICT, np.concatenate([[10], synth.hybrid_encode_bitpacked(k10, 10)]).astype(np.uint8),
        k10.size, dictionary=d64)
    # a key outside the dictionary in the middle of the page: count 0, "invalid index"
    bad = keys.copy()
    bad[n // 2 + 17] = 4095
    add("dict invalid index", K.KIND_DICT, np.concatenate([[12], synth.hybrid_encode_bitpacked(bad, 12)]).astype(np.uint8),
        n, dictionary=d32[:4000])
    # runs of 64 groups (two header bytes), of 5 groups (stride below the limit), of 1000 groups
    for gpr in (64, 5, 1000, 8):
        s = _bp_stream(keys[:150_000], 12, gpr)
        add("dict runs of %d groups" % gpr, K.KIND_DICT, np.concatenate([[12], s]).astype(np.uint8), 150_000,
            dictionary=d32)
    # equal runs first, then runs of every kind (the reference encoder on run-heavy keys), then equal runs again
    mixed = np.concatenate([keys[:100_000], np.repeat(rng.integers(0, 4096, size=2000), 37), keys[:90_000]])
    add("dict mixed", K.KIND_DICT, synth.dict_page(mixed, w=12), mixed.size, dictionary=d32)
    # damage: a run header in the middle replaced (other run length; zero-length run)
    for new in (0x7d, 0x01, 0x00):
        s = page.copy()
        s[1 + 757 * 200] = new
        add("dict header %#x in the middle" % new, K.KIND_DICT, s, n, dictionary=d32)
    # raw hybrid streams at several widths
    for w in (1, 3, 7, 16, 20, 32):
        v = rng.integers(0, 1 << min(w, 31), size=120_000)
        add("hybrid w=%d" % w, K.KIND_HYBRID, synth.hybrid_encode_bitpacked(v, w), v.size, bit_width=w)
    # not cut at all: RLE first
    rl = np.concatenate([np.zeros(64, np.int64), keys[:100_000]])
    add("dict rle first", K.KIND_DICT, synth.dict_page(rl, w=12), rl.size, dictionary=d32)
    return cases


@pytest.mark.parametrize("seg_bytes", [2048, 8192, 0])
def test_hybrid_streams_in_segments_vs_oracle(ctx, seg_bytes):
    rng = np.random.default_rng(900 + seg_bytes)
    cases = _hybrid_cases(rng)
    try:
        for max_pages in (1 << 30, 0):          # cut into segments, and not cut at all
            ctx.set_option(K.OPT_SEGMENT_MAX_PAGES, max_pages)
            ctx.set_option(K.OPT_SEGMENT_BYTES, seg_bytes)
            run_cases(ctx, cases, seed=8)
            for c in cases:
                _check_hybrid(c)
    finally:
        ctx.set_option(K.OPT_SEGMENT_MAX_PAGES, 1024)
        ctx.set_option(K.OPT_SEGMENT_BYTES, 0)


# ---------------------------------------------------------------------------
# hybrid streams of runs of any kind in segments: level streams, BOOLEAN RLE, run-heavy keys

def _runs(rng, n, hi, mean):
    """Values in runs of geometric length (mean `mean`), the way level streams and sorted keys look."""
    out = np.empty(n, dtype=np.int64)
    i = 0
    while i < n:
        ln = int(rng.geometric(1.0 / mean))
        out[i:i + ln] = rng.integers(0, hi)
        i += ln
    return out


def _check_run_stream(c):
    r = c.result
    if c.kind == K.KIND_LEVELS:
        mode = 1 if c.flags & K.FLAG_LENGTH_PREFIXED else 2
        want = O.levels_decode(c.src[: c.result_src_len], c.max_def, c.n, mode)
        assert r["status"] == want["status"], (c.tag, r, want["status"])
        assert r["not_null"] == want["not_null"], (c.tag, r, want["not_null"])
        if want["status"] == O.OK:
            assert r["n_decoded"] == c.n and np.array_equal(c.def_levels, want["levels"]), c.tag
            assert r["consumed"] == want["consumed"], (c.tag, r, want["consumed"])
        assert guard_ok(c.def_guard), c.tag
        return
    if c.kind == K.KIND_DICT:
        st, want, nd, consumed = O.dict_decode(c.src[: c.result_src_len], c.dictionary, c.n)
    else:
        w = 1 if c.kind == K.KIND_BOOL_RLE else c.bit_width
        prefixed = c.kind == K.KIND_BOOL_RLE or bool(c.flags & K.FLAG_LENGTH_PREFIXED)
        st, want, consumed = O.hybrid_decode(c.src[: c.result_src_len], w, c.n, prefixed)
        nd = want.size
    assert r["status"] == st, (c.tag, r, st, nd)
    assert r["n_decoded"] == nd, (c.tag, r, nd)
    if st != K.DICT_INVALID_INDEX:
        assert np.array_equal(c.values[:nd], want[:nd]), c.tag
    assert guard_ok(c.guard), c.tag
    if st == O.OK:
        assert r["consumed"] == consumed, (c.tag, r, consumed)


def _run_stream_cases(rng):
    cases = []

    def add(tag, kind, stream, n, **kw):
        c = Case(kind, stream, n, **kw)
        c.result_src_len = c.src.size
        c.tag = tag
        cases.append(c)
        return c
    d32 = rng.integers(-2**31, 2**31, size=4096, dtype=np.int64).astype(np.int32)
    n = 400_000
    # level streams the way a nested column has them: repetition levels (w = 1), definition levels (w = 2, 3)
    for max_level, p in ((1, 0.75), (3, 0.8), (7, 0.5)):
        lv = np.where(rng.random(n) < p, max_level, rng.integers(0, max_level + 1, size=n))
        s = synth.hybrid_encode(lv, int(max_level).bit_length())
        for pull in (n, n - 1, n // 2 + 3, 4096, n + 9):
            add("levels max %d raw pull %d" % (max_level, pull), K.KIND_LEVELS, s, pull, max_def=max_level)
        add("levels max %d prefixed" % max_level, K.KIND_LEVELS, synth.length_prefixed(s), n, max_def=max_level,
            flags=K.FLAG_LENGTH_PREFIXED)
        # a length prefix that says less than there is, one that says more
        short = synth.length_prefixed(s).copy()
        short[:4] = np.frombuffer(np.uint32(s.size // 2).tobytes(), np.uint8)
        add("levels max %d prefix says half" % max_level, K.KIND_LEVELS, short, n, max_def=max_level,
            flags=K.FLAG_LENGTH_PREFIXED)
        for cut in (s.size - 1, s.size - 100, s.size // 2, s.size // 3 + 1, 20000):
            add("levels max %d cut %d" % (max_level, cut), K.KIND_LEVELS, s[:cut].copy(), n, max_def=max_level)
    # long constant stretches: RLE runs with headers of three and four bytes
    lv = np.concatenate([np.full(50_000, 2), _runs(rng, 100_000, 3, 6), np.full(300_000, 1), _runs(rng, 100_000, 3, 3),
                         np.full(3_000_000, 2), _runs(rng, 60_000, 3, 10)])
    s = synth.hybrid_encode(lv, 2)
    add("levels long runs", K.KIND_LEVELS, s, lv.size, max_def=2)
    add("levels long runs, pull inside a long run", K.KIND_LEVELS, s, 1_000_000, max_def=2)
    # BOOLEAN RLE values and length-prefixed raw streams
    for mean in (3, 24, 300):
        bits = _runs(rng, n, 2, mean)
        s = synth.length_prefixed(synth.hybrid_encode(bits, 1))
        add("bool rle mean run %d" % mean, K.KIND_BOOL_RLE, s, n)
        add("bool rle mean run %d cut" % mean, K.KIND_BOOL_RLE, s[: s.size * 2 // 3].copy(), n)
        add("hybrid w=1 prefixed mean run %d" % mean, K.KIND_HYBRID, s, n, bit_width=1, flags=K.FLAG_LENGTH_PREFIXED)
    # run-heavy dictionary keys (bit-packed groups between RLE runs), also at widths whose RLE value takes 3 and 4 bytes
    keys = _runs(rng, n, 4096, 32)
    page = synth.dict_page(keys, w=12)
    for pull in (n, n // 3, n + 1):
        add("dict run-heavy pull %d" % pull, K.KIND_DICT, page, pull, dictionary=d32)
    for cut in (page.size - 1, page.size // 2, page.size // 5):
        add("dict run-heavy cut %d" % cut, K.KIND_DICT, page[:cut].copy(), n, dictionary=d32)
    bad = keys.copy()
    bad[n // 2] = 4095
    add("dict run-heavy invalid index", K.KIND_DICT, synth.dict_page(bad, w=12), n, dictionary=d32[:4000])
    for w in (5, 17, 20, 27, 32):
        v = _runs(rng, 150_000, 1 << min(w, 31), 12)
        add("hybrid w=%d runs" % w, K.KIND_HYBRID, synth.hybrid_encode(v, w), v.size, bit_width=w)
    # chains that never fall into step: every byte parses as "one repeat of the value 2" wherever a walk starts
    twos = np.full(60_000, 2, dtype=np.uint8)
    add("hybrid w=8 all 02", K.KIND_HYBRID, twos, 30_000, bit_width=8)
    add("hybrid w=8 all 02, odd length", K.KIND_HYBRID, twos[:-1].copy(), 29_999, bit_width=8)
    # ... and the real chain on the odd bytes (behind a dictionary page's width byte), where no guess lands
    add("dict w=8 all 02", K.KIND_DICT, np.concatenate([[8], twos]).astype(np.uint8), 30_000, dictionary=d32[:256])
    # damage in the middle: an empty run, a run value too large, random bytes from some point on
    lv = np.where(rng.random(n) < 0.8, 3, rng.integers(0, 4, size=n))
    s = synth.hybrid_encode(lv, 2)
    for at, val in ((s.size // 2, 0x00), (s.size // 2, 0x01), (s.size // 3, 0x80)):
        t = s.copy()
        t[at: at + 3] = val
        add("levels damaged at %d with %#x" % (at, val), K.KIND_LEVELS, t, n, max_def=3)
    t = s.copy()
    t[t.size // 2:] = rng.integers(0, 256, size=t.size - t.size // 2, dtype=np.uint8)
    add("levels random second half", K.KIND_LEVELS, t, n, max_def=3)
    add("hybrid w=2 random second half", K.KIND_HYBRID, t, n, bit_width=2)
    return cases


def _say(what):
    """Names the GPU step that comes next (a run that stalls is then known by its last line)."""
    import os
    print(what, flush=True)
    if os.path.isdir("gpurun_out"):
        with open("gpurun_out/segments_progress.log", "a") as f:
            f.write(what + "\n")


@pytest.mark.parametrize("seg_bytes", [2048, 4096, 0])
def test_run_streams_in_segments_vs_oracle(ctx, seg_bytes):
    rng = np.random.default_rng(1300 + seg_bytes)
    cases = _run_stream_cases(rng)
    families = {}
    for c in cases:                             # one plan per family of cases
        families.setdefault(" ".join(c.tag.split()[:2]), []).append(c)
    try:
        # not cut at all; cut into segments; cut, the decode finding the run headers itself (no table of them)
        for max_pages, tables in ((0, 1), (1 << 30, 1), (1 << 30, 0)):
            ctx.set_option(K.OPT_SEGMENT_MAX_PAGES, max_pages)
            ctx.set_option(K.OPT_SEGMENT_BYTES, seg_bytes)
            ctx.set_option(K.OPT_RUN_TABLES, tables)
            for name, group in families.items():
                _say("run streams: seg_bytes %d, max_pages %d, tables %d, %s (%d cases)" % (seg_bytes, max_pages, tables, name, len(group)))
                run_cases(ctx, group, seed=13)
                for c in group:
                    _check_run_stream(c)
            _say("run streams: seg_bytes %d, max_pages %d, tables %d, all in one plan" % (seg_bytes, max_pages, tables))
            run_cases(ctx, cases, seed=13)
            for c in cases:
                _check_run_stream(c)
    finally:
        ctx.set_option(K.OPT_SEGMENT_MAX_PAGES, 1024)
        ctx.set_option(K.OPT_SEGMENT_BYTES, 0)
        ctx.set_option(K.OPT_RUN_TABLES, 1)


def _hand_runs(rng, w, n_runs, rle_share, kinds=("rle8", "bp1")):
    """A hybrid stream put together run by run, with runs no encoder would close so early: RLE runs of 1..40
    repeats and bit-packed runs of 1..3 groups, so that a span of 64 bytes holds up to 32 run headers."""
    vb = (w + 7) // 8
    parts, values = [], []
    for _ in range(n_runs):
        if rng.random() < rle_share:
            cnt = int(rng.integers(1, 41))
            v = int(rng.integers(0, 1 << w))
            parts.append(np.array([cnt << 1], dtype=np.uint8))
            parts.append(np.frombuffer(int(v).to_bytes(vb, "little"), dtype=np.uint8))
            values.append(np.full(cnt, v, dtype=np.int64))
        else:
            g = int(rng.integers(1, 4))
            vals = rng.integers(0, 1 << w, size=8 * g, dtype=np.int64)
            parts.append(np.array([(g << 1) | 1], dtype=np.uint8))
            parts.append(synth.pack_bits(vals, w))
            values.append(vals)
    return np.concatenate(parts).astype(np.uint8), np.concatenate(values)


def test_spans_of_many_short_runs(ctx):
    """Streams whose runs are two or three bytes each: a span of 64 header bytes holds 16 to 32 of them, which is
    where the tiles follow the chain of run headers by pointer doubling instead of a step per run.  Whole streams,
    streams pulled short, cut streams, an RLE value too large for the width in the middle; small pages (one wave
    a stream) and the same cut into segments."""
    rng = np.random.default_rng(77001)
    cases = []

    def add(tag, kind, stream, n, **kw):
        c = Case(kind, stream, n, **kw)
        c.result_src_len = c.src.size
        c.tag = tag
        cases.append(c)

    for w in (1, 2, 3, 7, 8, 9, 12, 16):
        for share in (1.0, 0.85, 0.5):
            s, vals = _hand_runs(rng, w, 6000, share)
            n = vals.size
            st, want, _ = O.hybrid_decode(s, w, n, False)
            assert st == O.OK and np.array_equal(want, vals)
            add("hybrid w %d share %.2f" % (w, share), K.KIND_HYBRID, s, n, bit_width=w)
            add("hybrid w %d share %.2f pull" % (w, share), K.KIND_HYBRID, s, n // 2 + 5, bit_width=w)
            for cut in rng.integers(s.size // 4, s.size, size=3):
                add("hybrid w %d share %.2f cut %d" % (w, share, cut), K.KIND_HYBRID, s[: int(cut)].copy(), n, bit_width=w)
            if w < 8:
                bad = s.copy()      # the value byte of an RLE run somewhere in the middle gets a bit above the width
                at = int(np.flatnonzero((bad[: bad.size // 2] & 1) == 0)[-1]) + 1
                bad[at] |= 0x80
                add("hybrid w %d share %.2f bad value" % (w, share), K.KIND_HYBRID, bad, n, bit_width=w)
    for max_level in (1, 3):
        w = int(max_level).bit_length()
        s, vals = _hand_runs(rng, w, 20000, 0.9)
        vals = np.minimum(vals, max_level)
        s2, _ = s, None
        add("levels max %d hand runs" % max_level, K.KIND_LEVELS, s, vals.size, max_def=max_level)
        add("levels max %d hand runs cut" % max_level, K.KIND_LEVELS, s[: s.size * 2 // 3].copy(), vals.size,
            max_def=max_level)
    s, vals = _hand_runs(rng, 1, 30000, 0.95)
    add("bool rle hand runs", K.KIND_BOOL_RLE, synth.length_prefixed(s), vals.size)
    try:
        for max_pages, sb in ((0, 0), (1 << 30, 2048)):
            ctx.set_option(K.OPT_SEGMENT_MAX_PAGES, max_pages)
            ctx.set_option(K.OPT_SEGMENT_BYTES, sb)
            run_cases(ctx, cases, seed=5)
            for c in cases:
                _check_run_stream(c)
    finally:
        ctx.set_option(K.OPT_SEGMENT_MAX_PAGES, 1024)
        ctx.set_option(K.OPT_SEGMENT_BYTES, 0)



def _uleb(x):
    out = []
    while True:
        b = x & 0x7F
        x >>= 7
        out.append(b | (0x80 if x else 0))
        if not x:
            return np.array(out, dtype=np.uint8)


def _recipe_runs(rng, w, recipe, rounds):
    """A hybrid stream put together from a recipe of (kind, count) pairs, repeated: "rle" runs of `count`
    repeats (a header of three bytes from 8192 repeats on) and "bp" runs of `count` groups."""
    vb = (w + 7) // 8
    parts, values = [], []
    for _ in range(rounds):
        for kind, cnt in recipe:
            cnt = int(cnt() if callable(cnt) else cnt)
            if kind == "rle":
                v = int(rng.integers(0, 1 << w))
                parts += [_uleb(cnt << 1), np.frombuffer(v.to_bytes(vb, "little"), dtype=np.uint8)]
                values.append(np.full(cnt, v, dtype=np.int64))
            else:
                vals = rng.integers(0, 1 << w, size=8 * cnt, dtype=np.int64)
                parts += [_uleb((cnt << 1) | 1), synth.pack_bits(vals, w)]
                values.append(vals)
    return np.concatenate(parts).astype(np.uint8), np.concatenate(values)


def test_segments_whose_run_headers_the_tiles_cannot_all_take(ctx):
    """The decode of a segment takes the places of its run headers from the walk that proved the segment and
    builds whole tiles from them; what such a tile cannot hold goes to the other loops and comes back: more
    than 64 runs in a tile (runs of one and two repeats), headers of three bytes between short runs, bit-packed
    runs long enough for the bulk path, a pull that ends inside a tile, in a run, at a run's first value."""
    rng = np.random.default_rng(4242)
    cases = []

    def add(tag, kind, stream, n, **kw):
        c = Case(kind, stream, n, **kw)
        c.result_src_len = c.src.size
        c.tag = tag
        cases.append(c)

    small = lambda: rng.integers(1, 3)
    some = lambda: rng.integers(1, 30)
    recipes = {
        "ones and twos": ([("rle", small)], 60000),
        "ones and twos between groups": ([("rle", small), ("rle", small), ("rle", small), ("bp", 1)], 15000),
        "three-byte headers between short runs": ([("rle", some)] * 40 + [("rle", lambda: rng.integers(8192, 20000))] +
                                                  [("bp", lambda: rng.integers(1, 4))] * 7, 120),
        "bulk runs between short runs": ([("rle", some)] * 25 + [("bp", lambda: rng.integers(32, 200))] +
                                         [("bp", 2), ("rle", some)] * 5, 150),
        "two-byte headers": ([("rle", lambda: rng.integers(64, 400)), ("bp", lambda: rng.integers(1, 3))], 4000),
    }
    for name, (recipe, rounds) in recipes.items():
        for w in (1, 2, 12):
            s, vals = _recipe_runs(rng, w, recipe, rounds)
            n = vals.size
            st, want, _ = O.hybrid_decode(s, w, n, False)
            assert st == O.OK and np.array_equal(want, vals)
            for pull in (n, n - 1, n // 2, n // 2 + 1021, 1024 * (n // 3072), 1024 * (n // 3072) + 1):
                add("hybrid w %d %s pull %d" % (w, name, pull), K.KIND_HYBRID, s, pull, bit_width=w)
            add("hybrid w %d %s cut" % (w, name), K.KIND_HYBRID, s[: s.size * 3 // 5].copy(), n, bit_width=w)
            if w <= 2:
                lv = (1 << w) - 1
                add("levels max %d %s" % (lv, name), K.KIND_LEVELS, s, n, max_def=lv)
                add("levels max %d %s pull" % (lv, name), K.KIND_LEVELS, s, n // 2 + 7, max_def=lv)
            if w == 1:
                add("bool rle %s" % name, K.KIND_BOOL_RLE, synth.length_prefixed(s), n)
            if w == 12:
                d32 = rng.integers(-2**31, 2**31, size=4096, dtype=np.int64).astype(np.int32)
                page = np.concatenate([[12], s]).astype(np.uint8)
                add("dict w 12 %s" % name, K.KIND_DICT, page, n, dictionary=d32)
                add("dict w 12 %s short dictionary" % name, K.KIND_DICT, page, n, dictionary=d32[:4090])
    try:
        # cut with and without the table of run headers (PQ_OPT_RUN_TABLES), cut at the default length, not cut
        for max_pages, sb, tables in ((1 << 30, 2048, 1), (1 << 30, 2048, 0), (1 << 30, 0, 1), (0, 0, 1)):
            ctx.set_option(K.OPT_SEGMENT_MAX_PAGES, max_pages)
            ctx.set_option(K.OPT_SEGMENT_BYTES, sb)
            ctx.set_option(K.OPT_RUN_TABLES, tables)
            _say("recipes: max_pages %d seg_bytes %d tables %d (%d cases)" % (max_pages, sb, tables, len(cases)))
            run_cases(ctx, cases, seed=9)
            for c in cases:
                _check_run_stream(c)
    finally:
        ctx.set_option(K.OPT_SEGMENT_MAX_PAGES, 1024)
        ctx.set_option(K.OPT_SEGMENT_BYTES, 0)
        ctx.set_option(K.OPT_RUN_TABLES, 1)
