"""GPU parity for DELTA_LENGTH_BYTE_ARRAY and DELTA_BYTE_ARRAY (types/byte-array.go:168-219, 279-348):
the HIP path through the C ABI against the CPU oracle — same status, counts, consumed bytes, offsets
and value bytes — as streams, as the values of data pages, and through the single-call entry points.
The reference holds no test for these decoders (parity unpinned; see test_byte_array_oracle.py)."""
import numpy as np
import pytest

import oracle_lib as O
from byte_array_cases import damaged_dba, damaged_dlba, dba_stream, dlba_stream, random_values, total_len
from gpu_harness import Case, guard_ok, run_cases
from parquet_go_amd import capi, synth

pytestmark = pytest.mark.gpu

K = capi


@pytest.fixture(scope="module", params=["wave_per_page", "workgroup_per_page"])
def ctx(request):
    c = capi.Context(0)
    c.set_option(K.OPT_HYBRID_WAVE_MIN_PAGES, 0 if request.param == "wave_per_page" else 1 << 30)
    yield c
    c.close()


def _declared(stream):
    """Lengths the offsets area must have room for: the third varint of the stream's header,
    whether or not the rest of the header is any good."""
    pos, vals = 0, []
    for _ in range(3):
        x, shift = 0, 0
        while pos < len(stream):
            b = int(stream[pos])
            pos += 1
            x |= (b & 0x7F) << shift
            shift += 7
            if not b & 0x80:
                break
        vals.append(x)
    return min(vals[2], 1 << 20)


def _case(kind, stream, n, data_cap=None):
    stream = np.ascontiguousarray(stream, dtype=np.uint8)
    return Case(kind, stream, n, offs_cap=max(n, _declared(stream)), data_cap=data_cap)


def _check(c):
    if c.kind == K.KIND_DELTA_LENGTH_BA:
        want = O.delta_length_ba_decode(c.src, c.n)
    else:
        want = O.delta_ba_decode(c.src, c.n, c.data_cap)
    r = c.result
    assert guard_ok(c.guard) and guard_ok(c.area_guard)
    if want["status"] == O.OUT_OF_RANGE_CAP:
        assert r["status"] == K.ERR_OUT_OF_RANGE and r["n_decoded"] == 0
        return
    assert r["status"] == want["status"], (c.n, c.src.size, r, want["status"], want["n_decoded"])
    init_failed = want["init_status"] != O.OK
    assert (r["stage"] == K.STAGE_VALUES_INIT) == init_failed
    if init_failed:
        return
    assert r["n_decoded"] == want["n_decoded"]
    assert r["consumed"] == want["consumed"], (r, want["consumed"])
    nd = want["n_decoded"]
    assert np.array_equal(c.ba_offsets[: nd + 1], want["offsets"])
    assert np.array_equal(c.ba_data[: int(want["offsets"][nd])], want["data"])


@pytest.mark.parametrize("shape", ["short", "mixed", "sorted", "runs"])
def test_encoder_streams_vs_oracle(ctx, shape):
    rng = np.random.default_rng(100 + ["short", "mixed", "sorted", "runs"].index(shape))
    cases = []
    for n in (0, 1, 2, 3, 63, 64, 65, 127, 128, 129, 130, 200, 1000, 5000):
        vals = random_values(rng, n, shape)
        room = total_len(vals)
        ls, fs = synth.delta_length_ba_encode(vals), synth.delta_ba_encode(vals)
        for pull in sorted({n, n + 3, n // 2, max(n - 1, 0)}):
            cases.append(_case(K.KIND_DELTA_LENGTH_BA, ls, pull))
            cases.append(_case(K.KIND_DELTA_BA, fs, pull, data_cap=room))
        cases.append(_case(K.KIND_DELTA_BA, fs, n, data_cap=room + 37))
        if room:
            cases.append(_case(K.KIND_DELTA_BA, fs, n, data_cap=room - 1))
    run_cases(ctx, cases, seed=3)
    for c in cases:
        _check(c)
    good = [c for c in cases if c.n >= 2 and c.n % 128 != 1 and c.result["status"] == K.OK]
    assert len(good) > 20


def test_damaged_streams_vs_oracle(ctx):
    rng = np.random.default_rng(11)
    cases = [_case(K.KIND_DELTA_LENGTH_BA, s, n) for s, n in damaged_dlba(rng)]
    cases += [_case(K.KIND_DELTA_BA, s, n, data_cap=cap) for s, n, cap in damaged_dba(rng)]
    run_cases(ctx, cases, seed=4)
    for c in cases:
        _check(c)
    seen = {int(c.result["status"]) for c in cases}
    assert {K.OK, K.EOF, K.UNEXPECTED_EOF, K.BYTE_ARRAY_NEGATIVE_LEN, K.BYTE_ARRAY_COUNT_MISMATCH,
            K.BYTE_ARRAY_INVALID_PREFIX, K.ERR_OUT_OF_RANGE} <= seen


def test_truncated_at_every_byte(ctx):
    rng = np.random.default_rng(12)
    vals = random_values(rng, 150, "runs")
    ls, fs = synth.delta_length_ba_encode(vals), synth.delta_ba_encode(vals)
    cases = [_case(K.KIND_DELTA_LENGTH_BA, ls[:cut].copy(), 150) for cut in range(ls.size + 1)]
    cases += [_case(K.KIND_DELTA_BA, fs[:cut].copy(), 150, data_cap=total_len(vals)) for cut in range(fs.size + 1)]
    run_cases(ctx, cases, seed=5)
    for c in cases:
        _check(c)


def test_random_length_lists(ctx):
    """Length lists no encoder writes: random prefixes (legal and not), negative and huge entries."""
    rng = np.random.default_rng(14)
    cases = []
    for it in range(300):
        n = int(rng.integers(2, 400))
        if n % 128 == 1:
            n += 1
        sl = rng.integers(0, int(rng.choice([3, 20, 400])), size=n)
        body = rng.integers(0, 256, size=int(sl.sum()), dtype=np.uint8)
        pl = np.zeros(n, np.int64)
        prev = 0
        for i in range(n):
            pl[i] = int(rng.integers(0, prev + 1)) if rng.random() < 0.8 else 0
            prev = pl[i] + sl[i]
        mode = it % 6
        if mode == 1:
            pl[int(rng.integers(0, n))] += int(rng.integers(1, 50))      # very likely too long a prefix
        elif mode == 2:
            sl[int(rng.integers(0, n))] = -int(rng.integers(1, 9))
        elif mode == 3:
            pl[int(rng.integers(0, n))] = -int(rng.integers(1, 30))
        elif mode == 4:
            body = body[: int(rng.integers(0, body.size + 1))]
        room = int((np.maximum(pl, 0) + np.maximum(sl, 0)).sum()) + 64
        cases.append(_case(K.KIND_DELTA_BA, dba_stream(pl, sl, body), int(rng.integers(0, n + 4)), data_cap=room))
        cases.append(_case(K.KIND_DELTA_LENGTH_BA, dlba_stream(sl, body), int(rng.integers(0, n + 4))))
    run_cases(ctx, cases, seed=6)
    for c in cases:
        _check(c)


def test_offsets_area_too_small_is_refused(ctx):
    rng = np.random.default_rng(15)
    vals = random_values(rng, 300)
    for kind, stream in ((K.KIND_DELTA_LENGTH_BA, synth.delta_length_ba_encode(vals)),
                         (K.KIND_DELTA_BA, synth.delta_ba_encode(vals))):
        c = Case(kind, stream, 10, offs_cap=299, data_cap=total_len(vals))
        run_cases(ctx, [c], seed=7)
        assert c.result["status"] == K.ERR_OUT_OF_RANGE and c.result["stage"] == K.STAGE_VALUES_INIT
        assert guard_ok(c.guard) and guard_ok(c.area_guard)


def test_byte_array_values_of_data_pages(ctx):
    rng = np.random.default_rng(16)
    cases, wants = [], []
    for version in (1, 2):
        for n in (0, 9, 300, 4000):
            deff = np.where(rng.random(n) < 0.7, 2, rng.integers(0, 3, size=n))
            rep = rng.integers(0, 2, size=n)
            nn = int((deff == 2).sum())
            vals = random_values(rng, nn, "sorted" if n != 300 else "mixed")
            rs, ds = synth.hybrid_encode(rep, 1), synth.hybrid_encode(deff, 2)
            for kind, okind, enc in ((K.KIND_DELTA_LENGTH_BA, O.VALUES_DELTA_LENGTH_BA, synth.delta_length_ba_encode),
                                     (K.KIND_DELTA_BA, O.VALUES_DELTA_BA, synth.delta_ba_encode)):
                vs = enc(vals)
                if version == 1:
                    page, rl, dl = np.concatenate([synth.length_prefixed(rs), synth.length_prefixed(ds), vs]), 0, 0
                else:
                    page, rl, dl = np.concatenate([rs, ds, vs]), rs.size, ds.size
                for cut in (page.size, page.size - 2, page.size - vs.size + 3):
                    src = page[:cut].copy()
                    room = total_len(vals)
                    c = Case(K.KIND_PAGE_V1 if version == 1 else K.KIND_PAGE_V2, src, n, max_rep=1, max_def=2,
                             value_kind=kind, rep_len=rl, def_len=dl, offs_cap=max(n, nn), data_cap=room)
                    cases.append(c)
                    wants.append(O.page_read(src, version, n, 1, 2, okind, rep_len=rl, def_len=dl, data_cap=room))
    run_cases(ctx, cases, seed=8)
    n_ok = 0
    for c, want in zip(cases, wants):
        r = c.result
        assert (r["status"], r["stage"], r["detail"]) == (want["status"], want["stage"], want["detail"]), (r, want)
        assert guard_ok(c.guard) and guard_ok(c.area_guard) and guard_ok(c.def_guard) and guard_ok(c.rep_guard)
        if want["status"] == O.OK and c.n:
            n_ok += 1
            nn = want["not_null"]
            assert r["not_null"] == nn
            assert np.array_equal(c.rep_levels, want["rep"]) and np.array_equal(c.def_levels, want["def_"])
            off = want["values"][: 4 * (c.n + 1)].view(np.uint32)
            assert np.array_equal(c.ba_offsets[: nn + 1], off[: nn + 1])
            assert np.array_equal(c.ba_data[: int(off[nn])], want["values"][4 * (c.n + 1): 4 * (c.n + 1) + int(off[nn])])
    assert n_ok >= 8


def test_single_call_entry_points(ctx):
    rng = np.random.default_rng(17)
    vals = random_values(rng, 777, "runs")
    ls, fs = synth.delta_length_ba_encode(vals), synth.delta_ba_encode(vals)
    for pull in (0, 5, 777, 800):
        got, want = ctx.delta_length_ba_decode(ls, pull), O.delta_length_ba_decode(ls, pull)
        assert (got["status"], got["n_decoded"], got["consumed"]) == (want["status"], want["n_decoded"], want["consumed"])
        assert np.array_equal(got["offsets"], want["offsets"]) and np.array_equal(got["data"], want["data"])
        got, want = ctx.delta_ba_decode(fs, pull, total_len(vals)), O.delta_ba_decode(fs, pull, total_len(vals))
        assert (got["status"], got["n_decoded"], got["consumed"]) == (want["status"], want["n_decoded"], want["consumed"])
        assert np.array_equal(got["offsets"], want["offsets"]) and np.array_equal(got["data"], want["data"])
    bad = ctx.delta_length_ba_decode(ls[:3], 5)
    assert bad["init_failed"] == 1 and bad["n_decoded"] == 0
    with pytest.raises(capi.PqError):
        ctx.delta_ba_decode(fs, 777, total_len(vals) - 1)


def test_many_pages_of_strings(ctx):
    """A column chunk's worth: 2000 pages of 2000 values, every byte checked against the input."""
    rng = np.random.default_rng(18)
    words = [bytes(rng.integers(97, 123, size=int(rng.integers(3, 14)), dtype=np.uint8)) for _ in range(5000)]
    base_vals = sorted(words[:2000])
    ls, fs = synth.delta_length_ba_encode(base_vals), synth.delta_ba_encode(base_vals)
    room = total_len(base_vals)
    flat = np.frombuffer(b"".join(base_vals), dtype=np.uint8)
    offs = np.concatenate([[0], np.cumsum([len(v) for v in base_vals])]).astype(np.uint32)
    cases = []
    for p in range(2000):
        cases.append(Case(K.KIND_DELTA_BA if p % 2 else K.KIND_DELTA_LENGTH_BA, fs if p % 2 else ls, 2000,
                          offs_cap=2000, data_cap=room))
    run_cases(ctx, cases, seed=9)
    for c in cases:
        assert c.result["status"] == K.OK and c.result["n_decoded"] == 2000
        assert np.array_equal(c.ba_offsets, offs) and np.array_equal(c.ba_data[:room], flat)
        assert guard_ok(c.guard) and guard_ok(c.area_guard)


def test_values_of_megabytes(ctx):
    """A few very long values among short ones: the value-by-value path, offsets beyond 2^24."""
    rng = np.random.default_rng(19)
    vals = []
    for i in range(130):
        if i in (3, 64, 65, 129):
            vals.append(bytes(rng.integers(0, 256, size=int(rng.integers(1 << 20, 3 << 20)), dtype=np.uint8)))
        elif i in (4, 66):
            vals.append(vals[-1][: 1 << 19] + b"tail")          # a long shared prefix
        else:
            vals.append(bytes(rng.integers(0, 256, size=int(rng.integers(0, 30)), dtype=np.uint8)))
    room = total_len(vals)
    flat = np.frombuffer(b"".join(vals), dtype=np.uint8)
    offs = np.concatenate([[0], np.cumsum([len(v) for v in vals])]).astype(np.uint32)
    cases = [_case(K.KIND_DELTA_LENGTH_BA, synth.delta_length_ba_encode(vals), 130),
             _case(K.KIND_DELTA_BA, synth.delta_ba_encode(vals), 130, data_cap=room)]
    run_cases(ctx, cases, seed=11)
    for c in cases:
        assert c.result["status"] == K.OK and c.result["n_decoded"] == 130
        assert np.array_equal(c.ba_offsets[:131], offs) and np.array_equal(c.ba_data[:room], flat)
        assert guard_ok(c.guard) and guard_ok(c.area_guard)


@pytest.mark.parametrize("seed", [31, 32, 33, 34])
def test_values_around_the_output_window(ctx, seed):
    """Value lengths on either side of what a wave keeps of its output in LDS (2 KiB), long shared
    prefixes across them: batches of every size from 1 to 64, values that go through memory, values
    whose predecessor did, window wrap-around at every alignment of the output."""
    rng = np.random.default_rng(seed)
    vals, prev = [], b""
    for i in range(700):
        pick = rng.integers(0, 10)
        if pick < 5:
            n = int(rng.integers(0, 40))
        elif pick < 7:
            n = int(rng.integers(900, 1100))
        elif pick < 9:
            n = int(rng.integers(2030, 2070))
        else:
            n = int(rng.integers(3000, 5000))
        keep = int(rng.integers(0, min(len(prev), n) + 1)) if rng.integers(0, 4) else 0
        if rng.integers(0, 8) == 0:
            keep = min(len(prev), n)                  # the whole value in front, or the whole of this one
        prev = prev[:keep] + bytes(rng.integers(0, 256, size=n - keep, dtype=np.uint8))
        vals.append(prev)
    room = total_len(vals)
    flat = np.frombuffer(b"".join(vals), dtype=np.uint8)
    offs = np.concatenate([[0], np.cumsum([len(v) for v in vals])]).astype(np.uint32)
    fs, ls = synth.delta_ba_encode(vals), synth.delta_length_ba_encode(vals)
    cases = [_case(K.KIND_DELTA_BA, fs, 700, data_cap=room + k) for k in range(0, 48, 3)]
    cases += [_case(K.KIND_DELTA_BA, fs, n, data_cap=room) for n in (1, 5, 64, 65, 333, 699)]
    cases.append(_case(K.KIND_DELTA_LENGTH_BA, ls, 700))
    run_cases(ctx, cases, seed=seed)
    for c in cases:
        _check(c)
        nd = min(c.n, 700)
        assert c.result["status"] == K.OK and c.result["n_decoded"] == nd
        assert np.array_equal(c.ba_offsets[: nd + 1], offs[: nd + 1])
        assert np.array_equal(c.ba_data[: int(offs[nd])], flat[: int(offs[nd])])


def test_lengths_at_the_limits_of_the_one_pass_path(ctx):
    """DELTA_LENGTH_BYTE_ARRAY lists whose entries sit at 2^24 and whose totals pass 2^31 (the body is
    short, so DecodeValues runs out of bytes somewhere): the decoder has to notice that its 32-bit
    running sums are no longer safe and still stop at the value the reference stops at."""
    rng = np.random.default_rng(41)
    cases = []
    for it in range(120):
        n = int(rng.integers(2, 500))
        if n % 128 == 1:
            n += 1
        sl = rng.integers(0, 30, size=n).astype(np.int64)
        kind = it % 6
        where = int(rng.integers(0, n))
        if kind == 0:
            sl[where] = (1 << 24) - 1                  # the largest length the one pass takes
        elif kind == 1:
            sl[where] = 1 << 24
        elif kind == 2:
            sl[rng.integers(0, n, size=3)] = (1 << 31) - 1 - int(rng.integers(0, 3))   # totals beyond 2^32
        elif kind == 3:
            sl[:] = rng.integers((1 << 23) - 40, 1 << 23, size=n)      # each fine, the total passes 2^31
        elif kind == 4:
            sl[where] = -1
            sl[(where + 7) % n] = 1 << 30
        body_len = int(rng.choice([0, 1, 29, 300, int(np.clip(sl, 0, None)[: n // 2].sum() % 5000)]))
        body = rng.integers(0, 256, size=body_len, dtype=np.uint8)
        for pull in (n, int(rng.integers(0, n + 2))):
            cases.append(_case(K.KIND_DELTA_LENGTH_BA, dlba_stream(sl, body), pull))
    run_cases(ctx, cases, seed=12)
    for c in cases:
        _check(c)
    seen = {int(c.result["status"]) for c in cases}
    assert {K.EOF, K.UNEXPECTED_EOF, K.BYTE_ARRAY_NEGATIVE_LEN} <= seen


def _front_coded(rng, n, max_len, chain, alphabet=4):
    """n values of at most max_len bytes, each keeping some bytes of the one in front.  chain: the
    first `chain` bytes never change (no value but the first starts its suffix there), so whoever
    starts in the middle of the list has to look all the way back for them."""
    vals, prev = [], bytes(rng.integers(97, 97 + alphabet, size=max(chain, 1), dtype=np.uint8))
    vals.append(prev)
    for _ in range(n - 1):
        ln = int(rng.integers(chain, max_len + 1))
        keep = int(rng.integers(chain, min(len(prev), ln) + 1))
        prev = prev[:keep] + bytes(rng.integers(97, 97 + alphabet, size=ln - keep, dtype=np.uint8))
        vals.append(prev)
    return vals


@pytest.mark.parametrize("max_len", [1, 9, 16, 17, 32, 33, 64, 65, 128, 129, 256, 257])
def test_short_values_of_every_length_class(ctx, max_len):
    """Pages of values no longer than max_len (1 .. 257 bytes), front-coded: short and long lists,
    prefixes that reach back to the first value of the page, pulls that end in the middle, value
    areas with room to spare, a suffix area that runs out, a length list that fails in the middle."""
    rng = np.random.default_rng(500 + max_len)
    cases = []
    for n, chain in ((100, 0), (129, 1), (511, 0), (512, 0), (513, 1), (700, 0), (2047, 0), (2048, 1), (3000, 0),
                     (3000, min(3, max_len)), (20000 if max_len <= 33 else 5000, min(2, max_len))):
        vals = _front_coded(rng, n, max_len, chain)
        room = total_len(vals)
        fs = synth.delta_ba_encode(vals)
        for pull in (n, n - 1, n // 2 + 7, n // 4, 3):
            cases.append(_case(K.KIND_DELTA_BA, fs, pull, data_cap=room))
        for extra in (1, 5, 16):
            cases.append(_case(K.KIND_DELTA_BA, fs, n, data_cap=room + extra))
        cases.append(_case(K.KIND_DELTA_BA, fs[: fs.size - room // 3].copy(), n, data_cap=room))   # the bytes run out
    # sorted words, as a dictionary of strings would be written
    words = sorted(bytes(rng.integers(97, 123, size=int(rng.integers(min(3, max_len), max_len + 1)), dtype=np.uint8))
                   for _ in range(6000))
    cases.append(_case(K.KIND_DELTA_BA, synth.delta_ba_encode(words), 6000, data_cap=total_len(words)))
    # a prefix longer than the value in front, past the first cut
    n = 2000
    sl = rng.integers(0, max_len // 2 + 1, size=n)
    pl = np.zeros(n, np.int64)
    prev = 0
    for i in range(n):
        pl[i] = int(rng.integers(0, min(prev, max_len - sl[i]) + 1))
        prev = pl[i] + sl[i]
    body = rng.integers(0, 256, size=int(sl.sum()), dtype=np.uint8)
    cases.append(_case(K.KIND_DELTA_BA, dba_stream(pl, sl, body), n, data_cap=int((pl + sl).sum())))
    bad = pl.copy()
    bad[1500] = pl[1499] + sl[1499] + 1
    cases.append(_case(K.KIND_DELTA_BA, dba_stream(bad, sl, body), n, data_cap=int((bad + sl).sum())))
    neg = sl.copy()
    neg[1200] = -1
    cases.append(_case(K.KIND_DELTA_BA, dba_stream(pl, neg, body), n, data_cap=int((pl + sl).sum())))
    run_cases(ctx, cases, seed=max_len)
    for c in cases:
        _check(c)
    assert sum(1 for c in cases if c.result["status"] == K.OK and c.result["n_decoded"] >= 512) >= 20
