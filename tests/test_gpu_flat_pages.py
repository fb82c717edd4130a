"""GPU parity of data pages of flat columns (max_rep == 0) that a plan decodes as the value streams they
are, so that a few large pages are cut into segments like bare streams: descriptors built from walked
chunks (pq_chunks_page_descs, PQ_KIND_PAGE_V1 / _V2) and by hand, against the oracle's page reader
(dataPageReaderV1/V2.read + ReadValues, layout/page-v1.go:40-116, page-v2.go:42-149; both level decoders
are ConstDecoder(0) for such a column, layout/chunk.go:111-124)."""
import numpy as np
import pytest

import oracle_lib as O
import walk_cases as W
from gpu_harness import Case, guard_ok, run_cases
from parquet_go_amd import capi, synth

pytestmark = pytest.mark.gpu
K = capi

ORACLE_KIND = {K.KIND_DICT: O.VALUES_DICT, K.KIND_DELTA32: O.VALUES_DELTA32, K.KIND_DELTA64: O.VALUES_DELTA64,
               K.KIND_PLAIN: O.VALUES_PLAIN, K.KIND_BOOL_PLAIN: O.VALUES_BOOL_PLAIN, K.KIND_BOOL_RLE: O.VALUES_BOOL_RLE}


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def _values_stream(rng, vk, n, dictionary=None, dt=None, wmax=16):
    if vk == K.KIND_DICT:
        return synth.dict_page(rng.integers(0, dictionary.size, size=n), dict_len=dictionary.size)
    if vk == K.KIND_PLAIN:
        return rng.integers(0, 256, size=n * np.dtype(dt).itemsize, dtype=np.uint8)
    if vk == K.KIND_BOOL_PLAIN:
        return rng.integers(0, 256, size=(n + 7) // 8, dtype=np.uint8)
    if vk == K.KIND_BOOL_RLE:
        runs = np.repeat(rng.integers(0, 2, size=n // 20 + 1), rng.integers(1, 40, size=n // 20 + 1))[:n]
        bits = np.concatenate([runs, rng.integers(0, 2, size=n - runs.size)])
        return synth.length_prefixed(synth.hybrid_encode(bits, 1))
    bits = 32 if vk == K.KIND_DELTA32 else 64
    with np.errstate(over="ignore"):
        v = np.cumsum(rng.integers(-9, 1 << wmax, size=n)).astype(np.int32 if bits == 32 else np.int64)
    return synth.delta_encode(v, bits)


def _check(c):
    """The page's result, values and levels against the oracle's page reader."""
    want = O.page_read(c.src[: c.result_src_len], 1 if c.kind == K.KIND_PAGE_V1 else 2, c.n, 0, c.max_def,
                       ORACLE_KIND[c.value_kind], dictionary=c.dictionary, rep_len=c.rep_len, def_len=c.def_len,
                       value_dtype=c.plain_dtype)
    r = c.result
    got = (int(r["status"]), int(r["stage"]), int(r["detail"]), int(r["n_levels"]))
    assert got == (want["status"], want["stage"], want["detail"], want["n"]), (c.tag, r, want["status"], want["stage"])
    if want["status"] == O.OK or want["status"] == K.PAGE_VALUES_FAILED:
        assert int(r["not_null"]) == want["not_null"], c.tag
        assert int(r["n_decoded"]) == want["values_decoded"], (c.tag, r, want["values_decoded"])
    nd = want["values_decoded"] if want["stage"] in (K.STAGE_NONE, K.STAGE_VALUES) else 0
    if not (want["status"] == K.PAGE_VALUES_FAILED and want["detail"] == K.DICT_INVALID_INDEX):
        assert np.array_equal(c.values[:nd].view(np.uint8), want["values"][:nd].view(np.uint8)), c.tag
    assert guard_ok(c.guard) and guard_ok(c.def_guard) and guard_ok(c.rep_guard), c.tag
    if c.flags & K.FLAG_NO_LEVEL_OUTPUT:
        assert guard_ok(c.def_levels) and guard_ok(c.rep_levels), c.tag
    elif want["status"] == O.OK and c.n:
        assert not c.rep_levels.any() and not c.def_levels.any(), c.tag


def _page_case(tag, version, src, n, vk, rl=0, dl=0, flags=0, cut=None, **kw):
    c = Case(K.KIND_PAGE_V1 if version == 1 else K.KIND_PAGE_V2, src if cut is None else src[:cut].copy(), n, max_rep=0,
             max_def=0, value_kind=vk, rep_len=rl, def_len=dl, flags=flags, **kw)
    c.result_src_len = c.src.size
    c.tag = tag
    return c


def _cases(rng, n, cuts_per_page):
    d32 = rng.integers(-2**31, 2**31, size=4096, dtype=np.int64).astype(np.int32)
    d64 = rng.integers(-2**62, 2**62, size=700, dtype=np.int64)
    cases = []
    kinds = ((K.KIND_DELTA64, {}, {}), (K.KIND_DELTA32, {}, {}), (K.KIND_DICT, {"dictionary": d32}, {"dictionary": d32}),
             (K.KIND_DICT, {"dictionary": d64}, {"dictionary": d64}), (K.KIND_PLAIN, {"dt": np.int64}, {"plain_dtype": np.int64}),
             (K.KIND_PLAIN, {"dt": np.float32}, {"plain_dtype": np.float32}), (K.KIND_BOOL_PLAIN, {}, {}),
             (K.KIND_BOOL_RLE, {}, {}))
    for version in (1, 2):
        for vk, mk, kw in kinds:
            vs = _values_stream(rng, vk, n, **mk)
            # V2 pages of a flat column: the level bytes the header announces are skipped, whatever they are
            rl, dl = ((0, 0), (0, 7))[int(rng.integers(0, 2))] if version == 2 else (0, 0)
            src = np.concatenate([rng.integers(0, 256, size=rl + dl, dtype=np.uint8), vs])
            for flags in (K.FLAG_NO_LEVEL_OUTPUT, 0):
                tag = "v%d kind %d flags %d" % (version, vk, flags)
                cases.append(_page_case(tag, version, src, n, vk, rl, dl, flags, **kw))
                cases.append(_page_case(tag + " pull more", version, src, n + 5, vk, rl, dl, flags, **kw))
            for cut in sorted(set(rng.integers(0, src.size, size=cuts_per_page).tolist()) | {0, 1, 3, rl + dl, src.size - 1}):
                cases.append(_page_case("v%d kind %d cut %d of %d" % (version, vk, cut, src.size), version, src, n, vk, rl, dl,
                                        K.FLAG_NO_LEVEL_OUTPUT, cut=cut, **kw))
    return cases


@pytest.mark.parametrize("seg_bytes", [2048, 8192])
def test_flat_pages_in_segments_vs_oracle(ctx, seg_bytes):
    """Every value encoding, V1 and V2, with and without level output, whole and truncated anywhere — inside every
    segment of the pages that are cut — under two segment lengths and with the cut switched off."""
    rng = np.random.default_rng(4000 + seg_bytes)
    cases = _cases(rng, 30000, 40)
    try:
        for max_pages in (1 << 30, 0):
            ctx.set_option(K.OPT_SEGMENT_MAX_PAGES, max_pages)
            ctx.set_option(K.OPT_SEGMENT_BYTES, seg_bytes)
            run_cases(ctx, cases, seed=31)
            for c in cases:
                _check(c)
    finally:
        ctx.set_option(K.OPT_SEGMENT_MAX_PAGES, 1024)
        ctx.set_option(K.OPT_SEGMENT_BYTES, 0)


def test_damaged_flat_pages(ctx):
    """Errors of the value decoders' Init and DecodeValues as a page reports them: a bit width above 32, a key outside
    the dictionary, a delta header that does not parse, a miniblock width above 32 in a late segment."""
    rng = np.random.default_rng(4100)
    d32 = rng.integers(-2**31, 2**31, size=1000, dtype=np.int64).astype(np.int32)
    cases = []
    n = 40000
    for version in (1, 2):
        page = synth.dict_page(rng.integers(0, 1000, size=n), dict_len=1000)
        bad = page.copy()
        bad[0] = 33
        cases.append(_page_case("dict width 33", version, bad, n, K.KIND_DICT, dictionary=d32))
        cases.append(_page_case("dict key outside", version, page, n, K.KIND_DICT, dictionary=d32[:900]))
        for vk, bits in ((K.KIND_DELTA32, 32), (K.KIND_DELTA64, 64)):
            s = _values_stream(rng, vk, n)
            for at, val in ((0, 0x7f), (1, 3), (s.size // 2, None), (s.size - 300, None)):
                t = s.copy()
                if val is None:
                    t[at: at + 400] = 0xC7
                else:
                    t[at] = val
                cases.append(_page_case("delta%d damaged at %d" % (bits, at), version, t, n, vk))
        cases.append(_page_case("bool rle, no length", version, np.zeros(2, np.uint8), 10, K.KIND_BOOL_RLE))
    try:
        for max_pages, sb in ((1 << 30, 2048), (0, 0)):
            ctx.set_option(K.OPT_SEGMENT_MAX_PAGES, max_pages)
            ctx.set_option(K.OPT_SEGMENT_BYTES, sb)
            run_cases(ctx, cases, seed=32)
            for c in cases:
                _check(c)
    finally:
        ctx.set_option(K.OPT_SEGMENT_MAX_PAGES, 1024)
        ctx.set_option(K.OPT_SEGMENT_BYTES, 0)


def test_few_large_pages_of_walked_chunks(ctx):
    """A row group the way parquet-mr lays it out — a few columns, a few pages of 131 072 values each — from the chunk
    bytes on the device through pq_walk_chunks_dev, pq_chunks_page_descs and one plan: every page is a PQ_KIND_PAGE_V1 /
    _V2 descriptor, the delta pages are decoded in segments (pq_plan_segments), every value matches the oracle."""
    import torch
    rng = np.random.default_rng(4200)
    n, pages_per_chunk = 131072, 3
    d64 = rng.integers(-2**62, 2**62, size=3000, dtype=np.int64)
    H = synth.page_header
    chunks, truth = [], []
    for col, (ptype, vk, enc) in enumerate(((2, K.KIND_DELTA64, 5), (1, K.KIND_DELTA32, 5), (2, K.KIND_DICT, 8),
                                              (5, K.KIND_PLAIN, 0), (2, K.KIND_DELTA64, 5))):
        version = 1 + col % 2
        pages, bodies = [], []
        if vk == K.KIND_DICT:
            body = d64.view(np.uint8)
            pages.append((H(2, body.size, body.size, dictionary=(d64.size, 0)), body))
        for _ in range(pages_per_chunk):
            body = _values_stream(rng, vk, n, dictionary=d64, dt=np.float64)
            bodies.append(body)
            if version == 1:
                pages.append((H(0, body.size, body.size, data=(n, enc, 3, 3)), body))
            else:
                pages.append((H(3, body.size, body.size, v2=(n, 0, n, enc, 0, 0)), body))
        chunks.append(W.chunk(pages, physical_type=ptype))
        truth.append((vk, version, bodies))
    cap = 8
    descs_c = np.zeros(len(chunks), dtype=K.CHUNK_DTYPE)
    parts, pos = [], 0
    for i, c in enumerate(chunks):
        d = descs_c[i]
        d["src_off"], d["src_len"] = pos, c["src"].size
        parts.append(c["src"])
        pos += c["src"].size
        d["total_compressed_size"], d["data_page_offset"] = c["total"], c["data_page_offset"]
        d["physical_type"], d["type_length"] = c["physical_type"], -1
        d["info_first"], d["info_cap"] = i * cap, cap
    data = np.concatenate(parts)
    dev = torch.device("cuda:0")
    t_src = torch.from_numpy(np.concatenate([data, np.zeros(16, np.uint8)])).to(dev)
    t_infos = torch.zeros(len(chunks) * cap * K.PAGE_INFO_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    res = ctx.walk_chunks_dev(t_src.data_ptr(), data.size, descs_c, t_infos.data_ptr(), len(chunks) * cap)
    assert (res["status"] == K.OK).all(), res
    infos = t_infos.cpu().numpy().view(K.PAGE_INFO_DTYPE)
    layout = np.zeros(1, dtype=K.CHUNK_LAYOUT_DTYPE)
    descs, first, copies = capi.chunks_page_descs(descs_c, res, infos, layout)
    assert descs.size == len(chunks) * pages_per_chunk
    assert set(descs["kind"].tolist()) == {K.KIND_PAGE_V1, K.KIND_PAGE_V2}
    vb, db = int(layout["values_off"][0]) + 64, int(layout["dict_off"][0]) + 64
    t_val = torch.full((vb,), 0xA5, dtype=torch.uint8, device=dev)
    t_dict = torch.zeros(db, dtype=torch.uint8, device=dev)
    for c in copies:
        t_dict[int(c["dst"]): int(c["dst"]) + int(c["bytes"])] = t_src[int(c["src"]): int(c["src"]) + int(c["bytes"])]
    torch.cuda.synchronize()
    plan = ctx.plan(descs)
    plan.run(t_src.data_ptr(), data.size, t_val.data_ptr(), vb, 0, 0, t_dict.data_ptr(), db)
    results = plan.results()
    segs = plan.segments()
    plan.close()
    values = t_val.cpu().numpy()
    assert (results["status"] == K.OK).all(), results[results["status"] != 0][:3]
    assert (results["n_levels"] == n).all() and (results["not_null"] == n).all() and (results["n_decoded"] == n).all()
    # 6 + 3 delta pages of a few hundred KB each: cut, and all but a few segments decoded by waves of their own
    assert segs.size >= 9 * 20 and int((segs["decode"] != 0).sum()) >= segs.size - 2 * 9, (segs.size, segs["decode"])
    for i, (vk, version, bodies) in enumerate(truth):
        for k, body in enumerate(bodies):
            d = descs[int(first[i]) + k]
            want = O.page_read(body, version, n, 0, 0, ORACLE_KIND[vk], dictionary=d64 if vk == K.KIND_DICT else None,
                               value_dtype=np.float64 if vk == K.KIND_PLAIN else None)
            assert want["status"] == O.OK and want["not_null"] == n
            es = want["values"].dtype.itemsize
            got = values[int(d["dst_off"]): int(d["dst_off"]) + es * n]
            assert np.array_equal(got, want["values"][:n].view(np.uint8)), (i, k)
