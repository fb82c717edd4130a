"""GPU parity of large data pages of nested columns (max_rep > 0) that a plan runs as three units of work — the
repetition levels, the definition levels and the values, the level streams cut into segments — against the oracle's
page reader (dataPageReaderV1/V2.read + ReadValues, layout/page-v1.go:40-116, page-v2.go:42-149): status, stage,
detail, counts, levels and values, for every value encoding, whole and damaged at every stage."""
import numpy as np
import pytest

import oracle_lib as O
from gpu_harness import Case, run_cases
from parquet_go_amd import capi, synth
from test_gpu_parity import _check_page, _make_page

pytestmark = pytest.mark.gpu
K = capi


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def _say(what):
    import os
    print(what, flush=True)
    if os.path.isdir("gpurun_out"):
        with open("gpurun_out/nested_progress.log", "a") as f:
            f.write(what + "\n")


def _cases(rng, n):
    d32 = rng.integers(-2**31, 2**31, size=4096, dtype=np.int64).astype(np.int32)
    d64 = rng.integers(-2**62, 2**62, size=500, dtype=np.int64)
    cases = []
    kinds = ((K.KIND_DICT, d32, None), (K.KIND_DICT, d64, None), (K.KIND_DELTA32, None, None), (K.KIND_DELTA64, None, None),
             (K.KIND_PLAIN, None, np.int64), (K.KIND_PLAIN, None, np.float32), (K.KIND_BOOL_PLAIN, None, None),
             (K.KIND_BOOL_RLE, None, None))
    for version in (1, 2):
        kind = K.KIND_PAGE_V1 if version == 1 else K.KIND_PAGE_V2
        for k, (vk, dic, dt) in enumerate(kinds):
            max_rep, max_def = ((1, 3), (2, 2), (1, 1), (3, 7))[k % 4]
            src, rl, dl = _make_page(rng, version, n, max_rep, max_def, vk, dic, plain_dtype=dt)
            kw = dict(max_rep=max_rep, max_def=max_def, value_kind=vk, dictionary=dic, rep_len=rl, def_len=dl)
            if dt is not None:
                kw["plain_dtype"] = dt
            cases.append(Case(kind, src, n, **kw))
            if k % 2 == 0:      # damaged pages: every stage's error path, and cuts inside each section
                if version == 1:
                    rep_end = 4 + int(np.frombuffer(src[:4].tobytes(), np.uint32)[0])
                    def_end = rep_end + 4 + int(np.frombuffer(src[rep_end: rep_end + 4].tobytes(), np.uint32)[0])
                else:
                    rep_end, def_end = rl, rl + dl
                for cut in (0, 3, rep_end // 2, rep_end - 1, rep_end + 2, (rep_end + def_end) // 2, def_end - 1, def_end,
                            def_end + 1, (def_end + src.size) // 2, src.size - 1):
                    cases.append(Case(kind, src[:cut].copy(), n, **kw))
                bad = src.copy()            # a run header in the middle of the definition levels replaced
                bad[(rep_end + def_end) // 2: (rep_end + def_end) // 2 + 2] = 0
                cases.append(Case(kind, bad, n, **kw))
                if version == 2:            # zero-length level sections leave their decoders uninitialised
                    cases.append(Case(kind, src[rl:].copy(), n, **dict(kw, rep_len=0)))
                    cases.append(Case(kind, np.concatenate([src[:rl], src[rl + dl:]]), n, **dict(kw, def_len=0)))
    # a width-0 definition level stream (max_def == 0 under a repeated column) and pages below the size that is split
    src, rl, dl = _make_page(rng, 1, n, 1, 0, K.KIND_DELTA64, None)
    cases.append(Case(K.KIND_PAGE_V1, src, n, max_rep=1, max_def=0, value_kind=K.KIND_DELTA64, rep_len=rl, def_len=dl))
    src, rl, dl = _make_page(rng, 2, 30000, 1, 3, K.KIND_DICT, d32)
    cases.append(Case(K.KIND_PAGE_V2, src, 30000, max_rep=1, max_def=3, value_kind=K.KIND_DICT, dictionary=d32,
                      rep_len=rl, def_len=dl))
    return cases


@pytest.mark.parametrize("seg_bytes", [0, 2048])
def test_large_nested_pages_as_three_units(ctx, seg_bytes):
    rng = np.random.default_rng(5100 + seg_bytes)
    cases = _cases(rng, 140_000)
    try:
        for max_pages in (0, 1 << 30):          # one wave per page, and split into units
            ctx.set_option(K.OPT_SEGMENT_MAX_PAGES, max_pages)
            ctx.set_option(K.OPT_SEGMENT_BYTES, seg_bytes)
            _say("nested pages: seg_bytes %d, max_pages %d, %d cases" % (seg_bytes, max_pages, len(cases)))
            run_cases(ctx, cases, seed=51)
            for c in cases:
                _check_page(c)
            _say("nested pages: the same plan again and a second plan")
            run_cases(ctx, cases[::3], seed=52)
            for c in cases[::3]:
                _check_page(c)
    finally:
        ctx.set_option(K.OPT_SEGMENT_MAX_PAGES, 1024)
        ctx.set_option(K.OPT_SEGMENT_BYTES, 0)


def test_one_million_slot_nested_pages(ctx):
    """The large-page shape of the nested configuration: 1 M level slots a page (rep w = 1, def w = 2, dictionary
    values), a few pages, so that they are split."""
    rng = np.random.default_rng(5200)
    d32 = rng.integers(-2**31, 2**31, size=4096, dtype=np.int64).astype(np.int32)
    cases = []
    n = 1_000_000
    for version in (1, 2):
        rep = (rng.random(n) >= 0.25).astype(np.int64)
        deff = np.where(rng.random(n) < 0.8, 3, rng.integers(0, 3, size=n))
        keys = rng.integers(0, 4096, size=int((deff == 3).sum()))
        rs, ds = synth.hybrid_encode(rep, 1), synth.hybrid_encode(deff, 2)
        vs = np.concatenate([[12], synth.hybrid_encode_bitpacked(keys, 12)]).astype(np.uint8)
        if version == 1:
            src = np.concatenate([synth.length_prefixed(rs), synth.length_prefixed(ds), vs]).astype(np.uint8)
            rl = dl = 0
        else:
            src, rl, dl = np.concatenate([rs, ds, vs]).astype(np.uint8), rs.size, ds.size
        for cut in (None, src.size - 5000, src.size // 3):
            cases.append(Case(K.KIND_PAGE_V1 if version == 1 else K.KIND_PAGE_V2, src if cut is None else src[:cut].copy(), n,
                              max_rep=1, max_def=3, value_kind=K.KIND_DICT, dictionary=d32, rep_len=rl, def_len=dl))
    _say("nested pages: 1 M slots a page")
    run_cases(ctx, cases, seed=53)
    for c in cases:
        _check_page(c)
