"""The multi-row-group configuration on one GPU: 64 column chunks (4 row groups x 16 columns; even
columns INT64 DELTA_BINARY_PACKED, odd columns INT32 RLE_DICTIONARY at width 12; 800 pages x 20 000
values per chunk, 1.02 G values), the chunks of rank 0 of a one-rank plan (parquet_go_amd.shard)
handed to pq_decode_chunks in one call, from host memory and back, every value checked.  Then the
edges of that entry point on a small workload: pinned buffers, a job with a bad descriptor, pages
that fail, levels and dictionaries, more jobs than slots, none at all."""
import os
import sys

import numpy as np
import pytest

import oracle_lib as O
from parquet_go_amd import capi, shard, synth

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import bench_chunks as BC  # noqa: E402

pytestmark = pytest.mark.gpu
K = capi


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def test_64_chunks_from_host_memory_every_value(ctx):
    wl = BC.build_workload(row_groups=4, columns=16, pages=800, n=20000, distinct=50)
    assert len(wl["infos"]) == 64
    plan = shard.plan_shards(wl["infos"], 1)
    mine = [p.chunk for p in plan[0]]
    assert sorted(mine) == list(range(64)) and all(p.page_lo == 0 and p.page_hi == 800 for p in plan[0])
    jobs = BC.chunk_jobs(wl, mine, lambda nbytes: np.full(nbytes, 0xA5, dtype=np.uint8))
    outcome = ctx.decode_chunks(jobs)
    BC.check_jobs(wl, jobs, outcome, every_value=True)
    consumed = sum(int(res["consumed"].sum()) for _, res in outcome)
    assert consumed == sum(i.encoded_bytes for i in wl["infos"])


def _small_jobs(rng, n_jobs, alloc, vary_dict=False, max_n=5000):
    """Chunks of a few pages each: delta64, dictionary, and V1 pages with levels; -> (jobs, checks)
    vary_dict: every chunk has a dictionary of its own."""
    jobs, checks = [], []
    d0 = rng.integers(-2**31, 2**31 - 1000, size=300, dtype=np.int64).astype(np.int32)
    for j in range(n_jobs):
        d = (d0 + j).astype(np.int32) if vary_dict else d0
        kind = j % 3
        streams, want, descs = [], [], []
        n_pages = 1 + j % 5
        for p in range(n_pages):
            n = int(rng.integers(1, max_n))
            if n % 128 == 1:
                n += 1      # the reference's own decoder fails on the last value of such a stream
            if kind == 0:
                v = np.cumsum(rng.integers(-9, 1 << 14, size=n)).astype(np.int64)
                streams.append(synth.delta_encode(v, 64))
                want.append(("delta", v))
            elif kind == 1:
                keys = rng.integers(0, 300, size=n)
                streams.append(synth.dict_page(keys, w=9))
                want.append(("dict", d[keys]))
            else:
                rep = rng.integers(0, 2, size=n)
                deff = rng.integers(0, 3, size=n)
                keys = rng.integers(0, 300, size=int((deff == 2).sum()))
                page = np.concatenate([synth.length_prefixed(synth.hybrid_encode(rep, 1)),
                                       synth.length_prefixed(synth.hybrid_encode(deff, 2)),
                                       synth.dict_page(keys, w=9)]).astype(np.uint8)
                streams.append(page)
                want.append(("page", (rep, deff, d[keys])))
        lens = [s.size for s in streams]
        offs = np.concatenate([[0], np.cumsum(lens)])[:-1]
        descs = np.zeros(n_pages, dtype=K.DESC_DTYPE)
        vpos = lpos = 0
        for p in range(n_pages):
            q = descs[p]
            tag, w = want[p]
            q["src_off"], q["src_len"] = offs[p], lens[p]
            if tag == "delta":
                q["kind"], q["n_values"] = K.KIND_DELTA64, w.size
                q["dst_off"] = vpos
                vpos += 8 * w.size
            elif tag == "dict":
                q["kind"], q["n_values"], q["elem_size"], q["dict_len"] = K.KIND_DICT, w.size, 4, 300
                q["dst_off"] = vpos
                vpos += (4 * w.size + 7) // 8 * 8
            else:
                rep, deff, vals = w
                q["kind"], q["n_values"], q["max_rep"], q["max_def"] = K.KIND_PAGE_V1, rep.size, 1, 2
                q["value_kind"], q["elem_size"], q["dict_len"] = K.KIND_DICT, 4, 300
                q["dst_off"] = vpos
                vpos += (4 * rep.size + 7) // 8 * 8
                q["rep_off"], q["def_off"] = lpos, lpos + 4 * rep.size
                lpos += 8 * rep.size
        src = alloc(int(sum(lens)))
        src[:] = np.concatenate(streams)
        job = dict(descs=descs, src=src, values=alloc(max(vpos, 8)))
        if lpos:
            job["levels"] = alloc(lpos)
        if kind != 0:
            job["dict"] = alloc(d.nbytes)
            job["dict"][:] = d.view(np.uint8)
        jobs.append(job)
        checks.append(want)
    return jobs, checks


def _check_small(jobs, checks, outcome):
    for job, want, (status, res) in zip(jobs, checks, outcome):
        assert status == K.OK
        for p, (tag, w) in enumerate(want):
            q, r = job["descs"][p], res[p]
            assert r["status"] == K.OK, (tag, r)
            if tag == "page":
                rep, deff, vals = w
                lv = job["levels"]
                assert np.array_equal(lv[int(q["rep_off"]): int(q["rep_off"]) + 4 * rep.size].view(np.int32), rep)
                assert np.array_equal(lv[int(q["def_off"]): int(q["def_off"]) + 4 * rep.size].view(np.int32), deff)
                assert r["not_null"] == vals.size and r["n_levels"] == rep.size
                got = job["values"][int(q["dst_off"]): int(q["dst_off"]) + 4 * vals.size].view(np.int32)
                assert np.array_equal(got, vals)
            else:
                got = job["values"][int(q["dst_off"]): int(q["dst_off"]) + w.nbytes].view(w.dtype)
                assert r["n_decoded"] == w.size and np.array_equal(got, w)


@pytest.mark.parametrize("pinned", [False, True])
def test_chunk_jobs_of_every_kind(ctx, pinned):
    rng = np.random.default_rng(41)
    held = []

    def alloc(nbytes):
        if pinned:
            a = ctx.host_alloc(nbytes)
            held.append(a)
            return a
        return np.empty(nbytes, dtype=np.uint8)
    for n_jobs in (0, 1, 2, 3, 4, 11):
        jobs, checks = _small_jobs(rng, n_jobs, alloc)
        for _ in range(2):                      # the second call reuses slots, staging and plans' blocks
            for j in jobs:
                j["values"][:] = 0x5A
            outcome = ctx.decode_chunks(jobs, pinned=pinned)
            assert len(outcome) == n_jobs
            _check_small(jobs, checks, outcome)
    for a in held:
        ctx.host_free(a)


def test_a_refused_job_and_failing_pages_do_not_stop_the_others(ctx):
    rng = np.random.default_rng(43)
    jobs, checks = _small_jobs(rng, 6, lambda nbytes: np.empty(nbytes, dtype=np.uint8))
    bad = jobs[2]
    bad["descs"] = bad["descs"].copy()
    bad["descs"]["kind"][0] = 77                               # no such kind: the job is refused
    cut = jobs[3]                                              # a delta chunk: its first page loses its tail
    assert cut["descs"]["kind"][0] == K.KIND_DELTA64
    cut["descs"] = cut["descs"].copy()
    cut["descs"]["src_len"][0] = max(int(cut["descs"]["src_len"][0]) // 2, 1)
    outcome = ctx.decode_chunks(jobs)
    assert outcome[2][0] == K.ERR_BAD_ARGUMENT
    keep = [i for i in range(6) if i not in (2, 3)]
    _check_small([jobs[i] for i in keep], [checks[i] for i in keep], [outcome[i] for i in keep])
    status, res = outcome[3]
    assert status == K.OK
    q = cut["descs"][0]
    want = O.delta_decode64(cut["src"][int(q["src_off"]): int(q["src_off"]) + int(q["src_len"])], int(q["n_values"]))
    assert res[0]["status"] == want["status"] != K.OK and res[0]["n_decoded"] == want["n_decoded"]
    got = cut["values"][int(q["dst_off"]): int(q["dst_off"]) + 8 * want["n_decoded"]].view(np.int64)
    assert np.array_equal(got, want["values"])


def test_one_plan_over_many_chunks_in_buffers_of_their_own(ctx):
    """pq_plan_create_sets / pq_plan_run_sets: the chunks of a row group resident in separate device allocations,
    one plan, one set of launches — same values, levels and results as a plan per chunk, also when the buffers move
    between two runs of the plan, with a dictionary per chunk, large pages that are cut into segments and an
    empty set in the middle."""
    import torch
    rng = np.random.default_rng(47)
    dev = torch.device("cuda:0")
    jobs, checks = _small_jobs(rng, 13, lambda nbytes: np.empty(nbytes, dtype=np.uint8), vary_dict=True, max_n=90000)
    empty = dict(descs=np.zeros(0, dtype=K.DESC_DTYPE), src=np.zeros(0, np.uint8), values=np.zeros(0, np.uint8))
    jobs.insert(5, empty)
    checks.insert(5, [])
    descs = np.concatenate([j["descs"] for j in jobs])
    first = np.concatenate([[0], np.cumsum([j["descs"].size for j in jobs])]).astype(np.uint32)
    plan = ctx.plan(descs, first_desc=first)
    with pytest.raises(capi.PqError):       # a plan over several sets does not run on one
        plan.run(0, 0)
    for attempt in range(2):                # fresh allocations the second time: the buffers have moved
        pad = torch.empty(int(rng.integers(1, 1 << 20)), dtype=torch.uint8, device=dev)
        held, sets = [pad], []
        for j in jobs:
            t = {}
            for name in ("src", "values", "levels", "dict"):
                a = j.get(name)
                if a is not None and a.size:
                    src = np.concatenate([a, np.zeros(16, np.uint8)]) if name in ("src", "dict") else np.full(a.size, 0xA5, np.uint8)
                    t[name] = torch.from_numpy(src).to(dev)
            held.append(t)
            g = lambda name, j=j, t=t: (t[name].data_ptr(), j[name].size) if name in t else (0, 0)
            sets.append(g("src") + g("values") + g("levels") + g("dict"))
        torch.cuda.synchronize()
        plan.run_sets(sets)
        res = plan.results()
        outcome = []
        for k, j in enumerate(jobs):
            for name in ("values", "levels"):
                if name in held[1 + k]:
                    j[name] = held[1 + k][name].cpu().numpy()
            outcome.append((K.OK, res[int(first[k]): int(first[k + 1])]))
        _check_small(jobs, checks, outcome)
    with pytest.raises(capi.PqError):       # a set too small for what its descriptors reach
        short = list(sets)
        short[0] = (short[0][0], 1) + short[0][2:]
        plan.run_sets(short)
    with pytest.raises(capi.PqError):
        plan.run_sets(sets[:-1])
    plan.close()
