"""Re-entrancy of the C ABI: the reference is single-threaded, but a Go program calls from any OS thread and from
many goroutines at once (callers: DecodeValues, types/types.go:22-27; decodePackedArray, layout/helpers.go:68;
readPageData, file-reader.go:268-302).  The contract is one pq_ctx per calling thread.  Here four host threads,
each with a context of its own on GPU 0, call pq_decode_pages, pq_decode_chunks, pq_column_entries and
pq_walk_chunks at the same time on different inputs, a fifth creates and destroys contexts in a loop, and every
output is checked against the oracle.  (ctypes releases the interpreter lock for the duration of each call.)"""
import threading

import numpy as np
import pytest

import oracle_lib as O
import walk_cases as W
from parquet_go_amd import capi, synth

pytestmark = pytest.mark.gpu
K = capi


def _pages_workload(rng, n_pages):
    """Delta, dictionary and nested V1 pages in one descriptor table + the oracle's answers."""
    d = rng.integers(-2**31, 2**31, size=500, dtype=np.int64).astype(np.int32)
    descs = np.zeros(n_pages, dtype=K.DESC_DTYPE)
    parts, want = [], []
    spos = vpos = lpos = 0
    for p in range(n_pages):
        n = int(rng.integers(100, 30000))
        if n % 128 == 1:
            n += 1
        q = descs[p]
        kind = p % 3
        if kind == 0:
            v = np.cumsum(rng.integers(-9, 1 << 15, size=n)).astype(np.int64)
            s = synth.delta_encode(v, 64)
            q["kind"] = K.KIND_DELTA64
            es = 8
            want.append(("values", O.delta_decode64(s, n)["values"].view(np.uint8)))
        elif kind == 1:
            keys = rng.integers(0, 500, size=n)
            s = synth.dict_page(keys, w=9)
            q["kind"], q["elem_size"], q["dict_len"] = K.KIND_DICT, 4, 500
            es = 4
            want.append(("values", O.dict_decode(s, d, n)[1][:n].view(np.uint8)))
        else:
            rep = rng.integers(0, 2, size=n)
            deff = rng.integers(0, 3, size=n)
            keys = rng.integers(0, 500, size=int((deff == 2).sum()))
            s = np.concatenate([synth.length_prefixed(synth.hybrid_encode(rep, 1)),
                                synth.length_prefixed(synth.hybrid_encode(deff, 2)), synth.dict_page(keys, w=9)]).astype(np.uint8)
            q["kind"], q["value_kind"], q["elem_size"], q["dict_len"] = K.KIND_PAGE_V1, K.KIND_DICT, 4, 500
            q["max_rep"], q["max_def"] = 1, 2
            q["rep_off"], q["def_off"] = lpos, lpos + 4 * n
            lpos += 8 * n
            es = 4
            pr = O.page_read(s, 1, n, 1, 2, O.VALUES_DICT, dictionary=d)
            assert pr["status"] == O.OK
            want.append(("page", (pr["values"][: pr["not_null"]].view(np.uint8), pr["rep"], pr["def_"], pr["not_null"])))
        q["src_off"], q["src_len"], q["n_values"] = spos, s.size, n
        vpos = (vpos + 7) // 8 * 8
        q["dst_off"] = vpos
        vpos += es * n
        parts.append(s)
        spos += s.size
    return dict(descs=descs, src=np.concatenate(parts), dict=d.view(np.uint8).copy(), values_bytes=vpos + 8,
                levels_bytes=lpos + 4, want=want)


def _check_pages(wl, results, values, levels):
    levels = levels.view(np.int32)
    for p, (tag, w) in enumerate(wl["want"]):
        q, r = wl["descs"][p], results[p]
        assert int(r["status"]) == K.OK, (p, r)
        off = int(q["dst_off"])
        if tag == "values":
            assert np.array_equal(values[off: off + w.size], w), p
        else:
            vals, rep, deff, nn = w
            n = int(q["n_values"])
            assert int(r["not_null"]) == nn and int(r["n_levels"]) == n
            assert np.array_equal(values[off: off + vals.size], vals), p
            assert np.array_equal(levels[int(q["rep_off"]) // 4: int(q["rep_off"]) // 4 + n], rep)
            assert np.array_equal(levels[int(q["def_off"]) // 4: int(q["def_off"]) // 4 + n], deff)


def _walk_workload(rng):
    chunks = [c for _, c in W.good_and_bad_chunks(rng)]
    cap = 400
    descs = np.zeros(len(chunks), dtype=K.CHUNK_DTYPE)
    parts, pos = [], 0
    for i, c in enumerate(chunks):
        d = descs[i]
        d["src_off"], d["src_len"] = pos, c["src"].size
        parts.append(c["src"])
        pos += c["src"].size
        d["total_compressed_size"], d["data_page_offset"] = c["total"], c["data_page_offset"]
        d["has_dictionary_page_offset"] = 0 if c["dictionary_page_offset"] is None else 1
        d["dictionary_page_offset"] = c["dictionary_page_offset"] or 0
        d["physical_type"], d["type_length"] = c["physical_type"], c["type_length"]
        d["max_rep"], d["max_def"], d["codec"] = c["max_rep"], c["max_def"], c["codec"]
        d["info_first"], d["info_cap"] = i * cap, cap
    want = [O.walk_chunk(c["src"], c["total"], c["data_page_offset"], c["dictionary_page_offset"], c["physical_type"],
                         c["type_length"], c["max_rep"], c["max_def"], c["codec"], cap=cap) for c in chunks]
    return dict(src=np.concatenate(parts + [np.zeros(16, np.uint8)]), descs=descs, cap=cap, want=want)


def _worker(tid, rounds, errors):
    try:
        rng = np.random.default_rng(9000 + tid)
        ctx = capi.Context(0)
        try:
            pages = [_pages_workload(rng, 12 + 3 * tid) for _ in range(2)]
            walk = _walk_workload(rng)
            n = 200_000 + 1000 * tid
            deff = rng.integers(0, 4, size=n).astype(np.int32)
            rep = (rng.random(n) < 0.7).astype(np.int32)
            rep[0] = 0
            want_entries = O.column_entries(deff, rep, 3, 1, True)
            for it in range(rounds):
                wl = pages[it % 2]
                # pq_decode_pages
                res, values, levels = ctx.decode_pages(wl["descs"], wl["src"], wl["values_bytes"], wl["levels_bytes"], wl["dict"])
                _check_pages(wl, res, values, levels)
                # pq_decode_chunks: the same tables as five jobs through the pipeline slots
                jobs = []
                for k in range(5):
                    w2 = pages[(it + k) % 2]
                    jobs.append(dict(descs=w2["descs"], src=w2["src"], dict=w2["dict"],
                                     values=np.full(w2["values_bytes"], 0xA5, np.uint8),
                                     levels=np.full(w2["levels_bytes"], 0xA5, np.uint8), wl=w2))
                out = ctx.decode_chunks(jobs)
                for j, (st, results) in zip(jobs, out):
                    assert st == K.OK
                    _check_pages(j["wl"], results, j["values"], j["levels"])
                # pq_column_entries
                got = ctx.column_entries(deff, rep, 3, 1, True)
                assert (got["n_entries"], got["n_values"]) == (want_entries["n_entries"], want_entries["n_values"])
                assert np.array_equal(got["offsets"], want_entries["offsets"])
                assert np.array_equal(got["entry_def"], want_entries["entry_def"])
                assert got["validity"].tobytes() == want_entries["validity"].tobytes()
                # pq_walk_chunks
                res, infos = ctx.walk_chunks(walk["src"], walk["descs"], len(walk["want"]) * walk["cap"])
                for i, w in enumerate(walk["want"]):
                    assert int(res[i]["status"]) == w["status"] and int(res[i]["n_pages"]) == len(w["infos"])
                    assert infos[i * walk["cap"]: i * walk["cap"] + len(w["infos"])].tobytes() == w["infos"].tobytes()
        finally:
            ctx.close()
    except BaseException as e:  # noqa: BLE001 - handed to the main thread
        errors.append((tid, repr(e)))


def _churn(stop, errors):
    """Contexts created and destroyed while the others work; each does one small decode in between."""
    try:
        rng = np.random.default_rng(77)
        v = np.cumsum(rng.integers(0, 1000, size=3000)).astype(np.int64)
        s = synth.delta_encode(v, 64)
        count = 0
        while not stop.is_set() or count < 3:
            ctx = capi.Context(0)
            r = ctx.delta_decode(s, v.size, 64)
            assert r["status"] == K.OK and np.array_equal(r["values"], v)
            ctx.close()
            count += 1
    except BaseException as e:  # noqa: BLE001
        errors.append(("churn", repr(e)))


def test_four_threads_four_contexts_and_one_that_churns():
    capi.Context(0).close()     # the first context of a process sets the device up
    errors, stop = [], threading.Event()
    workers = [threading.Thread(target=_worker, args=(t, 3, errors)) for t in range(4)]
    churn = threading.Thread(target=_churn, args=(stop, errors))
    for t in workers:
        t.start()
    churn.start()
    for t in workers:
        t.join(timeout=280)
    stop.set()
    churn.join(timeout=60)
    assert not any(t.is_alive() for t in workers) and not churn.is_alive(), "a thread did not finish"
    assert not errors, errors


def test_groups_of_a_large_plan_side_by_side_and_one_after_the_other():
    """A plan whose delta pages, dictionary pages and nested data pages each have 16 MiB and more to move runs those
    groups on streams of the library's own, forked from the context's stream and joined to it again
    (PQ_OPT_CONCURRENT_GROUPS, on by default); the same plan with the option off runs them one after the other.
    Every page of both against the oracle, three runs each way (a second run must not see the first one's state)."""
    rng = np.random.default_rng(31337)
    wl = _pages_workload(rng, 1500)
    ctx = capi.Context(0)
    try:
        for on in (1, 0, 1):
            ctx.set_option(K.OPT_CONCURRENT_GROUPS, on)
            for _ in range(3 if on else 1):
                res, values, levels = ctx.decode_pages(wl["descs"], wl["src"], wl["values_bytes"], wl["levels_bytes"], wl["dict"])
                _check_pages(wl, res, values, levels)
    finally:
        ctx.close()

