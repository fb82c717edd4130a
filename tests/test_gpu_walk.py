"""GPU parity for the page-header walk (ChunkReader.readPages, layout/chunk.go:129-217) through the C ABI
against the CPU oracle — same status, page count, Count() and page records — and, end to end, real
Parquet files written by pyarrow: chunk bytes on the device -> pq_walk_chunks_dev -> pq_chunk_page_descs
-> plan -> values and levels equal to what pyarrow reads from the same file."""
import numpy as np
import pytest

import oracle_lib as O
import parquet_files as F
import walk_cases as W
from parquet_go_amd import capi

pytestmark = pytest.mark.gpu

K = capi


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def _walk_all(ctx, chunks, cap=400):
    """All chunks in one call, packed into one source buffer at odd offsets."""
    rng = np.random.default_rng(1)
    descs = np.zeros(len(chunks), dtype=K.CHUNK_DTYPE)
    parts, pos = [], 0
    for i, c in enumerate(chunks):
        gap = int(rng.integers(0, 40))
        parts.append(rng.integers(0, 256, size=gap, dtype=np.uint8))
        pos += gap
        d = descs[i]
        d["src_off"], d["src_len"] = pos, c["src"].size
        parts.append(c["src"])
        pos += c["src"].size
        d["total_compressed_size"] = c["total"]
        d["data_page_offset"] = c["data_page_offset"]
        d["has_dictionary_page_offset"] = 0 if c["dictionary_page_offset"] is None else 1
        d["dictionary_page_offset"] = c["dictionary_page_offset"] or 0
        d["physical_type"], d["type_length"] = c["physical_type"], c["type_length"]
        d["max_rep"], d["max_def"], d["codec"] = c["max_rep"], c["max_def"], c["codec"]
        d["info_first"], d["info_cap"] = i * cap, c.get("cap", cap)
    src = np.concatenate(parts + [np.zeros(16, np.uint8)])
    res, infos = ctx.walk_chunks(src, descs, len(chunks) * cap)
    return res, [infos[i * cap: i * cap + int(res[i]["n_pages"])] for i in range(len(chunks))]


def _oracle(c, cap=400):
    return O.walk_chunk(c["src"], c["total"], c["data_page_offset"], c["dictionary_page_offset"], c["physical_type"],
                        c["type_length"], c["max_rep"], c["max_def"], c["codec"], cap=c.get("cap", cap))


def _compare(ctx, chunks, names=None):
    res, infos = _walk_all(ctx, chunks)
    for i, c in enumerate(chunks):
        want = _oracle(c)
        tag = names[i] if names else i
        assert int(res[i]["status"]) == want["status"], (tag, res[i], want["status"])
        assert int(res[i]["n_pages"]) == len(want["infos"]), tag
        assert int(res[i]["count"]) == want["count"], (tag, res[i], want["count"])
        assert infos[i].tobytes() == want["infos"].tobytes(), tag


def test_synthetic_chunks_vs_oracle(ctx):
    pairs = W.good_and_bad_chunks(np.random.default_rng(21))
    small = dict(pairs)["three hundred pages"]
    pairs.append(("more pages than records", dict(small, cap=299)))
    _compare(ctx, [c for _, c in pairs], [n for n, _ in pairs])
    seen = {_oracle(c)["status"] for _, c in pairs}
    assert set(range(19, 31)) | {0, O.OUT_OF_RANGE_CAP} <= seen


def test_truncated_at_every_byte(ctx):
    chunks = dict(W.good_and_bad_chunks(np.random.default_rng(22)))
    cases = []
    for name in ("v1 with statistics and crc", "v2 pages", "dictionary then data, with offsets", "fields nobody knows",
                 "nesting the skip still allows"):
        cases += list(W.truncations(chunks[name]))
    _compare(ctx, cases)


def test_fuzzed_headers(ctx):
    """Random bytes, and good chunks with a few bytes flipped: whatever the bytes say, same answer."""
    rng = np.random.default_rng(23)
    good = [c for _, c in W.good_and_bad_chunks(rng) if c["src"].size > 40]
    cases = []
    for i in range(1500):
        if i % 3 == 0:
            src = rng.integers(0, 256, size=int(rng.integers(0, 300)), dtype=np.uint8)
            if src.size > 4 and i % 2:
                src[0] = 0x15                      # field 1, i32: looks like a page header's start
            c = W.chunk([(bytes(src), 0)], physical_type=int(rng.integers(0, 9)), max_rep=int(rng.integers(0, 2)))
        else:
            base = good[int(rng.integers(0, len(good)))]
            src = base["src"].copy()
            head = min(src.size, 120)
            for _ in range(int(rng.integers(1, 4))):
                src[int(rng.integers(0, head))] = int(rng.integers(0, 256))
            c = dict(base, src=src)
        c["total"] = int(rng.integers(0, 2 * c["src"].size + 2))
        cases.append(c)
    _compare(ctx, cases)


def test_unknown_fields_of_every_shape(ctx):
    """thrift.Skip over random nested structs, lists, sets and maps, whole and cut short."""
    rng = np.random.default_rng(24)
    chunks = W.chunks_with_unknown_fields(rng, 400)
    cut = []
    for c in chunks[:150]:
        for _ in range(4):
            d = dict(c)
            d["src"] = c["src"][: int(rng.integers(0, c["src"].size))].copy()
            cut.append(d)
    _compare(ctx, chunks + cut)


def test_statistics_of_every_shape(ctx):
    """The shapes the walk's fast path takes in one go, and the ones just outside: binaries whose length
    needs one and two varint bytes and that end on, one short of and one past the 64 bytes a span of
    lanes covers; counts of one to five varint bytes (five leave the fast path); up to ten fields, then
    eleven; a long-form field id; headers longer than what a window keeps behind a header's first byte;
    page bodies so short that several headers share a window; every page kind."""
    from parquet_go_amd import synth
    rng = np.random.default_rng(25)
    H = synth.page_header
    blob = lambda n: bytes(rng.integers(0, 256, size=n, dtype=np.uint8))
    stats = []
    for n in (0, 1, 2, 50, 58, 59, 60, 61, 62, 63, 64, 65, 66, 126, 127, 128, 129, 200, 300, 470, 520, 700, 1100):
        stats.append([(1, "binary", blob(n)), (2, "binary", b"ab"), (3, "i64", 5), (4, "i64", 77), (5, "binary", blob(n // 2)),
                      (6, "binary", b""), (7, "bool", True), (8, "bool", False)])
        stats.append([(5, "binary", blob(n)), (6, "binary", blob(n))])
    for count in (1, 127, 128, 1 << 14, 1 << 21, 1 << 27, 1 << 28, 1 << 40, -1, -(1 << 30)):
        stats.append([(3, "i64", count), (4, "i64", -count)])
    stats.append([(k, "bool", bool(k & 1)) for k in range(1, 11)])               # ten fields
    stats.append([(k, "bool", bool(k & 1)) for k in range(1, 12)])               # eleven
    stats.append([(1, "binary", b"x"), (40, "i64", 3)])                          # a field id that needs the long form
    stats.append([(1, "binary", b"x"), (2, "i32", 3)])                           # a type Statistics does not hold
    stats.append([])
    chunks = []
    for body in (0, 7, 64, 3000):
        pages = []
        for k, st in enumerate(stats):
            if k % 3 == 0:
                pages.append((H(0, body, body, data=(16, 0, 3, 3, st)), body))
            elif k % 3 == 1:
                pages.append((H(3, body, body, v2=(5, 1, 5, 0, 0, 0, bool(k & 2), st)), body))
            else:
                pages.append((H(3, body, body, v2=(5, 1, 5, 0, 0, 0, None, st)), body))
        chunks.append(W.chunk(pages))
        chunks.append(W.chunk(pages[::-1], max_def=1))
    cut = []
    for c in chunks[:2]:
        for _ in range(60):
            d = dict(c)
            d["src"] = c["src"][: int(rng.integers(0, c["src"].size))].copy()
            cut.append(d)
    _compare(ctx, chunks + cut)
    res, _ = _walk_all(ctx, chunks)
    assert (res["status"] == 0).all() and (res["n_pages"] == len(stats)).all()


@pytest.mark.parametrize("version,page_size", [("1.0", 1 << 30), ("2.0", 4000), ("1.0", 4000)])
def test_pyarrow_file_end_to_end(ctx, tmp_path, version, page_size):
    import torch

    n = 5000
    table = F.make_table(n, seed=4)
    path = F.tmp_path_for(tmp_path, "e")
    F.write_file(path, table, version, page_size)
    data, recs = F.chunk_records(path)
    cap = 64
    chunks = np.zeros(len(recs), dtype=K.CHUNK_DTYPE)
    for i, r in enumerate(recs):
        c = chunks[i]
        c["src_off"], c["src_len"] = r["start"], data.size - r["start"]
        c["total_compressed_size"], c["data_page_offset"] = r["total_compressed_size"], r["data_page_offset"]
        c["has_dictionary_page_offset"] = 0 if r["dictionary_page_offset"] is None else 1
        c["dictionary_page_offset"] = r["dictionary_page_offset"] or 0
        c["physical_type"], c["type_length"] = r["physical_type"], -1
        c["max_rep"], c["max_def"] = r["max_rep"], r["max_def"]
        c["info_first"], c["info_cap"] = i * cap, cap
    dev = torch.device("cuda:0")
    t_src = torch.from_numpy(np.concatenate([data, np.zeros(16, np.uint8)])).to(dev)
    t_infos = torch.zeros(len(recs) * cap * K.PAGE_INFO_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    res = ctx.walk_chunks_dev(t_src.data_ptr(), data.size, chunks, t_infos.data_ptr(), len(recs) * cap)
    infos = t_infos.cpu().numpy().view(K.PAGE_INFO_DTYPE)
    assert (res["status"] == K.OK).all(), res
    for i, r in enumerate(recs):                    # the walk agrees with the oracle and with the footer
        want = O.walk_chunk(data[r["start"]:], r["total_compressed_size"], r["data_page_offset"],
                            r["dictionary_page_offset"], r["physical_type"], -1, r["max_rep"], r["max_def"])
        got = infos[i * cap: i * cap + int(res[i]["n_pages"])]
        assert got.tobytes() == want["infos"].tobytes() and int(res[i]["count"]) == r["total_compressed_size"]

    # descriptors for every data page of the file, one plan, one run
    layout = np.zeros(1, dtype=K.CHUNK_LAYOUT_DTYPE)
    layout["delta_ba_room"] = 1 << 18
    all_descs, spans, dict_copies = [], [], []
    for i, r in enumerate(recs):
        got = infos[i * cap: i * cap + int(res[i]["n_pages"])]
        descs, dsrc, ddst, dbytes = capi.chunk_page_descs(chunks[i], got, layout)
        spans.append((len(all_descs), len(descs)))
        all_descs += list(descs)
        if dbytes:
            dict_copies.append((dsrc, ddst, dbytes))
    descs = np.array(all_descs, dtype=K.DESC_DTYPE)
    # the one-call form builds the same table
    layout2 = np.zeros(1, dtype=K.CHUNK_LAYOUT_DTYPE)
    layout2["delta_ba_room"] = 1 << 18
    descs2, first2, copies2 = capi.chunks_page_descs(chunks, res, infos, layout2)
    assert descs2.tobytes() == descs.tobytes() and layout2.tobytes() == layout.tobytes()
    assert [int(first2[i]) for i in range(len(recs))] == [s0 for s0, _ in spans] and int(first2[-1]) == len(descs)
    assert [(int(c["src"]), int(c["dst"]), int(c["bytes"])) for c in copies2] == dict_copies
    vb, lb, db = int(layout["values_off"][0]) + 64, int(layout["levels_off"][0]) + 64, int(layout["dict_off"][0]) + 64
    t_val = torch.full((vb,), 0xA5, dtype=torch.uint8, device=dev)
    t_lvl = torch.full((lb,), 0xA5, dtype=torch.uint8, device=dev)
    t_dict = torch.zeros(db, dtype=torch.uint8, device=dev)
    for dsrc, ddst, dbytes in dict_copies:          # dictionary page bodies, device to device
        t_dict[ddst: ddst + dbytes] = t_src[dsrc: dsrc + dbytes]
    torch.cuda.synchronize()
    plan = ctx.plan(descs)
    plan.run(t_src.data_ptr(), data.size, t_val.data_ptr(), vb, t_lvl.data_ptr(), lb, t_dict.data_ptr(), db)
    results = plan.results()
    plan.close()
    assert (results["status"] == K.OK).all(), results[results["status"] != 0][:3]
    values, levels = t_val.cpu().numpy(), t_lvl.cpu().numpy()
    nested = sum(int(descs[k]["n_values"]) for i, r in enumerate(recs) if r["max_rep"]
                 for k in range(spans[i][0], spans[i][0] + spans[i][1]))
    assert int(layout["levels_off"][0]) == 8 * nested          # only the nested column's levels take room
    assert (levels[8 * nested:] == 0xA5).all()

    for i, r in enumerate(recs):
        first, cnt = spans[i]
        want_vals, want_def, want_rep = F.expected_column(table, r, n)
        vals, defs, reps = [], [], []
        for k in range(first, first + cnt):
            d, pr = descs[k], results[k]
            nn = int(pr["not_null"]) if r["max_rep"] else int(d["n_values"])
            if d["value_kind"] in (K.KIND_DELTA_LENGTH_BA, K.KIND_DELTA_BA):
                off = values[int(d["dict_off"]): int(d["dict_off"]) + 4 * (nn + 1)].view(np.uint32)
                blob = values[int(d["dst_off"]):]
                vals.extend(bytes(blob[int(off[j]): int(off[j + 1])]) for j in range(nn))
            else:
                es = {K.KIND_BOOL_PLAIN: 4, K.KIND_BOOL_RLE: 4, K.KIND_DELTA32: 4, K.KIND_DELTA64: 8}.get(
                    int(d["value_kind"]), int(d["elem_size"]))
                vals.append(values[int(d["dst_off"]): int(d["dst_off"]) + es * nn])
            nl = int(d["n_values"])
            if r["max_rep"] == 0:           # flat column: the constant-zero levels are not written
                assert d["flags"] & K.FLAG_NO_LEVEL_OUTPUT
                continue
            defs.append(levels[int(d["def_off"]): int(d["def_off"]) + 4 * nl].view(np.int32))
            reps.append(levels[int(d["rep_off"]): int(d["rep_off"]) + 4 * nl].view(np.int32))
        if isinstance(want_vals, list):
            assert vals == want_vals, r["name"]
        else:
            got_bytes = np.concatenate(vals) if vals else np.zeros(0, np.uint8)
            assert np.array_equal(got_bytes, np.ascontiguousarray(want_vals).view(np.uint8)), r["name"]
        if want_def is not None:
            assert np.array_equal(np.concatenate(defs), want_def) and np.array_equal(np.concatenate(reps), want_rep)
