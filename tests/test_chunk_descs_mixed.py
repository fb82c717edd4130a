"""pq_chunks_page_descs (host bookkeeping, no device work) on a row group in which one chunk is of a
kind the library does not decode: a PLAIN BYTE_ARRAY column (what writers emit for strings by
default) between integer columns.  The chunk is reported, gets no descriptors and takes no room; the
others are laid out as if it were not there.  Page records come from the oracle's header walk."""
import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq

import oracle_lib as O
import parquet_files as F
from parquet_go_amd import capi

K = capi


def _file(tmp_path):
    rng = np.random.default_rng(3)
    n = 6000
    words = [bytes(rng.integers(97, 123, size=int(rng.integers(1, 12)), dtype=np.uint8)) for _ in range(200)]
    cols = {
        "a_i32": pa.array(rng.integers(-2**31, 2**31, size=n).astype(np.int32)),
        "b_str_plain": pa.array([words[i] for i in rng.integers(0, 200, size=n)], type=pa.binary()),
        "c_i64_delta": pa.array(np.cumsum(rng.integers(0, 1 << 20, size=n))),
        "d_i32_dict": pa.array(rng.integers(0, 300, size=n).astype(np.int32) * 7),
    }
    table = pa.table(cols)
    path = str(tmp_path / "mixed.parquet")
    pq.write_table(table, path, compression="NONE", use_dictionary=["d_i32_dict"],
                   column_encoding={"a_i32": "PLAIN", "b_str_plain": "PLAIN", "c_i64_delta": "DELTA_BINARY_PACKED"},
                   data_page_size=4000, row_group_size=n // 2, write_batch_size=512)
    return path


def _walked(path, cap=64):
    data, recs = F.chunk_records(path)
    chunks = np.zeros(len(recs), dtype=K.CHUNK_DTYPE)
    results = np.zeros(len(recs), dtype=K.CHUNK_RESULT_DTYPE)
    infos = np.zeros(len(recs) * cap, dtype=K.PAGE_INFO_DTYPE)
    for i, r in enumerate(recs):
        c = chunks[i]
        c["src_off"], c["src_len"] = r["start"], data.size - r["start"]
        c["total_compressed_size"], c["data_page_offset"] = r["total_compressed_size"], r["data_page_offset"]
        c["has_dictionary_page_offset"] = 0 if r["dictionary_page_offset"] is None else 1
        c["dictionary_page_offset"] = r["dictionary_page_offset"] or 0
        c["physical_type"], c["type_length"] = r["physical_type"], -1
        c["max_rep"], c["max_def"] = r["max_rep"], r["max_def"]
        c["info_first"], c["info_cap"] = i * cap, cap
        w = O.walk_chunk(data[r["start"]:], r["total_compressed_size"], r["data_page_offset"],
                         r["dictionary_page_offset"], r["physical_type"], -1, r["max_rep"], r["max_def"], cap=cap)
        assert w["status"] == O.OK                       # the walk itself accepts every chunk
        k = len(w["infos"])
        infos[i * cap: i * cap + k] = w["infos"].view(K.PAGE_INFO_DTYPE)
        results[i]["status"], results[i]["n_pages"], results[i]["count"] = 0, k, w["count"]
    return recs, chunks, results, infos


def test_one_undecodable_chunk_does_not_stop_the_row_group(tmp_path):
    recs, chunks, results, infos = _walked(_file(tmp_path))
    names = [r["name"] for r in recs]
    assert names.count("b_str_plain") == 2                # two row groups
    layout = np.zeros(1, dtype=K.CHUNK_LAYOUT_DTYPE)
    descs, first, copies, status, rc = K.chunks_page_descs(chunks, results, infos, layout, with_status=True)
    assert rc == K.OK
    for i, name in enumerate(names):
        if name == "b_str_plain":
            assert status[i] == K.ERR_BAD_ARGUMENT and first[i + 1] == first[i]
        else:
            assert status[i] == K.OK and first[i + 1] > first[i]
    assert int(first[-1]) == descs.size and len(copies) == names.count("d_i32_dict")

    # the same table as from the good chunks alone, one by one
    layout2 = np.zeros(1, dtype=K.CHUNK_LAYOUT_DTYPE)
    want, want_copies = [], []
    for i, r in enumerate(recs):
        if r["name"] == "b_str_plain":
            continue
        got = infos[int(chunks[i]["info_first"]): int(chunks[i]["info_first"]) + int(results[i]["n_pages"])]
        d, dsrc, ddst, dbytes = K.chunk_page_descs(chunks[i], got, layout2)
        want += list(d)
        if dbytes:
            want_copies.append((dsrc, ddst, dbytes))
    assert descs.tobytes() == np.array(want, dtype=K.DESC_DTYPE).tobytes()
    assert layout.tobytes() == layout2.tobytes()
    assert [(int(c["src"]), int(c["dst"]), int(c["bytes"])) for c in copies] == want_copies

    # the strict form of the binding refuses the row group and names the chunk
    try:
        K.chunks_page_descs(chunks, results, infos, np.zeros(1, dtype=K.CHUNK_LAYOUT_DTYPE))
        raise AssertionError("expected PqError")
    except K.PqError as e:
        assert e.code == K.ERR_BAD_ARGUMENT


def test_failed_walks_and_a_full_table_are_reported_per_chunk(tmp_path):
    recs, chunks, results, infos = _walked(_file(tmp_path))
    results = results.copy()
    results[0]["status"] = K.WALK_HEADER_FAILED           # a chunk whose walk failed
    layout = np.zeros(1, dtype=K.CHUNK_LAYOUT_DTYPE)
    descs, first, copies, status, rc = K.chunks_page_descs(chunks, results, infos, layout, with_status=True)
    assert rc == K.OK and status[0] == K.WALK_HEADER_FAILED and first[1] == first[0] == 0
    total = descs.size
    # room for fewer descriptors than the row group needs: the chunk that does not fit and all
    # behind it carry PQ_ERR_OUT_OF_RANGE, the table holds the chunks in front, counts are written
    layout = np.zeros(1, dtype=K.CHUNK_LAYOUT_DTYPE)
    descs2, first2, copies2, status2, rc2 = K.chunks_page_descs(chunks, results, infos, layout, with_status=True,
                                                               descs_cap=total - 1)
    assert rc2 == K.ERR_OUT_OF_RANGE
    bad = np.flatnonzero(status2 == K.ERR_OUT_OF_RANGE)
    assert bad.size >= 1 and (status2[bad[0]:] != K.OK).all()
    assert int(first2[-1]) == descs2.size < total and (first2[bad[0]:] == first2[-1]).all()
    assert descs2.tobytes() == descs[: descs2.size].tobytes()
