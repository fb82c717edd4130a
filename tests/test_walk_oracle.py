"""CPU: the oracle's page-header walk (ChunkReader.readPages, layout/chunk.go:129-217) and, with it, the
whole chunk path of the oracle against real Parquet files written by pyarrow — an independent source:
it is neither the reference nor derived from the restatement.  The reference has no test for this
loop (parity unpinned); the Thrift compact reader is a dependency that is not vendored."""
import numpy as np
import pytest

import oracle_lib as O
import parquet_files as F
import walk_cases as W

ENC_KIND = {(1, 0): O.VALUES_PLAIN, (2, 0): O.VALUES_PLAIN, (4, 0): O.VALUES_PLAIN, (5, 0): O.VALUES_PLAIN,
            (0, 0): O.VALUES_BOOL_PLAIN, (0, 3): O.VALUES_BOOL_RLE, (1, 5): O.VALUES_DELTA32, (2, 5): O.VALUES_DELTA64,
            (6, 6): O.VALUES_DELTA_LENGTH_BA, (6, 7): O.VALUES_DELTA_BA}
DTYPE = {1: np.int32, 2: np.int64, 4: np.float32, 5: np.float64}


def _walk(c, cap=4096):
    return O.walk_chunk(c["src"], c["total"], c["data_page_offset"], c["dictionary_page_offset"], c["physical_type"],
                        c["type_length"], c["max_rep"], c["max_def"], c["codec"], cap=cap)


def test_synthetic_chunks_expected_outcomes():
    rng = np.random.default_rng(21)
    got = {name: _walk(c) for name, c in W.good_and_bad_chunks(rng)}
    ok = lambda name, pages: got[name]["status"] == O.OK and len(got[name]["infos"]) == pages
    assert ok("three v1 pages", 3) and got["three v1 pages"]["infos"]["num_values"].tolist() == [100, 7, 1000]
    assert ok("v1 with statistics and crc", 2) and ok("v2 pages", 3)
    assert got["v2 pages"]["infos"]["is_compressed"].tolist() == [1, 1, 0]
    assert got["v2 pages"]["infos"]["def_len"].tolist() == [0, 20, 3]
    assert ok("dictionary then data, with offsets", 3) and ok("dictionary without a dictionary offset", 3)
    c = dict(W.good_and_bad_chunks(np.random.default_rng(21)))
    for name in ("three v1 pages", "v2 pages", "dictionary then data, with offsets", "fields nobody knows",
                 "three hundred pages", "header longer than the window"):
        assert got[name]["status"] == O.OK and got[name]["count"] == c[name]["src"].size, name
    assert ok("seek over a gap", 2) and got["seek over a gap"]["count"] == c["seek over a gap"]["src"].size
    assert got["seek to before the chunk"]["status"] == 30
    assert ok("seek past the end", 1)                        # Count() jumps past the total: the loop just ends
    assert ok("fields nobody knows", 2) and ok("index page header is read and dropped", 1)
    assert ok("byte arrays", 2) and ok("booleans", 2) and ok("fixed length", 1)
    assert ok("total size ends the loop early", 1) and ok("total size 0", 0)
    assert ok("other codec: sizes differ", 1) and ok("nesting the skip still allows", 1)
    assert ok("dictionary encoding PLAIN_DICTIONARY is fine", 2)
    assert ok("bit-packed levels on a flat column are not looked at", 1)
    assert ok("three hundred pages", 300) and ok("header longer than the window", 2)
    want = {"second dictionary": 20, "index page": 21, "page type 7": 21, "v1 without its header": 22,
            "v2 without its header": 22, "dictionary without its header": 22, "negative num values v1": 23,
            "negative num values v2": 23, "negative num values dict": 23, "negative rep length": 24,
            "negative def length": 24, "dictionary encoding RLE": 25, "bit-packed levels on a nested column": 26,
            "delta encoding for floats": 27, "delta length for int32 (v2)": 27, "fixed length without a length": 27,
            "boolean dictionary page": 27, "type 9": 27, "last block five bytes short": 28,
            "negative compressed size": 28, "sizes differ without a codec": 29, "v2 levels longer than the page": 28,
            "v2 levels longer than the source": 28, "required field missing": 19, "data header misses an encoding": 19,
            "wrong wire type for a known field is skipped, then missed": 19, "garbage": 19, "empty source": 19,
            "nesting deeper than the skip allows": 19, "negative binary length": 19}
    for name, status in want.items():
        assert got[name]["status"] == status, (name, got[name]["status"], status)
    assert len(got["second dictionary"]["infos"]) == 1 and len(got["last block five bytes short"]["infos"]) == 1
    assert _walk(c["three hundred pages"], cap=299)["status"] == O.OUT_OF_RANGE_CAP


def test_truncated_chunks_never_report_more_than_is_there():
    rng = np.random.default_rng(22)
    chunks = dict(W.good_and_bad_chunks(rng))
    for name in ("v1 with statistics and crc", "v2 pages", "dictionary then data, with offsets", "fields nobody knows"):
        full = _walk(chunks[name])
        for c in W.truncations(chunks[name]):
            r = _walk(c)
            n = len(r["infos"])
            assert np.array_equal(r["infos"], full["infos"][:n])
            if c["src"].size < chunks[name]["src"].size:
                assert r["status"] in (19, 28) and n < len(full["infos"])


def _decode_chunk(data, rec, infos):
    """Every data page of the chunk through the oracle's page reader -> (values, def, rep)."""
    base = rec["start"]
    dictionary = None
    vals, defs, reps = [], [], []
    for pi in infos:
        body = data[base + int(pi["data_off"]): base + int(pi["data_off"]) + int(pi["compressed_size"])]
        t = rec["physical_type"]
        if pi["type"] == 2:
            dictionary = body[: int(pi["num_values"]) * np.dtype(DTYPE[t]).itemsize].view(DTYPE[t])
            continue
        enc = 8 if pi["encoding"] == 2 else int(pi["encoding"])
        kind = O.VALUES_DICT if enc == 8 else ENC_KIND[(t, enc)]
        n = int(pi["num_values"])
        r = O.page_read(body, 1 if pi["type"] == 0 else 2, n, rec["max_rep"], rec["max_def"], kind,
                        dictionary=dictionary, rep_len=int(pi["rep_len"]), def_len=int(pi["def_len"]),
                        value_dtype=DTYPE.get(t) if kind == O.VALUES_PLAIN else None, data_cap=64 * body.size)
        assert r["status"] == O.OK, (rec["name"], r["status"], r["stage"], r["detail"])
        nn = r["not_null"]
        if kind in (O.VALUES_DELTA_LENGTH_BA, O.VALUES_DELTA_BA):
            off = r["values"][: 4 * (n + 1)].view(np.uint32)
            blob = r["values"][4 * (n + 1):]
            vals.extend(bytes(blob[int(off[i]): int(off[i + 1])]) for i in range(nn))
        else:
            vals.append(r["values"][:nn])
        defs.append(r["def_"])
        reps.append(r["rep"])
    return vals, defs, reps


@pytest.mark.parametrize("version,page_size", [("1.0", 1 << 30), ("2.0", 1 << 30), ("1.0", 4000), ("2.0", 4000)])
def test_pyarrow_files_walk_and_decode(tmp_path, version, page_size):
    n = 5000
    table = F.make_table(n, seed=3)
    path = F.tmp_path_for(tmp_path, "a")
    F.write_file(path, table, version, page_size)
    data, recs = F.chunk_records(path)
    assert len(recs) == table.num_columns
    for rec in recs:
        w = O.walk_chunk(data[rec["start"]:], rec["total_compressed_size"], rec["data_page_offset"],
                         rec["dictionary_page_offset"], rec["physical_type"], -1, rec["max_rep"], rec["max_def"])
        assert w["status"] == O.OK, (rec["name"], w["status"])
        assert w["count"] == rec["total_compressed_size"]
        infos = w["infos"]
        pages = infos[infos["type"] != 2]
        assert int(pages["num_values"].sum()) == rec["num_values"]
        assert set(pages["type"].tolist()) == {0 if version == "1.0" else 3}
        assert (infos["type"] == 2).sum() == (1 if rec["dictionary_page_offset"] is not None else 0)
        want_vals, want_def, want_rep = F.expected_column(table, rec, n)
        vals, defs, reps = _decode_chunk(data, rec, infos)
        if isinstance(want_vals, list):
            assert vals == want_vals, rec["name"]
        else:
            got = np.concatenate(vals) if vals else np.zeros(0)
            assert np.array_equal(got.view(np.uint8), np.ascontiguousarray(want_vals).view(np.uint8)), rec["name"]
        if want_def is not None:
            assert np.array_equal(np.concatenate(defs), want_def) and np.array_equal(np.concatenate(reps), want_rep)


def test_unknown_fields_of_every_shape_are_stepped_over():
    rng = np.random.default_rng(24)
    for c in W.chunks_with_unknown_fields(rng, 300):
        r = _walk(c)
        assert r["status"] == O.OK and len(r["infos"]) == 2 and r["count"] == c["src"].size
        assert r["infos"]["num_values"].tolist() == [8, 4]
