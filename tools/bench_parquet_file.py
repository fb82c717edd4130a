#!/usr/bin/env python3
"""A real Parquet file, uncompressed, written by pyarrow: file bytes in HBM -> page-header walk of every
column chunk -> descriptors -> one decode plan -> values and levels in HBM, checked against what pyarrow
reads.  Times each stage (HIP events / wall clock) and, beside them, pyarrow.parquet.read_table on the same
file on the host cores (a production CPU reader; it is not the reference, which is single-threaded Go).

    python tools/bench_parquet_file.py [--rows 4000000] [--page-size 1048576] [--row-groups 4]

Columns: REQUIRED int32 / int64 / float / double in PLAIN and dictionary encoding, int32 / int64 in
DELTA_BINARY_PACKED, binary in DELTA_LENGTH_BYTE_ARRAY, booleans.  Footer metadata is read with pyarrow
(the footer is outside the decode path)."""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import pyarrow as pa  # noqa: E402
import pyarrow.parquet as pq  # noqa: E402

from parquet_go_amd import capi  # noqa: E402

PHYSICAL = {"BOOLEAN": 0, "INT32": 1, "INT64": 2, "INT96": 3, "FLOAT": 4, "DOUBLE": 5, "BYTE_ARRAY": 6,
            "FIXED_LEN_BYTE_ARRAY": 7}


def make_table(n, seed=0):
    rng = np.random.default_rng(seed)
    words = np.array([bytes(rng.integers(97, 123, size=int(rng.integers(4, 20)), dtype=np.uint8)) for _ in range(4096)],
                     dtype=object)
    cols = {
        "i32_plain": rng.integers(-2**31, 2**31, size=n).astype(np.int32),
        "i64_plain": rng.integers(-2**62, 2**62, size=n),
        "f32_plain": rng.random(n).astype(np.float32),
        "f64_plain": rng.random(n),
        "bool_plain": rng.random(n) < 0.3,
        "i32_dict": (rng.integers(0, 4000, size=n) * 1000003).astype(np.int32),
        "i64_dict": rng.integers(0, 4000, size=n) * (1 << 40),
        "f64_dict": rng.integers(0, 3000, size=n) / 7.0,
        "i32_delta": np.cumsum(rng.integers(-40, 60000, size=n)).astype(np.int32),
        "i64_delta": np.cumsum(rng.integers(-(1 << 10), 1 << 16, size=n)),
        "i64_delta_b": np.cumsum(rng.integers(0, 1 << 24, size=n)),
    }
    arrays = [pa.array(v) for v in cols.values()]
    names = list(cols)
    arrays.append(pa.array(words[rng.integers(0, 4096, size=n)].tolist(), type=pa.binary()))
    names.append("bin_dlba")
    return pa.Table.from_arrays(arrays, schema=pa.schema([pa.field(k, a.type, nullable=False) for k, a in zip(names, arrays)]))


ENCODINGS = {"i32_delta": "DELTA_BINARY_PACKED", "i64_delta": "DELTA_BINARY_PACKED", "i64_delta_b": "DELTA_BINARY_PACKED",
             "bin_dlba": "DELTA_LENGTH_BYTE_ARRAY", "i32_plain": "PLAIN", "i64_plain": "PLAIN", "f32_plain": "PLAIN",
             "f64_plain": "PLAIN", "bool_plain": "PLAIN"}
DICT_COLUMNS = ["i32_dict", "i64_dict", "f64_dict"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=4_000_000)
    ap.add_argument("--row-groups", type=int, default=4)
    ap.add_argument("--page-size", type=int, default=1 << 20)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--per-column", action="store_true", help="also time each column's pages as a plan of their own")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    rows_per_group = a.rows // a.row_groups
    # a row count per group that keeps every page off N = 1 (mod 128): the reference's delta decoder
    # cannot read such a page (encoding/delta-binary-packed-decoder.go:103); pages are cut at multiples of 1024
    if rows_per_group % 1024 == 1:
        rows_per_group += 1
    table = make_table(rows_per_group * a.row_groups)
    path = os.path.join(tempfile.mkdtemp(), "bench.parquet")
    pq.write_table(table, path, compression="NONE", use_dictionary=DICT_COLUMNS, column_encoding=ENCODINGS,
                   data_page_version="1.0", data_page_size=a.page_size, write_statistics=True,
                   row_group_size=rows_per_group, write_batch_size=1024, dictionary_pagesize_limit=1 << 22)
    data = np.fromfile(path, dtype=np.uint8)
    pf = pq.ParquetFile(path)
    md = pf.metadata
    recs = []
    for rg in range(md.num_row_groups):
        for ci in range(md.num_columns):
            cm = md.row_group(rg).column(ci)
            has_dict = cm.has_dictionary_page and cm.dictionary_page_offset is not None
            recs.append(dict(rg=rg, ci=ci, start=cm.dictionary_page_offset if has_dict else cm.data_page_offset,
                             total=cm.total_compressed_size, dpo=cm.data_page_offset,
                             dict_off=cm.dictionary_page_offset if has_dict else None,
                             ptype=PHYSICAL[cm.physical_type], encodings=cm.encodings, nv=cm.num_values))
    cap = 256
    chunks = np.zeros(len(recs), dtype=capi.CHUNK_DTYPE)
    for i, r in enumerate(recs):
        c = chunks[i]
        c["src_off"], c["src_len"] = r["start"], data.size - r["start"]
        c["total_compressed_size"], c["data_page_offset"] = r["total"], r["dpo"]
        c["has_dictionary_page_offset"] = 0 if r["dict_off"] is None else 1
        c["dictionary_page_offset"] = r["dict_off"] or 0
        c["physical_type"], c["type_length"] = r["ptype"], -1
        c["info_first"], c["info_cap"] = i * cap, cap

    stream = torch.cuda.Stream(device=dev)
    ctx = capi.Context(0, stream.cuda_stream)
    h_src = torch.from_numpy(np.concatenate([data, np.zeros(64, np.uint8)])).pin_memory()
    t_src = torch.empty_like(h_src, device=dev)
    t_infos = torch.zeros(len(recs) * cap * capi.PAGE_INFO_DTYPE.itemsize, dtype=torch.uint8, device=dev)

    def h2d():
        with torch.cuda.stream(stream):
            t_src.copy_(h_src, non_blocking=True)
        stream.synchronize()

    def walk():
        res = ctx.walk_chunks_dev(t_src.data_ptr(), data.size, chunks, t_infos.data_ptr(), len(recs) * cap)
        assert (res["status"] == 0).all(), res
        return res

    def plan_chunks(res, infos):
        layout = np.zeros(1, dtype=capi.CHUNK_LAYOUT_DTYPE)
        descs, first, cps = capi.chunks_page_descs(chunks, res, infos, layout)
        spans = [(int(first[i]), int(first[i + 1] - first[i])) for i in range(len(recs))]
        copies = [(int(c["src"]), int(c["dst"]), int(c["bytes"])) for c in cps]
        return descs, spans, copies, layout

    def wall(fn, reps):
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            out = fn()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)), out

    t_h2d, _ = wall(h2d, a.steps)
    t_walk, res = wall(walk, a.steps)
    infos = t_infos.cpu().numpy().view(capi.PAGE_INFO_DTYPE)
    t_descs, (descs, spans, copies, layout) = wall(lambda: plan_chunks(res, infos), 2)
    vb, lb, db = int(layout["values_off"][0]) + 64, int(layout["levels_off"][0]) + 64, int(layout["dict_off"][0]) + 64
    t_val = torch.empty(vb, dtype=torch.uint8, device=dev)
    t_lvl = torch.empty(max(lb, 64), dtype=torch.uint8, device=dev)
    t_dict = torch.zeros(db, dtype=torch.uint8, device=dev)
    with torch.cuda.stream(stream):
        for dsrc, ddst, dbytes in copies:
            t_dict[ddst: ddst + dbytes] = t_src[dsrc: dsrc + dbytes]
    stream.synchronize()
    t0 = time.perf_counter()
    plan = ctx.plan(descs)
    t_plan_first = time.perf_counter() - t0      # allocates the context's device block and pinned image
    t_plans = []
    for _ in range(5):                             # a reader's steady state: one plan after another
        plan.close()
        t0 = time.perf_counter()
        plan = ctx.plan(descs)
        t_plans.append(time.perf_counter() - t0)
    t_plan = float(np.median(t_plans))

    def run():
        plan.run(t_src.data_ptr(), data.size, t_val.data_ptr(), vb, t_lvl.data_ptr(), lb, t_dict.data_ptr(), db)

    run()
    stream.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    with torch.cuda.stream(stream):
        evs[0].record()
        for i in range(a.steps):
            run()
            evs[i + 1].record()
    stream.synchronize()
    t_decode = float(np.median([evs[i].elapsed_time(evs[i + 1]) for i in range(a.steps)])) / 1e3
    results = plan.results()
    assert (results["status"] == 0).all(), results[results["status"] != 0][:3]
    if os.environ.get("PQ_BENCH_DEBUG"):
        print("vb", vb, "lb", lb, "db", db, "descs", descs.size, "spans", spans[:3], file=sys.stderr)
        print(descs[["kind", "value_kind", "dst_off", "n_values", "elem_size"]][48:54], file=sys.stderr)

    # ---- every value against pyarrow ----
    values = t_val.cpu().numpy()
    out_bytes = 0
    for i, r in enumerate(recs):
        first, cnt = spans[i]
        col = table.column(r["ci"]).slice(r["rg"] * rows_per_group, rows_per_group)
        name = table.schema.names[r["ci"]]
        parts = []
        for k in range(first, first + cnt):
            d = descs[k]
            nv = int(d["n_values"])
            if d["value_kind"] == capi.KIND_DELTA_LENGTH_BA:
                off = values[int(d["dict_off"]): int(d["dict_off"]) + 4 * (nv + 1)].view(np.uint32)
                parts.append((off, values[int(d["dst_off"]): int(d["dst_off"]) + int(off[nv])]))
                out_bytes += 4 * (nv + 1) + int(off[nv])
            else:
                es = {capi.KIND_BOOL_PLAIN: 4, capi.KIND_BOOL_RLE: 4, capi.KIND_DELTA32: 4, capi.KIND_DELTA64: 8}.get(
                    int(d["value_kind"]), int(d["elem_size"]))
                parts.append(values[int(d["dst_off"]): int(d["dst_off"]) + es * nv])
                out_bytes += es * nv
        if name == "bin_dlba":
            want = col.combine_chunks()
            wbuf = np.frombuffer(want.buffers()[2], dtype=np.uint8)
            woff = np.frombuffer(want.buffers()[1], dtype=np.int32)[want.offset: want.offset + len(want) + 1]
            got_bytes = np.concatenate([p[1] for p in parts])
            assert np.array_equal(got_bytes, wbuf[woff[0]: woff[-1]]), name
            lens = np.concatenate([np.diff(p[0].astype(np.int64)) for p in parts])
            assert np.array_equal(lens, np.diff(woff.astype(np.int64))), name
        else:
            want = np.array(col.to_pylist(), dtype=np.int32) if name == "bool_plain" else col.to_numpy()
            got_all, want_all = np.concatenate(parts), np.ascontiguousarray(want).view(np.uint8)
            if not np.array_equal(got_all, want_all):
                bad = np.flatnonzero(got_all != want_all) if got_all.size == want_all.size else np.array([-1])
                raise AssertionError("%s row group %d: %d of %d bytes differ (decoded %d), first at byte %d; %d pages"
                                     % (name, r["rg"], bad.size, want_all.size, got_all.size, int(bad[0]), cnt))

    per_column = {}
    if a.per_column:
        for ci in range(md.num_columns):
            sel = np.concatenate([np.arange(spans[i][0], spans[i][0] + spans[i][1]) for i, r in enumerate(recs) if r["ci"] == ci])
            sub = ctx.plan(descs[sel])
            fn = lambda: sub.run(t_src.data_ptr(), data.size, t_val.data_ptr(), vb, t_lvl.data_ptr(), lb, t_dict.data_ptr(), db)
            fn()
            stream.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            with torch.cuda.stream(stream):
                ev[0].record()
                for k in range(3):
                    fn()
                    ev[k + 1].record()
            stream.synchronize()
            per_column[table.schema.names[ci]] = {"pages": int(sel.size),
                                                  "ms": round(float(np.median([ev[k].elapsed_time(ev[k + 1]) for k in range(3)])), 4)}
            sub.close()

    # ---- pyarrow on the host cores, same file (in the page cache) ----
    t_pa, _ = wall(lambda: pq.read_table(path, use_threads=True), 3)
    t_pa1, _ = wall(lambda: pq.read_table(path, use_threads=False), 2)
    n_pages = int(res["n_pages"].sum())
    total_gpu = t_walk + t_descs + t_plan + t_decode
    print(json.dumps({
        "file_bytes": int(data.size), "decoded_bytes": int(out_bytes), "row_groups": md.num_row_groups,
        "columns": md.num_columns, "chunks": len(recs), "pages": n_pages, "rows": table.num_rows,
        "ms": {"h2d_file_pinned": round(t_h2d * 1e3, 3), "walk_all_chunks": round(t_walk * 1e3, 3),
               "descriptors_host": round(t_descs * 1e3, 3), "plan_create": round(t_plan * 1e3, 3),
               "plan_create_first_of_the_context": round(t_plan_first * 1e3, 3),
               "decode": round(t_decode * 1e3, 4)},
        "decode_out_GBps": round(out_bytes / t_decode / 1e9, 1),
        "decode_alg_frac_of_8TBps": round((data.size + out_bytes) / t_decode / 8e12, 4),
        "file_to_values_out_GBps_device_resident": round(out_bytes / total_gpu / 1e9, 1),
        "file_to_values_out_GBps_with_h2d": round(out_bytes / (total_gpu + t_h2d) / 1e9, 1),
        "pyarrow_read_table_out_GBps": {"threads": round(out_bytes / t_pa / 1e9, 2), "one_thread": round(out_bytes / t_pa1 / 1e9, 2),
                                        "cpu_count": len(os.sched_getaffinity(0))},
        "verified": "every value equal to pyarrow's", **({"per_column": per_column} if per_column else {})}), flush=True)
    plan.close()
    ctx.close()


if __name__ == "__main__":
    main()
