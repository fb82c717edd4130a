"""Real Parquet files as an independent check of the page-header walk and of the whole chunk path:
written by the pyarrow in this image (uncompressed), footer read by pyarrow too (the footer is outside
the path), column-chunk bytes handed to the walk.  pyarrow is not the oracle: it tells what values a
correct reader gets and how many pages / values a chunk holds.

Columns are picked so that the reference itself would read them: REQUIRED flat columns and a nested
list (the reference decodes levels only when max_rep > 0, layout/chunk.go:111-124), delta miniblock
widths <= 32 and page value counts N with N != 1 mod 128 (encoding/delta-binary-packed-decoder.go:78-88,
:103)."""
import os

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq

PHYSICAL = {"BOOLEAN": 0, "INT32": 1, "INT64": 2, "INT96": 3, "FLOAT": 4, "DOUBLE": 5, "BYTE_ARRAY": 6,
            "FIXED_LEN_BYTE_ARRAY": 7}


def make_table(n, seed=0):
    rng = np.random.default_rng(seed)
    words = [bytes(rng.integers(97, 123, size=int(rng.integers(1, 12)), dtype=np.uint8)) for _ in range(300)]
    lists = []
    for _ in range(n):
        u = rng.random()
        if u < 0.1:
            lists.append(None)
        elif u < 0.2:
            lists.append([])
        else:
            lists.append([None if rng.random() < 0.15 else int(x) for x in rng.integers(0, 50, size=int(rng.integers(1, 6)))])
    cols = {
        "i32_plain": pa.array(rng.integers(-2**31, 2**31, size=n).astype(np.int32)),
        "i64_plain": pa.array(rng.integers(-2**62, 2**62, size=n)),
        "f32_plain": pa.array(rng.random(n).astype(np.float32)),
        "f64_plain": pa.array(rng.random(n)),
        "bool_plain": pa.array(rng.random(n) < 0.3),
        "i32_dict": pa.array(rng.integers(0, 700, size=n).astype(np.int32) * 1000003),
        "i64_dict": pa.array(rng.integers(0, 50, size=n) * (1 << 40)),
        "f64_dict": pa.array(rng.integers(0, 90, size=n) / 7.0),
        "i32_delta": pa.array(np.cumsum(rng.integers(-40, 900, size=n)).astype(np.int32)),
        "i64_delta": pa.array(np.cumsum(rng.integers(-(1 << 20), 1 << 30, size=n))),
        "bin_dlba": pa.array([words[i] for i in rng.integers(0, 300, size=n)], type=pa.binary()),
        "bin_dba": pa.array(sorted(words[i] + b"/" + words[j] for i, j in rng.integers(0, 300, size=(n, 2))),
                            type=pa.binary()),
        "list_i32": pa.array(lists, type=pa.list_(pa.int32())),
    }
    fields = [pa.field(k, v.type, nullable=(k == "list_i32")) for k, v in cols.items()]
    return pa.Table.from_arrays(list(cols.values()), schema=pa.schema(fields))


ENCODINGS = {"i32_delta": "DELTA_BINARY_PACKED", "i64_delta": "DELTA_BINARY_PACKED",
             "bin_dlba": "DELTA_LENGTH_BYTE_ARRAY", "bin_dba": "DELTA_BYTE_ARRAY",
             "i32_plain": "PLAIN", "i64_plain": "PLAIN", "f32_plain": "PLAIN", "f64_plain": "PLAIN",
             "bool_plain": "PLAIN"}
DICT_COLUMNS = ["i32_dict", "i64_dict", "f64_dict", "list_i32"]


def write_file(path, table, version, page_size, row_group_size=None):
    pq.write_table(table, path, compression="NONE", use_dictionary=DICT_COLUMNS, column_encoding=ENCODINGS,
                   data_page_version=version, data_page_size=page_size, write_statistics=True,
                   row_group_size=row_group_size or table.num_rows, write_batch_size=1024)


def chunk_records(path):
    """-> (file bytes, list of dict per column chunk: the ColumnMetaData fields the walk needs, names, counts)."""
    data = np.fromfile(path, dtype=np.uint8)
    pf = pq.ParquetFile(path)
    md = pf.metadata
    out = []
    for rg in range(md.num_row_groups):
        for ci in range(md.num_columns):
            cm = md.row_group(rg).column(ci)
            sc = pf.schema.column(ci)
            has_dict = cm.has_dictionary_page and cm.dictionary_page_offset is not None
            start = cm.dictionary_page_offset if has_dict else cm.data_page_offset
            out.append(dict(name=cm.path_in_schema, row_group=rg, column=ci, start=start,
                            total_compressed_size=cm.total_compressed_size, data_page_offset=cm.data_page_offset,
                            dictionary_page_offset=cm.dictionary_page_offset if has_dict else None,
                            physical_type=PHYSICAL[cm.physical_type], max_rep=sc.max_repetition_level,
                            max_def=sc.max_definition_level, num_values=cm.num_values,
                            encodings=tuple(cm.encodings)))
    return data, out


def expected_column(table, rec, rows_per_group):
    """What a reader of this chunk returns: (values as numpy / list of bytes, def levels or None, rep levels or None)."""
    lo = rec["row_group"] * rows_per_group
    col = table.column(rec["column"]).slice(lo, rows_per_group)
    name = rec["name"].split(".")[0]
    if name == "list_i32":
        deff, rep, vals = [], [], []
        for row in col.to_pylist():
            if row is None:
                deff.append(0), rep.append(0)
            elif not row:
                deff.append(1), rep.append(0)
            else:
                for k, x in enumerate(row):
                    rep.append(0 if k == 0 else 1)
                    deff.append(2 if x is None else 3)
                    if x is not None:
                        vals.append(x)
        return np.array(vals, dtype=np.int32), np.array(deff, dtype=np.int32), np.array(rep, dtype=np.int32)
    if name.startswith("bin_"):
        return [bytes(x) for x in col.to_pylist()], None, None
    if name == "bool_plain":
        return np.array(col.to_pylist(), dtype=np.int32), None, None
    return col.to_numpy(), None, None


def tmp_path_for(tmpdir, tag):
    return os.path.join(str(tmpdir), "t_%s.parquet" % tag)
