"""Column chunks for the page-header walk tests: synthetic chunks assembled from the Thrift writer
in parquet_go_amd.synth (every header shape and every error class of layout/chunk.go:129-217) and
helpers shared by the CPU (oracle) and GPU (parity) tests."""
import numpy as np

from parquet_go_amd import synth

RLE = 3


def _u8(b):
    return np.frombuffer(bytes(b), dtype=np.uint8)


def chunk(pages, **meta):
    """pages: list of (header bytes, body length or body bytes).  -> dict(src, total, **meta)"""
    parts = []
    for h, body in pages:
        parts.append(_u8(h))
        parts.append(np.zeros(body, np.uint8) if isinstance(body, int) else np.asarray(body, np.uint8))
    src = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
    d = dict(src=src, total=int(src.size), data_page_offset=4, dictionary_page_offset=None, physical_type=1,
             type_length=-1, max_rep=0, max_def=0, codec=0)
    d.update(meta)
    return d


def good_and_bad_chunks(rng):
    """(name, chunk) pairs: well-formed chunks of every page kind, then one chunk per failure."""
    H = synth.page_header
    st = synth.statistics(rng)
    v1 = lambda n=100, size=400, enc=0, de=RLE, re=RLE, **kw: (H(0, size, size, data=(n, enc, de, re), **kw), size)
    v2 = lambda n=100, size=400, enc=0, dl=0, rl=0, **kw: (H(3, size, size, v2=(n, 0, n, enc, dl, rl), **kw), size)
    dic = lambda n=10, size=40, enc=0: (H(2, size, size, dictionary=(n, enc)), size)
    out = []
    out.append(("three v1 pages", chunk([v1(), v1(7, 28), v1(1000, 4000)])))
    out.append(("v1 with statistics and crc", chunk([(H(0, 64, 64, data=(16, 0, RLE, RLE, st), crc=-12345), 64), v1()])))
    out.append(("v2 pages", chunk([v2(), v2(50, 300, 0, 20, 30), (H(3, 90, 90, v2=(5, 1, 5, 0, 3, 4, False, st)), 90)],
                                  max_rep=1, max_def=2)))
    out.append(("dictionary then data, with offsets", None))  # filled below
    d0 = dic()
    pages = [d0, v1(100, 130, 8), v1(100, 131, 2)]
    c = chunk(pages, dictionary_page_offset=1000)
    c["data_page_offset"] = 1000 + len(d0[0]) + d0[1]
    out[-1] = ("dictionary then data, with offsets", c)
    # the dictionary offset is absent from the metadata: no seek, reading starts at data_page_offset
    out.append(("dictionary without a dictionary offset", chunk(pages)))
    # a gap between the dictionary page and the first data page: the seek crosses it and Count() with it
    gap = chunk([d0, (b"", 17), v1(100, 130, 8)], dictionary_page_offset=500)
    gap["data_page_offset"] = 500 + len(d0[0]) + d0[1] + 17
    out.append(("seek over a gap", gap))
    back = chunk([d0, v1(100, 130, 8)], dictionary_page_offset=500)
    back["data_page_offset"] = 500 - 20
    out.append(("seek to before the chunk", back))
    far = chunk([d0, v1(100, 130, 8)], dictionary_page_offset=500)
    far["data_page_offset"] = 500 + 10_000
    out.append(("seek past the end", far))
    unknown = [(9, "i32", 5), (10, ("list", "i64"), [1, -2, 3]), (11, ("map", "i32", "binary"), [(1, b"x"), (2, b"yz")]),
               (12, "double", 2.5), (13, "struct", [(1, ("list", "struct"), [[(1, "bool", True)], [(2, "byte", 7)]]),
                                                     (3, ("set", "bool"), [True, False])]),
               (300, "i16", -3), (301, ("list", "i32"), list(range(20))), (302, "bool", False), (303, ("map", "i32", "i32"), [])]
    out.append(("fields nobody knows", chunk([(H(0, 40, 40, data=(10, 0, RLE, RLE), extra=unknown), 40), v1()])))
    out.append(("index page header is read and dropped",
                chunk([(H(0, 8, 8, data=(2, 0, RLE, RLE), extra=[(6, "struct", [])]), 8)])))
    out.append(("byte arrays", chunk([v1(enc=6), v2(enc=7)], physical_type=6)))
    out.append(("booleans", chunk([v1(enc=0), v2(enc=3)], physical_type=0)))
    out.append(("fixed length", chunk([v1(enc=7)], physical_type=7, type_length=16)))
    out.append(("total size ends the loop early", dict(chunk([v1(), v1()]), total=1)))
    out.append(("total size 0", dict(chunk([v1()]), total=0)))
    out.append(("other codec: sizes differ", chunk([(H(0, 999, 400, data=(5, 0, RLE, RLE)), 400)], codec=1)))
    # ---- failures ----
    out.append(("second dictionary", chunk([dic(), dic()])))
    out.append(("index page", chunk([(H(1, 10, 10), 10)])))
    out.append(("page type 7", chunk([(H(7, 10, 10), 10)])))
    out.append(("v1 without its header", chunk([(H(0, 10, 10), 10)])))
    out.append(("v2 without its header", chunk([(H(3, 10, 10), 10)])))
    out.append(("dictionary without its header", chunk([(H(2, 10, 10), 10)])))
    out.append(("negative num values v1", chunk([v1(-1)])))
    out.append(("negative num values v2", chunk([v2(-5)])))
    out.append(("negative num values dict", chunk([dic(-1)])))
    out.append(("negative rep length", chunk([v2(rl=-1)])))
    out.append(("negative def length", chunk([v2(dl=-7)])))
    out.append(("dictionary encoding RLE", chunk([dic(enc=3)])))
    out.append(("dictionary encoding PLAIN_DICTIONARY is fine", chunk([dic(enc=2), v1(enc=2)])))
    out.append(("bit-packed levels on a nested column", chunk([v1(de=4)], max_rep=1, max_def=1)))
    out.append(("bit-packed levels on a flat column are not looked at", chunk([v1(de=4, re=4)], max_def=1)))
    out.append(("delta encoding for floats", chunk([v1(enc=5)], physical_type=4)))
    out.append(("delta length for int32 (v2)", chunk([v2(enc=6)])))
    out.append(("fixed length without a length", chunk([v1(enc=0)], physical_type=7)))
    out.append(("boolean dictionary page", chunk([dic()], physical_type=0)))
    out.append(("type 9", chunk([v1()], physical_type=9)))
    short = chunk([v1(), v1()])
    short["src"] = short["src"][:-5]
    out.append(("last block five bytes short", short))
    out.append(("negative compressed size", chunk([(H(0, 10, -10, data=(5, 0, RLE, RLE)), 10)])))
    out.append(("sizes differ without a codec", chunk([(H(0, 12, 10, data=(5, 0, RLE, RLE)), 10)])))
    out.append(("v2 levels longer than the page", chunk([(H(3, 20, 20, v2=(5, 0, 5, 0, 30, 0)), 20)], max_rep=1)))
    out.append(("v2 levels longer than the source", chunk([(H(3, 20, 20, v2=(5, 0, 5, 0, 300, 0)), 20)], max_rep=1)))
    out.append(("required field missing", chunk([(synth.thrift_struct([(1, "i32", 0), (3, "i32", 10)]), 10)])))
    out.append(("data header misses an encoding",
                chunk([(synth.thrift_struct([(1, "i32", 0), (2, "i32", 4), (3, "i32", 4),
                                             (5, "struct", [(1, "i32", 1), (2, "i32", 0), (3, "i32", 3)])]), 4)])))
    out.append(("wrong wire type for a known field is skipped, then missed",
                chunk([(synth.thrift_struct([(1, "i64", 0), (2, "i32", 4), (3, "i32", 4)]), 4)])))
    out.append(("garbage", dict(chunk([(bytes(rng.integers(0, 256, size=64, dtype=np.uint8)), 0)]), total=64)))
    out.append(("empty source", dict(chunk([]), total=100)))
    deep = [(1, "i32", 1)]
    for _ in range(70):
        deep = [(1, "struct", deep)]
    out.append(("nesting deeper than the skip allows", chunk([(H(0, 4, 4, data=(1, 0, RLE, RLE), extra=[(20, "struct", deep)]), 4)])))
    ok_deep = [(1, "i32", 1)]
    for _ in range(60):
        ok_deep = [(1, "struct", ok_deep)]
    out.append(("nesting the skip still allows", chunk([(H(0, 4, 4, data=(1, 0, RLE, RLE), extra=[(20, "struct", ok_deep)]), 4)])))
    out.append(("negative binary length", chunk([(H(0, 4, 4, data=(1, 0, RLE, RLE)), 4)])))
    hb = bytearray(H(0, 4, 4, data=(1, 0, RLE, RLE, [(1, "binary", b"abc")])))
    hb[hb.index(b"\x03abc")] = 0x80
    hb = bytes(hb).replace(b"\x80abc", b"\xff\xff\xff\xff\x0fabc")
    out[-1] = ("negative binary length", chunk([(hb, 4)]))
    many = chunk([v1(3, 12) for _ in range(300)])
    out.append(("three hundred pages", many))
    big = H(0, 50, 50, data=(9, 0, RLE, RLE, [(1, "binary", bytes(3000)), (5, "binary", bytes(rng.integers(0, 256, 5000, dtype=np.uint8)))]))
    out.append(("header longer than the window", chunk([(big, 50), v1()])))
    return out


def truncations(c, step=1):
    """The same chunk cut at every step-th byte."""
    for cut in range(0, c["src"].size + 1, step):
        d = dict(c)
        d["src"] = c["src"][:cut].copy()
        yield d


SCALARS = ["i32", "i64", "i16", "byte", "double", "binary", "bool"]


def random_kind(rng, depth):
    """A random Thrift kind as parquet_go_amd.synth.thrift_value names it; containers up to six deep."""
    kinds = list(SCALARS)
    if depth < 6:
        kinds += ["struct", "list", "set", "map"] * 2
    k = kinds[int(rng.integers(0, len(kinds)))]
    if k in ("list", "set"):
        return (k, SCALARS[int(rng.integers(0, 7))] if rng.random() < 0.6 else "struct")
    if k == "map":
        return ("map", ["i32", "i64", "binary", "byte"][int(rng.integers(0, 4))],
                ["i32", "binary", "struct", "double"][int(rng.integers(0, 4))])
    return k


def random_value(rng, kind, depth):
    """A value of the given kind: what the skip of an unknown field has to step over."""
    if kind in ("i32", "i16"):
        return int(rng.integers(-2**15, 2**15))
    if kind == "i64":
        return int(rng.integers(-2**62, 2**62))
    if kind == "byte":
        return int(rng.integers(0, 256))
    if kind == "double":
        return float(rng.random())
    if kind == "binary":
        return bytes(rng.integers(0, 256, size=int(rng.integers(0, 40)), dtype=np.uint8))
    if kind == "bool":
        return bool(rng.integers(0, 2))
    if kind == "struct":
        fields, fid = [], 0
        for _ in range(int(rng.integers(0, 5))):
            fid += int(rng.integers(1, 40))
            k = random_kind(rng, depth + 1)
            fields.append((fid, k, random_value(rng, k, depth + 1)))
        return fields
    if kind[0] in ("list", "set"):
        return [random_value(rng, kind[1], depth + 1) for _ in range(int(rng.integers(0, 20)))]
    return [(random_value(rng, kind[1], depth + 1), random_value(rng, kind[2], depth + 1))
            for _ in range(int(rng.integers(0, 6)))]


def random_thrift_value(rng, depth=0):
    k = random_kind(rng, depth)
    return k, random_value(rng, k, depth)


def chunks_with_unknown_fields(rng, count):
    """Well-formed V1 chunks whose headers carry random unknown fields at every level."""
    out = []
    for _ in range(count):
        extra, fid = [], 8
        for _ in range(int(rng.integers(1, 4))):
            fid += int(rng.integers(1, 30))
            k, v = random_thrift_value(rng)
            extra.append((fid, k, v))
        stats = [(int(rng.integers(7, 20)), *random_thrift_value(rng))]
        h = synth.page_header(0, 32, 32, data=(8, 0, RLE, RLE, stats), extra=extra)
        out.append(chunk([(h, 32), (synth.page_header(0, 16, 16, data=(4, 0, RLE, RLE)), 16)]))
    return out
