#!/usr/bin/env python3
"""Secondary measurements (not the headline bench): the other workload configurations,
device-resident, HIP-event timed.  Prints one JSON object per configuration.

  C1  INT32 RLE_DICTIONARY, width 12, 4096-entry dictionary   (pages x 20000 values)
  C1b same with run-heavy keys (geometric runs, mean 32)
  C3  plain bit-unpack sweep (int32 and int64 widths), 256 MiB encoded per width
  C4  nested V1 data pages: rep (w=1) + def (w=2) levels + dictionary values
  C2x INT64 delta at other miniblock widths / INT32 delta
  P   PLAIN int64 / int32 values, BOOLEAN PLAIN bits, BOOLEAN RLE runs
  B   DELTA_LENGTH_BYTE_ARRAY (random 4..20-byte values) and DELTA_BYTE_ARRAY (sorted words, shared prefixes):
      value bytes + uint32 offsets out
  W   page-header walk of 64 column chunks x 800 V1 pages (headers with statistics, 20 KB page bodies)
  L   level post-processing over 100 M slots: column entries (validity, offsets, returned definition
      levels) of a REPEATED and of an OPTIONAL column, PackedArray bytes of the levels
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from parquet_go_amd import capi, synth  # noqa: E402

DEV = torch.device("cuda:0")


def timed(fn, stream, steps=10, warmup=2):
    for _ in range(warmup):
        fn()
    stream.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    with torch.cuda.stream(stream):
        evs[0].record()
        for i in range(steps):
            fn()
            evs[i + 1].record()
    stream.synchronize()
    return float(np.median([evs[i].elapsed_time(evs[i + 1]) for i in range(steps)])) / 1e3


def pack_pages(streams):
    lens = np.array([s.size for s in streams], dtype=np.int64)
    offs = np.zeros(len(streams), dtype=np.int64)
    pos = 0
    for i, ln in enumerate(lens):
        offs[i] = pos
        pos += (int(ln) + 15) // 16 * 16
    src = np.zeros(pos + 16, dtype=np.uint8)
    for i, s in enumerate(streams):
        src[offs[i]: offs[i] + lens[i]] = s
    return src, offs, lens


def report(name, in_bytes, out_bytes, t, extra=None):
    d = {"config": name, "ms": round(t * 1e3, 4), "out_GBps": round(out_bytes / t / 1e9, 1),
         "alg_GBps": round((in_bytes + out_bytes) / t / 1e9, 1),
         "frac_of_8TBps": round((in_bytes + out_bytes) / t / 8e12, 4)}
    if extra:
        d.update(extra)
    print(json.dumps(d), flush=True)


def runny(rng, n, hi, mean_run):
    out = np.empty(n, dtype=np.int64)
    i = 0
    while i < n:
        ln = int(rng.geometric(1.0 / mean_run))
        out[i:i + ln] = rng.integers(0, hi)
        i += ln
    return out


def bench_dict(ctx, stream, pages, n, run_heavy, label, distinct_cap=None):
    rng = np.random.default_rng(1)
    d = synth.splitmix64(0xD1C7, 4096).astype(np.uint32).view(np.int32)
    streams, keys_all = [], []
    distinct = min(pages, (200 if run_heavy else pages) if distinct_cap is None else distinct_cap)
    for p in range(distinct):
        keys = runny(rng, n, 4096, 32) if run_heavy else rng.integers(0, 4096, size=n)
        keys_all.append(keys)
        streams.append(np.concatenate([[12], synth.hybrid_encode(keys, 12) if run_heavy
                                       else synth.hybrid_encode_bitpacked(keys, 12)]).astype(np.uint8))
    while len(streams) < pages:   # run-heavy pages are slow to synthesise: reuse them
        streams.append(streams[len(streams) % distinct])
        keys_all.append(keys_all[len(keys_all) % distinct])
    src, offs, lens = pack_pages(streams)
    descs = np.zeros(pages, dtype=capi.DESC_DTYPE)
    descs["kind"] = capi.KIND_DICT
    descs["src_off"], descs["src_len"], descs["n_values"] = offs, lens, n
    descs["dst_off"] = np.arange(pages, dtype=np.uint64) * np.uint64(4 * n)
    descs["elem_size"], descs["dict_len"] = 4, 4096
    t_src = torch.from_numpy(src).to(DEV)
    t_dict = torch.from_numpy(d.copy()).to(DEV)
    t_out = torch.empty(pages * n, dtype=torch.int32, device=DEV)
    plan = ctx.plan(descs)
    fn = lambda: plan.run(t_src.data_ptr(), src.size, t_out.data_ptr(), pages * n * 4, 0, 0,
                          t_dict.data_ptr(), 4096 * 4)
    t = timed(fn, stream)
    res = plan.results()
    assert (res["status"] == 0).all() and (res["n_decoded"] == n).all()
    got = t_out.cpu().numpy().reshape(pages, n)
    for p in (0, pages // 2, pages - 1):
        assert np.array_equal(got[p], d[keys_all[p]])
    report(label, int(lens.sum()), pages * n * 4, t, {"pages": pages, "values_per_page": n})
    plan.close()


def bitunpack_differs(t_src, t_out, w, bits):
    """How many values of t_out differ from the closed form of the little-endian bit string in t_src
    (value i = bits [i w, (i + 1) w)), computed with torch on the device, 16 M values at a time."""
    n = t_out.numel()
    step = 1 << 24
    bad = 0
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        bit = torch.arange(lo, hi, dtype=torch.int64, device=DEV) * w
        byte, sh = bit >> 3, bit & 7
        lo56 = torch.zeros(hi - lo, dtype=torch.int64, device=DEV)     # bytes 0..6 of the window
        for b in range(7):
            lo56 |= t_src[byte + b].to(torch.int64) << (8 * b)
        v = lo56 >> sh
        if w > 48:                                                     # up to 7 + 64 bits: two more bytes
            hi16 = t_src[byte + 7].to(torch.int64) | (t_src[byte + 8].to(torch.int64) << 8)
            v = v | (hi16 << (56 - sh))                                # wraps mod 2^64, as wanted
        if w < 64:
            v = v & ((1 << w) - 1)
        if bits == 32 and w == 32:
            v = (v ^ (1 << 31)) - (1 << 31)                            # as int32
        bad += int((t_out[lo:hi].to(torch.int64) != v).sum().item())
    return bad


def bench_bitunpack(ctx, stream, bits, widths, enc_bytes, out_cap=2 << 30, verify=False):
    rng = np.random.default_rng(4)
    src = rng.integers(0, 256, size=enc_bytes + 64, dtype=np.uint8)
    t_src = torch.from_numpy(src).to(DEV)
    es = bits // 8
    for w in widths:
        ng = min(enc_bytes // w, out_cap // (8 * es))     # cap the output
        t_out = torch.empty(ng * 8, dtype=torch.int32 if bits == 32 else torch.int64, device=DEV)
        fn = lambda: ctx.bitunpack_dev(t_src.data_ptr(), ng * w, w, ng, t_out.data_ptr(), bits)
        t = timed(fn, stream, steps=6, warmup=1)
        extra = {"groups": ng, "bits": bits, "width": w, "encoded_bytes": ng * w, "decoded_bytes": ng * 8 * es,
                 "output_capped": ng * w < enc_bytes - w}
        if verify:
            differ = bitunpack_differs(t_src, t_out, w, bits)
            assert differ == 0, ("bitunpack%d w=%d: %d values differ from the closed form" % (bits, w, differ))
            extra["verified"] = "every value equal to the torch closed form"
        report("C3 bitunpack%d w=%d" % (bits, w), ng * w, ng * 8 * es, t, extra)
        del t_out


def bench_nested(ctx, stream, pages, n, distinct=100):
    rng = np.random.default_rng(5)
    d = synth.splitmix64(7, 4096).astype(np.uint32).view(np.int32)
    streams, nn = [], []
    distinct = min(pages, distinct)
    for p in range(distinct):
        rep = (rng.random(n) >= 0.25).astype(np.int64)
        deff = np.where(rng.random(n) < 0.8, 3, rng.integers(0, 3, size=n))
        k = int((deff == 3).sum())
        keys = rng.integers(0, 4096, size=k)
        page = np.concatenate([synth.length_prefixed(synth.hybrid_encode(rep, 1)),
                               synth.length_prefixed(synth.hybrid_encode(deff, 2)),
                               [12], synth.hybrid_encode_bitpacked(keys, 12)]).astype(np.uint8)
        streams.append(page)
        nn.append(k)
    while len(streams) < pages:
        streams.append(streams[len(streams) % distinct])
        nn.append(nn[len(nn) % distinct])
    src, offs, lens = pack_pages(streams)
    descs = np.zeros(pages, dtype=capi.DESC_DTYPE)
    descs["kind"] = capi.KIND_PAGE_V1
    descs["src_off"], descs["src_len"], descs["n_values"] = offs, lens, n
    descs["max_rep"], descs["max_def"] = 1, 3
    descs["value_kind"], descs["elem_size"], descs["dict_len"] = capi.KIND_DICT, 4, 4096
    descs["dst_off"] = np.arange(pages, dtype=np.uint64) * np.uint64(4 * n)
    descs["rep_off"] = np.arange(pages, dtype=np.uint64) * np.uint64(8 * n)
    descs["def_off"] = descs["rep_off"] + np.uint64(4 * n)
    t_src = torch.from_numpy(src).to(DEV)
    t_dict = torch.from_numpy(d.copy()).to(DEV)
    t_val = torch.empty(pages * n, dtype=torch.int32, device=DEV)
    t_lvl = torch.empty(pages * n * 2, dtype=torch.int32, device=DEV)
    plan = ctx.plan(descs)
    fn = lambda: plan.run(t_src.data_ptr(), src.size, t_val.data_ptr(), pages * n * 4, t_lvl.data_ptr(),
                          pages * n * 8, t_dict.data_ptr(), 4096 * 4)
    t = timed(fn, stream, steps=6, warmup=1)
    res = plan.results()
    assert (res["status"] == 0).all(), res[res["status"] != 0][:5]
    assert (res["not_null"] == np.array(nn)).all()
    out_bytes = pages * n * 8 + 4 * int(np.sum(nn))
    report("C4 nested V1 pages (rep w=1, def w=2, dict w=12)", int(lens.sum()), out_bytes, t,
           {"pages": pages, "slots_per_page": n})
    plan.close()


def bench_levels_stream(ctx, stream, pages, n, max_level, p_max, label, distinct=8):
    """Bare level streams (PQ_KIND_LEVELS, raw): int32 levels out + the count of levels at the maximum."""
    rng = np.random.default_rng(12)
    distinct = min(pages, distinct)
    streams, want = [], []
    for _ in range(distinct):
        lv = np.where(rng.random(n) < p_max, max_level, rng.integers(0, max_level + 1, size=n))
        streams.append(synth.hybrid_encode(lv, int(max_level).bit_length()))
        want.append(lv.astype(np.int32))
    while len(streams) < pages:
        streams.append(streams[len(streams) % distinct])
        want.append(want[len(want) % distinct])
    src, offs, lens = pack_pages(streams)
    descs = np.zeros(pages, dtype=capi.DESC_DTYPE)
    descs["kind"] = capi.KIND_LEVELS
    descs["src_off"], descs["src_len"], descs["n_values"], descs["max_def"] = offs, lens, n, max_level
    descs["def_off"] = np.arange(pages, dtype=np.uint64) * np.uint64(4 * n)
    t_src = torch.from_numpy(src).to(DEV)
    t_lvl = torch.empty(pages * n, dtype=torch.int32, device=DEV)
    plan = ctx.plan(descs)
    fn = lambda: plan.run(t_src.data_ptr(), src.size, 0, 0, t_lvl.data_ptr(), pages * n * 4)
    t = timed(fn, stream, steps=6, warmup=1)
    res = plan.results()
    assert (res["status"] == 0).all() and (res["n_decoded"] == n).all()
    got = t_lvl.view(pages, n)
    for p in (0, pages // 2, pages - 1):
        assert np.array_equal(got[p].cpu().numpy(), want[p])
        assert int(res[p]["not_null"]) == int((want[p] == max_level).sum())
    report(label, int(lens.sum()), pages * n * 4, t, {"pages": pages, "slots_per_page": n})
    plan.close()


def bench_delta(ctx, stream, pages, n, bits, width, label="", as_pages=False, stride=-5):
    """as_pages: the same streams as PQ_KIND_PAGE_V1 descriptors of a flat REQUIRED column (what
    pq_chunks_page_descs makes of a walked chunk) instead of bare PQ_KIND_DELTA32/64.
    stride: added to every delta (the blocks' minDelta; 2^27 and more take five varint bytes)."""
    rng = np.random.default_rng(3)
    dt = np.int64 if bits == 64 else np.int32
    streams, keep = [], {}
    for p in range(pages):
        steps = rng.integers(0, 1 << width, size=n - 1, dtype=np.int64) + stride
        v = np.concatenate([[p * 7], p * 7 + np.cumsum(steps)]).astype(dt)
        if p in (0, pages // 2, pages - 1):
            keep[p] = v
        streams.append(synth.delta_encode(v, bits))
    src, offs, lens = pack_pages(streams)
    descs = np.zeros(pages, dtype=capi.DESC_DTYPE)
    descs["kind"] = capi.KIND_DELTA64 if bits == 64 else capi.KIND_DELTA32
    if as_pages:
        descs["value_kind"] = descs["kind"]
        descs["kind"], descs["flags"] = capi.KIND_PAGE_V1, capi.FLAG_NO_LEVEL_OUTPUT
    descs["src_off"], descs["src_len"], descs["n_values"] = offs, lens, n
    es = bits // 8
    descs["dst_off"] = np.arange(pages, dtype=np.uint64) * np.uint64(es * n)
    t_src = torch.from_numpy(src).to(DEV)
    t_out = torch.empty(pages * n * es, dtype=torch.uint8, device=DEV)
    plan = ctx.plan(descs)
    fn = lambda: plan.run(t_src.data_ptr(), src.size, t_out.data_ptr(), pages * n * es)
    t = timed(fn, stream)
    res = plan.results()
    assert (res["status"] == 0).all() and (res["n_decoded"] == n).all()
    got = t_out.view(torch.int64 if bits == 64 else torch.int32).view(pages, n)
    for p, v in keep.items():
        assert np.array_equal(got[p].cpu().numpy(), v), ("values differ", p)
    report("C2x delta%d width %d%s" % (bits, width, label), int(lens.sum()), pages * n * es, t,
           {"pages": pages, "values_per_page": n})
    plan.close()


def bench_plain(ctx, stream, pages, n, kind, es, label):
    """kind: KIND_PLAIN (es bytes per value), KIND_BOOL_PLAIN or KIND_BOOL_RLE (int32 0 / 1 out)."""
    rng = np.random.default_rng(6)
    distinct = min(pages, 64 if n <= 100_000 else 4)
    streams, want = [], []
    for p in range(distinct):
        if kind == capi.KIND_PLAIN:
            s = rng.integers(0, 256, size=n * es, dtype=np.uint8)
            want.append(s.view(np.int32 if es == 4 else np.int64))
        elif kind == capi.KIND_BOOL_PLAIN:
            s = rng.integers(0, 256, size=(n + 7) // 8, dtype=np.uint8)
            want.append(np.unpackbits(s, bitorder="little")[:n].astype(np.int32))
        else:
            bits = runny(rng, n, 2, 24)
            s = synth.length_prefixed(synth.hybrid_encode(bits, 1)).astype(np.uint8)
            want.append(bits.astype(np.int32))
        streams.append(s)
    while len(streams) < pages:
        streams.append(streams[len(streams) % distinct])
        want.append(want[len(want) % distinct])
    src, offs, lens = pack_pages(streams)
    oes = es if kind == capi.KIND_PLAIN else 4
    descs = np.zeros(pages, dtype=capi.DESC_DTYPE)
    descs["kind"] = kind
    descs["src_off"], descs["src_len"], descs["n_values"] = offs, lens, n
    descs["elem_size"] = es
    descs["dst_off"] = np.arange(pages, dtype=np.uint64) * np.uint64(oes * n)
    t_src = torch.from_numpy(src).to(DEV)
    t_out = torch.empty(pages * n * oes, dtype=torch.uint8, device=DEV)
    plan = ctx.plan(descs)
    fn = lambda: plan.run(t_src.data_ptr(), src.size, t_out.data_ptr(), pages * n * oes)
    t = timed(fn, stream)
    res = plan.results()
    assert (res["status"] == 0).all() and (res["n_decoded"] == n).all()
    got = t_out.cpu().numpy().view(np.int64 if oes == 8 else np.int32).reshape(pages, n)
    for p in (0, pages // 2, pages - 1):
        assert np.array_equal(got[p], want[p])
    report(label, int(lens.sum()), pages * n * oes, t, {"pages": pages, "values_per_page": n})
    plan.close()


def bench_byte_array(ctx, stream, pages, n, front, pull=None):
    """front: DELTA_BYTE_ARRAY over sorted 3..13-letter words; else DELTA_LENGTH_BYTE_ARRAY over random bytes.
    pull: values asked of each page (default all); 0 times Init alone, i.e. the two length streams."""
    rng = np.random.default_rng(8)
    distinct = min(pages, 8)
    streams, flats, offs_want = [], [], []
    for p in range(distinct):
        if front:
            vals = sorted(bytes(rng.integers(97, 123, size=int(rng.integers(3, 14)), dtype=np.uint8)) for _ in range(n))
            streams.append(synth.delta_ba_encode(vals))
        else:
            vals = [bytes(rng.integers(0, 256, size=int(rng.integers(4, 21)), dtype=np.uint8)) for _ in range(n)]
            streams.append(synth.delta_length_ba_encode(vals))
        flats.append(np.frombuffer(b"".join(vals), dtype=np.uint8))
        offs_want.append(np.concatenate([[0], np.cumsum([len(v) for v in vals])]).astype(np.uint32))
    room = (max(f.size for f in flats) + 15) // 16 * 16
    if not front:
        room = (max(s.size for s in streams) + 15) // 16 * 16
    while len(streams) < pages:
        streams.append(streams[len(streams) % distinct])
    src, offs, lens = pack_pages(streams)
    area = 4 * ((2 if front else 1) * n + 1)
    area = (area + 15) // 16 * 16
    descs = np.zeros(pages, dtype=capi.DESC_DTYPE)
    descs["kind"] = capi.KIND_DELTA_BA if front else capi.KIND_DELTA_LENGTH_BA
    descs["src_off"], descs["src_len"], descs["n_values"] = offs, lens, n if pull is None else pull
    descs["dst_off"] = np.arange(pages, dtype=np.uint64) * np.uint64(room + area)
    descs["dict_off"] = descs["dst_off"] + np.uint64(room)
    descs["dict_len"] = n
    descs["dst_cap"] = room if front else 0
    t_src = torch.from_numpy(src).to(DEV)
    t_out = torch.empty(pages * (room + area), dtype=torch.uint8, device=DEV)
    plan = ctx.plan(descs)
    fn = lambda: plan.run(t_src.data_ptr(), src.size, t_out.data_ptr(), t_out.numel())
    t = timed(fn, stream, steps=6, warmup=1)
    res = plan.results()
    if pull is not None:
        report("B %s, Init + %d values per page" % ("DELTA_BYTE_ARRAY" if front else "DELTA_LENGTH_BYTE_ARRAY", pull),
               int(lens.sum()), 0, t, {"pages": pages})
        plan.close()
        return
    assert (res["status"] == 0).all() and (res["n_decoded"] == n).all(), res[:4]
    got = t_out.cpu().numpy().reshape(pages, room + area)
    out_bytes = 0
    for p in range(pages):
        f, o = flats[p % distinct], offs_want[p % distinct]
        out_bytes += f.size + 4 * (n + 1)
        if p in (0, pages // 2, pages - 1):
            assert np.array_equal(got[p, :f.size], f)
            assert np.array_equal(got[p, room: room + 4 * (n + 1)].view(np.uint32), o)
    report("B DELTA_BYTE_ARRAY sorted words" if front else "B DELTA_LENGTH_BYTE_ARRAY 4..20-byte values",
           int(lens.sum()), out_bytes, t, {"pages": pages, "values_per_page": n,
                                           "value_bytes_per_page": int(flats[0].size)})
    plan.close()


def bench_walk(ctx, stream, n_chunks, pages_per_chunk, body, with_statistics=True):
    rng = np.random.default_rng(9)
    heads = [np.frombuffer(synth.page_header(0, body, body, data=(20000, 8, 3, 3, synth.statistics(rng))
                                             if with_statistics else (20000, 8, 3, 3)), dtype=np.uint8)
             for _ in range(16)]
    chunk_bytes = sum(heads[k % 16].size + body for k in range(pages_per_chunk))
    stride = (chunk_bytes + 255) // 256 * 256
    src = np.zeros(n_chunks * stride + 64, dtype=np.uint8)
    for c in range(n_chunks):
        pos = c * stride
        for k in range(pages_per_chunk):
            h = heads[k % 16]
            src[pos: pos + h.size] = h
            pos += h.size + body
    chunks = np.zeros(n_chunks, dtype=capi.CHUNK_DTYPE)
    chunks["src_off"] = np.arange(n_chunks, dtype=np.uint64) * np.uint64(stride)
    chunks["src_len"] = chunk_bytes
    chunks["total_compressed_size"] = chunk_bytes
    chunks["data_page_offset"] = 4
    chunks["physical_type"], chunks["type_length"] = 1, -1
    chunks["info_first"] = np.arange(n_chunks, dtype=np.uint32) * pages_per_chunk
    chunks["info_cap"] = pages_per_chunk
    t_src = torch.from_numpy(src).to(DEV)
    t_infos = torch.zeros(n_chunks * pages_per_chunk * capi.PAGE_INFO_DTYPE.itemsize, dtype=torch.uint8, device=DEV)
    times = []
    for _ in range(5):
        t0 = time.perf_counter()
        res = ctx.walk_chunks_dev(t_src.data_ptr(), src.size, chunks, t_infos.data_ptr(), n_chunks * pages_per_chunk)
        times.append(time.perf_counter() - t0)     # the call waits for the stream: upload, kernel, results
    assert (res["status"] == 0).all() and (res["n_pages"] == pages_per_chunk).all()
    infos = t_infos.cpu().numpy().view(capi.PAGE_INFO_DTYPE)
    assert (infos["num_values"] == 20000).all() and (infos["compressed_size"] == body).all()
    t = float(np.median(times[1:]))
    print(json.dumps({"config": "W page-header walk", "chunks": n_chunks, "pages_per_chunk": pages_per_chunk, "body": body,
                      "ms_per_call": round(t * 1e3, 3), "pages_per_s": round(n_chunks * pages_per_chunk / t),
                      "us_per_page_of_a_chunk": round(t * 1e6 / pages_per_chunk, 3),
                      "header_bytes": int(np.mean([h.size for h in heads]))}), flush=True)


def bench_levels(ctx, stream, n):
    g = torch.Generator(device=DEV)
    g.manual_seed(7)
    with torch.cuda.stream(stream):
        u = torch.rand(n, device=DEV, generator=g)
        rep = (u < 0.75).to(torch.int32)                       # lists of 4 on average
        u = torch.rand(n, device=DEV, generator=g)
        deff = torch.where((u < 0.9) | (rep == 1), 3, (u * 29).to(torch.int32) % 3).to(torch.int32)
        del u
        offsets = torch.empty(n + 1, dtype=torch.int32, device=DEV)
        validity = torch.empty((n + 31) // 32, dtype=torch.int32, device=DEV)
        edef = torch.empty(n, dtype=torch.int32, device=DEV)
        res = torch.zeros(4, dtype=torch.int32, device=DEV)
        packed = torch.empty((n + 7) // 8 * 2 + 16, dtype=torch.uint8, device=DEV)
    stream.synchronize()
    for repeated, label in ((True, "REPEATED (max_rep 1, max_def 3)"), (False, "OPTIONAL (max_def 3)")):
        fn = lambda: ctx.column_entries_dev(deff.data_ptr(), rep.data_ptr(), n, 3, 1, repeated, offsets.data_ptr(),
                                            validity.data_ptr(), edef.data_ptr(), res.data_ptr())
        t = timed(fn, stream, steps=6, warmup=1)
        _, ne, nv, _ = res.cpu().tolist()
        report("L column entries, %s" % label, n * (8 if repeated else 4), ne * 8 + 4 + (ne + 7) // 8, t,
               {"slots": n, "entries": ne, "values": nv})
    fn = lambda: ctx.levels_pack_dev(deff.data_ptr(), n, 2, packed.data_ptr(), packed.numel())
    t = timed(fn, stream, steps=6, warmup=1)
    report("L PackedArray bytes of int32 levels, width 2", n * 4, (n + 7) // 8 * 2, t, {"slots": n})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--pages", type=int, default=2000)
    ap.add_argument("--seg-bytes", type=int, default=0, help="PQ_OPT_SEGMENT_BYTES for every plan (0: the library chooses)")
    args = ap.parse_args()
    stream = torch.cuda.Stream(device=DEV)
    ctx = capi.Context(0, stream.cuda_stream)
    if args.seg_bytes:
        ctx.set_option(capi.OPT_SEGMENT_BYTES, args.seg_bytes)
    want = lambda k: (not args.only) or k in args.only.split(",")
    t0 = time.time()
    if want("c1"):
        bench_dict(ctx, stream, args.pages, 20000, False, "C1 dict w=12 uniform keys")
        bench_dict(ctx, stream, 1, 1000000, False, "C1 dict w=12, one 1M-value page")
    if want("c1b"):
        bench_dict(ctx, stream, args.pages, 20000, True, "C1b dict w=12 run-heavy keys (mean run 32)")
    if want("c3"):
        bench_bitunpack(ctx, stream, 32, [1, 4, 8, 12, 16, 24, 32], 256 << 20)
        bench_bitunpack(ctx, stream, 64, [8, 16, 33, 48, 64], 256 << 20)
    if args.only and want("c3full"):   # every width of both tables, 1 GiB encoded (output capped at 16 GiB)
        bench_bitunpack(ctx, stream, 32, list(range(1, 33)), 1 << 30, out_cap=16 << 30, verify=True)
        bench_bitunpack(ctx, stream, 64, list(range(1, 65)), 1 << 30, out_cap=16 << 30, verify=True)
    if want("c4"):
        bench_nested(ctx, stream, args.pages, 20000)
    if args.only and want("c4big"):   # the large-page variant: 500 pages x 1 M slots (10 distinct pages)
        bench_nested(ctx, stream, 500, 1_000_000, distinct=10)
    if args.only and want("hybig"):   # few large pages of run-dense hybrid streams: cut into segments against one wave a page
        for seg in (1024, 0):
            ctx.set_option(capi.OPT_SEGMENT_MAX_PAGES, seg)
            lab = ", segmented" if seg else ", one wave per page"
            bench_levels_stream(ctx, stream, 64, 1_000_000, 3, 0.8, "L def levels (max 3), 64 x 1 M slots" + lab)
            bench_levels_stream(ctx, stream, 64, 1_000_000, 1, 0.75, "L rep levels (max 1), 64 x 1 M slots" + lab)
            bench_dict(ctx, stream, 64, 1_000_000, True, "C1b dict w=12 run-heavy keys, 64 x 1 M values" + lab, distinct_cap=4)
            bench_plain(ctx, stream, 64, 1_000_000, capi.KIND_BOOL_RLE, 4, "P BOOLEAN RLE (mean run 24), 64 x 1 M values" + lab)
        ctx.set_option(capi.OPT_SEGMENT_MAX_PAGES, 1024)
    if want("c2x"):
        for w in (4, 8, 24, 30, 32):
            bench_delta(ctx, stream, args.pages, 20000, 64, w)
        bench_delta(ctx, stream, args.pages, 20000, 32, 12)
    if want("c2big"):    # the large-page variant: one wave per page leaves most of the machine idle
        bench_delta(ctx, stream, 100, 1_000_000, 64, 16)
        bench_delta(ctx, stream, 1000, 100_000, 64, 16)
        ctx.set_option(capi.OPT_DELTA_WAVE_MIN_PAGES, 1 << 30)    # the workgroup-per-page kernel instead
        bench_delta(ctx, stream, 100, 1_000_000, 64, 16)
        bench_delta(ctx, stream, 1000, 100_000, 64, 16)
        ctx.set_option(capi.OPT_DELTA_WAVE_MIN_PAGES, 1)
    if want("c2seg"):    # large pages cut into segments (the default) against one wave per page
        for seg in (1024, 0):
            ctx.set_option(capi.OPT_SEGMENT_MAX_PAGES, seg)
            lab = ", segmented" if seg else ", one wave per page"
            bench_delta(ctx, stream, 1, 1_000_000, 64, 16, lab)
            bench_delta(ctx, stream, 100, 1_000_000, 64, 16, lab)
            bench_delta(ctx, stream, 1000, 100_000, 64, 16, lab)
            bench_delta(ctx, stream, 100, 1_000_000, 32, 12, lab)
        ctx.set_option(capi.OPT_SEGMENT_MAX_PAGES, 1024)
    if args.only and want("c2stride"):   # blocks whose minDelta takes five varint bytes and more (nanosecond timestamps)
        bench_delta(ctx, stream, 100, 1_000_000, 64, 16, ", minDelta 10^9 (5 varint bytes), segmented", stride=10**9)
        bench_delta(ctx, stream, 8000, 20_000, 64, 16, ", minDelta 10^9 (5 varint bytes), 8000 pages", stride=10**9)
    if args.only and want("c2pages"):   # a parquet-mr style row group: 16 columns x 8 pages x 131 072 int64
        bench_delta(ctx, stream, 128, 131072, 64, 16, ", bare DELTA64 descriptors")
        bench_delta(ctx, stream, 128, 131072, 64, 16, ", PAGE_V1 descriptors of a flat column", as_pages=True)
        bench_delta(ctx, stream, 128, 131072, 32, 12, ", PAGE_V1 descriptors of a flat column", as_pages=True)
    if want("plain"):
        bench_plain(ctx, stream, args.pages, 20000, capi.KIND_PLAIN, 8, "P PLAIN int64")
        bench_plain(ctx, stream, args.pages, 20000, capi.KIND_PLAIN, 4, "P PLAIN int32")
        bench_plain(ctx, stream, args.pages, 20000, capi.KIND_BOOL_PLAIN, 4, "P BOOLEAN PLAIN")
        bench_plain(ctx, stream, args.pages, 20000, capi.KIND_BOOL_RLE, 4, "P BOOLEAN RLE (mean run 24)")
    if want("bytes"):
        bench_byte_array(ctx, stream, args.pages, 20000, False)
        bench_byte_array(ctx, stream, args.pages, 20000, True)
    if args.only and want("dba"):      # DELTA_BYTE_ARRAY alone, for counter passes
        bench_byte_array(ctx, stream, args.pages, 20000, True)
    if want("bytes_init"):
        bench_byte_array(ctx, stream, args.pages, 20000, False, pull=0)
        bench_byte_array(ctx, stream, args.pages, 20000, True, pull=0)
    if want("walk"):
        bench_walk(ctx, stream, 64, 800, 20000)
        bench_walk(ctx, stream, 1024, 100, 20000)
        bench_walk(ctx, stream, 64, 800, 20000, with_statistics=False)
    if want("walk_probe"):   # bodies so short that several headers share a window: the parse without the memory misses
        bench_walk(ctx, stream, 64, 800, 64)
        bench_walk(ctx, stream, 64, 800, 64, with_statistics=False)
    if want("levels"):
        bench_levels(ctx, stream, 100_000_000)
    print("elapsed %.1fs" % (time.time() - t0), file=sys.stderr)
    ctx.close()


if __name__ == "__main__":
    main()
