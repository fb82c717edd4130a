"""GPU parity at the full workload sizes, through properties that do not need the CPU oracle
to process the whole input:

* bit-unpack of 1 GiB of encoded bytes per width against the closed form of the encoding
  (value i = bits [i w, (i + 1) w) of the little-endian bit string), evaluated with torch
  integer ops on the device — an implementation independent of the kernels;
* the 100 M-value DELTA_BINARY_PACKED column of the headline benchmark: decode(encode(x)) == x;
* 500 M level slots of nested V1 pages: 100 distinct pages are checked against the oracle, the
  25 000 pages of the full run are replicas of them and must decode to identical bytes.
"""
import numpy as np
import pytest

import oracle_lib as O
from parquet_go_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def _closed_form(t_src, w, n_values, bits, torch):
    """Values of the little-endian bit string in t_src (uint8 on the device) as int64."""
    dev = t_src.device
    out = torch.empty(n_values, dtype=torch.int64, device=dev)
    step = 1 << 24
    src64 = torch.cat([t_src, torch.zeros(16, dtype=torch.uint8, device=dev)]).to(torch.int64)
    for lo in range(0, n_values, step):
        hi = min(n_values, lo + step)
        bit = torch.arange(lo, hi, dtype=torch.int64, device=dev) * w
        byte, sh = bit >> 3, bit & 7
        lo56 = torch.zeros(hi - lo, dtype=torch.int64, device=dev)     # bytes 0..6 of the window
        for b in range(7):
            lo56 |= src64[byte + b] << (8 * b)
        v = lo56 >> sh                                                 # bits sh .. 55 of the window
        if w > 48:                                                     # up to 7 + 64 bits: two more bytes
            hi16 = src64[byte + 7] | (src64[byte + 8] << 8)
            v = v | (hi16 << (56 - sh))                                # wraps mod 2^64, as wanted
        if w < 64:
            v = v & ((1 << w) - 1)
        if bits == 32 and w == 32:
            v = (v ^ (1 << 31)) - (1 << 31)                            # as int32
        out[lo:hi] = v
    return out


# every residue of the width mod 8 in both tables (9..16, 33..40), the ends of both, and a few more
@pytest.mark.parametrize("bits,w", [(32, w) for w in (1, 9, 10, 11, 12, 13, 14, 15, 16, 31, 32)] +
                         [(64, w) for w in (7, 33, 34, 35, 36, 37, 38, 39, 40, 47, 57, 64)])
def test_bitunpack_one_gib_encoded(ctx, bits, w):
    import torch
    dev = torch.device("cuda:0")
    enc = 1 << 30
    es = bits // 8
    n_groups = min(enc // w, (6 << 30) // (8 * es))          # cap the output at 6 GiB
    g = torch.Generator(device=dev)
    g.manual_seed(4 + w)
    t_src = torch.randint(0, 256, (n_groups * w + 64,), dtype=torch.uint8, device=dev, generator=g)
    t_dst = torch.empty(n_groups * 8, dtype=torch.int32 if bits == 32 else torch.int64, device=dev)
    torch.cuda.synchronize()   # the context launches on its own stream: torch's fills must have landed
    ctx.bitunpack_dev(t_src.data_ptr(), n_groups * w, w, n_groups, t_dst.data_ptr(), bits)
    ctx.synchronize()
    # the closed form needs 8-byte temporaries: check a prefix, a suffix and strided slices
    # of 64 Mi values each, which covers every alignment of the tiling
    n = n_groups * 8
    span = 1 << 26
    for start in sorted({0, max(0, n - span), (n // 2 // 8) * 8, (n // 3 // 8) * 8}):
        cnt = min(span, n - start)
        first_byte = start * w // 8          # start is a multiple of 8 values = whole groups
        want = _closed_form(t_src[first_byte: first_byte + cnt * w // 8 + 16], w, cnt, bits, torch)
        got = t_dst[start: start + cnt].to(torch.int64)
        assert torch.equal(got, want), (bits, w, start)
        del want, got
    # and the reference's own 8-value function on a few groups (CPU oracle)
    for gi in (0, n_groups // 2, n_groups - 1):
        grp = t_src[gi * w: gi * w + w].cpu().numpy()
        ref = (O.bitunpack32 if bits == 32 else O.bitunpack64)(grp, w, 1)
        assert np.array_equal(t_dst[gi * 8: gi * 8 + 8].cpu().numpy(), ref)


def test_delta_100m_values_round_trip(ctx):
    import torch
    import bench
    dev = torch.device("cuda:0")
    pages, n = 5000, 20000
    src, offs, lens, expected = bench.make_pages(pages, n, 16, seed=0xC2)
    descs = np.zeros(pages, dtype=capi.DESC_DTYPE)
    descs["kind"] = capi.KIND_DELTA64
    descs["src_off"], descs["src_len"], descs["n_values"] = offs, lens, n
    descs["dst_off"] = np.arange(pages, dtype=np.uint64) * np.uint64(8 * n)
    t_src = torch.from_numpy(src).to(dev)
    t_out = torch.full((pages * n,), -1, dtype=torch.int64, device=dev)
    plan = ctx.plan(descs)
    torch.cuda.synchronize()
    plan.run(t_src.data_ptr(), src.size, t_out.data_ptr(), pages * n * 8)
    res = plan.results()
    assert (res["status"] == 0).all() and (res["n_decoded"] == n).all()
    assert (res["consumed"] == lens).all()
    assert torch.equal(t_out.view(pages, n), torch.from_numpy(expected).to(dev))
    # the same pages through the other kernel selection
    ctx.set_option(capi.OPT_DELTA_WAVE_MIN_PAGES, 1 << 30)
    try:
        t_out.fill_(-1)
        torch.cuda.synchronize()
        plan2 = ctx.plan(descs[:500])
        plan2.run(t_src.data_ptr(), src.size, t_out.data_ptr(), pages * n * 8)
        assert (plan2.results()["status"] == 0).all()
        assert torch.equal(t_out[: 500 * n].view(500, n), torch.from_numpy(expected[:500]).to(dev))
        plan2.close()
    finally:
        ctx.set_option(capi.OPT_DELTA_WAVE_MIN_PAGES, 1)
    # a few pages against the oracle too
    for p in (0, pages // 2, pages - 1):
        r = O.delta_decode64(src[offs[p]: offs[p] + lens[p]], n)
        assert r["status"] == 0 and np.array_equal(r["values"], expected[p])
    plan.close()


def test_nested_pages_500m_slots(ctx):
    import torch
    dev = torch.device("cuda:0")
    distinct, n, replicas = 100, 20000, 250          # 25 000 pages x 20 000 slots = 500 M slots
    rng = np.random.default_rng(5)
    d = synth.splitmix64(7, 4096).astype(np.uint32).view(np.int32)
    blocks = []
    for p in range(distinct):
        rep = (rng.random(n) >= 0.25).astype(np.int64)
        deff = np.where(rng.random(n) < 0.8, 3, rng.integers(0, 3, size=n))
        keys = rng.integers(0, 4096, size=int((deff == 3).sum()))
        blocks.append(np.concatenate([synth.length_prefixed(synth.hybrid_encode(rep, 1)),
                                      synth.length_prefixed(synth.hybrid_encode(deff, 2)),
                                      [12], synth.hybrid_encode_bitpacked(keys, 12)]).astype(np.uint8))
    lens = np.array([b.size for b in blocks], dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum((lens + 15) // 16 * 16)])[:-1]
    src = np.zeros(int(offs[-1] + lens[-1] + 16), dtype=np.uint8)
    for b, o in zip(blocks, offs):
        src[o: o + b.size] = b
    # the oracle on the distinct pages
    want = []
    for p in range(distinct):
        o = O.page_read(blocks[p], version=1, num_values=n, max_rep=1, max_def=3, value_kind=O.VALUES_DICT,
                        dictionary=d)
        assert o["status"] == 0
        want.append(o)
    pages = distinct * replicas
    descs = np.zeros(pages, dtype=capi.DESC_DTYPE)
    descs["kind"] = capi.KIND_PAGE_V1
    descs["src_off"] = np.tile(offs, replicas)
    descs["src_len"] = np.tile(lens, replicas)
    descs["n_values"] = n
    descs["max_rep"], descs["max_def"] = 1, 3
    descs["value_kind"], descs["elem_size"], descs["dict_len"] = capi.KIND_DICT, 4, 4096
    idx = np.arange(pages, dtype=np.uint64)
    descs["dst_off"] = idx * np.uint64(4 * n)
    descs["rep_off"] = idx * np.uint64(8 * n)
    descs["def_off"] = descs["rep_off"] + np.uint64(4 * n)
    t_src = torch.from_numpy(src).to(dev)
    t_dict = torch.from_numpy(d.copy()).to(dev)
    t_val = torch.zeros(pages * n, dtype=torch.int32, device=dev)
    t_lvl = torch.empty(pages * n * 2, dtype=torch.int32, device=dev)
    plan = ctx.plan(descs)
    torch.cuda.synchronize()
    plan.run(t_src.data_ptr(), src.size, t_val.data_ptr(), pages * n * 4, t_lvl.data_ptr(), pages * n * 8,
             t_dict.data_ptr(), 4096 * 4)
    res = plan.results()
    assert (res["status"] == 0).all()
    assert (res["n_levels"] == n).all()
    nn = np.array([w["not_null"] for w in want])
    assert (res["not_null"] == np.tile(nn, replicas)).all()
    lv = t_lvl.view(replicas, distinct, 2, n)
    vals = t_val.view(replicas, distinct, n)
    # replica 0 against the oracle
    for p in range(distinct):
        assert np.array_equal(lv[0, p, 0].cpu().numpy(), want[p]["rep"])
        assert np.array_equal(lv[0, p, 1].cpu().numpy(), want[p]["def_"])
        k = int(nn[p])
        assert np.array_equal(vals[0, p, :k].cpu().numpy(), want[p]["values"][:k])
    # every replica against replica 0
    assert torch.equal(lv, lv[0:1].expand_as(lv))
    for p in range(distinct):
        k = int(nn[p])
        assert torch.equal(vals[:, p, :k], vals[0:1, p, :k].expand(replicas, k))
    plan.close()
