"""Level post-processing: PackedArray bytes of a level run, and the entries ColumnStore.Get yields.

CPU tests pin the oracle: the pack against the reference's PackedArray vector
(encoding/packed-array_test.go, tests/golden/packed_array.json) and the bit-packing tables, the
entries against hand-worked columns and a literal Python transcription of the call loop
(datastore/column.go:127-177).  The reference holds no test of ColumnStore.Get: parity of the
entries is pinned by the restated code only.  GPU tests compare the library with the oracle.
"""
import json
import os

import numpy as np
import pytest

import oracle_lib as O


def _get_loop(deff, rep, max_def, max_rep, repeated):
    """ColumnStore.Get, line by line, until "out of range"."""
    n = len(deff)
    read_pos, vals = 0, 0
    offsets, valid, edef = [], [], []
    while read_pos < n:
        dl = deff[read_pos]
        offsets.append(vals)
        if dl < max_def:
            read_pos += 1
            valid.append(0)
            edef.append(dl)
            continue
        vals += 1
        valid.append(1)
        edef.append(max_def)
        if not repeated:
            read_pos += 1
            continue
        while True:
            read_pos += 1
            if read_pos >= n or rep[read_pos] < max_rep:
                break
            vals += 1
    offsets.append(vals)
    return offsets, valid, edef


def _bits(validity, n):
    return np.unpackbits(np.asarray(validity, dtype=np.uint8), bitorder="little")[:n].tolist()


def _check_entries(got, deff, rep, max_def, max_rep, repeated):
    offsets, valid, edef = _get_loop(list(deff), list(rep) if rep is not None else None, max_def, max_rep, repeated)
    assert got["n_entries"] == len(valid)
    assert got["n_values"] == offsets[-1]
    assert got["offsets"].tolist() == offsets
    assert got["entry_def"].tolist() == edef
    assert _bits(got["validity"], len(valid)) == valid
    assert not np.any(np.unpackbits(got["validity"], bitorder="little")[len(valid):])


def _random_levels(rng, n, max_def, max_rep, consistent):
    rep = rng.integers(0, max_rep + 1, size=n) if max_rep else np.zeros(n, np.int64)
    if n:
        rep[0] = 0
    deff = np.where(rng.random(n) < 0.7, max_def, rng.integers(0, max_def + 1, size=n))
    if consistent and max_rep:
        deff = np.where(rep >= max_rep, max_def, deff)   # a continued list element is defined
    return deff.astype(np.int32), rep.astype(np.int32)


# ---------------------------------------------------------------------------
# oracle (CPU)

def test_oracle_pack_matches_packed_array_vector(golden_dir):
    v = json.load(open(os.path.join(golden_dir, "packed_array.json")))
    assert O.levels_pack(v["values"], v["w"]).tolist() == v["bytes"]
    assert O.levels_pack(v["values"] * 2, v["w"]).tolist() == v["bytes"] * 2
    # a short last group is zero-filled by Flush
    short = O.levels_pack(v["values"][:3], v["w"])
    assert short.tolist() == O.pack8_32(np.array(v["values"][:3] + [0] * 5, np.int32), v["w"]).tolist()
    assert O.levels_pack([], v["w"]).size == 0


def test_oracle_pack_is_groupwise_pack8():
    rng = np.random.default_rng(3)
    for w in range(0, 17):
        lv = rng.integers(0, 1 << w, size=8 * 9 + 5).astype(np.int32) if w else np.zeros(77, np.int32)
        got = O.levels_pack(lv, w)
        padded = np.concatenate([lv, np.zeros(3, np.int32)])
        want = np.concatenate([O.pack8_32(padded[i:i + 8], w) for i in range(0, 80, 8)]) if w else np.zeros(0, np.uint8)
        assert got.tolist() == want.tolist(), w


def test_oracle_entries_hand_worked():
    # OPTIONAL scalar: every slot is an entry
    r = O.column_entries([1, 0, 1], None, 1, 0, False)
    assert (r["n_entries"], r["n_values"]) == (3, 2)
    assert r["offsets"].tolist() == [0, 1, 1, 2] and r["entry_def"].tolist() == [1, 0, 1]
    assert _bits(r["validity"], 3) == [1, 0, 1]
    # REPEATED leaf: [a b c] null [d e]
    r = O.column_entries([1, 1, 1, 0, 1, 1], [0, 1, 1, 0, 0, 1], 1, 1, True)
    assert (r["n_entries"], r["n_values"]) == (3, 5)
    assert r["offsets"].tolist() == [0, 3, 3, 5] and r["entry_def"].tolist() == [1, 0, 1]
    assert _bits(r["validity"], 3) == [1, 0, 1]
    # a null never continues: the slot behind it opens an entry whatever its repetition level
    r = O.column_entries([0, 2, 2], [0, 1, 1], 2, 1, True)
    assert r["offsets"].tolist() == [0, 0, 2] and r["entry_def"].tolist() == [0, 2]
    # positions of a value entry take a value whatever their own definition level
    r = O.column_entries([2, 0, 2], [0, 1, 0], 2, 1, True)
    assert r["offsets"].tolist() == [0, 2, 3] and r["n_values"] == 3
    r = O.column_entries([], [], 3, 1, True)
    assert (r["n_entries"], r["n_values"]) == (0, 0) and r["offsets"].tolist() == [0]


@pytest.mark.parametrize("repeated", [False, True])
def test_oracle_entries_vs_call_loop(repeated):
    rng = np.random.default_rng(5 + repeated)
    for n in (1, 2, 63, 64, 65, 200, 3000):
        for max_def, max_rep in ((1, 1), (3, 1), (2, 2), (0, 0), (1, 0), (5, 3)):
            for consistent in (True, False):
                deff, rep = _random_levels(rng, n, max_def, max_rep, consistent)
                got = O.column_entries(deff, rep, max_def, max_rep, repeated)
                _check_entries(got, deff, rep, max_def, max_rep, repeated)


# ---------------------------------------------------------------------------
# library (GPU)

@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available()
    from parquet_go_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


@pytest.mark.gpu
def test_levels_pack_vs_oracle(ctx, golden_dir):
    v = json.load(open(os.path.join(golden_dir, "packed_array.json")))
    assert ctx.levels_pack(v["values"], v["w"]).tolist() == v["bytes"]
    rng = np.random.default_rng(8)
    for w in range(0, 17):
        for n in (0, 1, 7, 8, 9, 31, 32, 33, 255, 256, 257, 8191, 8192, 100003):
            lv = rng.integers(0, 1 << w, size=n).astype(np.int32) if w else np.zeros(n, np.int32)
            assert np.array_equal(ctx.levels_pack(lv, w), O.levels_pack(lv, w)), (w, n)


@pytest.mark.gpu
def test_column_entries_vs_oracle(ctx):
    rng = np.random.default_rng(9)
    for repeated in (False, True):
        for n in (0, 1, 2, 63, 64, 65, 2047, 2048, 2049, 4096, 70001, 300000):
            for max_def, max_rep in ((1, 1), (3, 1), (2, 2), (0, 0), (1, 0), (5, 3)):
                for consistent in (True, False):
                    deff, rep = _random_levels(rng, n, max_def, max_rep, consistent)
                    got = ctx.column_entries(deff, rep if repeated else None, max_def, max_rep, repeated)
                    want = O.column_entries(deff, rep, max_def, max_rep, repeated)
                    assert got["n_entries"] == want["n_entries"], (n, max_def, max_rep, repeated)
                    assert got["n_values"] == want["n_values"]
                    assert np.array_equal(got["offsets"], want["offsets"])
                    assert np.array_equal(got["entry_def"], want["entry_def"])
                    assert np.array_equal(got["validity"], want["validity"])


@pytest.mark.gpu
def test_column_entries_levels_beyond_a_byte(ctx):
    """Definition levels that do not fit a byte (a maximum above 255, stray large or negative
    levels in a column of small ones): the entries without a value return them unchanged."""
    rng = np.random.default_rng(19)
    n = 5 * 8192 + 3000
    for repeated in (False, True):
        for max_def, lo, hi in ((300, 0, 301), (65535, 0, 65536), (3, -4, 4), (3, 0, 4), (200, 0, 260)):
            deff = rng.integers(lo, hi, size=n).astype(np.int32)
            deff[rng.random(n) < 0.6] = max_def
            if max_def == 3 and lo == 0:
                deff[[100, 9000, 20000]] = [256, 1 << 20, -(1 << 31)]   # one stray level in three of the tiles
            rep = (rng.random(n) < 0.7).astype(np.int32)
            got = ctx.column_entries(deff, rep if repeated else None, max_def, 1, repeated)
            want = O.column_entries(deff, rep, max_def, 1, repeated)
            assert got["n_entries"] == want["n_entries"] and got["n_values"] == want["n_values"], (max_def, lo, repeated)
            assert np.array_equal(got["offsets"], want["offsets"])
            assert np.array_equal(got["entry_def"], want["entry_def"]), (max_def, lo, repeated)
            assert np.array_equal(got["validity"], want["validity"])


@pytest.mark.gpu
def test_column_entries_long_chains(ctx):
    """Chains that cross many wave steps and tiles: one list of everything, all nulls, a value
    entry whose continuation positions are undefined (they still take values), alternating."""
    n = 3 * 2048 * 5 + 77
    ones, zeros = np.ones(n, np.int32), np.zeros(n, np.int32)
    alt = (np.arange(n) & 1).astype(np.int32)
    first = zeros.copy()
    first[0] = 1
    cases = [(ones, ones, 1, 1), (zeros, ones, 1, 1), (first, ones, 1, 1), (ones, zeros, 1, 1), (alt, ones, 1, 1),
             (alt, alt, 1, 1), (1 - alt, alt, 1, 1), (ones, ones, 1, 0)]
    for deff, rep, max_def, max_rep in cases:
        got = ctx.column_entries(deff, rep, max_def, max_rep, True)
        want = O.column_entries(deff, rep, max_def, max_rep, True)
        assert got["n_entries"] == want["n_entries"] and got["n_values"] == want["n_values"]
        assert np.array_equal(got["offsets"], want["offsets"])
        assert np.array_equal(got["entry_def"], want["entry_def"])
        assert np.array_equal(got["validity"], want["validity"])


@pytest.mark.gpu
def test_level_assembly_full_size(ctx):
    """100 M slots on the device: entries checked through closed forms of a periodic pattern,
    the pack through unpacking it again."""
    import torch
    from parquet_go_amd import capi
    dev = torch.device("cuda:0")
    n = 100_000_000
    idx = torch.arange(n, device=dev, dtype=torch.int64)
    # lists of 4 values, every fifth entry a null: period 17 = 4 lists x 4 slots + 1 null
    ph = idx % 17
    deff = (ph != 16).to(torch.int32) * 2
    rep = ((ph != 16) & (ph % 4 != 0)).to(torch.int32)
    offsets = torch.empty(n + 1, dtype=torch.int32, device=dev)
    validity = torch.zeros((n + 31) // 32, dtype=torch.int32, device=dev)
    edef = torch.empty(n, dtype=torch.int32, device=dev)
    res = torch.zeros(4, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.column_entries_dev(deff.data_ptr(), rep.data_ptr(), n, 2, 1, True, offsets.data_ptr(), validity.data_ptr(),
                           edef.data_ptr(), res.data_ptr())
    ctx.synchronize()
    status, ne, nv, _ = res.cpu().tolist()
    periods, rem = divmod(n, 17)
    want_ne = periods * 5 + (rem + 3) // 4
    want_nv = periods * 16 + rem
    assert (status, ne, nv) == (0, want_ne, want_nv)
    k = torch.arange(ne + 1, device=dev, dtype=torch.int64)
    want_off = (k // 5) * 16 + torch.minimum(k % 5, torch.tensor(4, device=dev)) * 4
    want_off[ne] = nv
    assert torch.equal(offsets[: ne + 1].to(torch.int64), want_off)
    kk = k[:ne]
    assert torch.equal(edef[:ne], ((kk % 5 != 4).to(torch.int32) * 2))
    vbits = (validity.view(torch.uint8)[: (ne + 7) // 8].to(torch.int32).unsqueeze(1)
             >> torch.arange(8, device=dev, dtype=torch.int32)) & 1
    assert torch.equal(vbits.reshape(-1)[:ne], (kk % 5 != 4).to(torch.int32))
    del offsets, edef, vbits, want_off, k, kk
    # pack def levels (width 2) and unpack them with the library's own bit-unpack
    packed = torch.zeros((n + 7) // 8 * 2 + 16, dtype=torch.uint8, device=dev)
    ctx.levels_pack_dev(deff.data_ptr(), n, 2, packed.data_ptr(), packed.numel())
    back = torch.empty((n + 7) // 8 * 8, dtype=torch.int32, device=dev)
    ctx.bitunpack_dev(packed.data_ptr(), (n + 7) // 8 * 2, 2, (n + 7) // 8, back.data_ptr(), 32)
    ctx.synchronize()
    assert torch.equal(back[:n], deff)
    assert int(back[n:].abs().sum()) == 0
