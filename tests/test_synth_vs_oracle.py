"""The numpy page writers (parquet_go_amd.synth) must emit exactly what the oracle's
restated reference encoders emit, and the oracle's decoders must read it back."""
import numpy as np
import pytest

import oracle_lib as O
from parquet_go_amd import synth


def _rng(seed):
    return np.random.default_rng(seed)


@pytest.mark.parametrize("w", [0, 1, 2, 3, 5, 7, 8, 9, 12, 16, 17, 24, 31, 32])
def test_pack_bits_matches_pack8(w):
    rng = _rng(w)
    vals = rng.integers(0, 1 << w, size=8 * 37, dtype=np.uint64) if w else np.zeros(8 * 37, np.uint64)
    got = synth.pack_bits(vals, w)
    want = np.concatenate([O.pack8_32(vals[g * 8:(g + 1) * 8].astype(np.uint32).astype(np.int32), w)
                           for g in range(37)]) if w else np.zeros(0, np.uint8)
    assert got.tolist() == want.tolist()


@pytest.mark.parametrize("w", [33, 40, 47, 63, 64])
def test_pack_bits_matches_pack8_64(w):
    rng = _rng(w)
    vals = rng.integers(0, 1 << 63, size=8 * 19, dtype=np.uint64)
    if w < 64:
        vals &= np.uint64((1 << w) - 1)
    got = synth.pack_bits(vals, w)
    want = np.concatenate([O.pack8_64(vals[g * 8:(g + 1) * 8].astype(np.int64), w) for g in range(19)])
    assert got.tolist() == want.tolist()


def _runny(rng, n, w, mean_run):
    """values with geometric run lengths"""
    out = np.empty(n, dtype=np.int64)
    i = 0
    hi = max(1 << w, 1)
    while i < n:
        ln = int(rng.geometric(1.0 / mean_run))
        out[i:i + ln] = rng.integers(0, hi)
        i += ln
    return out


@pytest.mark.parametrize("w,mean_run,n", [(1, 3, 5000), (2, 6, 5000), (3, 1, 1000), (3, 12, 4003),
                                          (9, 32, 20000), (12, 1, 20000), (12, 40, 7777),
                                          (17, 9, 3001), (32, 20, 2048), (0, 1, 100), (5, 1, 9),
                                          (5, 100, 1), (4, 8, 8), (4, 8, 7), (4, 200, 1000)])
def test_hybrid_writer_equals_oracle_encoder(w, mean_run, n):
    rng = _rng(100 + w + n)
    vals = _runny(rng, n, w, mean_run) if w else np.zeros(n, np.int64)
    got = synth.hybrid_encode(vals, w)
    want = O.hybrid_encode(vals.astype(np.uint32).astype(np.int32), w)
    assert got.tolist() == want.tolist()
    st, dec, _ = O.hybrid_decode(got, w, n)
    assert st == O.OK
    assert dec.astype(np.uint32).tolist() == vals.astype(np.uint32).tolist()


def test_hybrid_bitpacked_fast_path_equals_general():
    rng = _rng(5)
    for n in (1, 8, 503, 504, 505, 20000):
        vals = rng.integers(0, 4096, size=n)
        # make sure no aligned group is constant
        assert synth.hybrid_encode_bitpacked(vals, 12).tolist() == synth.hybrid_encode(vals, 12).tolist()


@pytest.mark.parametrize("bits", [32, 64])
@pytest.mark.parametrize("n", [0, 1, 2, 8, 9, 128, 129, 130, 257, 1000, 20000])
@pytest.mark.parametrize("block,mb", [(128, 4), (256, 8), (128, 1), (512, 4)])
def test_delta_writer_equals_oracle_encoder(bits, n, block, mb):
    rng = _rng(n + bits + block)
    dt = np.int32 if bits == 32 else np.int64
    if bits == 32:
        vals = rng.integers(-2**31, 2**31, size=n, dtype=np.int64).astype(np.int32)
        vals[: n // 2] = np.cumsum(rng.integers(-5, 300, size=n // 2)).astype(np.int32)
    else:
        vals = np.cumsum(rng.integers(-1000, 1 << 20, size=n)).astype(np.int64)
        if n > 300:
            vals[200:300] = rng.integers(-2**63, 2**63 - 1, size=100, dtype=np.int64)
    got = synth.delta_encode(vals, bits, block, mb)
    rc, want = (O.delta_encode32 if bits == 32 else O.delta_encode64)(vals.astype(dt), block, mb)
    assert rc == 0
    assert got.tolist() == want.tolist()


def test_splitmix64_known_answer():
    # first outputs for seed 1234567 (reference implementation by Sebastiano Vigna)
    got = synth.splitmix64(1234567, 5).tolist()
    assert got == [6457827717110365317, 3203168211198807973, 9817491932198370423,
                   4593380528125082431, 16408922859458223821]
