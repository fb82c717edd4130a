"""CPU: the oracle's DELTA_LENGTH_BYTE_ARRAY / DELTA_BYTE_ARRAY restatement (types/byte-array.go:168-219,
279-348) against a second, independent transcription of the same loops in Python, and against the
values the restated encoders were fed.  The reference holds no test for these decoders: parity
unpinned, pinned only by the restated code and these cross-checks."""
import numpy as np
import pytest

import oracle_lib as O
from byte_array_cases import damaged_dba, damaged_dlba, random_values, total_len
from parquet_go_amd import synth


def _lens(stream):
    """Int32DeltaBPDecoder.Init + decodeInt32 -> (lens or None when Init fails, status, consumed)."""
    s = np.asarray(stream, dtype=np.uint8)
    head = O.delta_decode32(s, 0)
    if head["init_status"] != O.OK:
        return None, head["init_status"], head["consumed"]
    vc = head["values_count"]
    r = O.delta_decode32(s, vc)
    if r["status"] != O.OK or r["n_decoded"] != vc:
        return None, r["status"], r["consumed"]
    return r["values"].astype(np.int64), O.OK, r["consumed"]


def _model_dlba(stream, n):
    """types/byte-array.go:174-219, line by line."""
    s = np.asarray(stream, dtype=np.uint8)
    lens, st, pos = _lens(s)
    if lens is None:
        return dict(status=st, init=st, n=0, values=[])
    out = []
    for i in range(n):
        if i >= len(lens):
            return dict(status=O.EOF, init=O.OK, n=i, values=out, consumed=pos)
        size = int(lens[i])
        if size < 0:
            return dict(status=O.BYTE_ARRAY_NEGATIVE_LEN, init=O.OK, n=i, values=out, consumed=pos)
        if size:
            left = s.size - pos
            if left == 0:
                return dict(status=O.EOF, init=O.OK, n=i, values=out, consumed=pos)
            if left < size:
                return dict(status=O.UNEXPECTED_EOF, init=O.OK, n=i, values=out, consumed=s.size)
        out.append(bytes(s[pos: pos + size]))
        pos += size
    return dict(status=O.OK, init=O.OK, n=n, values=out, consumed=pos)


def _model_dba(stream, n):
    """types/byte-array.go:285-348, line by line (no capacity limit)."""
    s = np.asarray(stream, dtype=np.uint8)
    plens, st, pos = _lens(s)
    if plens is None:
        return dict(status=st, init=st, n=0, values=[])
    slens, st, pos2 = _lens(s[pos:])
    if slens is None:
        return dict(status=st, init=st, n=0, values=[])
    pos += pos2
    if len(plens) != len(slens):
        return dict(status=O.BYTE_ARRAY_COUNT_MISMATCH, init=O.BYTE_ARRAY_COUNT_MISMATCH, n=0, values=[])
    out, prev = [], b""
    for i in range(n):
        if i >= len(slens):
            return dict(status=O.EOF, init=O.OK, n=i, values=out, consumed=pos)
        size = int(slens[i])
        if size < 0:
            return dict(status=O.BYTE_ARRAY_NEGATIVE_LEN, init=O.OK, n=i, values=out, consumed=pos)
        if size:
            left = s.size - pos
            if left == 0:
                return dict(status=O.EOF, init=O.OK, n=i, values=out, consumed=pos)
            if left < size:
                return dict(status=O.UNEXPECTED_EOF, init=O.OK, n=i, values=out, consumed=s.size)
        suffix = bytes(s[pos: pos + size])
        pos += size
        p = int(plens[i])
        if p + size < 0:
            return dict(status=O.BYTE_ARRAY_NEGATIVE_LEN, init=O.OK, n=i, values=out, consumed=pos)
        if len(prev) < p:
            return dict(status=O.BYTE_ARRAY_INVALID_PREFIX, init=O.OK, n=0, values=[], consumed=pos)
        prev = (prev[:p] if p > 0 else b"") + suffix
        out.append(prev)
    return dict(status=O.OK, init=O.OK, n=n, values=out, consumed=pos)


def _split(r):
    off = r["offsets"]
    return [bytes(r["data"][int(off[i]): int(off[i + 1])]) for i in range(r["n_decoded"])]


def _agree(got, want):
    assert got["status"] == want["status"], (got["status"], want["status"])
    assert got["init_status"] == want["init"]
    assert got["n_decoded"] == want["n"]
    assert _split(got) == want["values"]
    if want["init"] == O.OK:
        assert got["consumed"] == want["consumed"]


@pytest.mark.parametrize("shape", ["short", "mixed", "sorted", "runs"])
def test_round_trip_through_the_restated_encoders(shape):
    rng = np.random.default_rng(["short", "mixed", "sorted", "runs"].index(shape))
    for n in (0, 2, 3, 64, 65, 127, 128, 130, 200, 258, 1000):
        vals = random_values(rng, n, shape)
        for enc, dec in ((synth.delta_length_ba_encode, lambda s, k: O.delta_length_ba_decode(s, k)),
                         (synth.delta_ba_encode, lambda s, k: O.delta_ba_decode(s, k, total_len(vals)))):
            stream = enc(vals)
            r = dec(stream, n)
            if n == 0:
                # the encoder writes a first value (0) even for an empty stream, the decoder does not read it
                # (encoding/delta-binary-packed-decoder.go:60-62): DELTA_LENGTH_BYTE_ARRAY leaves that byte
                # behind, DELTA_BYTE_ARRAY takes it for the block size of the suffix lengths
                if enc is synth.delta_length_ba_encode:
                    assert r["status"] == O.OK and r["n_decoded"] == 0 and r["consumed"] == stream.size - 1
                else:
                    assert r["init_status"] == O.INVALID_BLOCK_SIZE
                continue
            assert r["status"] == O.OK and r["init_status"] == O.OK, (shape, n, r["status"])
            assert _split(r) == vals
            assert r["consumed"] == stream.size
            more = dec(stream, n + 3)            # pulling past the last value: io.EOF, the values stay good
            assert more["status"] == O.EOF and more["n_decoded"] == n and _split(more) == vals


def test_streams_the_reference_cannot_read_back():
    """One value: its encoder writes no block and Init hits EOF; N = 1 mod 128: the last length
    needs a block that was never written (encoding/delta-binary-packed-decoder.go:103,134-138)."""
    rng = np.random.default_rng(5)
    for n in (1, 129, 257):
        vals = random_values(rng, n)
        r = O.delta_length_ba_decode(synth.delta_length_ba_encode(vals), n)
        assert r["init_status"] != O.OK and r["n_decoded"] == 0
        r = O.delta_ba_decode(synth.delta_ba_encode(vals), n, total_len(vals))
        assert r["init_status"] != O.OK and r["n_decoded"] == 0


def test_oracle_against_the_python_transcription():
    rng = np.random.default_rng(11)
    for stream, n in damaged_dlba(rng):
        _agree(O.delta_length_ba_decode(stream, n), _model_dlba(stream, n))
    for stream, n, cap in damaged_dba(rng):
        got = O.delta_ba_decode(stream, n, cap)
        want = _model_dba(stream, n)
        if total_len(want["values"]) > cap or got["status"] == O.OUT_OF_RANGE_CAP:
            # the library's own limit: the values the reference would return do not fit
            assert got["status"] == O.OUT_OF_RANGE_CAP and got["n_decoded"] == 0
            full = _model_dba(stream, n)
            assert total_len(full["values"]) > cap or full["status"] == O.BYTE_ARRAY_INVALID_PREFIX
            continue
        _agree(got, want)
    statuses = {O.delta_ba_decode(s, n, c)["status"] for s, n, c in damaged_dba(np.random.default_rng(11))}
    assert {O.OK, O.EOF, O.UNEXPECTED_EOF, O.BYTE_ARRAY_NEGATIVE_LEN, O.BYTE_ARRAY_COUNT_MISMATCH,
            O.BYTE_ARRAY_INVALID_PREFIX, O.OUT_OF_RANGE_CAP} <= statuses


def test_truncated_at_every_byte():
    rng = np.random.default_rng(12)
    vals = random_values(rng, 150, "runs")
    for enc, run, model in ((synth.delta_length_ba_encode, lambda s: O.delta_length_ba_decode(s, 150), _model_dlba),
                            (synth.delta_ba_encode, lambda s: O.delta_ba_decode(s, 150, total_len(vals)), _model_dba)):
        stream = enc(vals)
        for cut in range(stream.size + 1):
            _agree(run(stream[:cut].copy()), model(stream[:cut], 150))


def test_byte_array_values_in_data_pages():
    """ReadValues with byte-array values (layout/page-v1.go:79-116): the not-null count of the
    levels is what DecodeValues is asked for."""
    rng = np.random.default_rng(13)
    n = 300
    deff = np.where(rng.random(n) < 0.7, 2, rng.integers(0, 3, size=n))
    rep = rng.integers(0, 2, size=n)
    nn = int((deff == 2).sum())
    vals = random_values(rng, nn, "sorted")
    for kind, enc in ((O.VALUES_DELTA_LENGTH_BA, synth.delta_length_ba_encode), (O.VALUES_DELTA_BA, synth.delta_ba_encode)):
        page = np.concatenate([synth.length_prefixed(synth.hybrid_encode(rep, 1)),
                               synth.length_prefixed(synth.hybrid_encode(deff, 2)), enc(vals)])
        r = O.page_read(page, 1, n, 1, 2, kind, data_cap=total_len(vals))
        assert r["status"] == O.OK and r["not_null"] == nn and r["values_decoded"] == nn
        off = r["values"][: 4 * (n + 1)].view(np.uint32)
        data = r["values"][4 * (n + 1):]
        assert [bytes(data[int(off[i]): int(off[i + 1])]) for i in range(nn)] == vals
        short = O.page_read(page[:-2].copy(), 1, n, 1, 2, kind, data_cap=total_len(vals))
        assert short["status"] == O.PAGE_VALUES_FAILED and short["detail"] in (O.EOF, O.UNEXPECTED_EOF)
