"""The CPU oracle against every known-answer vector the reference's own tests hold
(fixtures under tests/golden, transcribed by tests/golden/make_fixtures.py)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O


def _load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


def test_bitpack32_vectors(golden_dir):
    vecs = _load(golden_dir, "bitpack32.json")["vectors"]
    assert len(vecs) == 127
    for v in vecs:
        got = O.unpack8_32(v["data"], v["w"])
        assert got.tolist() == v["values"], v
        assert O.pack8_32(v["values"], v["w"]).tolist() == v["data"], v


def test_bitpack64_vectors(golden_dir):
    vecs = _load(golden_dir, "bitpack64.json")["vectors"]
    assert len(vecs) == 317
    for v in vecs:
        got = O.unpack8_64(v["data"], v["w"])
        assert got.tolist() == v["values"], v
        assert O.pack8_64(v["values"], v["w"]).tolist() == v["data"], v


def test_hybrid_vectors(golden_dir):
    for v in _load(golden_dir, "hybrid.json")["vectors"]:
        w, values = v["w"], v["values"]
        if v["encoder"]:
            stream = O.hybrid_encode(values, w).tolist()
            if v["stream"] is not None:
                assert stream == v["stream"], v["name"]
            else:
                assert stream[:len(v["prefix"])] == v["prefix"], v["name"]
                if v["suffix"]:
                    assert stream[-len(v["suffix"]):] == v["suffix"], v["name"]
        else:
            stream = v["stream"]
        st, got, consumed = O.hybrid_decode(stream, w, len(values))
        assert st == O.OK, v["name"]
        assert got.tolist() == values, v["name"]
        if "consumed" in v:
            assert consumed == v["consumed"]
        if "padded_values" in v:
            st, got, _ = O.hybrid_decode(stream, w, len(v["padded_values"]))
            assert st == O.OK and got.tolist() == v["padded_values"]


def test_hybrid_group_boundary_tail(golden_dir):
    # bytes 03 39 at w=2: 1,2,3 then the zero-filled rest of the group, then EOF
    st, got, consumed = O.hybrid_decode([0x03, 0x39], 2, 9)
    assert got.tolist() == [1, 2, 3, 0, 0, 0, 0, 0]
    assert st == O.EOF and consumed == 2


@pytest.mark.parametrize("bits", [32, 64])
def test_delta_vectors(golden_dir, bits):
    d = _load(golden_dir, "delta.json")
    dec = O.delta_decode32 if bits == 32 else O.delta_decode64
    enc = O.delta_encode32 if bits == 32 else O.delta_encode64
    for v in d["decode"]:
        if v["init"] == "eof":
            r = dec(v["stream"], 1)
            assert r["init_status"] == O.EOF
            continue
        r = dec(v["stream"], len(v["values"]))
        assert r["init_status"] == O.OK and r["status"] == O.OK
        assert r["values"].tolist() == v["values"]
        if v.get("next") == "eof":
            r = dec(v["stream"], 1)
            assert r["status"] == O.EOF and r["n_decoded"] == 0
    for s in d["invalid_block_size"]:
        assert dec(s, 1)["init_status"] == O.INVALID_BLOCK_SIZE
    for s in d["invalid_miniblock_count"]:
        assert dec(s, 1)["init_status"] == O.INVALID_MINIBLOCK_COUNT
    for v in d["encode"]:
        rc, stream = enc(v["values"], v["block"], v["miniblocks"])
        assert rc == 0 and stream.tolist() == v["stream"], v["name"]
    # flush golden: deltas -2,-2,-2,1,1,1,1 are what values 7,5,3,1,2,3,4,5 produce
    fb = d["flush_body"]
    rc, stream = enc([7, 5, 3, 1, 2, 3, 4, 5], 128, 4)
    assert stream.tolist()[5:] == fb["body"]
    for bs in d["encoder_invalid_block_sizes"]:
        assert enc([1, 2], bs, 4)[0] == O.INVALID_BLOCK_SIZE
    for mc in d["encoder_invalid_miniblock_counts"]:
        assert enc([1, 2], 128, mc)[0] == O.INVALID_MINIBLOCK_COUNT


def test_packed_array_vector(golden_dir):
    v = _load(golden_dir, "packed_array.json")
    assert O.pack8_32(v["values"], v["w"]).tolist() == v["bytes"]
    # decodePackedArray keeps the most recent <= 8 values unflushed
    stream = O.hybrid_encode(v["values"] * 2, v["w"])
    framed = np.concatenate([np.frombuffer(np.uint32(stream.size).tobytes(), np.uint8), stream])
    r = O.levels_decode(framed, 7, 16, mode=1)
    assert r["status"] == O.OK
    assert r["packed"].tolist() == v["bytes"]
    assert r["tail"].tolist() == v["values"]
    assert r["not_null"] == 2


# ---- closed form, independent of the C code: S >> k*w mod 2^w over the LE group ----

@pytest.mark.parametrize("bits", [32, 64])
def test_unpack_closed_form_random(bits):
    rng = np.random.default_rng(11)
    for w in range(bits + 1):
        data = rng.integers(0, 256, size=max(w, 1), dtype=np.uint8)[:w]
        s = int.from_bytes(bytes(data.tolist()), "little")
        want = [(s >> (k * w)) & ((1 << w) - 1) for k in range(8)]
        if bits == 32:
            got = O.unpack8_32(data, w).astype(np.uint32).tolist()
        else:
            got = O.unpack8_64(data, w).astype(np.uint64).tolist()
        assert [int(x) for x in got] == want, w
        packed = (O.pack8_32(np.array(want, np.uint32).astype(np.int32), w) if bits == 32
                  else O.pack8_64(np.array(want, np.uint64).astype(np.int64), w))
        assert packed.tolist() == data.tolist(), w


def test_plain_decoder_vectors(golden_dir):
    """types/float_test.go:120-171, types/double_test.go:120-171: three values, and two values + io.EOF."""
    import json
    import os
    vs = json.load(open(os.path.join(golden_dir, "plain.json")))["vectors"]
    assert len(vs) == 4
    for v in vs:
        r = O.plain_decode(np.array(v["data"], dtype=np.uint8), v["n"], v["dtype"])
        assert r["status"] == (O.OK if v["status"] == "ok" else O.EOF)
        assert r["n_decoded"] == v["count"]
        assert r["values"].tolist() == v["values"]


def test_plain_and_bool_oracle_edges():
    # a partial last value is io.ErrUnexpectedEOF (binary.Read = io.ReadFull)
    r = O.plain_decode(np.arange(11, dtype=np.uint8), 3, np.int32)
    assert r["status"] == O.UNEXPECTED_EOF and r["n_decoded"] == 2 and r["consumed"] == 11
    r = O.plain_decode(np.zeros(0, np.uint8), 1, np.int64)
    assert r["status"] == O.EOF and r["n_decoded"] == 0
    # booleans: bit j of byte i is value 8 i + j; a missing byte is io.EOF at a multiple of 8
    r = O.bool_plain_decode(np.array([0b10100101, 0x01], dtype=np.uint8), 12)
    assert r["status"] == O.OK and r["values"].tolist() == [1, 0, 1, 0, 0, 1, 0, 1, 1, 0, 0, 0]
    assert r["consumed"] == 2
    r = O.bool_plain_decode(np.array([0xFF], dtype=np.uint8), 12)
    assert r["status"] == O.EOF and r["n_decoded"] == 8
