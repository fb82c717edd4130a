#!/usr/bin/env python3
"""Transcribe the reference's known-answer vectors into JSON fixtures.

Reads the reference's *_test.go files as TEXT (nothing is executed) and writes
the input/expected-output vectors they hold to tests/golden/*.json.  Run once in
the build container (the reference is not present on the GPU box):

    python tests/golden/make_fixtures.py [/root/reference]

Sources (paths relative to the reference root):
  encoding/bitpacking32_test.go:25-654     -> bitpack32.json  (127 vectors, w = 0..32)
  encoding/bitpacking64_test.go:25-1744    -> bitpack64.json  (317 vectors, w = 0..64)
  encoding/hybrid-decoder_test.go:11-35,
  encoding/hybrid-encoder_test.go:15-322   -> hybrid.json
  encoding/delta-binary-packed-decoder_test.go:29-153,157-293,
  encoding/delta-binary-packed-encoder_test.go:118-167,274-326 -> delta.json
  encoding/packed-array_test.go:211-229    -> packed_array.json
  types/float_test.go:120-171, types/double_test.go:120-171 -> plain.json
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

_VEC = re.compile(
    r"\{\s*(\d+)\s*,\s*\[\]byte\{([^}]*)\}\s*,\s*\[8\]int(?:32|64)\{([^}]*)\}\s*,?\s*\}", re.S)


def _ints(body):
    return [int(t, 0) for t in re.split(r"[\s,]+", body.strip()) if t]


def bitpack_vectors(path):
    text = open(path).read()
    out = []
    for m in _VEC.finditer(text):
        w, data, values = int(m.group(1)), _ints(m.group(2)), _ints(m.group(3))
        assert len(data) == w and len(values) == 8, (w, data, values)
        out.append({"w": w, "data": data, "values": values})
    return out


def main():
    enc = os.path.join(REF, "encoding")
    v32 = bitpack_vectors(os.path.join(enc, "bitpacking32_test.go"))
    v64 = bitpack_vectors(os.path.join(enc, "bitpacking64_test.go"))
    assert len(v32) == 127, len(v32)
    assert len(v64) == 317, len(v64)
    assert sorted({v["w"] for v in v32}) == list(range(33))
    assert sorted({v["w"] for v in v64}) == list(range(65))
    json.dump({"source": "encoding/bitpacking32_test.go:25-654", "vectors": v32},
              open(os.path.join(OUT, "bitpack32.json"), "w"))
    json.dump({"source": "encoding/bitpacking64_test.go:25-1744", "vectors": v64},
              open(os.path.join(OUT, "bitpack64.json"), "w"))

    # Hybrid vectors.  "stream" is what the reference test feeds / expects the
    # encoder to emit; "values" what the decoder must yield for n pulls.
    bp_only = [136, 16, 33, 66, 132, 8, 17, 34, 68] * 4 + [136, 0, 0]
    hybrid = {
        "source": "encoding/hybrid-decoder_test.go:11-35, encoding/hybrid-encoder_test.go:15-322",
        "vectors": [
            {"name": "GroupBoundary (hybrid-decoder_test.go:11-35): short final group zero-filled",
             "w": 2, "stream": [(1 << 1) | 1, (1 << 0) | (2 << 2) | (3 << 4)],
             "values": [1, 2, 3], "consumed": 2, "encoder": False},
            {"name": "RLEOnly (hybrid-encoder_test.go:15-54)",
             "w": 3, "stream": [200, 1, 4, 200, 1, 5],
             "values": [4] * 100 + [5] * 100, "encoder": True},
            {"name": "RepeatedZeros (hybrid-encoder_test.go:56-78)",
             "w": 3, "stream": [20, 0], "values": [0] * 10, "encoder": True},
            {"name": "BitWidthZeros (hybrid-encoder_test.go:80-101)",
             "w": 0, "stream": [20], "values": [0] * 10, "encoder": True},
            {"name": "BitPackingOnly (hybrid-encoder_test.go:103-133)",
             "w": 3, "stream": [27] + bp_only,
             "values": [i % 3 for i in range(100)], "encoder": True},
            {"name": "TransitionFromBitPackingToRLE (hybrid-encoder_test.go:166-218)",
             "w": 3, "stream": None, "prefix": [3], "suffix": [200, 1, 2],
             "values": [0, 1, 0, 1, 0, 2, 2, 2] + [2] * 100, "encoder": True},
            {"name": "PaddingZerosOnUnfinishedBitPackedRuns (hybrid-encoder_test.go:220-245)",
             "w": 5, "stream": None, "prefix": [5], "suffix": [],
             "values": list(range(1, 10)), "padded_values": list(range(1, 10)) + [0] * 7,
             "encoder": True},
            {"name": "SwitchingModes (hybrid-encoder_test.go:247-322)",
             "w": 9, "stream": None, "prefix": [50, 17, 0, 5], "suffix": [38, 6, 0, 16, 5, 0],
             "values": [17] * 25 + [7] * 7 + [8, 9, 10] + [6] * 25 + [5] * 8, "encoder": True},
            {"name": "BitPackingOverflow (hybrid-encoder_test.go:135-164): first header 63 groups",
             "w": 3, "stream": None, "prefix": [((504 // 8) << 1) | 1], "suffix": [],
             "values": [i % 3 for i in range(1000)], "encoder": True},
        ],
    }
    json.dump(hybrid, open(os.path.join(OUT, "hybrid.json"), "w"))

    golden = [128, 1, 4, 8, 14, 3, 2, 0, 0, 0, 192, 63, 0, 0, 0, 0, 0, 0]
    delta = {
        "source": "encoding/delta-binary-packed-decoder_test.go, encoding/delta-binary-packed-encoder_test.go",
        "decode": [
            {"name": "Next (decoder_test.go:120-139, 260-279)", "stream": golden,
             "values": [7, 5, 3, 1, 2, 3, 4, 5], "init": "ok"},
            {"name": "Next_NoValues (decoder_test.go:141-153, 281-293)",
             "stream": [128, 1, 4, 0, 0], "values": [], "init": "ok", "next": "eof"},
            {"name": "Init_EmptyBuffer (decoder_test.go:50-57)", "stream": [], "init": "eof"},
        ],
        "invalid_block_size": [[0, 1, 4, 0, 0], [127, 1, 4, 0, 0], [129, 1, 4, 0, 0]],
        "invalid_miniblock_count": [[128, 1, 0, 0, 0], [128, 1, 3, 0, 0], [128, 1, 128, 2, 0]],
        "encode": [
            {"name": "Close (encoder_test.go:118-137, 274-296)", "block": 128, "miniblocks": 4,
             "values": [7, 5, 3, 1, 2, 3, 4, 5], "stream": golden},
            {"name": "Init only (decoder_test.go:29-38): header of an empty stream",
             "block": 128, "miniblocks": 4, "values": [], "stream": [128, 1, 4, 0, 0]},
        ],
        "flush_body": {"name": "flush32/flush64 (encoder_test.go:153-167, 312-326)",
                       "deltas": [-2, -2, -2, 1, 1, 1, 1], "min_delta": -2,
                       "body": [3, 2, 0, 0, 0, 192, 63, 0, 0, 0, 0, 0, 0]},
        "encoder_invalid_block_sizes": [-1, 129],
        "encoder_invalid_miniblock_counts": [-1, 3, 8],
    }
    json.dump(delta, open(os.path.join(OUT, "delta.json"), "w"))

    packed = {"source": "encoding/packed-array_test.go:211-229",
              "w": 3, "values": [0, 1, 2, 3, 4, 5, 6, 7],
              "bytes": [0b10001000, 0b11000110, 0b11111010]}
    json.dump(packed, open(os.path.join(OUT, "packed_array.json"), "w"))
    # PLAIN decoder vectors: the byte strings are read from the test sources, the expected
    # values and counts are what the tests assert (1., 2., 3.; two values then io.EOF).
    plain = {"source": "types/float_test.go:120-171, types/double_test.go:120-171", "vectors": []}
    for name, dtype in (("float_test.go", "float32"), ("double_test.go", "float64")):
        text = open(os.path.join(REF, "types", name)).read()
        dec = text[text.index("PlainDecoder_DecodeValues(t"):]
        full = _ints(re.search(r"memory\.NewWriter\(\[\]byte\{([^}]*)\}\)", dec).group(1))
        short_src = dec[dec.index("DecodeValues_NotEnoughValues"):]
        short = _ints(re.search(r"memory\.NewWriter\(\[\]byte\{([^}]*)\}\)", short_src).group(1))
        plain["vectors"].append({"dtype": dtype, "data": full, "n": 3, "values": [1.0, 2.0, 3.0],
                                 "status": "ok", "count": 3})
        plain["vectors"].append({"dtype": dtype, "data": short, "n": 3, "values": [1.0, 2.0],
                                 "status": "eof", "count": 2})
    json.dump(plain, open(os.path.join(OUT, "plain.json"), "w"))
    print("wrote fixtures: %d int32 + %d int64 bit-pack vectors" % (len(v32), len(v64)))


if __name__ == "__main__":
    main()
