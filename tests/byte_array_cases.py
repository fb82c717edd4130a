"""Inputs for the DELTA_LENGTH_BYTE_ARRAY / DELTA_BYTE_ARRAY tests (types/byte-array.go:116-348):
value lists the reference's encoders would write, and streams assembled from arbitrary length
lists (negative lengths, prefixes longer than the value in front, unequal counts) that only a
damaged or foreign file could hold.  Shared by the CPU (oracle) and the GPU (parity) tests."""
import numpy as np

from parquet_go_amd import synth


def random_values(rng, n, shape="short"):
    """n byte strings.  short: 0..24 bytes; mixed: mostly short, some hundreds or thousands of bytes;
    sorted: short strings in order (long shared prefixes); runs: repeats and extensions of the value
    in front (prefix == whole previous value)."""
    if shape == "sorted":
        vals = sorted(bytes(rng.integers(97, 101, size=int(rng.integers(0, 20)), dtype=np.uint8)) for _ in range(n))
        return vals
    if shape == "runs":
        vals, cur = [], b""
        for _ in range(n):
            k = int(rng.integers(0, 4))
            if k == 0:
                cur = bytes(rng.integers(0, 256, size=int(rng.integers(0, 30)), dtype=np.uint8))
            elif k == 1:
                cur = cur + bytes(rng.integers(0, 256, size=int(rng.integers(0, 9)), dtype=np.uint8))
            elif k == 2:
                cur = cur[: int(rng.integers(0, len(cur) + 1))]
            vals.append(cur)
        return vals
    vals = []
    for _ in range(n):
        ln = int(rng.integers(0, 25))
        if shape == "mixed":
            u = rng.random()
            if u < 0.03:
                ln = int(rng.integers(150, 700))
            elif u < 0.04:
                ln = int(rng.integers(2000, 9000))
        vals.append(bytes(rng.integers(0, 256, size=ln, dtype=np.uint8)))
    return vals


def dlba_stream(lens, body, block=128, miniblocks=4):
    """A DELTA_LENGTH_BYTE_ARRAY stream from any int32 length list and any bytes behind it."""
    return np.concatenate([synth.delta_encode(np.asarray(lens, dtype=np.int32), 32, block, miniblocks),
                           np.asarray(body, dtype=np.uint8)]).astype(np.uint8)


def dba_stream(plens, slens, body, block=128, miniblocks=4):
    """A DELTA_BYTE_ARRAY stream from any prefix / suffix length lists."""
    return np.concatenate([synth.delta_encode(np.asarray(plens, dtype=np.int32), 32, block, miniblocks),
                           dlba_stream(slens, body, block, miniblocks)]).astype(np.uint8)


def total_len(values):
    return sum(len(v) for v in values)


def damaged_dlba(rng):
    """(stream, n) pairs whose decode ends in each error the reference can reach."""
    out = []
    body = rng.integers(0, 256, size=4000, dtype=np.uint8)
    lens = rng.integers(0, 20, size=200)
    out.append((dlba_stream(lens, body[: int(lens.sum())]), 200))           # exact
    out.append((dlba_stream(lens, body[: int(lens.sum()) - 1]), 200))       # last value one byte short
    out.append((dlba_stream(lens, body[: int(lens[:100].sum())]), 200))     # io.EOF at a value boundary
    out.append((dlba_stream(lens, body[:0]), 200))                          # no bytes at all
    neg = lens.copy()
    neg[77] = -5
    out.append((dlba_stream(neg, body), 200))                               # make([]byte, -5)
    neg[0] = -1
    out.append((dlba_stream(neg, body), 200))
    big = lens.copy()
    big[150] = 2**31 - 1
    out.append((dlba_stream(big, body), 200))
    out.append((dlba_stream(np.zeros(300, np.int64), body[:0]), 300))       # all empty values
    out.append((dlba_stream(lens, body, 256, 8), 200))                      # other block layouts
    out.append((dlba_stream(lens, body, 128, 1), 150))
    out.append((dlba_stream(lens[:129], body), 129))                        # N = 1 mod 128: Init fails
    out.append((dlba_stream(lens[:1], body), 1))                            # one value: no block, Init fails
    out.append((dlba_stream(lens[:2], body), 5))
    out.append((dlba_stream(lens[:0], body), 3))                            # zero values declared
    return out


def damaged_dba(rng):
    """(stream, n, data_cap) triples for ByteArrayDeltaDecoder's error paths."""
    out = []
    body = rng.integers(0, 256, size=6000, dtype=np.uint8)
    n = 200
    sl = rng.integers(1, 12, size=n)
    # legal prefixes: at most the length of the value in front
    pl = np.zeros(n, np.int64)
    prev = 0
    for i in range(n):
        pl[i] = int(rng.integers(0, prev + 1))
        prev = pl[i] + sl[i]
    room = int((pl + sl).sum())
    out.append((dba_stream(pl, sl, body[: int(sl.sum())]), n, room))
    out.append((dba_stream(pl, sl, body[: int(sl.sum())]), n, room - 1))        # does not fit
    out.append((dba_stream(pl, sl, body[: int(sl.sum()) - 3]), n, room))        # suffix bytes run out
    out.append((dba_stream(pl, sl, body[: int(sl[:64].sum())]), n, room))       # ... at a batch boundary
    bad = pl.copy()
    bad[90] = pl[89] + sl[89] + 1
    out.append((dba_stream(bad, sl, body), n, room + 100))                      # invalid prefix len -> count 0
    bad = pl.copy()
    bad[0] = 1
    out.append((dba_stream(bad, sl, body), n, room + 100))                      # first value with a prefix
    negp = pl.copy()
    negp[40] = -3                                                               # -3 + suffix >= 0: copies nothing
    for i in range(41, n):                                                      # keep the rest legal
        negp[i] = min(negp[i], (max(negp[i - 1], 0) + sl[i - 1]))
    out.append((dba_stream(negp, sl, body), n, room + 100))
    negp2 = pl.copy()
    negp2[40] = -100                                                            # make(.., 0, negative) panics
    out.append((dba_stream(negp2, sl, body), n, room + 100))
    negs = sl.copy()
    negs[130] = -1
    out.append((dba_stream(pl, negs, body), n, room + 100))
    out.append((dba_stream(pl[:150], sl, body), n, room))                       # counts differ
    out.append((dba_stream(pl, sl[:199], body), n, room))
    out.append((dba_stream(pl[:129], sl[:129], body), 129, room))               # N = 1 mod 128
    out.append((dba_stream(pl[:0], sl[:0], body), 4, 16))                       # zero values
    out.append((dba_stream(pl[:2], sl[:2], body), 9, room))
    out.append((dba_stream(pl, sl, body, 256, 4), n, room))                     # other layouts: serial lengths
    out.append((np.concatenate([synth.delta_encode(pl.astype(np.int32), 32, 128, 4),
                                dlba_stream(sl, body, 512, 8)]), n, room))
    huge = pl.copy()
    huge[199] = 2**31 - 1
    out.append((dba_stream(huge, sl, body), n, room + 100))
    return out
