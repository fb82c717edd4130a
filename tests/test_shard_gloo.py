"""Multi-rank partitioning of the decode work (no GPU): unit properties of the planner and
a world_size-2 run over gloo in which each rank decodes only its pieces (with the CPU oracle
standing in for the device) and rank 0 checks that the union is the whole file, once."""
import os
import socket
import sys
import zlib

import numpy as np
import pytest

from parquet_go_amd import shard, synth

HERE = os.path.dirname(os.path.abspath(__file__))


def test_split_pages_covers_everything():
    for n in (0, 1, 7, 8, 9, 1000):
        for w in (1, 2, 3, 8):
            parts = shard.split_pages(n, w)
            assert len(parts) == w and parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1


def test_assign_chunks_is_a_balanced_partition():
    rng = np.random.default_rng(11)
    for world in (1, 2, 4, 8):
        for n in (world, 16, 64):
            costs = rng.integers(1, 1000, size=n).tolist()
            owned = shard.assign_chunks(costs, world)
            flat = sorted(i for o in owned for i in o)
            assert flat == list(range(n))
            loads = [sum(costs[i] for i in o) for o in owned]
            # greedy LPT: no rank exceeds the mean by more than the largest chunk
            assert max(loads) <= sum(costs) / world + max(costs)
    assert shard.assign_chunks([5, 5, 5, 5], 2) == [[0, 2], [1, 3]]   # deterministic tie-breaks
    with pytest.raises(ValueError):
        shard.assign_chunks([1], 0)


def test_plan_shards_both_regimes():
    chunks = [shard.ChunkInfo(n_pages=10 + i, encoded_bytes=1000 * (i + 1), decoded_bytes=8000) for i in range(5)]
    for world in (1, 2, 4, 8):
        plan = shard.plan_shards(chunks, world)
        seen = {}
        for r, pieces in enumerate(plan):
            for p in pieces:
                for page in range(p.page_lo, p.page_hi):
                    assert (p.chunk, page) not in seen
                    seen[(p.chunk, page)] = r
        assert len(seen) == sum(c.n_pages for c in chunks)
        if world <= len(chunks):
            assert all(p.page_lo == 0 and p.page_hi == chunks[p.chunk].n_pages for pieces in plan for p in pieces)
        else:
            assert all(len(pieces) > 0 for pieces in plan)   # page-level split keeps every rank busy
        assert shard.imbalance(chunks, plan) < 1.6


def _build_file(seed=5):
    """A small multi-chunk 'file': even chunks INT64 delta, odd chunks INT32 dictionary keys."""
    rng = np.random.default_rng(seed)
    chunks, pages = [], []          # pages: (chunk, kind, stream, n, dictionary)
    for c in range(5):
        n_pages = 3 + c
        enc = dec = 0
        d = rng.integers(-2**31, 2**31, size=256, dtype=np.int64).astype(np.int32)
        for _ in range(n_pages):
            n = int(rng.integers(200, 900))
            if c % 2 == 0:
                v = np.cumsum(rng.integers(-50, 5000, size=n)).astype(np.int64)
                s = synth.delta_encode(v, 64)
                pages.append((c, "delta64", s, n, None))
                dec += 8 * n
            else:
                keys = rng.integers(0, 256, size=n)
                s = synth.dict_page(keys, w=8)
                pages.append((c, "dict", s, n, d))
                dec += 4 * n
            enc += s.size
        chunks.append(shard.ChunkInfo(n_pages, enc, dec))
    return chunks, pages


def _decode_page(page):
    import oracle_lib as O
    _, kind, s, n, d = page
    if kind == "delta64":
        r = O.delta_decode64(s, n)
        assert r["status"] == 0
        return zlib.crc32(r["values"].tobytes())
    st, vals, nd, _ = O.dict_decode(s, d, n)
    assert st == 0 and nd == n
    return zlib.crc32(vals.tobytes())


def _rank_main(rank, world, port, n_chunks_used, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        chunks, pages = _build_file()
        chunks = chunks[:n_chunks_used]
        pages = [p for p in pages if p[0] < n_chunks_used]
        first = np.concatenate([[0], np.cumsum([c.n_pages for c in chunks])])[:-1]
        plan = shard.plan_shards(chunks, world)          # same on every rank, nothing exchanged
        mine = {}
        for piece in plan[rank]:
            for pg in range(piece.page_lo, piece.page_hi):
                row = int(first[piece.chunk]) + pg
                mine[row] = _decode_page(pages[row])
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)           # verification only: the data path has no collective
        if rank == 0:
            merged = {}
            for g in gathered:
                assert not (set(g) & set(merged)), "a page was decoded by two ranks"
                merged.update(g)
            want = {row: _decode_page(p) for row, p in enumerate(pages)}
            q.put(("ok", merged == want, len(merged), [len(g) for g in gathered]))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("n_chunks_used", [5, 1])   # chunk-level and page-level partitioning
def test_two_ranks_cover_the_file_once(n_chunks_used):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, n_chunks_used, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    tag, same, n, per_rank = q.get(timeout=10)
    assert tag == "ok" and same
    assert n == sum(3 + c for c in range(n_chunks_used))
    assert all(k > 0 for k in per_rank)


if HERE not in sys.path:
    sys.path.insert(0, HERE)


def test_bench_forms_its_own_ranks_with_disjoint_pages():
    """`bench.py --gpus 2 --dry-run` starts two ranks by itself (no launcher), which agree over gloo
    on a partition of the job's pages: disjoint ranges that cover everything."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--dry-run",
                          "--pages", "37"], env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                      # rank 0 prints the one line
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["total_pages"] == 74
    ranks = sorted(rec["ranks"], key=lambda r: r["rank"])
    assert [r["rank"] for r in ranks] == [0, 1] and all(r["world"] == 2 for r in ranks)
    assert ranks[0]["page_lo"] == 0 and ranks[0]["page_hi"] == ranks[1]["page_lo"] and ranks[1]["page_hi"] == 74
    assert all(r["page_hi"] > r["page_lo"] for r in ranks)
