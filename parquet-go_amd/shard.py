"""Partitioning of decode work over the GPUs of one node.

Data pages share nothing except their column chunk's dictionary page (the reference
creates fresh decoders per page, layout/chunk.go:196-209, and reads column chunks
independently, file-reader.go:167-188), so the work shards with no exchange step:

* with at least as many column chunks as ranks, whole chunks are assigned (the dictionary
  travels only to the owning rank), balanced greedily by encoded + decoded bytes;
* with fewer chunks than ranks, every chunk is cut into contiguous page ranges, one per
  rank, and the (small) dictionary is replicated.

Everything here is deterministic and depends only on its arguments, so every rank computes
the same assignment without communicating.
"""
from dataclasses import dataclass
from typing import List, Sequence, Tuple

import numpy as np


@dataclass(frozen=True)
class ChunkInfo:
    """One column chunk: how many data pages it has and what it costs to decode."""
    n_pages: int
    encoded_bytes: int
    decoded_bytes: int

    @property
    def cost(self) -> int:
        # the decode is HBM-bound: time is proportional to bytes read + bytes written
        return int(self.encoded_bytes) + int(self.decoded_bytes)


@dataclass(frozen=True)
class Piece:
    """Pages [page_lo, page_hi) of chunk `chunk`, decoded by one rank."""
    chunk: int
    page_lo: int
    page_hi: int


def assign_chunks(costs: Sequence[int], world: int) -> List[List[int]]:
    """Longest-processing-time-first: chunks in decreasing cost order, each to the least
    loaded rank (ties: lower chunk index first, lower rank first).  Returns, per rank, the
    sorted list of chunk indices."""
    if world < 1:
        raise ValueError("world must be >= 1")
    order = sorted(range(len(costs)), key=lambda i: (-int(costs[i]), i))
    load = [0] * world
    owned: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        owned[r].append(i)
        load[r] += int(costs[i])
    return [sorted(o) for o in owned]


def split_pages(n_pages: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous, near-equal page ranges; the first n_pages % world ranks get one more."""
    base, extra = divmod(n_pages, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


def plan_shards(chunks: Sequence[ChunkInfo], world: int) -> List[List[Piece]]:
    """Per rank, the pieces it decodes.  Every page of every chunk appears in exactly one piece."""
    if len(chunks) >= world:
        owned = assign_chunks([c.cost for c in chunks], world)
        return [[Piece(i, 0, chunks[i].n_pages) for i in mine] for mine in owned]
    plan: List[List[Piece]] = [[] for _ in range(world)]
    for i, c in enumerate(chunks):
        for r, (lo, hi) in enumerate(split_pages(c.n_pages, world)):
            if hi > lo:
                plan[r].append(Piece(i, lo, hi))
    return plan


def local_descriptors(descs: np.ndarray, first_page_of_chunk: Sequence[int], pieces: Sequence[Piece]) -> np.ndarray:
    """Rows of a node-wide descriptor table (pages of chunk c are rows
    first_page_of_chunk[c] ... ) that one rank decodes, in piece order."""
    if not pieces:
        return descs[:0].copy()
    rows = np.concatenate([np.arange(first_page_of_chunk[p.chunk] + p.page_lo,
                                     first_page_of_chunk[p.chunk] + p.page_hi) for p in pieces])
    return descs[rows].copy()


def imbalance(chunks: Sequence[ChunkInfo], plan: Sequence[Sequence[Piece]]) -> float:
    """max rank cost / mean rank cost (1.0 = perfectly balanced), page-proportional inside chunks."""
    loads = []
    for pieces in plan:
        t = 0.0
        for p in pieces:
            c = chunks[p.chunk]
            t += c.cost * (p.page_hi - p.page_lo) / max(c.n_pages, 1)
        loads.append(t)
    mean = sum(loads) / len(loads)
    return max(loads) / mean if mean else 1.0
