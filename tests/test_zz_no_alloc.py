"""The allocation-free entry point (pq_column_entries_noalloc: static device workspace inside the
library) against pq_column_entries_dev on the same device buffers: bit-exact outputs on two calls
in a row, the second over a smaller input, plus the size limit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _levels(n, seed, dev):
    import torch
    g = torch.Generator(device="cpu").manual_seed(seed)
    deff = torch.randint(0, 4, (n,), generator=g, dtype=torch.int32)
    rep = (torch.rand(n, generator=g) < 0.7).to(torch.int32)
    if n:
        rep[0] = 0
    return deff.to(dev), rep.to(dev)


def _run(ctx, deff, rep, n, repeated, noalloc):
    import torch
    dev = deff.device
    offsets = torch.full((n + 1,), -1, dtype=torch.int32, device=dev)
    validity = torch.full(((n + 31) // 32 + 1,), -1, dtype=torch.int32, device=dev)
    edef = torch.full((max(n, 1),), -1, dtype=torch.int32, device=dev)
    res = torch.full((4,), -1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.column_entries_dev(deff.data_ptr(), rep.data_ptr(), n, 3, 1, repeated, offsets.data_ptr(),
                           validity.data_ptr(), edef.data_ptr(), res.data_ptr(), noalloc=noalloc)
    ctx.synchronize()
    status, ne, nv, _ = res.cpu().tolist()
    assert status == 0
    return ne, nv, offsets[: ne + 1].cpu(), validity[: (n + 31) // 32].cpu(), edef[:ne].cpu()


def test_noalloc_matches_allocating_entry_point_twice():
    import torch
    from parquet_go_amd import capi
    ctx = capi.Context(0)
    dev = torch.device("cuda:0")
    try:
        # second call smaller: the shared workspace still holds the first call's summaries
        for n, seed, repeated in ((3_000_001, 1, True), (70_003, 2, True), (5_000, 3, False), (0, 4, True)):
            deff, rep = _levels(n, seed, dev)
            want = _run(ctx, deff, rep, n, repeated, noalloc=False)
            got = _run(ctx, deff, rep, n, repeated, noalloc=True)
            assert got[:2] == want[:2], (n, got[:2], want[:2])
            for g, w in zip(got[2:], want[2:]):
                assert torch.equal(g, w), n
        # beyond the static workspace: a clear error, nothing launched
        n = 30_000_001
        one = torch.zeros(8, dtype=torch.int32, device=dev)
        with pytest.raises(capi.PqError) as e:
            ctx.column_entries_dev(one.data_ptr(), one.data_ptr(), n, 3, 1, True, one.data_ptr(), one.data_ptr(),
                                   one.data_ptr(), one.data_ptr(), noalloc=True)
        assert e.value.code == capi.ERR_OUT_OF_RANGE
    finally:
        ctx.close()


def test_noalloc_against_oracle():
    import oracle_lib as O
    import torch
    from parquet_go_amd import capi
    ctx = capi.Context(0)
    dev = torch.device("cuda:0")
    try:
        for n, seed in ((100_000, 7), (4_097, 8)):
            deff, rep = _levels(n, seed, dev)
            ne, nv, off, val, edef = _run(ctx, deff, rep, n, True, noalloc=True)
            want = O.column_entries(deff.cpu().numpy(), rep.cpu().numpy(), 3, 1, True)
            assert (ne, nv) == (want["n_entries"], want["n_values"])
            assert np.array_equal(off.numpy().view(np.uint32), want["offsets"])
            assert np.array_equal(edef.numpy(), want["entry_def"])
            assert val.numpy().view(np.uint8)[: (n + 31) // 32 * 4].tobytes() == want["validity"].tobytes()
    finally:
        ctx.close()
