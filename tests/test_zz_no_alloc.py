"""The allocation-free entry point (pq_column_entries_noalloc: static device workspace inside the
library) against pq_column_entries_dev on the same device buffers: bit-exact outputs on two calls
in a row, the second over a smaller input, plus the size limit; that it allocates no device memory;
and that calls from two contexts on two streams, back to back with no wait between them, are put in
order by the library (they share the one workspace of the GPU)."""
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


def test_noalloc_allocates_nothing():
    """A fresh context has no workspace of its own yet: the first call through the allocation-free entry point, over a
    large input, must leave the device's free memory where it was (the allocating entry point does not)."""
    import torch
    from parquet_go_amd import capi
    dev = torch.device("cuda:0")
    n = 20_000_000
    deff, rep = _levels(n, 11, dev)
    offsets = torch.empty(n + 1, dtype=torch.int32, device=dev)
    validity = torch.empty((n + 31) // 32 + 1, dtype=torch.int32, device=dev)
    edef = torch.empty(n, dtype=torch.int32, device=dev)
    res = torch.zeros(4, dtype=torch.int32, device=dev)
    ctx = capi.Context(0)
    try:
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info()[0]
        ctx.column_entries_dev(deff.data_ptr(), rep.data_ptr(), n, 3, 1, True, offsets.data_ptr(), validity.data_ptr(),
                               edef.data_ptr(), res.data_ptr(), noalloc=True)
        ctx.synchronize()
        assert torch.cuda.mem_get_info()[0] == free_before
        assert res.cpu().tolist()[0] == 0
        ne = int(res[1])
        # and the other entry point, on the same context, does take memory (the check above can tell)
        ctx.column_entries_dev(deff.data_ptr(), rep.data_ptr(), n, 3, 1, True, offsets.data_ptr(), validity.data_ptr(),
                               edef.data_ptr(), res.data_ptr(), noalloc=False)
        ctx.synchronize()
        assert torch.cuda.mem_get_info()[0] < free_before
        assert int(res[1]) == ne
    finally:
        ctx.close()


def test_noalloc_calls_of_two_contexts_are_put_in_order():
    """Two contexts, each with a stream of its own, call back to back without waiting: the second call clears the
    records the first one's workgroups wait on, so the library has to hold it back until the first is done."""
    import torch
    from parquet_go_amd import capi
    dev = torch.device("cuda:0")
    a, b = capi.Context(0), capi.Context(0)
    try:
        n1, n2 = 8_000_003, 1_000_001
        d1, r1 = _levels(n1, 21, dev)
        d2, r2 = _levels(n2, 22, dev)
        want1 = _run(a, d1, r1, n1, True, noalloc=False)
        want2 = _run(b, d2, r2, n2, True, noalloc=False)
        outs = []
        for n in (n1, n2):
            outs.append((torch.full((n + 1,), -1, dtype=torch.int32, device=dev),
                         torch.full(((n + 31) // 32 + 1,), -1, dtype=torch.int32, device=dev),
                         torch.full((n,), -1, dtype=torch.int32, device=dev),
                         torch.full((4,), -1, dtype=torch.int32, device=dev)))
        torch.cuda.synchronize()
        for rounds in range(3):
            for ctx, (d, r, n), o in ((a, (d1, r1, n1), outs[0]), (b, (d2, r2, n2), outs[1])):
                ctx.column_entries_dev(d.data_ptr(), r.data_ptr(), n, 3, 1, True, o[0].data_ptr(), o[1].data_ptr(),
                                       o[2].data_ptr(), o[3].data_ptr(), noalloc=True)
        a.synchronize()
        b.synchronize()
        for want, o, n in ((want1, outs[0], n1), (want2, outs[1], n2)):
            status, ne, nv, _ = o[3].cpu().tolist()
            assert (status, ne, nv) == (0, want[0], want[1])
            assert torch.equal(o[0][: ne + 1].cpu(), want[2]) and torch.equal(o[1][: (n + 31) // 32].cpu(), want[3])
            assert torch.equal(o[2][:ne].cpu(), want[4])
    finally:
        a.close()
        b.close()
