"""mpnn_aggregate_fwd_noalloc (the forward edge pass for callers that cannot allocate: its split-row slabs live in
a device array built into the library) against mpnn_aggregate_fwd with caller-provided slabs, through the C ABI.

Bar: bit-exact (same kernels, same order of additions).  Two calls in a row on one GPU, the second with a smaller
input, so that left-over contents of the shared array from the first call would show.  Collected last on purpose.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

NOALLOC_BYTES = 16 << 20   # MPNN_NOALLOC_SCRATCH_BYTES of include/mpnn_amd.h


def _case(dev, n, e, H, Fe, seed, dtype):
    from mpnn_amd.graph import MPGraph
    gen = torch.Generator().manual_seed(seed)
    dst = (torch.rand(e, generator=gen) ** 6 * n).long().clamp_(max=n - 1)   # skewed: hubs are cut into pieces
    src = torch.randint(0, n, (e,), generator=gen)
    graph = MPGraph(torch.stack([src, dst]).to(dev), n, chunk=64)
    ab = torch.randn(n, 2 * H, generator=gen).to(dev, dtype)
    ea = torch.rand(e, Fe, generator=gen).to(dev, dtype)
    we = (torch.randn(H, Fe, generator=gen) * 0.5).to(dev, dtype)
    ea_s, _ = graph.sorted_edge_attr(ea, dtype)
    return graph, ab, ea_s, we


def _run(lib, entry, graph, ab, ea_s, we, train):
    """Calls `entry` ('alloc' = mpnn_aggregate_fwd with fresh slabs, 'noalloc' = the built-in array)."""
    from mpnn_amd import _lib
    csr = graph.by_dst
    n, H = ab.size(0), ab.size(1) // 2
    dev, dtype = ab.device, ab.dtype
    A, B = ab[:, :H], ab[:, H:]
    out = torch.full((n, H), float("nan"), dtype=dtype, device=dev)
    cnt = torch.full((n, H), 77, dtype=torch.uint8, device=dev) if train else None
    big = torch.full((csr.n_split + csr.n_big, H), float("nan"), dtype=torch.float32, device=dev) if train else None
    words = lib.mpnn_mask_words(H, _lib.dtype_code(dtype)) if train else 0
    mask = torch.zeros((csr.num_edges, words), dtype=torch.int32, device=dev) if words else None
    head = (_lib.ptr(A), A.stride(0), _lib.ptr(B), B.stride(0), _lib.ptr(ea_s), ea_s.size(1), _lib.ptr(we),
            _lib.ptr(csr.col), _lib.ptr(csr.items), csr.n_items, _lib.ptr(csr.split), csr.n_split, csr.n_parts,
            _lib.ptr(out), out.stride(0), _lib.ptr(cnt), _lib.ptr(big), csr.n_big)
    tail = (_lib.ptr(mask), n, H, _lib.dtype_code(dtype), _lib.stream_handle(dev))
    if entry == "alloc":
        part_out = torch.full((csr.n_parts, H), float("nan"), dtype=torch.float32, device=dev)
        part_cnt = torch.full((csr.n_parts, H), float("nan"), dtype=torch.float32, device=dev) if train else None
        _lib.check(lib.mpnn_aggregate_fwd(*head, _lib.ptr(part_out), _lib.ptr(part_cnt), *tail), "mpnn_aggregate_fwd")
    else:
        _lib.check(lib.mpnn_aggregate_fwd_noalloc(*head, *tail), "mpnn_aggregate_fwd_noalloc")
    torch.cuda.synchronize(dev)
    if cnt is not None:      # the byte counts of the long rows are unspecified: compare them through the exact table
        rows = torch.cat([csr.split[:csr.n_split, 0], csr.items[csr.n_parts:csr.n_parts + csr.n_big, 0]]).long()
        cnt[rows] = 0
    return out, cnt, big, mask


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_noalloc_forward_matches_the_allocating_entry_point_twice_in_a_row(dtype):
    from mpnn_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    big = _case(dev, 4000, 400_000, 128, 4, seed=5, dtype=dtype)
    small = _case(dev, 500, 30_000, 64, 4, seed=6, dtype=dtype)
    assert big[0].by_dst.n_parts > small[0].by_dst.n_parts > 0
    for train in (True, False):
        for graph, ab, ea_s, we in (big, small):       # second call: smaller input over the first call's leftovers
            want = _run(lib, "alloc", graph, ab, ea_s, we, train)
            got = _run(lib, "noalloc", graph, ab, ea_s, we, train)
            for g, w, what in zip(got, want, ("out", "byte counts", "exact counts of the long rows", "mask bits")):
                if w is None:
                    assert g is None
                    continue
                assert torch.equal(g, w), f"{what} differs ({dtype}, train={train}, n_parts={graph.by_dst.n_parts})"
            assert not torch.isnan(got[0].float()).any()


def test_noalloc_rejects_more_pieces_than_the_array_holds():
    from mpnn_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    graph, ab, ea_s, we = _case(dev, 500, 30_000, 64, 4, seed=7, dtype=torch.float32)
    csr = graph.by_dst
    H = 64
    out = torch.zeros((500, H), device=dev)
    too_many = NOALLOC_BYTES // 4 // H + 1
    A, B = ab[:, :H], ab[:, H:]
    rc = lib.mpnn_aggregate_fwd_noalloc(
        _lib.ptr(A), A.stride(0), _lib.ptr(B), B.stride(0), _lib.ptr(ea_s), 4, _lib.ptr(we), _lib.ptr(csr.col),
        _lib.ptr(csr.items), csr.n_items, _lib.ptr(csr.split), csr.n_split, too_many, _lib.ptr(out), H, None, None, 0,
        None, 500, H, _lib.MPNN_F32, _lib.stream_handle(dev))
    assert rc == -4      # MPNN_E_CAPACITY
    assert b"scratch" in lib.mpnn_last_error()
    torch.cuda.synchronize(dev)
    assert float(out.abs().sum()) == 0.0     # nothing ran
