"""World-size-2 (and 3) tests of the multi-GPU path on CPU with the gloo backend.

The partition arithmetic, the halo all-gather / reduce-scatter autograd pair and the flat gradient
all-reduce are exercised for real; the per-rank edge kernel is replaced by a plain-torch stand-in
(the HIP kernel needs a GPU), injected through the hooks partition.py offers for this purpose.
The partitioned result must equal the single-process oracle on the whole graph.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import assert_close
from oracle import reference_oracle as O


def cpu_graph(edge_index, num_dst, num_src):
    return (edge_index, num_dst, num_src)


def cpu_aggregate(a, b, edge_attr, w_edge, graph):
    ei, num_dst, _ = graph
    msg = torch.relu((a[ei[1]] + b[ei[0]]) + edge_attr @ w_edge.t())
    return a.new_zeros(num_dst, a.size(1)).index_add_(0, ei[1], msg)


def cpu_project(x, w):
    return torch.nn.functional.linear(x, w)


def cpu_update(x, aggr, w_out):
    return torch.relu(torch.nn.functional.linear(torch.cat([x, aggr], 1), w_out))


def make_problem(mode, f_in=10):
    gen = torch.Generator().manual_seed(17)
    n, e, fe, h = 120, 1500, 4, 16
    dst = (torch.rand(e, generator=gen) ** 2 * n).long().clamp_(max=n - 1)
    src = torch.randint(0, n, (e,), generator=gen)
    ei = torch.stack([src, dst])
    x = torch.rand(n, f_in, generator=gen)
    ea = torch.rand(e, fe, generator=gen)
    y = torch.randint(0, 5, (n,), generator=gen)
    torch.manual_seed(3)
    from mpnn_amd import MPNN
    model = MPNN(f_in, h, 5, num_layers=2, mode=mode)
    return n, ei, x, ea, y, model


def _worker(rank, world, port, mode, out_dir, halo_mode="all_gather", first_layer=False):
    _init(rank, world, port, out_dir)
    try:
        from mpnn_amd.partition import Halo, PartitionedGraph, allreduce_gradients, partitioned_model
        torch.set_num_threads(1)
        n, ei, x, ea, y, model = make_problem(mode)
        pg = PartitionedGraph(ei, n, rank, world, graph_factory=cpu_graph)
        halo = Halo(pg, mode=halo_mode)
        x_loc = pg.local_rows(x).clone().requires_grad_(not first_layer)
        logits = partitioned_model(model, x_loc, pg, pg.local_edges(ea), aggregate_fn=cpu_aggregate,
                                   dense_fns=(cpu_project, cpu_update), halo=halo, x_all=x if first_layer else None)
        # sum-reduced loss so that the global loss is the sum of the local ones
        loss = torch.nn.functional.cross_entropy(logits, pg.local_rows(y), reduction="sum")
        loss.backward()
        allreduce_gradients(model)
        rows = {}
        if halo.mode == "rows":
            off = [0]
            for c in halo.need_counts:
                off.append(off[-1] + c)
            rows = {"need": [halo.need_local[off[q]:off[q + 1]].clone() for q in range(world)],
                    "serve": [halo.serve_idx[halo.serve_off[q]:halo.serve_off[q + 1]].clone() for q in range(world)]}
        torch.save({"bounds": pg.bounds, "logits": logits.detach(), "gx": x_loc.grad, "halo_mode": halo.mode,
                    "grads": {k: p.grad for k, p in model.named_parameters()}, "rows": rows,
                    "wire": halo.wire_bytes(16, 4), "ref_fraction": pg.referenced_fraction(),
                    "n_edges": pg.edge_ids.numel(), "n_pad": pg.n_pad},
                   os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def _init(rank, world, port, out_dir):
    """File rendezvous in the test's own directory: no fixed port that another process could take between the
    moment it was found free and the moment rank 0 binds it.  (MASTER_* stay set for code that reads them.)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")   # the container's hostname need not resolve
    dist.init_process_group("gloo", init_method=f"file://{os.path.join(out_dir, 'rendezvous')}", rank=rank,
                            world_size=world)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,mode,halo_mode,first_layer", [
    (2, "sum", "all_gather", False), (2, "cat", "all_gather", False), (3, "sum", "all_gather", False),
    (2, "sum", "rows", False), (3, "cat", "rows", False), (3, "sum", "auto", False),
    (2, "sum", "all_gather", True), (3, "cat", "rows", True)])
def test_partitioned_model_equals_whole_graph(world, mode, halo_mode, first_layer, tmp_path):
    """Every form of the exchange (padded all-gather / reduce-scatter, referenced rows only over all-to-all, and
    the exchange-free first layer on replicated features) gives the whole-graph logits and gradients."""
    mp.spawn(_worker, args=(world, _free_port(), mode, str(tmp_path), halo_mode, first_layer), nprocs=world, join=True)
    n, ei, x, ea, y, model = make_problem(mode)
    params = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    xg = x.clone().requires_grad_(True)
    want = O.model_forward(params, mode, xg, ei, ea, None)
    torch.nn.functional.cross_entropy(want, y, reduction="sum").backward()

    rowptr = O.csr_by_key(ei[1], ei[0], n)[0]
    parts = [torch.load(os.path.join(tmp_path, f"rank{r}.pt"), weights_only=False) for r in range(world)]
    assert parts[0]["bounds"] == O.dst_range_partition(rowptr, world)      # integer-exact partition
    assert sum(p["n_edges"] for p in parts) == ei.size(1)
    b = parts[0]["bounds"]
    assert parts[0]["n_pad"] == max(b[r + 1] - b[r] for r in range(world))
    assert len({p["halo_mode"] for p in parts}) == 1                        # "auto": every rank decided alike
    logits = torch.cat([p["logits"] for p in parts])
    assert_close(logits, want, 1e-5, what="logits")
    if not first_layer:
        gx = torch.cat([p["gx"] for p in parts])
        assert_close(gx, xg.grad, 1e-5, atol_scale=2.0, what="grad x")
    for k, v in params.items():
        for r in range(world):                                             # every rank holds the reduced gradient
            assert_close(parts[r]["grads"][k], v.grad, 1e-5, atol_scale=2.0, what=f"grad {k} on rank {r}")
    if parts[0]["halo_mode"] == "rows":
        # the row lists are integer structures: bit-exact against the oracle; what rank r needs of q is what q serves r
        for r in range(world):
            need = O.halo_row_lists(ei, b, r)
            for q in range(world):
                assert torch.equal(parts[r]["rows"]["need"][q], need[q]), f"rows rank {r} needs of {q}"
                assert torch.equal(parts[q]["rows"]["serve"][r], need[q]), f"rows rank {q} serves {r}"
            rows = sum(t.numel() for q, t in enumerate(need) if q != r)
            assert parts[r]["wire"] == {"mode": "rows", "fwd_bytes_received": rows * 16 * 4, "bwd_bytes_sent": rows * 16 * 4}
    else:
        rows = (world - 1) * parts[0]["n_pad"]
        assert parts[0]["wire"] == {"mode": "all_gather", "fwd_bytes_received": rows * 64, "bwd_bytes_sent": rows * 64}


def test_bounds_match_oracle_for_many_world_sizes():
    from mpnn_amd.partition import dst_range_bounds
    gen = torch.Generator().manual_seed(5)
    n, e = 777, 30000
    dst = (torch.rand(e, generator=gen) ** 4 * n).long().clamp_(max=n - 1)
    rowptr = O.csr_by_key(dst, torch.zeros(e, dtype=torch.int64), n)[0]
    for world in (1, 2, 3, 4, 8):
        assert dst_range_bounds(dst, n, world) == O.dst_range_partition(rowptr, world)


def _dp_worker(rank, world, port, out_dir):
    _init(rank, world, port, out_dir)
    try:
        from mpnn_amd.partition import allreduce_gradients
        torch.manual_seed(0)
        lin = torch.nn.Linear(6, 3)
        xs = torch.arange(24, dtype=torch.float32).view(4, 6) / 10
        lin(xs[rank::world]).pow(2).mean().backward()
        allreduce_gradients(lin, average=True)
        torch.save({k: p.grad for k, p in lin.named_parameters()}, os.path.join(out_dir, f"dp{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_data_parallel_gradient_allreduce(tmp_path):
    mp.spawn(_dp_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    torch.manual_seed(0)
    lin = torch.nn.Linear(6, 3)
    xs = torch.arange(24, dtype=torch.float32).view(4, 6) / 10
    (0.5 * (lin(xs[0::2]).pow(2).mean() + lin(xs[1::2]).pow(2).mean())).backward()
    for r in range(2):
        got = torch.load(os.path.join(tmp_path, f"dp{r}.pt"), weights_only=False)
        for k, p in lin.named_parameters():
            assert_close(got[k], p.grad, 1e-6, what=k)


# ---------------------------------------------------------------------------------------------
# the same check with the real HIP kernels: two ranks share the one GPU of the test box and exchange
# through gloo (device tensors staged through the host); only the RCCL transport itself is not covered
# ---------------------------------------------------------------------------------------------
def _gpu_worker(rank, world, port, mode, out_dir, halo_mode, first_layer, dtype):
    _init(rank, world, port, out_dir)
    try:
        from mpnn_amd.optim import FlatAdam
        from mpnn_amd.partition import Halo, PartitionedGraph, allreduce_gradients, partitioned_model
        dev = torch.device("cuda:0")
        n, ei, x, ea, y, model = make_problem(mode, f_in=16 if dtype == torch.bfloat16 else 10)
        model = model.to(dev)
        opt = FlatAdam(model.parameters(), lr=0.01) if dtype == torch.float32 else None
        model = model.to(dtype)
        pg = PartitionedGraph(ei.to(dev), n, rank, world)
        halo = Halo(pg, mode=halo_mode)
        xd = x.to(dev, dtype)
        x_loc = pg.local_rows(xd).clone().requires_grad_(not first_layer)
        logits = partitioned_model(model, x_loc, pg, pg.local_edges(ea.to(dev, dtype)).clone(), halo=halo,
                                   x_all=xd if first_layer else None)
        loss = torch.nn.functional.cross_entropy(logits.float(), pg.local_rows(y.to(dev)), reduction="sum")
        loss.backward()
        allreduce_gradients(opt if opt is not None else model)     # FlatAdam: the flat gradient buffer, in place
        torch.save({"bounds": pg.bounds, "logits": logits.detach().float().cpu(),
                    "gx": x_loc.grad.float().cpu() if not first_layer else None,
                    "grads": {k: p.grad.float().cpu() for k, p in model.named_parameters()}},
                   os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world,mode,halo_mode,first_layer,dtype", [
    (2, "sum", "all_gather", False, torch.float32), (3, "cat", "all_gather", False, torch.float32),
    (2, "sum", "rows", False, torch.float32), (3, "cat", "rows", True, torch.float32),
    (2, "sum", "all_gather", True, torch.float32), (2, "sum", "rows", False, torch.bfloat16),
    (2, "cat", "all_gather", False, torch.bfloat16)])
def test_partitioned_model_on_the_gpu_equals_whole_graph(world, mode, halo_mode, first_layer, dtype, tmp_path):
    """The fused per-rank layer node (real kernels, exchanges started asynchronously between its GEMMs) in every
    form of the exchange, float32 and bfloat16, against the whole-graph CPU oracle."""
    mp.spawn(_gpu_worker, args=(world, _free_port(), mode, str(tmp_path), halo_mode, first_layer, dtype),
             nprocs=world, join=True)
    bf = dtype == torch.bfloat16
    n, ei, x, ea, y, model = make_problem(mode, f_in=16 if bf else 10)
    if bf:      # the oracle sees the values the GPU holds
        x, ea = x.bfloat16().float(), ea.bfloat16().float()
        with torch.no_grad():
            for p in model.parameters():
                p.copy_(p.bfloat16().float())
    params = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    xg = x.clone().requires_grad_(True)
    want = O.model_forward(params, mode, xg, ei, ea, None)
    torch.nn.functional.cross_entropy(want, y, reduction="sum").backward()
    parts = [torch.load(os.path.join(tmp_path, f"rank{r}.pt"), weights_only=True) for r in range(world)]
    rtol = 2e-2 if bf else 1e-5
    assert_close(torch.cat([p["logits"] for p in parts]), want, rtol, what="logits")
    if not first_layer:
        assert_close(torch.cat([p["gx"] for p in parts]), xg.grad, rtol, atol_scale=10, what="grad x")
    for k, v in params.items():
        for p in parts:
            assert_close(p["grads"][k], v.grad, rtol, atol_scale=10, what=k)


@pytest.mark.gpu
def test_bench_two_ranks_as_the_driver_starts_it():
    """`python bench.py --gpus 2 ...` as a fresh child process - the command form of the multi-GPU scaling run (the
    script starts its own ranks through torch.distributed.run) - rehearsed over gloo with both ranks on one GPU
    at 1/20 scale: one JSON line from rank 0 with the rank count, the exchange volume and a positive rate."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu",
                        "--scale", "0.05", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["value"] > 0 and rec["scaling"] == "strong"
    assert rec["metric"] == "edges_per_sec_mpnn_layer_fwd_bwd" and rec["unit"] == "edges/s"
    assert rec["exchange_per_layer_rank0"]["fwd_bytes_received"] > 0
    assert "2 ranks" in rec["config"]["parallelism"] or "2" in rec["config"]["parallelism"]


@pytest.mark.gpu
@pytest.mark.parametrize("h,table_dtype,wire", [(128, torch.float32, torch.float32), (64, torch.float32, torch.bfloat16),
                                                (128, torch.bfloat16, torch.bfloat16), (20, torch.float32, torch.float32),
                                                (12, torch.bfloat16, torch.float32)])
def test_halo_pack_and_unpack_kernels(h, table_dtype, wire):
    """mpnn_halo_pack / mpnn_halo_unpack against the torch statements they replace: a row gather with conversion
    to the wire format, and per row the float32 sum of the received contributions in list order."""
    from mpnn_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(h)
    n, served = 5000, 12000
    table = torch.randn(n, 2 * h, generator=g).to(table_dtype).to(dev)[:, h:]        # a strided view, like B in [A | B]
    idx = torch.randint(0, n, (served,), generator=g).to(dev)
    send = torch.empty((served, h), dtype=wire, device=dev)
    _lib.check(lib.mpnn_halo_pack(_lib.ptr(table), table.stride(0), _lib.dtype_code(table_dtype), _lib.ptr(idx.to(torch.int32)),
                                  _lib.ptr(send), _lib.dtype_code(wire), served, h, _lib.stream_handle(dev)), "pack")
    assert torch.equal(send, table.index_select(0, idx).to(wire))
    # unpack: contributions of 3 senders, each naming a row at most once
    lists = [torch.randperm(n, generator=g)[:k] for k in (3000, 1, 4500)]
    serve_idx = torch.cat(lists).to(dev)
    recv = torch.randn(serve_idx.numel(), h, generator=g).to(wire).to(dev)
    order = torch.sort(serve_idx, stable=True).indices.to(torch.int32)
    rowptr = torch.zeros(n + 1, dtype=torch.int32, device=dev)
    rowptr[1:] = torch.cumsum(torch.bincount(serve_idx, minlength=n), 0).to(torch.int32)
    out = torch.full((n, h + 4), 7.0, dtype=table_dtype, device=dev)
    _lib.check(lib.mpnn_halo_unpack(_lib.ptr(recv), _lib.dtype_code(wire), _lib.ptr(rowptr), _lib.ptr(order), _lib.ptr(out),
                                    out.stride(0), _lib.dtype_code(table_dtype), n, h, _lib.stream_handle(dev)), "unpack")
    want = torch.zeros(n, h, dtype=torch.float32, device=dev)
    off = 0
    for lst in lists:                                   # sender order: the float32 sum of a row has this one order
        want.index_add_(0, lst.to(dev), recv[off:off + lst.numel()].float())
        off += lst.numel()
    assert torch.equal(out[:, :h], want.to(table_dtype))
    assert bool((out[:, h:] == 7.0).all())
