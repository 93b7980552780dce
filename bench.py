"""Benchmark of the MPNN layer hot path: edges/sec for one layer forward+backward.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload powerlaw|flickr|modelnet|qm9]
                  [--mode sum|cat] [--layer mpnn|egnn] [--dtype f32|bf16] [--no-cpu-baseline]

A step is one full forward + backward of one MPNN layer (node GEMMs, fused gather/message/
aggregate pass, update GEMM, and every gradient: x, all weights) on a synthetic graph whose
tensors are already in HBM.  With N > 1 the graph is partitioned by destination range and the
step includes the halo exchange of the source rows, the exchange of their gradient back to the owners and the
weight gradient all-reduce (one process per GPU); `exchange_per_layer_rank0` gives the bytes on the wire.  Started without a launcher (`python bench.py --gpus N`, no
WORLD_SIZE in the environment) the script starts N ranks itself through torch.distributed.run, as a child
process and before anything touches the GPU.

After the timed region the single-GPU run checks itself: one more forward + backward with an upstream
gradient that is zero outside a sample of target rows is compared, output rows and every gradient, with the
CPU oracle evaluated on the edges that touch the sample (`verified`); a mismatch makes the run fail.

Rank 0 prints ONE JSON line.  `roofline` describes the dominant HIP kernel (timed with events on
the stream it runs on, inside the timed region); `cpu_baseline` is the CPU oracle timed on the
host cores on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak

WORKLOADS = {
    # name: (generator kwargs, in_features, hidden, edge_features)
    "powerlaw": dict(F=128, H=128, Fe=4, desc="S3DIS-scale power-law graph, 4,000,000 nodes, 100,000,000 edges"),
    "flickr": dict(F=500, H=64, Fe=4, desc="Flickr-shaped graph, 89,250 nodes, 899,756 edges, layer 1 (500->64)"),
    "modelnet": dict(F=3, H=64, Fe=1, desc="ModelNet-shaped batch, 32 clouds x 1024 points, k-NN k=20"),
    "qm9": dict(F=11, H=64, Fe=4, desc="QM9-shaped batch of 1024 molecules (~18 atoms, ~38 directed bonds)"),
}


def make_data(name, device, scale=1.0, seed=0, with_features=True):
    """with_features=False (power-law graph only): edge list without node features and edge attributes - the
    ranks of a partitioned run draw their own share of those afterwards (synth.powerlaw_features / _edge_attr)."""
    from mpnn_amd import synth
    if name == "powerlaw":
        n = max(int(4_000_000 * scale), 1000)
        e = max(int(100_000_000 * scale), 10_000)
        return synth.powerlaw(device, seed, n, e, 128, 4, with_features=with_features)
    if name == "flickr":
        return synth.flickr(device, seed, max(int(89_250 * scale), 100), max(int(899_756 * scale) // 2 * 2, 200))
    if name == "modelnet":
        return synth.modelnet(device, seed, max(int(32 * scale), 1))
    if name == "qm9":
        return synth.qm9(device, seed, max(int(1024 * scale), 1))
    raise ValueError(name)


def layer_bytes_model(n, e, h, fe, s):
    """Three-pass algorithmic traffic of one layer fwd+bwd (gather, relu-mask recompute, no atomics):
    E*(16 + 4*H*s + 3*Fe*s) + N*(12 + 7*H*s)."""
    return e * (16 + 4 * h * s + 3 * fe * s) + n * (12 + 7 * h * s)


def kernel_bytes(kind, n_rows, e, h, fe, s, mask_words=0, f=0):
    """Compulsory bytes of one launch of this build's two edge kernels.

    aggregate_fwd       per edge: col, B[src] row, edge_attr (+ relu mask words written)
                        per row:  item record, A row, out row, cnt row (one byte per column)
    aggregate_bwd_src   recomputing form, per edge: col_t, A[dst] row, G[dst] row, edge_attr
                        mask-bit form,    per edge: col_t, tpos, mask words, G[dst] row, edge_attr
                        per row:  item record, (B row,) gB row
    """
    if kind == "egnn_fwd":         # per edge: col, B[src] row, edge_attr, pos[src]; per row: item, A, out, pos in/out
        return e * (4 + h * s + fe * s + 12) + n_rows * (16 + 2 * h * s + 24)
    if kind == "egnn_bwd_dst":     # the forward's reads again + per row: upstream gradient row and g pos in, gA and S out
        return e * (4 + h * s + fe * s + 12) + n_rows * (16 + 3 * h * s + 40)
    if kind == "egnn_bwd_src":     # per edge: col_t, A[dst] and G[dst] rows, edge_attr, pos[dst], g pos[dst]
        return e * (4 + 2 * h * s + fe * s + 24) + n_rows * (16 + 2 * h * s + 24)
    # whole layers in one launch (batches of small graphs): the rows of x in and of y out, the edge list once; the
    # backward reads x, gy and y, writes gx.  f stands for the input width.
    if kind == "island_layer_fwd":
        return e * (4 + fe * s) + n_rows * (4 + (f + h) * s)
    if kind == "island_layer_bwd":
        return e * (4 + fe * s) + n_rows * (4 + (2 * f + 2 * h) * s)
    # whole layers for point clouds: the forward like the island form; the backward walks both sorted views and
    # passes the gradient of the sums (float rows) from its first launch to its second, which gathers it per edge
    if kind == "region_layer_fwd":
        return e * (4 + fe * s) + n_rows * (4 + (f + h) * s)
    if kind == "region_layer_bwd":
        return e * (8 + 2 * fe * s + 4 * h) + n_rows * (8 + (2 * f + 2 * h) * s + 4 * h)
    if kind == "aggregate_fwd":
        return e * (4 + h * s + fe * s + 4 * mask_words) + n_rows * (16 + 2 * h * s + h)
    if kind == "aggregate_bwd_src":
        if mask_words:
            return e * (8 + 4 * mask_words + h * s + fe * s) + n_rows * (16 + h * s)
        return e * (4 + 2 * h * s + fe * s) + n_rows * (16 + 2 * h * s)
    raise ValueError(kind)


F32_MATRIX_PEAK_TFLOPS = 157.3   # v_mfma_f32_32x32x2f32, the instruction of the island kernels' dense products


def island_flops(kind, n_rows, e, f, h, fe):
    """Algorithmic floating-point operations of one launch of the whole-layer kernels (real rows and edges, not the
    32-row tiles they are padded to): forward = projection + messages + update; backward = the projection and the
    messages again, gz W_out, the two weight-gradient products, [gA | gB] Wn and the backward edge pass."""
    proj, upd, edge = 2 * n_rows * 2 * h * f, 2 * n_rows * h * (f + h), e * h * (2 * fe + 3)
    if kind in ("island_layer_fwd", "region_layer_fwd"):
        return proj + upd + edge
    return (proj + edge) + 2 * upd + 2 * proj + e * h * (2 * fe + 3 + 2 * fe + 2)


def pmc_traffic(workload, dtype, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (separate FETCH_SIZE / WRITE_SIZE
    runs of this script, corrected for gfx950 as profiles/*_pmc_traffic.json records), or None."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        try:
            with open(path) as fh:
                rec = json.load(fh)
        except (OSError, ValueError):
            continue
        if rec.get("dtype") != dtype:
            continue
        for v in rec.get("kernels", {}).get(workload, {}).values():
            if v.get("bench_kernel") == kernel:
                return v["hbm_bytes_per_launch"], (f"{os.path.relpath(path, ROOT)}: separate rocprofv3 --pmc passes of "
                                                   "this command in the builder's session, not counters of this run")
    return None, None


# Sample of the workload the CPU oracle is timed on: (scale of the generator, timed runs after one warm-up).
# The cat form materialises [E, 2F+Fe] rows and their gradient, six times the sum form's work per edge, so its
# sample of the power-law graph is smaller.
CPU_SAMPLE = {
    ("powerlaw", "sum"): (0.1, 1), ("powerlaw", "cat"): (0.02, 1),
    ("flickr", "sum"): (1.0, 5), ("flickr", "cat"): (1.0, 2),
    ("modelnet", "sum"): (1.0, 5), ("modelnet", "cat"): (1.0, 5),
    ("qm9", "sum"): (1.0, 5), ("qm9", "cat"): (1.0, 5),
}


def cpu_baseline(name, mode, layer="mpnn"):
    """CPU oracle (plain torch ops, the arithmetic the reference runs through PyG) on a bounded sample."""
    from oracle import reference_oracle as O
    cfg = WORKLOADS[name]
    scale, runs = CPU_SAMPLE[(name, mode)]
    d = make_data(name, "cpu", scale=scale, seed=0)
    x, ei, ea = d["x"].float(), d["edge_index"], d["edge_attr"].float()
    F_in, H, Fe = x.size(1), cfg["H"], ea.size(1)
    gen = torch.Generator().manual_seed(1)
    w_out = (torch.rand(H, H + F_in, generator=gen) - 0.5).requires_grad_(True)
    x = x.requires_grad_(True)
    r = torch.randn(x.size(0), H, generator=gen)
    if layer == "egnn":
        pos = (d["pos"].float() if "pos" in d else torch.randn(x.size(0), 3, generator=gen)).requires_grad_(True)
        rp = torch.randn(x.size(0), 3, generator=gen)
        ws = [(torch.rand(2 * H, F_in, generator=gen) - 0.5).requires_grad_(True),
              (torch.rand(H, Fe, generator=gen) - 0.5).requires_grad_(True), w_out,
              ((torch.rand(H, generator=gen) - 0.5) / H).requires_grad_(True)]

        def step():
            ab = torch.nn.functional.linear(x, ws[0])
            aggr, pos_out = O.egnn_layer(ab[:, :H], ab[:, H:], ea, ws[1], pos, ws[3], ei)
            out = O.update(x, aggr, ws[2])
            torch.autograd.grad((out * r).sum() + (pos_out * rp).sum(), [x, pos] + ws)
        what = "EGNN layer (sum-form message + coordinate update)"
    else:
        if mode == "cat":
            ws = [(torch.rand(H, 2 * F_in + Fe, generator=gen) - 0.5).requires_grad_(True), w_out]
            fn = lambda xx: O.layer_cat(xx, ei, ea, ws[0], ws[1])
        else:
            ws = [(torch.rand(2 * H, F_in, generator=gen) - 0.5).requires_grad_(True),
                  (torch.rand(H, Fe, generator=gen) - 0.5).requires_grad_(True), w_out]
            fn = lambda xx: O.layer_sum(xx, ei, ea, ws[0], ws[1], ws[2])

        def step():
            out = fn(x)
            torch.autograd.grad((out * r).sum(), [x] + ws)
        what = f"MPNNLayer{mode.capitalize()}"

    step()  # warm-up
    times = []
    for _ in range(runs):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    return {
        "value": ei.size(1) / med, "unit": "edges/s", "cores": torch.get_num_threads(), "kind": "port",
        "sample": f"{name} generator at scale {scale} (N={x.size(0)}, E={ei.size(1)}), {what} "
                  f"fwd+bwd with plain torch CPU ops, median of {len(times)} run(s) after 1 warm-up",
    }


def verify_layer(layer, mode, x, ei, ea, n_total):
    """One more forward + backward of `layer`, checked against the CPU oracle on a sample of rows.

    The upstream gradient is zero outside the sampled target rows (the largest hub, four more rows of the top 64 and
    2,000 random rows), so every gradient - input rows and all weights - depends only on the edges into the
    sample, which the oracle evaluates in float64 (oracle/sampled.py); the kernels still walk every edge."""
    from oracle import reference_oracle as O
    from oracle import sampled as S
    dev = x.device
    H = layer.out_channels
    rows = S.sample_rows(ei, n_total, 1, 2000, hubs=1, seed=7)
    gen = torch.Generator(device=dev).manual_seed(8)
    gy_rows = torch.randn(rows.numel(), H, generator=gen, device=dev)
    gy = torch.zeros(n_total, H, device=dev)
    gy[rows] = gy_rows
    params = list(layer.parameters())
    for p in params:
        p.grad = None
    xg = x.detach().float().requires_grad_(True)
    y = layer(xg, ei, ea.float())
    y.backward(gy)
    fn = O.layer_sum if mode == "sum" else O.layer_cat
    want = S.layer_sparse_upstream(fn, xg, ei, ea, [p.detach() for p in params], rows, gy_rows)
    nodes = want["nodes"]
    pairs = [("y[rows]", y.detach()[rows], want["y_rows"]), ("grad x[nodes]", xg.grad[nodes], want["gx"])]
    pairs += [(f"grad {k}", p.grad, w) for (k, _), p, w in zip(layer.named_parameters(), params, want["gw"])]
    rep = S.error_report(pairs, 1e-5)
    untouched = torch.ones(n_total, dtype=torch.bool, device=dev)
    untouched[nodes] = False
    stray = float(xg.grad[untouched].abs().max()) if bool(untouched.any()) else 0.0
    if stray != 0.0:
        rep["ok"], rep["worst"] = False, f"input-gradient rows no sampled edge reaches are not zero (max {stray})"
    for p in params:
        p.grad = None
    return {"ok": rep["ok"], "rows": int(rows.numel()), "input_rows": int(nodes.numel()),
            "edges": want["num_edges"], "entries": rep["entries"], "max_rel_err": rep["max_rel_err"],
            "outside_tolerance": rep["outside"], "tolerance": "|got-want| <= 1e-5*max|want| + 1e-5*|want| against "
            "the float64 CPU oracle; at most 1e-5 of a tensor's entries outside (relu kinks)",
            "what": "y rows, grad x, grad of every weight; sparse upstream gradient on the sampled target rows",
            "worst": rep["worst"]}


def verify_layer_bf16(layer, mode, x, ei, ea, n_total):
    """verify_layer for bfloat16 storage: the float64 oracle on the bfloat16-rounded inputs and weights, bar 2e-2.
    bfloat16 rows move about one relu decision in 300 against a float64 evaluation and each of those moves the
    gradients it feeds by a whole term, so a gradient entry outside the bar passes only within what the undecided
    relus that reach it explain (oracle/flip_bounds.py); the output rows get no such allowance."""
    from oracle import flip_bounds as FB
    from oracle import sampled as S
    dev, bf = x.device, torch.bfloat16
    H = layer.out_channels
    rows = S.sample_rows(ei, n_total, 1, 1000, hubs=1, seed=7)
    gen = torch.Generator(device=dev).manual_seed(8)
    gy_rows = torch.randn(rows.numel(), H, generator=gen, device=dev).to(bf)
    gy = torch.zeros(n_total, H, device=dev, dtype=bf)
    gy[rows] = gy_rows
    params = list(layer.parameters())
    for p in params:
        p.grad = None
    xb, eb = x.detach().to(bf), ea.to(bf)
    xg = xb.clone().requires_grad_(True)
    y = layer(xg, ei, eb)
    y.backward(gy)
    f = xb.size(1)
    if mode == "sum":
        wn, we, wo = (p.detach() for p in params)
    else:
        wn, we = FB.cat_to_sum_weights(params[0].detach(), f)
        wo = params[1].detach()
    want = S.layer_sum_sparse_upstream_with_flip_bounds(xb, ei, eb, wn, we, wo, rows, gy_rows, FB.DELTA[bf])
    nodes = want["nodes"]
    entries = [("y[rows]", y.detach()[rows].float(), (want["y_rows"], None, None)),
               ("grad x[nodes]", xg.grad[nodes].float(), want["gx"], 4.0)]
    if mode == "sum":
        entries += [(f"grad {k}", p.grad.float(), want[w], 4.0)
                    for (k, _), p, w in zip(layer.named_parameters(), params, ("gw_nodes", "gw_edge", "gw_out"))]
    else:
        gw = tuple(FB.cat_grad_from_sum(a, b, H) for a, b in zip(want["gw_nodes"], want["gw_edge"]))
        entries += [("grad linear", params[0].grad.float(), gw, 4.0), ("grad linear_out", params[1].grad.float(), want["gw_out"], 4.0)]
    rep = FB.report(entries, 2e-2)
    untouched = torch.ones(n_total, dtype=torch.bool, device=dev)
    untouched[nodes] = False
    stray = float(xg.grad[untouched].float().abs().max()) if bool(untouched.any()) else 0.0
    if stray != 0.0:
        rep["ok"], rep["worst"] = False, f"input-gradient rows no sampled edge reaches are not zero (max {stray})"
    for p in params:
        p.grad = None
    return {"ok": rep["ok"], "rows": int(rows.numel()), "input_rows": int(nodes.numel()), "edges": want["num_edges"],
            "entries": rep["entries"], "max_rel_err": rep["max_rel_err"], "undecided_message_relus": want["open"],
            "entries_passed_through_the_flip_allowance": rep["through_flips"],
            "tolerance": "|got-want| <= 2e-2*max|want| (x4 for gradients) + 2e-2*|want| against the float64 CPU oracle "
                         "on the bfloat16-rounded operands; a gradient entry beyond that only within the total (or 8 "
                         "sigma of the spread) of the terms behind relus that are zero to bfloat16 rounding",
            "what": "y rows, grad x, grad of every weight; sparse upstream gradient on the sampled target rows",
            "worst": rep["worst"]}


def verify_egnn_layer(layer, x, pos, ei, ea, dtype):
    """One more forward + backward of the EGNN layer against the float64 CPU oracle on the WHOLE batch (the batched
    workloads are small enough): outputs, coordinates and every gradient.  No reference code exists for this layer
    (parity unpinned): the oracle is the formula of oracle/reference_oracle.py::egnn_full_layer."""
    from oracle import reference_oracle as O
    from oracle import sampled as S
    dev = x.device
    H = layer.out_channels
    gen = torch.Generator(device=dev).manual_seed(9)
    r = torch.randn(x.size(0), H, generator=gen, device=dev)
    rp = torch.randn(x.size(0), 3, generator=gen, device=dev)
    params = list(layer.parameters())
    for p in params:
        p.grad = None
    xg = x.detach().to(dtype).requires_grad_(True)
    pg = pos.detach().float().requires_grad_(True)
    y, pos_out = layer(xg, pg, ei, ea.to(dtype))
    ((y.float() * r).sum() + (pos_out * rp).sum()).backward()
    ws = [p.detach().double().cpu().requires_grad_(True) for p in params]   # fused_linear, linear_3, linear_out, coord_linear
    xc, pc = xg.detach().double().cpu().requires_grad_(True), pg.detach().double().cpu().requires_grad_(True)
    want_y, want_pos = O.egnn_full_layer(xc, pc, ei.cpu(), ea.to(dtype).double().cpu(), ws[0], ws[1], ws[2], ws[3])
    ((want_y * r.double().cpu()).sum() + (want_pos * rp.double().cpu()).sum()).backward()
    bf = dtype == torch.bfloat16
    rtol = 2e-2 if bf else 1e-5
    # gradients with the floor the GPU tests use (tests/test_gpu_egnn.py): 4 * rtol * max|want|
    pairs = [("y", y, want_y), ("pos'", pos_out, want_pos), ("grad x", xg.grad, xc.grad, 4.0),
             ("grad pos", pg.grad, pc.grad, 4.0)]
    pairs += [(f"grad {k}", p.grad, w.grad, 4.0) for (k, _), p, w in zip(layer.named_parameters(), params, ws)]
    rep = S.error_report(pairs, rtol, kink_frac=2e-3 if bf else 1e-5)
    for p in params:
        p.grad = None
    return {"ok": rep["ok"], "rows": int(x.size(0)), "edges": int(ei.size(1)), "entries": rep["entries"],
            "max_rel_err": rep["max_rel_err"], "outside_tolerance": rep["outside"],
            "tolerance": f"|got-want| <= {rtol}*max|want| (4x for gradients) + {rtol}*|want| against the float64 CPU oracle "
                         "(whole batch); "
                         f"at most {'2e-3' if bf else '1e-5'} of a tensor's entries outside (relu decisions that "
                         "differ in the storage precision)",
            "what": "y, pos', grad x, grad pos, grad of every weight; dense upstream gradients; parity unpinned "
                    "(no EGNN in the reference)", "worst": rep["worst"]}


def model_step_time(mode, d, hidden, dev, steps=5, warmup=2):
    """Whole training step of the reference's loop (training/train.py:43-49) on the workload's graph: 2 layers + head
    (models/mpnn.py:86-101), cross-entropy over the masked rows, backward, Adam.  Returns seconds per step."""
    from mpnn_amd import MPNN
    from mpnn_amd.loss import masked_cross_entropy
    from mpnn_amd.optim import FlatAdam
    x, ei, ea, n = d["x"].detach().float(), d["edge_index"], d["edge_attr"].float(), d["num_nodes"]
    classes = 7
    gen = torch.Generator(device=dev).manual_seed(5)
    y = d.get("y")
    if y is None or y.dim() != 1 or y.numel() != n or y.dtype != torch.int64:
        y = torch.randint(0, classes, (n,), generator=gen, device=dev)
    mask = d.get("train_mask")
    if mask is None:
        mask = torch.rand(n, generator=gen, device=dev) < 0.5
    torch.manual_seed(0)
    model = MPNN(x.size(1), hidden, classes, num_layers=2, mode=mode).to(dev)
    opt = FlatAdam(model.parameters(), lr=0.01)

    def step():
        opt.zero_grad()
        loss = masked_cross_entropy(model(x, ei, ea), y, mask)
        loss.backward()
        opt.step()

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks through torch.distributed.run as a CHILD process
    (nothing in this process has touched the GPU yet, and it never will) and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    note(f"no launcher in the environment: starting {args.gpus} ranks with torch.distributed.run on port {port}")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def note(msg):
    """Progress line on stderr (stdout carries only the JSON result)."""
    print(f"[bench r{os.environ.get('RANK', '0')}] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="powerlaw", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="sum", choices=["sum", "cat"])
    ap.add_argument("--layer", default="mpnn", choices=["mpnn", "egnn"],
                    help="egnn: EGNNLayer (message pass + coordinate update in one edge pass; one GPU, or data "
                         "parallel with --workload qm9 | modelnet)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the workload (debugging only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the sampled check against the CPU oracle")
    ap.add_argument("--no-model-step", action="store_true", help="skip the whole-model training step timing")
    ap.add_argument("--hip-graph", action="store_true",
                    help="capture one step (forward+backward) into a HIP graph and time replays (single GPU)")
    ap.add_argument("--backward-on-caller", action="store_true",
                    help="run the backward on the calling thread (torch.autograd.set_multithreading_enabled(False)): "
                         "no hand-over to autograd's device thread, which matters for launch-bound eager steps")
    ap.add_argument("--first-layer", action="store_true",
                    help="time the layer as the FIRST layer of a model: x is data, no gradient with respect to x "
                         "(default: a middle layer, every gradient including the input's)")
    ap.add_argument("--halo", default="auto", choices=["auto", "all_gather", "rows"],
                    help="partitioned runs: exchange every source row (padded all-gather) or only the rows a rank's "
                         "edges read (all-to-all); auto picks rows when the ranks read less than 3/4 of the rows")
    ap.add_argument("--halo-wire", default="native", choices=["native", "bf16"],
                    help="partitioned runs: dtype of the rows on the wire (bf16 halves the bytes and leaves the "
                         "single-GPU numerics)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses GPU 0")
    args = ap.parse_args()

    if os.environ.get("BENCH_WATCHDOG"):  # debugging aid: dump every thread's stack and exit after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["BENCH_WATCHDOG"]), exit=True)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus differ")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    from mpnn_amd import MPNNLayerCat, MPNNLayerSum, ops
    from mpnn_amd.partition import Halo, PartitionedGraph, allreduce_gradients, partitioned_layer

    cfg = WORKLOADS[args.workload]
    dtype = torch.float32 if args.dtype == "f32" else torch.bfloat16
    s = 4 if args.dtype == "f32" else 2

    # batched small-graph workloads scale by data parallelism: every rank trains on its own batch and the only
    # exchange is the gradient all-reduce (weak scaling); single large graphs are partitioned (strong scaling)
    data_parallel = world > 1 and args.workload in ("qm9", "modelnet")
    note("generating the synthetic graph")
    # a rank of a partitioned power-law run draws only its own rows of x and its own edges' attributes (below)
    own_share = world > 1 and args.workload == "powerlaw" and not args.first_layer
    d = make_data(args.workload, dev, scale=args.scale, seed=rank if data_parallel else 0, with_features=not own_share)
    n_total, e_total = d["num_nodes"], d["edge_index"].size(1)
    H = cfg["H"]
    F_in, Fe = (128, 4) if own_share else (d["x"].size(1), d["edge_attr"].size(1))
    torch.manual_seed(0)
    if args.layer == "egnn":
        if world > 1 and not data_parallel:
            raise SystemExit("--layer egnn on several GPUs needs a batched workload (--workload qm9 | modelnet): "
                             "those scale by data parallelism; the partitioned single-graph path is MPNN only")
        from mpnn_amd import EGNNLayer
        layer = EGNNLayer(F_in, Fe, H).to(dev, dtype)
        pos = d["pos"].float() if "pos" in d else torch.randn(n_total, 3, device=dev)
        pos = pos.clone().requires_grad_(True)
    else:
        layer = (MPNNLayerSum if args.mode == "sum" else MPNNLayerCat)(F_in, Fe, H).to(dev, dtype)

    if world == 1 or data_parallel:
        x = d["x"].to(dtype).requires_grad_(not args.first_layer)
        ei, ea = d["edge_index"], d["edge_attr"].to(dtype)
        from mpnn_amd.graph import graph_for, make_graph
        graph_for(ei, n_total)            # built once and cached, outside the timed region (static graph)
        # cost of building the structures (CSR, transpose, work items) from edge_index, for per-batch graphs
        for _ in range(2):
            make_graph(ei, n_total)
        torch.cuda.synchronize()
        tb = time.perf_counter()
        for _ in range(5):
            make_graph(ei, n_total)
        torch.cuda.synchronize()
        graph_build_ms = (time.perf_counter() - tb) / 5 * 1e3
        n_rows_fwd, n_rows_bwd, e_local = n_total, n_total, e_total
        forward = (lambda: layer(x, pos, ei, ea)) if args.layer == "egnn" else (lambda: layer(x, ei, ea))
    else:
        graph_build_ms = None
        pg = PartitionedGraph(d["edge_index"], n_total, rank, world)
        # every rank generated the graph on its own: the halo exchanges assume identical partitions, so compare
        # them (a mismatch would otherwise show up as a hang inside a collective)
        sig = torch.tensor([pg.n_pad, sum(pg.bounds), int(d["edge_index"].sum().item() % (1 << 40))],
                           dtype=torch.int64, device=dev if args.backend == "nccl" else "cpu")
        lo_sig, hi_sig = sig.clone(), sig.clone()
        dist.all_reduce(lo_sig, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi_sig, op=dist.ReduceOp.MAX)
        if not torch.equal(lo_sig, hi_sig):
            raise SystemExit(f"rank {rank}: the ranks built different graphs or partitions ({sig.tolist()})")
        if own_share:
            from mpnn_amd import synth
            x = synth.powerlaw_features(dev, 0, n_total, F_in, pg.lo, pg.hi, dtype).requires_grad_(True)
            ea = synth.powerlaw_edge_attr(dev, 0, e_total, Fe, pg.edge_ids, dtype)
        else:
            x = pg.local_rows(d["x"]).to(dtype).clone().requires_grad_(not args.first_layer)
            ea = pg.local_edges(d["edge_attr"]).to(dtype).clone()
        if args.first_layer:
            # the layer reads the dataset's features, which every rank holds: it projects all source rows itself
            # and nothing is exchanged but the weight gradients
            halo, x_all = None, d["x"].to(dtype).clone()
            exchange = {"mode": "none (first layer on replicated features)", "fwd_bytes_received": 0, "bwd_bytes_sent": 0}
            n_rows_bwd = n_total
        else:
            halo = Halo(pg, mode=args.halo, wire_dtype=torch.bfloat16 if args.halo_wire == "bf16" else None)
            x_all = None
            exchange = halo.wire_bytes(H, s)
            n_rows_bwd = halo.graph.num_src
        exchange["referenced_fraction_of_rows"] = pg.referenced_fraction()
        n_rows_fwd, e_local = pg.n_local, pg.edge_ids.numel()
        forward = lambda: partitioned_layer(layer, x, pg, ea, halo=halo, x_all=x_all)
    keep = d if world == 1 else None      # the single-GPU run checks itself and times a model step afterwards
    del d
    torch.cuda.empty_cache()
    note("graph structures built; warming up")
    gen = torch.Generator(device=dev).manual_seed(2)
    r = torch.randn(x.size(0), H, generator=gen, device=dev).to(dtype)
    rp = torch.randn(x.size(0), 3, generator=gen, device=dev) if args.layer == "egnn" else None

    def step():
        for p in layer.parameters():
            p.grad = None
        x.grad = None
        out = forward()
        if args.layer == "egnn":
            pos.grad = None
            torch.autograd.backward(list(out), [r, rp])   # dense upstream gradients for features and coordinates
        else:
            out.backward(r)      # dense upstream gradient, as a following layer would supply
        if world > 1:
            allreduce_gradients(layer, average=data_parallel)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.backward_on_caller:
        torch.autograd.set_multithreading_enabled(False)
    for _ in range(args.warmup):
        step()
    fence()
    run_step = step
    if args.hip_graph and world == 1:
        # The library calls are plain stream-ordered launches (no allocation, no synchronisation), so a whole
        # step can be captured once and replayed; graph structures were built and cached before the capture.
        note("capturing one step into a HIP graph")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        hip_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(hip_graph):
            step()
        run_step = hip_graph.replay
        run_step()
        fence()
    note("timing")
    ops.EVENT_SINK = [] if run_step is step else None
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    cur = torch.cuda.current_stream(dev)
    t0 = time.perf_counter()
    marks[0].record(cur)
    for i in range(args.steps):
        run_step()
        marks[i + 1].record(cur)
    fence()
    elapsed = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    median_ms = step_ms[len(step_ms) // 2] if step_ms else None
    events, ops.EVENT_SINK = ops.EVENT_SINK or [], None
    if run_step is not step:
        # graph replays carry no events: time the kernels over a few eager steps after the timed region
        ops.EVENT_SINK = []
        for _ in range(5):
            step()
        fence()
        events, ops.EVENT_SINK = ops.EVENT_SINK, None
    e_job = e_total
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if data_parallel:   # every rank holds its own batch: the job's edges are the sum over the ranks
            t = torch.tensor([float(e_total)], dtype=torch.float64, device=t.device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            e_job = int(t.item())

    # per-kernel time from the events recorded on the launch stream during the timed steps
    kt = {}
    for name, a, b in events:
        kt.setdefault(name, []).append(a.elapsed_time(b) * 1e-3)
    kavg = {k: sum(v) / len(v) for k, v in kt.items()}

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        rows = {"aggregate_fwd": n_rows_fwd, "aggregate_bwd_src": n_rows_bwd, "egnn_fwd": n_rows_fwd,
                "egnn_bwd_dst": n_rows_fwd, "egnn_bwd_src": n_rows_bwd, "island_layer_fwd": n_rows_fwd,
                "island_layer_bwd": n_rows_fwd, "region_layer_fwd": n_rows_fwd, "region_layer_bwd": n_rows_fwd}
        per_kernel = {}
        from mpnn_amd import _lib
        mask_words = _lib.load().mpnn_mask_words(H, _lib.dtype_code(dtype)) if ops.USE_MASKBITS else 0
        for k, tk in kavg.items():
            b = kernel_bytes(k, rows[k], e_local, H, Fe, s, mask_words, F_in)
            per_kernel[k] = {"avg_ms": tk * 1e3, "algorithmic_bytes": b, "GBps": b / tk / 1e9}
        dom = max(kavg, key=kavg.get)
        traffic = traffic_src = None
        if world == 1 and args.scale == 1.0 and args.mode == "sum":
            traffic, traffic_src = pmc_traffic(args.workload, args.dtype, dom)
        roofline = {"bound": "hbm", "kernel": dom, "achieved": per_kernel[dom]["GBps"], "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": per_kernel[dom]["GBps"] / HBM_PEAK_GBS,
                    "traffic": traffic, "traffic_source": traffic_src}
        if dom.startswith(("island_layer", "region_layer")):
            # a batch of small graphs runs as whole-layer kernels whose tiles never leave the LDS: their bytes are a
            # few megabytes and the limit is the fp32 matrix pipe (and, at this size, launch latency)
            tf = island_flops(dom, rows[dom], e_local, F_in, H, Fe) / kavg[dom] / 1e12
            roofline = {"bound": "mfma", "kernel": dom, "achieved": tf, "peak": F32_MATRIX_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": tf / F32_MATRIX_PEAK_TFLOPS, "traffic": None,
                        "note": "algorithmic flops of the real rows and edges; the backward span includes the "
                                "reduction of the weight-gradient partials"}
        mp_time = sum(kavg.values())
        model_bytes = layer_bytes_model(n_rows_fwd, e_local, H, Fe, s)
        result = {
            "metric": "edges_per_sec_mpnn_layer_fwd_bwd",
            "value": e_job / (elapsed / args.steps),
            "unit": "edges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "ms_per_step_median_events": median_ms,
            "higher_is_better": True,
            "scaling": "weak" if data_parallel else "strong",
            "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.workload}: {cfg['desc']}" + (f" (scale {args.scale})" if args.scale != 1 else ""),
                       "layer": ("EGNNLayer" if args.layer == "egnn" else f"MPNNLayer{args.mode.capitalize()}")
                                + (" (first layer: no input gradient)" if args.first_layer else ""),
                       "num_nodes": n_total, "num_edges": e_total,
                       "in_features": F_in, "hidden": H, "edge_features": Fe,
                       "parallelism": "single GPU" if world == 1 else (
                           (f"data parallel x{world}: one batch per rank + gradient all-reduce" if data_parallel
                            else f"dst-range partition x{world} + halo exchange ({exchange['mode']})")
                           + f" ({dist.get_backend()} reports {dist.get_world_size()} ranks)"),
                       "launch": "HIP graph replay" if run_step is not step else (
                           "eager, backward on the calling thread" if args.backward_on_caller else "eager")},
            "roofline": roofline,
            "kernels": per_kernel,
            "message_passing_only": {
                "ms": mp_time * 1e3, "edges_per_sec": e_local / mp_time,
                "three_pass_model_bytes": model_bytes,
                "three_pass_model_frac_of_hbm_peak": model_bytes / mp_time / 1e9 / HBM_PEAK_GBS},
        }
        if world > 1 and not data_parallel:
            result["exchange_per_layer_rank0"] = exchange
        if graph_build_ms is not None:
            result["graph_build"] = {
                "ms": graph_build_ms,
                "note": "dst-sorted CSR + src-sorted transpose + work items from edge_index; paid once for a static "
                        "graph, once per mini-batch for batched workloads",
                "edges_per_sec_if_rebuilt_every_step": e_total / (elapsed / args.steps + graph_build_ms * 1e-3)}
        failed = None
        if world == 1 and keep is not None and args.layer == "mpnn" and not args.no_verify:
            note("checking a sample of rows against the CPU oracle")
            if args.dtype == "f32":
                result["verified"] = verify_layer(layer, args.mode, keep["x"], keep["edge_index"], keep["edge_attr"],
                                                  n_total)
                if not result["verified"]["ok"]:
                    failed = f"verification against the CPU oracle failed: {result['verified']['worst']}"
            else:
                result["verified"] = verify_layer_bf16(layer, args.mode, keep["x"], keep["edge_index"],
                                                       keep["edge_attr"], n_total)
                if not result["verified"]["ok"]:
                    failed = f"verification against the CPU oracle failed: {result['verified']['worst']}"
        if world == 1 and keep is not None and args.layer == "egnn" and not args.no_verify and e_total <= 4_000_000:
            note("checking the EGNN layer against the CPU oracle")
            kpos = keep["pos"].float() if "pos" in keep else pos.detach()
            result["verified"] = verify_egnn_layer(layer, keep["x"], kpos, keep["edge_index"], keep["edge_attr"], dtype)
            if not result["verified"]["ok"]:
                failed = f"verification against the CPU oracle failed: {result['verified']['worst']}"
        # (the node-level model fixes the edge width at 4, models/mpnn.py:72-81: workloads with other widths have no
        # such model step)
        if world == 1 and keep is not None and args.layer == "mpnn" and args.dtype == "f32" and not args.no_model_step \
                and keep["edge_attr"].size(1) == 4:
            note("timing the whole-model training step")
            del x, r
            layer = None
            torch.cuda.empty_cache()
            t_model = model_step_time(args.mode, keep, H, dev)
            result["model_step"] = {
                "ms": t_model * 1e3, "edges_per_sec": 2 * e_total / t_model,
                "what": "MPNN(2 layers, hidden %d) + head + masked cross-entropy + backward + Adam on the same graph "
                        "(training/train.py:43-49); edges_per_sec counts both layers" % H}
        keep = None
        if world == 1 and not args.no_cpu_baseline:
            note("timing the CPU oracle")
            result["cpu_baseline"] = cpu_baseline(args.workload, args.mode, args.layer)
            if args.layer == "mpnn":
                other = "cat" if args.mode == "sum" else "sum"
                result["cpu_baseline_other_form"] = cpu_baseline(args.workload, other)
        print(json.dumps(result), flush=True)
        if failed:
            raise SystemExit(failed)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
