"""Times the two edge kernels on a power-law graph (default 1/4 of the S3DIS-scale workload)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpnn_amd import synth, ops
from mpnn_amd.graph import MPGraph
from mpnn_amd.ops import aggregate_fused

dev = torch.device("cuda:0")
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
H = int(sys.argv[2]) if len(sys.argv) > 2 else 128
kind = sys.argv[3] if len(sys.argv) > 3 else "powerlaw"
n, e = int(4_000_000 * scale), int(100_000_000 * scale)
if kind == "powerlaw":
    d = synth.powerlaw(dev, 0, n, e, 8, 4)
elif kind == "regular":   # every row has exactly AGG_DEG incoming edges from uniform random sources
    deg = int(os.environ.get("AGG_DEG", "24")); e = n * deg
    dst = torch.arange(n, device=dev).repeat_interleave(deg)
    pm = torch.randperm(e, device=dev)
    d = {"edge_index": torch.stack([torch.randint(0, n, (e,), device=dev), dst])[:, pm].contiguous(),
         "edge_attr": torch.rand(e, 4, device=dev)}
else:
    d = {"edge_index": torch.randint(0, n, (2, e), device=dev), "edge_attr": torch.rand(e, 4, device=dev)}
order = os.environ.get("AGG_ORDER", "")
if order:   # experiment: edge order inside each row of the two sorted structures
    ei = d["edge_index"]
    perm = torch.argsort(ei[0] if order == "src" else ei[0] + ei[1], stable=True) if order in ("src", "sum") else torch.randperm(e, device=dev)
    if order == "both":   # forward rows ascending in src AND transpose rows ascending in dst: sort by (src, dst) pairs
        perm = torch.argsort(ei[1], stable=True); perm = perm[torch.argsort(ei[0][perm], stable=True)]
    d["edge_index"] = ei[:, perm].contiguous(); d["edge_attr"] = d["edge_attr"][perm].contiguous()
chunk = int(os.environ.get("AGG_CHUNK", "0"))
g = MPGraph(d["edge_index"], n, chunk=chunk or None, row_order=os.environ.get("AGG_ROWS", "items"))
print("chunk", chunk or "default", "items", g.by_dst.n_items, "parts", g.by_dst.n_parts, "| transpose items", g.by_src.n_items, "parts", g.by_src.n_parts)
dt = torch.bfloat16 if os.environ.get("AGG_DTYPE") == "bf16" else torch.float32
ab = torch.randn(n, 2 * H, device=dev).to(dt).requires_grad_(True)
we = torch.randn(H, 4, device=dev).to(dt).requires_grad_(True)
r = torch.randn(n, H, device=dev).to(dt)
d["edge_attr"] = d["edge_attr"].to(dt)
for it in range(3):
    ops.EVENT_SINK = []
    for _ in range(5):
        ab.grad = None; we.grad = None
        out = aggregate_fused(ab, d["edge_attr"], we, g)
        out.backward(r)
    torch.cuda.synchronize()
    ev = ops.EVENT_SINK; ops.EVENT_SINK = None
t = {}
for name, a, b in ev:
    t.setdefault(name, []).append(a.elapsed_time(b))
s = 2 if dt == torch.bfloat16 else 4
for k, v in t.items():
    ms = sorted(v)[len(v) // 2]
    if k == "aggregate_fwd":
        by = e * (4 + H * s + 4 * s) + n * (16 + 2 * H * s + 4 * H)
    else:
        by = e * (4 + 2 * H * s + 4 * s) + n * (16 + 2 * H * s)
    print(f"{kind} scale={scale} H={H} U={os.environ.get('MPNN_FWD_U','4')} {k}: {ms:.3f} ms  {by/ms/1e6:.0f} GB/s algorithmic  {e/ms/1e6:.2f} Gedges/s")
# inference-mode forward (no cnt / mask outputs)
with torch.no_grad():
    ops.EVENT_SINK = []
    for _ in range(5):
        aggregate_fused(ab.detach(), d["edge_attr"], we.detach(), g)
    torch.cuda.synchronize()
    ev = ops.EVENT_SINK; ops.EVENT_SINK = None
ms = sorted(a.elapsed_time(b) for _, a, b in ev)[len(ev) // 2]
by = e * (4 + H * s + 4 * s) + n * (16 + 2 * H * s)
print(f"{kind} scale={scale} H={H} inference fwd: {ms:.3f} ms  {by/ms/1e6:.0f} GB/s algorithmic")
