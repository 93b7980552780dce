"""One case of tests/fuzz_islands.py looked at closely: which entries of the input gradient leave the float64 oracle,
and how close to zero the float64 pre-activations of the edges between those nodes are (a relu decision within
float32 rounding of zero can fall on the other side on the GPU, which moves the gradient of both end points by one
message's share).  Usage: python tools/island_flip_debug.py SEED CASE egnn|sum|cat F H FE GRAPHS MAX_NODES DENSE(0|1)"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import torch
import torch.nn.functional as F
import test_gpu_islands as T

seed, c = int(sys.argv[1]), int(sys.argv[2])
kind, f, h, fe, count, max_nodes, dense = sys.argv[3], *map(int, sys.argv[4:10])
egnn = kind == "egnn"
orig = T.molecules
T.molecules = lambda s, cnt, max_nodes=max_nodes, fe=4, f=11, dense=bool(dense): orig(s, cnt, max_nodes, fe, f, bool(dense))
case = T._layer_case("sum" if egnn else kind, torch.float32, f, h, fe, seed=1000 * seed + c, count=count, egnn=egnn)
T.molecules = orig
got, got_pos, gx, gpos, grads, ei = T._run(case, torch.float32, egnn)
want = case["xc"].grad
err = (gx.double().cpu() - want).abs()
scale = float(want.abs().max())
bad = torch.nonzero(err > 4e-5 * scale + 1e-5 * want.abs())
print(f"grad x: scale {scale:.4g}, {bad.size(0)} entries outside, worst {float(err.max()):.3e}")
P = {k: v.detach() for k, v in case["params"].items()}
with torch.no_grad():
    x = case["xc"].detach()
    wn = P["fused_linear.weight"] if "fused_linear.weight" in P else None
    if wn is not None:
        xs, xt = F.linear(x, wn).chunk(2, dim=1)
        es = F.linear(case["ea"].double(), P["linear_3.weight"])
        src, dst = case["ei"][0], case["ei"][1]
        pre = xs[dst] + xt[src] + es
        mag = xs[dst].abs() + xt[src].abs() + es.abs()
        print(f"edges {pre.size(0)}; smallest |pre|/magnitude over all: {float((pre.abs() / mag).min()):.3e}")
        nodes = sorted(set(bad[:, 0].tolist()))
        print("bad nodes", nodes)
        for a in nodes:
            for b in nodes:
                sel = torch.nonzero((dst == a) & (src == b)).view(-1)
                for e in sel.tolist():
                    r = pre[e].abs() / mag[e]
                    j = int(r.argmin())
                    print(f"  edge {e}: {b}->{a}  smallest |pre|/magnitude {float(r[j]):.3e} (column {j}, pre {float(pre[e, j]):.3e})")
