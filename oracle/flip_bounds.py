"""Gradient checks that know about relu decisions at zero.  TEST INFRASTRUCTURE ONLY.

The layers are relu networks without biases (models/layers.py:56,69,128,140).  A pre-activation that is zero to
within the rounding of its summands may fall on either side in two correct evaluations (float32 against float64,
and far more often bfloat16 storage against float64); the forward output barely moves, but every gradient that
passes through that relu gains or loses one WHOLE term.  A blanket allowance ("5 % of the entries may miss the
bar") would hide a wrong column as well.  This module restates one layer's forward and backward explicitly in
float64 (same formulas as reference_oracle.layer_sum / egnn_full_layer, which the tests check it against) and
propagates, next to every gradient, how much of it hangs on such undecided relus:

  bound  sum of |term| over the undecided terms that reach the entry   (worst case: all of them taken the other way)
  var    sum of term^2 over the same terms                             (their spread if each falls independently)

An entry of the result under test may then deviate from the float64 gradient by the usual bar plus
min(bound, SIGMAS * sqrt(var)): entries that no undecided relu reaches get no allowance at all, entries of small
sums get at most the terms that can actually flip, and entries of million-term sums (weight gradients), whose
worst-case bound is vacuous, are held to the statistical spread of the flips instead.  A decision counts as
undecided when |pre-activation| <= delta * (sum of the magnitudes of its summands); delta is a few units of the
storage format's rounding (2^-8 for bfloat16 rows, 2e-6 for float32).

Nothing under mpnn_amd/ may import this module.
"""
from __future__ import annotations

import torch

SIGMAS = 8.0
DELTA = {torch.float32: 2e-6, torch.bfloat16: 2.0 ** -8}


def _sum_rows(index, values, n):
    return values.new_zeros((n,) + values.shape[1:]).index_add_(0, index, values)


class LayerTrace:
    """Float64 forward of one layer in the sum form, y = relu([x | aggr] W_out^T) with
    aggr[n] = sum_{e -> n} relu((A[n] + B[src e]) + W_e ea_e), [A | B] = x W_n^T (models/layers.py:104-110,128,140;
    MPNNLayerCat is the same with W_n = [W_i ; W_j] of its `linear`), optionally with the EGNN coordinate update
    pos' = pos + sum_{e -> n} (pos_n - pos_src) (w_pos . m_e).  Keeps what backward() needs."""

    def __init__(self, x, edge_index, edge_attr, w_nodes, w_edge, w_out, delta, pos=None, w_pos=None):
        d = lambda t: t.detach().double().cpu()
        self.x, self.ea = d(x), d(edge_attr)
        self.wn, self.we, self.wo = d(w_nodes), d(w_edge), d(w_out)
        self.src, self.dst = edge_index[0].cpu(), edge_index[1].cpu()
        n, f = self.x.shape
        h = self.wo.size(0)
        self.n, self.f, self.h = n, f, h
        ab = self.x @ self.wn.t()
        ab_mag = self.x.abs() @ self.wn.abs().t()
        es = self.ea @ self.we.t()
        pre = (ab[:, :h][self.dst] + ab[:, h:][self.src]) + es
        mag = ab_mag[:, :h][self.dst] + ab_mag[:, h:][self.src] + self.ea.abs() @ self.we.abs().t()
        self.mask = pre > 0
        self.open_msg = pre.abs() <= delta * mag          # undecided message relus
        self.m = pre * self.mask
        self.aggr = _sum_rows(self.dst, self.m, n)
        cat = torch.cat([self.x, self.aggr], 1)
        z = cat @ self.wo.t()
        zmag = cat.abs() @ self.wo.abs().t()
        self.zmask = z > 0
        self.open_out = z.abs() <= delta * zmag           # undecided update relus
        self.y = z * self.zmask
        self.cat = cat
        self.pos = self.w_pos = self.pos_out = None
        if pos is not None:
            self.pos, self.w_pos = d(pos), d(w_pos).reshape(-1)
            self.rel = self.pos[self.dst] - self.pos[self.src]
            self.sc = self.m @ self.w_pos
            self.pos_out = self.pos + _sum_rows(self.dst, self.rel * self.sc[:, None], n)

    def backward(self, gy, gy_bound=None, gy_var=None, gpos=None, gpos_bound=None, gpos_var=None):
        """Gradients of <y, gy> (+ <pos', gpos>) with their undecided shares.  gy_bound / gy_var: the shares the
        upstream gradient already carries (zeros at the loss).  Returns {name: (gradient, bound, var)} for x,
        w_nodes, w_edge, w_out and, with coordinates, pos and w_pos."""
        z0 = lambda t: torch.zeros_like(t)
        gy = gy.detach().double().cpu()
        ub = z0(gy) if gy_bound is None else gy_bound
        uv = z0(gy) if gy_var is None else gy_var
        f, h, n = self.f, self.h, self.n
        live = (self.zmask | self.open_out).double()
        op = self.open_out.double()
        gz = gy * self.zmask
        gz_b = ub * live + gy.abs() * op
        gz_v = uv * live + gy * gy * op
        wo_abs, wo_sq = self.wo.abs(), self.wo * self.wo
        cat_abs = self.cat.abs()
        out = {"w_out": (gz.t() @ self.cat, gz_b.t() @ cat_abs, gz_v.t() @ (self.cat * self.cat))}
        ga, ga_b, ga_v = gz @ self.wo[:, f:], gz_b @ wo_abs[:, f:], gz_v @ wo_sq[:, f:]
        gx, gx_b, gx_v = gz @ self.wo[:, :f], gz_b @ wo_abs[:, :f], gz_v @ wo_sq[:, :f]
        gm, gm_b, gm_v = ga[self.dst], ga_b[self.dst], ga_v[self.dst]
        if self.pos is not None:
            gpos = gpos.detach().double().cpu()
            pb = z0(gpos) if gpos_bound is None else gpos_bound
            pv = z0(gpos) if gpos_var is None else gpos_var
            gd, gd_b, gd_v = gpos[self.dst], pb[self.dst], pv[self.dst]
            t = (gd * self.rel).sum(1)
            t_b = (gd_b * self.rel.abs()).sum(1)
            t_v = (gd_v * self.rel * self.rel).sum(1)
            wp = self.w_pos
            gm = gm + t[:, None] * wp
            gm_b = gm_b + t_b[:, None] * wp.abs()
            gm_v = gm_v + t_v[:, None] * wp * wp
            out["w_pos"] = ((t[:, None] * self.m).sum(0), (t_b[:, None] * self.m.abs()).sum(0),
                            (t_v[:, None] * self.m * self.m).sum(0))
            s = self.sc[:, None]
            out["pos"] = (gpos + _sum_rows(self.dst, s * gd, n) - _sum_rows(self.src, s * gd, n),
                          pb + _sum_rows(self.dst, s.abs() * gd_b, n) + _sum_rows(self.src, s.abs() * gd_b, n),
                          pv + _sum_rows(self.dst, s * s * gd_v, n) + _sum_rows(self.src, s * s * gd_v, n))
        live = (self.mask | self.open_msg).double()
        op = self.open_msg.double()
        gp = gm * self.mask
        gp_b = gm_b * live + gm.abs() * op
        gp_v = gm_v * live + gm * gm * op
        out["w_edge"] = (gp.t() @ self.ea, gp_b.t() @ self.ea.abs(), gp_v.t() @ (self.ea * self.ea))
        cat2 = lambda v: torch.cat([_sum_rows(self.dst, v, n), _sum_rows(self.src, v, n)], 1)
        gab, gab_b, gab_v = cat2(gp), cat2(gp_b), cat2(gp_v)
        out["w_nodes"] = (gab.t() @ self.x, gab_b.t() @ self.x.abs(), gab_v.t() @ (self.x * self.x))
        out["x"] = (gx + gab @ self.wn, gx_b + gab_b @ self.wn.abs(), gx_v + gab_v @ (self.wn * self.wn))
        return out


def cat_to_sum_weights(w_linear, f):
    """[W_i | W_j | W_e] of MPNNLayerCat.linear (models/layers.py:22) as the sum form's (W_n, W_e)."""
    return torch.cat([w_linear[:, :f], w_linear[:, f:2 * f]], 0), w_linear[:, 2 * f:]


def cat_grad_from_sum(g_nodes, g_edge, h):
    """The gradient of MPNNLayerCat.linear.weight from the sum form's pieces, side by side."""
    return torch.cat([g_nodes[:h], g_nodes[h:], g_edge], 1)


def egnn_model(params: dict, x, pos, edge_index, edge_attr, batch, r, rp, delta):
    """Stack of EGNN layers -> relu -> mean pool -> head (reference_oracle.egnn_model_forward) with the loss
    <logits, r> + <pos_last, rp>.  Returns (logits, pos_last, {parameter name or 'x' / 'pos': (grad, bound, var)})."""
    d = lambda t: t.detach().double().cpu()
    traces, xk, pk, k = [], d(x), d(pos), 0
    while f"layers.{k}.linear_out.weight" in params:
        p = lambda s: params[f"layers.{k}.{s}.weight"]
        t = LayerTrace(xk, edge_index, edge_attr, p("fused_linear"), p("linear_3"), p("linear_out"), delta, pk,
                       p("coord_linear"))
        traces.append(t)
        xk, pk = t.y, t.pos_out            # (the model's second relu is the identity on a relu's output)
        k += 1
    batch = batch.cpu()
    b = int(batch.max()) + 1
    count = torch.bincount(batch, minlength=b).clamp(min=1).double()
    pooled = _sum_rows(batch, xk, b) / count[:, None]
    w, bias = d(params["output_layer.weight"]), d(params["output_layer.bias"])
    logits = pooled @ w.t() + bias
    r, rp = d(r), d(rp)
    grads = {"output_layer.weight": (r.t() @ pooled, None, None), "output_layer.bias": (r.sum(0), None, None)}
    gy = ((r @ w) / count[:, None])[batch]
    gy_b = gy_v = None
    gpos, gp_b, gp_v = rp, None, None
    for k in reversed(range(len(traces))):
        o = traces[k].backward(gy, gy_b, gy_v, gpos, gp_b, gp_v)
        for mine, theirs in (("w_nodes", "fused_linear"), ("w_edge", "linear_3"), ("w_out", "linear_out"), ("w_pos", "coord_linear")):
            g, bd, vr = o[mine]
            shape = params[f"layers.{k}.{theirs}.weight"].shape
            grads[f"layers.{k}.{theirs}.weight"] = (g.reshape(shape), bd.reshape(shape), vr.reshape(shape))
        gy, gy_b, gy_v = o["x"]
        gpos, gp_b, gp_v = o["pos"]
    grads["x"], grads["pos"] = (gy, gy_b, gy_v), (gpos, gp_b, gp_v)
    return logits, pk, grads


def report(entries, rtol: float) -> dict:
    """entries: (name, got, (want, bound, var)[, floor]).  An entry of `got` passes when
    |got - want| <= floor * rtol * max|want| + rtol * |want| + min(bound, SIGMAS * sqrt(var)) (bound / var None: no
    allowance).  Returns {"ok", "worst", "entries", "through_flips" (entries that needed the allowance),
    "max_rel_err" (largest deviation / max|want| among the entries that met the plain bar)}."""
    total = flips = 0
    max_rel, worst = 0.0, ""
    for name, got, (want, bound, var), *rest in entries:
        floor = rest[0] if rest else 1.0
        got, want = got.detach().double().cpu(), want.detach().double().cpu()
        if got.shape != want.shape:
            return {"ok": False, "worst": f"{name}: shape {tuple(got.shape)} vs {tuple(want.shape)}", "entries": total,
                    "through_flips": flips, "max_rel_err": float("inf")}
        if want.numel() == 0:
            continue
        scale = max(float(want.abs().max()), 1e-30)
        err = (got - want).abs()
        bar = rtol * floor * scale + rtol * want.abs()
        plain = err <= bar
        allow = torch.zeros_like(err) if bound is None else torch.minimum(bound, SIGMAS * var.sqrt())
        ok = plain | (err <= bar + allow * (1 + rtol))
        total += want.numel()
        flips += int((ok & ~plain).sum())
        if plain.any():
            rel = float(err[plain].max()) / scale
            if rel > max_rel:
                max_rel, worst = rel, name
        if not bool(ok.all()) or not bool(torch.isfinite(got).all()):
            i = int(torch.argmax((err - bar - allow) * (~ok)))
            return {"ok": False, "entries": total, "through_flips": flips, "max_rel_err": max_rel,
                    "worst": f"{name}: {int((~ok).sum())}/{want.numel()} entries outside rtol {rtol} and outside what "
                             f"undecided relus explain; worst got={got.flatten()[i].item():.8e} "
                             f"want={want.flatten()[i].item():.8e} allowance={allow.flatten()[i].item():.3e}"}
    return {"ok": True, "worst": worst, "entries": total, "through_flips": flips, "max_rel_err": max_rel}
