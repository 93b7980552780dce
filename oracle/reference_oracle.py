"""CPU oracle for the MPNN message-passing hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, with stock ``torch`` CPU ops, the arithmetic that the
reference executes for one MPNN layer and for the model stacks around it.  It
exists so that the HIP path can be checked against something; nothing in the
product package (``mpnn_amd/``) may import it.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` use it.

PARITY UNPINNED.  The reference (KutayTasci/MPNN) ships no tests, golden vectors
or logged outputs, and its layers subclass ``torch_geometric.nn.MessagePassing``
(version not pinned anywhere in the reference; the package is not installed in
this image), so the reference itself cannot be executed here.  The restatement
below therefore follows the reference's call sites line by line and the
published semantics of PyG's ``MessagePassing.propagate`` with ``aggr='add'``,
``flow='source_to_target'``:

  * ``x_j = x.index_select(0, edge_index[0])``  (row 0 = source)
  * ``x_i = x.index_select(0, edge_index[1])``  (row 1 = target = aggregation index)
  * ``aggr = zeros(N, H).scatter_add_(0, edge_index[1][:, None].expand(E, H), msg)``
  * ``update(aggr, x=x)``
  * ``global_mean_pool(x, batch)`` = per-graph sum / clamp(count, min=1)

The EGNN coordinate update has no counterpart in the reference at all; its
oracle is the formula ``pos_i += sum_j (pos_i - pos_j) * phi_x(m_ij)`` with
``phi_x`` a bias-free linear map H -> 1.

Reference lines followed (paths relative to the reference root):
  models/layers.py:20,82      aggr='add'
  models/layers.py:22-23      MPNNLayerCat parameters (bias=False)
  models/layers.py:38,56,69   Cat forward / message / update
  models/layers.py:84-87      MPNNLayerSum parameters (bias=False)
  models/layers.py:104-110    Sum forward (fused_linear, chunk, linear_3, propagate)
  models/layers.py:128,140    Sum message ((xs_i + xt_j) + es, relu) / update
  models/mpnn.py:50-55        MPNN_QM9 / MPNN_ModelNet stack (+ global_mean_pool)
  models/mpnn.py:97-101       MPNN stack
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------
# propagate() pieces
# ----------------------------------------------------------------------------
def gather_src(x: torch.Tensor, edge_index: torch.Tensor) -> torch.Tensor:
    """``x_j``: rows of x at the SOURCE of every edge (edge_index row 0)."""
    return x.index_select(0, edge_index[0])


def gather_dst(x: torch.Tensor, edge_index: torch.Tensor) -> torch.Tensor:
    """``x_i``: rows of x at the TARGET of every edge (edge_index row 1)."""
    return x.index_select(0, edge_index[1])


def scatter_sum(msg: torch.Tensor, edge_index: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """PyG SumAggregation: zeros(N,H).scatter_add_(0, dst broadcast, msg)."""
    idx = edge_index[1].view(-1, 1).expand_as(msg)
    return msg.new_zeros(num_nodes, msg.size(1)).scatter_add_(0, idx, msg)


# ----------------------------------------------------------------------------
# layers  (models/layers.py)
# ----------------------------------------------------------------------------
def cat_message(x, edge_index, edge_attr, w_lin):
    """models/layers.py:56 — relu(linear(cat[x_i, x_j, edge_attr]))."""
    x_i = gather_dst(x, edge_index)
    x_j = gather_src(x, edge_index)
    return F.relu(F.linear(torch.cat([x_i, x_j, edge_attr], dim=1), w_lin))


def sum_message(x, edge_index, edge_attr, w_fused, w_edge):
    """models/layers.py:104-110,128 — relu((xs_i + xt_j) + es)."""
    fused = F.linear(x, w_fused)
    xs, xt = fused.chunk(2, dim=1)
    es = F.linear(edge_attr, w_edge)
    xs_i = gather_dst(xs, edge_index)
    xt_j = gather_src(xt, edge_index)
    return F.relu(torch.add(torch.add(xs_i, xt_j), es))


def update(x, aggr, w_out):
    """models/layers.py:69,140 — relu(linear_out(cat[x, aggr]))."""
    return F.relu(F.linear(torch.cat([x, aggr], dim=1), w_out))


def layer_cat(x, edge_index, edge_attr, w_lin, w_out, return_aggr=False):
    """MPNNLayerCat.forward (models/layers.py:25-38)."""
    aggr = scatter_sum(cat_message(x, edge_index, edge_attr, w_lin), edge_index, x.size(0))
    out = update(x, aggr, w_out)
    return (out, aggr) if return_aggr else out


def layer_sum(x, edge_index, edge_attr, w_fused, w_edge, w_out, return_aggr=False):
    """MPNNLayerSum.forward (models/layers.py:91-110)."""
    aggr = scatter_sum(sum_message(x, edge_index, edge_attr, w_fused, w_edge), edge_index, x.size(0))
    out = update(x, aggr, w_out)
    return (out, aggr) if return_aggr else out


def aggregate_sum_form(a, b, edge_attr, w_edge, edge_index):
    """The engine's internal operator, in oracle form.

    out[n] = sum over edges e with dst(e)=n of relu((a[n] + b[src(e)]) + w_edge @ edge_attr[e])
    This is models/layers.py:128 followed by PyG's add-aggregation, with ``a`` playing
    ``xs`` and ``b`` playing ``xt``.
    """
    es = F.linear(edge_attr, w_edge)
    msg = F.relu(torch.add(torch.add(gather_dst(a, edge_index), gather_src(b, edge_index)), es))
    return scatter_sum(msg, edge_index, a.size(0))


# ----------------------------------------------------------------------------
# pooling + model stacks (models/mpnn.py)
# ----------------------------------------------------------------------------
def global_mean_pool(x: torch.Tensor, batch: torch.Tensor, size: int | None = None) -> torch.Tensor:
    """PyG global_mean_pool: sum by graph id, divided by clamp(count, 1)."""
    if size is None:
        size = int(batch.max().item()) + 1 if batch.numel() > 0 else 0
    idx = batch.view(-1, 1).expand_as(x)
    s = x.new_zeros(size, x.size(1)).scatter_add_(0, idx, x)
    cnt = torch.bincount(batch, minlength=size).clamp(min=1).to(x.dtype).view(-1, 1)
    return s / cnt


def model_forward(params: dict, mode: str, x, edge_index, edge_attr, batch=None):
    """MPNN / MPNN_QM9 / MPNN_ModelNet forward (models/mpnn.py:38-55, 86-101, 134-151).

    ``params`` uses the reference's state_dict keys: ``layers.{k}.linear.weight`` and
    ``layers.{k}.linear_out.weight`` (cat) or ``layers.{k}.fused_linear.weight``,
    ``layers.{k}.linear_3.weight``, ``layers.{k}.linear_out.weight`` (sum), plus
    ``output_layer.weight`` / ``output_layer.bias``.
    """
    k = 0
    while f"layers.{k}.linear_out.weight" in params:
        if mode == "cat":
            x = layer_cat(x, edge_index, edge_attr, params[f"layers.{k}.linear.weight"],
                          params[f"layers.{k}.linear_out.weight"])
        else:
            x = layer_sum(x, edge_index, edge_attr, params[f"layers.{k}.fused_linear.weight"],
                          params[f"layers.{k}.linear_3.weight"], params[f"layers.{k}.linear_out.weight"])
        x = F.relu(x)                      # models/mpnn.py:52,99,148 (second relu)
        k += 1
    if batch is not None:
        x = global_mean_pool(x, batch)     # models/mpnn.py:54,150
    return F.linear(x, params["output_layer.weight"], params["output_layer.bias"])


# ----------------------------------------------------------------------------
# integer structures (bit-exact targets)
# ----------------------------------------------------------------------------
def csr_by_key(key: torch.Tensor, other: torch.Tensor, num_nodes: int):
    """Stable sort of the edge list by ``key``.

    Returns (rowptr int32[N+1], col int32[E] = other in sorted order, eid int32[E] =
    original edge id in sorted order).  Within one row, edges keep their original order.
    """
    key = key.to(torch.int64)
    _, perm = torch.sort(key, stable=True)
    counts = torch.bincount(key, minlength=num_nodes)
    rowptr = torch.zeros(num_nodes + 1, dtype=torch.int64)
    rowptr[1:] = torch.cumsum(counts, 0)
    return rowptr.to(torch.int32), other[perm].to(torch.int32), perm.to(torch.int32)


def csr_build(edge_index: torch.Tensor, num_nodes: int) -> dict:
    """dst-sorted CSR, src-sorted transpose and the two degree vectors."""
    src, dst = edge_index[0], edge_index[1]
    rowptr, col, eid = csr_by_key(dst, src, num_nodes)
    rowptr_t, col_t, eid_t = csr_by_key(src, dst, num_nodes)
    return {
        "rowptr": rowptr, "col": col, "eid": eid,
        "rowptr_t": rowptr_t, "col_t": col_t, "eid_t": eid_t,
        "deg_in": (rowptr[1:] - rowptr[:-1]).to(torch.int32),
        "deg_out": (rowptr_t[1:] - rowptr_t[:-1]).to(torch.int32),
    }


def work_items(rowptr: torch.Tensor, chunk: int, max_parts: int):
    """Restatement of the engine's row-chunking rule (pure integer arithmetic).

    A row with deg <= chunk is one item.  A longer row is cut into pieces of
    ``max(chunk, ceil(deg / max_parts))`` edges.  Items of split rows come first
    (row order, piece order), then the unsplit rows by descending degree (row order
    among equal degrees).  Each item is
    (row, e0, e1, pslot) with pslot = -1 for unsplit rows and the running piece
    number otherwise.  Split rows are listed as (row, first pslot, pieces, first item).
    """
    rp = rowptr.to(torch.int64).tolist()
    n = len(rp) - 1
    split_items, plain_items, split_rows = [], [], []
    pslot = 0
    for r in range(n):
        deg = rp[r + 1] - rp[r]
        if deg <= chunk:
            continue
        size = max(chunk, -(-deg // max_parts))
        pieces = -(-deg // size)
        split_rows.append((r, pslot, pieces, len(split_items)))
        for c in range(pieces):
            e0 = rp[r] + c * size
            split_items.append((r, e0, min(e0 + size, rp[r + 1]), pslot))
            pslot += 1
    plain = [r for r in range(n) if rp[r + 1] - rp[r] <= chunk]
    plain.sort(key=lambda r: -(rp[r + 1] - rp[r]))  # list.sort is stable: ties stay in row order
    for r in plain:
        plain_items.append((r, rp[r], rp[r + 1], -1))
    items = torch.tensor(split_items + plain_items, dtype=torch.int32).reshape(-1, 4)
    split = torch.tensor(split_rows, dtype=torch.int32).reshape(-1, 4)
    return items, split, pslot


def rows_in_item_order(rowptr: torch.Tensor, col: torch.Tensor, eid: torch.Tensor, chunk: int):
    """A sorted view with its ROWS stored in the order work_items lists them for ``chunk`` (rows longer than chunk
    first in row order, then descending length, row order among equals); edges keep their order inside a row.
    Returns (order [N] = row stored at each position, rowptr indexed by position, col, eid), int32.  Plain loops."""
    rp = rowptr.to(torch.int64).tolist()
    n = len(rp) - 1
    order = sorted(range(n), key=lambda r: 0 if rp[r + 1] - rp[r] > chunk else 1 + chunk - (rp[r + 1] - rp[r]))
    newptr, pos = [0], []
    for r in order:
        pos.extend(range(rp[r], rp[r + 1]))
        newptr.append(newptr[-1] + rp[r + 1] - rp[r])
    pos = torch.tensor(pos, dtype=torch.int64)
    return (torch.tensor(order, dtype=torch.int32), torch.tensor(newptr, dtype=torch.int32),
            col[pos].to(torch.int32), eid[pos].to(torch.int32))


def dst_range_partition(rowptr: torch.Tensor, world: int):
    """Contiguous destination ranges balanced by in-edge count.

    bounds[r] = smallest node id n such that rowptr[n] >= r * E / world (integer
    arithmetic, ceil division), bounds[0] = 0, bounds[world] = N.
    """
    rp = rowptr.to(torch.int64)
    n = rp.numel() - 1
    e = int(rp[-1])
    bounds = [0]
    for r in range(1, world):
        target = -(-(r * e) // world)
        b = int(torch.searchsorted(rp, torch.tensor(target), right=False))
        bounds.append(min(max(b, bounds[-1]), n))
    bounds.append(n)
    return bounds


def halo_row_lists(edge_index: torch.Tensor, bounds: list, rank: int):
    """Rows of every peer that rank ``rank`` of a destination-range partition must receive: for the edges whose
    target lies in [bounds[rank], bounds[rank+1]), the sorted distinct source ids, grouped by the range that owns
    them and numbered inside that range.  Returns a list (one int64 tensor per peer).  Plain Python sets."""
    lo, hi = bounds[rank], bounds[rank + 1]
    src, dst = edge_index[0].tolist(), edge_index[1].tolist()
    need = sorted({s for s, d in zip(src, dst) if lo <= d < hi})
    out = []
    for q in range(len(bounds) - 1):
        out.append(torch.tensor([s - bounds[q] for s in need if bounds[q] <= s < bounds[q + 1]], dtype=torch.int64))
    return out


def islands(edge_index: torch.Tensor, num_nodes: int) -> list:
    """Finest split of the node range into runs of consecutive ids that no edge leaves: the cut after node r stands
    iff no edge has one end <= r and the other > r.  Returns the run boundaries [0, ..., num_nodes].  (The graphs
    of a PyG mini-batch, data/dataset.py:94-96, are unions of such runs; the engine's whole-layer kernels for
    small graphs work run by run.)"""
    lo = torch.minimum(edge_index[0], edge_index[1]).tolist()
    hi = torch.maximum(edge_index[0], edge_index[1]).tolist()
    spans = [0] * (num_nodes + 1)          # spans[r] = edges over the cut after r, as a difference array
    for a, b in zip(lo, hi):
        spans[a] += 1
        spans[b] -= 1
    out, open_edges = [0], 0
    for r in range(num_nodes):
        open_edges += spans[r]
        if open_edges == 0:
            out.append(r + 1)
    return out


# ----------------------------------------------------------------------------
# EGNN coordinate update (no reference counterpart)
# ----------------------------------------------------------------------------
def egnn_layer(a, b, edge_attr, w_edge, pos, w_pos, edge_index):
    """Message + coordinate update, oracle form.

    m_e   = relu((a[dst] + b[src]) + w_edge @ edge_attr[e])
    aggr  = sum_e->dst m_e
    pos'  = pos + sum_e->dst (pos[dst] - pos[src]) * (w_pos . m_e)
    """
    es = F.linear(edge_attr, w_edge)
    m = F.relu(torch.add(torch.add(gather_dst(a, edge_index), gather_src(b, edge_index)), es))
    aggr = scatter_sum(m, edge_index, a.size(0))
    s = (m * w_pos.view(1, -1)).sum(dim=1, keepdim=True)
    rel = gather_dst(pos, edge_index) - gather_src(pos, edge_index)
    return aggr, pos + scatter_sum(rel * s, edge_index, pos.size(0))


def egnn_full_layer(x, pos, edge_index, edge_attr, w_fused, w_edge, w_out, w_pos):
    """Whole EGNN layer: MPNNLayerSum's projection and update (models/layers.py:104-110,140) around egnn_layer.
    Returns (relu(linear_out([x | aggr])), pos')."""
    fused = F.linear(x, w_fused)
    xs, xt = fused.chunk(2, dim=1)
    aggr, pos_out = egnn_layer(xs, xt, edge_attr, w_edge, pos, w_pos.view(-1), edge_index)
    return update(x, aggr, w_out), pos_out


def egnn_model_forward(params: dict, x, pos, edge_index, edge_attr, batch):
    """Stack of EGNN layers -> relu -> global_mean_pool -> Linear head, the shape of models/mpnn.py:38-55 with the
    coordinates threaded through.  ``params``: ``layers.{k}.{fused_linear,linear_3,linear_out,coord_linear}.weight``
    and ``output_layer.{weight,bias}``.  Returns (logits, pos after the last layer)."""
    k = 0
    while f"layers.{k}.linear_out.weight" in params:
        x, pos = egnn_full_layer(x, pos, edge_index, edge_attr, params[f"layers.{k}.fused_linear.weight"],
                                 params[f"layers.{k}.linear_3.weight"], params[f"layers.{k}.linear_out.weight"],
                                 params[f"layers.{k}.coord_linear.weight"])
        x = F.relu(x)
        k += 1
    x = global_mean_pool(x, batch)
    return F.linear(x, params["output_layer.weight"], params["output_layer.bias"]), pos


# ----------------------------------------------------------------------------
# mini-batch collation (torch_geometric DataLoader semantics)
# ----------------------------------------------------------------------------
def collate(x, edge_index, edge_attr, node_ptr, edge_ptr, ids, y=None, pos=None) -> dict:
    """Batch of the graphs ``ids`` of a concatenated dataset, as torch_geometric's ``Batch.from_data_list`` builds
    it for the reference's loaders (data/dataset.py:94-96, 215-216): node rows, targets and edges of the listed
    graphs concatenated in list order, every graph's LOCAL edge_index shifted by the number of nodes before it,
    and ``batch[p]`` = position in the list of the graph node p came from.  ``edge_index`` here holds
    dataset-global node ids, so a graph's local ids are its ids minus node_ptr[g].  Plain Python loop."""
    xs, eis, eas, bs, ys, ps = [], [], [], [], [], []
    n_before = 0
    for slot, g in enumerate(int(i) for i in ids):
        n0, n1 = int(node_ptr[g]), int(node_ptr[g + 1])
        e0, e1 = int(edge_ptr[g]), int(edge_ptr[g + 1])
        xs.append(x[n0:n1])
        if pos is not None:
            ps.append(pos[n0:n1])
        if y is not None:
            ys.append(y[g:g + 1])
        eis.append(edge_index[:, e0:e1] - n0 + n_before)
        eas.append(edge_attr[e0:e1])
        bs.append(torch.full((n1 - n0,), slot, dtype=torch.int64))
        n_before += n1 - n0
    return {
        "x": torch.cat(xs), "edge_index": torch.cat(eis, dim=1), "edge_attr": torch.cat(eas), "batch": torch.cat(bs),
        "y": torch.cat(ys) if y is not None else None, "pos": torch.cat(ps) if pos is not None else None,
        "num_nodes": n_before,
    }


# ----------------------------------------------------------------------------
# loss and optimizer step of the training loop
# ----------------------------------------------------------------------------
def masked_cross_entropy(logits, labels, mask=None):
    """F.cross_entropy(out[mask], y[mask]) (training/train.py:47) written as a mask-weighted mean, float64."""
    x = logits.double()
    lse = torch.logsumexp(x, dim=1)
    picked = x.gather(1, labels.view(-1, 1)).squeeze(1)
    w = torch.ones_like(lse) if mask is None else mask.double()
    return ((lse - picked) * w).sum() / w.sum()


def adam_step(p, g, m, v, t, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8):
    """One torch.optim.Adam step (no weight decay, no amsgrad) on plain tensors, in the dtype of ``p``;
    ``t`` is the 1-based step number.  Returns (p, m, v)."""
    m = m + (g - m) * (1 - beta1)
    v = v * beta2 + (1 - beta2) * g * g
    step_size = lr / (1 - beta1 ** t)
    denom = v.sqrt() / (1 - beta2 ** t) ** 0.5 + eps
    return p - step_size * (m / denom), m, v


# ----------------------------------------------------------------------------
# k-NN graph of point clouds with the normalised distance feature
# ----------------------------------------------------------------------------
def knn_graph(pos: torch.Tensor, num_clouds: int, k: int):
    """Edges j -> i for the k nearest j of each point i of the same cloud (grouped by i, ascending distance, ties
    by lower j) and T.Distance(norm=True)'s feature (experiments/modelnet_exp.py:58): distance divided by the
    largest edge distance of the cloud.  float32, squared distance as ((dx*dx + dy*dy) + dz*dz)."""
    pts = pos.view(num_clouds, -1, 3).float()
    p = pts.size(1)
    src, dst, attr = [], [], []
    for c in range(num_clouds):
        d = pts[c][:, None, :] - pts[c][None, :, :]          # [i, j, 3] = p_i - p_j
        d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
        d2.fill_diagonal_(float("inf"))
        val, idx = torch.sort(d2, dim=1, stable=True)
        # correctly rounded float32 square root and quotient, obtained by rounding the float64 results (torch's
        # float32 CPU sqrt is a few-ulp approximation, and division by a 0-dim tensor multiplies by a reciprocal)
        dist = torch.sqrt(val[:, :k].double()).float()
        src.append(idx[:, :k].reshape(-1) + c * p)
        dst.append(torch.arange(p).repeat_interleave(k) + c * p)
        attr.append((dist.double() / dist.max().double()).float().reshape(-1))
    return torch.stack([torch.cat(src), torch.cat(dst)]), torch.cat(attr).view(-1, 1)
