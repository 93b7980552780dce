"""Device-resident datasets of small graphs and their mini-batches.

The reference batches molecules / point clouds with torch_geometric's DataLoader in 8 CPU worker
processes (data/dataset.py:94-96, 215-216) and iterates ``for data in train_loader`` (training/train.py:96-98);
every batch then pays for the index handling inside propagate() again.  Here the whole dataset lives on the
GPU, its (block-diagonal) edge list is sorted once, and a mini-batch - including the sorted index structure
the layers need - is written by one library call (mpnn_collate) with no sort and no device->host read.

``Batch`` carries the attributes the reference's trainers read from a PyG batch: x, edge_index, batch,
edge_attr, y, pos (training/train.py:21,73), and ``.to(device)``.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib
from .graph import CHUNK, CSR, Islands, MPGraph, forget_graph, register_graph
from .ops import register_pool


class Batch:
    """The concatenation of a list of graphs (PyG ``Batch.from_data_list`` semantics)."""

    def __init__(self, x, edge_index, batch, edge_attr, y, pos, num_graphs):
        self.x, self.edge_index, self.batch, self.edge_attr, self.y, self.pos = x, edge_index, batch, edge_attr, y, pos
        self.num_graphs = num_graphs
        self.num_nodes = x.size(0)
        self._parts = None
        self._graph = None     # MPGraph over the structures mpnn_collate wrote (stores whose rows are never split)
        self.capacity = None   # (nodes, edges) of a padded batch, see GraphStore.collate

    def to(self, device):
        if torch.device(device).type != self.x.device.type:
            raise _lib.MpnnError("a device-assembled batch cannot be moved off the GPU")
        return self


def slot_table(node_ptr: np.ndarray, edge_ptr: np.ndarray, ids) -> np.ndarray:
    """Host-side offsets of a batch: int32 [5, B + 1], the `slots` argument of mpnn_collate.

    Row 0 / 1: first batch node / edge of every slot (entry B = totals); row 2 / 3: those minus the first dataset
    node / edge of the slot's graph; row 4: the graph's dataset index."""
    ids = np.asarray(ids, dtype=np.int64).reshape(-1)
    if ids.size == 0:
        raise ValueError("empty batch")
    if ids.min() < 0 or ids.max() >= len(node_ptr) - 1:
        raise IndexError("graph index out of range")
    b = ids.size
    nn = node_ptr[ids + 1] - node_ptr[ids]
    ne = edge_ptr[ids + 1] - edge_ptr[ids]
    if nn.sum() >= 2 ** 31 or ne.sum() >= 2 ** 31:
        raise ValueError("batch exceeds 2^31 nodes or edges")
    t = np.zeros((5, b + 1), dtype=np.int64)
    t[0, 1:] = np.cumsum(nn)
    t[1, 1:] = np.cumsum(ne)
    t[2, :b] = t[0, :b] - node_ptr[ids]
    t[3, :b] = t[1, :b] - edge_ptr[ids]
    t[4, :b] = ids
    return t.astype(np.int32)


class GraphStore:
    """A dataset of G small graphs, concatenated on the device.

    x [N_all, F], edge_index [2, E_all] with dataset-global node ids, edge_attr [E_all, Fe], node_ptr / edge_ptr
    [G + 1] (host or device, int64): graph g owns nodes node_ptr[g]:node_ptr[g+1] and edges
    edge_ptr[g]:edge_ptr[g+1], and its edges only join its own nodes.  y has one row per graph.
    """

    def __init__(self, x, edge_index, edge_attr, node_ptr, edge_ptr, y=None, pos=None):
        _lib.require_device(x, "x")
        dev = x.device
        self.device = dev
        self.node_ptr = np.asarray(torch.as_tensor(node_ptr).cpu().numpy(), dtype=np.int64)
        self.edge_ptr = np.asarray(torch.as_tensor(edge_ptr).cpu().numpy(), dtype=np.int64)
        self.num_graphs = len(self.node_ptr) - 1
        n_all, e_all = int(self.node_ptr[-1]), int(self.edge_ptr[-1])
        if x.size(0) != n_all or edge_index.size(1) != e_all or edge_attr.size(0) != e_all:
            raise ValueError("node_ptr / edge_ptr totals do not match x / edge_index / edge_attr")
        if y is not None and y.size(0) != self.num_graphs:
            raise ValueError("y needs one row per graph")
        self.x = x.contiguous()
        self.pos = pos.contiguous() if pos is not None else None
        self.y = y.contiguous() if y is not None else None
        self.edge_attr = edge_attr.contiguous()
        self.src = edge_index[0].contiguous()
        self.dst = edge_index[1].contiguous()
        # every edge must stay inside its graph's node range (block-diagonal dataset)
        if e_all:
            ep = torch.as_tensor(self.edge_ptr, device=dev)
            npt = torch.as_tensor(self.node_ptr, device=dev)
            owner = torch.searchsorted(ep, torch.arange(e_all, device=dev), right=True) - 1
            lo, hi = npt[owner], npt[owner + 1]
            ok = (self.src >= lo) & (self.src < hi) & (self.dst >= lo) & (self.dst < hi)
            if not bool(ok.all()):
                raise ValueError("an edge leaves the node range of its graph")
        # one sort for the whole dataset
        self.graph_all = MPGraph(torch.stack([self.src, self.dst]), n_all)
        self.tpos_all = self.graph_all.tpos
        self.ea_dst_all, self.ea_src_all = self.graph_all.sorted_edge_attr(self.edge_attr, self.edge_attr.dtype)
        g = self.graph_all
        self.max_degree = int(max(g.by_dst.degree().max().item(), g.by_src.degree().max().item())) if e_all else 0
        # the fused path writes one work item per row, which is the general builder's result only while no row
        # is long enough to be split
        self.fused = self.max_degree <= CHUNK
        self.max_graph_nodes = int((self.node_ptr[1:] - self.node_ptr[:-1]).max()) if self.num_graphs else 0
        self._iota = torch.arange(0, dtype=torch.int32, device=dev)
        self._staging = {}     # page-locked staging buffers of the slot tables, by shape (see _upload)

    @classmethod
    def from_batched(cls, d: dict) -> "GraphStore":
        """Builds a store from one big batched sample (dict with x, edge_index, edge_attr, batch, y, pos):
        edges are regrouped by the graph of their target, keeping their relative order."""
        batch = d["batch"]
        num_graphs = int(batch.max().item()) + 1
        sizes = torch.bincount(batch, minlength=num_graphs)
        node_ptr = torch.zeros(num_graphs + 1, dtype=torch.int64, device=batch.device)
        node_ptr[1:] = torch.cumsum(sizes, 0)
        ei = d["edge_index"]
        eg = batch[ei[1]]
        order = torch.sort(eg, stable=True).indices
        esz = torch.bincount(eg, minlength=num_graphs)
        edge_ptr = torch.zeros(num_graphs + 1, dtype=torch.int64, device=batch.device)
        edge_ptr[1:] = torch.cumsum(esz, 0)
        return cls(d["x"], ei[:, order], d["edge_attr"][order], node_ptr, edge_ptr, d.get("y"), d.get("pos"))

    # ------------------------------------------------------------------------------------------
    def slot_table(self, ids) -> np.ndarray:
        return slot_table(self.node_ptr, self.edge_ptr, ids)

    def capacity_for(self, batch_size: int) -> tuple:
        """(nodes, edges) that hold any batch of ``batch_size`` DISTINCT graphs of this store (what a loader draws),
        with the one spare node a padded batch needs: the sums over the ``batch_size`` largest graphs.  A batch
        that repeats graphs may need more; collate() refuses a batch that does not fit."""
        k = min(int(batch_size), self.num_graphs)
        nn = np.sort(self.node_ptr[1:] - self.node_ptr[:-1])[::-1][:k].sum()
        ne = np.sort(self.edge_ptr[1:] - self.edge_ptr[:-1])[::-1][:k].sum()
        return int(nn) + 1, int(ne)

    def collate(self, ids, out: Batch | None = None, capacity: tuple | None = None) -> Batch:
        """Mini-batch of the graphs ``ids`` (host sequence), in that order.

        ``out``: a batch returned earlier with the same number of graphs, nodes and edges; its tensors are
        overwritten in place (fixed-size point-cloud batches), so code captured into a HIP graph that reads them
        sees the new batch on its next replay.

        ``capacity`` = (nodes, edges), e.g. ``capacity_for(batch_size)``: a PADDED batch whose tensors have that
        size whatever the graphs hold (molecules differ in size).  Rows past the real nodes are zero features that
        belong to one extra graph (``batch`` value = number of graphs) and have no edges; entries past the real
        edges are outside the sorted structure the layers use.  A model therefore returns one extra pooled row
        (the last; global_mean_pool sets it to zero and passes no gradient to the padding rows), and every real
        row equals the unpadded result.  A padded batch passed as ``out`` is refilled
        by any ``ids`` of the same count that fit - the real sizes travel in the slot table on the device, so the
        refill and the training step can be replayed from one HIP graph."""
        table = self.slot_table(ids)
        b = table.shape[1] - 1
        n, e = int(table[0, b]), int(table[1, b])
        dev = self.device
        if out is not None:
            if out._parts is None or out.num_graphs != b:
                raise ValueError("collate(out=...) needs a batch with the same number of graphs produced by this store")
            if out.capacity is not None:
                if n >= out.capacity[0] or e > out.capacity[1]:
                    raise ValueError(f"batch of {n} nodes / {e} edges exceeds the capacity {out.capacity}")
            elif (out.num_nodes, out.edge_index.size(1)) != (n, e):
                raise ValueError("collate(out=...) needs a batch of identical size (or a padded batch)")
            self._upload(table, out._parts["slots"])
            self._launch(out._parts, b, n, e, out.capacity)
            # The launch rewrote edge_index, edge_attr and batch through raw pointers, which leaves their version
            # counters (the caches' staleness test) unchanged: put the structures that were rewritten with them
            # back under their keys, or drop what was derived from the old contents.
            self._register(out, n)
            return out
        if capacity is not None:
            n_out, e_out = int(capacity[0]), int(capacity[1])
            if n >= n_out or e > e_out:
                raise ValueError(f"batch of {n} nodes / {e} edges exceeds the capacity {(n_out, e_out)}")
            if not self.fused:
                raise ValueError("padded batches need a store whose rows are never split (max degree <= 64)")
        else:
            n_out, e_out = n, e
        i32 = dict(dtype=torch.int32, device=dev)
        ints = torch.empty(2 * (n_out + 1) + 5 * e_out + (b + 2), **i32)   # rowptr x2, col/eid x2, tpos, pool rowptr
        items = torch.empty((2, max(n_out, 1), 4), **i32)
        o = 2 * (n_out + 1)
        ea3 = torch.empty((3, e_out) + tuple(self.edge_attr.shape[1:]), dtype=self.edge_attr.dtype, device=dev)
        parts = dict(
            slots=self._upload(table, torch.empty(table.shape, dtype=torch.int32, device=dev)),
            x=torch.empty((n_out,) + tuple(self.x.shape[1:]), dtype=self.x.dtype, device=dev),
            pos=torch.empty((n_out,) + tuple(self.pos.shape[1:]), dtype=self.pos.dtype, device=dev) if self.pos is not None else None,
            y=torch.empty((b,) + tuple(self.y.shape[1:]), dtype=self.y.dtype, device=dev) if self.y is not None else None,
            batch=torch.empty(n_out, dtype=torch.int64, device=dev),
            edge_index=torch.empty((2, e_out), dtype=torch.int64, device=dev),
            ea=ea3[0], ea_dst=ea3[1], ea_src=ea3[2],
            rowptr_dst=ints[:n_out + 1], rowptr_src=ints[n_out + 1:o], items_dst=items[0], items_src=items[1],
            pool_rowptr=ints[o + 5 * e_out:])
        for k, name in enumerate(("col_dst", "eid_dst", "col_src", "eid_src", "tpos")):
            parts[name] = ints[o + k * e_out:o + (k + 1) * e_out]
        self._launch(parts, b, n, e, capacity)
        res = Batch(parts["x"], parts["edge_index"], parts["batch"], parts["ea"], parts["y"], parts["pos"], b)
        res._parts = parts
        res.capacity = (n_out, e_out) if capacity is not None else None
        if self.fused:
            views = []
            for sfx in ("dst", "src"):
                c = CSR()
                c.num_rows, c.num_cols, c.num_edges = n_out, n_out, e_out
                c.rowptr, c.col, c.eid = parts["rowptr_" + sfx], parts["col_" + sfx], parts["eid_" + sfx]
                c.items, c.split = parts["items_" + sfx], None
                c.n_items, c.n_parts, c.n_split = n_out, 0, 0
                views.append(c)
            # the graphs of the batch are its islands (row 0 of the slot table = their first nodes); the rows of a
            # padded batch past the real nodes have no edges
            isl = Islands(parts["slots"][0], b, self.max_graph_nodes, tail_rows=n_out if capacity is not None else 0)
            res._graph = MPGraph.from_parts(views[0], views[1], parts["tpos"], isl)
        if self._iota.numel() < n_out:
            self._iota = torch.arange(max(n_out, 2 * self._iota.numel()), **i32)
        self._register(res, n)
        return res

    def _upload(self, table: np.ndarray, dst: torch.Tensor) -> torch.Tensor:
        """The slot table to the device without stopping the host.  A copy from ordinary host memory waits for the
        stream to reach it, which would serialise the host's work on the next batch with the GPU's work on this one;
        from page-locked memory it is queued like a kernel.  Four staging buffers per table shape take turns, each
        guarded by an event recorded after its copy."""
        if dst.device.type != "cuda":
            dst.copy_(torch.from_numpy(table))
            return dst
        ring = self._staging.get(table.shape)
        if ring is None:
            ring = self._staging[table.shape] = {"bufs": [], "events": [], "next": 0}
        if len(ring["bufs"]) < 4:
            ring["bufs"].append(torch.empty(table.shape, dtype=torch.int32, pin_memory=True))
            ring["events"].append(torch.cuda.Event())
            i = len(ring["bufs"]) - 1
        else:
            i = ring["next"]
            ring["events"][i].synchronize()      # its copy of four uploads ago: long done
        ring["next"] = (i + 1) % 4
        ring["bufs"][i].numpy()[...] = table
        dst.copy_(ring["bufs"][i], non_blocking=True)
        ring["events"][i].record(torch.cuda.current_stream(dst.device))
        return dst

    def _register(self, batch: Batch, n: int) -> None:
        """Makes the layers and the pooling find the structures mpnn_collate just wrote for ``batch`` (after a
        first assembly and after every refill through ``out=``)."""
        parts, b = batch._parts, batch.num_graphs
        n_out = parts["x"].size(0)
        if batch._graph is not None:
            batch._graph.seed_edge_attr(parts["ea"], parts["ea_dst"], parts["ea_src"])
            register_graph(parts["edge_index"], n_out, batch._graph)
        else:
            # rows too long for one work item each: the layers build the general structure from edge_index on
            # first use - never reuse one built from the buffer's earlier contents
            forget_graph(parts["edge_index"])
        if batch.capacity is not None:   # graphs 0 .. b-1 and the padding graph b
            register_pool(parts["batch"], b + 1, parts["pool_rowptr"], self._iota[:n_out], averaged=b)
        else:
            register_pool(parts["batch"], b, parts["slots"][0], self._iota[:n])

    def _launch(self, t: dict, b: int, n: int, e: int, capacity=None) -> None:
        ga = self.graph_all
        row_bytes = lambda a: int(np.prod(a.shape[1:], dtype=np.int64)) * a.element_size()
        P = _lib.ptr
        d = _lib.CollateDesc(
            slots=P(t["slots"]), num_slots=b, num_nodes=n, num_edges=e,
            x_all=P(self.x), x_row_bytes=row_bytes(self.x),
            pos_all=P(self.pos), pos_row_bytes=row_bytes(self.pos) if self.pos is not None else 0,
            y_all=P(self.y), y_row_bytes=row_bytes(self.y) if self.y is not None else 0,
            src_all=P(self.src), dst_all=P(self.dst),
            ea_all=P(self.edge_attr), ea_dst_all=P(self.ea_dst_all), ea_src_all=P(self.ea_src_all),
            ea_row_bytes=row_bytes(self.edge_attr),
            rowptr_dst_all=P(ga.by_dst.rowptr), col_dst_all=P(ga.by_dst.col), eid_dst_all=P(ga.by_dst.eid),
            rowptr_src_all=P(ga.by_src.rowptr), col_src_all=P(ga.by_src.col), eid_src_all=P(ga.by_src.eid),
            tpos_all=P(self.tpos_all),
            x=P(t["x"]), pos=P(t["pos"]), y=P(t["y"]), batch=P(t["batch"]), edge_index=P(t["edge_index"]),
            ea=P(t["ea"]), ea_dst=P(t["ea_dst"]), ea_src=P(t["ea_src"]),
            rowptr_dst=P(t["rowptr_dst"]), col_dst=P(t["col_dst"]), eid_dst=P(t["eid_dst"]), items_dst=P(t["items_dst"]),
            rowptr_src=P(t["rowptr_src"]), col_src=P(t["col_src"]), eid_src=P(t["eid_src"]), items_src=P(t["items_src"]),
            tpos=P(t["tpos"]),
            node_capacity=capacity[0] if capacity is not None else 0,
            edge_capacity=capacity[1] if capacity is not None else 0,
            pool_rowptr=P(t["pool_rowptr"]) if capacity is not None else None)
        _lib.check(_lib.load().mpnn_collate(ctypes.byref(d), _lib.stream_handle(self.device)), "mpnn_collate")


class GraphLoader:
    """``for data in loader`` over mini-batches of a GraphStore (DataLoader(batch_size, shuffle) of the reference,
    data/dataset.py:94-96): the last batch may be short; a shuffling loader draws a new order every epoch."""

    def __init__(self, store: GraphStore, batch_size: int = 32, shuffle: bool = False, seed: int = 0, indices=None):
        self.store, self.batch_size, self.shuffle = store, int(batch_size), shuffle
        self.indices = np.arange(store.num_graphs) if indices is None else np.asarray(indices, dtype=np.int64)
        self._rng = np.random.default_rng(seed)

    def __len__(self):
        return (len(self.indices) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        order = self._rng.permutation(self.indices) if self.shuffle else self.indices
        for i in range(0, len(order), self.batch_size):
            yield self.store.collate(order[i:i + self.batch_size])
