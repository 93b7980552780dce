"""Graph structure on the device: destination-sorted CSR, source-sorted transpose, work items.

Built once per ``edge_index`` tensor by the HIP library and cached on that tensor, so the
unchanged training loops (which pass the same ``data.edge_index`` every step, reference
training/train.py:46) pay for the sort once; mini-batch loops (training/train.py:96-98)
pay per batch.
"""
from __future__ import annotations

import collections
import os

import torch

from . import _lib

# Row-splitting rule of mpnn_items_build: rows longer than `chunk` edges are cut into at most
# MAX_PARTS pieces of at least `chunk` edges.  A piece costs a partial row written and summed again, so pieces
# should be as long as the parallelism of the edge list allows: `chunk` grows with the number of edges
# (the largest power of two <= E / 65536) between CHUNK and CHUNK_MAX.
CHUNK = 64
CHUNK_MAX = 1024
MAX_PARTS = 256
BYTE_COUNT_MAX = 255    # a target row of at most this many edges has exact one-byte counts of positive messages
# Edge lists of at least this many edges get their sorted views laid out in work-item order by graph_for()
# (see MPGraph, ``row_order``).
ITEM_ORDER_MIN_EDGES = 1 << 22


# Graphs of at most this many nodes are searched for islands when their structure is built (one more launch of
# the build, its result read back with the item counts); see Islands.  MPNN_AMD_ISLANDS=0 switches the search and
# with it the one-launch layers off.
ISLAND_SEARCH_MAX_NODES = 1 << 18
ISLAND_GIVE_UP_NODES = 33        # one more than the largest island the one-wavefront layers take
ISLAND_SEARCH_GIVE_UP = 2049     # one more than the largest island any one-launch layer takes (islands.REGION_MAX_NODES)
# An edge list with at most this many edges per node on average is taken for a batch of small graphs: its structure
# is built in two steps (destination view + island search first, the rest only if a general kernel asks for it).
SMALL_GRAPHS_MAX_DEGREE = 8
ISLANDS_ENABLED = os.environ.get("MPNN_AMD_ISLANDS", "1") != "0"


class Islands:
    """Runs of consecutive nodes that no edge leaves (the graphs of a collated batch): island i = nodes
    ptr[i] .. ptr[i+1]-1.  ``max_nodes`` bounds every island; rows from ptr[count] up to the graph's node count
    (padding of a fixed-capacity batch) have no edges and are processed ``tail_units`` tiles at most."""

    __slots__ = ("ptr", "count", "max_nodes", "max_edges", "tail_rows")

    def __init__(self, ptr, count, max_nodes, max_edges=0, tail_rows=0):
        self.ptr, self.count, self.max_nodes, self.max_edges, self.tail_rows = ptr, count, max_nodes, max_edges, tail_rows


def default_chunk(num_edges: int) -> int:
    """Shortest split piece for an edge list of this size (pure integer rule, mirrored by the tests)."""
    c = CHUNK
    while 2 * c <= CHUNK_MAX and 2 * c * 65536 <= num_edges:
        c *= 2
    return c


class CSR:
    """One sorted view of the edge list (see mpnn_csr_build / mpnn_items_build)."""

    __slots__ = ("num_rows", "num_cols", "num_edges", "rowptr", "col", "eid", "items", "split",
                 "n_items", "n_parts", "n_split", "order", "n_big", "averaged")

    def __init__(self):
        self.order = None
        self.n_big = 0      # unsplit rows with more than 255 edges: the first n_big unsplit work items
        self.averaged = None   # pooling segments: rows that are real graphs (None = all), see ops.register_pool

    def degree(self) -> torch.Tensor:
        """Edges per row, indexed by row id."""
        deg = self.rowptr[1:] - self.rowptr[:-1]
        order = getattr(self, "order", None)
        if order is None:
            return deg
        out = torch.empty_like(deg)
        out[order.long()] = deg          # rowptr is indexed by storage position; order[j] = row stored at position j
        return out


def _build_csr(key: torch.Tensor, other: torch.Tensor, num_rows: int, num_cols: int,
               chunk: int, max_parts: int, item_order: bool = False, with_items: bool = True) -> CSR:
    """Sorted view + (with_items) its work items.  Returns (csr, pending record for _finish), or - without items -
    (csr, (status, workspace)) for a later _build_items."""
    lib = _lib.load()
    dev = key.device
    E = key.numel()
    stream = _lib.stream_handle(dev)
    csr = CSR()
    csr.order = None
    csr.num_rows, csr.num_cols, csr.num_edges = num_rows, num_cols, E
    csr.rowptr = torch.empty(num_rows + 1, dtype=torch.int32, device=dev)
    csr.col = torch.empty(E, dtype=torch.int32, device=dev)
    csr.eid = torch.empty(E, dtype=torch.int32, device=dev)
    status = torch.zeros(4, dtype=torch.int32, device=dev)
    ws_bytes = lib.mpnn_csr_workspace_bytes(num_rows, E)
    if ws_bytes < 0:
        raise _lib.MpnnError(f"mpnn_csr_workspace_bytes({num_rows}, {E}) failed")
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
    _lib.check(lib.mpnn_csr_build(_lib.ptr(key), _lib.ptr(other), E, num_rows, num_cols, _lib.ptr(csr.rowptr),
                                  _lib.ptr(csr.col), _lib.ptr(csr.eid), _lib.ptr(status), _lib.ptr(ws), ws.numel(),
                                  stream), "mpnn_csr_build")
    if item_order:
        _rows_into_item_order(csr, status, chunk)
    if not with_items:
        return csr, (status, ws)
    return csr, _build_items(csr, chunk, max_parts, status, ws)


def _build_items(csr: CSR, chunk: int, max_parts: int, status: torch.Tensor, ws: torch.Tensor):
    lib = _lib.load()
    dev = csr.rowptr.device
    num_rows, E = csr.num_rows, csr.num_edges
    stream = _lib.stream_handle(dev)
    items_cap = num_rows + E // chunk + 1
    split_cap = E // (chunk + 1) + 1
    csr.items = torch.empty((items_cap, 4), dtype=torch.int32, device=dev)
    csr.split = torch.empty((split_cap, 4), dtype=torch.int32, device=dev)
    meta = torch.zeros(8, dtype=torch.int32, device=dev)
    ws2_bytes = lib.mpnn_items_workspace_bytes(num_rows)
    if ws2_bytes < 0:
        raise _lib.MpnnError(f"mpnn_items_workspace_bytes({num_rows}) failed")
    if ws2_bytes > ws.numel():
        ws = torch.empty(ws2_bytes, dtype=torch.uint8, device=dev)
    _lib.check(lib.mpnn_items_build(_lib.ptr(csr.rowptr), num_rows, chunk, max_parts, _lib.ptr(csr.items),
                                    items_cap, _lib.ptr(csr.split), split_cap, _lib.ptr(meta), _lib.ptr(ws),
                                    ws.numel(), stream), "mpnn_items_build")
    # unsplit rows whose per-column counts can exceed a byte (mpnn_aggregate_fwd keeps exact counts for them)
    if chunk > BYTE_COUNT_MAX:
        deg = csr.rowptr[1:] - csr.rowptr[:-1]
        n_big = ((deg > BYTE_COUNT_MAX) & (deg <= chunk)).sum().to(torch.int32).view(1)
    else:
        n_big = torch.zeros(1, dtype=torch.int32, device=dev)
    return torch.cat([meta[:4], status[:1], n_big])


def _rows_into_item_order(csr: CSR, status: torch.Tensor, chunk: int) -> None:
    """Moves the rows of a freshly sorted view into the order in which mpnn_items_build lists them: split rows
    first in row order, then descending degree, row order among equals.  Edges keep their order inside a row."""
    if int(status[0]) != 0:   # the arrays are not trustworthy: report it before indexing with them
        raise IndexError(f"edge_index holds node ids outside [0, {csr.num_rows}) x [0, {csr.num_cols})")
    lib = _lib.load()
    n, e = csr.num_rows, csr.num_edges
    dev = csr.rowptr.device
    order = torch.empty(n, dtype=torch.int32, device=dev)
    rowptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
    col, eid = torch.empty_like(csr.col), torch.empty_like(csr.eid)
    ws_bytes = lib.mpnn_csr_item_order_workspace_bytes(n)
    if ws_bytes < 0:
        raise _lib.MpnnError(f"mpnn_csr_item_order_workspace_bytes({n}) failed")
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
    _lib.check(lib.mpnn_csr_item_order(_lib.ptr(csr.rowptr), _lib.ptr(csr.col), _lib.ptr(csr.eid), n, e, chunk,
                                       _lib.ptr(order), _lib.ptr(rowptr), _lib.ptr(col), _lib.ptr(eid), _lib.ptr(ws),
                                       ws.numel(), _lib.stream_handle(dev)), "mpnn_csr_item_order")
    csr.rowptr, csr.col, csr.eid, csr.order = rowptr, col, eid, order


def _finish(csrs, pending, extra=None):
    """One host read for all CSRs built together: item counts, overflow flag, range-check flag.  ``extra``: one more
    6-entry device record to bring along; its host copy is returned."""
    host = torch.stack(pending + ([extra] if extra is not None else [])).tolist()
    _apply(csrs, host)
    return host[len(csrs)] if extra is not None else None


def _apply(csrs, host) -> None:
    for csr, h in zip(csrs, host):
        if h[4] != 0:
            raise IndexError(f"edge_index holds node ids outside [0, {csr.num_rows}) x [0, {csr.num_cols})")
        if h[3] != 0:
            raise _lib.MpnnError("mpnn_items_build overflowed its output capacity")
        csr.n_items, csr.n_parts, csr.n_split, csr.n_big = h[0], h[1], h[2], h[5]


def _rows_to_ids(csr: CSR) -> None:
    """After an items build over storage positions: the work items and split records name rows by id again."""
    order = csr.order
    if csr.n_items:
        csr.items[:csr.n_items, 0] = order[csr.items[:csr.n_items, 0].long()].to(torch.int32)
    if csr.n_split:
        csr.split[:csr.n_split, 0] = order[csr.split[:csr.n_split, 0].long()].to(torch.int32)


def build_csr(key: torch.Tensor, other: torch.Tensor, num_rows: int, num_cols: int,
              chunk: int | None = None, max_parts: int = MAX_PARTS) -> CSR:
    chunk = default_chunk(key.numel()) if chunk is None else chunk
    csr, pending = _build_csr(key, other, num_rows, num_cols, chunk, max_parts)
    _finish([csr], [pending])
    return csr


class MPGraph:
    """CSR by destination (forward) and by source (backward) of one ``edge_index``.

    ``edge_index`` is int64 ``[2, E]``, row 0 = source, row 1 = target, unsorted, duplicates
    allowed (PyG convention used by the reference, models/layers.py:38,110).
    ``num_dst`` / ``num_src`` differ only for a rank-local piece of a partitioned graph.

    ``row_order="items"`` stores the rows of both views in the order the work items list them instead of by row
    id: the kernels walk the views through the items' (row, first, last) records only, so the index lists, the
    sorted edge attributes and the relu-mask words of consecutive work items then lie next to each other in memory
    (one sequential stream per array instead of a scattered piece per row).  ``rowptr`` is then indexed by storage
    position (``CSR.order[j]`` = id of the row stored at position j); edges inside a row keep their order, so every
    result is bit-identical to the ``"natural"`` layout.
    """

    def __init__(self, edge_index: torch.Tensor, num_dst: int, num_src: int | None = None,
                 chunk: int | None = None, max_parts: int = MAX_PARTS, row_order: str = "natural"):
        _lib.require_device(edge_index, "edge_index")
        if edge_index.dim() != 2 or edge_index.size(0) != 2:
            raise ValueError(f"edge_index must have shape [2, E], got {tuple(edge_index.shape)}")
        if edge_index.dtype != torch.int64:
            raise TypeError(f"edge_index must be int64, got {edge_index.dtype}")
        num_src = num_dst if num_src is None else num_src
        src = edge_index[0].contiguous()
        dst = edge_index[1].contiguous()
        self.num_dst, self.num_src, self.num_edges = int(num_dst), int(num_src), src.numel()
        chunk = default_chunk(self.num_edges) if chunk is None else chunk
        if row_order not in ("natural", "items"):
            raise ValueError(f"row_order must be 'natural' or 'items', got {row_order!r}")
        by_items = row_order == "items" and self.num_edges > 0
        self._ea_key = None
        self._ea_sorted = None
        self._scratch = {}
        self._tpos = None
        self._deferred = None
        self.islands = None
        search = not by_items and self.num_dst == self.num_src and ISLANDS_ENABLED and \
            0 < self.num_dst <= ISLAND_SEARCH_MAX_NODES
        if search and self.num_edges <= SMALL_GRAPHS_MAX_DEGREE * self.num_dst:
            # Probably a batch of small graphs.  The one-launch layers need the destination view and the islands only:
            # build those, read the result, and leave the work items and the whole source view for the first caller
            # that takes the general path (none, for layers the island kernels cover).
            self._by_dst, (status, ws) = _build_csr(dst, src, self.num_dst, self.num_src, chunk, max_parts, False, False)
            ptr, meta = self._search_islands(self._by_dst)
            bad, count, largest, most_edges = torch.cat([status[:1], meta[:3]]).tolist()   # the one host read
            if bad != 0:
                raise IndexError(f"edge_index holds node ids outside [0, {self.num_dst}) x [0, {self.num_src})")
            if count > 0:
                self.islands = Islands(ptr, count, largest, most_edges)
            if count > 0 and largest < ISLAND_GIVE_UP_NODES:
                self._by_src = None
                self._deferred = (src, dst, chunk, max_parts, status, ws, src._version, dst._version)
                return
            p1 = _build_items(self._by_dst, chunk, max_parts, status, ws)
            self._by_src, p2 = _build_csr(src, dst, self.num_src, self.num_dst, chunk, max_parts, False)
            _finish([self._by_dst, self._by_src], [p1, p2])
            return
        self._by_dst, p1 = _build_csr(dst, src, self.num_dst, self.num_src, chunk, max_parts, by_items)
        self._by_src, p2 = _build_csr(src, dst, self.num_src, self.num_dst, chunk, max_parts, by_items)
        found = self._search_islands(self._by_dst) if search else None
        meta = _finish([self._by_dst, self._by_src], [p1, p2], found[1] if found else None)
        if found and meta[0] > 0:
            self.islands = Islands(found[0], meta[0], meta[1], meta[2])
        if by_items:
            _rows_to_ids(self._by_dst)
            _rows_to_ids(self._by_src)

    def _complete(self) -> None:
        """Builds what a two-phase build left out: the work items of the destination view and the source view."""
        src, dst, chunk, max_parts, status, ws, v_src, v_dst = self._deferred
        if torch.cuda.is_current_stream_capturing():
            # the build ends with a host read of the item counts, which a capture cannot hold
            raise RuntimeError("the graph structure of this edge_index is only half built (destination view and "
                               "islands) and a kernel of the general path asks for the rest during a HIP-graph "
                               "capture; call MPGraph.complete() (or run the step once) before capturing")
        if (src._version, dst._version) != (v_src, v_dst):
            # the destination view was sorted from the old contents: a source view of the new ones would not match it
            raise RuntimeError("edge_index was modified in place after its graph structure was built; "
                               "build a new structure (graph_for does, it keys on the tensor's version)")
        self._deferred = None
        p1 = _build_items(self._by_dst, chunk, max_parts, status, ws)
        self._by_src, p2 = _build_csr(src, dst, self.num_src, self.num_dst, chunk, max_parts, False)
        _finish([self._by_dst, self._by_src], [p1, p2])

    def complete(self) -> "MPGraph":
        """Finishes a two-phase build now (work items, source view), e.g. before a HIP-graph capture whose kernels
        may take the general path.  A no-op on a fully built structure."""
        if self._deferred is not None:
            self._complete()
        return self

    @property
    def by_dst(self) -> CSR:
        """Destination-sorted view with its work items (what the general kernels walk)."""
        if self._deferred is not None:
            self._complete()
        return self._by_dst

    @by_dst.setter
    def by_dst(self, csr: CSR) -> None:
        self._by_dst = csr

    @property
    def by_src(self) -> CSR:
        if self._deferred is not None:
            self._complete()
        return self._by_src

    @by_src.setter
    def by_src(self, csr: CSR) -> None:
        self._by_src = csr

    @property
    def dst_rows(self) -> CSR:
        """rowptr / col / eid of the destination view without asking for its work items."""
        return self._by_dst

    def _search_islands(self, csr: CSR):
        """Launches the island search over the destination view; (island_ptr, 6-entry result record)."""
        n = self.num_dst
        lib = _lib.load()
        dev = csr.rowptr.device
        ptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
        meta = torch.empty(6, dtype=torch.int32, device=dev)    # the call clears the 4 entries it owns
        ws = torch.empty(lib.mpnn_islands_workspace_bytes(n), dtype=torch.uint8, device=dev)
        # a row that alone spans more nodes than any one-launch layer takes ends the search at once (citation
        # graphs: the common case among the graphs that have no small islands)
        _lib.check(lib.mpnn_islands_build(_lib.ptr(csr.rowptr), _lib.ptr(csr.col), n,
                                          ISLAND_SEARCH_GIVE_UP, _lib.ptr(ptr), _lib.ptr(meta), _lib.ptr(ws), ws.numel(),
                                          _lib.stream_handle(dev)), "mpnn_islands_build")
        return ptr, meta

    @classmethod
    def from_parts(cls, by_dst: CSR, by_src: CSR, tpos: torch.Tensor | None = None,
                   islands: Islands | None = None) -> "MPGraph":
        """Wraps index structures that were produced elsewhere (the batch assembler writes them directly)."""
        g = cls.__new__(cls)
        g.num_dst, g.num_src, g.num_edges = by_dst.num_rows, by_src.num_rows, by_dst.num_edges
        g._deferred = None
        g.by_dst, g.by_src = by_dst, by_src
        g._ea_key = None
        g._ea_sorted = None
        g._scratch = {}
        g._tpos = tpos
        g.islands = islands if ISLANDS_ENABLED else None
        return g

    def seed_edge_attr(self, edge_attr: torch.Tensor, ea_dst: torch.Tensor, ea_src: torch.Tensor) -> None:
        """Registers the two sorted layouts of ``edge_attr`` when the caller already holds them."""
        self._ea_key = (edge_attr.data_ptr(), edge_attr._version, tuple(edge_attr.shape), edge_attr.dtype,
                        edge_attr.dtype)
        self._ea_holder = edge_attr
        self._ea_sorted = (ea_dst, ea_src)

    @property
    def tpos(self) -> torch.Tensor:
        """tpos[p] = position in the destination-sorted CSR of the edge at transpose position p."""
        if self._tpos is None:
            e = self.num_edges
            inv = torch.empty(e, dtype=torch.int32, device=self.by_dst.eid.device)
            inv[self.by_dst.eid.long()] = torch.arange(e, dtype=torch.int32, device=inv.device)
            self._tpos = inv[self.by_src.eid.long()].contiguous()
        return self._tpos

    def sorted_edge_attr(self, edge_attr: torch.Tensor, dtype: torch.dtype, views: str = "both"):
        """edge_attr laid out in CSR order and in transpose order (cached per tensor version).  ``views="dst"``:
        a 1-tuple with the CSR order only (all the one-launch layers read; it does not ask for the source view)."""
        key = (edge_attr.data_ptr(), edge_attr._version, tuple(edge_attr.shape), edge_attr.dtype, dtype)
        if self._ea_key == key:
            return self._ea_sorted
        if edge_attr.dim() != 2 or edge_attr.size(0) != self.num_edges:
            raise ValueError(f"edge_attr must have shape [{self.num_edges}, Fe], got {tuple(edge_attr.shape)}")
        lib = _lib.load()
        if views == "dst":
            hit = self._scratch.get("ea_dst")
            if hit is not None and hit[0] == key:
                return (hit[2],)
            src = edge_attr.detach().to(dtype).contiguous()
            out = torch.empty_like(src)
            _lib.check(lib.mpnn_gather_rows(_lib.ptr(src), _lib.ptr(self._by_dst.eid), _lib.ptr(out), self.num_edges,
                                            src.size(1) * src.element_size(), _lib.stream_handle(src.device)),
                       "mpnn_gather_rows")
            self._scratch["ea_dst"] = (key, edge_attr, out)   # (holds the tensor: its address cannot be reused)
            return (out,)
        src = edge_attr.detach().to(dtype).contiguous()
        row_bytes = src.size(1) * src.element_size()
        outs = []
        for csr in (self.by_dst, self.by_src):
            out = torch.empty_like(src)
            _lib.check(lib.mpnn_gather_rows(_lib.ptr(src), _lib.ptr(csr.eid), _lib.ptr(out), self.num_edges,
                                            row_bytes, _lib.stream_handle(src.device)), "mpnn_gather_rows")
            outs.append(out)
        self._ea_key = key
        self._ea_holder = edge_attr  # keeps the address from being reused while the cache entry lives
        self._ea_sorted = tuple(outs)
        return self._ea_sorted

    def scratch(self, name: str, shape, dtype, device) -> torch.Tensor:
        """Reusable scratch slabs (partial sums of split rows, gradient partials), one set per stream: calls on
        one stream reuse them in stream order, calls on different streams never share them."""
        key = (name, tuple(shape), dtype, _lib.stream_handle(device))
        buf = self._scratch.get(key)
        if buf is None:
            buf = torch.empty(shape, dtype=dtype, device=device)
            self._scratch[key] = buf
        return buf


def make_graph(edge_index: torch.Tensor, num_dst: int, num_src: int | None = None) -> MPGraph:
    """MPGraph in the layout graph_for() uses for an edge list of this size: rows in work-item order once the
    list is long enough for the scattered per-row pieces to cost more than the one-off reordering."""
    big = edge_index.dim() == 2 and edge_index.size(1) >= ITEM_ORDER_MIN_EDGES
    return MPGraph(edge_index, num_dst, num_src, row_order="items" if big else "natural")


_CACHE: "collections.OrderedDict[tuple, tuple]" = collections.OrderedDict()
_CACHE_SIZE = 8


def graph_for_eager(edge_index: torch.Tensor, num_nodes: int) -> MPGraph:
    """Cached MPGraph of ``edge_index``.

    The cache entry holds a reference to the tensor, so its storage address cannot be
    recycled for another edge list while the entry is alive; an in-place edit bumps
    ``_version`` and misses.
    """
    key = (edge_index.data_ptr(), edge_index._version, edge_index.shape, edge_index.stride(), edge_index.device,
           int(num_nodes))
    hit = _CACHE.get(key)
    if hit is not None:
        _CACHE.move_to_end(key)
        return hit[1]
    g = make_graph(edge_index, num_nodes)
    _CACHE[key] = (edge_index, g)
    while len(_CACHE) > _CACHE_SIZE:
        _CACHE.popitem(last=False)
    return g


# for callers that may be traced (the same function behind the marker that keeps a tracing compiler out; callers
# that know they run eagerly save the marker's few microseconds per call)
graph_for = torch.compiler.disable(graph_for_eager)


def register_graph(edge_index: torch.Tensor, num_nodes: int, g: MPGraph) -> None:
    """Puts a ready-made MPGraph into the cache under the key graph_for() will compute for ``edge_index``."""
    key = (edge_index.data_ptr(), edge_index._version, edge_index.shape, edge_index.stride(), edge_index.device,
           int(num_nodes))
    _CACHE[key] = (edge_index, g)
    _CACHE.move_to_end(key)
    while len(_CACHE) > _CACHE_SIZE:
        _CACHE.popitem(last=False)


def forget_graph(edge_index: torch.Tensor) -> None:
    """Drops every cache entry built for the storage of ``edge_index``.  For callers that rewrite an edge list
    through raw device pointers: such writes do not bump the tensor's version counter, so an entry built from the
    earlier contents would still match."""
    ptr = edge_index.data_ptr()
    for key in [k for k in _CACHE if k[0] == ptr]:
        del _CACHE[key]


def clear_cache() -> None:
    _CACHE.clear()
