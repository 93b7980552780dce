// Mini-batch assembly on the device: the concatenation of a list of small graphs (node offset added to
// edge_index, `batch` vector, stacked features) together with the batch's destination-sorted CSR, its
// source-sorted transpose, the edge attributes in both orders, the transpose positions and the work items.
//
// The reference leaves batching to torch_geometric's DataLoader in 8 CPU worker processes
// (data/dataset.py:94-96, 215-216) and the model then re-derives the index structure inside propagate()
// every step (models/layers.py:38,110).  Here the per-graph structure is sorted ONCE for the whole dataset
// (graphs are block-diagonal: the CSR of a batch is the concatenation of its graphs' CSR segments with
// offsets), so a batch costs two copy kernels and no sort.
#include "common.h"

namespace mpnn {

struct CollateArgs {
  mpnn_collate_desc d;
};

// largest b in [0, n) with off[b] <= i   (off is ascending, off[0] == 0, i < off[n])
__device__ __forceinline__ int find_slot(const int32_t* __restrict__ off, int n, int32_t i) {
  int lo = 0, hi = n;  // invariant: off[lo] <= i < off[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (off[mid] <= i) lo = mid; else hi = mid;
  }
  return lo;
}

template <typename W>
__device__ __forceinline__ void copy_row(const void* src, int64_t srow, void* dst, int64_t drow, int row_bytes) {
  const int words = row_bytes / (int)sizeof(W);
  const W* s = reinterpret_cast<const W*>(src) + srow * words;
  W* o = reinterpret_cast<W*>(dst) + drow * words;
  for (int i = 0; i < words; ++i) o[i] = s[i];
}

template <typename W>
__device__ __forceinline__ void zero_row(void* dst, int64_t drow, int row_bytes) {
  const int words = row_bytes / (int)sizeof(W);
  W* o = reinterpret_cast<W*>(dst) + drow * words;
  for (int i = 0; i < words; ++i) o[i] = 0;
}

// PADDED: outputs of capacity size, real totals read from the slot table on the device (see the header).
template <typename W, bool PADDED>
__global__ void __launch_bounds__(kBlock) k_collate_nodes(CollateArgs a) {
  const mpnn_collate_desc& d = a.d;
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int B = d.num_slots;
  const int64_t n_real = PADDED ? (int64_t)d.slots[B] : d.num_nodes;
  const int64_t e_real = PADDED ? (int64_t)d.slots[(B + 1) + B] : d.num_edges;
  const int64_t n_out = PADDED ? d.node_capacity : d.num_nodes;
  if (p == n_out) {
    d.rowptr_dst[p] = (int32_t)e_real;
    d.rowptr_src[p] = (int32_t)e_real;
  }
  if (p < B && d.y_all) copy_row<W>(d.y_all, d.slots[4 * (B + 1) + p], d.y, p, d.y_row_bytes);  // p as slot
  if (PADDED && d.pool_rowptr) {
    if (p <= B) d.pool_rowptr[p] = d.slots[p];
    if (p == B + 1) d.pool_rowptr[p] = (int32_t)n_out;
  }
  if (p >= n_out) return;
  if (PADDED && p >= n_real) {   // padding row
    d.batch[p] = B;
    d.rowptr_dst[p] = (int32_t)e_real;
    d.rowptr_src[p] = (int32_t)e_real;
    reinterpret_cast<int4*>(d.items_dst)[p] = make_int4((int32_t)p, (int32_t)e_real, (int32_t)e_real, -1);
    reinterpret_cast<int4*>(d.items_src)[p] = make_int4((int32_t)p, (int32_t)e_real, (int32_t)e_real, -1);
    zero_row<W>(d.x, p, d.x_row_bytes);
    if (d.pos_all) zero_row<W>(d.pos, p, d.pos_row_bytes);
    return;
  }
  const int b = find_slot(d.slots, B, (int32_t)p);
  const int32_t dn = d.slots[2 * (B + 1) + b], de = d.slots[3 * (B + 1) + b];
  const int64_t v = p - dn;
  d.batch[p] = b;
  {
    const int32_t r0 = d.rowptr_dst_all[v] + de, r1 = d.rowptr_dst_all[v + 1] + de;
    d.rowptr_dst[p] = r0;
    reinterpret_cast<int4*>(d.items_dst)[p] = make_int4((int32_t)p, r0, r1, -1);
  }
  {
    const int32_t r0 = d.rowptr_src_all[v] + de, r1 = d.rowptr_src_all[v + 1] + de;
    d.rowptr_src[p] = r0;
    reinterpret_cast<int4*>(d.items_src)[p] = make_int4((int32_t)p, r0, r1, -1);
  }
  copy_row<W>(d.x_all, v, d.x, p, d.x_row_bytes);
  if (d.pos_all) copy_row<W>(d.pos_all, v, d.pos, p, d.pos_row_bytes);
}

template <typename W, bool PADDED>
__global__ void __launch_bounds__(kBlock) k_collate_edges(CollateArgs a) {
  const mpnn_collate_desc& d = a.d;
  const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int B = d.num_slots;
  const int64_t e_real = PADDED ? (int64_t)d.slots[(B + 1) + B] : d.num_edges;
  const int64_t e_out = PADDED ? d.edge_capacity : d.num_edges;   // also the row length of edge_index
  if (q >= e_out) return;
  if (PADDED && q >= e_real) {   // padding entry: outside every CSR row
    const int64_t pad = d.node_capacity - 1;
    d.edge_index[q] = pad;
    d.edge_index[e_out + q] = pad;
    d.col_dst[q] = (int32_t)pad;
    d.col_src[q] = (int32_t)pad;
    d.eid_dst[q] = (int32_t)q;
    d.eid_src[q] = (int32_t)q;
    d.tpos[q] = (int32_t)q;
    zero_row<W>(d.ea, q, d.ea_row_bytes);
    zero_row<W>(d.ea_dst, q, d.ea_row_bytes);
    zero_row<W>(d.ea_src, q, d.ea_row_bytes);
    return;
  }
  const int b = find_slot(d.slots + (B + 1), B, (int32_t)q);
  const int32_t dn = d.slots[2 * (B + 1) + b], de = d.slots[3 * (B + 1) + b];
  const int64_t e = q - de;
  d.edge_index[q] = d.src_all[e] + dn;
  d.edge_index[e_out + q] = d.dst_all[e] + dn;
  d.col_dst[q] = d.col_dst_all[e] + dn;
  d.eid_dst[q] = d.eid_dst_all[e] + de;
  d.col_src[q] = d.col_src_all[e] + dn;
  d.eid_src[q] = d.eid_src_all[e] + de;
  d.tpos[q] = d.tpos_all[e] + de;
  copy_row<W>(d.ea_all, e, d.ea, q, d.ea_row_bytes);
  copy_row<W>(d.ea_dst_all, e, d.ea_dst, q, d.ea_row_bytes);
  copy_row<W>(d.ea_src_all, e, d.ea_src, q, d.ea_row_bytes);
}

}  // namespace mpnn

using namespace mpnn;

extern "C" int32_t mpnn_collate(const mpnn_collate_desc* desc, mpnn_stream_t stream) {
  if (!desc) return fail(MPNN_E_INVALID, "mpnn_collate: null descriptor");
  const mpnn_collate_desc& d = *desc;
  if (d.num_slots < 1 || d.num_nodes < 0 || d.num_edges < 0 || d.num_nodes >= INT32_MAX || d.num_edges >= INT32_MAX)
    return fail(MPNN_E_INVALID, "mpnn_collate: slots=%d nodes=%lld edges=%lld", d.num_slots, (long long)d.num_nodes,
                (long long)d.num_edges);
  if (!d.slots || !d.x_all || !d.x || !d.batch || !d.rowptr_dst_all || !d.rowptr_src_all || !d.rowptr_dst ||
      !d.rowptr_src || !d.items_dst || !d.items_src)
    return fail(MPNN_E_INVALID, "mpnn_collate: null node-side pointer");
  if ((d.node_capacity > 0 ? d.edge_capacity : d.num_edges) > 0 &&
      (!d.src_all || !d.dst_all || !d.edge_index || !d.ea_all || !d.ea_dst_all || !d.ea_src_all || !d.ea || !d.ea_dst ||
       !d.ea_src || !d.col_dst_all || !d.eid_dst_all || !d.col_src_all || !d.eid_src_all || !d.tpos_all || !d.col_dst ||
       !d.eid_dst || !d.col_src || !d.eid_src || !d.tpos))
    return fail(MPNN_E_INVALID, "mpnn_collate: null edge-side pointer");
  if (d.pos_all && !d.pos) return fail(MPNN_E_INVALID, "mpnn_collate: pos_all without pos");
  if (d.y_all && (!d.y || d.y_row_bytes < 2 || (d.y_row_bytes & 1)))
    return fail(MPNN_E_INVALID, "mpnn_collate: y_all needs y and a row size that is a positive multiple of 2");
  if (d.x_row_bytes < 2 || (d.x_row_bytes & 1) || d.ea_row_bytes < 2 || (d.ea_row_bytes & 1) ||
      (d.pos_all && (d.pos_row_bytes < 2 || (d.pos_row_bytes & 1))))
    return fail(MPNN_E_INVALID, "mpnn_collate: row sizes must be positive multiples of 2 bytes");
  if (((uintptr_t)d.items_dst | (uintptr_t)d.items_src) & 15)
    return fail(MPNN_E_INVALID, "mpnn_collate: items arrays must be 16-byte aligned");
  const bool padded = d.node_capacity > 0;
  if (padded && (d.edge_capacity < 0 || d.node_capacity >= INT32_MAX || d.edge_capacity >= INT32_MAX))
    return fail(MPNN_E_INVALID, "mpnn_collate: capacities %lld nodes, %lld edges", (long long)d.node_capacity,
                (long long)d.edge_capacity);
  const int64_t n_out = padded ? d.node_capacity : d.num_nodes, e_out = padded ? d.edge_capacity : d.num_edges;
  const bool w4 = !((d.x_row_bytes | d.ea_row_bytes | (d.pos_all ? d.pos_row_bytes : 0) | (d.y_all ? d.y_row_bytes : 0)) & 3);
  CollateArgs a{d};
  int64_t node_threads = n_out + 1;
  if (node_threads < d.num_slots + 2) node_threads = d.num_slots + 2;
  const unsigned gn = (unsigned)((node_threads + kBlock - 1) / kBlock);
  if (padded) {
    if (w4) k_collate_nodes<uint32_t, true><<<gn, kBlock, 0, stream>>>(a);
    else k_collate_nodes<uint16_t, true><<<gn, kBlock, 0, stream>>>(a);
  } else {
    if (w4) k_collate_nodes<uint32_t, false><<<gn, kBlock, 0, stream>>>(a);
    else k_collate_nodes<uint16_t, false><<<gn, kBlock, 0, stream>>>(a);
  }
  MPNN_LAUNCH_CHECK("k_collate_nodes");
  if (e_out > 0) {
    const unsigned ge = (unsigned)((e_out + kBlock - 1) / kBlock);
    if (padded) {
      if (w4) k_collate_edges<uint32_t, true><<<ge, kBlock, 0, stream>>>(a);
      else k_collate_edges<uint16_t, true><<<ge, kBlock, 0, stream>>>(a);
    } else {
      if (w4) k_collate_edges<uint32_t, false><<<ge, kBlock, 0, stream>>>(a);
      else k_collate_edges<uint16_t, false><<<ge, kBlock, 0, stream>>>(a);
    }
    MPNN_LAUNCH_CHECK("k_collate_edges");
  }
  return MPNN_OK;
}
