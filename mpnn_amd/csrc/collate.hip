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
__global__ void __launch_bounds__(kBlock) k_collate_nodes(CollateArgs a) {
  const mpnn_collate_desc& d = a.d;
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int B = d.num_slots;
  if (p == d.num_nodes) {
    d.rowptr_dst[p] = (int32_t)d.num_edges;
    d.rowptr_src[p] = (int32_t)d.num_edges;
  }
  if (p < B && d.y_all) copy_row<W>(d.y_all, d.slots[4 * (B + 1) + p], d.y, p, d.y_row_bytes);  // p as slot
  if (p >= d.num_nodes) return;
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

template <typename W>
__global__ void __launch_bounds__(kBlock) k_collate_edges(CollateArgs a) {
  const mpnn_collate_desc& d = a.d;
  const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (q >= d.num_edges) return;
  const int B = d.num_slots;
  const int b = find_slot(d.slots + (B + 1), B, (int32_t)q);
  const int32_t dn = d.slots[2 * (B + 1) + b], de = d.slots[3 * (B + 1) + b];
  const int64_t e = q - de;
  d.edge_index[q] = d.src_all[e] + dn;
  d.edge_index[d.num_edges + q] = d.dst_all[e] + dn;
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
  if (d.num_edges > 0 &&
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
  const bool w4 = !((d.x_row_bytes | d.ea_row_bytes | (d.pos_all ? d.pos_row_bytes : 0) | (d.y_all ? d.y_row_bytes : 0)) & 3);
  CollateArgs a{d};
  const int64_t node_threads = (d.num_nodes + 1 > d.num_slots) ? d.num_nodes + 1 : d.num_slots;
  const unsigned gn = (unsigned)((node_threads + kBlock - 1) / kBlock);
  if (w4) k_collate_nodes<uint32_t><<<gn, kBlock, 0, stream>>>(a);
  else k_collate_nodes<uint16_t><<<gn, kBlock, 0, stream>>>(a);
  MPNN_LAUNCH_CHECK("k_collate_nodes");
  if (d.num_edges > 0) {
    const unsigned ge = (unsigned)((d.num_edges + kBlock - 1) / kBlock);
    if (w4) k_collate_edges<uint32_t><<<ge, kBlock, 0, stream>>>(a);
    else k_collate_edges<uint16_t><<<ge, kBlock, 0, stream>>>(a);
    MPNN_LAUNCH_CHECK("k_collate_edges");
  }
  return MPNN_OK;
}
