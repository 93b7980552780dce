/*
 * mpnn_amd.h — C ABI of the MI355X (gfx950) message-passing engine.
 *
 * The reference (KutayTasci/MPNN) has no FFI of its own: its hot path is the
 * Python call  MessagePassing.propagate(edge_index, x=..., edge_attr=...)  made at
 * models/layers.py:38 (MPNNLayerCat) and models/layers.py:110 (MPNNLayerSum), plus
 * global_mean_pool(x, batch) at models/mpnn.py:54,150.  Each entry point below names
 * the part of that call it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the comment says "host";
 *   - every function returns 0 on success and a negative MPNN_E_* code otherwise;
 *     mpnn_last_error() returns a thread-local description of the last failure;
 *   - all work is enqueued on `stream`; no function synchronises the device, allocates
 *     or frees device memory; scratch space is passed in by the caller
 *     (sizes come from the *_workspace_bytes functions);
 *   - dtype: MPNN_F32 (float) or MPNN_BF16 (bfloat16 storage, float accumulation);
 *   - feature matrices are row-major with an explicit leading dimension `ld*` counted
 *     in ELEMENTS, so the two halves of a [N,2H] fused projection can be passed
 *     without a copy (models/layers.py:105, `fused_out.chunk(2, dim=1)`).
 */
#ifndef MPNN_AMD_H
#define MPNN_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* mpnn_stream_t; /* == hipStream_t */

enum { MPNN_F32 = 0, MPNN_BF16 = 1, MPNN_U8 = 2 /* only as the kind of a GEMM epilogue multiplier */ };

enum {
  MPNN_OK = 0,
  MPNN_E_INVALID = -1,   /* bad argument (null pointer, size, alignment, unsupported shape) */
  MPNN_E_WORKSPACE = -2, /* workspace too small */
  MPNN_E_HIP = -3,       /* a HIP runtime call or kernel launch failed */
  MPNN_E_CAPACITY = -4   /* an output array's stated capacity is too small */
};

#define MPNN_ABI_VERSION 17

/* Library self-description. */
int32_t mpnn_abi_version(void);
const char* mpnn_last_error(void);

/* ---------------------------------------------------------------------------
 * Graph structure (integer-exact).
 *
 * Replaces the implicit index handling inside PyG's propagate():
 * `index_select(·, 0, edge_index[0|1])` and `scatter_add_(0, edge_index[1], ·)`
 * (call sites models/layers.py:38,110).
 *
 * mpnn_csr_build sorts the edge list by `key` (stable: edges with equal key keep
 * their original order) and produces
 *     rowptr[num_keys+1]  first sorted position of every key value (int32)
 *     col[E]       `other` of the edge at each sorted position      (int32)
 *     eid[E]       original edge id at each sorted position         (int32)
 * Call it with (key=dst=edge_index[1], other=src=edge_index[0]) for the
 * destination-sorted CSR the forward pass walks, and with the roles swapped for
 * the source-sorted transpose the backward pass walks.
 * Keys must lie in [0, num_keys) and `other` values in [0, num_other) (the two differ
 * when a rank owns only a range of destination rows but reads every source row).
 * status[0] is zeroed, then set to 1 on the device if any value is out of range; such
 * edges are still placed (clamped) so no access is out of bounds, and the caller is
 * expected to read status back and raise.
 * Limits: num_keys, num_other, E < 2^31.
 * ------------------------------------------------------------------------- */
int64_t mpnn_csr_workspace_bytes(int64_t num_keys, int64_t num_edges);
int32_t mpnn_csr_build(const int64_t* key, const int64_t* other, int64_t num_edges,
                       int64_t num_keys, int64_t num_other, int32_t* rowptr, int32_t* col,
                       int32_t* eid, int32_t* status, void* workspace, int64_t workspace_bytes,
                       mpnn_stream_t stream);

/* Work decomposition of a CSR into items of bounded length.
 *
 * A row with deg <= chunk is one item; a longer row is cut into pieces of
 * max(chunk, ceil(deg / max_parts)) edges whose partial sums are combined in piece
 * order afterwards (fixed order => bitwise reproducible results, no atomics).
 *   items[4*t .. 4*t+3] = { row, e0, e1, pslot }   pslot = -1 for an unsplit row,
 *                                                    else the piece's partial-sum slot
 *   split[4*s .. 4*s+3] = { row, first pslot, pieces, first item }
 *   meta[0..2]          = { number of items, number of partial slots, number of split rows }
 * Items of split rows come first (row order, piece order), then unsplit rows by descending
 * degree (row order among equal degrees), so that the lane groups of a wave get items of like length.
 * Capacity needed: items <= N + E/chunk + 1 ; split <= E/(chunk+1) + 1.
 */
int64_t mpnn_items_workspace_bytes(int64_t num_nodes);
int32_t mpnn_items_build(const int32_t* rowptr, int64_t num_nodes, int32_t chunk,
                         int32_t max_parts, int32_t* items, int64_t items_capacity,
                         int32_t* split, int64_t split_capacity, int32_t* meta,
                         void* workspace, int64_t workspace_bytes, mpnn_stream_t stream);

/* Rows of a sorted view (rowptr/col/eid from mpnn_csr_build) moved into the order in which mpnn_items_build lists
 * them for the same `chunk`: rows longer than chunk first (row order), then by descending length, row order among
 * equals.  Edges keep their order inside a row.  The edge kernels reach a view only through the work items'
 * {row, first, last} records, so with the rows in item order the index list, the sorted edge attributes and the
 * relu-mask words of consecutive items lie next to each other (one sequential stream per array).
 *   order[j]      id of the row stored at position j (int32 [N])
 *   rowptr_out    [N + 1], indexed by storage position
 *   col_out/eid_out [E], must not alias col/eid
 * Same index handling as mpnn_csr_build replaces (models/layers.py:38,110); no reference counterpart of its own. */
int64_t mpnn_csr_item_order_workspace_bytes(int64_t num_nodes);
int32_t mpnn_csr_item_order(const int32_t* rowptr, const int32_t* col, const int32_t* eid,
                            int64_t num_nodes, int64_t num_edges, int32_t chunk, int32_t* order,
                            int32_t* rowptr_out, int32_t* col_out, int32_t* eid_out, void* workspace,
                            int64_t workspace_bytes, mpnn_stream_t stream);

/* dst[r, :] = src[index[r], :] for rows of `row_bytes` bytes (row_bytes % 2 == 0).
 * Used to lay edge_attr out in CSR order once per graph (eid / eid_t from above). */
int32_t mpnn_gather_rows(const void* src, const int32_t* index, void* dst, int64_t rows,
                         int64_t row_bytes, mpnn_stream_t stream);

/* ---------------------------------------------------------------------------
 * Mini-batch assembly on the device.
 *
 * Replaces torch_geometric's DataLoader collation, which the reference runs in 8 CPU worker
 * processes (data/dataset.py:94-96, 215-216; loops training/train.py:96-98), AND the per-batch
 * rebuild of the index structure: a dataset of small graphs is block-diagonal, so it is sorted
 * once as a whole (mpnn_csr_build over the concatenated dataset) and the CSR of a batch is the
 * concatenation of its graphs' CSR segments with offsets.  One call writes, for the graphs placed
 * in the batch slots:
 *     x, pos            node rows stacked in slot order
 *     batch[p]          slot of node p (int64, the `batch` vector of PyG)
 *     edge_index[2][E]  dataset edge list of each graph + the slot's node offset (int64)
 *     ea, ea_dst, ea_src  edge attributes in edge order / destination-sorted / source-sorted order
 *     rowptr, col, eid, items of both sorted views, and tpos (transpose position -> sorted position)
 * Work items are one per row in row order ({row, e0, e1, -1}); the caller must use the general builder
 * for a dataset that holds a row longer than its chunk length.
 *
 * slots (device, int32 [5][num_slots + 1]): row 0 = first batch node of slot b, row 1 = first batch
 * edge, row 2 = row 0 minus the first DATASET node of the slot's graph, row 3 = row 1 minus its first
 * dataset edge, row 4 = the graph's index in the dataset (selects its row of y_all); entry num_slots
 * of rows 0 and 1 holds the batch totals.
 * *_all arrays are dataset-level: src_all/dst_all/col_*_all hold dataset-global node ids, eid_*_all and
 * tpos_all dataset-global edge positions.  Row sizes are in bytes (multiples of 2).
 * The descriptor itself is read on the host.
 * ------------------------------------------------------------------------- */
typedef struct mpnn_collate_desc {
  const int32_t* slots;
  int32_t num_slots;
  int64_t num_nodes, num_edges; /* batch totals */
  const void* x_all;
  int32_t x_row_bytes;
  const void* pos_all; /* optional */
  int32_t pos_row_bytes;
  const void* y_all; /* optional, one row per dataset graph */
  int32_t y_row_bytes;
  const int64_t* src_all;
  const int64_t* dst_all;
  const void* ea_all;
  const void* ea_dst_all;
  const void* ea_src_all;
  int32_t ea_row_bytes;
  const int32_t* rowptr_dst_all;
  const int32_t* col_dst_all;
  const int32_t* eid_dst_all;
  const int32_t* rowptr_src_all;
  const int32_t* col_src_all;
  const int32_t* eid_src_all;
  const int32_t* tpos_all;
  /* outputs */
  void* x;
  void* pos;
  void* y; /* [num_slots] rows */
  int64_t* batch;
  int64_t* edge_index;
  void* ea;
  void* ea_dst;
  void* ea_src;
  int32_t* rowptr_dst;
  int32_t* col_dst;
  int32_t* eid_dst;
  int32_t* items_dst;
  int32_t* rowptr_src;
  int32_t* col_src;
  int32_t* eid_src;
  int32_t* items_src;
  int32_t* tpos;
  /* Padded mode (node_capacity > 0): the outputs have node_capacity rows / edge_capacity edge entries whatever
   * the batch holds, and the real totals are READ ON THE DEVICE from slots[0][num_slots] and slots[1][num_slots]
   * (num_nodes / num_edges above are ignored), so a launch recorded in a HIP graph assembles batches of any size
   * up to the capacity.  node_capacity must exceed the real node count.  Padding rows: x, pos = 0,
   * batch = num_slots (a padding graph), empty CSR rows and work items; padding edge entries: both ends
   * = node_capacity - 1 in edge_index and col (they belong to no CSR row), eid = tpos = own position,
   * attributes 0.  pool_rowptr (optional, [num_slots + 2]): first node of every graph, then the real total,
   * then node_capacity - the segment boundaries for pooling over batch, padding graph last. */
  int64_t node_capacity, edge_capacity;
  int32_t* pool_rowptr;
} mpnn_collate_desc;
int32_t mpnn_collate(const mpnn_collate_desc* desc, mpnn_stream_t stream);

/* ---------------------------------------------------------------------------
 * Fused message + sum aggregation, forward.
 *
 * Replaces, for one layer, PyG's gather -> message() -> scatter_add_ chain:
 *   models/layers.py:128  relu((xs_i + xt_j) + es)           (MPNNLayerSum.message)
 *   models/layers.py:56   relu(W [x_i | x_j | e])            (MPNNLayerCat.message, after
 *                         splitting W column-wise into W_i, W_j, W_e and projecting nodes first)
 *   models/layers.py:108  es = linear_3(edge_attr)           (recomputed per edge in registers)
 *   aggr='add'            models/layers.py:20,82
 *
 *   out[n, :] = sum_{p in [rowptr[n], rowptr[n+1])}
 *                  relu((A[n, :] + B[col[p], :]) + We @ ea_sorted[p, :])
 *
 * A, B      [N, H] with leading dimensions ldA, ldB (elements)
 * ea_sorted [E, Fe] edge attributes ALREADY in CSR order (see mpnn_gather_rows)
 * We        [H, Fe] row-major (the Linear weight layout), same dtype as A
 * items/split/n_*  from mpnn_items_build for this CSR
 * out       [N, H] leading dimension ldo
 * cnt       optional uint8 [N, H] (row stride H): number of edges of row n whose message is > 0 in
 *           column h, saturated at 255 (needed by the backward: d out / d A = grad * cnt); may be NULL.
 *           A byte per count quarters the bytes this pass writes and the backward reads; exact for every
 *           row of at most 255 edges.  The rows that can hold more keep exact float counts in
 * cnt_big   float [n_split + n_big, H]: row s < n_split = the row of split record s; row n_split + j = the
 *           row of item n_parts + j, j < n_big, where n_big = the number of UNSPLIT rows with more than 255
 *           edges (they are the first unsplit items: unsplit items are listed by descending length).
 *           mpnn_count_fixup rewrites grad * cnt for exactly these rows.  The byte counts of these rows are
 *           unspecified.  May be NULL when cnt is NULL or n_split + n_big == 0.
 * part_out, part_cnt  float [n_parts, H] scratch for split rows (part_cnt may be NULL
 *           iff cnt is NULL); may be NULL when n_parts == 0
 * maskbits  optional uint32 [E, mpnn_mask_words(H, dtype)], 16-byte aligned: per CSR position the
 *           relu mask of the message, one bit per column, in a layout private to this call and
 *           mpnn_aggregate_bwd_src_masked, which consumes it (bfloat16 rows of 33..128 columns: one byte
 *           per 8 columns, bit 7 - v of byte l <=> column 8 l + v, 8 or 16 bytes per edge; otherwise
 *           word v, bit l <=> column l * words + v); a NaN message counts as positive in the bfloat16
 *           byte layout (torch's relu backward would give it 0; the forward output is NaN either way);
 *           needs cnt != NULL; pass NULL when
 *           mpnn_mask_words(H, dtype) == 0 (H too wide or not a multiple of the vector width)
 * Requirements: H % 4 == 0 (f32) or H % 8 == 0 (bf16) for the vector path, otherwise a
 * scalar path is used; base pointers 16-byte aligned and ld* % vector width == 0 for
 * the vector path. 1 <= Fe <= 8.
 * ------------------------------------------------------------------------- */
int32_t mpnn_aggregate_fwd(const void* A, int64_t ldA, const void* B, int64_t ldB,
                           const void* ea_sorted, int32_t Fe, const void* We,
                           const int32_t* col, const int32_t* items, int64_t n_items,
                           const int32_t* split, int64_t n_split, int64_t n_parts,
                           void* out, int64_t ldo, uint8_t* cnt, float* cnt_big, int64_t n_big,
                           float* part_out, float* part_cnt, uint32_t* maskbits, int64_t num_nodes,
                           int32_t H, int32_t dtype, mpnn_stream_t stream);

/* gA[row, :] = G[row, :] * cnt_big[slot, :] for the rows listed by cnt_big's slots (see mpnn_aggregate_fwd):
 * run after the GEMM whose epilogue wrote gA = G * cnt from the byte counts, it makes d out / d A exact for the
 * rows whose counts can exceed a byte.  Part of the autograd backward of the aggregation (training/train.py:23,48,76). */
int32_t mpnn_count_fixup(const void* G, int64_t ldG, const float* cnt_big, const int32_t* items,
                         const int32_t* split, int64_t n_split, int64_t n_parts, int64_t n_big,
                         void* gA, int64_t ldgA, int32_t H, int32_t dtype, mpnn_stream_t stream);

/* mpnn_aggregate_fwd for callers that cannot allocate device memory at all (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc): the partial-sum slabs of split rows (part_out, part_cnt)
 * live in a device array built into the library instead of being passed in.  No counterpart in the reference.
 * Every other argument, the arithmetic and the results are those of mpnn_aggregate_fwd (bit for bit).
 *   - limit: align64(n_parts * H) floats, twice that when cnt != NULL, must fit MPNN_NOALLOC_SCRATCH_BYTES;
 *     beyond it the call fails with MPNN_E_CAPACITY and touches nothing;
 *   - the call itself performs no allocation, free or synchronisation; the first call on a device looks the
 *     array's address up (hipGetSymbolAddress), so make that first call outside a stream capture;
 *   - `stream` must belong to the CURRENT device (hipGetDevice): the array used is the current device's copy;
 *   - each GPU holds one array for the whole process: calls that overlap on one GPU (different streams) race on
 *     it - order them on one stream or use mpnn_aggregate_fwd with separate slabs.  Contents are not kept
 *     between calls and nothing needs clearing (every piece writes its slab row before it is read). */
#define MPNN_NOALLOC_SCRATCH_BYTES (16 << 20)
int32_t mpnn_aggregate_fwd_noalloc(const void* A, int64_t ldA, const void* B, int64_t ldB,
                                   const void* ea_sorted, int32_t Fe, const void* We,
                                   const int32_t* col, const int32_t* items, int64_t n_items,
                                   const int32_t* split, int64_t n_split, int64_t n_parts,
                                   void* out, int64_t ldo, uint8_t* cnt, float* cnt_big, int64_t n_big,
                                   uint32_t* maskbits, int64_t num_nodes, int32_t H, int32_t dtype,
                                   mpnn_stream_t stream);

/* ---------------------------------------------------------------------------
 * Backward of the aggregation with respect to B (the per-source table) and We.
 *
 * Replaces what autograd derives for the chain above (triggered by loss.backward(),
 * training/train.py:23,48,76): grad_msg = grad_aggr[dst]; masked by msg > 0;
 * index_add_ by src; and grad of linear_3.
 *
 * Walks the SOURCE-sorted transpose (rowptr_t/col_t hold destination ids):
 *   gB[s, :]  = sum_{p in row s} mask_p * G[col_t[p], :]
 *   gWe[h, f] = sum_p mask_p[h] * G[col_t[p], h] * ea_sorted_t[p, f]
 *   mask_p    = ((A[col_t[p], :] + B[s, :]) + We @ ea_sorted_t[p, :]) > 0
 * The gradient with respect to A needs no edge pass: gA = G * cnt (elementwise).
 *
 * G            [Nd, H] gradient of the loss w.r.t. `out` of the forward (ldG)
 * ea_sorted_t  [E, Fe] edge attributes in TRANSPOSE order
 * gB           [Ns, H] (ldgB)
 * gWe          [H, Fe] same dtype as A
 * part_gB      float [n_parts, H] scratch for split rows (NULL when n_parts == 0)
 * gwe_scratch  float, at least mpnn_bwd_scratch_floats(H, Fe) elements
 * ------------------------------------------------------------------------- */
int64_t mpnn_bwd_scratch_floats(int32_t H, int32_t Fe);
int32_t mpnn_mask_words(int32_t H, int32_t dtype);
int32_t mpnn_aggregate_bwd_src(const void* A, int64_t ldA, const void* B, int64_t ldB,
                               const void* G, int64_t ldG, const void* ea_sorted_t,
                               int32_t Fe, const void* We, const int32_t* col_t,
                               const int32_t* items_t, int64_t n_items_t,
                               const int32_t* split_t, int64_t n_split_t, int64_t n_parts_t,
                               void* gB, int64_t ldgB, void* gWe, float* part_gB,
                               float* gwe_scratch, int64_t num_src_nodes, int32_t H,
                               int32_t dtype, mpnn_stream_t stream);

/* Same gradients from the mask bits the forward stored instead of recomputing the pre-activation:
 * one row gather per edge (G[dst]) instead of two.
 *   tpos[p] = position, in the destination-sorted CSR, of the edge at transpose position p
 *             (tpos = inverse(eid)[eid_t]); maskbits as written by mpnn_aggregate_fwd.
 *   gB[s, :]  = sum_{p in row s} bits(maskbits[tpos[p]]) * G[col_t[p], :]
 *   gWe[h, f] = sum_p bit_h(maskbits[tpos[p]]) * G[col_t[p], h] * ea_sorted_t[p, f]
 */
int32_t mpnn_aggregate_bwd_src_masked(const void* G, int64_t ldG, const uint32_t* maskbits,
                                      const int32_t* tpos, const void* ea_sorted_t, int32_t Fe,
                                      const int32_t* col_t, const int32_t* items_t,
                                      int64_t n_items_t, const int32_t* split_t, int64_t n_split_t,
                                      int64_t n_parts_t, void* gB, int64_t ldgB, void* gWe,
                                      float* part_gB, float* gwe_scratch, int64_t num_src_nodes,
                                      int32_t H, int32_t dtype, mpnn_stream_t stream);

/* ---------------------------------------------------------------------------
 * EGNN coordinate update fused into the aggregation pass.
 *
 * No counterpart in the reference (its models/layers.py holds only MPNNLayerCat and
 * MPNNLayerSum); the operator extends the message pass of models/layers.py:128 by
 *     m_e    = relu((A[dst] + B[src]) + We @ edge_attr[e])
 *     out_n  = sum_{e -> n} m_e
 *     pos'_n = pos_n + sum_{e -> n} (pos_n - pos_src(e)) * (w_pos . m_e)
 * Coordinates and their gradients are float [*, 3] whatever the feature dtype.
 * Requires the vector path: H % 4 == 0 (f32) / H % 8 == 0 (bf16), H <= 256 / 512,
 * 16-byte aligned feature matrices.  part_small is float [n_parts, 4] scratch.
 *
 * Backward, with G = d loss / d out and gpos_out = d loss / d pos':
 *   mpnn_egnn_bwd_dst (destination-sorted CSR):  gA, S[n] = sum_{e -> n} (w_pos . m_e), g_wpos
 *   mpnn_egnn_bwd_src (source-sorted transpose): gB, gWe, gpos_src[s] = -sum_{e from s} (w_pos . m_e) gpos_out[dst]
 *   d loss / d pos = gpos_out * (1 + S) + gpos_src     (elementwise, left to the caller)
 * scratch: float, at least mpnn_bwd_scratch_floats(H, Fe) elements.
 * ------------------------------------------------------------------------- */
int32_t mpnn_egnn_fwd(const void* A, int64_t ldA, const void* B, int64_t ldB,
                      const void* ea_sorted, int32_t Fe, const void* We, const void* w_pos,
                      const float* pos_dst, const float* pos_src, const int32_t* col,
                      const int32_t* items, int64_t n_items, const int32_t* split,
                      int64_t n_split, int64_t n_parts, void* out, int64_t ldo,
                      float* pos_out, float* part_out, float* part_small, int64_t num_nodes,
                      int32_t H, int32_t dtype, mpnn_stream_t stream);
int32_t mpnn_egnn_bwd_dst(const void* A, int64_t ldA, const void* B, int64_t ldB,
                          const void* G, int64_t ldG, const float* gpos_out,
                          const void* ea_sorted, int32_t Fe, const void* We, const void* w_pos,
                          const float* pos_dst, const float* pos_src, const int32_t* col,
                          const int32_t* items, int64_t n_items, const int32_t* split,
                          int64_t n_split, int64_t n_parts, void* gA, int64_t ldgA, float* S,
                          void* g_wpos, float* part_gA, float* part_small, float* scratch,
                          int64_t num_nodes, int32_t H, int32_t dtype, mpnn_stream_t stream);
int32_t mpnn_egnn_bwd_src(const void* A, int64_t ldA, const void* B, int64_t ldB,
                          const void* G, int64_t ldG, const float* gpos_out,
                          const void* ea_sorted_t, int32_t Fe, const void* We, const void* w_pos,
                          const float* pos_dst, const float* pos_src, const int32_t* col_t,
                          const int32_t* items_t, int64_t n_items_t, const int32_t* split_t,
                          int64_t n_split_t, int64_t n_parts_t, void* gB, int64_t ldgB,
                          float* gpos_src, void* gWe, float* part_gB, float* part_small,
                          float* scratch, int64_t num_src_nodes, int32_t H, int32_t dtype,
                          mpnn_stream_t stream);

/* ---------------------------------------------------------------------------
 * Dense fp32 GEMMs of the layer on the matrix cores.
 *
 * Replace the node-level Linear calls of the reference and their autograd backward:
 *   models/layers.py:104 fused_linear(x); models/layers.py:56 linear(...) after the column split;
 *   models/layers.py:69,140 relu(linear_out(cat[x, aggr])) as  x W_x^T + aggr W_a^T  without the cat.
 *
 * mpnn_gemm_nt_f32:  Y[N,M] = act( X1[N,K1] W1[M,K1]^T + X2[N,K2] W2[M,K2]^T ),  act = relu if `relu` != 0.
 *   The second term is skipped when K2 == 0.  maskI (optional, same shape as XI): XI is read as
 *   XI * (maskI > 0) -- the relu backward fused into the operand load.  With a pre-transposed weight the
 *   same entry point computes input gradients (dX = G W  ==  G (W^T)^T).
 *   Y2 (optional, with emul [N,M]): second output Y2 = Y * emul written by the same epilogue
 *   (d aggr and d A = d aggr * cnt leave one GEMM together).  emul_kind: MPNN_F32 (float) or MPNN_U8 (one
 *   byte per entry, the counts of mpnn_aggregate_fwd); ldemul in elements of that kind.
 *   addend (optional, [N,M], not together with Y2): Y = act(products + addend).  With it the update of
 *   models/layers.py:69,140 becomes relu(P + aggr W_a^T), where P = x W_x^T left the projection's GEMM as extra
 *   output columns (x [W_n ; W_x]^T): x is read once per forward instead of twice.
 *   Arithmetic: by default every float32 operand is split into three bfloat16 pieces and the products run on
 *   the bfloat16 matrix cores with float32 accumulation (six piece products per product; error against a
 *   float64 product no larger than the float32 matrix cores'); MPNN_GEMM=f32 in the environment selects the
 *   float32 matrix-core instructions, which unaligned operands (K % 4 != 0) always take.
 *   scratch (optional, 16-byte aligned, at least mpnn_gemm_nt_scratch_bytes(M, K1, K2) bytes): with it the
 *   weights are split once per call into chunk images that the kernel copies straight into LDS, instead of
 *   being split again by every block for every 128-row step.
 * mpnn_gemm_tn_f32:  dW[M,K] = (G * (mask > 0))[N,M]^T X[N,K]  (weight gradient; rows are summed in a fixed
 *   order through `scratch`, float, at least mpnn_gemm_tn_scratch_floats(N, M, K) elements).
 *   X2 (optional, [N,K2], K2 = 0 without it): dW[M, K+K2] = (G * (mask > 0))^T [X | X2] in one pass over G
 *   (the gradient of linear_out over its two inputs x and aggr); scratch is sized with K + K2.
 *   Gout (optional, [N,M], needs 16-byte aligned operands): G * (mask > 0) is also written out, so that the
 *   other GEMMs of the backward read one matrix instead of two.  Runs on the float32 matrix-core instructions.
 * All matrices row-major with leading dimensions in elements.
 * ------------------------------------------------------------------------- */
int32_t mpnn_gemm_nt_f32(const float* X1, int64_t ldx1, const float* mask1, int64_t ldm1,
                         const float* W1, int64_t ldw1, int32_t K1, const float* X2, int64_t ldx2,
                         const float* mask2, int64_t ldm2, const float* W2, int64_t ldw2, int32_t K2,
                         float* Y, int64_t ldy, const void* emul, int32_t emul_kind, int64_t ldemul,
                         float* Y2, int64_t ldy2, const float* addend, int64_t ldadd, int64_t N, int32_t M,
                         int32_t relu, void* scratch, mpnn_stream_t stream);
int64_t mpnn_gemm_nt_scratch_bytes(int32_t M, int32_t K1, int32_t K2);
int64_t mpnn_gemm_tn_scratch_floats(int64_t N, int32_t M, int32_t K);
int32_t mpnn_gemm_tn_f32(const float* G, int64_t ldg, const float* mask, int64_t ldm,
                         const float* X, int64_t ldx, const float* X2, int64_t ldx2, float* dW,
                         int64_t lddw, float* scratch, float* Gout, int64_t ldgo, int64_t N, int32_t M,
                         int32_t K, int32_t K2, mpnn_stream_t stream);

/* bfloat16 form of mpnn_gemm_nt_f32 for bfloat16 models (bf16 storage, float32 accumulation, no operand mask):
 *   Y[N,M] = act( X1 W1^T + X2 W2^T ),  optionally  Y2 = Y * emul  with emul [N,M] float32 or bytes (emul_kind).
 * M, K1, K2 multiples of 8; X, W, emul 16-byte aligned, Y and Y2 8-byte aligned, with row strides that keep
 * every row so. */
int32_t mpnn_gemm_nt_bf16(const void* X1, int64_t ldx1, const void* W1, int64_t ldw1, int32_t K1,
                          const void* X2, int64_t ldx2, const void* W2, int64_t ldw2, int32_t K2, void* Y,
                          int64_t ldy, const void* emul, int32_t emul_kind, int64_t ldemul, void* Y2,
                          int64_t ldy2, int64_t N, int32_t M, int32_t relu, mpnn_stream_t stream);
/* bfloat16 form of mpnn_gemm_tn_f32 (the weight gradients of the same Linear calls): G, mask, X, Gout and dW
 * bfloat16, float32 accumulation over the rows in a fixed order, one rounding to bfloat16 at the end.
 * X2 / K2 as in mpnn_gemm_tn_f32.  `scratch`: at least mpnn_gemm_tn_bf16_scratch_floats(N, M, K + K2) floats.  M and K multiples of 8; G, mask, X, Gout
 * 16-byte aligned with row strides that are multiples of 8 elements. */
int64_t mpnn_gemm_tn_bf16_scratch_floats(int64_t N, int32_t M, int32_t K);
int32_t mpnn_gemm_tn_bf16(const void* G, int64_t ldg, const void* mask, int64_t ldm, const void* X,
                          int64_t ldx, const void* X2, int64_t ldx2, void* dW, int64_t lddw, float* scratch,
                          void* Gout, int64_t ldgo, int64_t N, int32_t M, int32_t K, int32_t K2,
                          mpnn_stream_t stream);

/* ---------------------------------------------------------------------------
 * Rows exchanged between the ranks of a partitioned layer (no counterpart in the reference, which runs on one
 * device; the exchange stands in for PyG's gathers at models/layers.py:38,110 when source rows live on other ranks).
 *
 * mpnn_halo_pack:    send[i, :] = table[index[i], :]   (rows this rank serves, in the order of the peers' lists),
 *                    converted between float32 and bfloat16 on the way when the dtypes differ (the wire format).
 * mpnn_halo_unpack:  out[r, :] = sum over k in [rowptr[r], rowptr[r+1]) of recv[pos[k], :]: the contributions
 *                    received for this rank's rows, added in list order (float32 sums, no atomics: one fixed
 *                    order); rows without contributions are written as zeros.  recv rows are H wide, contiguous.
 * ------------------------------------------------------------------------- */
int32_t mpnn_halo_pack(const void* table, int64_t ldt, int32_t table_dtype, const int32_t* index, void* send,
                       int32_t send_dtype, int64_t rows, int32_t H, mpnn_stream_t stream);
int32_t mpnn_halo_unpack(const void* recv, int32_t recv_dtype, const int32_t* rowptr, const int32_t* pos,
                         void* out, int64_t ldo, int32_t out_dtype, int64_t rows, int32_t H,
                         mpnn_stream_t stream);

/* ---------------------------------------------------------------------------
 * Segment mean over an index list: out[g, :] = (1/max(cnt_g,1)) * sum_{p in seg g} x[col[p], :]
 * and its backward gx[col[p], :] = gout[g, :] / max(cnt_g, 1).
 *
 * Replaces global_mean_pool(x, batch) at models/mpnn.py:54,150.  The CSR is built
 * with mpnn_csr_build(key = batch, other = arange(N)), so unsorted `batch` works too.
 *
 * num_averaged <= num_segments: the segments from num_averaged on are padding (the spare
 * graph that holds the unused rows of a fixed-capacity batch, mpnn_collate): their output
 * rows are zeros, their rows are not read and take a zero gradient.  Pass num_segments
 * for the plain mean of every segment.
 * ------------------------------------------------------------------------- */
int32_t mpnn_segment_mean_fwd(const void* x, int64_t ldx, const int32_t* rowptr,
                              const int32_t* col, void* out, int64_t ldo,
                              int64_t num_segments, int64_t num_averaged, int32_t H,
                              int32_t dtype, mpnn_stream_t stream);
int32_t mpnn_segment_mean_bwd(const void* gout, int64_t ldg, const int32_t* rowptr,
                              const int32_t* col, void* gx, int64_t ldgx,
                              int64_t num_segments, int64_t num_averaged, int32_t H,
                              int32_t dtype, mpnn_stream_t stream);

/* ---------------------------------------------------------------------------
 * Loss and optimizer step (each a fixed sequence of launches, nothing read back to the host,
 * so a whole training step can be captured into a HIP graph).
 *
 * mpnn_masked_ce: mean cross-entropy over the rows with mask != 0 (mask == NULL: every row), what the
 * reference computes as F.cross_entropy(out[data.train_mask], data.y[data.train_mask])
 * (training/train.py:47 with experiments/flickr_exp.py:31).
 *   logits [n, c] float, labels [n] int64, mask [n] bytes (torch.bool storage); a label of -100 (F.cross_entropy's
 *   default ignore_index) leaves its row out of the sum, the count and the gradient; any other label outside
 *   [0, c), which torch rejects, makes the loss NaN
 *   grad   [n, c]  softmax - onehot on masked rows, 0 elsewhere - NOT yet divided by the row count
 *   result [2]     { loss, 1 / number of masked rows }   (the gradient of the loss is grad * result[1])
 *   scratch        mpnn_masked_ce_scratch_floats() floats
 *
 * mpnn_adam_step: one step of torch.optim.Adam(lr, betas, eps) without weight decay or amsgrad
 * (experiments/flickr_exp.py:26, qm9_exp.py:33, modelnet_exp.py:49) over one flat float buffer.
 *   state [3] (device): state[0] holds the number of steps taken so far (0 before the first call) and is
 *   incremented by the call; state[1..2] are overwritten with the bias-corrected step size and sqrt(1-beta2^t).
 * ------------------------------------------------------------------------- */
int64_t mpnn_masked_ce_scratch_floats(void);
int32_t mpnn_masked_ce(const float* logits, int64_t ld, const int64_t* labels, const uint8_t* mask, int64_t n,
                       int32_t c, float* grad, int64_t ldg, float* result, float* scratch,
                       mpnn_stream_t stream);
int32_t mpnn_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                       float* state, float lr, float beta1, float beta2, float eps, mpnn_stream_t stream);

/* ---------------------------------------------------------------------------
 * k-nearest-neighbour graph of a batch of equally sized point clouds + normalised distance feature.
 *
 * Stands where the reference derives point-cloud edges and the T.Distance(norm=True) feature on the CPU inside
 * its dataset transforms (experiments/modelnet_exp.py:58, data/dataset.py:193).
 *   pos        [num_clouds * points, 3] float, clouds stored one after the other
 *   edge_index [2][num_clouds * points * k] int64: edges j -> i for the k nearest j of each point i of the
 *              same cloud, grouped by i, ascending distance, ties by lower j
 *   edge_attr  [num_clouds * points * k] float: distance / largest edge distance of the cloud
 *   workspace  mpnn_knn_workspace_floats(...) floats
 * Limits: 1 <= k <= 32, k < points <= 8192.
 * ------------------------------------------------------------------------- */
int64_t mpnn_knn_workspace_floats(int32_t num_clouds, int32_t points, int32_t k);
int32_t mpnn_knn_graph(const float* pos, int32_t num_clouds, int32_t points, int32_t k, int64_t* edge_index,
                       float* edge_attr, float* workspace, mpnn_stream_t stream);

/* ---------------------------------------------------------------------------
 * Whole layers for batches of small graphs, one wavefront per graph.
 *
 * The mini-batches of the molecule workloads (DataLoader batches of data/dataset.py:94-96, consumed by
 * training/train.py:94-98) are disjoint unions of graphs of a few dozen nodes.  An ISLAND is a run of consecutive
 * node ids that no edge leaves; a layer (models/layers.py:91-140, both forms after the column split of
 * `linear.weight`) never moves data between islands, so one wavefront can run the whole layer for one island out
 * of LDS: projection, messages, sum per destination, update - and in the backward every gradient - in ONE launch
 * each way (plus a fixed-order sum of per-block weight-gradient partials), instead of a dozen launches whose
 * working sets are a few kilobytes.  Results are deterministic; they agree with the general path to rounding.
 *
 * mpnn_islands_build finds the islands of a destination-sorted CSR (mpnn_csr_build output in row-id order):
 *   island_ptr [num_nodes + 1]: island i = nodes island_ptr[i] .. island_ptr[i+1]-1 (ascending, island_ptr[0] = 0)
 *   meta [4]: { number of islands, nodes of the largest island, edges of the largest island, gave up }
 *   give_up > 0: stop as soon as one row alone spans that many nodes or more (its island is at least that large -
 *   a k-NN cloud, a citation graph); meta then reads { 0, 0, 0, 1 } and island_ptr is unspecified.  0 = always finish.
 *   workspace: mpnn_islands_workspace_bytes(num_nodes) bytes.
 * A batch assembled by mpnn_collate can use its graph offsets (slot table row 0) as island_ptr directly.
 *
 * mpnn_island_layer_fwd / _bwd run MPNNLayerSum.forward and its autograd backward (pos == NULL), or the EGNN
 * layer (messages as above plus pos'[n] = pos[n] + sum_{e -> n} (pos[n] - pos[src]) * (w_pos . m_e)) when pos is
 * given.  Limits: H = 64 or 128, 1 <= Fe <= 8, every island at most 32 nodes (the height of a matrix-core tile;
 * `rows` must be 32), and a tile must fit the LDS (mpnn_island_layer_waves(...) >= 1).  Rows from
 * island_ptr[n_islands] up to num_nodes must have no edges (the padding rows of a fixed-capacity batch); they are
 * processed 32 at a time by `tail_units` further wavefronts (tail_units * 32 >= the number of such rows; 0 when
 * there are none).
 * The backward recomputes the forward intermediates from x, so nothing but y needs to be kept.
 *   rows       32
 *   x [N, F] (ldx), ea [E, Fe] in CSR order, w_nodes [2H, F], w_edge [H, Fe], w_out [H, F + H], w_pos [H]: `dtype`
 *   pos, pos_out, gpos_out (gradient arriving at pos'), gpos (gradient of pos): float [N, 3]
 *   y [N, H] (ldy): output of the forward, input of the backward
 *   gy [N, H] (ldgy); gx [N, F] (ldgx) or NULL when the gradient of x is not wanted
 *   g_wn [2H, F], g_we [H, Fe], g_wout [H, F + H], g_wpos [H]: `dtype`, contiguous, each may be NULL
 *   partial    mpnn_island_layer_partial_floats(...) floats of scratch (backward)
 *   gx_part    float [N, F] scratch (backward, only when gx is given)
 * ------------------------------------------------------------------------- */
typedef struct mpnn_island_layer {
  const int32_t* rowptr; const int32_t* col; const int32_t* island_ptr;
  int32_t n_islands; int32_t tail_units; int32_t rows; int64_t num_nodes;
  const void* x; int64_t ldx;
  const void* ea; const void* w_nodes; const void* w_edge; const void* w_out;
  const float* pos; const void* w_pos;
  int32_t F; int32_t H; int32_t Fe; int32_t dtype;
  void* y; int64_t ldy; float* pos_out;
  const void* gy; int64_t ldgy; const float* gpos_out;
  void* gx; int64_t ldgx; float* gpos;
  void* g_wn; void* g_we; void* g_wout; void* g_wpos;
  float* partial; float* gx_part;
} mpnn_island_layer;
int64_t mpnn_islands_workspace_bytes(int64_t num_nodes);
int32_t mpnn_islands_build(const int32_t* rowptr, const int32_t* col, int64_t num_nodes, int32_t give_up,
                           int32_t* island_ptr, int32_t* meta, void* workspace, int64_t workspace_bytes,
                           mpnn_stream_t stream);
/* wavefronts (= islands) per block that fit the LDS for this shape; 0 = the shape is not supported */
int32_t mpnn_island_layer_waves(int32_t F, int32_t H, int32_t Fe, int32_t egnn, int32_t backward);
int64_t mpnn_island_layer_partial_floats(int32_t n_islands, int32_t tail_units, int32_t F, int32_t H, int32_t Fe,
                                         int32_t egnn);
int32_t mpnn_island_layer_fwd(const mpnn_island_layer* desc, mpnn_stream_t stream);
int32_t mpnn_island_layer_bwd(const mpnn_island_layer* desc, mpnn_stream_t stream);

/* ---------------------------------------------------------------------------
 * Whole layers for batches of point clouds: islands of up to 2048 nodes whose rows of x are at most 4 wide.
 *
 * The k-NN batches of the point-cloud workload (experiments/modelnet_exp.py:58, `data.pos` as x) are islands of a
 * thousand nodes with 3 input features.  The x of a whole island then fits the LDS, and the projection rows of an
 * edge's two ends (models/layers.py:104-105) are recomputed from it per edge instead of being fetched: the forward
 * of MPNNLayerSum (or of the EGNN layer) is ONE launch (projection, messages, sums, update), the backward three (destination tiles,
 * source rows, fixed-order sum of the per-block weight-gradient partials).  Results are deterministic and agree
 * with the general path to rounding.
 *
 * The descriptor is the one of the island layers with rows = 32, tail_units = 0 (the islands cover every node).
 * With pos (and w_pos, pos_out; gpos_out, gpos, g_wpos in the backward) the calls run the EGNN layer, as the island
 * entry points do.  Limits: H = 64, 1 <= F <= 4, 1 <= Fe <= 8, max_nodes (a bound on the largest island) <= 2048
 * (<= 1024 with coordinates: their tables share the LDS); mpnn_region_layer_fits(...) tells.  The backward also reads
 * the source-sorted view (rowptr_t [N + 1], col_t [E]: destinations of the edges grouped by source; ea_t [E, Fe] in
 * that order, `dtype`) and needs
 *   partial   mpnn_region_layer_partial_floats(...) floats of scratch
 *   gx_part   float [N, F] scratch (only when gx is given)
 *   ga        float [N, H] scratch, N more floats behind it when pos is given
 *   aggr      float [N, H]: the sums over incoming edges, written by the forward when given and handed to the
 *             backward, which otherwise (NULL) walks the edges once more to recompute them
 * Without gx and without gpos (both NULL: a model's first layer, whose inputs want no gradient) the backward is two
 * launches: the pass over the source rows is not needed, the destination pass collects the whole of g_wn (when
 * Fe + 3 F (+ 1 with coordinates) <= 17).  gpos alone may be NULL as well (it is then not written).
 * ------------------------------------------------------------------------- */
int32_t mpnn_region_layer_fits(int32_t F, int32_t H, int32_t Fe, int32_t max_nodes, int32_t egnn);
int64_t mpnn_region_layer_partial_floats(int32_t n_islands, int32_t max_nodes, int32_t F, int32_t H, int32_t Fe,
                                         int32_t egnn);
int32_t mpnn_region_layer_fwd(const mpnn_island_layer* desc, int32_t max_nodes, float* aggr, mpnn_stream_t stream);
int32_t mpnn_region_layer_bwd(const mpnn_island_layer* desc, int32_t max_nodes, const int32_t* rowptr_t,
                              const int32_t* col_t, const void* ea_t, const float* aggr, float* ga,
                              mpnn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MPNN_AMD_H */
