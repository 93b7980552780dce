// Graph structure for the message-passing engine: stable sort of the edge list into a CSR
// (and, with the roles of the two endpoint arrays swapped, its transpose), and the split of
// CSR rows into bounded work items.  Everything here is integer arithmetic and must be
// bit-exact against a stable host sort.
//
// Replaces the implicit index handling of PyG's MessagePassing.propagate() used by the
// reference at models/layers.py:38 and models/layers.py:110.
#include <cstring>
#include <cstdlib>
#include <rocprim/rocprim.hpp>

#include "common.h"

namespace mpnn {

char* error_buffer() {
  static thread_local char buf[512] = "";
  return buf;
}

// ---------------------------------------------------------------------------------------
// CSR build
// ---------------------------------------------------------------------------------------
// keys32[e] = key of edge e (clamped into range), payload[e] = (other of edge e, clamped) << 32 | e: the neighbour
// id travels through the sort beside the edge id, so no random gather by edge id follows the sort.
__global__ void k_prepare_keys(const int64_t* __restrict__ key, const int64_t* __restrict__ other,
                               int64_t E, int64_t N, int64_t M, int32_t* __restrict__ keys32,
                               uint64_t* __restrict__ payload, int32_t* __restrict__ status) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  bool bad = false;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += stride) {
    int64_t k = key[e];
    int64_t o = other[e];
    if (k < 0 || k >= N || o < 0 || o >= M) bad = true;
    k = k < 0 ? 0 : (k >= N ? N - 1 : k);
    o = o < 0 ? 0 : (o >= M ? M - 1 : o);
    keys32[e] = (int32_t)k;
    payload[e] = ((uint64_t)(uint32_t)o << 32) | (uint32_t)e;
  }
  if (bad) status[0] = 1;  // every writer stores the same value
}

// After the sort, one thread per position p in [0, E]:
//   col[p], eid[p] = the two halves of the payload;
//   rowptr[k] = first sorted position whose key is >= k: the thread owns the key values in (key[p-1], key[p]], so
//   every rowptr entry has exactly one writer.
__global__ void k_unpack_sorted(const int32_t* __restrict__ sorted, const uint64_t* __restrict__ payload, int64_t E,
                                int64_t N, int32_t* __restrict__ rowptr, int32_t* __restrict__ col,
                                int32_t* __restrict__ eid) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p <= E; p += stride) {
    int64_t prev = p > 0 ? sorted[p - 1] : -1;
    int64_t cur = p < E ? sorted[p] : N;
    for (int64_t k = prev + 1; k <= cur; ++k) rowptr[k] = (int32_t)p;
    if (p < E) {
      const uint64_t v = payload[p];
      col[p] = (int32_t)(v >> 32);
      eid[p] = (int32_t)(uint32_t)v;
    }
  }
}

// rocPRIM sorts up to 2^20 items by merge sort: some twenty launches of a few microseconds each for the edge list of
// a mini-batch, most of what a per-batch graph build costs.  The keys here are narrow (node ids, degree classes), so
// the one-sweep radix sort needs one or two digit passes; take it from MPNN_SORT_MERGE_LIMIT items on.
#ifndef MPNN_SORT_MERGE_LIMIT
#define MPNN_SORT_MERGE_LIMIT 65536
#endif
using SortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config,
                                              MPNN_SORT_MERGE_LIMIT>;

static unsigned key_bits(int64_t N) {
  unsigned b = 1;
  while (b < 31 && ((int64_t)1 << b) < N) ++b;
  return b;
}

static hipError_t sort_temp_bytes(int64_t E, int64_t N, size_t& bytes) {
  bytes = 0;
  int32_t* kn = nullptr;
  uint64_t* vn = nullptr;
  return rocprim::radix_sort_pairs<SortConfig>(nullptr, bytes, kn, kn, vn, vn, (size_t)E, 0u, key_bits(N), (hipStream_t)0);
}

static int grid_for(int64_t n) {
  int64_t g = (n + kBlock - 1) / kBlock;
  if (g < 1) g = 1;
  if (g > kMaxBlocks) g = kMaxBlocks;
  return (int)g;
}

// ---------------------------------------------------------------------------------------
// Work items
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int piece_size(int deg, int chunk, int max_parts) {
  int s = (deg + max_parts - 1) / max_parts;
  return s > chunk ? s : chunk;
}

// counts[0][n] = partial slots of row n (0 if unsplit), counts[1][n] = 1 if row n is split.
// Entry n == N is a zero terminator so that an exclusive scan over N+1 entries ends in the total.
// key[n] orders the rows for the item list: split rows first (key 0, row order), then unsplit rows by
// DESCENDING degree (key = 1 + chunk - deg), row order among equal degrees (the sort is stable).
__global__ void k_item_counts(const int32_t* __restrict__ rowptr, int64_t N, int chunk, int max_parts,
                              int32_t* __restrict__ c_parts, int32_t* __restrict__ c_split,
                              int32_t* __restrict__ key) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n <= N; n += stride) {
    int parts = 0, split = 0;
    if (n < N) {
      int deg = rowptr[n + 1] - rowptr[n];
      if (deg > chunk) {
        int ps = piece_size(deg, chunk, max_parts);
        parts = (deg + ps - 1) / ps;
        split = 1;
      }
      key[n] = split ? 0 : 1 + chunk - deg;
    }
    c_parts[n] = parts;
    c_split[n] = split;
  }
}

__global__ void k_item_fill_rows(const int32_t* __restrict__ rowptr, int64_t N, int chunk, int max_parts,
                                 const int32_t* __restrict__ o_parts, const int32_t* __restrict__ o_split,
                                 const int32_t* __restrict__ order, int4* __restrict__ items, int64_t items_cap,
                                 int4* __restrict__ split, int64_t split_cap, int32_t* __restrict__ meta) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int n_parts = o_parts[N];
  const int n_split = o_split[N];
  bool overflow = false;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < N; j += stride) {
    const int n = order[j];  // j-th row in (split first, then descending degree) order
    int r0 = rowptr[n], r1 = rowptr[n + 1];
    int deg = r1 - r0;
    if (deg <= chunk) {
      int64_t idx = (int64_t)n_parts + j - n_split;
      if (idx < items_cap) items[idx] = make_int4(n, r0, r1, -1);
      else overflow = true;
    } else {
      int ps = piece_size(deg, chunk, max_parts);
      int pieces = (deg + ps - 1) / ps;
      int64_t idx = o_split[n];  // == j: split rows keep row order at the front
      if (idx < split_cap) split[idx] = make_int4(n, o_parts[n], pieces, o_parts[n]);
      else overflow = true;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    meta[0] = n_parts + (int)N - n_split;
    meta[1] = n_parts;
    meta[2] = n_split;
  }
  if (overflow) meta[3] = 1;
}

// One wave per split row writes that row's pieces.
__global__ void k_item_fill_pieces(const int32_t* __restrict__ rowptr, int64_t N, int chunk, int max_parts,
                                   const int32_t* __restrict__ o_split, const int4* __restrict__ split,
                                   int64_t split_cap, int4* __restrict__ items, int64_t items_cap,
                                   int32_t* __restrict__ meta) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  int64_t n_split = o_split[N];
  if (n_split > split_cap) n_split = split_cap;
  bool overflow = false;
  for (int64_t s = wave; s < n_split; s += nwaves) {
    int4 sr = split[s];
    int row = sr.x, poff = sr.y, pieces = sr.z;
    int r0 = rowptr[row], r1 = rowptr[row + 1];
    int ps = piece_size(r1 - r0, chunk, max_parts);
    for (int c = lane; c < pieces; c += 64) {
      int e0 = r0 + c * ps;
      int e1 = e0 + ps < r1 ? e0 + ps : r1;
      int64_t idx = (int64_t)poff + c;
      if (idx < items_cap) items[idx] = make_int4(row, e0, e1, (int)idx);
      else overflow = true;
    }
  }
  if (overflow) meta[3] = 1;
}

static unsigned order_bits(int chunk) {
  unsigned b = 1;
  while (b < 31 && (1 << b) < chunk + 2) ++b;
  return b;
}

static hipError_t order_temp_bytes(int64_t N, int chunk, size_t& bytes) {
  bytes = 0;
  int32_t* kn = nullptr;
  return rocprim::radix_sort_pairs<SortConfig>(nullptr, bytes, kn, kn, rocprim::counting_iterator<int32_t>(0), kn, (size_t)N, 0u,
                                   order_bits(chunk), (hipStream_t)0);
}

static hipError_t scan_temp_bytes(int64_t n, size_t& bytes) {
  bytes = 0;
  int32_t* p = nullptr;
  return rocprim::exclusive_scan(nullptr, bytes, p, p, 0, (size_t)n, rocprim::plus<int32_t>(), (hipStream_t)0);
}

// ---------------------------------------------------------------------------------------
// Rows of a sorted view moved into work-item order
// ---------------------------------------------------------------------------------------
// key[n] as in k_item_counts (split rows 0, then 1 + chunk - deg); deg[n].
__global__ void k_row_order_keys(const int32_t* __restrict__ rowptr, int64_t N, int chunk, int32_t* __restrict__ key) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += stride) {
    const int deg = rowptr[n + 1] - rowptr[n];
    key[n] = deg > chunk ? 0 : 1 + chunk - deg;
  }
}

// deg_o[j] = degree of the row stored at position j (entry N = 0: the terminator of the exclusive scan)
__global__ void k_degrees_in_order(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ order, int64_t N,
                                   int32_t* __restrict__ deg_o) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j <= N; j += stride)
    deg_o[j] = j < N ? rowptr[order[j] + 1] - rowptr[order[j]] : 0;
}

constexpr int kPermTile = 2048;   // output positions per block

// Output position p belongs to the stored row j with newptr[j] <= p < newptr[j+1]; its entry comes from
// position rowptr[order[j]] + (p - newptr[j]) of the old arrays.  A block finds the rows its tile of positions
// touches once, then every thread searches inside that small window: balanced whatever the degrees are.
__global__ void __launch_bounds__(kBlock) k_permute_segments(const int32_t* __restrict__ rowptr,
                                                             const int32_t* __restrict__ order,
                                                             const int32_t* __restrict__ newptr, int64_t N, int64_t E,
                                                             const int32_t* __restrict__ col,
                                                             const int32_t* __restrict__ eid,
                                                             int32_t* __restrict__ col_out,
                                                             int32_t* __restrict__ eid_out) {
  __shared__ int64_t win[2];
  const int64_t tiles = (E + kPermTile - 1) / kPermTile;
  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int64_t p0 = t * kPermTile;
    const int64_t p1 = p0 + kPermTile < E ? p0 + kPermTile : E;
    if (threadIdx.x < 2) {
      // last row j with newptr[j] <= q, for q = p0 (thread 0) and q = p1 - 1 (thread 1)
      const int64_t q = threadIdx.x == 0 ? p0 : p1 - 1;
      int64_t lo = 0, hi = N;      // invariant: newptr[lo] <= q < newptr[hi]  (newptr[N] = E > q)
      while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (newptr[mid] <= q) lo = mid; else hi = mid;
      }
      win[threadIdx.x] = lo;
    }
    __syncthreads();
    const int64_t j0 = win[0], j1 = win[1];
    for (int64_t p = p0 + threadIdx.x; p < p1; p += kBlock) {
      int64_t lo = j0, hi = j1 + 1;
      while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (newptr[mid] <= p) lo = mid; else hi = mid;
      }
      const int64_t src = (int64_t)rowptr[order[lo]] + (p - newptr[lo]);
      col_out[p] = col[src];
      eid_out[p] = eid[src];
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------
// Row gather
// ---------------------------------------------------------------------------------------
template <typename W>
__global__ void k_gather_rows(const W* __restrict__ src, const int32_t* __restrict__ index,
                              W* __restrict__ dst, int64_t rows, int words) {
  const int64_t total = rows * words;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    int64_t r = i / words;
    int w = (int)(i - r * words);
    dst[i] = src[(int64_t)index[r] * words + w];
  }
}

}  // namespace mpnn

using namespace mpnn;

extern "C" {

int32_t mpnn_abi_version(void) { return MPNN_ABI_VERSION; }
const char* mpnn_last_error(void) { return error_buffer(); }

int64_t mpnn_csr_workspace_bytes(int64_t N, int64_t E) {
  if (N < 0 || E < 0 || N >= INT32_MAX || E >= INT32_MAX) return -1;
  size_t temp = 0;
  if (sort_temp_bytes(E, N, temp) != hipSuccess) return -1;
  // keys in / out, payload in / out, rocPRIM temporary storage (each rounded up to 256 B)
  return 2 * align_up((E + 1) * 4, 256) + 2 * align_up((E + 1) * 8, 256) + align_up((int64_t)temp + 256, 256);
}

int32_t mpnn_csr_build(const int64_t* key, const int64_t* other, int64_t E, int64_t N, int64_t M, int32_t* rowptr,
                       int32_t* col, int32_t* eid, int32_t* status, void* workspace, int64_t workspace_bytes,
                       mpnn_stream_t stream) {
  if (N < 0 || M < 0 || E < 0 || N >= INT32_MAX || M >= INT32_MAX || E >= INT32_MAX)
    return fail(MPNN_E_INVALID, "mpnn_csr_build: num_keys=%lld num_other=%lld E=%lld outside [0, 2^31)", (long long)N,
                (long long)M, (long long)E);
  if (!rowptr || !status || (E > 0 && (!key || !other || !col || !eid)))
    return fail(MPNN_E_INVALID, "mpnn_csr_build: null pointer argument");
  if (E > 0 && (N == 0 || M == 0)) return fail(MPNN_E_INVALID, "mpnn_csr_build: edges given for a graph with no nodes");
  MPNN_HIP_CHECK(hipMemsetAsync(status, 0, sizeof(int32_t), stream));
  if (E == 0) {
    MPNN_HIP_CHECK(hipMemsetAsync(rowptr, 0, (size_t)(N + 1) * 4, stream));
    return MPNN_OK;
  }
  if (!workspace) return fail(MPNN_E_WORKSPACE, "mpnn_csr_build: workspace is null");
  Arena arena(workspace, workspace_bytes);
  int32_t* keys_in = arena.take<int32_t>(E + 1);
  int32_t* keys_out = arena.take<int32_t>(E + 1);
  uint64_t* pay_in = arena.take<uint64_t>(E + 1);
  uint64_t* pay_out = arena.take<uint64_t>(E + 1);
  size_t temp = 0;
  MPNN_HIP_CHECK(sort_temp_bytes(E, N, temp));
  char* temp_ptr = arena.take<char>((int64_t)temp + 256);
  if (!keys_in || !keys_out || !pay_in || !pay_out || !temp_ptr)
    return fail(MPNN_E_WORKSPACE, "mpnn_csr_build: workspace of %lld bytes is too small (need %lld)",
                (long long)workspace_bytes, (long long)mpnn_csr_workspace_bytes(N, E));

  k_prepare_keys<<<grid_for(E), kBlock, 0, stream>>>(key, other, E, N, M, keys_in, pay_in, status);
  MPNN_LAUNCH_CHECK("k_prepare_keys");
  // Radix sort is stable, and the payload's low half is the original edge id, so equal keys stay in edge order.
  MPNN_HIP_CHECK(rocprim::radix_sort_pairs<SortConfig>(temp_ptr, temp, keys_in, keys_out, pay_in, pay_out, (size_t)E, 0u,
                                           key_bits(N), stream));
  k_unpack_sorted<<<grid_for(E + 1), kBlock, 0, stream>>>(keys_out, pay_out, E, N, rowptr, col, eid);
  MPNN_LAUNCH_CHECK("k_unpack_sorted");
  return MPNN_OK;
}

// The radix sort's temporary storage is sized for the widest key the rule can produce (chunk < 2^30).
static int64_t items_temp_bytes(int64_t N) {
  size_t t1 = 0, t2 = 0;
  if (scan_temp_bytes(N + 1, t1) != hipSuccess) return -1;
  if (order_temp_bytes(N, (1 << 30) - 2, t2) != hipSuccess) return -1;
  return (int64_t)(t1 > t2 ? t1 : t2);
}

int64_t mpnn_items_workspace_bytes(int64_t N) {
  if (N < 0 || N >= INT32_MAX) return -1;
  const int64_t temp = items_temp_bytes(N);
  if (temp < 0) return -1;
  return 7 * align_up((N + 1) * 4, 256) + align_up(temp + 256, 256);
}

int32_t mpnn_items_build(const int32_t* rowptr, int64_t N, int32_t chunk, int32_t max_parts, int32_t* items,
                         int64_t items_capacity, int32_t* split, int64_t split_capacity, int32_t* meta,
                         void* workspace, int64_t workspace_bytes, mpnn_stream_t stream) {
  if (N < 0 || N >= INT32_MAX) return fail(MPNN_E_INVALID, "mpnn_items_build: N=%lld out of range", (long long)N);
  if (chunk < 1 || max_parts < 2)
    return fail(MPNN_E_INVALID, "mpnn_items_build: chunk=%d (>=1) max_parts=%d (>=2)", chunk, max_parts);
  if (!rowptr || !meta || !workspace || (N > 0 && !items) || (split_capacity > 0 && !split))
    return fail(MPNN_E_INVALID, "mpnn_items_build: null pointer argument");
  if (((uintptr_t)items & 15) || ((uintptr_t)split & 15))
    return fail(MPNN_E_INVALID, "mpnn_items_build: items/split must be 16-byte aligned");
  Arena arena(workspace, workspace_bytes);
  int32_t* c_parts = arena.take<int32_t>(N + 1);
  int32_t* c_split = arena.take<int32_t>(N + 1);
  int32_t* o_parts = arena.take<int32_t>(N + 1);
  int32_t* o_split = arena.take<int32_t>(N + 1);
  int32_t* key = arena.take<int32_t>(N + 1);
  int32_t* key_sorted = arena.take<int32_t>(N + 1);
  int32_t* order = arena.take<int32_t>(N + 1);
  const int64_t temp_need = items_temp_bytes(N);
  if (temp_need < 0) return fail(MPNN_E_HIP, "mpnn_items_build: rocPRIM size query failed");
  size_t temp = (size_t)temp_need;
  char* temp_ptr = arena.take<char>((int64_t)temp + 256);
  if (!c_parts || !c_split || !o_parts || !o_split || !key || !key_sorted || !order || !temp_ptr)
    return fail(MPNN_E_WORKSPACE, "mpnn_items_build: workspace of %lld bytes is too small (need %lld)",
                (long long)workspace_bytes, (long long)mpnn_items_workspace_bytes(N));
  MPNN_HIP_CHECK(hipMemsetAsync(meta, 0, 8 * sizeof(int32_t), stream));
  k_item_counts<<<grid_for(N + 1), kBlock, 0, stream>>>(rowptr, N, chunk, max_parts, c_parts, c_split, key);
  MPNN_LAUNCH_CHECK("k_item_counts");
  size_t t1 = temp;
  MPNN_HIP_CHECK(rocprim::exclusive_scan(temp_ptr, t1, c_parts, o_parts, 0, (size_t)(N + 1),
                                         rocprim::plus<int32_t>(), stream));
  size_t t2 = temp;
  MPNN_HIP_CHECK(rocprim::exclusive_scan(temp_ptr, t2, c_split, o_split, 0, (size_t)(N + 1),
                                         rocprim::plus<int32_t>(), stream));
  if (N > 0) {
    size_t t3 = temp;
    MPNN_HIP_CHECK(rocprim::radix_sort_pairs<SortConfig>(temp_ptr, t3, key, key_sorted, rocprim::counting_iterator<int32_t>(0),
                                             order, (size_t)N, 0u, order_bits(chunk), stream));
  }
  k_item_fill_rows<<<grid_for(N), kBlock, 0, stream>>>(rowptr, N, chunk, max_parts, o_parts, o_split, order,
                                                       reinterpret_cast<int4*>(items), items_capacity,
                                                       reinterpret_cast<int4*>(split), split_capacity, meta);
  MPNN_LAUNCH_CHECK("k_item_fill_rows");
  if (split_capacity > 0) {
    int64_t blocks = (split_capacity * 64 + kBlock - 1) / kBlock;
    if (blocks > kMaxBlocks) blocks = kMaxBlocks;
    k_item_fill_pieces<<<(int)blocks, kBlock, 0, stream>>>(rowptr, N, chunk, max_parts, o_split,
                                                           reinterpret_cast<const int4*>(split), split_capacity,
                                                           reinterpret_cast<int4*>(items), items_capacity, meta);
    MPNN_LAUNCH_CHECK("k_item_fill_pieces");
  }
  return MPNN_OK;
}

int64_t mpnn_csr_item_order_workspace_bytes(int64_t N) {
  if (N < 0 || N >= INT32_MAX) return -1;
  const int64_t temp = items_temp_bytes(N);
  if (temp < 0) return -1;
  return 3 * align_up((N + 1) * 4, 256) + align_up(temp + 256, 256);
}

int32_t mpnn_csr_item_order(const int32_t* rowptr, const int32_t* col, const int32_t* eid, int64_t N, int64_t E,
                            int32_t chunk, int32_t* order, int32_t* rowptr_out, int32_t* col_out, int32_t* eid_out,
                            void* workspace, int64_t workspace_bytes, mpnn_stream_t stream) {
  if (N < 0 || N >= INT32_MAX || E < 0 || E >= INT32_MAX || chunk < 1)
    return fail(MPNN_E_INVALID, "mpnn_csr_item_order: N=%lld E=%lld chunk=%d", (long long)N, (long long)E, chunk);
  if (!rowptr || !rowptr_out || !workspace || (N > 0 && !order) || (E > 0 && (!col || !eid || !col_out || !eid_out)))
    return fail(MPNN_E_INVALID, "mpnn_csr_item_order: null pointer argument");
  if (col == col_out || eid == eid_out) return fail(MPNN_E_INVALID, "mpnn_csr_item_order: outputs must not alias inputs");
  Arena arena(workspace, workspace_bytes);
  int32_t* key = arena.take<int32_t>(N + 1);
  int32_t* key_sorted = arena.take<int32_t>(N + 1);
  int32_t* deg_o = arena.take<int32_t>(N + 1);
  const int64_t temp_need = items_temp_bytes(N);
  if (temp_need < 0) return fail(MPNN_E_HIP, "mpnn_csr_item_order: rocPRIM size query failed");
  char* temp_ptr = arena.take<char>(temp_need + 256);
  if (!key || !key_sorted || !deg_o || !temp_ptr)
    return fail(MPNN_E_WORKSPACE, "mpnn_csr_item_order: workspace of %lld bytes is too small (need %lld)",
                (long long)workspace_bytes, (long long)mpnn_csr_item_order_workspace_bytes(N));
  if (N == 0) {
    MPNN_HIP_CHECK(hipMemsetAsync(rowptr_out, 0, sizeof(int32_t), stream));
    return MPNN_OK;
  }
  k_row_order_keys<<<grid_for(N), kBlock, 0, stream>>>(rowptr, N, chunk, key);
  MPNN_LAUNCH_CHECK("k_row_order_keys");
  size_t t = (size_t)temp_need;
  MPNN_HIP_CHECK(rocprim::radix_sort_pairs<SortConfig>(temp_ptr, t, key, key_sorted, rocprim::counting_iterator<int32_t>(0), order,
                                           (size_t)N, 0u, order_bits(chunk), stream));
  k_degrees_in_order<<<grid_for(N + 1), kBlock, 0, stream>>>(rowptr, order, N, deg_o);
  MPNN_LAUNCH_CHECK("k_degrees_in_order");
  t = (size_t)temp_need;
  MPNN_HIP_CHECK(rocprim::exclusive_scan(temp_ptr, t, deg_o, rowptr_out, 0, (size_t)(N + 1), rocprim::plus<int32_t>(),
                                         stream));
  if (E > 0) {
    int64_t blocks = (E + kPermTile - 1) / kPermTile;
    if (blocks > 8 * kMaxBlocks) blocks = 8 * kMaxBlocks;
    k_permute_segments<<<(int)blocks, kBlock, 0, stream>>>(rowptr, order, rowptr_out, N, E, col, eid, col_out, eid_out);
    MPNN_LAUNCH_CHECK("k_permute_segments");
  }
  return MPNN_OK;
}

int32_t mpnn_gather_rows(const void* src, const int32_t* index, void* dst, int64_t rows, int64_t row_bytes,
                         mpnn_stream_t stream) {
  if (rows < 0 || row_bytes <= 0 || (row_bytes & 1))
    return fail(MPNN_E_INVALID, "mpnn_gather_rows: rows=%lld row_bytes=%lld (need a positive multiple of 2)",
                (long long)rows, (long long)row_bytes);
  if (rows == 0) return MPNN_OK;
  if (!src || !index || !dst) return fail(MPNN_E_INVALID, "mpnn_gather_rows: null pointer argument");
  if ((row_bytes & 3) == 0 && (((uintptr_t)src | (uintptr_t)dst) & 3) == 0) {
    int words = (int)(row_bytes / 4);
    k_gather_rows<uint32_t><<<grid_for(rows * words), kBlock, 0, stream>>>(
        static_cast<const uint32_t*>(src), index, static_cast<uint32_t*>(dst), rows, words);
  } else {
    int words = (int)(row_bytes / 2);
    k_gather_rows<uint16_t><<<grid_for(rows * words), kBlock, 0, stream>>>(
        static_cast<const uint16_t*>(src), index, static_cast<uint16_t*>(dst), rows, words);
  }
  MPNN_LAUNCH_CHECK("k_gather_rows");
  return MPNN_OK;
}

}  // extern "C"
