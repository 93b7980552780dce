// Whole-layer kernels for batches of small graphs (molecules): forward in one launch, backward in one launch plus
// a fixed-order sum of the weight-gradient partials.
//
// An ISLAND is a run of consecutive node ids that no edge leaves: every edge with an end inside has both ends
// inside.  A collated batch of molecules is a list of islands of a few dozen nodes each (the graphs of the batch,
// data/dataset.py:94-96 builds such batches with the DataLoader), and a layer (models/layers.py:91-140) never
// moves data between islands.  One wavefront therefore runs the whole layer for one island out of its share of the
// LDS: the rows of x, the projection [A | B], the sums over incoming edges and, in the backward, the gradients of
// all of them never leave the CU.  Lane l owns output column l (and l + 64 for 128 hidden channels) of every dense
// product, so a product is a chain of fused multiply-adds over the contraction index in ascending order, and an
// edge sum is a chain over the island's destination-sorted edges in CSR order: every result is deterministic.
//
// The weight gradients are sums over ALL nodes / edges of the batch.  The waves of a block first finish their
// islands' node-level gradients, then the block multiplies the rows of all its islands together (16 output rows x
// 128 columns per wave and task, operands read from every wave's tile) and writes one partial per block;
// k_island_reduce adds the partials in a fixed order.
#include <limits.h>

#include "common.h"
#include "rowio.h"

namespace mpnn {
namespace {

constexpr int kIslandLds = 160 * 1024;   // LDS of one CU
constexpr int kIslandHeader = 16;        // floats in front of the tiles: rows held by each wave of the block
constexpr int kIslandFeMax = 8;
constexpr int kScanBlock = 1024;
constexpr int kRowsInRegisters = 32;   // island search: rows per thread that can stay in registers

// ---------------------------------------------------------------------------------------------------------------
// structure: the islands of a destination-sorted CSR
// ---------------------------------------------------------------------------------------------------------------
// Inclusive scan over the block's 1024 threads: a shuffle scan inside every wave, then the 16 wave totals.
template <typename Op>
__device__ __forceinline__ int block_scan_inclusive(int v, int* sm, Op op) {   // sm: 16 ints
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(v, off, 64);
    if (lane >= off) v = op(o, v);
  }
  if (lane == 63) sm[wv] = v;
  __syncthreads();
  for (int i = 0; i < wv; ++i) v = op(sm[wv - 1 - i], v);   // (op is commutative and associative: min, max, +)
  __syncthreads();
  return v;
}

// lo[r] / hi[r] = smallest / largest node id among row r and the sources of its edges.
// A row that alone spans `give_up` nodes or more (give_up > 0) settles that the graph has an island at least that
// large: meta[3] is set and the search stops there (every writer stores the same 1).
__global__ void __launch_bounds__(256) k_island_bounds(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                       int n, int32_t* __restrict__ lo, int32_t* __restrict__ hi,
                                                       int give_up, int32_t* __restrict__ meta) {
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
    int a = r, b = r;
    for (int e = rowptr[r]; e < rowptr[r + 1]; ++e) {
      const int c = col[e];
      a = c < a ? c : a;
      b = c > b ? c : b;
    }
    lo[r] = a;
    hi[r] = b;
    if (give_up > 0 && b - a + 1 >= give_up) meta[3] = 1;
  }
}

// A cut after node r (between r and r + 1) separates two islands iff no edge spans it: no row <= r reaches a
// node > r (prefix maximum of hi) and no row > r reaches a node <= r (suffix minimum of lo).  One block; thread t
// owns the rows [t k, (t + 1) k), k = ceil(n / 1024): it scans them itself and the block scans the threads' totals,
// so the number of barriers does not grow with n.  `lo` is overwritten by the suffix minima.
__global__ void __launch_bounds__(kScanBlock) k_islands_build(const int32_t* __restrict__ rowptr, int n, int32_t* lo,
                                                              const int32_t* __restrict__ hi,
                                                              int32_t* __restrict__ island_ptr, int32_t* __restrict__ meta) {
  __shared__ int sm[16];
  __shared__ int max_nodes, max_edges;
  const int t = threadIdx.x;
  const int k = (n + kScanBlock - 1) / kScanBlock;
  const int r0 = t * k < n ? t * k : n, r1 = (t + 1) * k < n ? (t + 1) * k : n;
  auto imin = [](int a, int b) { return a < b ? a : b; };
  auto imax = [](int a, int b) { return a > b ? a : b; };
  auto iadd = [](int a, int b) { return a + b; };
  if (meta[3] != 0) {   // (the whole block reads the same word) an island too large for the caller: nothing to list
    if (t == 0) { meta[0] = 0; meta[1] = 0; meta[2] = 0; }
    return;
  }
  if (t == 0) { max_nodes = 0; max_edges = 0; island_ptr[0] = 0; }
  __shared__ int totals[kScanBlock];
  int upto;
  if (k <= kRowsInRegisters) {
    // the thread's rows live in registers: one round of loads, everything after it is arithmetic
    int vlo[kRowsInRegisters], vhi[kRowsInRegisters];
#pragma unroll
    for (int i = 0; i < kRowsInRegisters; ++i) {
      const bool ok = r0 + i < r1;
      const int r = ok ? r0 + i : (n > 0 ? n - 1 : 0);
      const int a = n > 0 ? lo[r] : 0, b2 = n > 0 ? hi[r] : 0;
      vlo[i] = ok ? a : INT_MAX;
      vhi[i] = ok ? b2 : -1;
    }
    int mine = INT_MAX, top = -1;
#pragma unroll
    for (int i = 0; i < kRowsInRegisters; ++i) { mine = imin(mine, vlo[i]); top = imax(top, vhi[i]); }
    totals[kScanBlock - 1 - t] = mine;   // a scan over the threads in reverse order = a forward scan of the mirror
    __syncthreads();
    const int mirrored = block_scan_inclusive(totals[t], sm, imin);
    totals[t] = mirrored;
    __syncthreads();
    int after = t + 1 < kScanBlock ? totals[kScanBlock - 2 - t] : INT_MAX;   // min over the threads after t
    __syncthreads();
#pragma unroll
    for (int i = kRowsInRegisters - 1; i >= 0; --i) {   // vlo[i] <- minimum over the rows after r0 + i
      const int v = vlo[i];
      vlo[i] = after;
      after = imin(after, v);
    }
    const int incl = block_scan_inclusive(top, sm, imax);
    totals[t] = incl;
    __syncthreads();
    int before = t > 0 ? totals[t - 1] : -1;   // max over the threads before t
    __syncthreads();
    int cuts = 0;
    unsigned cut_bits = 0;
#pragma unroll
    for (int i = 0; i < kRowsInRegisters; ++i) {
      before = imax(before, vhi[i]);
      const bool cut = r0 + i < r1 && before <= r0 + i && vlo[i] > r0 + i;
      cuts += cut ? 1 : 0;
      cut_bits |= cut ? 1u << i : 0u;
    }
    upto = block_scan_inclusive(cuts, sm, iadd);
    int pos = upto - cuts;
#pragma unroll
    for (int i = 0; i < kRowsInRegisters; ++i)
      if (cut_bits >> i & 1u) island_ptr[++pos] = r0 + i + 1;
  } else {
    // suffix minimum of lo over the rows AFTER r: the thread's own rows from the back, then the later threads' totals
    int mine = INT_MAX;
#pragma unroll 8
    for (int r = r0; r < r1; ++r) mine = imin(mine, lo[r]);
    // a scan over the threads in reverse order = a forward scan of the mirrored values
    totals[kScanBlock - 1 - t] = mine;
    __syncthreads();
    const int mirrored = block_scan_inclusive(totals[t], sm, imin);   // min over threads >= (1023 - t)
    totals[t] = mirrored;
    __syncthreads();
    int after = t + 1 < kScanBlock ? totals[kScanBlock - 2 - t] : INT_MAX;   // min over threads > t
    __syncthreads();
    for (int r = r1 - 1; r >= r0; --r) {
      const int v = lo[r];
      lo[r] = after;
      after = imin(after, v);
    }
    // prefix maximum of hi up to and including r
    int top = -1;
#pragma unroll 8
    for (int r = r0; r < r1; ++r) top = imax(top, hi[r]);
    const int incl = block_scan_inclusive(top, sm, imax);
    totals[t] = incl;
    __syncthreads();
    const int before0 = t > 0 ? totals[t - 1] : -1;   // max over threads < t (kept: totals is rewritten below)
    __syncthreads();
    int before = before0;
    int cuts = 0;
    for (int r = r0; r < r1; ++r) {
      before = imax(before, hi[r]);
      cuts += (before <= r && lo[r] > r) ? 1 : 0;
    }
    upto = block_scan_inclusive(cuts, sm, iadd);
    int pos = upto - cuts;   // cuts of the threads before this one
    before = before0;
    for (int r = r0; r < r1; ++r) {
      before = imax(before, hi[r]);
      if (before <= r && lo[r] > r) island_ptr[++pos] = r + 1;
    }
  }
  totals[t] = upto;
  __syncthreads();
  const int islands = totals[kScanBlock - 1];
  int mn = 0, me = 0;
  for (int i = t; i < islands; i += kScanBlock) {
    const int b = island_ptr[i], e = island_ptr[i + 1];
    mn = imax(mn, e - b);
    me = imax(me, rowptr[e] - rowptr[b]);
  }
  atomicMax(&max_nodes, mn);   // integer maxima: the order does not matter
  atomicMax(&max_edges, me);
  __syncthreads();
  if (t == 0) { meta[0] = islands; meta[1] = max_nodes; meta[2] = max_edges; }
}

// ---------------------------------------------------------------------------------------------------------------
// the layer
// ---------------------------------------------------------------------------------------------------------------
struct IslandLayer {
  const int32_t* rowptr; const int32_t* col; const int32_t* island_ptr;
  int32_t n_islands, tail_units, rows;
  int64_t N;
  const void* x; int64_t ldx;
  const void* ea; const void* w_nodes; const void* w_edge; const void* w_out;
  const float* pos; const void* w_pos;
  int32_t F, H, Fe;
  void* y; int64_t ldy; float* pos_out;
  const void* gy; int64_t ldgy; const float* gpos_out;
  void* gx; int64_t ldgx; float* gpos;
  float* partial; int64_t partial_stride; float* gx_part;
};

// Phase timestamps of every wave for tools/island_prof.py (built only with -DMPNN_ISLAND_PROF).
#ifdef MPNN_ISLAND_PROF
__device__ long long* g_island_prof = nullptr;
#define ISLAND_PROF(slot) \
  do { if (g_island_prof && (threadIdx.x & 63) == 0) \
         g_island_prof[((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (slot)] = wall_clock64(); } while (0)
#define REGION_PROF(slot) \
  do { if (g_island_prof && (threadIdx.x & 63) == 0) \
         g_island_prof[((int64_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + (slot)] = wall_clock64(); } while (0)
#else
#define ISLAND_PROF(slot) do {} while (0)
#define REGION_PROF(slot) do {} while (0)
#endif

struct Unit { int n0, n; };

// Work unit u of a launch: island u, or - past the islands - a tile of `rows` edgeless rows of the tail
// [island_ptr[n_islands], N) (the padding rows of a fixed-capacity batch).
__device__ __forceinline__ Unit island_unit(const IslandLayer& a, int u) {
  Unit r{0, 0};
  if (u < a.n_islands) {
    r.n0 = a.island_ptr[u];
    r.n = a.island_ptr[u + 1] - r.n0;
  } else if (u < a.n_islands + a.tail_units) {
    const int64_t s = a.island_ptr[a.n_islands] + (int64_t)(u - a.n_islands) * a.rows;
    if (s < a.N) { r.n0 = (int)s; r.n = a.N - s < a.rows ? (int)(a.N - s) : a.rows; }
  }
  if (r.n > a.rows) r.n = a.rows;   // the host checked the largest island; never write past the tile
  if (r.n < 0) r.n = 0;
  return r;
}

// Units that hold rows: the islands and the tiles of the tail that start below N.  Blocks past them have nothing
// to do (the host sizes the grid for the longest possible tail without reading it back).
__device__ __forceinline__ int island_active_units(const int32_t* island_ptr, int n_islands, int tail_units, int rows,
                                                   int64_t N) {
  int64_t tail = tail_units > 0 ? N - island_ptr[n_islands] : 0;
  if (tail < 0) tail = 0;
  int64_t t = (tail + rows - 1) / rows;
  if (t > tail_units) t = tail_units;
  return n_islands + (int)t;
}

__device__ __forceinline__ void wave_sync() {   // LDS written by other lanes of this wave becomes readable
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Sum over the 64 lanes, the same bits in every lane: data-parallel-primitive adds inside each row of 16 lanes
// (lane pairs, quads, mirrored halves, mirrored row), then the rows' sums passed on to the next row, and the total
// read from the last lane.  Six vector adds instead of six round trips through the LDS crossbar.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true);
  return v + __int_as_float(moved);
}
__device__ __forceinline__ float wave_sum(float v) {
  v = dpp_add<0xB1, 0xf>(v);    // quad_perm [1, 0, 3, 2]
  v = dpp_add<0x4E, 0xf>(v);    // quad_perm [2, 3, 0, 1]
  v = dpp_add<0x141, 0xf>(v);   // row_half_mirror
  v = dpp_add<0x140, 0xf>(v);   // row_mirror: every lane holds the sum of its row of 16
  v = dpp_add<0x142, 0xa>(v);   // row_bcast15 into rows 1 and 3: rows 0+1, 2+3
  v = dpp_add<0x143, 0xc>(v);   // row_bcast31 into rows 2 and 3: row 3 holds all four
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

template <typename T>
__device__ __forceinline__ float4 load4(const T* p);
template <>
__device__ __forceinline__ float4 load4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <>
__device__ __forceinline__ float4 load4<bf16_t>(const bf16_t* p) {
  const uint2 t = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xffff0000u), __uint_as_float(t.y << 16),
                     __uint_as_float(t.y & 0xffff0000u));
}

// The dense products run on the float32 matrix cores (v_mfma_f32_32x32x2f32): an island's tile is 32 rows tall,
// which is the height of one MFMA tile, and a matrix-core operand is one value per lane - no operand has to be
// broadcast to all lanes through the LDS, which is what bounds the same products written as per-lane fma chains.
// Operand layout: A[m][k] sits in lane (m, k) = (lane % 32, lane / 32), B[k][n] in lane (n, k), and the
// accumulator's register i of lane (n, h) is D[8 (i / 4) + 4 h + i % 4][n].  The two contraction indices of one
// instruction may be any two, as long as A and B agree: lane half h takes the 4 consecutive indices 8 s + 4 h .. + 3
// of step s (one 16-byte read) and feeds them to 4 instructions.
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int acc_row(int i, int lh) { return 8 * (i >> 2) + 4 * lh + (i & 3); }

__device__ __forceinline__ void mfma4(f32x16& c, const float4& a, const float4& b) {
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, c, 0, 0, 0);
}

// acc[t] += in[32 x K] * W[c0 + 32 t .. + 31][K]^T: the tile's rows times the transposed weight rows of 32 NT output
// columns (the Linear layout, models/layers.py:84-87).  Contraction index k sits in tile column k below `split`
// and k + pad from there on (the tile holds [x | zero pad | aggr]).  VEC: K, split and the row starts are multiples
// of 4 and pad = 0.  The loads of a 64-wide chunk of the contraction are issued together, ahead of its MFMAs.
// XG: the A operand comes straight from the global rows of x (`xrow` = this lane's row, `row_ok` = it exists)
// instead of from the tile - for shapes whose tile would not leave room for four waves per block.
template <typename T, int NT, bool VEC, bool XG = false>
__device__ __forceinline__ void mm_wt(f32x16 (&acc)[NT], const float* in, int pitch, int K, int split, int pad, const T* W,
                                      int64_t ldw, int c0, int li, int lh, const T* xrow = nullptr, bool row_ok = false) {
  auto four = [&](auto get, int k) __attribute__((always_inline)) {   // get(k) for k .. k + 3, zero past K
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {   // the load itself is unconditional (index clamped): no branch per element
      const float t = get(k + q < K ? k + q : K - 1);
      v[q] = k + q < K ? t : 0.f;
    }
    return make_float4(v[0], v[1], v[2], v[3]);
  };
  for (int k0 = 0; k0 < K; k0 += 64) {
    float4 af[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (k0 + 8 * s >= K) break;   // (uniform) steps past the end of the contraction: nothing to load or multiply
      const int k = k0 + 8 * s + 4 * lh;
      if constexpr (XG) {
        af[s] = load4<T>(xrow + (k < K ? k : K - 4));
        if (k >= K || !row_ok) af[s] = make_float4(0.f, 0.f, 0.f, 0.f);
      } else if constexpr (VEC) {
        af[s] = *reinterpret_cast<const float4*>(in + li * pitch + (k < K ? k : K - 4));
        if (k >= K) af[s] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      else af[s] = four([&](int kk) __attribute__((always_inline)) { return in[li * pitch + (kk < split ? kk : kk + pad)]; }, k);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const T* wr = W + (int64_t)(c0 + 32 * t + li) * ldw;
      float4 bf[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        if (k0 + 8 * s >= K) break;
        const int k = k0 + 8 * s + 4 * lh;
        if constexpr (VEC) {
          bf[s] = load4<T>(wr + (k < K ? k : K - 4));
          if (k >= K) bf[s] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        else bf[s] = four([&](int kk) __attribute__((always_inline)) { return scalar_to_f32<T>(wr[kk]); }, k);
      }
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        if (k0 + 8 * s >= K) break;
        mfma4(acc[t], af[s], bf[s]);
      }
    }
  }
}

// acc[t] += in[32 x C] * W[C][col0 + 32 t .. + 31]: the tile's rows times weight columns (C a multiple of 32); the
// lanes of a half read consecutive columns of one weight row.  Columns from `ncols` on count as zero.
template <typename T, int NT>
__device__ __forceinline__ void mm_w(f32x16 (&acc)[NT], const float* in, int pitch, int C, const T* W, int64_t ldw,
                                     int col0, int ncols, int li, int lh) {
  for (int c0 = 0; c0 < C; c0 += 32) {
    float4 af[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) af[s] = *reinterpret_cast<const float4*>(in + li * pitch + c0 + 8 * s + 4 * lh);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = col0 + 32 * t + li;
      const bool ok = col < ncols;
      float4 bf[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const T* wp = W + (int64_t)(c0 + 8 * s + 4 * lh) * ldw + (ok ? col : 0);   // unconditional load, then select
        const float4 v = make_float4(scalar_to_f32<T>(wp[0]), scalar_to_f32<T>(wp[ldw]), scalar_to_f32<T>(wp[2 * ldw]),
                                     scalar_to_f32<T>(wp[3 * ldw]));
        bf[s] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) mfma4(acc[t], af[s], bf[s]);
    }
  }
}

// One 32 x 32 tile of a block-wide weight-gradient product: out[c0 + m][col0 + n] = sum over the rows of every
// wave's tile of left[row][c0 + m] * right[row][col0 + n] (right column j sits at tile column j below `split`, at
// j + pad from there on; columns from K on are not written).  XG: right columns below `split` are columns of x,
// read from global memory (island w2 starts at node n0s[w2]).  Waves in order, rows in ascending pairs; rows past
// a tile's last node hold zeros.  All operand reads of one island go out together, ahead of its MFMAs.
template <typename T, bool XG>
__device__ __forceinline__ void block_outer(float* out, int K, int c0, int col0, const float* tiles, int wave_floats,
                                            const int* nrows, const int* n0s, int wpb, int left, int ldl, int right,
                                            int ldr, int split, int pad, const T* x, int64_t ldx, int li, int lh) {
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int col = col0 + li;
  const bool ok = col < K;
  const bool from_x = XG && col0 < split;   // uniform: a 32-column tile lies on one side (split is a multiple of 32
                                            // whenever XG is chosen) or is cut by `ok`
  const int roff = right + (ok ? (col < split ? col : col + pad) : 0);
  if (from_x) {
    // global operand: the rows of all (at most 4) islands of the block requested before the first multiply
    float bx[4][16];
#pragma unroll
    for (int w2 = 0; w2 < 4; ++w2) {
      const int nr = w2 < wpb ? nrows[w2] : 0;
      const int64_t base = w2 < wpb ? n0s[w2] : 0;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = 2 * q + lh;
        const bool live = ok && row < nr;
        const float v = scalar_to_f32<T>(x[live ? (base + row) * ldx + col : 0]);
        bx[w2][q] = live ? v : 0.f;
      }
    }
#pragma unroll
    for (int w2 = 0; w2 < 4; ++w2) {
      if (w2 >= wpb) break;
      const float* tb = tiles + w2 * wave_floats;
      const int nr = nrows[w2];
      float av[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) av[q] = tb[left + (2 * q + lh) * ldl + c0 + li];
#pragma unroll
      for (int q = 0; q < 16; ++q)
        if (2 * q < nr) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bx[w2][q], acc, 0, 0, 0);
    }
  } else {
    for (int w2 = 0; w2 < wpb; ++w2) {
      const float* tb = tiles + w2 * wave_floats;
      const int nr = nrows[w2];
      if (nr == 0) continue;
      float av[16], bv[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = 2 * q + lh;   // < 32: the tile is 32 rows tall
        av[q] = tb[left + row * ldl + c0 + li];
        bv[q] = ok ? tb[roff + row * ldr] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < 16; ++q)
        if (2 * q < nr) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], acc, 0, 0, 0);
    }
  }
  if (ok) {
#pragma unroll
    for (int i = 0; i < 16; ++i) out[(int64_t)(c0 + acc_row(i, lh)) * K + col] = acc[i];
  }
}

// Per-wave LDS, in floats; every tile is kTileRows tall.  xa [32][px]: the rows of x, zero columns up to a
// multiple of 4, then the sums over edges; ab [32][pab]: the projection [A | B]; (backward) gz [32][pz]; the
// staging area of the edge loops; (EGNN) pos, and in the backward the gradients arriving at pos' and collected for
// pos, 4 floats per node.  Row pitches are odd multiples of 4 floats, so that the 16-byte operand reads of 8
// consecutive rows fall into different banks.
constexpr int kTileRows = 32;
constexpr int kStageEdges = 32;
__host__ __device__ inline int island_pitch(int width) { return width % 8 == 4 ? width : width + 4; }   // width % 4 == 0
// The tile [x | aggr] is read 4 bytes per lane when the widths do not allow 16-byte reads (row = lane): its rows
// then want an ODD pitch (32 rows in 32 different banks; the 16-byte pitch rule puts rows 8 apart into one bank).
// The tile is laid out for the wider of the two pitches either way.
__host__ __device__ inline int island_x_room(int width) { return island_pitch(width) + 1; }
__host__ __device__ inline int island_x_pitch(int width, bool vec) { return island_pitch(width) + (vec ? 0 : 1); }
__host__ __device__ inline int island_stage_floats(int Fe) { return kTileRows + 8 + kStageEdges + kStageEdges * kIslandFeMax; }
__host__ __device__ inline int island_wave_floats(int F, int H, int Fe, bool egnn, bool backward, bool xg = false) {
  const int Fp = xg ? 0 : (F + 3) & ~3;   // xg: the rows of x stay in global memory
  return kTileRows * (island_x_room(Fp + H) + island_pitch(2 * H) + (backward ? island_pitch(H) : 0)) +
         (island_stage_floats(Fe) + 3) / 4 * 4 + (egnn ? kTileRows * 4 * (backward ? 3 : 1) : 0);
}

// Rows [n0, n0 + n) of x into the tile's first F columns (zero rows below, zero pad columns).  The iterations are
// independent: unrolled, their loads are in flight together.
template <typename T, bool VEC>
__device__ __forceinline__ void load_rows(float* xa, int px, const T* x, int64_t ldx, int n0, int n, int F, int Fp, int lane) {
  if constexpr (VEC) {   // F % 4 == 0 (Fp = F): 4 columns per lane and step
    const int per_row = F >> 2;
#pragma unroll 8
    for (int i = lane; i < kTileRows * per_row; i += 64) {
      const int r = i / per_row, k = 4 * (i - r * per_row);
      const int rc = r < n ? r : n - 1;
      const T* p = x + (int64_t)(n0 + rc) * ldx + k;
      float4 v = make_float4(scalar_to_f32<T>(p[0]), scalar_to_f32<T>(p[1]), scalar_to_f32<T>(p[2]), scalar_to_f32<T>(p[3]));
      if (r >= n) v = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(xa + r * px + k) = v;
    }
  } else {
#pragma unroll 4
    for (int i = lane; i < kTileRows * Fp; i += 64) {
      const int r = i / Fp, k = i - r * Fp;
      const int rc = r < n ? r : n - 1, kc = k < F ? k : F - 1;
      const float v = scalar_to_f32<T>(x[(int64_t)(n0 + rc) * ldx + kc]);
      xa[r * px + k] = (r < n && k < F) ? v : 0.f;
    }
  }
}

// The lane's rows of the edge weight (and of the coordinate weight), held in registers.
template <typename T, int CPL, bool EGNN>
struct EdgeWeights {
  float we[CPL][kIslandFeMax];
  float wp[CPL];
  __device__ __forceinline__ void load(const IslandLayer& a, int lane) {
    const T* w = static_cast<const T*>(a.w_edge);
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
#pragma unroll
      for (int d = 0; d < kIslandFeMax; ++d) we[j][d] = d < a.Fe ? scalar_to_f32<T>(w[(lane + 64 * j) * a.Fe + d]) : 0.f;
      wp[j] = EGNN ? scalar_to_f32<T>(static_cast<const T*>(a.w_pos)[lane + 64 * j]) : 0.f;
    }
  }
  // We[c, :] . e as an fma chain in feature order (the order of the general edge kernels)
  // (all 8 terms: weights and attributes past the real width are zeros, and adding an exact zero changes nothing)
  __device__ __forceinline__ float term(int j, const float (&ev)[kIslandFeMax]) const {
    float s = we[j][0] * ev[0];
#pragma unroll
    for (int d = 1; d < kIslandFeMax; ++d) s = fmaf(we[j][d], ev[d], s);
    return s;
  }
};

// Walks the edges of rows [n0, n0 + n) in CSR order: begin(r) when row r starts, edge(r, s, ev, f) for every edge
// into r from node col_base + s with attributes ev, end(r) when the row is complete; rows without edges get begin
// and end too (n >= 1).
// f = fetch(s) is what the edge needs of its source's rows, read ahead of use: the fetches of G edges go out
// together, one group ahead of the group being used (a wave that runs almost alone on its SIMD has nothing else
// to cover their latency; G = 4 covers an LDS round trip, fetches from global memory want 16).  The row pointers,
// source ids and attributes pass through LDS in coalesced pieces of ST edges, and the piece after the one being
// walked is already on its way from global memory, in registers.
template <typename T, typename Fetched, int ST = kStageEdges, int G = 4, typename Begin, typename Fetch, typename Edge,
          typename End>
__device__ __forceinline__ void walk_edges(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                           const T* __restrict__ ea, int Fe, float* stage, int n0, int n, int col_base,
                                           int lane, Begin begin, Fetch fetch, Edge edge, End end) {
  constexpr int NV = ST * kIslandFeMax / 64;   // attribute values per lane and piece
  int* rp = reinterpret_cast<int*>(stage);
  int* ecol = rp + kTileRows + 8;
  float* eat = reinterpret_cast<float*>(ecol + ST);
  for (int i = lane; i <= n; i += 64) rp[i] = rowptr[n0 + i];
  wave_sync();
  const int e_first = __builtin_amdgcn_readfirstlane(rp[0]), e_last = __builtin_amdgcn_readfirstlane(rp[n]);
  int r = 0;
  int row_end = __builtin_amdgcn_readfirstlane(rp[1]);
  begin(0);
  int pc = 0;        // the next piece: source id of edge `lane`, attribute values lane + 64 k (rows of 8 floats,
  float pv[NV];      // zeros past the real width)
  auto request = [&](int eb) __attribute__((always_inline)) {
    const int cnt = e_last - eb < ST ? e_last - eb : ST;   // (<= 0 past the last piece: nothing is read)
    const int e = eb + (lane < cnt ? lane : 0);
    pc = lane < cnt ? col[e] - col_base : 0;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = lane + 64 * k, q = i >> 3, d = i & 7;
      const bool live = q < cnt && d < Fe;
      const float v = scalar_to_f32<T>(ea[live ? (int64_t)(eb + q) * Fe + d : 0]);
      pv[k] = live ? v : 0.f;
    }
  };
  if (e_first < e_last) request(e_first);
  for (int eb = e_first; eb < e_last; eb += ST) {
    const int cnt = e_last - eb < ST ? e_last - eb : ST;
    if (lane < ST) ecol[lane] = pc;
#pragma unroll
    for (int k = 0; k < NV; ++k) eat[lane + 64 * k] = pv[k];
    wave_sync();
    if (eb + ST < e_last) request(eb + ST);
    auto fetch_group = [&](int i0, Fetched (&f)[G], int (&sv)[G]) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < G; ++q) sv[q] = ecol[i0 + q < cnt ? i0 + q : cnt - 1];
#pragma unroll
      for (int q = 0; q < G; ++q) {
        sv[q] = __builtin_amdgcn_readfirstlane(sv[q]);
        f[q] = fetch(sv[q]);
      }
    };
    Fetched f[G], fn[G];
    int sv[G], svn[G];
    fetch_group(0, f, sv);
    for (int i0 = 0; i0 < cnt; i0 += G) {
      if (i0 + G < cnt) fetch_group(i0 + G, fn, svn);
#pragma unroll
      for (int q = 0; q < G; ++q) {
        if (i0 + q < cnt) {
          const int i = i0 + q;
          const float4 lo = *reinterpret_cast<const float4*>(eat + i * kIslandFeMax);
          const float4 hi = *reinterpret_cast<const float4*>(eat + i * kIslandFeMax + 4);
          const float ev[kIslandFeMax] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
          while (eb + i >= row_end) {
            end(r);
            ++r;
            row_end = __builtin_amdgcn_readfirstlane(rp[r + 1]);
            begin(r);
          }
          edge(r, sv[q], ev, f[q]);
        }
      }
#pragma unroll
      for (int q = 0; q < G; ++q) { f[q] = fn[q]; sv[q] = svn[q]; }
    }
    wave_sync();
  }
  for (;;) {
    end(r);
    if (++r >= n) break;
    begin(r);
  }
}

// Stores a wave's accumulator tile: put(row, value) for the lane's 16 rows of its column.
template <typename Put>
__device__ __forceinline__ void for_acc(const f32x16& acc, int lh, Put put) {
#pragma unroll
  for (int i = 0; i < 16; ++i) put(acc_row(i, lh), acc[i]);
}

// [A | B] = x Wn^T for the tile's rows (models/layers.py:104-105).
template <typename T, int CPL, bool VEC, bool XG = false>
__device__ __forceinline__ void project(const IslandLayer& a, const float* xa, int px, float* ab, int pab, int li, int lh,
                                        int n0 = 0, int n = 0) {
  constexpr int NT = 4 * CPL;   // 2 H / 32 column tiles
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  const T* xrow = XG ? static_cast<const T*>(a.x) + (int64_t)(n0 + (li < n ? li : n - 1)) * a.ldx : nullptr;
  mm_wt<T, NT, VEC, XG>(acc, xa, px, a.F, a.F, 0, static_cast<const T*>(a.w_nodes), a.F, 0, li, lh, xrow, li < n);
#pragma unroll
  for (int t = 0; t < NT; ++t)
    for_acc(acc[t], lh, [&](int row, float v) __attribute__((always_inline)) { ab[row * pab + 32 * t + li] = v; });
}

template <int CPL>
struct SourceRow { float b[CPL]; float px, py, pz; };   // the lane's columns of B[s]; the coordinates of s (EGNN)
template <int CPL>
using SourceRowBwd = SourceRow<CPL>;

// Messages of the island's edges summed per destination (models/layers.py:128 and the add-aggregation) into the
// tile's aggr columns, plus the coordinate update of the EGNN form: pos'[n] = pos[n] + sum (pos[n] - pos[src]) *
// (w_pos . m).
template <typename T, int CPL, bool EGNN>
__device__ __forceinline__ void edges_forward(const IslandLayer& a, const EdgeWeights<T, CPL, EGNN>& ew, const float* ab,
                                              int pab, float* ag, int pag, const float* ps, float* stage, int n0, int n,
                                              int lane, bool write_pos) {
  constexpr int H = 64 * CPL;
  float av[CPL], sum[CPL];
  // coordinate update of a row: sum_e rel_e (w_pos . m_e) = sum_c w_pos[c] (sum_e rel_e m_e[c]) - the lane keeps the
  // inner sums of its columns, and the sum over the columns is taken once per row instead of once per edge
  float vx = 0.f, vy = 0.f, vz = 0.f;
  const int Fe = a.Fe;
  float* const pos_out = a.pos_out;
  float prx = 0.f, pry = 0.f, prz = 0.f;
  walk_edges<T, SourceRow<CPL>>(
      a.rowptr, a.col, static_cast<const T*>(a.ea), Fe, stage, n0, n, n0, lane,
      [&](int r) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < CPL; ++j) { av[j] = ab[r * pab + lane + 64 * j]; sum[j] = 0.f; }
        vx = vy = vz = 0.f;
        if constexpr (EGNN) { prx = ps[4 * r]; pry = ps[4 * r + 1]; prz = ps[4 * r + 2]; }
      },
      [&](int s) __attribute__((always_inline)) {
        SourceRow<CPL> f;
#pragma unroll
        for (int j = 0; j < CPL; ++j) f.b[j] = ab[s * pab + H + lane + 64 * j];
        f.px = EGNN ? ps[4 * s] : 0.f;
        f.py = EGNN ? ps[4 * s + 1] : 0.f;
        f.pz = EGNN ? ps[4 * s + 2] : 0.f;
        return f;
      },
      [&](int r, int s, const float (&ev)[kIslandFeMax], const SourceRow<CPL>& f) __attribute__((always_inline)) {
        float wm = 0.f;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
          const float pre = (av[j] + f.b[j]) + ew.term(j, ev);
          const float m = fmaxf(pre, 0.f);
          sum[j] += m;
          if constexpr (EGNN) wm = fmaf(ew.wp[j], m, wm);
        }
        if constexpr (EGNN) {
          vx = fmaf(prx - f.px, wm, vx);
          vy = fmaf(pry - f.py, wm, vy);
          vz = fmaf(prz - f.pz, wm, vz);
        }
      },
      [&](int r) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < CPL; ++j) ag[r * pag + lane + 64 * j] = sum[j];
        if constexpr (EGNN) {
          if (write_pos) {
            const float dx = wave_sum(vx), dy = wave_sum(vy), dz = wave_sum(vz);
            if (lane == 0) {
              pos_out[(int64_t)(n0 + r) * 3] = prx + dx;
              pos_out[(int64_t)(n0 + r) * 3 + 1] = pry + dy;
              pos_out[(int64_t)(n0 + r) * 3 + 2] = prz + dz;
            }
          }
        }
      });
}

template <typename T, int CPL, bool EGNN, bool VEC>
__global__ void __launch_bounds__(256) k_island_fwd(IslandLayer a) {
  extern __shared__ __align__(16) float lds[];
  constexpr int H = 64 * CPL;
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wpb = blockDim.x >> 6;
  const int F = a.F, Fp = (F + 3) & ~3;
  const int px = island_x_pitch(Fp + H, VEC), pab = island_pitch(2 * H);
  float* xa = lds + kIslandHeader + w * island_wave_floats(F, H, a.Fe, EGNN, false);
  float* ab = xa + kTileRows * island_x_room(Fp + H);
  float* stage = ab + kTileRows * pab;
  float* ps = stage + (island_stage_floats(a.Fe) + 3) / 4 * 4;
  const Unit un = island_unit(a, blockIdx.x * wpb + w);
  if (un.n == 0) return;
  const int n = un.n, n0 = un.n0;
  ISLAND_PROF(0);
  load_rows<T, VEC>(xa, px, static_cast<const T*>(a.x), a.ldx, n0, n, F, Fp, lane);
  if ((int)blockIdx.x * wpb + w >= a.n_islands) {
    // A tile of the edgeless tail: when its rows of x are zeros (the padding of a fixed-capacity batch) the layer's
    // output is relu(0) = 0 exactly and the coordinates pass through - nothing to multiply.
    wave_sync();
    bool nz = false;
    for (int i = lane; i < n * Fp; i += 64) nz |= xa[(i / Fp) * px + i % Fp] != 0.f;
    if (!__any(nz)) {
      T* y = static_cast<T*>(a.y);
      for (int i = lane; i < n * H; i += 64) y[(int64_t)(n0 + i / H) * a.ldy + (i & (H - 1))] = f32_to_scalar<T>(0.f);
      if constexpr (EGNN)
        for (int i = lane; i < n * 3; i += 64) a.pos_out[(int64_t)n0 * 3 + i] = a.pos[(int64_t)n0 * 3 + i];
      return;
    }
  }
  if constexpr (EGNN)
    for (int i = lane; i < n * 4; i += 64) ps[i] = (i & 3) < 3 ? a.pos[(int64_t)(n0 + (i >> 2)) * 3 + (i & 3)] : 0.f;
  EdgeWeights<T, CPL, EGNN> ew;
  ew.load(a, lane);
  wave_sync();
  ISLAND_PROF(1);
  project<T, CPL, VEC>(a, xa, px, ab, pab, li, lh);
  wave_sync();
  ISLAND_PROF(2);
  edges_forward<T, CPL, EGNN>(a, ew, ab, pab, xa + Fp, px, ps, stage, n0, n, lane, true);
  for (int i = lane; i < (kTileRows - n) * H; i += 64) xa[(n + i / H) * px + Fp + (i & (H - 1))] = 0.f;
  wave_sync();
  ISLAND_PROF(3);
  // y = relu([x | aggr] W_out^T) (models/layers.py:140)
  {
    constexpr int NT = 2 * CPL;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    mm_wt<T, NT, VEC>(acc, xa, px, F + H, F, Fp - F, static_cast<const T*>(a.w_out), F + H, 0, li, lh);
    T* y = static_cast<T*>(a.y);
    const int64_t ldy = a.ldy;
#pragma unroll
    for (int t = 0; t < NT; ++t)
      for_acc(acc[t], lh, [&](int row, float v) __attribute__((always_inline)) {
        if (row < n) y[(int64_t)(n0 + row) * ldy + 32 * t + li] = f32_to_scalar<T>(fmaxf(v, 0.f));
      });
  }
  ISLAND_PROF(4);
}

// Partial weight gradients of one block: [gW_out H x (F+H)] [gWn 2H x F] [gWe H x Fe] [g w_pos H].
__host__ __device__ inline int64_t island_partial_floats(int F, int H, int Fe, bool egnn) {
  const int64_t v = (int64_t)H * (F + H) + (int64_t)2 * H * F + (int64_t)H * Fe + (egnn ? H : 0);
  return (v + 63) / 64 * 64;
}

template <typename T, int CPL, bool EGNN, bool VEC, bool XG>
__global__ void __launch_bounds__(256) k_island_bwd(IslandLayer a) {
  extern __shared__ __align__(16) float lds[];
  constexpr int H = 64 * CPL;
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wpb = blockDim.x >> 6;
  const int F = a.F, Fp = XG ? 0 : (F + 3) & ~3;   // XG: no columns of x in the tile
  const int px = island_x_pitch(Fp + H, VEC), pab = island_pitch(2 * H), pz = island_pitch(H);
  const int wave_floats = island_wave_floats(F, H, a.Fe, EGNN, true, XG);
  int* nrows = reinterpret_cast<int*>(lds);
  int* n0s = nrows + 4;
  float* tiles = lds + kIslandHeader;
  const int o_ab = kTileRows * island_x_room(Fp + H), o_gz = o_ab + kTileRows * pab, o_stage = o_gz + kTileRows * pz;
  const int o_ps = o_stage + (island_stage_floats(a.Fe) + 3) / 4 * 4;
  float* xa = tiles + w * wave_floats;
  float* ab = xa + o_ab;
  float* ag = xa + Fp;     // pitch px: sums over edges, later the gradient that reaches them
  float* gz = xa + o_gz;   // gy * [y > 0], later the gradient of B
  float* stage = xa + o_stage;
  float* ps = xa + o_ps;   // pos | gradient of pos' | gradient of pos (EGNN)
  float* gp = ps + 4 * kTileRows;
  float* gq = gp + 4 * kTileRows;
  float* part = a.partial + (int64_t)blockIdx.x * a.partial_stride;
  const int64_t o_wn = (int64_t)H * (F + H), o_we = o_wn + (int64_t)2 * H * F;   // g w_pos follows gWe

  if ((int)blockIdx.x * wpb >= island_active_units(a.island_ptr, a.n_islands, a.tail_units, kTileRows, a.N)) return;   // whole block
  const Unit un = island_unit(a, blockIdx.x * wpb + w);
  const int n0 = un.n0;
  int n = un.n;
  ISLAND_PROF(0);
  EdgeWeights<T, CPL, EGNN> ew;
  ew.load(a, lane);
  if (n > 0) {
    if constexpr (!XG) load_rows<T, VEC>(xa, px, static_cast<const T*>(a.x), a.ldx, n0, n, F, Fp, lane);
    if constexpr (EGNN) {
      for (int i = lane; i < n * 4; i += 64) {
        const bool c = (i & 3) < 3;
        const int64_t g = (int64_t)(n0 + (i >> 2)) * 3 + (i & 3);
        ps[i] = c ? a.pos[g] : 0.f;
        gp[i] = c ? a.gpos_out[g] : 0.f;
        gq[i] = 0.f;
      }
    }
    // gz = gy * [y > 0] (models/layers.py:140 backwards)
    const T* gy = static_cast<const T*>(a.gy);
    const T* y = static_cast<const T*>(a.y);
    const int64_t ldy = a.ldy, ldgy = a.ldgy;
    const bool wide = ldy % 4 == 0 && ldgy % 4 == 0 && (((uintptr_t)gy | (uintptr_t)y) & 15) == 0;   // (uniform)
    if (wide) {   // 4 columns per lane and load; the loads of 8 steps (both arrays) are in flight together
#pragma unroll 8
      for (int i = lane; i < kTileRows * (H / 4); i += 64) {
        const int r = i / (H / 4), c = 4 * (i % (H / 4));
        const int rc = r < n ? r : n - 1;
        const float4 yv = load4<T>(y + (int64_t)(n0 + rc) * ldy + c);
        const float4 gv = load4<T>(gy + (int64_t)(n0 + rc) * ldgy + c);
        const bool in = r < n;
        *reinterpret_cast<float4*>(gz + r * pz + c) =
            make_float4(in && yv.x > 0.f ? gv.x : 0.f, in && yv.y > 0.f ? gv.y : 0.f, in && yv.z > 0.f ? gv.z : 0.f,
                        in && yv.w > 0.f ? gv.w : 0.f);
      }
    } else {
#pragma unroll 8
      for (int i = lane; i < kTileRows * H; i += 64) {   // both loads unconditional (row clamped): all in flight together
        const int r = i / H, c = i & (H - 1);
        const int rc = r < n ? r : n - 1;
        const float yv = scalar_to_f32<T>(y[(int64_t)(n0 + rc) * ldy + c]);
        const float gv = scalar_to_f32<T>(gy[(int64_t)(n0 + rc) * ldgy + c]);
        gz[r * pz + c] = (r < n && yv > 0.f) ? gv : 0.f;
      }
    }
    wave_sync();
    if ((int)blockIdx.x * wpb + w >= a.n_islands) {
      // A tile of the edgeless tail that no gradient reaches through the relu (the padding rows of a
      // fixed-capacity batch: their y is 0): every product below would be a product with zeros.  gx = 0, the
      // coordinates' gradient passes through, and the tile takes no part in the weight gradients.
      bool nz = false;
      for (int i = lane; i < n * H; i += 64) nz |= gz[(i / H) * pz + (i & (H - 1))] != 0.f;
      if (!__any(nz)) {
        if (a.gx) {
          T* gx = static_cast<T*>(a.gx);
          for (int i = lane; i < n * F; i += 64) gx[(int64_t)(n0 + i / F) * a.ldgx + i % F] = f32_to_scalar<T>(0.f);
        }
        if constexpr (EGNN)
          for (int i = lane; i < n * 3; i += 64) a.gpos[(int64_t)n0 * 3 + i] = a.gpos_out[(int64_t)n0 * 3 + i];
        n = 0;
      }
    }
  }
  if (lane == 0) { nrows[w] = n; n0s[w] = n0; }
  if (n > 0) {
    ISLAND_PROF(1);
    project<T, CPL, VEC, XG>(a, xa, px, ab, pab, li, lh, n0, n);
    wave_sync();
    ISLAND_PROF(2);
    edges_forward<T, CPL, EGNN>(a, ew, ab, pab, ag, px, ps, stage, n0, n, lane, false);
    for (int i = lane; i < (kTileRows - n) * H; i += 64) ag[(n + i / H) * px + (i & (H - 1))] = 0.f;
    ISLAND_PROF(3);
  }
  __syncthreads();
  ISLAND_PROF(4);
  {
    // a block whose tiles all dropped out (padding rows of a fixed-capacity batch): its share of every weight gradient
    // is zero - write that and leave
    int live = 0;
    for (int w2 = 0; w2 < wpb; ++w2) live |= nrows[w2];
    if (live == 0) {
      for (int64_t i = threadIdx.x; i < a.partial_stride; i += blockDim.x) part[i] = 0.f;
      return;
    }
  }
  // gW_out = gz^T [x | aggr] over the rows of the whole block
  {
    const int ktiles = (F + H + 31) / 32;
    const int tasks = (H / 32) * ktiles;
    for (int t = w; t < tasks; t += wpb)
      block_outer<T, XG>(part, F + H, 32 * (t / ktiles), 32 * (t % ktiles), tiles, wave_floats, nrows, n0s, wpb, o_gz, pz,
                         0, px, F, Fp - F, static_cast<const T*>(a.x), a.ldx, li, lh);
  }
  ISLAND_PROF(5);
  __syncthreads();
  ISLAND_PROF(6);
  float gwe[CPL][kIslandFeMax], gwp[CPL];
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    gwp[j] = 0.f;
#pragma unroll
    for (int d = 0; d < kIslandFeMax; ++d) gwe[j][d] = 0.f;
  }
  if (n > 0) {
    // [gx (first part) | ga] = gz W_out: the aggr columns always, the x columns when gx is wanted
    {
      const T* wo = static_cast<const T*>(a.w_out);
      float* const gx_part = a.gx_part;
      const int first = a.gx ? 0 : (F / 32) * 32;    // the 32-column tile that holds column F
      for (int col0 = first; col0 < F + H; col0 += 64) {
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        mm_w<T, 2>(acc, gz, pz, H, wo, F + H, col0, F + H, li, lh);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int col = col0 + 32 * t + li;
          if (col >= F + H) continue;
          if (col >= F) {
            for_acc(acc[t], lh, [&](int row, float v) __attribute__((always_inline)) { ag[row * px + col - F] = v; });
          } else if (gx_part) {
            for_acc(acc[t], lh, [&](int row, float v) __attribute__((always_inline)) {
              if (row < n) gx_part[(int64_t)(n0 + row) * F + col] = v;
            });
          }
        }
      }
    }
    wave_sync();
    ISLAND_PROF(7);
    // gz is dead: its slab collects the gradient of B
    for (int i = lane; i < kTileRows * pz; i += 64) gz[i] = 0.f;
    wave_sync();
    float av[CPL], gar[CPL], ga_sum[CPL];
    float gpx = 0.f, gpy = 0.f, gpz = 0.f, sc_sum = 0.f;
    const int Fe = a.Fe;
    float prx = 0.f, pry = 0.f, prz = 0.f, gpl = 0.f;
    walk_edges<T, SourceRowBwd<CPL>>(
        a.rowptr, a.col, static_cast<const T*>(a.ea), Fe, stage, n0, n, n0, lane,
        [&](int r) __attribute__((always_inline)) {
#pragma unroll
          for (int j = 0; j < CPL; ++j) {
            av[j] = ab[r * pab + lane + 64 * j];
            gar[j] = ag[r * px + lane + 64 * j];
            ga_sum[j] = 0.f;
          }
          sc_sum = 0.f;
          if constexpr (EGNN) {
            gpx = gp[4 * r]; gpy = gp[4 * r + 1]; gpz = gp[4 * r + 2];
            prx = ps[4 * r]; pry = ps[4 * r + 1]; prz = ps[4 * r + 2];
            gpl = gp[4 * r + (lane < 3 ? lane : 3)];
          }
        },
        [&](int s) __attribute__((always_inline)) {
          SourceRowBwd<CPL> f;
#pragma unroll
          for (int j = 0; j < CPL; ++j) f.b[j] = ab[s * pab + H + lane + 64 * j];
          f.px = EGNN ? ps[4 * s] : 0.f;
          f.py = EGNN ? ps[4 * s + 1] : 0.f;
          f.pz = EGNN ? ps[4 * s + 2] : 0.f;
          return f;
        },
        [&](int r, int s, const float (&ev)[kIslandFeMax], const SourceRowBwd<CPL>& f) __attribute__((always_inline)) {
          float pre[CPL], m[CPL], dot = 0.f;
#pragma unroll
          for (int j = 0; j < CPL; ++j) {
            pre[j] = (av[j] + f.b[j]) + ew.term(j, ev);
            m[j] = fmaxf(pre[j], 0.f);
            if constexpr (EGNN) dot = fmaf(ew.wp[j], m[j], dot);
          }
          float q = 0.f;
          if constexpr (EGNN) {
            // pos'[r] += (pos[r] - pos[s]) * sc, sc = w_pos . m: q = d loss / d sc
            const float sc = wave_sum(dot);
            q = gpx * (prx - f.px);
            q = fmaf(gpy, pry - f.py, q);
            q = fmaf(gpz, prz - f.pz, q);
            sc_sum += sc;
            if (lane < 3) gq[4 * s + lane] -= gpl * sc;
          }
#pragma unroll
          for (int j = 0; j < CPL; ++j) {
            float g = EGNN ? fmaf(q, ew.wp[j], gar[j]) : gar[j];
            g = pre[j] > 0.f ? g : 0.f;
            ga_sum[j] += g;
            gz[s * pz + lane + 64 * j] += g;
            if constexpr (EGNN) gwp[j] = fmaf(q, m[j], gwp[j]);
#pragma unroll
            for (int d = 0; d < kIslandFeMax; ++d) gwe[j][d] = fmaf(g, ev[d], gwe[j][d]);
          }
        },
        [&](int r) __attribute__((always_inline)) {
#pragma unroll
          for (int j = 0; j < CPL; ++j) ab[r * pab + lane + 64 * j] = ga_sum[j];   // A[r] is dead: gradient of A[r]
          if constexpr (EGNN) {
            if (lane < 3) gq[4 * r + lane] += gpl * (1.f + sc_sum);
          }
        });
    wave_sync();
    for (int i = lane; i < kTileRows * H; i += 64) ab[(i / H) * pab + H + (i & (H - 1))] = gz[(i / H) * pz + (i & (H - 1))];
    if constexpr (EGNN)
      for (int i = lane; i < n * 3; i += 64) a.gpos[(int64_t)n0 * 3 + i] = gq[4 * (i / 3) + i % 3];
    wave_sync();
    ISLAND_PROF(8);
    if (a.gx) {   // gx = gz Wx + [gA | gB] Wn
      const T* wn = static_cast<const T*>(a.w_nodes);
      T* gx = static_cast<T*>(a.gx);
      const float* const gx_part = a.gx_part;
      const int64_t ldgx = a.ldgx;
      for (int col0 = 0; col0 < F; col0 += 64) {
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int col = col0 + 32 * t + li;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int row = acc_row(i, lh);
            acc[t][i] = (col < F && row < n) ? gx_part[(int64_t)(n0 + row) * F + col] : 0.f;
          }
        }
        mm_w<T, 2>(acc, ab, pab, 2 * H, wn, F, col0, F, li, lh);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int col = col0 + 32 * t + li;
          if (col < F)
            for_acc(acc[t], lh, [&](int row, float v) __attribute__((always_inline)) {
              if (row < n) gx[(int64_t)(n0 + row) * ldgx + col] = f32_to_scalar<T>(v);
            });
        }
      }
    }
  }
  ISLAND_PROF(9);
  // the lane's rows of gWe and g w_pos go through the (now dead) gz slab of its wave, and the block adds the waves'
  // shares in wave order
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
#pragma unroll
    for (int d = 0; d < kIslandFeMax; ++d)
      if (d < a.Fe) gz[(lane + 64 * j) * a.Fe + d] = gwe[j][d];
    if constexpr (EGNN) gz[H * a.Fe + lane + 64 * j] = gwp[j];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < H * a.Fe + (EGNN ? H : 0); e += blockDim.x) {
    float t = 0.f;
    for (int w2 = 0; w2 < wpb; ++w2) t += tiles[w2 * wave_floats + o_gz + e];
    part[o_we + e] = t;
  }
  ISLAND_PROF(10);
  // gWn = [gA | gB]^T x over the rows of the whole block
  {
    const int ktiles = (F + 31) / 32;
    const int tasks = (2 * H / 32) * ktiles;
    for (int t = w; t < tasks; t += wpb)
      block_outer<T, XG>(part + o_wn, F, 32 * (t / ktiles), 32 * (t % ktiles), tiles, wave_floats, nrows, n0s, wpb, o_ab,
                         pab, 0, px, F, 0, static_cast<const T*>(a.x), a.ldx, li, lh);
  }
  ISLAND_PROF(11);
}

// Sums the blocks' partials in a fixed order: a thread takes one element and every fourth block in ascending order
// (eight loads in flight), then the four sums of an element are added in ascending order.
struct IslandReduce {
  const float* partial; int64_t stride; const int32_t* island_ptr; int64_t N; int32_t n_islands, tail_units, rows;
  int32_t wpb, F, H, Fe, egnn;
  void* g_wout; void* g_wn; void* g_we; void* g_wp;
};

template <typename T>
__global__ void __launch_bounds__(256) k_island_reduce(IslandReduce a) {
  __shared__ float sm[4][64];
  const int el = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int64_t n1 = (int64_t)a.H * (a.F + a.H), n2 = (int64_t)2 * a.H * a.F, n3 = (int64_t)a.H * a.Fe;
  const int64_t total = n1 + n2 + n3 + (a.egnn ? a.H : 0);
  const int64_t e = (int64_t)blockIdx.x * 64 + el;
  float s = 0.f;
  if (e < total) {
    const int units = (a.n_islands + a.tail_units) > 0
                          ? island_active_units(a.island_ptr, a.n_islands, a.tail_units, a.rows, a.N) : 0;
    const int blocks = (units + a.wpb - 1) / a.wpb;   // the blocks that wrote a partial
    const float* p = a.partial + e;
#pragma unroll 8
    for (int b = g; b < blocks; b += 4) s += p[(int64_t)b * a.stride];
  }
  sm[g][el] = s;
  __syncthreads();
  if (g == 0 && e < total) {
    const float t = ((sm[0][el] + sm[1][el]) + sm[2][el]) + sm[3][el];
    T* dst;
    if (e < n1) dst = a.g_wout ? static_cast<T*>(a.g_wout) + e : nullptr;
    else if (e < n1 + n2) dst = a.g_wn ? static_cast<T*>(a.g_wn) + (e - n1) : nullptr;
    else if (e < n1 + n2 + n3) dst = a.g_we ? static_cast<T*>(a.g_we) + (e - n1 - n2) : nullptr;
    else dst = a.g_wp ? static_cast<T*>(a.g_wp) + (e - n1 - n2 - n3) : nullptr;
    if (dst) *dst = f32_to_scalar<T>(t);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Regions: islands of up to kRegionMax nodes over rows of x at most 4 wide (batches of point clouds)
// ---------------------------------------------------------------------------------------------------------------
// A k-NN cloud (experiments/modelnet_exp.py:58) is an island of a thousand nodes - far more than one tile - but its
// rows of x are the 3 coordinates, so the x of the WHOLE island fits the LDS many times over.  The projection rows
// A[r] = Wn_A x[r] and B[s] = Wn_B x[s] of an edge (models/layers.py:104-105) are then cheaper to recompute per edge
// from the island's table of x (F multiply-adds per column) than to fetch (a 256-byte row of B per edge from HBM
// or L2), and the table serves sources anywhere in the island.  A block of 8 waves takes 4 tiles of 32 consecutive
// rows of one island, two waves per tile: each walks the edges of 16 of the tile's rows, and in the matrix-core
// products the pair splits the output columns.
//
// The edge walk: a wave is four groups of 16 lanes, each group walks one row at a time, lane l of a group owning
// columns 4 l .. 4 l + 3 (H = 64).  What an edge needs besides the lane's weights - its attributes and its other
// end's row of x - is the same for the 16 lanes; lane k of the group loads it for the k-th edge of the row (one
// coalesced read of the index and attribute arrays, one read of the x table) and the group takes edge k's values
// from lane k through the LDS crossbar (ds_bpermute), which moves registers and touches no LDS bank.
//
// Backward: the pass over destination tiles (k_region_bwd) recomputes the sums, then gz, gW_out, ga = gz Wa and -
// walking the incoming edges again - gA, gWe and the A-half of gWn; it leaves ga in global memory.  The gradient of
// B[s] is a sum over the OUTGOING edges of s, so a second pass (k_region_bwd_src) walks the source-sorted view: it
// fetches ga[d], recomputes the edge's relu mask from the x table and collects the B-half of gWn and of gx.  The
// weight-gradient partials of both passes share one slab per block and k_region_reduce adds them in block order.
constexpr int kRegionMax = 2048;    // nodes of the largest island
constexpr int kRegionF = 4;         // widest x
constexpr int kRegionTiles = 4;     // tiles per block
constexpr int kRegionWaves = 2 * kRegionTiles;
constexpr int kRegionRows = kRegionTiles * kTileRows;   // rows per block
constexpr int kRegionH = 64;        // 16 lanes x 4 columns

struct RegionLayer {
  IslandLayer l;
  int32_t cap;      // rows of the x table: the largest island
  int32_t groups;   // blocks per island
  const int32_t* rowptr_t; const int32_t* col_t; const void* ea_t;   // source-sorted view
  float* ga;        // [N, H] gradient arriving at the sums
  float* aggr;      // [N, H] the sums: written by the forward, read by the backward (null: not kept / recomputed)
  float* srow;      // [N] EGNN backward: w_pos . aggr[n], from the first pass to the second
  int32_t wnb;      // backward without gx (and without coordinates): the first pass also collects the B half of gWn,
                    // and the pass over the sources is not launched
};

struct RegionBlock { int n0, rn, base; bool live; };   // island [n0, n0 + rn), the block's rows start at local row base

__device__ __forceinline__ RegionBlock region_block(const int32_t* island_ptr, int groups, int cap, int b) {
  RegionBlock r;
  const int isl = b / groups;
  r.n0 = island_ptr[isl];
  r.rn = island_ptr[isl + 1] - r.n0;
  if (r.rn > cap) r.rn = cap;   // the host checked the largest island; never index past the table
  r.base = (b - isl * groups) * kRegionRows;
  r.live = r.base < r.rn;
  return r;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 fma2(f32x2 a, float b, f32x2 c) {   // v_pk_fma_f32, b for both halves
  return __builtin_elementwise_fma(a, (f32x2)(b), c);
}

// The lane's rows of the node weight (A and B halves) and of the edge weight: columns 4 (lane % 16) + 2 p and + 1
// as pair p, so that a multiply-add serves two columns (v_pk_fma_f32).  FEW: attribute values taken per edge (1,
// or 8 with zeros past the real width); FW: values taken per row of x (3 or 4).
template <typename T, int FEW, int FW>
struct ThinWeights {
  f32x2 wa[2][FW], wb[2][FW], we[2][FEW];
  f32x2 wp[2];   // the coordinate weight (EGNN form; zeros otherwise)
  __device__ __forceinline__ void load(const IslandLayer& a, int lane) {
    const T* wn = static_cast<const T*>(a.w_nodes);
    const T* w = static_cast<const T*>(a.w_edge);
    const T* wpos = static_cast<const T*>(a.w_pos);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = 4 * (lane & 15) + j;
      wp[j >> 1][j & 1] = wpos ? scalar_to_f32<T>(wpos[c]) : 0.f;
#pragma unroll
      for (int f = 0; f < FW; ++f) {
        wa[j >> 1][f][j & 1] = f < a.F ? scalar_to_f32<T>(wn[c * a.F + f]) : 0.f;
        wb[j >> 1][f][j & 1] = f < a.F ? scalar_to_f32<T>(wn[(kRegionH + c) * a.F + f]) : 0.f;
      }
#pragma unroll
      for (int d = 0; d < FEW; ++d) we[j >> 1][d][j & 1] = d < a.Fe ? scalar_to_f32<T>(w[c * a.Fe + d]) : 0.f;
    }
  }
  // A[node][pair p of the lane] = Wn_A[c] . x in feature order; the terms past the real width multiply zeros
  __device__ __forceinline__ f32x2 a_of(int p, const float (&x)[FW]) const {
    f32x2 s = wa[p][0] * x[0];
#pragma unroll
    for (int f = 1; f < FW; ++f) s = fma2(wa[p][f], x[f], s);
    return s;
  }
  __device__ __forceinline__ f32x2 b_of(int p, const float (&x)[FW]) const {   // B[node], likewise
    f32x2 s = wb[p][0] * x[0];
#pragma unroll
    for (int f = 1; f < FW; ++f) s = fma2(wb[p][f], x[f], s);
    return s;
  }
  // the message before its relu: (A[dst] + B[src]) as in the reference (models/layers.py:128), then the edge term
  // We . e added to it one multiply-add per edge feature, in feature order.  Written out as explicit multiply-adds
  // (nothing left to the compiler's contraction), so that the same bits come out wherever it is computed and the
  // passes agree on every relu decision.
  __device__ __forceinline__ f32x2 pre(int p, f32x2 a, f32x2 b, const float (&ev)[FEW]) const {
    f32x2 t = a + b;
#pragma unroll
    for (int d = 0; d < FEW; ++d) t = fma2(we[p][d], ev[d], t);
    return t;
  }
};

// The island's rows of a [N, width <= 4] array as a table of 4 floats per node (zeros past the width).  (Eight loads
// per thread are requested before the first is stored: a loop of load-then-store would pay one trip to memory per
// iteration.)
template <typename S>
__device__ __forceinline__ void region_rows(float* xs, const S* x, int64_t ld, int width, const RegionBlock& rb) {
  const int total = rb.rn * kRegionF;
  for (int i0 = threadIdx.x; i0 < total; i0 += 8 * blockDim.x) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = i0 + k * blockDim.x;
      const int ic = i < total ? i : 0;
      const int r = ic >> 2, f = ic & 3;
      const float t = scalar_to_f32<S>(x[(int64_t)(rb.n0 + r) * ld + (f < width ? f : 0)]);
      v[k] = f < width ? t : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = i0 + k * blockDim.x;
      if (i < total) xs[i] = v[k];
    }
  }
}
template <typename T>
__device__ __forceinline__ void region_table(float* xs, const IslandLayer& a, const RegionBlock& rb) {
  region_rows<T>(xs, static_cast<const T*>(a.x), a.ldx, a.F, rb);
}

__device__ __forceinline__ float group_value(float v, int byte_addr) {   // v of the lane at byte_addr / 4
  return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_addr, __float_as_int(v)));
}
__device__ __forceinline__ int wave_max4(int v) {   // the largest of the four groups' (group-uniform) values
  const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
  const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
  const int ab = a > b ? a : b, cd = c > d ? c : d;
  return ab > cd ? ab : cd;
}
// Sum over the 16 lanes of a group, the same bits in each of them.
__device__ __forceinline__ float group_sum(float v) {
  v = dpp_add<0xB1, 0xf>(v);    // quad_perm [1, 0, 3, 2]
  v = dpp_add<0x4E, 0xf>(v);    // quad_perm [2, 3, 0, 1]
  v = dpp_add<0x141, 0xf>(v);   // row_half_mirror
  return dpp_add<0x140, 0xf>(v);   // row_mirror
}
// Sum over the four groups (lanes l, l + 16, l + 32, l + 48), the same bits in each of them.
__device__ __forceinline__ float across_groups(float v, int lane) {
  v += group_value(v, 4 * (lane ^ 16));
  return v + group_value(v, 4 * (lane ^ 32));
}

template <int FEW, int FW, bool GATHER, int NX>
struct ThinEdges {   // four consecutive edges of a group's row
  float x[4][FW], e[4][FEW];
  float p[4][NX > 0 ? NX : 1][3];   // the other end's rows of the NX further node tables (3 values each)
  float4 g[4];                      // GATHER: the lane's columns of the gradient row of the edge's other end
};

template <int FEW>
struct ThinPiece { int c[2]; float e[2][FEW]; };   // 32 edges of a group's row: lane l holds edges l and 16 + l

// Walks the edges of rows [n0, n0 + nr), nr <= 16, of a sorted view whose other ends lie in the island that starts
// at node col_base and has last + 1 nodes.  Group g of the wave takes rows g, g + 4, g + 8, g + 12: begin(r, ok)
// when its row r starts (ok: the row exists), edge(x, e, live, grad) for every edge with the other end's row of x,
// the edge's attributes, live = 1.f (0.f for the padding of a group whose row is shorter than another group's: its
// values are finite and must not count), and with GATHER the lane's four columns of gather[other end]; end(r, ok)
// when the row is complete.  The lanes of a group see the same arguments.  The index and attribute reads of the
// piece after the one being walked - of the next four rows, when this is the rows' last - are in flight meanwhile.
// NX further tables of 4 floats per node (xt0, xt1: coordinates, their gradient) hand the edge function the first
// three values of the other end's row of each as its last argument.
template <typename T, int FEW, int FW, bool GATHER, int NX = 0, typename Begin, typename Edge, typename End>
__device__ __forceinline__ void thin_walk(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                          const T* __restrict__ ea, int Fe, const float* xs, int n0, int nr, int col_base,
                                          int last, int lane, const float* gather, const float* xt0, const float* xt1,
                                          Begin begin, Edge edge, End end) {
  constexpr bool AHEAD = FEW == 1 && NX < 2;   // a second set of edge values, one group of four ahead, when it fits
  const int l = lane & 15, grp = lane >> 4;
  const int gbase = 4 * (lane & 48);   // byte address of the group's first lane for ds_bpermute
  const int rpv = rowptr[n0 + (lane < nr ? lane : nr)];   // lane i: where row i starts (i <= nr <= 16)
  const int e_end = __builtin_amdgcn_readlane(rpv, nr);   // one past the wave's last edge
  const int slots = (nr + 3) >> 2;
  struct Rows { int e0, e1, pieces; bool ok; };
  auto rows_of = [&](int slot) __attribute__((always_inline)) {
    Rows w;
    const int r = 4 * slot + grp;
    w.ok = r < nr;
    w.e0 = __builtin_amdgcn_ds_bpermute(4 * (w.ok ? r : nr), rpv);
    w.e1 = __builtin_amdgcn_ds_bpermute(4 * (w.ok ? r + 1 : nr), rpv);
    w.pieces = wave_max4((w.e1 - w.e0 + 31) >> 5);
    return w;
  };
  auto request = [&](const Rows& w, int p) __attribute__((always_inline)) {
    ThinPiece<FEW> t;
    const int eb = w.e0 + 32 * p;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const bool have = eb + 16 * h + l < w.e1;
      const int64_t e = have ? eb + 16 * h + l : e_end - 1;   // (the wave has edges, or nothing is requested)
      t.c[h] = col[e];
#pragma unroll
      for (int d = 0; d < FEW; ++d) {
        const float v = scalar_to_f32<T>(ea[e * Fe + (d < Fe ? d : 0)]);
        t.e[h][d] = (have && d < Fe) ? v : 0.f;
      }
    }
    return t;
  };
  Rows cur_rows = rows_of(0);
  ThinPiece<FEW> coming{};
  if (cur_rows.pieces > 0) coming = request(cur_rows, 0);
  for (int slot = 0; slot < slots; ++slot) {
    const Rows w = cur_rows;
    Rows next_rows = w;
    if (slot + 1 < slots) next_rows = rows_of(slot + 1);
    const int r = 4 * slot + grp;
    begin(r, w.ok);
    for (int p = 0; p < w.pieces; ++p) {
      const ThinPiece<FEW> here = coming;
      int cnt = w.e1 - (w.e0 + 32 * p);
      cnt = cnt < 0 ? 0 : (cnt > 32 ? 32 : cnt);
      float sx[2][FW], sp[2][NX > 0 ? NX : 1][3];
      int sc[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int c = here.c[h] - col_base;
        c = c < 0 ? 0 : (c > last ? last : c);
        sc[h] = c;
        const float4 xv = *reinterpret_cast<const float4*>(xs + 4 * c);
        const float xa4[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
        for (int f = 0; f < FW; ++f) sx[h][f] = xa4[f];
#pragma unroll
        for (int t = 0; t < NX; ++t) {
          const float4 pv = *reinterpret_cast<const float4*>((t == 0 ? xt0 : xt1) + 4 * c);
          sp[h][t][0] = pv.x; sp[h][t][1] = pv.y; sp[h][t][2] = pv.z;
        }
      }
      if (p + 1 < w.pieces) coming = request(w, p + 1);
      else if (slot + 1 < slots && next_rows.pieces > 0) coming = request(next_rows, 0);
      const int kmax = wave_max4(cnt);
      auto take = [&](int k0, ThinEdges<FEW, FW, GATHER, NX>& t) __attribute__((always_inline)) {
        const bool hi = k0 >= 16;   // (uniform) which of the lane's two staged edges the four come from
        const int addr = gbase + 4 * (k0 & 15);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int f = 0; f < FW; ++f) t.x[q][f] = group_value(hi ? sx[1][f] : sx[0][f], addr + 4 * q);
#pragma unroll
          for (int d = 0; d < FEW; ++d) t.e[q][d] = group_value(hi ? here.e[1][d] : here.e[0][d], addr + 4 * q);
#pragma unroll
          for (int u = 0; u < NX; ++u)
#pragma unroll
            for (int k = 0; k < 3; ++k) t.p[q][u][k] = group_value(hi ? sp[1][u][k] : sp[0][u][k], addr + 4 * q);
          if constexpr (GATHER) {
            const int node = __builtin_amdgcn_ds_bpermute(addr + 4 * q, hi ? sc[1] : sc[0]);
            t.g[q] = *reinterpret_cast<const float4*>(gather + (int64_t)node * kRegionH + 4 * l);
          }
        }
      };
      if constexpr (AHEAD) {
        ThinEdges<FEW, FW, GATHER, NX> cur, nxt;   // the next four edges' values are on their way while these are used
        take(0, cur);
        for (int k0 = 0; k0 < kmax; k0 += 4) {
          if (k0 + 4 < kmax) take(k0 + 4, nxt);
#pragma unroll
          for (int q = 0; q < 4; ++q) edge(cur.x[q], cur.e[q], k0 + q < cnt ? 1.f : 0.f, cur.g[q], cur.p[q]);
          cur = nxt;
        }
      } else {   // (wide attributes, two further tables: one set of edge values fits the registers)
        for (int k0 = 0; k0 < kmax; k0 += 4) {
          ThinEdges<FEW, FW, GATHER, NX> cur;
          take(k0, cur);
#pragma unroll
          for (int q = 0; q < 4; ++q) edge(cur.x[q], cur.e[q], k0 + q < cnt ? 1.f : 0.f, cur.g[q], cur.p[q]);
        }
      }
    }
    if (w.pieces == 0 && slot + 1 < slots && next_rows.pieces > 0) coming = request(next_rows, 0);
    end(r, w.ok);
    cur_rows = next_rows;
  }
}

// Sums of the messages into rows [row0, row0 + nr) of the island (local ids), written to sums[r * pitch + c] for
// r = 0 .. nr - 1.  EGNN: also pos'[n] = pos[n] + sum_e (pos[n] - pos[src e]) (w_pos . m_e), written to pos_out;
// sum_e rel_e (w_pos . m_e) = sum_c w_pos[c] (sum_e rel_e m_e[c]), so the lane keeps the inner sums of its columns
// and the sum over the columns is taken once per row.
template <typename T, int FEW, int FW, bool EGNN>
__device__ __forceinline__ void region_sums(const IslandLayer& a, const ThinWeights<T, FEW, FW>& tw, const float* xs,
                                            const float* ps, const RegionBlock& rb, int row0, int nr, float* sums, int pitch,
                                            int lane) {
  f32x2 av[2], sum[2], vv[2][3];
  float pr[3] = {0.f, 0.f, 0.f};
  float* const pos_out = a.pos_out;
  thin_walk<T, FEW, FW, false, EGNN ? 1 : 0>(
      a.rowptr, a.col, static_cast<const T*>(a.ea), a.Fe, xs, rb.n0 + row0, nr, rb.n0, rb.rn - 1, lane, nullptr, ps, nullptr,
      [&](int r, bool ok) __attribute__((always_inline)) {
        const float4 xr = *reinterpret_cast<const float4*>(xs + 4 * (row0 + (ok ? r : 0)));
        const float x4[4] = {xr.x, xr.y, xr.z, xr.w};
        float xv[FW];
#pragma unroll
        for (int f = 0; f < FW; ++f) xv[f] = x4[f];
#pragma unroll
        for (int p = 0; p < 2; ++p) { av[p] = tw.a_of(p, xv); sum[p] = (f32x2)(0.f); }
        if constexpr (EGNN) {
          const float4 p4 = *reinterpret_cast<const float4*>(ps + 4 * (row0 + (ok ? r : 0)));
          pr[0] = p4.x; pr[1] = p4.y; pr[2] = p4.z;
#pragma unroll
          for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int k = 0; k < 3; ++k) vv[p][k] = (f32x2)(0.f);
        }
      },
      [&](const float (&xv)[FW], const float (&ev)[FEW], float live, const float4&, const float (&ex)[1][3])
          __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const f32x2 m = __builtin_elementwise_max(tw.pre(p, av[p], tw.b_of(p, xv), ev), (f32x2)(0.f));
          if constexpr (EGNN) {
            const f32x2 ml = m * live;
            sum[p] += ml;
#pragma unroll
            for (int k = 0; k < 3; ++k) vv[p][k] = fma2(ml, pr[k] - ex[0][k], vv[p][k]);
          } else {
            sum[p] = fma2(m, live, sum[p]);
          }
        }
      },
      [&](int r, bool ok) __attribute__((always_inline)) {
        if (ok) *reinterpret_cast<float4*>(sums + r * pitch + 4 * (lane & 15)) = make_float4(sum[0][0], sum[0][1], sum[1][0], sum[1][1]);
        if constexpr (EGNN) {
          float mine = 0.f;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const f32x2 t = tw.wp[0] * vv[0][k] + tw.wp[1] * vv[1][k];
            const float v = group_sum(t[0] + t[1]);
            mine = (lane & 15) == k ? pr[k] + v : mine;
          }
          if (ok && (lane & 15) < 3) pos_out[(int64_t)(rb.n0 + row0 + r) * 3 + (lane & 15)] = mine;
        }
      });
}

constexpr int kRegionWp = kRegionF + kRegionH;   // row of the update weight in LDS: [x part, zeros up to 4 | aggr part]
// (with coordinates a second table of 4 floats per node sits behind the x table; the pass over the sources keeps a
// third, the gradient arriving at the new coordinates)
__host__ __device__ inline int64_t region_lds_floats(int cap, bool backward, bool egnn) {
  const int px = island_pitch(kRegionF + kRegionH), pz = island_pitch(kRegionH);
  return kIslandHeader + (int64_t)kRegionF * cap * (egnn ? 2 : 1) + kRegionTiles * kTileRows * (px + (backward ? pz : 0)) +
         kRegionH * kRegionWp + (backward ? 2 * kRegionH * kRegionH : 0);
}
__host__ __device__ inline int64_t region_src_lds_floats(int cap, bool egnn) {
  return kIslandHeader + (int64_t)kRegionF * cap * (egnn ? 3 : 1) + (int64_t)kRegionWaves * kRegionH * kRegionF;
}

// W_out [H, F + H] into LDS as float rows of 68 with the x part padded to 4 columns: the tile's rows [x | 0 | aggr]
// and the weight rows then line up, and both operands of the update products are read 16 bytes at a time.
template <typename T>
struct RegionWeights {
  static constexpr int total = kRegionH * kRegionWp;
  static constexpr int per = (total + 64 * kRegionWaves - 1) / (64 * kRegionWaves);
  float v[per];
  __device__ __forceinline__ void request(const IslandLayer& a) {   // all of a thread's loads go out together
    const T* w = static_cast<const T*>(a.w_out);
    const int F = a.F, K = a.F + kRegionH;
#pragma unroll
    for (int k = 0; k < per; ++k) {
      const int i = threadIdx.x + k * 64 * kRegionWaves;
      const int ic = i < total ? i : 0;
      const int c = ic / kRegionWp, kk = ic - c * kRegionWp;
      const bool pad = kk >= F && kk < kRegionF;
      const float t = scalar_to_f32<T>(w[c * K + (pad ? 0 : (kk < F ? kk : kk - (kRegionF - F)))]);
      v[k] = pad ? 0.f : t;
    }
  }
  __device__ __forceinline__ void store(float* wl) const {
#pragma unroll
    for (int k = 0; k < per; ++k) {
      const int i = threadIdx.x + k * 64 * kRegionWaves;
      if (i < total) wl[i] = v[k];
    }
  }
};

template <typename T, int FEW, int FW, bool EGNN>
__global__ void __launch_bounds__(64 * kRegionWaves) k_region_fwd(RegionLayer ra) {
  extern __shared__ __align__(16) float lds[];
  const IslandLayer& a = ra.l;
  constexpr int H = kRegionH;
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile = w >> 1, half = w & 1;
  const int px = island_pitch(kRegionF + H);
  float* xs = lds + kIslandHeader;
  float* ps = xs + kRegionF * ra.cap;                       // coordinates of the island's nodes (EGNN)
  float* tiles = ps + (EGNN ? kRegionF * ra.cap : 0);
  float* xa = tiles + tile * kTileRows * px;
  float* wl = tiles + kRegionTiles * kTileRows * px;
  const RegionBlock rb = region_block(a.island_ptr, ra.groups, ra.cap, blockIdx.x);
  if (!rb.live) return;   // whole block
  REGION_PROF(0);
  // everything the block reads before its first barrier is requested before anything is waited for
  ThinWeights<T, FEW, FW> tw;
  tw.load(a, lane);
  RegionWeights<T> wo;
  wo.request(a);
  region_table<T>(xs, a, rb);
  if constexpr (EGNN) region_rows<float>(ps, a.pos, 3, 3, rb);
  wo.store(wl);
  __syncthreads();
  REGION_PROF(1);
  const int t0 = rb.base + kTileRows * tile;                 // the tile's first row (local to the island)
  const int tn = rb.rn - t0 < kTileRows ? (rb.rn - t0 < 0 ? 0 : rb.rn - t0) : kTileRows;
  const int r0 = 16 * half;                                  // this wave's rows of the tile
  const int nr = tn - r0 < 16 ? (tn - r0 < 0 ? 0 : tn - r0) : 16;
  {
    const int r = r0 + (lane >> 2), f = lane & 3;
    xa[r * px + f] = r < tn ? xs[4 * (t0 + r) + f] : 0.f;
  }
  if (nr > 0) region_sums<T, FEW, FW, EGNN>(a, tw, xs, ps, rb, t0 + r0, nr, xa + r0 * px + kRegionF, px, lane);
  REGION_PROF(2);
  for (int i = lane; i < (16 - nr) * H; i += 64) xa[(r0 + nr + i / H) * px + kRegionF + (i & (H - 1))] = 0.f;
  if (ra.aggr)   // kept for the backward
    for (int i = lane; i < nr * (H / 4); i += 64) {
      const int r = i / (H / 4), c = 4 * (i % (H / 4));
      *reinterpret_cast<float4*>(ra.aggr + (int64_t)(rb.n0 + t0 + r0 + r) * H + c) =
          *reinterpret_cast<const float4*>(xa + (r0 + r) * px + kRegionF + c);
    }
  __syncthreads();
  REGION_PROF(3);
  if (tn == 0) return;
  // y = relu([x | aggr] W_out^T) (models/layers.py:140): the pair splits the output columns
  f32x16 acc[1];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[0][i] = 0.f;
  const int c0 = 32 * half;
  mm_wt<float, 1, true>(acc, xa, px, kRegionWp, kRegionWp, 0, wl, kRegionWp, c0, li, lh);
  REGION_PROF(4);
  T* y = static_cast<T*>(a.y);
  const int64_t ldy = a.ldy;
  const int n_first = rb.n0 + t0;
  for_acc(acc[0], lh, [&](int row, float v) __attribute__((always_inline)) {
    if (row < tn) y[(int64_t)(n_first + row) * ldy + c0 + li] = f32_to_scalar<T>(fmaxf(v, 0.f));
  });
  REGION_PROF(5);
}

template <typename T, int FEW, int FW, bool EGNN, bool WNB>
__global__ void __launch_bounds__(64 * kRegionWaves) k_region_bwd(RegionLayer ra) {
  extern __shared__ __align__(16) float lds[];
  const IslandLayer& a = ra.l;
  constexpr int H = kRegionH;
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5, l = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile = w >> 1, half = w & 1;
  const int F = a.F, px = island_pitch(kRegionF + H), pz = island_pitch(H);
  const int tile_floats = kTileRows * (px + pz), o_gz = kTileRows * px;
  int* nrows = reinterpret_cast<int*>(lds);
  int* n0s = nrows + 4;
  float* xs = lds + kIslandHeader;
  float* ps = xs + kRegionF * ra.cap;                       // coordinates of the island's nodes (EGNN)
  float* tiles = ps + (EGNN ? kRegionF * ra.cap : 0);
  float* xa = tiles + tile * tile_floats;
  float* ag = xa + kRegionF;
  float* gz = xa + o_gz;
  float* wl = tiles + kRegionTiles * tile_floats;
  float* meet = wl + kRegionH * kRegionWp;   // [2][H][H]: the two halves of gW_out's aggr part
  const RegionBlock rb = region_block(a.island_ptr, ra.groups, ra.cap, blockIdx.x);
  if (!rb.live) return;   // whole block; k_region_reduce skips its slab
  float* part = a.partial + (int64_t)blockIdx.x * a.partial_stride;
  const int64_t o_wn = (int64_t)H * (F + H), o_we = o_wn + (int64_t)2 * H * F;
  REGION_PROF(0);
  ThinWeights<T, FEW, FW> tw;
  tw.load(a, lane);
  RegionWeights<T> wo;
  wo.request(a);
  const int t0 = rb.base + kTileRows * tile;
  const int tn = rb.rn - t0 < kTileRows ? (rb.rn - t0 < 0 ? 0 : rb.rn - t0) : kTileRows;
  const int r0 = 16 * half;
  const int nr = tn - r0 < 16 ? (tn - r0 < 0 ? 0 : tn - r0) : 16;
  const int n_first = rb.n0 + t0;
  if (half == 0 && lane == 0) { nrows[tile] = tn; n0s[tile] = n_first; }
  {
    // gz = gy * [y > 0] (models/layers.py:140 backwards), this wave's 16 rows of the tile
    const T* gy = static_cast<const T*>(a.gy);
    const T* y = static_cast<const T*>(a.y);
    const int64_t ldy = a.ldy, ldgy = a.ldgy;
    const bool wide = ldy % 4 == 0 && ldgy % 4 == 0 && (((uintptr_t)gy | (uintptr_t)y) & 15) == 0;   // (uniform)
    if (wide) {   // 4 columns per lane and load, all eight loads of a lane in flight together
      float4 yv[4], gv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = lane + 64 * k, r = r0 + i / (H / 4), c = 4 * (i % (H / 4));
        const int rc = r < tn ? r : (tn > 0 ? tn - 1 : 0);
        const int64_t node = tn > 0 ? n_first + rc : rb.n0;
        yv[k] = load4<T>(y + node * ldy + c);
        gv[k] = load4<T>(gy + node * ldgy + c);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = lane + 64 * k, r = r0 + i / (H / 4), c = 4 * (i % (H / 4));
        const bool in = r < tn;
        *reinterpret_cast<float4*>(gz + r * pz + c) =
            make_float4(in && yv[k].x > 0.f ? gv[k].x : 0.f, in && yv[k].y > 0.f ? gv[k].y : 0.f,
                        in && yv[k].z > 0.f ? gv[k].z : 0.f, in && yv[k].w > 0.f ? gv[k].w : 0.f);
      }
    } else {
#pragma unroll 8
      for (int i = lane; i < 16 * H; i += 64) {
        const int r = r0 + i / H, c = i & (H - 1);
        const int rc = r < tn ? r : (tn > 0 ? tn - 1 : 0);
        const int64_t node = tn > 0 ? n_first + rc : rb.n0;
        const float yv = scalar_to_f32<T>(y[node * ldy + c]);
        const float gv = scalar_to_f32<T>(gy[node * ldgy + c]);
        gz[r * pz + c] = (r < tn && yv > 0.f) ? gv : 0.f;
      }
    }
  }
  region_table<T>(xs, a, rb);
  float gpv[3] = {0.f, 0.f, 0.f};   // EGNN: lane i holds the gradient arriving at the new coordinates of the wave's row i
  if constexpr (EGNN) {
    region_rows<float>(ps, a.pos, 3, 3, rb);
    if (nr > 0) {
      const int64_t node = n_first + r0 + ((lane & 15) < nr ? (lane & 15) : nr - 1);
#pragma unroll
      for (int q = 0; q < 3; ++q) gpv[q] = a.gpos_out[node * 3 + q];
    }
  }
  wo.store(wl);
  __syncthreads();
  REGION_PROF(1);
  {
    const int r = r0 + (lane >> 2), f = lane & 3;
    xa[r * px + f] = r < tn ? xs[4 * (t0 + r) + f] : 0.f;
  }
  if (ra.aggr) {   // the sums the forward kept
#pragma unroll 4
    for (int i = lane; i < nr * (H / 4); i += 64) {
      const int r = i / (H / 4), c = 4 * (i % (H / 4));
      *reinterpret_cast<float4*>(ag + (r0 + r) * px + c) =
          *reinterpret_cast<const float4*>(ra.aggr + (int64_t)(n_first + r0 + r) * H + c);
    }
  } else if (nr > 0) {
    region_sums<T, FEW, FW, false>(a, tw, xs, ps, rb, t0 + r0, nr, ag + r0 * px, px, lane);
  }
  for (int i = lane; i < (16 - nr) * H; i += 64) ag[(r0 + nr + i / H) * px + (i & (H - 1))] = 0.f;
  __syncthreads();
  REGION_PROF(2);
  if constexpr (EGNN && !WNB) {
    // w_pos . aggr[n] = the sum over the row's edges of w_pos . m_e: what the coordinate gradient needs of the
    // messages; handed to the pass over the sources, which writes that gradient
    for (int slot = 0; 4 * slot < nr; ++slot) {
      const int r = 4 * slot + (lane >> 4);
      const float4 a4 = *reinterpret_cast<const float4*>(ag + (r0 + (r < nr ? r : 0)) * px + 4 * l);
      const f32x2 t = tw.wp[0] * f32x2{a4.x, a4.y} + tw.wp[1] * f32x2{a4.z, a4.w};
      const float v = group_sum(t[0] + t[1]);
      if (r < nr && l == 0) ra.srow[n_first + r0 + r] = v;
    }
  }
  // gW_out's aggr part = gz^T aggr over the rows of the whole block: 2 x 2 output tiles of 32 x 32, the contraction
  // over the block's four row tiles cut in two, so that each of the 8 waves multiplies one output tile over two row
  // tiles; the halves meet in LDS.  (The x part, F columns, is collected by the edge walk below.)
  {
    const int c0 = 32 * (w & 1), col0 = 32 * ((w >> 1) & 1), pp = w >> 2;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int w2 = 2 * pp; w2 < 2 * pp + 2; ++w2) {
      const float* tb = tiles + w2 * tile_floats;
      const int nr2 = nrows[w2];
      if (nr2 == 0) continue;
      float av[16], bv[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = 2 * q + lh;
        av[q] = tb[o_gz + row * pz + c0 + li];
        bv[q] = tb[row * px + kRegionF + col0 + li];
      }
#pragma unroll
      for (int q = 0; q < 16; ++q)
        if (2 * q < nr2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], acc, 0, 0, 0);
    }
    float* st = meet + pp * H * H;
#pragma unroll
    for (int i = 0; i < 16; ++i) st[(c0 + acc_row(i, lh)) * H + col0 + li] = acc[i];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < H * H; e += blockDim.x) part[(e / H) * (F + H) + F + (e & (H - 1))] = meet[e] + meet[H * H + e];
  REGION_PROF(3);
  // ga = gz Wa (the aggr columns of W_out), each wave of the pair one tile of 32 columns; it replaces the sums in
  // the tile.  (gz Wx, the first part of gx, is formed per row in the walk.)
  if (tn > 0) {
    f32x16 acc[1];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[0][i] = 0.f;
    mm_w<float, 1>(acc, gz, pz, H, wl, kRegionWp, kRegionF + 32 * half, kRegionWp, li, lh);
    for_acc(acc[0], lh, [&](int row, float v) __attribute__((always_inline)) { ag[row * px + 32 * half + li] = v; });
  }
  f32x2 wx[2][FW];   // the lane's rows of the x part of W_out
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int f = 0; f < FW; ++f) wx[j >> 1][f][j & 1] = wl[(4 * l + j) * kRegionWp + f];
  __syncthreads();
  REGION_PROF(4);
  f32x2 gwe[2][FEW], gwa[2][FW], gwx[2][FW], gwb[2][FW], gwp[2];
  gwp[0] = gwp[1] = (f32x2)(0.f);
#pragma unroll
  for (int p = 0; p < 2; ++p) {
#pragma unroll
    for (int d = 0; d < FEW; ++d) gwe[p][d] = (f32x2)(0.f);
#pragma unroll
    for (int f = 0; f < FW; ++f) gwa[p][f] = gwx[p][f] = gwb[p][f] = (f32x2)(0.f);
  }
  constexpr bool wnb = WNB;   // no gx (and no gradient of the coordinates) wanted: gB[s] itself is not needed, only gB^T x
  if (nr > 0) {
    // ga of this wave's rows for the pass over the sources
    for (int i = lane; i < nr * (H / 4); i += 64) {
      const int r = i / (H / 4), c = 4 * (i % (H / 4));
      *reinterpret_cast<float4*>(ra.ga + (int64_t)(n_first + r0 + r) * H + c) =
          *reinterpret_cast<const float4*>(ag + (r0 + r) * px + c);
    }
    REGION_PROF(5);
    f32x2 av[2], gar[2], gzr[2], ga_sum[2];
    float xr[FW], pr[3] = {0.f, 0.f, 0.f}, gp[3] = {0.f, 0.f, 0.f};
    float* const gx_part = a.gx ? a.gx_part : nullptr;
    thin_walk<T, FEW, FW, false, EGNN ? 1 : 0>(
        a.rowptr, a.col, static_cast<const T*>(a.ea), a.Fe, xs, n_first + r0, nr, rb.n0, rb.rn - 1, lane, nullptr, ps, nullptr,
        [&](int r, bool ok) __attribute__((always_inline)) {
          const int rr = r0 + (ok ? r : 0);
          if constexpr (EGNN) {
            const float4 p4 = *reinterpret_cast<const float4*>(ps + 4 * (t0 + rr));
            pr[0] = p4.x; pr[1] = p4.y; pr[2] = p4.z;
#pragma unroll
            for (int q = 0; q < 3; ++q) gp[q] = group_value(gpv[q], 4 * (ok ? r : 0));
          }
          const float4 x4 = *reinterpret_cast<const float4*>(xs + 4 * (t0 + rr));
          const float xv[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
          for (int f = 0; f < FW; ++f) xr[f] = xv[f];
          const float4 g4 = *reinterpret_cast<const float4*>(ag + rr * px + 4 * l);
          gar[0] = f32x2{g4.x, g4.y};
          gar[1] = f32x2{g4.z, g4.w};
          const float4 z4 = *reinterpret_cast<const float4*>(gz + rr * pz + 4 * l);
          gzr[0] = f32x2{z4.x, z4.y};
          gzr[1] = f32x2{z4.z, z4.w};
#pragma unroll
          for (int p = 0; p < 2; ++p) { av[p] = tw.a_of(p, xr); ga_sum[p] = (f32x2)(0.f); }
        },
        [&](const float (&xv)[FW], const float (&ev)[FEW], float live, const float4&, const float (&ex)[1][3])
            __attribute__((always_inline)) {
          // EGNN: pos'[r] += (pos[r] - pos[s]) sc, sc = w_pos . m; q = d loss / d sc reaches the message through w_pos
          float q = 0.f;
          if constexpr (EGNN) {
            q = gp[0] * (pr[0] - ex[0][0]);
            q = fmaf(gp[1], pr[1] - ex[0][1], q);
            q = fmaf(gp[2], pr[2] - ex[0][2], q);
          }
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const f32x2 t = tw.pre(p, av[p], tw.b_of(p, xv), ev);
            const f32x2 up = EGNN ? fma2(tw.wp[p], q, gar[p]) : gar[p];
            f32x2 g;
            g[0] = t[0] > 0.f ? up[0] : 0.f;
            g[1] = t[1] > 0.f ? up[1] : 0.f;
            if constexpr (EGNN) gwp[p] = fma2(__builtin_elementwise_max(t, (f32x2)(0.f)), q * live, gwp[p]);
            ga_sum[p] = fma2(g, live, ga_sum[p]);
            if constexpr (wnb) {   // the edge's share of gWn's B half, g x[src]^T, taken here instead of in a pass by sources
              const f32x2 gl = g * live;
#pragma unroll
              for (int f = 0; f < FW; ++f) gwb[p][f] = fma2(gl, xv[f], gwb[p][f]);
            }
#pragma unroll
            for (int d = 0; d < FEW; ++d) gwe[p][d] = fma2(g, ev[d], gwe[p][d]);   // (padding carries zero attributes)
          }
        },
        [&](int r, bool ok) __attribute__((always_inline)) {
          // ga_sum = gradient of A[r]: its share of gWn and of gx[r]; gz[r]: its share of gW_out's x part and of gx[r]
          float mine = 0.f;
#pragma unroll
          for (int f = 0; f < FW; ++f) {
            f32x2 t2 = (f32x2)(0.f);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
              if (ok) {
                gwa[p][f] = fma2(ga_sum[p], xr[f], gwa[p][f]);
                gwx[p][f] = fma2(gzr[p], xr[f], gwx[p][f]);
              }
              t2 = __builtin_elementwise_fma(gzr[p], wx[p][f], t2);
              t2 = __builtin_elementwise_fma(ga_sum[p], tw.wa[p][f], t2);
            }
            if (gx_part) {
              const float v = group_sum(t2[0] + t2[1]);
              mine = l == f ? v : mine;
            }
          }
          if (gx_part && ok && l < F) gx_part[(int64_t)(n_first + r0 + r) * F + l] = mine;
        });
  }
  REGION_PROF(6);
  __syncthreads();
  REGION_PROF(7);
  // the groups' shares of gWe and of the A half of gWn are added inside the wave, then go through this wave's half
  // of the (dead) gz slab, and the block adds the waves' shares in wave order
  {
    float* slab = gz + r0 * pz;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = 4 * l + j;
#pragma unroll
      for (int d = 0; d < FEW; ++d) {
        const float v = across_groups(gwe[j >> 1][d][j & 1], lane);
        if (lane < 16 && d < a.Fe) slab[c * a.Fe + d] = v;
      }
#pragma unroll
      for (int f = 0; f < FW; ++f) {
        const float v = across_groups(gwa[j >> 1][f][j & 1], lane);
        const float u = across_groups(gwx[j >> 1][f][j & 1], lane);
        if (lane < 16 && f < F) {
          slab[H * a.Fe + c * F + f] = v;
          slab[H * (a.Fe + F) + c * F + f] = u;
        }
      }
      if constexpr (EGNN) {
        const float v = across_groups(gwp[j >> 1][j & 1], lane);
        if (lane < 16) slab[H * (a.Fe + 2 * F) + c] = v;
      }
      if constexpr (wnb) {
#pragma unroll
        for (int f = 0; f < FW; ++f) {
          const float v = across_groups(gwb[j >> 1][f][j & 1], lane);
          if (lane < 16 && f < F) slab[H * (a.Fe + 2 * F + (EGNN ? 1 : 0)) + c * F + f] = v;
        }
      }
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < H * (a.Fe + 2 * F + (EGNN ? 1 : 0) + (wnb ? F : 0)); e += blockDim.x) {
    float t = 0.f;
    for (int w2 = 0; w2 < kRegionWaves; ++w2) t += tiles[(w2 >> 1) * tile_floats + o_gz + (w2 & 1) * 16 * pz + e];
    if (e < H * a.Fe) part[o_we + e] = t;
    else if (e < H * (a.Fe + F)) part[o_wn + (e - H * a.Fe)] = t;
    else if (e < H * (a.Fe + 2 * F)) {   // gW_out[c][f], f < F
      const int i = e - H * (a.Fe + F);
      part[(i / F) * (F + H) + i % F] = t;
    } else if (EGNN && e < H * (a.Fe + 2 * F + 1)) {   // g w_pos follows gWe
      part[o_we + H * a.Fe + (e - H * (a.Fe + 2 * F))] = t;
    } else {   // the B half of gWn follows the A half
      part[o_wn + H * F + (e - H * (a.Fe + 2 * F + (EGNN ? 1 : 0)))] = t;
    }
  }
  REGION_PROF(8);
}

// The pass over the sources: 16 rows per wave.  gB[s] = sum over the edges s -> d of ga[d] where the message was
// positive; gWn's B half += gB[s] x[s]^T, gx[s] = (what the first pass left) + gB[s] Wn_B.  EGNN: the gradient of the
// coordinates, gpos[s] = gp[s] (1 + w_pos . aggr[s]) - sum over the edges s -> d of gp[d] (w_pos . m_e) with gp the
// gradient arriving at the new coordinates; the last sum by columns, as in the forward.
template <typename T, int FEW, int FW, bool EGNN>
__global__ void __launch_bounds__(64 * kRegionWaves) k_region_bwd_src(RegionLayer ra) {
  extern __shared__ __align__(16) float lds[];
  const IslandLayer& a = ra.l;
  constexpr int H = kRegionH;
  const int lane = threadIdx.x & 63, l = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int F = a.F;
  float* xs = lds + kIslandHeader;
  float* ps = xs + kRegionF * ra.cap;       // EGNN: coordinates, and behind them the gradient arriving at the new ones
  float* gps = ps + kRegionF * ra.cap;
  float* slabs = xs + kRegionF * ra.cap * (EGNN ? 3 : 1);   // [waves][H * 4]
  const RegionBlock rb = region_block(a.island_ptr, ra.groups, ra.cap, blockIdx.x);
  if (!rb.live) return;
  float* part = a.partial + (int64_t)blockIdx.x * a.partial_stride;
  const int64_t o_wnb = (int64_t)H * (F + H) + (int64_t)H * F;
  REGION_PROF(9);
  ThinWeights<T, FEW, FW> tw;
  tw.load(a, lane);
  region_table<T>(xs, a, rb);
  if constexpr (EGNN) {
    region_rows<float>(ps, a.pos, 3, 3, rb);
    region_rows<float>(gps, a.gpos_out, 3, 3, rb);
  }
  __syncthreads();
  REGION_PROF(10);
  const int s0 = rb.base + 16 * w;
  const int nr = rb.rn - s0 < 16 ? (rb.rn - s0 < 0 ? 0 : rb.rn - s0) : 16;
  f32x2 gwb[2][FW];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int f = 0; f < FW; ++f) gwb[p][f] = (f32x2)(0.f);
  if (nr > 0) {
    f32x2 gb[2], bv[2], uu[2][3];
    float xr[FW], pr[3] = {0.f, 0.f, 0.f}, gs[3] = {0.f, 0.f, 0.f}, s_row = 0.f;
    float sv = 0.f;   // EGNN: lane i holds w_pos . aggr of the wave's row i
    if constexpr (EGNN) sv = ra.srow[rb.n0 + s0 + ((lane & 15) < nr ? (lane & 15) : nr - 1)];
    float* const gpos = a.gpos;
    const float* const gx_part = a.gx_part;
    T* const gx = static_cast<T*>(a.gx);
    const int64_t ldgx = a.ldgx;
    thin_walk<T, FEW, FW, true, EGNN ? 2 : 0>(
        ra.rowptr_t, ra.col_t, static_cast<const T*>(ra.ea_t), a.Fe, xs, rb.n0 + s0, nr, rb.n0, rb.rn - 1, lane,
        ra.ga + (int64_t)rb.n0 * H, ps, gps,
        [&](int r, bool ok) __attribute__((always_inline)) {
          const float4 x4 = *reinterpret_cast<const float4*>(xs + 4 * (s0 + (ok ? r : 0)));
          const float xv[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
          for (int f = 0; f < FW; ++f) xr[f] = xv[f];
#pragma unroll
          for (int p = 0; p < 2; ++p) { bv[p] = tw.b_of(p, xr); gb[p] = (f32x2)(0.f); }
          if constexpr (EGNN) {
            const int row = s0 + (ok ? r : 0);
            const float4 p4 = *reinterpret_cast<const float4*>(ps + 4 * row);
            const float4 q4 = *reinterpret_cast<const float4*>(gps + 4 * row);
            pr[0] = p4.x; pr[1] = p4.y; pr[2] = p4.z;
            gs[0] = q4.x; gs[1] = q4.y; gs[2] = q4.z;
            s_row = group_value(sv, 4 * (ok ? r : 0));
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
              for (int q = 0; q < 3; ++q) uu[p][q] = (f32x2)(0.f);
          }
        },
        [&](const float (&xv)[FW], const float (&ev)[FEW], float live, const float4& g4, const float (&ex)[EGNN ? 2 : 1][3])
            __attribute__((always_inline)) {
          const f32x2 gd[2] = {f32x2{g4.x, g4.y}, f32x2{g4.z, g4.w}};
          // EGNN: ex[0] = pos[d], ex[1] = gp[d]; q = gp[d] . (pos[d] - pos[s]) as in the first pass
          float q = 0.f;
          if constexpr (EGNN) {
            q = ex[EGNN ? 1 : 0][0] * (ex[0][0] - pr[0]);
            q = fmaf(ex[EGNN ? 1 : 0][1], ex[0][1] - pr[1], q);
            q = fmaf(ex[EGNN ? 1 : 0][2], ex[0][2] - pr[2], q);
          }
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            // the other end is the edge's destination: A from its row of x, B from this row's - the first pass's bits
            const f32x2 t = tw.pre(p, tw.a_of(p, xv), bv[p], ev);
            const f32x2 up = EGNN ? fma2(tw.wp[p], q, gd[p]) : gd[p];
            f32x2 g;
            g[0] = t[0] > 0.f ? up[0] : 0.f;
            g[1] = t[1] > 0.f ? up[1] : 0.f;
            gb[p] = fma2(g, live, gb[p]);
            if constexpr (EGNN) {
              const f32x2 ml = __builtin_elementwise_max(t, (f32x2)(0.f)) * live;
#pragma unroll
              for (int k = 0; k < 3; ++k) uu[p][k] = fma2(ml, ex[EGNN ? 1 : 0][k], uu[p][k]);
            }
          }
        },
        [&](int r, bool ok) __attribute__((always_inline)) {
          float mine = 0.f;
#pragma unroll
          for (int f = 0; f < FW; ++f) {
            f32x2 t2 = (f32x2)(0.f);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
              if (ok) gwb[p][f] = fma2(gb[p], xr[f], gwb[p][f]);
              t2 = __builtin_elementwise_fma(gb[p], tw.wb[p][f], t2);
            }
            if (gx) {
              const float v = group_sum(t2[0] + t2[1]);
              mine = l == f ? v : mine;
            }
          }
          if (gx && ok && l < F) {
            const int64_t node = rb.n0 + s0 + r;
            gx[node * ldgx + l] = f32_to_scalar<T>(gx_part[node * F + l] + mine);
          }
          if constexpr (EGNN) {
            float out = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              const f32x2 t = tw.wp[0] * uu[0][k] + tw.wp[1] * uu[1][k];
              const float v = group_sum(t[0] + t[1]);
              out = l == k ? fmaf(gs[k], 1.f + s_row, -v) : out;
            }
            if (gpos && ok && l < 3) gpos[(int64_t)(rb.n0 + s0 + r) * 3 + l] = out;
          }
        });
  }
  REGION_PROF(11);
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int f = 0; f < FW; ++f) {
      const float v = across_groups(gwb[j >> 1][f][j & 1], lane);
      if (lane < 16 && f < F) slabs[w * H * kRegionF + (4 * l + j) * F + f] = v;
    }
  __syncthreads();
  for (int e = threadIdx.x; e < H * F; e += blockDim.x) {
    float t = 0.f;
    for (int w2 = 0; w2 < kRegionWaves; ++w2) t += slabs[w2 * H * kRegionF + e];
    part[o_wnb + e] = t;
  }
  REGION_PROF(12);
}

// Sums the live blocks' partials in a fixed order: a thread takes one element and every sixteenth block in
// ascending order, then the sixteen sums of an element are added in ascending order.
struct RegionReduce {
  const float* partial; int64_t stride; const int32_t* island_ptr; int32_t n_islands, groups, cap;
  int32_t F, H, Fe, egnn;
  void* g_wout; void* g_wn; void* g_we; void* g_wp;
};

constexpr int kReduceWays = 16;

template <typename T>
__global__ void __launch_bounds__(64 * kReduceWays) k_region_reduce(RegionReduce a) {
  __shared__ float sm[kReduceWays][64];
  const int el = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int64_t n1 = (int64_t)a.H * (a.F + a.H), n2 = (int64_t)2 * a.H * a.F, n3 = (int64_t)a.H * a.Fe;
  const int64_t total = n1 + n2 + n3 + (a.egnn ? a.H : 0);
  const int64_t e = (int64_t)blockIdx.x * 64 + el;
  float s = 0.f;
  if (e < total) {
    const int blocks = a.n_islands * a.groups;
    const float* p = a.partial + e;
#pragma unroll 4
    for (int b = g; b < blocks; b += kReduceWays) {
      const float v = p[(int64_t)b * a.stride];   // (a slab that was never written may hold anything: select)
      s += region_block(a.island_ptr, a.groups, a.cap, b).live ? v : 0.f;
    }
  }
  sm[g][el] = s;
  __syncthreads();
  if (g == 0 && e < total) {
    float t = sm[0][el];
#pragma unroll
    for (int i = 1; i < kReduceWays; ++i) t += sm[i][el];
    T* dst;
    if (e < n1) dst = a.g_wout ? static_cast<T*>(a.g_wout) + e : nullptr;
    else if (e < n1 + n2) dst = a.g_wn ? static_cast<T*>(a.g_wn) + (e - n1) : nullptr;
    else if (e < n1 + n2 + n3) dst = a.g_we ? static_cast<T*>(a.g_we) + (e - n1 - n2) : nullptr;
    else dst = a.g_wp ? static_cast<T*>(a.g_wp) + (e - n1 - n2 - n3) : nullptr;
    if (dst) *dst = f32_to_scalar<T>(t);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
// Waves (= islands) per block.  The backward keeps the rows of x out of the tile (xg) when that is what lets four
// waves share a CU's LDS; it needs 4-wide loads of x and an input width that is a multiple of the tile width.
int island_waves_per_block(int F, int H, int Fe, bool egnn, bool backward, bool xg = false) {
  const int64_t per_wave = (int64_t)island_wave_floats(F, H, Fe, egnn, backward, xg) * 4;
  const int64_t w = (kIslandLds - kIslandHeader * 4) / per_wave;
  return w > 4 ? 4 : (int)w;
}

bool island_wants_xg(int F, int H, int Fe, bool egnn) {
  return F % 32 == 0 && island_waves_per_block(F, H, Fe, egnn, true, false) < 4 &&
         island_waves_per_block(F, H, Fe, egnn, true, true) > island_waves_per_block(F, H, Fe, egnn, true, false);
}

int64_t island_lds_bytes(int F, int H, int Fe, bool egnn, bool backward, int wpb, bool xg = false) {
  return kIslandHeader * 4 + (int64_t)island_wave_floats(F, H, Fe, egnn, backward, xg) * 4 * wpb;
}

template <typename K>
void allow_lds(K kern, int kernel_id, int64_t bytes) {   // once per kernel and device, with that device current
  static bool done[128][64];
  int dev = 0;
  if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && !__atomic_load_n(&done[kernel_id][dev], __ATOMIC_ACQUIRE)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kIslandLds) ==
        hipSuccess)
      __atomic_store_n(&done[kernel_id][dev], true, __ATOMIC_RELEASE);
  }
}

template <typename T, int CPL, bool EGNN, bool VEC>
void launch_island(const IslandLayer& a, bool backward, bool xg, int blocks, int wpb, int64_t lds, hipStream_t stream) {
  const int id = (sizeof(T) == 2) | (CPL == 2) << 1 | EGNN << 2 | VEC << 3 | backward << 4 | xg << 5;
  if (backward && xg) {
    if constexpr (VEC) {
      allow_lds(k_island_bwd<T, CPL, EGNN, true, true>, id, lds);
      k_island_bwd<T, CPL, EGNN, true, true><<<blocks, 64 * wpb, lds, stream>>>(a);
    }
  } else if (backward) {
    allow_lds(k_island_bwd<T, CPL, EGNN, VEC, false>, id, lds);
    k_island_bwd<T, CPL, EGNN, VEC, false><<<blocks, 64 * wpb, lds, stream>>>(a);
  } else {
    allow_lds(k_island_fwd<T, CPL, EGNN, VEC>, id, lds);
    k_island_fwd<T, CPL, EGNN, VEC><<<blocks, 64 * wpb, lds, stream>>>(a);
  }
}

template <typename T>
void dispatch_island(const IslandLayer& a, bool egnn, bool vec, bool backward, bool xg, int blocks, int wpb, int64_t lds,
                     hipStream_t stream) {
  const bool two = a.H == 128;
#define MPNN_ISLAND_GO(CPL, EG, VE) launch_island<T, CPL, EG, VE>(a, backward, xg, blocks, wpb, lds, stream)
  if (two) {
    if (egnn) vec ? MPNN_ISLAND_GO(2, true, true) : MPNN_ISLAND_GO(2, true, false);
    else vec ? MPNN_ISLAND_GO(2, false, true) : MPNN_ISLAND_GO(2, false, false);
  } else {
    if (egnn) vec ? MPNN_ISLAND_GO(1, true, true) : MPNN_ISLAND_GO(1, true, false);
    else vec ? MPNN_ISLAND_GO(1, false, true) : MPNN_ISLAND_GO(1, false, false);
  }
#undef MPNN_ISLAND_GO
}

int32_t check_layer(const mpnn_island_layer* d, bool backward, const char* who) {
  if (!d) return fail(MPNN_E_INVALID, "%s: descriptor is null", who);
  if (d->dtype != MPNN_F32 && d->dtype != MPNN_BF16) return fail(MPNN_E_INVALID, "%s: bad dtype %d", who, d->dtype);
  if (d->H != 64 && d->H != 128) return fail(MPNN_E_INVALID, "%s: H must be 64 or 128, got %d", who, d->H);
  if (d->F < 1 || d->Fe < 1 || d->Fe > kIslandFeMax) return fail(MPNN_E_INVALID, "%s: needs F >= 1 and 1 <= Fe <= 8", who);
  if (d->num_nodes < 0 || d->num_nodes > INT32_MAX || d->n_islands < 0 || d->tail_units < 0)
    return fail(MPNN_E_INVALID, "%s: bad sizes", who);
  if (d->rows != kTileRows) return fail(MPNN_E_INVALID, "%s: rows must be %d (the height of a tile)", who, kTileRows);
  const bool egnn = d->pos != nullptr;
  if (island_waves_per_block(d->F, d->H, d->Fe, egnn, backward) < 1)
    return fail(MPNN_E_CAPACITY, "%s: a tile for F %d, H %d does not fit the LDS", who, d->F, d->H);
  if (d->n_islands + d->tail_units == 0) return MPNN_OK;
  if (!d->rowptr || !d->col || !d->island_ptr || !d->x || !d->ea || !d->w_nodes || !d->w_edge || !d->w_out || !d->y)
    return fail(MPNN_E_INVALID, "%s: null pointer argument", who);
  if (d->ldx < d->F || d->ldy < d->H) return fail(MPNN_E_INVALID, "%s: leading dimension too small", who);
  if (egnn && !d->w_pos) return fail(MPNN_E_INVALID, "%s: pos without w_pos", who);
  if (!backward && egnn && !d->pos_out) return fail(MPNN_E_INVALID, "%s: pos without pos_out", who);
  if (backward) {
    if (!d->gy || !d->partial || d->ldgy < d->H) return fail(MPNN_E_INVALID, "%s: gy / partial missing", who);
    if (d->gx && (!d->gx_part || d->ldgx < d->F)) return fail(MPNN_E_INVALID, "%s: gx needs gx_part and ldgx >= F", who);
    if (egnn && !d->gpos_out) return fail(MPNN_E_INVALID, "%s: pos without gpos_out", who);
  }
  return MPNN_OK;
}

IslandLayer to_args(const mpnn_island_layer* d) {
  IslandLayer a{};
  a.rowptr = d->rowptr; a.col = d->col; a.island_ptr = d->island_ptr;
  a.n_islands = d->n_islands; a.tail_units = d->tail_units; a.rows = d->rows; a.N = d->num_nodes;
  a.x = d->x; a.ldx = d->ldx; a.ea = d->ea; a.w_nodes = d->w_nodes; a.w_edge = d->w_edge; a.w_out = d->w_out;
  a.pos = d->pos; a.w_pos = d->w_pos; a.F = d->F; a.H = d->H; a.Fe = d->Fe;
  a.y = d->y; a.ldy = d->ldy; a.pos_out = d->pos_out;
  a.gy = d->gy; a.ldgy = d->ldgy; a.gpos_out = d->gpos_out; a.gx = d->gx; a.ldgx = d->ldgx; a.gpos = d->gpos;
  a.partial = d->partial; a.gx_part = d->gx_part;
  return a;
}

// 4-wide loads of the weights and tile rows: every pointer they touch 16-byte aligned, every row length a multiple of 4
bool island_vec(const mpnn_island_layer* d) {
  const auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
  return d->F % 4 == 0 && al(d->w_nodes) && al(d->w_out);
}


// --- regions ---
int32_t check_region(const mpnn_island_layer* d, int32_t max_nodes, bool backward, const char* who) {
  if (!d) return fail(MPNN_E_INVALID, "%s: descriptor is null", who);
  if (d->dtype != MPNN_F32 && d->dtype != MPNN_BF16) return fail(MPNN_E_INVALID, "%s: bad dtype %d", who, d->dtype);
  if (d->H != 64) return fail(MPNN_E_INVALID, "%s: H must be 64, got %d", who, d->H);
  if (d->F < 1 || d->F > kRegionF || d->Fe < 1 || d->Fe > kIslandFeMax)
    return fail(MPNN_E_INVALID, "%s: needs 1 <= F <= %d and 1 <= Fe <= 8", who, kRegionF);
  if (max_nodes < 1 || max_nodes > kRegionMax)
    return fail(MPNN_E_CAPACITY, "%s: islands of up to %d nodes, the limit is %d", who, max_nodes, kRegionMax);
  const bool egnn = d->pos != nullptr;
  if (region_lds_floats(max_nodes, true, egnn) * 4 > kIslandLds || region_src_lds_floats(max_nodes, egnn) * 4 > kIslandLds)
    return fail(MPNN_E_CAPACITY, "%s: islands of up to %d nodes%s do not fit the LDS", who, max_nodes,
                egnn ? " with coordinates" : "");
  if (d->num_nodes < 0 || d->num_nodes > INT32_MAX || d->n_islands < 0 || d->tail_units != 0 || d->rows != kTileRows)
    return fail(MPNN_E_INVALID, "%s: bad sizes (rows must be %d, tail_units 0)", who, kTileRows);
  if ((int64_t)d->n_islands * ((max_nodes + kRegionRows - 1) / kRegionRows) > INT32_MAX)
    return fail(MPNN_E_CAPACITY, "%s: too many islands", who);
  if (d->n_islands == 0) return MPNN_OK;
  if (!d->rowptr || !d->col || !d->island_ptr || !d->x || !d->ea || !d->w_nodes || !d->w_edge || !d->w_out || !d->y)
    return fail(MPNN_E_INVALID, "%s: null pointer argument", who);
  if (d->ldx < d->F || d->ldy < d->H) return fail(MPNN_E_INVALID, "%s: leading dimension too small", who);
  if (egnn && !d->w_pos) return fail(MPNN_E_INVALID, "%s: pos without w_pos", who);
  if (!backward && egnn && !d->pos_out) return fail(MPNN_E_INVALID, "%s: pos without pos_out", who);
  if (backward) {
    if (!d->gy || !d->partial || d->ldgy < d->H) return fail(MPNN_E_INVALID, "%s: gy / partial missing", who);
    if (d->gx && (!d->gx_part || d->ldgx < d->F)) return fail(MPNN_E_INVALID, "%s: gx needs gx_part and ldgx >= F", who);
    if (egnn && !d->gpos_out) return fail(MPNN_E_INVALID, "%s: pos without gpos_out", who);
  }
  return MPNN_OK;
}

RegionLayer region_args(const mpnn_island_layer* d, int32_t max_nodes) {
  RegionLayer r{};
  r.l = to_args(d);
  r.cap = max_nodes;
  r.groups = (max_nodes + kRegionRows - 1) / kRegionRows;
  return r;
}

template <typename T, int FEW, int FW, bool EGNN>
void launch_region(const RegionLayer& r, int kind, int blocks, int64_t lds, hipStream_t stream) {
  const int id = 64 + ((sizeof(T) == 2) | (FEW == 1) << 1 | (FW == 4) << 2 | kind << 3 | EGNN << 5);
  if (kind == 0) {
    allow_lds(k_region_fwd<T, FEW, FW, EGNN>, id, lds);
    k_region_fwd<T, FEW, FW, EGNN><<<blocks, 64 * kRegionWaves, lds, stream>>>(r);
  } else if (kind == 1 && r.wnb) {
    allow_lds(k_region_bwd<T, FEW, FW, EGNN, true>, id | 24, lds);   // (kind 3 in the id)
    k_region_bwd<T, FEW, FW, EGNN, true><<<blocks, 64 * kRegionWaves, lds, stream>>>(r);
  } else if (kind == 1) {
    allow_lds(k_region_bwd<T, FEW, FW, EGNN, false>, id, lds);
    k_region_bwd<T, FEW, FW, EGNN, false><<<blocks, 64 * kRegionWaves, lds, stream>>>(r);
  } else {
    allow_lds(k_region_bwd_src<T, FEW, FW, EGNN>, id, lds);
    k_region_bwd_src<T, FEW, FW, EGNN><<<blocks, 64 * kRegionWaves, lds, stream>>>(r);
  }
}

template <typename T, bool EGNN>
void dispatch_region_e(const RegionLayer& r, int kind, int blocks, int64_t lds, hipStream_t stream) {
  const bool fe1 = r.l.Fe == 1, f4 = r.l.F == 4;
  if (fe1) f4 ? launch_region<T, 1, 4, EGNN>(r, kind, blocks, lds, stream) : launch_region<T, 1, 3, EGNN>(r, kind, blocks, lds, stream);
  else f4 ? launch_region<T, kIslandFeMax, 4, EGNN>(r, kind, blocks, lds, stream)
          : launch_region<T, kIslandFeMax, 3, EGNN>(r, kind, blocks, lds, stream);
}

template <typename T>
void dispatch_region_t(const RegionLayer& r, int kind, int blocks, int64_t lds, hipStream_t stream) {
  if (r.l.pos) dispatch_region_e<T, true>(r, kind, blocks, lds, stream);
  else dispatch_region_e<T, false>(r, kind, blocks, lds, stream);
}

void dispatch_region(const RegionLayer& r, int dtype, int kind, int blocks, int64_t lds, hipStream_t stream) {
  if (dtype == MPNN_F32) dispatch_region_t<float>(r, kind, blocks, lds, stream);
  else dispatch_region_t<bf16_t>(r, kind, blocks, lds, stream);
}

}  // namespace
}  // namespace mpnn

using namespace mpnn;

#ifdef MPNN_ISLAND_PROF
extern "C" int32_t mpnn_island_prof_buffer(long long* buffer) {   // [blocks * 4 * 16] device timestamps, or null
  MPNN_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(mpnn::g_island_prof), &buffer, sizeof(buffer)));
  return MPNN_OK;
}
#endif

extern "C" {

int64_t mpnn_islands_workspace_bytes(int64_t num_nodes) {
  if (num_nodes < 0 || num_nodes > INT32_MAX) return -1;
  return 2 * align_up((num_nodes + 1) * 4, 256);
}

int32_t mpnn_islands_build(const int32_t* rowptr, const int32_t* col, int64_t num_nodes, int32_t give_up,
                           int32_t* island_ptr, int32_t* meta, void* workspace, int64_t workspace_bytes,
                           mpnn_stream_t stream) {
  if (num_nodes < 0 || num_nodes > INT32_MAX) return fail(MPNN_E_INVALID, "mpnn_islands_build: bad num_nodes");
  if (!rowptr || !island_ptr || !meta || !workspace) return fail(MPNN_E_INVALID, "mpnn_islands_build: null pointer argument");
  if (workspace_bytes < mpnn_islands_workspace_bytes(num_nodes))
    return fail(MPNN_E_CAPACITY, "mpnn_islands_build: workspace too small");
  int32_t* lo = static_cast<int32_t*>(workspace);
  int32_t* hi = lo + align_up((num_nodes + 1) * 4, 256) / 4;
  MPNN_HIP_CHECK(hipMemsetAsync(meta, 0, 4 * sizeof(int32_t), stream));
  if (num_nodes > 0) {
    int64_t blocks = (num_nodes + 255) / 256;
    if (blocks > kMaxBlocks) blocks = kMaxBlocks;
    k_island_bounds<<<(int)blocks, 256, 0, stream>>>(rowptr, col, (int)num_nodes, lo, hi, give_up, meta);
    MPNN_LAUNCH_CHECK("k_island_bounds");
  }
  k_islands_build<<<1, kScanBlock, 0, stream>>>(rowptr, (int)num_nodes, lo, hi, island_ptr, meta);
  MPNN_LAUNCH_CHECK("k_islands_build");
  return MPNN_OK;
}

int32_t mpnn_island_layer_waves(int32_t F, int32_t H, int32_t Fe, int32_t egnn, int32_t backward) {
  if (F < 1 || (H != 64 && H != 128) || Fe < 1 || Fe > kIslandFeMax) return 0;
  const int w = island_waves_per_block(F, H, Fe, egnn != 0, backward != 0);
  return w < 0 ? 0 : w;
}

int64_t mpnn_island_layer_partial_floats(int32_t n_islands, int32_t tail_units, int32_t F, int32_t H, int32_t Fe,
                                         int32_t egnn) {
  if (F < 1 || (H != 64 && H != 128) || n_islands < 0 || tail_units < 0 || Fe < 1 || Fe > kIslandFeMax) return -1;
  int64_t most = -1;
  for (int xg = 0; xg < 2; ++xg) {   // which form runs depends on the alignment of x: room for either
    const int wpb = island_waves_per_block(F, H, Fe, egnn != 0, true, xg != 0);
    if (wpb < 1) continue;
    const int64_t blocks = ((int64_t)n_islands + tail_units + wpb - 1) / wpb;
    const int64_t v = blocks * island_partial_floats(F, H, Fe, egnn != 0);
    if (v > most) most = v;
  }
  return most;
}

int32_t mpnn_island_layer_fwd(const mpnn_island_layer* d, mpnn_stream_t stream) {
  if (int32_t rc = check_layer(d, false, "mpnn_island_layer_fwd")) return rc;
  const int units = d->n_islands + d->tail_units;
  if (units == 0) return MPNN_OK;
  const bool egnn = d->pos != nullptr;
  const int wpb = island_waves_per_block(d->F, d->H, d->Fe, egnn, false);
  const int blocks = (units + wpb - 1) / wpb;
  const int64_t lds = island_lds_bytes(d->F, d->H, d->Fe, egnn, false, wpb);
  const IslandLayer a = to_args(d);
  if (d->dtype == MPNN_F32) dispatch_island<float>(a, egnn, island_vec(d), false, false, blocks, wpb, lds, stream);
  else dispatch_island<bf16_t>(a, egnn, island_vec(d), false, false, blocks, wpb, lds, stream);
  MPNN_LAUNCH_CHECK("k_island_fwd");
  return MPNN_OK;
}

int32_t mpnn_island_layer_bwd(const mpnn_island_layer* d, mpnn_stream_t stream) {
  if (int32_t rc = check_layer(d, true, "mpnn_island_layer_bwd")) return rc;
  const bool egnn = d->pos != nullptr;
  const int units = d->n_islands + d->tail_units;
  const bool vec = island_vec(d);
  // 4-wide loads of x itself (the forward copies it into the tile element by element when it must)
  const bool x_vec = vec && ((uintptr_t)d->x & 15) == 0 && d->ldx % 4 == 0;
  const bool xg = x_vec && island_wants_xg(d->F, d->H, d->Fe, egnn);
  const int wpb = island_waves_per_block(d->F, d->H, d->Fe, egnn, true, xg);
  const int blocks = (units + wpb - 1) / wpb;
  IslandLayer a = to_args(d);
  a.partial_stride = island_partial_floats(d->F, d->H, d->Fe, egnn);
  if (units > 0) {
    const int64_t lds = island_lds_bytes(d->F, d->H, d->Fe, egnn, true, wpb, xg);
    if (d->dtype == MPNN_F32) dispatch_island<float>(a, egnn, vec, true, xg, blocks, wpb, lds, stream);
    else dispatch_island<bf16_t>(a, egnn, vec, true, xg, blocks, wpb, lds, stream);
    MPNN_LAUNCH_CHECK("k_island_bwd");
  }
  IslandReduce r{d->partial, a.partial_stride, d->island_ptr, d->num_nodes, d->n_islands, d->tail_units, d->rows,
                 wpb, d->F, d->H, d->Fe, egnn ? 1 : 0, d->g_wout, d->g_wn, d->g_we, d->g_wpos};
  const int64_t total = (int64_t)d->H * (d->F + d->H) + (int64_t)2 * d->H * d->F + (int64_t)d->H * d->Fe + (egnn ? d->H : 0);
  const int rblocks = (int)((total + 63) / 64);
  if (d->dtype == MPNN_F32) k_island_reduce<float><<<rblocks, 256, 0, stream>>>(r);
  else k_island_reduce<bf16_t><<<rblocks, 256, 0, stream>>>(r);
  MPNN_LAUNCH_CHECK("k_island_reduce");
  return MPNN_OK;
}

int32_t mpnn_region_layer_fits(int32_t F, int32_t H, int32_t Fe, int32_t max_nodes, int32_t egnn) {
  return H == 64 && F >= 1 && F <= kRegionF && Fe >= 1 && Fe <= kIslandFeMax && max_nodes >= 1 && max_nodes <= kRegionMax &&
         region_lds_floats(max_nodes, true, egnn != 0) * 4 <= kIslandLds &&
         region_src_lds_floats(max_nodes, egnn != 0) * 4 <= kIslandLds;
}

int64_t mpnn_region_layer_partial_floats(int32_t n_islands, int32_t max_nodes, int32_t F, int32_t H, int32_t Fe,
                                         int32_t egnn) {
  if (!mpnn_region_layer_fits(F, H, Fe, max_nodes, egnn) || n_islands < 0) return -1;
  return (int64_t)n_islands * ((max_nodes + kRegionRows - 1) / kRegionRows) * island_partial_floats(F, H, Fe, egnn != 0);
}

int32_t mpnn_region_layer_fwd(const mpnn_island_layer* d, int32_t max_nodes, float* aggr, mpnn_stream_t stream) {
  if (int32_t rc = check_region(d, max_nodes, false, "mpnn_region_layer_fwd")) return rc;
  if (d->n_islands == 0) return MPNN_OK;
  RegionLayer r = region_args(d, max_nodes);
  r.aggr = aggr;
  dispatch_region(r, d->dtype, 0, d->n_islands * r.groups, region_lds_floats(r.cap, false, d->pos != nullptr) * 4, stream);
  MPNN_LAUNCH_CHECK("k_region_fwd");
  return MPNN_OK;
}

int32_t mpnn_region_layer_bwd(const mpnn_island_layer* d, int32_t max_nodes, const int32_t* rowptr_t,
                              const int32_t* col_t, const void* ea_t, const float* aggr, float* ga,
                              mpnn_stream_t stream) {
  if (int32_t rc = check_region(d, max_nodes, true, "mpnn_region_layer_bwd")) return rc;
  if (d->n_islands > 0 && (!rowptr_t || !col_t || !ea_t || !ga))
    return fail(MPNN_E_INVALID, "mpnn_region_layer_bwd: the source-sorted view and the ga scratch are required");
  const bool egnn = d->pos != nullptr;
  RegionLayer r = region_args(d, max_nodes);
  r.l.partial_stride = island_partial_floats(d->F, d->H, d->Fe, egnn);
  r.rowptr_t = rowptr_t; r.col_t = col_t; r.ea_t = ea_t; r.ga = ga;
  r.srow = ga + d->num_nodes * d->H;   // (read and written with coordinates only)
  r.aggr = const_cast<float*>(aggr);
  const int blocks = d->n_islands * r.groups;
  // Without gx and without the gradient of the coordinates (a model's first layer) nothing needs gB[s] itself: the
  // first pass then takes gB^T x per edge and the pass over the sources is skipped (when the wave's share fits its
  // slab: Fe + 3 F (+ 1 with coordinates) <= 17).
  r.wnb = !d->gx && !(egnn && d->gpos) && d->Fe + 3 * d->F + (egnn ? 1 : 0) <= 17;
  if (blocks > 0) {
    dispatch_region(r, d->dtype, 1, blocks, region_lds_floats(r.cap, true, egnn) * 4, stream);
    MPNN_LAUNCH_CHECK("k_region_bwd");
    if (!r.wnb) {
      dispatch_region(r, d->dtype, 2, blocks, region_src_lds_floats(r.cap, egnn) * 4, stream);
      MPNN_LAUNCH_CHECK("k_region_bwd_src");
    }
  }
  RegionReduce rr{d->partial, r.l.partial_stride, d->island_ptr, d->n_islands, r.groups, r.cap, d->F, d->H, d->Fe,
                  egnn ? 1 : 0, d->g_wout, d->g_wn, d->g_we, d->g_wpos};
  const int64_t total = (int64_t)d->H * (d->F + d->H) + (int64_t)2 * d->H * d->F + (int64_t)d->H * d->Fe + (egnn ? d->H : 0);
  const int rblocks = (int)((total + 63) / 64);
  if (d->dtype == MPNN_F32) k_region_reduce<float><<<rblocks, 64 * kReduceWays, 0, stream>>>(rr);
  else k_region_reduce<bf16_t><<<rblocks, 64 * kReduceWays, 0, stream>>>(rr);
  MPNN_LAUNCH_CHECK("k_region_reduce");
  return MPNN_OK;
}

}  // extern "C"
