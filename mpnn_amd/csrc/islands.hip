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

// ---------------------------------------------------------------------------------------------------------------
// structure: the islands of a destination-sorted CSR
// ---------------------------------------------------------------------------------------------------------------
template <typename Op>
__device__ __forceinline__ int block_scan_inclusive(int v, int* sm, Op op) {   // sm: 2 * kScanBlock ints
  const int t = threadIdx.x;
  int cur = 0;
  sm[t] = v;
  __syncthreads();
  for (int off = 1; off < kScanBlock; off <<= 1) {
    int x = sm[cur * kScanBlock + t];
    if (t >= off) x = op(sm[cur * kScanBlock + t - off], x);
    cur ^= 1;
    sm[cur * kScanBlock + t] = x;
    __syncthreads();
  }
  const int r = sm[cur * kScanBlock + t];
  __syncthreads();
  return r;
}

// A cut after node r (between r and r + 1) separates two islands iff no edge spans it: no row <= r has a source
// > r (prefix maximum of the rows' largest node id) and no row > r has a source <= r (suffix minimum of the rows'
// smallest node id).  One block: the scans run over chunks of 1024 rows with a carry.
__global__ void __launch_bounds__(kScanBlock) k_islands_build(const int32_t* __restrict__ rowptr,
                                                              const int32_t* __restrict__ col, int n,
                                                              int32_t* __restrict__ island_ptr, int32_t* __restrict__ meta,
                                                              int32_t* __restrict__ sufmin) {
  __shared__ int sm[2 * kScanBlock];
  __shared__ int carry, count, max_nodes, max_edges;
  const int t = threadIdx.x;
  const int chunks = (n + kScanBlock - 1) / kScanBlock;
  auto imin = [](int a, int b) { return a < b ? a : b; };
  auto imax = [](int a, int b) { return a > b ? a : b; };
  auto iadd = [](int a, int b) { return a + b; };
  if (t == 0) { carry = INT_MAX; count = 0; max_nodes = 0; max_edges = 0; island_ptr[0] = 0; }
  __syncthreads();
  for (int ch = chunks - 1; ch >= 0; --ch) {   // thread t takes the chunk's rows from the back: a prefix scan = suffix
    const int r = ch * kScanBlock + (kScanBlock - 1 - t);
    int lo = INT_MAX;
    if (r < n) {
      lo = r;
      for (int e = rowptr[r]; e < rowptr[r + 1]; ++e) lo = imin(lo, col[e]);
    }
    const int incl = block_scan_inclusive(lo, sm, imin);
    const int c = carry;
    sm[t] = incl;
    __syncthreads();
    const int excl = t > 0 ? imin(sm[t - 1], c) : c;   // rows after r
    if (r < n) sufmin[r] = excl;
    __syncthreads();
    if (t == kScanBlock - 1) carry = imin(incl, c);
    __syncthreads();
  }
  if (t == 0) carry = -1;
  __syncthreads();
  for (int ch = 0; ch < chunks; ++ch) {
    const int r = ch * kScanBlock + t;
    int hi = -1;
    if (r < n) {
      hi = r;
      for (int e = rowptr[r]; e < rowptr[r + 1]; ++e) hi = imax(hi, col[e]);
    }
    const int pm = imax(block_scan_inclusive(hi, sm, imax), carry);
    const int cut = r < n && pm <= r && sufmin[r] > r;
    const int pos = block_scan_inclusive(cut, sm, iadd);
    const int base = count;
    if (cut) island_ptr[base + pos] = r + 1;
    __syncthreads();
    if (t == kScanBlock - 1) { carry = pm; count = base + pos; }
    __syncthreads();
  }
  const int islands = count;
  int mn = 0, me = 0;
  for (int i = t; i < islands; i += kScanBlock) {
    const int b = island_ptr[i], e = island_ptr[i + 1];
    mn = imax(mn, e - b);
    me = imax(me, rowptr[e] - rowptr[b]);
  }
  atomicMax(&max_nodes, mn);   // integer maxima: the order does not matter
  atomicMax(&max_edges, me);
  __syncthreads();
  if (t == 0) { meta[0] = islands; meta[1] = max_nodes; meta[2] = max_edges; meta[3] = 0; }
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
#else
#define ISLAND_PROF(slot) do {} while (0)
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

__device__ __forceinline__ float wave_sum(float v) {   // the same bits in every lane
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
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

// A wave runs alone on its SIMD (the tiles fill the LDS), so nothing hides the latency of a load but the wave's
// own instruction stream: the dense loops below fetch the weights of a whole chunk of the contraction first (all
// loads in flight together, addresses clamped instead of branches) and multiply afterwards.

// acc[i][j] += sum_k in[i * ldin + col(k)] * W[(lane + 64 j) * ldw + k], k ascending: 8 rows of an LDS tile times the
// transposed weight rows of the lane's output columns (the Linear layout, models/layers.py:84-87).  Tile column of
// contraction index k: k below `split`, k + pad from there on (the tile holds [x | zero pad | aggr]).
template <typename T, int CP, bool VEC>
__device__ __forceinline__ void rows_times_wt(float (&acc)[8][CP], const float* in, int ldin, int K, int split, int pad,
                                              const T* W, int64_t ldw, int lane) {
  constexpr int Q = CP >= 4 ? 8 : 16;
  const T* wrow[CP];
#pragma unroll
  for (int j = 0; j < CP; ++j) wrow[j] = W + (int64_t)(lane + 64 * j) * ldw;
  if constexpr (VEC) {   // K, split and the row starts are multiples of 4, pad = 0
    for (int k0 = 0; k0 < K; k0 += 4 * Q) {
      float4 w[CP][Q];
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const int k = k0 + 4 * q < K ? k0 + 4 * q : K - 4;
#pragma unroll
        for (int j = 0; j < CP; ++j) w[j][q] = load4<T>(wrow[j] + k);
      }
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        if (k0 + 4 * q < K) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float4 v = *reinterpret_cast<const float4*>(in + i * ldin + k0 + 4 * q);
#pragma unroll
            for (int j = 0; j < CP; ++j) {
              acc[i][j] = fmaf(v.x, w[j][q].x, acc[i][j]);
              acc[i][j] = fmaf(v.y, w[j][q].y, acc[i][j]);
              acc[i][j] = fmaf(v.z, w[j][q].z, acc[i][j]);
              acc[i][j] = fmaf(v.w, w[j][q].w, acc[i][j]);
            }
          }
        }
      }
    }
  } else {
    for (int k0 = 0; k0 < K; k0 += Q) {
      float w[CP][Q];
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const int k = k0 + q < K ? k0 + q : K - 1;
#pragma unroll
        for (int j = 0; j < CP; ++j) w[j][q] = scalar_to_f32<T>(wrow[j][k]);
      }
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const int k = k0 + q;
        if (k < K) {
          const int xc = k < split ? k : k + pad;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float v = in[i * ldin + xc];
#pragma unroll
            for (int j = 0; j < CP; ++j) acc[i][j] = fmaf(v, w[j][q], acc[i][j]);
          }
        }
      }
    }
  }
}

// acc[i][j] += sum_c in[i * ldin + c] * W[c * ldw + col[j]], c ascending (C a multiple of 32): 8 rows of an LDS tile
// times weight columns, the lanes reading consecutive columns of one weight row.
template <typename T>
__device__ __forceinline__ void rows_times_w(float (&acc)[8][2], const float* in, int ldin, int C, const T* W,
                                             int64_t ldw, const int (&col)[2], const bool (&ok)[2]) {
  for (int c0 = 0; c0 < C; c0 += 32) {
    float w[32][2];
#pragma unroll
    for (int q = 0; q < 32; ++q)
#pragma unroll
      for (int j = 0; j < 2; ++j) w[q][j] = ok[j] ? scalar_to_f32<T>(W[(int64_t)(c0 + q) * ldw + col[j]]) : 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(in + i * ldin + c0 + 4 * q);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = fmaf(v.x, w[4 * q][j], acc[i][j]);
          acc[i][j] = fmaf(v.y, w[4 * q + 1][j], acc[i][j]);
          acc[i][j] = fmaf(v.z, w[4 * q + 2][j], acc[i][j]);
          acc[i][j] = fmaf(v.w, w[4 * q + 3][j], acc[i][j]);
        }
      }
    }
  }
}

// One task of a block-wide weight-gradient product: out[(c0 + i) * K + col] = sum over the rows of every wave's tile
// of left[row][c0 + i] * right[row][col], i < 16, col = kbase + lane + 64 j.  Right column `col` sits at tile column
// col below `split` and at col + pad from there on.  Waves in order, rows in order.
__device__ __forceinline__ void block_outer(float* out, int K, int c0, int kbase, const float* tiles, int wave_floats,
                                            const int* nrows, int wpb, int left, int ldl, int right, int ldr, int split,
                                            int pad, int lane) {
  float acc[16][2];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i][0] = acc[i][1] = 0.f;
  int off[2], col[2];
  bool ok[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    col[j] = kbase + lane + 64 * j;
    ok[j] = col[j] < K;
    off[j] = ok[j] ? right + (col[j] < split ? col[j] : col[j] + pad) : right;
  }
  for (int w2 = 0; w2 < wpb; ++w2) {
    const float* tb = tiles + w2 * wave_floats;
    const int nr = nrows[w2];
    for (int r = 0; r < nr; ++r) {
      float lv[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(tb + left + r * ldl + c0 + 4 * q);
        lv[4 * q] = v.x; lv[4 * q + 1] = v.y; lv[4 * q + 2] = v.z; lv[4 * q + 3] = v.w;
      }
      float rv[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) rv[j] = ok[j] ? tb[off[j] + r * ldr] : 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc[i][0] = fmaf(lv[i], rv[0], acc[i][0]);
        acc[i][1] = fmaf(lv[i], rv[1], acc[i][1]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (ok[j]) out[(int64_t)(c0 + i) * K + col[j]] = acc[i][j];
}

// Per-wave LDS, in floats.  xa [R][P]: the rows of x, zero columns up to a multiple of 4, then the sums over edges
// (P = Fp + H); ab [R][2H]; (backward) gz [R][H]; the staging area of the edge loops; (EGNN) pos, and in the
// backward the gradients arriving at pos' and collected for pos, 4 floats per node.
constexpr int kStageEdges = 64;
__host__ __device__ inline int island_stage_floats(int rows) { return rows + 8 + kStageEdges + kStageEdges * kIslandFeMax; }
__host__ __device__ inline int island_wave_floats(int rows, int F, int H, bool egnn, bool backward) {
  const int Fp = (F + 3) & ~3;
  return rows * (Fp + H) + rows * 2 * H + (backward ? rows * H : 0) + island_stage_floats(rows) +
         (egnn ? rows * 4 * (backward ? 3 : 1) : 0);
}

// Rows [n0, n0 + n) of x into the tile's first F columns (zero rows up to rn, zero pad columns).
template <typename T>
__device__ __forceinline__ void load_rows(float* xa, int P, const T* x, int64_t ldx, int n0, int n, int rn, int F, int Fp,
                                          int lane) {
  for (int i = lane; i < rn * Fp; i += 64) {
    const int r = i / Fp, k = i - r * Fp;
    xa[r * P + k] = (r < n && k < F) ? scalar_to_f32<T>(x[(int64_t)(n0 + r) * ldx + k]) : 0.f;
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
  __device__ __forceinline__ float term(int j, const float (&ev)[kIslandFeMax], int Fe) const {
    float s = we[j][0] * ev[0];
#pragma unroll
    for (int d = 1; d < kIslandFeMax; ++d)
      if (d < Fe) s = fmaf(we[j][d], ev[d], s);
    return s;
  }
};

// Walks the island's edges in CSR order: begin(r) when row r starts, edge(r, s, ev) for every edge into r from local
// node s with attributes ev, end(r) when the row is complete; rows without edges get begin and end too.  The row
// pointers, source ids and attributes pass through LDS in coalesced pieces of 64 edges, so that the walk itself
// waits for no global load.
template <typename T, typename Begin, typename Edge, typename End>
__device__ __forceinline__ void walk_edges(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                           const T* __restrict__ ea, int Fe, int rows, float* stage, int n0, int n, int lane,
                                           Begin begin, Edge edge, End end) {
  int* rp = reinterpret_cast<int*>(stage);
  int* ecol = rp + rows + 8;
  float* eat = reinterpret_cast<float*>(ecol + kStageEdges);
  for (int i = lane; i <= n; i += 64) rp[i] = rowptr[n0 + i];
  wave_sync();
  const int e_first = __builtin_amdgcn_readfirstlane(rp[0]), e_last = __builtin_amdgcn_readfirstlane(rp[n]);
  int r = 0;
  int row_end = __builtin_amdgcn_readfirstlane(rp[1]);
  begin(0);
  for (int eb = e_first; eb < e_last; eb += kStageEdges) {
    const int cnt = e_last - eb < kStageEdges ? e_last - eb : kStageEdges;
    if (lane < cnt) ecol[lane] = col[eb + lane] - n0;
    for (int i = lane; i < cnt * Fe; i += 64) {
      const int q = i / Fe;
      eat[q * kIslandFeMax + (i - q * Fe)] = scalar_to_f32<T>(ea[(int64_t)eb * Fe + i]);
    }
    wave_sync();
    for (int i = 0; i < cnt; ++i) {
      while (eb + i >= row_end) {
        end(r);
        ++r;
        row_end = __builtin_amdgcn_readfirstlane(rp[r + 1]);
        begin(r);
      }
      const int s = __builtin_amdgcn_readfirstlane(ecol[i]);
      float ev[kIslandFeMax];
      const float4 lo = *reinterpret_cast<const float4*>(eat + i * kIslandFeMax);
      const float4 hi = *reinterpret_cast<const float4*>(eat + i * kIslandFeMax + 4);
      ev[0] = lo.x; ev[1] = lo.y; ev[2] = lo.z; ev[3] = lo.w; ev[4] = hi.x; ev[5] = hi.y; ev[6] = hi.z; ev[7] = hi.w;
      edge(r, s, ev);
    }
    wave_sync();
  }
  for (;;) {
    end(r);
    if (++r >= n) break;
    begin(r);
  }
}

// [A | B] = x Wn^T for the tile's rows (models/layers.py:104-105).
template <typename T, int CPL, bool VEC>
__device__ __forceinline__ void project(const IslandLayer& a, const float* xa, int P, float* ab, int rn, int lane) {
  constexpr int H = 64 * CPL;
  for (int r0 = 0; r0 < rn; r0 += 8) {
    float acc[8][2 * CPL];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 2 * CPL; ++j) acc[i][j] = 0.f;
    rows_times_wt<T, 2 * CPL, VEC>(acc, xa + r0 * P, P, a.F, a.F, 0, static_cast<const T*>(a.w_nodes), a.F, lane);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 2 * CPL; ++j) ab[(r0 + i) * 2 * H + lane + 64 * j] = acc[i][j];
  }
}

// Messages of the island's edges summed per destination (models/layers.py:128 and the add-aggregation) into the
// tile's aggr columns, plus the coordinate update of the EGNN form: pos'[n] = pos[n] + sum (pos[n] - pos[src]) *
// (w_pos . m).
template <typename T, int CPL, bool EGNN>
__device__ __forceinline__ void edges_forward(const IslandLayer& a, const EdgeWeights<T, CPL, EGNN>& ew, const float* ab,
                                              float* ag, int P, const float* ps, float* stage, int n0, int n, int lane,
                                              bool write_pos) {
  constexpr int H = 64 * CPL;
  float av[CPL], sum[CPL];
  float pacc = 0.f;                      // lane c < 3: component c of the row's coordinate update
  const int pc = lane < 3 ? lane : 3;    // (the fourth float of a node's coordinates is a zero pad)
  const int Fe = a.Fe;
  float* const pos_out = a.pos_out;
  walk_edges<T>(
      a.rowptr, a.col, static_cast<const T*>(a.ea), Fe, a.rows, stage, n0, n, lane,
      [&](int r) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < CPL; ++j) { av[j] = ab[r * 2 * H + lane + 64 * j]; sum[j] = 0.f; }
        pacc = 0.f;
      },
      [&](int r, int s, const float (&ev)[kIslandFeMax]) __attribute__((always_inline)) {
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
          const float pre = (av[j] + ab[s * 2 * H + H + lane + 64 * j]) + ew.term(j, ev, Fe);
          const float m = fmaxf(pre, 0.f);
          sum[j] += m;
          if constexpr (EGNN) dot = fmaf(ew.wp[j], m, dot);
        }
        if constexpr (EGNN) {
          const float sc = wave_sum(dot);
          pacc = fmaf(ps[4 * r + pc] - ps[4 * s + pc], sc, pacc);
        }
      },
      [&](int r) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < CPL; ++j) ag[r * P + lane + 64 * j] = sum[j];
        if constexpr (EGNN) {
          if (write_pos && lane < 3) pos_out[(int64_t)(n0 + r) * 3 + lane] = ps[4 * r + lane] + pacc;
        }
      });
}

template <typename T, int CPL, bool EGNN, bool VEC>
__global__ void __launch_bounds__(256) k_island_fwd(IslandLayer a) {
  extern __shared__ __align__(16) float lds[];
  constexpr int H = 64 * CPL;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wpb = blockDim.x >> 6;
  const int F = a.F, Fp = (F + 3) & ~3, P = Fp + H, R = a.rows;
  float* xa = lds + kIslandHeader + w * island_wave_floats(R, F, H, EGNN, false);
  float* ab = xa + R * P;
  float* stage = ab + R * 2 * H;
  float* ps = stage + island_stage_floats(R);
  const Unit un = island_unit(a, blockIdx.x * wpb + w);
  if (un.n == 0) return;
  const int n = un.n, n0 = un.n0, rn = (n + 7) & ~7;
  ISLAND_PROF(0);
  load_rows<T>(xa, P, static_cast<const T*>(a.x), a.ldx, n0, n, rn, F, Fp, lane);
  if constexpr (EGNN)
    for (int i = lane; i < n * 4; i += 64) ps[i] = (i & 3) < 3 ? a.pos[(int64_t)(n0 + (i >> 2)) * 3 + (i & 3)] : 0.f;
  EdgeWeights<T, CPL, EGNN> ew;
  ew.load(a, lane);
  wave_sync();
  ISLAND_PROF(1);
  project<T, CPL, VEC>(a, xa, P, ab, rn, lane);
  wave_sync();
  ISLAND_PROF(2);
  edges_forward<T, CPL, EGNN>(a, ew, ab, xa + Fp, P, ps, stage, n0, n, lane, true);
  for (int i = lane; i < (rn - n) * H; i += 64) xa[(n + i / H) * P + Fp + (i & (H - 1))] = 0.f;
  wave_sync();
  ISLAND_PROF(3);
  // y = relu([x | aggr] W_out^T) (models/layers.py:140)
  T* y = static_cast<T*>(a.y);
  for (int r0 = 0; r0 < rn; r0 += 8) {
    float acc[8][CPL];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < CPL; ++j) acc[i][j] = 0.f;
    rows_times_wt<T, CPL, VEC>(acc, xa + r0 * P, P, F + H, F, Fp - F, static_cast<const T*>(a.w_out), F + H, lane);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (r0 + i < n) {
#pragma unroll
        for (int j = 0; j < CPL; ++j)
          y[(int64_t)(n0 + r0 + i) * a.ldy + lane + 64 * j] = f32_to_scalar<T>(fmaxf(acc[i][j], 0.f));
      }
  }
  ISLAND_PROF(4);
}

// Partial weight gradients of one block: [gW_out H x (F+H)] [gWn 2H x F] [gWe wpb x H x Fe] [g w_pos wpb x H].
__host__ __device__ inline int64_t island_partial_floats(int F, int H, int Fe, int wpb, bool egnn) {
  const int64_t v = (int64_t)H * (F + H) + (int64_t)2 * H * F + (int64_t)wpb * H * Fe + (egnn ? (int64_t)wpb * H : 0);
  return (v + 63) / 64 * 64;
}

template <typename T, int CPL, bool EGNN, bool VEC>
__global__ void __launch_bounds__(256) k_island_bwd(IslandLayer a) {
  extern __shared__ __align__(16) float lds[];
  constexpr int H = 64 * CPL;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wpb = blockDim.x >> 6;
  const int F = a.F, Fp = (F + 3) & ~3, P = Fp + H, R = a.rows;
  const int wave_floats = island_wave_floats(R, F, H, EGNN, true);
  int* nrows = reinterpret_cast<int*>(lds);
  float* tiles = lds + kIslandHeader;
  const int o_ab = R * P, o_gz = o_ab + R * 2 * H, o_stage = o_gz + R * H, o_ps = o_stage + island_stage_floats(R);
  float* xa = tiles + w * wave_floats;
  float* ab = xa + o_ab;
  float* ag = xa + Fp;     // pitch P: sums over edges, later the gradient that reaches them
  float* gz = xa + o_gz;   // gy * [y > 0], later the gradient of B
  float* stage = xa + o_stage;
  float* ps = xa + o_ps;   // pos | gradient of pos' | gradient of pos (EGNN)
  float* gp = ps + 4 * R;
  float* gq = gp + 4 * R;
  float* part = a.partial + (int64_t)blockIdx.x * a.partial_stride;
  const int64_t o_wn = (int64_t)H * (F + H), o_we = o_wn + (int64_t)2 * H * F, o_wp = o_we + (int64_t)wpb * H * a.Fe;

  if ((int)blockIdx.x * wpb >= island_active_units(a.island_ptr, a.n_islands, a.tail_units, R, a.N)) return;   // whole block
  const Unit un = island_unit(a, blockIdx.x * wpb + w);
  const int n = un.n, n0 = un.n0, rn = (n + 7) & ~7;
  if (lane == 0) nrows[w] = n;
  ISLAND_PROF(0);
  EdgeWeights<T, CPL, EGNN> ew;
  ew.load(a, lane);
  if (n > 0) {
    load_rows<T>(xa, P, static_cast<const T*>(a.x), a.ldx, n0, n, rn, F, Fp, lane);
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
    for (int i = lane; i < rn * H; i += 64) {
      const int r = i / H, c = i & (H - 1);
      float g = 0.f;
      if (r < n && scalar_to_f32<T>(y[(int64_t)(n0 + r) * a.ldy + c]) > 0.f)
        g = scalar_to_f32<T>(gy[(int64_t)(n0 + r) * a.ldgy + c]);
      gz[i] = g;
    }
    wave_sync();
    ISLAND_PROF(1);
    project<T, CPL, VEC>(a, xa, P, ab, rn, lane);
    wave_sync();
    ISLAND_PROF(2);
    edges_forward<T, CPL, EGNN>(a, ew, ab, ag, P, ps, stage, n0, n, lane, false);
    for (int i = lane; i < (rn - n) * H; i += 64) ag[(n + i / H) * P + (i & (H - 1))] = 0.f;
    ISLAND_PROF(3);
  }
  __syncthreads();
  ISLAND_PROF(4);
  // gW_out = gz^T [x | aggr] over the rows of the whole block
  {
    const int kgroups = (F + H + 127) / 128;
    const int tasks = (H / 16) * kgroups;
    for (int t = w; t < tasks; t += wpb)
      block_outer(part, F + H, 16 * (t / kgroups), 128 * (t % kgroups), tiles, wave_floats, nrows, wpb, o_gz, H, 0, P, F,
                  Fp - F, lane);
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
    const T* wo = static_cast<const T*>(a.w_out);
    const int kfirst = a.gx ? 0 : F;
    for (int kb = kfirst; kb < F + H; kb += 128) {
      int col[2];
      bool ok[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) { col[j] = kb + lane + 64 * j; ok[j] = col[j] < F + H; }
      for (int r0 = 0; r0 < rn; r0 += 8) {
        float acc[8][2];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i][0] = acc[i][1] = 0.f;
        rows_times_w<T>(acc, gz + r0 * H, H, H, wo, F + H, col, ok);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if (!ok[j]) continue;
            if (col[j] >= F) ag[(r0 + i) * P + col[j] - F] = acc[i][j];
            else if (r0 + i < n) a.gx_part[(int64_t)(n0 + r0 + i) * F + col[j]] = acc[i][j];
          }
      }
    }
    wave_sync();
    ISLAND_PROF(7);
    // gz is dead: its slab collects the gradient of B
    for (int i = lane; i < rn * H; i += 64) gz[i] = 0.f;
    wave_sync();
    float av[CPL], gar[CPL], ga_sum[CPL];
    float gpx = 0.f, gpy = 0.f, gpz = 0.f, sc_sum = 0.f;
    const int Fe = a.Fe;
    walk_edges<T>(
        a.rowptr, a.col, static_cast<const T*>(a.ea), Fe, R, stage, n0, n, lane,
        [&](int r) __attribute__((always_inline)) {
#pragma unroll
          for (int j = 0; j < CPL; ++j) {
            av[j] = ab[r * 2 * H + lane + 64 * j];
            gar[j] = ag[r * P + lane + 64 * j];
            ga_sum[j] = 0.f;
          }
          sc_sum = 0.f;
          if constexpr (EGNN) { gpx = gp[4 * r]; gpy = gp[4 * r + 1]; gpz = gp[4 * r + 2]; }
        },
        [&](int r, int s, const float (&ev)[kIslandFeMax]) __attribute__((always_inline)) {
          float pre[CPL], m[CPL], dot = 0.f;
#pragma unroll
          for (int j = 0; j < CPL; ++j) {
            pre[j] = (av[j] + ab[s * 2 * H + H + lane + 64 * j]) + ew.term(j, ev, Fe);
            m[j] = fmaxf(pre[j], 0.f);
            if constexpr (EGNN) dot = fmaf(ew.wp[j], m[j], dot);
          }
          float q = 0.f;
          if constexpr (EGNN) {
            // pos'[r] += (pos[r] - pos[s]) * sc, sc = w_pos . m: q = d loss / d sc
            const float sc = wave_sum(dot);
            q = gpx * (ps[4 * r] - ps[4 * s]);
            q = fmaf(gpy, ps[4 * r + 1] - ps[4 * s + 1], q);
            q = fmaf(gpz, ps[4 * r + 2] - ps[4 * s + 2], q);
            sc_sum += sc;
            if (lane < 3) gq[4 * s + lane] -= gp[4 * r + lane] * sc;
          }
#pragma unroll
          for (int j = 0; j < CPL; ++j) {
            float g = EGNN ? fmaf(q, ew.wp[j], gar[j]) : gar[j];
            g = pre[j] > 0.f ? g : 0.f;
            ga_sum[j] += g;
            gz[s * H + lane + 64 * j] += g;
            if constexpr (EGNN) gwp[j] = fmaf(q, m[j], gwp[j]);
#pragma unroll
            for (int d = 0; d < kIslandFeMax; ++d)
              if (d < Fe) gwe[j][d] = fmaf(g, ev[d], gwe[j][d]);
          }
        },
        [&](int r) __attribute__((always_inline)) {
#pragma unroll
          for (int j = 0; j < CPL; ++j) ab[r * 2 * H + lane + 64 * j] = ga_sum[j];   // A[r] is dead: gradient of A[r]
          if constexpr (EGNN) {
            if (lane < 3) gq[4 * r + lane] += gp[4 * r + lane] * (1.f + sc_sum);
          }
        });
    wave_sync();
    for (int i = lane; i < rn * H; i += 64) ab[(i / H) * 2 * H + H + (i & (H - 1))] = gz[i];
    if constexpr (EGNN)
      for (int i = lane; i < n * 3; i += 64) a.gpos[(int64_t)n0 * 3 + i] = gq[4 * (i / 3) + i % 3];
    wave_sync();
    ISLAND_PROF(8);
    if (a.gx) {   // gx = gz Wx + [gA | gB] Wn
      const T* wn = static_cast<const T*>(a.w_nodes);
      T* gx = static_cast<T*>(a.gx);
      for (int kb = 0; kb < F; kb += 128) {
        int col[2];
        bool ok[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) { col[j] = kb + lane + 64 * j; ok[j] = col[j] < F; }
        for (int r0 = 0; r0 < rn; r0 += 8) {
          float acc[8][2];
#pragma unroll
          for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = (ok[j] && r0 + i < n) ? a.gx_part[(int64_t)(n0 + r0 + i) * F + col[j]] : 0.f;
          rows_times_w<T>(acc, ab + r0 * 2 * H, 2 * H, 2 * H, wn, F, col, ok);
#pragma unroll
          for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              if (ok[j] && r0 + i < n) gx[(int64_t)(n0 + r0 + i) * a.ldgx + col[j]] = f32_to_scalar<T>(acc[i][j]);
        }
      }
    }
  }
  ISLAND_PROF(9);
  // the lane's rows of gWe and g w_pos: one slot per wave
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
#pragma unroll
    for (int d = 0; d < kIslandFeMax; ++d)
      if (d < a.Fe) part[o_we + ((int64_t)w * H + lane + 64 * j) * a.Fe + d] = gwe[j][d];
    if constexpr (EGNN) part[o_wp + (int64_t)w * H + lane + 64 * j] = gwp[j];
  }
  __syncthreads();
  ISLAND_PROF(10);
  // gWn = [gA | gB]^T x over the rows of the whole block
  {
    const int kgroups = (F + 127) / 128;
    const int tasks = (2 * H / 16) * kgroups;
    for (int t = w; t < tasks; t += wpb)
      block_outer(part + o_wn, F, 16 * (t / kgroups), 128 * (t % kgroups), tiles, wave_floats, nrows, wpb, o_ab, 2 * H, 0, P,
                  F, 0, lane);
  }
  ISLAND_PROF(11);
}

// Sums the blocks' partials in a fixed order: 16 lanes per element take every 16th partial in ascending order, then
// lane group 0 adds the 16 sums in ascending order.
struct IslandReduce {
  const float* partial; int64_t stride; const int32_t* island_ptr; int64_t N; int32_t n_islands, tail_units, rows;
  int32_t wpb, F, H, Fe, egnn;
  void* g_wout; void* g_wn; void* g_we; void* g_wp;
};

template <typename T>
__global__ void __launch_bounds__(256) k_island_reduce(IslandReduce a) {
  __shared__ float sm[16][17];
  const int el = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int64_t n1 = (int64_t)a.H * (a.F + a.H), n2 = (int64_t)2 * a.H * a.F, n3 = (int64_t)a.H * a.Fe;
  const int64_t total = n1 + n2 + n3 + (a.egnn ? a.H : 0);
  const int64_t e = (int64_t)blockIdx.x * 16 + el;
  float s = 0.f;
  T* dst = nullptr;
  if (e < total) {
    int64_t off, slot_stride = 0;
    int slots = 1;
    if (e < n1) { off = e; dst = a.g_wout ? static_cast<T*>(a.g_wout) + e : nullptr; }
    else if (e < n1 + n2) { off = e; dst = a.g_wn ? static_cast<T*>(a.g_wn) + (e - n1) : nullptr; }
    else if (e < n1 + n2 + n3) {
      off = e; slots = a.wpb; slot_stride = n3;
      dst = a.g_we ? static_cast<T*>(a.g_we) + (e - n1 - n2) : nullptr;
    } else {
      off = n1 + n2 + (int64_t)a.wpb * n3 + (e - n1 - n2 - n3); slots = a.wpb; slot_stride = a.H;
      dst = a.g_wp ? static_cast<T*>(a.g_wp) + (e - n1 - n2 - n3) : nullptr;
    }
    if (dst) {
      const int units = (a.n_islands + a.tail_units) > 0
                            ? island_active_units(a.island_ptr, a.n_islands, a.tail_units, a.rows, a.N) : 0;
      const int64_t slabs = (int64_t)((units + a.wpb - 1) / a.wpb) * slots;   // the blocks that wrote a partial
      for (int64_t i = g; i < slabs; i += 16)
        s += a.partial[(i / slots) * a.stride + off + (i % slots) * slot_stride];
    }
  }
  sm[g][el] = s;
  __syncthreads();
  if (g == 0 && dst) {
    float t = sm[0][el];
#pragma unroll
    for (int i = 1; i < 16; ++i) t += sm[i][el];
    *dst = f32_to_scalar<T>(t);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
int island_waves_per_block(int rows, int F, int H, bool egnn, bool backward) {
  const int64_t per_wave = (int64_t)island_wave_floats(rows, F, H, egnn, backward) * 4;
  const int64_t w = (kIslandLds - kIslandHeader * 4) / per_wave;
  return w > 4 ? 4 : (int)w;
}

int64_t island_lds_bytes(int rows, int F, int H, bool egnn, bool backward, int wpb) {
  return kIslandHeader * 4 + (int64_t)island_wave_floats(rows, F, H, egnn, backward) * 4 * wpb;
}

template <typename K>
void allow_lds(K kern, int kernel_id, int64_t bytes) {   // once per kernel and device, with that device current
  static bool done[64][64];
  int dev = 0;
  if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && !__atomic_load_n(&done[kernel_id][dev], __ATOMIC_ACQUIRE)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kIslandLds) ==
        hipSuccess)
      __atomic_store_n(&done[kernel_id][dev], true, __ATOMIC_RELEASE);
  }
}

template <typename T, int CPL, bool EGNN, bool VEC>
void launch_island(const IslandLayer& a, bool backward, int blocks, int wpb, int64_t lds, hipStream_t stream) {
  const int id = (sizeof(T) == 2) | (CPL == 2) << 1 | EGNN << 2 | VEC << 3 | backward << 4;
  if (backward) {
    allow_lds(k_island_bwd<T, CPL, EGNN, VEC>, id, lds);
    k_island_bwd<T, CPL, EGNN, VEC><<<blocks, 64 * wpb, lds, stream>>>(a);
  } else {
    allow_lds(k_island_fwd<T, CPL, EGNN, VEC>, id, lds);
    k_island_fwd<T, CPL, EGNN, VEC><<<blocks, 64 * wpb, lds, stream>>>(a);
  }
}

template <typename T>
void dispatch_island(const IslandLayer& a, bool egnn, bool vec, bool backward, int blocks, int wpb, int64_t lds,
                     hipStream_t stream) {
  const bool two = a.H == 128;
#define MPNN_ISLAND_GO(CPL, EG, VE) launch_island<T, CPL, EG, VE>(a, backward, blocks, wpb, lds, stream)
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
  if (d->rows < 8 || d->rows % 8) return fail(MPNN_E_INVALID, "%s: rows must be a positive multiple of 8", who);
  const bool egnn = d->pos != nullptr;
  if (island_waves_per_block(d->rows, d->F, d->H, egnn, backward) < 1)
    return fail(MPNN_E_CAPACITY, "%s: a tile of %d rows (F %d, H %d) does not fit the LDS", who, d->rows, d->F, d->H);
  if (d->n_islands + d->tail_units == 0) return MPNN_OK;
  if (!d->rowptr || !d->col || !d->island_ptr || !d->x || !d->ea || !d->w_nodes || !d->w_edge || !d->w_out || !d->y)
    return fail(MPNN_E_INVALID, "%s: null pointer argument", who);
  if (d->ldx < d->F || d->ldy < d->H) return fail(MPNN_E_INVALID, "%s: leading dimension too small", who);
  if (egnn && !d->w_pos) return fail(MPNN_E_INVALID, "%s: pos without w_pos", who);
  if (!backward && egnn && !d->pos_out) return fail(MPNN_E_INVALID, "%s: pos without pos_out", who);
  if (backward) {
    if (!d->gy || !d->partial || d->ldgy < d->H) return fail(MPNN_E_INVALID, "%s: gy / partial missing", who);
    if (d->gx && (!d->gx_part || d->ldgx < d->F)) return fail(MPNN_E_INVALID, "%s: gx needs gx_part and ldgx >= F", who);
    if (egnn && (!d->gpos_out || !d->gpos)) return fail(MPNN_E_INVALID, "%s: pos without gpos_out / gpos", who);
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
  return align_up((num_nodes + 1) * 4, 256);
}

int32_t mpnn_islands_build(const int32_t* rowptr, const int32_t* col, int64_t num_nodes, int32_t* island_ptr,
                           int32_t* meta, void* workspace, int64_t workspace_bytes, mpnn_stream_t stream) {
  if (num_nodes < 0 || num_nodes > INT32_MAX) return fail(MPNN_E_INVALID, "mpnn_islands_build: bad num_nodes");
  if (!rowptr || !island_ptr || !meta || !workspace) return fail(MPNN_E_INVALID, "mpnn_islands_build: null pointer argument");
  if (workspace_bytes < mpnn_islands_workspace_bytes(num_nodes))
    return fail(MPNN_E_CAPACITY, "mpnn_islands_build: workspace too small");
  k_islands_build<<<1, kScanBlock, 0, stream>>>(rowptr, col, (int)num_nodes, island_ptr, meta, static_cast<int32_t*>(workspace));
  MPNN_LAUNCH_CHECK("k_islands_build");
  return MPNN_OK;
}

int32_t mpnn_island_layer_waves(int32_t rows, int32_t F, int32_t H, int32_t egnn, int32_t backward) {
  if (rows < 8 || rows % 8 || F < 1 || (H != 64 && H != 128)) return 0;
  const int w = island_waves_per_block(rows, F, H, egnn != 0, backward != 0);
  return w < 0 ? 0 : w;
}

int64_t mpnn_island_layer_partial_floats(int32_t n_islands, int32_t tail_units, int32_t rows, int32_t F, int32_t H,
                                         int32_t Fe, int32_t egnn) {
  const int wpb = mpnn_island_layer_waves(rows, F, H, egnn, 1);
  if (wpb < 1 || n_islands < 0 || tail_units < 0 || Fe < 1 || Fe > kIslandFeMax) return -1;
  const int64_t blocks = ((int64_t)n_islands + tail_units + wpb - 1) / wpb;
  return blocks * island_partial_floats(F, H, Fe, wpb, egnn != 0);
}

int32_t mpnn_island_layer_fwd(const mpnn_island_layer* d, mpnn_stream_t stream) {
  if (int32_t rc = check_layer(d, false, "mpnn_island_layer_fwd")) return rc;
  const int units = d->n_islands + d->tail_units;
  if (units == 0) return MPNN_OK;
  const bool egnn = d->pos != nullptr;
  const int wpb = island_waves_per_block(d->rows, d->F, d->H, egnn, false);
  const int blocks = (units + wpb - 1) / wpb;
  const int64_t lds = island_lds_bytes(d->rows, d->F, d->H, egnn, false, wpb);
  const IslandLayer a = to_args(d);
  if (d->dtype == MPNN_F32) dispatch_island<float>(a, egnn, island_vec(d), false, blocks, wpb, lds, stream);
  else dispatch_island<bf16_t>(a, egnn, island_vec(d), false, blocks, wpb, lds, stream);
  MPNN_LAUNCH_CHECK("k_island_fwd");
  return MPNN_OK;
}

int32_t mpnn_island_layer_bwd(const mpnn_island_layer* d, mpnn_stream_t stream) {
  if (int32_t rc = check_layer(d, true, "mpnn_island_layer_bwd")) return rc;
  const bool egnn = d->pos != nullptr;
  const int units = d->n_islands + d->tail_units;
  const int wpb = island_waves_per_block(d->rows, d->F, d->H, egnn, true);
  const int blocks = (units + wpb - 1) / wpb;
  IslandLayer a = to_args(d);
  a.partial_stride = island_partial_floats(d->F, d->H, d->Fe, wpb, egnn);
  if (units > 0) {
    const int64_t lds = island_lds_bytes(d->rows, d->F, d->H, egnn, true, wpb);
    if (d->dtype == MPNN_F32) dispatch_island<float>(a, egnn, island_vec(d), true, blocks, wpb, lds, stream);
    else dispatch_island<bf16_t>(a, egnn, island_vec(d), true, blocks, wpb, lds, stream);
    MPNN_LAUNCH_CHECK("k_island_bwd");
  }
  IslandReduce r{d->partial, a.partial_stride, d->island_ptr, d->num_nodes, d->n_islands, d->tail_units, d->rows,
                 wpb, d->F, d->H, d->Fe, egnn ? 1 : 0, d->g_wout, d->g_wn, d->g_we, d->g_wpos};
  const int64_t total = (int64_t)d->H * (d->F + d->H) + (int64_t)2 * d->H * d->F + (int64_t)d->H * d->Fe + (egnn ? d->H : 0);
  const int rblocks = (int)((total + 15) / 16);
  if (d->dtype == MPNN_F32) k_island_reduce<float><<<rblocks, 256, 0, stream>>>(r);
  else k_island_reduce<bf16_t><<<rblocks, 256, 0, stream>>>(r);
  MPNN_LAUNCH_CHECK("k_island_reduce");
  return MPNN_OK;
}

}  // extern "C"
