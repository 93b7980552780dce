// Device helpers shared by the edge kernels: per-lane edge-projection weights, edge-attribute
// loads, the W_e . e fma chain, relu with the reference's NaN behaviour, and launch geometry.
#pragma once
#include "common.h"
#include "rowio.h"

namespace mpnn {

constexpr int kMaxFe = 8;  // edge-attribute width handled by the generic (runtime Fe) instantiation

// VEC floats to a float row; 16-byte stores when the lane owns a multiple of 4 columns.
template <int VEC>
__device__ __forceinline__ void store_f32(float* p, const float (&v)[VEC]) {
  if constexpr (VEC % 4 == 0) {
#pragma unroll
    for (int q = 0; q < VEC / 4; ++q)
      reinterpret_cast<float4*>(p)[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
  } else {
#pragma unroll
    for (int q = 0; q < VEC; ++q) p[q] = v[q];
  }
}

// relu as the reference computes it: NaN propagates, 0 for x <= 0.
__device__ __forceinline__ float relu_f(float x) { return x <= 0.f ? 0.f : x; }

// Per-lane weights of the edge projection: w[v][f] = We[c0 + v][f].
template <typename T, int VEC, int FE>
__device__ __forceinline__ void load_edge_weights(const T* We, int fe, int c0, bool act, float (&w)[VEC][FE]) {
#pragma unroll
  for (int v = 0; v < VEC; ++v)
#pragma unroll
    for (int f = 0; f < FE; ++f)
      w[v][f] = (act && f < fe) ? scalar_to_f32<T>(We[(int64_t)(c0 + v) * fe + f]) : 0.f;
}

template <typename T, int FE, bool RTFE>
__device__ __forceinline__ void load_edge_attr(const T* ea, int64_t p, int fe, float (&e)[FE]) {
  if constexpr (RTFE) {
#pragma unroll
    for (int f = 0; f < FE; ++f) e[f] = f < fe ? scalar_to_f32<T>(ea[p * fe + f]) : 0.f;
  } else {
    EdgeIO<T, FE>::load(ea + p * FE, e);
  }
}

// ce[v] = sum_f w[v][f] * e[f], as an explicit fma chain in f order.
template <int VEC, int FE>
__device__ __forceinline__ void edge_term(const float (&w)[VEC][FE], const float (&e)[FE], float (&ce)[VEC]) {
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    float s = w[v][0] * e[0];
#pragma unroll
    for (int f = 1; f < FE; ++f) s = fmaf(w[v][f], e[f], s);
    ce[v] = s;
  }
}


// Block-level sum of per-lane partials gw[v][f] (lane owns columns c0..c0+VEC-1) in a fixed order:
// groups of a wave by ds_bpermute in group order, then the 4 waves through LDS in wave order.
// Writes partial[(blockIdx.x * H + c) * fe + f].  lds must hold 4 * tile * FE floats.
template <int VEC, int FE>
__device__ __forceinline__ void block_reduce_lane_partials(const float (&gw)[VEC][FE], float* lds, float* partial,
                                                           int H, int fe, int lpr_log2) {
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  const int lpr = 1 << lpr_log2;
  const int lig = lane & (lpr - 1);
  const int grp = lane >> lpr_log2;
  const int gpw = 64 >> lpr_log2;
  const int tile = lpr * VEC;
#pragma unroll
  for (int v = 0; v < VEC; ++v)
#pragma unroll
    for (int f = 0; f < FE; ++f) {
      float tot = 0.f;
      for (int g = 0; g < gpw; ++g) tot += __shfl(gw[v][f], g * lpr + lig, 64);
      if (grp == 0) lds[(wid * tile + lig * VEC + v) * FE + f] = tot;
    }
  __syncthreads();
  for (int i = threadIdx.x; i < tile * FE; i += kBlock) {
    const int c = blockIdx.y * tile + i / FE;
    const int f = i % FE;
    if (c < H && f < fe) {
      float tot = 0.f;
      for (int k = 0; k < kBlock / 64; ++k) tot += lds[k * tile * FE + i];
      partial[((int64_t)blockIdx.x * H + c) * fe + f] = tot;
    }
  }
}

// Combine the pieces of split rows in piece order.  One wave per split row: group g adds
// pieces g, g + gpw, ... in order, then the groups' sums are added in group order.
template <typename T, int VEC>
__global__ void __launch_bounds__(kBlock) k_reduce_split(const float* __restrict__ parts, const int4* __restrict__ split,
                                                         int64_t n_split, T* __restrict__ out, int64_t ldo, int H,
                                                         int lpr_log2, bool by_slot) {
  const int lane = threadIdx.x & 63;
  const int lpr = 1 << lpr_log2;
  const int lig = lane & (lpr - 1);
  const int grp = lane >> lpr_log2;
  const int gpw = 64 >> lpr_log2;
  const int c0 = blockIdx.y * (lpr * VEC) + lig * VEC;
  const bool act = c0 < H;
  const int cc = act ? c0 : 0;
  const int64_t wave = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (kBlock / 64);
  for (int64_t s = wave; s < n_split; s += nwaves) {
    const int4 sr = split[s];
    const int64_t row = by_slot ? s : (int64_t)sr.x;   // by_slot: a table with one row per split record
    const int poff = sr.y, pieces = sr.z;
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
    for (int c = grp; c < pieces; c += gpw) {
      const float* p = parts + (int64_t)(poff + c) * H + cc;
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[v] += p[v];
    }
    float tot[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) tot[v] = 0.f;
    for (int g = 0; g < gpw; ++g) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) tot[v] += __shfl(acc[v], g * lpr + lig, 64);
    }
    if (grp == 0 && act) {
      T* o = out + row * ldo + c0;
#pragma unroll
      for (int v = 0; v < VEC; ++v) o[v] = f32_to_scalar<T>(tot[v]);
    }
  }
}

// gWe[i] = sum over blocks of partial[b][i], b ascending: 256 threads take strided subsets
// sequentially, then a fixed-shape tree in LDS.
template <typename T>
__global__ void __launch_bounds__(kBlock) k_reduce_gwe(const float* __restrict__ partial, int nblocks, int count,
                                                       T* __restrict__ gWe) {
  __shared__ float red[kBlock];
  const int i = blockIdx.x;
  float s = 0.f;
  for (int b = threadIdx.x; b < nblocks; b += kBlock) s += partial[(int64_t)b * count + i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = kBlock / 2; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) gWe[i] = f32_to_scalar<T>(red[0]);
}


// Host-side launch geometry of the row kernels.
struct Geometry {
  int vec;       // columns per lane
  int lpr_log2;  // log2 lanes per row group
  int tiles;     // column tiles (gridDim.y)
};

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// Chooses the vector path when every row access of every listed matrix is 16-byte aligned.
inline Geometry geometry(int H, int dtype, bool vec_ok) {
  Geometry g;
  const int full = dtype == MPNN_BF16 ? 8 : 4;
  g.vec = (vec_ok && H % full == 0) ? full : 1;
  int lanes = (H + g.vec - 1) / g.vec;
  int k = 0;
  while ((1 << k) < lanes && k < 6) ++k;
  g.lpr_log2 = k;
  const int tile = (1 << k) * g.vec;
  g.tiles = (H + tile - 1) / tile;
  return g;
}

inline int row_grid(int64_t n_items, int lpr_log2, int64_t max_blocks = kMaxBlocks) {
  const int per_block = (kBlock / 64) * (64 >> lpr_log2);
  int64_t g = (n_items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > max_blocks) g = max_blocks;
  return (int)g;
}

template <typename T, int VEC>
inline int32_t launch_reduce_split(const float* parts, const int32_t* split, int64_t n_split, T* out, int64_t ldo,
                                   int H, const Geometry& g, hipStream_t stream, bool by_slot = false) {
  int64_t blocks = (n_split + 3) / 4;
  if (blocks > kMaxBlocks) blocks = kMaxBlocks;
  k_reduce_split<T, VEC><<<dim3((unsigned)blocks, g.tiles), kBlock, 0, stream>>>(
      parts, reinterpret_cast<const int4*>(split), n_split, out, ldo, H, g.lpr_log2, by_slot);
  MPNN_LAUNCH_CHECK("k_reduce_split");
  return MPNN_OK;
}

}  // namespace mpnn
