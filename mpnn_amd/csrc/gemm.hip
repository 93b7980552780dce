// Tall-skinny fp32 GEMMs of the MPNN layer on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32:
// f32 in, f32 accumulate, bit-for-bit a k-ordered fmaf chain).
//
// The dense work around the edge pass is three shapes, all with a huge row count N and small
// feature widths (reference models/layers.py:56,69,104,108,140 and their autograd backward):
//
//   NT   Y[N,M]  = relu?( X1[N,K1] W1[M,K1]^T + X2[N,K2] W2[M,K2]^T )     projection, update, and (with
//                                                                         pre-transposed weights) every
//                                                                         input-gradient GEMM
//   TN   dW[M,K] = (G (.) mask)[N,M]^T X[N,K]                             weight gradients, reduction over N
//
// Both operands of the NT form are K-contiguous, so a lane fetches 4 consecutive k of its row with one
// ds_read_b128 and feeds 4 MFMAs (k-step t of a lane half h uses k = 8q + 4h + t for BOTH operands, a
// permutation of the k order inside 8-wide groups that leaves every dot product's term set intact).
// The TN form reduces over rows: each block walks a strided set of 32-row slabs, keeps a 128x128 output
// tile in accumulators and writes one partial tile; partials are summed in slab order by
// k_sum_partials (fixed order, no atomics).
#include "common.h"

namespace mpnn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBK = 32;         // k-chunk staged per step (floats)
constexpr int kLd = kBK + 4;    // LDS row stride: 36 floats = 9 sixteen-byte slots (odd => conflict-free b128)
constexpr int kBN = 128;        // rows of X per block step (4 waves x 32)

struct NtSegment {
  const float* X; int64_t ldx;        // [N, K]
  const float* mask; int64_t ldm;     // optional [N, K]: X is used where mask > 0, else 0
  const float* W; int64_t ldw;        // [M, K]
  int K;
};

struct NtArgs {
  NtSegment seg[2];
  int nseg;
  float* Y; int64_t ldy;
  const float* emul; int64_t ldemul;  // optional [N, M]: second output Y2 = Y * emul
  float* Y2; int64_t ldy2;
  int64_t N; int M;
  int relu;
};

// Branch-free 4-float load at columns k..k+3 of a row of K floats.  The address is clamped into the row so
// the load can always be issued (a branch around a load makes the compiler drain every outstanding load
// first); columns >= K are zeroed afterwards with select4().  ALIGNED requires K % 4 == 0 and k % 4 == 0.
template <bool ALIGNED>
__device__ __forceinline__ float4 load_chunk4(const float* row, int k, int K) {
  if (ALIGNED) {
    const int kk = k < K ? k : K - 4;
    return *reinterpret_cast<const float4*>(row + kk);
  }
  const int last = K - 1;
  float4 v;
  v.x = row[k < last ? k : last];
  v.y = row[k + 1 < last ? k + 1 : last];
  v.z = row[k + 2 < last ? k + 2 : last];
  v.w = row[k + 3 < last ? k + 3 : last];
  return v;
}

// bit j set <=> column k + j exists (and the row does)
__device__ __forceinline__ int valid4(int k, int K, bool row_ok) {
  if (!row_ok) return 0;
  return (k < K ? 1 : 0) | (k + 1 < K ? 2 : 0) | (k + 2 < K ? 4 : 0) | (k + 3 < K ? 8 : 0);
}

__device__ __forceinline__ float4 select4(float4 v, int vm) {
  v.x = (vm & 1) ? v.x : 0.f; v.y = (vm & 2) ? v.y : 0.f;
  v.z = (vm & 4) ? v.z : 0.f; v.w = (vm & 8) ? v.w : 0.f;
  return v;
}

__device__ __forceinline__ float4 relu_mask4(float4 v, float4 m) {
  v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f;
  v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
  return v;
}

// MT = 32-column output tiles per wave, WC = wave columns: a block is 4 x WC waves; wave (wr, wc) owns rows
// [32 wr, 32 wr + 32) of the 128-row step and columns [32 MT wc, 32 MT (wc + 1)) of the block's column slice.
// k-chunks are software-pipelined: while chunk c is multiplied out of one LDS buffer, chunk c + 1 is in
// flight from global memory into registers and is written to the other buffer before the (single) barrier.
// TRANS swaps the MFMA operand roles so that the accumulator holds the transpose of the output tile: a lane then
// owns 4 consecutive output columns of one row per register quad and the epilogue moves 16 bytes per access.
// That pays when the epilogue also reads emul and writes Y2 (three arrays); for a single output the
// 4-byte, whole-line pattern of the plain layout is faster.
template <int MT, int WC, bool ALIGNED, bool TRANS>
__global__ void __launch_bounds__(kBlock * WC) k_gemm_nt(NtArgs a) {
  constexpr int NTHREADS = kBlock * WC;
  constexpr int WROWS = 32 * MT * WC;
  constexpr int XQ = (kBN * (kBK / 4)) / NTHREADS;                   // float4 of X per thread and chunk
  constexpr int WQ = (WROWS * (kBK / 4) + NTHREADS - 1) / NTHREADS;  // float4 of W per thread and chunk
  constexpr int STAGE = (kBN + WROWS) * kLd;
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wr = wid & 3, wc = wid >> 2;
  const int li = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.y * WROWS;
  const int nch0 = (a.seg[0].K + kBK - 1) / kBK;
  const int nch = nch0 + (a.nseg > 1 ? (a.seg[1].K + kBK - 1) / kBK : 0);

  // The (row step, k-chunk) pairs of this block form one flat sequence, so the first chunk of the next
  // row step is already in flight while the last chunk of the current one is multiplied and stored.
  const int64_t nsteps_all = (a.N + kBN - 1) / kBN;
  const int64_t my_steps = blockIdx.x < nsteps_all ? (nsteps_all - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
  const int64_t total = my_steps * nch;

  f32x16 acc[MT];
#pragma unroll
  for (int c = 0; c < MT; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  float4 xr[XQ], mr[XQ], wreg[WQ];
  int xv[XQ], wv[WQ];
  bool masked = false;

  auto fetch = [&](int64_t row0, int c) {
    const NtSegment& sg = a.seg[c < nch0 ? 0 : 1];
    const int k0 = (c < nch0 ? c : c - nch0) * kBK;
    masked = sg.mask != nullptr;
#pragma unroll
    for (int q = 0; q < XQ; ++q) {
      const int idx = tid + NTHREADS * q;
      const int c4 = (idx & 7) * 4;
      int64_t row = row0 + (idx >> 3);
      row = row < a.N ? row : a.N - 1;   // rows past the end compute garbage that is never stored
      xr[q] = load_chunk4<ALIGNED>(sg.X + row * sg.ldx, k0 + c4, sg.K);
      if (masked) mr[q] = load_chunk4<ALIGNED>(sg.mask + row * sg.ldm, k0 + c4, sg.K);
      xv[q] = valid4(k0 + c4, sg.K, true);
    }
#pragma unroll
    for (int q = 0; q < WQ; ++q) {
      const int idx = tid + NTHREADS * q;
      const int c4 = (idx & 7) * 4;
      int r = m0 + (idx >> 3);
      r = r < a.M ? r : a.M - 1;
      wreg[q] = load_chunk4<ALIGNED>(sg.W + (int64_t)r * sg.ldw, k0 + c4, sg.K);
      wv[q] = valid4(k0 + c4, sg.K, true);
    }
  };
  auto stash = [&](int buf) {
    float* Xs = lds + buf * STAGE;
    float* Ws = Xs + kBN * kLd;
#pragma unroll
    for (int q = 0; q < XQ; ++q) {
      const int idx = tid + NTHREADS * q;
      float4 v = select4(xr[q], xv[q]);
      if (masked) v = relu_mask4(v, mr[q]);  // relu backward fused into the operand
      *reinterpret_cast<float4*>(Xs + (idx >> 3) * kLd + (idx & 7) * 4) = v;
    }
#pragma unroll
    for (int q = 0; q < WQ; ++q) {
      const int idx = tid + NTHREADS * q;
      if ((idx >> 3) < WROWS)
        *reinterpret_cast<float4*>(Ws + (idx >> 3) * kLd + (idx & 7) * 4) = select4(wreg[q], wv[q]);
    }
  };

  if (total > 0) {
    fetch((int64_t)blockIdx.x * kBN, 0);
    stash(0);
  }
  __syncthreads();
  int64_t step = 0;
  int c = 0;
  for (int64_t it = 0; it < total; ++it) {
    const int buf = (int)(it & 1);
    const int64_t row0 = (blockIdx.x + step * gridDim.x) * kBN;
    const bool last_chunk = c + 1 == nch;
    if (it + 1 < total) fetch(last_chunk ? row0 + (int64_t)gridDim.x * kBN : row0, last_chunk ? 0 : c + 1);
    const float* Xs = lds + buf * STAGE;
    const float* Ws = Xs + kBN * kLd;
#pragma unroll
    for (int kq = 0; kq < kBK / 8; ++kq) {
      const float4 av = *reinterpret_cast<const float4*>(Xs + (wr * 32 + li) * kLd + kq * 8 + lh * 4);
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const float4 bv = *reinterpret_cast<const float4*>(Ws + ((wc * MT + t) * 32 + li) * kLd + kq * 8 + lh * 4);
        if (TRANS) {
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv.x, av.x, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv.y, av.y, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv.z, av.z, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv.w, av.w, acc[t], 0, 0, 0);
        } else {
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[t], 0, 0, 0);
        }
      }
    }
    if (last_chunk && TRANS) {
      // Transposed tile: output row = lane & 31, column within the 32-wide tile = (reg & 3) + 8 * (reg >> 2)
      // + 4 * (lane >> 5); registers 4i..4i+3 are 4 consecutive columns.  The host picks this variant only
      // when every output row starts on a 16-byte boundary and M % 4 == 0.
      const int64_t row = row0 + wr * 32 + li;
#pragma unroll
      for (int t = 0; t < MT; ++t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int col = m0 + (wc * MT + t) * 32 + 8 * i + 4 * lh;
          float4 v = make_float4(acc[t][4 * i], acc[t][4 * i + 1], acc[t][4 * i + 2], acc[t][4 * i + 3]);
          if (a.relu) {
            v.x = v.x <= 0.f ? 0.f : v.x; v.y = v.y <= 0.f ? 0.f : v.y;
            v.z = v.z <= 0.f ? 0.f : v.z; v.w = v.w <= 0.f ? 0.f : v.w;
          }
          if (row < a.N && col < a.M) {
            *reinterpret_cast<float4*>(a.Y + row * a.ldy + col) = v;
            if (a.Y2) {
              const float4 e = *reinterpret_cast<const float4*>(a.emul + row * a.ldemul + col);
              *reinterpret_cast<float4*>(a.Y2 + row * a.ldy2 + col) = make_float4(v.x * e.x, v.y * e.y, v.z * e.z, v.w * e.w);
            }
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
      }
      ++step;
      c = 0;
    } else if (last_chunk) {
      // C layout of the 32x32 tile: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const int col = m0 + (wc * MT + t) * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = row0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (col < a.M && row < a.N) {
            float v = acc[t][r];
            if (a.relu) v = v <= 0.f ? 0.f : v;
            a.Y[row * a.ldy + col] = v;
            if (a.Y2) a.Y2[row * a.ldy2 + col] = v * a.emul[row * a.ldemul + col];
          }
          acc[t][r] = 0.f;
        }
      }
      ++step;
      c = 0;
    } else {
      ++c;
    }
    if (it + 1 < total) stash(buf ^ 1);
    __syncthreads();
  }
}

struct TnArgs {
  const float* G; int64_t ldg;        // [N, M]
  const float* mask; int64_t ldm;     // optional [N, M]: G is used where mask > 0
  const float* X; int64_t ldx;        // [N, K]
  float* partial;                     // [gridDim.x][M][K]
  int64_t N; int M; int K;
};

// Output tile 128 (m) x 128 (k) per block; wave (wm, wk) owns a 64 x 64 quadrant (2 x 2 MFMA tiles).
// Row slabs are software-pipelined like the k-chunks of k_gemm_nt (two LDS buffers, one barrier per slab).
template <bool ALIGNED>
__global__ void __launch_bounds__(kBlock) k_gemm_tn(TnArgs a) {
  __shared__ float Gs[2][32][128];
  __shared__ float Xs[2][32][128];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = (wid >> 1) * 64, wk = (wid & 1) * 64;
  const int m0 = blockIdx.y * 128, k0 = blockIdx.z * 128;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 gr[4], mr[4], xr[4];
  int gv[4], xv[4];
  auto fetch = [&](int64_t n0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // 32 rows x 32 float4 per matrix
      const int idx = tid + kBlock * q;
      const int c4 = (idx & 31) * 4;
      int64_t row = n0 + (idx >> 5);
      const bool row_ok = row < a.N;   // rows past the end must contribute exact zeros to the sums
      row = row_ok ? row : a.N - 1;
      gr[q] = load_chunk4<ALIGNED>(a.G + row * a.ldg, m0 + c4, a.M);
      if (a.mask) mr[q] = load_chunk4<ALIGNED>(a.mask + row * a.ldm, m0 + c4, a.M);
      xr[q] = load_chunk4<ALIGNED>(a.X + row * a.ldx, k0 + c4, a.K);
      gv[q] = valid4(m0 + c4, a.M, row_ok);
      xv[q] = valid4(k0 + c4, a.K, row_ok);
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = tid + kBlock * q;
      float4 g = select4(gr[q], gv[q]);
      if (a.mask) g = relu_mask4(g, mr[q]);
      *reinterpret_cast<float4*>(&Gs[buf][idx >> 5][(idx & 31) * 4]) = g;
      *reinterpret_cast<float4*>(&Xs[buf][idx >> 5][(idx & 31) * 4]) = select4(xr[q], xv[q]);
    }
  };

  const int64_t step = (int64_t)gridDim.x * 32;
  int64_t n0 = (int64_t)blockIdx.x * 32;
  int buf = 0;
  if (n0 < a.N) {
    fetch(n0);
    stash(0);
  }
  __syncthreads();
  for (; n0 < a.N; n0 += step, buf ^= 1) {
    const bool more = n0 + step < a.N;
    if (more) fetch(n0 + step);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int n = 2 * t + lh;
      const float a0 = Gs[buf][n][wm + li], a1 = Gs[buf][n][wm + 32 + li];
      const float b0 = Xs[buf][n][wk + li], b1 = Xs[buf][n][wk + 32 + li];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (more) stash(buf ^ 1);
    __syncthreads();
  }
  float* out = a.partial + (int64_t)blockIdx.x * a.M * a.K;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = k0 + wk + j * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < a.M && col < a.K) out[(int64_t)row * a.K + col] = acc[i][j][r];
      }
    }
}

// dW[m, k] = sum over slabs of partial[b][m][k] in a fixed order: each wave of a block takes the slabs
// b = wave, wave + 4, ... (ascending) for 64 consecutive outputs, then the four wave sums are added in wave order.
__global__ void __launch_bounds__(kBlock) k_sum_partials(const float* __restrict__ partial, int nslabs, int M, int K,
                                                         float* __restrict__ dW, int64_t ldw) {
  __shared__ float red[kBlock];
  const int64_t count = (int64_t)M * K;
  const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
  for (int64_t base = (int64_t)blockIdx.x * 64; base < count; base += (int64_t)gridDim.x * 64) {
    const int64_t i = base + o;
    float s = 0.f;
    if (i < count)
      for (int b = g; b < nslabs; b += kBlock / 64) s += partial[(int64_t)b * count + i];
    red[threadIdx.x] = s;
    __syncthreads();
    if (g == 0 && i < count) {
      float tot = red[o];
      for (int k = 1; k < kBlock / 64; ++k) tot += red[k * 64 + o];
      dW[(i / K) * ldw + (i % K)] = tot;
    }
    __syncthreads();
  }
}

// Number of row-slab walkers (gridDim.x) of the TN kernel: enough blocks to fill the chip (two 64 KiB-LDS
// blocks per CU over all output tiles), but at least 8 slabs of 32 rows per walker so that the partial tiles
// written and re-read stay small next to the operands.
static int tn_slabs(int64_t N, int M, int K) {
  const int tiles = ((M + 127) / 128) * ((K + 127) / 128);
  int64_t slabs = 512 / tiles;
  const int64_t need = ((N + 31) / 32 + 7) / 8;
  if (slabs > need) slabs = need;
  if (slabs < 1) slabs = 1;
  return (int)slabs;
}

static bool nt_aligned(const NtArgs& a) {
  for (int s = 0; s < a.nseg; ++s) {
    const NtSegment& g = a.seg[s];
    if (((uintptr_t)g.X & 15) || ((uintptr_t)g.W & 15) || (g.ldx & 3) || (g.ldw & 3) || (g.K & 3)) return false;
    if (g.mask && (((uintptr_t)g.mask & 15) || (g.ldm & 3))) return false;
  }
  return true;
}

template <int MT, int WC>
static void launch_nt(const NtArgs& a, bool aligned, hipStream_t stream) {
  int64_t blocks = (a.N + kBN - 1) / kBN;
  if (blocks > 1024) blocks = 1024;
  constexpr int WROWS = 32 * MT * WC;
  dim3 grid((unsigned)blocks, (a.M + WROWS - 1) / WROWS);
  const size_t lds = (size_t)2 * (kBN + WROWS) * kLd * sizeof(float);
  // transposed accumulator tile only for the three-array epilogue, and only when 16-byte accesses are legal
  const bool trans = a.Y2 && (a.M & 3) == 0 && ((a.ldy | a.ldy2 | a.ldemul) & 3) == 0 &&
                     (((uintptr_t)a.Y | (uintptr_t)a.Y2 | (uintptr_t)a.emul) & 15) == 0;
  if (trans) {
    if (aligned) k_gemm_nt<MT, WC, true, true><<<grid, kBlock * WC, lds, stream>>>(a);
    else k_gemm_nt<MT, WC, false, true><<<grid, kBlock * WC, lds, stream>>>(a);
  } else {
    if (aligned) k_gemm_nt<MT, WC, true, false><<<grid, kBlock * WC, lds, stream>>>(a);
    else k_gemm_nt<MT, WC, false, false><<<grid, kBlock * WC, lds, stream>>>(a);
  }
}

}  // namespace mpnn

using namespace mpnn;

extern "C" {

int32_t mpnn_gemm_nt_f32(const float* X1, int64_t ldx1, const float* mask1, int64_t ldm1, const float* W1,
                         int64_t ldw1, int32_t K1, const float* X2, int64_t ldx2, const float* mask2, int64_t ldm2,
                         const float* W2, int64_t ldw2, int32_t K2, float* Y, int64_t ldy, const float* emul,
                         int64_t ldemul, float* Y2, int64_t ldy2, int64_t N, int32_t M, int32_t relu,
                         mpnn_stream_t stream) {
  if (N < 0 || M < 1 || K1 < 1 || K2 < 0) return fail(MPNN_E_INVALID, "mpnn_gemm_nt_f32: bad sizes");
  if (N == 0) return MPNN_OK;
  if (!X1 || !W1 || !Y || (K2 > 0 && (!X2 || !W2))) return fail(MPNN_E_INVALID, "mpnn_gemm_nt_f32: null pointer argument");
  if (ldx1 < K1 || ldw1 < K1 || ldy < M || (K2 > 0 && (ldx2 < K2 || ldw2 < K2)))
    return fail(MPNN_E_INVALID, "mpnn_gemm_nt_f32: leading dimension too small");
  NtArgs a;
  a.seg[0] = NtSegment{X1, ldx1, mask1, ldm1, W1, ldw1, K1};
  a.seg[1] = NtSegment{X2, ldx2, mask2, ldm2, W2, ldw2, K2};
  a.nseg = K2 > 0 ? 2 : 1;
  if (Y2 && (!emul || ldemul < M || ldy2 < M)) return fail(MPNN_E_INVALID, "mpnn_gemm_nt_f32: Y2 needs emul and ld >= M");
  a.Y = Y; a.ldy = ldy; a.emul = emul; a.ldemul = ldemul; a.Y2 = Y2; a.ldy2 = ldy2;
  a.N = N; a.M = M; a.relu = relu;
  const bool aligned = nt_aligned(a);
  if (M <= 32) launch_nt<1, 1>(a, aligned, stream);
  else if (M <= 64) launch_nt<2, 1>(a, aligned, stream);
  else if (M <= 128) launch_nt<4, 1>(a, aligned, stream);
  else launch_nt<4, 2>(a, aligned, stream);
  MPNN_LAUNCH_CHECK("k_gemm_nt");
  return MPNN_OK;
}

int64_t mpnn_gemm_tn_scratch_floats(int64_t N, int32_t M, int32_t K) {
  if (N < 0 || M < 1 || K < 1) return -1;
  return (int64_t)tn_slabs(N, M, K) * M * K;
}

int32_t mpnn_gemm_tn_f32(const float* G, int64_t ldg, const float* mask, int64_t ldm, const float* X, int64_t ldx,
                         float* dW, int64_t lddw, float* scratch, int64_t N, int32_t M, int32_t K,
                         mpnn_stream_t stream) {
  if (N < 0 || M < 1 || K < 1) return fail(MPNN_E_INVALID, "mpnn_gemm_tn_f32: bad sizes");
  if (!dW) return fail(MPNN_E_INVALID, "mpnn_gemm_tn_f32: dW is null");
  if (lddw < K) return fail(MPNN_E_INVALID, "mpnn_gemm_tn_f32: lddw below K");
  if (N == 0) {
    MPNN_HIP_CHECK(hipMemset2DAsync(dW, (size_t)lddw * 4, 0, (size_t)K * 4, (size_t)M, stream));
    return MPNN_OK;
  }
  if (!G || !X || !scratch) return fail(MPNN_E_INVALID, "mpnn_gemm_tn_f32: null pointer argument");
  if (ldg < M || ldx < K || (mask && ldm < M)) return fail(MPNN_E_INVALID, "mpnn_gemm_tn_f32: leading dimension too small");
  TnArgs a{G, ldg, mask, ldm, X, ldx, scratch, N, M, K};
  const int slabs = tn_slabs(N, M, K);
  const bool aligned = !(((uintptr_t)G & 15) || ((uintptr_t)X & 15) || (ldg & 3) || (ldx & 3) || (M & 3) || (K & 3) ||
                         (mask && (((uintptr_t)mask & 15) || (ldm & 3))));
  dim3 grid(slabs, (M + 127) / 128, (K + 127) / 128);
  if (aligned) k_gemm_tn<true><<<grid, kBlock, 0, stream>>>(a);
  else k_gemm_tn<false><<<grid, kBlock, 0, stream>>>(a);
  MPNN_LAUNCH_CHECK("k_gemm_tn");
  int64_t blocks = ((int64_t)M * K + 63) / 64;
  if (blocks > kMaxBlocks) blocks = kMaxBlocks;
  k_sum_partials<<<(int)blocks, kBlock, 0, stream>>>(scratch, slabs, M, K, dW, lddw);
  MPNN_LAUNCH_CHECK("k_sum_partials");
  return MPNN_OK;
}

}  // extern "C"
