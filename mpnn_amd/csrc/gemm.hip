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
  int64_t N; int M;
  int relu;
};

template <bool ALIGNED>
__device__ __forceinline__ float4 load_chunk4(const float* row, int k, int K) {
  if (ALIGNED) {
    if (k + 3 < K) return *reinterpret_cast<const float4*>(row + k);
  }
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (k < K) v.x = row[k];
  if (k + 1 < K) v.y = row[k + 1];
  if (k + 2 < K) v.z = row[k + 2];
  if (k + 3 < K) v.w = row[k + 3];
  return v;
}

// MT = number of 32-column output tiles per wave (M <= 32 * MT per gridDim.y slice).
template <int MT, bool ALIGNED>
__global__ void __launch_bounds__(kBlock) k_gemm_nt(NtArgs a) {
  extern __shared__ float lds[];
  float* Xs = lds;                      // [kBN][kLd]
  float* Ws = lds + kBN * kLd;          // [32*MT][kLd]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.y * (32 * MT);

  for (int64_t row0 = (int64_t)blockIdx.x * kBN; row0 < a.N; row0 += (int64_t)gridDim.x * kBN) {
    f32x16 acc[MT];
#pragma unroll
    for (int c = 0; c < MT; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

    for (int s = 0; s < a.nseg; ++s) {
      const NtSegment sg = a.seg[s];
      for (int k0 = 0; k0 < sg.K; k0 += kBK) {
        __syncthreads();
        // stage X chunk: 128 rows x 8 float4
#pragma unroll
        for (int q = 0; q < (kBN * (kBK / 4)) / kBlock; ++q) {
          const int idx = tid + kBlock * q;
          const int r = idx >> 3, c4 = (idx & 7) * 4;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (row0 + r < a.N) {
            v = load_chunk4<ALIGNED>(sg.X + (row0 + r) * sg.ldx, k0 + c4, sg.K);
            if (sg.mask) {
              const float4 mk = load_chunk4<ALIGNED>(sg.mask + (row0 + r) * sg.ldm, k0 + c4, sg.K);
              v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f;
              v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
            }
          }
          *reinterpret_cast<float4*>(Xs + r * kLd + c4) = v;
        }
        // stage W chunk: 32*MT rows x 8 float4
#pragma unroll
        for (int q = 0; q < (32 * MT * (kBK / 4) + kBlock - 1) / kBlock; ++q) {
          const int idx = tid + kBlock * q;
          const int r = idx >> 3, c4 = (idx & 7) * 4;
          if (r < 32 * MT) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m0 + r < a.M) v = load_chunk4<ALIGNED>(sg.W + (int64_t)(m0 + r) * sg.ldw, k0 + c4, sg.K);
            *reinterpret_cast<float4*>(Ws + r * kLd + c4) = v;
          }
        }
        __syncthreads();
#pragma unroll
        for (int kq = 0; kq < kBK / 8; ++kq) {
          const float4 av = *reinterpret_cast<const float4*>(Xs + (wid * 32 + li) * kLd + kq * 8 + lh * 4);
#pragma unroll
          for (int c = 0; c < MT; ++c) {
            const float4 bv = *reinterpret_cast<const float4*>(Ws + (c * 32 + li) * kLd + kq * 8 + lh * 4);
            acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[c], 0, 0, 0);
            acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[c], 0, 0, 0);
            acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[c], 0, 0, 0);
            acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[c], 0, 0, 0);
          }
        }
      }
    }
    // C layout of the 32x32 tile: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int c = 0; c < MT; ++c) {
      const int col = m0 + c * 32 + li;
      if (col < a.M) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = row0 + wid * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (row < a.N) {
            float v = acc[c][r];
            if (a.relu) v = v <= 0.f ? 0.f : v;
            a.Y[row * a.ldy + col] = v;
          }
        }
      }
    }
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
template <bool ALIGNED>
__global__ void __launch_bounds__(kBlock) k_gemm_tn(TnArgs a) {
  __shared__ float Gs[32][128];
  __shared__ float Xs[32][128];
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

  for (int64_t n0 = (int64_t)blockIdx.x * 32; n0 < a.N; n0 += (int64_t)gridDim.x * 32) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < (32 * 32) / kBlock; ++q) {  // 32 rows x 32 float4 per matrix
      const int idx = tid + kBlock * q;
      const int r = idx >> 5, c4 = (idx & 31) * 4;
      float4 g = make_float4(0.f, 0.f, 0.f, 0.f), x = g;
      if (n0 + r < a.N) {
        g = load_chunk4<ALIGNED>(a.G + (n0 + r) * a.ldg, m0 + c4, a.M);
        if (a.mask) {
          const float4 mk = load_chunk4<ALIGNED>(a.mask + (n0 + r) * a.ldm, m0 + c4, a.M);
          g.x = mk.x > 0.f ? g.x : 0.f; g.y = mk.y > 0.f ? g.y : 0.f;
          g.z = mk.z > 0.f ? g.z : 0.f; g.w = mk.w > 0.f ? g.w : 0.f;
        }
        x = load_chunk4<ALIGNED>(a.X + (n0 + r) * a.ldx, k0 + c4, a.K);
      }
      *reinterpret_cast<float4*>(&Gs[r][c4]) = g;
      *reinterpret_cast<float4*>(&Xs[r][c4]) = x;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int n = 2 * t + lh;
      const float a0 = Gs[n][wm + li], a1 = Gs[n][wm + 32 + li];
      const float b0 = Xs[n][wk + li], b1 = Xs[n][wk + 32 + li];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
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

// dW[m, k] = sum over slabs b (ascending) of partial[b][m][k]
__global__ void __launch_bounds__(kBlock) k_sum_partials(const float* __restrict__ partial, int nslabs, int M, int K,
                                                         float* __restrict__ dW, int64_t ldw) {
  const int64_t count = (int64_t)M * K;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += (int64_t)gridDim.x * kBlock) {
    float s = 0.f;
    for (int b = 0; b < nslabs; ++b) s += partial[(int64_t)b * count + i];
    dW[(i / K) * ldw + (i % K)] = s;
  }
}

static int tn_slabs(int64_t N, int M, int K) {
  const int tiles = ((M + 127) / 128) * ((K + 127) / 128);
  int slabs = 768 / tiles;  // about 3 resident blocks per CU in total
  if (slabs < 1) slabs = 1;
  const int64_t need = (N + 31) / 32;
  if (slabs > need) slabs = (int)(need < 1 ? 1 : need);
  return slabs;
}

static bool nt_aligned(const NtArgs& a) {
  for (int s = 0; s < a.nseg; ++s) {
    const NtSegment& g = a.seg[s];
    if (((uintptr_t)g.X & 15) || ((uintptr_t)g.W & 15) || (g.ldx & 3) || (g.ldw & 3)) return false;
    if (g.mask && (((uintptr_t)g.mask & 15) || (g.ldm & 3))) return false;
  }
  return true;
}

template <int MT>
static void launch_nt(const NtArgs& a, bool aligned, hipStream_t stream) {
  int64_t blocks = (a.N + kBN - 1) / kBN;
  if (blocks > 1024) blocks = 1024;
  dim3 grid((unsigned)blocks, (a.M + 32 * MT - 1) / (32 * MT));
  const size_t lds = (size_t)(kBN + 32 * MT) * kLd * sizeof(float);
  if (aligned) k_gemm_nt<MT, true><<<grid, kBlock, lds, stream>>>(a);
  else k_gemm_nt<MT, false><<<grid, kBlock, lds, stream>>>(a);
}

}  // namespace mpnn

using namespace mpnn;

extern "C" {

int32_t mpnn_gemm_nt_f32(const float* X1, int64_t ldx1, const float* mask1, int64_t ldm1, const float* W1,
                         int64_t ldw1, int32_t K1, const float* X2, int64_t ldx2, const float* mask2, int64_t ldm2,
                         const float* W2, int64_t ldw2, int32_t K2, float* Y, int64_t ldy, int64_t N, int32_t M,
                         int32_t relu, mpnn_stream_t stream) {
  if (N < 0 || M < 1 || K1 < 1 || K2 < 0) return fail(MPNN_E_INVALID, "mpnn_gemm_nt_f32: bad sizes");
  if (N == 0) return MPNN_OK;
  if (!X1 || !W1 || !Y || (K2 > 0 && (!X2 || !W2))) return fail(MPNN_E_INVALID, "mpnn_gemm_nt_f32: null pointer argument");
  if (ldx1 < K1 || ldw1 < K1 || ldy < M || (K2 > 0 && (ldx2 < K2 || ldw2 < K2)))
    return fail(MPNN_E_INVALID, "mpnn_gemm_nt_f32: leading dimension too small");
  NtArgs a;
  a.seg[0] = NtSegment{X1, ldx1, mask1, ldm1, W1, ldw1, K1};
  a.seg[1] = NtSegment{X2, ldx2, mask2, ldm2, W2, ldw2, K2};
  a.nseg = K2 > 0 ? 2 : 1;
  a.Y = Y; a.ldy = ldy; a.N = N; a.M = M; a.relu = relu;
  const bool aligned = nt_aligned(a);
  if (M <= 32) launch_nt<1>(a, aligned, stream);
  else if (M <= 64) launch_nt<2>(a, aligned, stream);
  else if (M <= 128) launch_nt<4>(a, aligned, stream);
  else launch_nt<8>(a, aligned, stream);
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
  const bool aligned = !(((uintptr_t)G & 15) || ((uintptr_t)X & 15) || (ldg & 3) || (ldx & 3) ||
                         (mask && (((uintptr_t)mask & 15) || (ldm & 3))));
  dim3 grid(slabs, (M + 127) / 128, (K + 127) / 128);
  if (aligned) k_gemm_tn<true><<<grid, kBlock, 0, stream>>>(a);
  else k_gemm_tn<false><<<grid, kBlock, 0, stream>>>(a);
  MPNN_LAUNCH_CHECK("k_gemm_tn");
  int64_t blocks = ((int64_t)M * K + kBlock - 1) / kBlock;
  if (blocks > kMaxBlocks) blocks = kMaxBlocks;
  k_sum_partials<<<(int)blocks, kBlock, 0, stream>>>(scratch, slabs, M, K, dW, lddw);
  MPNN_LAUNCH_CHECK("k_sum_partials");
  return MPNN_OK;
}

}  // extern "C"
