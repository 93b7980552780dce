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
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace mpnn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

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
  const uint8_t* emul8;               // the same factor stored as bytes (emul is null then)
  float* Y2; int64_t ldy2;
  const float* C; int64_t ldc;        // optional [N, M]: added to the product before the activation
  int64_t N; int M;
  int relu;
  const char* wimg;   // optional: the weights split once per call into chunk images (k_split_weights)
  int ws_staged;      // k_gemm_nt_ws: 16-byte stores through LDS patches (aligned rows, M % 4 == 0)
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
// STAGED (the epilogue that also reads emul and writes Y2): every wave passes its accumulator tiles through a
// private 32 x 32 patch of the LDS buffer that has just been multiplied (one extra barrier per row step) and
// moves them to memory as whole 128-byte row segments, 16 bytes per lane - three arrays are touched there, and
// the 4-byte whole-line pattern of the plain epilogue (or 16 bytes per lane in 32 different rows) would cost as
// much as a separate elementwise pass.  For a single output the plain epilogue is faster.
//
// SPLIT computes the same float32 products on the bfloat16 matrix cores, which run 16 times the rate of the
// float32 ones: every operand value v is written to LDS as three bfloat16 pieces v = v1 + v2 + v3 (v1 = the upper
// 16 bits of v, v2 = the upper 16 bits of v - v1, v3 = v - v1 - v2; the three 8-bit significands hold all 24 bits
// of v, the sum is exact), and a product x*w is accumulated in float32 as x3*w1 + x1*w3 + x2*w2 + x2*w1 + x1*w2 +
// x1*w1 - the dropped terms (x2*w3, x3*w2, x3*w3) are at most 2^-23, typically 2^-25, of |x*w|, i.e. at the level
// of one float32 rounding.  Each piece product is exact in float32 (measured: error against float64 no larger
// than the float32 MFMA's).
// k-chunks are 16 wide (one v_mfma_f32_32x32x16_bf16 step); an LDS row is 3 pieces x 16 k x 2 B + 16 B pad.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Stand-in "mask" of an unmasked segment inside a kernel compiled with mask support: every lane reads these
// 16 bytes (a cache hit), which keeps the load sequence of a chunk free of branches.
__device__ const float kOnes[4] = {1.f, 1.f, 1.f, 1.f};

constexpr int kSplitBK = 16;
constexpr int kSplitRowB = 3 * kSplitBK * 2 + 16;   // 112 bytes = 7 sixteen-byte slots (odd => conflict-free b128)

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  const f32x2 v = {lo, hi};
  const bf16x2 b = __builtin_convertvector(v, bf16x2);   // round to nearest even, one v_cvt_pk_bf16_f32
  return __builtin_bit_cast(uint32_t, b);
}

// The three bfloat16 pieces of 4 consecutive floats, each as 4 packed values (8 bytes).  A piece is the upper
// half of the float (truncation); the remainder after taking 8 significant bits away is exact in float32, and
// the third remainder has at most 8 significant bits left, so v1 + v2 + v3 == v exactly.  Only full-rate
// integer/float instructions: one v_perm_b32 packs the upper halves of two floats.
__device__ __forceinline__ void split4(float4 v, uint2 (&p)[3]) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const uint32_t ux = __float_as_uint(v.x), uy = __float_as_uint(v.y), uz = __float_as_uint(v.z), uw = __float_as_uint(v.w);
    p[i].x = __builtin_amdgcn_perm(uy, ux, 0x07060302u);
    p[i].y = __builtin_amdgcn_perm(uw, uz, 0x07060302u);
    if (i < 2) {
      v.x -= __uint_as_float(ux & 0xffff0000u); v.y -= __uint_as_float(uy & 0xffff0000u);
      v.z -= __uint_as_float(uz & 0xffff0000u); v.w -= __uint_as_float(uw & 0xffff0000u);
    }
  }
}

template <int MT, int WC, bool ALIGNED, bool STAGED, bool SPLIT, bool MASKED, bool PRE = false>
__global__ void __launch_bounds__(kBlock * WC, 2) k_gemm_nt(NtArgs a) {
  static_assert(!SPLIT || ALIGNED, "the split form needs 16-byte aligned rows");
  static_assert(!PRE || SPLIT, "pre-split weight images belong to the split form");
  constexpr int NTHREADS = kBlock * WC;
  constexpr int WROWS = 32 * MT * WC;
  constexpr int BK = SPLIT ? kSplitBK : kBK;
  constexpr int QPR = BK / 4;                                       // float4 per row and chunk
  constexpr int XQ = (kBN * QPR) / NTHREADS;                        // float4 of X per thread and chunk
  // W per thread and chunk: float4 of the matrix itself, or (PRE) 16-byte parts of its chunk image, whose rows
  // are the 96 payload bytes of an LDS row (3 pieces x 16 k x 2 B)
  constexpr int WPARTS = WROWS * 6;
  constexpr int WQ = PRE ? (WPARTS + NTHREADS - 1) / NTHREADS : (WROWS * QPR + NTHREADS - 1) / NTHREADS;
  constexpr int ROWB = SPLIT ? kSplitRowB : kLd * 4;                // bytes per LDS row
  constexpr int STAGEB = (kBN + WROWS) * ROWB;
  extern __shared__ float lds[];
  char* const ldsb = reinterpret_cast<char*>(lds);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wr = wid & 3, wc = wid >> 2;
  const int li = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.y * WROWS;
  const int nch0 = (a.seg[0].K + BK - 1) / BK;
  const int nch = nch0 + (a.nseg > 1 ? (a.seg[1].K + BK - 1) / BK : 0);

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

  // Loads of X (HBM) run D - 1 chunks ahead of the multiply: a ring of D register sets (set i % D holds chunk i).
  // With the narrow chunks of the split form one chunk ahead covers only a fraction of the memory latency.
  // The weight tile comes out of the L2 and runs two chunks ahead in a ring of 2 sets (set = chunk parity).  Loads
  // return in issue order, so within an iteration the weight load is issued BEFORE the X load: waiting for the
  // weights of chunk i + 1 then leaves the younger X loads in flight.
  constexpr int D = (SPLIT && !MASKED) ? 4 : 2;   // mask operands double the registers of a set
  float4 xr[D][XQ], mr[MASKED ? D : 1][XQ], wreg[PRE ? 1 : 2][PRE ? 1 : WQ];
  u32x4 wpart[PRE ? 2 : 1][PRE ? WQ : 1];   // PRE: raw 16-byte parts of the weight chunk image
  int xk[D], xK[D], wk[2], wK[2];   // k origin and row length of the chunk a set holds (for the tail mask)
  bool masked[D];
  int64_t f_row0 = (int64_t)blockIdx.x * kBN;   // position of the next X fetch in the (row step, chunk) sequence
  int f_c = 0;

  auto fetch = [&](auto S) {
    constexpr int st = decltype(S)::value;
    const int c = f_c;
    const int64_t row0 = f_row0;
    if (++f_c == nch) { f_c = 0; f_row0 += (int64_t)gridDim.x * kBN; }
    const NtSegment& sg = a.seg[c < nch0 ? 0 : 1];
    const int k0 = (c < nch0 ? c : c - nch0) * BK;
    masked[st] = MASKED && sg.mask != nullptr;
    xk[st] = k0;
    xK[st] = sg.K;
    if constexpr (ALIGNED) {
      // uniform base pointer + a 32-bit per-thread offset: the loads take the scalar-base addressing form and
      // cost no per-thread 64-bit arithmetic.  Rows past the end re-read the last row (never stored).
      const int64_t left = a.N - 1 - row0;
      const int rt_max = left < kBN - 1 ? (left < 0 ? 0 : (int)left) : kBN - 1;
      const int64_t rbase = row0 <= a.N - 1 ? row0 : a.N - 1;
      const float* xb = sg.X + rbase * sg.ldx;
      const float* mb = masked[st] ? sg.mask + rbase * sg.ldm : kOnes;
      const uint32_t ldx = (uint32_t)sg.ldx, ldm = masked[st] ? (uint32_t)sg.ldm : 0u;
      const bool tail = k0 + BK > sg.K;
#pragma unroll
      for (int q = 0; q < XQ; ++q) {
        const int idx = tid + NTHREADS * q;
        int kk = k0 + (idx % QPR) * 4;
        if (tail) kk = kk < sg.K ? kk : sg.K - 4;
        int rt = idx / QPR;
        rt = rt < rt_max ? rt : rt_max;
        xr[st][q] = *reinterpret_cast<const float4*>(xb + ((uint32_t)rt * ldx + (uint32_t)kk));
        if constexpr (MASKED)
          mr[st][q] = *reinterpret_cast<const float4*>(mb + ((uint32_t)rt * ldm + (masked[st] ? (uint32_t)kk : 0u)));
      }
    } else {
#pragma unroll
      for (int q = 0; q < XQ; ++q) {
        const int idx = tid + NTHREADS * q;
        const int c4 = (idx % QPR) * 4;
        int64_t row = row0 + idx / QPR;
        row = row < a.N ? row : a.N - 1;   // rows past the end compute garbage that is never stored
        xr[st][q] = load_chunk4<ALIGNED>(sg.X + row * sg.ldx, k0 + c4, sg.K);
        if constexpr (MASKED) {
          const float* mp = masked[st] ? sg.mask + row * sg.ldm : kOnes;
          mr[st][q] = load_chunk4<ALIGNED>(mp, masked[st] ? k0 + c4 : 0, masked[st] ? sg.K : 4);
        }
      }
    }
  };
  auto fetch_w = [&](auto P, int c) {
    constexpr int ws = decltype(P)::value;
    const NtSegment& sg = a.seg[c < nch0 ? 0 : 1];
    wk[ws] = (c < nch0 ? c : c - nch0) * BK;
    wK[ws] = sg.K;
    if constexpr (PRE) {
      const char* img = a.wimg + ((int64_t)blockIdx.y * nch + c) * (WPARTS * 16);
#pragma unroll
      for (int q = 0; q < WQ; ++q) {
        int idx = tid + NTHREADS * q;
        idx = idx < WPARTS ? idx : WPARTS - 1;
        wpart[ws][q] = *reinterpret_cast<const u32x4*>(img + idx * 16);
      }
    } else if constexpr (ALIGNED) {
      const int r_max = a.M - 1 - m0 < WROWS - 1 ? a.M - 1 - m0 : WROWS - 1;
      const float* wb = sg.W + (int64_t)m0 * sg.ldw;
      const uint32_t ldw = (uint32_t)sg.ldw;
      const bool tail = wk[ws] + BK > sg.K;
#pragma unroll
      for (int q = 0; q < WQ; ++q) {
        const int idx = tid + NTHREADS * q;
        int kk = wk[ws] + (idx % QPR) * 4;
        if (tail) kk = kk < sg.K ? kk : sg.K - 4;
        int r = idx / QPR;
        r = r < r_max ? r : r_max;
        wreg[ws][q] = *reinterpret_cast<const float4*>(wb + ((uint32_t)r * ldw + (uint32_t)kk));
      }
    } else {
#pragma unroll
      for (int q = 0; q < WQ; ++q) {
        const int idx = tid + NTHREADS * q;
        int r = m0 + idx / QPR;
        r = r < a.M ? r : a.M - 1;
        wreg[ws][q] = load_chunk4<ALIGNED>(sg.W + (int64_t)r * sg.ldw, wk[ws] + (idx % QPR) * 4, sg.K);
      }
    }
  };
  auto put = [&](char* tile, int idx, float4 v) {
    char* dst = tile + (idx / QPR) * ROWB;
    if constexpr (SPLIT) {
      uint2 p[3];
      split4(v, p);
#pragma unroll
      for (int i = 0; i < 3; ++i) *reinterpret_cast<uint2*>(dst + i * (kSplitBK * 2) + (idx % QPR) * 8) = p[i];
    } else {
      *reinterpret_cast<float4*>(dst + (idx % QPR) * 16) = v;
    }
  };
  auto stash = [&](auto S, int buf) {
    constexpr int st = decltype(S)::value;
    char* Xs = ldsb + buf * STAGEB;
    char* Ws = Xs + kBN * ROWB;
    constexpr int ws = st & 1;
    const bool x_tail = xk[st] + BK > xK[st], w_tail = wk[ws] + BK > wK[ws];   // uniform: only a row's last chunk masks
#pragma unroll
    for (int q = 0; q < XQ; ++q) {
      const int idx = tid + NTHREADS * q;
      float4 v = xr[st][q];
      if (x_tail) v = select4(v, valid4(xk[st] + (idx % QPR) * 4, xK[st], true));
      if constexpr (MASKED) v = relu_mask4(v, mr[st][q]);  // relu backward fused into the operand (ones if unmasked)
      put(Xs, idx, v);
    }
    if constexpr (PRE) {   // already split and zero-padded: a plain copy into the LDS rows
#pragma unroll
      for (int q = 0; q < WQ; ++q) {
        const int idx = tid + NTHREADS * q;
        if (idx < WPARTS) *reinterpret_cast<u32x4*>(Ws + (idx / 6) * ROWB + (idx % 6) * 16) = wpart[ws][q];
      }
    } else {
#pragma unroll
      for (int q = 0; q < WQ; ++q) {
        const int idx = tid + NTHREADS * q;
        float4 v = wreg[ws][q];
        if (w_tail) v = select4(v, valid4(wk[ws] + (idx % QPR) * 4, wK[ws], true));
        if (idx / QPR < WROWS) put(Ws, idx, v);
      }
    }
  };

  int64_t step = 0;
  int c = 0;
  // one chunk: start the load of chunk it + D - 1 (into the set chunk it - 1 has just left), multiply chunk it out
  // of LDS buffer it & 1, store the tile at the end of a row step, then move chunk it + 1 from its registers
  // into the other LDS buffer
  auto body = [&](auto S, int64_t it) __attribute__((always_inline)) {
    constexpr int st = decltype(S)::value;            // == it % D
    using Prev = std::integral_constant<int, (st + D - 1) % D>;
    using Next = std::integral_constant<int, (st + 1) % D>;
    const int buf = (int)(it & 1);
    const int64_t row0 = (blockIdx.x + step * gridDim.x) * kBN;
    const bool last_chunk = c + 1 == nch;
    // Both loads are issued unconditionally (past the end of the block's work they re-read valid rows and are
    // never used): a load under a branch makes the compiler wait for EVERY outstanding load at the next use,
    // which would drain the ring each chunk.
    {
      int cw = c + 2;
      while (cw >= nch) cw -= nch;
      fetch_w(std::integral_constant<int, st & 1>{}, cw);
    }
    fetch(Prev{});
    const char* Xs = ldsb + buf * STAGEB;
    const char* Ws = Xs + kBN * ROWB;
    if constexpr (SPLIT) {
      // lane (li, lh) holds k = 8 lh .. 8 lh + 7 of row li of each piece: one 16-byte read per piece
      const char* xrow = Xs + (wr * 32 + li) * ROWB + lh * 16;
      const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(xrow);
      const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(xrow + 2 * kSplitBK);
      const bf16x8 a3 = *reinterpret_cast<const bf16x8*>(xrow + 4 * kSplitBK);
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const char* wrow = Ws + ((wc * MT + t) * 32 + li) * ROWB + lh * 16;
        const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(wrow);
        const bf16x8 b2 = *reinterpret_cast<const bf16x8*>(wrow + 2 * kSplitBK);
        const bf16x8 b3 = *reinterpret_cast<const bf16x8*>(wrow + 4 * kSplitBK);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[t], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int kq = 0; kq < kBK / 8; ++kq) {
        const float4 av = *reinterpret_cast<const float4*>(Xs + (wr * 32 + li) * ROWB + (kq * 8 + lh * 4) * 4);
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          const float4 bv = *reinterpret_cast<const float4*>(Ws + ((wc * MT + t) * 32 + li) * ROWB + (kq * 8 + lh * 4) * 4);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[t], 0, 0, 0);
        }
      }
    }
    if (last_chunk && STAGED) {
      // C layout of a 32x32 tile: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).  The
      // buffer just multiplied is free once every wave has left the loop above; the next write into it is the
      // stash of the following iteration, which comes after this iteration's closing barrier.  The host picks
      // this variant only when every output row starts on a 16-byte boundary and M % 4 == 0.
      __syncthreads();
      constexpr int PITCH = 36;   // floats: 9 sixteen-byte slots, conflict-free 16-byte reads
      float* patch = reinterpret_cast<float*>(ldsb + buf * STAGEB) + wid * (32 * PITCH);
      const int prow = lane >> 3, pc4 = (lane & 7) * 4;   // a lane moves 4 columns of rows prow, prow + 8, ..
#pragma unroll
      for (int t = 0; t < MT; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[t][r];
          if (a.relu) v = v <= 0.f ? 0.f : v;
          patch[((r & 3) + 8 * (r >> 2) + 4 * lh) * PITCH + li] = v;
          acc[t][r] = 0.f;
        }
        __builtin_amdgcn_wave_barrier();
        const int col = m0 + (wc * MT + t) * 32 + pc4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int pr = prow + 8 * k;
          const int64_t row = row0 + wr * 32 + pr;
          const float4 v = *reinterpret_cast<const float4*>(patch + pr * PITCH + pc4);
          if (row < a.N && col < a.M) {
            *reinterpret_cast<float4*>(a.Y + row * a.ldy + col) = v;
            if (a.Y2) {
              float4 e;
              if (a.emul8) {   // uniform over the launch
                const uint32_t w = *reinterpret_cast<const uint32_t*>(a.emul8 + row * a.ldemul + col);
                e = make_float4((float)(w & 255u), (float)((w >> 8) & 255u), (float)((w >> 16) & 255u), (float)(w >> 24));
              } else {
                e = *reinterpret_cast<const float4*>(a.emul + row * a.ldemul + col);
              }
              *reinterpret_cast<float4*>(a.Y2 + row * a.ldy2 + col) = make_float4(v.x * e.x, v.y * e.y, v.z * e.z, v.w * e.w);
            }
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      ++step;
      c = 0;
    } else if (last_chunk) {
      // C layout of the 32x32 tile: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
      constexpr bool WITH_Y2 = !SPLIT;   // the split form leaves a second output to the staged variant
      const bool interior = row0 + kBN <= a.N && m0 + WROWS <= a.M;   // uniform over the block
      if (interior && !(WITH_Y2 && a.Y2)) {
        // whole tile inside the matrix: one running row pointer, the MT column tiles at immediate offsets
        float* p = a.Y + (row0 + wr * 32 + 4 * lh) * a.ldy + (m0 + wc * MT * 32 + li);
        const bool relu = a.relu != 0;
        if (a.C) {   // uniform: the addend's rows are read in the same pattern, all of a tile column before its sums
          const float* cp = a.C + (row0 + wr * 32 + 4 * lh) * a.ldc + (m0 + wc * MT * 32 + li);
#pragma unroll
          for (int t = 0; t < MT; ++t) {
            float cv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) cv[r] = cp[((r & 3) + 8 * (r >> 2)) * a.ldc + t * 32];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] += cv[r];
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
          for (int t = 0; t < MT; ++t) {
            float v = acc[t][r];
            v = (relu && v <= 0.f) ? 0.f : v;
            p[t * 32] = v;
            acc[t][r] = 0.f;
          }
          p += ((r & 3) == 3 ? 5 : 1) * a.ldy;
        }
      } else {
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          const int col = m0 + (wc * MT + t) * 32 + li;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int64_t row = row0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (col < a.M && row < a.N) {
              float v = acc[t][r];
              if (a.C) v += a.C[row * a.ldc + col];
              if (a.relu) v = v <= 0.f ? 0.f : v;
              a.Y[row * a.ldy + col] = v;
              if constexpr (WITH_Y2) {
                if (a.Y2)
                  a.Y2[row * a.ldy2 + col] =
                      v * (a.emul8 ? (float)a.emul8[row * a.ldemul + col] : a.emul[row * a.ldemul + col]);
              }
            }
            acc[t][r] = 0.f;
          }
        }
      }
      ++step;
      c = 0;
    } else {
      ++c;
    }
    if (it + 1 < total) stash(Next{}, buf ^ 1);
    __syncthreads();
  };

  // prologue: chunks 0 .. D-2 in flight, chunk 0 in LDS
  if (total == 0) return;   // uniform over the block
  fetch_w(std::integral_constant<int, 0>{}, 0);
  fetch(std::integral_constant<int, 0>{});
  fetch_w(std::integral_constant<int, 1>{}, 1 % nch);
  if constexpr (D > 2) {
    fetch(std::integral_constant<int, 1>{});
    fetch(std::integral_constant<int, 2>{});
  }
  stash(std::integral_constant<int, 0>{}, 0);
  __syncthreads();
  for (int64_t it = 0; it < total; it += D) {
    body(std::integral_constant<int, 0>{}, it);
    if (it + 1 < total) body(std::integral_constant<int, 1>{}, it + 1);
    if constexpr (D > 2) {
      if (it + 2 < total) body(std::integral_constant<int, 2>{}, it + 2);
      if (it + 3 < total) body(std::integral_constant<int, 3>{}, it + 3);
    }
  }
}

// One block per (column block, 16-wide chunk): the chunk's weight rows as their three bfloat16 pieces, 96 bytes per
// row in the order of an LDS row of k_gemm_nt's split form; rows and columns past the end are zeros.
// `pitch` = bytes per image row: 96 (payload only) or the 112 of an LDS row, pad included, for the kernel that copies
// whole images into LDS with linear DMA pieces.
__global__ void __launch_bounds__(kBlock) k_split_weights(NtArgs a, char* __restrict__ out, int wrows, int pitch) {
  const int nch0 = (a.seg[0].K + kSplitBK - 1) / kSplitBK;
  const int nch = nch0 + (a.nseg > 1 ? (a.seg[1].K + kSplitBK - 1) / kSplitBK : 0);
  const int cb = blockIdx.x / nch, c = blockIdx.x % nch;
  const NtSegment& sg = a.seg[c < nch0 ? 0 : 1];
  const int k0 = (c < nch0 ? c : c - nch0) * kSplitBK;
  char* img = out + (int64_t)blockIdx.x * wrows * pitch;
  for (int idx = threadIdx.x; idx < wrows * (kSplitBK / 4); idx += kBlock) {
    const int r = idx / (kSplitBK / 4), q = idx % (kSplitBK / 4);
    const int m = cb * wrows + r, kk = k0 + q * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m < a.M && kk < sg.K) v = *reinterpret_cast<const float4*>(sg.W + (int64_t)m * sg.ldw + kk);
    uint2 pz[3];
    split4(v, pz);
#pragma unroll
    for (int i = 0; i < 3; ++i) *reinterpret_cast<uint2*>(img + r * pitch + i * (kSplitBK * 2) + q * 8) = pz[i];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The split form with the work divided between waves: a block is 8 waves, 4 PRODUCERS and 4 CONSUMERS, one of each
// per SIMD, one block per CU.  Producers fetch X (a ring of 8 register sets: the loads run 8 k-chunks ahead of their
// use), split it into its three bfloat16 pieces and write the pieces to LDS; consumers copy the chunk images of the
// pre-split weights (k_split_weights, rows at the LDS pitch) from global memory straight into LDS with DMA loads
// (no registers, no LDS store instructions), read fragments and issue matrix instructions - nothing else.
// Measured with per-wave cycle counters (tools/ws_prof.py): while a consumer is not waiting at a barrier or storing
// its tile it issues one v_mfma_f32_32x32x16_bf16 every 49 cycles, which is the rate a bare loop of that
// instruction reaches on this chip (tools/mfma_probe.hip: 40 per microsecond and SIMD, 1.34 PFLOP/s over the chip -
// not the 32 cycles of the data sheet); the rest of the time is the barrier per chunk (8-13 %), the epilogue and
// the last, partly empty round of row steps.
// LDS holds a ring of 3 stages (X tile 128 rows + W tile 64 NC rows, 112-byte rows); during iteration i the
// consumers multiply chunk i out of registers that were filled from stage i % 3 during iteration i - 1, request
// the fragments of chunk i + 1 and start the copy of weight image i + 2, while the producers write the X pieces of
// chunk i + 2 into stage (i + 2) % 3; one block barrier per chunk.  A consumer wave owns 64 rows x 32 NC columns
// (2 x NC matrix tiles: 6 + 3 NC fragment reads for 12 NC matrix instructions per chunk).  The (row step, k-chunk)
// pairs of a block form one flat sequence as in k_gemm_nt.  Same piece products in the same order as k_gemm_nt:
// the two kernels agree bit for bit.
constexpr int kWsStages = 3;
constexpr int kWsRound = 24;   // copies of the producers' loop body: register sets (8 or 4) x stages

// Per-wave cycle counts of k_gemm_nt_ws for tools/ws_prof.py (built only with -DMPNN_WS_PROF): total, spent at
// barriers, spent in the epilogue.
#ifdef MPNN_WS_PROF
__device__ long long* g_ws_prof = nullptr;
#define WS_NOW() ((long long)__builtin_readcyclecounter())
#define WS_SYNC() do { const long long t0_ = WS_NOW(); __syncthreads(); prof_wait += WS_NOW() - t0_; } while (0)
#else
#define WS_SYNC() __syncthreads()
#endif

// (stages + one 32 x 32 patch, pitch 36 floats, per consumer wave for the epilogue)
template <int NC>
constexpr int ws_lds_bytes() { return kWsStages * (kBN + 64 * NC) * kSplitRowB + 4 * 32 * 36 * 4; }

template <int NC, bool MASKED>
__global__ void __launch_bounds__(512, 1) k_gemm_nt_ws(NtArgs a) {
  constexpr int BK = kSplitBK;
  constexpr int WROWS = 64 * NC;
  constexpr int ROWB = kSplitRowB;
  constexpr int STAGEB = (kBN + WROWS) * ROWB;
  constexpr int NPROD = 256;
  constexpr int XQ = (kBN * (BK / 4)) / NPROD;          // float4 of X per producer thread and chunk (2)
  constexpr int WIMG = WROWS * ROWB;                    // bytes of a weight chunk image, LDS row pitch included
  constexpr int WPIECES = WIMG / 1024;                  // 1-KiB pieces of it: one DMA instruction of a wave each
  static_assert(WIMG % 1024 == 0, "the weight image must be whole DMA pieces");
  constexpr int D = MASKED ? 4 : 8;                     // register sets of the X ring (chunks in flight)
  extern __shared__ float lds[];
  char* const ldsb = reinterpret_cast<char*>(lds);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.y * WROWS;
  const int nch0 = (a.seg[0].K + BK - 1) / BK;
  const int nch = nch0 + (a.nseg > 1 ? (a.seg[1].K + BK - 1) / BK : 0);
  const int64_t nsteps_all = (a.N + kBN - 1) / kBN;
  const int64_t my_steps = blockIdx.x < nsteps_all ? (nsteps_all - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
  const int64_t total = my_steps * nch;
  if (total == 0) return;   // uniform over the block
#ifdef MPNN_WS_PROF
  long long prof_wait = 0, prof_epi = 0;
  const long long prof_t0 = WS_NOW();
  auto prof_done = [&]() {
    if (g_ws_prof && lane == 0) {
      long long* o = g_ws_prof + ((int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + wid) * 4;
      o[0] = WS_NOW() - prof_t0; o[1] = prof_wait; o[2] = prof_epi; o[3] = total;
    }
  };
#endif

  if (wid >= 4) {
    // ------------------------------------------------------------------------------------------ producers
    const int pt = tid - 256;
    float4 xr[D][XQ], mr[MASKED ? D : 1][XQ];
    int xk[D], xK[D];
    int64_t f_row0 = (int64_t)blockIdx.x * kBN;   // position of the next X fetch in the (row step, chunk) sequence
    int f_c = 0;
    auto fetch_x = [&](auto S) {
      constexpr int st = decltype(S)::value;
      const int c = f_c;
      const int64_t row0 = f_row0;
      if (++f_c == nch) { f_c = 0; f_row0 += (int64_t)gridDim.x * kBN; }
      const NtSegment& sg = a.seg[c < nch0 ? 0 : 1];
      const int k0 = (c < nch0 ? c : c - nch0) * BK;
      const bool masked = MASKED && sg.mask != nullptr;
      xk[st] = k0;
      xK[st] = sg.K;
      // uniform base + 32-bit lane offset; rows past the end re-read the last row (never stored), chunks past
      // the end of the block's work re-read valid rows (never used): every load is issued unconditionally
      const int64_t left = a.N - 1 - row0;
      const int rt_max = left < kBN - 1 ? (left < 0 ? 0 : (int)left) : kBN - 1;
      const int64_t rbase = row0 <= a.N - 1 ? row0 : a.N - 1;
      const float* xb = sg.X + rbase * sg.ldx;
      const float* mb = masked ? sg.mask + rbase * sg.ldm : kOnes;
      const uint32_t ldx = (uint32_t)sg.ldx, ldm = masked ? (uint32_t)sg.ldm : 0u;
      const bool tail = k0 + BK > sg.K;
#pragma unroll
      for (int q = 0; q < XQ; ++q) {
        const int idx = pt + NPROD * q;
        int kk = k0 + (idx & 3) * 4;
        if (tail) kk = kk < sg.K ? kk : sg.K - 4;
        int rt = idx >> 2;
        rt = rt < rt_max ? rt : rt_max;
        xr[st][q] = *reinterpret_cast<const float4*>(xb + ((uint32_t)rt * ldx + (uint32_t)kk));
        if constexpr (MASKED)
          mr[st][q] = *reinterpret_cast<const float4*>(mb + ((uint32_t)rt * ldm + (masked ? (uint32_t)kk : 0u)));
      }
    };
    auto stash = [&](auto S, int stage) {
      constexpr int st = decltype(S)::value;
      char* Xs = ldsb + stage * STAGEB;
      const bool x_tail = xk[st] + BK > xK[st];   // uniform: only a row's last chunk masks
#pragma unroll
      for (int q = 0; q < XQ; ++q) {
        const int idx = pt + NPROD * q;
        float4 v = xr[st][q];
        if (x_tail) v = select4(v, valid4(xk[st] + (idx & 3) * 4, xK[st], true));
        if constexpr (MASKED) v = relu_mask4(v, mr[st][q]);
        uint2 p[3];
        split4(v, p);
        char* dst = Xs + (idx >> 2) * ROWB + (idx & 3) * 8;
#pragma unroll
        for (int i = 0; i < 3; ++i) *reinterpret_cast<uint2*>(dst + i * (BK * 2)) = p[i];
      }
    };
    // produce(s): chunk s goes from its register set s % D into stage s % 3, and the freed set starts the loads
    // of chunk s + D
    auto produce = [&](auto J, int64_t s) __attribute__((always_inline)) {
      constexpr int j = decltype(J)::value;   // == s % 24
      using XS = std::integral_constant<int, j % D>;
      if (s < total) stash(XS{}, j % kWsStages);
      fetch_x(XS{});
    };
    fetch_x(std::integral_constant<int, 0>{});
    fetch_x(std::integral_constant<int, 1>{});
    fetch_x(std::integral_constant<int, 2>{});
    fetch_x(std::integral_constant<int, 3>{});
    if constexpr (D > 4) {
      fetch_x(std::integral_constant<int, 4>{});
      fetch_x(std::integral_constant<int, 5>{});
      fetch_x(std::integral_constant<int, 6>{});
      fetch_x(std::integral_constant<int, 7>{});
    }
    produce(std::integral_constant<int, 0>{}, 0);
    produce(std::integral_constant<int, 1>{}, 1);
    WS_SYNC();
    // All loads of the prologue land before the loop is entered (s_waitcnt vmcnt(0), once per block): the wait counts
    // inside the loop are the worse of the two ways into it, and the compiler orders the prologue's loads as it
    // likes - without this the first copy waits for almost the whole ring on every round.
    __builtin_amdgcn_s_waitcnt(0x0F70);
    // The loop always runs whole rounds of 24 copies (register sets x stages; the consumers execute the same
    // number of barriers): a copy that could be skipped would give the compiler a path on which the loads just issued
    // are the next ones used (skip -> back edge -> first copy), and it would then wait for the whole ring at the top
    // of every round.  Chunks past the end are fetched from valid addresses and never written to LDS.
    for (int64_t it = 0; it < total; it += kWsRound) {
      produce(std::integral_constant<int, 2>{}, it + 2);
      WS_SYNC();
      produce(std::integral_constant<int, 3>{}, it + 3);
      WS_SYNC();
      produce(std::integral_constant<int, 4>{}, it + 4);
      WS_SYNC();
      produce(std::integral_constant<int, 5>{}, it + 5);
      WS_SYNC();
      produce(std::integral_constant<int, 6>{}, it + 6);
      WS_SYNC();
      produce(std::integral_constant<int, 7>{}, it + 7);
      WS_SYNC();
      produce(std::integral_constant<int, 8>{}, it + 8);
      WS_SYNC();
      produce(std::integral_constant<int, 9>{}, it + 9);
      WS_SYNC();
      produce(std::integral_constant<int, 10>{}, it + 10);
      WS_SYNC();
      produce(std::integral_constant<int, 11>{}, it + 11);
      WS_SYNC();
      produce(std::integral_constant<int, 12>{}, it + 12);
      WS_SYNC();
      produce(std::integral_constant<int, 13>{}, it + 13);
      WS_SYNC();
      produce(std::integral_constant<int, 14>{}, it + 14);
      WS_SYNC();
      produce(std::integral_constant<int, 15>{}, it + 15);
      WS_SYNC();
      produce(std::integral_constant<int, 16>{}, it + 16);
      WS_SYNC();
      produce(std::integral_constant<int, 17>{}, it + 17);
      WS_SYNC();
      produce(std::integral_constant<int, 18>{}, it + 18);
      WS_SYNC();
      produce(std::integral_constant<int, 19>{}, it + 19);
      WS_SYNC();
      produce(std::integral_constant<int, 20>{}, it + 20);
      WS_SYNC();
      produce(std::integral_constant<int, 21>{}, it + 21);
      WS_SYNC();
      produce(std::integral_constant<int, 22>{}, it + 22);
      WS_SYNC();
      produce(std::integral_constant<int, 23>{}, it + 23);
      WS_SYNC();
      produce(std::integral_constant<int, 0>{}, it + 24);
      WS_SYNC();
      produce(std::integral_constant<int, 1>{}, it + 25);
      WS_SYNC();
    }
#ifdef MPNN_WS_PROF
    prof_done();
#endif
    return;
  }

  // -------------------------------------------------------------------------------------------- consumers
  const int li = lane & 31, lh = lane >> 5;
  const int cr = wid >> 1, cc = wid & 1;
  f32x16 acc[2][NC];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < NC; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;
  int64_t step = 0;
  int c = 0;
  // Fragments of chunk i + 1 are requested while chunk i is multiplied (the LDS ring holds chunk i + 1 complete
  // since the barrier before iteration i), each into the registers of the fragment it replaces as soon as the last
  // matrix instruction that reads those has been issued: the W fragments of a column tile after that tile's
  // products, the X fragments of a row tile after its products with the last column tile.  Without this every
  // chunk exposes one LDS round trip and the barrier (a third of the time).
  bf16x8 av[2][3], bv[NC][3];
  // per-lane parts of the fragment addresses; the stage offset is added once per chunk, tiles and pieces are
  // immediate offsets.  Lane (li, lh) holds k = 8 lh .. 8 lh + 7 of row li of each piece: one 16-byte read per piece
  const int a_lane = (cr * 64 + li) * ROWB + lh * 16;
  const int b_lane = kBN * ROWB + (cc * NC * 32 + li) * ROWB + lh * 16;
  auto read_a = [&](int i, const char* sb) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < 3; ++p) av[i][p] = *reinterpret_cast<const bf16x8*>(sb + a_lane + i * 32 * ROWB + p * (2 * BK));
  };
  auto read_b = [&](int t, const char* sb) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < 3; ++p) bv[t][p] = *reinterpret_cast<const bf16x8*>(sb + b_lane + t * 32 * ROWB + p * (2 * BK));
  };
  auto epilogue = [&]() __attribute__((always_inline)) {
    // C layout of a 32 x 32 tile: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    const int64_t row0 = (blockIdx.x + step * gridDim.x) * kBN + cr * 64;
    ++step;
    const bool relu = a.relu != 0;
    if (a.ws_staged) {
      // Every tile passes through the wave's own 32 x 32 patch of LDS (behind the stages) and leaves as 16 bytes per
      // lane, 8 lanes to a 128-byte row segment: a quarter of the store instructions of the 4-byte form, which is
      // what an epilogue costs (the host picks this when rows are 16-byte aligned and M % 4 == 0).
      constexpr int PITCH = 36;   // floats: 9 sixteen-byte slots, conflict-free 16-byte reads
      float* patch = reinterpret_cast<float*>(ldsb + kWsStages * STAGEB) + wid * (32 * PITCH);
      const int prow = lane >> 3, pc4 = (lane & 7) * 4;   // a lane moves 4 columns of rows prow, prow + 8, ..
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < NC; ++t) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            patch[((r & 3) + 8 * (r >> 2) + 4 * lh) * PITCH + li] = acc[i][t][r];
            acc[i][t][r] = 0.f;
          }
          __builtin_amdgcn_wave_barrier();
          const int col = m0 + (cc * NC + t) * 32 + pc4;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int pr = prow + 8 * k;
            const int64_t row = row0 + i * 32 + pr;
            float4 v = *reinterpret_cast<const float4*>(patch + pr * PITCH + pc4);
            if (row < a.N && col < a.M) {
              if (a.C) {
                const float4 c4 = *reinterpret_cast<const float4*>(a.C + row * a.ldc + col);
                v.x += c4.x; v.y += c4.y; v.z += c4.z; v.w += c4.w;
              }
              if (relu) {
                v.x = v.x <= 0.f ? 0.f : v.x; v.y = v.y <= 0.f ? 0.f : v.y;
                v.z = v.z <= 0.f ? 0.f : v.z; v.w = v.w <= 0.f ? 0.f : v.w;
              }
              *reinterpret_cast<float4*>(a.Y + row * a.ldy + col) = v;
            }
          }
          __builtin_amdgcn_wave_barrier();
        }
      return;
    }
    const int col0 = m0 + cc * NC * 32 + li;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int t = 0; t < NC; ++t) {
        const int col = col0 + t * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (col < a.M && row < a.N) {
            float v = acc[i][t][r];
            if (a.C) v += a.C[row * a.ldc + col];
            if (relu) v = v <= 0.f ? 0.f : v;
            a.Y[row * a.ldy + col] = v;
          }
          acc[i][t][r] = 0.f;
        }
      }
  };
  // The weight images go from global memory straight into LDS (no registers): every consumer wave copies its share
  // of the 1-KiB pieces of chunk i + 2 at the start of iteration i, and the barrier at the end of the iteration
  // (whose fence waits for the copies) publishes them; they are first read in iteration i + 1.
  int w_c = 2 % nch;   // chunk of the next image to copy
  auto copy_w = [&](int c, int stage) __attribute__((always_inline)) {
    const char* img = a.wimg + ((int64_t)blockIdx.y * nch + c) * WIMG + lane * 16;
    char* dst = ldsb + stage * STAGEB + kBN * ROWB;
#pragma unroll
    for (int j = 0; j < (WPIECES + 3) / 4; ++j) {
      const int piece = wid + 4 * j;
      if (piece < WPIECES)   // uniform over the wave
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(img + piece * 1024),
                                         (__attribute__((address_space(3))) void*)(dst + piece * 1024), 16, 0, 0);
    }
  };
  copy_w(0, 0);
  copy_w(1 % nch, 1);
  WS_SYNC();
  read_a(0, ldsb);
  read_a(1, ldsb);
#pragma unroll
  for (int t = 0; t < NC; ++t) read_b(t, ldsb);
  int stage = 0;
  const int64_t rounds = (total + kWsRound - 1) / kWsRound * kWsRound;   // the producers' loop runs whole rounds
  for (int64_t it = 0; it < rounds; ++it) {
    if (it >= total) {   // uniform: only the barrier of a padding iteration
      WS_SYNC();
      continue;
    }
    stage = stage + 1 == kWsStages ? 0 : stage + 1;
    const char* nb = ldsb + stage * STAGEB;   // the next chunk's stage
    copy_w(w_c, stage + 1 == kWsStages ? 0 : stage + 1);   // chunk it + 2 (past the end: a valid image, never read)
    w_c = w_c + 1 == nch ? 0 : w_c + 1;
#pragma unroll
    for (int t = 0; t < NC; ++t) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {   // smallest terms first
        f32x16 v = acc[i][t];
        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i][2], bv[t][0], v, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i][0], bv[t][2], v, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i][1], bv[t][1], v, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i][1], bv[t][0], v, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i][0], bv[t][1], v, 0, 0, 0);
        acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i][0], bv[t][0], v, 0, 0, 0);
        if (t == NC - 1) {
          __builtin_amdgcn_sched_barrier(0);   // the reads below reuse the registers these products read
          read_a(i, nb);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      read_b(t, nb);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (++c == nch) {   // the row step is complete
      c = 0;
#ifdef MPNN_WS_PROF
      const long long te = WS_NOW();
      epilogue();
      prof_epi += WS_NOW() - te;
#else
      epilogue();
#endif
    }
    WS_SYNC();
  }
#ifdef MPNN_WS_PROF
  prof_done();
#endif
}

struct TnArgs {
  const float* G; int64_t ldg;        // [N, M]
  const float* mask; int64_t ldm;     // optional [N, M]: G is used where mask > 0
  const float* X; int64_t ldx;        // [N, K]
  const float* X2; int64_t ldx2;      // optional [N, K2]: dW = G^T [X | X2], the column tiles of X2 follow those of X
  float* gout; int64_t ldgo;          // optional [N, M]: G * (mask > 0), written by the blocks of the first K tile
  float* partial;                     // [gridDim.x][M][K + K2]
  int64_t N; int M; int K; int K2;
};

// The right-hand matrix a block's column tile (blockIdx.z) lies in, the tile's first column inside it, and the
// tile's first column in dW.
struct TnTile {
  const float* X; int64_t ldx; int K; int k0; int kout0;
};
__device__ __forceinline__ TnTile tn_tile(const TnArgs& a) {
  const int z1 = (a.K + 127) / 128;
  if (a.X2 && (int)blockIdx.z >= z1) {
    const int k0 = ((int)blockIdx.z - z1) * 128;
    return TnTile{a.X2, a.ldx2, a.K2, k0, a.K + k0};
  }
  return TnTile{a.X, a.ldx, a.K, (int)blockIdx.z * 128, (int)blockIdx.z * 128};
}

#ifndef MPNN_TN_DEPTH
#define MPNN_TN_DEPTH 2
#endif
constexpr int kTnDepth = MPNN_TN_DEPTH;   // register sets of the slab ring of k_gemm_tn (split, unmasked form)

// The three bfloat16 pieces of 8 floats, each piece as one MFMA operand.
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8 (&p)[3]) {
  uint2 lo[3], hi[3];
  split4(make_float4(v[0], v[1], v[2], v[3]), lo);
  split4(make_float4(v[4], v[5], v[6], v[7]), hi);
#pragma unroll
  for (int i = 0; i < 3; ++i) p[i] = __builtin_bit_cast(bf16x8, make_uint4(lo[i].x, lo[i].y, hi[i].x, hi[i].y));
}

// Which of the block's 4 x 4 MFMA tiles (32 x 32 each) a wave computes.  Dense: the wave's 64 x 64 quadrant (slot
// s = 2 i + j is tile (2 (wid >> 1) + i, 2 (wid & 1) + j): fragments are shared inside the quadrant).  When at most
// half of the 128 x 128 tile lies inside the matrix (narrow weight gradients: 64 hidden channels, 11 or 3 input
// features) the tiles that do are dealt round-robin over the 4 waves instead, so that no wave multiplies zeros
// while another one does all the work.
template <bool SPARSE>
struct TnSlots {
  int mt, kt, wid;
  __device__ __forceinline__ bool on(int s) const { return !SPARSE || s * 4 + wid < mt * kt; }
  __device__ __forceinline__ int ti(int s) const { return SPARSE ? (on(s) ? (s * 4 + wid) / kt : 0) : 2 * (wid >> 1) + (s >> 1); }
  __device__ __forceinline__ int tj(int s) const { return SPARSE ? (on(s) ? (s * 4 + wid) % kt : 0) : 2 * (wid & 1) + (s & 1); }
};
template <bool SPARSE>
__device__ __forceinline__ TnSlots<SPARSE> tn_slots(int m_left, int k_left, int wid) {
  TnSlots<SPARSE> t;
  t.mt = m_left >= 128 ? 4 : (m_left + 31) / 32;
  t.kt = k_left >= 128 ? 4 : (k_left + 31) / 32;
  t.wid = wid;
  return t;
}
// The host picks the SPARSE instantiation when no block of the launch has more than 8 tiles inside the matrix.
inline bool tn_sparse(int M, int K, int K2) {
  const int mt = M >= 128 ? 4 : (M + 31) / 32;
  const int kw = K > K2 ? K : K2;
  const int kt = kw >= 128 ? 4 : (kw + 31) / 32;
  return mt * kt <= 8;
}

// Output tile 128 (m) x 128 (k) per block; wave (wm, wk) owns a 64 x 64 quadrant (2 x 2 MFMA tiles); 32-row slabs
// alternate between two LDS buffers with one barrier per slab.
// SPLIT: the float32 slabs lie in LDS as loaded; a lane reads the 8 contraction values of its output column with
// 4-byte LDS reads (lane half h, value j = row 2 j + h of the 16-row step: the order inside one MFMA is free as
// long as both operands share it), splits them into three bfloat16 pieces in registers and multiplies on the
// bfloat16 matrix cores (six piece products, as in k_gemm_nt).
// Row slabs run through a ring of D register sets (the slab being multiplied is in LDS, the next D - 1 are in
// flight; every load is issued unconditionally with clamped rows so that the wait counts keep the ring full).
template <bool ALIGNED, bool SPLIT, bool MASKED, bool SPARSE>
__global__ void __launch_bounds__(kBlock, 2) k_gemm_tn(TnArgs a) {
  constexpr int D = kTnDepth > 2 && SPLIT && !MASKED ? kTnDepth : 2;
  __shared__ float Gs[2][32][128];
  __shared__ float Xs[2][32][128];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = (wid >> 1) * 64, wk = (wid & 1) * 64;
  const TnTile xt = tn_tile(a);
  const int m0 = blockIdx.y * 128, k0 = xt.k0;
  const TnSlots<SPARSE> slots = tn_slots<SPARSE>(a.M - m0, xt.K - k0, wid);

  f32x16 acc0, acc1, acc2, acc3;   // slot s; dense: s = 2 i + j of the wave's quadrant
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; acc3[r] = 0.f; }

  // a thread moves 4 float4 per matrix and slab: rows tid/32 + 8 q, columns 4 (tid % 32) ..
  const int pr = tid >> 5, pc = (tid & 31) * 4;
  float4 gr[D][4], mr[MASKED ? D : 1][4], xr[D][4];
  auto fetch = [&](auto S, int64_t n0) {
    constexpr int st = decltype(S)::value;
    if constexpr (ALIGNED) {   // scalar slab base + 32-bit lane offsets (a slab spans < 2^31 elements)
      const int64_t left = a.N - 1 - n0;
      const int rt_max = left < 31 ? (left < 0 ? 0 : (int)left) : 31;
      const int64_t rbase = n0 < a.N ? n0 : a.N - 1;
      const float* gb = a.G + rbase * a.ldg;
      const float* mb = MASKED ? a.mask + rbase * a.ldm : nullptr;
      const float* xb = xt.X + rbase * xt.ldx;
      const uint32_t gc = m0 + pc < a.M ? m0 + pc : a.M - 4, xc = k0 + pc < xt.K ? k0 + pc : xt.K - 4;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t r = pr + 8 * q < rt_max ? pr + 8 * q : rt_max;
        gr[st][q] = *reinterpret_cast<const float4*>(gb + (r * (uint32_t)a.ldg + gc));
        if constexpr (MASKED) mr[st][q] = *reinterpret_cast<const float4*>(mb + (r * (uint32_t)a.ldm + gc));
        xr[st][q] = *reinterpret_cast<const float4*>(xb + (r * (uint32_t)xt.ldx + xc));
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int64_t row = n0 + pr + 8 * q;
        row = row < a.N ? row : a.N - 1;
        gr[st][q] = load_chunk4<ALIGNED>(a.G + row * a.ldg, m0 + pc, a.M);
        if constexpr (MASKED) mr[st][q] = load_chunk4<ALIGNED>(a.mask + row * a.ldm, m0 + pc, a.M);
        xr[st][q] = load_chunk4<ALIGNED>(xt.X + row * xt.ldx, k0 + pc, xt.K);
      }
    }
  };
  auto stash = [&](auto S, int buf, int64_t nslab) {
    constexpr int st = decltype(S)::value;
    const bool emit = ALIGNED && a.gout && blockIdx.z == 0;   // uniform
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = pr + 8 * q;
      const bool row_ok = nslab + r < a.N;   // rows past the end must contribute exact zeros to the sums
      float4 g = select4(gr[st][q], valid4(m0 + pc, a.M, row_ok));
      if constexpr (MASKED) g = relu_mask4(g, mr[st][q]);
      if (emit) {   // the masked gradient, one whole 512-byte row per 32 lanes
        if (row_ok && m0 + pc < a.M) *reinterpret_cast<float4*>(a.gout + (nslab + r) * a.ldgo + m0 + pc) = g;
      }
      *reinterpret_cast<float4*>(&Gs[buf][r][pc]) = g;
      *reinterpret_cast<float4*>(&Xs[buf][r][pc]) = select4(xr[st][q], valid4(k0 + pc, xt.K, row_ok));
    }
  };
  auto multiply = [&](int buf) __attribute__((always_inline)) {
    if constexpr (SPARSE) {
      auto one = [&](int sl, f32x16& c) __attribute__((always_inline)) {
        if (!slots.on(sl)) return;   // uniform over the wave
        const int cm = 32 * slots.ti(sl) + li, ck = 32 * slots.tj(sl) + li;
        if constexpr (SPLIT) {
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            float av[8], bv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              av[j] = Gs[buf][16 * t + 2 * j + lh][cm];
              bv[j] = Xs[buf][16 * t + 2 * j + lh][ck];
            }
            bf16x8 ap[3], bp[3];
            split8(av, ap);
            split8(bv, bp);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[2], bp[0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], bp[2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[1], bp[1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[1], bp[0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], bp[1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], bp[0], c, 0, 0, 0);
          }
        } else {
#pragma unroll
          for (int t = 0; t < 16; ++t) {
            const int n = 2 * t + lh;
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(Gs[buf][n][cm], Xs[buf][n][ck], c, 0, 0, 0);
          }
        }
      };
      one(0, acc0); one(1, acc1); one(2, acc2); one(3, acc3);
      return;
    }
    if constexpr (SPLIT) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        bf16x8 ap[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          float av[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) av[j] = Gs[buf][16 * t + 2 * j + lh][wm + 32 * i + li];
          split8(av, ap[i]);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float bv[8];
          bf16x8 bp[3];
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) bv[jj] = Xs[buf][16 * t + 2 * jj + lh][wk + 32 * j + li];
          split8(bv, bp);
#pragma unroll
          for (int i = 0; i < 2; ++i) {   // smallest terms first
            f32x16& c = i == 0 ? (j == 0 ? acc0 : acc1) : (j == 0 ? acc2 : acc3);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i][2], bp[0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i][0], bp[2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i][1], bp[1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i][1], bp[0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i][0], bp[1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i][0], bp[0], c, 0, 0, 0);
          }
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int n = 2 * t + lh;
        const float a0 = Gs[buf][n][wm + li], a1 = Gs[buf][n][wm + 32 + li];
        const float b0 = Xs[buf][n][wk + li], b1 = Xs[buf][n][wk + 32 + li];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc3, 0, 0, 0);
      }
    }
  };

  const int64_t step = (int64_t)gridDim.x * 32;
  const int64_t first = (int64_t)blockIdx.x * 32;
  const int64_t total = first < a.N ? (a.N - first + step - 1) / step : 0;   // slabs of this block
  auto body = [&](auto S, int64_t it) __attribute__((always_inline)) {
    constexpr int st = decltype(S)::value;   // == it % D: the set slab `it` left when it was stashed
    using Next = std::integral_constant<int, (st + 1) % D>;
    const int64_t n0 = first + it * step;
    const int buf = (int)(it & 1);
    fetch(S, n0 + D * step);
    multiply(buf);
    if (it + 1 < total) stash(Next{}, buf ^ 1, n0 + step);
    __syncthreads();
  };
  if (total > 0) {   // uniform over the block
    fetch(std::integral_constant<int, 0>{}, first);
    fetch(std::integral_constant<int, 1>{}, first + step);
    if constexpr (D > 2) fetch(std::integral_constant<int, 2>{}, first + 2 * step);
    stash(std::integral_constant<int, 0>{}, 0, first);
    __syncthreads();
    for (int64_t it = 0; it < total; it += D) {
      body(std::integral_constant<int, 0>{}, it);
      if (it + 1 < total) body(std::integral_constant<int, 1>{}, it + 1);
      if constexpr (D > 2)
        if (it + 2 < total) body(std::integral_constant<int, 2>{}, it + 2);
    }
  }
  const int ktot = a.K + a.K2;
  float* out = a.partial + (int64_t)blockIdx.x * a.M * ktot;
  auto put = [&](int sl, const f32x16& c) __attribute__((always_inline)) {
    if (!slots.on(sl)) return;
    const int cl = k0 + 32 * slots.tj(sl) + li;   // column inside the tile's matrix
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + 32 * slots.ti(sl) + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (row < a.M && cl < xt.K) out[(int64_t)row * ktot + (xt.kout0 - k0) + cl] = c[r];
    }
  };
  put(0, acc0); put(1, acc1); put(2, acc2); put(3, acc3);
}

// The split form with hardware-transposed fragment reads (aligned operands).  Slabs of 16 rows are split into
// their three bfloat16 pieces once, when they are written to LDS (piece planes [16 rows][128 columns] of bf16,
// 288-byte row pitch), and ds_read_b64_tr_b16 hands every lane 4 contraction values of its output column per
// read: inside a group of 16 lanes, lane 4a + c receives element c of the 8-byte rows addressed by lanes
// a, a + 4, a + 8, a + 12.  Lanes a + 4k therefore address row k of a 4-row block at columns 4a .. 4a + 3, and
// the group ends up with 16 consecutive columns x 4 rows.  Lane half h takes rows 8h .. 8h + 7 of the slab in two
// reads, which is the operand layout of v_mfma_f32_32x32x16_bf16.  The multiply loop has no VALU work left.
constexpr int kTrRows = 16;
constexpr int kTrPitch = 128 * 2 + 64;            // bytes; 4 consecutive rows start 16 banks apart: the 64-byte row pieces of
                                                  // a transposed read then fall on disjoint banks
constexpr int kTrPlane = kTrRows * kTrPitch;      // one piece of one matrix
constexpr int kTrStage = 6 * kTrPlane;            // 3 pieces of G, 3 pieces of X

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_ptr;

__device__ __forceinline__ bf16x8 read_tr8(const char* p) {   // rows +0..3 and +4..7 of the lane's 8
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(p));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(p + 4 * kTrPitch));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <bool MASKED, bool SPARSE>
__global__ void __launch_bounds__(kBlock, 2) k_gemm_tn_tr(TnArgs a) {
  constexpr int D = MASKED ? 3 : 4;
  __shared__ __attribute__((aligned(16))) char lds[2 * kTrStage];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = (wid >> 1) * 64, wk = (wid & 1) * 64;
  const TnTile xt = tn_tile(a);
  const int m0 = blockIdx.y * 128, k0 = xt.k0;
  const TnSlots<SPARSE> slots = tn_slots<SPARSE>(a.M - m0, xt.K - k0, wid);

  f32x16 acc0, acc1, acc2, acc3;   // slot s; dense: s = 2 i + j of the wave's quadrant
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; acc3[r] = 0.f; }

  // a thread moves 2 float4 per matrix and slab: rows tid/32 + 8 q, columns 4 (tid % 32) ..
  const int pr = tid >> 5, pc = (tid & 31) * 4;
  const bool cols_full = m0 + 128 <= a.M && k0 + 128 <= xt.K;
  const uint32_t gc = m0 + pc < a.M ? m0 + pc : a.M - 4, xc = k0 + pc < xt.K ? k0 + pc : xt.K - 4;
  float4 gr[D][2], mr[MASKED ? D : 1][2], xr[D][2];
  auto fetch = [&](auto S, int64_t n0) {   // scalar slab base + 32-bit lane offsets
    constexpr int st = decltype(S)::value;
    const int64_t left = a.N - 1 - n0;
    const int rt_max = left < kTrRows - 1 ? (left < 0 ? 0 : (int)left) : kTrRows - 1;
    const int64_t rbase = n0 < a.N ? n0 : a.N - 1;
    const float* gb = a.G + rbase * a.ldg;
    const float* mb = MASKED ? a.mask + rbase * a.ldm : nullptr;
    const float* xb = xt.X + rbase * xt.ldx;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const uint32_t r = pr + 8 * q < rt_max ? pr + 8 * q : rt_max;
      gr[st][q] = *reinterpret_cast<const float4*>(gb + (r * (uint32_t)a.ldg + gc));
      if constexpr (MASKED) mr[st][q] = *reinterpret_cast<const float4*>(mb + (r * (uint32_t)a.ldm + gc));
      xr[st][q] = *reinterpret_cast<const float4*>(xb + (r * (uint32_t)xt.ldx + xc));
    }
  };
  auto stash = [&](auto S, int buf, int64_t nslab) {
    constexpr int st = decltype(S)::value;
    const bool emit = a.gout && blockIdx.z == 0;                  // uniform
    const bool full = cols_full && nslab + kTrRows <= a.N;        // uniform: nothing to zero
    char* base = lds + buf * kTrStage;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int r = pr + 8 * q;
      float4 g = gr[st][q], x = xr[st][q];
      if (!full) {   // rows and columns past the end must contribute exact zeros to the sums
        const bool row_ok = nslab + r < a.N;
        g = select4(g, valid4(m0 + pc, a.M, row_ok));
        x = select4(x, valid4(k0 + pc, xt.K, row_ok));
      }
      if constexpr (MASKED) g = relu_mask4(g, mr[st][q]);
      if (emit && nslab + r < a.N && m0 + pc < a.M)   // the masked gradient, one whole 512-byte row per 32 lanes
        *reinterpret_cast<float4*>(a.gout + (nslab + r) * a.ldgo + m0 + pc) = g;
      uint2 gp[3], xp[3];
      split4(g, gp);
      split4(x, xp);
      char* dst = base + r * kTrPitch + pc * 2;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        *reinterpret_cast<uint2*>(dst + i * kTrPlane) = gp[i];
        *reinterpret_cast<uint2*>(dst + (3 + i) * kTrPlane) = xp[i];
      }
    }
  };
  // this lane's address inside a 32-column tile of a piece plane (see the header comment)
  const int s16 = lane & 15;
  const int frag_off = (8 * lh + (s16 >> 2)) * kTrPitch + (16 * ((lane >> 4) & 1) + 4 * (s16 & 3)) * 2;
  auto multiply = [&](int buf) __attribute__((always_inline)) {
    if constexpr (SPARSE) {
      auto one = [&](int sl, f32x16& c) __attribute__((always_inline)) {
        if (!slots.on(sl)) return;   // uniform over the wave
        const char* gb = lds + buf * kTrStage + frag_off + 64 * slots.ti(sl);
        const char* xb = lds + buf * kTrStage + 3 * kTrPlane + frag_off + 64 * slots.tj(sl);
        bf16x8 ap[3], bp[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          ap[p] = read_tr8(gb + p * kTrPlane);
          bp[p] = read_tr8(xb + p * kTrPlane);
        }
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[2], bp[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], bp[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[1], bp[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[1], bp[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], bp[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], bp[0], c, 0, 0, 0);
      };
      one(0, acc0); one(1, acc1); one(2, acc2); one(3, acc3);
      return;
    }
    const char* gbase = lds + buf * kTrStage + frag_off + wm * 2;
    const char* xbase = lds + buf * kTrStage + 3 * kTrPlane + frag_off + wk * 2;
    bf16x8 ap[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p) ap[i][p] = read_tr8(gbase + p * kTrPlane + i * 64);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bf16x8 bp[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) bp[p] = read_tr8(xbase + p * kTrPlane + j * 64);
#pragma unroll
      for (int i = 0; i < 2; ++i) {   // smallest terms first
        f32x16& c = i == 0 ? (j == 0 ? acc0 : acc1) : (j == 0 ? acc2 : acc3);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i][2], bp[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i][0], bp[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i][1], bp[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i][1], bp[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i][0], bp[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i][0], bp[0], c, 0, 0, 0);
      }
    }
  };

  const int64_t step = (int64_t)gridDim.x * kTrRows;
  const int64_t first = (int64_t)blockIdx.x * kTrRows;
  const int64_t total = first < a.N ? (a.N - first + step - 1) / step : 0;   // slabs of this block
  auto body = [&](auto S, int64_t it) __attribute__((always_inline)) {
    constexpr int st = decltype(S)::value;   // == it % D: the set slab `it` left when it was stashed
    using Next = std::integral_constant<int, (st + 1) % D>;
    const int64_t n0 = first + it * step;
    const int buf = (int)(it & 1);
    fetch(S, n0 + D * step);
    multiply(buf);
    if (it + 1 < total) stash(Next{}, buf ^ 1, n0 + step);
    __syncthreads();
  };
  if (total > 0) {   // uniform over the block
    fetch(std::integral_constant<int, 0>{}, first);
    fetch(std::integral_constant<int, 1>{}, first + step);
    fetch(std::integral_constant<int, 2>{}, first + 2 * step);
    if constexpr (D > 3) fetch(std::integral_constant<int, 3>{}, first + 3 * step);
    stash(std::integral_constant<int, 0>{}, 0, first);
    __syncthreads();
    for (int64_t it = 0; it < total; it += D) {
      body(std::integral_constant<int, 0>{}, it);
      if (it + 1 < total) body(std::integral_constant<int, 1>{}, it + 1);
      if (it + 2 < total) body(std::integral_constant<int, 2>{}, it + 2);
      if constexpr (D > 3)
        if (it + 3 < total) body(std::integral_constant<int, 3>{}, it + 3);
    }
  }
  const int ktot = a.K + a.K2;
  float* out = a.partial + (int64_t)blockIdx.x * a.M * ktot;
  auto put = [&](int sl, const f32x16& c) __attribute__((always_inline)) {
    if (!slots.on(sl)) return;
    const int cl = k0 + 32 * slots.tj(sl) + li;   // column inside the tile's matrix
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + 32 * slots.ti(sl) + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (row < a.M && cl < xt.K) out[(int64_t)row * ktot + (xt.kout0 - k0) + cl] = c[r];
    }
  };
  put(0, acc0); put(1, acc1); put(2, acc2); put(3, acc3);
}

// The same kernel with an output tile of 192 x 128 (wave (wm, wk) owns 96 x 64: 3 x 2 matrix tiles) for weight
// gradients of 129 .. 192 rows - [gA | gB | gz]^T x of a layer with 64 hidden channels.  With 128-row tiles such a
// product pads to 256 rows: a third more matrix instructions, and X is read once per row tile.  Unmasked, aligned.
constexpr int kTr192PitchG = 192 * 2 + 64;                       // bytes; 4 consecutive rows start 16 banks apart
constexpr int kTr192PlaneG = kTrRows * kTr192PitchG;
constexpr int kTr192Stage = 3 * kTr192PlaneG + 3 * kTrPlane;      // 3 pieces of G (192 wide), 3 pieces of X (128 wide)

__global__ void __launch_bounds__(kBlock, 2) k_gemm_tn_tr192(TnArgs a) {
  constexpr int D = 3;   // register sets of the slab ring (a fourth spills)
  extern __shared__ float lds_f[];
  char* const lds = reinterpret_cast<char*>(lds_f);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = (wid >> 1) * 96, wk = (wid & 1) * 64;
  const TnTile xt = tn_tile(a);
  const int m0 = blockIdx.y * 192, k0 = xt.k0;

  f32x16 acc[3][2];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // a thread moves 3 float4 of G (rows (tid + 256 q) / 48, 48 float4 to a row) and 2 of X per slab
  const int xpr = tid >> 5, xpc = (tid & 31) * 4;
  int gpr[3], gpc[3];
  uint32_t gcol[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int idx = tid + kBlock * q;
    gpr[q] = idx / 48;
    gpc[q] = (idx % 48) * 4;
    gcol[q] = m0 + gpc[q] < a.M ? m0 + gpc[q] : a.M - 4;
  }
  const uint32_t xc = k0 + xpc < xt.K ? k0 + xpc : xt.K - 4;
  const bool cols_full = m0 + 192 <= a.M && k0 + 128 <= xt.K;
  float4 gr[D][3], xr[D][2];
  auto fetch = [&](auto S, int64_t n0) {   // scalar slab base + 32-bit lane offsets, every load unconditional
    constexpr int st = decltype(S)::value;
    const int64_t left = a.N - 1 - n0;
    const int rt_max = left < kTrRows - 1 ? (left < 0 ? 0 : (int)left) : kTrRows - 1;
    const int64_t rbase = n0 < a.N ? n0 : a.N - 1;
    const float* gb = a.G + rbase * a.ldg;
    const float* xb = xt.X + rbase * xt.ldx;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const uint32_t r = gpr[q] < rt_max ? gpr[q] : rt_max;
      gr[st][q] = *reinterpret_cast<const float4*>(gb + (r * (uint32_t)a.ldg + gcol[q]));
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const uint32_t r = xpr + 8 * q < rt_max ? xpr + 8 * q : rt_max;
      xr[st][q] = *reinterpret_cast<const float4*>(xb + (r * (uint32_t)xt.ldx + xc));
    }
  };
  auto stash = [&](auto S, int buf, int64_t nslab) {
    constexpr int st = decltype(S)::value;
    const bool full = cols_full && nslab + kTrRows <= a.N;        // uniform: nothing to zero
    char* base = lds + buf * kTr192Stage;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      float4 g = gr[st][q];
      if (!full) g = select4(g, valid4(m0 + gpc[q], a.M, nslab + gpr[q] < a.N));   // exact zeros past the end
      uint2 gp[3];
      split4(g, gp);
      char* dst = base + gpr[q] * kTr192PitchG + gpc[q] * 2;
#pragma unroll
      for (int i = 0; i < 3; ++i) *reinterpret_cast<uint2*>(dst + i * kTr192PlaneG) = gp[i];
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int r = xpr + 8 * q;
      float4 x = xr[st][q];
      if (!full) x = select4(x, valid4(k0 + xpc, xt.K, nslab + r < a.N));
      uint2 xp[3];
      split4(x, xp);
      char* dst = base + 3 * kTr192PlaneG + r * kTrPitch + xpc * 2;
#pragma unroll
      for (int i = 0; i < 3; ++i) *reinterpret_cast<uint2*>(dst + i * kTrPlane) = xp[i];
    }
  };
  // this lane's address inside a 32-column tile of a piece plane (see k_gemm_tn_tr)
  const int s16 = lane & 15;
  const int col_off = (16 * ((lane >> 4) & 1) + 4 * (s16 & 3)) * 2;
  const int row_sel = 8 * lh + (s16 >> 2);
  auto read_g = [&](const char* p) {   // rows +0..3 and +4..7 of the lane's 8, G pitch
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(p));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(p + 4 * kTr192PitchG));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto multiply = [&](int buf) __attribute__((always_inline)) {
    const char* gbase = lds + buf * kTr192Stage + row_sel * kTr192PitchG + col_off + wm * 2;
    const char* xbase = lds + buf * kTr192Stage + 3 * kTr192PlaneG + row_sel * kTrPitch + col_off + wk * 2;
    bf16x8 ap[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p) ap[i][p] = read_g(gbase + p * kTr192PlaneG + i * 64);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bf16x8 bp[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) bp[p] = read_tr8(xbase + p * kTrPlane + j * 64);
#pragma unroll
      for (int i = 0; i < 3; ++i) {   // smallest terms first
        f32x16 c = acc[i][j];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i][2], bp[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i][0], bp[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i][1], bp[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i][1], bp[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i][0], bp[1], c, 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i][0], bp[0], c, 0, 0, 0);
      }
    }
  };

  const int64_t step = (int64_t)gridDim.x * kTrRows;
  const int64_t first = (int64_t)blockIdx.x * kTrRows;
  const int64_t total = first < a.N ? (a.N - first + step - 1) / step : 0;   // slabs of this block
  auto body = [&](auto S, int64_t it) __attribute__((always_inline)) {
    constexpr int st = decltype(S)::value;   // == it % D: the set slab `it` left when it was stashed
    using Next = std::integral_constant<int, (st + 1) % D>;
    const int64_t n0 = first + it * step;
    const int buf = (int)(it & 1);
    fetch(S, n0 + D * step);
    multiply(buf);
    if (it + 1 < total) stash(Next{}, buf ^ 1, n0 + step);
    __syncthreads();
  };
  if (total > 0) {   // uniform over the block
    fetch(std::integral_constant<int, 0>{}, first);
    fetch(std::integral_constant<int, 1>{}, first + step);
    fetch(std::integral_constant<int, 2>{}, first + 2 * step);
    stash(std::integral_constant<int, 0>{}, 0, first);
    __syncthreads();
    for (int64_t it = 0; it < total; it += D) {
      body(std::integral_constant<int, 0>{}, it);
      if (it + 1 < total) body(std::integral_constant<int, 1>{}, it + 1);
      if (it + 2 < total) body(std::integral_constant<int, 2>{}, it + 2);
    }
  }
  const int ktot = a.K + a.K2;
  float* out = a.partial + (int64_t)blockIdx.x * a.M * ktot;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int cl = k0 + wk + 32 * j + li;   // column inside the tile's matrix
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < a.M && cl < xt.K) out[(int64_t)row * ktot + (xt.kout0 - k0) + cl] = acc[i][j][r];
      }
    }
}

// Two output tiles per block of 8 waves, sharing one operand slab in LDS: PAIR_K = one 128-row tile of G^T
// against two 128-column tiles of [X | X2] (G and its mask are read once for both), otherwise two row tiles
// against one column tile (X is read once for both).  Waves 0-3 own the first tile, waves 4-7 the second.
// Separate blocks for the two tiles read the shared operand twice from HBM: their slabs drift apart and the
// working set of all resident blocks does not fit the L2.
constexpr int kTr2Stage = 9 * kTrPlane;   // 3 tile-wide matrices x 3 pieces

struct TnPane {   // one 128-column wide part of an operand
  const float* p; int64_t ld; int width; int c0; int out0; bool ok;
};

template <bool MASKED, bool PAIR_K>
__global__ void __launch_bounds__(2 * kBlock) k_gemm_tn_tr2(TnArgs a) {
  constexpr int NG = PAIR_K ? 1 : 2, NX = PAIR_K ? 2 : 1;
  constexpr int D = 4;
  extern __shared__ float lds_f[];
  char* const lds = reinterpret_cast<char*>(lds_f);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int tw = wid >> 2;                                  // which of the two tiles this wave owns
  const int wm = ((wid >> 1) & 1) * 64, wk = (wid & 1) * 64;

  TnPane gp[NG], xp[NX];
#pragma unroll
  for (int t = 0; t < NG; ++t) {
    const int m0 = ((PAIR_K ? 1 : 2) * (int)blockIdx.y + t) * 128;
    gp[t] = TnPane{a.G, a.ldg, a.M, m0, m0, m0 < a.M};
  }
  const int z1 = (a.K + 127) / 128, z2 = a.X2 ? (a.K2 + 127) / 128 : 0;
#pragma unroll
  for (int t = 0; t < NX; ++t) {
    const int z = (PAIR_K ? 2 : 1) * (int)blockIdx.z + t;
    if (z < z1) xp[t] = TnPane{a.X, a.ldx, a.K, z * 128, z * 128, true};
    else if (z < z1 + z2) xp[t] = TnPane{a.X2, a.ldx2, a.K2, (z - z1) * 128, a.K + (z - z1) * 128, true};
    else xp[t] = TnPane{a.X, a.ldx, a.K, 0, 0, false};
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // a thread moves one float4 per pane and slab: row tid / 32, columns 4 (tid % 32) ..
  const int pr = tid >> 5, pc = (tid & 31) * 4;
  uint32_t gcol[NG], xcol[NX];
#pragma unroll
  for (int t = 0; t < NG; ++t) gcol[t] = (gp[t].ok && gp[t].c0 + pc < gp[t].width) ? gp[t].c0 + pc : 0;
#pragma unroll
  for (int t = 0; t < NX; ++t) xcol[t] = (xp[t].ok && xp[t].c0 + pc < xp[t].width) ? xp[t].c0 + pc : 0;
  float4 gr[D][NG], mr[MASKED ? D : 1][NG], xr[D][NX];
  auto fetch = [&](auto S, int64_t n0) {   // unconditional loads, scalar slab base + 32-bit lane offsets
    constexpr int st = decltype(S)::value;
    const int64_t left = a.N - 1 - n0;
    const int rt_max = left < kTrRows - 1 ? (left < 0 ? 0 : (int)left) : kTrRows - 1;
    const int64_t rbase = n0 < a.N ? n0 : a.N - 1;
    const uint32_t r = pr < rt_max ? pr : rt_max;
#pragma unroll
    for (int t = 0; t < NG; ++t) {
      gr[st][t] = *reinterpret_cast<const float4*>(gp[t].p + rbase * gp[t].ld + (r * (uint32_t)gp[t].ld + gcol[t]));
      if constexpr (MASKED)
        mr[st][t] = *reinterpret_cast<const float4*>(a.mask + rbase * a.ldm + (r * (uint32_t)a.ldm + gcol[t]));
    }
#pragma unroll
    for (int t = 0; t < NX; ++t)
      xr[st][t] = *reinterpret_cast<const float4*>(xp[t].p + rbase * xp[t].ld + (r * (uint32_t)xp[t].ld + xcol[t]));
  };
  auto stash = [&](auto S, int buf, int64_t nslab) {
    constexpr int st = decltype(S)::value;
    const bool emit = a.gout && blockIdx.z == 0;   // uniform
    const bool row_ok = nslab + pr < a.N;          // rows and columns past the end must contribute exact zeros
    char* dst = lds + buf * kTr2Stage + pr * kTrPitch + pc * 2;
#pragma unroll
    for (int t = 0; t < NG; ++t) {
      const bool in = gp[t].ok && gp[t].c0 + pc < gp[t].width;   // widths are multiples of 4
      float4 g = select4(gr[st][t], (row_ok && in) ? 15 : 0);
      if constexpr (MASKED) g = relu_mask4(g, mr[st][t]);
      if (emit && row_ok && in)   // the masked gradient, one whole 512-byte row per 32 lanes
        *reinterpret_cast<float4*>(a.gout + (nslab + pr) * a.ldgo + gp[t].c0 + pc) = g;
      uint2 pz[3];
      split4(g, pz);
#pragma unroll
      for (int i = 0; i < 3; ++i) *reinterpret_cast<uint2*>(dst + (3 * t + i) * kTrPlane) = pz[i];
    }
#pragma unroll
    for (int t = 0; t < NX; ++t) {
      const bool in = xp[t].ok && xp[t].c0 + pc < xp[t].width;
      uint2 pz[3];
      split4(select4(xr[st][t], (row_ok && in) ? 15 : 0), pz);
#pragma unroll
      for (int i = 0; i < 3; ++i) *reinterpret_cast<uint2*>(dst + (3 * (NG + t) + i) * kTrPlane) = pz[i];
    }
  };
  const int s16 = lane & 15;
  const int frag_off = (8 * lh + (s16 >> 2)) * kTrPitch + (16 * ((lane >> 4) & 1) + 4 * (s16 & 3)) * 2;
  const int gsel = PAIR_K ? 0 : tw, xsel = PAIR_K ? tw : 0;
  auto multiply = [&](int buf) __attribute__((always_inline)) {
    const char* gbase = lds + buf * kTr2Stage + 3 * gsel * kTrPlane + frag_off + wm * 2;
    const char* xbase = lds + buf * kTr2Stage + 3 * (NG + xsel) * kTrPlane + frag_off + wk * 2;
    bf16x8 ap[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p) ap[i][p] = read_tr8(gbase + p * kTrPlane + i * 64);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bf16x8 bp[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) bp[p] = read_tr8(xbase + p * kTrPlane + j * 64);
#pragma unroll
      for (int i = 0; i < 2; ++i) {   // smallest terms first
        f32x16 c = acc[i][j];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i][2], bp[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i][0], bp[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i][1], bp[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i][1], bp[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i][0], bp[1], c, 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i][0], bp[0], c, 0, 0, 0);
      }
    }
  };

  const int64_t step = (int64_t)gridDim.x * kTrRows;
  const int64_t first = (int64_t)blockIdx.x * kTrRows;
  const int64_t total = first < a.N ? (a.N - first + step - 1) / step : 0;   // slabs of this block
  auto body = [&](auto S, int64_t it) __attribute__((always_inline)) {
    constexpr int st = decltype(S)::value;   // == it % D: the set slab `it` left when it was stashed
    using Next = std::integral_constant<int, (st + 1) % D>;
    const int64_t n0 = first + it * step;
    const int buf = (int)(it & 1);
    fetch(S, n0 + D * step);
    multiply(buf);
    if (it + 1 < total) stash(Next{}, buf ^ 1, n0 + step);
    __syncthreads();
  };
  if (total > 0) {   // uniform over the block
    fetch(std::integral_constant<int, 0>{}, first);
    fetch(std::integral_constant<int, 1>{}, first + step);
    fetch(std::integral_constant<int, 2>{}, first + 2 * step);
    fetch(std::integral_constant<int, 3>{}, first + 3 * step);
    stash(std::integral_constant<int, 0>{}, 0, first);
    __syncthreads();
    for (int64_t it = 0; it < total; it += D) {
      body(std::integral_constant<int, 0>{}, it);
      if (it + 1 < total) body(std::integral_constant<int, 1>{}, it + 1);
      if (it + 2 < total) body(std::integral_constant<int, 2>{}, it + 2);
      if (it + 3 < total) body(std::integral_constant<int, 3>{}, it + 3);
    }
  }
  const TnPane& gw = gp[gsel];
  const TnPane& xw = xp[xsel];
  if (!gw.ok || !xw.ok) return;
  const int ktot = a.K + a.K2;
  float* out = a.partial + (int64_t)blockIdx.x * a.M * ktot;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int cl = xw.c0 + wk + j * 32 + li;   // column inside the pane's matrix
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = gw.c0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < a.M && cl < xw.width) out[(int64_t)row * ktot + (xw.out0 - xw.c0) + cl] = acc[i][j][r];
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
    if (i < count) {
#pragma unroll 8
      for (int b = g; b < nslabs; b += kBlock / 64) s += partial[(int64_t)b * count + i];   // 8 loads in flight
    }
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
// blocks per CU over all output tiles), but at least 4 slabs of 32 rows per walker so that the partial tiles
// written and re-read stay small next to the operands (8 was 10-20 % slower on matrices of 30-90 thousand rows,
// whose walkers then cover half the chip; 2 is no faster than 4).
static int tn_slabs(int64_t N, int M, int K) {
  const int mtiles = M > 128 && M <= 192 ? 1 : (M + 127) / 128;   // (k_gemm_tn_tr192 takes 129 .. 192 rows as one tile)
  const int tiles = mtiles * ((K + 127) / 128);
  int64_t slabs = 512 / tiles;
  const int64_t need = ((N + 31) / 32 + 3) / 4;   // in 32-row units whatever the kernel's slab height
  if (slabs > need) slabs = need;
  if (slabs < 1) slabs = 1;
  return (int)slabs;
}

// MPNN_GEMM=f32 selects the float32 matrix-core instructions; the default is the 3-piece bfloat16 split form.
// Read on every call (one getenv) so that a test can compare the two forms inside one process.
static bool gemm_split() {
  const char* e = getenv("MPNN_GEMM");
  return !(e && (e[0] == 'f' || e[0] == 'F'));
}

// MPNN_GEMM_TN=regs keeps the split weight-gradient form that splits in registers after 4-byte LDS reads
// (kept for comparison); the default splits once on the way into LDS and reads fragments transposed.
static bool tn_lds_split() {
  const char* e = getenv("MPNN_GEMM_TN");
  return e && e[0] == 'r';
}

// MPNN_GEMM_NT=nosplit keeps the per-chunk weight split inside the kernel even when scratch is given (comparison).
static bool nt_presplit() {
  const char* e = getenv("MPNN_GEMM_NT");
  return !(e && e[0] == 'n');
}

template <bool MASKED, bool PAIR_K>
static void launch_tn_pair(dim3 grid, const TnArgs& a, hipStream_t stream) {
  // more than 64 KiB of dynamic LDS needs the attribute; set once per kernel (the call is far slower than a launch)
  // the attribute belongs to the kernel's image on ONE device: once per kernel and device, with that device current
  static bool done[64];
  int dev = 0;
  if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && !__atomic_load_n(&done[dev], __ATOMIC_ACQUIRE)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_tn_tr2<MASKED, PAIR_K>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kTr2Stage) == hipSuccess)
      __atomic_store_n(&done[dev], true, __ATOMIC_RELEASE);
    // on failure the launch below fails for want of LDS and the caller's launch check reports it
  }
  k_gemm_tn_tr2<MASKED, PAIR_K><<<grid, 2 * kBlock, 2 * kTr2Stage, stream>>>(a);
}

// MPNN_GEMM_TN=single keeps one output tile per block (kept for comparison).
static bool tn_pairs() {
  const char* e = getenv("MPNN_GEMM_TN");
  return !(e && e[0] == 's');
}

static bool nt_aligned(const NtArgs& a) {
  for (int s = 0; s < a.nseg; ++s) {
    const NtSegment& g = a.seg[s];
    if (((uintptr_t)g.X & 15) || ((uintptr_t)g.W & 15) || (g.ldx & 3) || (g.ldw & 3) || (g.K & 3)) return false;
    if (g.mask && (((uintptr_t)g.mask & 15) || (g.ldm & 3))) return false;
  }
  return true;
}

// Room for the chunk images of the weights: column blocks of at most 256 rows in either kernel (k_gemm_nt,
// k_gemm_nt_ws), whatever the block width chosen.
static int64_t nt_scratch_bytes(int M, int K1, int K2) {
  const int64_t nch = (K1 + kSplitBK - 1) / kSplitBK + (K2 + kSplitBK - 1) / kSplitBK;
  return (int64_t)((M + 255) / 256) * nch * 256 * kSplitRowB;   // (the padded rows of k_gemm_nt_ws's images)
}

// MPNN_GEMM_NT=old keeps every product on k_gemm_nt, MPNN_GEMM_NT=ws sends every product that k_gemm_nt_ws can
// take to it (comparison).  Default: the widths k_gemm_nt would pad to 256 columns (129 .. 192).
static int nt_wave_specialised() {
  const char* e = getenv("MPNN_GEMM_NT");
  return e && e[0] == 'o' ? 0 : (e && e[0] == 'w' ? 2 : 1);
}

template <int NC, bool MASKED>
static void launch_nt_ws(dim3 grid, const NtArgs& a, hipStream_t stream) {
  // more than 64 KiB of dynamic LDS needs the attribute, once per kernel and device, with that device current
  static bool done[64];
  int dev = 0;
  if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && !__atomic_load_n(&done[dev], __ATOMIC_ACQUIRE)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_nt_ws<NC, MASKED>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, ws_lds_bytes<NC>()) == hipSuccess)
      __atomic_store_n(&done[dev], true, __ATOMIC_RELEASE);
    // on failure the launch below fails for want of LDS and the caller's launch check reports it
  }
  k_gemm_nt_ws<NC, MASKED><<<grid, 512, ws_lds_bytes<NC>(), stream>>>(a);
}

// The wave-specialised kernel takes the products it is built for: aligned operands, split arithmetic, one output,
// room for the weight images, and enough columns to occupy four consumer waves.  Returns false otherwise.
static bool try_nt_ws(NtArgs& a, bool aligned, hipStream_t stream) {
  const int mode = nt_wave_specialised();
  if (!aligned || !gemm_split() || a.Y2 || !a.wimg || !mode || a.M <= 64) return false;
  const int nblk = (a.M + 255) / 256;
  const int nc = ((a.M + nblk - 1) / nblk + 63) / 64;   // 2, 3 or 4
  if (mode == 1 && nc != 3) return false;
  const int wrows = 64 * nc;
  const int nch = (a.seg[0].K + kSplitBK - 1) / kSplitBK + (a.nseg > 1 ? (a.seg[1].K + kSplitBK - 1) / kSplitBK : 0);
  int64_t steps = (a.N + kBN - 1) / kBN;
  const int64_t slots = 256 / nblk > 0 ? 256 / nblk : 1;   // one block per CU
  dim3 grid((unsigned)(steps < slots ? steps : slots), nblk);
  k_split_weights<<<nblk * nch, kBlock, 0, stream>>>(a, const_cast<char*>(a.wimg), wrows, kSplitRowB);
  a.ws_staged = (a.M & 3) == 0 && (a.ldy & 3) == 0 && ((uintptr_t)a.Y & 15) == 0 &&
                (!a.C || ((a.ldc & 3) == 0 && ((uintptr_t)a.C & 15) == 0));
  const bool has_mask = a.seg[0].mask || (a.nseg > 1 && a.seg[1].mask);
  if (nc <= 2) has_mask ? launch_nt_ws<2, true>(grid, a, stream) : launch_nt_ws<2, false>(grid, a, stream);
  else if (nc == 3) has_mask ? launch_nt_ws<3, true>(grid, a, stream) : launch_nt_ws<3, false>(grid, a, stream);
  else has_mask ? launch_nt_ws<4, true>(grid, a, stream) : launch_nt_ws<4, false>(grid, a, stream);
  return true;
}

template <int MT, int WC>
static void launch_nt(NtArgs& a, bool aligned, hipStream_t stream) {
  int64_t blocks = (a.N + kBN - 1) / kBN;
  if (blocks > 1024) blocks = 1024;
  constexpr int WROWS = 32 * MT * WC;
  dim3 grid((unsigned)blocks, (a.M + WROWS - 1) / WROWS);
  // LDS-staged epilogue only for the three-array case, and only when 16-byte accesses are legal
  const bool staged = a.Y2 && (a.M & 3) == 0 && ((a.ldy | a.ldy2 | a.ldemul) & 3) == 0 &&
                      (((uintptr_t)a.Y | (uintptr_t)a.Y2 | (uintptr_t)a.emul | (uintptr_t)a.emul8) & 15) == 0;
  const bool has_mask = a.seg[0].mask || (a.nseg > 1 && a.seg[1].mask);
  auto go = [&](auto kern, size_t lds) { kern<<<grid, kBlock * WC, lds, stream>>>(a); };
  constexpr size_t kPatches = (size_t)4 * WC * 32 * 36 * sizeof(float);   // one 32 x 32 patch (pitch 36) per wave
  constexpr size_t lds_split = (size_t)2 * (kBN + WROWS) * kSplitRowB;
  constexpr bool split_can_stage = (kBN + WROWS) * kSplitRowB >= kPatches;   // not the narrowest tile (MT = 1)
  const bool split_form = aligned && gemm_split() && (!a.Y2 || (staged && split_can_stage));
  // splitting the weights in a launch of its own pays once blocks take several 128-row steps each
  const bool pre = split_form && a.wimg != nullptr && nt_presplit() && a.N >= (int64_t)128 * 1024;
  if (pre) {
    const int nch = (a.seg[0].K + kSplitBK - 1) / kSplitBK + (a.nseg > 1 ? (a.seg[1].K + kSplitBK - 1) / kSplitBK : 0);
    k_split_weights<<<grid.y * nch, kBlock, 0, stream>>>(a, const_cast<char*>(a.wimg), WROWS, 96);
  } else {
    a.wimg = nullptr;
  }
  if (aligned && gemm_split() && !a.Y2) {
    if (pre) has_mask ? go(k_gemm_nt<MT, WC, true, false, true, true, true>, lds_split) : go(k_gemm_nt<MT, WC, true, false, true, false, true>, lds_split);
    else has_mask ? go(k_gemm_nt<MT, WC, true, false, true, true>, lds_split) : go(k_gemm_nt<MT, WC, true, false, true, false>, lds_split);
    return;
  }
  if constexpr (split_can_stage) {
    if (aligned && gemm_split() && staged) {
      if (pre) has_mask ? go(k_gemm_nt<MT, WC, true, true, true, true, true>, lds_split) : go(k_gemm_nt<MT, WC, true, true, true, false, true>, lds_split);
      else has_mask ? go(k_gemm_nt<MT, WC, true, true, true, true>, lds_split) : go(k_gemm_nt<MT, WC, true, true, true, false>, lds_split);
      return;
    }
  }
  const size_t lds = (size_t)2 * (kBN + WROWS) * kLd * sizeof(float);
  static_assert((kBN + WROWS) * kLd * sizeof(float) >= kPatches, "the staged epilogue needs one LDS buffer for its patches");
  if (staged) {
    if (aligned) go(k_gemm_nt<MT, WC, true, true, false, true>, lds);
    else go(k_gemm_nt<MT, WC, false, true, false, true>, lds);
  } else if (has_mask) {
    if (aligned) go(k_gemm_nt<MT, WC, true, false, false, true>, lds);
    else go(k_gemm_nt<MT, WC, false, false, false, true>, lds);
  } else {
    if (aligned) go(k_gemm_nt<MT, WC, true, false, false, false>, lds);
    else go(k_gemm_nt<MT, WC, false, false, false, false>, lds);
  }
}

}  // namespace mpnn

using namespace mpnn;

#ifdef MPNN_WS_PROF
extern "C" int32_t mpnn_ws_prof_buffer(long long* buffer) {   // [blocks * 8 * 4] cycle counts, or null
  MPNN_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(mpnn::g_ws_prof), &buffer, sizeof(buffer)));
  return MPNN_OK;
}
#endif

extern "C" {

int32_t mpnn_gemm_nt_f32(const float* X1, int64_t ldx1, const float* mask1, int64_t ldm1, const float* W1,
                         int64_t ldw1, int32_t K1, const float* X2, int64_t ldx2, const float* mask2, int64_t ldm2,
                         const float* W2, int64_t ldw2, int32_t K2, float* Y, int64_t ldy, const void* emul,
                         int32_t emul_kind, int64_t ldemul, float* Y2, int64_t ldy2, const float* addend,
                         int64_t ldadd, int64_t N, int32_t M, int32_t relu, void* scratch, mpnn_stream_t stream) {
  if (emul_kind != MPNN_F32 && emul_kind != MPNN_U8) return fail(MPNN_E_INVALID, "mpnn_gemm_nt_f32: emul_kind %d", emul_kind);
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
  a.Y = Y; a.ldy = ldy; a.ldemul = ldemul; a.Y2 = Y2; a.ldy2 = ldy2;
  a.emul = emul_kind == MPNN_F32 ? static_cast<const float*>(emul) : nullptr;
  a.emul8 = emul_kind == MPNN_U8 ? static_cast<const uint8_t*>(emul) : nullptr;
  a.N = N; a.M = M; a.relu = relu;
  if (addend && (ldadd < M || Y2)) return fail(MPNN_E_INVALID, "mpnn_gemm_nt_f32: addend needs ldadd >= M and no second output");
  a.C = addend; a.ldc = ldadd; a.ws_staged = 0;
  if (scratch && ((uintptr_t)scratch & 15)) return fail(MPNN_E_INVALID, "mpnn_gemm_nt_f32: scratch must be 16-byte aligned");
  a.wimg = static_cast<const char*>(scratch);
  const bool aligned = nt_aligned(a);
  if (try_nt_ws(a, aligned, stream)) {
    MPNN_LAUNCH_CHECK("k_gemm_nt_ws");
    return MPNN_OK;
  }
  if (M <= 32) launch_nt<1, 1>(a, aligned, stream);
  else if (M <= 64) launch_nt<2, 1>(a, aligned, stream);
  else if (M <= 128) launch_nt<4, 1>(a, aligned, stream);
  else launch_nt<4, 2>(a, aligned, stream);
  MPNN_LAUNCH_CHECK("k_gemm_nt");
  return MPNN_OK;
}

int64_t mpnn_gemm_nt_scratch_bytes(int32_t M, int32_t K1, int32_t K2) {
  if (M < 1 || K1 < 1 || K2 < 0) return -1;
  return nt_scratch_bytes(M, K1, K2);
}

int64_t mpnn_gemm_tn_scratch_floats(int64_t N, int32_t M, int32_t K) {   // K = K + K2 of the call
  if (N < 0 || M < 1 || K < 1) return -1;
  return (int64_t)tn_slabs(N, M, K) * M * K;   // two matrices make one tile more at most: fewer slabs, never more
}

int32_t mpnn_gemm_tn_f32(const float* G, int64_t ldg, const float* mask, int64_t ldm, const float* X, int64_t ldx,
                         const float* X2, int64_t ldx2, float* dW, int64_t lddw, float* scratch, float* Gout,
                         int64_t ldgo, int64_t N, int32_t M, int32_t K, int32_t K2, mpnn_stream_t stream) {
  if (N < 0 || M < 1 || K < 1 || K2 < 0) return fail(MPNN_E_INVALID, "mpnn_gemm_tn_f32: bad sizes");
  if (!dW) return fail(MPNN_E_INVALID, "mpnn_gemm_tn_f32: dW is null");
  const int ktot = K + K2;
  if (lddw < ktot) return fail(MPNN_E_INVALID, "mpnn_gemm_tn_f32: lddw below K + K2");
  if (N == 0) {
    MPNN_HIP_CHECK(hipMemset2DAsync(dW, (size_t)lddw * 4, 0, (size_t)ktot * 4, (size_t)M, stream));
    return MPNN_OK;
  }
  if (!G || !X || !scratch || (K2 > 0 && !X2)) return fail(MPNN_E_INVALID, "mpnn_gemm_tn_f32: null pointer argument");
  if (ldg < M || ldx < K || (mask && ldm < M) || (K2 > 0 && ldx2 < K2))
    return fail(MPNN_E_INVALID, "mpnn_gemm_tn_f32: leading dimension too small");
  TnArgs a{G, ldg, mask, ldm, X, ldx, K2 > 0 ? X2 : nullptr, ldx2, Gout, ldgo, scratch, N, M, K, K2};
  const int ztiles = (K + 127) / 128 + (K2 + 127) / 128;
  const int slabs = tn_slabs(N, M, ztiles * 128);
  const bool aligned = !(((uintptr_t)G & 15) || ((uintptr_t)X & 15) || (ldg & 3) || (ldx & 3) || (M & 3) || (K & 3) ||
                         (mask && (((uintptr_t)mask & 15) || (ldm & 3))) ||
                         (K2 > 0 && (((uintptr_t)X2 & 15) || (ldx2 & 3) || (K2 & 3))));
  if (Gout && (!aligned || ldgo < M || ((uintptr_t)Gout & 15) || (ldgo & 3)))
    return fail(MPNN_E_INVALID, "mpnn_gemm_tn_f32: Gout needs 16-byte aligned operands and ldgo >= M");
  dim3 grid(slabs, (M + 127) / 128, ztiles);
  auto go = [&](auto kern) { kern<<<grid, kBlock, 0, stream>>>(a); };
  // 129 .. 192 rows: one 192-row tile instead of two of 128 (MPNN_GEMM_TN=single keeps the 128-row tiles)
  if (aligned && gemm_split() && !tn_lds_split() && tn_pairs() && !mask && !Gout && M > 128 && M <= 192 &&
      N < ((int64_t)1 << 19)) {
    static bool done[64];
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && !__atomic_load_n(&done[dev], __ATOMIC_ACQUIRE)) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_tn_tr192), hipFuncAttributeMaxDynamicSharedMemorySize,
                              2 * kTr192Stage) == hipSuccess)
        __atomic_store_n(&done[dev], true, __ATOMIC_RELEASE);
    }
    k_gemm_tn_tr192<<<dim3(slabs, 1, ztiles), kBlock, 2 * kTr192Stage, stream>>>(a);
  } else
  // Pairing pays once a block walks many slabs (it halves the traffic of the shared operand); on short matrices
  // the 8-wave blocks' start-up and barriers cost more than the re-read, which the caches absorb there.
  if (aligned && gemm_split() && !tn_lds_split() && tn_pairs() && (ztiles > 1 || grid.y > 1) && N >= ((int64_t)1 << 19)) {
    // two tiles per 8-wave block along the dimension that has more than one (columns first: they share G and mask)
    const bool pair_k = ztiles > 1;
    dim3 grid2(slabs, pair_k ? grid.y : (grid.y + 1) / 2, pair_k ? (ztiles + 1) / 2 : ztiles);
    if (pair_k) mask ? launch_tn_pair<true, true>(grid2, a, stream) : launch_tn_pair<false, true>(grid2, a, stream);
    else mask ? launch_tn_pair<true, false>(grid2, a, stream) : launch_tn_pair<false, false>(grid2, a, stream);
  } else if (tn_sparse(M, K, K2)) {   // narrow products: the tiles inside the matrix dealt over all waves
    if (aligned && gemm_split() && !tn_lds_split()) mask ? go(k_gemm_tn_tr<true, true>) : go(k_gemm_tn_tr<false, true>);
    else if (aligned && gemm_split()) mask ? go(k_gemm_tn<true, true, true, true>) : go(k_gemm_tn<true, true, false, true>);
    else if (aligned) mask ? go(k_gemm_tn<true, false, true, true>) : go(k_gemm_tn<true, false, false, true>);
    else mask ? go(k_gemm_tn<false, false, true, true>) : go(k_gemm_tn<false, false, false, true>);
  } else if (aligned && gemm_split() && !tn_lds_split()) mask ? go(k_gemm_tn_tr<true, false>) : go(k_gemm_tn_tr<false, false>);
  else if (aligned && gemm_split()) mask ? go(k_gemm_tn<true, true, true, false>) : go(k_gemm_tn<true, true, false, false>);
  else if (aligned) mask ? go(k_gemm_tn<true, false, true, false>) : go(k_gemm_tn<true, false, false, false>);
  else mask ? go(k_gemm_tn<false, false, true, false>) : go(k_gemm_tn<false, false, false, false>);
  MPNN_LAUNCH_CHECK("k_gemm_tn");
  int64_t blocks = ((int64_t)M * ktot + 63) / 64;
  if (blocks > kMaxBlocks) blocks = kMaxBlocks;
  k_sum_partials<<<(int)blocks, kBlock, 0, stream>>>(scratch, slabs, M, ktot, dW, lddw);
  MPNN_LAUNCH_CHECK("k_sum_partials");
  return MPNN_OK;
}

}  // extern "C"
