// Tall-skinny bfloat16 GEMM of the MPNN layer for bfloat16 models (bf16 storage, float32 accumulation on
// v_mfma_f32_32x32x16_bf16):
//
//   Y[N,M] = relu?( X1[N,K1] W1[M,K1]^T + X2[N,K2] W2[M,K2]^T ),   optionally  Y2 = Y * emul  (emul float32)
//
// the node-level Linear calls of the reference (models/layers.py:56,69,104,140) and, with pre-transposed
// weights, the input-gradient products of their backward.  Same block geometry as the float32 kernel
// (128-row steps, 4 x WC waves, wave = 32 rows x 32 MT columns), 64-wide k-chunks double-buffered in LDS, loads
// one chunk ahead in registers, issued unconditionally with a scalar base and a 32-bit lane offset.
// The MFMA operand roles are swapped (accumulator = transposed output tile) so that a lane ends up with 4
// consecutive output columns per register quad and stores them as one 8-byte word.
// Requirements (checked on the host): K1, K2, M multiples of 8; rows 16-byte aligned.
#include <type_traits>

#include "common.h"
#include "rowio.h"

namespace mpnn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kBfBK = 64;                  // k-chunk (elements)
constexpr int kBfRowB = kBfBK * 2 + 16;    // LDS row: 144 bytes = 9 sixteen-byte slots (odd => conflict-free b128)
constexpr int kBfBN = 128;                 // rows of X per block step

struct NtBf16Seg {
  const bf16_t* X; int64_t ldx;   // [N, K]
  const bf16_t* W; int64_t ldw;   // [M, K]
  int K;
};

struct NtBf16Args {
  NtBf16Seg seg[2];
  int nseg;
  bf16_t* Y; int64_t ldy;
  const float* emul; int64_t ldemul;   // optional [N, M] float32: second output Y2 = Y * emul
  const uint8_t* emul8;                // the same factor stored as bytes (emul is null then)
  bf16_t* Y2; int64_t ldy2;
  int64_t N; int M;
  int relu;
};

__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));   // round to nearest even
}

template <int MT, int WC>
__global__ void __launch_bounds__(kBlock * WC, 2) k_gemm_nt_bf16(NtBf16Args a) {
  constexpr int NTHREADS = kBlock * WC;
  constexpr int WROWS = 32 * MT * WC;
  constexpr int QPR = kBfBK / 8;                                      // 16-byte loads per row and chunk
  constexpr int XQ = (kBfBN * QPR) / NTHREADS;
  constexpr int WQ = (WROWS * QPR + NTHREADS - 1) / NTHREADS;
  constexpr int STAGEB = (kBfBN + WROWS) * kBfRowB;
  extern __shared__ float lds[];
  char* const ldsb = reinterpret_cast<char*>(lds);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wr = wid & 3, wc = wid >> 2;
  const int li = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.y * WROWS;
  const int nch0 = (a.seg[0].K + kBfBK - 1) / kBfBK;
  const int nch = nch0 + (a.nseg > 1 ? (a.seg[1].K + kBfBK - 1) / kBfBK : 0);
  const int64_t nsteps_all = (a.N + kBfBN - 1) / kBfBN;
  const int64_t my_steps = blockIdx.x < nsteps_all ? (nsteps_all - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
  const int64_t total = my_steps * nch;
  if (total == 0) return;   // uniform over the block

  f32x16 acc[MT];
#pragma unroll
  for (int c = 0; c < MT; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

  uint4 xr[2][XQ], wreg[2][WQ];
  int ck[2], cK[2];                              // k origin and row length of the chunk a set holds
  int64_t f_row0 = (int64_t)blockIdx.x * kBfBN;  // position of the next fetch in the (row step, chunk) sequence
  int f_c = 0;

  auto fetch = [&](auto S) {
    constexpr int st = decltype(S)::value;
    const int c = f_c;
    const int64_t row0 = f_row0;
    if (++f_c == nch) { f_c = 0; f_row0 += (int64_t)gridDim.x * kBfBN; }
    const NtBf16Seg& sg = a.seg[c < nch0 ? 0 : 1];
    const int k0 = (c < nch0 ? c : c - nch0) * kBfBK;
    ck[st] = k0;
    cK[st] = sg.K;
    const int64_t left = a.N - 1 - row0;
    const int rt_max = left < kBfBN - 1 ? (left < 0 ? 0 : (int)left) : kBfBN - 1;
    const int64_t rbase = row0 <= a.N - 1 ? row0 : a.N - 1;
    const bf16_t* xb = sg.X + rbase * sg.ldx;
    const bf16_t* wb = sg.W + (int64_t)m0 * sg.ldw;
    const int r_max = a.M - 1 - m0 < WROWS - 1 ? a.M - 1 - m0 : WROWS - 1;
    const uint32_t ldx = (uint32_t)sg.ldx, ldw = (uint32_t)sg.ldw;
    const bool tail = k0 + kBfBK > sg.K;
#pragma unroll
    for (int q = 0; q < WQ; ++q) {   // weights first (L2): waiting for them leaves the X loads in flight
      const int idx = tid + NTHREADS * q;
      int kk = k0 + (idx % QPR) * 8;
      if (tail) kk = kk < sg.K ? kk : sg.K - 8;
      int r = idx / QPR;
      r = r < r_max ? r : r_max;
      wreg[st][q] = *reinterpret_cast<const uint4*>(wb + ((uint32_t)r * ldw + (uint32_t)kk));
    }
#pragma unroll
    for (int q = 0; q < XQ; ++q) {
      const int idx = tid + NTHREADS * q;
      int kk = k0 + (idx % QPR) * 8;
      if (tail) kk = kk < sg.K ? kk : sg.K - 8;
      int rt = idx / QPR;
      rt = rt < rt_max ? rt : rt_max;
      xr[st][q] = *reinterpret_cast<const uint4*>(xb + ((uint32_t)rt * ldx + (uint32_t)kk));
    }
  };
  auto stash = [&](auto S, int buf) {
    constexpr int st = decltype(S)::value;
    char* Xs = ldsb + buf * STAGEB;
    char* Ws = Xs + kBfBN * kBfRowB;
    const bool tail = ck[st] + kBfBK > cK[st];   // uniform: columns past the row's end must multiply as zeros
    const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int q = 0; q < XQ; ++q) {
      const int idx = tid + NTHREADS * q;
      uint4 v = xr[st][q];
      if (tail && ck[st] + (idx % QPR) * 8 >= cK[st]) v = zero;
      *reinterpret_cast<uint4*>(Xs + (idx / QPR) * kBfRowB + (idx % QPR) * 16) = v;
    }
#pragma unroll
    for (int q = 0; q < WQ; ++q) {
      const int idx = tid + NTHREADS * q;
      uint4 v = wreg[st][q];
      if (tail && ck[st] + (idx % QPR) * 8 >= cK[st]) v = zero;
      if (idx / QPR < WROWS) *reinterpret_cast<uint4*>(Ws + (idx / QPR) * kBfRowB + (idx % QPR) * 16) = v;
    }
  };

  int64_t step = 0;
  int c = 0;
  auto body = [&](auto S, int64_t it) {
    constexpr int st = decltype(S)::value;   // == it & 1: LDS buffer and the register set chunk it + 2 goes to
    using Next = std::integral_constant<int, st ^ 1>;
    const int64_t row0 = (blockIdx.x + step * gridDim.x) * kBfBN;
    const bool last_chunk = c + 1 == nch;
    fetch(S);   // chunk it + 2 into the set chunk it left when it was stashed (unconditional, see header)
    const char* Xs = ldsb + st * STAGEB;
    const char* Ws = Xs + kBfBN * kBfRowB;
    const char* xrow = Xs + (wr * 32 + li) * kBfRowB + lh * 16;
#pragma unroll
    for (int s = 0; s < kBfBK / 16; ++s) {
      const bf16x8 av = *reinterpret_cast<const bf16x8*>(xrow + s * 32);
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const bf16x8 bv = *reinterpret_cast<const bf16x8*>(Ws + ((wc * MT + t) * 32 + li) * kBfRowB + lh * 16 + s * 32);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv, av, acc[t], 0, 0, 0);
      }
    }
    if (last_chunk && (a.Y2 || MT == 4)) {
      // The tiles go through a wave-private patch of the LDS buffer that
      // has just been multiplied (free once every wave has left the loop above; its next writer is the stash of
      // the following iteration, behind this iteration's closing barrier) and leave as whole row segments,
      // 8 columns per lane.  TPB tiles share a patch when the buffer is large enough.  With the second output
      // (Y2 = Y * emul) three arrays are touched here; for a single output of 4 tiles per wave it is still the
      // faster way out - the registers hold a tile transposed, 8 bytes per lane in 32 different rows per store,
      // and the 2 GB of a 256-column product written that way took half as long again (1.00 vs 0.66 ms).
      __syncthreads();
      constexpr int TPB = (MT == 4 && WC == 1) ? 2 : 1;
      constexpr int PITCH = 32 * TPB + 4;                 // floats; 16-byte slots per row odd
      static_assert((size_t)STAGEB >= (size_t)4 * WC * 32 * PITCH * sizeof(float), "patches must fit one LDS buffer");
      constexpr int LPRW = 4 * TPB, RPP = 64 / LPRW;      // lanes per patch row, rows per pass
      float* patch = reinterpret_cast<float*>(ldsb + st * STAGEB) + wid * (32 * PITCH);
      const int prow = lane / LPRW, pc8 = (lane % LPRW) * 8;
#pragma unroll
      for (int tb = 0; tb < MT; tb += TPB) {
#pragma unroll
        for (int tt = 0; tt < TPB; ++tt) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {   // transposed tile: row = lane & 31, columns 8 i + 4 (lane >> 5) .. + 3
            float4 v = make_float4(acc[tb + tt][4 * i], acc[tb + tt][4 * i + 1], acc[tb + tt][4 * i + 2], acc[tb + tt][4 * i + 3]);
            if (a.relu) {
              v.x = v.x <= 0.f ? 0.f : v.x; v.y = v.y <= 0.f ? 0.f : v.y;
              v.z = v.z <= 0.f ? 0.f : v.z; v.w = v.w <= 0.f ? 0.f : v.w;
            }
            *reinterpret_cast<float4*>(patch + li * PITCH + tt * 32 + 8 * i + 4 * lh) = v;
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[tb + tt][r] = 0.f;
        }
        __builtin_amdgcn_wave_barrier();
        const int col = m0 + (wc * MT + tb) * 32 + pc8;
#pragma unroll
        for (int k = 0; k < 32 / RPP; ++k) {
          const int pr = prow + RPP * k;
          const int64_t row = row0 + wr * 32 + pr;
          const float4 v0 = *reinterpret_cast<const float4*>(patch + pr * PITCH + pc8);
          const float4 v1 = *reinterpret_cast<const float4*>(patch + pr * PITCH + pc8 + 4);
          if (row < a.N && col < a.M) {   // M is a multiple of 8: a lane's 8 columns are inside or outside
            *reinterpret_cast<uint4*>(a.Y + row * a.ldy + col) =
                make_uint4(pack2_bf16(v0.x, v0.y), pack2_bf16(v0.z, v0.w), pack2_bf16(v1.x, v1.y), pack2_bf16(v1.z, v1.w));
            if (!a.Y2) continue;   // uniform over the launch
            float4 e0, e1;
            if (a.emul8) {   // uniform over the launch
              const uint2 w = *reinterpret_cast<const uint2*>(a.emul8 + row * a.ldemul + col);
              e0 = make_float4((float)(w.x & 255u), (float)((w.x >> 8) & 255u), (float)((w.x >> 16) & 255u), (float)(w.x >> 24));
              e1 = make_float4((float)(w.y & 255u), (float)((w.y >> 8) & 255u), (float)((w.y >> 16) & 255u), (float)(w.y >> 24));
            } else {
              e0 = *reinterpret_cast<const float4*>(a.emul + row * a.ldemul + col);
              e1 = *reinterpret_cast<const float4*>(a.emul + row * a.ldemul + col + 4);
            }
            *reinterpret_cast<uint4*>(a.Y2 + row * a.ldy2 + col) =
                make_uint4(pack2_bf16(v0.x * e0.x, v0.y * e0.y), pack2_bf16(v0.z * e0.z, v0.w * e0.w),
                           pack2_bf16(v1.x * e1.x, v1.y * e1.y), pack2_bf16(v1.z * e1.z, v1.w * e1.w));
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      ++step;
      c = 0;
    } else if (last_chunk) {
      // transposed tile: output row = lane & 31, column within the 32-wide tile = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
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
            *reinterpret_cast<uint2*>(a.Y + row * a.ldy + col) = make_uint2(pack2_bf16(v.x, v.y), pack2_bf16(v.z, v.w));
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
      }
      ++step;
      c = 0;
    } else {
      ++c;
    }
    if (it + 1 < total) stash(Next{}, st ^ 1);
    __syncthreads();
  };

  // prologue: chunk 0 in LDS, chunk 1 in flight in set 1 (set 0 is refilled with chunk 2 by the first body)
  fetch(std::integral_constant<int, 0>{});
  fetch(std::integral_constant<int, 1>{});
  stash(std::integral_constant<int, 0>{}, 0);
  __syncthreads();
  for (int64_t it = 0; it < total; it += 2) {
    body(std::integral_constant<int, 0>{}, it);
    if (it + 1 < total) body(std::integral_constant<int, 1>{}, it + 1);
  }
}

template <int MT, int WC>
static void launch_nt_bf16(const NtBf16Args& a, hipStream_t stream) {
  int64_t blocks = (a.N + kBfBN - 1) / kBfBN;
  if (blocks > 1024) blocks = 1024;
  constexpr int WROWS = 32 * MT * WC;
  dim3 grid((unsigned)blocks, (a.M + WROWS - 1) / WROWS);
  const size_t lds = (size_t)2 * (kBfBN + WROWS) * kBfRowB;
  if (lds > 64 * 1024) {
    // the attribute belongs to the kernel's image on ONE device: set once per kernel and device, with that device
    // current (the call is far slower than a launch)
    static bool done[64];
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && !__atomic_load_n(&done[dev], __ATOMIC_ACQUIRE)) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_nt_bf16<MT, WC>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess)
        __atomic_store_n(&done[dev], true, __ATOMIC_RELEASE);
      // on failure the launch below fails for want of LDS and the caller's launch check reports it
    }
  }
  k_gemm_nt_bf16<MT, WC><<<grid, kBlock * WC, lds, stream>>>(a);
}


// ---------------------------------------------------------------------------------------------------------
// Weight-gradient product for bfloat16 models (the reductions over the rows in the backward of the Linear
// calls at models/layers.py:56,69,104,140):
//
//   dW[M,K] = (G * [mask > 0])^T X        G, mask [N,M], X [N,K] bfloat16; float32 accumulation
//
// The contraction runs over the rows, which are the slow index of both operands, while a lane of
// v_mfma_f32_32x32x16_bf16 wants 8 contraction values of one output row.  Slabs of 32 rows go to LDS as they
// lie in memory (16-byte stores) and ds_read_b64_tr_b16 hands every lane 4 contraction values of its output
// column per read: inside a group of 16 lanes, lane 4a + c receives element c of the 8-byte rows addressed by
// lanes a, a + 4, a + 8, a + 12, so lanes a + 4k address row k of a 4-row block at columns 4a .. 4a + 3 and the
// group ends up with 16 consecutive columns x 4 rows.  Lane half h takes rows 8h .. 8h + 7 of a 16-row step in
// two reads.  288-byte row pitch: 4 consecutive rows start 8 banks apart.
constexpr int kTnRows = 32;
constexpr int kTnRowB = 128 * 2 + 32;

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_ptr;

struct TnBf16Args {
  const bf16_t* G; int64_t ldg;      // [N, M]
  const bf16_t* mask; int64_t ldm;   // optional [N, M]
  const bf16_t* X; int64_t ldx;      // [N, K]
  const bf16_t* X2; int64_t ldx2;    // optional [N, K2]: dW = G^T [X | X2], the column tiles of X2 follow those of X
  bf16_t* gout; int64_t ldgo;        // optional [N, M]: G * [mask > 0], written by the blocks of the first K tile
  float* partial;                    // [gridDim.x][M][K + K2]
  int64_t N; int M; int K; int K2;
};

// Two packed bfloat16 of g where the matching entries of y are > 0 (NaN counts as > 0, like `y <= 0 ? 0 : g`).
// Both halves at once on packed 16-bit integer instructions, and no compare: a compare's result can be used two
// wait states later at the earliest, and this runs 8 times per thread and slab.
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t relu_mask2(uint32_t g, uint32_t y) {
  const u16x2 h = __builtin_bit_cast(u16x2, y);
  // as a signed half word h > 0 (positive numbers, +inf, +NaN) <=> -h is negative and h is not (-0 = 0x8000 fails both)
  const s16x2 pos = __builtin_bit_cast(s16x2, (u16x2)((u16x2)(0) - h) & ~h) >> 15;
  // -NaN: above 0xff80 (-inf); the saturating difference is 1..0x7f there and 0 everywhere else
  const u16x2 over = __builtin_elementwise_sub_sat(h, (u16x2)(0xff80));
  const s16x2 nan = __builtin_bit_cast(s16x2, (u16x2)((u16x2)(0) - over)) >> 15;
  return g & __builtin_bit_cast(uint32_t, pos | nan);
}

// 8 contraction values of one output column: rows +0..3 and +4..7 of the lane's 8.
__device__ __forceinline__ bf16x8 read_col8(const char* p) {
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(p));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(p + 4 * kTnRowB));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// Output tile 128 (m) x 128 (k) per 4 waves; wave (wm, wk) owns a 64 x 64 quadrant (2 x 2 MFMA tiles); blockIdx.x
// walks the row slabs with stride gridDim.x and leaves one partial tile.  PAIR: a block of 8 waves owns TWO column
// tiles of [X | X2] (waves 0-3 the first, 4-7 the second) against one slab of G in LDS: G and its mask are read once
// for both, where separate blocks read them once each (their slabs drift apart, and the L2 catches only part).
template <bool MASKED, bool PAIR>
__global__ void __launch_bounds__(PAIR ? 2 * kBlock : kBlock) k_gemm_tn_bf16(TnBf16Args a) {
  constexpr int NX = PAIR ? 2 : 1;   // column tiles per block
  constexpr int Q = PAIR ? 1 : 2;    // 16-byte pieces per thread, matrix and slab
  __shared__ __attribute__((aligned(16))) char Gs[2][kTnRows * kTnRowB];
  __shared__ __attribute__((aligned(16))) char Xs[2][NX][kTnRows * kTnRowB];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int tw = PAIR ? wid >> 2 : 0;   // the column tile this wave multiplies
  const int wm = ((wid >> 1) & 1) * 64, wk = (wid & 1) * 64;
  // per column tile: the right-hand matrix it lies in, its first column there and in dW
  const int z1 = (a.K + 127) / 128, z2 = a.X2 ? (a.K2 + 127) / 128 : 0;
  const bf16_t* xm[NX];
  int64_t ldxm[NX];
  int km[NX], k0[NX], kout0[NX];
  bool tile_ok[NX];
#pragma unroll
  for (int t = 0; t < NX; ++t) {
    const int zt = NX * (int)blockIdx.z + t;
    tile_ok[t] = zt < z1 + z2;
    const bool second = tile_ok[t] && zt >= z1;
    xm[t] = second ? a.X2 : a.X;
    ldxm[t] = second ? a.ldx2 : a.ldx;
    km[t] = second ? a.K2 : a.K;
    k0[t] = tile_ok[t] ? (second ? zt - z1 : zt) * 128 : 0;
    kout0[t] = second ? a.K + k0[t] : k0[t];
  }
  const int m0 = blockIdx.y * 128;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // a thread moves Q 16-byte pieces per matrix and slab: rows tid / 16 (+ 16), columns 8 (tid % 16) ..
  const int pr = tid >> 4, pc = (tid & 15) * 8;
  const bool g_ok = m0 + pc < a.M;   // widths are multiples of 8: a piece is in or out
  const int gcol = g_ok ? m0 + pc : 0;
  bool x_ok[NX];
  int xcol[NX];
#pragma unroll
  for (int t = 0; t < NX; ++t) {
    x_ok[t] = tile_ok[t] && k0[t] + pc < km[t];
    xcol[t] = x_ok[t] ? k0[t] + pc : 0;
  }
  uint4 gr[Q], mr[Q], xr[Q][NX];
  bool rv[Q];
  auto fetch = [&](int64_t n0) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      int64_t row = n0 + pr + 16 * q;
      rv[q] = row < a.N;   // rows past the end must contribute exact zeros to the sums
      row = rv[q] ? row : a.N - 1;
      gr[q] = *reinterpret_cast<const uint4*>(a.G + row * a.ldg + gcol);
      if constexpr (MASKED) mr[q] = *reinterpret_cast<const uint4*>(a.mask + row * a.ldm + gcol);
#pragma unroll
      for (int t = 0; t < NX; ++t) xr[q][t] = *reinterpret_cast<const uint4*>(xm[t] + row * ldxm[t] + xcol[t]);
    }
  };
  auto stash = [&](int buf, int64_t nslab) __attribute__((always_inline)) {
    const bool emit = a.gout && blockIdx.z == 0;   // uniform
    // (zeroing by an all-ones / all-zeros word per piece: the compiler turns a select between two 16-byte values
    // into a two-entry table in scratch memory, read back with a lane-indexed load for every piece)
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const uint32_t gk = (rv[q] && g_ok) ? ~0u : 0u;
      uint4 g = make_uint4(gr[q].x & gk, gr[q].y & gk, gr[q].z & gk, gr[q].w & gk);
      if constexpr (MASKED) {
        g.x = relu_mask2(g.x, mr[q].x); g.y = relu_mask2(g.y, mr[q].y);
        g.z = relu_mask2(g.z, mr[q].z); g.w = relu_mask2(g.w, mr[q].w);
      }
      const int r = pr + 16 * q;
      if (emit && rv[q] && g_ok)   // the masked gradient, one whole 256-byte row per 16 lanes
        *reinterpret_cast<uint4*>(a.gout + (nslab + r) * a.ldgo + gcol) = g;
      *reinterpret_cast<uint4*>(&Gs[buf][r * kTnRowB + pc * 2]) = g;
#pragma unroll
      for (int t = 0; t < NX; ++t) {
        const uint32_t xk = (rv[q] && x_ok[t]) ? ~0u : 0u;
        *reinterpret_cast<uint4*>(&Xs[buf][t][r * kTnRowB + pc * 2]) =
            make_uint4(xr[q][t].x & xk, xr[q][t].y & xk, xr[q][t].z & xk, xr[q][t].w & xk);
      }
    }
  };

  // this lane's address inside a 32-column tile of a 16-row step (see the header comment)
  const int s16 = lane & 15;
  const int frag_off = (8 * lh + (s16 >> 2)) * kTnRowB + (16 * ((lane >> 4) & 1) + 4 * (s16 & 3)) * 2;
  const int64_t step = (int64_t)gridDim.x * kTnRows;
  int64_t n0 = (int64_t)blockIdx.x * kTnRows;
  int buf = 0;
  if (n0 < a.N) {
    fetch(n0);
    stash(0, n0);
  }
  __syncthreads();
  for (; n0 < a.N; n0 += step, buf ^= 1) {
    const bool more = n0 + step < a.N;
    if (more) fetch(n0 + step);
    const char* gp = &Gs[buf][frag_off + wm * 2];
    const char* xp = &Xs[buf][tw][frag_off + wk * 2];
#pragma unroll
    for (int s = 0; s < kTnRows / 16; ++s) {
      const bf16x8 a0 = read_col8(gp + s * 16 * kTnRowB), a1 = read_col8(gp + s * 16 * kTnRowB + 64);
      const bf16x8 b0 = read_col8(xp + s * 16 * kTnRowB), b1 = read_col8(xp + s * 16 * kTnRowB + 64);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (more) stash(buf ^ 1, n0 + step);
    __syncthreads();
  }
  // this wave's column tile (scalars picked by a select: an indexed local array would live in scratch memory)
  const bool my_ok = PAIR && tw ? tile_ok[NX - 1] : tile_ok[0];
  const int my_k0 = PAIR && tw ? k0[NX - 1] : k0[0], my_km = PAIR && tw ? km[NX - 1] : km[0];
  const int my_kout0 = PAIR && tw ? kout0[NX - 1] : kout0[0];
  if (!my_ok) return;
  const int ktot = a.K + a.K2;
  float* out = a.partial + (int64_t)blockIdx.x * a.M * ktot;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int cl = my_k0 + wk + j * 32 + li;   // column inside the tile's matrix
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < a.M && cl < my_km) out[(int64_t)row * ktot + (my_kout0 - my_k0) + cl] = acc[i][j][r];
      }
    }
}

// dW[m, k] = sum over slabs of partial[b][m][k] in a fixed order (wave w of a block adds the slabs w, w + 4, ..
// ascending for 64 consecutive outputs, then the four wave sums are added in wave order), rounded once to bfloat16.
__global__ void __launch_bounds__(kBlock) k_sum_partials_bf16(const float* __restrict__ partial, int nslabs, int M, int K,
                                                              bf16_t* __restrict__ dW, int64_t ldw) {
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
      dW[(i / K) * ldw + (i % K)] = f32_to_bf16(tot);
    }
    __syncthreads();
  }
}

// Row-slab walkers (gridDim.x): four 40 KiB-LDS blocks per CU over all output tiles, but at least 8 slabs per
// walker so that the partial tiles stay small next to the operands.
static int tn_bf16_slabs(int64_t N, int M, int K) {
  const int tiles = ((M + 127) / 128) * ((K + 127) / 128);
  int64_t slabs = 1024 / tiles;
  const int64_t need = ((N + kTnRows - 1) / kTnRows + 7) / 8;
  if (slabs > need) slabs = need;
  if (slabs < 1) slabs = 1;
  return (int)slabs;
}

}  // namespace mpnn

using namespace mpnn;

extern "C" int32_t mpnn_gemm_nt_bf16(const void* X1, int64_t ldx1, const void* W1, int64_t ldw1, int32_t K1,
                                     const void* X2, int64_t ldx2, const void* W2, int64_t ldw2, int32_t K2, void* Y,
                                     int64_t ldy, const void* emul, int32_t emul_kind, int64_t ldemul, void* Y2,
                                     int64_t ldy2, int64_t N, int32_t M, int32_t relu, mpnn_stream_t stream) {
  if (emul_kind != MPNN_F32 && emul_kind != MPNN_U8) return fail(MPNN_E_INVALID, "mpnn_gemm_nt_bf16: emul_kind %d", emul_kind);
  if (emul_kind == MPNN_U8 && (ldemul & 7)) return fail(MPNN_E_INVALID, "mpnn_gemm_nt_bf16: byte emul rows must be 8-byte aligned");
  if (N < 0 || M < 8 || K1 < 8 || K2 < 0) return fail(MPNN_E_INVALID, "mpnn_gemm_nt_bf16: bad sizes");
  if ((M | K1 | K2) & 7) return fail(MPNN_E_INVALID, "mpnn_gemm_nt_bf16: M, K1 and K2 must be multiples of 8");
  if (N == 0) return MPNN_OK;
  if (!X1 || !W1 || !Y || (K2 > 0 && (!X2 || !W2))) return fail(MPNN_E_INVALID, "mpnn_gemm_nt_bf16: null pointer argument");
  if (ldx1 < K1 || ldw1 < K1 || ldy < M || (K2 > 0 && (ldx2 < K2 || ldw2 < K2)))
    return fail(MPNN_E_INVALID, "mpnn_gemm_nt_bf16: leading dimension too small");
  if (Y2 && (!emul || ldemul < M || ldy2 < M)) return fail(MPNN_E_INVALID, "mpnn_gemm_nt_bf16: Y2 needs emul and ld >= M");
  const uintptr_t ptrs = (uintptr_t)X1 | (uintptr_t)W1 | (uintptr_t)X2 | (uintptr_t)W2 | (uintptr_t)emul;
  if (Y2 && ((((uintptr_t)Y | (uintptr_t)Y2) & 15) || ((ldy | ldy2) & 7)))
    return fail(MPNN_E_INVALID, "mpnn_gemm_nt_bf16: with Y2, Y and Y2 must be 16-byte aligned with row strides that are multiples of 8");
  if ((ptrs & 15) || (((uintptr_t)Y | (uintptr_t)Y2) & 7) || ((ldx1 | ldw1 | ldx2 | ldw2) & 7) || ((ldy | ldy2 | ldemul) & 3))
    return fail(MPNN_E_INVALID, "mpnn_gemm_nt_bf16: operands must be 16-byte aligned with row strides that keep them so");
  NtBf16Args a;
  a.seg[0] = NtBf16Seg{static_cast<const bf16_t*>(X1), ldx1, static_cast<const bf16_t*>(W1), ldw1, K1};
  a.seg[1] = NtBf16Seg{static_cast<const bf16_t*>(X2), ldx2, static_cast<const bf16_t*>(W2), ldw2, K2};
  a.nseg = K2 > 0 ? 2 : 1;
  a.Y = static_cast<bf16_t*>(Y); a.ldy = ldy;
  a.emul = emul_kind == MPNN_F32 ? static_cast<const float*>(emul) : nullptr;
  a.emul8 = emul_kind == MPNN_U8 ? static_cast<const uint8_t*>(emul) : nullptr;
  a.ldemul = ldemul;
  a.Y2 = static_cast<bf16_t*>(Y2); a.ldy2 = ldy2;
  a.N = N; a.M = M; a.relu = relu;
  if (M <= 32) launch_nt_bf16<1, 1>(a, stream);
  else if (M <= 64) launch_nt_bf16<2, 1>(a, stream);
  else if (M <= 128) launch_nt_bf16<4, 1>(a, stream);
  else launch_nt_bf16<4, 2>(a, stream);
  MPNN_LAUNCH_CHECK("k_gemm_nt_bf16");
  return MPNN_OK;
}

extern "C" int64_t mpnn_gemm_tn_bf16_scratch_floats(int64_t N, int32_t M, int32_t K) {
  if (N < 0 || M < 1 || K < 1) return -1;
  return (int64_t)tn_bf16_slabs(N, M, K) * M * K;
}

extern "C" int32_t mpnn_gemm_tn_bf16(const void* G, int64_t ldg, const void* mask, int64_t ldm, const void* X,
                                     int64_t ldx, const void* X2, int64_t ldx2, void* dW, int64_t lddw, float* scratch,
                                     void* Gout, int64_t ldgo, int64_t N, int32_t M, int32_t K, int32_t K2,
                                     mpnn_stream_t stream) {
  if (N < 0 || M < 8 || K < 8 || K2 < 0) return fail(MPNN_E_INVALID, "mpnn_gemm_tn_bf16: bad sizes");
  if ((M | K | K2) & 7) return fail(MPNN_E_INVALID, "mpnn_gemm_tn_bf16: M, K and K2 must be multiples of 8");
  if (!dW) return fail(MPNN_E_INVALID, "mpnn_gemm_tn_bf16: dW is null");
  const int ktot = K + K2;
  if (lddw < ktot) return fail(MPNN_E_INVALID, "mpnn_gemm_tn_bf16: lddw below K + K2");
  if (N == 0) {
    MPNN_HIP_CHECK(hipMemset2DAsync(dW, (size_t)lddw * 2, 0, (size_t)ktot * 2, (size_t)M, stream));
    return MPNN_OK;
  }
  if (!G || !X || !scratch || (K2 > 0 && !X2)) return fail(MPNN_E_INVALID, "mpnn_gemm_tn_bf16: null pointer argument");
  if (ldg < M || ldx < K || (mask && ldm < M) || (Gout && ldgo < M) || (K2 > 0 && ldx2 < K2))
    return fail(MPNN_E_INVALID, "mpnn_gemm_tn_bf16: leading dimension too small");
  if (K2 == 0) { X2 = nullptr; ldx2 = 0; }
  const uintptr_t ptrs = (uintptr_t)G | (uintptr_t)mask | (uintptr_t)X | (uintptr_t)X2 | (uintptr_t)Gout;
  if ((ptrs & 15) || ((ldg | ldm | ldx | ldx2 | ldgo) & 7))
    return fail(MPNN_E_INVALID, "mpnn_gemm_tn_bf16: operands must be 16-byte aligned with row strides that keep them so");
  TnBf16Args a{static_cast<const bf16_t*>(G), ldg, static_cast<const bf16_t*>(mask), ldm, static_cast<const bf16_t*>(X), ldx,
               static_cast<const bf16_t*>(X2), ldx2, static_cast<bf16_t*>(Gout), ldgo, scratch, N, M, K, K2};
  const int ztiles = (K + 127) / 128 + (K2 + 127) / 128;
  const int slabs = tn_bf16_slabs(N, M, ztiles * 128);
  if (ztiles >= 2) {   // blocks of 8 waves that own two column tiles each and read G (and its mask) once for both
    dim3 grid(slabs, (M + 127) / 128, (ztiles + 1) / 2);
    if (mask) k_gemm_tn_bf16<true, true><<<grid, 2 * kBlock, 0, stream>>>(a);
    else k_gemm_tn_bf16<false, true><<<grid, 2 * kBlock, 0, stream>>>(a);
  } else {
    dim3 grid(slabs, (M + 127) / 128, ztiles);
    if (mask) k_gemm_tn_bf16<true, false><<<grid, kBlock, 0, stream>>>(a);
    else k_gemm_tn_bf16<false, false><<<grid, kBlock, 0, stream>>>(a);
  }
  MPNN_LAUNCH_CHECK("k_gemm_tn_bf16");
  int64_t blocks = ((int64_t)M * ktot + 63) / 64;
  if (blocks > kMaxBlocks) blocks = kMaxBlocks;
  k_sum_partials_bf16<<<(int)blocks, kBlock, 0, stream>>>(scratch, slabs, M, ktot, static_cast<bf16_t*>(dW), lddw);
  MPNN_LAUNCH_CHECK("k_sum_partials_bf16");
  return MPNN_OK;
}
