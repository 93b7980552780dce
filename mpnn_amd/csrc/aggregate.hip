// Fused message + sum aggregation over a destination-sorted CSR, and its backward over the
// source-sorted transpose, for gfx950 (wave64).
//
// Replaces, for one layer, the gather -> message() -> scatter_add_ chain that PyG's
// MessagePassing.propagate() runs for the reference (call sites models/layers.py:38,110;
// message bodies models/layers.py:56,128) and the autograd backward of that chain.
//
// Work layout
//   * A feature row of H elements is owned by a GROUP of LPR = 2^k lanes (k chosen so that
//     LPR * VEC >= min(H, 64 * VEC)); each lane moves VEC consecutive columns with one 16-byte
//     access (4 floats or 8 bf16).  A wave holds 64 / LPR groups, a 256-thread block 4 waves.
//   * A group walks one work item (a CSR row, or a bounded piece of a long row) edge by edge
//     and keeps the running sum in registers, so the sum order inside an item is the CSR
//     (= original edge) order and no atomics are needed.  Pieces of a split row go to a
//     float scratch slab and are combined in piece order by k_reduce_split.
//   * Source indices of an item are fetched LPR at a time with one coalesced load and handed
//     round the group with ds_bpermute; U row gathers per lane are kept in flight.
#include <stdlib.h>

#include <type_traits>

#include "edgeops.h"

#ifndef MPNN_FWD_WAVES
#define MPNN_FWD_WAVES 4  // minimum waves per SIMD the forward kernel is compiled for (register budget)
#endif

namespace mpnn {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// Streams that are read once (index lists, mask words) can bypass the cache hierarchy's retention so that the
// gathered rows, some of which are re-read often (targets of high in-degree), keep the L2.
#define NT_LOAD(p) __builtin_nontemporal_load(p)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// edge_term for pairs of columns, as vector arithmetic (v_pk_mul_f32 / v_pk_fma_f32): per column the same multiply
// and fma chain in f order.
template <int VEC, int FE>
__device__ __forceinline__ void edge_term_pairs(const float (&w)[VEC][FE], const float (&e)[FE], float (&ce)[VEC]) {
#pragma unroll
  for (int q = 0; q < VEC / 2; ++q) {
    f32x2 s = f32x2{w[2 * q][0], w[2 * q + 1][0]} * f32x2{e[0], e[0]};
#pragma unroll
    for (int f = 1; f < FE; ++f)
      s = __builtin_elementwise_fma(f32x2{w[2 * q][f], w[2 * q + 1][f]}, f32x2{e[f], e[f]}, s);
    ce[2 * q] = s.x;
    ce[2 * q + 1] = s.y;
  }
}

// x[v] = relu(x[v]) and bits = bits << 8 | (bit 7 - v = [x[v] > 0]) for the 8 columns of a lane, as 24 instructions:
// 8 compares into 8 scalar register pairs, 8 selects, 8 add-with-carry (bits + bits + carry shifts the compare's
// result in).  Written out because on gfx950 a compare's scalar result may be read by a vector instruction only two
// wait states later (the compiler pads each compare-select pair with s_nop 1, which doubles its cost); with the
// compares grouped, every reader is at least 8 instructions behind its compare and nothing is padded.
// The select keeps x unless 0 >= x, so NaN propagates as in torch.relu; a NaN message counts as positive here.
__device__ __forceinline__ void relu_shift_in8(float (&x)[8], uint32_t& bits) {
  unsigned long long c0, c1, c2, c3, c4, c5, c6, c7, carry_out;
  asm("v_cmp_nge_f32_e64 %[c0], 0, %[x0]\n\t"
      "v_cmp_nge_f32_e64 %[c1], 0, %[x1]\n\t"
      "v_cmp_nge_f32_e64 %[c2], 0, %[x2]\n\t"
      "v_cmp_nge_f32_e64 %[c3], 0, %[x3]\n\t"
      "v_cmp_nge_f32_e64 %[c4], 0, %[x4]\n\t"
      "v_cmp_nge_f32_e64 %[c5], 0, %[x5]\n\t"
      "v_cmp_nge_f32_e64 %[c6], 0, %[x6]\n\t"
      "v_cmp_nge_f32_e64 %[c7], 0, %[x7]\n\t"
      "v_cndmask_b32_e64 %[x0], 0, %[x0], %[c0]\n\t"
      "v_cndmask_b32_e64 %[x1], 0, %[x1], %[c1]\n\t"
      "v_cndmask_b32_e64 %[x2], 0, %[x2], %[c2]\n\t"
      "v_cndmask_b32_e64 %[x3], 0, %[x3], %[c3]\n\t"
      "v_cndmask_b32_e64 %[x4], 0, %[x4], %[c4]\n\t"
      "v_cndmask_b32_e64 %[x5], 0, %[x5], %[c5]\n\t"
      "v_cndmask_b32_e64 %[x6], 0, %[x6], %[c6]\n\t"
      "v_cndmask_b32_e64 %[x7], 0, %[x7], %[c7]\n\t"
      "v_addc_co_u32_e64 %[b], %[co], %[b], %[b], %[c0]\n\t"
      "v_addc_co_u32_e64 %[b], %[co], %[b], %[b], %[c1]\n\t"
      "v_addc_co_u32_e64 %[b], %[co], %[b], %[b], %[c2]\n\t"
      "v_addc_co_u32_e64 %[b], %[co], %[b], %[b], %[c3]\n\t"
      "v_addc_co_u32_e64 %[b], %[co], %[b], %[b], %[c4]\n\t"
      "v_addc_co_u32_e64 %[b], %[co], %[b], %[b], %[c5]\n\t"
      "v_addc_co_u32_e64 %[b], %[co], %[b], %[b], %[c6]\n\t"
      "v_addc_co_u32_e64 %[b], %[co], %[b], %[b], %[c7]"
      : [x0] "+v"(x[0]), [x1] "+v"(x[1]), [x2] "+v"(x[2]), [x3] "+v"(x[3]), [x4] "+v"(x[4]), [x5] "+v"(x[5]),
        [x6] "+v"(x[6]), [x7] "+v"(x[7]), [b] "+v"(bits), [c0] "=&s"(c0), [c1] "=&s"(c1), [c2] "=&s"(c2),
        [c3] "=&s"(c3), [c4] "=&s"(c4), [c5] "=&s"(c5), [c6] "=&s"(c6), [c7] "=&s"(c7), [co] "=&s"(carry_out));
}

// Byte transpose across the 4 W lanes of a group (W = 2 or 4 words per lane): in, byte k of lane l's words = M[l][k];
// out, byte l of lane k's words = M[l][k].  Exchanges with the lanes 8 (W = 4 only), 4, 2 and 1 apart: word pairs,
// words, half words, bytes.
template <int W>
__device__ __forceinline__ void transpose_bytes(uint32_t (&w)[W], int lig) {
  constexpr int LANES = 4 * W;
  if constexpr (W == 4) {
    const bool b8 = lig & 8;
    const uint32_t g0 = __shfl_xor(b8 ? w[0] : w[2], 8, LANES), g1 = __shfl_xor(b8 ? w[1] : w[3], 8, LANES);
    if (b8) { w[0] = g0; w[1] = g1; } else { w[2] = g0; w[3] = g1; }
  }
  const bool b4 = lig & 4, b2 = lig & 2, b1 = lig & 1;
#pragma unroll
  for (int p = 0; p < W; p += 2) {
    const uint32_t got = __shfl_xor(b4 ? w[p] : w[p + 1], 4, LANES);
    if (b4) w[p] = got; else w[p + 1] = got;
  }
  // v_perm_b32: selector values 0-3 pick bytes of the second operand (own), 4-7 of the first (the partner's)
  const uint32_t sel2 = b2 ? 0x03020706u : 0x05040100u;   // [own 0 1 | partner 0 1]  or  [partner 2 3 | own 2 3]
  const uint32_t sel1 = b1 ? 0x03070105u : 0x06020400u;   // [own 0, partner 0, own 2, partner 2]  or  [partner 1, own 1, partner 3, own 3]
#pragma unroll
  for (int p = 0; p < W; ++p) w[p] = __builtin_amdgcn_perm(__shfl_xor(w[p], 2, LANES), w[p], sel2);
#pragma unroll
  for (int p = 0; p < W; ++p) w[p] = __builtin_amdgcn_perm(__shfl_xor(w[p], 1, LANES), w[p], sel1);
}

// VEC counts as bytes (saturated at 255): one 4- or 8-byte store when the lane owns that many columns.
template <int VEC>
__device__ __forceinline__ void store_count_bytes(uint8_t* p, const int (&cn)[VEC]) {
  if constexpr (VEC % 4 == 0) {
#pragma unroll
    for (int q = 0; q < VEC / 4; ++q) {
      uint32_t w = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = cn[4 * q + k];
        w |= (uint32_t)(c > 255 ? 255 : c) << (8 * k);
      }
      reinterpret_cast<uint32_t*>(p)[q] = w;
    }
  } else {
#pragma unroll
    for (int v = 0; v < VEC; ++v) p[v] = (uint8_t)(cn[v] > 255 ? 255 : cn[v]);
  }
}

template <typename T>
struct FwdArgs {
  const T* A; int64_t ldA;
  const T* B; int64_t ldB;
  const T* ea; int fe;
  const T* We;
  const int32_t* col;
  const int4* items; int64_t n_items;
  T* out; int64_t ldo;
  uint8_t* cnt;        // [N][H] bytes: positive messages per target row and column, saturated at 255
  float* cnt_big;      // [n_split + n_big][H]: the exact counts of the rows that can exceed 255 (see mpnn_amd.h)
  int64_t first_plain; // index of the first item of an unsplit row (= number of partial slots)
  int64_t n_big;       // unsplit rows with more than 255 edges: the items first_plain .. first_plain + n_big - 1
  int64_t n_split;
  float* part_out;
  float* part_cnt;
  uint32_t* maskbits;  // optional [E][mpnn_mask_words] words: bit `lig` of word v = (message column lig*VEC+v of this
                       // edge > 0); 16-bit words, two per 32-bit word, when a row takes at most 16 lanes of 8 columns;
                       // TRAIN >= 3 (8 or 16 lanes of 8 columns): byte `lig` of the edge's 8 or 16, bit 7 - v =
                       // column lig*8+v
  int H; int lpr_log2;
};

// TRAIN: 0 = output only, 1 = + positive-message counts, 2 = + relu mask bits per edge (votes across the lanes),
// 3 / 4 = the same for rows of 8 / 16 lanes x 8 columns with every lane keeping the bits of its own columns (one
// byte per edge and lane; the counts are taken from the bits, word by word).
// Waves per SIMD the kernel is compiled for: the per-lane edge weights take VEC * FE registers, so the widest
// instantiation (bfloat16 rows with the generic edge width) gets a larger register budget instead of spilling
// hundreds of bytes.  (bfloat16 rows with Fe = 4 are faster at 4 waves with a 36-byte spill than at 3 without.)
constexpr int fwd_waves(int vec, int fe) { return vec * fe >= 64 ? 2 : MPNN_FWD_WAVES; }

template <typename T, int VEC, int FE, bool RTFE, int TRAIN, int U>
__global__ void __launch_bounds__(kBlock, fwd_waves(VEC, FE)) k_agg_fwd(FwdArgs<T> a) {
  const int lane = threadIdx.x & 63;
  const int lpr = 1 << a.lpr_log2;
  const int lig = lane & (lpr - 1);
  const int grp = lane >> a.lpr_log2;
  const int gpw = 64 >> a.lpr_log2;
  const int c0 = blockIdx.y * (lpr * VEC) + lig * VEC;
  const bool act = c0 < a.H;
  const int cc = act ? c0 : 0;  // inactive lanes shadow column 0 and never store

  const int gfirst = grp << a.lpr_log2;   // first lane of this lane group

  float w[VEC][FE];
  load_edge_weights<T, VEC, FE>(a.We, a.fe, c0, act, w);
  const int ldb = (int)a.ldB;   // (checked on the host) row index times a 32-bit stride: one multiply-add per gather,
                                // where the 64-bit stride takes three quarter-rate multiplies

  // TRAIN >= 3: the mask words of a finished chunk wait here (lane l: its own bytes of the chunk's edges) until the
  // next chunk's gathers are in flight; the exchange that turns them into per-edge records (three dependent trips
  // through the LDS crossbar) then runs in the shadow of those gathers instead of at the end of its own chunk.
  constexpr int OW = TRAIN == 3 ? 2 : (TRAIN == 4 ? 4 : 0);   // mask words per lane and chunk (4 edges each)
  constexpr bool OWN = OW > 0;
  uint32_t pend[OWN ? OW : 1];
#pragma unroll
  for (int k = 0; k < (OWN ? OW : 1); ++k) pend[k] = 0u;
  int pend_base = 0, pend_nb = 0;
  auto flush_mask = [&]() __attribute__((always_inline)) {
    if constexpr (OWN) {
      // lane k of the group gets the bytes of edge pend_base + k (byte l from lane l): one coalesced 8- or 16-byte
      // store per edge
      transpose_bytes<OW>(pend, lig);
      if (lig < pend_nb) {
        uint32_t* mp = a.maskbits + (int64_t)(pend_base + lig) * OW;
        if constexpr (OW == 2) {
          u32x2 w2 = {pend[0], pend[1]};
          __builtin_nontemporal_store(w2, reinterpret_cast<u32x2*>(mp));
        } else {
          u32x4 w4 = {pend[0], pend[1], pend[2], pend[3]};
          __builtin_nontemporal_store(w4, reinterpret_cast<u32x4*>(mp));
        }
      }
      pend_nb = 0;
    }
  };

  const int64_t wave = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const int64_t stride = (int64_t)gridDim.x * (kBlock / 64) * gpw;
  for (int64_t tb = wave * gpw; tb < a.n_items; tb += stride) {
    const int64_t t = tb + grp;
    const bool valid = t < a.n_items;
    int4 it = valid ? a.items[t] : make_int4(0, 0, 0, -1);
    const int row = it.x, e0 = it.y, e1 = it.z;

    float ar[VEC], acc[VEC];
    int cn[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) { ar[v] = 0.f; acc[v] = 0.f; cn[v] = 0; }
    if (valid) RowIO<T, VEC>::load(a.A + (int64_t)row * a.ldA + cc, ar);

    // Lane `lig` of the group fetches the source index and the attributes of edge base + lig (two coalesced
    // streams, one chunk ahead); the inner loop hands them round with ds_bpermute, so its only memory
    // instructions are the U row gathers and nothing delays their issue.
    int nxt = 0;
    float enx[FE];
#pragma unroll
    for (int f = 0; f < FE; ++f) enx[f] = 0.f;
    if (e0 + lig < e1) {
      nxt = NT_LOAD(a.col + e0 + lig);
      load_edge_attr<T, FE, RTFE>(a.ea, (int64_t)(e0 + lig), a.fe, enx);
    }
    for (int base = e0; base < e1; base += lpr) {
      const int cur = nxt;
      float ecur[FE];
#pragma unroll
      for (int f = 0; f < FE; ++f) ecur[f] = enx[f];
      const int nb = (e1 - base) < lpr ? (e1 - base) : lpr;
      const int pn = base + lpr + lig;
      if (pn < e1) {
        nxt = NT_LOAD(a.col + pn);
        load_edge_attr<T, FE, RTFE>(a.ea, (int64_t)pn, a.fe, enx);
      }
      uint32_t mb[VEC < 4 ? 4 : VEC];  // relu mask words of edge base + lig, collected over the chunk
#pragma unroll
      for (int v = 0; v < (VEC < 4 ? 4 : VEC); ++v) mb[v] = 0u;
      // OWN: the bits of the lane's own columns, 8 per edge, shifted in from below (edge j + u of a group of 4 ends in
      // byte 3 - u, column v in bit 7 - v of its byte); own_w[k]: edges 4 k .. 4 k + 3 of the chunk
      uint32_t own = 0u, own_w[OWN ? OW : 1];
#pragma unroll
      for (int k = 0; k < (OWN ? OW : 1); ++k) own_w[k] = 0u;
      static_assert(!OWN || (VEC == 8 && U == 4), "own-bits form: 8 columns per lane, words of 4 edges");
      for (int j = 0; j < nb; j += U) {
        typename RowIO<T, VEC>::Raw braw[U];   // rows in flight stay in their storage format
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int jj = (j + u) < nb ? (j + u) : (nb - 1);
          const int s = __shfl(cur, jj, lpr);
          braw[u] = RowIO<T, VEC>::load_raw(a.B + (int64_t)s * ldb + cc);
        }
        if (OWN && j == 0) flush_mask();   // the previous chunk's words, while this chunk's rows arrive
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const bool live = (j + u) < nb;
          const int jj = live ? (j + u) : (nb - 1);
          float ev[FE];
#pragma unroll
          for (int f = 0; f < FE; ++f) ev[f] = __shfl(ecur[f], jj, lpr);
          if (live) {   // slots past the chunk's end only cost the (cached) load: their lanes sit out the arithmetic
            float ce[VEC], bv[VEC];
            if constexpr (OWN) edge_term_pairs<VEC, FE>(w, ev, ce);
            else edge_term<VEC, FE>(w, ev, ce);
            RowIO<T, VEC>::unpack(braw[u], bv);
            if constexpr (OWN) {
              // the pre-activations of column pairs on packed instructions, written as vector arithmetic: their
              // results go into the block below lane register by lane register, and the compiler does not pair up
              // scalar arithmetic that feeds it (same operations in the same order as the scalar form)
              float m[VEC];
#pragma unroll
              for (int q = 0; q < VEC / 2; ++q) {
                const f32x2 t = (f32x2{ar[2 * q], ar[2 * q + 1]} + f32x2{bv[2 * q], bv[2 * q + 1]}) +
                                f32x2{ce[2 * q], ce[2 * q + 1]};
                m[2 * q] = t.x;
                m[2 * q + 1] = t.y;
              }
              relu_shift_in8(m, own);
#pragma unroll
              for (int q = 0; q < VEC / 2; ++q) {
                const f32x2 s2 = f32x2{acc[2 * q], acc[2 * q + 1]} + f32x2{m[2 * q], m[2 * q + 1]};
                acc[2 * q] = s2.x;
                acc[2 * q + 1] = s2.y;
              }
            }
#pragma unroll
            for (int v = 0; v < (OWN ? 0 : VEC); ++v) {
              const float m = relu_f((ar[v] + bv[v]) + ce[v]);
              acc[v] += m;
              if (TRAIN) {
                const bool pos = m > 0.f;
                cn[v] += pos ? 1 : 0;
                if (TRAIN == 2) {
                  // The wave's vote shifted down to the group's first lane (higher bits belong to other groups;
                  // the reader tests bit `lig` only; lanes that sit out vote 0).  Lane j + u of the group keeps
                  // the words of this edge.
                  const unsigned long long vote = __ballot(pos);
                  const uint32_t word = (uint32_t)(vote >> gfirst);
                  mb[v] = (lig == j + u) ? word : mb[v];
                }
              }
            }
          } else if (OWN) {
            own <<= 8;   // a slot past the row's end: no bits, but the word keeps its alignment
          }
        }
        if constexpr (OWN) {   // a word of 4 edges is complete: count its bits per column, keep it in edge order
#pragma unroll
          for (int v = 0; v < VEC; ++v) cn[v] += __popc(own & (0x80808080u >> v));
          const uint32_t done = __builtin_bswap32(own);   // byte k = edge j + k
#pragma unroll
          for (int k = 0; k < OW; ++k) own_w[k] = (j == 4 * k) ? done : own_w[k];
        }
      }
      if constexpr (OWN) {
#pragma unroll
        for (int k = 0; k < OW; ++k) pend[k] = own_w[k];
        pend_base = base;
        pend_nb = nb;
      }
      if (TRAIN == 2 && VEC % 4 == 0) {
        // one coalesced store per chunk: lane lig writes the VEC words of edge base + lig
        if (lig < nb) {
          if (VEC == 8 && lpr <= 16) {   // 16 lanes per row: two 16-bit words share a 32-bit one (16 bytes per edge)
            uint32_t* mp = a.maskbits + (int64_t)(base + lig) * 4;
            u32x4 w4 = {(mb[0] & 0xffffu) | (mb[1] << 16), (mb[2] & 0xffffu) | (mb[3] << 16),
                        (mb[4 % VEC] & 0xffffu) | (mb[5 % VEC] << 16), (mb[6 % VEC] & 0xffffu) | (mb[7 % VEC] << 16)};
            __builtin_nontemporal_store(w4, reinterpret_cast<u32x4*>(mp));
          } else {
            uint32_t* mp = a.maskbits + (int64_t)(base + lig) * VEC;
#pragma unroll
            for (int v = 0; v < VEC; v += 4) {
              u32x4 w4 = {mb[v], mb[v + 1], mb[v + 2], mb[v + 3]};
              __builtin_nontemporal_store(w4, reinterpret_cast<u32x4*>(mp + v));
            }
          }
        }
      }
    }

    if (valid && act) {
      if (it.w < 0) {
        RowIO<T, VEC>::store(a.out + (int64_t)row * a.ldo + c0, acc);
        if (TRAIN) {
          store_count_bytes<VEC>(a.cnt + (int64_t)row * a.H + c0, cn);
          if (t < a.first_plain + a.n_big) {   // a long unsplit row: its exact counts go to the side table
            float cf[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) cf[v] = (float)cn[v];
            store_f32<VEC>(a.cnt_big + (a.n_split + (t - a.first_plain)) * a.H + c0, cf);
          }
        }
      } else {
        float* po = a.part_out + (int64_t)it.w * a.H + c0;
#pragma unroll
        for (int v = 0; v < VEC; ++v) po[v] = acc[v];
        if (TRAIN) {
          float* pc = a.part_cnt + (int64_t)it.w * a.H + c0;
#pragma unroll
          for (int v = 0; v < VEC; ++v) pc[v] = (float)cn[v];
        }
      }
    }
  }
  if (OWN) flush_mask();   // the last chunk's words
}

// ---------------------------------------------------------------------------------------
// Backward over the source-sorted transpose.
// ---------------------------------------------------------------------------------------
template <typename T>
struct BwdArgs {
  const T* A; int64_t ldA;
  const T* B; int64_t ldB;
  const T* G; int64_t ldG;
  const T* ea; int fe;
  const T* We;
  const int32_t* col;
  const int4* items; int64_t n_items;
  T* gB; int64_t ldgB;
  float* part_gB;
  float* gwe_partial;  // [gridDim.x][H][fe]
  int H; int lpr_log2;
};

template <typename T, int VEC, int FE, bool RTFE, int U>
__global__ void __launch_bounds__(kBlock) k_agg_bwd_src(BwdArgs<T> a) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  const int lpr = 1 << a.lpr_log2;
  const int lig = lane & (lpr - 1);
  const int grp = lane >> a.lpr_log2;
  const int gpw = 64 >> a.lpr_log2;
  const int tile = lpr * VEC;
  const int c0 = blockIdx.y * tile + lig * VEC;
  const bool act = c0 < a.H;
  const int cc = act ? c0 : 0;

  float w[VEC][FE], gw[VEC][FE];
  load_edge_weights<T, VEC, FE>(a.We, a.fe, c0, act, w);
#pragma unroll
  for (int v = 0; v < VEC; ++v)
#pragma unroll
    for (int f = 0; f < FE; ++f) gw[v][f] = 0.f;

  const int64_t wave = (int64_t)blockIdx.x * (kBlock / 64) + wid;
  const int64_t stride = (int64_t)gridDim.x * (kBlock / 64) * gpw;
  for (int64_t tb = wave * gpw; tb < a.n_items; tb += stride) {
    const int64_t t = tb + grp;
    const bool valid = t < a.n_items;
    int4 it = valid ? a.items[t] : make_int4(0, 0, 0, -1);
    const int row = it.x, e0 = it.y, e1 = it.z;

    float br[VEC], acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) { br[v] = 0.f; acc[v] = 0.f; }
    if (valid) RowIO<T, VEC>::load(a.B + (int64_t)row * a.ldB + cc, br);

    int nxt = (e0 + lig < e1) ? a.col[e0 + lig] : 0;
    for (int base = e0; base < e1; base += lpr) {
      const int cur = nxt;
      const int nb = (e1 - base) < lpr ? (e1 - base) : lpr;
      const int pn = base + lpr + lig;
      nxt = pn < e1 ? a.col[pn] : 0;
      for (int j = 0; j < nb; j += U) {
        float av[U][VEC], gv[U][VEC], ev[U][FE];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int jj = (j + u) < nb ? (j + u) : (nb - 1);
          const int d = __shfl(cur, jj, lpr);
          RowIO<T, VEC>::load(a.A + (int64_t)d * (int)a.ldA + cc, av[u]);   // (32-bit strides: checked on the host)
          RowIO<T, VEC>::load(a.G + (int64_t)d * (int)a.ldG + cc, gv[u]);
          load_edge_attr<T, FE, RTFE>(a.ea, (int64_t)(base + jj), a.fe, ev[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const bool live = (j + u) < nb;
          float ce[VEC];
          edge_term<VEC, FE>(w, ev[u], ce);
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            const float pre = (av[u][v] + br[v]) + ce[v];
            const float m = (live && pre > 0.f) ? gv[u][v] : 0.f;
            acc[v] += m;
#pragma unroll
            for (int f = 0; f < FE; ++f) gw[v][f] = fmaf(m, ev[u][f], gw[v][f]);
          }
        }
      }
    }

    if (valid && act) {
      if (it.w < 0) {
        RowIO<T, VEC>::store(a.gB + (int64_t)row * a.ldgB + c0, acc);
      } else {
        float* po = a.part_gB + (int64_t)it.w * a.H + c0;
#pragma unroll
        for (int v = 0; v < VEC; ++v) po[v] = acc[v];
      }
    }
  }

  block_reduce_lane_partials<VEC, FE>(gw, lds, a.gwe_partial, a.H, a.fe, a.lpr_log2);
}

// Backward over the source-sorted transpose using the relu mask bits the forward stored: one row
// gather per edge (G[dst]) instead of two, no recomputation of the pre-activation.
//   tpos[p] = position of transposed edge p in the destination-sorted CSR (where its mask words live)
template <typename T>
struct BwdMaskArgs {
  const T* G; int64_t ldG;
  const uint32_t* maskbits;
  const int32_t* tpos;
  const T* ea; int fe;
  const int32_t* col;
  const int4* items; int64_t n_items;
  T* gB; int64_t ldgB;
  float* part_gB;
  float* gwe_partial;
  int H; int lpr_log2;
};

// FMT: the layout of an edge's mask the forward wrote: 0 = VEC words, 1 = VEC 16-bit words, 2 = one byte per lane
// (rows of 8 or 16 lanes x 8 columns: byte `lig`, bit 7 - v = column lig * 8 + v).
template <typename T, int VEC, int FE, bool RTFE, int U, int FMT>
__global__ void __launch_bounds__(kBlock) k_agg_bwd_src_masked(BwdMaskArgs<T> a) {
  constexpr bool PACK = FMT == 1;
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  const int lpr = 1 << a.lpr_log2;
  const int lig = lane & (lpr - 1);
  const int grp = lane >> a.lpr_log2;
  const int gpw = 64 >> a.lpr_log2;
  const int c0 = lig * VEC;
  const bool act = c0 < a.H;
  const int cc = act ? c0 : 0;

  float gw[VEC][FE];
#pragma unroll
  for (int v = 0; v < VEC; ++v)
#pragma unroll
    for (int f = 0; f < FE; ++f) gw[v][f] = 0.f;

  const int64_t wave = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const int64_t stride = (int64_t)gridDim.x * (kBlock / 64) * gpw;
  for (int64_t tb = wave * gpw; tb < a.n_items; tb += stride) {
    const int64_t t = tb + grp;
    const bool valid = t < a.n_items;
    int4 it = valid ? a.items[t] : make_int4(0, 0, 0, -1);
    const int row = it.x, e0 = it.y, e1 = it.z;
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;

    int nxt = 0, nxtp = 0;
    if (e0 + lig < e1) { nxt = NT_LOAD(a.col + e0 + lig); nxtp = NT_LOAD(a.tpos + e0 + lig); }
    for (int base = e0; base < e1; base += lpr) {
      const int cur = nxt, curp = nxtp;
      const int nb = (e1 - base) < lpr ? (e1 - base) : lpr;
      const int pn = base + lpr + lig;
      nxt = 0; nxtp = 0;
      if (pn < e1) { nxt = NT_LOAD(a.col + pn); nxtp = NT_LOAD(a.tpos + pn); }
      for (int j = 0; j < nb; j += U) {
        float gv[U][VEC], ev[U][FE];
        uint32_t mw[U][VEC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int jj = (j + u) < nb ? (j + u) : (nb - 1);
          const int d = __shfl(cur, jj, lpr);
          const int pp = __shfl(curp, jj, lpr);
          RowIO<T, VEC>::load(a.G + (int64_t)d * (int)a.ldG + cc, gv[u]);   // (32-bit stride: checked on the host)
          if constexpr (FMT == 2) {
            // the word that holds this lane's byte (4 lanes per word: the group reads its 8 or 16 bytes as one piece)
            mw[u][0] = NT_LOAD(a.maskbits + (int64_t)pp * (lpr >> 2) + (lig >> 2));
          } else if constexpr (PACK) {   // 16-bit words, two per 32-bit word (see the forward kernel); compile-time: a
                                  // load under a run-time branch would make the compiler wait for every load in flight
            const u32x4 w = NT_LOAD(reinterpret_cast<const u32x4*>(a.maskbits + (int64_t)pp * 4));
            mw[u][0] = w.x; mw[u][1 % VEC] = w.x >> 16; mw[u][2 % VEC] = w.y; mw[u][3 % VEC] = w.y >> 16;
            mw[u][4 % VEC] = w.z; mw[u][5 % VEC] = w.z >> 16; mw[u][6 % VEC] = w.w; mw[u][7 % VEC] = w.w >> 16;
          } else {
            const uint32_t* mp = a.maskbits + (int64_t)pp * VEC;
#pragma unroll
            for (int v = 0; v < VEC; v += 4) {
              const u32x4 w = NT_LOAD(reinterpret_cast<const u32x4*>(mp + v));
              mw[u][v] = w.x;
              if (v + 1 < VEC) mw[u][v + 1] = w.y;
              if (v + 2 < VEC) mw[u][v + 2] = w.z;
              if (v + 3 < VEC) mw[u][v + 3] = w.w;
            }
          }
          load_edge_attr<T, FE, RTFE>(a.ea, (int64_t)(base + jj), a.fe, ev[u]);
        }
        // every load of the U edges goes out before the first of them is waited for (left to itself the scheduler
        // takes this form's edges one after the other: load, wait, compute)
        if constexpr (FMT == 2) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const bool live = (j + u) < nb;
          // (FMT == 2: the bits of this lane's columns, byte lig & 3 of the word; a slot past the row's end extracts
          // a field of width 0.  Not `live ? field : 0`: the compiler turns a select with a load behind one operand
          // into a branch around the load, and a wait for every gather in flight with it.)
          const uint32_t own = __builtin_amdgcn_ubfe(mw[u][0], 8 * (lig & 3), live ? 8u : 0u);
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            float m;
            if constexpr (FMT == 2) {
              // the column's bit spread over a word (0 or all ones) and'ed onto the gradient: two instructions, and
              // no compare whose result a select would have to wait for
              const int keep = __builtin_amdgcn_sbfe((int)own, 7 - v, 1);
              m = __uint_as_float(__float_as_uint(gv[u][v]) & (uint32_t)keep);
            } else {
              m = (live && ((mw[u][v] >> lig) & 1u)) ? gv[u][v] : 0.f;
            }
            acc[v] += m;
#pragma unroll
            for (int f = 0; f < FE; ++f) gw[v][f] = fmaf(m, ev[u][f], gw[v][f]);
          }
        }
      }
    }
    if (valid && act) {
      if (it.w < 0) {
        RowIO<T, VEC>::store(a.gB + (int64_t)row * a.ldgB + c0, acc);
      } else {
        float* po = a.part_gB + (int64_t)it.w * a.H + c0;
#pragma unroll
        for (int v = 0; v < VEC; ++v) po[v] = acc[v];
      }
    }
  }
  block_reduce_lane_partials<VEC, FE>(gw, lds, a.gwe_partial, a.H, a.fe, a.lpr_log2);
}

// gA[row, :] = G[row, :] * cnt_big[slot, :] for the rows whose counts can exceed a byte: slot s < n_split is the
// row of split record s, slot n_split + j the row of item first_plain + j (the longest unsplit rows).
template <typename T>
__global__ void __launch_bounds__(kBlock) k_count_fixup(const T* __restrict__ G, int64_t ldG,
                                                        const float* __restrict__ cnt_big,
                                                        const int4* __restrict__ items, const int4* __restrict__ split,
                                                        int64_t n_split, int64_t first_plain, int64_t n_big,
                                                        T* __restrict__ gA, int64_t ldgA, int H) {
  const int64_t slots = n_split + n_big;
  for (int64_t s = blockIdx.x; s < slots; s += gridDim.x) {
    const int64_t row = s < n_split ? split[s].x : items[first_plain + (s - n_split)].x;
    for (int c = threadIdx.x; c < H; c += kBlock)
      gA[row * ldgA + c] = f32_to_scalar<T>(scalar_to_f32<T>(G[row * ldG + c]) * cnt_big[s * H + c]);
  }
}

// ---------------------------------------------------------------------------------------
// Segment mean (global_mean_pool, models/mpnn.py:54,150): one block per segment.
// ---------------------------------------------------------------------------------------
template <typename T, int VEC>
__global__ void __launch_bounds__(kBlock) k_segment_mean_fwd(const T* __restrict__ x, int64_t ldx,
                                                             const int32_t* __restrict__ rowptr,
                                                             const int32_t* __restrict__ col, T* __restrict__ out,
                                                             int64_t ldo, int64_t nseg, int64_t navg, int H,
                                                             int lpr_log2) {
  extern __shared__ float lds[];
  const int lpr = 1 << lpr_log2;
  const int lig = threadIdx.x & (lpr - 1);
  const int grp = threadIdx.x >> lpr_log2;
  const int ngrp = kBlock >> lpr_log2;
  const int tile = lpr * VEC;
  const int c0 = blockIdx.y * tile + lig * VEC;
  const bool act = c0 < H;
  const int cc = act ? c0 : 0;
  for (int64_t s = blockIdx.x; s < nseg; s += gridDim.x) {
    if (s >= navg) {   // a padding segment: its row is zeros, its (possibly very many) rows are not read
      if (grp == 0 && act) {
        float z[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) z[v] = 0.f;
        RowIO<T, VEC>::store(out + s * ldo + c0, z);
      }
      continue;
    }
    const int r0 = rowptr[s], r1 = rowptr[s + 1];
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
    // four rows in flight per lane group (a point cloud's segment has a thousand rows and one block); the rows are
    // added in the same order as one by one
    int p = r0 + grp;
    for (; p + 3 * ngrp < r1; p += 4 * ngrp) {
      float xv[4][VEC];
#pragma unroll
      for (int u = 0; u < 4; ++u) RowIO<T, VEC>::load(x + (int64_t)col[p + u * ngrp] * ldx + cc, xv[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] += xv[u][v];
    }
    for (; p < r1; p += ngrp) {
      float xv[VEC];
      RowIO<T, VEC>::load(x + (int64_t)col[p] * ldx + cc, xv);
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[v] += xv[v];
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) lds[grp * tile + lig * VEC + v] = acc[v];
    __syncthreads();
    if (grp == 0 && act) {
      float tot[VEC];
#pragma unroll
      for (int v = 0; v < VEC; ++v) tot[v] = 0.f;
      for (int g = 0; g < ngrp; ++g)
#pragma unroll
        for (int v = 0; v < VEC; ++v) tot[v] += lds[g * tile + lig * VEC + v];
      const float cnt = (float)((r1 - r0) > 1 ? (r1 - r0) : 1);
#pragma unroll
      for (int v = 0; v < VEC; ++v) tot[v] = tot[v] / cnt;
      RowIO<T, VEC>::store(out + s * ldo + c0, tot);
    }
    __syncthreads();
  }
}

template <typename T, int VEC>
__global__ void __launch_bounds__(kBlock) k_segment_mean_bwd(const T* __restrict__ gout, int64_t ldg,
                                                             const int32_t* __restrict__ rowptr,
                                                             const int32_t* __restrict__ col, T* __restrict__ gx,
                                                             int64_t ldgx, int64_t nseg, int64_t navg, int H,
                                                             int lpr_log2) {
  const int lpr = 1 << lpr_log2;
  const int lig = threadIdx.x & (lpr - 1);
  const int grp = threadIdx.x >> lpr_log2;
  const int ngrp = kBlock >> lpr_log2;
  const int c0 = blockIdx.y * (lpr * VEC) + lig * VEC;
  if (c0 >= H) return;
  if (navg < nseg) {
    // the rows of the padding segments take no gradient; there may be many of them in one segment, so all blocks
    // share the zero fill
    float z[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) z[v] = 0.f;
    const int z1 = rowptr[nseg];
    for (int64_t p = rowptr[navg] + (int64_t)blockIdx.x * ngrp + grp; p < z1; p += (int64_t)gridDim.x * ngrp)
      RowIO<T, VEC>::store(gx + (int64_t)col[p] * ldgx + c0, z);
  }
  for (int64_t s = blockIdx.x; s < navg; s += gridDim.x) {
    const int r0 = rowptr[s], r1 = rowptr[s + 1];
    if (r1 <= r0) continue;
    float g[VEC];
    RowIO<T, VEC>::load(gout + s * ldg + c0, g);
    const float cnt = (float)(r1 - r0);
#pragma unroll
    for (int v = 0; v < VEC; ++v) g[v] = g[v] / cnt;
    int p = r0 + grp;
    for (; p + 3 * ngrp < r1; p += 4 * ngrp) {   // the four row ids requested together
      int rows[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) rows[u] = col[p + u * ngrp];
#pragma unroll
      for (int u = 0; u < 4; ++u) RowIO<T, VEC>::store(gx + (int64_t)rows[u] * ldgx + c0, g);
    }
    for (; p < r1; p += ngrp) RowIO<T, VEC>::store(gx + (int64_t)col[p] * ldgx + c0, g);
  }
}

// ---------------------------------------------------------------------------------------
// Host-side dispatch
// ---------------------------------------------------------------------------------------
// Row gathers kept in flight per lane by the forward kernel: 4.  With 8 columns per lane (bfloat16 rows) that
// spills some 40-60 registers at 4 waves per SIMD, and is still faster than 2 gathers without spills or 4 gathers
// at 3 waves (5.9 vs 6.8 ms inference, 8.0 vs 8.4 ms training on the 100 M-edge graph): the pass needs both the
// occupancy and the loads in flight.  MPNN_FWD_U=2|4 overrides (tuning).
static int fwd_unroll(int vec) {
  static int forced = [] {
    const char* e = getenv("MPNN_FWD_U");
    const int v = e ? atoi(e) : 0;
    return (v == 2 || v == 4) ? v : 0;
  }();
  if (forced) return forced;
  return 4;
}

template <typename T, int VEC, int TRAIN, int U>
static int32_t launch_fwd_u(const FwdArgs<T>& a, dim3 grid, hipStream_t stream) {
  if (a.fe == 4) k_agg_fwd<T, VEC, 4, false, TRAIN, U><<<grid, kBlock, 0, stream>>>(a);
  else if (a.fe == 1) k_agg_fwd<T, VEC, 1, false, TRAIN, U><<<grid, kBlock, 0, stream>>>(a);
  else k_agg_fwd<T, VEC, kMaxFe, true, TRAIN, U><<<grid, kBlock, 0, stream>>>(a);
  MPNN_LAUNCH_CHECK("k_agg_fwd");
  return MPNN_OK;
}

template <typename T, int VEC, int TRAIN>
static int32_t launch_fwd(const FwdArgs<T>& a, dim3 grid, hipStream_t stream) {
  if (fwd_unroll(VEC) == 2) return launch_fwd_u<T, VEC, TRAIN, 2>(a, grid, stream);
  return launch_fwd_u<T, VEC, TRAIN, 4>(a, grid, stream);
}

template <typename T, int VEC>
static int32_t run_fwd(const void* A, int64_t ldA, const void* B, int64_t ldB, const void* ea, int32_t Fe,
                       const void* We, const int32_t* col, const int32_t* items, int64_t n_items,
                       const int32_t* split, int64_t n_split, int64_t n_parts, void* out, int64_t ldo, uint8_t* cnt,
                       float* cnt_big, int64_t n_big, float* part_out, float* part_cnt, uint32_t* maskbits, int32_t H,
                       const Geometry& g, hipStream_t stream) {
  FwdArgs<T> a;
  a.A = static_cast<const T*>(A); a.ldA = ldA;
  a.B = static_cast<const T*>(B); a.ldB = ldB;
  a.ea = static_cast<const T*>(ea); a.fe = Fe;
  a.We = static_cast<const T*>(We);
  a.col = col;
  a.items = reinterpret_cast<const int4*>(items); a.n_items = n_items;
  a.out = static_cast<T*>(out); a.ldo = ldo;
  a.cnt = cnt; a.cnt_big = cnt_big; a.first_plain = n_parts; a.n_big = n_big; a.n_split = n_split;
  a.part_out = part_out; a.part_cnt = part_cnt;
  a.maskbits = maskbits;
  a.H = H; a.lpr_log2 = g.lpr_log2;
  // One item per lane group while the grid allows it: the items arrive largest first, so the hardware's
  // in-order block dispatch balances the CUs by itself; the forward kernel keeps no per-block state that
  // would bound the grid.
  static const int64_t max_blocks = [] { const char* e = getenv("MPNN_FWD_BLOCKS"); return e ? atoll(e) : (int64_t)1 << 16; }();
  dim3 grid(row_grid(n_items, g.lpr_log2, max_blocks), g.tiles);
  int32_t rc;
  if (!cnt) rc = launch_fwd<T, VEC, 0>(a, grid, stream);
  else if (!maskbits) rc = launch_fwd<T, VEC, 1>(a, grid, stream);
  else if constexpr (VEC == 8) {
    // rows of 8 or 16 lanes: every lane keeps the bits of its own columns (4 gathers in flight: a mask word is 4 edges)
    rc = g.lpr_log2 == 3 ? launch_fwd_u<T, VEC, 3, 4>(a, grid, stream)
       : g.lpr_log2 == 4 ? launch_fwd_u<T, VEC, 4, 4>(a, grid, stream) : launch_fwd<T, VEC, 2>(a, grid, stream);
  } else if constexpr (VEC % 4 == 0) rc = launch_fwd<T, VEC, 2>(a, grid, stream);
  else return fail(MPNN_E_INVALID, "mpnn_aggregate_fwd: mask bits need the 16-byte vector path");
  if (rc != MPNN_OK) return rc;
  if (n_split > 0) {
    rc = launch_reduce_split<T, VEC>(part_out, split, n_split, static_cast<T*>(out), ldo, H, g, stream);
    if (rc != MPNN_OK) return rc;
    // the exact counts of split rows go to the side table, one row per split record
    if (cnt) rc = launch_reduce_split<float, VEC>(part_cnt, split, n_split, cnt_big, H, H, g, stream, /*by_slot=*/true);
  }
  return rc;
}

template <typename T, int VEC>
static int32_t run_bwd(const void* A, int64_t ldA, const void* B, int64_t ldB, const void* G, int64_t ldG,
                       const void* ea, int32_t Fe, const void* We, const int32_t* col, const int32_t* items,
                       int64_t n_items, const int32_t* split, int64_t n_split, void* gB, int64_t ldgB, void* gWe,
                       float* part_gB, float* scratch, int32_t H, const Geometry& g, hipStream_t stream) {
  constexpr int U = 2;
  BwdArgs<T> a;
  a.A = static_cast<const T*>(A); a.ldA = ldA;
  a.B = static_cast<const T*>(B); a.ldB = ldB;
  a.G = static_cast<const T*>(G); a.ldG = ldG;
  a.ea = static_cast<const T*>(ea); a.fe = Fe;
  a.We = static_cast<const T*>(We);
  a.col = col;
  a.items = reinterpret_cast<const int4*>(items); a.n_items = n_items;
  a.gB = static_cast<T*>(gB); a.ldgB = ldgB;
  a.part_gB = part_gB; a.gwe_partial = scratch;
  a.H = H; a.lpr_log2 = g.lpr_log2;
  dim3 grid(row_grid(n_items, g.lpr_log2), g.tiles);
  const int tile = (1 << g.lpr_log2) * VEC;
  const int fe_t = Fe == 4 ? 4 : (Fe == 1 ? 1 : kMaxFe);
  const size_t lds = (size_t)(kBlock / 64) * tile * fe_t * sizeof(float);
  if (Fe == 4) k_agg_bwd_src<T, VEC, 4, false, U><<<grid, kBlock, lds, stream>>>(a);
  else if (Fe == 1) k_agg_bwd_src<T, VEC, 1, false, U><<<grid, kBlock, lds, stream>>>(a);
  else k_agg_bwd_src<T, VEC, kMaxFe, true, U><<<grid, kBlock, lds, stream>>>(a);
  MPNN_LAUNCH_CHECK("k_agg_bwd_src");
  if (n_split > 0) {
    int32_t rc = launch_reduce_split<T, VEC>(part_gB, split, n_split, static_cast<T*>(gB), ldgB, H, g, stream);
    if (rc != MPNN_OK) return rc;
  }
  k_reduce_gwe<T><<<H * Fe, kBlock, 0, stream>>>(scratch, (int)grid.x, H * Fe, static_cast<T*>(gWe));
  MPNN_LAUNCH_CHECK("k_reduce_gwe");
  return MPNN_OK;
}

template <typename T, int VEC>
static int32_t run_bwd_masked(const void* G, int64_t ldG, const uint32_t* maskbits, const int32_t* tpos,
                              const void* ea, int32_t Fe, const int32_t* col, const int32_t* items, int64_t n_items,
                              const int32_t* split, int64_t n_split, void* gB, int64_t ldgB, void* gWe,
                              float* part_gB, float* scratch, int32_t H, const Geometry& g, hipStream_t stream) {
  constexpr int U = 4;
  BwdMaskArgs<T> a;
  a.G = static_cast<const T*>(G); a.ldG = ldG;
  a.maskbits = maskbits; a.tpos = tpos;
  a.ea = static_cast<const T*>(ea); a.fe = Fe;
  a.col = col;
  a.items = reinterpret_cast<const int4*>(items); a.n_items = n_items;
  a.gB = static_cast<T*>(gB); a.ldgB = ldgB;
  a.part_gB = part_gB; a.gwe_partial = scratch;
  a.H = H; a.lpr_log2 = g.lpr_log2;
  dim3 grid(row_grid(n_items, g.lpr_log2), 1);
  const int tile = (1 << g.lpr_log2) * VEC;
  const int fe_t = Fe == 4 ? 4 : (Fe == 1 ? 1 : kMaxFe);
  const size_t lds = (size_t)(kBlock / 64) * tile * fe_t * sizeof(float);
  auto go = [&](auto fmt) {
    constexpr int FMT = decltype(fmt)::value;
    if (Fe == 4) k_agg_bwd_src_masked<T, VEC, 4, false, U, FMT><<<grid, kBlock, lds, stream>>>(a);
    else if (Fe == 1) k_agg_bwd_src_masked<T, VEC, 1, false, U, FMT><<<grid, kBlock, lds, stream>>>(a);
    else k_agg_bwd_src_masked<T, VEC, kMaxFe, true, U, FMT><<<grid, kBlock, lds, stream>>>(a);
  };
  if (VEC == 8 && (g.lpr_log2 == 3 || g.lpr_log2 == 4)) go(std::integral_constant<int, 2>{});   // the layouts run_fwd chooses
  else if (VEC == 8 && g.lpr_log2 <= 4) go(std::integral_constant<int, 1>{});
  else go(std::integral_constant<int, 0>{});
  MPNN_LAUNCH_CHECK("k_agg_bwd_src_masked");
  if (n_split > 0) {
    int32_t rc = launch_reduce_split<T, VEC>(part_gB, split, n_split, static_cast<T*>(gB), ldgB, H, g, stream);
    if (rc != MPNN_OK) return rc;
  }
  k_reduce_gwe<T><<<H * Fe, kBlock, 0, stream>>>(scratch, (int)grid.x, H * Fe, static_cast<T*>(gWe));
  MPNN_LAUNCH_CHECK("k_reduce_gwe");
  return MPNN_OK;
}

template <typename T, int VEC>
static int32_t run_segmean(bool fwd, const void* in, int64_t ldin, const int32_t* rowptr, const int32_t* col,
                           void* out, int64_t ldout, int64_t nseg, int64_t navg, int32_t H, const Geometry& g,
                           hipStream_t stream) {
  int64_t blocks = nseg > kMaxBlocks ? kMaxBlocks : nseg;
  dim3 grid((unsigned)blocks, g.tiles);
  if (fwd) {
    const size_t lds = (size_t)(kBlock >> g.lpr_log2) * (1 << g.lpr_log2) * VEC * sizeof(float);
    k_segment_mean_fwd<T, VEC><<<grid, kBlock, lds, stream>>>(static_cast<const T*>(in), ldin, rowptr, col,
                                                              static_cast<T*>(out), ldout, nseg, navg, H, g.lpr_log2);
  } else {
    k_segment_mean_bwd<T, VEC><<<grid, kBlock, 0, stream>>>(static_cast<const T*>(in), ldin, rowptr, col,
                                                            static_cast<T*>(out), ldout, nseg, navg, H, g.lpr_log2);
  }
  MPNN_LAUNCH_CHECK("k_segment_mean");
  return MPNN_OK;
}

static int32_t segmean(bool fwd, const void* in, int64_t ldin, const int32_t* rowptr, const int32_t* col, void* out,
                       int64_t ldout, int64_t nseg, int64_t navg, int32_t H, int32_t dtype, hipStream_t stream) {
  if (dtype != MPNN_F32 && dtype != MPNN_BF16) return fail(MPNN_E_INVALID, "mpnn_segment_mean: dtype %d", dtype);
  if (H < 1 || nseg < 0) return fail(MPNN_E_INVALID, "mpnn_segment_mean: H=%d nseg=%lld", H, (long long)nseg);
  if (navg < 0 || navg > nseg)
    return fail(MPNN_E_INVALID, "mpnn_segment_mean: num_averaged=%lld outside [0, %lld]", (long long)navg, (long long)nseg);
  if (nseg == 0) return MPNN_OK;
  if (!in || !rowptr || !col || !out) return fail(MPNN_E_INVALID, "mpnn_segment_mean: null pointer argument");
  if (ldin < H || ldout < H) return fail(MPNN_E_INVALID, "mpnn_segment_mean: leading dimension below H");
  const int full = dtype == MPNN_BF16 ? 8 : 4;
  const bool vec_ok = aligned16(in) && aligned16(out) && ldin % full == 0 && ldout % full == 0;
  const Geometry g = geometry(H, dtype, vec_ok);
  if (dtype == MPNN_F32) {
    if (g.vec == 4) return run_segmean<float, 4>(fwd, in, ldin, rowptr, col, out, ldout, nseg, navg, H, g, stream);
    return run_segmean<float, 1>(fwd, in, ldin, rowptr, col, out, ldout, nseg, navg, H, g, stream);
  }
  if (g.vec == 8) return run_segmean<bf16_t, 8>(fwd, in, ldin, rowptr, col, out, ldout, nseg, navg, H, g, stream);
  return run_segmean<bf16_t, 1>(fwd, in, ldin, rowptr, col, out, ldout, nseg, navg, H, g, stream);
}

}  // namespace mpnn

using namespace mpnn;

extern "C" {

int32_t mpnn_aggregate_fwd(const void* A, int64_t ldA, const void* B, int64_t ldB, const void* ea_sorted, int32_t Fe,
                           const void* We, const int32_t* col, const int32_t* items, int64_t n_items,
                           const int32_t* split, int64_t n_split, int64_t n_parts, void* out, int64_t ldo, uint8_t* cnt,
                           float* cnt_big, int64_t n_big, float* part_out, float* part_cnt, uint32_t* maskbits,
                           int64_t num_nodes, int32_t H, int32_t dtype, mpnn_stream_t stream) {
  if (dtype != MPNN_F32 && dtype != MPNN_BF16) return fail(MPNN_E_INVALID, "mpnn_aggregate_fwd: dtype %d", dtype);
  if (n_big < 0 || (cnt && n_split + n_big > 0 && !cnt_big))
    return fail(MPNN_E_INVALID, "mpnn_aggregate_fwd: %lld long rows but no cnt_big table", (long long)(n_split + n_big));
  if (H < 1 || Fe < 1 || Fe > kMaxFe)
    return fail(MPNN_E_INVALID, "mpnn_aggregate_fwd: H=%d (>=1), Fe=%d (1..%d)", H, Fe, kMaxFe);
  if (num_nodes < 0 || n_items < 0 || n_split < 0 || n_parts < 0)
    return fail(MPNN_E_INVALID, "mpnn_aggregate_fwd: negative size");
  if (num_nodes == 0 || n_items == 0) return MPNN_OK;
  if (!A || !B || !We || !items || !out) return fail(MPNN_E_INVALID, "mpnn_aggregate_fwd: null pointer argument");
  if (ldA < H || ldB < H || ldo < H) return fail(MPNN_E_INVALID, "mpnn_aggregate_fwd: leading dimension below H");
  if (ldA > INT32_MAX || ldB > INT32_MAX) return fail(MPNN_E_INVALID, "mpnn_aggregate_fwd: leading dimension above 2^31 - 1");
  if (n_parts > 0 && (!part_out || (cnt && !part_cnt) || !split))
    return fail(MPNN_E_INVALID, "mpnn_aggregate_fwd: split rows present but scratch slabs are null");
  if (((uintptr_t)items & 15) || ((uintptr_t)split & 15))
    return fail(MPNN_E_INVALID, "mpnn_aggregate_fwd: items/split must be 16-byte aligned");
  const int full = dtype == MPNN_BF16 ? 8 : 4;
  const bool vec_ok = aligned16(A) && aligned16(B) && aligned16(out) && aligned16(cnt) && aligned16(cnt_big) &&
                      aligned16(ea_sorted) && ldA % full == 0 && ldB % full == 0 && ldo % full == 0;
  const Geometry g = geometry(H, dtype, vec_ok);
  if (maskbits && (!cnt || g.vec == 1 || g.tiles != 1 || g.lpr_log2 > 5 || ((uintptr_t)maskbits & 15)))
    return fail(MPNN_E_INVALID, "mpnn_aggregate_fwd: maskbits needs cnt, the vector path and mpnn_mask_words(H, dtype) > 0");
  if (dtype == MPNN_F32) {
    if (g.vec == 4)
      return run_fwd<float, 4>(A, ldA, B, ldB, ea_sorted, Fe, We, col, items, n_items, split, n_split, n_parts, out, ldo,
                               cnt, cnt_big, n_big, part_out, part_cnt, maskbits, H, g, stream);
    return run_fwd<float, 1>(A, ldA, B, ldB, ea_sorted, Fe, We, col, items, n_items, split, n_split, n_parts, out, ldo, cnt,
                             cnt_big, n_big, part_out, part_cnt, maskbits, H, g, stream);
  }
  if (g.vec == 8)
    return run_fwd<bf16_t, 8>(A, ldA, B, ldB, ea_sorted, Fe, We, col, items, n_items, split, n_split, n_parts, out, ldo,
                              cnt, cnt_big, n_big, part_out, part_cnt, maskbits, H, g, stream);
  return run_fwd<bf16_t, 1>(A, ldA, B, ldB, ea_sorted, Fe, We, col, items, n_items, split, n_split, n_parts, out, ldo, cnt,
                            cnt_big, n_big, part_out, part_cnt, maskbits, H, g, stream);
}

}  // extern "C"

// ---------------------------------------------------------------------------------------
// Forward entry point that allocates nothing: the partial-sum slabs of split rows live in a
// file-scope device array instead of caller-provided scratch.
// ---------------------------------------------------------------------------------------
namespace mpnn {

constexpr int64_t kNoAllocFloats = MPNN_NOALLOC_SCRATCH_BYTES / (int64_t)sizeof(float);
static __device__ __align__(256) float scratch[kNoAllocFloats];

// Every GPU holds its own copy of `scratch` at its own address; the name above is only a host-side handle.
// The device address is looked up once per device (with that device current) and kept here.
constexpr int kMaxDevices = 64;
static float* g_scratch_ptr[kMaxDevices];

static int32_t noalloc_scratch(float** out) {
  int dev = -1;
  MPNN_HIP_CHECK(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices) return fail(MPNN_E_INVALID, "mpnn_aggregate_fwd_noalloc: device %d out of range", dev);
  float* p = __atomic_load_n(&g_scratch_ptr[dev], __ATOMIC_ACQUIRE);
  if (!p) {
    // two threads racing here both obtain the same address for this device
    MPNN_HIP_CHECK(hipGetSymbolAddress(reinterpret_cast<void**>(&p), HIP_SYMBOL(scratch)));
    if (!p) return fail(MPNN_E_HIP, "mpnn_aggregate_fwd_noalloc: hipGetSymbolAddress returned a null address");
    __atomic_store_n(&g_scratch_ptr[dev], p, __ATOMIC_RELEASE);
  }
  *out = p;
  return MPNN_OK;
}

}  // namespace mpnn

extern "C" {

int32_t mpnn_aggregate_fwd_noalloc(const void* A, int64_t ldA, const void* B, int64_t ldB, const void* ea_sorted,
                                   int32_t Fe, const void* We, const int32_t* col, const int32_t* items,
                                   int64_t n_items, const int32_t* split, int64_t n_split, int64_t n_parts, void* out,
                                   int64_t ldo, uint8_t* cnt, float* cnt_big, int64_t n_big, uint32_t* maskbits,
                                   int64_t num_nodes, int32_t H, int32_t dtype, mpnn_stream_t stream) {
  if (H < 1 || n_parts < 0) return fail(MPNN_E_INVALID, "mpnn_aggregate_fwd_noalloc: H=%d n_parts=%lld", H, (long long)n_parts);
  float* part_out = nullptr;
  float* part_cnt = nullptr;
  if (n_parts > 0) {
    // slab layout inside the array: [n_parts][H] partial sums, then (training) [n_parts][H] partial counts,
    // each slab starting on a 256-byte boundary
    // (n_parts * H cannot overflow below: pieces beyond the array's capacity are rejected first)
    const int64_t slab = n_parts <= kNoAllocFloats / H ? align_up(n_parts * (int64_t)H, 64) : kNoAllocFloats + 1;
    const int64_t need = slab * (cnt ? 2 : 1);
    if (need > kNoAllocFloats)
      return fail(MPNN_E_CAPACITY,
                  "mpnn_aggregate_fwd_noalloc: %lld split pieces of %d columns need %lld bytes of scratch, the built-in "
                  "array holds %lld; use mpnn_aggregate_fwd with caller-provided slabs",
                  (long long)n_parts, H, (long long)(n_parts <= kNoAllocFloats / H ? need * 4 : -1),
                  (long long)MPNN_NOALLOC_SCRATCH_BYTES);
    float* base = nullptr;
    const int32_t rc = noalloc_scratch(&base);
    if (rc != MPNN_OK) return rc;
    // Nothing has to start at zero: every piece writes its whole slab row before k_reduce_split reads it.
    part_out = base;
    part_cnt = cnt ? base + slab : nullptr;
  }
  return mpnn_aggregate_fwd(A, ldA, B, ldB, ea_sorted, Fe, We, col, items, n_items, split, n_split, n_parts, out, ldo,
                            cnt, cnt_big, n_big, part_out, part_cnt, maskbits, num_nodes, H, dtype, stream);
}

int32_t mpnn_count_fixup(const void* G, int64_t ldG, const float* cnt_big, const int32_t* items, const int32_t* split,
                         int64_t n_split, int64_t n_parts, int64_t n_big, void* gA, int64_t ldgA, int32_t H,
                         int32_t dtype, mpnn_stream_t stream) {
  if (dtype != MPNN_F32 && dtype != MPNN_BF16) return fail(MPNN_E_INVALID, "mpnn_count_fixup: dtype %d", dtype);
  if (H < 1 || n_split < 0 || n_parts < 0 || n_big < 0 || ldG < H || ldgA < H)
    return fail(MPNN_E_INVALID, "mpnn_count_fixup: H=%d n_split=%lld n_big=%lld ldG=%lld ldgA=%lld", H,
                (long long)n_split, (long long)n_big, (long long)ldG, (long long)ldgA);
  const int64_t slots = n_split + n_big;
  if (slots == 0) return MPNN_OK;
  if (!G || !cnt_big || !gA || (n_split > 0 && !split) || (n_big > 0 && !items))
    return fail(MPNN_E_INVALID, "mpnn_count_fixup: null pointer argument");
  if (((uintptr_t)items & 15) || ((uintptr_t)split & 15))
    return fail(MPNN_E_INVALID, "mpnn_count_fixup: items/split must be 16-byte aligned");
  const int blocks = (int)(slots < kMaxBlocks ? slots : kMaxBlocks);
  if (dtype == MPNN_F32)
    k_count_fixup<float><<<blocks, kBlock, 0, stream>>>(static_cast<const float*>(G), ldG, cnt_big,
                                                        reinterpret_cast<const int4*>(items),
                                                        reinterpret_cast<const int4*>(split), n_split, n_parts, n_big,
                                                        static_cast<float*>(gA), ldgA, H);
  else
    k_count_fixup<bf16_t><<<blocks, kBlock, 0, stream>>>(static_cast<const bf16_t*>(G), ldG, cnt_big,
                                                         reinterpret_cast<const int4*>(items),
                                                         reinterpret_cast<const int4*>(split), n_split, n_parts, n_big,
                                                         static_cast<bf16_t*>(gA), ldgA, H);
  MPNN_LAUNCH_CHECK("k_count_fixup");
  return MPNN_OK;
}

int32_t mpnn_mask_words(int32_t H, int32_t dtype) {
  if (H < 1 || (dtype != MPNN_F32 && dtype != MPNN_BF16)) return 0;
  const Geometry g = geometry(H, dtype, true);
  if (g.vec == 1 || g.tiles != 1 || g.lpr_log2 > 5) return 0;
  if (g.vec == 8 && g.lpr_log2 == 3) return 2;          // rows of 8 lanes x 8 columns: one byte per lane
  return (g.vec == 8 && g.lpr_log2 <= 4) ? 4 : g.vec;   // rows of at most 16 lanes pack two 16-bit words per word
}

int64_t mpnn_bwd_scratch_floats(int32_t H, int32_t Fe) {
  if (H < 1 || Fe < 1) return -1;
  return (int64_t)kMaxBlocks * H * Fe;
}

int32_t mpnn_aggregate_bwd_src(const void* A, int64_t ldA, const void* B, int64_t ldB, const void* G, int64_t ldG,
                               const void* ea_sorted_t, int32_t Fe, const void* We, const int32_t* col_t,
                               const int32_t* items_t, int64_t n_items_t, const int32_t* split_t, int64_t n_split_t,
                               int64_t n_parts_t, void* gB, int64_t ldgB, void* gWe, float* part_gB,
                               float* gwe_scratch, int64_t num_src_nodes, int32_t H, int32_t dtype,
                               mpnn_stream_t stream) {
  if (dtype != MPNN_F32 && dtype != MPNN_BF16) return fail(MPNN_E_INVALID, "mpnn_aggregate_bwd_src: dtype %d", dtype);
  if (H < 1 || Fe < 1 || Fe > kMaxFe)
    return fail(MPNN_E_INVALID, "mpnn_aggregate_bwd_src: H=%d (>=1), Fe=%d (1..%d)", H, Fe, kMaxFe);
  if (num_src_nodes < 0 || n_items_t < 0 || n_split_t < 0 || n_parts_t < 0)
    return fail(MPNN_E_INVALID, "mpnn_aggregate_bwd_src: negative size");
  if (!gWe || !gwe_scratch) return fail(MPNN_E_INVALID, "mpnn_aggregate_bwd_src: gWe/gwe_scratch is null");
  if (num_src_nodes == 0 || n_items_t == 0) {
    MPNN_HIP_CHECK(hipMemsetAsync(gWe, 0, (size_t)H * Fe * (dtype == MPNN_BF16 ? 2 : 4), stream));
    return MPNN_OK;
  }
  if (!A || !B || !G || !We || !items_t || !gB)
    return fail(MPNN_E_INVALID, "mpnn_aggregate_bwd_src: null pointer argument");
  if (ldA < H || ldB < H || ldG < H || ldgB < H)
    return fail(MPNN_E_INVALID, "mpnn_aggregate_bwd_src: leading dimension below H");
  if (ldA > INT32_MAX || ldG > INT32_MAX)
    return fail(MPNN_E_INVALID, "mpnn_aggregate_bwd_src: leading dimension above 2^31 - 1");
  if (n_parts_t > 0 && (!part_gB || !split_t))
    return fail(MPNN_E_INVALID, "mpnn_aggregate_bwd_src: split rows present but scratch slab is null");
  if (((uintptr_t)items_t & 15) || ((uintptr_t)split_t & 15))
    return fail(MPNN_E_INVALID, "mpnn_aggregate_bwd_src: items/split must be 16-byte aligned");
  const int full = dtype == MPNN_BF16 ? 8 : 4;
  const bool vec_ok = aligned16(A) && aligned16(B) && aligned16(G) && aligned16(gB) && aligned16(ea_sorted_t) &&
                      ldA % full == 0 && ldB % full == 0 && ldG % full == 0 && ldgB % full == 0;
  const Geometry g = geometry(H, dtype, vec_ok);
  if (dtype == MPNN_F32) {
    if (g.vec == 4)
      return run_bwd<float, 4>(A, ldA, B, ldB, G, ldG, ea_sorted_t, Fe, We, col_t, items_t, n_items_t, split_t,
                               n_split_t, gB, ldgB, gWe, part_gB, gwe_scratch, H, g, stream);
    return run_bwd<float, 1>(A, ldA, B, ldB, G, ldG, ea_sorted_t, Fe, We, col_t, items_t, n_items_t, split_t,
                             n_split_t, gB, ldgB, gWe, part_gB, gwe_scratch, H, g, stream);
  }
  if (g.vec == 8)
    return run_bwd<bf16_t, 8>(A, ldA, B, ldB, G, ldG, ea_sorted_t, Fe, We, col_t, items_t, n_items_t, split_t,
                              n_split_t, gB, ldgB, gWe, part_gB, gwe_scratch, H, g, stream);
  return run_bwd<bf16_t, 1>(A, ldA, B, ldB, G, ldG, ea_sorted_t, Fe, We, col_t, items_t, n_items_t, split_t,
                            n_split_t, gB, ldgB, gWe, part_gB, gwe_scratch, H, g, stream);
}

int32_t mpnn_aggregate_bwd_src_masked(const void* G, int64_t ldG, const uint32_t* maskbits, const int32_t* tpos,
                                      const void* ea_sorted_t, int32_t Fe, const int32_t* col_t,
                                      const int32_t* items_t, int64_t n_items_t, const int32_t* split_t,
                                      int64_t n_split_t, int64_t n_parts_t, void* gB, int64_t ldgB, void* gWe,
                                      float* part_gB, float* gwe_scratch, int64_t num_src_nodes, int32_t H,
                                      int32_t dtype, mpnn_stream_t stream) {
  const char* who = "mpnn_aggregate_bwd_src_masked";
  if (dtype != MPNN_F32 && dtype != MPNN_BF16) return fail(MPNN_E_INVALID, "%s: dtype %d", who, dtype);
  if (H < 1 || Fe < 1 || Fe > kMaxFe) return fail(MPNN_E_INVALID, "%s: H=%d (>=1), Fe=%d (1..%d)", who, H, Fe, kMaxFe);
  if (num_src_nodes < 0 || n_items_t < 0 || n_split_t < 0 || n_parts_t < 0)
    return fail(MPNN_E_INVALID, "%s: negative size", who);
  if (!gWe || !gwe_scratch) return fail(MPNN_E_INVALID, "%s: gWe/gwe_scratch is null", who);
  if (mpnn_mask_words(H, dtype) == 0) return fail(MPNN_E_INVALID, "%s: H=%d has no mask-bit layout", who, H);
  if (num_src_nodes == 0 || n_items_t == 0) {
    MPNN_HIP_CHECK(hipMemsetAsync(gWe, 0, (size_t)H * Fe * (dtype == MPNN_BF16 ? 2 : 4), stream));
    return MPNN_OK;
  }
  if (!G || !maskbits || !tpos || !items_t || !gB) return fail(MPNN_E_INVALID, "%s: null pointer argument", who);
  if (ldG < H || ldgB < H) return fail(MPNN_E_INVALID, "%s: leading dimension below H", who);
  if (ldG > INT32_MAX) return fail(MPNN_E_INVALID, "%s: leading dimension above 2^31 - 1", who);
  if (n_parts_t > 0 && (!part_gB || !split_t)) return fail(MPNN_E_INVALID, "%s: split rows present but scratch slab is null", who);
  const int full = dtype == MPNN_BF16 ? 8 : 4;
  if (!aligned16(G) || !aligned16(gB) || !aligned16(ea_sorted_t) || !aligned16(maskbits) || !aligned16(items_t) ||
      !aligned16(split_t) || ldG % full != 0 || ldgB % full != 0)
    return fail(MPNN_E_INVALID, "%s: operands must be 16-byte aligned with ld %% %d == 0", who, full);
  const Geometry g = geometry(H, dtype, true);
  if (dtype == MPNN_F32)
    return run_bwd_masked<float, 4>(G, ldG, maskbits, tpos, ea_sorted_t, Fe, col_t, items_t, n_items_t, split_t,
                                    n_split_t, gB, ldgB, gWe, part_gB, gwe_scratch, H, g, stream);
  return run_bwd_masked<bf16_t, 8>(G, ldG, maskbits, tpos, ea_sorted_t, Fe, col_t, items_t, n_items_t, split_t,
                                   n_split_t, gB, ldgB, gWe, part_gB, gwe_scratch, H, g, stream);
}

int32_t mpnn_segment_mean_fwd(const void* x, int64_t ldx, const int32_t* rowptr, const int32_t* col, void* out,
                              int64_t ldo, int64_t num_segments, int64_t num_averaged, int32_t H, int32_t dtype,
                              mpnn_stream_t stream) {
  return segmean(true, x, ldx, rowptr, col, out, ldo, num_segments, num_averaged, H, dtype, stream);
}

int32_t mpnn_segment_mean_bwd(const void* gout, int64_t ldg, const int32_t* rowptr, const int32_t* col, void* gx,
                              int64_t ldgx, int64_t num_segments, int64_t num_averaged, int32_t H, int32_t dtype,
                              mpnn_stream_t stream) {
  return segmean(false, gout, ldg, rowptr, col, gx, ldgx, num_segments, num_averaged, H, dtype, stream);
}

}  // extern "C"
