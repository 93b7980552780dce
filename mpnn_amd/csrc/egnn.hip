// EGNN coordinate update fused into the aggregation pass, and its backward, for gfx950.
//
// The reference has no EGNN layer; the operator is defined by the formula
//     m_e    = relu((A[dst] + B[src]) + W_e . edge_attr[e])
//     aggr_n = sum_{e -> n} m_e
//     pos'_n = pos_n + sum_{e -> n} (pos_n - pos_src(e)) * (w_pos . m_e)
// i.e. the message pass of models/layers.py:128 plus a coordinate update whose per-edge scalar is a
// bias-free linear map of the message.  Same work layout as aggregate.hip: a group of LPR lanes owns
// one work item (CSR row or piece of a long row) and walks it in edge order; the per-edge scalar
// w_pos . m_e is a fixed-shape xor-shuffle tree over the group's lanes.  Coordinates are float.
//
// Backward (upstream gradients G = d/d aggr, gp = d/d pos'):
//     t_e   = gp[dst] . (pos[dst] - pos[src])                    d loss / d (w_pos . m_e)
//     gm_e  = (m_e > 0) * (G[dst] + w_pos * t_e)                 d loss / d pre-activation
//     gA_n  = sum_{e -> n} gm_e          S_n = sum_{e -> n} (w_pos . m_e)      (dst pass)
//     gw_pos = sum_e t_e * m_e                                                    (dst pass)
//     gB_s  = sum_{e from s} gm_e        gW_e = sum_e gm_e (x) edge_attr[e]       (src pass)
//     gpos_src_s = - sum_{e from s} (w_pos . m_e) * gp[dst]                       (src pass)
//     d loss / d pos = gp * (1 + S) + gpos_src                                    (caller, elementwise)
#include "edgeops.h"

namespace mpnn {

template <typename T>
struct EgnnArgs {
  const T* A; int64_t ldA;
  const T* B; int64_t ldB;
  const T* G; int64_t ldG;      // backward only
  const float* gp;              // backward only: [Nd, 3]
  const T* ea; int fe;
  const T* We;
  const T* wp;                  // [H]
  const float* pos_dst;         // [Nd, 3]
  const float* pos_src;         // [Ns, 3]
  const int32_t* col;
  const int4* items; int64_t n_items;
  T* out; int64_t ldo;          // fwd: aggr; bwd_dst: gA; bwd_src: gB
  float* small_out;             // fwd: pos' [Nd,3]; bwd_dst: S [Nd]; bwd_src: gpos_src [Ns,3]
  float* part_out;              // [P, H]
  float* part_small;            // [P, 4]
  float* lane_partial;          // bwd_dst: gw_pos partial [grid][H]; bwd_src: gW_e partial [grid][H][fe]
  int H; int lpr_log2;
};

template <typename T, int VEC>
__device__ __forceinline__ void load_wp(const T* wp, int c0, bool act, float (&w)[VEC]) {
#pragma unroll
  for (int v = 0; v < VEC; ++v) w[v] = act ? scalar_to_f32<T>(wp[c0 + v]) : 0.f;
}

// Sum over the lanes of a group, the same xor tree in every lane (result identical in all lanes).
__device__ __forceinline__ float group_sum(float x, int lpr) {
  for (int off = lpr >> 1; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
  return x;
}

__device__ __forceinline__ void load3(const float* p, float (&v)[3]) { v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; }

// MODE 0: forward.  MODE 1: backward over the destination-sorted CSR.
template <typename T, int VEC, int FE, bool RTFE, int MODE, int U>
__global__ void __launch_bounds__(kBlock) k_egnn_dst(EgnnArgs<T> a) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  const int lpr = 1 << a.lpr_log2;
  const int lig = lane & (lpr - 1);
  const int grp = lane >> a.lpr_log2;
  const int gpw = 64 >> a.lpr_log2;
  const int c0 = lig * VEC;
  const bool act = c0 < a.H;
  const int cc = act ? c0 : 0;

  float w[VEC][FE], wp[VEC], gwp[VEC][1];
  load_edge_weights<T, VEC, FE>(a.We, a.fe, c0, act, w);
  load_wp<T, VEC>(a.wp, c0, act, wp);
#pragma unroll
  for (int v = 0; v < VEC; ++v) gwp[v][0] = 0.f;

  const int64_t wave = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const int64_t stride = (int64_t)gridDim.x * (kBlock / 64) * gpw;
  for (int64_t tb = wave * gpw; tb < a.n_items; tb += stride) {
    const int64_t t = tb + grp;
    const bool valid = t < a.n_items;
    int4 it = valid ? a.items[t] : make_int4(0, 0, 0, -1);
    const int row = it.x, e0 = it.y, e1 = it.z;

    float ar[VEC], gr[VEC], acc[VEC], pd[3] = {0.f, 0.f, 0.f}, gpr[3] = {0.f, 0.f, 0.f}, sm[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int v = 0; v < VEC; ++v) { ar[v] = 0.f; gr[v] = 0.f; acc[v] = 0.f; }
    if (valid) {
      RowIO<T, VEC>::load(a.A + (int64_t)row * a.ldA + cc, ar);
      load3(a.pos_dst + (int64_t)row * 3, pd);
      if (MODE == 1) {
        RowIO<T, VEC>::load(a.G + (int64_t)row * a.ldG + cc, gr);
        load3(a.gp + (int64_t)row * 3, gpr);
      }
    }

    int nxt = (e0 + lig < e1) ? a.col[e0 + lig] : 0;
    for (int base = e0; base < e1; base += lpr) {
      const int cur = nxt;
      const int nb = (e1 - base) < lpr ? (e1 - base) : lpr;
      const int pn = base + lpr + lig;
      nxt = pn < e1 ? a.col[pn] : 0;
      for (int j = 0; j < nb; j += U) {
        float bv[U][VEC], ev[U][FE], ps[U][3];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int jj = (j + u) < nb ? (j + u) : (nb - 1);
          const int s = __shfl(cur, jj, lpr);
          RowIO<T, VEC>::load(a.B + (int64_t)s * a.ldB + cc, bv[u]);
          load_edge_attr<T, FE, RTFE>(a.ea, (int64_t)(base + jj), a.fe, ev[u]);
          load3(a.pos_src + (int64_t)s * 3, ps[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const bool live = (j + u) < nb;
          float ce[VEC], m[VEC];
          edge_term<VEC, FE>(w, ev[u], ce);
          float dot = 0.f;
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            m[v] = relu_f((ar[v] + bv[u][v]) + ce[v]);
            m[v] = live ? m[v] : 0.f;
            dot = fmaf(wp[v], m[v], dot);
          }
          const float se = group_sum(dot, lpr);
          if (MODE == 0) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] += m[v];
#pragma unroll
            for (int k = 0; k < 3; ++k) sm[k] = fmaf(pd[k] - ps[u][k], se, sm[k]);
          } else {
            float te = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) te = fmaf(gpr[k], pd[k] - ps[u][k], te);
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
              acc[v] += (m[v] > 0.f) ? fmaf(wp[v], te, gr[v]) : 0.f;
              gwp[v][0] = fmaf(te, m[v], gwp[v][0]);
            }
            sm[0] += se;
          }
        }
      }
    }

    if (valid) {
      if (it.w < 0) {
        if (act) RowIO<T, VEC>::store(a.out + (int64_t)row * a.ldo + c0, acc);
        if (lig == 0) {
          if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) a.small_out[(int64_t)row * 3 + k] = pd[k] + sm[k];
          } else {
            a.small_out[row] = sm[0];
          }
        }
      } else {
        if (act) {
          float* po = a.part_out + (int64_t)it.w * a.H + c0;
#pragma unroll
          for (int v = 0; v < VEC; ++v) po[v] = acc[v];
        }
        if (lig == 0) {
#pragma unroll
          for (int k = 0; k < 3; ++k) a.part_small[(int64_t)it.w * 4 + k] = sm[k];
        }
      }
    }
  }
  if (MODE == 1) block_reduce_lane_partials<VEC, 1>(gwp, lds, a.lane_partial, a.H, 1, a.lpr_log2);
}

// Backward over the source-sorted transpose.
template <typename T, int VEC, int FE, bool RTFE, int U>
__global__ void __launch_bounds__(kBlock) k_egnn_bwd_src(EgnnArgs<T> a) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  const int lpr = 1 << a.lpr_log2;
  const int lig = lane & (lpr - 1);
  const int grp = lane >> a.lpr_log2;
  const int gpw = 64 >> a.lpr_log2;
  const int c0 = lig * VEC;
  const bool act = c0 < a.H;
  const int cc = act ? c0 : 0;

  float w[VEC][FE], gw[VEC][FE], wp[VEC];
  load_edge_weights<T, VEC, FE>(a.We, a.fe, c0, act, w);
  load_wp<T, VEC>(a.wp, c0, act, wp);
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

    float br[VEC], acc[VEC], psr[3] = {0.f, 0.f, 0.f}, gps[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int v = 0; v < VEC; ++v) { br[v] = 0.f; acc[v] = 0.f; }
    if (valid) {
      RowIO<T, VEC>::load(a.B + (int64_t)row * a.ldB + cc, br);
      load3(a.pos_src + (int64_t)row * 3, psr);
    }

    int nxt = (e0 + lig < e1) ? a.col[e0 + lig] : 0;
    for (int base = e0; base < e1; base += lpr) {
      const int cur = nxt;
      const int nb = (e1 - base) < lpr ? (e1 - base) : lpr;
      const int pn = base + lpr + lig;
      nxt = pn < e1 ? a.col[pn] : 0;
      for (int j = 0; j < nb; j += U) {
        float av[U][VEC], gv[U][VEC], ev[U][FE], pdv[U][3], gpv[U][3];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int jj = (j + u) < nb ? (j + u) : (nb - 1);
          const int d = __shfl(cur, jj, lpr);
          RowIO<T, VEC>::load(a.A + (int64_t)d * a.ldA + cc, av[u]);
          RowIO<T, VEC>::load(a.G + (int64_t)d * a.ldG + cc, gv[u]);
          load_edge_attr<T, FE, RTFE>(a.ea, (int64_t)(base + jj), a.fe, ev[u]);
          load3(a.pos_dst + (int64_t)d * 3, pdv[u]);
          load3(a.gp + (int64_t)d * 3, gpv[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const bool live = (j + u) < nb;
          float ce[VEC], m[VEC];
          edge_term<VEC, FE>(w, ev[u], ce);
          float dot = 0.f;
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            m[v] = relu_f((av[u][v] + br[v]) + ce[v]);
            m[v] = live ? m[v] : 0.f;
            dot = fmaf(wp[v], m[v], dot);
          }
          const float se = group_sum(dot, lpr);
          float te = 0.f;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            te = fmaf(gpv[u][k], pdv[u][k] - psr[k], te);
            gps[k] = fmaf(-se, gpv[u][k], gps[k]);
          }
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            const float gm = (m[v] > 0.f) ? fmaf(wp[v], te, gv[u][v]) : 0.f;
            acc[v] += gm;
#pragma unroll
            for (int f = 0; f < FE; ++f) gw[v][f] = fmaf(gm, ev[u][f], gw[v][f]);
          }
        }
      }
    }

    if (valid) {
      if (it.w < 0) {
        if (act) RowIO<T, VEC>::store(a.out + (int64_t)row * a.ldo + c0, acc);
        if (lig == 0) {
#pragma unroll
          for (int k = 0; k < 3; ++k) a.small_out[(int64_t)row * 3 + k] = gps[k];
        }
      } else {
        if (act) {
          float* po = a.part_out + (int64_t)it.w * a.H + c0;
#pragma unroll
          for (int v = 0; v < VEC; ++v) po[v] = acc[v];
        }
        if (lig == 0) {
#pragma unroll
          for (int k = 0; k < 3; ++k) a.part_small[(int64_t)it.w * 4 + k] = gps[k];
        }
      }
    }
  }
  block_reduce_lane_partials<VEC, FE>(gw, lds, a.lane_partial, a.H, a.fe, a.lpr_log2);
}

// Per split row: sum of the [pieces, 4] small partials in piece order, plus an optional addend row.
__global__ void k_reduce_split_small(const float* __restrict__ parts, const int4* __restrict__ split, int64_t n_split,
                                     const float* __restrict__ addend, float* __restrict__ out, int ncomp) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n_split; s += stride) {
    const int4 sr = split[s];
    float tot[3] = {0.f, 0.f, 0.f};
    for (int c = 0; c < sr.z; ++c)
      for (int k = 0; k < ncomp; ++k) tot[k] += parts[(int64_t)(sr.y + c) * 4 + k];
    for (int k = 0; k < ncomp; ++k)
      out[(int64_t)sr.x * ncomp + k] = (addend ? addend[(int64_t)sr.x * ncomp + k] : 0.f) + tot[k];
  }
}

enum { kEgnnFwd = 0, kEgnnBwdDst = 1, kEgnnBwdSrc = 2 };

template <typename T, int VEC, int FE, bool RTFE>
static void launch_pass(int pass, const EgnnArgs<T>& a, dim3 grid, size_t lds, hipStream_t stream) {
  if (pass == kEgnnFwd) k_egnn_dst<T, VEC, FE, RTFE, 0, 4><<<grid, kBlock, 0, stream>>>(a);
  else if (pass == kEgnnBwdDst) k_egnn_dst<T, VEC, FE, RTFE, 1, 4><<<grid, kBlock, lds, stream>>>(a);
  else k_egnn_bwd_src<T, VEC, FE, RTFE, 2><<<grid, kBlock, lds, stream>>>(a);
}

template <typename T, int VEC>
static int32_t run_pass(int pass, EgnnArgs<T> a, const int32_t* split, int64_t n_split, void* lane_out,
                        const Geometry& g, hipStream_t stream) {
  a.lpr_log2 = g.lpr_log2;
  const dim3 grid(row_grid(a.n_items, g.lpr_log2), 1);
  const int tile = (1 << g.lpr_log2) * VEC;
  const int fe_t = a.fe == 4 ? 4 : (a.fe == 1 ? 1 : kMaxFe);
  const int lane_fe = pass == kEgnnBwdDst ? 1 : fe_t;
  const size_t lds = (size_t)(kBlock / 64) * tile * lane_fe * sizeof(float);
  if (a.fe == 4) launch_pass<T, VEC, 4, false>(pass, a, grid, lds, stream);
  else if (a.fe == 1) launch_pass<T, VEC, 1, false>(pass, a, grid, lds, stream);
  else launch_pass<T, VEC, kMaxFe, true>(pass, a, grid, lds, stream);
  MPNN_LAUNCH_CHECK("k_egnn");
  if (n_split > 0) {
    int32_t rc = launch_reduce_split<T, VEC>(a.part_out, split, n_split, a.out, a.ldo, a.H, g, stream);
    if (rc != MPNN_OK) return rc;
    const int ncomp = pass == kEgnnBwdDst ? 1 : 3;
    int64_t blocks = (n_split + kBlock - 1) / kBlock;
    if (blocks > kMaxBlocks) blocks = kMaxBlocks;
    k_reduce_split_small<<<(int)blocks, kBlock, 0, stream>>>(a.part_small, reinterpret_cast<const int4*>(split),
                                                             n_split, pass == kEgnnFwd ? a.pos_dst : nullptr,
                                                             a.small_out, ncomp);
    MPNN_LAUNCH_CHECK("k_reduce_split_small");
  }
  if (pass != kEgnnFwd) {
    const int count = pass == kEgnnBwdDst ? a.H : a.H * a.fe;
    k_reduce_gwe<T><<<count, kBlock, 0, stream>>>(a.lane_partial, (int)grid.x, count, static_cast<T*>(lane_out));
    MPNN_LAUNCH_CHECK("k_reduce_gwe");
  }
  return MPNN_OK;
}

static int32_t egnn_dispatch(int pass, const char* who, const void* A, int64_t ldA, const void* B, int64_t ldB,
                             const void* G, int64_t ldG, const float* gp, const void* ea, int32_t Fe, const void* We,
                             const void* wp, const float* pos_dst, const float* pos_src, const int32_t* col,
                             const int32_t* items, int64_t n_items, const int32_t* split, int64_t n_split,
                             int64_t n_parts, void* out, int64_t ldo, float* small_out, void* lane_out,
                             float* part_out, float* part_small, float* scratch, int64_t rows, int32_t H,
                             int32_t dtype, hipStream_t stream) {
  if (dtype != MPNN_F32 && dtype != MPNN_BF16) return fail(MPNN_E_INVALID, "%s: dtype %d", who, dtype);
  if (H < 1 || Fe < 1 || Fe > kMaxFe) return fail(MPNN_E_INVALID, "%s: H=%d (>=1), Fe=%d (1..%d)", who, H, Fe, kMaxFe);
  if (rows < 0 || n_items < 0 || n_split < 0 || n_parts < 0) return fail(MPNN_E_INVALID, "%s: negative size", who);
  const int full = dtype == MPNN_BF16 ? 8 : 4;
  if (H % full != 0 || H > 64 * full)
    return fail(MPNN_E_INVALID, "%s: H=%d must be a multiple of %d and at most %d", who, H, full, 64 * full);
  const size_t esz = dtype == MPNN_BF16 ? 2 : 4;
  if (rows == 0 || n_items == 0) {
    if (pass == kEgnnBwdDst && lane_out) MPNN_HIP_CHECK(hipMemsetAsync(lane_out, 0, (size_t)H * esz, stream));
    if (pass == kEgnnBwdSrc && lane_out) MPNN_HIP_CHECK(hipMemsetAsync(lane_out, 0, (size_t)H * Fe * esz, stream));
    return MPNN_OK;
  }
  if (!A || !B || !We || !wp || !pos_dst || !pos_src || !items || !out || !small_out)
    return fail(MPNN_E_INVALID, "%s: null pointer argument", who);
  if (pass != kEgnnFwd && (!G || !gp || !lane_out || !scratch))
    return fail(MPNN_E_INVALID, "%s: null gradient/scratch argument", who);
  if (ldA < H || ldB < H || ldo < H || (pass != kEgnnFwd && ldG < H))
    return fail(MPNN_E_INVALID, "%s: leading dimension below H", who);
  if (n_parts > 0 && (!part_out || !part_small || !split))
    return fail(MPNN_E_INVALID, "%s: split rows present but scratch slabs are null", who);
  if (((uintptr_t)items & 15) || ((uintptr_t)split & 15))
    return fail(MPNN_E_INVALID, "%s: items/split must be 16-byte aligned", who);
  const bool vec_ok = aligned16(A) && aligned16(B) && aligned16(out) && aligned16(ea) && aligned16(G) &&
                      ldA % full == 0 && ldB % full == 0 && ldo % full == 0 && ldG % full == 0;
  if (!vec_ok) return fail(MPNN_E_INVALID, "%s: feature matrices must be 16-byte aligned with ld %% %d == 0", who, full);
  const Geometry g = geometry(H, dtype, true);
  if (dtype == MPNN_F32) {
    EgnnArgs<float> a{static_cast<const float*>(A), ldA, static_cast<const float*>(B), ldB,
                      static_cast<const float*>(G), ldG, gp, static_cast<const float*>(ea), Fe,
                      static_cast<const float*>(We), static_cast<const float*>(wp), pos_dst, pos_src, col,
                      reinterpret_cast<const int4*>(items), n_items, static_cast<float*>(out), ldo, small_out,
                      part_out, part_small, scratch, H, 0};
    return run_pass<float, 4>(pass, a, split, n_split, lane_out, g, stream);
  }
  EgnnArgs<bf16_t> a{static_cast<const bf16_t*>(A), ldA, static_cast<const bf16_t*>(B), ldB,
                     static_cast<const bf16_t*>(G), ldG, gp, static_cast<const bf16_t*>(ea), Fe,
                     static_cast<const bf16_t*>(We), static_cast<const bf16_t*>(wp), pos_dst, pos_src, col,
                     reinterpret_cast<const int4*>(items), n_items, static_cast<bf16_t*>(out), ldo, small_out,
                     part_out, part_small, scratch, H, 0};
  return run_pass<bf16_t, 8>(pass, a, split, n_split, lane_out, g, stream);
}

}  // namespace mpnn

using namespace mpnn;

extern "C" {

int32_t mpnn_egnn_fwd(const void* A, int64_t ldA, const void* B, int64_t ldB, const void* ea_sorted, int32_t Fe,
                      const void* We, const void* w_pos, const float* pos_dst, const float* pos_src,
                      const int32_t* col, const int32_t* items, int64_t n_items, const int32_t* split,
                      int64_t n_split, int64_t n_parts, void* out, int64_t ldo, float* pos_out, float* part_out,
                      float* part_small, int64_t num_nodes, int32_t H, int32_t dtype, mpnn_stream_t stream) {
  return egnn_dispatch(kEgnnFwd, "mpnn_egnn_fwd", A, ldA, B, ldB, nullptr, ldA, nullptr, ea_sorted, Fe, We, w_pos,
                       pos_dst, pos_src, col, items, n_items, split, n_split, n_parts, out, ldo, pos_out, nullptr,
                       part_out, part_small, nullptr, num_nodes, H, dtype, stream);
}

int32_t mpnn_egnn_bwd_dst(const void* A, int64_t ldA, const void* B, int64_t ldB, const void* G, int64_t ldG,
                          const float* gpos_out, const void* ea_sorted, int32_t Fe, const void* We, const void* w_pos,
                          const float* pos_dst, const float* pos_src, const int32_t* col, const int32_t* items,
                          int64_t n_items, const int32_t* split, int64_t n_split, int64_t n_parts, void* gA,
                          int64_t ldgA, float* S, void* g_wpos, float* part_gA, float* part_small, float* scratch,
                          int64_t num_nodes, int32_t H, int32_t dtype, mpnn_stream_t stream) {
  return egnn_dispatch(kEgnnBwdDst, "mpnn_egnn_bwd_dst", A, ldA, B, ldB, G, ldG, gpos_out, ea_sorted, Fe, We, w_pos,
                       pos_dst, pos_src, col, items, n_items, split, n_split, n_parts, gA, ldgA, S, g_wpos, part_gA,
                       part_small, scratch, num_nodes, H, dtype, stream);
}

int32_t mpnn_egnn_bwd_src(const void* A, int64_t ldA, const void* B, int64_t ldB, const void* G, int64_t ldG,
                          const float* gpos_out, const void* ea_sorted_t, int32_t Fe, const void* We,
                          const void* w_pos, const float* pos_dst, const float* pos_src, const int32_t* col_t,
                          const int32_t* items_t, int64_t n_items_t, const int32_t* split_t, int64_t n_split_t,
                          int64_t n_parts_t, void* gB, int64_t ldgB, float* gpos_src, void* gWe, float* part_gB,
                          float* part_small, float* scratch, int64_t num_src_nodes, int32_t H, int32_t dtype,
                          mpnn_stream_t stream) {
  return egnn_dispatch(kEgnnBwdSrc, "mpnn_egnn_bwd_src", A, ldA, B, ldB, G, ldG, gpos_out, ea_sorted_t, Fe, We, w_pos,
                       pos_dst, pos_src, col_t, items_t, n_items_t, split_t, n_split_t, n_parts_t, gB, ldgB, gpos_src,
                       gWe, part_gB, part_small, scratch, num_src_nodes, H, dtype, stream);
}

}  // extern "C"
