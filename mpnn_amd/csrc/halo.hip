// Packing and unpacking of the rows a partitioned layer exchanges between ranks (no reference counterpart: the
// reference runs on one device; the exchange stands in for the gathers of PyG's propagate, call sites
// models/layers.py:38,110, when the source rows of a rank's edges live on other ranks).
//
//   mpnn_halo_pack     send[i, :] = convert(table[index[i], :])          the rows this rank serves to its peers
//   mpnn_halo_unpack   out[r, :]  = sum_{k in [rowptr[r], rowptr[r+1])} convert(recv[pos[k], :])
//                      the gradient contributions received for this rank's rows, added per row in the order of the
//                      list (sender rank, then position: one fixed order, no atomics), rows nobody sent are zeros
// Rows move as 16-byte pieces when widths and addresses allow, element by element otherwise; sums are float32.
#include "common.h"
#include "rowio.h"

namespace mpnn {
namespace {

template <typename TI, typename TO, int VEC>
__global__ void __launch_bounds__(kBlock) k_halo_pack(const TI* __restrict__ table, int64_t ldt, const int32_t* __restrict__ index,
                                                      TO* __restrict__ send, int64_t rows, int H) {
  const int per_row = (H + VEC - 1) / VEC;
  const int64_t total = rows * per_row;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / per_row;
    const int c = (int)(i - r * per_row) * VEC;
    float v[VEC];
    RowIO<TI, VEC>::load(table + (int64_t)index[r] * ldt + c, v);
    RowIO<TO, VEC>::store(send + r * H + c, v);
  }
}

template <typename TI, typename TO, int VEC>
__global__ void __launch_bounds__(kBlock) k_halo_unpack(const TI* __restrict__ recv, const int32_t* __restrict__ rowptr,
                                                        const int32_t* __restrict__ pos, TO* __restrict__ out, int64_t ldo,
                                                        int64_t rows, int H) {
  const int per_row = (H + VEC - 1) / VEC;
  const int64_t total = rows * per_row;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / per_row;
    const int c = (int)(i - r * per_row) * VEC;
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
    for (int p = rowptr[r]; p < rowptr[r + 1]; ++p) {
      float v[VEC];
      RowIO<TI, VEC>::load(recv + (int64_t)pos[p] * H + c, v);
#pragma unroll
      for (int k = 0; k < VEC; ++k) acc[k] += v[k];
    }
    RowIO<TO, VEC>::store(out + r * ldo + c, acc);
  }
}

inline int grid_of(int64_t items) {
  int64_t b = (items + kBlock - 1) / kBlock;
  return (int)(b < 1 ? 1 : (b > 4 * kMaxBlocks ? 4 * kMaxBlocks : b));
}

inline bool wide(const void* a, int64_t lda, const void* b, int64_t ldb, int H, int vec_a_bytes, int vec_b_bytes) {
  return ((uintptr_t)a % 16 == 0) && ((uintptr_t)b % 16 == 0) && (lda * vec_a_bytes) % 16 == 0 &&
         (ldb * vec_b_bytes) % 16 == 0 && H % 8 == 0;
}

}  // namespace
}  // namespace mpnn

using namespace mpnn;

extern "C" {

int32_t mpnn_halo_pack(const void* table, int64_t ldt, int32_t table_dtype, const int32_t* index, void* send,
                       int32_t send_dtype, int64_t rows, int32_t H, mpnn_stream_t stream) {
  if (rows < 0 || H < 1 || ldt < H) return fail(MPNN_E_INVALID, "mpnn_halo_pack: rows=%lld H=%d ldt=%lld", (long long)rows, H, (long long)ldt);
  if (rows == 0) return MPNN_OK;
  if (!table || !index || !send) return fail(MPNN_E_INVALID, "mpnn_halo_pack: null pointer argument");
  const bool ti = table_dtype == MPNN_BF16, to = send_dtype == MPNN_BF16;
  if ((table_dtype != MPNN_F32 && !ti) || (send_dtype != MPNN_F32 && !to)) return fail(MPNN_E_INVALID, "mpnn_halo_pack: dtype");
  // one element per thread unless both sides take whole 16-byte pieces of 8 (bfloat16) / 4 (float32) columns
  if (!ti && !to && wide(table, ldt, send, H, H, 4, 4))
    k_halo_pack<float, float, 4><<<grid_of(rows * (H / 4)), kBlock, 0, stream>>>((const float*)table, ldt, index, (float*)send, rows, H);
  else if (ti && to && wide(table, ldt, send, H, H, 2, 2))
    k_halo_pack<bf16_t, bf16_t, 8><<<grid_of(rows * (H / 8)), kBlock, 0, stream>>>((const bf16_t*)table, ldt, index, (bf16_t*)send, rows, H);
  else if (!ti && !to)
    k_halo_pack<float, float, 1><<<grid_of(rows * H), kBlock, 0, stream>>>((const float*)table, ldt, index, (float*)send, rows, H);
  else if (!ti && to)
    k_halo_pack<float, bf16_t, 1><<<grid_of(rows * H), kBlock, 0, stream>>>((const float*)table, ldt, index, (bf16_t*)send, rows, H);
  else if (ti && !to)
    k_halo_pack<bf16_t, float, 1><<<grid_of(rows * H), kBlock, 0, stream>>>((const bf16_t*)table, ldt, index, (float*)send, rows, H);
  else
    k_halo_pack<bf16_t, bf16_t, 1><<<grid_of(rows * H), kBlock, 0, stream>>>((const bf16_t*)table, ldt, index, (bf16_t*)send, rows, H);
  MPNN_LAUNCH_CHECK("k_halo_pack");
  return MPNN_OK;
}

int32_t mpnn_halo_unpack(const void* recv, int32_t recv_dtype, const int32_t* rowptr, const int32_t* pos, void* out,
                         int64_t ldo, int32_t out_dtype, int64_t rows, int32_t H, mpnn_stream_t stream) {
  if (rows < 0 || H < 1 || ldo < H) return fail(MPNN_E_INVALID, "mpnn_halo_unpack: rows=%lld H=%d ldo=%lld", (long long)rows, H, (long long)ldo);
  if (rows == 0) return MPNN_OK;
  if (!rowptr || !out) return fail(MPNN_E_INVALID, "mpnn_halo_unpack: null pointer argument");
  const bool ti = recv_dtype == MPNN_BF16, to = out_dtype == MPNN_BF16;
  if ((recv_dtype != MPNN_F32 && !ti) || (out_dtype != MPNN_F32 && !to)) return fail(MPNN_E_INVALID, "mpnn_halo_unpack: dtype");
  if (!ti && !to && wide(recv, H, out, ldo, H, 4, 4))
    k_halo_unpack<float, float, 4><<<grid_of(rows * (H / 4)), kBlock, 0, stream>>>((const float*)recv, rowptr, pos, (float*)out, ldo, rows, H);
  else if (ti && to && wide(recv, H, out, ldo, H, 2, 2))
    k_halo_unpack<bf16_t, bf16_t, 8><<<grid_of(rows * (H / 8)), kBlock, 0, stream>>>((const bf16_t*)recv, rowptr, pos, (bf16_t*)out, ldo, rows, H);
  else if (!ti && !to)
    k_halo_unpack<float, float, 1><<<grid_of(rows * H), kBlock, 0, stream>>>((const float*)recv, rowptr, pos, (float*)out, ldo, rows, H);
  else if (!ti && to)
    k_halo_unpack<float, bf16_t, 1><<<grid_of(rows * H), kBlock, 0, stream>>>((const float*)recv, rowptr, pos, (bf16_t*)out, ldo, rows, H);
  else if (ti && !to)
    k_halo_unpack<bf16_t, float, 1><<<grid_of(rows * H), kBlock, 0, stream>>>((const bf16_t*)recv, rowptr, pos, (float*)out, ldo, rows, H);
  else
    k_halo_unpack<bf16_t, bf16_t, 1><<<grid_of(rows * H), kBlock, 0, stream>>>((const bf16_t*)recv, rowptr, pos, (bf16_t*)out, ldo, rows, H);
  MPNN_LAUNCH_CHECK("k_halo_unpack");
  return MPNN_OK;
}

}  // extern "C"
