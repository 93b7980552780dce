// Per-CU load throughput probe: every block (one per CU, 256 threads) sums 16-byte loads from
//   mode 0: a small buffer shared by all blocks (L2-resident, `small_kb` KiB), whole 1-KiB pieces per wave
//   mode 1: its own slice of a big buffer (HBM / Infinity Cache), rows of `row_bytes` with `piece` contiguous bytes
//           fetched per row and step (the shape of an operand tile load)
//   mode 2: both at once, alternating
// with `depth` loads in flight per lane.  Prints bytes per clock and CU.  usage: l2_probe mode small_kb piece depth waves
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int DEPTH>
__global__ void __launch_bounds__(512) probe(const float4* __restrict__ small, size_t small_n, const float4* __restrict__ big,
                                              size_t rows_per_block, int row_f4, int piece_f4, int mode, int iters,
                                              float* out, long long* cycles) {
  const int tid = threadIdx.x, nt = blockDim.x;
  float4 acc[DEPTH];
  for (int d = 0; d < DEPTH; ++d) acc[d] = make_float4(0, 0, 0, 0);
  const long long t0 = __builtin_readcyclecounter();
  const float4* mine = big + (size_t)blockIdx.x * rows_per_block * row_f4;
  const int lanes_per_row = piece_f4;                  // lanes covering one row piece
  const int rows_per_step = nt / lanes_per_row;
  size_t s_off = tid, b_step = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      float4 v;
      if (mode == 0 || (mode == 2 && (d & 1))) {
        v = small[s_off];
        s_off += nt;
        if (s_off >= small_n) s_off -= small_n;
      } else {
        // step b_step: rows [r0, r0 + rows_per_step), piece p of each row
        const size_t pieces_per_row = row_f4 / piece_f4;
        const size_t r0 = (b_step / pieces_per_row) * rows_per_step % rows_per_block;
        const size_t p = b_step % pieces_per_row;
        v = mine[(r0 + tid / lanes_per_row) * row_f4 + p * piece_f4 + tid % lanes_per_row];
        ++b_step;
      }
      acc[d].x += v.x; acc[d].y += v.y; acc[d].z += v.z; acc[d].w += v.w;
    }
  }
  float s = 0;
  for (int d = 0; d < DEPTH; ++d) s += acc[d].x + acc[d].y + acc[d].z + acc[d].w;
  if (s == 123.456f) out[0] = s;
  if (tid == 0) cycles[blockIdx.x] = __builtin_readcyclecounter() - t0;
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  const int small_kb = argc > 2 ? atoi(argv[2]) : 512;
  const int piece = argc > 3 ? atoi(argv[3]) : 64;       // bytes
  const int depth = argc > 4 ? atoi(argv[4]) : 8;
  const int waves = argc > 5 ? atoi(argv[5]) : 4;
  const int blocks = 256, row_bytes = 2048;
  const size_t rows_per_block = 2048;                    // 4 MiB per block, 1 GiB in all
  const size_t small_n = (size_t)small_kb * 1024 / 16;
  float4 *small, *big; float* out; long long* cyc;
  hipMalloc(&small, small_n * 16); hipMemset(small, 0, small_n * 16);
  hipMalloc(&big, (size_t)blocks * rows_per_block * row_bytes); hipMemset(big, 0, (size_t)blocks * rows_per_block * row_bytes);
  hipMalloc(&out, 4); hipMalloc(&cyc, blocks * 8);
  const int iters = 4096 / depth;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
#define GO(D) probe<D><<<blocks, waves * 64, 0, 0>>>(small, small_n, big, rows_per_block, row_bytes / 16, piece / 16, mode, iters, out, cyc)
    if (depth == 4) GO(4); else if (depth == 8) GO(8); else if (depth == 16) GO(16); else GO(32);
    hipEventRecord(b); hipEventSynchronize(b);
  }
  float ms; hipEventElapsedTime(&ms, a, b);
  std::vector<long long> h(blocks); hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
  double mean = 0; for (auto v : h) mean += v; mean /= blocks;
  const double bytes_per_block = (double)iters * depth * waves * 64 * 16;
  printf("mode %d small %d KiB piece %d B depth %d waves %d: %.1f us, %.0f cycles(100MHz?) -> %.2f GB/s per CU, %.2f TB/s chip\n",
         mode, small_kb, piece, depth, waves, ms * 1e3, mean, bytes_per_block / (ms * 1e-3) / 1e9, bytes_per_block * blocks / (ms * 1e-3) / 1e12);
  return 0;
}
