// What one wave per SIMD gets out of the bf16 matrix pipe: v_mfma_f32_32x32x16_bf16 issued back to back
//   mode 0: 4 accumulators, chains of 6 dependent instructions per accumulator (the order of the split GEMMs)
//   mode 1: one accumulator, every instruction depends on the one before
//   mode 2: mode 0 with a second wave per SIMD running float/integer VALU work (the operand split) all the time
//   mode 3: mode 0 with 3 ds_read_b128 after every 12 matrix instructions
//   mode 4: mode 2 + mode 3
// Prints matrix instructions per microsecond and SIMD, and s_memtime ticks per microsecond.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ void __launch_bounds__(512) probe(int mode, int iters, float* out, long long* ticks) {
  __shared__ float lds[8192];
  const int wid = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 1.0f;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  if (wid >= 4) {   // the VALU wave of modes 2 and 4
    float v = threadIdx.x * 0.001f, w = 1.0f;
    unsigned u = threadIdx.x;
    for (int it = 0; it < iters * 24; ++it) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        u = __builtin_amdgcn_perm(u, __float_as_uint(v), 0x07060302u);
        v -= __uint_as_float(__float_as_uint(v) & 0xffff0000u);
        w = w * 1.0001f + v;
      }
    }
    if (v + w == 123.f) out[1] = v + (float)u;
    return;
  }
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
  f32x16 c0, c1, c2, c3;
  for (int r = 0; r < 16; ++r) { c0[r] = 0; c1[r] = 0; c2[r] = 0; c3[r] = 0; }
  const float4* lp = reinterpret_cast<const float4*>(lds) + (threadIdx.x & 63);
  float4 s = make_float4(0, 0, 0, 0);
  for (int it = 0; it < iters; ++it) {
    if (mode == 1) {
#pragma unroll
      for (int k = 0; k < 24; ++k) c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
    } else {
#pragma unroll
      for (int k = 0; k < 6; ++k) c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
#pragma unroll
      for (int k = 0; k < 6; ++k) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
      if (mode == 3 || mode == 4) { float4 x = lp[0], y = lp[64], z = lp[128]; s.x += x.x + y.y + z.z; }
#pragma unroll
      for (int k = 0; k < 6; ++k) c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
#pragma unroll
      for (int k = 0; k < 6; ++k) c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
      if (mode == 3 || mode == 4) { float4 x = lp[192], y = lp[256], z = lp[320]; s.y += x.x + y.y + z.z; }
    }
  }
  float t = s.x + s.y;
  for (int r = 0; r < 16; ++r) t += c0[r] + c1[r] + c2[r] + c3[r];
  if (t == 123.456f) out[0] = t;
  if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + wid] = __builtin_readcyclecounter() - t0;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 400;
  float* out; long long* ticks;
  hipMalloc(&out, 8); hipMalloc(&ticks, 256 * 4 * 8);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int mode = 0; mode < 5; ++mode) {
    const int threads = (mode == 2 || mode == 4) ? 512 : 256;
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(a);
      probe<<<256, threads>>>(mode, iters, out, ticks);
      hipEventRecord(b); hipEventSynchronize(b);
      hipEventElapsedTime(&ms, a, b);
    }
    long long h[1024]; hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < 1024; ++i) mean += h[i]; mean /= 1024;
    const double us = ms * 1e3, mfma = 24.0 * iters;
    printf("mode %d: %.1f us; %.1f matrix instructions per us and SIMD (%.1f ns each); wave: %.0f ticks = %.2f ticks/ns, %.1f ticks per instruction\n",
           mode, us, mfma / us, us * 1e3 / mfma, mean, mean / (us * 1e3), mean / mfma);
  }
  return 0;
}
