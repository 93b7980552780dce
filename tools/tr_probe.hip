// Prints what every lane receives from ds_read_b64_tr_b16 when lane L points at the 4 consecutive 16-bit
// elements 4L .. 4L+3 of LDS (element value = its index).  Build: hipcc --offload-arch=gfx950 -O2 tools/tr_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[256];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + 4 * threadIdx.x));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)v[j];
}
int main() {
  uint16_t* d; uint16_t h[256];
  if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 1;
  k<<<1, 64>>>(d);
  if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int j = 0; j < 4; ++j) printf("  (L%2d,e%d)", h[l * 4 + j] / 4, h[l * 4 + j] % 4);
    printf("\n");
  }
  return 0;
}
