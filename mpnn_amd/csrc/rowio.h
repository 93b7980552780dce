// Device-side row access helpers: a lane owns VEC consecutive columns of a feature row and
// moves them with one 4/8/16-byte access; arithmetic is always float.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mpnn {

typedef uint16_t bf16_t;  // bfloat16 storage

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// Round to nearest even; NaN stays NaN (quiet).
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x0040u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

template <typename T, int VEC>
struct RowIO;

// `Raw` is the lane's share of a row as it sits in memory; keeping gathered rows in this form until they are
// consumed halves the registers a bfloat16 row in flight occupies.
template <>
struct RowIO<float, 4> {
  typedef float4 Raw;
  static __device__ __forceinline__ Raw load_raw(const float* p) { return *reinterpret_cast<const float4*>(p); }
  static __device__ __forceinline__ void unpack(const Raw& t, float (&v)[4]) { v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
  static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
    float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
};

template <>
struct RowIO<float, 1> {
  typedef float Raw;
  static __device__ __forceinline__ Raw load_raw(const float* p) { return *p; }
  static __device__ __forceinline__ void unpack(const Raw& t, float (&v)[1]) { v[0] = t; }
  static __device__ __forceinline__ void load(const float* p, float (&v)[1]) { v[0] = *p; }
  static __device__ __forceinline__ void store(float* p, const float (&v)[1]) { *p = v[0]; }
};

template <>
struct RowIO<bf16_t, 8> {
  typedef uint4 Raw;
  static __device__ __forceinline__ Raw load_raw(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }
  static __device__ __forceinline__ void unpack(const Raw& t, float (&v)[8]) {
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
    v[4] = __uint_as_float(t.z << 16); v[5] = __uint_as_float(t.z & 0xffff0000u);
    v[6] = __uint_as_float(t.w << 16); v[7] = __uint_as_float(t.w & 0xffff0000u);
  }
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
    uint4 t = *reinterpret_cast<const uint4*>(p);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
    v[4] = __uint_as_float(t.z << 16); v[5] = __uint_as_float(t.z & 0xffff0000u);
    v[6] = __uint_as_float(t.w << 16); v[7] = __uint_as_float(t.w & 0xffff0000u);
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
    uint4 t;
    t.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
    t.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
    t.z = (uint32_t)f32_to_bf16(v[4]) | ((uint32_t)f32_to_bf16(v[5]) << 16);
    t.w = (uint32_t)f32_to_bf16(v[6]) | ((uint32_t)f32_to_bf16(v[7]) << 16);
    *reinterpret_cast<uint4*>(p) = t;
  }
};

template <>
struct RowIO<bf16_t, 1> {
  typedef bf16_t Raw;
  static __device__ __forceinline__ Raw load_raw(const bf16_t* p) { return *p; }
  static __device__ __forceinline__ void unpack(const Raw& t, float (&v)[1]) { v[0] = bf16_to_f32(t); }
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[1]) { v[0] = bf16_to_f32(*p); }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[1]) { *p = f32_to_bf16(v[0]); }
};

// Edge attributes of one edge: FE values of T, contiguous.
template <typename T, int FE>
struct EdgeIO {
  static __device__ __forceinline__ void load(const T* p, float (&v)[FE]) {
#pragma unroll
    for (int f = 0; f < FE; ++f) {
      if constexpr (sizeof(T) == 4) v[f] = (float)p[f];
      else v[f] = bf16_to_f32((bf16_t)p[f]);
    }
  }
};
template <>
struct EdgeIO<float, 4> {
  static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
    float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
};
template <>
struct EdgeIO<bf16_t, 4> {
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[4]) {
    uint2 t = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
  }
};

template <typename T>
__device__ __forceinline__ float scalar_to_f32(T v);
template <>
__device__ __forceinline__ float scalar_to_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ float scalar_to_f32<bf16_t>(bf16_t v) { return bf16_to_f32(v); }

template <typename T>
__device__ __forceinline__ T f32_to_scalar(float v);
template <>
__device__ __forceinline__ float f32_to_scalar<float>(float v) { return v; }
template <>
__device__ __forceinline__ bf16_t f32_to_scalar<bf16_t>(float v) { return f32_to_bf16(v); }

}  // namespace mpnn
