// Shared device/host helpers for the gfx950 kernels.  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/frmap_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

struct BF16 {
  using elem = __bf16;
  using vec8 = bf16x8_t;
  static __device__ __forceinline__ f32x4_t mfma(vec8 a, vec8 b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ float to_f32(elem v) { return (float)v; }
  static __device__ __forceinline__ elem from_f32(float v) { return (elem)v; }
};
struct F16 {
  using elem = _Float16;
  using vec8 = f16x8_t;
  static __device__ __forceinline__ f32x4_t mfma(vec8 a, vec8 b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ float to_f32(elem v) { return (float)v; }
  static __device__ __forceinline__ elem from_f32(float v) { return (elem)v; }
};

// pack 4 fp32 -> 4 T (8 bytes)
template <typename TT>
__device__ __forceinline__ u32x2_t pack4(float a, float b, float c, float d) {
  typename TT::elem e[4] = {TT::from_f32(a), TT::from_f32(b), TT::from_f32(c), TT::from_f32(d)};
  u32x2_t r;
  __builtin_memcpy(&r, e, 8);
  return r;
}
template <typename TT>
__device__ __forceinline__ void unpack4(u32x2_t v, float* o) {
  typename TT::elem e[4];
  __builtin_memcpy(e, &v, 8);
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = TT::to_f32(e[i]);
}
template <typename TT>
__device__ __forceinline__ void unpack8(u32x4_t v, float* o) {
  typename TT::elem e[8];
  __builtin_memcpy(e, &v, 16);
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = TT::to_f32(e[i]);
}
template <typename TT>
__device__ __forceinline__ u32x4_t pack8(const float* f) {
  typename TT::elem e[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) e[i] = TT::from_f32(f[i]);
  u32x4_t r;
  __builtin_memcpy(&r, e, 16);
  return r;
}

// exact floor(n / d) for n, d < 65536 with magic = ceil(2^32 / d)
__host__ __device__ inline uint32_t frmap_magic(uint32_t d) {
  return (uint32_t)(((1ull << 32) + d - 1) / d);
}
__device__ __forceinline__ uint32_t fast_div(uint32_t n, uint32_t magic) { return __umulhi(n, magic); }

// host-side error plumbing
void frmap_set_error(const char* fmt, ...);
#define FRMAP_REQUIRE(cond, ...)        \
  do {                                  \
    if (!(cond)) {                      \
      frmap_set_error(__VA_ARGS__);     \
      return -1;                        \
    }                                   \
  } while (0)
#define FRMAP_LAUNCH_CHECK()                                           \
  do {                                                                 \
    hipError_t e__ = hipGetLastError();                                \
    if (e__ != hipSuccess) {                                           \
      frmap_set_error("launch failed: %s", hipGetErrorString(e__));    \
      return -2;                                                       \
    }                                                                  \
  } while (0)
