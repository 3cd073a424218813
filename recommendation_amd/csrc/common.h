// Shared host/device helpers for libmi355rec (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/mi355rec.h"

#define MI_WAVE 64

void mi_set_error(const char* fmt, ...);
int mi_fm_check_table(const mi_fm_table* t, int opt, const char* who);

#define MI_REQUIRE(cond, ...)        \
  do {                               \
    if (!(cond)) {                   \
      mi_set_error(__VA_ARGS__);     \
      return MI_ERR_INVALID_ARG;     \
    }                                \
  } while (0)

#define MI_HIP(call)                                                          \
  do {                                                                        \
    hipError_t e_ = (call);                                                   \
    if (e_ != hipSuccess) {                                                   \
      mi_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),     \
                   __FILE__, __LINE__);                                       \
      return MI_ERR_HIP;                                                      \
    }                                                                         \
  } while (0)

#define MI_LAUNCH_CHECK() MI_HIP(hipGetLastError())

static inline int64_t mi_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

#ifdef __HIPCC__
// id loads for either index width
template <typename IDX>
__device__ __forceinline__ int64_t mi_ld_id(const void* p, int64_t i) {
  return (int64_t) reinterpret_cast<const IDX*>(p)[i];
}

__device__ __forceinline__ float mi_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float mi_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// round-to-nearest-even f32 -> bf16 through the hardware conversion
__device__ __forceinline__ unsigned short mi_f2bf(float f) {
  __bf16 h = (__bf16) f;
  return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float mi_bf2f(unsigned short u) {
  return __uint_as_float(((unsigned int) u) << 16);
}
#endif
