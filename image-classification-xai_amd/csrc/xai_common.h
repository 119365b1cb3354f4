// Shared helpers for the gfx950 kernels of libxai_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "xai_hip.h"

#define XAI_EXPORT extern "C" __attribute__((visibility("default")))

#define XAI_REQUIRE_PTR(p) \
  do {                     \
    if ((p) == nullptr) return XAI_E_NULL; \
  } while (0)
#define XAI_REQUIRE(cond, code) \
  do {                          \
    if (!(cond)) return (code); \
  } while (0)

// hipGetLastError after a launch: >0 on failure, 0 on success.
static inline int xai_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? XAI_OK : static_cast<int>(e);
}

static inline bool xai_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static inline int64_t xai_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Compute units of the current device (256 on MI355X).  One definition for the whole library (abi.hip): the per-device
// value is queried once and published through a std::atomic, so concurrent first calls from several host threads race
// only on storing the same number.
int xai_cu_count();

constexpr int kWave = 64;  // gfx950 wavefront

// Wave-wide reductions over 64 lanes (butterfly through DPP/bpermute shuffles).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, kWave));
  return v;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
