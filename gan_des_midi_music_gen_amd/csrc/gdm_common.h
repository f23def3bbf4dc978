// Shared device/host helpers for libgdm_hip.so (gfx950 only; no other target is supported or compiled).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/gdm.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#define GDM_WAVE 64

// ---- host-side error plumbing (thread-local message, negative return codes) ------------------------------------
void gdm_set_error(const char* fmt, ...);

#define GDM_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      gdm_set_error(__VA_ARGS__);         \
      return GDM_EINVAL;                  \
    }                                     \
  } while (0)

#define GDM_LAUNCH_OK(name)                                                        \
  do {                                                                             \
    hipError_t e__ = hipGetLastError();                                            \
    if (e__ != hipSuccess) {                                                       \
      gdm_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));        \
      return GDM_ELAUNCH;                                                          \
    }                                                                              \
  } while (0)

static inline bool gdm_dtype_ok(int d) { return d == GDM_F32 || d == GDM_BF16; }
static inline size_t gdm_dtype_size(int d) { return d == GDM_BF16 ? 2 : 4; }

// ---- device helpers --------------------------------------------------------------------------------------------
__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(__bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 from_f32<__bf16>(float v) { return (__bf16)v; }

__device__ __forceinline__ float load_as_f32(const void* p, int dtype, int64_t i) {
  return dtype == GDM_BF16 ? (float)((const __bf16*)p)[i] : ((const float*)p)[i];
}
__device__ __forceinline__ void store_from_f32(void* p, int dtype, int64_t i, float v) {
  if (dtype == GDM_BF16) ((__bf16*)p)[i] = (__bf16)v; else ((float*)p)[i] = v;
}

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  switch (act) {
    case GDM_ACT_RELU: return v > 0.f ? v : 0.f;
    case GDM_ACT_LEAKY: return v > 0.f ? v : v * slope;
    case GDM_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    default: return v;
  }
}
// derivative of the activation expressed through its OUTPUT o (what the forward saved)
__device__ __forceinline__ float act_grad_from_out(float o, int act, float slope) {
  switch (act) {
    case GDM_ACT_RELU: return o > 0.f ? 1.f : 0.f;
    case GDM_ACT_LEAKY: return o > 0.f ? 1.f : slope;
    case GDM_ACT_SIGMOID: return o * (1.f - o);
    default: return 1.f;
  }
}

// One Adam element -- THE expression every kernel that applies the optimizer uses (pointwise.hip's three kernels and the
// model-2 slab sum that applies the step itself).  The multiply-adds are written out: left to the compiler, contraction
// differed from kernel to kernel and the same update came out one ulp apart.
__device__ __forceinline__ void adam_element(float& p, float& m, float& v, float g, float gscale, float w1, float beta2,
                                             float omb2, float eps, float step_size, float bc2_sqrt) {
#pragma clang fp contract(off)
  const float gj = g * gscale;
  const float d = gj - m;
  const float mj = (w1 < 0.5f) ? __builtin_fmaf(w1, d, m) : __builtin_fmaf(-(1.f - w1), d, gj);
  const float vj = __builtin_fmaf(omb2 * gj, gj, v * beta2);
  const float denom = sqrtf(vj) / bc2_sqrt + eps;
  p = __builtin_fmaf(-step_size, mj / denom, p);
  m = mj;
  v = vj;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// MFMA wrappers: D(16x16) += A(16xK) * B(Kx16); lane l holds A[l&15][kslice(l>>4)], B[kslice(l>>4)][l&15];
// D: col = l&15, row = 4*(l>>4) + reg.
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
