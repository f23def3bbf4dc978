// Adam with on-the-fly transposition for a parameter whose gradient / operand copy live in the (N, P, C) layout
// (model 1's fc1.weight): the tile body shared by pointwise.hip's stand-alone kernel and simnn_disc.hip's one-launch
// optimizer step.
#pragma once
#include "gdm_common.h"

// derived terms of step `step` in double, like torch computes them on the host: lr / (1 - beta1^t), sqrt(1 - beta2^t)
__device__ __forceinline__ void adam_derived(const float* hyper, int step, float& step_size, float& bc2_sqrt) {
  const double b1 = (double)hyper[2], b2 = (double)hyper[3];
  step_size = (float)((double)hyper[1] / (1.0 - pow(b1, (double)step)));
  bc2_sqrt = (float)sqrt(1.0 - pow(b2, (double)step));
}

// One (128 p x 32 c) tile of row n of an (N, C, P) parameter (gdm_adam_step_dev_pc, gdm_simnn_adam_step), in two parts
// with a workgroup barrier between them (the caller's): the gather of the gradient tile needs nothing but addresses,
// so a caller that still waits for the step's bias-correction terms issues it first.
__device__ __forceinline__ void adam_pc_gather(float (&tile)[32][132], const float* __restrict__ g_pc, int C, int P,
                                               int bx, int by, int bz) {
  const int t = threadIdx.x, tx = t & 31, ty = t >> 5;
  const int p0 = bx * 128, c0 = by * 32;
  const int64_t base = (int64_t)bz * C * P;
#pragma unroll
  for (int i = 0; i < 16; ++i) {                                     // gradient tile, read along c
    const int pl = ty + 8 * i, pp = p0 + pl, c = c0 + tx;
    tile[tx][pl] = (pp < P && c < C) ? g_pc[base + (int64_t)pp * C + c] : 0.f;
  }
}

// step_size / bc2_sqrt: the derived bias-correction terms of THIS step.
template <typename TS>
__device__ __forceinline__ void adam_pc_update(float (&tile)[32][132], float* __restrict__ p, float* __restrict__ m,
                                               float* __restrict__ v, int C, int P, TS* __restrict__ shadow_pc,
                                               const float* __restrict__ hyper, int vec_ok, float step_size,
                                               float bc2_sqrt, int bx, int by, int bz) {
  const float w1 = 1.0f - hyper[2], beta2 = hyper[3], omb2 = 1.0f - hyper[3], eps = hyper[4], gscale = hyper[5];
  const int t = threadIdx.x, tx = t & 31, ty = t >> 5;
  const int p0 = bx * 128, c0 = by * 32, n = bz;
  const int64_t base = (int64_t)n * C * P;
  const bool full = vec_ok && p0 + 128 <= P && c0 + 32 <= C;      // 16-byte accesses to p, m, v
#pragma unroll
  for (int i = 0; i < 4; ++i) {                                      // p, m, v along p: four elements per thread
    const int cl = ty + 8 * i, c = c0 + cl, pp = p0 + 4 * tx;
    const int64_t k = base + (int64_t)c * P + pp;
    f32x4 gg = *(const f32x4*)&tile[cl][4 * tx], pv, mv, vv;
    if (full) {
#ifndef GDM_ADAM_DEFAULT_POLICY       /* p, m, v are read once and written once per step: non-temporal */
      pv = __builtin_nontemporal_load((const f32x4*)(p + k)); mv = __builtin_nontemporal_load((const f32x4*)(m + k));
      vv = __builtin_nontemporal_load((const f32x4*)(v + k));
#else
      pv = *(const f32x4*)(p + k); mv = *(const f32x4*)(m + k); vv = *(const f32x4*)(v + k);
#endif
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool ok = c < C && pp + e < P;
        pv[e] = ok ? p[k + e] : 0.f; mv[e] = ok ? m[k + e] : 0.f; vv[e] = ok ? v[k + e] : 0.f;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float pj = pv[e], mj = mv[e], vj = vv[e];
      adam_element(pj, mj, vj, gg[e], gscale, w1, beta2, omb2, eps, step_size, bc2_sqrt);
      pv[e] = pj; mv[e] = mj; vv[e] = vj;
    }
    if (full) {
#ifndef GDM_ADAM_DEFAULT_POLICY       /* p, m, v are read once and written once per step: non-temporal */
      __builtin_nontemporal_store(pv, (f32x4*)(p + k)); __builtin_nontemporal_store(mv, (f32x4*)(m + k));
      __builtin_nontemporal_store(vv, (f32x4*)(v + k));
#else
      *(f32x4*)(p + k) = pv; *(f32x4*)(m + k) = mv; *(f32x4*)(v + k) = vv;
#endif
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c < C && pp + e < P) { p[k + e] = pv[e]; m[k + e] = mv[e]; v[k + e] = vv[e]; }
    }
    *(f32x4*)&tile[cl][4 * tx] = pv;                                  // the elements this thread read: no hazard
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {                                     // updated weight, written along c
    const int pl = ty + 8 * i, pp = p0 + pl, c = c0 + tx;
    if (pp < P && c < C) shadow_pc[base + (int64_t)pp * C + c] = from_f32<TS>(tile[tx][pl]);
  }
}


template <typename TS>
__device__ __forceinline__ void adam_pc_tile(float (&tile)[32][132], float* __restrict__ p, const float* __restrict__ g_pc,
                                             float* __restrict__ m, float* __restrict__ v, int C, int P,
                                             TS* __restrict__ shadow_pc, const float* __restrict__ hyper, int vec_ok,
                                             float step_size, float bc2_sqrt, int bx, int by, int bz) {
  adam_pc_gather(tile, g_pc, C, P, bx, by, bz);
  __syncthreads();
  adam_pc_update<TS>(tile, p, m, v, C, P, shadow_pc, hyper, vec_ok, step_size, bc2_sqrt, bx, by, bz);
}
