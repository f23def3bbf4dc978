// Shared by gemm.hip (generic strided kernel) and gemm_bf16.hip (vectorised bf16-MFMA kernel).
#pragma once
#include "gdm_common.h"

struct GemmArgs {
  const void* A; int64_t sam, sak;
  const void* B; int64_t sbk, sbn;
  void* C; int c_dtype; int64_t scm, scn;
  int M, N, K;
  const float* bias_n; const float* bias_m; int act; float slope;
  int split_k, k_per_split; float* ws;
};

__device__ __forceinline__ void gemm_epilogue_store(const GemmArgs& g, int m, int n, float v) {
  if (g.bias_n) v += g.bias_n[n];
  if (g.bias_m) v += g.bias_m[m];
  v = apply_act(v, g.act, g.slope);
  store_from_f32(g.C, g.c_dtype, (int64_t)m * g.scm + (int64_t)n * g.scn, v);
}

// true if the vectorised kernel can take this problem (layouts contiguous along k or along m/n, 16-byte aligned
// rows, row-major C); launches it (and the split-K reduce is left to the caller).  Defined in gemm_bf16.hip.
bool gdm_gemm_bf16_fast_ok(const GemmArgs& g, int a_dtype, int b_dtype);
int gdm_gemm_bf16_fast_launch(const GemmArgs& g, int a_dtype, int b_dtype, hipStream_t s);
constexpr int GDM_GEMM_FAST_KT = 64;   // split-K slabs are multiples of the widest K tile
