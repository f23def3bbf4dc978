// Wavefront-level fused kernels around the GEMMs: BCE-with-logits (+ model 1's sigmoid chain), Adam, train-mode
// batch norm + activation (forward/backward), bias+activation, column sums, casts.
// All reductions are fixed-order (no float atomics) so a training run is bit-reproducible on one device.
#include "gdm_common.h"
#include "adam_pc.h"

namespace {

// ------------------------------------------------------------------------------------------------ BCE with logits
// mean_i( max(x,0) - x*y + log1p(exp(-|x|)) ); nn.BCEWithLogitsLoss() at SIMNN.py:257 / network_tests.py:248.
__global__ __launch_bounds__(1024) void bce_kernel(const float* __restrict__ x, float target, int n, float gscale,
                                                   float* __restrict__ loss, float* __restrict__ dx, int fuse_sig,
                                                   int accumulate) {
  __shared__ float red[1024];
  const int t = threadIdx.x;
  float s = 0.f;
  for (int i = t; i < n; i += 1024) {
    const float v = x[i];
    s += fmaxf(v, 0.f) - v * target + log1pf(expf(-fabsf(v)));
    if (dx) {
      float g = (1.0f / (1.0f + expf(-v)) - target) * gscale / (float)n;
      if (fuse_sig) g *= v * (1.f - v);   // v is itself sigmoid(z): chain to d/dz
      dx[i] = g;
    }
  }
  red[t] = s;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (t < o) red[t] += red[t + o];
    __syncthreads();
  }
  if (t == 0) loss[0] = (accumulate ? loss[0] : 0.f) + red[0] / (float)n;
}

// ------------------------------------------------------------------------------------------------------------ Adam
// torch.optim.Adam single-tensor update (lerp form of exp_avg, sqrt/bias-correction/eps order as torch).
template <bool VEC>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                   float w1, float beta2, float one_minus_b2, float step_size,
                                                   float bc2_sqrt, float eps, float gscale) {
  const int64_t stride = (int64_t)gridDim.x * 256 * 4;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
    if (VEC && i + 3 < n) {
      f32x4 pp = *(const f32x4*)(p + i), gg = *(const f32x4*)(g + i), mm = *(const f32x4*)(m + i),
            vv = *(const f32x4*)(v + i);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float pj = pp[j], mj = mm[j], vj = vv[j];
        adam_element(pj, mj, vj, gg[j], gscale, w1, beta2, one_minus_b2, eps, step_size, bc2_sqrt);
        pp[j] = pj; mm[j] = mj; vv[j] = vj;
      }
      *(f32x4*)(p + i) = pp;
      *(f32x4*)(m + i) = mm;
      *(f32x4*)(v + i) = vv;
    } else {
      for (int64_t k = i; k < n && k < i + 4; ++k) {
        adam_element(p[k], m[k], v[k], g[k], gscale, w1, beta2, one_minus_b2, eps, step_size, bc2_sqrt);
      }
    }
  }
}

// Device-resident optimizer state for graph replay: hyper[0] = step (int bits), [1] lr, [2] beta1, [3] beta2, [4] eps,
// [5] grad_scale, [6] step_size (derived), [7] sqrt(bias_correction2) (derived).  adam_prep advances the step and
// derives the two bias-correction terms in double, like torch does on the host.
__global__ void adam_prep_kernel(float* __restrict__ hyper) {
  int step = __float_as_int(hyper[0]) + 1;
  hyper[0] = __int_as_float(step);
  float ss, bq;
  adam_derived(hyper, step, ss, bq);
  hyper[6] = ss;
  hyper[7] = bq;
}

// Adam on a parameter whose GRADIENT arrives in the channels-last ("permuted") layout and whose updated value is also
// wanted there as a GEMM operand: model 1's fc1.weight (128, 32, P).  The weight gradient GEMM writes dW (N, P, C) in the
// flatten order of the channels-last feature map; the parameter and its moments stay in the reference's (N, C, P) order
// (state_dict / checkpoint compatible); the forward and dX GEMMs read a (N, P, C) copy in the activation dtype.
// Instead of two transposing passes around Adam (gradient -> parameter order, new weight -> operand order: 235 MB of
// HBM traffic per step) the transposes happen inside this kernel, through LDS: a workgroup owns a (128 p x 32 c) tile
// of one row n, reads the gradient tile along c (128-byte runs), updates p/m/v along p (512-byte runs) and writes the
// new weight back along c.
// (the Adam arithmetic is the same expression order as adam_dev_kernel: bit-identical results)
template <typename TS>
__global__ __launch_bounds__(256) void adam_dev_pc_kernel(float* __restrict__ p, const float* __restrict__ g_pc,
                                                          float* __restrict__ m, float* __restrict__ v, int C, int P,
                                                          TS* __restrict__ shadow_pc,
                                                          const float* __restrict__ hyper, int vec_ok) {
  __shared__ __attribute__((aligned(16))) float tile[32][132];        // [c][p], rows 16-byte aligned
  adam_pc_tile<TS>(tile, p, g_pc, m, v, C, P, shadow_pc, hyper, vec_ok, hyper[6], hyper[7], blockIdx.x, blockIdx.y,
                   blockIdx.z);
}

template <bool VEC>
__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                       const float* __restrict__ hyper) {
  const float w1 = 1.0f - hyper[2], beta2 = hyper[3], omb2 = 1.0f - hyper[3], eps = hyper[4], gscale = hyper[5];
  const float step_size = hyper[6], bc2_sqrt = hyper[7];
  const int64_t stride = (int64_t)gridDim.x * 256 * 4;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
    if (VEC && i + 3 < n) {
      f32x4 pp = *(const f32x4*)(p + i), gg = *(const f32x4*)(g + i), mm = *(const f32x4*)(m + i),
            vv = *(const f32x4*)(v + i);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float pj = pp[j], mj = mm[j], vj = vv[j];
        adam_element(pj, mj, vj, gg[j], gscale, w1, beta2, omb2, eps, step_size, bc2_sqrt);
        pp[j] = pj; mm[j] = mj; vv[j] = vj;
      }
      *(f32x4*)(p + i) = pp;
      *(f32x4*)(m + i) = mm;
      *(f32x4*)(v + i) = vv;
    } else {
      for (int64_t k = i; k < n && k < i + 4; ++k) {
        adam_element(p[k], m[k], v[k], g[k], gscale, w1, beta2, omb2, eps, step_size, bc2_sqrt);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------- batch norm fwd
// Stage 1: per (row-chunk, column) Welford triple (count, mean, M2); lanes = consecutive columns, the 4 waves of a
// workgroup interleave rows and are merged in wave order (Chan's formula).
struct Welford { float n, mean, m2; };
__device__ __forceinline__ void welford_merge(Welford& a, const Welford& b) {
  if (b.n == 0.f) return;
  const float n = a.n + b.n;
  const float d = b.mean - a.mean;
  a.mean += d * (b.n / n);
  a.m2 += b.m2 + d * d * (a.n * b.n / n);
  a.n = n;
}

// A thread's share of a chunk is at most a few dozen values: it keeps SHIFTED sums (shift = its first value, so the
// sums stay small whatever the channel's mean is) with eight loads in flight, and converts them to a Welford triple
// once.  `CW` = lanes per row (power of two >= min(C, 64)): with fewer than 64 channels a wave covers 64 / CW rows per
// load, so no lane idles.
__device__ __forceinline__ Welford welford_from_shifted(float n, float shift, float s, float ss) {
  Welford a{n, 0.f, 0.f};
  if (n > 0.f) {
    a.mean = shift + s / n;
    a.m2 = fmaxf(ss - s * (s / n), 0.f);
  }
  return a;
}

__global__ __launch_bounds__(256) void bn_partial_stats(const float* __restrict__ y, int rows, int C, int chunk_rows,
                                                        int CW, float* __restrict__ ws) {
  __shared__ Welford sh[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int rpl = 64 / CW;                                   // rows per wave-wide load
  const int c = blockIdx.x * 64 + (lane & (CW - 1)), sub = lane / CW;
  const int r0 = blockIdx.y * chunk_rows, r1 = min(rows, r0 + chunk_rows);
  const int step = 4 * rpl;                                  // the 4 waves interleave row groups
  float n = 0.f, shift = 0.f, s = 0.f, ss = 0.f;
  if (c < C) {
    int r = r0 + wv * rpl + sub;
    if (r < r1) shift = y[(int64_t)r * C + c];
    for (; r + 7 * step < r1; r += 8 * step) {
      float x[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) x[q] = y[(int64_t)(r + q * step) * C + c];
#pragma unroll
      for (int q = 0; q < 8; ++q) { const float d = x[q] - shift; s += d; ss += d * d; }
      n += 8.f;
    }
    for (; r < r1; r += step) { const float d = y[(int64_t)r * C + c] - shift; s += d; ss += d * d; n += 1.f; }
  }
  sh[wv][lane] = welford_from_shifted(n, shift, s, ss);
  __syncthreads();
  if (wv == 0 && sub == 0 && c < C) {
    Welford t{0.f, 0.f, 0.f};
    for (int w = 0; w < 4; ++w)
      for (int q = 0; q < rpl; ++q) welford_merge(t, sh[w][q * CW + lane]);       // fixed order
    float* o = ws + ((int64_t)blockIdx.y * C + c) * 3;
    o[0] = t.n; o[1] = t.mean; o[2] = t.m2;
  }
}

// Stage 2: a 1024-thread workgroup owns 64 channels; wave w merges chunks w, w+16, ... in order, then the 16 partial
// triples are merged in wave order (fixed order => deterministic).
__global__ __launch_bounds__(1024) void bn_finalize(const float* __restrict__ ws, int chunks, int rows, int C,
                                                    float momentum, float eps, float* __restrict__ running_mean,
                                                    float* __restrict__ running_var, int64_t* __restrict__ nbt,
                                                    float* __restrict__ save_mean, float* __restrict__ save_invstd) {
  __shared__ Welford part[16][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) nbt[0] += 1;
  Welford t{0.f, 0.f, 0.f};
  if (c < C)
    for (int k = wv; k < chunks; k += 16 * 8) {                 // eight triples in flight, merged in chunk order
      Welford b[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int kk = k + 16 * q;
        const float* o = ws + ((int64_t)min(kk, chunks - 1) * C + c) * 3;
        b[q] = Welford{kk < chunks ? o[0] : 0.f, o[1], o[2]};
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) welford_merge(t, b[q]);
    }
  part[wv][lane] = t;
  __syncthreads();
  if (wv != 0 || c >= C) return;
  t = part[0][lane];
#pragma unroll
  for (int w = 1; w < 16; ++w) welford_merge(t, part[w][lane]);
  const float var_b = t.m2 / (float)rows;
  save_mean[c] = t.mean;
  save_invstd[c] = 1.0f / sqrtf(var_b + eps);
  if (running_mean) {
    const float var_u = t.m2 / (float)max(rows - 1, 1);
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * t.mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * var_u;
  }
}

__global__ __launch_bounds__(256) void bn_eval_stats(const float* __restrict__ running_mean,
                                                     const float* __restrict__ running_var, int C, float eps,
                                                     float* __restrict__ save_mean, float* __restrict__ save_invstd) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  save_mean[c] = running_mean[c];
  save_invstd[c] = 1.0f / sqrtf(running_var[c] + eps);
}

// (IT: 32-bit element indices when they fit -- a 64-bit modulo per element made this pass instruction-bound)
template <typename IT>
__global__ __launch_bounds__(256) void bn_apply(const float* __restrict__ y, int64_t total64, int C,
                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                                int act, void* __restrict__ out, int out_dtype) {
  const IT total = (IT)total64, stride = (IT)gridDim.x * 256;
  for (IT i = (IT)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % C);
    const float alpha = invstd[c] * gamma[c];
    const float v = (y[i] - mean[c]) * alpha + beta[c];
    store_from_f32(out, out_dtype, i, apply_act(v, act, 0.f));
  }
}

// ------------------------------------------------------------------------------------------------- batch norm bwd
// g = dout * act'(out); s1 = sum g, s2 = sum g*xhat (two-stage, fixed order); dy = gamma*invstd*(g - s1/R - xhat*s2/R)
__global__ __launch_bounds__(256) void bn_bwd_partial(const void* __restrict__ dout, const void* __restrict__ out,
                                                      int dtype, const float* __restrict__ y, int rows, int C,
                                                      int chunk_rows, const float* __restrict__ mean,
                                                      const float* __restrict__ invstd, int act,
                                                      float* __restrict__ ws) {
  __shared__ float sh1[4][64], sh2[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int r0 = blockIdx.y * chunk_rows, r1 = min(rows, r0 + chunk_rows);
  float s1 = 0.f, s2 = 0.f;
  if (c < C) {
    const float mu = mean[c], is = invstd[c];
    for (int r = r0 + wv; r < r1; r += 4) {
      const int64_t i = (int64_t)r * C + c;
      const float g = load_as_f32(dout, dtype, i) * act_grad_from_out(load_as_f32(out, dtype, i), act, 0.f);
      s1 += g;
      s2 += g * ((y[i] - mu) * is);
    }
  }
  sh1[wv][lane] = s1; sh2[wv][lane] = s2;
  __syncthreads();
  if (wv == 0 && c < C) {
    float* o = ws + ((int64_t)blockIdx.y * C + c) * 2;
    o[0] = ((sh1[0][lane] + sh1[1][lane]) + sh1[2][lane]) + sh1[3][lane];
    o[1] = ((sh2[0][lane] + sh2[1][lane]) + sh2[2][lane]) + sh2[3][lane];
  }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize(const float* __restrict__ ws, int chunks, int C,
                                                       float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float s1 = 0.f, s2 = 0.f;
  for (int k = 0; k < chunks; ++k) {
    s1 += ws[((int64_t)k * C + c) * 2];
    s2 += ws[((int64_t)k * C + c) * 2 + 1];
  }
  dbeta[c] = s1;
  dgamma[c] = s2;
}

__global__ __launch_bounds__(256) void bn_bwd_apply(const void* __restrict__ dout, const void* __restrict__ out,
                                                    int dtype, const float* __restrict__ y, int64_t total, int rows,
                                                    int C, const float* __restrict__ gamma,
                                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                                    const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                    int act, float* __restrict__ dy) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  const float inv_r = 1.0f / (float)rows;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % C);
    const float g = load_as_f32(dout, dtype, i) * act_grad_from_out(load_as_f32(out, dtype, i), act, 0.f);
    const float xhat = (y[i] - mean[c]) * invstd[c];
    dy[i] = gamma[c] * invstd[c] * (g - dbeta[c] * inv_r - xhat * dgamma[c] * inv_r);
  }
}

// ------------------------------------------------------------------------------------------------ small pointwise
__global__ __launch_bounds__(256) void bias_act_kernel(const float* __restrict__ x, const float* __restrict__ bias,
                                                       int64_t total, int cols, int act, float slope,
                                                       void* __restrict__ out, int out_dtype) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    float v = x[i];
    if (bias) v += bias[i % cols];
    store_from_f32(out, out_dtype, i, apply_act(v, act, slope));
  }
}

__global__ __launch_bounds__(256) void act_bwd_kernel(const void* __restrict__ dout, const void* __restrict__ out,
                                                      int dtype, int64_t n, int act, float slope,
                                                      void* __restrict__ dx) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float g = load_as_f32(dout, dtype, i) * act_grad_from_out(load_as_f32(out, dtype, i), act, slope);
    store_from_f32(dx, dtype, i, g);
  }
}

__global__ __launch_bounds__(256) void colsum_partial(const void* __restrict__ x, int dtype, int rows, int C,
                                                      int chunk_rows, float* __restrict__ ws) {
  __shared__ float sh[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int r0 = blockIdx.y * chunk_rows, r1 = min(rows, r0 + chunk_rows);
  float s = 0.f;
  if (c < C)
    for (int r = r0 + wv; r < r1; r += 4) s += load_as_f32(x, dtype, (int64_t)r * C + c);
  sh[wv][lane] = s;
  __syncthreads();
  if (wv == 0 && c < C) ws[(int64_t)blockIdx.y * C + c] = ((sh[0][lane] + sh[1][lane]) + sh[2][lane]) + sh[3][lane];
}

__global__ __launch_bounds__(256) void colsum_final(const float* __restrict__ ws, int chunks, int C,
                                                    float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int k = 0; k < chunks; ++k) s += ws[(int64_t)k * C + c];
  out[c] = s;
}

__global__ __launch_bounds__(256) void cast_kernel(const void* __restrict__ src, int sd, void* __restrict__ dst,
                                                   int dd, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
    store_from_f32(dst, dd, i, load_as_f32(src, sd, i));
}

inline int row_chunks(int rows) {
  int c = (rows + 255) / 256;        // >= 256 rows per chunk; at most 64 chunks (one round of the finalize kernel)
  return c < 1 ? 1 : (c > 64 ? 64 : c);
}
inline unsigned grid_for(int64_t total) {
  int64_t b = (total + 255) / 256;
  return (unsigned)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int gdm_bce_with_logits(const float* x, float target, int n, float grad_scale, float* loss, float* dx,
                                   int fuse_sigmoid_backward, int accumulate_loss, void* stream) {
  GDM_REQUIRE(x && loss, "gdm_bce_with_logits: null pointer");
  GDM_REQUIRE(n > 0 && n <= 65536, "gdm_bce_with_logits: n=%d out of range (1..65536)", n);
  hipLaunchKernelGGL(bce_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, target, n, grad_scale, loss, dx,
                     fuse_sigmoid_backward, accumulate_loss);
  GDM_LAUNCH_OK("gdm_bce_with_logits");
  return GDM_OK;
}

extern "C" int gdm_adam_step(float* p, const float* g, float* m, float* v, int64_t n, int step, float lr, float beta1,
                             float beta2, float eps, float grad_scale, void* stream) {
  GDM_REQUIRE(p && g && m && v, "gdm_adam_step: null pointer");
  GDM_REQUIRE(n > 0 && step >= 1, "gdm_adam_step: bad n=%lld or step=%d", (long long)n, step);
  const bool aligned = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 8192) blocks = 8192;
  if (aligned)
    hipLaunchKernelGGL(adam_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n,
                       1.0f - beta1, beta2, 1.0f - beta2, step_size, bc2_sqrt, eps, grad_scale);
  else
    hipLaunchKernelGGL(adam_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n,
                       1.0f - beta1, beta2, 1.0f - beta2, step_size, bc2_sqrt, eps, grad_scale);
  GDM_LAUNCH_OK("gdm_adam_step");
  return GDM_OK;
}

extern "C" int gdm_adam_step_dev_pc(float* p, const float* g_pc, float* m, float* v, int N, int C, int P,
                                    void* shadow_pc, int shadow_dtype, float* hyper, int advance_step, void* stream) {
  GDM_REQUIRE(p && g_pc && m && v && shadow_pc && hyper, "gdm_adam_step_dev_pc: null pointer");
  GDM_REQUIRE(N > 0 && C > 0 && P > 0 && N <= 65535 && (C + 31) / 32 <= 65535 && gdm_dtype_ok(shadow_dtype),
              "gdm_adam_step_dev_pc: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (advance_step) hipLaunchKernelGGL(adam_prep_kernel, dim3(1), dim3(1), 0, s, hyper);
  const dim3 grid((P + 127) / 128, (C + 31) / 32, N);
  const int vec_ok = (P % 4 == 0) && ((((uintptr_t)p | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
  if (shadow_dtype == GDM_BF16)
    hipLaunchKernelGGL(adam_dev_pc_kernel<__bf16>, grid, dim3(256), 0, s, p, g_pc, m, v, C, P, (__bf16*)shadow_pc, hyper,
                       vec_ok);
  else
    hipLaunchKernelGGL(adam_dev_pc_kernel<float>, grid, dim3(256), 0, s, p, g_pc, m, v, C, P, (float*)shadow_pc, hyper,
                       vec_ok);
  GDM_LAUNCH_OK("gdm_adam_step_dev_pc");
  return GDM_OK;
}

extern "C" int gdm_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float* hyper, void* stream) {
  GDM_REQUIRE(p && g && m && v && hyper && n > 0, "gdm_adam_step_dev: bad arguments");
  const bool aligned = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 8192) blocks = 8192;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(adam_prep_kernel, dim3(1), dim3(1), 0, s, hyper);
  if (aligned) hipLaunchKernelGGL(adam_dev_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, p, g, m, v, n, hyper);
  else hipLaunchKernelGGL(adam_dev_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, p, g, m, v, n, hyper);
  GDM_LAUNCH_OK("gdm_adam_step_dev");
  return GDM_OK;
}

extern "C" size_t gdm_bn_workspace_bytes(int rows, int channels) {
  return (size_t)row_chunks(rows) * (size_t)channels * 3 * sizeof(float);
}

extern "C" int gdm_bn_act_fwd(const float* y, int rows, int channels, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                              float eps, int act, void* out, int out_dtype, float* save_mean, float* save_invstd,
                              int training, void* workspace, size_t workspace_bytes, void* stream) {
  GDM_REQUIRE(y && gamma && beta && out && save_mean && save_invstd, "gdm_bn_act_fwd: null pointer");
  GDM_REQUIRE(rows > 0 && channels > 0 && gdm_dtype_ok(out_dtype), "gdm_bn_act_fwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const int C = channels;
  if (training) {
    GDM_REQUIRE(rows > 1, "gdm_bn_act_fwd: training-mode batch norm needs more than 1 value per channel");
    const int chunks = row_chunks(rows);
    if (!workspace || workspace_bytes < gdm_bn_workspace_bytes(rows, C)) {
      gdm_set_error("gdm_bn_act_fwd: workspace too small");
      return GDM_EWORKSPACE;
    }
    const int chunk_rows = (rows + chunks - 1) / chunks;
    int cw = 64;
    while (cw / 2 >= C && cw > 1) cw /= 2;                      // lanes per row: power of two >= min(C, 64)
    hipLaunchKernelGGL(bn_partial_stats, dim3((C + 63) / 64, chunks), dim3(256), 0, s, y, rows, C, chunk_rows, cw,
                       (float*)workspace);
    hipLaunchKernelGGL(bn_finalize, dim3((C + 63) / 64), dim3(1024), 0, s, (const float*)workspace, chunks, rows, C,
                       momentum, eps, running_mean, running_var, num_batches_tracked, save_mean, save_invstd);
  } else {
    GDM_REQUIRE(running_mean && running_var, "gdm_bn_act_fwd: eval mode needs running statistics");
    hipLaunchKernelGGL(bn_eval_stats, dim3((C + 255) / 256), dim3(256), 0, s, running_mean, running_var, C, eps,
                       save_mean, save_invstd);
  }
  const int64_t total = (int64_t)rows * C;
  if (total < ((int64_t)1 << 31))
    hipLaunchKernelGGL(bn_apply<int>, dim3(grid_for(total)), dim3(256), 0, s, y, total, C, gamma, beta, save_mean,
                       save_invstd, act, out, out_dtype);
  else
    hipLaunchKernelGGL(bn_apply<int64_t>, dim3(grid_for(total)), dim3(256), 0, s, y, total, C, gamma, beta, save_mean,
                       save_invstd, act, out, out_dtype);
  GDM_LAUNCH_OK("gdm_bn_act_fwd");
  return GDM_OK;
}

// The two halves of training-mode batch norm as entry points of their own, for statistics that span more than one
// call: gdm_bn_partials leaves the per-row-chunk Welford triples (n, mean, M2) of y in `partials`
// (gdm_bn_partial_chunks(rows) x channels x 3 floats); partials of several batches (ranks) concatenated along the chunk
// axis go through gdm_bn_finalize (fixed-order merge -> mean, invstd, running statistics); gdm_bn_apply normalises.
extern "C" int gdm_bn_partial_chunks(int rows) { return row_chunks(rows); }

extern "C" int gdm_bn_partials(const float* y, int rows, int channels, float* partials, void* stream) {
  GDM_REQUIRE(y && partials && rows > 0 && channels > 0, "gdm_bn_partials: bad arguments");
  const int C = channels, chunks = row_chunks(rows), chunk_rows = (rows + chunks - 1) / chunks;
  int cw = 64;
  while (cw / 2 >= C && cw > 1) cw /= 2;
  hipLaunchKernelGGL(bn_partial_stats, dim3((C + 63) / 64, chunks), dim3(256), 0, (hipStream_t)stream, y, rows, C,
                     chunk_rows, cw, partials);
  GDM_LAUNCH_OK("gdm_bn_partials");
  return GDM_OK;
}

extern "C" int gdm_bn_apply(const float* y, int rows, int channels, const float* gamma, const float* beta,
                            const float* mean, const float* invstd, int act, void* out, int out_dtype, void* stream) {
  GDM_REQUIRE(y && gamma && beta && mean && invstd && out && rows > 0 && channels > 0 && gdm_dtype_ok(out_dtype),
              "gdm_bn_apply: bad arguments");
  const int64_t total = (int64_t)rows * channels;
  hipStream_t s = (hipStream_t)stream;
  if (total < ((int64_t)1 << 31))
    hipLaunchKernelGGL(bn_apply<int>, dim3(grid_for(total)), dim3(256), 0, s, y, total, channels, gamma, beta, mean, invstd,
                       act, out, out_dtype);
  else
    hipLaunchKernelGGL(bn_apply<int64_t>, dim3(grid_for(total)), dim3(256), 0, s, y, total, channels, gamma, beta, mean,
                       invstd, act, out, out_dtype);
  GDM_LAUNCH_OK("gdm_bn_apply");
  return GDM_OK;
}

extern "C" int gdm_bn_finalize(const float* ws, int chunks, int rows, int channels, float momentum, float eps,
                               float* running_mean, float* running_var, int64_t* num_batches_tracked, float* save_mean,
                               float* save_invstd, void* stream) {
  GDM_REQUIRE(ws && save_mean && save_invstd && chunks > 0 && rows > 1 && channels > 0, "gdm_bn_finalize: bad arguments");
  hipLaunchKernelGGL(bn_finalize, dim3((channels + 63) / 64), dim3(1024), 0, (hipStream_t)stream, ws, chunks, rows,
                     channels, momentum, eps, running_mean, running_var, num_batches_tracked, save_mean, save_invstd);
  GDM_LAUNCH_OK("gdm_bn_finalize");
  return GDM_OK;
}

extern "C" int gdm_bn_stats(const float* y, int rows, int channels, float* running_mean, float* running_var,
                            int64_t* num_batches_tracked, float momentum, float eps, float* save_mean,
                            float* save_invstd, void* workspace, size_t workspace_bytes, void* stream) {
  GDM_REQUIRE(y && save_mean && save_invstd && rows > 1 && channels > 0, "gdm_bn_stats: bad arguments");
  if (!workspace || workspace_bytes < gdm_bn_workspace_bytes(rows, channels)) {
    gdm_set_error("gdm_bn_stats: workspace too small");
    return GDM_EWORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  const int C = channels, chunks = row_chunks(rows), chunk_rows = (rows + chunks - 1) / chunks;
  int cw = 64;
  while (cw / 2 >= C && cw > 1) cw /= 2;
  hipLaunchKernelGGL(bn_partial_stats, dim3((C + 63) / 64, chunks), dim3(256), 0, s, y, rows, C, chunk_rows, cw,
                     (float*)workspace);
  hipLaunchKernelGGL(bn_finalize, dim3((C + 63) / 64), dim3(1024), 0, s, (const float*)workspace, chunks, rows, C,
                     momentum, eps, running_mean, running_var, num_batches_tracked, save_mean, save_invstd);
  GDM_LAUNCH_OK("gdm_bn_stats");
  return GDM_OK;
}

extern "C" int gdm_bn_act_bwd(const void* dout, const void* out, int out_dtype, const float* y, int rows, int channels,
                              const float* gamma, const float* save_mean, const float* save_invstd, int act,
                              float* dy, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                              void* stream) {
  GDM_REQUIRE(dout && out && y && gamma && save_mean && save_invstd && dy && dgamma && dbeta,
              "gdm_bn_act_bwd: null pointer");
  GDM_REQUIRE(rows > 0 && channels > 0 && gdm_dtype_ok(out_dtype), "gdm_bn_act_bwd: bad arguments");
  const int C = channels, chunks = row_chunks(rows);
  if (!workspace || workspace_bytes < gdm_bn_workspace_bytes(rows, C)) {
    gdm_set_error("gdm_bn_act_bwd: workspace too small");
    return GDM_EWORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  const int chunk_rows = (rows + chunks - 1) / chunks;
  hipLaunchKernelGGL(bn_bwd_partial, dim3((C + 63) / 64, chunks), dim3(256), 0, s, dout, out, out_dtype, y, rows, C,
                     chunk_rows, save_mean, save_invstd, act, (float*)workspace);
  hipLaunchKernelGGL(bn_bwd_finalize, dim3((C + 255) / 256), dim3(256), 0, s, (const float*)workspace, chunks, C,
                     dgamma, dbeta);
  const int64_t total = (int64_t)rows * C;
  hipLaunchKernelGGL(bn_bwd_apply, dim3(grid_for(total)), dim3(256), 0, s, dout, out, out_dtype, y, total, rows, C,
                     gamma, save_mean, save_invstd, dgamma, dbeta, act, dy);
  GDM_LAUNCH_OK("gdm_bn_act_bwd");
  return GDM_OK;
}

extern "C" int gdm_bias_act_fwd(const float* x, const float* bias, int rows, int cols, int act, float slope, void* out,
                                int out_dtype, void* stream) {
  GDM_REQUIRE(x && out && rows > 0 && cols > 0 && gdm_dtype_ok(out_dtype), "gdm_bias_act_fwd: bad arguments");
  const int64_t total = (int64_t)rows * cols;
  hipLaunchKernelGGL(bias_act_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, bias, total, cols,
                     act, slope, out, out_dtype);
  GDM_LAUNCH_OK("gdm_bias_act_fwd");
  return GDM_OK;
}

extern "C" int gdm_act_bwd(const void* dout, const void* out, int dtype, int64_t n, int act, float slope, void* dx,
                           void* stream) {
  GDM_REQUIRE(dout && out && dx && n > 0 && gdm_dtype_ok(dtype), "gdm_act_bwd: bad arguments");
  hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dout, out, dtype, n, act,
                     slope, dx);
  GDM_LAUNCH_OK("gdm_act_bwd");
  return GDM_OK;
}

extern "C" int gdm_colsum(const void* x, int dtype, int rows, int cols, float* out, void* workspace,
                          size_t workspace_bytes, void* stream) {
  GDM_REQUIRE(x && out && rows > 0 && cols > 0 && gdm_dtype_ok(dtype), "gdm_colsum: bad arguments");
  const int chunks = row_chunks(rows);
  if (!workspace || workspace_bytes < (size_t)chunks * cols * sizeof(float)) {
    gdm_set_error("gdm_colsum: workspace too small (need %zu)", (size_t)chunks * cols * sizeof(float));
    return GDM_EWORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  const int chunk_rows = (rows + chunks - 1) / chunks;
  hipLaunchKernelGGL(colsum_partial, dim3((cols + 63) / 64, chunks), dim3(256), 0, s, x, dtype, rows, cols, chunk_rows,
                     (float*)workspace);
  hipLaunchKernelGGL(colsum_final, dim3((cols + 255) / 256), dim3(256), 0, s, (const float*)workspace, chunks, cols,
                     out);
  GDM_LAUNCH_OK("gdm_colsum");
  return GDM_OK;
}

extern "C" int gdm_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream) {
  GDM_REQUIRE(src && dst && n > 0 && gdm_dtype_ok(src_dtype) && gdm_dtype_ok(dst_dtype), "gdm_cast: bad arguments");
  hipLaunchKernelGGL(cast_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, src, src_dtype, dst,
                     dst_dtype, n);
  GDM_LAUNCH_OK("gdm_cast");
  return GDM_OK;
}

// =====================================================================================================================
// Model 1 discriminator head, forward + loss + backward in ONE single-workgroup launch (GAN_DES/SIMNN.py:140-141 and
// the BCE terms at 289/311/329).  Replaces the N = 1 / M = 1 GEMMs (fc2 forward, dW_fc2, dh1), two BCE launches,
// the ReLU backward and two bias column sums:
//   z[b] = h1[b,:] . w2 + b2;  p[b] = sigmoid(z[b]);  loss = sum over the two label halves of mean BCEWithLogits(p, y)
//   dz[b] = (sigmoid(p[b]) - y) / n_half * p(1-p)          (the reference feeds the sigmoid OUTPUT to a logits loss)
//   dh1[b,j] = dz[b] * w2[j] * (h1[b,j] > 0);  dw2[j] = sum_b dz[b] h1[b,j];  db2 = sum_b dz[b];  db1[j] = sum_b dh1[b,j]
// rows [0, n0) carry label y0, rows [n0, n) label y1 (n0 == n: a single label).  All reductions in fixed order.
// =====================================================================================================================
namespace {
constexpr int HEAD_ROWS = 8;    // batch rows per workgroup (a launch is latency-bound: many short workgroups)
// partial layout per workgroup: [0,128) dw2, [128,256) db1, [256] db2, [257] loss
__global__ __launch_bounds__(256) void simnn_head_kernel(const float* __restrict__ h1, const float* __restrict__ w2,
                                                         const float* __restrict__ b2, int n, int n0, float y0,
                                                         float y1, float* __restrict__ prob, void* __restrict__ dh1,
                                                         int dh1_bf16, float* __restrict__ partials) {
  __shared__ float dz_s[HEAD_ROWS], lt_s[HEAD_ROWS], part[2][256];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int r0 = blockIdx.x * HEAD_ROWS, r1 = min(n, r0 + HEAD_ROWS);
  const float wa = w2[lane], wb = w2[lane + 64], bias = b2[0];
  float hv[HEAD_ROWS / 4][2];
#pragma unroll
  for (int k = 0; k < HEAD_ROWS / 4; ++k) {       // all loads first
    const int b = r0 + wv + 4 * k;
    hv[k][0] = b < r1 ? h1[(int64_t)b * 128 + lane] : 0.f;
    hv[k][1] = b < r1 ? h1[(int64_t)b * 128 + 64 + lane] : 0.f;
  }
#pragma unroll
  for (int k = 0; k < HEAD_ROWS / 4; ++k) {
    const int b = r0 + wv + 4 * k, bl = wv + 4 * k;
    const float s = wave_sum(hv[k][0] * wa + hv[k][1] * wb);
    if (lane == 0) {
      float lt = 0.f, dz = 0.f;
      if (b < r1) {
        const float z = s + bias;
        const float p = 1.0f / (1.0f + expf(-z));
        const bool first = b < n0;
        const float y = first ? y0 : y1;
        const float cnt = (float)(first ? n0 : n - n0);
        prob[b] = p;
        lt = (fmaxf(p, 0.f) - p * y + log1pf(expf(-fabsf(p)))) / cnt;
        dz = (1.0f / (1.0f + expf(-p)) - y) / cnt * p * (1.f - p);
      }
      lt_s[bl] = lt;
      dz_s[bl] = dz;
    }
  }
  __syncthreads();
  float* po = partials + (int64_t)blockIdx.x * 258;
  if (t == 0) {
    float l = 0.f, sdz = 0.f;
#pragma unroll
    for (int b = 0; b < HEAD_ROWS; ++b) { l += lt_s[b]; sdz += dz_s[b]; }
    po[256] = sdz;
    po[257] = l;
  }
  // backward for this row slice: thread (grp = t / 128, j = t % 128) walks rows grp, grp+2, ...
  const int j = t & 127, grp = t >> 7;
  const float wj = w2[j];
  float s_w = 0.f, s_b1 = 0.f;
  float hh[HEAD_ROWS / 2];
#pragma unroll
  for (int k = 0; k < HEAD_ROWS / 2; ++k) {
    const int b = r0 + grp + 2 * k;
    hh[k] = b < r1 ? h1[(int64_t)b * 128 + j] : 0.f;
  }
#pragma unroll
  for (int k = 0; k < HEAD_ROWS / 2; ++k) {
    const int b = r0 + grp + 2 * k;
    const float d = dz_s[grp + 2 * k];
    const float g = hh[k] > 0.f ? d * wj : 0.f;
    if (dh1 != nullptr && b < r1) {
      if (dh1_bf16) ((__bf16*)dh1)[(int64_t)b * 128 + j] = (__bf16)g;      // the two fc1 GEMMs' operand in bf16 mode
      else ((float*)dh1)[(int64_t)b * 128 + j] = g;
    }
    s_w = fmaf(d, hh[k], s_w);
    s_b1 += g;
  }
  part[0][t] = s_w;
  part[1][t] = s_b1;
  __syncthreads();
  if (t < 128) {
    po[t] = part[0][t] + part[0][128 + t];
    po[128 + t] = part[1][t] + part[1][128 + t];
  }
}

__global__ __launch_bounds__(320) void simnn_head_final(const float* __restrict__ partials, int groups,
                                                        float* __restrict__ loss, int accumulate_loss,
                                                        float* __restrict__ dw2, float* __restrict__ db2,
                                                        float* __restrict__ db1) {
  const int t = threadIdx.x;
  if (t >= 258) return;
  // eight independent partial sums (fixed assignment and order): the loads of a trip issue back to back instead of
  // one dependent load per group
  float p8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int g = 0;
  for (; g + 8 <= groups; g += 8) {
#pragma unroll
    for (int q = 0; q < 8; ++q) p8[q] += partials[(int64_t)(g + q) * 258 + t];
  }
  for (; g < groups; ++g) p8[0] += partials[(int64_t)g * 258 + t];      // (no run-time register index)
  const float s = ((p8[0] + p8[1]) + (p8[2] + p8[3])) + ((p8[4] + p8[5]) + (p8[6] + p8[7]));
  if (t == 257) loss[0] = (accumulate_loss ? loss[0] : 0.f) + s;
  else if (dw2 != nullptr) {
    if (t < 128) dw2[t] = s;
    else if (t < 256) db1[t - 128] = s;
    else db2[0] = s;
  }
}
}  // namespace

extern "C" size_t gdm_simnn_head_workspace_bytes(int n) {
  return (size_t)((n + HEAD_ROWS - 1) / HEAD_ROWS) * 258 * sizeof(float);
}

extern "C" int gdm_simnn_head(const float* h1, const float* w2, const float* b2, int n, int n0, float y0, float y1,
                              float* prob, float* loss, int accumulate_loss, void* dh1, int dh1_dtype, float* dw2,
                              float* db2, float* db1, void* workspace, size_t workspace_bytes, void* stream) {
  GDM_REQUIRE(h1 && w2 && b2 && prob && loss, "gdm_simnn_head: null pointer");
  GDM_REQUIRE(n > 0 && n <= 65536 && n0 > 0 && n0 <= n, "gdm_simnn_head: bad batch n=%d n0=%d", n, n0);
  GDM_REQUIRE(dh1 == nullptr || (dw2 && db2 && db1), "gdm_simnn_head: gradient outputs missing");
  GDM_REQUIRE(gdm_dtype_ok(dh1_dtype), "gdm_simnn_head: bad dh1 dtype");
  if (!workspace || workspace_bytes < gdm_simnn_head_workspace_bytes(n)) {
    gdm_set_error("gdm_simnn_head: workspace too small");
    return GDM_EWORKSPACE;
  }
  const int groups = (n + HEAD_ROWS - 1) / HEAD_ROWS;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(simnn_head_kernel, dim3(groups), dim3(256), 0, s, h1, w2, b2, n, n0, y0, y1, prob, dh1,
                     dh1_dtype == GDM_BF16 ? 1 : 0, (float*)workspace);
  hipLaunchKernelGGL(simnn_head_final, dim3(1), dim3(320), 0, s, (const float*)workspace, groups, loss,
                     accumulate_loss, dh1 ? dw2 : nullptr, db2, db1);
  GDM_LAUNCH_OK("gdm_simnn_head");
  return GDM_OK;
}

// ---- concatenation along dim 1 for a few small jobs in one launch (the generators' inputs of model 2) -----------------
namespace {
struct ConcatJobs {
  gdm_concat_job j[4];
  int n;
};
__global__ __launch_bounds__(256) void concat_cols_kernel(ConcatJobs js) {
  const gdm_concat_job& q = js.j[blockIdx.y];
  const int K = q.Ka + q.Kb, total = q.M * K;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int m = i / K, k = i - m * K;
    q.out[i] = k < q.Ka ? q.a[(size_t)m * q.Ka + k] : q.b[(size_t)m * q.Kb + (k - q.Ka)];
  }
}
}  // namespace

extern "C" int gdm_concat_cols_multi(const gdm_concat_job* jobs, int n_jobs, void* stream) {
  GDM_REQUIRE(jobs && n_jobs >= 1 && n_jobs <= 4, "gdm_concat_cols_multi: 1..4 jobs per launch");
  ConcatJobs js{};
  js.n = n_jobs;
  int most = 0;
  for (int i = 0; i < n_jobs; ++i) {
    const gdm_concat_job& q = jobs[i];
    GDM_REQUIRE(q.a && q.out && q.M >= 1 && q.Ka >= 1 && q.Kb >= 0 && (q.b || q.Kb == 0), "gdm_concat_cols_multi: bad job %d", i);
    js.j[i] = q;
    most = q.M * (q.Ka + q.Kb) > most ? q.M * (q.Ka + q.Kb) : most;
  }
  const int bx = (most + 255) / 256 < 64 ? (most + 255) / 256 : 64;
  hipLaunchKernelGGL(concat_cols_kernel, dim3(bx, n_jobs), dim3(256), 0, (hipStream_t)stream, js);
  GDM_LAUNCH_OK("gdm_concat_cols_multi");
  return GDM_OK;
}

// ---- non-finite detection (anomaly mode / check_finite) ----------------------------------------------------------------
// The library is compiled with -fno-honor-nans (a NaN has no defined effect on the arithmetic: the issue-bound conv
// epilogues save a canonicalisation per fmaxf), so "did a NaN or Inf enter or leave the step" is answered by looking at
// the BITS of the tensors: exponent all ones.  counter[0] += number of non-finite elements (integer atomics: exact).
namespace {
__global__ __launch_bounds__(256) void nonfinite_count_kernel(const void* __restrict__ x, int dtype, int64_t n,
                                                              int* __restrict__ counter) {
  int c = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const uint32_t bits = dtype == GDM_BF16 ? (uint32_t)((const uint16_t*)x)[i] << 16 : ((const uint32_t*)x)[i];
    c += (bits & 0x7f800000u) == 0x7f800000u ? 1 : 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0 && c != 0) atomicAdd(counter, c);
}
}  // namespace

extern "C" int gdm_nonfinite_count(const void* x, int dtype, int64_t n, int* counter, void* stream) {
  GDM_REQUIRE(x && counter && n >= 0 && gdm_dtype_ok(dtype), "gdm_nonfinite_count: bad arguments");
  if (n == 0) return GDM_OK;
  int64_t blocks = (n + 256 * 8 - 1) / (256 * 8);
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  hipLaunchKernelGGL(nonfinite_count_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, dtype, n, counter);
  GDM_LAUNCH_OK("gdm_nonfinite_count");
  return GDM_OK;
}
