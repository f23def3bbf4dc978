// One generator block of model 2 in one launch: Linear -> BatchNorm1d (train or eval) -> activation
// (MMGAN_MIDI_DES/network_tests.py:75-80, 110-115), for batches of up to 256 rows.
//
// A 256-thread workgroup owns ALL M rows of a 32-column strip of the output, so the batch statistics of its columns
// never leave the workgroup: y = x W^T + b is accumulated on v_mfma_f32_16x16x32_bf16 (wave w: rows [64w, 64w+64),
// 4 x 2 accumulator tiles) with SPLIT operands -- x = xh + xl, W = wh + wl in bf16, y += xh wh + xh wl + xl wh (the
// dropped xl wl term is 2^-16 relative) -- because these layers are latency-bound (<= 13 MFLOP per launch), so three
// MFMAs per tile cost nothing, while plain bf16 operands put 5e-2 of error on the generated DES parameters (beat times
// up to ~27 s enter un-normalised, network_tests.py:119): the generated matrices are the product of this path.  The column mean is reduced
// registers -> lanes -> waves (LDS), the variance is a second pass over the SAME registers (exact two-pass, no E[x^2]-E[x]^2 cancellation), running statistics are updated by the
// owning workgroup, and the normalised + activated strip is stored.  Replaces GEMM + 3 batch-norm launches per layer.
#include "gdm_common.h"

namespace {

constexpr int LB_M = 256, LB_N = 32, LB_KT = 32, LB_LD = LB_KT + 8;

// GP = groups in flight per workgroup (1 or 2): with GP = 2 the workgroup has 512 threads, waves 0-3 run group g and waves
// 4-7 group g+1 through the same barriers (the launch is a latency chain -- staging, MFMA, two reduction rounds, stores --
// so two groups one after the other cost twice the launch, side by side barely more than one); the running statistics
// still take the groups' updates in order, applied by the first group's threads once both batches' statistics are in LDS.
// Up to two independent blocks ("jobs": different inputs, weights and shapes) share one launch -- the workgroups
// [0, tiles0) belong to job 0, the rest to job 1: model 2's two generators have the same depth, so their k-th blocks
// go out together and an iteration's generator work is 4 launches instead of 8.
struct LbJobs {
  gdm_linear_bn_job j[2];
  int tiles0;
};

template <bool VEC, int GP>
__global__ __launch_bounds__(256 * GP) void linear_bn_act_kernel(LbJobs jobs, float momentum, float eps, int act,
                                                                 int training) {
  const int job = (int)blockIdx.x >= jobs.tiles0 ? 1 : 0;
  const gdm_linear_bn_job& jb = jobs.j[job];
  const int tile = (int)blockIdx.x - (job ? jobs.tiles0 : 0);
  const float* x = jb.x;
  const float* __restrict__ w = jb.w;
  const float* __restrict__ bias = jb.bias;
  const float* __restrict__ gamma = jb.gamma;
  const float* __restrict__ beta = jb.beta;
  float* __restrict__ running_mean = jb.running_mean;
  float* __restrict__ running_var = jb.running_var;
  int64_t* __restrict__ nbt = jb.num_batches_tracked;
  const int M = jb.M, N = jb.N, K = jb.K, groups = jb.groups, stat_repeats = jb.stat_repeats;
  float* y_out = jb.y_out;
  float* out = jb.out;
  float* __restrict__ save_mean = jb.save_mean;
  float* __restrict__ save_invstd = jb.save_invstd;
  __shared__ __attribute__((aligned(16))) __bf16 As_[GP][2][LB_M * LB_LD];     // [group slot][hi | lo]
  __shared__ __attribute__((aligned(16))) __bf16 Bs_[GP][2][LB_N * LB_LD];
  __shared__ float colred_[GP][4][LB_N];
  __shared__ float colstat_[GP][3][LB_N];                                       // mean | invstd | sum of squared deviations
  const int gsel = threadIdx.x >> 8, t = threadIdx.x & 255, l = t & 63, wv = t >> 6, lr = l & 15, lg = l >> 4;
  auto& As = As_[gsel];
  auto& Bs = Bs_[gsel];
  auto& colred = colred_[gsel];
  auto& colstat = colstat_[gsel];
  const int n0 = tile * LB_N;
  if (tile == 0 && threadIdx.x == 0 && training && nbt) nbt[0] += (int64_t)groups * stat_repeats;
  // `groups` independent batches of M rows share the weights and are normalised with their OWN batch statistics, one
  // after the other (a generator's two forwards of one training iteration, network_tests.py:294 and 312, in one launch:
  // the running statistics take the updates in that order); `stat_repeats` applies a group's running-statistics update
  // that many times (the second forward of the beat generator sees exactly the first one's inputs).
  const float* const x_all = x;
  float* const out_all = out;
  float* const y_all = y_out;
  for (int grp0 = 0; grp0 < groups; grp0 += GP) {
  const bool live = grp0 + gsel < groups;          // an odd tail: the second slot repeats the last group and stores nothing
  const int grp = live ? grp0 + gsel : groups - 1;
  x = x_all + (int64_t)grp * M * K;
  out = out_all + (int64_t)grp * M * N;
  y_out = y_all ? y_all + (int64_t)grp * M * N : nullptr;
  float* const save_mean_g = save_mean + (int64_t)grp * N;
  float* const save_invstd_g = save_invstd + (int64_t)grp * N;
  __syncthreads();                          // the previous group's LDS reads are done

  f32x4 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Staging: 16-byte loads (K % 4 == 0 and 16-byte aligned rows, else a scalar path), all loads of a k-tile in flight
  // together and one k-tile ahead of the MFMAs (register prefetch): these layers are latency-bound, not bandwidth-bound.
  // VEC (chosen on the host): K % 4 == 0 and 16-byte aligned operands -> every chunk is inside or outside the row as a whole
  f32x4 ra[8], rb;
  // Small batches (the reference trains model 2 with 16 rows): staging pass i covers rows [32 i, 32 i + 32) and
  // accumulator tile (wave, i) rows [64 wv + 16 i, + 16) -- passes and tiles that hold no row of the batch are skipped
  // (wave-uniform tests), so a k-tile of a 16-row batch costs one staging pass and 6 MFMAs in one wave instead of eight
  // passes and 24 MFMAs in each of the four.
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (32 * i >= M) continue;
      const int idx = t + 256 * i, m = idx >> 3, k = k0 + (idx & 7) * 4;
      ra[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (m < M) {
        const float* p = x + (int64_t)m * K + k;
        if constexpr (VEC) {
          if (k < K) ra[i] = *(const f32x4*)p;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) ra[i][e] = (k + e < K) ? p[e] : 0.f;
        }
      }
    }
    {
      const int n = t >> 3, k = k0 + (t & 7) * 4;
      rb = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (n0 + n < N) {
        const float* p = w + (int64_t)(n0 + n) * K + k;
        if constexpr (VEC) {
          if (k < K) rb = *(const f32x4*)p;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) rb[e] = (k + e < K) ? p[e] : 0.f;
        }
      }
    }
  };
  load_tile(0);
  for (int k0 = 0; k0 < K; k0 += LB_KT) {
    __syncthreads();                       // the previous tile's fragment reads are done
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (32 * i >= M) continue;
      const int idx = t + 256 * i;
      bf16x4 h, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) { h[e] = (__bf16)ra[i][e]; lo[e] = (__bf16)(ra[i][e] - (float)h[e]); }
      *(bf16x4*)&As[0][(idx >> 3) * LB_LD + (idx & 7) * 4] = h;
      *(bf16x4*)&As[1][(idx >> 3) * LB_LD + (idx & 7) * 4] = lo;
    }
    {
      bf16x4 h, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) { h[e] = (__bf16)rb[e]; lo[e] = (__bf16)(rb[e] - (float)h[e]); }
      *(bf16x4*)&Bs[0][(t >> 3) * LB_LD + (t & 7) * 4] = h;
      *(bf16x4*)&Bs[1][(t >> 3) * LB_LD + (t & 7) * 4] = lo;
    }
    __syncthreads();
    if (k0 + LB_KT < K) load_tile(k0 + LB_KT);
    bf16x8 b[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h) b[h][j] = *(const bf16x8*)&Bs[h][(16 * j + lr) * LB_LD + 8 * lg];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (64 * wv + 16 * i >= M) continue;           // no row of the batch in this tile: its accumulators stay zero
      const bf16x8 ah = *(const bf16x8*)&As[0][(64 * wv + 16 * i + lr) * LB_LD + 8 * lg];
      const bf16x8 al = *(const bf16x8*)&As[1][(64 * wv + 16 * i + lr) * LB_LD + 8 * lg];
#pragma unroll
      for (int j = 0; j < 2; ++j) {      // small terms first
        acc[i][j] = mfma16(al, b[0][j], acc[i][j]);
        acc[i][j] = mfma16(ah, b[1][j], acc[i][j]);
        acc[i][j] = mfma16(ah, b[0][j], acc[i][j]);
      }
    }
  }
  // C layout: col = n (16j + lr), row = m (64wv + 16i + 4lg + r).  Add the Linear bias.
  float bn_[2], mean[2], invstd[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + 16 * j + lr;
    bn_[j] = (bias && n < N) ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] += bn_[j];
  }
  if (training) {
    // ---- pass 1: column means over the M valid rows
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += (64 * wv + 16 * i + 4 * lg + r < M) ? acc[i][j][r] : 0.f;
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      if (lg == 0) colred[wv][16 * j + lr] = s;
    }
    __syncthreads();
    if (t < LB_N) colstat[0][t] = (((colred[0][t] + colred[1][t]) + colred[2][t]) + colred[3][t]) / (float)M;
    __syncthreads();
    // ---- pass 2: centred sum of squares from the same registers
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      mean[j] = colstat[0][16 * j + lr];
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float dlt = acc[i][j][r] - mean[j];
          s += (64 * wv + 16 * i + 4 * lg + r < M) ? dlt * dlt : 0.f;
        }
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      if (lg == 0) colred[wv][16 * j + lr] = s;
    }
    __syncthreads();
    if (t < LB_N) {
      const float m2 = ((colred[0][t] + colred[1][t]) + colred[2][t]) + colred[3][t];
      const float var_b = m2 / (float)M;
      colstat[1][t] = 1.0f / sqrtf(var_b + eps);
      colstat[2][t] = m2;
      const int n = n0 + t;
      if (n < N && live) {
        save_mean_g[n] = colstat[0][t];
        save_invstd_g[n] = colstat[1][t];
      }
    }
    __syncthreads();
    if (gsel == 0 && t < LB_N && n0 + t < N && running_mean) {
      const int n = n0 + t;
      float rm = running_mean[n], rv = running_var[n];
      for (int g = 0; g < GP && grp0 + g < groups; ++g)
        for (int rep = 0; rep < stat_repeats; ++rep) {
          rm = (1.f - momentum) * rm + momentum * colstat_[g][0][t];
          rv = (1.f - momentum) * rv + momentum * (colstat_[g][2][t] / (float)max(M - 1, 1));
        }
      running_mean[n] = rm;
      running_var[n] = rv;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) invstd[j] = colstat[1][16 * j + lr];
  } else {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + 16 * j + lr;
      mean[j] = n < N ? running_mean[n] : 0.f;
      invstd[j] = n < N ? 1.0f / sqrtf(running_var[n] + eps) : 0.f;
      if (n < N && lg == 0 && wv == 0 && live) { save_mean_g[n] = mean[j]; save_invstd_g[n] = invstd[j]; }
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + 16 * j + lr;
    if (n >= N || !live) continue;
    const float alpha = invstd[j] * gamma[n], bt = beta[n];
    if (M == LB_M && act == GDM_ACT_SIGMOID && !y_out) {
      // the generators' case (full batch, sigmoid, forward only): no per-element branch, so the 16 stores of a column
      // issue back to back (with a branch around every store the compiler waited for each one: ~12 us per launch)
      float o[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[i][r] = apply_act((acc[i][j][r] - mean[j]) * alpha + bt, GDM_ACT_SIGMOID, 0.f);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(int64_t)(64 * wv + 16 * i + 4 * lg + r) * N + n] = o[i][r];
      continue;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 64 * wv + 16 * i + 4 * lg + r;
        if (m < M) {
          const float yv = acc[i][j][r];
          if (y_out) y_out[(int64_t)m * N + n] = yv;
          out[(int64_t)m * N + n] = apply_act((yv - mean[j]) * alpha + bt, act, 0.f);
        }
      }
  }
  }   // groups
}

}  // namespace

extern "C" int gdm_linear_bn_act_max_rows(void) { return LB_M; }

namespace {
int launch_jobs(const gdm_linear_bn_job* jobs, int n_jobs, float momentum, float eps, int act, int training,
                hipStream_t s, const char* who) {
  LbJobs lj{};
  bool vec = true;
  int gp = 1, tiles = 0;
  for (int i = 0; i < n_jobs; ++i) {
    const gdm_linear_bn_job& j = jobs[i];
    GDM_REQUIRE(j.x && j.w && j.gamma && j.beta && j.out && j.save_mean && j.save_invstd, "%s: null pointer", who);
    GDM_REQUIRE(j.groups >= 1 && j.stat_repeats >= 1, "%s: groups and stat_repeats must be >= 1", who);
    GDM_REQUIRE(j.M >= 1 && j.M <= LB_M && j.N >= 1 && j.K >= 1, "%s: M=%d outside 1..%d (or bad N/K)", who, j.M, LB_M);
    GDM_REQUIRE(!training || j.M > 1, "%s: training-mode batch norm needs more than 1 row", who);
    GDM_REQUIRE(training || (j.running_mean && j.running_var), "%s: eval mode needs running statistics", who);
    vec = vec && (j.K % 4 == 0) && ((((uintptr_t)j.x | (uintptr_t)j.w) & 15) == 0);
    if (j.groups >= 2) gp = 2;
    lj.j[i] = j;
    if (i == 0) lj.tiles0 = (j.N + LB_N - 1) / LB_N;
    tiles += (j.N + LB_N - 1) / LB_N;
  }
  const dim3 grid(tiles);
#define GDM_LB_LAUNCH(VEC_, GP_)                                                                                          \
  hipLaunchKernelGGL((linear_bn_act_kernel<VEC_, GP_>), grid, dim3(256 * GP_), 0, s, lj, momentum, eps, act, training)
  if (gp == 2) { if (vec) GDM_LB_LAUNCH(true, 2); else GDM_LB_LAUNCH(false, 2); }
  else { if (vec) GDM_LB_LAUNCH(true, 1); else GDM_LB_LAUNCH(false, 1); }
#undef GDM_LB_LAUNCH
  GDM_LAUNCH_OK(who);
  return GDM_OK;
}
}  // namespace

extern "C" int gdm_linear_bn_act_fwd(const float* x, const float* w, const float* bias, const float* gamma,
                                     const float* beta, float* running_mean, float* running_var,
                                     int64_t* num_batches_tracked, float momentum, float eps, int act, int training,
                                     int M, int N, int K, float* y_out, float* out, float* save_mean,
                                     float* save_invstd, int groups, int stat_repeats, void* stream) {
  const gdm_linear_bn_job j{x, w, bias, gamma, beta, running_mean, running_var, num_batches_tracked, y_out, out,
                            save_mean, save_invstd, M, N, K, groups, stat_repeats};
  return launch_jobs(&j, 1, momentum, eps, act, training, (hipStream_t)stream, "gdm_linear_bn_act_fwd");
}

extern "C" int gdm_linear_bn_act_fwd_multi(const gdm_linear_bn_job* jobs, int n_jobs, float momentum, float eps, int act,
                                           int training, void* stream) {
  GDM_REQUIRE(jobs && n_jobs >= 1 && n_jobs <= 2, "gdm_linear_bn_act_fwd_multi: 1 or 2 jobs per launch");
  return launch_jobs(jobs, n_jobs, momentum, eps, act, training, (hipStream_t)stream, "gdm_linear_bn_act_fwd_multi");
}
