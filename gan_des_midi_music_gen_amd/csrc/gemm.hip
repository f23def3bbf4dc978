// gdm_gemm: strided MFMA GEMM with fused bias/activation epilogue and deterministic split-K.
// Replaces aten::addmm / aten::mm under nn.Linear (GAN_DES/SIMNN.py:126-127,140-141; MMGAN network_tests.py:77,112,
// 139,154,160) and the first ConvTranspose2d of model 1's generator (SIMNN.py:70, a [B,100]x[100,2048] product).
//
// Tiling: 64x64 output tile per 256-thread workgroup (4 waves as 2x2, each wave 32x32 = 2x2 MFMA 16x16 tiles).
//   fp32 compute: v_mfma_f32_16x16x4_f32, K tile 32 (8 MFMA k-steps), LDS rows padded to 34 floats
//                 (row stride == 2 mod 32 banks -> ds_read_b32 of 16 rows x 2 k is conflict-free)
//   bf16 compute: v_mfma_f32_16x16x32_bf16, K tile 64 (2 k-steps), LDS rows padded to 72 bf16 (144 B, 16-B aligned
//                 so each lane's 8-element fragment is one ds_read_b128)
// Both operand tiles are stored k-contiguous ([m][k], [n][k]) whatever the global strides are; the global->LDS map is
// chosen per operand so that consecutive lanes touch consecutive addresses (k-fastest when the k stride is 1,
// m/n-fastest otherwise).
#include "gemm_common.h"

namespace {

constexpr int BM = 64, BN = 64, NT = 256;

template <typename CT> struct GemmCfg;
template <> struct GemmCfg<float> { static constexpr int KT = 32, LD = 34; };
template <> struct GemmCfg<__bf16> { static constexpr int KT = 64, LD = 72; };

template <typename CT, typename TA, typename TB>
__global__ __launch_bounds__(NT) void gemm_kernel(GemmArgs g) {
  constexpr int KT = GemmCfg<CT>::KT, LD = GemmCfg<CT>::LD;
  __shared__ __attribute__((aligned(16))) CT smem[(BM + BN) * LD];
  CT* As = smem;
  CT* Bs = smem + BM * LD;
  const int t = threadIdx.x, l = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;
  const int lr = l & 15, lg = l >> 4;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int kbeg = blockIdx.z * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);
  const TA* __restrict__ A = (const TA*)g.A;
  const TB* __restrict__ B = (const TB*)g.B;
  const bool a_kfast = (g.sak == 1), b_kfast = (g.sbk == 1);
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Register prefetch: the loads of k-tile i+1 are issued right after tile i has been handed to LDS, so they fly during
  // tile i's MFMAs (these launches run one or two workgroups per CU: without it every k-tile paid a full HBM round trip
  // in front of 32 MFMAs -- exact-fp32 fc1 products ran at a third of what the matrix pipe allows).
  constexpr int PER = BM * KT / NT;
  float va[PER], vb[PER];
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int idx = t + NT * i;
      const int m = a_kfast ? idx / KT : idx % BM;
      const int k = a_kfast ? idx % KT : idx / BM;
      const int gm = m0 + m, gk = k0 + k;
      va[i] = (gm < g.M && gk < kend) ? to_f32(A[(int64_t)gm * g.sam + (int64_t)gk * g.sak]) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int idx = t + NT * i;
      const int n = b_kfast ? idx / KT : idx % BN;
      const int k = b_kfast ? idx % KT : idx / BN;
      const int gn = n0 + n, gk = k0 + k;
      vb[i] = (gn < g.N && gk < kend) ? to_f32(B[(int64_t)gk * g.sbk + (int64_t)gn * g.sbn]) : 0.f;
    }
  };
  if (kbeg < kend) load_tile(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += KT) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int idx = t + NT * i;
      const int m = a_kfast ? idx / KT : idx % BM;
      const int k = a_kfast ? idx % KT : idx / BM;
      As[m * LD + k] = from_f32<CT>(va[i]);
      const int n = b_kfast ? idx / KT : idx % BN;
      const int kb = b_kfast ? idx % KT : idx / BN;
      Bs[n * LD + kb] = from_f32<CT>(vb[i]);
    }
    __syncthreads();
    if (k0 + KT < kend) load_tile(k0 + KT);
    if constexpr (sizeof(CT) == 4) {
#pragma unroll
      for (int ks = 0; ks < KT / 4; ++ks) {
        float a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = As[(wm * 32 + i * 16 + lr) * LD + ks * 4 + lg];
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = Bs[(wn * 32 + j * 16 + lr) * LD + ks * 4 + lg];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < KT / 32; ++ks) {
        bf16x8 a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = *(const bf16x8*)&As[(wm * 32 + i * 16 + lr) * LD + ks * 32 + lg * 8];
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = *(const bf16x8*)&Bs[(wn * 32 + j * 16 + lr) * LD + ks * 32 + lg * 8];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
      }
    }
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 32 + i * 16 + lg * 4 + r;
        const int n = n0 + wn * 32 + j * 16 + lr;
        if (m < g.M && n < g.N) {
          if (g.split_k > 1) g.ws[((int64_t)blockIdx.z * g.M + m) * g.N + n] = acc[i][j][r];
          else gemm_epilogue_store(g, m, n, acc[i][j][r]);
        }
      }
}

// Sums the split-K slabs in a fixed order, then applies the epilogue.  The kernel is a chain of memory latencies over
// few outputs, so it is spread wide: VEC (N % 4 == 0, row-major C): a workgroup owns 64 output vectors (4 consecutive n
// each); its four waves each sum one contiguous quarter of the slabs (4 interleaved partial sums so that the loads
// pipeline) and wave 0 adds the four quarter sums in order.
template <bool VEC>
__global__ __launch_bounds__(256) void gemm_splitk_reduce(GemmArgs g) {
  const int64_t total = (int64_t)g.M * g.N;
  if constexpr (VEC) {
    __shared__ f32x4 part[4][64];
    const int lane = threadIdx.x & 63, zg = threadIdx.x >> 6;
    const int64_t idx = ((int64_t)blockIdx.x * 64 + lane) * 4;
    const bool live = idx < total;
    const int q4 = (g.split_k + 3) / 4;
    const int z0 = zg * q4, z1 = min(g.split_k, z0 + q4);
    f32x4 s[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) s[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (live) {
      int z = z0;
      for (; z + 3 < z1; z += 4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) s[q] += *(const f32x4*)(g.ws + (int64_t)(z + q) * total + idx);
      }
      for (; z < z1; ++z) s[0] += *(const f32x4*)(g.ws + (int64_t)z * total + idx);
    }
    part[zg][lane] = (s[0] + s[1]) + (s[2] + s[3]);
    __syncthreads();
    if (zg == 0 && live) {
      const f32x4 v = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
      const int m = (int)(idx / g.N), n = (int)(idx % g.N);
#pragma unroll
      for (int r = 0; r < 4; ++r) gemm_epilogue_store(g, m, n + r, v[r]);
    }
  } else {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    int z = 0;
    for (; z + 3 < g.split_k; z += 4) {
#pragma unroll
      for (int q = 0; q < 4; ++q) s[q] += g.ws[(int64_t)(z + q) * total + idx];
    }
    for (; z < g.split_k; ++z) s[0] += g.ws[(int64_t)z * total + idx];
    gemm_epilogue_store(g, (int)(idx / g.N), (int)(idx % g.N), (s[0] + s[1]) + (s[2] + s[3]));
  }
}

int launch_splitk_reduce(const GemmArgs& g, hipStream_t s) {
  if (g.split_k > 1) {
    const int64_t total = (int64_t)g.M * g.N;
    if (g.N % 4 == 0) {
      hipLaunchKernelGGL(gemm_splitk_reduce<true>, dim3((unsigned)((total / 4 + 63) / 64)), dim3(256), 0, s, g);
    } else {
      hipLaunchKernelGGL(gemm_splitk_reduce<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, g);
    }
    GDM_LAUNCH_OK("gdm_gemm(split-k reduce)");
  }
  return GDM_OK;
}

template <typename CT>
int launch_gemm(const GemmArgs& g, int a_dtype, int b_dtype, hipStream_t s) {
  dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, g.split_k), block(NT);
  if (a_dtype == GDM_F32 && b_dtype == GDM_F32) hipLaunchKernelGGL((gemm_kernel<CT, float, float>), grid, block, 0, s, g);
  else if (a_dtype == GDM_F32) hipLaunchKernelGGL((gemm_kernel<CT, float, __bf16>), grid, block, 0, s, g);
  else if (b_dtype == GDM_F32) hipLaunchKernelGGL((gemm_kernel<CT, __bf16, float>), grid, block, 0, s, g);
  else hipLaunchKernelGGL((gemm_kernel<CT, __bf16, __bf16>), grid, block, 0, s, g);
  GDM_LAUNCH_OK("gdm_gemm");
  return launch_splitk_reduce(g, s);
}

}  // namespace

extern "C" int gdm_gemm(const void* A, int a_dtype, int64_t sam, int64_t sak, const void* B, int b_dtype, int64_t sbk,
                        int64_t sbn, void* C, int c_dtype, int64_t scm, int64_t scn, int M, int N, int K,
                        const float* bias_n, const float* bias_m, int act, float slope, int compute_dtype, int split_k,
                        void* workspace, size_t workspace_bytes, void* stream) {
  GDM_REQUIRE(A && B && C, "gdm_gemm: null operand");
  GDM_REQUIRE(M > 0 && N > 0 && K > 0, "gdm_gemm: bad dims M=%d N=%d K=%d", M, N, K);
  GDM_REQUIRE(gdm_dtype_ok(a_dtype) && gdm_dtype_ok(b_dtype) && gdm_dtype_ok(c_dtype) && gdm_dtype_ok(compute_dtype),
              "gdm_gemm: bad dtype");
  GDM_REQUIRE(act >= GDM_ACT_NONE && act <= GDM_ACT_SIGMOID, "gdm_gemm: bad activation %d", act);
  GDM_REQUIRE(split_k >= 1 && split_k <= 65535, "gdm_gemm: bad split_k %d", split_k);
  GemmArgs probe{A, sam, sak, B, sbk, sbn, C, c_dtype, scm, scn, M, N, K, bias_n, bias_m, act, slope, 1, K, nullptr};
  const bool fast = compute_dtype == GDM_BF16 && gdm_gemm_bf16_fast_ok(probe, a_dtype, b_dtype);
  const int KT = fast ? GDM_GEMM_FAST_KT : (compute_dtype == GDM_BF16 ? GemmCfg<__bf16>::KT : GemmCfg<float>::KT);
  int tiles = (K + KT - 1) / KT;
  if (split_k > tiles) split_k = tiles;
  int per = (tiles + split_k - 1) / split_k;
  split_k = (tiles + per - 1) / per;  // no empty slab
  if (split_k > 1) {
    if (workspace == nullptr || workspace_bytes < (size_t)split_k * M * N * sizeof(float)) {
      gdm_set_error("gdm_gemm: split_k=%d needs %zu workspace bytes, got %zu", split_k,
                    (size_t)split_k * M * N * sizeof(float), workspace_bytes);
      return GDM_EWORKSPACE;
    }
  }
  GemmArgs g{A, sam, sak, B, sbk, sbn, C, c_dtype, scm, scn, M, N, K, bias_n, bias_m, act, slope,
             split_k, per * KT, (float*)workspace};
  hipStream_t s = (hipStream_t)stream;
  if (fast) {
    int rc = gdm_gemm_bf16_fast_launch(g, a_dtype, b_dtype, s);
    if (rc != GDM_OK) return rc;
    return launch_splitk_reduce(g, s);
  }
  return compute_dtype == GDM_BF16 ? launch_gemm<__bf16>(g, a_dtype, b_dtype, s)
                                   : launch_gemm<float>(g, a_dtype, b_dtype, s);
}
