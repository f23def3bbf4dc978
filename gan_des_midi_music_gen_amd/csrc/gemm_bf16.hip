// Vectorised bf16-MFMA GEMM for the weight-/activation-streaming products of the hot path (M <= a few hundred, K or N
// in the tens of thousands): model 1's fc1 forward / dW / dX (GAN_DES/SIMNN.py:126,140 and their autograd mm's), the
// generators' ConvTranspose2d-as-GEMM products and model 2's conv GEMMs.
//
//   128x128 output tile per 256-thread workgroup (4 waves as 2x2, each 64x64 = 4x4 v_mfma_f32_16x16x32_bf16 tiles),
//   K tile 32 (one MFMA k-step), global loads are 16 bytes per lane for every layout, register-prefetched one tile
//   ahead of the LDS image that is being consumed.
//
// Operand layouts (chosen on the host from the strides; anything else falls back to the generic kernel in gemm.hip):
//   K-major  (k stride 1):   LDS image [row][k]   (80-B rows), fragment = one ds_read_b128
//   R-major  (row stride 1): LDS image [k][row]   (288-B rows), fragment = two ds_read_b64_tr_b16 (hardware
//                            transpose: the contraction index is the slow axis in memory)
// fp32 operands are converted to bf16 on the way into LDS.  The MFMA is issued with the operands swapped (D^T), so a
// lane ends up with 4 CONSECUTIVE n of one output row: row-major C is written 16 B (fp32) / 8 B (bf16) per lane.
#include "gemm_common.h"

namespace {

constexpr int BM = 128, BN = 128, KT = GDM_GEMM_FAST_KT, NT = 256;
constexpr int LDK = KT + 8;    // K-major image row (elements)
constexpr int LDR = 128 + 16;  // R-major image row (elements)
constexpr int IMG = (128 * LDK > KT * LDR) ? 128 * LDK : KT * LDR;

__device__ __forceinline__ bf16x4 lds_tr16(const __bf16* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p);
}

// One operand's staging registers: 2 x 16 B (bf16 source) or 4 x 16 B (fp32 source) per lane and tile.
template <typename T> struct Stage { static constexpr int N = sizeof(T) == 2 ? 2 : 4; f32x4 v[N]; };

// ---- global -> registers.  `rows` = extent of the non-contracted axis (M or N), tile origin (r0, k0).
template <typename T, bool KMAJ>
__device__ __forceinline__ void stage_load(Stage<T>& st, const T* __restrict__ base, int64_t ld, int rows, int r0,
                                           int k0, int kend) {
  constexpr int EPC = 16 / sizeof(T);               // elements per 16-byte chunk
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < Stage<T>::N; ++i) {
    const int c = t + NT * i;
    int r, k, nvalid;
    const T* p;
    if constexpr (KMAJ) {
      constexpr int CPR = KT / EPC;                 // chunks per row
      r = r0 + c / CPR;
      k = k0 + (c % CPR) * EPC;
      nvalid = (r < rows) ? kend - k : 0;
      p = base + (int64_t)r * ld + k;
    } else {
      constexpr int CPK = 128 / EPC;                // chunks per k-row
      k = k0 + c / CPK;
      r = r0 + (c % CPK) * EPC;
      nvalid = (k < kend) ? rows - r : 0;
      p = base + (int64_t)k * ld + r;
    }
    if (nvalid >= EPC) {
      st.v[i] = *(const f32x4*)p;
    } else {
      T tmp[EPC];
#pragma unroll
      for (int e = 0; e < EPC; ++e) tmp[e] = (e < nvalid) ? p[e] : from_f32<T>(0.f);
      st.v[i] = *(const f32x4*)tmp;
    }
  }
}

// ---- registers -> LDS image (bf16)
template <typename T, bool KMAJ>
__device__ __forceinline__ void stage_store(const Stage<T>& st, __bf16* __restrict__ img) {
  constexpr int EPC = 16 / sizeof(T);
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < Stage<T>::N; ++i) {
    const int c = t + NT * i;
    int off;
    if constexpr (KMAJ) {
      constexpr int CPR = KT / EPC;
      off = (c / CPR) * LDK + (c % CPR) * EPC;
    } else {
      constexpr int CPK = 128 / EPC;
      off = (c / CPK) * LDR + (c % CPK) * EPC;
    }
    if constexpr (sizeof(T) == 2) {
      *(f32x4*)(img + off) = st.v[i];
    } else {
      bf16x4 h;
#pragma unroll
      for (int e = 0; e < 4; ++e) h[e] = (__bf16)st.v[i][e];
      *(bf16x4*)(img + off) = h;
    }
  }
}

template <bool KMAJ>
__device__ __forceinline__ bf16x8 frag_read(const __bf16* __restrict__ img, int row0, int lr, int lg) {
  if constexpr (KMAJ) {
    return *(const bf16x8*)&img[(row0 + lr) * LDK + 8 * lg];
  } else {
    const int q = lr >> 2, p = lr & 3;
    const bf16x4 lo = lds_tr16(&img[(8 * lg + q) * LDR + row0 + 4 * p]);
    const bf16x4 hi = lds_tr16(&img[(8 * lg + 4 + q) * LDR + row0 + 4 * p]);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  }
}

template <typename TA, bool A_KMAJ, typename TB, bool B_KMAJ>
__global__ __launch_bounds__(NT) void gemm_bf16_fast(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) __bf16 smem[2 * IMG];
  __bf16* As = smem;
  __bf16* Bs = smem + IMG;
  const int t = threadIdx.x, l = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;
  const int lr = l & 15, lg = l >> 4;
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so the MT row tiles
  // that stream the SAME B panel (n tile, k slice) are given ids 8 apart -> they run on one XCD and share its L2.
  const int MT = (g.M + BM - 1) / BM, NTl = (g.N + BN - 1) / BN;
  const int id = blockIdx.x;
  const int grp = id / (8 * MT), within = id % (8 * MT);
  const int outer = grp * 8 + (within & 7), mt = within >> 3;
  if (outer >= NTl * g.split_k) return;
  const int nt = outer % NTl, zs = outer / NTl;
  const int m0 = mt * BM, n0 = nt * BN;
  const int kbeg = zs * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);
  const TA* __restrict__ A = (const TA*)g.A;
  const TB* __restrict__ B = (const TB*)g.B;
  const int64_t lda = A_KMAJ ? g.sam : g.sak;
  const int64_t ldb = B_KMAJ ? g.sbn : g.sbk;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  Stage<TA> sa;
  Stage<TB> sb;
  stage_load<TA, A_KMAJ>(sa, A, lda, g.M, m0, kbeg, kend);
  stage_load<TB, B_KMAJ>(sb, B, ldb, g.N, n0, kbeg, kend);
  for (int k0 = kbeg; k0 < kend; k0 += KT) {
    __syncthreads();                         // previous tile's fragment reads are done
    stage_store<TA, A_KMAJ>(sa, As);
    stage_store<TB, B_KMAJ>(sb, Bs);
    __syncthreads();
    if (k0 + KT < kend) {                    // prefetch the next tile while this one is multiplied
      stage_load<TA, A_KMAJ>(sa, A, lda, g.M, m0, k0 + KT, kend);
      stage_load<TB, B_KMAJ>(sb, B, ldb, g.N, n0, k0 + KT, kend);
    }
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = frag_read<A_KMAJ>(As, wm * 64 + 16 * i, lr, lg);
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = frag_read<B_KMAJ>(Bs, wn * 64 + 16 * j, lr, lg);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(b[j], a[i], acc[i][j]);   // D^T: lane -> (m = lr, n = 4*lg + r)
  }

  const bool vec_ok = (g.N % 4 == 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + 16 * i + lr;
    if (m >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + 16 * j + 4 * lg;
      if (n >= g.N) continue;
      f32x4 v = acc[i][j];
      if (g.split_k > 1) {
        float* dst = g.ws + ((int64_t)zs * g.M + m) * g.N + n;
        if (vec_ok) *(f32x4*)dst = v;
        else
          for (int r = 0; r < 4 && n + r < g.N; ++r) dst[r] = v[r];
        continue;
      }
      if (vec_ok) {
        if (g.bias_n) { const f32x4 bn = *(const f32x4*)(g.bias_n + n); v += bn; }
        if (g.bias_m) { const float bm = g.bias_m[m]; v += (f32x4){bm, bm, bm, bm}; }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], g.act, g.slope);
        const int64_t ci = (int64_t)m * g.scm + n;
        if (g.c_dtype == GDM_BF16) {
          bf16x4 h;
#pragma unroll
          for (int r = 0; r < 4; ++r) h[r] = (__bf16)v[r];
          *(bf16x4*)((__bf16*)g.C + ci) = h;
        } else {
          *(f32x4*)((float*)g.C + ci) = v;
        }
      } else {
        for (int r = 0; r < 4 && n + r < g.N; ++r) gemm_epilogue_store(g, m, n + r, v[r]);
      }
    }
  }
}

template <typename TA, bool AK, typename TB>
void launch_b(const GemmArgs& g, bool b_kmaj, dim3 grid, hipStream_t s) {
  if (b_kmaj) hipLaunchKernelGGL((gemm_bf16_fast<TA, AK, TB, true>), grid, dim3(NT), 0, s, g);
  else hipLaunchKernelGGL((gemm_bf16_fast<TA, AK, TB, false>), grid, dim3(NT), 0, s, g);
}
template <typename TA, bool AK>
void launch_a(const GemmArgs& g, int b_dtype, bool b_kmaj, dim3 grid, hipStream_t s) {
  if (b_dtype == GDM_BF16) launch_b<TA, AK, __bf16>(g, b_kmaj, grid, s);
  else launch_b<TA, AK, float>(g, b_kmaj, grid, s);
}

inline bool operand_ok(const void* p, int dtype, int64_t s_row, int64_t s_k, bool* kmaj) {
  const int64_t esz = dtype == GDM_BF16 ? 2 : 4;
  if (((uintptr_t)p & 15) != 0) return false;
  if (s_k == 1 && s_row >= 1 && (s_row * esz) % 16 == 0) { *kmaj = true; return true; }
  if (s_row == 1 && s_k >= 1 && (s_k * esz) % 16 == 0) { *kmaj = false; return true; }
  return false;
}

}  // namespace

bool gdm_gemm_bf16_fast_ok(const GemmArgs& g, int a_dtype, int b_dtype) {
  bool ak, bk;
  if (!operand_ok(g.A, a_dtype, g.sam, g.sak, &ak)) return false;
  if (!operand_ok(g.B, b_dtype, g.sbn, g.sbk, &bk)) return false;
  if (g.scn != 1) return false;
  const int64_t csz = g.c_dtype == GDM_BF16 ? 2 : 4;
  if (g.N % 4 == 0) {   // vector epilogue: rows of C and the bias must allow 4-element accesses
    if ((g.scm * csz) % (4 * csz) != 0 || ((uintptr_t)g.C & (4 * csz - 1)) != 0) return false;
    if (g.bias_n && ((uintptr_t)g.bias_n & 15) != 0) return false;
  }
  // tiny problems gain nothing from 128x128 tiles
  if ((int64_t)g.M * g.N < 64 * 64) return false;
  return true;
}

int gdm_gemm_bf16_fast_launch(const GemmArgs& g, int a_dtype, int b_dtype, hipStream_t s) {
  bool ak = true, bk = true;
  operand_ok(g.A, a_dtype, g.sam, g.sak, &ak);
  operand_ok(g.B, b_dtype, g.sbn, g.sbk, &bk);
  const int MT = (g.M + BM - 1) / BM, NTl = (g.N + BN - 1) / BN;
  const int outer = NTl * g.split_k;
  dim3 grid((unsigned)(((outer + 7) / 8) * 8 * MT));
  if (a_dtype == GDM_BF16) {
    if (ak) launch_a<__bf16, true>(g, b_dtype, bk, grid, s); else launch_a<__bf16, false>(g, b_dtype, bk, grid, s);
  } else {
    if (ak) launch_a<float, true>(g, b_dtype, bk, grid, s); else launch_a<float, false>(g, b_dtype, bk, grid, s);
  }
  GDM_LAUNCH_OK("gdm_gemm(bf16 fast path)");
  return GDM_OK;
}
