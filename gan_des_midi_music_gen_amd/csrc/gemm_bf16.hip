// Vectorised bf16-MFMA GEMM for the weight-/activation-streaming products of the hot path (M <= a few hundred, K or N
// in the tens of thousands): model 1's fc1 forward / dW / dX (GAN_DES/SIMNN.py:126,140 and their autograd mm's), the
// generators' ConvTranspose2d-as-GEMM products and model 2's conv GEMMs.
//
//   128x128 output tile per 256-thread workgroup (4 waves as 2x2, each 64x64 = 4x4 v_mfma_f32_16x16x32_bf16 tiles).
//   These products are bandwidth-bound, so what decides their speed is bytes in flight.  Operand tiles are fetched
//   ahead into registers with 16-byte buffer loads (out-of-range chunks read zeros: no branch around a load, counted
//   waits) and the LDS images are XOR-swizzled so that fragment reads and stores are bank-conflict-free.  Two variants:
//   K tile 64 fetched TWO tiles ahead with double-buffered LDS (one barrier per tile) for skinny long-K products that
//   put few workgroups on a CU, and K tile 32 fetched one tile ahead (16 KB LDS) for products with many short ones.
//
// Operand layouts (chosen on the host from the strides; anything else falls back to the generic kernel in gemm.hip):
//   K-major  (k stride 1):   LDS image [row][KT k]   fragment = one ds_read_b128
//                            16-byte piece p of row r sits at piece p ^ kswz(r)
//   R-major  (row stride 1): LDS image [k][128 rows] fragment = two ds_read_b64_tr_b16 (hardware transpose: the
//                            contraction index is the slow axis in memory)
//                            16-element block b of k-row k sits at block b ^ ((k & 3) | ((k >> 1) & 4))
// fp32 operands are converted to bf16 on the way into LDS.  The MFMA is issued with the operands swapped (D^T), so a
// lane ends up with 4 CONSECUTIVE n of one output row: row-major C is written 16 B (fp32) / 8 B (bf16) per lane.
#include <cstdlib>
#include <type_traits>
#include "gemm_common.h"

namespace {

constexpr int BM = 128, BN = 128, NT = 256;

using rsrc_t = __amdgpu_buffer_rsrc_t;
constexpr uint32_t BUF_OOB = 0x80000000u;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, uint32_t bytes) {
  const uint64_t a = (uint64_t)p;      // descriptor words must be provably wave-uniform (no waterfall loop)
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0,
                                           __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// Lanes of one wave exchange data through LDS without a workgroup barrier (the wave-private epilogue scratch): the
// compiler reasons per thread and would otherwise move the scratch stores under the (per-lane) condition of the loads.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ bf16x4 lds_tr16(const __bf16* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p);
}

// K-major image swizzle: 16-byte piece p of row r sits at piece p ^ kswz(r).  Found by exhaustive search over the
// ds_read_b128 lane groups ({0-3,12-15,20-27}, ...): conflict-free for 16 consecutive rows at any row offset.
// activation with a compile-time selector (same formulas as apply_act)
template <int ACT> __device__ __forceinline__ float act_const(float v, float slope) {
  if constexpr (ACT == GDM_ACT_RELU) return v > 0.f ? v : 0.f;
  else if constexpr (ACT == GDM_ACT_LEAKY) return v > 0.f ? v : v * slope;
  else if constexpr (ACT == GDM_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
  else return v;
}

template <int KT> __device__ __forceinline__ int kswz(int row) { return KT == 64 ? ((row >> 1) & 7) : ((row >> 1) & 2); }
// R-major image swizzle: 16-element block b of k-row k sits at block b ^ rswz(k) (ds_read_b64_tr_b16 reads k-rows
// {q, 8+q} x 16 rows per 32-lane half: eight distinct blocks).
__device__ __forceinline__ int rswz(int k) { return (k & 3) | ((k >> 1) & 4); }

// One operand tile in flight: 16-byte chunks per lane.
template <typename T, int KT> struct Stage {
  static constexpr int N = 128 * KT * (int)sizeof(T) / 16 / NT;
  f32x4 v[N];
};

// Per-lane constants of an operand's staging pattern (the chunk -> (row, k) map never changes).
template <typename T, int KT> struct Lane {
  uint32_t off[Stage<T, KT>::N];     // byte offset of the chunk inside the tile at k0 = 0
  int kl[Stage<T, KT>::N];           // k of the chunk relative to the tile; -1: row outside the matrix
  int lds[Stage<T, KT>::N];          // element offset in the LDS image
};

template <typename T, int KT, bool KMAJ>
__device__ __forceinline__ void lane_init(Lane<T, KT>& ln, int64_t ld, int rows, int r0) {
  constexpr int EPC = 16 / sizeof(T);               // elements per 16-byte chunk
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < Stage<T, KT>::N; ++i) {
    const int c = t + NT * i;
    int rl, kl;
    if constexpr (KMAJ) {
      constexpr int CPR = KT / EPC;                 // chunks per row
      rl = c / CPR; kl = (c % CPR) * EPC;
      ln.off[i] = (uint32_t)(((int64_t)(r0 + rl) * ld + kl) * (int64_t)sizeof(T));
      ln.lds[i] = rl * KT + 8 * ((kl >> 3) ^ kswz<KT>(rl)) + (kl & 7);
    } else {
      constexpr int CPK = 128 / EPC;                // chunks per k-row
      kl = c / CPK; rl = (c % CPK) * EPC;
      ln.off[i] = (uint32_t)(((int64_t)kl * ld + r0 + rl) * (int64_t)sizeof(T));
      ln.lds[i] = kl * 128 + (rl ^ (16 * rswz(kl)));
    }
    ln.kl[i] = (r0 + rl < rows) ? kl : -1;          // rows % EPC == 0: a chunk is inside or outside as a whole
  }
}

// ---- global -> registers, tile origin k0 (chunks at or past kend, or in rows past the matrix, read zeros)
template <typename T, int KT, bool KMAJ>
__device__ __forceinline__ void stage_load(Stage<T, KT>& st, const Lane<T, KT>& ln, rsrc_t rs, int64_t ld, int k0,
                                           int kend) {
  const uint32_t kbytes = (uint32_t)((KMAJ ? (int64_t)k0 : (int64_t)k0 * ld) * (int64_t)sizeof(T));
#pragma unroll
  for (int i = 0; i < Stage<T, KT>::N; ++i) {
    const bool ok = ln.kl[i] >= 0 && k0 + ln.kl[i] < kend;
    st.v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? kbytes + ln.off[i] : BUF_OOB, 0, 0));
  }
}

// ---- registers -> LDS image (bf16)
template <typename T, int KT>
__device__ __forceinline__ void stage_store(const Stage<T, KT>& st, const Lane<T, KT>& ln, __bf16* __restrict__ img) {
#pragma unroll
  for (int i = 0; i < Stage<T, KT>::N; ++i) {
    if constexpr (sizeof(T) == 2) {
      *(f32x4*)(img + ln.lds[i]) = st.v[i];
    } else {
      bf16x4 h;
#pragma unroll
      for (int e = 0; e < 4; ++e) h[e] = (__bf16)st.v[i][e];
      *(bf16x4*)(img + ln.lds[i]) = h;
    }
  }
}

// fragment of rows [row0, row0+16) for MFMA k-step kk (32 k) of the tile
template <int KT, bool KMAJ>
__device__ __forceinline__ bf16x8 frag_read(const __bf16* __restrict__ img, int row0, int kk, int lr, int lg) {
  if constexpr (KMAJ) {
    const int row = row0 + lr;
    return *(const bf16x8*)&img[row * KT + 8 * ((4 * kk + lg) ^ kswz<KT>(row))];
  } else {
    const int q = lr >> 2, p = lr & 3;
    const int kr = 32 * kk + 8 * lg + q;
    const int n = (row0 + 4 * p) ^ (16 * (q | (4 * (lg & 1))));     // = rswz(kr) = rswz(kr + 4)
    const bf16x4 lo = lds_tr16(&img[kr * 128 + n]);
    const bf16x4 hi = lds_tr16(&img[(kr + 4) * 128 + n]);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  }
}

// DEPTH = tiles fetched ahead = LDS buffers.  DEPTH 2 / KT 64: 64 KB of LDS and ~64 KB in flight per workgroup, for
// products with few workgroups per CU (long K, split-K).  DEPTH 1 / KT 32: 16 KB of LDS, for products whose many short
// workgroups hide each other's latency.
template <typename TA, bool A_KMAJ, typename TB, bool B_KMAJ, int KT, int DEPTH>
__global__ __launch_bounds__(NT) void gemm_bf16_fast(GemmArgs g) {
  constexpr int IMG = 128 * KT;                                         // elements of one operand image
  __shared__ __attribute__((aligned(16))) __bf16 smem[2 * DEPTH * IMG];    // [buffer][A | B]
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
  // split-K slices are INTERLEAVED k tiles (slice z takes tiles z, z + split_k, ...), not contiguous ranges: the
  // workgroups of one output tile run side by side, so together they sweep each operand row contiguously and a DRAM
  // page is used up while it is open (with contiguous slices every workgroup pulls 128-byte pieces from pages of its
  // own, 128 KB apart per row).  The slab sum is the same set of products in a different, still fixed, order.
  const int kstep = KT * g.split_k;
  const int kbeg = zs * KT;
  const int kend = g.K;
  const int64_t lda = A_KMAJ ? g.sam : g.sak;
  const int64_t ldb = B_KMAJ ? g.sbn : g.sbk;
  // whole-operand descriptors (sizes checked < 2 GiB on the host)
  const rsrc_t ra = make_rsrc(g.A, (uint32_t)((A_KMAJ ? (int64_t)(g.M - 1) * lda + g.K : (int64_t)(g.K - 1) * lda + g.M) *
                                             (int64_t)sizeof(TA)));
  const rsrc_t rb = make_rsrc(g.B, (uint32_t)((B_KMAJ ? (int64_t)(g.N - 1) * ldb + g.K : (int64_t)(g.K - 1) * ldb + g.N) *
                                             (int64_t)sizeof(TB)));
  Lane<TA, KT> la;
  Lane<TB, KT> lb;
  lane_init<TA, KT, A_KMAJ>(la, lda, g.M, m0);
  lane_init<TB, KT, B_KMAJ>(lb, ldb, g.N, n0);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto multiply = [&](const __bf16* As, const __bf16* Bs) {
#pragma unroll
    for (int kk = 0; kk < KT / 32; ++kk) {
      bf16x8 a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = frag_read<KT, A_KMAJ>(As, wm * 64 + 16 * i, kk, lr, lg);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = frag_read<KT, B_KMAJ>(Bs, wn * 64 + 16 * j, kk, lr, lg);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(b[j], a[i], acc[i][j]);   // D^T: lane -> (m = lr, n = 4*lg + r)
    }
  };
  Stage<TA, KT> sa[DEPTH];
  Stage<TB, KT> sb[DEPTH];
  if constexpr (DEPTH == 2) {
    auto tile = [&](Stage<TA, KT>& xa, Stage<TB, KT>& xb, __bf16* As, int k0) {
      __bf16* Bs = As + IMG;
      stage_store<TA, KT>(xa, la, As);         // waits for THIS stage's loads only: the other stage's were issued later
      stage_store<TB, KT>(xb, lb, Bs);
      __syncthreads();                         // one barrier per tile: the other LDS buffer is what laggards still read
      stage_load<TA, KT, A_KMAJ>(xa, la, ra, lda, k0 + 2 * kstep, kend);   // two tiles ahead, always issued (zeros past kend)
      stage_load<TB, KT, B_KMAJ>(xb, lb, rb, ldb, k0 + 2 * kstep, kend);
      multiply(As, Bs);
    };
    stage_load<TA, KT, A_KMAJ>(sa[0], la, ra, lda, kbeg, kend);
    stage_load<TB, KT, B_KMAJ>(sb[0], lb, rb, ldb, kbeg, kend);
    stage_load<TA, KT, A_KMAJ>(sa[1], la, ra, lda, kbeg + kstep, kend);
    stage_load<TB, KT, B_KMAJ>(sb[1], lb, rb, ldb, kbeg + kstep, kend);
    // tiles are taken in pairs so that both register stages have a fixed place in the loop; an odd tail tile is zeros
    for (int k0 = kbeg; k0 < kend; k0 += 2 * kstep) {
      tile(sa[0], sb[0], smem, k0);
      tile(sa[DEPTH - 1], sb[DEPTH - 1], smem + 2 * IMG * (DEPTH - 1), k0 + kstep);
    }
  } else {
    __bf16* As = smem;
    __bf16* Bs = smem + IMG;
    stage_load<TA, KT, A_KMAJ>(sa[0], la, ra, lda, kbeg, kend);
    stage_load<TB, KT, B_KMAJ>(sb[0], lb, rb, ldb, kbeg, kend);
    for (int k0 = kbeg; k0 < kend; k0 += kstep) {
      __syncthreads();                         // previous tile's fragment reads are done
      stage_store<TA, KT>(sa[0], la, As);
      stage_store<TB, KT>(sb[0], lb, Bs);
      __syncthreads();
      stage_load<TA, KT, A_KMAJ>(sa[0], la, ra, lda, k0 + kstep, kend);    // next tile, always issued (zeros past kend)
      stage_load<TB, KT, B_KMAJ>(sb[0], lb, rb, ldb, k0 + kstep, kend);
      multiply(As, Bs);
    }
  }

  // ---- epilogue.  The activation is a compile-time constant of the (four) epilogue bodies: a run-time switch per
  //      element costs more instructions than the whole K loop of a short product.
  auto epilogue = [&](auto act_tag) {
    constexpr int ACT = decltype(act_tag)::value;
    const bool vec_ok = (g.N % 4 == 0);
    if (g.split_k > 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + 16 * i + lr;
        if (m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = n0 + wn * 64 + 16 * j + 4 * lg;
          if (n >= g.N) continue;
          float* dst = g.ws + ((int64_t)zs * g.M + m) * g.N + n;
#ifndef GDM_GEMM_DEFAULT_STORES
          // (outputs and split-K slabs are consumed by a later kernel: non-temporal, like the activation streams of
          //  simnn_disc.hip -- 0.7 % per iteration in same-box A/B)
          if (vec_ok) __builtin_nontemporal_store(acc[i][j], (f32x4*)dst);
#else
          if (vec_ok) *(f32x4*)dst = acc[i][j];
#endif
          else
            for (int r = 0; r < 4 && n + r < g.N; ++r) dst[r] = acc[i][j][r];
        }
      }
      return;
    }
    if (vec_ok && m0 + BM <= g.M && n0 + BN <= g.N &&
        (g.c_dtype != GDM_BF16 || (g.scm % 8 == 0 && ((uintptr_t)g.C & 15) == 0))) {
      // Interior tile: turn the accumulator layout (a lane owns 4 consecutive n of ONE row: 16 rows x 64 bytes per
      // store) into row-contiguous stores through a wave-private 16 x 64 fp32 LDS scratch (XOR-swizzled by the row):
      // one store instruction writes 4 rows x 256 bytes (fp32 C) or 8 rows x 128 bytes (bf16 C).
      __syncthreads();                                   // every wave is done reading the operand images
      float* scr = (float*)smem + w * (16 * 64);
      const bool is_bf16 = g.c_dtype == GDM_BF16;
      // this lane's (row, column) inside a 16 x 64 pass, and its bias vectors
      const int prow = is_bf16 ? (l >> 3) : (l >> 4), pcol = is_bf16 ? 8 * (l & 7) : 4 * (l & 15);
      const int nn = n0 + wn * 64 + pcol;
      f32x4 bn0 = {0.f, 0.f, 0.f, 0.f}, bn1 = bn0;
      if (g.bias_n) {
        bn0 = *(const f32x4*)(g.bias_n + nn);
        if (is_bf16) bn1 = *(const f32x4*)(g.bias_n + nn + 4);
      }
      const bool has_bias = g.bias_n || g.bias_m;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        wave_lds_sync();                     // the previous pass's scratch reads are done
#pragma unroll
        for (int j = 0; j < 4; ++j) *(f32x4*)&scr[lr * 64 + 4 * ((4 * j + lg) ^ lr)] = acc[i][j];
        wave_lds_sync();
        if (is_bf16) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int row = 8 * h + prow, c8 = l & 7;
            f32x4 v0 = *(const f32x4*)&scr[row * 64 + 4 * ((2 * c8) ^ row)];
            f32x4 v1 = *(const f32x4*)&scr[row * 64 + 4 * ((2 * c8 + 1) ^ row)];
            const int m = m0 + wm * 64 + 16 * i + row;
            if (has_bias) {
              const float bm = g.bias_m ? g.bias_m[m] : 0.f;
              v0 += bn0 + (f32x4){bm, bm, bm, bm};
              v1 += bn1 + (f32x4){bm, bm, bm, bm};
            }
            bf16x8 hv;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              hv[r] = (__bf16)act_const<ACT>(v0[r], g.slope);
              hv[4 + r] = (__bf16)act_const<ACT>(v1[r], g.slope);
            }
#ifndef GDM_GEMM_DEFAULT_STORES
            __builtin_nontemporal_store(hv, (bf16x8*)((__bf16*)g.C + (int64_t)m * g.scm + nn));
#else
            *(bf16x8*)((__bf16*)g.C + (int64_t)m * g.scm + nn) = hv;
#endif
          }
        } else {
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            const int row = 4 * h + prow, pc = l & 15;
            f32x4 v = *(const f32x4*)&scr[row * 64 + 4 * (pc ^ row)];
            const int m = m0 + wm * 64 + 16 * i + row;
            if (has_bias) {
              const float bm = g.bias_m ? g.bias_m[m] : 0.f;
              v += bn0 + (f32x4){bm, bm, bm, bm};
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = act_const<ACT>(v[r], g.slope);
#ifndef GDM_GEMM_DEFAULT_STORES
            __builtin_nontemporal_store(v, (f32x4*)((float*)g.C + (int64_t)m * g.scm + nn));
#else
            *(f32x4*)((float*)g.C + (int64_t)m * g.scm + nn) = v;
#endif
          }
        }
      }
      return;
    }
    // edge tiles / odd N
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wm * 64 + 16 * i + lr;
      if (m >= g.M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + 16 * j + 4 * lg;
        if (n >= g.N) continue;
        f32x4 v = acc[i][j];
        for (int r = 0; r < 4 && n + r < g.N; ++r) {
          float y = v[r];
          if (g.bias_n) y += g.bias_n[n + r];
          if (g.bias_m) y += g.bias_m[m];
          store_from_f32(g.C, g.c_dtype, (int64_t)m * g.scm + n + r, act_const<ACT>(y, g.slope));
        }
      }
    }
  };
  switch (g.act) {
    case GDM_ACT_RELU: epilogue(std::integral_constant<int, GDM_ACT_RELU>{}); break;
    case GDM_ACT_LEAKY: epilogue(std::integral_constant<int, GDM_ACT_LEAKY>{}); break;
    case GDM_ACT_SIGMOID: epilogue(std::integral_constant<int, GDM_ACT_SIGMOID>{}); break;
    default: epilogue(std::integral_constant<int, GDM_ACT_NONE>{}); break;
  }
}

// variant 0: K tile 32, one tile ahead, 16 KB LDS (many workgroups per CU); 1: K tile 64, two tiles ahead, 64 KB LDS
template <typename TA, bool AK, typename TB, bool BK>
void launch_v(const GemmArgs& g, int variant, dim3 grid, hipStream_t s) {
  if (variant == 1) hipLaunchKernelGGL((gemm_bf16_fast<TA, AK, TB, BK, 64, 2>), grid, dim3(NT), 0, s, g);
  else hipLaunchKernelGGL((gemm_bf16_fast<TA, AK, TB, BK, 32, 1>), grid, dim3(NT), 0, s, g);
}
template <typename TA, bool AK, typename TB>
void launch_b(const GemmArgs& g, bool b_kmaj, int variant, dim3 grid, hipStream_t s) {
  if (b_kmaj) launch_v<TA, AK, TB, true>(g, variant, grid, s);
  else launch_v<TA, AK, TB, false>(g, variant, grid, s);
}
template <typename TA, bool AK>
void launch_a(const GemmArgs& g, int b_dtype, bool b_kmaj, int variant, dim3 grid, hipStream_t s) {
  if (b_dtype == GDM_BF16) launch_b<TA, AK, __bf16>(g, b_kmaj, variant, grid, s);
  else launch_b<TA, AK, float>(g, b_kmaj, variant, grid, s);
}

// rows = extent of the non-contracted axis.  16-byte chunks must be inside or outside the matrix as a whole, and the
// operand must be addressable through one 32-bit buffer descriptor.
inline bool operand_ok(const void* p, int dtype, int64_t s_row, int64_t s_k, int rows, int K, bool* kmaj) {
  const int64_t esz = dtype == GDM_BF16 ? 2 : 4, epc = 16 / esz;
  if (((uintptr_t)p & 15) != 0) return false;
  if (s_k == 1 && s_row >= 1 && (s_row * esz) % 16 == 0 && K % epc == 0) {
    *kmaj = true;
    return ((int64_t)(rows - 1) * s_row + K) * esz < ((int64_t)1 << 31);
  }
  if (s_row == 1 && s_k >= 1 && (s_k * esz) % 16 == 0 && rows % epc == 0) {
    *kmaj = false;
    return ((int64_t)(K - 1) * s_k + rows) * esz < ((int64_t)1 << 31);
  }
  return false;
}

}  // namespace

bool gdm_gemm_bf16_fast_ok(const GemmArgs& g, int a_dtype, int b_dtype) {
  bool ak, bk;
  if (!operand_ok(g.A, a_dtype, g.sam, g.sak, g.M, g.K, &ak)) return false;
  if (!operand_ok(g.B, b_dtype, g.sbn, g.sbk, g.N, g.K, &bk)) return false;
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
  operand_ok(g.A, a_dtype, g.sam, g.sak, g.M, g.K, &ak);
  operand_ok(g.B, b_dtype, g.sbn, g.sbk, g.N, g.K, &bk);
  const int MT = (g.M + BM - 1) / BM, NTl = (g.N + BN - 1) / BN;
  const int outer = NTl * g.split_k;
  dim3 grid((unsigned)(((outer + 7) / 8) * 8 * MT));
  // split-K products (few workgroups per CU, long K loops): bytes in flight per workgroup decide -> deep variant
  int variant = (g.split_k > 1 && (int64_t)outer * MT <= 512 && g.k_per_split >= 256) ? 1 : 0;
  static const char* force = getenv("GDM_GEMM_VARIANT");      // experiments only
  if (force && force[0]) variant = force[0] == '1';
  if (a_dtype == GDM_BF16) {
    if (ak) launch_a<__bf16, true>(g, b_dtype, bk, variant, grid, s); else launch_a<__bf16, false>(g, b_dtype, bk, variant, grid, s);
  } else {
    if (ak) launch_a<float, true>(g, b_dtype, bk, variant, grid, s); else launch_a<float, false>(g, b_dtype, bk, variant, grid, s);
  }
  GDM_LAUNCH_OK("gdm_gemm(bf16 fast path)");
  return GDM_OK;
}
