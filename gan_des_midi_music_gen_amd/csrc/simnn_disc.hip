// Convolution trunk of model 1's discriminator (GAN_DES/SIMNN.py:123-125, 136-139) for gfx950.
//
//   x (B,H,W) fp32 --conv1 k2 p1 +ReLU +pool2--> p1 (B,H1,W1,16) channels-last
//                  --conv2 k3 p1 +ReLU +pool2--> p2 (B,32,H2,W2) channel-major (= the reference's flatten order)
//
// conv1 (K = 4) is an HBM-bound stencil: one lane per pooled pixel, all 16 channels in registers.
// conv2 (K = 144) is an implicit GEMM on MFMA with M = output channels, N = 16 consecutive pixels of a row,
// the input halo band staged once in LDS as [row][col][channel]; ReLU, the 2x2 max-pool and its argmax code are
// applied to the accumulator tile (row pair in two accumulators, column pair by a lane swap) before anything is
// written, so the full-resolution conv outputs never touch HBM.  Backward kernels rebuild the sparse full-resolution
// gradient (one non-zero per pooling window) in LDS from the pooled gradient + the 1-byte code.
//
// T = float : exact-fp32 mode, v_mfma_f32_16x16x4_f32, LDS pixel records padded to avoid bank conflicts
// T = __bf16: bf16 storage + v_mfma_f32_16x16x32_bf16, fp32 accumulation
#include <cstdlib>
#include "gdm_common.h"
#include "adam_pc.h"

namespace {

// Diagnostic build only (-DGDM_STAMPS, tools/stamps.py): per-workgroup cycle totals of the phases of a persistent
// kernel's tile loop, taken with s_memtime by every wave and written by wave 0 to a buffer nothing else reads.
#ifdef GDM_STAMPS
__device__ unsigned long long gdm_stamp_buf[1024 * 8];
struct Stamps {
  uint64_t last, ph[8];
  __device__ __forceinline__ void start() {
    for (int k = 0; k < 8; ++k) ph[k] = 0;
    __builtin_amdgcn_sched_barrier(0);
    last = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
  }
  __device__ __forceinline__ void mark(int k) {
    __builtin_amdgcn_sched_barrier(0);
    const uint64_t now = __builtin_amdgcn_s_memtime();
    ph[k] += now - last;
    last = now;
    __builtin_amdgcn_sched_barrier(0);
  }
  __device__ __forceinline__ void flush() {
    if (threadIdx.x == 0)
      for (int k = 0; k < 8; ++k) gdm_stamp_buf[(blockIdx.x & 1023) * 8 + k] = ph[k];
  }
};
#define STAMP_DECL Stamps stamps_; stamps_.start()
#define STAMP(k) stamps_.mark(k)
#define STAMP_FLUSH stamps_.flush()
#define STAMP_ARG , Stamps& stamps_
#define STAMP_PASS , stamps_
#else
#define STAMP_DECL
#define STAMP(k)
#define STAMP_FLUSH
#define STAMP_ARG
#define STAMP_PASS
#endif

constexpr int COLS = 64;   // conv-output columns handled per workgroup (column super-tile)

// index (in 16-bit fields) of code1's field for pooled pixel pw of row `row` (= image * H1 + pooled row), group g
__device__ __forceinline__ uint32_t code1_field(uint32_t row, int Q1, int pw, int g) {
  return ((row * (uint32_t)Q1 + (uint32_t)(pw >> 2)) * 4u + (uint32_t)g) * 4u + (uint32_t)(pw & 3);
}

template <typename T> struct Px;  // LDS pixel-record strides (elements) for 16- and 32-channel records
template <> struct Px<float> { static constexpr int S16 = 17, S32 = 33; };
template <> struct Px<__bf16> { static constexpr int S16 = 16, S32 = 32; };

// Tile staging goes through buffer descriptors: a lane that falls outside the image hands the load an offset past
// num_records and the hardware returns zeros (stores are dropped), so halo handling needs no branch and no select.
// Branch-free staging matters twice: the loads of a tile issue back to back, and hipcc's s_waitcnt bookkeeping stays
// exact (with loads inside exec-masked branches it drained the whole queue -- vmcnt(0) -- right after issuing a
// prefetch, which turned "prefetch" into "wait").  All tensors addressed this way are < 2 GiB (checked on the host).
using rsrc_t = __amdgpu_buffer_rsrc_t;
constexpr uint32_t BUF_OOB = 0x80000000u;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, uint32_t bytes) {
  // descriptor words must be provably wave-uniform, or every buffer op gets wrapped in a waterfall loop
  const uint64_t a = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0,
                                           __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#ifndef GDM_IN_LOAD_AUX
#define GDM_IN_LOAD_AUX 0            // cache policy of the x / code1 loads: non-temporal (2) was measured 3 % SLOWER (x is
                                     // read again by the backward of the same iteration: it wants to stay in the Infinity Cache)
#endif
template <int AUX = 0>
__device__ __forceinline__ f32x4 buf_load16(rsrc_t r, uint32_t off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, AUX));
}
template <int AUX = 0>
__device__ __forceinline__ uint64_t buf_load8(rsrc_t r, uint32_t off) {
  return __builtin_bit_cast(uint64_t, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, AUX));
}
template <int AUX = 0>
__device__ __forceinline__ float buf_load4(rsrc_t r, uint32_t off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, AUX));
}
// Cache policy of the big write-once activation streams (conv1: p1 + code1, conv2: p2 + code2; 84 + 75 MB per 256 samples):
// non-temporal.  They are consumed by a LATER kernel, and on this 8-XCD part a kernel boundary writes the XCD's dirty L2
// lines back before the dependent kernel starts; streaming them out as they are produced took 0.5-0.9 % off the
// iteration in same-box A/B (the kernels themselves run as before).  -DGDM_ACT_STORE_AUX=0 restores the default policy.
#ifndef GDM_ACT_STORE_AUX
#define GDM_ACT_STORE_AUX 2
#endif
#ifndef GDM_ACT_STORE_AUX2
#define GDM_ACT_STORE_AUX2 GDM_ACT_STORE_AUX
#endif
template <int AUX = 0>
__device__ __forceinline__ void buf_store16(rsrc_t r, uint32_t off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, AUX);
}
template <int AUX = 0>
__device__ __forceinline__ void buf_store8(rsrc_t r, uint32_t off, uint64_t v) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, off, 0, AUX);
}
__device__ __forceinline__ void buf_store4(rsrc_t r, uint32_t off, uint32_t v) {
  __builtin_amdgcn_raw_buffer_store_b32(v, r, off, 0, 0);
}
template <int AUX = 0>
__device__ __forceinline__ void buf_store2(rsrc_t r, uint32_t off, uint32_t v) {
  __builtin_amdgcn_raw_buffer_store_b16((unsigned short)v, r, off, 0, AUX);
}

// =====================================================================================================================
// conv1 forward: Conv2d(1,16,k2,s1,p1) + ReLU + MaxPool2d(2)
// code1 = one 16-bit field per (pooled pixel, 4-channel group g): nibble k (bits 4k..4k+3) belongs to channel 4g+k:
//   bits [1:0] = argmax position (dy*2+dx, first max in scan order like aten::max_pool2d_with_indices),
//   bit 2 = channel is live (pooled value > 0, i.e. ReLU passes gradient), bit 3 = 0.
// Fields are stored "quad-major": [image][pooled row][quad = pw / 4][group g][pw % 4], Q1 = ceil(W1 / 4) quads per row
// (8 bytes per pixel).  The forward's lane (pixel, group) writes one field, as before; in the fused backward a lane owns
// ONE channel and FOUR neighbouring pixels (the transposed MFMA result), and the four fields of its channel group are
// then 8 contiguous bytes -- one load, no cross-lane transpose on either side.  Pixels >= W1 of a row's last quad
// hold 0 (dead).
//
// The 2x2 stencil is a [16 channels x 4 taps] x [4 taps x pixels] product: one exact-fp32 v_mfma_f32_16x16x4_f32 per
// pooling position and 16 pooled pixels, bias as the accumulator's initial value.  A wave takes units of 16 pooled
// pixels of one pooled row; in the result lane (lr, lg) holds channels 4lg..4lg+3 of pixel lr for all four positions,
// so max-pool, argmax and ReLU are in-lane and the lane stores 8/16 bytes of p1 and one 16-bit code field.  (As a VALU
// stencil this layer cost ~10 instructions per output value and was issue-bound at 40 % of its memory roofline.)
// =====================================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void conv1_fwd_kernel(const float* __restrict__ x0, const float* __restrict__ x1,
                                                        int bsplit, const float* __restrict__ w,
                                                        const float* __restrict__ bias, int B, int H, int W, int H1,
                                                        int W1, int n_rows, T* __restrict__ p1,
                                                        uint16_t* __restrict__ code1) {
  const int t = threadIdx.x, l = t & 63, lr = l & 15, lg = l >> 4;
  const int Q1 = (W1 + 3) >> 2;
  // the wave index is uniform: keep everything derived from it in scalar registers
  const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (t >> 6)), n_waves = gridDim.x * 4;
  const int upr = (W1 + 15) >> 4;                            // units per pooled row
  // the batch is the concatenation of two input tensors (images 0 .. bsplit-1 | bsplit .. B-1: real | generated)
  const rsrc_t xr0 = make_rsrc(x0, (uint32_t)bsplit * H * W * 4);
  const rsrc_t xr1 = make_rsrc(x1, (uint32_t)(B - bsplit) * H * W * 4);
  const rsrc_t pr = make_rsrc(p1, (uint32_t)B * H1 * W1 * 16 * sizeof(T));
  const rsrc_t cr = make_rsrc(code1, (uint32_t)B * H1 * Q1 * 32);
  // A operand: lane (row = channel lr, k = tap lg); accumulator rows 4lg + r = channels
  const float aw = w[lr * 4 + lg];
  const f32x4 b4 = {bias[4 * lg], bias[4 * lg + 1], bias[4 * lg + 2], bias[4 * lg + 3]};
  // B operand: lane (k = tap lg = kh*2+kw, column = pixel lr) reads x[2ph-1+kh+dy][2pw-1+kw+dx] for position (dy,dx)
  const int r_lo = (lg >> 1) - 1, c_lo = 2 * lr + (lg & 1) - 1;
  uint32_t loff[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) loff[p] = (uint32_t)(((r_lo + (p >> 1)) * W + c_lo + (p & 1)) * 4);   // may wrap: added to base

  // Work is dealt by pooled ROW: wave w takes rows w, w + n_waves, ... of the (b, ph) row space and walks each row's
  // units left to right in CHUNKS of DEPTH units that share one row record: everything that depends on the row (base
  // offsets, row-interior flag, validity) is computed once per row, behind a scalar branch, and a unit costs a handful
  // of scalar instructions.  The CU's ONE scalar unit is what this kernel queues for: with per-unit position
  // arithmetic with carries (~100 SALU per unit) it was slower than its VALU work; with four independently advancing
  // slots (~45 SALU per unit, a row change every other step of each slot) a phase-stamp build still showed 36 % of a
  // wave's time in "bookkeeping + next loads" (`tools/stamps.py c1`).
  struct Row { int ph, b; uint32_t xrow, pixrow, crow; bool interior, valid, second; };
  const int dph = n_waves % H1, db = n_waves / H1;
  auto set_row = [&](Row& q) {
    const int b = min(q.b, B - 1);                           // past the end: re-read the last image (stores are dropped)
    q.second = b >= bsplit;
    q.xrow = (uint32_t)((((q.second ? b - bsplit : b) * H + 2 * q.ph) * W) * 4);
    q.pixrow = (uint32_t)((b * H1 + q.ph) * W1);
    q.crow = (uint32_t)(b * H1 + q.ph);
    q.interior = q.ph > 0 && 2 * q.ph + 1 < H;
    q.valid = q.b < B;
  };
  auto load = [&](const Row& q, int seg, float (&xv)[4]) {
    const rsrc_t xr = q.second ? xr1 : xr0;                  // scalar select
    const uint32_t base = q.xrow + 128u * (uint32_t)seg;
    if (q.interior && seg > 0 && 32 * seg + 32 < W) {        // interior unit (scalar test)
      // the lane's 2x2 patch as two 8-byte loads (4-byte aligned): memory instructions, not bytes, are what the
      // address unit charges for
#pragma unroll
      for (int dy = 0; dy < 2; ++dy) {
        const uint64_t two = buf_load8<GDM_IN_LOAD_AUX>(xr, base + loff[2 * dy]);
        xv[2 * dy] = __builtin_bit_cast(float, (uint32_t)two);
        xv[2 * dy + 1] = __builtin_bit_cast(float, (uint32_t)(two >> 32));
      }
      return;
    }
    // edge unit (also: a unit past the end of its row, seg >= upr -- its stores are dropped)
    const int row0 = 2 * q.ph + r_lo, col0 = 32 * seg + c_lo;
    const bool rok[2] = {(unsigned)row0 < (unsigned)H, (unsigned)(row0 + 1) < (unsigned)H};
    const bool cok[2] = {(unsigned)col0 < (unsigned)W, (unsigned)(col0 + 1) < (unsigned)W};
#pragma unroll
    for (int p = 0; p < 4; ++p) xv[p] = buf_load4<GDM_IN_LOAD_AUX>(xr, (rok[p >> 1] && cok[p & 1]) ? base + loff[p] : BUF_OOB);
  };
  STAMP_DECL;
  auto finish = [&](const Row& q, int seg, const float (&xv)[4]) {
    f32x4 acc[4];
#ifdef GDM_STAMPS
    asm volatile("" :: "v"(xv[0]), "v"(xv[1]), "v"(xv[2]), "v"(xv[3]));      // the unit's loads have landed
    STAMP(0);
#endif
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw, xv[p], b4, 0, 0, 0);
#ifdef GDM_STAMPS
    asm volatile("" :: "v"(acc[3][3]));                                        // the MFMAs have finished
    STAMP(1);
#endif
    uint32_t field = 0;
    float best[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v0 = acc[0][r], v1 = acc[1][r], v2 = acc[2][r], v3 = acc[3][r];
      const float m = fmaxf(fmaxf(v0, v1), fmaxf(v2, v3));
      uint32_t pos = 3u;                        // select chain, last write wins = first maximum in scan order
      pos = v2 == m ? 2u : pos;
      pos = v1 == m ? 1u : pos;
      pos = v0 == m ? 0u : pos;
      best[r] = fmaxf(m, 0.f);
      field |= (m > 0.f ? pos | 4u : pos) << (4 * r);
    }
    const int pw = 16 * seg + lr;
    const uint32_t pix = q.pixrow + (uint32_t)pw;
    const bool ok = q.valid && pw < W1;
#ifdef GDM_STAMPS
    asm volatile("" :: "v"(field), "v"(best[0]), "v"(best[3]));
    STAMP(2);
#endif
    if constexpr (sizeof(T) == 2) {
      bf16x4 h;
#pragma unroll
      for (int r = 0; r < 4; ++r) h[r] = (__bf16)best[r];
      buf_store8<GDM_ACT_STORE_AUX>(pr, ok ? pix * 32u + 8u * lg : BUF_OOB, __builtin_bit_cast(uint64_t, h));
    } else {
      buf_store16(pr, ok ? pix * 64u + 16u * lg : BUF_OOB, (f32x4){best[0], best[1], best[2], best[3]});
    }
    // the whole last quad is written (zeros beyond W1): its consumers load four pixels' fields at once
    buf_store2<GDM_ACT_STORE_AUX2>(cr, (q.valid && pw < 4 * Q1) ? code1_field(q.crow, Q1, pw, lg) * 2u : BUF_OOB,
                                   ok ? field : 0u);
    STAMP(3);
  };
  if (wave >= n_rows) return;
  // DEPTH units in flight per wave: while the units of this chunk are finished, the loads of the next chunk (same row
  // or the wave's next row) are issued into the registers they free.  Every trip issues the same loads and stores
  // (units past the end of a row or of the batch read valid memory and their stores are dropped), so the waits between
  // them are exact counts.  The sched_barriers keep the compiler from sinking the refill loads below the next unit's
  // MFMAs (which would expose their latency again).
#ifndef GDM_C1_DEPTH
#define GDM_C1_DEPTH 4
#endif
  constexpr int DEPTH = GDM_C1_DEPTH;
  Row rc;
  float xv[DEPTH][4];
  rc.ph = wave % H1; rc.b = wave / H1;
  set_row(rc);
  int seg0 = 0;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) load(rc, d, xv[d]);
  while (rc.valid) {
    Row rn = rc;
    int seg0n = seg0 + DEPTH;
    if (seg0n >= upr) {                         // the wave's next row
      seg0n = 0;
      rn.ph += dph; rn.b += db;
      if (rn.ph >= H1) { rn.ph -= H1; ++rn.b; }
      set_row(rn);
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      finish(rc, seg0 + d, xv[d]);
      load(rn, seg0n + d, xv[d]);
      STAMP(4);
      __builtin_amdgcn_sched_barrier(0);
    }
    rc = rn;
    seg0 = seg0n;
  }
  STAMP_FLUSH;
}

// =====================================================================================================================
// conv1 backward (weights): dW1[c][kh][kw] = sum live * dp1[c] * x[2ph+dy-1+kh][2pw+dx-1+kw], db1[c] = sum live*dp1[c]
// Two-stage fixed-order reduction: per-workgroup slab of 80 floats, then a 1-block final sum.
// =====================================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void conv1_bwd_weight_kernel(const T* __restrict__ dp1,
                                                               const uint16_t* __restrict__ code1,
                                                               const float* __restrict__ x, int B, int H, int W,
                                                               int H1, int W1, float* __restrict__ slabs) {
  __shared__ float red[4][80];
  float acc[80];
#pragma unroll
  for (int i = 0; i < 80; ++i) acc[i] = 0.f;
  const int64_t total = (int64_t)B * H1 * W1;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int pw = (int)(idx % W1);
    const int ph = (int)((idx / W1) % H1);
    const int b = (int)(idx / ((int64_t)W1 * H1));
    const float* xb = x + (int64_t)b * H * W;
    float in[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int ih = 2 * ph - 1 + r;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int iw = 2 * pw - 1 + s;
        in[r][s] = (ih >= 0 && ih < H && iw >= 0 && iw < W) ? xb[(int64_t)ih * W + iw] : 0.f;
      }
    }
    uint32_t fields[4];                                                         // code1 format: see conv1_fwd_kernel
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) fields[g4] = code1[code1_field((uint32_t)(b * H1 + ph), (W1 + 3) >> 2, pw, g4)];
    const T* g16 = dp1 + idx * 16;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const uint32_t nib = fields[c >> 2] >> (4 * (c & 3));
      const bool live = (nib >> 2) & 1;
      const float g = live ? to_f32(g16[c]) : 0.f;
      const int pos = (int)(nib & 3);
      const bool dy = pos >> 1, dx = pos & 1;
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int kw = 0; kw < 2; ++kw) {
          const float a = dx ? in[kh][kw + 1] : in[kh][kw];
          const float bsel = dx ? in[kh + 1][kw + 1] : in[kh + 1][kw];
          acc[c * 4 + kh * 2 + kw] = fmaf(g, dy ? bsel : a, acc[c * 4 + kh * 2 + kw]);
        }
      acc[64 + c] += g;
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 80; ++i) {
    const float s = wave_sum(acc[i]);
    if (lane == 0) red[wv][i] = s;
  }
  __syncthreads();
  if (threadIdx.x < 80)
    slabs[(int64_t)blockIdx.x * 80 + threadIdx.x] =
        ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// =====================================================================================================================
// conv1 backward (data): dx[i][j] = sum_{c,kh,kw} w[c][kh][kw] * dy[c][i+1-kh][j+1-kw], where dy is the sparse
// full-resolution gradient of the (H+1)x(W+1) conv output: dy[c][oh][ow] = dp1[oh/2][ow/2][c] if channel c of that
// pooled pixel is live and its argmax position is (oh&1, ow&1), else 0 (conv rows/columns beyond 2*H1 / 2*W1 were
// dropped by the floor pooling).  Not on the reference's training path (the discriminator's inputs are data or detached
// bridge outputs, SIMNN.py:283,299-306): this completes the module's autograd (aten::convolution_backward input grad,
// SIMNN.py:136).  One thread per input pixel; the <= 4 pooled pixels it touches are read straight from HBM/L2.
// =====================================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void conv1_bwd_data_kernel(const T* __restrict__ dp1,
                                                             const uint16_t* __restrict__ code1,
                                                             const float* __restrict__ w, int B, int H, int W, int H1,
                                                             int W1, float* __restrict__ dx) {
  __shared__ float ws[64];
  if (threadIdx.x < 64) ws[threadIdx.x] = w[threadIdx.x];
  __syncthreads();
  const int64_t total = (int64_t)B * H * W;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int j = (int)(idx % W), i = (int)((idx / W) % H), b = (int)(idx / ((int64_t)W * H));
    float acc = 0.f;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int kw = 0; kw < 2; ++kw) {
        const int oh = i + 1 - kh, ow = j + 1 - kw;          // >= 0 always
        const int ph = oh >> 1, pw = ow >> 1;
        if (ph >= H1 || pw >= W1) continue;
        const int64_t pix = ((int64_t)b * H1 + ph) * W1 + pw;
        uint32_t fields[4];                                                       // code1 format: see conv1_fwd_kernel
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) fields[g4] = code1[code1_field((uint32_t)(b * H1 + ph), (W1 + 3) >> 2, pw, g4)];
        const uint32_t pos_here = (uint32_t)((oh & 1) * 2 + (ow & 1));
        const T* g16 = dp1 + pix * 16;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          const uint32_t nib = fields[c >> 2] >> (4 * (c & 3));
          const bool hit = ((nib >> 2) & 1u) && (nib & 3u) == pos_here;
          acc += hit ? ws[c * 4 + kh * 2 + kw] * to_f32(g16[c]) : 0.f;
        }
      }
    dx[idx] = acc;
  }
}

// =====================================================================================================================
// Fixed-order sum of `nslabs` slabs of `width` floats (deterministic replacement for float atomics), with the
// gradient's final placement fused into the last level.
// A 1024-thread workgroup owns 64 consecutive elements of one slab group; wave w adds slabs w, w+16, ... of its group
// (coalesced 256-B rows), the 16 partials are added in wave order.  Wide slabs (conv2's 4640 floats x up to 1024 slabs)
// take two levels (groups of <= 256 slabs, then the group partials in order); 80-float slabs take one launch.
// SINK: 0 = out[group][i];  1 = conv1 gradients (dw[64], db[16], optional accumulate);  2 = conv2 gradients
// (slab order [o 32][tap 9][ci 16] -> dw[o][ci][tap], then db[32]).
// =====================================================================================================================
template <int SINK>
__global__ __launch_bounds__(1024) void slab_sum_kernel(const float* __restrict__ slabs, int nslabs, int per_group,
                                                        int width, float* __restrict__ out, float* __restrict__ dw,
                                                        float* __restrict__ db, int accumulate) {
  __shared__ float part[16][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  const int k0 = blockIdx.y * per_group, k1 = min(nslabs, k0 + per_group);
  // wave wv adds slabs k0 + wv + 16 j; eight independent partial sums so that eight row loads are in flight at once
  float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (i < width) {
    int k = k0 + wv;
    for (; k + 16 * 7 < k1; k += 16 * 8) {
#pragma unroll
      for (int q = 0; q < 8; ++q) s8[q] += slabs[(int64_t)(k + 16 * q) * width + i];
    }
    for (; k < k1; k += 16) s8[0] += slabs[(int64_t)k * width + i];      // (no run-time register index)
  }
  part[wv][lane] = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
  __syncthreads();
  if (wv == 0 && i < width) {
    float t = part[0][lane];
#pragma unroll
    for (int w = 1; w < 16; ++w) t += part[w][lane];
    if constexpr (SINK == 0) {
      out[(int64_t)blockIdx.y * width + i] = t;
    } else if constexpr (SINK == 1) {
      float* dst = i < 64 ? dw + i : db + (i - 64);
      *dst = accumulate ? *dst + t : t;
    } else {
      if (i < 4608) {
        const int ci = i & 15, tap = (i >> 4) % 9, o = i / 144;
        dw[(o * 16 + ci) * 9 + tap] = t;
      } else {
        db[i - 4608] = t;
      }
    }
  }
}

// sums `nslabs` slabs into their sink using `scratch` (>= slab_groups(nslabs, width) * width floats)
inline int slab_groups(int nslabs, int width) {
  if (width <= 128 || nslabs <= 64) return 1;
  const int g = (nslabs + 255) / 256;
  return g > 64 ? 64 : g;
}
template <int SINK>
inline void launch_slab_sum(const float* slabs, int nslabs, int width, float* scratch, float* dw, float* db,
                            int accumulate, hipStream_t s) {
  const int groups = slab_groups(nslabs, width);
  const int per = (nslabs + groups - 1) / groups;
  const unsigned gx = (unsigned)((width + 63) / 64);
  if (groups == 1) {
    hipLaunchKernelGGL(slab_sum_kernel<SINK>, dim3(gx, 1), dim3(1024), 0, s, slabs, nslabs, per, width, (float*)nullptr,
                       dw, db, accumulate);
  } else {
    hipLaunchKernelGGL(slab_sum_kernel<0>, dim3(gx, groups), dim3(1024), 0, s, slabs, nslabs, per, width, scratch,
                       (float*)nullptr, (float*)nullptr, 0);
    hipLaunchKernelGGL(slab_sum_kernel<SINK>, dim3(gx, 1), dim3(1024), 0, s, (const float*)scratch, groups, groups, width,
                       (float*)nullptr, dw, db, accumulate);
  }
}

// =====================================================================================================================
// conv2 block.  Tile = ROWS (4) conv rows x COLS (128) conv columns of one image per 256-thread workgroup; wave w owns
// columns [32w, 32w+32).  Activations/gradients around it are channels-last:
//   p1, dp1 (B,H1,W1,16)   p2, dp2 (B,H2,W2,32)   code2 (B,H2,W2,16) uint8: one byte per channel PAIR (2j, 2j+1) =
//   8 * (c_even + 5 * c_odd) with c = 0..3 argmax position, 4 = ReLU-dead -- i.e. the byte offset of the pair's entry
//   in the backward kernels' 8-byte selector tables (Code2Tables), so that rebuilding the sparse full-resolution
//   gradient costs one v_bfe, two table reads and four v_perm per two channels and four positions
// Weights are consumed from a pre-packed image (gdm_simnn_conv2_pack, rebuilt after every optimizer step):
//   forward  image Wf[32][KPF]: k = tap*16 + ci                       (zero padded to the MFMA k granularity)
//   backward image Wb[16][KPB]: k = tap'*32 + o with tap' = 8 - tap   (the flipped kernel of the data gradient)
// =====================================================================================================================
constexpr int ROWS = 4;
template <typename T> struct C2 {
  static constexpr int KPF = sizeof(T) == 2 ? 168 : 146;
  static constexpr int KPB = sizeof(T) == 2 ? 296 : 290;
  static constexpr int WF_ELEMS = 32 * KPF, WB_ELEMS = 16 * KPB;
  static constexpr int S16 = Px<T>::S16, S32 = Px<T>::S32;
  static constexpr int WP = COLS + 2;
  // records per parity plane of a band row in conv2 forward's even/odd-split image: WP / 2 = 33, padded to 34 -- with 33
  // the four pixels of an 8-lane ds_write_b128 group (columns c, c+1, c+2, c+3 -> planes 0, 1, 0, 1) put two of them on the
  // same 32 store banks (33 * 32 B = 32 mod 128): every band store was 2-way conflicted; 34 * 32 B = 64 mod 128 separates them
  static constexpr int HP = WP / 2 + 1;
  static constexpr int IN_ELEMS = (ROWS + 2) * 2 * HP * S16;     // p1 halo band (split image: 2 planes per row)
};
constexpr int XROWS = 2 * ROWS + 1;                          // input-window rows behind ROWS rows of the p1 geometry

template <typename T>
__global__ __launch_bounds__(256) void conv2_pack_kernel(const float* __restrict__ w, T* __restrict__ wf,
                                                         T* __restrict__ wb) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < C2<T>::WF_ELEMS) {
    const int o = i / C2<T>::KPF, k = i % C2<T>::KPF;
    float v = 0.f;
    if (k < 144) v = w[(o * 16 + (k & 15)) * 9 + (k >> 4)];
    wf[i] = from_f32<T>(v);
  }
  if (i < C2<T>::WB_ELEMS) {
    const int ci = i / C2<T>::KPB, k = i % C2<T>::KPB;
    float v = 0.f;
    if (k < 288) v = w[((k & 31) * 16 + ci) * 9 + (8 - (k >> 5))];
    wb[i] = from_f32<T>(v);
  }
}

// ---- the discriminator's whole optimizer step in ONE launch -------------------------------------------------------------
// disc_opt.step() (GAN_DES/SIMNN.py:316) was adam_prep -> Adam(small parameters) -> Adam(fc1.weight, transposing) ->
// conv2 re-pack: four launches at the end of the iteration's critical chain (~15 us of launch tails for 5 us of work).
// Here the grid is fc1.weight's (p, c, n) tiles, and workgroup 0 first updates the 4.9 k small parameters and rebuilds
// conv2's packed MFMA images from the weights it has just written.  Same arithmetic (adam_element,
// adam_derived): bit-identical to the four launches.
//
// The device record `rec` (GDM_SIMNN_ADAM_RECORD_INTS ints, zero before the first launch) carries what a launch needs
// besides `hyper`:
//   * completion counters in two levels -- rec[0] counts finished GROUPS, rec[REC_GROUP0 + 16 g] the finished workgroups
//     of group g = block % 64, each on its own cache line: the workgroup that finishes last advances hyper's step
//     counter (all have read it by then).  (Two levels keep 2 k returning atomics off ONE address, where L2 serialises
//     them as the kernel ends; at this count the single counter measured the same, 61.0 vs 61.6 us.)
//   * the bias-correction terms of a step, cached: slot (step & 1) = {step, step_size, sqrt(bias_correction2)}.  The
//     workgroup 0 derives the NEXT step's terms (two double-precision pow) into the other slot, which nobody reads
//     during this launch; a workgroup whose slot does not carry its step (first launch, record rewritten by the host:
//     the host zeroes `rec` then) derives them itself.
constexpr int REC_SLOT0 = 16, REC_GROUP0 = 32, REC_GROUPS = 64;
constexpr int REC_INTS = REC_GROUP0 + 16 * REC_GROUPS;
template <typename T>
__global__ __launch_bounds__(256, 8) void simnn_adam_kernel(float* __restrict__ p, const float* __restrict__ g_pc,
                                                         float* __restrict__ m, float* __restrict__ v, int C, int P,
                                                         T* __restrict__ shadow_pc, int vec_ok, int tiles_x, int tiles_y,
                                                         int small_vec_ok, float* __restrict__ ps,
                                                         const float* __restrict__ gs, float* __restrict__ ms,
                                                         float* __restrict__ vs, int n_small,
                                                         const float* __restrict__ w2, T* __restrict__ wf,
                                                         T* __restrict__ wb, float* __restrict__ hyper, int* rec) {
  __shared__ __attribute__((aligned(16))) float lds[4608];          // the tile; workgroup 0: conv2.weight before that
  static_assert(sizeof(lds) >= sizeof(float[32][132]), "tile");
  float (&tile)[32][132] = *reinterpret_cast<float (*)[32][132]>(lds);
  __shared__ float hy[2];
  STAMP_DECL;
  // hyper's step counter and BOTH slots are read at once (one round trip, not a dependent chain of three -- with 2 k
  // workgroups streaming, a round trip is ~2 us), and a tile workgroup looks at them only after it has issued its
  // gradient gather.
  const int step = __float_as_int(hyper[0]) + 1;
  const int tag0 = rec[REC_SLOT0], tag1 = rec[REC_SLOT0 + 4];
  const float ss0 = __int_as_float(rec[REC_SLOT0 + 1]), bq0 = __int_as_float(rec[REC_SLOT0 + 2]);
  const float ss1 = __int_as_float(rec[REC_SLOT0 + 5]), bq1 = __int_as_float(rec[REC_SLOT0 + 6]);
  float step_size, bc2_sqrt;
  auto resolve = [&]() {
    const bool odd = step & 1;
    step_size = odd ? ss1 : ss0;
    bc2_sqrt = odd ? bq1 : bq0;
    if ((odd ? tag1 : tag0) != step) {                     // uniform; first launch, or the host has rewritten the record
      if (threadIdx.x == 0) adam_derived(hyper, step, hy[0], hy[1]);
      __syncthreads();
      step_size = hy[0]; bc2_sqrt = hy[1];
    }
  };
  const int blk = blockIdx.x;
  STAMP(0);
  // Workgroup 0 (dispatched first) does the small work BEFORE its tile.  The 2 k tile workgroups are exactly the chip's
  // 8 x 256 slots and run side by side for the whole kernel; a workgroup of its own for the small work either waits for
  // a slot (as the last block: 61 us instead of 47) or takes one, and the tile workgroup it displaces then runs alone
  // behind all the others (55 us).
  if (blk == 0) {
    resolve();
    // ~5 k elements + ~10 k packed values in one workgroup, while 2 k others saturate the memory system: a dependent
    // round trip costs ~5 us here, so what counts is their NUMBER.  The small range is read in rounds of 2 x 16 bytes
    // per thread and array (all eight loads of a round in flight; more would push the kernel past 64 VGPRs: 7 workgroups
    // per CU instead of 8, and the 2 k tile workgroups no longer run in one round), the updated conv2.weight stays in LDS
    // and the packed images are built from there.
    const float w1 = 1.0f - hyper[2], beta2 = hyper[3], omb2 = 1.0f - hyper[3], eps = hyper[4], gscale = hyper[5];
    const int w2_off = (int)(w2 - ps);
    auto keep_w2 = [&](int i, float val) {
      const unsigned j = (unsigned)(i - w2_off);
      if (j < 4608u) lds[j] = val;
    };
    const int n4 = small_vec_ok ? n_small / 4 : 0;
    constexpr int R = 2;
    for (int v0 = threadIdx.x; v0 < n4; v0 += 256 * R) {
      f32x4 pj[R], mj[R], vj[R], gj[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int iv = min(v0 + 256 * r, n4 - 1);
        pj[r] = ((const f32x4*)ps)[iv]; mj[r] = ((const f32x4*)ms)[iv]; vj[r] = ((const f32x4*)vs)[iv];
        gj[r] = ((const f32x4*)gs)[iv];
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int iv = v0 + 256 * r;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float a = pj[r][e], bm = mj[r][e], bv = vj[r][e];
          adam_element(a, bm, bv, gj[r][e], gscale, w1, beta2, omb2, eps, step_size, bc2_sqrt);
          pj[r][e] = a; mj[r][e] = bm; vj[r][e] = bv;
          if (iv < n4) keep_w2(4 * iv + e, a);
        }
        if (iv < n4) { ((f32x4*)ps)[iv] = pj[r]; ((f32x4*)ms)[iv] = mj[r]; ((f32x4*)vs)[iv] = vj[r]; }
      }
    }
    for (int i = 4 * n4 + threadIdx.x; i < n_small; i += 256) {      // the last n_small % 4 elements (all, if unaligned)
      float a = ps[i], bm = ms[i], bv = vs[i];
      adam_element(a, bm, bv, gs[i], gscale, w1, beta2, omb2, eps, step_size, bc2_sqrt);
      ps[i] = a; ms[i] = bm; vs[i] = bv;
      keep_w2(i, a);
    }
    __syncthreads();                                       // lds[0 .. 4607] = the updated conv2.weight
    STAMP(1);
    constexpr int U = 4;                                   // independent LDS reads per trip (one per trip: 15 us)
    for (int i0 = threadIdx.x; i0 < C2<T>::WF_ELEMS; i0 += 256 * U) {
      float val[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = min(i0 + 256 * u, C2<T>::WF_ELEMS - 1), o = i / C2<T>::KPF, k = i % C2<T>::KPF;
        val[u] = k < 144 ? lds[(o * 16 + (k & 15)) * 9 + (k >> 4)] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (i0 + 256 * u < C2<T>::WF_ELEMS) wf[i0 + 256 * u] = from_f32<T>(val[u]);
    }
    for (int i0 = threadIdx.x; i0 < C2<T>::WB_ELEMS; i0 += 256 * U) {
      float val[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = min(i0 + 256 * u, C2<T>::WB_ELEMS - 1), ci = i / C2<T>::KPB, k = i % C2<T>::KPB;
        val[u] = k < 288 ? lds[((k & 31) * 16 + ci) * 9 + (8 - (k >> 5))] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (i0 + 256 * u < C2<T>::WB_ELEMS) wb[i0 + 256 * u] = from_f32<T>(val[u]);
    }
    STAMP(2);
    if (threadIdx.x == 255) {                              // the next step's terms, into the slot this launch does not read
      float ss, bq;
      adam_derived(hyper, step + 1, ss, bq);
      int* nxt = rec + REC_SLOT0 + 4 * ((step + 1) & 1);
      nxt[1] = __float_as_int(ss); nxt[2] = __float_as_int(bq); nxt[0] = step + 1;
    }
    __syncthreads();                                       // the tile below reuses lds
    STAMP(3);
  }
  {
    const int bx = blk % tiles_x, by = (blk / tiles_x) % tiles_y, bz = blk / (tiles_x * tiles_y);
    adam_pc_gather(tile, g_pc, C, P, bx, by, bz);
    resolve();
    __syncthreads();
    adam_pc_update<T>(tile, p, m, v, C, P, shadow_pc, hyper, vec_ok, step_size, bc2_sqrt, bx, by, bz);
  }
  __syncthreads();
  STAMP(4);
  STAMP_FLUSH;
  if (threadIdx.x == 0) {
    const int grp = (int)blockIdx.x % REC_GROUPS, n_grp = min((int)gridDim.x, REC_GROUPS);
    const int in_grp = ((int)gridDim.x - grp + REC_GROUPS - 1) / REC_GROUPS;
    if (atomicAdd(rec + REC_GROUP0 + 16 * grp, 1) == in_grp - 1) {
      atomicExch(rec + REC_GROUP0 + 16 * grp, 0);          // nobody else touches this counter before the next launch
      if (atomicAdd(rec, 1) == n_grp - 1) {
        atomicExch(rec, 0);
        hyper[0] = __int_as_float(step);
        hyper[6] = step_size;
        hyper[7] = bc2_sqrt;
      }
    }
  }
}

template <typename T>
__device__ __forceinline__ void copy_to_lds(T* __restrict__ dst, const T* __restrict__ src, int nelems) {
  const int chunks = nelems * (int)sizeof(T) / 16;
  for (int i = threadIdx.x; i < chunks; i += 256) ((f32x4*)dst)[i] = ((const f32x4*)src)[i];
}

// p1 halo band (rows r_first .. r_first+NR-1, cols c_first .. c_first+WP-1) -> LDS [row][col][S16], zero outside.
// Two phases so that every global load of the tile is in flight before the first LDS store.
template <typename T, int NR> struct P1Stage {
  static constexpr int PIECES = (sizeof(T) == 2) ? 2 : 4;          // 16-byte pieces per pixel record
  static constexpr int ITERS = (NR * C2<T>::WP * PIECES + 255) / 256;
  f32x4 v[ITERS];
};

template <typename T, int NR>
__device__ __forceinline__ void p1_band_load(P1Stage<T, NR>& st, rsrc_t p1r, uint32_t img_off, int H1, int W1,
                                             int r_first, int c_first) {
  constexpr int WP = C2<T>::WP, PIECES = P1Stage<T, NR>::PIECES;
#pragma unroll
  for (int k = 0; k < P1Stage<T, NR>::ITERS; ++k) {
    const int i = threadIdx.x + 256 * k;
    const int piece = i % PIECES, pix = i / PIECES;
    const int cl = pix % WP, rl = pix / WP;
    const int r = r_first + rl, c = c_first + cl;
    const bool ok = i < NR * WP * PIECES && r >= 0 && r < H1 && c >= 0 && c < W1;
    const uint32_t off = img_off + (uint32_t)(r * W1 + c) * (16 * sizeof(T)) + piece * 16;   // bytes
    st.v[k] = buf_load16(p1r, ok ? off : BUF_OOB);
  }
}

// SPLIT: records of a band row are stored even columns first, then odd columns ([row][parity][WP/2][S16]), so that a
// reader whose lanes walk every second column (conv2 forward: a lane owns one pooled column) still steps one record
// per lane -- the conflict-free pattern of the plain layout.
template <typename T, int NR, bool SPLIT = false>
__device__ __forceinline__ void p1_band_store(const P1Stage<T, NR>& st, T* __restrict__ in_s) {
  constexpr int S16 = C2<T>::S16, WP = C2<T>::WP, PIECES = P1Stage<T, NR>::PIECES, EPP = 16 / PIECES;
#pragma unroll
  for (int k = 0; k < P1Stage<T, NR>::ITERS; ++k) {
    const int i = threadIdx.x + 256 * k;
    if (i >= NR * WP * PIECES) continue;
    const int piece = i % PIECES, pix = i / PIECES;
    int rec = pix;
    if constexpr (SPLIT) {
      const int cl = pix % WP, rl = pix / WP;
      rec = (rl * 2 + (cl & 1)) * C2<T>::HP + (cl >> 1);
    }
    T* dst = in_s + rec * S16 + piece * EPP;
    if constexpr (sizeof(T) == 2) *(f32x4*)dst = st.v[k];
    else { dst[0] = st.v[k][0]; dst[1] = st.v[k][1]; dst[2] = st.v[k][2]; dst[3] = st.v[k][3]; }
  }
}

typedef short s16x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------------- conv2 forward
// One tile (4 conv rows x 64 columns of image b) from the LDS band in_s: MFMA implicit GEMM + bias/ReLU/pool epilogue.
// Wave (rp, half) computes conv rows 2rp, 2rp+1 x 32 columns x 32 channels.  The mapping is chosen so that the 2x2
// pooling window never leaves a lane: column tile j holds the band columns 32*half + 2*lr + j (lane lr = pooled column),
// so {acc[i][d][j]} over (d, j) ARE the window; and accumulator row 4*lg + r of channel tile i is channel 8*lg + 4*i + r,
// so a lane ends up with 8 consecutive channels of one pooled pixel = one 16-byte store (+ one 8-byte code store).
// The bias is the accumulator's initial value.
template <typename T>
__device__ __forceinline__ void conv2_fwd_tile(const T* __restrict__ in_s, const T* __restrict__ w_s,
                                               const float (&bo)[2][4], int b, int rq, int c0, int H2, int W2,
                                               rsrc_t p2r, rsrc_t code2r STAMP_ARG) {
  constexpr int S16 = C2<T>::S16, KP = C2<T>::KPF, HP = C2<T>::HP;
  const int t = threadIdx.x, l = t & 63, wv = t >> 6, lr = l & 15, lg = l >> 4;
  const int rp = wv >> 1, half = wv & 1;            // wave -> (pooled row of the tile, 32-column half)
  f32x4 acc[2][2][2];                               // [channel tile][row of the pair][column parity]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][d][j] = (f32x4){bo[i][0], bo[i][1], bo[i][2], bo[i][3]};
  const int arow = 8 * (lr >> 2) + (lr & 3);        // + 4*i: the channel this lane's A row stands for
  const int pcol = 16 * half + lr;                  // pooled column within the tile
  if constexpr (sizeof(T) == 2) {
    // fragments of k-step ks+1 are read from LDS before the 8 MFMAs of k-step ks are issued (register double buffer),
    // so that the LDS latency sits under the matrix pipe instead of in front of every MFMA pair
    bf16x8 a[2][2], bb[2][2][2];
    auto frags = [&](int ks, bf16x8 (&aa)[2], bf16x8 (&bx)[2][2]) {
      int tap = 2 * ks + (lg >> 1);
      tap = tap > 8 ? 8 : tap;               // k >= 144: weights are zero, read any valid record
      const int kh = tap / 3, kw = tap % 3;
#pragma unroll
      for (int i = 0; i < 2; ++i) aa[i] = *(const bf16x8*)&w_s[(arow + 4 * i) * KP + 32 * ks + 8 * lg];
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int q = j + kw;               // band column 2*pcol + q
          bx[d][j] = *(const bf16x8*)&in_s[(((2 * rp + d + kh) * 2 + (q & 1)) * HP + pcol + (q >> 1)) * S16 +
                                           8 * (lg & 1)];
        }
    };
    frags(0, a[0], bb[0]);
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
      if (ks + 1 < 5) frags(ks + 1, a[(ks + 1) & 1], bb[(ks + 1) & 1]);
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[i][d][j] = mfma16(a[ks & 1][i], bb[ks & 1][d][j], acc[i][d][j]);
    }
  } else {
    // exact-fp32 mode: four k-steps (one tap) at a time -- the eight weight reads and sixteen activation reads of a tap are
    // issued together, ahead of its 32 MFMAs (two dependent LDS reads in front of every MFMA pair left the fp32 matrix
    // pipe waiting)
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      const int kh = tap / 3, kw = tap % 3;
      float a[4][2], bb[4][2][2];
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) {
        const int ks = 4 * tap + k4, ci = 4 * k4 + lg;
#pragma unroll
        for (int i = 0; i < 2; ++i) a[k4][i] = w_s[(arow + 4 * i) * KP + 4 * ks + lg];
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int q = j + kw;
            bb[k4][d][j] = in_s[(((2 * rp + d + kh) * 2 + (q & 1)) * HP + pcol + (q >> 1)) * S16 + ci];
          }
      }
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][d][j] = mfma16(a[k4][i], bb[k4][d][j], acc[i][d][j]);
    }
  }
  STAMP(2);
  // ---- epilogue, all in-lane: first maximum of the window in scan order (as aten::max_pool2d_with_indices), ReLU,
  //      code = window position, or 4 when the pooled value is not positive (ReLU passes no gradient)
  float best[8];
  uint32_t codes = 0;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v0 = acc[i][0][0][r], v1 = acc[i][0][1][r], v2 = acc[i][1][0][r], v3 = acc[i][1][1][r];
      const float m = fmaxf(fmaxf(v0, v1), fmaxf(v2, v3));
      uint32_t pos = 3u;                      // select chain, last write wins = first maximum (no branches)
      pos = v2 == m ? 2u : pos;
      pos = v1 == m ? 1u : pos;
      pos = v0 == m ? 0u : pos;
      best[4 * i + r] = fmaxf(m, 0.f);
      const uint32_t c = m > 0.f ? pos : 4u;
      // pair byte 8 * (c_even + 5 * c_odd) of channels 2j, 2j+1 (j = (4i + r) / 2) = byte j of the word (<= 192)
      codes += ((r & 1) ? c * 40u : c * 8u) << (8 * ((4 * i + r) >> 1));
    }
  const int ph = (ROWS / 2) * rq + rp, pw = (c0 >> 1) + pcol;
  const bool ok = ph < H2 && pw < W2;
  const uint32_t gi = (uint32_t)((b * H2 + ph) * W2 + pw) * 32 + 8 * lg;
  if constexpr (sizeof(T) == 2) {
    bf16x8 h;
#pragma unroll
    for (int e = 0; e < 8; ++e) h[e] = (__bf16)best[e];
    buf_store16<GDM_ACT_STORE_AUX>(p2r, ok ? gi * 2u : BUF_OOB, __builtin_bit_cast(f32x4, h));
  } else {
    buf_store16(p2r, ok ? gi * 4u : BUF_OOB, (f32x4){best[0], best[1], best[2], best[3]});
    buf_store16(p2r, ok ? gi * 4u + 16u : BUF_OOB, (f32x4){best[4], best[5], best[6], best[7]});
  }
  __builtin_amdgcn_raw_buffer_store_b32(codes, code2r, ok ? gi >> 1 : BUF_OOB, 0, GDM_ACT_STORE_AUX);
}

// Persistent over tiles; the weight image and the biases are fetched once per workgroup; input bands are prefetched
// TWO tiles ahead in two register sets (a tile's MFMA + epilogue is ~4x shorter than an HBM round trip under load, so
// one tile of look-ahead leaves the workgroup waiting for memory most of the time).
template <typename T>
__global__ __launch_bounds__(256) void conv2_fwd_kernel(const T* __restrict__ p1, const T* __restrict__ wf,
                                                        const float* __restrict__ bias, int H1, int W1, int H2,
                                                        int W2, int n_ctiles, int n_tiles, T* __restrict__ p2,
                                                        uint8_t* __restrict__ code2) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
  T* in_s = (T*)dyn_smem;
  T* w_s = in_s + C2<T>::IN_ELEMS;
  const int lg = (threadIdx.x & 63) >> 4;
  const int nrq = (2 * H2 + ROWS - 1) / ROWS, G = gridDim.x;
  // Every path through the loop issues the SAME number of global loads (tile indices past the end are clamped to the
  // last tile instead of skipping the loads): the hardware counts memory operations in order, so only then can the
  // compiler wait for "all but the other register set's loads" (vmcnt(N)) instead of draining the queue (vmcnt(0)).
  STAMP_DECL;
  const int B = n_tiles / (n_ctiles * nrq);
  const rsrc_t p1r = make_rsrc(p1, (uint32_t)B * H1 * W1 * 16 * sizeof(T));
  const rsrc_t p2r = make_rsrc(p2, (uint32_t)B * H2 * W2 * 32 * sizeof(T));
  const rsrc_t code2r = make_rsrc(code2, (uint32_t)B * H2 * W2 * 16);
  auto issue = [&](P1Stage<T, ROWS + 2>& st, int u) {
    u = min(u, n_tiles - 1);
    const int ct = u % n_ctiles, rq = (u / n_ctiles) % nrq, b = u / (n_ctiles * nrq);
    p1_band_load(st, p1r, (uint32_t)b * H1 * W1 * 16 * sizeof(T), H1, W1, ROWS * rq - 1, ct * COLS - 1);
  };
  auto run = [&](P1Stage<T, ROWS + 2>& st, int u, const float (&bo)[2][4]) {
    const int ct = u % n_ctiles, rq = (u / n_ctiles) % nrq, b = u / (n_ctiles * nrq);
    p1_band_store<T, ROWS + 2, true>(st, in_s);
    STAMP(6);
    __syncthreads();
    STAMP(0);
    issue(st, u + 2 * G);                                   // this register set is free again: two tiles ahead
    STAMP(1);
    conv2_fwd_tile<T>(in_s, w_s, bo, b, rq, ct * COLS, H2, W2, p2r, code2r STAMP_PASS);
    STAMP(3);
    __syncthreads();                                        // the band may be overwritten
    STAMP(4);
  };
  copy_to_lds(w_s, wf, C2<T>::WF_ELEMS);
  float bo[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) bo[i][r] = bias[8 * lg + 4 * i + r];     // channel map of conv2_fwd_tile
  P1Stage<T, ROWS + 2> sa, sb;
  // Workgroups are dealt round-robin over the 8 XCDs (each with its own L2): give every XCD a CONTIGUOUS run of tile
  // ids per round, so that the tiles that share halo rows / columns (vertical neighbours are n_ctiles ids apart) are
  // fetched through the same L2.  With tile id = workgroup id every neighbour lived on another XCD and the 1.55x halo
  // over-read of the 6 x 66 band went to HBM in full (157 MB fetched for 101 MB of p1).
  int u = blockIdx.x;                                       // host guarantees gridDim.x <= n_tiles
  if ((G & 7) == 0) u = (u & 7) * (G >> 3) + (u >> 3);
  issue(sa, u);
  issue(sb, u + G);
  STAMP(5);
  for (; u + G < n_tiles; u += 2 * G) {
    run(sa, u, bo);
    run(sb, u + G, bo);
  }
  if (u < n_tiles) run(sa, u, bo);
  STAMP_FLUSH;
}

// ---------------------------------------------------------------------------------- conv2 backward (data [+ conv1 dW])
// dp1[ih][iw][ci] = sum_{ah,aw,o} dc2[ih-1+ah][iw-1+aw][o] * Wb[ci][(ah*3+aw)*32 + o]        (K = 288)
// where dc2 is the sparse full-resolution gradient  dc2[r][c][o] = (code2[r/2][c/2][o] == 2(r&1)+(c&1)) ? dp2[..] : 0.
//
// A persistent 256-thread workgroup walks STRIPS (image b, 64-column tile ct) top to bottom in steps of 4 output
// rows; wave w owns the 16 columns of column tile w (4 rows = 4 accumulators).  dc2 lives in an 8-row LDS ring: step rq needs conv rows
// 4rq-1 .. 4rq+4 and only the two pooled rows 2rq+1, 2rq+2 (conv rows 4rq+2 .. 4rq+5) are new, so every dp2/code2
// element of the strip is fetched and expanded once (a stand-alone 4-row tile with halo re-expands 2.1x as much).  A
// strip starts with the pseudo step rq = -1 (pooled rows -1 [zeros] and 0, no output).  The pooled rows of the next
// step are fetched into registers right after this step's expansion (software prefetch across steps and strips);
// every staging access is a buffer load/store whose out-of-image lanes read zeros / are dropped, so the step body has
// no branch around memory operations.  The bf16 ring keeps a pixel's four 16-byte channel groups XOR-swizzled by its
// column (group ^ ((col >> 1) & 2)): the B-fragment ds_read_b128 of 16 neighbouring pixels (64-byte records, lane
// groups {0-3,12-15,20-27}, ...) is then bank-conflict-free for every column offset (searched exhaustively).
//
// FUSE: instead of (or besides) writing dp1, route it through conv1's ReLU/pool code and contract it with the input
// window held in LDS:  dW1[c][kh][kw] += live * dp1[c] * x[2ih+dy-1+kh][2iw+dx-1+kw],  db1[c] += live * dp1[c];
// the workgroup's 80 partial sums go to one slab (summed in fixed order by slab_sum_kernel).
// Selector tables for code2's pair bytes (bf16 kernels): entry e = c_even + 5 * c_odd at byte offset 8e of
//   table A: {selector for position 0, position 1}   table B: {position 2, position 3}   table C: {live channels, -}
// A selector moves the even channel's bf16 gradient (bytes 0,1 of the packed pair) and/or the odd channel's (bytes 2,3)
// to its place when that channel's argmax is the position, and writes zero (0x0c) otherwise.  25 entries x 8 B: two
// entries share a 16-byte LDS slot only 16 entries apart -> a table read is at most 2-way conflicted.  Table B sits
// more than 2040 bytes behind table A so that the compiler cannot merge the two 8-byte reads of a pair into one
// ds_read2_b64 (8 LDS cycles per wave instead of 2 + 2).
// Tables A' / B' hold the same selectors with the two pooling COLUMNS swapped ({position 1, 0} / {3, 2}): lanes whose
// pooled pixel has an odd column index expand through them and store their first record one column to the right, their
// second one to the left -- two neighbouring pooled pixels (128 bytes apart in a 64-byte-record image, i.e. on the same
// 32 store banks) then hit opposite halves of the bank window: every expansion ds_write_b128 was 2-way conflicted.
// (A' is shifted by 13 entries against A inside the 256-byte bank window: an A read and an A' read of the same half-wave
// then collide only for entry pairs (e, e + 13), not for equal entries -- "both channels dead" is by far the most common.)
constexpr int C2T_SWAP = 256 + 104, C2T_C = 576, C2T_B = 2304, C2T_BYTES = (C2T_B + C2T_SWAP + 256 + 15) / 16 * 16;
static_assert(C2T_BYTES % 16 == 0, "what follows the tables in LDS is read with 16-byte accesses (a misaligned "
                                   "ds_read_b128 is replayed at 64 cycles per wave instruction: the kernel ran 45 % slower)");
__device__ __forceinline__ void code2_tables_init(uint32_t* tab) {
  const int e = threadIdx.x;
  if (e < 25) {
    const uint32_t c0 = e % 5, c1 = e / 5;
    auto sel = [&](uint32_t pos) { return (c0 == pos ? 0x0100u : 0x0c0cu) | (c1 == pos ? 0x03020000u : 0x0c0c0000u); };
    tab[2 * e] = sel(0); tab[2 * e + 1] = sel(1);
    tab[C2T_B / 4 + 2 * e] = sel(2); tab[C2T_B / 4 + 2 * e + 1] = sel(3);
    tab[C2T_SWAP / 4 + 2 * e] = sel(1); tab[C2T_SWAP / 4 + 2 * e + 1] = sel(0);
    tab[(C2T_B + C2T_SWAP) / 4 + 2 * e] = sel(3); tab[(C2T_B + C2T_SWAP) / 4 + 2 * e + 1] = sel(2);
    tab[C2T_C / 4 + 2 * e] = (c0 != 4 ? 0x0100u : 0x0c0cu) | (c1 != 4 ? 0x03020000u : 0x0c0c0000u);
    tab[C2T_C / 4 + 2 * e + 1] = 0x0c0c0c0cu;
  }
}
// the four position-masked copies of one packed channel pair gw whose pair byte is `off8` (already a table offset)
__device__ __forceinline__ void code2_expand_pair(const unsigned char* tab, uint32_t off8, uint32_t gw, uint32_t (&out)[4]) {
  const u32x2 sa = *(const u32x2*)(tab + off8), sb = *(const u32x2*)(tab + C2T_B + off8);
  out[0] = __builtin_amdgcn_perm(0u, gw, sa[0]);
  out[1] = __builtin_amdgcn_perm(0u, gw, sa[1]);
  out[2] = __builtin_amdgcn_perm(0u, gw, sb[0]);
  out[3] = __builtin_amdgcn_perm(0u, gw, sb[1]);
}
// fp32 kernels: the two channel codes of a pair byte
__device__ __forceinline__ void code2_pair_codes(uint32_t byte8, uint32_t& c_even, uint32_t& c_odd) {
  const uint32_t n = byte8 >> 3;            // 0..24
  c_odd = (n * 13u) >> 6;                   // n / 5
  c_even = n - 5u * c_odd;
}

constexpr int BD_COLS = 64;
constexpr int BD_RING = 8;                     // conv rows in the LDS ring
constexpr int BD_WPX = BD_COLS + 4;            // stored columns: band column cl = -1 .. 66 lives at index cl + 1
constexpr int BD_NPC = BD_COLS / 2 + 2;        // pooled columns touching the band (c0/2 - 1 .. c0/2 + 32)
constexpr int BD_ITEMS = 2 * BD_NPC;           // pooled pixels expanded per step
constexpr int BD_DCIT = (BD_ITEMS * 4 + 255) / 256;
#ifndef GDM_BD_XW
#define GDM_BD_XW 256
#endif
constexpr int BD_XW = GDM_BD_XW;               // x-window row stride in LDS (>= BD_XCOLS = 136, multiple of 4)
constexpr int BD_XCOLS = 2 * BD_COLS + 8;      // x-window columns 2c0-4 .. 2c0+131 (16-byte aligned start)
// bf16 FUSE ("MF") epilogue: conv1's weight gradient is one more MFMA product (see conv2_bwd_data_kernel).  The input
// window lives in LDS as four bf16 planes [hi|lo part][kw] of XROWS rows: plane(kw)[row][m] = x[row][m + 3 + kw], so
// that the 8 window values a lane needs for its 4 pixels x 2 pooling columns are ONE aligned 16-byte read.  Row stride
// 288 B and plane stride = 64 (mod 256) B put the 16 (plane, kh, lane group) combinations of a read on 16 distinct
// 16-byte slots of the 256-byte bank row.
constexpr int XP_ROW = 144;                     // bf16 elements per plane row (128 used)
constexpr int XP_PLANE = XROWS * XP_ROW + 16;   // elements (2624 B)
constexpr int XP_ONES = 8 * XP_ROW;             // a block of 1.0: the B operand column that sums the bias gradient
constexpr int XP_ELEMS = 4 * XP_PLANE + XP_ONES;
template <typename T> struct BD {
  static constexpr int DC_ELEMS = BD_RING * BD_WPX * C2<T>::S32;
  static constexpr size_t TAB = sizeof(T) == 2 ? C2T_BYTES : 0;      // code2 selector tables (bf16)
  static constexpr size_t lds_bytes(bool fuse) {
    if (!fuse) return (size_t)(DC_ELEMS + C2<T>::WB_ELEMS) * sizeof(T) + TAB;
    const size_t xbytes = sizeof(T) == 2 ? (size_t)XP_ELEMS * 2 + 64 : (size_t)(XROWS * BD_XW) * 4;
    return (size_t)(DC_ELEMS + C2<T>::WB_ELEMS) * sizeof(T) + TAB + xbytes + (size_t)(4 * 80) * 4;
  }
};

// Everything a workgroup fetches from HBM for one step, held in registers between "issue" and "consume".
template <typename T, bool FUSE, bool XVEC> struct BdStepRegs {
  static constexpr int XIT = FUSE ? (XVEC ? (XROWS * (BD_XCOLS / 4) + 255) / 256 : (XROWS * BD_XCOLS + 255) / 256) : 1;
  static constexpr bool MF = FUSE && sizeof(T) == 2;
  f32x4 g[BD_DCIT][sizeof(T) == 2 ? 1 : 2];
  uint32_t cd[BD_DCIT];                       // four pair bytes = the item's 8 channels
  f32x4 xv4[XVEC ? XIT : 1];
  float xv[XVEC ? (MF ? XIT : 1) : XIT];     // XVEC && MF: the window value left of each vector (column bc - 1)
  uint64_t codes[FUSE ? 4 : 1];
};

struct BdRsrc {
  rsrc_t dp2, code2, dp1, code1;
};

// Per-thread constants of the staging pattern (they depend on the lane only, never on the step): computed once so
// that issuing a step's loads costs a handful of VALU per load.
template <bool FUSE, bool XVEC> struct BdLane {
  static constexpr int XIT = FUSE ? (XVEC ? (XROWS * (BD_XCOLS / 4) + 255) / 256 : (XROWS * BD_XCOLS + 255) / 256) : 1;
  uint32_t dc_off[BD_DCIT];      // (prow * W2 + pcol) * 32 + 8 og
  int dc_prow[BD_DCIT], dc_pcol[BD_DCIT];   // prow = 99 marks a lane without an item
  uint32_t x_off[XIT];           // (br * W + bc) * 4
  int x_br[XIT], x_bc[XIT];      // br = 99 marks a lane without an element
};

template <typename T, bool FUSE, bool XVEC>
__device__ __forceinline__ void bd_issue(BdStepRegs<T, FUSE, XVEC>& rg, const BdLane<FUSE, XVEC>& ln, int b, int c0,
                                         int rq, const BdRsrc& rs, int H1, int W1, int H2, int W2,
                                         const float* __restrict__ x0, const float* __restrict__ x1, int bsplit, int H,
                                         int W) {
  constexpr bool MF = FUSE && sizeof(T) == 2;
  const int t = threadIdx.x, lr = t & 15, lg = (t >> 4) & 3, wv = t >> 6;
  const int pr0 = 2 * rq + 1, pc0 = (c0 >> 1) - 1;
  const uint32_t base = (uint32_t)b * H2 * W2 * 32 + (uint32_t)(pr0 * W2 + pc0) * 32;   // may wrap: only used when valid
#pragma unroll
  for (int k = 0; k < BD_DCIT; ++k) {
    const bool ok = (unsigned)(pr0 + ln.dc_prow[k]) < (unsigned)H2 && (unsigned)(pc0 + ln.dc_pcol[k]) < (unsigned)W2;
    const uint32_t gi = base + ln.dc_off[k];
    rg.cd[k] = __builtin_bit_cast(uint32_t, buf_load4(rs.code2, ok ? gi >> 1 : BUF_OOB));
    rg.g[k][0] = buf_load16(rs.dp2, ok ? gi * (uint32_t)sizeof(T) : BUF_OOB);
    if constexpr (sizeof(T) == 4) rg.g[k][1] = buf_load16(rs.dp2, ok ? gi * 4u + 16u : BUF_OOB);
  }
  if constexpr (FUSE) {
    // the input image of sample b lives in one of two tensors (real | generated): one descriptor per step
    const float* xb = (b < bsplit) ? x0 + (int64_t)b * H * W : x1 + (int64_t)(b - bsplit) * H * W;
    const rsrc_t xr_ = make_rsrc(xb, (uint32_t)H * W * 4);
    const int xr0 = 2 * ROWS * rq - 1, xc0 = 2 * c0 - 4;
    const uint32_t xbase = (uint32_t)(xr0 * W + xc0) * 4u;
#pragma unroll
    for (int k = 0; k < BdLane<FUSE, XVEC>::XIT; ++k) {
      // XVEC: W % 4 == 0, a 4-column vector is inside or outside the image as a whole.  Row and column validity are
      // merged ARITHMETICALLY (OR of the out-of-range bit): with `row_ok && col_ok` shared by two loads the compiler
      // turned the row test into a branch around them and put s_waitcnt vmcnt(0) in front of the second version of each
      // load -- every step then waited for everything in flight, the look-ahead was gone.
      const uint32_t rbad = (unsigned)(xr0 + ln.x_br[k]) < (unsigned)H ? 0u : BUF_OOB;
      const uint32_t off = ((unsigned)(xc0 + ln.x_bc[k]) < (unsigned)W ? xbase + ln.x_off[k] : BUF_OOB) | rbad;
      if constexpr (XVEC) {
        rg.xv4[k] = buf_load16<GDM_IN_LOAD_AUX>(xr_, off);
        if constexpr (MF) {
          // the kw = 0 plane is the kw = 1 plane shifted by one column: each vector also needs its left neighbour
          const uint32_t offm = ((unsigned)(xc0 + ln.x_bc[k] - 1) < (unsigned)W ? xbase + ln.x_off[k] - 4u : BUF_OOB) | rbad;
          rg.xv[k] = buf_load4<GDM_IN_LOAD_AUX>(xr_, offm);
        }
      } else {
        rg.xv[k] = buf_load4<GDM_IN_LOAD_AUX>(xr_, off);
      }
    }
    const int Q1 = (W1 + 3) >> 2;
    if constexpr (MF) {
      // lane (channel lr, pixel quad lg of this wave's 16 columns): the four pixels' fields of channel group lr / 4
      const int quad = (c0 >> 2) + 4 * wv + lg;
      const uint32_t cbase = (((uint32_t)(b * H1 + ROWS * rq) * Q1 + quad) * 4u + (uint32_t)(lr >> 2)) * 8u;
#pragma unroll
      for (int ir = 0; ir < 4; ++ir) {
        const bool ok = (unsigned)(ROWS * rq + ir) < (unsigned)H1 && quad < Q1;
        rg.codes[ir] = buf_load8<GDM_IN_LOAD_AUX>(rs.code1, ok ? cbase + (uint32_t)(ir * Q1) * 32u : BUF_OOB);
      }
    } else {
      // lane (pixel lr, channel group lg): one 16-bit field per row, rows 4rq .. 4rq+3
      const int iw = c0 + 16 * wv + lr;
#pragma unroll
      for (int ir = 0; ir < 4; ++ir) {
        const bool ok = (unsigned)(ROWS * rq + ir) < (unsigned)H1 && iw < W1;
        const uint32_t fi = code1_field((uint32_t)(b * H1 + ROWS * rq + ir), Q1, iw, lg) * 2u;
        rg.codes[ir] = (uint64_t)__builtin_amdgcn_raw_buffer_load_b16(rs.code1, ok ? fi : BUF_OOB, 0, GDM_IN_LOAD_AUX);
      }
    }
  }
}

// Registers of step rq -> ring rows 4rq+2 .. 4rq+5.  Lane = (pooled pixel, 8-channel group): four 16-byte records.
// (OOB items loaded zeros: pair byte 0 = "both channels at position 0" of a zero gradient -> zeros everywhere.)
// slot0 = ring row of the step's first new conv row (even), RING = rows in the ring.
template <typename T, bool FUSE, bool XVEC, int RING = BD_RING>
__device__ __forceinline__ void bd_expand(const BdStepRegs<T, FUSE, XVEC>& rg, const BdLane<FUSE, XVEC>& ln, int slot0,
                                          T* __restrict__ dc_s, const unsigned char* __restrict__ tab) {
  constexpr int S32 = C2<T>::S32;
  const int og = threadIdx.x & 3;
#pragma unroll
  for (int k = 0; k < BD_DCIT; ++k) {
    const int prow = ln.dc_prow[k], pcol = ln.dc_pcol[k];
    if (prow > 1) continue;
    int slot = slot0 + 2 * prow;                                         // even: slot + 1 never wraps
    if (slot >= RING) slot -= RING;
    const uint32_t cd = rg.cd[k];
    if constexpr (sizeof(T) == 2) {
      const u32x4 gv = __builtin_bit_cast(u32x4, rg.g[k][0]);
      uint32_t ex[4][4];                                                  // [channel pair][position]
      const int odd = pcol & 1;                                           // odd pooled column: columns swapped (tables A'/B')
      const unsigned char* tab_l = tab + (odd ? C2T_SWAP : 0);
#pragma unroll
      for (int w = 0; w < 4; ++w) code2_expand_pair(tab_l, (cd >> (8 * w)) & 0xffu, gv[w], ex[w]);
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int sc = 2 * pcol + (dx ^ odd);                                 // stored column
        const int piece = og ^ ((sc >> 1) & 2);                               // swizzled 16-byte slot of the record
        T* dst = dc_s + (slot * BD_WPX + sc) * S32 + 8 * piece;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
          *(u32x4*)(dst + dy * BD_WPX * S32) = (u32x4){ex[0][2 * dy + dx], ex[1][2 * dy + dx], ex[2][2 * dy + dx],
                                                       ex[3][2 * dy + dx]};
      }
    } else {
      float g[8];
      uint32_t c[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        g[e] = rg.g[k][0][e]; g[4 + e] = rg.g[k][1][e];
        code2_pair_codes((cd >> (8 * e)) & 0xffu, c[2 * e], c[2 * e + 1]);
      }
#pragma unroll
      for (int pos = 0; pos < 4; ++pos) {
        T* dst = dc_s + ((slot + (pos >> 1)) * BD_WPX + 2 * pcol + (pos & 1)) * S32 + 8 * og;
#pragma unroll
        for (int e = 0; e < 8; ++e) dst[e] = (int)c[e] == pos ? g[e] : 0.f;
      }
    }
  }
}

// bf16 FUSE ("MF"): the data-gradient MFMA is issued with its operands SWAPPED (A = gradient fragments: rows = pixels,
// B = weights: columns = input channels), so a lane of the result holds ONE channel (lr) and FOUR neighbouring pixels
// (4 lg + r) of each of the step's four rows.  That is the A-operand layout of one more MFMA product,
//     S[c][n] += sum_k A[c][k] * X[k][n],      k = (pixel 4lg + r, pooling column dx)  for a fixed (row ir, pooling row dy)
//     A[c][k] = dp1[c][pixel] if channel c of that pixel is live and its argmax is (dy, dx), else 0
//     X[k][n] = x[2 ih + dy - 1 + kh][2 iw + dx - 1 + kw]  for n = tap (kh, kw) -- 8 CONSECUTIVE window values,
// i.e. conv1's weight gradient as the weight gradient of the full-resolution convolution (K = full-resolution pixels,
// one non-zero per pooling window): 8 MFMAs per wave and step.  A is built from the accumulators with one cvt_pk, one
// 8-byte LDS table read (argmax code -> two v_perm selectors) and two v_perm per value; X is one aligned 16-byte read
// from the bf16 planes (columns 0-3 of the result: high parts of x, 4-7: low parts -- x stays exact to 2^-17 --,
// column 8: a block of ones = the bias gradient).  The round-1/2 epilogue (position-dependent 2x2 gathers from an fp32
// window + 20 FMAs per value: 58 of the kernel's 117 us, 37 % of its LDS cycles bank conflicts) is kept for fp32 only.
// DP1: the data gradient itself is written out (always without FUSE; with FUSE only for the module's input-gradient
// path -- the training step never needs it, and as a run-time test the 16 stores stayed in the step body).
template <typename T, bool FUSE, bool XVEC, bool DP1>
__global__ __launch_bounds__(256) void conv2_bwd_data_kernel(const T* __restrict__ dp2,
                                                             const uint8_t* __restrict__ code2,
                                                             const T* __restrict__ wb, int B, int H1, int W1, int H2,
                                                             int W2, int n_ctiles, int nseg, int seg_len,
                                                             int n_strips, T* __restrict__ dp1,
                                                             const uint16_t* __restrict__ code1,
                                                             const float* __restrict__ x0,
                                                             const float* __restrict__ x1, int bsplit, int H, int W,
                                                             float* __restrict__ slabs) {
  constexpr int S32 = C2<T>::S32, KP = C2<T>::KPB, XW = BD_XW;
  constexpr bool MF = FUSE && sizeof(T) == 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
  T* dc_s = (T*)dyn_smem;
  T* w_s = dc_s + BD<T>::DC_ELEMS;
  unsigned char* tab_s = (unsigned char*)(w_s + C2<T>::WB_ELEMS);     // bf16: code2 selector tables
  float* x_s = (float*)(tab_s + BD<T>::TAB);              // FUSE, fp32: [XROWS][XW]
  __bf16* xp_s = (__bf16*)(tab_s + BD<T>::TAB);           // MF: four planes + ones block, then the selector table
  uint32_t* tbl_s = (uint32_t*)(xp_s + XP_ELEMS);         // MF: 8 x {selector for dy = 0, selector for dy = 1}
  float* red = MF ? (float*)(tbl_s + 16) : x_s + XROWS * XW;   // FUSE: [4][80]
  const int t = threadIdx.x, l = t & 63, wv = t >> 6, lr = l & 15, lg = l >> 4;
  const int nrq = (H1 + ROWS - 1) / ROWS, G = gridDim.x;

  float a1[MF ? 1 : 4][4], bs[4];
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f};                         // MF: S[c = 4lg + r][n = lr]
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    bs[r] = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) a1[MF ? 0 : r][q] = 0.f;
  }
  STAMP_DECL;
  BdRsrc rs;
  rs.dp2 = make_rsrc(dp2, (uint32_t)B * H2 * W2 * 32 * sizeof(T));
  rs.code2 = make_rsrc(code2, (uint32_t)B * H2 * W2 * 16);
  rs.dp1 = make_rsrc(dp1, DP1 ? (uint32_t)B * H1 * W1 * 16 * sizeof(T) : 0u);
  rs.code1 = make_rsrc(code1, FUSE ? (uint32_t)B * H1 * ((W1 + 3) >> 2) * 32 : 0u);
  copy_to_lds(w_s, wb, C2<T>::WB_ELEMS);
  if constexpr (sizeof(T) == 2) code2_tables_init((uint32_t*)tab_s);
  if constexpr (MF) {
    // argmax code (bits 1:0 position, bit 2 live) -> v_perm selectors that place the bf16 gradient (bytes 0,1 of the
    // source) in the low (dx = 0) or high (dx = 1) half of the A dword of pooling row dy, or nowhere (0x0c = 0x00)
    if (t < 8) {
      const uint32_t none = 0x0c0c0c0cu, lo = 0x0c0c0100u, hi = 0x01000c0cu;
      const bool live = t >= 4;
      const int pos = t & 3;
      tbl_s[2 * t] = (live && (pos >> 1) == 0) ? ((pos & 1) ? hi : lo) : none;
      tbl_s[2 * t + 1] = (live && (pos >> 1) == 1) ? ((pos & 1) ? hi : lo) : none;
    }
    for (int i = t; i < XP_ONES / 2; i += 256) ((uint32_t*)(xp_s + 4 * XP_PLANE))[i] = 0x3f803f80u;   // bf16 1.0 pairs
    // plane cells no step ever writes (row padding, columns the window does not reach) must hold finite values:
    // they are read by lanes whose result columns are discarded
    for (int i = t; i < 4 * XP_PLANE / 2; i += 256) ((uint32_t*)xp_s)[i] = 0u;
  }

  BdLane<FUSE, XVEC> ln;
#pragma unroll
  for (int k = 0; k < BD_DCIT; ++k) {
    const int i = t + 256 * k, item = i >> 2, og = i & 3;
    const bool has = item < BD_ITEMS;
    ln.dc_prow[k] = has ? item / BD_NPC : 99;
    ln.dc_pcol[k] = item % BD_NPC;
    ln.dc_off[k] = (uint32_t)((item / BD_NPC) * W2 + item % BD_NPC) * 32 + 8 * og;
  }
  if constexpr (FUSE) {
#pragma unroll
    for (int k = 0; k < BdLane<FUSE, XVEC>::XIT; ++k) {
      const int i = t + 256 * k;
      constexpr int PER_ROW = XVEC ? BD_XCOLS / 4 : BD_XCOLS;
      const int br = i / PER_ROW, bc = (i % PER_ROW) * (XVEC ? 4 : 1);
      ln.x_br[k] = br < XROWS ? br : 99;
      ln.x_bc[k] = bc;
      ln.x_off[k] = (uint32_t)(br * W + bc) * 4u;
    }
  }

  // work item s = (image b, row segment seg, column tile ct); a segment is seg_len steps and starts with a pseudo step
  auto place = [&](int s_, int& b_, int& c0_, int& rq_first, int& rq_end) {
    const int ct = s_ % n_ctiles, sg = (s_ / n_ctiles) % nseg;
    b_ = s_ / (n_ctiles * nseg);
    c0_ = ct * BD_COLS;
    rq_first = sg * seg_len;
    rq_end = min(rq_first + seg_len, nrq);
  };
  // Step positions.  The loads of a step are issued TWO steps ahead into one of two register sets: with one step of
  // look-ahead a step could not be shorter than one HBM round trip under load (~2 us) -- halving the step's VALU work
  // (round 3: 418 -> 270 instructions per wave) did not move the kernel by a microsecond until the look-ahead doubled.
  struct Pos { int s, b, c0, rq, rq_first, rq_end; };
  auto advance = [&](const Pos& q) {
    Pos n = q;
    if (q.s >= n_strips) return n;                          // past the end: keep re-reading the last rows
    n.rq = q.rq + 1;
    if (n.rq == q.rq_end) {
      n.s = q.s + G;
      if (n.s < n_strips) {
        place(n.s, n.b, n.c0, n.rq_first, n.rq_end);
        n.rq = n.rq_first - 1;
      } else {
        n.rq = q.rq;                                        // nothing left: its loads re-read cache-hot rows
      }
    }
    return n;
  };
  // (fp32: one register set, one step of look-ahead -- a second set does not fit 256 VGPRs beside the 72 weight registers)
  constexpr bool AHEAD2 = sizeof(T) == 2;
  BdStepRegs<T, FUSE, XVEC> rg_a, rg_b;
  Pos p0, p1;
  p0.s = blockIdx.x;                                        // host guarantees gridDim.x <= n_strips
  place(p0.s, p0.b, p0.c0, p0.rq_first, p0.rq_end);
  p0.rq = p0.rq_first - 1;
  p1 = advance(p0);
  bd_issue<T, FUSE, XVEC>(rg_a, ln, p0.b, p0.c0, p0.rq, rs, H1, W1, H2, W2, x0, x1, bsplit, H, W);
  if constexpr (AHEAD2) bd_issue<T, FUSE, XVEC>(rg_b, ln, p1.b, p1.c0, p1.rq, rs, H1, W1, H2, W2, x0, x1, bsplit, H, W);
  __syncthreads();                                          // weight image complete

  // Wave w computes the 4 output rows of column tile w (16 columns).  A dc2 row fragment (one per tap column aw) feeds
  // the up to three output rows it touches, and the weight fragments of all nine taps stay in registers for the whole
  // kernel (bf16): 18 LDS fragment reads per 36 MFMAs.
  bf16x8 afr[sizeof(T) == 2 ? 9 : 1];
  float afw[sizeof(T) == 4 ? 72 : 1];
  int cb[3];
  if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int k = 0; k < 72; ++k) afw[k] = w_s[lr * KP + 4 * k + lg];
  }
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int ks = 0; ks < 9; ++ks) afr[ks] = *(const bf16x8*)&w_s[lr * KP + 32 * ks + 8 * lg];
#pragma unroll
    for (int aw = 0; aw < 3; ++aw) {
      const int sc = 16 * wv + lr + aw + 1;
      cb[aw] = sc * S32 + 8 * (lg ^ ((sc >> 1) & 2));
    }
  }
  // MF: per-lane constants of the epilogue
  //   xb_off: element offset of this lane's X fragment for (ir, dy) = (0, 0); result column n = lr: n < 4 high plane of
  //           tap (kh, kw) = (n >> 1, n & 1), 4..7 the low plane, >= 8 the block of ones (only column 8 is used)
  //   sh0/sh1: rotation that brings the lane's channel nibble of pixel r (even / odd: low / high half word) to bits 5:3
  const int xb_off = lr < 8 ? (2 * (lr >> 2) + (lr & 1)) * XP_PLANE + ((lr >> 1) & 1) * XP_ROW + 32 * wv + 8 * lg
                            : 4 * XP_PLANE;
  const uint32_t sh0 = 29u + 4u * (lr & 3), sh1 = sh0 + 16u;
  STAMP(6);
  auto step = [&](BdStepRegs<T, FUSE, XVEC>& rg, const Pos& cur, const Pos& nxt) {
    const int b = cur.b, c0 = cur.c0, rq = cur.rq, rq_first = cur.rq_first;
    // ---- consume the prefetched registers into the LDS images of this step
    STAMP(3);
#ifdef GDM_STAMPS
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");     // (stamp builds: this set's 12 loads have landed; the other set's 12 fly)
    STAMP(6);
#endif
    bd_expand<T, FUSE, XVEC>(rg, ln, (ROWS * rq + 2) & (BD_RING - 1), dc_s, tab_s);
    uint64_t codes[FUSE ? 4 : 1];
    if constexpr (FUSE) {
      if constexpr (MF) {
        // fp32 window values -> bf16 high / low parts in the two column-shifted planes
        auto split = [](float v0, float v1, uint32_t& hi, uint32_t& lo) {
          const bf16x2 h = {(__bf16)v0, (__bf16)v1};
          hi = __builtin_bit_cast(uint32_t, h);
          const float r0 = v0 - __builtin_bit_cast(float, hi << 16), r1 = v1 - __builtin_bit_cast(float, hi & 0xffff0000u);
          const bf16x2 lw = {(__bf16)r0, (__bf16)r1};
          lo = __builtin_bit_cast(uint32_t, lw);
        };
#pragma unroll
        for (int k = 0; k < BdLane<FUSE, XVEC>::XIT; ++k) {
          if constexpr (XVEC) {
            // vector = window columns bc .. bc+3 -> plane kw=1 cells m = bc-4 .. bc-1; with the left neighbour in
            // front (bc-1 .. bc+2) the same cells of plane kw=0
            if (ln.x_br[k] < XROWS && ln.x_bc[k] >= 4) {
              const f32x4 v = rg.xv4[k];
              const float vm = rg.xv[k];
              __bf16* cell = xp_s + ln.x_br[k] * XP_ROW + ln.x_bc[k] - 4;
              uint32_t h00, h01, l00, l01, h10, h11, l10, l11;
              split(vm, v[0], h00, l00);
              split(v[1], v[2], h01, l01);
              split(v[0], v[1], h10, l10);
              split(v[2], v[3], h11, l11);
              *(u32x2*)(cell) = (u32x2){h00, h01};
              *(u32x2*)(cell + XP_PLANE) = (u32x2){h10, h11};
              *(u32x2*)(cell + 2 * XP_PLANE) = (u32x2){l00, l01};
              *(u32x2*)(cell + 3 * XP_PLANE) = (u32x2){l10, l11};
            }
          } else {
            // one window value: cell m = bc - 3 of plane kw=0 and m = bc - 4 of plane kw=1
            if (ln.x_br[k] < XROWS && ln.x_bc[k] >= 3) {
              const float v = rg.xv[k];
              const __bf16 h = (__bf16)v, lw = (__bf16)(v - (float)h);
              __bf16* cell = xp_s + ln.x_br[k] * XP_ROW + ln.x_bc[k] - 3;
              cell[0] = h;
              cell[2 * XP_PLANE] = lw;
              if (ln.x_bc[k] >= 4) {
                cell[XP_PLANE - 1] = h;
                cell[3 * XP_PLANE - 1] = lw;
              }
            }
          }
        }
      } else {
#pragma unroll
        for (int k = 0; k < BdLane<FUSE, XVEC>::XIT; ++k) {
          if (ln.x_br[k] < XROWS) {
            if constexpr (XVEC) *(f32x4*)&x_s[ln.x_br[k] * XW + ln.x_bc[k]] = rg.xv4[k];
            else x_s[ln.x_br[k] * XW + ln.x_bc[k]] = rg.xv[k];
          }
        }
      }
#pragma unroll
      for (int ir = 0; ir < 4; ++ir) codes[ir] = rg.codes[ir];
    }
    STAMP(7);
    __syncthreads();
    STAMP(0);
    // this register set is free again: the loads of the step after next fly during this step and the next one; issued on
    // EVERY step (past the end they re-read cache-hot rows), so the step body has no branch around memory operations
    bd_issue<T, FUSE, XVEC>(rg, ln, nxt.b, nxt.c0, nxt.rq, rs, H1, W1, H2, W2, x0, x1, bsplit, H, W);
    STAMP(1);

    if (rq >= rq_first) {
      f32x4 acc[4];                                         // [output row of the step]
#pragma unroll
      for (int ir = 0; ir < 4; ++ir) acc[ir] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if constexpr (sizeof(T) == 2) {
        // band row rr (conv row 4rq-1+rr) contributes to output row ir = rr - ah with tap row ah; the fragments of
        // row rr+1 are read before the MFMAs of row rr issue.  Per accumulator the taps arrive in ascending order.
        bf16x8 bb[2][3];
        auto row_frags = [&](int rr, bf16x8 (&bx)[3]) {
          const int ro = ((ROWS * rq - 1 + rr) & (BD_RING - 1)) * BD_WPX * S32;
#pragma unroll
          for (int aw = 0; aw < 3; ++aw) bx[aw] = *(const bf16x8*)&dc_s[ro + cb[aw]];
        };
        row_frags(0, bb[0]);
#pragma unroll
        for (int rr = 0; rr < ROWS + 2; ++rr) {
          if (rr + 1 < ROWS + 2) row_frags(rr + 1, bb[(rr + 1) & 1]);
#pragma unroll
          for (int ah = 2; ah >= 0; --ah) {
            const int ir = rr - ah;
            if (ir < 0 || ir >= ROWS) continue;
#pragma unroll
            for (int aw = 0; aw < 3; ++aw) {
              // MF: operands swapped -> result row = pixel 4lg + r, column = input channel lr (same sums, transposed)
              if constexpr (MF) acc[ir] = mfma16(bb[rr & 1][aw], afr[3 * ah + aw], acc[ir]);
              else acc[ir] = mfma16(afr[3 * ah + aw], bb[rr & 1][aw], acc[ir]);
            }
          }
        }
      } else {
        // exact-fp32 mode: the weight fragments of all nine taps live in registers (72 per lane, loaded once per
        // kernel), so a v_mfma_f32_16x16x4_f32 costs one LDS read (the gradient value, shared by up to three output
        // rows) instead of two; the eight reads of a (row, tap column) are issued together ahead of their MFMAs
#pragma unroll
        for (int rr = 0; rr < ROWS + 2; ++rr) {
          const int ro = ((ROWS * rq - 1 + rr) & (BD_RING - 1)) * BD_WPX;
#pragma unroll
          for (int aw = 0; aw < 3; ++aw) {
            float bb[8];
#pragma unroll
            for (int o4 = 0; o4 < 8; ++o4) bb[o4] = dc_s[(ro + 16 * wv + lr + aw + 1) * S32 + 4 * o4 + lg];
#pragma unroll
            for (int o4 = 0; o4 < 8; ++o4)
#pragma unroll
              for (int ah = 2; ah >= 0; --ah) {
                const int ir = rr - ah;
                if (ir < 0 || ir >= ROWS) continue;
                acc[ir] = mfma16(afw[(3 * ah + aw) * 8 + o4], bb[o4], acc[ir]);
              }
          }
        }
      }
      STAMP(2);
      if constexpr (MF) {
        // C layout (swapped): col (lr) = input channel ci, row (4*lg + r) = pixel of the wave's 16 columns
        if constexpr (DP1) {
#pragma unroll
          for (int ir = 0; ir < 4; ++ir) {
            const int ih = ROWS * rq + ir;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int iw = c0 + 16 * wv + 4 * lg + r;
              const bool ok = ih < H1 && iw < W1;
              const uint32_t di = (uint32_t)((b * H1 + ih) * W1 + iw) * 16 + lr;
              const __bf16 v = (__bf16)acc[ir][r];
              buf_store2(rs.dp1, ok ? di * 2u : BUF_OOB, (uint32_t)__builtin_bit_cast(unsigned short, v));
            }
          }
        }
        const unsigned char* xp_b = (const unsigned char*)xp_s + 2 * xb_off;
        const unsigned char* tb = (const unsigned char*)tbl_s;
        u32x2 sel[2][4];
        auto selectors = [&](int ir, u32x2 (&so)[4]) {
          const uint32_t clo = (uint32_t)codes[ir], chi = (uint32_t)(codes[ir] >> 32);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const uint32_t word = r < 2 ? clo : chi;
            // rotate right by (nibble offset - 3) mod 32 (the instruction takes the shift modulo 32)
            const uint32_t idx8 = __builtin_amdgcn_alignbit(word, word, (r & 1) ? sh1 : sh0) & 0x38u;
            so[r] = *(const u32x2*)(tb + idx8);
          }
        };
        selectors(0, sel[0]);
#pragma unroll
        for (int ir = 0; ir < 4; ++ir) {
          if (ir + 1 < 4) selectors(ir + 1, sel[(ir + 1) & 1]);      // table reads of the next row fly under this one
          const bf16x8 xf0 = *(const bf16x8*)(xp_b + (2 * ir) * (XP_ROW * 2));
          const bf16x8 xf1 = *(const bf16x8*)(xp_b + (2 * ir + 1) * (XP_ROW * 2));
          u32x4 a0, a1v;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float gv = acc[ir][r];
            const bf16x2 gp = {(__bf16)gv, (__bf16)gv};
            const uint32_t gg = __builtin_bit_cast(uint32_t, gp);
            a0[r] = __builtin_amdgcn_perm(0u, gg, sel[ir & 1][r][0]);
            a1v[r] = __builtin_amdgcn_perm(0u, gg, sel[ir & 1][r][1]);
          }
          s1 = mfma16(__builtin_bit_cast(bf16x8, a0), xf0, s1);
          s1 = mfma16(__builtin_bit_cast(bf16x8, a1v), xf1, s1);
        }
      } else {
        // C layout: col (lr) = pixel, row (4*lg + r) = input channel ci
        const int iw = c0 + 16 * wv + lr;
        if constexpr (DP1) {
#pragma unroll
          for (int ir = 0; ir < 4; ++ir) {
            const int ih = ROWS * rq + ir;
            const bool ok = ih < H1 && iw < W1;
            const uint32_t di = (uint32_t)((b * H1 + ih) * W1 + iw) * 16 + 4 * lg;
            if constexpr (sizeof(T) == 2) {
              bf16x4 v;
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = (__bf16)acc[ir][r];
              buf_store8(rs.dp1, ok ? di * 2u : BUF_OOB, __builtin_bit_cast(uint64_t, v));
            } else {
              buf_store16(rs.dp1, ok ? di * 4u : BUF_OOB, acc[ir]);
            }
          }
        }
        if constexpr (FUSE) {
          // x window of pixel (row ir, column cl = 16 wv + lr) starts at x_s[2 ir][2 cl + 3]; position (dy, dx) moves it
          // by dy rows and dx columns: offset = dx + 256 dy = (pos * 129) & 0x101.  The 16 gathers of row ir+1 are issued
          // before the FMAs of row ir (register double buffer).
          const float* xcol = x_s + 2 * (16 * wv + lr) + 3;
          float xw[2][4][4];
          auto gather = [&](int ir, float (&xo)[4][4]) {
            const uint32_t pf = (uint32_t)codes[ir];                          // nibbles of channels 4lg..4lg+3
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const uint32_t pos = (pf >> (4 * r)) & 3u;
              const float* xp = xcol + 2 * ir * XW + (XW == 256 ? ((pos * 129u) & 0x101u) : (pos & 1u) + XW * (pos >> 1));
              xo[r][0] = xp[0]; xo[r][1] = xp[1]; xo[r][2] = xp[XW]; xo[r][3] = xp[XW + 1];
            }
          };
          gather(0, xw[0]);
#pragma unroll
          for (int ir = 0; ir < 4; ++ir) {
            if (ir + 1 < 4) gather(ir + 1, xw[(ir + 1) & 1]);
            const uint32_t lv = (uint32_t)codes[ir];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const uint32_t live = (uint32_t)((int32_t)(lv << (29 - 4 * r)) >> 31);        // bit 4r+2 -> 0 or ~0
              const float av = acc[ir][r];     // (bit_cast straight from a vector element reads element 0)
              const float g = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, av) & live);
#pragma unroll
              for (int q = 0; q < 4; ++q) a1[r][q] = fmaf(g, xw[ir & 1][r][q], a1[r][q]);
              bs[r] += g;
            }
          }
        }
      }
      STAMP(4);
    }
    __syncthreads();     // every wave is done with this step's LDS images
    STAMP(5);
  };
  while (p0.s < n_strips) {
    if constexpr (AHEAD2) {
      const Pos p2 = advance(p1);
      step(rg_a, p0, p2);
      const Pos p3 = advance(p2);
      if (p1.s < n_strips) step(rg_b, p1, p3);
      p0 = p2;
      p1 = p3;
    } else {
      step(rg_a, p0, p1);
      p0 = p1;
      p1 = advance(p1);
    }
  }
  STAMP_FLUSH;
  if constexpr (MF) {
    // S[c = 4lg + r][n = lr]: taps = columns 0..3 (+ their low-part twins 4..7), bias gradient = column 8
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = s1[r];
      const float tw = __shfl_down(v, 4, 64);          // lane lr + 4 of the same lane group (lr < 4: no wrap)
      if (lr < 4) red[wv * 80 + (4 * lg + r) * 4 + lr] = v + tw;
      if (lr == 8) red[wv * 80 + 64 + 4 * lg + r] = v;
    }
    __syncthreads();
    if (t < 80) slabs[(int64_t)blockIdx.x * 80 + t] = ((red[t] + red[80 + t]) + red[160 + t]) + red[240 + t];
  } else if constexpr (FUSE) {
    // one reduction per workgroup (not per tile): over the 16 lanes that share a channel group, then over the waves
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v = a1[r][q];
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
        if (lr == 0) red[wv * 80 + (4 * lg + r) * 4 + q] = v;
      }
      float v = bs[r];
      v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
      if (lr == 0) red[wv * 80 + 64 + 4 * lg + r] = v;
    }
    __syncthreads();
    if (t < 80) slabs[(int64_t)blockIdx.x * 80 + t] = ((red[t] + red[80 + t]) + red[160 + t]) + red[240 + t];
  }
}

// -------------------------------------------------------------------------------------- conv2 backward (weights)
// dW2[o][ci][kh][kw] = sum_{b,r,c} dc2[r][c][o] * p1[r-1+kh][c-1+kw][ci];  db2[o] = sum dc2
// GEMM with M = 32 (o), N = 9 taps x 16 ci, K = pixels.  Like the data-gradient kernel, a persistent workgroup walks
// strips (image, 64-column tile) top to bottom in steps of 4 conv rows; wave w contracts row w of the step.  p1 lives
// in an 8-row LDS ring (a step needs rows 4rq-1 .. 4rq+4, only 4rq+1 .. 4rq+4 are new), dc2 rows are rebuilt per step
// from the two pooled rows they come from; the next step's rows are fetched into registers during the MFMAs.  bf16
// reads both operands with the transposing ds_read_b64_tr_b16 (the contraction index is the pixel, the LDS images are
// channel-contiguous); both images are swizzled by the column's bit 3 -- a half-wave's tr16 read touches columns
// {c..c+3} and {c+8..c+11}, which without it fall on the same banks (with it: none left, checked exhaustively).
// Accumulators (2 x 9 tiles per wave) live in registers across all steps, then waves are summed through LDS in fixed
// order and the workgroup writes one slab.
__device__ __forceinline__ bf16x4 lds_tr16(const __bf16* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p);
}

constexpr int BW_RING = 8;                    // p1 rows in the LDS ring
constexpr int BW_WPX = 72;                    // p1 columns per ring row (band columns 0..65, padded to a multiple of 8)
template <typename T> struct BW {
  static constexpr int P_ELEMS = BW_RING * BW_WPX * C2<T>::S16;
  static constexpr int DC_ELEMS = ROWS * COLS * C2<T>::S32;
  static constexpr int P1IT = 3;              // 16-byte chunks of the 4 new rows: 2 x 256 (columns c0..c0+63) + halo
  // column swizzles (bf16 only): band column c of the p1 ring, 16-byte channel group og of dc2 column c
  static __device__ __forceinline__ int pcol(int c) { return sizeof(T) == 2 ? (c ^ (4 * ((c >> 3) & 1))) : c; }
  static __device__ __forceinline__ int dcpiece(int og, int c) { return sizeof(T) == 2 ? (og ^ (2 * ((c >> 3) & 1))) : og; }
};

template <typename T>
__global__ __launch_bounds__(256) void conv2_bwd_weight_kernel(const T* __restrict__ dp2,
                                                               const uint8_t* __restrict__ code2,
                                                               const T* __restrict__ p1, int B, int H1, int W1,
                                                               int H2, int W2, int n_ctiles, int nseg, int seg_len,
                                                               int n_items, float* __restrict__ slabs) {
  constexpr int S16 = C2<T>::S16, S32 = C2<T>::S32, WPX = BW_WPX;
  constexpr int PIECES = 2, EPP = 8;          // a p1 pixel record is staged as two 8-channel halves (16 B bf16, 32 B fp32)
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
  T* dc_s = (T*)dyn_smem;
  T* p_s = dc_s + BW<T>::DC_ELEMS;
  const unsigned char* tab_s = (const unsigned char*)(p_s + BW<T>::P_ELEMS);     // bf16: code2 selector tables
  if constexpr (sizeof(T) == 2) code2_tables_init((uint32_t*)(p_s + BW<T>::P_ELEMS));    // (first barrier: in the loop)
  const int t = threadIdx.x, l = t & 63, wv = t >> 6, lr = l & 15, lg = l >> 4;
  const int nrq = (H1 + ROWS - 1) / ROWS, G = gridDim.x;
  f32x4 acc[2][9];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int n = 0; n < 9; ++n) acc[i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bsum[e] = 0.f;

  const rsrc_t dp2r = make_rsrc(dp2, (uint32_t)B * H2 * W2 * 32 * sizeof(T));
  const rsrc_t code2r = make_rsrc(code2, (uint32_t)B * H2 * W2 * 16);
  const rsrc_t p1r = make_rsrc(p1, (uint32_t)B * H1 * W1 * 16 * sizeof(T));

  // ---- per-lane staging constants
  // dc2: lane = (pooled pixel item = t >> 2 of the 2 x 32 pooled block, 8-channel group og)
  const int og = t & 3, dprow = (t >> 2) >> 5, dpcol = (t >> 2) & 31;
  const uint32_t dc_goff = (uint32_t)(dprow * W2 + dpcol) * 32 + 8 * og;
  // p1: chunk k -> (new row 0..3, band column 0..65, piece); chunks 0..511 cover band columns 1..64, 512..527 the halo
  int pr_row[BW<T>::P1IT], pr_col[BW<T>::P1IT], pr_lds[BW<T>::P1IT];
  uint32_t pr_goff[BW<T>::P1IT];
#pragma unroll
  for (int k = 0; k < BW<T>::P1IT; ++k) {
    const int i = t + 256 * k;
    int row, col, piece;
    if (k < 2) {                                  // 4 rows x 64 columns x 2 halves = 512 chunks
      piece = i % PIECES; col = 1 + (i / PIECES) % 64; row = i / (PIECES * 64);
    } else {
      const int j = i - 512;                      // halo columns 0 and 65
      piece = j % PIECES; col = ((j / PIECES) & 1) ? 65 : 0; row = j / (2 * PIECES);
    }
    const bool has = row < 4;
    pr_row[k] = has ? row : 99;
    pr_col[k] = col;
    pr_goff[k] = (uint32_t)((row * W1 + col) * 16 + piece * EPP) * (uint32_t)sizeof(T);
    pr_lds[k] = BW<T>::pcol(col) * S16 + piece * EPP;       // + slot * WPX * S16
  }

  struct Regs {
    f32x4 g[sizeof(T) == 2 ? 1 : 2];
    uint32_t cd;
    f32x4 p[(sizeof(T) == 2 ? 1 : 2) * BW<T>::P1IT];
  } rg;
  auto issue = [&](int b, int c0, int rq, bool with_dc) {
    {   // dc2: pooled rows 2rq, 2rq+1, pooled columns c0/2 .. c0/2+31
      const int pr = 2 * rq + dprow, pc = (c0 >> 1) + dpcol;
      const bool ok = with_dc && pr >= 0 && pr < H2 && pc < W2;
      const uint32_t gi = (uint32_t)b * H2 * W2 * 32 + (uint32_t)(2 * rq * W2 + (c0 >> 1)) * 32 + dc_goff;
      rg.cd = __builtin_bit_cast(uint32_t, buf_load4(code2r, ok ? gi >> 1 : BUF_OOB));
      rg.g[0] = buf_load16(dp2r, ok ? gi * (uint32_t)sizeof(T) : BUF_OOB);
      if constexpr (sizeof(T) == 4) rg.g[1] = buf_load16(dp2r, ok ? gi * 4u + 16u : BUF_OOB);
    }
    // p1: rows 4rq+1 .. 4rq+4, band columns 0..65 = image columns c0-1 .. c0+64
    const int r0 = ROWS * rq + 1, cc0 = c0 - 1;
    const uint32_t base = (uint32_t)((b * H1 + r0) * W1 + cc0) * 16 * (uint32_t)sizeof(T);
#pragma unroll
    for (int k = 0; k < BW<T>::P1IT; ++k) {
      const bool ok = (unsigned)(r0 + pr_row[k]) < (unsigned)H1 && (unsigned)(cc0 + pr_col[k]) < (unsigned)W1;
      rg.p[k] = buf_load16(p1r, ok ? base + pr_goff[k] : BUF_OOB);
      if constexpr (sizeof(T) == 4) {
        // fp32: a half record is 32 bytes -> second 16 bytes
        rg.p[BW<T>::P1IT + k] = buf_load16(p1r, ok ? base + pr_goff[k] + 16u : BUF_OOB);
      }
    }
  };

  auto place = [&](int s_, int& b_, int& c0_, int& rq_first, int& rq_end) {
    const int ct = s_ % n_ctiles, sg = (s_ / n_ctiles) % nseg;
    b_ = s_ / (n_ctiles * nseg);
    c0_ = ct * COLS;
    rq_first = sg * seg_len;
    rq_end = min(rq_first + seg_len, nrq);
  };
  int s = blockIdx.x, b, c0, rq_first, rq_end;            // host guarantees gridDim.x <= n_items
  place(s, b, c0, rq_first, rq_end);
  int rq = rq_first - 1;                                   // pseudo step: brings in p1 rows 4 rq_first - 1 and 4 rq_first
  STAMP_DECL;
  issue(b, c0, rq, false);
  STAMP(4);

  // per-lane fragment addresses that never change: swizzled p1 band columns for (segment, kw, lo/hi)
  int pcol_off[2][3][2];
  const int q = lr >> 2, p4 = lr & 3;
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int sg2 = 0; sg2 < 2; ++sg2)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int hi = 0; hi < 2; ++hi)
          pcol_off[sg2][kw][hi] = BW<T>::pcol(32 * sg2 + 8 * lg + kw + q + 4 * hi) * S16 + 4 * p4;
  }

  while (s < n_items) {
#ifdef GDM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(5);
#endif
    // ---- registers -> LDS: dc2 rows of this step, p1 ring rows 4rq+1 .. 4rq+4
    if (rq >= rq_first) {
      const uint32_t cd = rg.cd;
      if constexpr (sizeof(T) == 2) {
        const u32x4 gv = __builtin_bit_cast(u32x4, rg.g[0]);
        uint32_t ex[4][4];                                   // [channel pair][position]
        const int odd = dpcol & 1;                           // odd pooled column: columns swapped (see Code2 tables)
        const unsigned char* tab_l = tab_s + (odd ? C2T_SWAP : 0);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const uint32_t off8 = (cd >> (8 * w)) & 0xffu;
          code2_expand_pair(tab_l, off8, gv[w], ex[w]);
          // bias gradient: channels whose pooled value was live
          const uint32_t g = __builtin_amdgcn_perm(0u, gv[w], *(const uint32_t*)(tab_s + C2T_C + off8));
          bsum[2 * w] += __builtin_bit_cast(float, g << 16);
          bsum[2 * w + 1] += __builtin_bit_cast(float, g & 0xffff0000u);
        }
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const int col = 2 * dpcol + (dx ^ odd);
          T* dst = dc_s + (2 * dprow * COLS + col) * S32 + 8 * BW<T>::dcpiece(og, col);
#pragma unroll
          for (int dy = 0; dy < 2; ++dy)
            *(u32x4*)(dst + dy * COLS * S32) = (u32x4){ex[0][2 * dy + dx], ex[1][2 * dy + dx], ex[2][2 * dy + dx],
                                                       ex[3][2 * dy + dx]};
        }
      } else {
        float g[8];
        uint32_t c[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          g[e] = rg.g[0][e]; g[4 + e] = rg.g[1][e];
          code2_pair_codes((cd >> (8 * e)) & 0xffu, c[2 * e], c[2 * e + 1]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) bsum[e] += c[e] < 4 ? g[e] : 0.f;
#pragma unroll
        for (int pos = 0; pos < 4; ++pos) {
          T* dst = dc_s + ((2 * dprow + (pos >> 1)) * COLS + 2 * dpcol + (pos & 1)) * S32 + 8 * og;
#pragma unroll
          for (int e = 0; e < 8; ++e) dst[e] = (int)c[e] == pos ? g[e] : 0.f;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < BW<T>::P1IT; ++k) {
      if (pr_row[k] > 3) continue;
      const int slot = (ROWS * rq + 1 + pr_row[k]) & (BW_RING - 1);
      T* dst = p_s + slot * WPX * S16 + pr_lds[k];
      if constexpr (sizeof(T) == 2) {
        *(f32x4*)dst = rg.p[k];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) { dst[e] = rg.p[k][e]; dst[4 + e] = rg.p[BW<T>::P1IT + k][e]; }
      }
    }
    STAMP(6);
    __syncthreads();
    STAMP(0);
    // ---- next step's loads (always issued: the very last step re-reads its own rows)
    int sn = s, bn = b, c0n = c0, rqn = rq + 1, rq_first_n = rq_first, rq_end_n = rq_end;
    if (rqn == rq_end) {
      sn = s + G;
      if (sn < n_items) {
        place(sn, bn, c0n, rq_first_n, rq_end_n);
        rqn = rq_first_n - 1;
      } else {
        rqn = rq;
      }
    }
    issue(bn, c0n, rqn, rqn >= rq_first_n);
    STAMP(1);

    if (rq >= rq_first) {
      // wave wv contracts row wv of the step (64 pixels = 2 bf16 k-steps / 16 fp32 k-steps)
      const int d = wv;
      if constexpr (sizeof(T) == 2) {
        int prow[3];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) prow[kh] = ((ROWS * rq + d - 1 + kh) & (BW_RING - 1)) * WPX * S16;
#pragma unroll
        for (int sgm = 0; sgm < COLS / 32; ++sgm) {
          const int cb = 32 * sgm + 8 * lg;   // this lane group's 8 pixels: cols cb .. cb+7 of row d
          bf16x8 a[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int pc8 = 16 * (i ^ (lg & 1)) + 4 * p4;     // swizzled 8-byte piece (columns cb+q, cb+4+q share bit 3)
            const bf16x4 lo = lds_tr16(&dc_s[(d * COLS + cb + q) * S32 + pc8]);
            const bf16x4 hi = lds_tr16(&dc_s[(d * COLS + cb + 4 + q) * S32 + pc8]);
            a[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          }
          bf16x8 bb[2];
          auto bfrag = [&](int tap) {
            const int kh = tap / 3, kw = tap % 3;
            const bf16x4 lo = lds_tr16(&p_s[prow[kh] + pcol_off[sgm][kw][0]]);
            const bf16x4 hi = lds_tr16(&p_s[prow[kh] + pcol_off[sgm][kw][1]]);
            return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          };
          bb[0] = bfrag(0);
#pragma unroll
          for (int tap = 0; tap < 9; ++tap) {      // the next tap's fragment is read before this tap's MFMAs issue
            if (tap + 1 < 9) bb[(tap + 1) & 1] = bfrag(tap + 1);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][tap] = mfma16(a[i], bb[tap & 1], acc[i][tap]);
          }
        }
      } else {
#pragma unroll 2
        for (int ks = 0; ks < COLS / 4; ++ks) {
          const int cpix = 4 * ks + lg;
          float a[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) a[i] = dc_s[(d * COLS + cpix) * S32 + 16 * i + lr];
#pragma unroll
          for (int tap = 0; tap < 9; ++tap) {
            const int kh = tap / 3, kw = tap % 3;
            const int slot = (ROWS * rq + d - 1 + kh) & (BW_RING - 1);
            const float bb = p_s[(slot * WPX + cpix + kw) * S16 + lr];
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][tap] = mfma16(a[i], bb, acc[i][tap]);
          }
        }
      }
    }
    STAMP(2);
    __syncthreads();   // this step's readers are done: the LDS images may be rebuilt
    STAMP(3);
    s = sn; b = bn; c0 = c0n; rq = rqn; rq_first = rq_first_n; rq_end = rq_end_n;
  }
  STAMP_FLUSH;
  // ---- cross-wave reduction in fixed order (wave 0 stores, waves 1..3 add in turn: each element is touched by the
  //      same lane position in every wave) and slab write.  slab layout: [o 32][tap 9][ci 16] then 32 bias sums.
  __syncthreads();
  float* red = (float*)dyn_smem;   // 4608 floats (+ 2048 for the bias sums)
#pragma unroll 1
  for (int turn = 0; turn < 4; ++turn) {
    if (wv == turn) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int idx = ((16 * i + 4 * lg + r) * 9 + tap) * 16 + lr;   // C row = o, C col = ci
            red[idx] = turn == 0 ? acc[i][tap][r] : red[idx] + acc[i][tap][r];
          }
    }
    __syncthreads();
  }
  float* slab = slabs + (int64_t)blockIdx.x * (4608 + 32);
  for (int i = t; i < 4608; i += 256) slab[i] = red[i];
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 8; ++e) red[t * 8 + e] = bsum[e];
  __syncthreads();
  if (t < 32) {
    const int og = t >> 3, e = t & 7;
    float s = 0.f;
    for (int k = 0; k < 64; ++k) s += red[(4 * k + og) * 8 + e];
    slab[4608 + t] = s;
  }
}


inline int conv1_slabs(int64_t total) {
  int64_t b = (total + 256 * 8 - 1) / (256 * 8);
  return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}
// conv2 backward-data work decomposition: items = (image, row segment, 64-column tile).  Whole-height strips when
// they fill the chip (2 resident workgroups per CU x 256 CUs); small batches are cut into row segments (each pays one
// pseudo step).  A pure function of the shapes: the slab count of the fused variant must be reproducible by _finish.
struct BdPlan { int n_ctiles, nseg, seg_len, n_items, blocks; };
// experiments only: grid caps of the persistent kernels from the environment (tools/ sweeps), else the tuned default
inline int tuned_cap(const char* name, int dflt) {
  const char* e = getenv(name);
  const int v = e ? atoi(e) : 0;
  return v > 0 ? v : dflt;
}
inline BdPlan bd_plan(int B, int H1, int W1, bool fuse) {
  BdPlan p;
  const int nrq = (H1 + ROWS - 1) / ROWS;
  p.n_ctiles = (W1 + BD_COLS - 1) / BD_COLS;
  const int64_t strips = (int64_t)B * p.n_ctiles;
  int nseg = (int)((512 + strips - 1) / strips);
  const int max_seg = nrq / 2 > 1 ? nrq / 2 : 1;            // at least two steps per segment
  nseg = nseg < 1 ? 1 : (nseg > max_seg ? max_seg : nseg);
  p.seg_len = (nrq + nseg - 1) / nseg;
  p.nseg = (nrq + p.seg_len - 1) / p.seg_len;
  p.n_items = (int)(strips * p.nseg);
  static const int cap_fuse = tuned_cap("GDM_BD_CAP", 512);
  const int cap = fuse ? cap_fuse : 768;      // (measured at 2B = 512: 384 -> 132 us, 512 -> 118, 640 -> 121, 768 -> 121)
  p.blocks = p.n_items < cap ? p.n_items : cap;
  return p;
}
// conv2 backward-weight: same items, 3 resident workgroups per CU, small batches cut finer
inline BdPlan bw_plan(int B, int H1, int W1) {
  BdPlan p;
  const int nrq = (H1 + ROWS - 1) / ROWS;
  p.n_ctiles = (W1 + COLS - 1) / COLS;
  const int64_t strips = (int64_t)B * p.n_ctiles;
  // measured at 2B = 512 (1024 strips): 768 workgroups (3 per CU) 73 us, 1024 (4 per CU) 84 us, 512 78 us; finer items
  // (more segments per strip) only add pseudo steps
  static const int cap = tuned_cap("GDM_BW_CAP", 768);
  static const int nseg_exp = tuned_cap("GDM_BW_NSEG", 0);          // experiments only
  int nseg = nseg_exp > 0 ? nseg_exp : (int)((1024 + strips - 1) / strips);
  const int max_seg = nrq / 2 > 1 ? nrq / 2 : 1;
  nseg = nseg < 1 ? 1 : (nseg > max_seg ? max_seg : nseg);
  p.seg_len = (nrq + nseg - 1) / nseg;
  p.nseg = (nrq + p.seg_len - 1) / p.seg_len;
  p.n_items = (int)(strips * p.nseg);
  p.blocks = p.n_items < cap ? p.n_items : cap;
  return p;
}

// the conv2 kernels address their tensors through 32-bit buffer offsets (see make_rsrc): largest tensor < 2 GiB
inline bool fits_buffer_addressing(int B, int H1, int W1) { return (int64_t)B * H1 * W1 * 16 * 4 < (int64_t)1 << 31; }

// Raise a kernel's dynamic-LDS limit once per kernel and process (not a stream operation: kept out of graph capture
// by doing it on the first, un-captured launch only; the size per kernel never changes).  Keyed by the kernel's
// address: template instantiations that share a signature share K.
template <typename K>
inline void allow_lds(K kernel, size_t bytes) {
  static const void* done[16];
  static int n_done = 0;
  for (int i = 0; i < n_done; ++i)
    if (done[i] == (const void*)kernel) return;
  (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (n_done < 16) done[n_done++] = (const void*)kernel;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------- C ABI
#define DISPATCH_T(dtype, CALL)                  \
  if ((dtype) == GDM_BF16) { using T = __bf16; CALL; } else { using T = float; CALL; }

extern "C" int gdm_simnn_conv1_fwd_pair(const float* x0, const float* x1, int bsplit, const float* w, const float* bias,
                                        int B, int H, int W, void* p1, uint64_t* code1, int dtype, void* stream) {
  GDM_REQUIRE(x0 && w && bias && p1 && code1, "gdm_simnn_conv1_fwd: null pointer");
  GDM_REQUIRE(B > 0 && H >= 1 && W >= 1 && gdm_dtype_ok(dtype), "gdm_simnn_conv1_fwd: bad arguments");
  GDM_REQUIRE(bsplit >= 1 && bsplit <= B && (bsplit == B || x1 != nullptr), "gdm_simnn_conv1_fwd: second input pointer missing");
  const int H1 = (H + 1) / 2, W1 = (W + 1) / 2;
  GDM_REQUIRE((int64_t)B * H1 * W1 * 64 < ((int64_t)1 << 31) && (int64_t)B * H * W * 4 < ((int64_t)1 << 31),
              "gdm_simnn_conv1_fwd: batch of %d %dx%d inputs exceeds 2 GiB per tensor", B, H, W);
  const int64_t n_rows = (int64_t)B * H1;                                // a wave walks whole pooled rows
  int64_t blocks = (n_rows + 3) / 4;                                     // 4 waves per workgroup
  static const int cap1 = tuned_cap("GDM_C1_CAP", 1536);               // persistent: 6 workgroups per CU (2048: +0.6 % per iteration)
  if (blocks > cap1) blocks = cap1;
  if (bsplit == B) x1 = x0;                                              // never read
  DISPATCH_T(dtype, hipLaunchKernelGGL(conv1_fwd_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                                       x0, x1, bsplit, w, bias, B, H, W, H1, W1, (int)n_rows, (T*)p1, (uint16_t*)code1));
  GDM_LAUNCH_OK("gdm_simnn_conv1_fwd");
  return GDM_OK;
}

extern "C" int gdm_simnn_conv1_fwd(const float* x, const float* w, const float* bias, int B, int H, int W, void* p1,
                                   uint64_t* code1, int dtype, void* stream) {
  return gdm_simnn_conv1_fwd_pair(x, nullptr, B, w, bias, B, H, W, p1, code1, dtype, stream);
}

extern "C" size_t gdm_simnn_conv1_bwd_weight_workspace_bytes(int B, int H, int W) {
  const int64_t total = (int64_t)B * ((H + 1) / 2) * ((W + 1) / 2);
  return (size_t)(conv1_slabs(total) + 65) * 80 * sizeof(float);
}

extern "C" int gdm_simnn_conv1_bwd_weight(const void* dp1, const uint64_t* code1, const float* x, int B, int H, int W,
                                          float* dw, float* db, int dtype, int accumulate, void* workspace,
                                          size_t workspace_bytes, void* stream) {
  GDM_REQUIRE(dp1 && code1 && x && dw && db, "gdm_simnn_conv1_bwd_weight: null pointer");
  GDM_REQUIRE(B > 0 && H >= 1 && W >= 1 && gdm_dtype_ok(dtype), "gdm_simnn_conv1_bwd_weight: bad arguments");
  if (!workspace || workspace_bytes < gdm_simnn_conv1_bwd_weight_workspace_bytes(B, H, W)) {
    gdm_set_error("gdm_simnn_conv1_bwd_weight: workspace too small");
    return GDM_EWORKSPACE;
  }
  const int H1 = (H + 1) / 2, W1 = (W + 1) / 2;
  const int nslabs = conv1_slabs((int64_t)B * H1 * W1);
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_T(dtype, hipLaunchKernelGGL(conv1_bwd_weight_kernel<T>, dim3(nslabs), dim3(256), 0, s, (const T*)dp1,
                                       (const uint16_t*)code1, x, B, H, W, H1, W1, (float*)workspace));
  float* scratch = (float*)workspace + (size_t)nslabs * 80;
  launch_slab_sum<1>((const float*)workspace, nslabs, 80, scratch, dw, db, accumulate, s);
  GDM_LAUNCH_OK("gdm_simnn_conv1_bwd_weight");
  return GDM_OK;
}

extern "C" int gdm_simnn_conv1_bwd_data(const void* dp1, const uint64_t* code1, const float* w, int B, int H, int W,
                                        float* dx, int dtype, void* stream) {
  GDM_REQUIRE(dp1 && code1 && w && dx, "gdm_simnn_conv1_bwd_data: null pointer");
  GDM_REQUIRE(B > 0 && H >= 1 && W >= 1 && gdm_dtype_ok(dtype), "gdm_simnn_conv1_bwd_data: bad arguments");
  const int H1 = (H + 1) / 2, W1 = (W + 1) / 2;
  const int64_t total = (int64_t)B * H * W;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  DISPATCH_T(dtype, hipLaunchKernelGGL(conv1_bwd_data_kernel<T>, dim3((unsigned)blocks), dim3(256), 0,
                                       (hipStream_t)stream, (const T*)dp1, (const uint16_t*)code1, w, B, H, W, H1, W1, dx));
  GDM_LAUNCH_OK("gdm_simnn_conv1_bwd_data");
  return GDM_OK;
}

extern "C" size_t gdm_simnn_conv2_pack_bytes(int dtype) {
  return dtype == GDM_BF16 ? (size_t)(C2<__bf16>::WF_ELEMS + C2<__bf16>::WB_ELEMS) * 2
                           : (size_t)(C2<float>::WF_ELEMS + C2<float>::WB_ELEMS) * 4;
}

extern "C" int gdm_simnn_conv2_pack(const float* w, int dtype, void* pack, void* stream) {
  GDM_REQUIRE(w && pack && gdm_dtype_ok(dtype), "gdm_simnn_conv2_pack: bad arguments");
  GDM_REQUIRE(((uintptr_t)pack & 15) == 0, "gdm_simnn_conv2_pack: pack buffer must be 16-byte aligned");
  DISPATCH_T(dtype, hipLaunchKernelGGL(conv2_pack_kernel<T>, dim3((C2<T>::WF_ELEMS + 255) / 256), dim3(256), 0,
                                       (hipStream_t)stream, w, (T*)pack, (T*)pack + C2<T>::WF_ELEMS));
  GDM_LAUNCH_OK("gdm_simnn_conv2_pack");
  return GDM_OK;
}

extern "C" int gdm_simnn_adam_step(float* p_big, const float* g_big_pc, float* m_big, float* v_big, int N, int C, int P,
                                   void* shadow_pc, float* p_small, const float* g_small, float* m_small, float* v_small,
                                   int n_small, const float* conv2_weight, void* pack, int dtype, float* hyper, int* done,
                                   void* stream) {
  GDM_REQUIRE(p_big && g_big_pc && m_big && v_big && shadow_pc && p_small && g_small && m_small && v_small && conv2_weight &&
              pack && hyper && done, "gdm_simnn_adam_step: null pointer");
  GDM_REQUIRE(N > 0 && C > 0 && P > 0 && n_small > 0 && gdm_dtype_ok(dtype), "gdm_simnn_adam_step: bad arguments");
  GDM_REQUIRE(conv2_weight >= p_small && conv2_weight + 4608 <= p_small + n_small,
              "gdm_simnn_adam_step: conv2.weight must lie inside the small-parameter range (it is re-packed from there)");
  const int tx = (P + 127) / 128, ty = (C + 31) / 32;
  const int64_t nbig = (int64_t)tx * ty * N;
  GDM_REQUIRE(nbig < ((int64_t)1 << 30), "gdm_simnn_adam_step: parameter too large");
  const int vec_ok = (P % 4 == 0) && ((((uintptr_t)p_big | (uintptr_t)m_big | (uintptr_t)v_big) & 15) == 0);
  const int small_vec_ok = ((((uintptr_t)p_small | (uintptr_t)g_small | (uintptr_t)m_small | (uintptr_t)v_small) & 15) == 0);
  DISPATCH_T(dtype, hipLaunchKernelGGL(simnn_adam_kernel<T>, dim3((unsigned)nbig), dim3(256), 0, (hipStream_t)stream, p_big,
                                       g_big_pc, m_big, v_big, C, P, (T*)shadow_pc, vec_ok, tx, ty, small_vec_ok, p_small, g_small,
                                       m_small, v_small, n_small, conv2_weight, (T*)pack, (T*)pack + C2<T>::WF_ELEMS, hyper,
                                       done));
  static_assert(REC_INTS == GDM_SIMNN_ADAM_RECORD_INTS, "include/gdm.h states the record's size");
  GDM_LAUNCH_OK("gdm_simnn_adam_step");
  return GDM_OK;
}

extern "C" int gdm_simnn_conv2_fwd(const void* p1, const void* pack, const float* bias, int B, int H1, int W1, void* p2,
                                   uint8_t* code2, int dtype, void* stream) {
  GDM_REQUIRE(p1 && pack && bias && p2 && code2, "gdm_simnn_conv2_fwd: null pointer");
  GDM_REQUIRE(B > 0 && H1 >= 2 && W1 >= 2 && gdm_dtype_ok(dtype), "gdm_simnn_conv2_fwd: bad arguments");
  GDM_REQUIRE(fits_buffer_addressing(B, H1, W1), "gdm_simnn_conv2_fwd: batch of %d %dx%d maps exceeds 2 GiB per tensor", B, H1, W1);
  const int H2 = H1 / 2, W2 = W1 / 2;
  const int n_ctiles = (2 * W2 + COLS - 1) / COLS;
  const int n_tiles = B * ((2 * H2 + ROWS - 1) / ROWS) * n_ctiles;
  static const int cap = tuned_cap("GDM_C2F_CAP", 768);          // persistent: 3 workgroups per CU
  dim3 grid((unsigned)(n_tiles < cap ? n_tiles : cap));
  hipStream_t s = (hipStream_t)stream;
  if (dtype == GDM_BF16) {
    const size_t sm = (size_t)(C2<__bf16>::IN_ELEMS + C2<__bf16>::WF_ELEMS) * 2;
    hipLaunchKernelGGL(conv2_fwd_kernel<__bf16>, grid, dim3(256), sm, s, (const __bf16*)p1, (const __bf16*)pack, bias,
                       H1, W1, H2, W2, n_ctiles, n_tiles, (__bf16*)p2, code2);
  } else {
    const size_t sm = (size_t)(C2<float>::IN_ELEMS + C2<float>::WF_ELEMS) * 4;
    allow_lds(conv2_fwd_kernel<float>, sm);
    hipLaunchKernelGGL(conv2_fwd_kernel<float>, grid, dim3(256), sm, s, (const float*)p1, (const float*)pack, bias, H1,
                       W1, H2, W2, n_ctiles, n_tiles, (float*)p2, code2);
  }
  GDM_LAUNCH_OK("gdm_simnn_conv2_fwd");
  return GDM_OK;
}

namespace {
template <typename T, bool FUSE>
int launch_bwd_data(const void* dp2, const uint8_t* code2, const void* pack, int B, int H1, int W1, void* dp1,
                    const uint64_t* code1, const float* x0, const float* x1, int bsplit, int H, int W, float* slabs,
                    hipStream_t s) {
  const int H2 = H1 / 2, W2 = W1 / 2;
  const BdPlan pl = bd_plan(B, H1, W1, FUSE);
  const size_t sm = BD<T>::lds_bytes(FUSE);
  // 16-byte x-window loads need rows that start on 16-byte boundaries
  const bool xvec = FUSE && W % 4 == 0 && (((uintptr_t)x0 | (uintptr_t)x1) & 15) == 0;
#define GDM_BD_LAUNCH(XV, D1)                                                                                          \
  allow_lds(conv2_bwd_data_kernel<T, FUSE, XV, D1>, sm);                                                               \
  hipLaunchKernelGGL((conv2_bwd_data_kernel<T, FUSE, XV, D1>), dim3(pl.blocks), dim3(256), sm, s, (const T*)dp2, code2, \
                     (const T*)pack + C2<T>::WF_ELEMS, B, H1, W1, H2, W2, pl.n_ctiles, pl.nseg, pl.seg_len, pl.n_items, \
                     (T*)dp1, (const uint16_t*)code1, x0, x1, bsplit, H, W, slabs)
  if constexpr (FUSE) {
    if (dp1 != nullptr) {
      if (xvec) { GDM_BD_LAUNCH(true, true); } else { GDM_BD_LAUNCH(false, true); }
    } else {
      if (xvec) { GDM_BD_LAUNCH(true, false); } else { GDM_BD_LAUNCH(false, false); }
    }
  } else {
    GDM_BD_LAUNCH(false, true);
  }
#undef GDM_BD_LAUNCH
  GDM_LAUNCH_OK("gdm_simnn_conv2_bwd_data");
  return GDM_OK;
}
}  // namespace

extern "C" int gdm_simnn_conv2_bwd_data(const void* dp2, const uint8_t* code2, const void* pack, int B, int H1, int W1,
                                        void* dp1, int dtype, void* stream) {
  GDM_REQUIRE(dp2 && code2 && pack && dp1, "gdm_simnn_conv2_bwd_data: null pointer");
  GDM_REQUIRE(B > 0 && H1 >= 2 && W1 >= 2 && gdm_dtype_ok(dtype), "gdm_simnn_conv2_bwd_data: bad arguments");
  GDM_REQUIRE(fits_buffer_addressing(B, H1, W1), "gdm_simnn_conv2_bwd_data: batch of %d %dx%d maps exceeds 2 GiB per tensor", B, H1, W1);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == GDM_BF16)
    return launch_bwd_data<__bf16, false>(dp2, code2, pack, B, H1, W1, dp1, nullptr, nullptr, nullptr, 0, 0, 0,
                                          nullptr, s);
  return launch_bwd_data<float, false>(dp2, code2, pack, B, H1, W1, dp1, nullptr, nullptr, nullptr, 0, 0, 0, nullptr,
                                       s);
}

extern "C" size_t gdm_simnn_conv2_bwd_fused_workspace_bytes(int B, int H1, int W1) {
  return (size_t)(bd_plan(B, H1, W1, true).blocks + 65) * 80 * sizeof(float);
}

extern "C" int gdm_simnn_conv2_bwd_fused(const void* dp2, const uint8_t* code2, const void* pack, int B, int H1, int W1,
                                         const uint64_t* code1, const float* x0, const float* x1, int bsplit, int H,
                                         int W, void* dp1_or_null, int dtype, void* workspace, size_t workspace_bytes,
                                         void* stream) {
  GDM_REQUIRE(dp2 && code2 && pack && code1 && x0, "gdm_simnn_conv2_bwd_fused: null pointer");
  GDM_REQUIRE(B > 0 && gdm_dtype_ok(dtype), "gdm_simnn_conv2_bwd_fused: bad arguments");
  GDM_REQUIRE(H1 == (H + 1) / 2 && W1 == (W + 1) / 2 && H1 >= 2 && W1 >= 2,
              "gdm_simnn_conv2_bwd_fused: (H1,W1)=(%d,%d) does not belong to a %dx%d input", H1, W1, H, W);
  GDM_REQUIRE(bsplit >= 0 && bsplit <= B && (bsplit == B || x1 != nullptr),
              "gdm_simnn_conv2_bwd_fused: second input pointer missing");
  GDM_REQUIRE(fits_buffer_addressing(B, H1, W1), "gdm_simnn_conv2_bwd_fused: batch of %d %dx%d maps exceeds 2 GiB per tensor",
              B, H1, W1);
  if (!workspace || workspace_bytes < gdm_simnn_conv2_bwd_fused_workspace_bytes(B, H1, W1)) {
    gdm_set_error("gdm_simnn_conv2_bwd_fused: workspace too small");
    return GDM_EWORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  float* slabs = (float*)workspace;
  int rc = dtype == GDM_BF16
               ? launch_bwd_data<__bf16, true>(dp2, code2, pack, B, H1, W1, dp1_or_null, code1, x0, x1, bsplit, H, W,
                                               slabs, s)
               : launch_bwd_data<float, true>(dp2, code2, pack, B, H1, W1, dp1_or_null, code1, x0, x1, bsplit, H, W,
                                              slabs, s);
  return rc;
}

extern "C" int gdm_simnn_conv2_bwd_fused_finish(int B, int H1, int W1, float* dw1, float* db1, void* workspace,
                                                size_t workspace_bytes, void* stream) {
  GDM_REQUIRE(dw1 && db1 && B > 0 && H1 >= 2 && W1 >= 2, "gdm_simnn_conv2_bwd_fused_finish: bad arguments");
  if (!workspace || workspace_bytes < gdm_simnn_conv2_bwd_fused_workspace_bytes(B, H1, W1)) {
    gdm_set_error("gdm_simnn_conv2_bwd_fused_finish: workspace too small");
    return GDM_EWORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  const int nblocks = bd_plan(B, H1, W1, true).blocks;
  float* slabs = (float*)workspace;
  float* scratch = slabs + (size_t)nblocks * 80;
  launch_slab_sum<1>(slabs, nblocks, 80, scratch, dw1, db1, 0, s);
  GDM_LAUNCH_OK("gdm_simnn_conv2_bwd_fused_finish");
  return GDM_OK;
}

extern "C" size_t gdm_simnn_conv2_bwd_weight_workspace_bytes(int B, int H1, int W1) {
  return (size_t)(bw_plan(B, H1, W1).blocks + 65) * (4608 + 32) * sizeof(float);
}

extern "C" int gdm_simnn_conv2_bwd_weight(const void* dp2, const uint8_t* code2, const void* p1, int B, int H1, int W1,
                                          float* dw, float* db, int dtype, void* workspace, size_t workspace_bytes,
                                          void* stream) {
  GDM_REQUIRE(dp2 && code2 && p1 && dw && db, "gdm_simnn_conv2_bwd_weight: null pointer");
  GDM_REQUIRE(B > 0 && H1 >= 2 && W1 >= 2 && gdm_dtype_ok(dtype), "gdm_simnn_conv2_bwd_weight: bad arguments");
  GDM_REQUIRE(fits_buffer_addressing(B, H1, W1), "gdm_simnn_conv2_bwd_weight: batch of %d %dx%d maps exceeds 2 GiB per tensor", B, H1, W1);
  if (!workspace || workspace_bytes < gdm_simnn_conv2_bwd_weight_workspace_bytes(B, H1, W1)) {
    gdm_set_error("gdm_simnn_conv2_bwd_weight: workspace too small");
    return GDM_EWORKSPACE;
  }
  const int H2 = H1 / 2, W2 = W1 / 2;
  const BdPlan pl = bw_plan(B, H1, W1);
  const int nblocks = pl.blocks;
  hipStream_t s = (hipStream_t)stream;
  const size_t red_bytes = (size_t)(4608 + 2048) * sizeof(float);
  if (dtype == GDM_BF16) {
    size_t sm = (size_t)(BW<__bf16>::DC_ELEMS + BW<__bf16>::P_ELEMS) * 2 + C2T_BYTES;
    if (sm < red_bytes) sm = red_bytes;
    allow_lds(conv2_bwd_weight_kernel<__bf16>, sm);
    hipLaunchKernelGGL(conv2_bwd_weight_kernel<__bf16>, dim3(nblocks), dim3(256), sm, s, (const __bf16*)dp2, code2,
                       (const __bf16*)p1, B, H1, W1, H2, W2, pl.n_ctiles, pl.nseg, pl.seg_len, pl.n_items,
                       (float*)workspace);
  } else {
    size_t sm = (size_t)(BW<float>::DC_ELEMS + BW<float>::P_ELEMS) * 4;
    if (sm < red_bytes) sm = red_bytes;
    allow_lds(conv2_bwd_weight_kernel<float>, sm);
    hipLaunchKernelGGL(conv2_bwd_weight_kernel<float>, dim3(nblocks), dim3(256), sm, s, (const float*)dp2, code2,
                       (const float*)p1, B, H1, W1, H2, W2, pl.n_ctiles, pl.nseg, pl.seg_len, pl.n_items,
                       (float*)workspace);
  }
  float* scratch = (float*)workspace + (size_t)nblocks * (4608 + 32);
  launch_slab_sum<2>((const float*)workspace, nblocks, 4608 + 32, scratch, dw, db, 0, s);
  GDM_LAUNCH_OK("gdm_simnn_conv2_bwd_weight");
  return GDM_OK;
}

#ifdef GDM_STAMPS
extern "C" int gdm_debug_read_stamps(unsigned long long* host_out, int n) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(gdm_stamp_buf), sizeof(unsigned long long) * n);
}
#endif
