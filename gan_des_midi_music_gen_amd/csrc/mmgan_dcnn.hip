// Model 2's CNN discriminator (MMGAN_MIDI_DES/network_tests.py:147-160) as ONE persistent kernel: a 512-thread
// workgroup keeps a whole piano-roll sample and every activation of it in LDS (160 KB per CU on MI355X) and runs
//   conv1 k4 s2 p1 + LeakyReLU -> conv2 k4 s2 p1 + LeakyReLU -> fc -> BCE-with-logits        (forward + loss)
//   d fc, dW_fc, d conv2 (weights + data), LeakyReLU', d conv1 (weights)                      (backward)
// on it, so HBM sees the input planes once (51 KB per sample at T = 50) and nothing else; weight gradients are
// accumulated in registers across the samples a workgroup processes and leave as one slab per workgroup (summed in
// fixed order afterwards: deterministic, no atomics).  bf16 operands (piano-roll velocities/durations are small
// integers: exact), fp32 accumulation; the exact-fp32 parity path for this model stays on the GEMM + im2col lowering.
//
// All convolutions are implicit GEMMs on v_mfma_f32_16x16x32_bf16:
//   conv1 fwd   M=16 oc,  N=16 pixels of a row, K=32  (kh,kw,ci)          B operand = 8 contiguous bf16 of the input row
//   conv2 fwd   M=32 oc,  N=16 pixels,          K=256 (tap,ci)            B operand = half a 16-channel record
//   conv2 dW    M=32 oc,  N=16 ci per tap,      K=pixels                  both operands via ds_read_b64_tr_b16
//   conv2 dX    M=16 ci,  N=16 same-parity pixels of a row, K=128 (2x2 taps of the parity class, 32 oc)
//   conv1 dW    M=16 oc,  N=2x16 (kh,kw,ci),    K=pixels                  both operands via ds_read_b64_tr_b16
#include <cstdlib>
#include "gdm_common.h"

namespace {

constexpr int H = 128, OH1 = 64, OH2 = 32, NTHREADS = 512, NWAVES = 8;
constexpr int W1K = 40, W2FK = 264, W2BK = 136;                      // padded K strides of the packed weight images

struct Dims {
  int T, OW1, OW2, G1, G2, XWP, W1P, W2P, KFC;
  __host__ __device__ constexpr explicit Dims(int t)
      : T(t), OW1(t / 2), OW2((t / 2 - 2) / 2 + 1), G1((t / 2 + 3) / 4), G2(((t / 2 - 2) / 2 + 1) / 4), XWP(t + 2),
        W1P(4 * ((t / 2 + 3) / 4) + 1), W2P((t / 2 - 2) / 2 + 3), KFC(32 * OH2 * ((t / 2 - 2) / 2 + 1)) {}
  // image sizes in bf16 elements, rounded to 16 bytes so that every image starts 16-byte aligned
  __host__ __device__ constexpr int xs_elems() const { return ((H + 3) * XWP * 2 + 7) & ~7; }
  __host__ __device__ constexpr int h1_elems() const { return ((OH1 + 2) * W1P * 16 + 7) & ~7; }
  __host__ __device__ constexpr int d2_elems() const { return ((OH2 + 2) * W2P * 32 + 7) & ~7; }
};
constexpr int W_ELEMS = 16 * W1K + 32 * W2FK + 4 * 16 * W2BK;        // weight images kept in LDS
// slab layout (floats): [0] loss, [1] dbfc, [2..18) db1, [18..50) db2, [50..52) unused, [52..564) dW1, [564..8756) dW2,
// then dWfc (KFC): the three matrices start on 16-byte boundaries (vector stores)
constexpr int S_LOSS = 0, S_DBFC = 1, S_DB1 = 2, S_DB2 = 18, S_PAD = 50, S_DW1 = 52, S_DW2 = 564, S_DWFC = 8756;

__device__ __forceinline__ bf16x4 tr16(const __bf16* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p);
}
__device__ __forceinline__ bf16x8 cat8(bf16x4 lo, bf16x4 hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }
__device__ __forceinline__ bf16x8 ld8_b64x2(const __bf16* p) {       // 8 bf16 from an 8-byte aligned LDS address
  return cat8(*(const bf16x4*)p, *(const bf16x4*)(p + 4));
}
__device__ __forceinline__ float leaky(float v) { return fmaxf(v, 0.2f * v); }   // = v > 0 ? v : 0.2 v (two VALU, no select)

// -DGDM_DCNN_STAMPS: thread 0 of workgroup 0 sums the shader-clock span of every phase over its samples and prints them
// (measuring builds only; tools/calls)
#ifdef GDM_DCNN_STAMPS
#define DSTAMP_DECL uint64_t st_last = __builtin_amdgcn_s_memtime(), st_ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; int st_n = 0
#define DSTAMP(k) do { if (threadIdx.x == 0) { const uint64_t now_ = __builtin_amdgcn_s_memtime(); st_ph[k] += now_ - st_last; st_last = now_; } } while (0)
#define DSTAMP_FLUSH st_last = __builtin_amdgcn_s_memtime()
#define DSTAMP_END do { if (threadIdx.x == 0 && blockIdx.x == 0) { const uint64_t ep_ = __builtin_amdgcn_s_memtime() - st_last;       \
  printf("dcnn stamps (cycles per sample, %d samples; P8 = prologue, total):", st_n);                                                \
  for (int k_ = 0; k_ < 10; ++k_) printf(" P%d=%llu", k_, (unsigned long long)(st_ph[k_] / (k_ != 8 && st_n > 0 ? st_n : 1)));       \
  printf(" epilogue=%llu\n", (unsigned long long)ep_); } } while (0)
#else
#define DSTAMP_DECL
#define DSTAMP(k)
#define DSTAMP_FLUSH
#define DSTAMP_END
#endif

// ---- weight pack: bf16 images + permuted fc weight + fp32 biases ---------------------------------------------------
// pack (bytes): [W1img 16xW1K | W2f 32xW2FK | W2b 4x16xW2BK | wfp KFC] bf16, then [b1 16 | b2 32 | bfc 1] fp32
__global__ __launch_bounds__(256) void dcnn_pack_kernel(const float* __restrict__ w1, const float* __restrict__ b1,
                                                        const float* __restrict__ w2, const float* __restrict__ b2,
                                                        const float* __restrict__ wfc, const float* __restrict__ bfc,
                                                        int T, __bf16* __restrict__ pk) {
  const Dims d(T);
  const int total = W_ELEMS + d.KFC;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total + 49; i += gridDim.x * 256) {
    if (i >= total) {                                              // biases (fp32 tail, 4-byte aligned: total is even)
      float* bt = (float*)(pk + total);
      const int j = i - total;
      bt[j] = j < 16 ? b1[j] : (j < 48 ? b2[j - 16] : bfc[0]);
      continue;
    }
    float v = 0.f;
    if (i < 16 * W1K) {                                            // conv1: k = kh*8 + kw*2 + ci
      const int o = i / W1K, k = i % W1K;
      if (k < 32) v = w1[((o * 2 + (k & 1)) * 4 + (k >> 3)) * 4 + ((k >> 1) & 3)];
    } else if (i < 16 * W1K + 32 * W2FK) {                         // conv2 forward: k = tap*16 + ci, tap = kh*4 + kw
      const int j = i - 16 * W1K, o = j / W2FK, k = j % W2FK;
      if (k < 256) v = w2[(o * 16 + (k & 15)) * 16 + (k >> 4)];
    } else if (i < W_ELEMS) {                                      // conv2 data gradient, parity class cl = ph*2 + pw:
      const int j = i - 16 * W1K - 32 * W2FK;                      //   [cl][ci][k = t2*32 + o], kh = ph + 2a, kw = pw + 2b
      const int cl = j / (16 * W2BK), ci = (j / W2BK) % 16, k = j % W2BK;
      if (k < 128) {
        const int t2 = k >> 5, o = k & 31, kh = (cl >> 1) + 2 * (t2 >> 1), kw = (cl & 1) + 2 * (t2 & 1);
        v = w2[(o * 16 + ci) * 16 + kh * 4 + kw];
      }
    } else {                                                       // fc weight in channels-last order: k' = pix*32 + c
      const int j = i - W_ELEMS, c = j & 31, pix = j >> 5;
      v = wfc[c * (OH2 * d.OW2) + pix];
    }
    pk[i] = (__bf16)v;
  }
}

// ---- the fused per-sample kernel ---------------------------------------------------------------------------------------
// The roll length T is a template parameter (six values pass supported()): every LDS offset is then lane term +
// immediate, and the divisions by row lengths are multiplications.  With T as a run-time value the phases spent more
// issue cycles on index arithmetic than on their MFMAs and LDS reads.
template <int T>
__global__ __launch_bounds__(NTHREADS) void dcnn_fused_kernel(const float* __restrict__ xa, int bsplit,
                                                              const float* __restrict__ p0,
                                                              const float* __restrict__ p1, int B, float ya,
                                                              float yb, const __bf16* __restrict__ pk,
                                                              float* __restrict__ logits, float* __restrict__ slabs,
                                                              int slab_width, int want_grad) {
  constexpr Dims d(T);
  DSTAMP_DECL;
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
  __bf16* xs = (__bf16*)dyn_smem;                 // [(H+3)][XWP][2]   index ((r+1)*XWP + (c+1))*2 + ch
  __bf16* h1s = xs + d.xs_elems();                // [(OH1+2)][W1P][16]
  __bf16* d2s = h1s + d.h1_elems();               // [(OH2+2)][W2P][32]
  __bf16* w1s = d2s + d.d2_elems();               // 16 x W1K
  __bf16* w2fs = w1s + 16 * W1K;                  // 32 x W2FK
  __bf16* w2bs = w2fs + 32 * W2FK;                // 4 x 16 x W2BK
  float* red = (float*)(w2bs + 4 * 16 * W2BK);    // 64 floats of block-reduction scratch
  float* bias_s = red + 64;                       // b1 16 | b2 32 | bfc 1 (read at the phase that needs them: a register
                                                  // each for the whole kernel pushed the T = 50 instance into scratch)
  const int t = threadIdx.x, l = t & 63, wv = t >> 6, lr = l & 15, lg = l >> 4, q4 = lr >> 2, p4 = lr & 3;
  const __bf16* wfp = pk + W_ELEMS;
  const float* biases = (const float*)(pk + W_ELEMS + d.KFC);
  constexpr int XWP = d.XWP, W1P = d.W1P, W2P = d.W2P, OW1 = d.OW1, OW2 = d.OW2;

  // zero every LDS image once (halos stay zero for the whole kernel), then bring the weight images in
  {
    const int nz = (d.xs_elems() + d.h1_elems() + d.d2_elems()) / 8;
    for (int i = t; i < nz; i += NTHREADS) ((f32x4*)xs)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int i = t; i < W_ELEMS / 8; i += NTHREADS) ((f32x4*)w1s)[i] = ((const f32x4*)pk)[i];
    if (t < 49) bias_s[t] = biases[t];
  }
  __syncthreads();

  // gradient accumulators that live across all samples of this workgroup
  f32x4 acc_w2[2][2], acc_w1[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    acc_w1[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2; ++j) acc_w2[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  constexpr int FC_CH = 3;                         // 16-byte chunks of the fc dot product per thread (KFC <= 12288)
  float acc_fc[FC_CH][8], db2p[8], db1p[4], loss_acc = 0.f, dbfc_acc = 0.f;
#pragma unroll
  for (int i = 0; i < FC_CH; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc_fc[i][e] = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) db2p[e] = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) db1p[r] = 0.f;
  constexpr int n_chunks = d.KFC / 8;

  // input staging: thread t owns float4 chunks t, t + 512, ... of BOTH planes (H*T/4 <= 512 * XCH chunks per plane)
  constexpr int XCH = 4;
  constexpr int per_plane = H * T / 4;
  static_assert(per_plane <= NTHREADS * XCH && OW1 <= 32, "input staging registers / conv1 column tiles");
  f32x4 xr[XCH][2];
  int x_lds[XCH], x_wrap = 0;                      // LDS element index of the chunk's first pixel; bit 4k+e: pixel e is in the next row
#pragma unroll
  for (int k = 0; k < XCH; ++k) {
    const int j = min(t + NTHREADS * k, per_plane - 1), r0 = (4 * j) / T, c0 = (4 * j) % T;
#pragma unroll
    for (int e = 1; e < 4; ++e) x_wrap |= (c0 + e >= T ? 1 : 0) << (4 * k + e);
    x_lds[k] = ((r0 + 1) * XWP + (c0 + 1)) * 2;
  }
  auto x_issue = [&](int b) {
    const float* pl0 = b < bsplit ? xa + (int64_t)b * 2 * H * T : p0 + (int64_t)(b - bsplit) * H * T;
    const float* pl1 = b < bsplit ? pl0 + H * T : p1 + (int64_t)(b - bsplit) * H * T;
#pragma unroll
    for (int k = 0; k < XCH; ++k) {
      const int j = min(t + NTHREADS * k, per_plane - 1);
      xr[k][0] = *(const f32x4*)(pl0 + 4 * j);
      xr[k][1] = *(const f32x4*)(pl1 + 4 * j);
    }
  };
  auto x_store = [&]() {
#pragma unroll
    for (int k = 0; k < XCH; ++k) {
      if (t + NTHREADS * k >= per_plane) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // a chunk may run over the end of its image row (T % 4 != 0): the next row starts 2 halo pixels further on
        const int idx = x_lds[k] + 2 * e + ((x_wrap >> (4 * k + e)) & 1) * 4;
        bf16x2 v;
        v[0] = (__bf16)xr[k][0][e];
        v[1] = (__bf16)xr[k][1][e];
        *(bf16x2*)&xs[idx] = v;
      }
    }
  };
  // the fc weight chunks of this thread never change: registers for the whole kernel
  bf16x8 wfc_c[FC_CH];
#pragma unroll
  for (int i = 0; i < FC_CH; ++i) {
    const int qd = t + NTHREADS * i;
    wfc_c[i] = __builtin_bit_cast(bf16x8, (f32x4){0.f, 0.f, 0.f, 0.f});
    if (qd < n_chunks) wfc_c[i] = *(const bf16x8*)&wfp[(int64_t)qd * 8];
  }
  if ((int)blockIdx.x < B) x_issue(blockIdx.x);
  DSTAMP(8);                                         // prologue
  const int lr0 = lr, lg0 = lg, q40 = q4, p40 = p4, wv0 = wv;
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    // An opaque zero added to the lane coordinates: the phases' (loop-invariant) LDS offsets are then recomputed per
    // sample with a few VALU each instead of being hoisted out of this loop, where ~70 of them lived in scratch memory
    int opq = 0;
    asm volatile("" : "+s"(opq));
    const int lr = lr0 + opq, lg = lg0 + opq, q4 = q40 + opq, p4 = p40 + opq, wv = wv0 + opq;
    DSTAMP(9);
#ifdef GDM_DCNN_STAMPS
    ++st_n;
#endif
    // ---- P0: input planes (already in registers) -> xs (bf16, channel-interleaved: one 4-byte store per pixel).  The
    //          NEXT sample's planes are requested at the end of P6 (or after P3 without gradients): late enough that
    //          the 32 staging registers are free during the register-hungry phases, ~3 us before they are needed
    x_store();
    __syncthreads();
    DSTAMP(0);
    // Index arithmetic of all phases below: every LDS address is written as (lane term, a few VALU per sample) + (a
    // compile-time offset that lands in the instruction's offset field).  Left to "row = unit / tiles" style indexing
    // the phases issued 3-4x more address VALU than MFMA + LDS instructions and were bound by exactly that.
    //
    // ---- P1: conv1 + LeakyReLU -> h1s.  unit = (output row, 16-column tile); wave wv owns units wv + 8j: with two
    //          column tiles that is tile wv & 1 of rows (wv >> 1) + 4j, with one tile rows wv + 8j.  EIGHT units are in
    //          flight per wave (reads, MFMAs, stores of the eight are independent chains: one unit at a time is a
    //          ~500-cycle latency chain)
    {
      constexpr int tsh = OW1 > 16 ? 1 : 0, NJ = (OH1 << tsh) / NWAVES, RSTEP = NWAVES >> tsh;   // units per wave, row step
      constexpr int UB = 8;
      static_assert(NJ % UB == 0, "conv1 units are dealt in whole batches");
      const int ow = 16 * (wv & tsh) + lr, owc = min(ow, OW1 - 1), oh0 = wv >> tsh;
      const bf16x8 a_w1 = *(const bf16x8*)&w1s[lr * W1K + 8 * lg];
      const f32x4 b1v = *(const f32x4*)&bias_s[4 * lg];
      // k = 8*lg + j: kh = lg, (kw, ci) = j  -> 8 contiguous bf16 starting at input (2oh-1+lg, 2ow-1, 0)
      const __bf16* xb = &xs[((2 * oh0 + lg) * XWP + 2 * owc) * 2];
      __bf16* hb = &h1s[((oh0 + 1) * W1P + (ow + 1)) * 16 + 4 * lg];
#pragma unroll
      for (int j0 = 0; j0 < NJ; j0 += UB) {
        bf16x8 bb[UB];
#pragma unroll
        for (int q = 0; q < UB; ++q) bb[q] = ld8_b64x2(xb + (j0 + q) * (2 * RSTEP * XWP * 2));
        f32x4 c[UB];
#pragma unroll
        for (int q = 0; q < UB; ++q) c[q] = mfma16(a_w1, bb[q], b1v);       // the bias is the accumulator's initial value
        if (ow < OW1) {
#pragma unroll
          for (int q = 0; q < UB; ++q) {
            bf16x4 hv;
#pragma unroll
            for (int r = 0; r < 4; ++r) hv[r] = (__bf16)leaky(c[q][r]);
            *(bf16x4*)(hb + (j0 + q) * (RSTEP * W1P * 16)) = hv;
          }
        }
      }
    }
    __syncthreads();
    DSTAMP(1);
    // ---- P2: conv2 + LeakyReLU -> d2s (holds h2 until the gradient overwrites it).  unit = output row (OW2 <= 16);
    //          the wave's four rows (wv, wv+8, ...) advance together: eight independent accumulation chains, a
    //          weight fragment is read once for the four rows, the fragments of k-step s+1 are read before the MFMAs of s.
    //          k-step ks holds taps 2ks + h (h = lg >> 1): kh = ks >> 1, kw = 2 (ks & 1) + h
    {
      static_assert(OH2 == 4 * NWAVES, "a wave owns four conv2 output rows");
      const int owc = min(lr, OW2 - 1);
      const __bf16* hb = &h1s[((2 * wv) * W1P + 2 * owc + (lg >> 1)) * 16 + 8 * (lg & 1)];
      const __bf16* wb = &w2fs[lr * W2FK + 8 * lg];
      f32x4 c[4][2];
      {
        const f32x4 b2v[2] = {*(const f32x4*)&bias_s[16 + 4 * lg], *(const f32x4*)&bias_s[32 + 4 * lg]};
#pragma unroll
        for (int q = 0; q < 4; ++q) { c[q][0] = b2v[0]; c[q][1] = b2v[1]; }  // the bias is the accumulator's initial value
      }
      auto frags = [&](int ks, bf16x8 (&a)[2], bf16x8 (&bb)[4]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = *(const bf16x8*)(wb + i * 16 * W2FK + 32 * ks);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          bb[q] = *(const bf16x8*)(hb + ((2 * NWAVES * q + (ks >> 1)) * W1P + 2 * (ks & 1)) * 16);
      };
      bf16x8 a[2][2], bb[2][4];
      frags(0, a[0], bb[0]);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        if (ks + 1 < 8) frags(ks + 1, a[(ks + 1) & 1], bb[(ks + 1) & 1]);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int i = 0; i < 2; ++i) c[q][i] = mfma16(a[ks & 1][i], bb[ks & 1][q], c[q][i]);
      }
      if (lr < OW2) {
        __bf16* sb = &d2s[((wv + 1) * W2P + (lr + 1)) * 32 + 4 * lg];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            bf16x4 hv;
#pragma unroll
            for (int r = 0; r < 4; ++r) hv[r] = (__bf16)leaky(c[q][i][r]);
            *(bf16x4*)(sb + NWAVES * q * W2P * 32 + 16 * i) = hv;
          }
      }
    }
    __syncthreads();
    DSTAMP(2);
    // ---- P3: fc dot product (channels-last flatten order, permuted weight) -> logit, loss, dl
    bf16x8 h2c[FC_CH];
    float part = 0.f;
#pragma unroll
    for (int i = 0; i < FC_CH; ++i) {
      const int qd = t + NTHREADS * i;
      h2c[i] = __builtin_bit_cast(bf16x8, (f32x4){0.f, 0.f, 0.f, 0.f});
      if (qd < n_chunks) {
        const int pix = qd >> 2, c8 = (qd & 3) * 8, oh = pix / OW2, ow = pix % OW2;
        h2c[i] = *(const bf16x8*)&d2s[((oh + 1) * W2P + (ow + 1)) * 32 + c8];
#pragma unroll
        for (int e = 0; e < 8; ++e) part = fmaf((float)h2c[i][e], (float)wfc_c[i][e], part);
      }
    }
    part = wave_sum(part);
    if (l == 0) red[wv] = part;
    __syncthreads();
    float z = bias_s[48];
#pragma unroll
    for (int w = 0; w < NWAVES; ++w) z += red[w];
    const bool first = b < bsplit;
    const float y = first ? ya : yb, cnt = (float)(first ? bsplit : B - bsplit);
    const float dl = (1.0f / (1.0f + expf(-z)) - y) / cnt;
    if (t == 0) {
      logits[b] = z;
      loss_acc += (fmaxf(z, 0.f) - z * y + log1pf(expf(-fabsf(z)))) / cnt;
      dbfc_acc += dl;
    }
    DSTAMP(3);
    if (!want_grad) x_issue(min(b + (int)gridDim.x, B - 1));       // past the end: re-reads the last sample (never stored)
    if (want_grad) {
      // ---- P4: dW_fc accumulation and the gradient of conv2's output (through its LeakyReLU), in place in d2s
#pragma unroll
      for (int i = 0; i < FC_CH; ++i) {
        const int qd = t + NTHREADS * i;
        if (qd < n_chunks) {
          const int pix = qd >> 2, c8 = (qd & 3) * 8, oh = pix / OW2, ow = pix % OW2;
          bf16x8 g;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float h = (float)h2c[i][e];
            acc_fc[i][e] = fmaf(dl, h, acc_fc[i][e]);
            const float gy = dl * (float)wfc_c[i][e] * (h > 0.f ? 1.f : 0.2f);
            db2p[e] += gy;
            g[e] = (__bf16)gy;
          }
          *(bf16x8*)&d2s[((oh + 1) * W2P + (ow + 1)) * 32 + c8] = g;
        }
      }
      __syncthreads();
      DSTAMP(4);
      // ---- P5: conv2 weight gradient: contraction over pixels, 8 groups of 4 row-adjacent pixels per k-step; wave wv
      //          owns taps 2wv, 2wv+1 (both 16-channel m-tiles): kh = wv >> 1, kw = 2 (wv & 1) + tt.  k-step s covers
      //          the 8 rows 8 (s / G2) .. +7 of column group s % G2; lane group lg takes rows +2lg and +2lg+1.  The
      //          fragments of k-step s+1 are read before the MFMAs of k-step s (two register sets)
      {
        constexpr int G2 = d.G2, NS = (OH2 / 8) * G2;
        const __bf16* ab = &d2s[((2 * lg + 1) * W2P + q4 + 1) * 32 + 4 * p4];
        const __bf16* bb_ = &h1s[((4 * lg + (wv >> 1)) * W1P + 2 * q4 + 2 * (wv & 1)) * 16 + 4 * p4];
        auto frags = [&](int st, bf16x8 (&a)[2], bf16x8 (&bb)[2]) {
          const int rblk = st / G2, cg = st % G2;                         // compile-time after unrolling
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const __bf16* pa = ab + (8 * rblk * W2P + 4 * cg) * 32 + 16 * i;
            a[i] = cat8(tr16(pa), tr16(pa + W2P * 32));
          }
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) {
            const __bf16* pb = bb_ + (16 * rblk * W1P + 8 * cg + tt) * 16;
            bb[tt] = cat8(tr16(pb), tr16(pb + 2 * W1P * 16));
          }
        };
        bf16x8 a[2][2], bb[2][2];
        frags(0, a[0], bb[0]);
#pragma unroll
        for (int st = 0; st < NS; ++st) {
          if (st + 1 < NS) frags(st + 1, a[(st + 1) & 1], bb[(st + 1) & 1]);
#pragma unroll
          for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc_w2[i][tt] = mfma16(a[st & 1][i], bb[st & 1][tt], acc_w2[i][tt]);
        }
      }
      __syncthreads();
      DSTAMP(5);
      // ---- P6: conv2 data gradient (stride-2 transposed conv by parity class) + LeakyReLU' of conv1, in place in h1s.
      //          unit = (input row ih, column parity pc) = 16 same-parity pixels.  Wave wv owns ONE parity class (its
      //          four weight fragments are read once per sample): column parity pc = wv & 1, row parity rp = (wv >> 1) & 1,
      //          and the 16 rows ih = rp + 2j + 32 (wv >> 2).  kh = ph + 2a, kw = pw + 2b for tap t2 = 2a + b:
      //              oh = (ih + 1 - ph) / 2 - a = c + j - a,      ow = (iw + 1 - pw) / 2 - b
      //          so unit j's a = 1 fragments are unit j-1's a = 0 fragments: a batch of four units reads 8 new d2s
      //          fragments (+2 carried over) instead of 16, and its four MFMA chains are independent
      {
        const int pc = wv & 1, rp = (wv >> 1) & 1, half = wv >> 2, ph = (rp + 1) & 1, pw = (pc + 1) & 1, cl = ph * 2 + pw;
        const int iw = pc + 2 * lr, iwc = min(iw, OW1 - 1 - ((OW1 - 1 - pc) & 1));
        bf16x8 a[4];
#pragma unroll
        for (int t2 = 0; t2 < 4; ++t2) a[t2] = *(const bf16x8*)&w2bs[(cl * 16 + lr) * W2BK + 32 * t2 + 8 * lg];
        // d2s record of row c - 1 (= unit 0's a = 1 row), column ow(b = 1): every fragment is a non-negative offset away
        const int crow = ((rp + 1 - ph) >> 1) + 16 * half;                 // c: dy2 row of unit 0, a = 0
        const __bf16* db = &d2s[((crow - 1 + 1) * W2P + (((iwc + 1 - pw) >> 1) - 1 + 1)) * 32 + 8 * lg];
        __bf16* hb = &h1s[((rp + 32 * half + 1) * W1P + (min(iw, OW1 - 1) + 1)) * 16 + 4 * lg];
        bf16x8 rowf[5][2];                                                 // [dy2 row c+4m-1 .. c+4m+3][b]
#pragma unroll
        for (int bq = 0; bq < 2; ++bq) rowf[0][bq] = *(const bf16x8*)(db + (1 - bq) * 32);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          f32x4 c[4];
          bf16x4 hv[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            c[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
            hv[q] = *(const bf16x4*)(hb + 2 * (q + 4 * m) * W1P * 16);
#pragma unroll
            for (int bq = 0; bq < 2; ++bq) rowf[q + 1][bq] = *(const bf16x8*)(db + ((4 * m + q + 1) * W2P + 1 - bq) * 32);
          }
#pragma unroll
          for (int t2 = 0; t2 < 4; ++t2)
#pragma unroll
            for (int q = 0; q < 4; ++q) c[q] = mfma16(a[t2], rowf[q + 1 - (t2 >> 1)][t2 & 1], c[q]);
          if (iw < OW1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              bf16x4 gv;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float gy = c[q][r] * ((float)hv[q][r] > 0.f ? 1.f : 0.2f);
                db1p[r] += gy;
                gv[r] = (__bf16)gy;
              }
              *(bf16x4*)(hb + 2 * (q + 4 * m) * W1P * 16) = gv;
            }
          }
#pragma unroll
          for (int bq = 0; bq < 2; ++bq) rowf[0][bq] = rowf[4][bq];
        }
        x_issue(min(b + (int)gridDim.x, B - 1));                          // next sample's planes (see P0)
      }
      __syncthreads();
      DSTAMP(6);
      // ---- P7: conv1 weight gradient: contraction over pixels (groups of 4 row-adjacent pixels, 8 per k-step): wave wv
      //          owns the rows 8wv .. 8wv+7 (lane group lg: rows +2lg, +2lg+1), k-step m is column group m; every wave
      //          accumulates both 16-column halves of (kh,kw,ci); fragments of k-step m+1 are read before the MFMAs of m.
      //          (Column groups past OW1 hold zeros in h1s, so what the x fragments read there does not matter.)
      {
        constexpr int G1 = d.G1;
        static_assert(OH1 == 8 * NWAVES, "a wave owns eight conv1 output rows");
        const __bf16* ab = &h1s[((8 * wv + 2 * lg + 1) * W1P + q4 + 1) * 16 + 4 * p4];
        // column block 4p..4p+3 of n = (kh_local = p>>1, kw = 2(p&1)+{0,1}, ci): 4 contiguous bf16 of the input row
        const __bf16* xb = &xs[((2 * (8 * wv + 2 * lg) + (p4 >> 1)) * XWP + 2 * q4 + 2 * (p4 & 1)) * 2];
        auto frags = [&](int m, bf16x8& a, bf16x8 (&bb)[2]) {
          a = cat8(tr16(ab + 4 * m * 16), tr16(ab + 4 * m * 16 + W1P * 16));
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            const __bf16* px = xb + (2 * nt * XWP + 8 * m) * 2;
            bb[nt] = cat8(tr16(px), tr16(px + 2 * XWP * 2));
          }
        };
        bf16x8 a[2], bb[2][2];
        frags(0, a[0], bb[0]);
#pragma unroll
        for (int m = 0; m < G1; ++m) {
          if (m + 1 < G1) frags(m + 1, a[(m + 1) & 1], bb[(m + 1) & 1]);
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc_w1[nt] = mfma16(a[m & 1], bb[m & 1][nt], acc_w1[nt]);
        }
      }
    }
    __syncthreads();     // all LDS images are free for the next sample
    DSTAMP(7);
  }
  DSTAMP_FLUSH;

  // ---- epilogue: one slab per workgroup -------------------------------------------------------------------------------
  float* slab = slabs + (int64_t)blockIdx.x * slab_width;
  if (t == 0) slab[S_LOSS] = loss_acc;
  if (!want_grad) {
    DSTAMP_END;
    return;
  }
  if (t == 0) slab[S_DBFC] = dbfc_acc;
  // Everything leaves through LDS (the activation images are dead) so that the slab is written with 16-byte, lane-
  // contiguous stores and the per-channel bias sums are shuffles + an 8-wave sum instead of a 512-step serial loop
  // (that loop alone was ~6 us of every launch).  All sums in fixed order.
  float* scr = (float*)dyn_smem;
  float* scr_w2 = scr;                              // 8192: dW2 in torch layout (o, ci, kh, kw)
  float* scr_w1 = scr + 8192;                       // 8 x 512: per-wave dW1 partials [o][n = (kh, kw, ci)]
  float* scr_b1 = scr + 8192 + 4096;                // 8 x 16
  float* scr_b2 = scr_b1 + 128;                     // 8 x 32
  // dW2: wave wv holds taps 2wv, 2wv+1: C row = o (4*lg + r within m-tile i), C col = ci (lr)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = 16 * i + 4 * lg + r, tap = 2 * wv + tt;
        scr_w2[(o * 16 + lr) * 16 + tap] = acc_w2[i][tt][r];
      }
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) scr_w1[wv * 512 + (4 * lg + r) * 32 + 16 * nt + lr] = acc_w1[nt][r];
  // db1[4*lg + r]: lanes of one lg group (16 values of lr); db2[8*(l & 3) + e]: lanes with equal l & 3
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float v = db1p[r];
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
    if (lr == 0) scr_b1[wv * 16 + 4 * lg + r] = v;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float v = db2p[e];
    v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
    if (l < 4) scr_b2[wv * 32 + 8 * l + e] = v;
  }
  // dWfc (channels-last order; un-permuted by the finish kernel): 32 contiguous bytes per lane and chunk
#pragma unroll
  for (int i = 0; i < FC_CH; ++i) {
    const int qd = t + NTHREADS * i;
    if (qd < n_chunks) {
      f32x4* dst = (f32x4*)(slab + S_DWFC + qd * 8);
      dst[0] = (f32x4){acc_fc[i][0], acc_fc[i][1], acc_fc[i][2], acc_fc[i][3]};
      dst[1] = (f32x4){acc_fc[i][4], acc_fc[i][5], acc_fc[i][6], acc_fc[i][7]};
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8192 / 4 / NTHREADS; ++k) ((f32x4*)(slab + S_DW2))[t + NTHREADS * k] = ((const f32x4*)scr_w2)[t + NTHREADS * k];
  {
    float s_ = 0.f;                                 // dW1: 512 outputs, one per thread
#pragma unroll
    for (int w = 0; w < NWAVES; ++w) s_ += scr_w1[w * 512 + t];
    const int o = t >> 5, n = t & 31, kh = n >> 3, kw = (n >> 1) & 3, ci = n & 1;
    slab[S_DW1 + ((o * 2 + ci) * 4 + kh) * 4 + kw] = s_;                     // torch layout (o, ci, kh, kw)
  }
  if (t < 16) {
    float s_ = 0.f;
#pragma unroll
    for (int w = 0; w < NWAVES; ++w) s_ += scr_b1[w * 16 + t];
    slab[S_DB1 + t] = s_;
  } else if (t >= 64 && t < 96) {
    float s_ = 0.f;
#pragma unroll
    for (int w = 0; w < NWAVES; ++w) s_ += scr_b2[w * 32 + t - 64];
    slab[S_DB2 + t - 64] = s_;
  } else if (t >= 128 && t < 130) {
    slab[S_PAD + t - 128] = 0.f;
  }
  DSTAMP_END;
}

// ---- slab reduction + unpack into the torch gradient tensors -----------------------------------------------------------
// Fixed-order sum over slabs (wave w adds slabs w, w+16, ...; four independent partial sums so that the row loads
// overlap).  FINISH: the last level writes straight into the loss slot / the six gradient tensors (slab element i ->
// its torch position; dwfc is stored pixel-major in the slab and channel-major in torch).
struct DcnnSinks {
  float* loss; int accumulate_loss; int want_grad;
  float *dw1, *db1, *dw2, *db2, *dwfc, *dbfc;
  int T;
};
// FINISH == 2: the optimizer step rides the last level of the sum (one rank: nothing has to be exchanged between the
// gradient and Adam).  The thread that owns a gradient element also owns its parameter: Adam on (p, m, v) -- the same
// expression order as adam_dev_kernel -- and the updated value goes straight to its place(s) in the packed bf16 weight
// images the next launch of the fused kernel reads (conv2's weight has two: forward image and parity-class backward
// image).  Replaces slab sum -> adam_prep -> Adam -> re-pack (4 launches on the iteration's critical chain) by one.
// The step counter of the device hyper-parameter record is advanced by the LAST workgroup to finish: every workgroup
// has read the record before it can count itself finished.
struct DcnnUpdate {
  float *p[6], *m[6], *v[6];        // w1, b1, w2, b2, wfc, bfc: parameter, exp_avg, exp_avg_sq
  float* hyper;                     // {step (int bits), lr, beta1, beta2, eps, grad_scale, step_size, sqrt(bc2)}
  int* done;                        // workgroups-finished counter, zero between launches
  __bf16* pack;
};

template <int FINISH>
__global__ __launch_bounds__(1024) void dcnn_slab_sum(const float* __restrict__ slabs, int nslabs, int per_group,
                                                      int width, float* __restrict__ out, DcnnSinks k, DcnnUpdate u) {
  __shared__ float part[16][64];
  __shared__ float hy[2];
  if constexpr (FINISH == 2) {
    if (threadIdx.x == 1023) {         // derived terms in double, like adam_prep_kernel (and torch on the host)
      const int step = __float_as_int(u.hyper[0]) + 1;
      const double b1 = (double)u.hyper[2], b2 = (double)u.hyper[3];
      hy[0] = (float)((double)u.hyper[1] / (1.0 - pow(b1, (double)step)));
      hy[1] = (float)sqrt(1.0 - pow(b2, (double)step));
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  const int k0 = blockIdx.y * per_group, k1 = min(nslabs, k0 + per_group);
  float s4[4] = {0.f, 0.f, 0.f, 0.f};
  if (i < width) {
    int q = k0 + wv;
    for (; q + 48 < k1; q += 64) {
#pragma unroll
      for (int u = 0; u < 4; ++u) s4[u] += slabs[(int64_t)(q + 16 * u) * width + i];
    }
    for (; q < k1; q += 16) s4[0] += slabs[(int64_t)q * width + i];
  }
  part[wv][lane] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
  __syncthreads();
  if (wv == 0 && i < width) {
    float tsum = part[0][lane];
#pragma unroll
    for (int w = 1; w < 16; ++w) tsum += part[w][lane];
    if constexpr (FINISH == 0) {
      out[(int64_t)blockIdx.y * width + i] = tsum;
    } else {
      const Dims d(k.T);
      int which = -1, e = 0, pk0 = -1, pk1 = -1;                   // parameter, element, packed position(s)
      const int total = W_ELEMS + d.KFC;                           // bf16 elements in front of the fp32 bias tail
      if (i == S_LOSS) { k.loss[0] = (k.accumulate_loss ? k.loss[0] : 0.f) + tsum; }
      else if (!k.want_grad) {}
      else if (i == S_DBFC) { k.dbfc[0] = tsum; which = 5; e = 0; pk0 = 48; }
      else if (i < S_DB2) { e = i - S_DB1; k.db1[e] = tsum; which = 1; pk0 = e; }
      else if (i < S_PAD) { e = i - S_DB2; k.db2[e] = tsum; which = 3; pk0 = 16 + e; }
      else if (i < S_DW1) {}
      else if (i < S_DW2) {
        e = i - S_DW1; k.dw1[e] = tsum; which = 0;
        pk0 = (e >> 5) * W1K + ((e >> 2) & 3) * 8 + (e & 3) * 2 + ((e >> 4) & 1);          // [o][kh*8 + kw*2 + ci]
      } else if (i < S_DWFC) {
        e = i - S_DW2; k.dw2[e] = tsum; which = 2;
        const int o = e >> 8, ci = (e >> 4) & 15, kh = (e >> 2) & 3, kw = e & 3;
        pk0 = 16 * W1K + o * W2FK + (kh * 4 + kw) * 16 + ci;                                 // forward image
        pk1 = 16 * W1K + 32 * W2FK + (((kh & 1) * 2 + (kw & 1)) * 16 + ci) * W2BK + ((kh >> 1) * 2 + (kw >> 1)) * 32 + o;
      } else {                                     // slab: k' = pix*32 + c   ->   torch: c*P + pix
        const int kp = i - S_DWFC, P = OH2 * d.OW2;
        if (kp < d.KFC) { e = (kp & 31) * P + (kp >> 5); k.dwfc[e] = tsum; which = 4; pk0 = W_ELEMS + kp; }
      }
      if constexpr (FINISH == 2) {
        if (which >= 0) {
          const float w1 = 1.0f - u.hyper[2], beta2 = u.hyper[3], omb2 = 1.0f - u.hyper[3], eps = u.hyper[4];
          const float step_size = hy[0], bc2_sqrt = hy[1];
          float* pp = u.p[which] + e; float* pm = u.m[which] + e; float* pv = u.v[which] + e;
          float pn = *pp, mj = *pm, vj = *pv;
          adam_element(pn, mj, vj, tsum, u.hyper[5], w1, beta2, omb2, eps, step_size, bc2_sqrt);
          *pp = pn; *pm = mj; *pv = vj;
          if (which & 1) ((float*)(u.pack + total))[pk0] = pn;               // biases: fp32 tail of the pack
          else {
            u.pack[pk0] = (__bf16)pn;
            if (pk1 >= 0) u.pack[pk1] = (__bf16)pn;
          }
        }
      }
    }
  }
  if constexpr (FINISH == 2) {
    __syncthreads();
    if (threadIdx.x == 0) {
      if (atomicAdd(u.done, 1) == (int)(gridDim.x * gridDim.y) - 1) {
        u.hyper[0] = __int_as_float(__float_as_int(u.hyper[0]) + 1);
        u.hyper[6] = hy[0];
        u.hyper[7] = hy[1];
        *u.done = 0;
      }
    }
  }
}


template <int T>
void launch_fused(int nb, size_t lds, hipStream_t s, const float* xa, int bsplit, const float* p0, const float* p1, int B,
                  float ya, float yb, const __bf16* pack, float* logits, float* slabs, int width, int want_grad) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)dcnn_fused_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL((dcnn_fused_kernel<T>), dim3(nb), dim3(NTHREADS), lds, s, xa, bsplit, p0, p1, B, ya, yb, pack, logits,
                     slabs, width, want_grad);
}

inline size_t lds_bytes(const Dims& d) {
  return (size_t)(d.xs_elems() + d.h1_elems() + d.d2_elems() + W_ELEMS) * 2 + (64 + 64) * sizeof(float);
}
// One persistent workgroup per CU (151 KB of LDS each) -- on 7/8 of the CUs: a workgroup of this kernel owns its CU's
// LDS, so nothing else can run beside it there; the training step runs the generators' latency-bound launches on
// side streams, and they need somewhere to land (measured, model-2 step at B = 256: 256 workgroups 0.319 ms,
// 224 -> 0.255 ms, 192 -> 0.252 ms, 160 -> 0.275 ms; alone the kernel is ~12 % slower on 224).
inline int n_blocks(int B) {
  static int cap = 0;
  if (cap == 0) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      cus = 256;
    cap = cus * 7 / 8 > 0 ? cus * 7 / 8 : 1;
    if (const char* e = getenv("GDM_DCNN_CAP")) { if (atoi(e) > 0) cap = atoi(e); }     // experiments only
  }
  // the fewest workgroups that finish in the same number of sample rounds (512 samples on 224 workgroups take three
  // rounds; so do 171, which leaves 85 CUs instead of 32 to the kernels running beside this one)
  const int rounds = (B + cap - 1) / cap;
  return (B + rounds - 1) / rounds;
}
inline int slab_width(const Dims& d, int want_grad) { return want_grad ? S_DWFC + d.KFC : 1; }
inline bool supported(int T) {
  if (T < 8 || T % 2) return false;   // even T: the channel-interleaved input rows stay 8-byte aligned
  const Dims d(T);
  // kernel instances exist for the six even roll lengths with OW2 in {4, 8, 12}: 16, 18, 32, 34, 48, 50
  if (T != 16 && T != 18 && T != 32 && T != 34 && T != 48 && T != 50) return false;
  return d.OW2 >= 4 && d.OW2 <= 16 && d.OW2 % 4 == 0 && d.KFC <= 8 * NTHREADS * 3 && (OH2 * d.G2) % 8 == 0 &&
         (OH1 * d.G1) % 8 == 0 && lds_bytes(d) <= 160 * 1024 && (size_t)(4096 + 2048 + 4096) * 4 <= lds_bytes(d);
}

}  // namespace

extern "C" int gdm_dcnn_fused_supported(int T) { return supported(T) ? 1 : 0; }

extern "C" size_t gdm_dcnn_pack_bytes(int T) {
  const Dims d(T);
  return (size_t)(W_ELEMS + d.KFC) * 2 + 49 * sizeof(float) + 16;
}

extern "C" int gdm_dcnn_pack(const float* w1, const float* b1, const float* w2, const float* b2, const float* wfc,
                             const float* bfc, int T, void* pack, void* stream) {
  GDM_REQUIRE(w1 && b1 && w2 && b2 && wfc && bfc && pack, "gdm_dcnn_pack: null pointer");
  GDM_REQUIRE(supported(T), "gdm_dcnn_pack: roll length T=%d is outside the fused kernel's range", T);
  GDM_REQUIRE(((uintptr_t)pack & 15) == 0, "gdm_dcnn_pack: pack buffer must be 16-byte aligned");
  const Dims d(T);
  hipLaunchKernelGGL(dcnn_pack_kernel, dim3((W_ELEMS + d.KFC + 49 + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     w1, b1, w2, b2, wfc, bfc, T, (__bf16*)pack);
  GDM_LAUNCH_OK("gdm_dcnn_pack");
  return GDM_OK;
}

extern "C" size_t gdm_dcnn_fused_workspace_bytes(int B, int T, int want_grad) {
  const Dims d(T);
  return (size_t)(n_blocks(B) + 65) * slab_width(d, want_grad) * sizeof(float);
}

static int dcnn_fused_impl(const float* xa, int bsplit, const float* p0, const float* p1, int B, int T, float ya,
                           float yb, const void* pack, float* logits, float* loss, int accumulate_loss,
                           int want_grad, float* dw1, float* db1, float* dw2, float* db2, float* dwfc, float* dbfc,
                           const DcnnUpdate* upd, void* workspace, size_t workspace_bytes, void* stream) {
  GDM_REQUIRE(pack && logits && loss && B > 0, "gdm_dcnn_fused: null pointer / empty batch");
  GDM_REQUIRE(supported(T), "gdm_dcnn_fused: roll length T=%d is outside the fused kernel's range", T);
  GDM_REQUIRE(bsplit >= 0 && bsplit <= B && (bsplit == 0 || xa) && (bsplit == B || (p0 && p1)),
              "gdm_dcnn_fused: input pointers do not cover the batch");
  GDM_REQUIRE(!want_grad || (dw1 && db1 && dw2 && db2 && dwfc && dbfc), "gdm_dcnn_fused: gradient outputs missing");
  GDM_REQUIRE(((uintptr_t)workspace & 15) == 0, "gdm_dcnn_fused: workspace must be 16-byte aligned");
  if (!workspace || workspace_bytes < gdm_dcnn_fused_workspace_bytes(B, T, want_grad)) {
    gdm_set_error("gdm_dcnn_fused: workspace too small");
    return GDM_EWORKSPACE;
  }
  const Dims d(T);
  hipStream_t s = (hipStream_t)stream;
  const int nb = n_blocks(B), width = slab_width(d, want_grad);
  float* slabs = (float*)workspace;
  switch (T) {
#define GDM_DCNN_CASE(TT)                                                                                               \
    case TT:                                                                                                            \
      launch_fused<TT>(nb, lds_bytes(d), s, xa, bsplit, p0, p1, B, ya, yb, (const __bf16*)pack, logits, slabs, width,     \
                       want_grad);                                                                                      \
      break;
    GDM_DCNN_CASE(16) GDM_DCNN_CASE(18) GDM_DCNN_CASE(32) GDM_DCNN_CASE(34) GDM_DCNN_CASE(48) GDM_DCNN_CASE(50)
#undef GDM_DCNN_CASE
    default:
      gdm_set_error("gdm_dcnn_fused: no kernel instance for this roll length");
      return GDM_EINVAL;
  }
  // one level: 64 columns x 16 waves per block, wave w adds slabs w, w+16, ... (fixed order); a two-level tree was two
  // dependent launches for the same 14-21 MB of reads
  const unsigned gx = (unsigned)((width + 63) / 64);
  const DcnnSinks sinks{loss, accumulate_loss, want_grad, dw1, db1, dw2, db2, dwfc, dbfc, T};
  if (upd) {
    hipLaunchKernelGGL(dcnn_slab_sum<2>, dim3(gx, 1), dim3(1024), 0, s, (const float*)slabs, nb, nb, width,
                       (float*)nullptr, sinks, *upd);
  } else {
    hipLaunchKernelGGL(dcnn_slab_sum<1>, dim3(gx, 1), dim3(1024), 0, s, (const float*)slabs, nb, nb, width,
                       (float*)nullptr, sinks, DcnnUpdate{});
  }
  GDM_LAUNCH_OK("gdm_dcnn_fused");
  return GDM_OK;
}

extern "C" int gdm_dcnn_fused(const float* xa, int bsplit, const float* p0, const float* p1, int B, int T, float ya,
                              float yb, const void* pack, float* logits, float* loss, int accumulate_loss,
                              int want_grad, float* dw1, float* db1, float* dw2, float* db2, float* dwfc, float* dbfc,
                              void* workspace, size_t workspace_bytes, void* stream) {
  return dcnn_fused_impl(xa, bsplit, p0, p1, B, T, ya, yb, pack, logits, loss, accumulate_loss, want_grad, dw1, db1, dw2,
                         db2, dwfc, dbfc, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int gdm_dcnn_fused_adam(const float* xa, int bsplit, const float* p0, const float* p1, int B, int T, float ya,
                                   float yb, void* pack, float* logits, float* loss, int accumulate_loss, float* dw1,
                                   float* db1, float* dw2, float* db2, float* dwfc, float* dbfc,
                                   const gdm_dcnn_adam* opt, void* workspace, size_t workspace_bytes, void* stream) {
  GDM_REQUIRE(opt && opt->hyper && opt->done, "gdm_dcnn_fused_adam: optimizer record missing");
  DcnnUpdate u{};
  for (int q = 0; q < 6; ++q) {
    GDM_REQUIRE(opt->param[q] && opt->exp_avg[q] && opt->exp_avg_sq[q], "gdm_dcnn_fused_adam: null optimizer tensor %d", q);
    u.p[q] = opt->param[q]; u.m[q] = opt->exp_avg[q]; u.v[q] = opt->exp_avg_sq[q];
  }
  u.hyper = opt->hyper;
  u.done = opt->done;
  u.pack = (__bf16*)pack;
  return dcnn_fused_impl(xa, bsplit, p0, p1, B, T, ya, yb, pack, logits, loss, accumulate_loss, 1, dw1, db1, dw2, db2,
                         dwfc, dbfc, &u, workspace, workspace_bytes, stream);
}
