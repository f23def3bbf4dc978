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
  __host__ __device__ explicit Dims(int t) {
    T = t; OW1 = t / 2; OW2 = (OW1 - 2) / 2 + 1; G1 = (OW1 + 3) / 4; G2 = OW2 / 4;
    XWP = t + 2; W1P = 4 * G1 + 1; W2P = OW2 + 2; KFC = 32 * OH2 * OW2;
  }
  // image sizes in bf16 elements, rounded to 16 bytes so that every image starts 16-byte aligned
  __host__ __device__ int xs_elems() const { return ((H + 3) * XWP * 2 + 7) & ~7; }
  __host__ __device__ int h1_elems() const { return ((OH1 + 2) * W1P * 16 + 7) & ~7; }
  __host__ __device__ int d2_elems() const { return ((OH2 + 2) * W2P * 32 + 7) & ~7; }
};
constexpr int W_ELEMS = 16 * W1K + 32 * W2FK + 4 * 16 * W2BK;        // weight images kept in LDS
// slab layout (floats): [0] loss, [1] dbfc, [2..18) db1, [18..50) db2, [50..562) dW1, [562..8754) dW2, then dWfc (KFC)
constexpr int S_LOSS = 0, S_DBFC = 1, S_DB1 = 2, S_DB2 = 18, S_DW1 = 50, S_DW2 = 562, S_DWFC = 8754;

__device__ __forceinline__ bf16x4 tr16(const __bf16* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p);
}
__device__ __forceinline__ bf16x8 cat8(bf16x4 lo, bf16x4 hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }
__device__ __forceinline__ bf16x8 ld8_b64x2(const __bf16* p) {       // 8 bf16 from an 8-byte aligned LDS address
  return cat8(*(const bf16x4*)p, *(const bf16x4*)(p + 4));
}
__device__ __forceinline__ float leaky(float v) { return v > 0.f ? v : 0.2f * v; }

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
__global__ __launch_bounds__(NTHREADS) void dcnn_fused_kernel(const float* __restrict__ xa, int bsplit,
                                                              const float* __restrict__ p0,
                                                              const float* __restrict__ p1, int B, int T, float ya,
                                                              float yb, const __bf16* __restrict__ pk,
                                                              float* __restrict__ logits, float* __restrict__ slabs,
                                                              int slab_width, int want_grad) {
  const Dims d(T);
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
  __bf16* xs = (__bf16*)dyn_smem;                 // [(H+3)][XWP][2]   index ((r+1)*XWP + (c+1))*2 + ch
  __bf16* h1s = xs + d.xs_elems();                // [(OH1+2)][W1P][16]
  __bf16* d2s = h1s + d.h1_elems();               // [(OH2+2)][W2P][32]
  __bf16* w1s = d2s + d.d2_elems();               // 16 x W1K
  __bf16* w2fs = w1s + 16 * W1K;                  // 32 x W2FK
  __bf16* w2bs = w2fs + 32 * W2FK;                // 4 x 16 x W2BK
  float* red = (float*)(w2bs + 4 * 16 * W2BK);    // 64 floats of block-reduction scratch
  const int t = threadIdx.x, l = t & 63, wv = t >> 6, lr = l & 15, lg = l >> 4, q4 = lr >> 2, p4 = lr & 3;
  const __bf16* wfp = pk + W_ELEMS;
  const float* biases = (const float*)(pk + W_ELEMS + d.KFC);
  const int XWP = d.XWP, W1P = d.W1P, W2P = d.W2P, OW1 = d.OW1, OW2 = d.OW2;

  // zero every LDS image once (halos stay zero for the whole kernel), then bring the weight images in
  {
    const int nz = (d.xs_elems() + d.h1_elems() + d.d2_elems()) / 8;
    for (int i = t; i < nz; i += NTHREADS) ((f32x4*)xs)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int i = t; i < W_ELEMS / 8; i += NTHREADS) ((f32x4*)w1s)[i] = ((const f32x4*)pk)[i];
  }
  __syncthreads();
  const bf16x8 a_w1 = *(const bf16x8*)&w1s[lr * W1K + 8 * lg];     // conv1's weight fragment lives in registers
  float b1v[4], b2v[2][4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    b1v[r] = biases[4 * lg + r];
    b2v[0][r] = biases[16 + 4 * lg + r];
    b2v[1][r] = biases[32 + 4 * lg + r];
  }
  const float bfc = biases[48];

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
  const int n_chunks = d.KFC / 8;

  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    // ---- P0: input planes -> xs (bf16, channel-interleaved)
    {
      const float* pl0 = b < bsplit ? xa + (int64_t)b * 2 * H * T : p0 + (int64_t)(b - bsplit) * H * T;
      const float* pl1 = b < bsplit ? pl0 + H * T : p1 + (int64_t)(b - bsplit) * H * T;
      const int per_plane = H * T / 4;             // float4 chunks (H*T is a multiple of 4)
      for (int i = t; i < 2 * per_plane; i += NTHREADS) {
        const int ch = i >= per_plane, j = ch ? i - per_plane : i;
        const f32x4 v = *(const f32x4*)((ch ? pl1 : pl0) + 4 * j);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int idx = 4 * j + e, r = idx / T, c = idx % T;
          xs[((r + 1) * XWP + (c + 1)) * 2 + ch] = (__bf16)v[e];
        }
      }
    }
    __syncthreads();
    // ---- P1: conv1 + LeakyReLU -> h1s.  unit = (output row, 16-column tile)
    {
      const int ntile = (OW1 + 15) / 16;
      for (int u = wv; u < OH1 * ntile; u += NWAVES) {
        const int oh = u / ntile, ow = 16 * (u % ntile) + lr, owc = min(ow, OW1 - 1);
        // k = 8*lg + j: kh = lg, (kw, ci) = j  -> 8 contiguous bf16 starting at input (2oh-1+lg, 2ow-1, 0)
        const bf16x8 bb = ld8_b64x2(&xs[((2 * oh + lg) * XWP + 2 * owc) * 2]);
        const f32x4 c = mfma16(a_w1, bb, (f32x4){0.f, 0.f, 0.f, 0.f});
        if (ow < OW1) {
          bf16x4 hv;
#pragma unroll
          for (int r = 0; r < 4; ++r) hv[r] = (__bf16)leaky(c[r] + b1v[r]);
          *(bf16x4*)&h1s[((oh + 1) * W1P + (ow + 1)) * 16 + 4 * lg] = hv;
        }
      }
    }
    __syncthreads();
    // ---- P2: conv2 + LeakyReLU -> d2s (holds h2 until the gradient overwrites it).  unit = output row (OW2 <= 16)
    for (int oh = wv; oh < OH2; oh += NWAVES) {
      const int owc = min(lr, OW2 - 1);
      f32x4 c[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const int tap = 2 * ks + (lg >> 1), kh = tap >> 2, kw = tap & 3;
        const bf16x8 bb = *(const bf16x8*)&h1s[((2 * oh + kh) * W1P + 2 * owc + kw) * 16 + 8 * (lg & 1)];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const bf16x8 a = *(const bf16x8*)&w2fs[(16 * i + lr) * W2FK + 32 * ks + 8 * lg];
          c[i] = mfma16(a, bb, c[i]);
        }
      }
      if (lr < OW2) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          bf16x4 hv;
#pragma unroll
          for (int r = 0; r < 4; ++r) hv[r] = (__bf16)leaky(c[i][r] + b2v[i][r]);
          *(bf16x4*)&d2s[((oh + 1) * W2P + (lr + 1)) * 32 + 16 * i + 4 * lg] = hv;
        }
      }
    }
    __syncthreads();
    // ---- P3: fc dot product (channels-last flatten order, permuted weight) -> logit, loss, dl
    bf16x8 h2c[FC_CH], wfc_c[FC_CH];
    float part = 0.f;
#pragma unroll
    for (int i = 0; i < FC_CH; ++i) {
      const int qd = t + NTHREADS * i;
      h2c[i] = __builtin_bit_cast(bf16x8, (f32x4){0.f, 0.f, 0.f, 0.f});
      wfc_c[i] = h2c[i];
      if (qd < n_chunks) {
        const int pix = qd >> 2, c8 = (qd & 3) * 8, oh = pix / OW2, ow = pix % OW2;
        h2c[i] = *(const bf16x8*)&d2s[((oh + 1) * W2P + (ow + 1)) * 32 + c8];
        wfc_c[i] = *(const bf16x8*)&wfp[(int64_t)qd * 8];
#pragma unroll
        for (int e = 0; e < 8; ++e) part = fmaf((float)h2c[i][e], (float)wfc_c[i][e], part);
      }
    }
    part = wave_sum(part);
    if (l == 0) red[wv] = part;
    __syncthreads();
    float z = bfc;
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
      // ---- P5: conv2 weight gradient: contraction over pixels, 8 groups of 4 row-adjacent pixels per k-step;
      //          wave wv owns taps 2wv, 2wv+1 (both 16-channel m-tiles)
      const int G2 = d.G2, ngroups2 = OH2 * G2;
      for (int g0 = 0; g0 < ngroups2; g0 += 8) {
        const int ga = g0 + 2 * lg, gb = ga + 1;
        const int ra = ga / G2, ca = (ga % G2) * 4 + q4, rb = gb / G2, cb = (gb % G2) * 4 + q4;
        bf16x8 a[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
          a[i] = cat8(tr16(&d2s[((ra + 1) * W2P + ca + 1) * 32 + 16 * i + 4 * p4]),
                      tr16(&d2s[((rb + 1) * W2P + cb + 1) * 32 + 16 * i + 4 * p4]));
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const int tap = 2 * wv + tt, kh = tap >> 2, kw = tap & 3;
          const bf16x8 bb = cat8(tr16(&h1s[((2 * ra + kh) * W1P + 2 * ca + kw) * 16 + 4 * p4]),
                                 tr16(&h1s[((2 * rb + kh) * W1P + 2 * cb + kw) * 16 + 4 * p4]));
#pragma unroll
          for (int i = 0; i < 2; ++i) acc_w2[i][tt] = mfma16(a[i], bb, acc_w2[i][tt]);
        }
      }
      __syncthreads();
      // ---- P6: conv2 data gradient (stride-2 transposed conv by parity class) + LeakyReLU' of conv1, in place in h1s
      for (int u = wv; u < OH1 * 2; u += NWAVES) {
        const int ih = u >> 1, pc = u & 1, iw = pc + 2 * lr, iwc = min(iw, OW1 - 1 - ((OW1 - 1 - pc) & 1));
        const int ph = (ih + 1) & 1, pw = (pc + 1) & 1, cl = ph * 2 + pw;
        f32x4 c = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t2 = 0; t2 < 4; ++t2) {
          const int kh = ph + 2 * (t2 >> 1), kw = pw + 2 * (t2 & 1);
          const int oh = (ih + 1 - kh) / 2, ow = (iwc + 1 - kw) / 2;     // exact: parities match by construction
          const bf16x8 bb = *(const bf16x8*)&d2s[((oh + 1) * W2P + (ow + 1)) * 32 + 8 * lg];
          const bf16x8 a = *(const bf16x8*)&w2bs[(cl * 16 + lr) * W2BK + 32 * t2 + 8 * lg];
          c = mfma16(a, bb, c);
        }
        if (iw < OW1) {
          __bf16* hp = &h1s[((ih + 1) * W1P + (iw + 1)) * 16 + 4 * lg];
          const bf16x4 hv = *(const bf16x4*)hp;
          bf16x4 gv;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float gy = c[r] * ((float)hv[r] > 0.f ? 1.f : 0.2f);
            db1p[r] += gy;
            gv[r] = (__bf16)gy;
          }
          *(bf16x4*)hp = gv;
        }
      }
      __syncthreads();
      // ---- P7: conv1 weight gradient: contraction over pixels (groups of 4 row-adjacent pixels, 8 per k-step);
      //          the k-steps are dealt over the waves, every wave accumulates both 16-column halves of (kh,kw,ci)
      const int G1 = d.G1, ngroups1 = OH1 * G1;
      for (int g0 = 8 * wv; g0 < ngroups1; g0 += 8 * NWAVES) {
        const int ga = g0 + 2 * lg, gb = ga + 1;
        const int ra = ga / G1, ca = (ga % G1) * 4 + q4, rb = gb / G1, cb = (gb % G1) * 4 + q4;
        const bf16x8 a = cat8(tr16(&h1s[((ra + 1) * W1P + ca + 1) * 16 + 4 * p4]),
                              tr16(&h1s[((rb + 1) * W1P + cb + 1) * 16 + 4 * p4]));
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          // column block 4p..4p+3 of n = (kh_local = p>>1, kw = 2(p&1)+{0,1}, ci): 4 contiguous bf16 of the input row
          const int kh = 2 * nt + (p4 >> 1), kwo = 2 * (p4 & 1);
          const bf16x8 bb = cat8(tr16(&xs[((2 * ra + kh) * XWP + 2 * ca + kwo) * 2]),
                                 tr16(&xs[((2 * rb + kh) * XWP + 2 * cb + kwo) * 2]));
          acc_w1[nt] = mfma16(a, bb, acc_w1[nt]);
        }
      }
    }
    __syncthreads();     // all LDS images are free for the next sample
  }

  // ---- epilogue: one slab per workgroup -------------------------------------------------------------------------------
  float* slab = slabs + (int64_t)blockIdx.x * slab_width;
  if (t == 0) slab[S_LOSS] = loss_acc;
  if (!want_grad) return;
  if (t == 0) slab[S_DBFC] = dbfc_acc;
  // dW2: wave wv holds taps 2wv, 2wv+1: C row = o (4*lg + r within m-tile i), C col = ci (lr)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = 16 * i + 4 * lg + r, tap = 2 * wv + tt;
        slab[S_DW2 + (o * 16 + lr) * 16 + tap] = acc_w2[i][tt][r];          // torch layout (o, ci, kh, kw)
      }
  // dWfc (channels-last order; un-permuted by the finish kernel)
#pragma unroll
  for (int i = 0; i < FC_CH; ++i) {
    const int qd = t + NTHREADS * i;
    if (qd < n_chunks) {
#pragma unroll
      for (int e = 0; e < 8; ++e) slab[S_DWFC + qd * 8 + e] = acc_fc[i][e];
    }
  }
  // dW1 (8 waves x 2 tiles), db1, db2: cross-wave / cross-lane sums through LDS in fixed order
  float* scr = (float*)dyn_smem;                   // the activation images are dead now
  __syncthreads();
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) scr[wv * 512 + (4 * lg + r) * 32 + 16 * nt + lr] = acc_w1[nt][r];   // [o][n=(kh,kw,ci)]
#pragma unroll
  for (int r = 0; r < 4; ++r) scr[4096 + t * 4 + r] = db1p[r];
#pragma unroll
  for (int e = 0; e < 8; ++e) scr[4096 + 2048 + t * 8 + e] = db2p[e];
  __syncthreads();
  {
    float s = 0.f;                                  // dW1: 512 outputs, one per thread
#pragma unroll
    for (int w = 0; w < NWAVES; ++w) s += scr[w * 512 + t];
    const int o = t >> 5, n = t & 31, kh = n >> 3, kw = (n >> 1) & 3, ci = n & 1;
    slab[S_DW1 + ((o * 2 + ci) * 4 + kh) * 4 + kw] = s;                      // torch layout (o, ci, kh, kw)
  }
  if (t < 16) {                                     // db1[c]: lanes with lg = c/4 of every wave hold register c%4
    const int c = t, g = c >> 2, r = c & 3;
    float s = 0.f;
    for (int th = 0; th < NTHREADS; ++th)
      if (((th & 63) >> 4) == g) s += scr[4096 + th * 4 + r];
    slab[S_DB1 + c] = s;
  } else if (t >= 64 && t < 96) {                   // db2[c]: threads with (t & 3) == c/8 hold element c%8
    const int c = t - 64, g = c >> 3, e = c & 7;
    float s = 0.f;
    for (int th = g; th < NTHREADS; th += 4) s += scr[4096 + 2048 + th * 8 + e];
    slab[S_DB2 + c] = s;
  }
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

template <bool FINISH>
__global__ __launch_bounds__(1024) void dcnn_slab_sum(const float* __restrict__ slabs, int nslabs, int per_group,
                                                      int width, float* __restrict__ out, DcnnSinks k) {
  __shared__ float part[16][64];
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
    if constexpr (!FINISH) {
      out[(int64_t)blockIdx.y * width + i] = tsum;
    } else {
      if (i == S_LOSS) { k.loss[0] = (k.accumulate_loss ? k.loss[0] : 0.f) + tsum; return; }
      if (!k.want_grad) return;
      if (i == S_DBFC) k.dbfc[0] = tsum;
      else if (i < S_DB2) k.db1[i - S_DB1] = tsum;
      else if (i < S_DW1) k.db2[i - S_DB2] = tsum;
      else if (i < S_DW2) k.dw1[i - S_DW1] = tsum;
      else if (i < S_DWFC) k.dw2[i - S_DW2] = tsum;
      else {                                       // slab: k' = pix*32 + c   ->   torch: c*P + pix
        const Dims d(k.T);
        const int kp = i - S_DWFC, P = OH2 * d.OW2;
        if (kp < d.KFC) k.dwfc[(kp & 31) * P + (kp >> 5)] = tsum;
      }
    }
  }
}


inline size_t lds_bytes(const Dims& d) {
  return (size_t)(d.xs_elems() + d.h1_elems() + d.d2_elems() + W_ELEMS) * 2 + 64 * sizeof(float);
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

extern "C" int gdm_dcnn_fused(const float* xa, int bsplit, const float* p0, const float* p1, int B, int T, float ya,
                              float yb, const void* pack, float* logits, float* loss, int accumulate_loss,
                              int want_grad, float* dw1, float* db1, float* dw2, float* db2, float* dwfc, float* dbfc,
                              void* workspace, size_t workspace_bytes, void* stream) {
  GDM_REQUIRE(pack && logits && loss && B > 0, "gdm_dcnn_fused: null pointer / empty batch");
  GDM_REQUIRE(supported(T), "gdm_dcnn_fused: roll length T=%d is outside the fused kernel's range", T);
  GDM_REQUIRE(bsplit >= 0 && bsplit <= B && (bsplit == 0 || xa) && (bsplit == B || (p0 && p1)),
              "gdm_dcnn_fused: input pointers do not cover the batch");
  GDM_REQUIRE(!want_grad || (dw1 && db1 && dw2 && db2 && dwfc && dbfc), "gdm_dcnn_fused: gradient outputs missing");
  if (!workspace || workspace_bytes < gdm_dcnn_fused_workspace_bytes(B, T, want_grad)) {
    gdm_set_error("gdm_dcnn_fused: workspace too small");
    return GDM_EWORKSPACE;
  }
  const Dims d(T);
  hipStream_t s = (hipStream_t)stream;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)dcnn_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  const int nb = n_blocks(B), width = slab_width(d, want_grad);
  float* slabs = (float*)workspace;
  hipLaunchKernelGGL(dcnn_fused_kernel, dim3(nb), dim3(NTHREADS), lds_bytes(d), s, xa, bsplit, p0, p1, B, T, ya, yb,
                     (const __bf16*)pack, logits, slabs, width, want_grad);
  float* scratch = slabs + (size_t)nb * width;      // up to 64 group partials
  const int groups = nb <= 32 ? 1 : (nb + 31) / 32;
  const int per = (nb + groups - 1) / groups;
  const unsigned gx = (unsigned)((width + 63) / 64);
  const DcnnSinks sinks{loss, accumulate_loss, want_grad, dw1, db1, dw2, db2, dwfc, dbfc, T};
  if (groups == 1) {
    hipLaunchKernelGGL(dcnn_slab_sum<true>, dim3(gx, 1), dim3(1024), 0, s, (const float*)slabs, nb, per, width,
                       (float*)nullptr, sinks);
  } else {
    hipLaunchKernelGGL(dcnn_slab_sum<false>, dim3(gx, groups), dim3(1024), 0, s, (const float*)slabs, nb, per, width,
                       scratch, sinks);
    hipLaunchKernelGGL(dcnn_slab_sum<true>, dim3(gx, 1), dim3(1024), 0, s, (const float*)scratch, groups, groups, width,
                       (float*)nullptr, sinks);
  }
  GDM_LAUNCH_OK("gdm_dcnn_fused");
  return GDM_OK;
}
