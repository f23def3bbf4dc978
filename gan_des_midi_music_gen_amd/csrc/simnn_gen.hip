// Model 1's generator forward (GAN_DES/SIMNN.py:97-112) for the training loop, where it is a side chain whose only
// consumers are the DES bridge and the BatchNorm running statistics: layers 2..4 as three fused kernels instead of
// GEMM + col2im + 3 batch-norm launches each (the chain was 134 us of latency-bound launches per iteration and cost the
// pipelined step ~90 us: tools/overlap_probe.py).
//
//   y1 (B*16, 128) raw  --[BN1+ReLU on load] ConvT(128->64,k4,s2,p1)--> y2 (B*64, 64) raw + BN partials
//   y2                  --[BN2+ReLU on load] ConvT(64->32,k4,s2,p1)---> y3 (B*256, 32) raw + BN partials
//   y3                  --[BN3+ReLU on load] ConvT(32->1,k5,s1,p0) + sigmoid --> out (B, 400)
//
// A stride-2 k4 p1 transposed convolution is four 2x2-tap convolutions, one per output-parity class (qy, qx):
//   out[s, 2i+qy, 2j+qx, co] = sum_{a,b in {0,1}} sum_ci in[s, i+qy-a, j+qx-b, ci] * w[ci, co, 1-qy+2a, 1-qx+2b]
// i.e. per class a GEMM with K = 4*Cin: M = co (weights, A operand), N = pixels (B operand from an LDS image of the
// normalised input with a zero halo), v_mfma_f32_16x16x32_bf16 (bf16 operands, fp32 accumulation: the same roundings as
// the GEMM + col2im path it replaces).  A workgroup = (class, group of S samples); it also leaves an exact two-pass
// (mean, M2) partial per output channel, merged in fixed order by the existing bn_finalize.  Activations stay
// channels-last fp32 in HBM between the kernels (8 MB at most).
#include "gdm_common.h"

namespace {

// ---- weight pack: Wp[class 4][co][k = (a*2+b)*CIN + ci] bf16 for conv2 and conv3 ---------------------------------------
template <int CIN, int COUT>
__device__ __forceinline__ void pack_one(const float* __restrict__ w, __bf16* __restrict__ dst, int i) {
  // i indexes [cl][co][k]
  constexpr int K = 4 * CIN;
  const int k = i % K, co = (i / K) % COUT, cl = i / (K * COUT);
  const int ci = k % CIN, t = k / CIN, a = t >> 1, b = t & 1, qy = cl >> 1, qx = cl & 1;
  const int kh = 1 - qy + 2 * a, kw = 1 - qx + 2 * b;
  dst[i] = (__bf16)w[((ci * COUT + co) * 4 + kh) * 4 + kw];              // torch ConvTranspose2d weight (Cin, Cout, KH, KW)
}
constexpr int GP_W2 = 4 * 64 * 4 * 128, GP_W3 = 4 * 32 * 4 * 64;          // elements
constexpr int L1_K = 128, GP_W1 = 16 * 128 * L1_K;                        // conv1: [n = pos*128 + co][k = ci, zero padded]
__global__ __launch_bounds__(256) void gen_pack_kernel(const float* __restrict__ w1, int noise_dim,
                                                       const float* __restrict__ w2, const float* __restrict__ w3,
                                                       __bf16* __restrict__ pack) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < GP_W2) pack_one<128, 64>(w2, pack, i);
  else if (i < GP_W2 + GP_W3) pack_one<64, 32>(w3, pack + GP_W2, i - GP_W2);
  else if (i < GP_W2 + GP_W3 + GP_W1) {
    const int j = i - GP_W2 - GP_W3, k = j % L1_K, n = j / L1_K, co = n % 128, pos = n / 128;
    pack[i] = (__bf16)(k < noise_dim ? w1[((size_t)k * 128 + co) * 16 + pos] : 0.f);   // (Cin, 128, 4, 4)
  }
}

// ---- first layer: ConvT(noise_dim -> 128, k4, s1, p0) on a 1x1 input = a (B, noise_dim) x (noise_dim, 16*128) product,
// with its BatchNorm statistics: a 256-thread workgroup owns L1_CH = 4 channels at all 16 positions for the WHOLE batch
// (B <= 256), so the channel statistics are exact two-pass sums inside the workgroup and no partial / finalize launch
// follows; 32 workgroups (with 16 channels per workgroup the 8 of them took 22 us: each wrote 256 KB and held 32
// accumulator tiles per wave).  M = n (weights, A), N = b (noise, B operand), K = noise_dim padded to 128.
// A tile of wave wv: rows 4 * pos_local + co (positions 4wv .. 4wv+3, the 4 channels): in the result lane (lr, lg) holds
// the 4 channels of position 4wv + lg for sample lr of a batch tile -- one 16-byte store.
constexpr int L1_CH = 4;
__global__ __launch_bounds__(256) void gen_l1_kernel(const float* __restrict__ noise, int B, int noise_dim,
                                                     const __bf16* __restrict__ w1p, float* __restrict__ y1,
                                                     float momentum, float eps, float* __restrict__ running_mean,
                                                     float* __restrict__ running_var, int64_t* __restrict__ nbt,
                                                     float* __restrict__ save_mean, float* __restrict__ save_invstd) {
  constexpr int SK = L1_K + 8;                                             // padded LDS row (see convt_s2_bn_kernel)
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
  __bf16* x_s = (__bf16*)dyn_smem;                                         // [256 b][SK]
  __bf16* w_s = x_s + 256 * SK;                                            // [16 pos][4 co][SK]
  __shared__ float red[4][L1_CH], cmean[L1_CH];
  const int t = threadIdx.x, l = t & 63, wv = t >> 6, lr = l & 15, lg = l >> 4;
  const int c0 = blockIdx.x * L1_CH;
  // stage: weights of this channel block (64 rows x 128 k, 16-byte chunks), noise rows converted to bf16 (zero padded);
  // all global loads are issued before the first conversion
  {
    f32x4 wr[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = t + 256 * k, row = i / (L1_K / 8), c8 = i % (L1_K / 8), pos = row >> 2, co = row & 3;
      wr[k] = *(const f32x4*)&w1p[((size_t)(pos * 128 + c0 + co)) * L1_K + 8 * c8];
    }
    const bool vec = (noise_dim % 4 == 0) && (((uintptr_t)noise & 15) == 0);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      f32x4 xr[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int i = t + 256 * (k + 16 * half), b = i / (L1_K / 4), k4 = 4 * (i % (L1_K / 4));
        xr[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (b < B && k4 < noise_dim) {
          const float* src = noise + (size_t)b * noise_dim + k4;
          if (vec) xr[k] = *(const f32x4*)src;
          else
#pragma unroll
            for (int e = 0; e < 4; ++e) xr[k][e] = (k4 + e < noise_dim) ? src[e] : 0.f;
        }
      }
      if (half == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i = t + 256 * k, row = i / (L1_K / 8), c8 = i % (L1_K / 8);
          *(f32x4*)&w_s[row * SK + 8 * c8] = wr[k];
        }
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int i = t + 256 * (k + 16 * half), b = i / (L1_K / 4), k4 = 4 * (i % (L1_K / 4));
        bf16x4 h;
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = (__bf16)xr[k][e];
        *(bf16x4*)&x_s[b * SK + k4] = h;
      }
    }
  }
  __syncthreads();
  f32x4 acc[16];                                                           // [batch tile]
#pragma unroll
  for (int bt = 0; bt < 16; ++bt) acc[bt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < L1_K / 32; ++ks) {
    const bf16x8 af = *(const bf16x8*)&w_s[(16 * wv + lr) * SK + ks * 32 + 8 * lg];
#pragma unroll
    for (int bt = 0; bt < 16; ++bt) {
      const bf16x8 bf = *(const bf16x8*)&x_s[(bt * 16 + lr) * SK + ks * 32 + 8 * lg];
      acc[bt] = mfma16(af, bf, acc[bt]);
    }
  }
  // C: col (lr) = b within the batch tile, rows 4*lg + r = (position 4wv + lg, channel c0 + r)
#pragma unroll
  for (int bt = 0; bt < 16; ++bt) {
    const int b = bt * 16 + lr;
    if (b < B) *(f32x4*)&y1[((size_t)b * 16 + 4 * wv + lg) * 128 + c0] = acc[bt];
  }
  auto reduce = [&](auto value) {                                          // channel r: over tiles, the 64 lanes -> red[wv][r]
#pragma unroll
    for (int r = 0; r < L1_CH; ++r) {
      float sum = 0.f;
#pragma unroll
      for (int bt = 0; bt < 16; ++bt) sum += (bt * 16 + lr < B) ? value(acc[bt][r], r) : 0.f;
      sum += __shfl_xor(sum, 1, 64); sum += __shfl_xor(sum, 2, 64); sum += __shfl_xor(sum, 4, 64);
      sum += __shfl_xor(sum, 8, 64); sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);
      if (l == 0) red[wv][r] = sum;
    }
  };
  const float n = (float)B * 16.f;
  reduce([&](float v, int) { return v; });
  __syncthreads();
  if (t < L1_CH) cmean[t] = (((red[0][t] + red[1][t]) + red[2][t]) + red[3][t]) / n;
  __syncthreads();
  reduce([&](float v, int c) { const float d = v - cmean[c]; return d * d; });
  __syncthreads();
  if (t < L1_CH) {
    const float m2 = ((red[0][t] + red[1][t]) + red[2][t]) + red[3][t];
    const int c = c0 + t;
    save_mean[c] = cmean[t];
    save_invstd[c] = 1.0f / sqrtf(m2 / n + eps);
    if (running_mean) {
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * cmean[t];
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (m2 / fmaxf(n - 1.f, 1.f));
    }
  }
  if (blockIdx.x == 0 && t == 0 && nbt) nbt[0] += 1;
}

// ---- ConvT k4 s2 p1 with BatchNorm+ReLU applied to its input on load ----------------------------------------------------
// IH = input height = width (4 or 8); S = samples per workgroup; 256 threads.
template <int CIN, int COUT, int IH, int S>
__global__ __launch_bounds__(256) void convt_s2_bn_kernel(const float* __restrict__ yin, const float* __restrict__ mean,
                                                          const float* __restrict__ invstd,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int B,
                                                          const __bf16* __restrict__ wp, float* __restrict__ yout,
                                                          float* __restrict__ ws) {
  constexpr int K = 4 * CIN, HP = IH + 2, NPX = S * IH * IH, PT = NPX / 16, CT = COUT / 16, OH = 2 * IH;
  constexpr int TILES = PT * CT, TPW = TILES / 4;                          // C tiles per wave
  // LDS strides padded by 16 bytes: the fragment reads of 16 neighbouring pixels / weight rows (ds_read_b128) otherwise
  // all start on the same bank (power-of-two record sizes): 16-way conflicts made the first version 3x slower
  constexpr int SIN = CIN + 8, SW = K + 8;
  static_assert(NPX % 16 == 0 && TILES % 4 == 0 && CIN % 32 == 0, "tile shapes");
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
  __bf16* in_s = (__bf16*)dyn_smem;                                        // [S][HP][HP][SIN], zero halo
  __bf16* w_s = in_s + S * HP * HP * SIN;                                  // [COUT][SW] of this class
  float* red = (float*)(w_s + COUT * SW);                                  // [4 waves][COUT]
  __shared__ float sm[CIN], sc[CIN], sh[CIN], cmean[COUT];
  const int t = threadIdx.x, l = t & 63, wv = t >> 6, lr = l & 15, lg = l >> 4;
  const int cl = blockIdx.x & 3, grp = blockIdx.x >> 2, qy = cl >> 1, qx = cl & 1;
  const int s0 = grp * S;
  for (int c = t; c < CIN; c += 256) {
    sm[c] = mean[c];
    sc[c] = invstd[c] * gamma[c];
    sh[c] = beta[c];
  }
  // Staging is one HBM/L2 round trip deep: all global loads of the workgroup (this class's weights, the S input samples)
  // are issued into registers first, the LDS image is cleared while they fly, then weights and normalised inputs go to LDS.
  constexpr int WV = COUT * K / 8 / 256, XV = S * IH * IH * (CIN / 4) / 256;
  static_assert((COUT * K / 8) % 256 == 0 && (S * IH * IH * (CIN / 4)) % 256 == 0, "staging loops are exact");
  f32x4 wr[WV], xr[XV];
#pragma unroll
  for (int k = 0; k < WV; ++k) wr[k] = ((const f32x4*)(wp + (size_t)cl * COUT * K))[t + 256 * k];
#pragma unroll
  for (int k = 0; k < XV; ++k) {
    const int i = t + 256 * k, c4 = i % (CIN / 4), px = i / (CIN / 4), s = px / (IH * IH);
    xr[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (s0 + s < B) xr[k] = *(const f32x4*)(yin + ((size_t)s0 * IH * IH + px) * CIN + 4 * c4);
  }
  for (int i = t; i < S * HP * HP * SIN / 8; i += 256) ((f32x4*)in_s)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < WV; ++k) {
    const int i = t + 256 * k, row = i / (K / 8), c8 = i % (K / 8);
    *(f32x4*)&w_s[row * SW + 8 * c8] = wr[k];
  }
  __syncthreads();                                  // image cleared, BatchNorm scale / shift visible
#pragma unroll
  for (int k = 0; k < XV; ++k) {
    const int i = t + 256 * k, c4 = i % (CIN / 4), px = i / (CIN / 4), s = px / (IH * IH), iy = (px / IH) % IH, ix = px % IH;
    bf16x4 h;
#pragma unroll
    for (int e = 0; e < 4; ++e)        // bn_apply's expression, then ReLU
      h[e] = (__bf16)fmaxf((xr[k][e] - sm[4 * c4 + e]) * sc[4 * c4 + e] + sh[4 * c4 + e], 0.f);
    if (s0 + s < B) *(bf16x4*)&in_s[((s * HP + iy + 1) * HP + ix + 1) * SIN + 4 * c4] = h;
  }
  __syncthreads();
  // C tile q of this wave -> (pixel tile, channel tile); all channel tiles of a pixel tile sit in one wave
  f32x4 acc[TPW];
#pragma unroll
  for (int q = 0; q < TPW; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int PTW = TPW / CT;                                            // pixel tiles per wave
  int boff[PTW];                                                           // LDS element offset of this lane's pixel (tap 0,0)
#pragma unroll
  for (int p = 0; p < PTW; ++p) {
    const int px = (wv * PTW + p) * 16 + lr, s = px / (IH * IH), i = (px / IH) % IH, j = px % IH;
    boff[p] = ((s * HP + i + qy + 1) * HP + j + qx + 1) * SIN;
  }
#pragma unroll 2
  for (int ks = 0; ks < K / 32; ++ks) {
    const int tap = (ks * 32) / CIN, ci0 = (ks * 32) % CIN, a = tap >> 1, b = tap & 1;
    bf16x8 af[CT], bf[PTW];
#pragma unroll
    for (int c = 0; c < CT; ++c) af[c] = *(const bf16x8*)&w_s[(c * 16 + lr) * SW + ks * 32 + 8 * lg];
#pragma unroll
    for (int p = 0; p < PTW; ++p) bf[p] = *(const bf16x8*)&in_s[boff[p] - (a * HP + b) * SIN + ci0 + 8 * lg];
#pragma unroll
    for (int p = 0; p < PTW; ++p)
#pragma unroll
      for (int c = 0; c < CT; ++c) acc[p * CT + c] = mfma16(af[c], bf[p], acc[p * CT + c]);
  }
  // C layout: col (lr) = pixel, rows 4*lg + r = channel.  Store 4 consecutive channels of the lane's pixel.
#pragma unroll
  for (int p = 0; p < PTW; ++p) {
    const int px = (wv * PTW + p) * 16 + lr, s = px / (IH * IH), i = (px / IH) % IH, j = px % IH;
    if (s0 + s < B) {
      float* dst = yout + ((size_t)(s0 + s) * OH * OH + (2 * i + qy) * OH + 2 * j + qx) * COUT + 4 * lg;
#pragma unroll
      for (int c = 0; c < CT; ++c) *(f32x4*)(dst + 16 * c) = acc[p * CT + c];
    }
  }
  // ---- BatchNorm partial of this workgroup: exact two-pass (mean, M2) per output channel over its valid pixels
  const int n_valid = min(S, max(0, B - s0)) * IH * IH;
  auto reduce = [&](auto value) {                                          // -> red[wv][co] for co = 16c + 4lg + r
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float sum = 0.f;
#pragma unroll
        for (int p = 0; p < PTW; ++p) {
          const int px = (wv * PTW + p) * 16 + lr;
          sum += (s0 + px / (IH * IH) < B) ? value(acc[p * CT + c][r], 16 * c + 4 * lg + r) : 0.f;
        }
        sum += __shfl_xor(sum, 1, 64); sum += __shfl_xor(sum, 2, 64);
        sum += __shfl_xor(sum, 4, 64); sum += __shfl_xor(sum, 8, 64);
        if (lr == 0) red[wv * COUT + 16 * c + 4 * lg + r] = sum;
      }
  };
  reduce([&](float v, int) { return v; });
  __syncthreads();
  if (t < COUT) cmean[t] = (((red[t] + red[COUT + t]) + red[2 * COUT + t]) + red[3 * COUT + t]) / (float)max(n_valid, 1);
  __syncthreads();
  reduce([&](float v, int co) { const float d = v - cmean[co]; return d * d; });
  __syncthreads();
  if (t < COUT) {
    const float m2 = ((red[t] + red[COUT + t]) + red[2 * COUT + t]) + red[3 * COUT + t];
    float* o = ws + ((size_t)blockIdx.x * COUT + t) * 3;
    o[0] = (float)n_valid; o[1] = cmean[t]; o[2] = m2;
  }
}

// ---- last layer: BN3+ReLU on load, ConvT(32 -> 1, k5, s1, p0) 16x16 -> 20x20, sigmoid -----------------------------------
// out[oy][ox] = sum_{kh,kw,ci} in[oy-kh][ox-kw][ci] * w[ci][kh][kw].  One workgroup per sample, fp32 VALU (N = 1 output
// channel leaves an MFMA 15/16 empty); the sample (32 KB) and the weights (3.2 KB, [tap][ci]) live in LDS.
__global__ __launch_bounds__(512) void convt_k5_bn_sigmoid_kernel(const float* __restrict__ yin,
                                                                  const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta,
                                                                  const float* __restrict__ w4, float* __restrict__ out) {
  constexpr int C = 32, IH = 16, OH = 20, SP = C + 4;                      // pixel records padded to 144 bytes: the lanes of a
  __shared__ __attribute__((aligned(16))) float in_s[IH * IH * SP];        // wave read neighbouring pixels (128-byte records
                                                                           // put them all on one bank: 111 us instead of 7)
  __shared__ __attribute__((aligned(16))) float w_s[25 * C];
  __shared__ float sm[C], sc[C], sh[C];
  const int t = threadIdx.x, b = blockIdx.x;
  if (t < C) {
    sm[t] = mean[t];
    sc[t] = invstd[t] * gamma[t];
    sh[t] = beta[t];
  }
  for (int i = t; i < 25 * C; i += 512) w_s[i] = w4[(i % C) * 25 + i / C];        // (Cin, 1, 5, 5) -> [tap][ci]
  __syncthreads();
  const f32x4* src = (const f32x4*)(yin + (size_t)b * IH * IH * C);
  for (int i = t; i < IH * IH * C / 4; i += 512) {
    f32x4 v = src[i];
    const int c4 = (i % (C / 4)) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf((v[e] - sm[c4 + e]) * sc[c4 + e] + sh[c4 + e], 0.f);
    *(f32x4*)&in_s[(i / (C / 4)) * SP + c4] = v;
  }
  __syncthreads();
  // one output pixel per thread (400 of the 512), tap loops NOT unrolled: with an outer loop over outputs the compiler
  // hoisted all 800 weight reads into registers (512 VGPRs + 2 KB of scratch per lane, 111 us per launch)
  const int o = t;
  if (o < OH * OH) {
    const int oy = o / OH, ox = o % OH;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 1
    for (int kh = 0; kh < 5; ++kh) {
      const int iy = oy - kh;
      if ((unsigned)iy >= (unsigned)IH) continue;
#pragma unroll 1
      for (int kw = 0; kw < 5; ++kw) {
        const int ix = ox - kw;
        if ((unsigned)ix >= (unsigned)IH) continue;
        const f32x4* xp = (const f32x4*)&in_s[(iy * IH + ix) * SP];
        const f32x4* wq = (const f32x4*)&w_s[(kh * 5 + kw) * C];
#pragma unroll
        for (int q = 0; q < C / 4; ++q) {
          const f32x4 xv = xp[q], wv = wq[q];
          a0 = fmaf(xv[0], wv[0], a0); a1 = fmaf(xv[1], wv[1], a1);
          a2 = fmaf(xv[2], wv[2], a2); a3 = fmaf(xv[3], wv[3], a3);
        }
      }
    }
    out[(size_t)b * OH * OH + o] = sigmoid_f((a0 + a1) + (a2 + a3));
  }
}

template <typename K>
inline void allow_dyn_lds(K kernel, size_t bytes) {
  static const void* done[4];
  static int n_done = 0;
  for (int i = 0; i < n_done; ++i)
    if (done[i] == (const void*)kernel) return;
  (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (n_done < 4) done[n_done++] = (const void*)kernel;
}

template <int CIN, int COUT, int IH, int S>
int launch_convt_s2(const float* yin, const float* mean, const float* invstd, const float* gamma, const float* beta,
                    int B, const __bf16* wp, float* yout, float* ws, hipStream_t s) {
  const size_t sm = ((size_t)S * (IH + 2) * (IH + 2) * (CIN + 8) + (size_t)COUT * (4 * CIN + 8)) * 2 + (size_t)4 * COUT * 4;
  allow_dyn_lds(convt_s2_bn_kernel<CIN, COUT, IH, S>, sm);
  const int groups = (B + S - 1) / S;
  hipLaunchKernelGGL((convt_s2_bn_kernel<CIN, COUT, IH, S>), dim3(4 * groups), dim3(256), sm, s, yin, mean, invstd, gamma,
                     beta, B, wp, yout, ws);
  return 4 * groups;
}

}  // namespace

extern "C" size_t gdm_simnn_gen_pack_bytes(void) { return (size_t)(GP_W2 + GP_W3 + GP_W1) * 2; }

extern "C" int gdm_simnn_gen_pack(const float* w1, int noise_dim, const float* w2, const float* w3, void* pack,
                                  void* stream) {
  GDM_REQUIRE(w1 && w2 && w3 && pack && ((uintptr_t)pack & 15) == 0 && noise_dim >= 1 && noise_dim <= L1_K,
              "gdm_simnn_gen_pack: bad arguments (noise_dim <= %d)", L1_K);
  hipLaunchKernelGGL(gen_pack_kernel, dim3((GP_W2 + GP_W3 + GP_W1 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w1,
                     noise_dim, w2, w3, (__bf16*)pack);
  GDM_LAUNCH_OK("gdm_simnn_gen_pack");
  return GDM_OK;
}

extern "C" int gdm_simnn_gen_first(const float* noise, int B, int noise_dim, const void* pack, float* y1, float momentum,
                                   float eps, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                   float* save_mean, float* save_invstd, void* stream) {
  GDM_REQUIRE(noise && pack && y1 && save_mean && save_invstd, "gdm_simnn_gen_first: null pointer");
  GDM_REQUIRE(B > 1 && B <= 256 && noise_dim >= 1 && noise_dim <= L1_K && ((uintptr_t)y1 & 15) == 0,
              "gdm_simnn_gen_first: batch %d outside 2..256 (the workgroup owns the whole batch) or noise_dim > %d", B, L1_K);
  const size_t sm = (size_t)(256 + 16 * L1_CH) * (L1_K + 8) * 2;
  allow_dyn_lds(gen_l1_kernel, sm);
  hipLaunchKernelGGL(gen_l1_kernel, dim3(128 / L1_CH), dim3(256), sm, (hipStream_t)stream, noise, B, noise_dim,
                     (const __bf16*)pack + GP_W2 + GP_W3, y1, momentum, eps, running_mean, running_var,
                     num_batches_tracked, save_mean, save_invstd);
  GDM_LAUNCH_OK("gdm_simnn_gen_first");
  return GDM_OK;
}

extern "C" int gdm_simnn_gen_convt_chunks(int layer, int B) { return layer == 2 ? 4 * ((B + 7) / 8) : 4 * ((B + 3) / 4); }

extern "C" int gdm_simnn_gen_convt_bn(int layer, const float* yin, const float* mean, const float* invstd,
                                      const float* gamma, const float* beta, int B, const void* pack, float* yout,
                                      float* ws_partials, void* stream) {
  GDM_REQUIRE(yin && mean && invstd && gamma && beta && pack && yout && ws_partials && B > 0 && (layer == 2 || layer == 3),
              "gdm_simnn_gen_convt_bn: bad arguments");
  GDM_REQUIRE((((uintptr_t)yin | (uintptr_t)yout | (uintptr_t)pack) & 15) == 0, "gdm_simnn_gen_convt_bn: 16-byte alignment");
  hipStream_t s = (hipStream_t)stream;
  if (layer == 2) launch_convt_s2<128, 64, 4, 8>(yin, mean, invstd, gamma, beta, B, (const __bf16*)pack, yout, ws_partials, s);
  else launch_convt_s2<64, 32, 8, 4>(yin, mean, invstd, gamma, beta, B, (const __bf16*)pack + GP_W2, yout, ws_partials, s);
  GDM_LAUNCH_OK("gdm_simnn_gen_convt_bn");
  return GDM_OK;
}

extern "C" int gdm_simnn_gen_last(const float* yin, const float* mean, const float* invstd, const float* gamma,
                                  const float* beta, const float* w4, int B, float* out, void* stream) {
  GDM_REQUIRE(yin && mean && invstd && gamma && beta && w4 && out && B > 0 && ((uintptr_t)yin & 15) == 0,
              "gdm_simnn_gen_last: bad arguments");
  hipLaunchKernelGGL(convt_k5_bn_sigmoid_kernel, dim3(B), dim3(512), 0, (hipStream_t)stream, yin, mean, invstd, gamma,
                     beta, w4, out);
  GDM_LAUNCH_OK("gdm_simnn_gen_last");
  return GDM_OK;
}
