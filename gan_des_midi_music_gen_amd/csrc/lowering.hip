// Patch lowering for the small convolutions that run as GEMMs: model 2's DiscriminatorCNN (Conv2d k4 s2 p1,
// MMGAN_MIDI_DES/network_tests.py:150-151) and model 1's generator (ConvTranspose2d, GAN_DES/SIMNN.py:70-81).
//   im2col : cols[(b,oh,ow), (c,kh,kw)] = src[b, oh*s-p+kh, ow*s-p+kw, c]   (Conv2d forward and dW operand)
//   col2im : dst[b,h,w,c] = sum_{kh,kw : (h+p-kh) % s == 0, ...} cols[(b,oh,ow),(c,kh,kw)]   (Conv2d dX and
//            ConvTranspose2d forward), gather form: one lane per destination element, fixed summation order.
// The k order (c, kh, kw) is torch's own weight order, so Conv2d weights (Cout, Cin*KH*KW) and ConvTranspose2d
// weights (Cin, Cout*KH*KW) are used as GEMM operands in place, without a permuted copy.
// Activations are channels-last, except the planar (NCHW fp32) piano-roll input / image output which is read or
// written with its own index map.  gdm_permute_pc swaps the last two axes of a (B, P, C) array (NHWC <-> NCHW).
#include "gdm_common.h"

namespace {

__global__ __launch_bounds__(256) void im2col_kernel(const void* __restrict__ src, int sd, int planar, int B, int H,
                                                     int W, int C, int KH, int KW, int stride, int pad, int OH, int OW,
                                                     void* __restrict__ cols, int cd) {
  const int K = KH * KW * C;
  const int64_t total = (int64_t)B * OH * OW * K;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int k = (int)(i % K);
    const int64_t row = i / K;
    const int kw = k % KW, kh = (k / KW) % KH, c = k / (KW * KH);
    const int ow = (int)(row % OW), oh = (int)((row / OW) % OH), b = (int)(row / ((int64_t)OW * OH));
    const int ih = oh * stride - pad + kh, iw = ow * stride - pad + kw;
    float v = 0.f;
    if (ih >= 0 && ih < H && iw >= 0 && iw < W) {
      const int64_t si = planar ? (((int64_t)b * C + c) * H + ih) * W + iw : (((int64_t)b * H + ih) * W + iw) * C + c;
      v = load_as_f32(src, sd, si);
    }
    store_from_f32(cols, cd, i, v);
  }
}

// IT = index type: int when every index fits 32 bits (always, for this path's tensors) -- 64-bit divisions cost ~100
// instructions each and this kernel does four per element; int64_t otherwise.
template <typename IT>
__global__ __launch_bounds__(256) void col2im_kernel(const void* __restrict__ cols, int cd, int B, int H, int W, int C,
                                                     int KH, int KW, int stride, int pad, int OH, int OW,
                                                     void* __restrict__ dst, int dd, int flags) {
  const int K = KH * KW * C;
  const bool planar = flags & 1, tap_major = flags & 2;
  const int act = (flags >> 4) & 3;             // fused activation (the generator's final sigmoid, SIMNN.py:110)
  const IT total = (IT)B * H * W * C;
  for (IT i = (IT)blockIdx.x * 256 + threadIdx.x; i < total; i += (IT)gridDim.x * 256) {
    const int c = (int)(i % C);
    const IT pix = i / C;
    const int w = (int)(pix % W);
    const IT row = pix / W;
    const int h = (int)(row % H), b = (int)(row / H);
    float s = 0.f;
    // only the taps that can hit this output: kh = (h + pad) mod stride, + stride, ... with 0 <= oh < OH (a stride-2
    // 4x4 kernel has 2x2 of them, not 16: the kernel is bound by this index arithmetic, not by memory)
    const int hp = h + pad, wp = w + pad;
    int kh0 = hp % stride, kw0 = wp % stride;
    const int kh_min = hp - (OH - 1) * stride, kw_min = wp - (OW - 1) * stride;
    if (kh_min > kh0) kh0 += (kh_min - kh0 + stride - 1) / stride * stride;
    if (kw_min > kw0) kw0 += (kw_min - kw0 + stride - 1) / stride * stride;
    const int kh1 = min(KH - 1, hp), kw1 = min(KW - 1, wp);
    for (int kh = kh0; kh <= kh1; kh += stride) {
      const int oh = (hp - kh) / stride;
      for (int kw = kw0; kw <= kw1; kw += stride) {
        const int ow = (wp - kw) / stride;
        // tap-major columns ((kh,kw) slow, c fast): the lanes of a wave (consecutive c) read consecutive elements
        const int col = tap_major ? (kh * KW + kw) * C + c : (c * KH + kh) * KW + kw;
        s += load_as_f32(cols, cd, (int64_t)(((IT)b * OH + oh) * OW + ow) * K + col);
      }
    }
    const int64_t di = planar ? (((int64_t)b * C + c) * H + h) * W + w : (int64_t)i;
    store_from_f32(dst, dd, di, apply_act(s, act, 0.f));
  }
}

// dst[b][c][p] = src[b][p][c] (src viewed as (B, P, C)), 32x32 tiles through LDS so that both the read (along c) and
// the write (along p) are coalesced; converts between fp32 and bf16 on the way.
__global__ __launch_bounds__(256) void permute_pc_kernel(const void* __restrict__ src, int sd, int P, int C,
                                                         void* __restrict__ dst, int dd) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32, b = blockIdx.z;
  const int64_t base = (int64_t)b * P * C;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = p0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = (p < P && c < C) ? load_as_f32(src, sd, base + (int64_t)p * C + c) : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, p = p0 + tx;
    if (p < P && c < C) store_from_f32(dst, dd, base + (int64_t)c * P + p, tile[tx][ty + 8 * i]);
  }
}

// 2x2 / stride-2 max-pool on channels-last activations (F.max_pool2d(x, 2, 2), GAN_DES/SIMNN.py:156,158: the SimNN
// branch; model 1's discriminator pools inside its conv kernels).  idx = window position 0..3 of the FIRST maximum in
// scan order (aten::max_pool2d_with_indices); the backward routes the gradient there, zeros elsewhere (also into the
// odd last row/column the floor pooling drops).
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const void* __restrict__ src, int sd, int B, int H, int W,
                                                           int C, void* __restrict__ dst, uint8_t* __restrict__ idx) {
  const int OH = H / 2, OW = W / 2;
  const int64_t total = (int64_t)B * OH * OW * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int ow = (int)((i / C) % OW), oh = (int)((i / ((int64_t)C * OW)) % OH), b = (int)(i / ((int64_t)C * OW * OH));
    const int64_t base = (((int64_t)b * H + 2 * oh) * W + 2 * ow) * C + c;
    const float v0 = load_as_f32(src, sd, base), v1 = load_as_f32(src, sd, base + C);
    const float v2 = load_as_f32(src, sd, base + (int64_t)W * C), v3 = load_as_f32(src, sd, base + (int64_t)W * C + C);
    float m = v0;
    int pos = 0;
    if (v1 > m) { m = v1; pos = 1; }
    if (v2 > m) { m = v2; pos = 2; }
    if (v3 > m) { m = v3; pos = 3; }
    store_from_f32(dst, sd, i, m);
    if (idx) idx[i] = (uint8_t)pos;
  }
}

__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const void* __restrict__ dout, int dd,
                                                           const uint8_t* __restrict__ idx, int B, int H, int W, int C,
                                                           void* __restrict__ dx) {
  const int OH = H / 2, OW = W / 2;
  const int64_t total = (int64_t)B * H * W * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int w = (int)((i / C) % W), h = (int)((i / ((int64_t)C * W)) % H), b = (int)(i / ((int64_t)C * W * H));
    float g = 0.f;
    const int oh = h >> 1, ow = w >> 1;
    if (oh < OH && ow < OW) {
      const int64_t o = (((int64_t)b * OH + oh) * OW + ow) * C + c;
      if (idx[o] == (uint8_t)((h & 1) * 2 + (w & 1))) g = load_as_f32(dout, dd, o);
    }
    store_from_f32(dx, dd, i, g);
  }
}

inline unsigned grid_for(int64_t total) {
  int64_t b = (total + 255) / 256;
  return (unsigned)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int gdm_im2col(const void* src, int src_dtype, int src_planar, int B, int H, int W, int C, int KH, int KW,
                          int stride, int pad, int OH, int OW, void* cols, int cols_dtype, void* stream) {
  GDM_REQUIRE(src && cols, "gdm_im2col: null pointer");
  GDM_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0 && OH > 0 && OW > 0,
              "gdm_im2col: bad geometry");
  GDM_REQUIRE(gdm_dtype_ok(src_dtype) && gdm_dtype_ok(cols_dtype), "gdm_im2col: bad dtype");
  const int64_t total = (int64_t)B * OH * OW * KH * KW * C;
  hipLaunchKernelGGL(im2col_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, src_dtype,
                     src_planar, B, H, W, C, KH, KW, stride, pad, OH, OW, cols, cols_dtype);
  GDM_LAUNCH_OK("gdm_im2col");
  return GDM_OK;
}

extern "C" int gdm_col2im(const void* cols, int cols_dtype, int B, int H, int W, int C, int KH, int KW, int stride,
                          int pad, int OH, int OW, void* dst, int dst_dtype, int dst_planar, void* stream) {
  GDM_REQUIRE(cols && dst, "gdm_col2im: null pointer");
  GDM_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0 && OH > 0 && OW > 0,
              "gdm_col2im: bad geometry");
  GDM_REQUIRE(gdm_dtype_ok(dst_dtype) && gdm_dtype_ok(cols_dtype), "gdm_col2im: bad dtype");
  const int64_t total = (int64_t)B * H * W * C;
  if (total < ((int64_t)1 << 31) && (int64_t)B * OH * OW < ((int64_t)1 << 31))
    hipLaunchKernelGGL(col2im_kernel<int>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, cols, cols_dtype, B,
                       H, W, C, KH, KW, stride, pad, OH, OW, dst, dst_dtype, dst_planar);
  else
    hipLaunchKernelGGL(col2im_kernel<int64_t>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, cols,
                       cols_dtype, B, H, W, C, KH, KW, stride, pad, OH, OW, dst, dst_dtype, dst_planar);
  GDM_LAUNCH_OK("gdm_col2im");
  return GDM_OK;
}

// The same transpose with a 128 (p) x 32 (c) tile: the writes along p are 512 contiguous bytes per channel row (fp32)
// instead of 128 -- the large fc1 weight / weight-gradient permutes are bandwidth-bound and DRAM likes long runs.
__global__ __launch_bounds__(256) void permute_pc_wide_kernel(const void* __restrict__ src, int sd, int P, int C,
                                                              void* __restrict__ dst, int dd) {
  __shared__ float tile[128][33];
  const int t = threadIdx.x;
  const int p0 = blockIdx.x * 128, c0 = blockIdx.y * 32, b = blockIdx.z;
  const int64_t base = (int64_t)b * P * C;
  {
    const int tx = t & 31, ty = t >> 5;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int p = p0 + ty + 8 * i, c = c0 + tx;
      tile[ty + 8 * i][tx] = (p < P && c < C) ? load_as_f32(src, sd, base + (int64_t)p * C + c) : 0.f;
    }
  }
  __syncthreads();
  {
    const int tx = t & 127, ty = t >> 7;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = c0 + ty + 2 * i, p = p0 + tx;
      if (p < P && c < C) store_from_f32(dst, dd, base + (int64_t)c * P + p, tile[tx][ty + 2 * i]);
    }
  }
}

extern "C" int gdm_permute_pc(const void* src, int src_dtype, int B, int P, int C, void* dst, int dst_dtype,
                              void* stream) {
  GDM_REQUIRE(src && dst && B > 0 && P > 0 && C > 0 && gdm_dtype_ok(src_dtype) && gdm_dtype_ok(dst_dtype),
              "gdm_permute_pc: bad arguments");
  GDM_REQUIRE(B <= 65535 && (C + 31) / 32 <= 65535, "gdm_permute_pc: extent too large");
  if (P >= 512) {
    hipLaunchKernelGGL(permute_pc_wide_kernel, dim3((P + 127) / 128, (C + 31) / 32, B), dim3(256), 0, (hipStream_t)stream,
                       src, src_dtype, P, C, dst, dst_dtype);
  } else {
    hipLaunchKernelGGL(permute_pc_kernel, dim3((P + 31) / 32, (C + 31) / 32, B), dim3(256), 0, (hipStream_t)stream, src,
                       src_dtype, P, C, dst, dst_dtype);
  }
  GDM_LAUNCH_OK("gdm_permute_pc");
  return GDM_OK;
}

extern "C" int gdm_maxpool2_fwd(const void* src, int dtype, int B, int H, int W, int C, void* dst, uint8_t* idx_or_null,
                                void* stream) {
  GDM_REQUIRE(src && dst && B > 0 && H >= 2 && W >= 2 && C > 0 && gdm_dtype_ok(dtype), "gdm_maxpool2_fwd: bad arguments");
  const int64_t total = (int64_t)B * (H / 2) * (W / 2) * C;
  hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, dtype, B, H, W,
                     C, dst, idx_or_null);
  GDM_LAUNCH_OK("gdm_maxpool2_fwd");
  return GDM_OK;
}

extern "C" int gdm_maxpool2_bwd(const void* dout, int dtype, const uint8_t* idx, int B, int H, int W, int C, void* dx,
                                void* stream) {
  GDM_REQUIRE(dout && idx && dx && B > 0 && H >= 2 && W >= 2 && C > 0 && gdm_dtype_ok(dtype),
              "gdm_maxpool2_bwd: bad arguments");
  const int64_t total = (int64_t)B * H * W * C;
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dout, dtype, idx, B,
                     H, W, C, dx);
  GDM_LAUNCH_OK("gdm_maxpool2_bwd");
  return GDM_OK;
}
