// Mel-spectrogram featuriser (reference GAN_DES/util.py:37-61 -> torchaudio MelSpectrogram + AmplitudeToDB): the step
// that produces model 1's discriminator input.  The two contractions (DFT and mel filter bank) are fp32 GEMMs
// (gdm_gemm, exact-fp32 MFMA); this file holds the three data-movement / pointwise kernels around them.
#include "gdm_common.h"

namespace {

// Centred STFT frames with reflect padding:  out[(b * frames + f)][n] = x_b[reflect(f * hop + n - n_fft / 2)].
// (The Hann window is folded into the DFT matrix on the host.)  Four consecutive n per lane: 16-byte stores.
__global__ __launch_bounds__(256) void stft_frames_kernel(const float* __restrict__ x, int64_t L, int64_t x_stride,
                                                          int hop, int n_fft, int frames, int64_t total4,
                                                          float* __restrict__ out) {
  const int per_row = n_fft / 4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
    const int n = (int)(i % per_row) * 4;
    const int64_t row = i / per_row;
    const int f = (int)(row % frames);
    const int64_t b = row / frames;
    const float* xb = x + b * x_stride;
    const int64_t s0 = (int64_t)f * hop + n - n_fft / 2;
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int64_t s = s0 + e;
      s = s < 0 ? -s : s;
      s = s >= L ? 2 * (L - 1) - s : s;
      v[e] = xb[s];
    }
    *(f32x4*)(out + row * n_fft + n) = v;
  }
}

// c (rows, 2 * nfreq) = [re | im]  ->  p (rows, ldp) = re^2 + im^2 (columns nfreq .. ldp-1 are written as zeros)
__global__ __launch_bounds__(256) void power_spectrum_kernel(const float* __restrict__ c, int64_t rows, int nfreq,
                                                             int ldp, float* __restrict__ p) {
  const int64_t total = rows * ldp;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int k = (int)(i % ldp);
    const int64_t r = i / ldp;
    float v = 0.f;
    if (k < nfreq) {
      const float re = c[r * 2 * nfreq + k], im = c[r * 2 * nfreq + nfreq + k];
      v = re * re + im * im;
    }
    p[i] = v;
  }
}

// One workgroup per window: mel (frames, n_mels) -> dB, raised to (window max - top_db), written as (n_mels, frames).
__global__ __launch_bounds__(1024) void power_to_db_kernel(const float* __restrict__ mel, int frames, int n_mels,
                                                           float top_db, float amin, float* __restrict__ out) {
  extern __shared__ float db_s[];              // [n_mels][frames + 1] (padded: transposed write without bank conflicts)
  __shared__ float red[16];
  const int t = threadIdx.x, n = frames * n_mels, ld = frames + 1;
  const float* m = mel + (int64_t)blockIdx.x * n;
  float mx = -INFINITY;
  for (int i = t; i < n; i += 1024) {
    const int f = i / n_mels, k = i % n_mels;              // coalesced read along the mel axis
    const float d = 10.0f * log10f(fmaxf(m[i], amin));
    db_s[k * ld + f] = d;
    mx = fmaxf(mx, d);
  }
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((t & 63) == 0) red[t >> 6] = mx;
  __syncthreads();
  mx = red[0];
#pragma unroll
  for (int w = 1; w < 16; ++w) mx = fmaxf(mx, red[w]);
  const float floor_db = top_db >= 0.f ? mx - top_db : -INFINITY;
  float* o = out + (int64_t)blockIdx.x * n;
  for (int i = t; i < n; i += 1024) {
    const int k = i / frames, f = i % frames;              // coalesced write along the time axis
    o[i] = fmaxf(db_s[k * ld + f], floor_db);
  }
}

inline unsigned blocks_for(int64_t total) {
  int64_t b = (total + 255) / 256;
  return (unsigned)(b > 16384 ? 16384 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int gdm_stft_frames(const float* x, int B, int64_t L, int64_t x_stride, int hop, int n_fft, int frames,
                               float* out, void* stream) {
  GDM_REQUIRE(x && out, "gdm_stft_frames: null pointer");
  GDM_REQUIRE(B > 0 && hop > 0 && n_fft >= 4 && n_fft % 4 == 0 && frames > 0, "gdm_stft_frames: bad arguments");
  GDM_REQUIRE(L > n_fft / 2, "gdm_stft_frames: reflect padding needs more than n_fft/2 = %d samples, got %lld", n_fft / 2,
              (long long)L);
  GDM_REQUIRE((int64_t)(frames - 1) * hop <= L, "gdm_stft_frames: %d frames of hop %d exceed %lld samples", frames, hop,
              (long long)L);
  GDM_REQUIRE(((uintptr_t)out & 15) == 0, "gdm_stft_frames: output must be 16-byte aligned");
  const int64_t total4 = (int64_t)B * frames * (n_fft / 4);
  hipLaunchKernelGGL(stft_frames_kernel, dim3(blocks_for(total4)), dim3(256), 0, (hipStream_t)stream, x, L, x_stride, hop,
                     n_fft, frames, total4, out);
  GDM_LAUNCH_OK("gdm_stft_frames");
  return GDM_OK;
}

extern "C" int gdm_power_spectrum(const float* c, int64_t rows, int nfreq, int ldp, float* p, void* stream) {
  GDM_REQUIRE(c && p && rows > 0 && nfreq > 0 && ldp >= nfreq, "gdm_power_spectrum: bad arguments");
  hipLaunchKernelGGL(power_spectrum_kernel, dim3(blocks_for(rows * ldp)), dim3(256), 0, (hipStream_t)stream, c, rows,
                     nfreq, ldp, p);
  GDM_LAUNCH_OK("gdm_power_spectrum");
  return GDM_OK;
}

extern "C" int gdm_power_to_db(const float* mel, int B, int frames, int n_mels, float top_db, float amin, float* out,
                               void* stream) {
  GDM_REQUIRE(mel && out && B > 0 && frames > 0 && n_mels > 0 && amin > 0.f, "gdm_power_to_db: bad arguments");
  const size_t sm = (size_t)n_mels * (frames + 1) * sizeof(float);
  GDM_REQUIRE(sm <= 150 * 1024, "gdm_power_to_db: a %d x %d window does not fit the LDS staging (150 KB)", n_mels, frames);
  static bool attr_done = false;
  if (!attr_done) {
    attr_done = true;
    (void)hipFuncSetAttribute((const void*)power_to_db_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  }
  hipLaunchKernelGGL(power_to_db_kernel, dim3(B), dim3(1024), sm, (hipStream_t)stream, mel, frames, n_mels, top_db, amin,
                     out);
  GDM_LAUNCH_OK("gdm_power_to_db");
  return GDM_OK;
}
