// Batched DES-matrix prologue: the numpy arithmetic at the head of
//   matrix_to_midi  MMGAN_MIDI_DES/matrix_sim_process.py:33-117   (64x64 generator output, dim = 61, 3 parameter rows)
//   matrix_to_wav   GAN_DES/matrix_sim_process.py:21-95           (20x20 generator output, dim = 15, 5 parameter rows)
// which turns a generated matrix into what the discrete-event simulator is constructed from.  It is integer/byte and
// float64 work on 16 KB per sample: HBM-bound, one workgroup per sample, the sample staged once in LDS.
//
// The reference interleaves the arithmetic with draws from numpy's GLOBAL legacy RNG (random sources, the random
// column that absorbs a row's rounding residue, the per-sample reseed); how much of the stream a draw consumes depends
// on the matrix (candidate lists), so the stream has to be advanced on the host, sample by sample.  The split:
//   gdm_des_scan     everything that does not depend on a draw: |m|, the threshold mask of the source row, the integer
//                    instrument / note-level rows, the two normalised distribution rows of model 1, and one 64-bit
//                    mask per row saying which entries are exactly zero (what the host needs to build the candidate
//                    lists and to keep the RNG stream identical);
//   (host)           sources and residue columns drawn in the reference's order (matrix_sim_process.py host code);
//   gdm_des_routing  given sources and residue columns: zero source columns and the diagonal, float64 row sums in
//                    numpy's pairwise order, normalise (0/0 -> 0), add 1 - sum(row) to the chosen column, diagonal
//                    +1 (source) / -1 (server)  -> the (dim, dim) float64 routing matrix of every sample.
// Sums follow numpy's summation order exactly (pairwise_sum with 8 partial sums for 8 <= n <= 128, sequential below;
// Python's built-in sum() for the two float32 rows), divisions are IEEE, nothing is contracted into FMAs: the float64
// matrix is bit-identical to the reference's on the fixtures.
#pragma clang fp contract(off)
#include "gdm_common.h"

namespace {

constexpr int DES_MAX_S = 64;

__device__ __forceinline__ bool nonfinite_bits(float v) {
  return (__builtin_bit_cast(uint32_t, v) & 0x7f800000u) == 0x7f800000u;   // no isnan/isinf: built with -fno-honor-nans
}

// sample b -> LDS tile[S][S+1] of |m| (fp32), coalesced
__device__ __forceinline__ void load_abs_tile(const float* __restrict__ g, int64_t stride, int b, int S,
                                              float (*tile)[DES_MAX_S + 1], int* bad) {
  const float* src = g + (int64_t)b * stride;
  int local_bad = 0;
  for (int i = threadIdx.x; i < S * S; i += blockDim.x) {
    const float v = src[i];
    local_bad |= nonfinite_bits(v) ? 1 : 0;
    tile[i / S][i % S] = fabsf(v);
  }
  if (local_bad) atomicOr(bad, 1);
}

__global__ __launch_bounds__(256) void des_scan_kernel(const float* __restrict__ g, int64_t stride, int B, int S,
                                                       int dim, float thr, int note_mod, int norm_aux,
                                                       uint8_t* __restrict__ thr_mask, int32_t* __restrict__ instruments,
                                                       int32_t* __restrict__ note_levels,
                                                       uint64_t* __restrict__ zero_mask, float* __restrict__ aux,
                                                       int32_t* __restrict__ flags) {
  __shared__ float tile[DES_MAX_S][DES_MAX_S + 1];
  __shared__ int bad;
  const int b = blockIdx.x, t = threadIdx.x;
  if (t == 0) bad = 0;
  __syncthreads();
  load_abs_tile(g, stride, b, S, tile, &bad);
  __syncthreads();
  if (t < S && thr_mask) thr_mask[(int64_t)b * S + t] = tile[dim][t] > thr ? 1 : 0;       // np.where(matrix[dim] > thr)
  if (t < dim) {
    // int(matrix[dim + 1, i] * 126): float32 product, truncation
    instruments[(int64_t)b * dim + t] = (int)(tile[dim + 1][t] * 126.0f);
    const int nl = (int)(tile[dim + 2][t] * 126.0f);
    note_levels[(int64_t)b * dim + t] = note_mod ? max(0, nl % 128) : nl;
    uint64_t z = 0;
    for (int x = 0; x < dim; ++x) z |= (uint64_t)(tile[t][x] == 0.0f ? 1 : 0) << x;
    zero_mask[(int64_t)b * dim + t] = z;
  }
  if (norm_aux && t < 2) {
    // matrix[r] = matrix[r] / sum(matrix[r]) with Python's sum(): sequential float32 adds over ALL S entries
    const int r = dim + 3 + t;
    float s = 0.0f;
    for (int x = 0; x < S; ++x) s = s + tile[r][x];
    for (int x = 0; x < dim; ++x) aux[((int64_t)b * 2 + t) * dim + x] = tile[r][x] / s;
  }
  __syncthreads();
  if (t == 0) flags[b] = bad;
}

// numpy's pairwise_sum for n <= 128 doubles (n < 8: sequential from 0; else 8 partial sums + tail)
template <typename F>
__device__ __forceinline__ double np_pairwise(int n, F at) {
  if (n < 8) {
    double res = 0.0;
    for (int i = 0; i < n; ++i) res += at(i);
    return res;
  }
  double r[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = at(j);
  int i = 8;
  for (; i < n - (n % 8); i += 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] += at(i + j);
  }
  double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; ++i) res += at(i);
  return res;
}

__global__ __launch_bounds__(256) void des_routing_kernel(const float* __restrict__ g, int64_t stride, int B, int S,
                                                          int dim, const uint8_t* __restrict__ src_mask,
                                                          const int32_t* __restrict__ residue_col,
                                                          double* __restrict__ out) {
  __shared__ float tile[DES_MAX_S][DES_MAX_S + 1];
  __shared__ double q[DES_MAX_S][DES_MAX_S + 1];
  __shared__ uint8_t srcs[DES_MAX_S];
  __shared__ int bad;
  const int b = blockIdx.x, t = threadIdx.x;
  if (t == 0) bad = 0;
  if (t < dim) srcs[t] = src_mask[(int64_t)b * dim + t];
  __syncthreads();
  load_abs_tile(g, stride, b, S, tile, &bad);
  __syncthreads();
  if (t < dim) {
    const int i = t;
    // sim_matrix[:, sources] = 0, diagonal = 0, astype(float64)
    auto a = [&](int x) -> double { return (srcs[x] || x == i) ? 0.0 : (double)tile[i][x]; };
    const double rs = np_pairwise(dim, a);                                // row_sums = sim_matrix.sum(axis=1)
    for (int x = 0; x < dim; ++x) q[i][x] = rs == 0.0 ? 0.0 : a(x) / rs;    // 0/0 -> nan -> 0
    const double s2 = np_pairwise(dim, [&](int x) -> double { return q[i][x]; });
    const int col = residue_col[(int64_t)b * dim + i];
    if (col >= 0 && col < dim) q[i][col] += 1.0 - s2;                      // sim_matrix[i, choice] += 1 - sim_matrix[i].sum()
    q[i][i] = srcs[i] ? 1.0 : -1.0;
  }
  __syncthreads();
  double* dst = out + (int64_t)b * dim * dim;
  for (int k = t; k < dim * dim; k += blockDim.x) dst[k] = q[k / dim][k % dim];
}

}  // namespace

extern "C" int gdm_des_scan(const float* g, int64_t sample_stride, int B, int S, int dim, float threshold, int note_mod,
                            int norm_aux, uint8_t* thr_mask_or_null, int32_t* instruments, int32_t* note_levels,
                            uint64_t* zero_mask, float* aux_or_null, int32_t* flags, void* stream) {
  GDM_REQUIRE(g && instruments && note_levels && zero_mask && flags, "gdm_des_scan: null pointer");
  GDM_REQUIRE(B > 0 && S >= 4 && S <= DES_MAX_S && dim >= 1 && dim + 3 <= S && sample_stride >= (int64_t)S * S,
              "gdm_des_scan: bad geometry (S=%d, dim=%d; S <= %d, dim + 3 <= S)", S, dim, DES_MAX_S);
  GDM_REQUIRE(!norm_aux || (aux_or_null && dim + 5 <= S), "gdm_des_scan: the two distribution rows need dim + 5 <= S");
  hipLaunchKernelGGL(des_scan_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, g, sample_stride, B, S, dim, threshold,
                     note_mod, norm_aux, thr_mask_or_null, instruments, note_levels, zero_mask, aux_or_null, flags);
  GDM_LAUNCH_OK("gdm_des_scan");
  return GDM_OK;
}

extern "C" int gdm_des_routing(const float* g, int64_t sample_stride, int B, int S, int dim, const uint8_t* src_mask,
                               const int32_t* residue_col, double* out, void* stream) {
  GDM_REQUIRE(g && src_mask && residue_col && out, "gdm_des_routing: null pointer");
  GDM_REQUIRE(B > 0 && S >= 1 && S <= DES_MAX_S && dim >= 1 && dim <= S && sample_stride >= (int64_t)S * S,
              "gdm_des_routing: bad geometry (S=%d, dim=%d; S <= %d)", S, dim, DES_MAX_S);
  hipLaunchKernelGGL(des_routing_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, g, sample_stride, B, S, dim,
                     src_mask, residue_col, out);
  GDM_LAUNCH_OK("gdm_des_routing");
  return GDM_OK;
}
