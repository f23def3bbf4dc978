// Micro-benchmark: what read bandwidth does the fc1-forward access pattern allow, independent of the GEMM kernel?
// A (512 x 65536 bf16, row-major) is read by 256 workgroups exactly as gemm_bf16_fast's deep variant reads it
// (128-row x 64-k tiles, interleaved split-K slices), with UNR tiles in flight per workgroup and nothing else.
//   hipcc -O3 --offload-arch=gfx950 tools/membench/rowtile_read.hip -o gpurun_out/rowtile_read && gpurun_out/rowtile_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int UNR, bool LINEAR>
__global__ __launch_bounds__(256) void reader(const uint16_t* __restrict__ a, int M, int K, int splits, float* out) {
  const int t = threadIdx.x, wg = blockIdx.x;
  const int mt = wg % (M / 128), zs = wg / (M / 128);
  f32x4 acc = {0, 0, 0, 0};
  const int ntiles = K / 64 / splits;
  if (LINEAR) {
    // the same bytes per workgroup, read as one contiguous range
    const size_t per = (size_t)M * K * 2 / gridDim.x;
    const char* base = (const char*)a + (size_t)wg * per;
    for (size_t off = (size_t)t * 16; off < per; off += 256 * 16 * UNR) {
      f32x4 v[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) v[u] = off + (size_t)u * 4096 < per ? *(const f32x4*)(base + off + (size_t)u * 4096) : acc;
#pragma unroll
      for (int u = 0; u < UNR; ++u) acc += v[u];
    }
  } else {
    for (int it = 0; it < ntiles; it += UNR / 4) {
      f32x4 v[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int tile = it + u / 4, c = t + 256 * (u % 4);          // 1024 chunks of 16 B per tile
        const int r = c >> 3, pc = c & 7;
        const size_t k0 = ((size_t)tile * splits + zs) * 64;
        v[u] = tile < ntiles ? *(const f32x4*)(a + (size_t)(mt * 128 + r) * K + k0 + pc * 8) : acc;
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) acc += v[u];
    }
  }
  if (acc[0] == 123.456f) out[0] = acc[1] + acc[2] + acc[3];
}

template <int UNR, bool LINEAR>
void run(const uint16_t* a, int M, int K, int splits, float* out, const char* name) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = M / 128 * splits;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((reader<UNR, LINEAR>), dim3(grid), dim3(256), 0, 0, a, M, K, splits, out);
  hipEventRecord(e0);
  const int n = 20;
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL((reader<UNR, LINEAR>), dim3(grid), dim3(256), 0, 0, a, M, K, splits, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-46s grid %4d: %7.1f us  %6.2f TB/s\n", name, grid, ms / n * 1e3, (double)M * K * 2 / (ms / n * 1e-3) / 1e12);
}

int main() {
  const int M = 512, K = 65536;
  uint16_t* a; float* out;
  hipMalloc(&a, (size_t)M * K * 2); hipMemset(a, 0, (size_t)M * K * 2);
  hipMalloc(&out, 16);
  run<4, false>(a, M, K, 64, out, "row tiles, 1 tile in flight, 64 slices");
  run<8, false>(a, M, K, 64, out, "row tiles, 2 tiles in flight, 64 slices");
  run<16, false>(a, M, K, 64, out, "row tiles, 4 tiles in flight, 64 slices");
  run<16, false>(a, M, K, 128, out, "row tiles, 4 tiles in flight, 128 slices");
  run<16, false>(a, M, K, 256, out, "row tiles, 4 tiles in flight, 256 slices");
  run<8, true>(a, M, K, 64, out, "linear, 8 x 16 B per lane in flight");
  run<16, true>(a, M, K, 256, out, "linear, 16 x 16 B per lane in flight");
  return 0;
}
