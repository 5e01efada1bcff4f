// What the fp32 matrix pipe sustains with nothing else in the kernel: v_mfma_f32_16x16x4_f32 on register operands only, C
// independent accumulator chains per wave, W waves per SIMD.  hipcc --offload-arch=gfx950 -O3 probe_mfma_peak.hip -o probe_mfma_peak.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int C>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float a0, float b0) {
  f32x4 acc[C];
  for (int c = 0; c < C; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x, b = b0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16 / C; ++u)
#pragma unroll
      for (int c = 0; c < C; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < C; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

// the same with operands that differ per lane and per instruction (random bits from memory, 16 A and 16 B registers in turn):
// what the pipe sustains when its inputs toggle like real data
__global__ __launch_bounds__(256) void probe_rand(float* out, const float* rnd, int iters) {
  f32x4 acc[4];
  for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a[16], b[16];
  for (int i = 0; i < 16; ++i) {
    a[i] = rnd[(threadIdx.x * 32 + i) & 8191];
    b[i] = rnd[(threadIdx.x * 32 + 16 + i + blockIdx.x) & 8191];
  }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[(u * 5) & 15], acc[u & 3], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

static void run_rand(int wgs_per_cu, int iters) {
  float *d, *r;
  hipMalloc(&d, 4096);
  hipMalloc(&r, 8192 * 4);
  std::vector<float> h(8192);
  unsigned x = 12345u;
  for (auto& v : h) {
    x = x * 1664525u + 1013904223u;
    v = ((int)(x >> 8) - (1 << 23)) / (float)(1 << 23) * 1e-3f;  // small magnitudes: the sums stay finite
  }
  hipMemcpy(r, h.data(), 8192 * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int grid = 256 * wgs_per_cu;
  hipLaunchKernelGGL(probe_rand, dim3(grid), dim3(256), 0, 0, d, r, 100);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe_rand, dim3(grid), dim3(256), 0, 0, d, r, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double flop = (double)grid * 4 * iters * 16 * 2048.0;
  printf("random operands, %d waves per SIMD, %d MFMAs per wave: %.3f ms -> %.1f TFLOP/s (%.3f of 157.3)\n", wgs_per_cu, iters * 16, best,
         flop / best / 1e9, flop / best / 1e9 / 157.3);
}

template <int C>
static void run(int wgs_per_cu, int iters) {
  float* d;
  hipMalloc(&d, 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int grid = 256 * wgs_per_cu;
  hipLaunchKernelGGL(probe<C>, dim3(grid), dim3(256), 0, 0, d, 100, 1.0f, 0.5f);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<C>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double flop = (double)grid * 4 * iters * 16 * 2048.0;
  printf("chains %d, %d waves per SIMD, %d MFMAs per wave: %.3f ms -> %.1f TFLOP/s (%.3f of 157.3)\n", C, wgs_per_cu, iters * 16, best,
         flop / best / 1e9, flop / best / 1e9 / 157.3);
  hipFree(d);
}

int main() {
  for (int w = 1; w <= 4; ++w) {
    run<1>(w, 4000);
    run<2>(w, 4000);
    run<4>(w, 4000);
    run<8>(w, 4000);
  }
  run<4>(2, 40000);  // what the clock does under a longer load
  run_rand(1, 4000);
  run_rand(2, 4000);
  run_rand(3, 4000);
  run_rand(2, 40000);
  return 0;
}
