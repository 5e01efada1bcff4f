// Discovers the operand / result lane map of v_mfma_f32_16x16x1_4b_f32 on the device it runs on:
// A lane value 2 l + 1, B lane value 2^l, so a result (2 la + 1) 2^lb names the two source lanes exactly.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(float* o) {
  const int l = threadIdx.x;
  f16v acc = {0};
  acc = __builtin_amdgcn_mfma_f32_16x16x1f32((float)(2 * l + 1), ldexpf(1.f, l), acc, 0, 0, 0);
  for (int i = 0; i < 16; ++i) o[l * 16 + i] = acc[i];
}
int main() {
  float* d;
  hipMalloc(&d, 64 * 16 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[64 * 16];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 16; ++r) {
      int e;
      const float m = frexpf(h[l * 16 + r], &e);  // v = m 2^e, m in [0.5, 1)
      // strip factors of two from the integer value
      double v = h[l * 16 + r];
      int lb = 0;
      while (v > 1 && fmod(v, 2.0) == 0) { v /= 2; ++lb; }
      const int la = ((int)v - 1) / 2;
      // expected: block = r / 4, row i = 4 (l / 16) + r % 4, col j = l % 16; A lane = i + 16 block, B lane = j + 16 block
      const int blk = r / 4, i = 4 * (l / 16) + r % 4, j = l % 16;
      const int ok = (la == i + 16 * blk) && (lb == j + 16 * blk);
      bad += !ok;
      if (!ok && bad < 40) printf("lane %d reg %d: A lane %d, B lane %d (expected %d, %d)\n", l, r, la, lb, i + 16 * blk, j + 16 * blk);
      (void)m;
    }
  printf("mfma_16x16x1_4b layout: %s (%d mismatches)\n", bad ? "DIFFERENT from the assumed map" : "as assumed", bad);
  return 0;
}
