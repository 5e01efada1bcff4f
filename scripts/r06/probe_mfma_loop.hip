// What costs the fp32 matrix pipe its time in a tile loop shaped like tzr_gemm_rows_kernel's (96 MFMAs per turn over 96 distinct
// weight registers, 4 accumulator chains, 2 waves per SIMD), one ingredient at a time: register operands only / operands read from
// LDS / + V vector-ALU instructions per turn / + an LDS store and a workgroup barrier per turn.
// hipcc --offload-arch=gfx950 -O3 probe_mfma_loop.hip -o probe_mfma_loop.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int LDSR, int VALU, int SYNC, int ACCZ>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void loop(float* out, const float* rnd, int turns) {
  __shared__ __attribute__((aligned(16))) float T[2][16 * 100];
  float W[4][24];
  for (int j = 0; j < 4; ++j)
    for (int s = 0; s < 24; ++s) W[j][s] = rnd[(threadIdx.x * 97 + j * 24 + s) & 8191];
  for (int i = threadIdx.x; i < 2 * 16 * 100; i += 256) (&T[0][0])[i] = rnd[i & 8191];
  __syncthreads();
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  float bsrc[24];
  for (int s = 0; s < 24; ++s) bsrc[s] = rnd[(threadIdx.x * 31 + s) & 8191];
  f32x4 keep = f32x4{0.f, 0.f, 0.f, 0.f};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = rnd[(threadIdx.x + i) & 8191];
  int buf = 0;
  for (int t = 0; t < turns; ++t) {
    f32x4 acc[4];
    for (int j = 0; j < 4; ++j) acc[j] = ACCZ ? f32x4{0.f, 0.f, 0.f, 0.f} : keep;
    const float* A = &T[buf][r * 100 + 4 * q];
#pragma unroll
    for (int e = 0; e < 6; ++e) {
      float a4[4];
      if (LDSR) {
        const float4 av = *reinterpret_cast<const float4*>(A + 16 * e);
        a4[0] = av.x, a4[1] = av.y, a4[2] = av.z, a4[3] = av.w;
      } else {
        a4[0] = bsrc[4 * e], a4[1] = bsrc[4 * e + 1], a4[2] = bsrc[4 * e + 2], a4[3] = bsrc[4 * e + 3];
      }
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(W[j][4 * e + c], a4[c], acc[j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < VALU; ++i) v[i & 7] = v[i & 7] * 1.0001f + v[(i + 1) & 7];  // (v_fma: a VALU instruction each)
    keep = f32x4{acc[0][0] + acc[1][1], acc[2][2], acc[3][3], acc[0][1]};
    if (SYNC) {
      T[buf ^ 1][threadIdx.x * 4 % 1600] = keep[0];
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      buf ^= 1;
    }
  }
  float s = keep[0] + keep[1] + keep[2] + keep[3];
  for (int i = 0; i < 8; ++i) s += v[i];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

// The same loop on real memory: N = 458 752 rows of 96 floats in, 256 floats out, tiles of 16 rows grid-strided over 512 workgroups
// (tzr_gemm_rows_kernel's shape 96 -> 256): GLD = the next tile's rows prefetched from global memory and staged into LDS,
// GST = the tile's 16 x 256 results stored, RELU = max(., 0) on them
template <int GLD, int GST, int RELU>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void loop2(float* out, const float* in, const float* rnd, int ntiles) {
  __shared__ __attribute__((aligned(16))) float T[2][16 * 100];
  float W[4][24];
  for (int j = 0; j < 4; ++j)
    for (int s = 0; s < 24; ++s) W[j][s] = rnd[(threadIdx.x * 97 + j * 24 + s) & 8191];
  for (int i = threadIdx.x; i < 2 * 16 * 100; i += 256) (&T[0][0])[i] = rnd[i & 8191];
  __syncthreads();
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4, wv = threadIdx.x >> 6;
  // the thread's two 16-byte pieces of a tile: 16 rows x 24 pieces = 384 pieces
  const int p0 = threadIdx.x, p1 = (threadIdx.x + 256) % 384;
  const unsigned o0 = (p0 / 24) * 96 + 4 * (p0 % 24), o1 = (p1 / 24) * 96 + 4 * (p1 % 24);
  const unsigned l0 = (p0 / 24) * 100 + 4 * (p0 % 24), l1 = (p1 / 24) * 100 + 4 * (p1 % 24);
  const unsigned so = r * 256 + wv * 64 + 4 * q;
  int buf = 0;
  f32x4 keep = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int tn = t + (int)gridDim.x < ntiles ? t + (int)gridDim.x : ntiles - 1;
    float4 n0, n1;
    if (GLD) {
      const float* base = in + (size_t)tn * 16 * 96;
      n0 = *reinterpret_cast<const float4*>(base + o0);
      n1 = *reinterpret_cast<const float4*>(base + o1);
    }
    f32x4 acc[4];
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* A = &T[buf][r * 100 + 4 * q];
#pragma unroll
    for (int e = 0; e < 6; ++e) {
      const float4 av = *reinterpret_cast<const float4*>(A + 16 * e);
      const float a4[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(W[j][4 * e + c], a4[c], acc[j], 0, 0, 0);
    }
    if (GLD) {
      *reinterpret_cast<float4*>(&T[buf ^ 1][l0]) = n0;
      *reinterpret_cast<float4*>(&T[buf ^ 1][l1]) = n1;
    }
    if (GST) {
      float* op = out + (size_t)t * 16 * 256 + so;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 o = acc[j];
        if (RELU) o = f32x4{fmaxf(o[0], 0.f), fmaxf(o[1], 0.f), fmaxf(o[2], 0.f), fmaxf(o[3], 0.f)};
        *reinterpret_cast<f32x4*>(op + 16 * j) = o;
      }
    } else {
      keep = f32x4{acc[0][0] + acc[1][1] + keep[0], acc[2][2], acc[3][3], acc[0][1]};
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    buf ^= 1;
  }
  float s = keep[0] + keep[1] + keep[2] + keep[3];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int GLD, int GST, int RELU>
static void run2(const char* what, float* out, const float* in, const float* r) {
  const int ntiles = 458752 / 16;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((loop2<GLD, GST, RELU>), dim3(512), dim3(256), 0, 0, out, in, r, ntiles);
  (void)hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 7; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((loop2<GLD, GST, RELU>), dim3(512), dim3(256), 0, 0, out, in, r, ntiles);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double flop = 2.0 * 458752 * 96 * 256;
  printf("%-70s %.1f us  %.3f of 157.3 TFLOP/s\n", what, best * 1e3, flop / best / 1e9 / 157.3);
}

template <int LDSR, int VALU, int SYNC, int ACCZ>
static void run(const char* what, float* d, const float* r, int turns) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((loop<LDSR, VALU, SYNC, ACCZ>), dim3(512), dim3(256), 0, 0, d, r, 10);
  (void)hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((loop<LDSR, VALU, SYNC, ACCZ>), dim3(512), dim3(256), 0, 0, d, r, turns);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double flop = 512.0 * 4 * turns * 96 * 2048.0;
  printf("%-70s %.3f ms  %.3f of 157.3 TFLOP/s\n", what, best, flop / best / 1e9 / 157.3);
}

int main() {
  float *d, *r;
  (void)hipMalloc(&d, 4096);
  (void)hipMalloc(&r, 8192 * 4);
  std::vector<float> h(8192);
  unsigned x = 12345u;
  for (auto& v : h) {
    x = x * 1664525u + 1013904223u;
    v = ((int)(x >> 8) - (1 << 23)) / (float)(1 << 23) * 1e-3f;
  }
  (void)hipMemcpy(r, h.data(), 8192 * 4, hipMemcpyHostToDevice);
  const int T = 2000;
  run<0, 0, 0, 0>("register operands, accumulators carried", d, r, T);
  run<0, 0, 0, 1>("register operands, accumulators zeroed per turn", d, r, T);
  run<1, 0, 0, 1>("operands from LDS (6 ds_read_b128 per turn)", d, r, T);
  run<0, 16, 0, 1>("registers + 16 VALU per turn", d, r, T);
  run<0, 32, 0, 1>("registers + 32 VALU per turn", d, r, T);
  run<0, 64, 0, 1>("registers + 64 VALU per turn", d, r, T);
  run<0, 0, 1, 1>("registers + LDS store + barrier per turn", d, r, T);
  run<1, 0, 1, 1>("LDS operands + LDS store + barrier per turn", d, r, T);
  run<1, 32, 1, 1>("LDS operands + 32 VALU + LDS store + barrier per turn", d, r, T);
  float *in, *out;
  (void)hipMalloc(&in, (size_t)458752 * 96 * 4);
  (void)hipMalloc(&out, (size_t)458752 * 256 * 4);
  (void)hipMemset(in, 0, (size_t)458752 * 96 * 4);
  run2<0, 0, 0>("96 -> 256 shape: LDS operands + barrier only", out, in, r);
  run2<1, 0, 0>("  + next tile's rows from global memory, staged", out, in, r);
  run2<0, 1, 0>("  + results stored (no loads)", out, in, r);
  run2<1, 1, 0>("  + loads and stores", out, in, r);
  run2<1, 1, 1>("  + loads, ReLU, stores", out, in, r);
  return 0;
}
