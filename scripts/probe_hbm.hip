// Calibration probe (not part of the product): achievable bandwidth and rocprofv3 FETCH_SIZE /
// WRITE_SIZE readings of the access patterns the embedding kernels are made of, on known byte
// counts.  Build: hipcc --offload-arch=gfx950 -O3 scripts/probe_hbm.hip -o /tmp/probe_hbm
//   patterns (N random rows out of R, ROW bytes each, LG = ROW/16 lanes per row):
//     gather     read row -> coalesced write of N*ROW bytes        (forward)
//     scatter    coalesced read -> write row                       (pure scattered stores)
//     rmw        read row, modify, write row                       (backward update)
//     copy       streaming float4 copy (the 6.3 TB/s reference)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int LG>
__global__ void k_gather(const float4* __restrict__ tab, const int64_t* __restrict__ ids, float4* __restrict__ out, int64_t n) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n * LG; k += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = k / LG; const int c = k % LG;
    out[k] = tab[ids[j] * LG + c];
  }
}
template <int LG, int UNR>
__global__ void k_gather_u(const float4* __restrict__ tab, const int64_t* __restrict__ ids, float4* __restrict__ out, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k0 < n * LG; k0 += stride * UNR) {
    float4 v[UNR]; int64_t id[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) { const int64_t k = k0 + u * stride; id[u] = k < n * LG ? ids[k / LG] : 0; }
#pragma unroll
    for (int u = 0; u < UNR; ++u) { const int64_t k = k0 + u * stride; v[u] = tab[id[u] * LG + (k % LG)]; }
#pragma unroll
    for (int u = 0; u < UNR; ++u) { const int64_t k = k0 + u * stride; if (k < n * LG) out[k] = v[u]; }
  }
}
template <int LG>
__global__ void k_scatter(float4* __restrict__ tab, const int64_t* __restrict__ ids, const float4* __restrict__ in, int64_t n) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n * LG; k += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = k / LG; const int c = k % LG;
    tab[ids[j] * LG + c] = in[k];
  }
}
template <int LG>
__global__ void k_rmw(float4* __restrict__ tab, const int64_t* __restrict__ ids, int64_t n) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n * LG; k += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = k / LG; const int c = k % LG;
    float4 v = tab[ids[j] * LG + c];
    v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f;
    tab[ids[j] * LG + c] = v;
  }
}
// 128-byte row handled by 4 lanes: two float4 per lane at +0 and +64 bytes (the interleaved [w|m] row)
__global__ void k_rmw128_by4(float4* __restrict__ tab, const int64_t* __restrict__ ids, int64_t n) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n * 4; k += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = k / 4; const int c = k % 4;
    float4* r = tab + ids[j] * 8;
    float4 w = r[c], m = r[4 + c];
    m.x += w.x; m.y += w.y; m.z += w.z; m.w += w.w; w.x += 1.f;
    r[4 + c] = m; r[c] = w;
  }
}
__global__ void k_copy(const float4* __restrict__ a, float4* __restrict__ b, int64_t n) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) b[k] = a[k];
}

template <class F>
static float timeit(F f, int iters = 5) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for (int i = 0; i < iters; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / iters * 1e3f;
}

int main(int argc, char** argv) {
  const int64_t R = 40000000, N = argc > 1 ? atoll(argv[1]) : 1703936;  // rows of 128 B
  float4* tab; CK(hipMalloc(&tab, R * 128)); CK(hipMemset(tab, 0, R * 128));
  float4 *buf, *buf2; CK(hipMalloc(&buf, N * 128)); CK(hipMalloc(&buf2, N * 128)); CK(hipMemset(buf, 0, N * 128));
  std::vector<int64_t> h(N); uint64_t s = 88172645463325252ull;
  for (auto& x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (int64_t)(s % (uint64_t)R); }
  int64_t* ids; CK(hipMalloc(&ids, N * 8)); CK(hipMemcpy(ids, h.data(), N * 8, hipMemcpyHostToDevice));
  const int G = 2048 * 4, T = 256;
  auto rep = [&](const char* name, double bytes, float us) { printf("%-34s %8.1f us  %7.1f GB/s (algorithmic %.1f MB)\n", name, us, bytes / us / 1e3, bytes / 1e6); };
  rep("copy 2x%N*128B", 2.0 * N * 128, timeit([&] { hipLaunchKernelGGL(k_copy, dim3(G), dim3(T), 0, 0, buf, buf2, N * 8); }));
  rep("gather64 (R rows of 64B)", N * (64.0 + 64 + 8), timeit([&] { hipLaunchKernelGGL(k_gather<4>, dim3(G), dim3(T), 0, 0, tab, ids, buf, N); }));
  rep("gather64 unroll4", N * (64.0 + 64 + 8), timeit([&] { hipLaunchKernelGGL((k_gather_u<4, 4>), dim3(G), dim3(T), 0, 0, tab, ids, buf, N); }));
  rep("gather64 unroll8", N * (64.0 + 64 + 8), timeit([&] { hipLaunchKernelGGL((k_gather_u<4, 8>), dim3(G), dim3(T), 0, 0, tab, ids, buf, N); }));
  rep("gather128", N * (128.0 + 128 + 8), timeit([&] { hipLaunchKernelGGL(k_gather<8>, dim3(G), dim3(T), 0, 0, tab, ids, buf, N); }));
  rep("scatter64", N * (64.0 + 64 + 8), timeit([&] { hipLaunchKernelGGL(k_scatter<4>, dim3(G), dim3(T), 0, 0, tab, ids, buf, N); }));
  rep("scatter128", N * (128.0 + 128 + 8), timeit([&] { hipLaunchKernelGGL(k_scatter<8>, dim3(G), dim3(T), 0, 0, tab, ids, buf, N); }));
  rep("rmw64", N * (64.0 + 64 + 8), timeit([&] { hipLaunchKernelGGL(k_rmw<4>, dim3(G), dim3(T), 0, 0, tab, ids, N); }));
  rep("rmw128 (8 lanes x 16B)", N * (128.0 + 128 + 8), timeit([&] { hipLaunchKernelGGL(k_rmw<8>, dim3(G), dim3(T), 0, 0, tab, ids, N); }));
  rep("rmw128 (4 lanes x 2x16B)", N * (128.0 + 128 + 8), timeit([&] { hipLaunchKernelGGL(k_rmw128_by4, dim3(G), dim3(T), 0, 0, tab, ids, N); }));
  return 0;
}
