// Probe (not part of the product): do neighbouring 8-byte elements of one 128-B line, written by
// workgroups on DIFFERENT XCDs inside one launch, survive -- with and without kernels of another
// HIP stream starting and finishing meanwhile?  This is the store pattern of the tile-parallel
// heavy-bucket sort (pooled_bwd.hip); NOTES.md "Side-stream plan: wrong results" suspected it.
//   build: hipcc --offload-arch=gfx950 -O3 scripts/probe_xcd_lines.hip -o scripts/probe_xcd_lines.bin
//   run:   scripts/probe_xcd_lines.bin [iterations]
// Every configuration: R iterations of { (optional) reader kernel that leaves clean copies of the
// lines in every XCD's L2; writer kernel: workgroup w stores element i iff (i / run) % G == w, after
// (optionally) reading the whole previous contents like a tile worker reads its bucket; checker
// kernel in the same stream counts wrong elements }.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void k_reader(const uint2* __restrict__ buf, int n, unsigned* sink) {
  unsigned acc = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc += buf[i].x;
  if (acc == 0xFFFFFFFFu) *sink = acc;
}

__global__ void k_writer(uint2* __restrict__ buf, const uint2* __restrict__ prev, int n, int run, int preread,
                         unsigned epoch, unsigned* sink) {
  const int G = gridDim.x, w = blockIdx.x;
  unsigned acc = 0;
  if (preread)
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc += prev[i].x;  // the whole "bucket", like a tile worker
  // elements of this workgroup: blocks of `run` consecutive elements, every G-th block
  for (int blk = w; blk * run < n; blk += G)
    for (int j = threadIdx.x; j < run; j += blockDim.x) {
      const int i = blk * run + j;
      if (i < n) buf[i] = make_uint2((unsigned)i ^ (acc & 0u), epoch);
    }
  if (acc == 0xFFFFFFFFu) *sink = acc;
}

__global__ void k_check(const uint2* __restrict__ buf, int n, unsigned epoch, unsigned* bad) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint2 v = buf[i];
    if (v.x != (unsigned)i || v.y != epoch) atomicAdd(bad, 1u);
  }
}

__global__ void k_noise(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = a[i];
    v.x += 1.f;
    b[i] = v;
  }
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 200;
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  const int n = 40 * 1024;  // one heavy bucket: 40 tiles of 1024 lookups
  uint2 *buf, *prev;
  unsigned *bad, *sink;
  CK(hipMalloc(&buf, sizeof(uint2) * n));
  CK(hipMalloc(&prev, sizeof(uint2) * n));
  CK(hipMalloc(&bad, 4));
  CK(hipMalloc(&sink, 4));
  CK(hipMemset(prev, 1, sizeof(uint2) * n));
  const size_t nn = (size_t)16 << 20;  // 256 MB per noise buffer
  float4 *na, *nb;
  CK(hipMalloc(&na, nn * 16));
  CK(hipMalloc(&nb, nn * 16));
  CK(hipMemset(na, 0, nn * 16));
  for (int noise = 0; noise < 2; ++noise)
    for (int preread = 0; preread < 2; ++preread)
      for (int reader = 0; reader < 2; ++reader)
        for (int run : {1, 3, 16}) {
          unsigned total_bad = 0, bad_iters = 0;
          for (int it = 0; it < iters; ++it) {
            const unsigned epoch = 1000u * (unsigned)(noise * 4 + preread * 2 + reader) + (unsigned)it + 1u;
            CK(hipMemsetAsync(bad, 0, 4, s1));
            if (reader) hipLaunchKernelGGL(k_reader, dim3(1024), dim3(256), 0, s1, buf, n, sink);
            if (noise)
              for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(k_noise, dim3(2048), dim3(256), 0, s2, na, nb, nn / 8);
            hipLaunchKernelGGL(k_writer, dim3(40), dim3(256), 0, s1, buf, prev, n, run, preread, epoch, sink);
            hipLaunchKernelGGL(k_check, dim3(64), dim3(256), 0, s1, buf, n, epoch, bad);
            unsigned h = 0;
            CK(hipMemcpyAsync(&h, bad, 4, hipMemcpyDeviceToHost, s1));
            CK(hipStreamSynchronize(s1));
            total_bad += h;
            bad_iters += h ? 1 : 0;
          }
          CK(hipDeviceSynchronize());
          printf("noise %d preread %d reader %d run %2d: %u bad elements in %u of %d iterations\n", noise, preread,
                 reader, run, total_bad, bad_iters, iters);
          fflush(stdout);
        }
  return 0;
}
