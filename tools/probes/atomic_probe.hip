// Probe: throughput of the f32 atomic flush of a stream-K weight-gradient tile (tools/probes; not product code).
//   G workgroups x 256 threads, each adds a 96 x 384 f32 tile (144 values per lane, MFMA D layout) into
//   `nreg` distinct regions (region = wg % nreg): nreg = 1 -> G-deep same-address chains, nreg = G -> no contention.
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o atomic_probe.bin atomic_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>   // 0: agent-scope relaxed (unsafe fp atomics), 1: workgroup scope, 2: plain store (no atomic), 3: linear lane order atomics
__global__ __launch_bounds__(256) void flush_kernel(float* out, int nreg, int ld, long long region_stride) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  float* C = out + (long long)(blockIdx.x % nreg) * region_stride;
  const float v = 1.0f + lane * 1e-3f;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 12; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wm * 48 + i * 16 + (lane >> 4) * 4 + r, col = wn * 192 + j * 16 + (lane & 15);
        float* p = C + row * ld + col;
        if (MODE == 0) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE == 1) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == 2) *p = v;
        else if (MODE == 3) {
          float* q = C + ((i * 12 + j) * 4 + r) * 256 + threadIdx.x;     // 64 consecutive dwords per wave instruction
          __hip_atomic_fetch_add(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {          // transposed tile: lane&15 runs along ROWS of a [384][96] result, a lane's 4 values are 4 consecutive columns
          float* q = C + (wn * 192 + j * 16 + (lane & 15)) * 96 + wm * 48 + i * 16 + (lane >> 4) * 4 + r;
          __hip_atomic_fetch_add(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
}

template <int MODE>
static float run(float* d, int G, int nreg, int iters) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(flush_kernel<MODE>, dim3(G), dim3(256), 0, 0, d, nreg, 384, 96LL * 384);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(flush_kernel<MODE>, dim3(G), dim3(256), 0, 0, d, nreg, 384, 96LL * 384);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters * 1e3f;
}

int main() {
  float* d; const int GMAX = 2048;
  CK(hipMalloc(&d, (size_t)GMAX * 96 * 384 * 4)); CK(hipMemset(d, 0, (size_t)GMAX * 96 * 384 * 4));
  const int Gs[] = {64, 256};
  const int NR[] = {1, 2, 4, 8, 16, 32, 64, 256, 1024};
  printf("# us per launch; each WG flushes 36864 f32 (147 KB)\n");
  printf("%6s %6s %10s %10s %10s %10s %10s\n", "G", "nreg", "agent", "wgscope", "store", "linear", "transp");
  for (int G : Gs)
    for (int nr : NR) {
      if (nr > G) continue;
      printf("%6d %6d %10.1f %10.1f %10.1f %10.1f %10.1f\n", G, nr, run<0>(d, G, nr, 20), run<1>(d, G, nr, 20), run<2>(d, G, nr, 20), run<3>(d, G, nr, 20), run<4>(d, G, nr, 20));
      fflush(stdout);
    }
  return 0;
}
