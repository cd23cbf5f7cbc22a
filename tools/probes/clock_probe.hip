// What s_memtime counts, and the shader clock under load: a pure MFMA loop (8 independent 16x16x32 bf16 accumulators per wave, 4 or 8 waves
// per CU, every CU busy), alone and with a second kernel streaming HBM beside it on another stream.  Prints TFLOP/s, s_memtime ticks per
// nanosecond of wall clock, and ticks per MFMA.  MI355X: 4 waves per CU 1740 TFLOP/s, 1.97 ticks per ns, 19 ticks per MFMA and SIMD; 8 waves
// per CU 1938 TFLOP/s, 1.16 ticks per ns, 10 ticks per MFMA and SIMD -- fewer ticks than the 16 cycles an MFMA occupies its pipe: s_memtime is
// not a shader-cycle counter here, stamps are good for shares inside one kernel only.  The same loop on 32x32x16: 1780-1790 TFLOP/s.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/clock_probe.hip -o tools/probes/clock_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ __launch_bounds__(512) void mfma32_loop(int iters, unsigned long long* ticks, float* sink) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  bf16x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
  if (s == 123.456f) sink[0] = s;
}
__global__ __launch_bounds__(512) void mfma_loop(int iters, unsigned long long* ticks, float* sink) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
  if (s == 123.456f) sink[0] = s;
}
__global__ __launch_bounds__(256) void stream_read(const uint4* src, long long n, float* sink) {
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) { const uint4 v = src[i]; s += __uint_as_float(v.x ^ v.w); }
  if (s == 123.456f) sink[0] = s;
}
int main() {
  unsigned long long* ticks; float* sink; uint4* big;
  const long long nbig = (1LL << 30) / 16;
  hipMalloc(&ticks, 256 * 8); hipMalloc(&sink, 64); hipMalloc(&big, nbig * 16); hipMemset(big, 1, nbig * 16);
  hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 40000;
  for (int waves = 4; waves <= 8; waves *= 2)
    for (int with_hbm = 0; with_hbm < 2; ++with_hbm) {
      hipLaunchKernelGGL(mfma_loop, dim3(256), dim3(64 * waves), 0, s1, 1000, ticks, sink);
      hipDeviceSynchronize();
      if (with_hbm) for (int r = 0; r < 40; ++r) hipLaunchKernelGGL(stream_read, dim3(2048), dim3(256), 0, s2, big, nbig, sink);
      hipEventRecord(e0, s1);
      hipLaunchKernelGGL(mfma_loop, dim3(256), dim3(64 * waves), 0, s1, iters, ticks, sink);
      hipEventRecord(e1, s1);
      hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h[256]; hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
      double tk = 0; for (int i = 0; i < 256; ++i) tk += h[i]; tk /= 256;
      const double flop = 256.0 * waves * iters * 8 * 16384.0;
      printf("%d waves/CU%s: %.0f TFLOP/s, s_memtime %.3f ticks per ns of the launch, %.2f ticks per MFMA per SIMD\n", waves, with_hbm ? " + HBM read stream beside it" : "",
             flop / ms / 1e9, tk / (ms * 1e6), tk / (iters * 8.0 * (waves / 4)));
    }
  for (int waves = 4; waves <= 8; waves *= 2) {
    hipLaunchKernelGGL(mfma32_loop, dim3(256), dim3(64 * waves), 0, s1, 1000, ticks, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0, s1);
    hipLaunchKernelGGL(mfma32_loop, dim3(256), dim3(64 * waves), 0, s1, iters, ticks, sink);
    hipEventRecord(e1, s1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("32x32x16: %d waves/CU: %.0f TFLOP/s\n", waves, 256.0 * waves * iters * 4 * 32768.0 / ms / 1e9);
  }
  return 0;
}
