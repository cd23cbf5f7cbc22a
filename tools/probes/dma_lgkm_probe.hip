// Does a wave's LDS traffic wait for its own pending LDS-DMA (global_load_lds_dwordx4)?  One wave per workgroup, one workgroup per CU;
// each iteration issues NP DMA pieces (1 KiB contiguous each) from addresses that miss every cache (a 1 GiB buffer, pieces 263 KiB apart), keeps at most KEEP
// pieces in flight (counted vmcnt), and -- depending on `mode` -- touches LDS:
//   0: nothing else                         1: ds_read_b32 of an unrelated LDS word + s_waitcnt lgkmcnt(0) AFTER the DMA issue
//   2: bare s_waitcnt lgkmcnt(0) after the DMA issue      3: ds_read_b32 issued BEFORE the DMA, waited for after it
//   4: ds_add_u32 (no return) after the DMA issue, no wait      5: s_memtime + s_waitcnt lgkmcnt(0) after the DMA issue (what a cycle stamp does)
// GATHER: the lanes of a piece read 14-piece pixels (12 x 16 bytes of a 192-byte pixel + 2 from a zero page), pixels 24 KiB apart -- the halo
// pattern of tools/probes/conv_ws5_async.hip.txt -- instead of 1 KiB contiguous
// Last part: the same stream with 1 / 2 / 4 issuing waves per CU, L2-resident (every workgroup walks the same 2 MiB) and cache-missing.
// MI355X: 17.9 / 34.5 / 60.1 GB/s per CU out of L2; 18.1 / 25.2 / 26.4 GB/s per CU cache-missing (4.62 / 6.45 / 6.75 TB/s chip-wide).
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/dma_lgkm_probe.hip -o /tmp/dma_probe ; run: /tmp/dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ void glds16(const char* g, uint32_t lds_byte) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(lds_byte) : "memory");
}
__device__ uint4 zero_page[4];
template <int MODE, int NP, int KEEP, bool GATHER>
__global__ __launch_bounds__(256) void probe(const char* src, long long span, int iters, uint32_t* sink) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)lds);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
  const uint32_t word = lds0 + 120 * 1024 + 4 * threadIdx.x;
  uint32_t acc = 0;
  long long off = (span > (64LL << 20) ? ((long long)(blockIdx.x * nwv + wv) * 7919 * 4096) % span : (long long)wv * 263 * 1024 % span) + lane * 16;
  for (int it = 0; it < iters; ++it) {
    uint32_t v = 0;
    if (MODE == 3) asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(word) : "memory");
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const char* a = src + off;
      if (GATHER) { const int px = lane / 14, part = lane % 14; a = part < 12 ? src + (off - lane * 16) + px * 24576 + part * 16 : reinterpret_cast<const char*>(zero_page); }
      glds16(a, __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(wv * 24 + (it * NP + p) % 24) * 1024u));
      off += 1024LL * 263; if (off >= span) off -= span;
    }
    if (MODE == 1) asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(word) : "memory");
    if (MODE == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (MODE == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v) :: "memory");
    if (MODE == 4) asm volatile("ds_add_u32 %0, %1" :: "v"(word), "v"(1u) : "memory");
    if (MODE == 5) { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); v = (uint32_t)t & 1u; }
    acc += v;
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(KEEP) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (acc == 0xdeadbeef) sink[0] = acc;
}
template <int MODE, bool GATHER> float run(const char* src, long long span, uint32_t* sink, int iters, int nw = 1) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((probe<MODE, 6, 42, GATHER>), dim3(256), dim3(64 * nw), 128 * 1024, 0, src, span, 50, sink);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL((probe<MODE, 6, 42, GATHER>), dim3(256), dim3(64 * nw), 128 * 1024, 0, src, span, iters, sink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
  const long long span = 1LL << 30;
  char* src; uint32_t* sink;
  hipMalloc(&src, span + (1 << 20)); hipMemset(src, 1, span + (1 << 20)); hipMalloc(&sink, 64);
  const int iters = 2000;
  const char* names[6] = {"DMA only", "ds_read + lgkmcnt(0) after the DMA issue", "bare lgkmcnt(0) after the DMA issue", "ds_read before, waited after", "ds_add after, no wait", "s_memtime + lgkmcnt(0) after"};
  for (int rep = 0; rep < 2; ++rep) {
    float t[6] = {run<0, false>(src, span, sink, iters), run<1, false>(src, span, sink, iters), run<2, false>(src, span, sink, iters), run<3, false>(src, span, sink, iters), run<4, false>(src, span, sink, iters), run<5, false>(src, span, sink, iters)};
    for (int m = 0; m < 6; ++m) printf("contiguous pieces, mode %d (%s): %.1f ns per iteration (6 pieces, <= 42 in flight)\n", m, names[m], t[m] * 1e6 / iters);
    float u[3] = {run<0, true>(src, span, sink, iters), run<1, true>(src, span, sink, iters), run<5, true>(src, span, sink, iters)};
    printf("halo-pattern pieces: DMA only %.1f | + ds_read + lgkmcnt(0) %.1f | + s_memtime + lgkmcnt(0) %.1f ns per iteration\n", u[0] * 1e6 / iters, u[1] * 1e6 / iters, u[2] * 1e6 / iters);
  }
  // the same stream out of L2: every workgroup walks the SAME 2 MiB (a weight slice all CUs stage), contiguous 1 KiB pieces
  for (int rep = 0; rep < 2; ++rep) {
    const float t = run<0, false>(src, 2LL << 20, sink, iters);
    printf("L2-resident 2 MiB shared by all workgroups, DMA only: %.1f ns per 6-piece round = %.1f GB/s per CU, %.2f TB/s chip-wide\n", t * 1e6 / iters, 6144.0 / (t * 1e6 / iters), 256 * 6144.0 / (t * 1e6 / iters) / 1e3);
  }
  for (int nw = 1; nw <= 4; nw *= 2) {
    hipFuncSetAttribute((const void*)probe<0, 6, 42, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    const float t = run<0, false>(src, 2LL << 20, sink, iters, nw), u = run<0, false>(src, span, sink, iters, nw);
    printf("%d wave(s) per CU issuing: L2-resident %.1f GB/s per CU | cache-missing %.1f GB/s per CU (%.2f TB/s chip-wide)\n", nw, nw * 6144.0 / (t * 1e6 / iters), nw * 6144.0 / (u * 1e6 / iters), 256 * nw * 6144.0 / (u * 1e6 / iters) / 1e3);
  }
  return 0;
}
