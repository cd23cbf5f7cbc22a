// Probe: HBM throughput of a read-modify-write pass over an [F][H][W][48] bf16 tensor (96 bytes per pixel, 403 MB at F = 64, 256 x 256) as a function of
// the TILE SHAPE a persistent workgroup walks (tools/probes; not product code).  outconv_bwd_mfma2 walks 16 x 16 pixel tiles (16 runs of 1536
// contiguous bytes, 24.5 KB apart) and moves 4.45 TB/s; the element-wise kernels, which walk the tensor linearly, 5.5 - 6.5.
// One tile per pass, next tile's 6 x 16 bytes per thread prefetched into registers under the current tile's stores (the kernel's structure).
// build: hipcc --offload-arch=gfx950 -O3 -o tile_stream_probe.bin tile_stream_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int C = 48, H = 256, W = 256;

template <int TH, int TW, int DEPTH>      // TH x TW = 256 pixels; DEPTH tiles in flight per workgroup
__global__ __launch_bounds__(256, 3) void tile_rw(const uint4* __restrict__ X, uint4* __restrict__ Y, int F) {
  const int tiles_x = W / TW, tiles_y = H / TH, ntiles = F * tiles_x * tiles_y;
  const int tid = threadIdx.x;
  int rel[6];
#pragma unroll
  for (int u = 0; u < 6; ++u) { const int q = tid + u * 256, px = q / 6, ch = q % 6; rel[u] = ((px / TW) * W + px % TW) * 6 + ch; }
  uint4 p[DEPTH][6];
  auto base = [&](int t) { const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, f = t / (tiles_x * tiles_y); return ((long long)(f * H + ty * TH) * W + tx * TW) * 6; };
  int tile = blockIdx.x;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    const long long b = base(min(tile + d * (int)gridDim.x, ntiles - 1));
#pragma unroll
    for (int u = 0; u < 6; ++u) p[d][u] = X[b + rel[u]];
  }
  for (; tile < ntiles; tile += DEPTH * gridDim.x) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int t = tile + d * gridDim.x;
      uint4 v[6];
#pragma unroll
      for (int u = 0; u < 6; ++u) { v[u] = p[d][u]; v[u].x ^= 0x00010001u; }
      const long long bn = base(min(t + DEPTH * (int)gridDim.x, ntiles - 1));
#pragma unroll
      for (int u = 0; u < 6; ++u) p[d][u] = X[bn + rel[u]];
      if (t < ntiles) {
        const long long b = base(t);
#pragma unroll
        for (int u = 0; u < 6; ++u) Y[b + rel[u]] = v[u];
      }
    }
  }
}
// one tile per workgroup, no persistence: load 6, store 6.  OCC = launch-bounds occupancy hint; NT tiles per workgroup in sequence (no prefetch)
template <int TH, int TW, int OCC, int NT, int LDSKB = 0, int MODE = 0>
__global__ __launch_bounds__(256, OCC) void tile_once(const uint4* __restrict__ X, uint4* __restrict__ Y, int F) {
  const int tiles_x = W / TW, tiles_y = H / TH, ntiles = F * tiles_x * tiles_y;
  const int tid = threadIdx.x;
  if (LDSKB > 0) {                     // occupy LDS so that only 160 / LDSKB workgroups fit a CU; the tile goes THROUGH it (stage, barrier, read back)
    __shared__ uint4 pad[(LDSKB > 0 ? LDSKB : 1) * 64];
    for (int i = 0; i < NT; ++i) {
      const int t = blockIdx.x * NT + i;
      if (t >= ntiles) return;
      const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, f = t / (tiles_x * tiles_y);
      const long long b = ((long long)(f * H + ty * TH) * W + tx * TW) * 6;
      uint4 v[6];
#pragma unroll
      for (int u = 0; u < 6; ++u) { const int q = tid + u * 256, px = q / 6, ch = q % 6; v[u] = X[b + ((px / TW) * W + px % TW) * 6 + ch]; }
      if (MODE == 0) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 6; ++u) pad[tid + u * 256] = v[u];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 6; ++u) v[u] = pad[(tid + u * 256 + 7) % 1536];
      } else if (MODE == 1) {
        if (v[0].x == 0x12345678u) pad[tid] = v[0];       // (keeps the allocation)
        __syncthreads();
      } else if (MODE == 2) {
#pragma unroll
        for (int u = 0; u < 6; ++u) pad[tid + u * 256] = v[u];
#pragma unroll
        for (int u = 0; u < 6; ++u) v[u] = pad[tid + u * 256];
      } else {
        if (v[0].x == 0x12345678u) pad[tid] = v[0];
      }
#pragma unroll
      for (int u = 0; u < 6; ++u) { const int q = tid + u * 256, px = q / 6, ch = q % 6; uint4 o = v[u]; o.x ^= 0x00010001u; Y[b + ((px / TW) * W + px % TW) * 6 + ch] = o; }
    }
    return;
  }
  for (int i = 0; i < NT; ++i) {
    const int t = blockIdx.x * NT + i;
    if (t >= ntiles) return;
    const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, f = t / (tiles_x * tiles_y);
    const long long b = ((long long)(f * H + ty * TH) * W + tx * TW) * 6;
    uint4 v[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) { const int q = tid + u * 256, px = q / 6, ch = q % 6; v[u] = X[b + ((px / TW) * W + px % TW) * 6 + ch]; }
#pragma unroll
    for (int u = 0; u < 6; ++u) { const int q = tid + u * 256, px = q / 6, ch = q % 6; v[u].x ^= 0x00010001u; Y[b + ((px / TW) * W + px % TW) * 6 + ch] = v[u]; }
  }
}
// one 16 x 16 tile per workgroup, loads linear (16 bytes per lane, whole lines); stores in the MFMA D layout of outconv_bwd: lane (g, ln) of wave w
// writes 24 bytes (16 + 8) at pixel (4 w + rr, ln), byte 24 g, rr = 0..3 -- or (LIN) the linear pattern of the loads
template <bool LIN>
__global__ __launch_bounds__(256) void tile_once_d(const uint4* __restrict__ X, uint4* __restrict__ Y, int F) {
  const int tiles_x = W / 16, tiles_y = H / 16;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, ln = lane & 15;
  const int t = blockIdx.x;
  const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, f = t / (tiles_x * tiles_y);
  const long long b = ((long long)(f * H + ty * 16) * W + tx * 16) * 6;
  uint4 v[6];
#pragma unroll
  for (int u = 0; u < 6; ++u) { const int c = lane + u * 64; v[u] = X[b + ((4 * w + c / 96) * W) * 6 + c % 96]; }
  if (LIN) {
#pragma unroll
    for (int u = 0; u < 6; ++u) { const int c = lane + u * 64; uint4 o = v[u]; o.x ^= 0x00010001u; Y[b + ((4 * w + c / 96) * W) * 6 + c % 96] = o; }
  } else {
    char* yb = reinterpret_cast<char*>(Y + b);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      char* o = yb + ((long long)(4 * w + rr) * W + ln) * 96 + 24 * g;
      uint4 a = v[rr]; a.x ^= 0x00010001u;
      *reinterpret_cast<uint2*>(o) = make_uint2(a.x, a.y);
      *reinterpret_cast<uint2*>(o + 8) = make_uint2(a.z, a.w);
      *reinterpret_cast<uint2*>(o + 16) = make_uint2(v[4].x + rr, v[5].y);
    }
  }
}
template <bool LIN> static void run_once_d(const uint4* x, uint4* y, int F) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int G = F * (H / 16) * (W / 16);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((tile_once_d<LIN>), dim3(G), dim3(256), 0, 0, x, y, F);
  CK(hipEventRecord(e0));
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((tile_once_d<LIN>), dim3(G), dim3(256), 0, 0, x, y, F);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
  printf("once, wave strips, stores %s: %7.1f us  %5.2f TB/s\n", LIN ? "linear 16 B / lane   " : "24 B / lane (D layout)", ms * 1e3, 2.0 * F * H * W * C * 2 / ms / 1e9);
}
template <int TH, int TW, int OCC, int NT, int LDSKB = 0, int MODE = 0> static void run_once(const uint4* x, uint4* y, int F) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int G = (F * (H / TH) * (W / TW) + NT - 1) / NT;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((tile_once<TH, TW, OCC, NT, LDSKB, MODE>), dim3(G), dim3(256), 0, 0, x, y, F);
  CK(hipEventRecord(e0));
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((tile_once<TH, TW, OCC, NT, LDSKB, MODE>), dim3(G), dim3(256), 0, 0, x, y, F);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
  printf("once tile %3d x %3d occ %d tiles/wg %d lds %2d KB mode %d grid %5d: %7.1f us  %5.2f TB/s\n", TH, TW, OCC, NT, LDSKB, MODE, G, ms * 1e3, 2.0 * F * H * W * C * 2 / ms / 1e9);
}
// wave-specialised persistent form: waves 0..3 only LOAD (tile -> LDS, NB-deep ring), waves 4..7 only STORE (LDS -> tile).  vmcnt counts a
// wave's loads and stores in order, so a wave that does both waits for its previous tile's store acknowledgements before it sees its next
// tile's loads; here the loaders' waits cover loads only and the storers never wait on vmcnt at all.
template <int NB>
__global__ __launch_bounds__(512, 1) void tile_ws(const uint4* __restrict__ X, uint4* __restrict__ Y, int F) {
  constexpr int TH = 16, TW = 16;
  __shared__ uint4 ring[NB][1536];
  const int tiles_x = W / TW, tiles_y = H / TH, ntiles = F * tiles_x * tiles_y;
  const int tid = threadIdx.x & 255, role = threadIdx.x >> 8;
  int rel[6];
#pragma unroll
  for (int u = 0; u < 6; ++u) { const int q = tid + u * 256, px = q / 6, ch = q % 6; rel[u] = ((px / TW) * W + px % TW) * 6 + ch; }
  auto base = [&](int t) { const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, f = t / (tiles_x * tiles_y); return ((long long)(f * H + ty * TH) * W + tx * TW) * 6; };
  const int nmine = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  // iteration i: loaders fill slot i % NB with tile i (i < nmine); storers drain slot (i - 1) % NB (tile i - 1); one barrier per iteration
  uint4 v[6];
  if (role == 0 && nmine > 0) {
    const long long b = base(blockIdx.x);
#pragma unroll
    for (int u = 0; u < 6; ++u) v[u] = X[b + rel[u]];
  }
  for (int i = 0; i <= nmine; ++i) {
    if (role == 0) {
      if (i < nmine) {
#pragma unroll
        for (int u = 0; u < 6; ++u) ring[i % NB][tid + u * 256] = v[u];
        if (i + 1 < nmine) {
          const long long b = base(blockIdx.x + (i + 1) * gridDim.x);
#pragma unroll
          for (int u = 0; u < 6; ++u) v[u] = X[b + rel[u]];
        }
      }
    } else if (i > 0) {
      const long long b = base(blockIdx.x + (i - 1) * gridDim.x);
#pragma unroll
      for (int u = 0; u < 6; ++u) { uint4 o = ring[(i - 1) % NB][tid + u * 256]; o.x ^= 0x00010001u; Y[b + rel[u]] = o; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
}
template <int NB> static void run_ws(const uint4* x, uint4* y, int F, int G) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((tile_ws<NB>), dim3(G), dim3(512), 0, 0, x, y, F);
  CK(hipEventRecord(e0));
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((tile_ws<NB>), dim3(G), dim3(512), 0, 0, x, y, F);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
  printf("wave-specialised ring %d grid %4d: %7.1f us  %5.2f TB/s\n", NB, G, ms * 1e3, 2.0 * F * H * W * C * 2 / ms / 1e9);
}
__global__ __launch_bounds__(256) void linear_rw(const uint4* __restrict__ X, uint4* __restrict__ Y, long long n) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) { uint4 v = X[i]; v.x ^= 0x00010001u; Y[i] = v; }
}

template <int TH, int TW, int DEPTH> static void run(const uint4* x, uint4* y, int F, int G) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((tile_rw<TH, TW, DEPTH>), dim3(G), dim3(256), 0, 0, x, y, F);
  CK(hipEventRecord(e0));
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((tile_rw<TH, TW, DEPTH>), dim3(G), dim3(256), 0, 0, x, y, F);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
  const double bytes = 2.0 * F * H * W * C * 2;
  printf("tile %3d x %3d depth %d grid %4d: %7.1f us  %5.2f TB/s\n", TH, TW, DEPTH, G, ms * 1e3, bytes / ms / 1e9);
}
int main() {
  const int F = 64;
  const long long n16 = (long long)F * H * W * 6;
  uint4 *x, *y; CK(hipMalloc(&x, n16 * 16)); CK(hipMalloc(&y, n16 * 16));
  CK(hipMemset(x, 1, n16 * 16));
  run_once_d<true>(x, y, F); run_once_d<false>(x, y, F); run_once_d<true>(x, y, F); run_once_d<false>(x, y, F);
  for (int G : {256}) { run_ws<2>(x, y, F, G); }
  for (int G : {768, 2048}) {
    run<16, 16, 1>(x, y, F, G); run<4, 64, 1>(x, y, F, G);
  }
  run_once<16, 16, 1, 1>(x, y, F); run_once<16, 16, 3, 1>(x, y, F); run_once<16, 16, 8, 1>(x, y, F);
  run_once<16, 16, 8, 1, 24, 0>(x, y, F); run_once<16, 16, 8, 1, 24, 1>(x, y, F); run_once<16, 16, 8, 1, 24, 2>(x, y, F); run_once<16, 16, 8, 1, 24, 3>(x, y, F);
  run_once<16, 16, 8, 1, 40, 1>(x, y, F); run_once<16, 16, 8, 1, 40, 3>(x, y, F); run_once<16, 16, 8, 1, 4, 1>(x, y, F); run_once<16, 16, 8, 1, 4, 3>(x, y, F);
  run_once<16, 16, 8, 4>(x, y, F); run_once<16, 16, 8, 16>(x, y, F); run_once<16, 16, 3, 4>(x, y, F); run_once<4, 64, 8, 1>(x, y, F); run_once<4, 64, 8, 4>(x, y, F);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int G : {2048, 8192, 65536}) {
    hipLaunchKernelGGL(linear_rw, dim3(G), dim3(256), 0, 0, x, y, n16);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(linear_rw, dim3(G), dim3(256), 0, 0, x, y, n16);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
    printf("linear grid %5d: %7.1f us  %5.2f TB/s\n", G, ms * 1e3, 2.0 * n16 * 16 / ms / 1e9);
  }
  return 0;
}
