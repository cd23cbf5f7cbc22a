// Probe: semantics of ds_read_b64_tr_b16 (gfx950).  Fills LDS with index values, every lane passes its own address,
// prints which LDS element each (lane, j) received.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void probe(int* out, int mode) {
  __shared__ __attribute__((aligned(16))) short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x;
  const int g = l >> 4, p = l & 15;
  // mode 0: canonical: lane p of group g points at element g*64 + p*4 (4 consecutive shorts)
  // mode 1: tile [pixel][channel] with row stride LD=72: lane p loads pixel (g*4 + p/4), channels 4*(p%4)..+3
  int addr = mode == 0 ? (g * 64 + p * 4) : ((g * 4 + p / 4) * 72 + 4 * (p % 4));
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + addr));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  int* d; hipMalloc(&d, 64 * 4 * sizeof(int));
  int h[256];
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" %5d", h[l * 4 + j]); printf("\n"); }
  }
  return 0;
}
