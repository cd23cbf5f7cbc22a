#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) short s4;
__global__ void probe(int ld, int mode, float* out) {
  __shared__ __attribute__((aligned(16))) uint16_t tile[64 * 72];
  for (int i = threadIdx.x; i < 64 * ld; i += 64) {
    int r = i / ld, c = i % ld;
    float v = mode == 0 ? (float)r : (float)c;
    tile[i] = (uint16_t)(__float_as_uint(v) >> 16);
  }
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, p = lane & 15;
  const uint16_t* a = tile + (0 + 4 * g + (p >> 2)) * ld + 0 + 4 * (p & 3);
  const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(a));
  const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(a + 16 * ld));
  for (int e = 0; e < 4; ++e) {
    out[lane * 8 + e] = __uint_as_float(((uint32_t)(uint16_t)lo[e]) << 16);
    out[lane * 8 + 4 + e] = __uint_as_float(((uint32_t)(uint16_t)hi[e]) << 16);
  }
}
int main() {
  float* d; hipMalloc(&d, 64 * 8 * 4);
  float h[512];
  for (int ld : {52, 56, 100, 72}) for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, ld, mode, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("ld %d mode %s\n", ld, mode == 0 ? "k-row (expect 4g+e | 16+4g+e)" : "col (expect lane&15)");
    for (int l : {0, 1, 2, 3, 4, 5, 15, 16, 17, 33, 63}) { printf(" lane %2d:", l); for (int e = 0; e < 8; ++e) printf(" %4.0f", h[l * 8 + e]); printf("\n"); }
  }
  return 0;
}
