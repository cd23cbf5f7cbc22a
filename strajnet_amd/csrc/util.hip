// Error plumbing + streaming elementwise kernels (HBM-bound: 16-byte vector access, grid-stride).
#include "common.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

void stj_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
int stj_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    stj_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return STJ_ELAUNCH;
  }
  return STJ_OK;
}
extern "C" const char* stj_last_error(void) { return g_err; }
extern "C" int stj_abi_version(void) { return 1; }

// One 16-byte vector per thread up to 64 M vectors: short-lived workgroups stream faster than a grid-stride loop over a capped grid
// (tools/probes/tile_stream_probe.hip, 403 MB read + 403 MB written: 5.0-5.3 TB/s with 2048 workgroups looping, 6.0-6.2 one-shot).
#ifndef EW_GRID_CAP
#define EW_GRID_CAP 262144
#endif
static inline int ew_grid(long long n_vec) {
  long long b = (n_vec + 255) / 256;
  return (int)(b < 1 ? 1 : (b > EW_GRID_CAP ? EW_GRID_CAP : b));
}

// ---- dtype cast ---------------------------------------------------------------------------
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void cast_kernel(const TS* src, TD* dst, long long n) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) stf(dst + i, ldf(src + i));
}
extern "C" int stj_cast(const void* src, int sdtype, void* dst, int ddtype, long long n, hipStream_t stream) {
  if (n <= 0) return STJ_OK;
  int g = ew_grid(n);
  if (sdtype == STJ_F32 && ddtype == STJ_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16>), dim3(g), dim3(256), 0, stream, (const float*)src, (bf16*)dst, n);
  else if (sdtype == STJ_BF16 && ddtype == STJ_F32) hipLaunchKernelGGL((cast_kernel<bf16, float>), dim3(g), dim3(256), 0, stream, (const bf16*)src, (float*)dst, n);
  else if (sdtype == STJ_F32 && ddtype == STJ_F32) hipLaunchKernelGGL((cast_kernel<float, float>), dim3(g), dim3(256), 0, stream, (const float*)src, (float*)dst, n);
  else if (sdtype == STJ_BF16 && ddtype == STJ_BF16) hipLaunchKernelGGL((cast_kernel<bf16, bf16>), dim3(g), dim3(256), 0, stream, (const bf16*)src, (bf16*)dst, n);
  else if (sdtype == STJ_F32 && ddtype == STJ_F16) hipLaunchKernelGGL((cast_kernel<float, f16>), dim3(g), dim3(256), 0, stream, (const float*)src, (f16*)dst, n);
  else if (sdtype == STJ_F16 && ddtype == STJ_F32) hipLaunchKernelGGL((cast_kernel<f16, float>), dim3(g), dim3(256), 0, stream, (const f16*)src, (float*)dst, n);
  else if (sdtype == STJ_F16 && ddtype == STJ_F16) hipLaunchKernelGGL((cast_kernel<f16, f16>), dim3(g), dim3(256), 0, stream, (const f16*)src, (f16*)dst, n);
  else { stj_set_error("stj_cast: bad dtypes"); return STJ_EINVAL; }
  return stj_check_launch("stj_cast");
}

// ---- unary activations ----------------------------------------------------------------------
// op: 1 gelu (saved = x), 2 elu (saved = y), 3 tanh*scale (saved = y)   (enum U_*: common.h)

// (tanh_sel / unary_f / unary_g: common.h -- the fused FG-MSA offset head evaluates the same expressions)
template <typename T, int OP>
__global__ __launch_bounds__(256) void unary_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long long n, float p0) {
  constexpr int VN = Vec<T>::N;
  constexpr bool FAST = sizeof(T) == 2;
  const long long nv = n / VN;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < nv; i += gridDim.x * 256ll) {
    float v[VN];
    ld16(x + i * VN, v);
#pragma unroll
    for (int e = 0; e < VN; ++e) v[e] = unary_f<OP, FAST>(v[e], p0);
    st16(y + i * VN, v);
  }
  for (long long i = nv * VN + blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) stf(y + i, unary_f<OP, FAST>(ldf(x + i), p0));
}
template <typename T, int OP>
__global__ __launch_bounds__(256) void unary_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ s, T* __restrict__ dx, long long n, float p0) {
  constexpr int VN = Vec<T>::N;
  constexpr bool FAST = sizeof(T) == 2;
  const long long nv = n / VN;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < nv; i += gridDim.x * 256ll) {
    float a[VN], b[VN];
    ld16(dy + i * VN, a);
    ld16(s + i * VN, b);
#pragma unroll
    for (int e = 0; e < VN; ++e) a[e] = unary_g<OP, FAST>(a[e], b[e], p0);
    st16(dx + i * VN, a);
  }
  for (long long i = nv * VN + blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll)
    stf(dx + i, unary_g<OP, FAST>(ldf(dy + i), ldf(s + i), p0));
}
#define UNARY_LAUNCH(KERN, TT, ...)                                                                              \
  do {                                                                                                           \
    if (op == U_GELU) hipLaunchKernelGGL((KERN<TT, U_GELU>), dim3(g), dim3(256), 0, stream, __VA_ARGS__);        \
    else if (op == U_ELU) hipLaunchKernelGGL((KERN<TT, U_ELU>), dim3(g), dim3(256), 0, stream, __VA_ARGS__);     \
    else hipLaunchKernelGGL((KERN<TT, U_TANHS>), dim3(g), dim3(256), 0, stream, __VA_ARGS__);                    \
  } while (0)
extern "C" int stj_unary_fwd(const void* x, void* y, long long n, int op, float p0, int dtype, hipStream_t stream) {
  if (n <= 0) return STJ_OK;
  if (op < U_GELU || op > U_TANHS) { stj_set_error("stj_unary_fwd: bad op %d", op); return STJ_EINVAL; }
  if (((uintptr_t)x | (uintptr_t)y) & 15) { stj_set_error("stj_unary_fwd: pointers must be 16-byte aligned"); return STJ_EINVAL; }
  int g = ew_grid(n / 8);
  if (dtype == STJ_BF16) UNARY_LAUNCH(unary_fwd_kernel, bf16, (const bf16*)x, (bf16*)y, n, p0);
  else if (dtype == STJ_F16) UNARY_LAUNCH(unary_fwd_kernel, f16, (const f16*)x, (f16*)y, n, p0);
  else UNARY_LAUNCH(unary_fwd_kernel, float, (const float*)x, (float*)y, n, p0);
  return stj_check_launch("stj_unary_fwd");
}
extern "C" int stj_unary_bwd(const void* dy, const void* saved, void* dx, long long n, int op, float p0, int dtype, hipStream_t stream) {
  if (n <= 0) return STJ_OK;
  if (op < U_GELU || op > U_TANHS) { stj_set_error("stj_unary_bwd: bad op %d", op); return STJ_EINVAL; }
  if (((uintptr_t)dy | (uintptr_t)saved | (uintptr_t)dx) & 15) { stj_set_error("stj_unary_bwd: pointers must be 16-byte aligned"); return STJ_EINVAL; }
  int g = ew_grid(n / 8);
  if (dtype == STJ_BF16) UNARY_LAUNCH(unary_bwd_kernel, bf16, (const bf16*)dy, (const bf16*)saved, (bf16*)dx, n, p0);
  else if (dtype == STJ_F16) UNARY_LAUNCH(unary_bwd_kernel, f16, (const f16*)dy, (const f16*)saved, (f16*)dx, n, p0);
  else UNARY_LAUNCH(unary_bwd_kernel, float, (const float*)dy, (const float*)saved, (float*)dx, n, p0);
  return stj_check_launch("stj_unary_bwd");
}

// Backward of  y = ELU(pre) + r  [and y2 = y + r2]  (stj_upconv_fwd_res): g = dy (+ dy2), dpre = g * ELU'(pre) with the ELU output
// recovered as y - r (ELU' = 1 for u > 0, u + 1 otherwise: continuous, so the rounding of y does not matter).  g is written too when
// dy2 is given (it is the gradient of r; dy2 itself is the gradient of r2).
template <typename T>
__global__ __launch_bounds__(256) void elu_res_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ dy2, const T* __restrict__ y,
                                                          const T* __restrict__ r, T* __restrict__ dpre, T* __restrict__ gsum, long long n) {
  constexpr int VN = Vec<T>::N;
  const long long nv = n / VN;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < nv; i += gridDim.x * 256ll) {
    float g[VN], b[VN], c[VN];
    ld16(dy + i * VN, g);
    if (dy2) {
      ld16(dy2 + i * VN, b);
#pragma unroll
      for (int e = 0; e < VN; ++e) g[e] += b[e];
      __attribute__((aligned(16))) T rounded[VN];
      st16(rounded, g);
      *reinterpret_cast<uint4*>(gsum + i * VN) = *reinterpret_cast<const uint4*>(rounded);
      ld16(rounded, g);                              // the rounded sum, as the consumer of gsum sees it
    }
    ld16(y + i * VN, b);
    if (r) ld16(r + i * VN, c);                      // r == NULL: y is the ELU output itself
    else {
#pragma unroll
      for (int e = 0; e < VN; ++e) c[e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < VN; ++e) { const float u = b[e] - c[e]; g[e] *= u > 0.f ? 1.f : u + 1.f; }
    st16(dpre + i * VN, g);
  }
}
extern "C" int stj_elu_res_bwd(const void* dy, const void* dy2, const void* y, const void* r, void* dpre, void* gsum, long long n, int dtype,
                               hipStream_t stream) {
  if (n <= 0) return STJ_OK;
  if (n % 8) { stj_set_error("stj_elu_res_bwd: n must be a multiple of 8"); return STJ_EINVAL; }
  if ((dy2 == nullptr) != (gsum == nullptr)) { stj_set_error("stj_elu_res_bwd: dy2 and gsum go together"); return STJ_EINVAL; }
  if (((uintptr_t)dy | (uintptr_t)dy2 | (uintptr_t)y | (uintptr_t)r | (uintptr_t)dpre | (uintptr_t)gsum) & 15) { stj_set_error("stj_elu_res_bwd: pointers must be 16-byte aligned"); return STJ_EINVAL; }
  const int g = ew_grid(n / 8);
  if (dtype == STJ_BF16) hipLaunchKernelGGL(elu_res_bwd_kernel<bf16>, dim3(g), dim3(256), 0, stream, (const bf16*)dy, (const bf16*)dy2, (const bf16*)y, (const bf16*)r, (bf16*)dpre, (bf16*)gsum, n);
  else if (dtype == STJ_F16) hipLaunchKernelGGL(elu_res_bwd_kernel<f16>, dim3(g), dim3(256), 0, stream, (const f16*)dy, (const f16*)dy2, (const f16*)y, (const f16*)r, (f16*)dpre, (f16*)gsum, n);
  else hipLaunchKernelGGL(elu_res_bwd_kernel<float>, dim3(g), dim3(256), 0, stream, (const float*)dy, (const float*)dy2, (const float*)y, (const float*)r, (float*)dpre, (float*)gsum, n);
  return stj_check_launch("stj_elu_res_bwd");
}

// Backward junction of a decoder level whose skips are ELU outputs themselves (modules.py:750-765: y1 = ELU(upconv(x)) + r1, y2 = y1 + r2,
// r = ELU(time-collapsed Conv3D of an encoder stage)): g = dy1 (+ dy2, rounded like the separate add), and ALL THREE ELU' products in the
// one pass -- dpre = g ELU'(y), dr1 = g ELU'(r1), dr2 = dy2 ELU'(r2) -- instead of stj_elu_res_bwd followed by one stj_unary_bwd per skip
// (three more passes over [F,H,W,C] tensors at the two-skip level, one more launch per skip).  Same values as that sequence, bit for bit.
template <typename T>
__global__ __launch_bounds__(256) void skip_junction_bwd_kernel(const T* __restrict__ dy1, const T* __restrict__ dy2, const T* __restrict__ y,
                                                                const T* __restrict__ r1, const T* __restrict__ r2, T* __restrict__ dpre,
                                                                T* __restrict__ dr1, T* __restrict__ dr2, long long n) {
  constexpr int VN = Vec<T>::N;
  const long long nv = n / VN;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < nv; i += gridDim.x * 256ll) {
    float g[VN], b[VN], s[VN], o[VN];
    ld16(dy1 + i * VN, g);
    if (dy2) {
      ld16(dy2 + i * VN, b);
      ld16(r2 + i * VN, s);
#pragma unroll
      for (int e = 0; e < VN; ++e) { g[e] += b[e]; o[e] = b[e] * (s[e] > 0.f ? 1.f : s[e] + 1.f); }
      st16(dr2 + i * VN, o);
      __attribute__((aligned(16))) T rounded[VN];
      st16(rounded, g);
      ld16(rounded, g);                              // the rounded sum, as the consumers of a separate add would see it
    }
    ld16(y + i * VN, s);
#pragma unroll
    for (int e = 0; e < VN; ++e) o[e] = g[e] * (s[e] > 0.f ? 1.f : s[e] + 1.f);
    st16(dpre + i * VN, o);
    ld16(r1 + i * VN, s);
#pragma unroll
    for (int e = 0; e < VN; ++e) o[e] = g[e] * (s[e] > 0.f ? 1.f : s[e] + 1.f);
    st16(dr1 + i * VN, o);
  }
}
extern "C" int stj_skip_junction_bwd(const void* dy1, const void* dy2, const void* y, const void* r1, const void* r2, void* dpre, void* dr1,
                                     void* dr2, long long n, int dtype, hipStream_t stream) {
  if (n <= 0) return STJ_OK;
  if (n % 8) { stj_set_error("stj_skip_junction_bwd: n must be a multiple of 8"); return STJ_EINVAL; }
  if (!dy1 || !y || !r1 || !dpre || !dr1) { stj_set_error("stj_skip_junction_bwd: null pointer"); return STJ_EINVAL; }
  if ((dy2 == nullptr) != (r2 == nullptr) || (dy2 == nullptr) != (dr2 == nullptr)) { stj_set_error("stj_skip_junction_bwd: dy2, r2 and dr2 go together"); return STJ_EINVAL; }
  if (((uintptr_t)dy1 | (uintptr_t)dy2 | (uintptr_t)y | (uintptr_t)r1 | (uintptr_t)r2 | (uintptr_t)dpre | (uintptr_t)dr1 | (uintptr_t)dr2) & 15) {
    stj_set_error("stj_skip_junction_bwd: pointers must be 16-byte aligned"); return STJ_EINVAL;
  }
  if (!stj_dtype_ok(dtype)) { stj_set_error("stj_skip_junction_bwd: bad dtype %d", dtype); return STJ_EINVAL; }
  const int g = ew_grid(n / 8);
#define SJ_GO(TT) hipLaunchKernelGGL(skip_junction_bwd_kernel<TT>, dim3(g), dim3(256), 0, stream, (const TT*)dy1, (const TT*)dy2, (const TT*)y, (const TT*)r1, (const TT*)r2, (TT*)dpre, (TT*)dr1, (TT*)dr2, n)
  if (dtype == STJ_BF16) SJ_GO(bf16); else if (dtype == STJ_F16) SJ_GO(f16); else SJ_GO(float);
#undef SJ_GO
  return stj_check_launch("stj_skip_junction_bwd");
}

// ---- max over a middle axis: x[outer][T][C] -> y[outer][C], idx (argmax, int8-in-int32) ---------
// GlobalMaxPooling1D over the 11 time steps (reference trajNet.py:34,44).
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* x, T* y, int* idx, long long outer, int Tn, int C) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < outer * C; i += gridDim.x * 256ll) {
    long long o = i / C; int c = (int)(i % C);
    float best = -INFINITY; int bi = 0;
    for (int t = 0; t < Tn; ++t) {
      float v = ldf(x + (o * Tn + t) * C + c);
      if (v > best) { best = v; bi = t; }
    }
    stf(y + i, best);
    idx[i] = bi;
  }
}
// TF reduce_max gradient: split evenly among ties (indicators / num_selected).
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* dy, const T* x, const T* y, T* dx, long long outer, int Tn, int C) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < outer * C; i += gridDim.x * 256ll) {
    long long o = i / C; int c = (int)(i % C);
    const float m = ldf(y + i), g = ldf(dy + i);
    int cnt = 0;
    for (int t = 0; t < Tn; ++t) cnt += ldf(x + (o * Tn + t) * C + c) == m ? 1 : 0;
    const float gg = g / (float)(cnt > 0 ? cnt : 1);
    for (int t = 0; t < Tn; ++t) stf(dx + (o * Tn + t) * C + c, ldf(x + (o * Tn + t) * C + c) == m ? gg : 0.f);
  }
}
extern "C" int stj_maxpool_fwd(const void* x, void* y, int* idx, long long outer, int Tn, int C, int dtype, hipStream_t stream) {
  if (outer <= 0) return STJ_OK;
  int g = ew_grid(outer * C);
  if (dtype == STJ_BF16) hipLaunchKernelGGL(maxpool_fwd_kernel<bf16>, dim3(g), dim3(256), 0, stream, (const bf16*)x, (bf16*)y, idx, outer, Tn, C);
  else if (dtype == STJ_F16) hipLaunchKernelGGL(maxpool_fwd_kernel<f16>, dim3(g), dim3(256), 0, stream, (const f16*)x, (f16*)y, idx, outer, Tn, C);
  else hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(g), dim3(256), 0, stream, (const float*)x, (float*)y, idx, outer, Tn, C);
  return stj_check_launch("stj_maxpool_fwd");
}
extern "C" int stj_maxpool_bwd(const void* dy, const void* x, const void* y, void* dx, long long outer, int Tn, int C, int dtype, hipStream_t stream) {
  if (outer <= 0) return STJ_OK;
  int g = ew_grid(outer * C);
  if (dtype == STJ_BF16) hipLaunchKernelGGL(maxpool_bwd_kernel<bf16>, dim3(g), dim3(256), 0, stream, (const bf16*)dy, (const bf16*)x, (const bf16*)y, (bf16*)dx, outer, Tn, C);
  else if (dtype == STJ_F16) hipLaunchKernelGGL(maxpool_bwd_kernel<f16>, dim3(g), dim3(256), 0, stream, (const f16*)dy, (const f16*)x, (const f16*)y, (f16*)dx, outer, Tn, C);
  else hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(g), dim3(256), 0, stream, (const float*)dy, (const float*)x, (const float*)y, (float*)dx, outer, Tn, C);
  return stj_check_launch("stj_maxpool_bwd");
}

// ---- raw record decode (reference train.py:87-103 / inference.py:84-96 _parse_image_function: tf.io.decode_raw + reshape +
// centre crop + cast, per feature) --------------------------------------------------------------------------------------
// src: [n_outer][H][W][C] elements of `kind` exactly as the bytes sit in the TFRecord feature (0: bool / uint8 -> (v != 0),
// 1: int8, 2: float32, 3: float64); dst f32 [n_outer][Ho][Wo][C] = scale * src[:, y0:y0+Ho, x0:x0+Wo, :].
template <int KIND>
__global__ __launch_bounds__(256) void decode_raw_kernel(const void* __restrict__ src, float* __restrict__ dst, long long n_out, int H, int W,
                                                         int C, int y0, int x0, int Ho, int Wo, float scale) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n_out; i += gridDim.x * 256ll) {
    const int c = (int)(i % C); long long t = i / C;
    const int x = (int)(t % Wo); t /= Wo;
    const int y = (int)(t % Ho); const long long n = t / Ho;
    const long long s = ((n * H + y0 + y) * W + x0 + x) * C + c;
    float v;
    if constexpr (KIND == 0) v = reinterpret_cast<const unsigned char*>(src)[s] != 0 ? 1.f : 0.f;
    else if constexpr (KIND == 1) v = (float)reinterpret_cast<const signed char*>(src)[s];
    else if constexpr (KIND == 2) v = reinterpret_cast<const float*>(src)[s];
    else v = (float)reinterpret_cast<const double*>(src)[s];
    dst[i] = v * scale;
  }
}
extern "C" int stj_decode_raw(const void* src, int kind, float* dst, long long n_outer, int H, int W, int C, int y0, int x0, int Ho,
                              int Wo, float scale, hipStream_t stream) {
  if (n_outer <= 0) return STJ_OK;
  if (kind < 0 || kind > 3 || y0 < 0 || x0 < 0 || y0 + Ho > H || x0 + Wo > W || C <= 0) { stj_set_error("stj_decode_raw: bad kind / crop"); return STJ_EINVAL; }
  const long long n_out = n_outer * Ho * Wo * C;
  const int g = ew_grid(n_out);
  if (kind == 0) hipLaunchKernelGGL(decode_raw_kernel<0>, dim3(g), dim3(256), 0, stream, src, dst, n_out, H, W, C, y0, x0, Ho, Wo, scale);
  else if (kind == 1) hipLaunchKernelGGL(decode_raw_kernel<1>, dim3(g), dim3(256), 0, stream, src, dst, n_out, H, W, C, y0, x0, Ho, Wo, scale);
  else if (kind == 2) hipLaunchKernelGGL(decode_raw_kernel<2>, dim3(g), dim3(256), 0, stream, src, dst, n_out, H, W, C, y0, x0, Ho, Wo, scale);
  else hipLaunchKernelGGL(decode_raw_kernel<3>, dim3(g), dim3(256), 0, stream, src, dst, n_out, H, W, C, y0, x0, Ho, Wo, scale);
  return stj_check_launch("stj_decode_raw");
}

// ------------------------------------------------------------------------------------------------ Conv3D time-kernel collapse
// The decoder's Conv3D(8,1,1) SAME skips see the same frame at every time step (modules.py:750-765), so output time t only needs
// W_t = sum_{j = max(0,3-t)}^{min(7,10-t)} W[j].  fwd: W f32 [8][n] -> Wz T [8][n]; bwd: dW[j] += sum over
// the t whose window contains j of dWz[t] -- one small launch each instead of an 8x8 matmul, a cast and an add.
template <typename T>
__global__ __launch_bounds__(256) void time_collapse_kernel(const float* __restrict__ W, T* __restrict__ Wz, long long n) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) {
    float w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = W[j * n + i];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j >= 3 - t && j <= 10 - t) a += w[j];
      stf(Wz + t * n + i, a);
    }
  }
}
__global__ __launch_bounds__(256) void time_fold_kernel(const float* __restrict__ dWz, float* __restrict__ dW, long long n) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) {
    float g[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) g[t] = dWz[t * n + i];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float a = 0.f;
#pragma unroll
      for (int t = 0; t < 8; ++t)
        if (j >= 3 - t && j <= 10 - t) a += g[t];
      dW[j * n + i] += a;
    }
  }
}
// ---- trajNet input plumbing and branch sums (trajNet.py:125-187) -------------------------------------------------------------
// One launch instead of cat / != 0 / any / three casts / two strided copies: tr = [obs | occ] [B,A,Tn,8] f32 ->
//   x5 [B*A*Tn,5] (node features), v3 [B*A,3] (vector features of step 0), vt [B*A,Tn] (step valid: tr[...,0] != 0), cmi / cmf [B*A]
//   (agent valid: any step valid) as int32 and in the activation dtype.
template <typename T>
__global__ __launch_bounds__(256) void agent_prep_kernel(const float* __restrict__ obs, const float* __restrict__ occ, int n_obs, int n_occ, int B,
                                                         int Tn, T* __restrict__ x5, T* __restrict__ v3, int* __restrict__ vt,
                                                         int* __restrict__ cmi, T* __restrict__ cmf) {
  // one thread per (agent, step) -- a thread per agent walking its steps was a chain of 11 dependent round trips, 30 us on the
  // branch's critical path; the step-0 thread also scans the agent's 11 validity flags (independent loads)
  const int A = n_obs + n_occ;
  for (int it = blockIdx.x * 256 + threadIdx.x; it < B * A * Tn; it += gridDim.x * 256) {
    const int i = it / Tn, t = it % Tn;
    const int b = i / A, a = i % A;
    const float* src = a < n_obs ? obs + ((long long)b * n_obs + a) * Tn * 8 : occ + ((long long)b * n_occ + (a - n_obs)) * Tn * 8;
    const float4 lo = *reinterpret_cast<const float4*>(src + t * 8), hi = *reinterpret_cast<const float4*>(src + t * 8 + 4);
    T* o = x5 + (long long)it * 5;
    stf(o, lo.x); stf(o + 1, lo.y); stf(o + 2, lo.z); stf(o + 3, lo.w); stf(o + 4, hi.x);
    vt[it] = lo.x != 0.f;
    if (t == 0) {
      stf(v3 + i * 3, hi.y); stf(v3 + i * 3 + 1, hi.z); stf(v3 + i * 3 + 2, hi.w);
      int any = 0;
      for (int u = 0; u < Tn; ++u) any |= src[u * 8] != 0.f;
      cmi[i] = any;
      stf(cmf + i, any ? 1.f : 0.f);
    }
  }
}
// concat = enc * cm ; qin = concat + embed   (trajNet.py:166-170; enc [B,A,C], embed [A,C] broadcast over B, cm [B,A])
template <typename T>
__global__ __launch_bounds__(256) void agent_mix_fwd_kernel(const T* __restrict__ enc, const T* __restrict__ embed, const T* __restrict__ cm,
                                                            T* __restrict__ concat, T* __restrict__ qin, int B, int A, int C) {
  const long long n = (long long)B * A * C;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) {
    const int c = (int)(i % C); const long long ba = i / C; const int a = (int)(ba % A);
    T t;
    stf(&t, ldf(enc + i) * ldf(cm + ba));
    concat[i] = t;
    stf(qin + i, ldf(&t) + ldf(embed + (long long)a * C + c));       // the sum of the ROUNDED product, like the two separate ops
  }
}
// d_enc = (dconcat + dqin) * cm ; d_embed[a,c] = sum_b dqin[b,a,c]   (thread = (a,c), loop over b: no atomics)
template <typename T>
__global__ __launch_bounds__(256) void agent_mix_bwd_kernel(const T* __restrict__ dconcat, const T* __restrict__ dqin, const T* __restrict__ cm,
                                                            T* __restrict__ denc, T* __restrict__ dembed, int B, int A, int C) {
  for (int j = blockIdx.x * 256 + threadIdx.x; j < A * C; j += gridDim.x * 256) {
    const int a = j / C;
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
      const long long i = (long long)b * A * C + j;
      const float dq = dqin ? ldf(dqin + i) : 0.f;
      s += dq;
      stf(denc + i, ((dconcat ? ldf(dconcat + i) : 0.f) + dq) * ldf(cm + (long long)b * A + a));
    }
    if (dembed) stf(dembed + j, s);
  }
}
// out = enc + value + embed (trajNet.py:171); bwd: d_embed[a,c] = sum_b dout[b,a,c] (d_enc = d_value = dout)
template <typename T>
__global__ __launch_bounds__(256) void agent_sum_fwd_kernel(const T* __restrict__ enc, const T* __restrict__ value, const T* __restrict__ embed,
                                                            T* __restrict__ out, int B, int A, int C) {
  const long long n = (long long)B * A * C;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) {
    T t;
    stf(&t, ldf(enc + i) + ldf(value + i));                            // (enc + value) rounded, then + embed: the order of the two adds
    stf(out + i, ldf(&t) + ldf(embed + i % ((long long)A * C)));
  }
}
template <typename T>
__global__ __launch_bounds__(256) void agent_sum_bwd_kernel(const T* __restrict__ dout, T* __restrict__ dembed, int B, int A, int C) {
  for (int j = blockIdx.x * 256 + threadIdx.x; j < A * C; j += gridDim.x * 256) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += ldf(dout + (long long)b * A * C + j);
    stf(dembed + j, s);
  }
}
extern "C" int stj_agent_prep(const float* obs, const float* occ, int n_obs, int n_occ, int B, int Tn, void* x5, void* v3, int* vt, int* cmi,
                              void* cmf, int dtype, hipStream_t stream) {
  const int rows = B * (n_obs + n_occ) * Tn;
  if (rows <= 0) return STJ_OK;
  if (((uintptr_t)obs | (uintptr_t)occ) & 15) { stj_set_error("stj_agent_prep: obs / occ must be 16-byte aligned"); return STJ_EINVAL; }
  const int g = (rows + 255) / 256;
#define TT_ARGS(TT) obs, occ, n_obs, n_occ, B, Tn, (TT*)x5, (TT*)v3, vt, cmi, (TT*)cmf
  if (dtype == STJ_BF16) hipLaunchKernelGGL(agent_prep_kernel<bf16>, dim3(g), dim3(256), 0, stream, TT_ARGS(bf16));
  else if (dtype == STJ_F16) hipLaunchKernelGGL(agent_prep_kernel<f16>, dim3(g), dim3(256), 0, stream, TT_ARGS(f16));
  else if (dtype == STJ_F32) hipLaunchKernelGGL(agent_prep_kernel<float>, dim3(g), dim3(256), 0, stream, TT_ARGS(float));
  else { stj_set_error("stj_agent_prep: bad dtype %d", dtype); return STJ_EINVAL; }
#undef TT_ARGS
  return stj_check_launch("stj_agent_prep");
}
#define AGENT_EW(NAME, KERN, G, ARGS_BF, ARGS_H, ARGS_F)                                                             \
  if (dtype == STJ_BF16) hipLaunchKernelGGL(KERN<bf16>, dim3(G), dim3(256), 0, stream, ARGS_BF);                     \
  else if (dtype == STJ_F16) hipLaunchKernelGGL(KERN<f16>, dim3(G), dim3(256), 0, stream, ARGS_H);                   \
  else if (dtype == STJ_F32) hipLaunchKernelGGL(KERN<float>, dim3(G), dim3(256), 0, stream, ARGS_F);                 \
  else { stj_set_error(NAME ": bad dtype %d", dtype); return STJ_EINVAL; }                                           \
  return stj_check_launch(NAME)
extern "C" int stj_agent_mix_fwd(const void* enc, const void* embed, const void* cm, void* concat, void* qin, int B, int A, int C, int dtype,
                                 hipStream_t stream) {
  if ((long long)B * A * C <= 0) return STJ_OK;
  const int g = ew_grid((long long)B * A * C);
#define A_(TT) (const TT*)enc, (const TT*)embed, (const TT*)cm, (TT*)concat, (TT*)qin, B, A, C
  AGENT_EW("stj_agent_mix_fwd", agent_mix_fwd_kernel, g, A_(bf16), A_(f16), A_(float));
#undef A_
}
extern "C" int stj_agent_mix_bwd(const void* dconcat, const void* dqin, const void* cm, void* denc, void* dembed, int B, int A, int C, int dtype,
                                 hipStream_t stream) {
  if ((long long)B * A * C <= 0) return STJ_OK;
  const int g = (A * C + 255) / 256;
#define A_(TT) (const TT*)dconcat, (const TT*)dqin, (const TT*)cm, (TT*)denc, (TT*)dembed, B, A, C
  AGENT_EW("stj_agent_mix_bwd", agent_mix_bwd_kernel, g, A_(bf16), A_(f16), A_(float));
#undef A_
}
extern "C" int stj_agent_sum_fwd(const void* enc, const void* value, const void* embed, void* out, int B, int A, int C, int dtype, hipStream_t stream) {
  if ((long long)B * A * C <= 0) return STJ_OK;
  const int g = ew_grid((long long)B * A * C);
#define A_(TT) (const TT*)enc, (const TT*)value, (const TT*)embed, (TT*)out, B, A, C
  AGENT_EW("stj_agent_sum_fwd", agent_sum_fwd_kernel, g, A_(bf16), A_(f16), A_(float));
#undef A_
}
extern "C" int stj_agent_sum_bwd(const void* dout, void* dembed, int B, int A, int C, int dtype, hipStream_t stream) {
  if ((long long)B * A * C <= 0) return STJ_OK;
  const int g = (A * C + 255) / 256;
#define A_(TT) (const TT*)dout, (TT*)dembed, B, A, C
  AGENT_EW("stj_agent_sum_bwd", agent_sum_bwd_kernel, g, A_(bf16), A_(f16), A_(float));
#undef A_
}

// ---- tail of TrajNet.call (trajNet.py:171-187): out = enc + value + embed, then obs_norm on the first n0 agents of a scene and occ_norm on
// the rest, as ONE launch per direction.  (Layer by layer: the sum, two slice copies, two LayerNorm launches and a concat -- six dependent
// ~5 us launches at the very end of the agent chain, on which the cross-attention waits; backward ten.)  A wave owns a row (forward) or
// all B rows of one agent (backward: d_embed[a] and the parameter gradients accumulate in registers); a lane 2 adjacent columns of every 128.
template <typename T, int NJ>
__global__ __launch_bounds__(256) void agent_out_fwd_kernel(const T* __restrict__ enc, const T* __restrict__ value, const T* __restrict__ embed,
                                                            const float* __restrict__ g0, const float* __restrict__ b0, const float* __restrict__ g1,
                                                            const float* __restrict__ b1, T* __restrict__ out, T* __restrict__ y, float* __restrict__ mean,
                                                            float* __restrict__ rstd, int B, int A, int n0, float eps) {
  constexpr int C = 128 * NJ;
  const int lane = threadIdx.x & 63;
  const long long row = blockIdx.x * 4ll + (threadIdx.x >> 6);
  if (row >= (long long)B * A) return;
  const int a = (int)(row % A);
  const float* gm = a < n0 ? g0 : g1;
  const float* bt = a < n0 ? b0 : b1;
  float v[NJ][2];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int c = 128 * j + 2 * lane + e;
      T t;
      stf(&t, ldf(enc + row * C + c) + ldf(value + row * C + c));          // (enc + value) rounded, then + embed: the order of the two adds
      stf(&t, ldf(&t) + ldf(embed + (long long)a * C + c));
      stf(out + row * C + c, ldf(&t));
      v[j][e] = ldf(&t);
      s += v[j][e];
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mu = s * (1.f / C);
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int e = 0; e < 2; ++e) { const float d = v[j][e] - mu; q += d * d; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rs = rsqrtf(q * (1.f / C) + eps);
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int c = 128 * j + 2 * lane + e;
      stf(y + row * C + c, (v[j][e] - mu) * rs * gm[c] + bt[c]);
    }
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}
template <typename T, int NJ>
__global__ __launch_bounds__(256) void agent_out_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ out, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ g0, const float* __restrict__ g1,
                                                            T* __restrict__ dout, T* __restrict__ dembed, float* __restrict__ dg0, float* __restrict__ db0,
                                                            float* __restrict__ dg1, float* __restrict__ db1, int B, int A, int n0) {
  constexpr int C = 128 * NJ;
  const int lane = threadIdx.x & 63;
  const int a = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (a >= A) return;
  const float* gm = a < n0 ? g0 : g1;
  float de[NJ][2], dg[NJ][2], db[NJ][2], gv[NJ][2];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int e = 0; e < 2; ++e) { de[j][e] = dg[j][e] = db[j][e] = 0.f; gv[j][e] = gm[128 * j + 2 * lane + e]; }
  for (int b = 0; b < B; ++b) {
    const long long row = (long long)b * A + a;
    const float mu = mean[row], rs = rstd[row];
    float xh[NJ][2], t[NJ][2];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int c = 128 * j + 2 * lane + e;
        const float d = ldf(dy + row * C + c);
        xh[j][e] = (ldf(out + row * C + c) - mu) * rs;
        dg[j][e] += d * xh[j][e];
        db[j][e] += d;
        t[j][e] = d * gv[j][e];
        s1 += t[j][e]; s2 += t[j][e] * xh[j][e];
      }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    s1 *= (1.f / C); s2 *= (1.f / C);
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int c = 128 * j + 2 * lane + e;
        T r;
        stf(&r, rs * (t[j][e] - s1 - xh[j][e] * s2));
        dout[row * C + c] = r;
        de[j][e] += ldf(&r);                                               // d_embed sums the ROUNDED row gradients, like the stand-alone sum kernel
      }
  }
  float* dgp = a < n0 ? dg0 : dg1;
  float* dbp = a < n0 ? db0 : db1;
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int c = 128 * j + 2 * lane + e;
      stf(dembed + (long long)a * C + c, de[j][e]);
      atomicAdd(dgp + c, dg[j][e]);
      atomicAdd(dbp + c, db[j][e]);
    }
}
extern "C" int stj_agent_out_fwd(const void* enc, const void* value, const void* embed, const float* g0, const float* b0, const float* g1,
                                 const float* b1, void* out, void* y, float* mean, float* rstd, int B, int A, int n0, int C, float eps, int dtype,
                                 hipStream_t stream) {
  if ((long long)B * A <= 0) return STJ_OK;
  if (C != 384 || n0 < 0 || n0 > A) { stj_set_error("stj_agent_out_fwd: C must be 384 and 0 <= n0 <= A (got C = %d, n0 = %d)", C, n0); return STJ_EUNSUPPORTED; }
  const int g = (B * A + 3) / 4;
#define A_(TT) (const TT*)enc, (const TT*)value, (const TT*)embed, g0, b0, g1, b1, (TT*)out, (TT*)y, mean, rstd, B, A, n0, eps
  if (dtype == STJ_BF16) hipLaunchKernelGGL((agent_out_fwd_kernel<bf16, 3>), dim3(g), dim3(256), 0, stream, A_(bf16));
  else if (dtype == STJ_F16) hipLaunchKernelGGL((agent_out_fwd_kernel<f16, 3>), dim3(g), dim3(256), 0, stream, A_(f16));
  else if (dtype == STJ_F32) hipLaunchKernelGGL((agent_out_fwd_kernel<float, 3>), dim3(g), dim3(256), 0, stream, A_(float));
  else { stj_set_error("stj_agent_out_fwd: bad dtype %d", dtype); return STJ_EINVAL; }
#undef A_
  return stj_check_launch("stj_agent_out_fwd");
}
extern "C" int stj_agent_out_bwd(const void* dy, const void* out, const float* mean, const float* rstd, const float* g0, const float* g1, void* dout,
                                 void* dembed, float* dg0, float* db0, float* dg1, float* db1, int B, int A, int n0, int C, int dtype,
                                 hipStream_t stream) {
  if ((long long)B * A <= 0) return STJ_OK;
  if (C != 384 || n0 < 0 || n0 > A) { stj_set_error("stj_agent_out_bwd: C must be 384 and 0 <= n0 <= A (got C = %d, n0 = %d)", C, n0); return STJ_EUNSUPPORTED; }
  const int g = (A + 3) / 4;
#define A_(TT) (const TT*)dy, (const TT*)out, mean, rstd, g0, g1, (TT*)dout, (TT*)dembed, dg0, db0, dg1, db1, B, A, n0
  if (dtype == STJ_BF16) hipLaunchKernelGGL((agent_out_bwd_kernel<bf16, 3>), dim3(g), dim3(256), 0, stream, A_(bf16));
  else if (dtype == STJ_F16) hipLaunchKernelGGL((agent_out_bwd_kernel<f16, 3>), dim3(g), dim3(256), 0, stream, A_(f16));
  else if (dtype == STJ_F32) hipLaunchKernelGGL((agent_out_bwd_kernel<float, 3>), dim3(g), dim3(256), 0, stream, A_(float));
  else { stj_set_error("stj_agent_out_bwd: bad dtype %d", dtype); return STJ_EINVAL; }
#undef A_
  return stj_check_launch("stj_agent_out_bwd");
}

extern "C" int stj_time_collapse(const float* W, void* Wz, long long n, int dtype, hipStream_t stream) {
  if (n <= 0) return STJ_OK;
  const int g = ew_grid(n);
  if (dtype == STJ_BF16) hipLaunchKernelGGL(time_collapse_kernel<bf16>, dim3(g), dim3(256), 0, stream, W, (bf16*)Wz, n);
  else if (dtype == STJ_F16) hipLaunchKernelGGL(time_collapse_kernel<f16>, dim3(g), dim3(256), 0, stream, W, (f16*)Wz, n);
  else if (dtype == STJ_F32) hipLaunchKernelGGL(time_collapse_kernel<float>, dim3(g), dim3(256), 0, stream, W, (float*)Wz, n);
  else { stj_set_error("stj_time_collapse: bad dtype %d", dtype); return STJ_EINVAL; }
  return stj_check_launch("stj_time_collapse");
}
extern "C" int stj_time_fold(const float* dWz, float* dW, long long n, hipStream_t stream) {
  if (n <= 0) return STJ_OK;
  hipLaunchKernelGGL(time_fold_kernel, dim3(ew_grid(n)), dim3(256), 0, stream, dWz, dW, n);
  return stj_check_launch("stj_time_fold");
}

// Partial-gradient copies -> flat gradient buffer, and the copies re-zeroed for the next backward pass, in ONE launch (was index_add_ + a fill,
// two dependent launches between the last weight-gradient launch and the optimizer): g[idx[i]] += parts[i]; parts[i] = 0.
__global__ __launch_bounds__(256) void fold_parts_kernel(float* __restrict__ g, const long long* __restrict__ idx, float* __restrict__ parts, long long n) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) {
    const float v = parts[i];
    parts[i] = 0.f;
    if (v != 0.f) atomicAdd(g + idx[i], v);
  }
}
extern "C" int stj_fold_parts(float* g, const long long* idx, float* parts, long long n, hipStream_t stream) {
  if (n <= 0) return STJ_OK;
  if (!g || !idx || !parts) { stj_set_error("stj_fold_parts: null pointer"); return STJ_EINVAL; }
  hipLaunchKernelGGL(fold_parts_kernel, dim3(ew_grid(n)), dim3(256), 0, stream, g, idx, parts, n);
  return stj_check_launch("stj_fold_parts");
}

// ------------------------------------------------------------------------------------------------ host CRC-32C
// TFRecord framing (train.py:75-78 tf.data.TFRecordDataset) and the TF checkpoint bundle (train.py:358,366,372
// save_weights / load_weights) both checksum with CRC-32C (Castagnoli, reflected 0x82F63B78).  Slice-by-8 on the host: the
// 53 MB of weights or a 35 MB example take tens of milliseconds instead of the tens of seconds of a Python byte loop.
namespace {
struct Crc32cTables {
  uint32_t t[8][256];
  Crc32cTables() {
    for (uint32_t i = 0; i < 256; i++) {
      uint32_t c = i;
      for (int k = 0; k < 8; k++) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
      t[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; i++)
      for (int s = 1; s < 8; s++) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xFF];
  }
};
}  // namespace
extern "C" int stj_crc32c(const void* data, long long n, unsigned int* crc) {
  if (!crc || (n > 0 && !data) || n < 0) { stj_set_error("stj_crc32c: null pointer / negative size"); return STJ_EINVAL; }
  static const Crc32cTables T;
  const unsigned char* p = (const unsigned char*)data;
  uint32_t c = ~*crc;
  while (n > 0 && ((uintptr_t)p & 7)) { c = T.t[0][(c ^ *p++) & 0xFF] ^ (c >> 8); n--; }
  while (n >= 8) {
    uint64_t w;
    memcpy(&w, p, 8);
    w ^= c;
    c = T.t[7][w & 0xFF] ^ T.t[6][(w >> 8) & 0xFF] ^ T.t[5][(w >> 16) & 0xFF] ^ T.t[4][(w >> 24) & 0xFF] ^
        T.t[3][(w >> 32) & 0xFF] ^ T.t[2][(w >> 40) & 0xFF] ^ T.t[1][(w >> 48) & 0xFF] ^ T.t[0][w >> 56];
    p += 8; n -= 8;
  }
  while (n-- > 0) c = T.t[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
  *crc = ~c;
  return STJ_OK;
}
