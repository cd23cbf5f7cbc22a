// Stochastic regularisers of the training step (HBM-bound elementwise work, one counter-based RNG):
//   * Keras Dropout / tfa-MHA attention dropout (reference trajNet.py:33,71,75,77,195,209,211): y = x * mask / (1 - p),
//     mask = U[0,1) >= p, one draw per element;
//   * DropPath (reference modules.py:137-151): y = x / keep * floor(keep + U), one draw per SAMPLE (inner = elements/sample),
//     i.e. the same rule with p = 1 - keep;  the residual add that always follows is fused (res).
// Randomness: Philox-4x32-10, key = (seed, step), counter = (draw/4, site).  Nothing is stored: backward re-derives the mask
// from (seed, step, site), and stj_dropout_mask exports it so the tests can hand the very same masks to the oracle.
// `state` is a device int64[2] = {seed, step}; stj_rng_advance bumps step on the stream (hipGraph replays get fresh masks).
#include "common.h"
#include "rng.h"

// one draw per element: a thread owns 4 consecutive elements == one Philox block
template <typename T>
__global__ __launch_bounds__(256) void dropout_elem_kernel(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y,
                                                           long long n, float p, float scale, const long long* __restrict__ state, int site) {
  const long long ng = (n + 3) / 4;
  for (long long gi = blockIdx.x * 256ll + threadIdx.x; gi < ng; gi += gridDim.x * 256ll) {
    bool k[4];
    keep4(state, site, gi, p, k);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long long i = gi * 4 + e;
      if (i < n) {
        float v = k[e] ? ldf(x + i) * scale : 0.f;
        if (res) v += ldf(res + i);
        stf(y + i, v);
      }
    }
  }
}
// one draw per run of `inner` elements (DropPath: inner = elements per sample)
template <typename T>
__global__ __launch_bounds__(256) void dropout_group_kernel(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y,
                                                            long long n, long long inner, float p, float scale,
                                                            const long long* __restrict__ state, int site) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) {
    const long long d = i / inner;
    bool k[4];
    keep4(state, site, d >> 2, p, k);
    float v = k[d & 3] ? ldf(x + i) * scale : 0.f;
    if (res) v += ldf(res + i);
    stf(y + i, v);
  }
}
// 16-byte vector path (n, inner multiples of the vector width, 16-byte aligned pointers): a thread owns VN consecutive
// elements = VN / 4 Philox blocks (per-element draws) or one shared draw (per-sample draws)
template <typename T, bool GROUP>
__global__ __launch_bounds__(256) void dropout_vec_kernel(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y,
                                                          long long n, long long inner, float p, float scale,
                                                          const long long* __restrict__ state, int site) {
  constexpr int VN = Vec<T>::N;
  const long long nv = n / VN;
  for (long long vi = blockIdx.x * 256ll + threadIdx.x; vi < nv; vi += gridDim.x * 256ll) {
    float v[VN], r[VN];
    ld16(x + vi * VN, v);
    if (res) ld16(res + vi * VN, r);
    bool k[VN];
    if constexpr (GROUP) {
      const long long d = vi * VN / inner;
      bool k4[4];
      keep4(state, site, d >> 2, p, k4);
#pragma unroll
      for (int e = 0; e < VN; ++e) k[e] = k4[d & 3];
    } else {
#pragma unroll
      for (int q = 0; q < VN / 4; ++q) keep4(state, site, vi * (VN / 4) + q, p, k + 4 * q);
    }
#pragma unroll
    for (int e = 0; e < VN; ++e) v[e] = (k[e] ? v[e] * scale : 0.f) + (res ? r[e] : 0.f);
    st16(y + vi * VN, v);
  }
}
__global__ __launch_bounds__(256) void dropout_mask_kernel(unsigned char* mask, long long ndraw, float p, const long long* state, int site) {
  const long long ng = (ndraw + 3) / 4;
  for (long long gi = blockIdx.x * 256ll + threadIdx.x; gi < ng; gi += gridDim.x * 256ll) {
    bool k[4];
    keep4(state, site, gi, p, k);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (gi * 4 + e < ndraw) mask[gi * 4 + e] = k[e] ? 1 : 0;
  }
}
__global__ void rng_advance_kernel(long long* state) { state[1] += 1; }

static inline int rng_grid(long long n) {
  long long b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

extern "C" int stj_rng_advance(long long* state, hipStream_t stream) {
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, stream, state);
  return stj_check_launch("stj_rng_advance");
}
// the same, and the advanced state copied to snap[2] (the snapshot a forward pass's backward kernels re-derive their masks from): one launch
// instead of the advance + a device copy at the head of every step
__global__ void rng_advance_snap_kernel(long long* state, long long* snap) { const long long s = state[1] + 1; state[1] = s; snap[0] = state[0]; snap[1] = s; }
extern "C" int stj_rng_advance_snap(long long* state, long long* snap, hipStream_t stream) {
  if (!state || !snap) { stj_set_error("stj_rng_advance_snap: null pointer"); return STJ_EINVAL; }
  hipLaunchKernelGGL(rng_advance_snap_kernel, dim3(1), dim3(1), 0, stream, state, snap);
  return stj_check_launch("stj_rng_advance_snap");
}
// y = (res ? res : 0) + keep(draw(i)) * x / (1 - p),  draw(i) = i / inner.  x, res, y: n elements of `dtype`; y may alias x.
extern "C" int stj_dropout(const void* x, const void* res, void* y, long long n, long long inner, float p, const long long* state,
                           int site, int dtype, hipStream_t stream) {
  if (n <= 0) return STJ_OK;
  if (!(p >= 0.f && p < 1.f) || inner < 1 || !state) { stj_set_error("stj_dropout: need 0 <= p < 1, inner >= 1, state != NULL"); return STJ_EINVAL; }
  const float scale = 1.0f / (1.0f - p);
  const int vn = stj_is16(dtype) ? 8 : 4;
  const bool vec = n % vn == 0 && (inner == 1 || inner % vn == 0) && !((((uintptr_t)x) | ((uintptr_t)res) | ((uintptr_t)y)) & 15);
  if (vec) {
    const int g = rng_grid(n / vn);
    if (dtype == STJ_BF16 && inner == 1) hipLaunchKernelGGL((dropout_vec_kernel<bf16, false>), dim3(g), dim3(256), 0, stream, (const bf16*)x, (const bf16*)res, (bf16*)y, n, inner, p, scale, state, site);
    else if (dtype == STJ_BF16) hipLaunchKernelGGL((dropout_vec_kernel<bf16, true>), dim3(g), dim3(256), 0, stream, (const bf16*)x, (const bf16*)res, (bf16*)y, n, inner, p, scale, state, site);
    else if (dtype == STJ_F16 && inner == 1) hipLaunchKernelGGL((dropout_vec_kernel<f16, false>), dim3(g), dim3(256), 0, stream, (const f16*)x, (const f16*)res, (f16*)y, n, inner, p, scale, state, site);
    else if (dtype == STJ_F16) hipLaunchKernelGGL((dropout_vec_kernel<f16, true>), dim3(g), dim3(256), 0, stream, (const f16*)x, (const f16*)res, (f16*)y, n, inner, p, scale, state, site);
    else if (inner == 1) hipLaunchKernelGGL((dropout_vec_kernel<float, false>), dim3(g), dim3(256), 0, stream, (const float*)x, (const float*)res, (float*)y, n, inner, p, scale, state, site);
    else hipLaunchKernelGGL((dropout_vec_kernel<float, true>), dim3(g), dim3(256), 0, stream, (const float*)x, (const float*)res, (float*)y, n, inner, p, scale, state, site);
  } else if (inner == 1) {
    const int g = rng_grid((n + 3) / 4);
    if (dtype == STJ_BF16) hipLaunchKernelGGL(dropout_elem_kernel<bf16>, dim3(g), dim3(256), 0, stream, (const bf16*)x, (const bf16*)res, (bf16*)y, n, p, scale, state, site);
    else if (dtype == STJ_F16) hipLaunchKernelGGL(dropout_elem_kernel<f16>, dim3(g), dim3(256), 0, stream, (const f16*)x, (const f16*)res, (f16*)y, n, p, scale, state, site);
    else hipLaunchKernelGGL(dropout_elem_kernel<float>, dim3(g), dim3(256), 0, stream, (const float*)x, (const float*)res, (float*)y, n, p, scale, state, site);
  } else {
    const int g = rng_grid(n);
    if (dtype == STJ_BF16) hipLaunchKernelGGL(dropout_group_kernel<bf16>, dim3(g), dim3(256), 0, stream, (const bf16*)x, (const bf16*)res, (bf16*)y, n, inner, p, scale, state, site);
    else if (dtype == STJ_F16) hipLaunchKernelGGL(dropout_group_kernel<f16>, dim3(g), dim3(256), 0, stream, (const f16*)x, (const f16*)res, (f16*)y, n, inner, p, scale, state, site);
    else hipLaunchKernelGGL(dropout_group_kernel<float>, dim3(g), dim3(256), 0, stream, (const float*)x, (const float*)res, (float*)y, n, inner, p, scale, state, site);
  }
  return stj_check_launch("stj_dropout");
}
// the keep decisions (1 = kept) of the first `ndraw` draws of a site, as bytes
extern "C" int stj_dropout_mask(unsigned char* mask, long long ndraw, float p, const long long* state, int site, hipStream_t stream) {
  if (ndraw <= 0) return STJ_OK;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(rng_grid((ndraw + 3) / 4)), dim3(256), 0, stream, mask, ndraw, p, state, site);
  return stj_check_launch("stj_dropout_mask");
}

// =====================================================================================================
// Fused Nadam step over the flat parameter / gradient buffers (reference train.py:197,224: tf.keras.optimizers.Nadam(1e-4),
// Keras defaults beta1 .9, beta2 .999, eps 1e-7; momentum-cache schedule mu_t = beta1 (1 - 0.5 * 0.96^(0.004 t))):
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2
//   w -= lr * ( (1-mu_t) / (1 - prod_t) * g + mu_{t+1} / (1 - prod_{t+1}) * m ) / ( sqrt(v / (1 - b2^t)) + eps )
// The scalar coefficients are computed by the host wrapper (they depend only on t); one pass over 4 flat f32 arrays.
// =====================================================================================================
__global__ __launch_bounds__(256) void nadam_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long long n, float lr, float b1, float b2, float eps,
                                                    float cg, float cm, float vhat_scale, float gscale) {
  const long long nv = n / 4;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < nv; i += gridDim.x * 256ll) {
    float4 W = reinterpret_cast<float4*>(w)[i], G = reinterpret_cast<const float4*>(g)[i];
    float4 M = reinterpret_cast<float4*>(m)[i], V = reinterpret_cast<float4*>(v)[i];
    float* wp = &W.x; float* gp = &G.x; float* mp = &M.x; float* vp = &V.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gg = gp[e] * gscale;
      mp[e] = b1 * mp[e] + (1.f - b1) * gg;
      vp[e] = b2 * vp[e] + (1.f - b2) * gg * gg;
      wp[e] -= lr * (cg * gg + cm * mp[e]) / (sqrtf(vp[e] * vhat_scale) + eps);
    }
    reinterpret_cast<float4*>(w)[i] = W; reinterpret_cast<float4*>(m)[i] = M; reinterpret_cast<float4*>(v)[i] = V;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long long i = nv * 4 + threadIdx.x;
    const float gg = g[i] * gscale;
    m[i] = b1 * m[i] + (1.f - b1) * gg;
    v[i] = b2 * v[i] + (1.f - b2) * gg * gg;
    w[i] -= lr * (cg * gg + cm * m[i]) / (sqrtf(v[i] * vhat_scale) + eps);
  }
}
// cg = (1 - mu_t) / (1 - prod_t), cm = mu_{t+1} / (1 - prod_{t+1}), vhat_scale = 1 / (1 - b2^t); gscale multiplies g first (e.g. 1).
extern "C" int stj_nadam_step(float* w, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps,
                              float cg, float cm, float vhat_scale, float gscale, hipStream_t stream) {
  if (n <= 0) return STJ_OK;
  if ((((uintptr_t)w) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) { stj_set_error("stj_nadam_step: buffers must be 16-byte aligned"); return STJ_EINVAL; }
  long long b = (n / 4 + 255) / 256;
  const int grid = (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
  hipLaunchKernelGGL(nadam_kernel, dim3(grid), dim3(256), 0, stream, w, g, m, v, n, lr, b1, b2, eps, cg, cm, vhat_scale, gscale);
  return stj_check_launch("stj_nadam_step");
}
