// Fused PatchEmbed (reference modules.py:430-446: Conv2D k = 4, s = 4, VALID -> reshape -> LayerNormalization(1e-5)) and the
// stem sums / norms that follow it (modules.py:572-590: vec + maps -> all_patch_norm; :576-578: flow embedding -> flow_norm):
//
//   pre = cols(src) @ W + b                       cols never exists in HBM unless the caller asks for it (training: dW = cols^T dpre)
//   x2  = LN(pre; gamma, beta) [+ add]            "add" = the other embedding the caller sums this one with
//   y   = LN(x2; gamma2, beta2)                   optional second norm (all_patch_norm / flow_norm)
//
// One workgroup = 64 consecutive tokens (one token row of a 256-pixel raster line group): the 4 x 256 x Cin f32 pixels are read
// from the raster ONCE, coalesced along the line (the stride-2 channel pick of ogm[..., 0] and the f32 -> T cast happen on the
// way into LDS), the weights [16 Cin, 96] sit un-transposed in LDS and become MFMA fragments through the transpose read.  The
// product is taken as D[channel][token] (A = W^T, B = tokens): a lane then holds 24 channels of ONE token, so both LayerNorm
// reductions are two cross-lane adds (lanes l, l ^ 16, l ^ 32, l ^ 48 share a token) and every store is 4 consecutive channels.
// `pre` and `x2` are rounded to the storage type BEFORE they are normalised: the backward pass (stj_layernorm_bwd on the saved
// tensors) then sees exactly the rows the statistics were taken from, as in the layer-by-layer path this kernel replaces.
//
// Bounds (B = 8, cfg-256): 33.5 MB of rasters in, 3 x 6.3 MB of tokens out (+ the saved pre / x2 / cols in training): HBM-bound,
// ~10 us of traffic for what was 11 launches (3 im2col, 3 GEMM, 5 LN) at the head of the step's critical path.
#include "common.h"

namespace pe {

template <typename T> __device__ __forceinline__ float rnd(float x) { T t; stf(&t, x); return ldf(&t); }   // value as stored in T

constexpr int TOK = 64;       // tokens per workgroup
constexpr int CO = 96;        // embed_dim (kernel specialisation)

template <typename T> struct Geo {
  static constexpr int KS = Mma<T>::KSTEP;
  static constexpr int WPAD = sizeof(T) == 2 ? 16 : 4;      // 96 + 16 elements = 56 dwords = 8 * odd: conflict-free transpose reads
  static constexpr int LDW = CO + WPAD;
  __host__ __device__ static constexpr int kpad(int K) { return (K + KS - 1) / KS * KS; }
  __host__ __device__ static constexpr int lda(int K) { return kpad(K) + LdsPad<T>::P; }
  __host__ __device__ static constexpr size_t lds_bytes(int K) { return (size_t)(TOK * lda(K) + kpad(K) * LDW) * sizeof(T); }
};

struct Args {
  const float* src; const void* w; const float* bias; const float* gamma; const float* beta; const void* add;
  const float* gamma2; const float* beta2;
  void* cols; void* pre; void* x2; void* y; float* mean; float* rstd; float* mean2; float* rstd2;
  int B, H, W; long long pix_stride; int ch_stride; float eps;
};

template <typename T, int CIN>
__global__ __launch_bounds__(256) void patch_embed_fwd_kernel(Args a) {
  typedef Geo<T> G;
  constexpr int K = 16 * CIN, KP = G::kpad(K), LDA = G::lda(K), LDW = G::LDW, VN = Vec<T>::N;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* As = reinterpret_cast<T*>(smem);                  // [TOK][LDA]   tokens x k (k contiguous)
  T* Ws = As + TOK * LDA;                              // [KP][LDW]    k x channel (as stored: Keras kernel [4,4,Cin,96] flattened)
  __shared__ int tokbase[TOK];                         // element offset of a token's top-left pixel (< 2^31: checked by the launcher), -1 past the end
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Ph = a.H / 4, Pw = a.W / 4;
  const long long M = (long long)a.B * Ph * Pw, m0 = (long long)blockIdx.x * TOK;

  if (tid < TOK) {                                     // 32-bit arithmetic (a 64-bit division is a few hundred instructions)
    const unsigned m = (unsigned)m0 + tid;
    const unsigned pj = m % (unsigned)Pw, q = m / (unsigned)Pw, pi = q % (unsigned)Ph, b = q / (unsigned)Ph;
    tokbase[tid] = (long long)m < M ? (int)(((b * a.H + 4 * pi) * a.W + 4 * pj) * (unsigned)a.pix_stride) : -1;
  }
  __syncthreads();
  // ---- one round trip for everything the tile needs: all weight chunks and all token elements of a thread are loaded
  // UNCONDITIONALLY (clamped addresses, values selected afterwards: a guarded load is a branch the wait counters cannot look
  // past) before the first LDS store -- a load-convert-store loop exposes one HBM latency per element -------------------
  // element (t, d = dy * 4 + dx, c) <- src[((b * H + 4 pi + dy) * W + 4 pj + dx) * pix_stride + c * ch_stride]; a thread's
  // elements i = tid + 256 j are ordered (dy, t, dx, c): consecutive threads walk along a raster line
  {
    const T* w = reinterpret_cast<const T*>(a.w);
    constexpr int CPR = CO / VN, WN = (KP * CPR + 255) / 256;      // 16-byte weight chunks per row / per thread
    constexpr int PER = TOK * K / 256;                             // 44 / 12 / 8 token elements per thread
    static_assert(TOK * K % 256 == 0, "whole elements per thread");
    uint4 wv[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int i = min(tid + 256 * j, KP * CPR - 1), k = min(i / CPR, K - 1), c = (i % CPR) * VN;
      wv[j] = *reinterpret_cast<const uint4*>(w + (size_t)k * CO + c);
    }
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int i = tid + 256 * j, k = i / CPR, c = (i % CPR) * VN;
      if (i < KP * CPR) *reinterpret_cast<uint4*>(Ws + k * LDW + c) = k < K ? wv[j] : make_uint4(0, 0, 0, 0);
    }
    constexpr int NB = PER > 22 ? 22 : PER;                        // ogm: two batches of 22 (all 44 at once: 266 registers, one workgroup per CU)
    static_assert(PER % NB == 0, "whole batches");
#pragma unroll 1
    for (int j0 = 0; j0 < PER; j0 += NB) {
      float v[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int i = tid + 256 * (j0 + j);
        const int c = i % CIN; int r = i / CIN;
        const int dx = r & 3; r >>= 2;
        const int t = r % TOK, dy = r / TOK;
        const int base = tokbase[t];
        v[j] = a.src[(unsigned)(max(base, 0) + (dy * a.W + dx) * (int)a.pix_stride + c * a.ch_stride)];     // 32-bit offset from a scalar base
        v[j] = base < 0 ? 0.f : v[j];
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int i = tid + 256 * (j0 + j);
        const int c = i % CIN; int r = i / CIN;
        const int dx = r & 3; r >>= 2;
        const int t = r % TOK, dy = r / TOK;
        stf(As + t * LDA + (dy * 4 + dx) * CIN + c, v[j]);
      }
    }
  }
  if (KP > K)
    for (int i = tid; i < TOK * (KP - K); i += 256) stf(As + (i / (KP - K)) * LDA + K + i % (KP - K), 0.f);
  __syncthreads();

  // ---- cols (training only): the staged tile is 64 consecutive rows of the [M, K] matrix ------------------------------------
  if (a.cols) {
    T* cols = reinterpret_cast<T*>(a.cols);
    constexpr int CPR = K / VN;
    static_assert(K % VN == 0, "whole 16-byte chunks per cols row");
    for (int i = tid; i < TOK * CPR; i += 256) {
      const int t = i / CPR, c = (i % CPR) * VN;
      if (m0 + t < M) *reinterpret_cast<uint4*>(cols + (m0 + t) * K + c) = *reinterpret_cast<const uint4*>(As + t * LDA + c);
    }
  }

  // ---- D[channel][token] = W^T tokens: wave = 16 tokens x 96 channels ---------------------------------------------------------
  f32x4 acc[6];
#pragma unroll
  for (int n = 0; n < 6; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
  for (int k0 = 0; k0 < KP; k0 += G::KS) {
    const typename Mma<T>::Frag tok = Mma<T>::load(As, LDA, wave * 16, k0, lane);
#pragma unroll
    for (int n = 0; n < 6; ++n) acc[n] = Mma<T>::mma(Mma<T>::load_tr(Ws, LDW, n * 16, k0, lane), tok, acc[n]);
  }

  // ---- epilogue: lane = token (lane & 15) of the wave's 16, channels 16 n + 4 g + r ---------------------------------------------
  const int g = lane >> 4;
  const long long m = m0 + wave * 16 + (lane & 15);
  const bool live = m < M;
  float v[6][4];
#pragma unroll
  for (int n = 0; n < 6; ++n) {
    const float4 bv = *reinterpret_cast<const float4*>(a.bias + n * 16 + 4 * g);
    v[n][0] = acc[n][0] + bv.x; v[n][1] = acc[n][1] + bv.y; v[n][2] = acc[n][2] + bv.z; v[n][3] = acc[n][3] + bv.w;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[n][r] = rnd<T>(v[n][r]);       // the stored (and normalised) value is the rounded one
    if (a.pre && live) st4(reinterpret_cast<T*>(a.pre) + m * CO + n * 16 + 4 * g, v[n]);
  }
  auto norm = [&](const float* gamma, const float* beta, float* mean, float* rstd) {
    float s = 0.f;
#pragma unroll
    for (int n = 0; n < 6; ++n) s += (v[n][0] + v[n][1]) + (v[n][2] + v[n][3]);
    s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
    const float mu = s * (1.f / CO);
    float q = 0.f;
#pragma unroll
    for (int n = 0; n < 6; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float d = v[n][r] - mu; q += d * d; }
    q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
    const float rs = rsqrtf(q * (1.f / CO) + a.eps);
#pragma unroll
    for (int n = 0; n < 6; ++n) {
      const float4 gv = *reinterpret_cast<const float4*>(gamma + n * 16 + 4 * g), bv = *reinterpret_cast<const float4*>(beta + n * 16 + 4 * g);
      v[n][0] = (v[n][0] - mu) * rs * gv.x + bv.x; v[n][1] = (v[n][1] - mu) * rs * gv.y + bv.y;
      v[n][2] = (v[n][2] - mu) * rs * gv.z + bv.z; v[n][3] = (v[n][3] - mu) * rs * gv.w + bv.w;
    }
    if (mean && live && g == 0) { mean[m] = mu; rstd[m] = rs; }
  };
  norm(a.gamma, a.beta, a.mean, a.rstd);
  if (a.add && live) {
#pragma unroll
    for (int n = 0; n < 6; ++n) {
      float r[4];
      ld4(reinterpret_cast<const T*>(a.add) + m * CO + n * 16 + 4 * g, r);
      v[n][0] += r[0]; v[n][1] += r[1]; v[n][2] += r[2]; v[n][3] += r[3];
    }
  }
  if (a.gamma2) {
#pragma unroll
    for (int n = 0; n < 6; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[n][r] = rnd<T>(v[n][r]);
    if (a.x2 && live) {
#pragma unroll
      for (int n = 0; n < 6; ++n) st4(reinterpret_cast<T*>(a.x2) + m * CO + n * 16 + 4 * g, v[n]);
    }
    norm(a.gamma2, a.beta2, a.mean2, a.rstd2);
  }
  if (live) {
#pragma unroll
    for (int n = 0; n < 6; ++n) st4(reinterpret_cast<T*>(a.y) + m * CO + n * 16 + 4 * g, v[n]);
  }
}

template <typename T, int CIN>
static int launch(const Args& a, hipStream_t stream) {
  static PerDevice<bool> attr;
  const size_t lds = Geo<T>::lds_bytes(16 * CIN);
  if (!attr) {
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)patch_embed_fwd_kernel<T, CIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      stj_set_error("stj_patch_embed_fwd: cannot reserve %zu bytes of LDS", lds);
      return STJ_ELAUNCH;
    }
    attr = true;
  }
  const long long M = (long long)a.B * (a.H / 4) * (a.W / 4);
  hipLaunchKernelGGL((patch_embed_fwd_kernel<T, CIN>), dim3((unsigned)((M + TOK - 1) / TOK)), dim3(256), lds, stream, a);
  return stj_check_launch("stj_patch_embed_fwd");
}

template <typename T>
static int by_cin(const Args& a, int Cin, hipStream_t stream) {
  switch (Cin) {
    case 11: return launch<T, 11>(a, stream);
    case 3: return launch<T, 3>(a, stream);
    case 2: return launch<T, 2>(a, stream);
  }
  return STJ_EUNSUPPORTED;
}

}  // namespace pe

extern "C" int stj_patch_embed_supported(int Cin, int Cout, int dtype) {
  return (Cin == 11 || Cin == 3 || Cin == 2) && Cout == pe::CO && stj_dtype_ok(dtype);
}

extern "C" int stj_patch_embed_fwd(const float* src, const void* w, const float* bias, const float* gamma, const float* beta, const void* add,
                                   const float* gamma2, const float* beta2, void* cols, void* pre, void* x2, void* y, float* mean, float* rstd,
                                   float* mean2, float* rstd2, int B, int H, int W, int Cin, long long pix_stride, int ch_stride, int Cout,
                                   float eps, int dtype, hipStream_t stream) {
  if (!stj_patch_embed_supported(Cin, Cout, dtype)) {
    stj_set_error("stj_patch_embed_fwd: Cin %d / Cout %d / dtype %d not built (Cin in {11, 3, 2}, Cout = 96)", Cin, Cout, dtype);
    return STJ_EUNSUPPORTED;
  }
  if (!src || !w || !bias || !gamma || !beta || !y || B <= 0 || H <= 0 || W <= 0 || (H & 3) || (W & 3) || (!gamma2) != (!beta2) ||
      ((uintptr_t)w & 15) || ((uintptr_t)y & 15) || ((uintptr_t)add & 15) || ((uintptr_t)cols & 15) || ((uintptr_t)pre & 15) || ((uintptr_t)x2 & 15) ||
      ((uintptr_t)bias & 15) || ((uintptr_t)gamma & 15) || ((uintptr_t)beta & 15) || ((uintptr_t)gamma2 & 15) || ((uintptr_t)beta2 & 15)) {
    stj_set_error("stj_patch_embed_fwd: bad arguments (null / unaligned pointer, H or W not a multiple of 4)");
    return STJ_EINVAL;
  }
  if ((long long)B * H * W * pix_stride >= (1ll << 31)) {
    stj_set_error("stj_patch_embed_fwd: raster of %lld elements (32-bit element offsets inside the kernel)", (long long)B * H * W * pix_stride);
    return STJ_EUNSUPPORTED;
  }
  pe::Args a{src, w, bias, gamma, beta, add, gamma2, beta2, cols, pre, x2, y, mean, rstd, mean2, rstd2, B, H, W, pix_stride, ch_stride, eps};
  if (dtype == STJ_BF16) return pe::by_cin<bf16>(a, Cin, stream);
  if (dtype == STJ_F16) return pe::by_cin<f16>(a, Cin, stream);
  return pe::by_cin<float>(a, Cin, stream);
}
