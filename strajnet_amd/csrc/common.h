// Common device helpers for the STrajNet gfx950 (CDNA4 / MI355X) kernels.
// Wavefront = 64 lanes.  MFMA tiles used: v_mfma_f32_16x16x32_bf16 (bf16 storage mode), v_mfma_f32_16x16x32_f16 (fp16
// storage mode: the inference path of BASELINE config 4) and v_mfma_f32_16x16x4_f32 (exact-f32 parity mode).  C/D fragment map for both:
//   col = lane & 15, row = (lane >> 4) * 4 + reg.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define STJ_WAVE 64

struct bf16 { uint16_t v; };
struct f16 { uint16_t v; };    // IEEE binary16 storage; same fragment layouts as bf16, f32 accumulation

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(8))) float f32x8_t;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ float bf2f(uint16_t x) { return __uint_as_float(((uint32_t)x) << 16); }
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
// two f32 -> packed bf16 pair (lo = a, hi = b), round-to-nearest-even: one v_cvt_pk_bf16_f32 on gfx950
__device__ __forceinline__ uint32_t pack2bf(float a, float b) {
  f32x2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ uint16_t f2bf(float f) { return (uint16_t)(pack2bf(f, 0.f) & 0xffffu); }
// fp16 counterparts (round-to-nearest-even v_cvt_f16_f32; values beyond 65504 become inf, as in any fp16 pipeline)
__device__ __forceinline__ float h2f(uint16_t x) { return (float)__builtin_bit_cast(_Float16, x); }
__device__ __forceinline__ uint32_t pack2h(float a, float b) {
  f32x2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
}
__device__ __forceinline__ uint16_t f2h(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
// the two 16-bit storage types through one name: pack two f32 / unpack a 32-bit pair
template <typename T> __device__ __forceinline__ uint32_t pack2(float a, float b);
template <> __device__ __forceinline__ uint32_t pack2<bf16>(float a, float b) { return pack2bf(a, b); }
template <> __device__ __forceinline__ uint32_t pack2<f16>(float a, float b) { return pack2h(a, b); }
template <typename T> __device__ __forceinline__ void unpack2(uint32_t w, float& lo, float& hi);
template <> __device__ __forceinline__ void unpack2<bf16>(uint32_t w, float& lo, float& hi) {
  lo = __uint_as_float(w << 16); hi = __uint_as_float(w & 0xffff0000u);
}
template <> __device__ __forceinline__ void unpack2<f16>(uint32_t w, float& lo, float& hi) {
  const f16x2_t h = __builtin_bit_cast(f16x2_t, w);
  lo = (float)h[0]; hi = (float)h[1];
}

template <typename T> __device__ __forceinline__ float ldf(const T* p);
template <> __device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldf<bf16>(const bf16* p) { return bf2f(p->v); }
template <> __device__ __forceinline__ float ldf<f16>(const f16* p) { return h2f(p->v); }
template <typename T> __device__ __forceinline__ void stf(T* p, float x);
template <> __device__ __forceinline__ void stf<float>(float* p, float x) { *p = x; }
template <> __device__ __forceinline__ void stf<bf16>(bf16* p, float x) { p->v = f2bf(x); }
template <> __device__ __forceinline__ void stf<f16>(f16* p, float x) { p->v = f2h(x); }

// ---- vector (16-byte) global access converted to/from float ---------------------------
template <typename T> struct Vec;   // elements per 16 bytes
template <> struct Vec<float> { static constexpr int N = 4; };
template <> struct Vec<bf16> { static constexpr int N = 8; };
template <> struct Vec<f16> { static constexpr int N = 8; };

template <typename T> __device__ __forceinline__ void ld16(const T* p, float* out);
template <> __device__ __forceinline__ void ld16<float>(const float* p, float* out) {
  float4 v = *reinterpret_cast<const float4*>(p);
  out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
}
template <> __device__ __forceinline__ void ld16<bf16>(const bf16* p, float* out) {
  uint4 v = *reinterpret_cast<const uint4*>(p);
  uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    out[2 * i] = __uint_as_float(w[i] << 16);
    out[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}
template <> __device__ __forceinline__ void ld16<f16>(const f16* p, float* out) {
  const f32x8_t v = __builtin_convertvector(__builtin_bit_cast(f16x8_t, *reinterpret_cast<const uint4*>(p)), f32x8_t);
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = v[i];
}
template <typename T> __device__ __forceinline__ void st16(T* p, const float* in);
template <> __device__ __forceinline__ void st16<float>(float* p, const float* in) {
  *reinterpret_cast<float4*>(p) = make_float4(in[0], in[1], in[2], in[3]);
}
template <> __device__ __forceinline__ void st16<bf16>(bf16* p, const float* in) {
  *reinterpret_cast<uint4*>(p) = make_uint4(pack2bf(in[0], in[1]), pack2bf(in[2], in[3]), pack2bf(in[4], in[5]), pack2bf(in[6], in[7]));
}
template <> __device__ __forceinline__ void st16<f16>(f16* p, const float* in) {
  *reinterpret_cast<uint4*>(p) = make_uint4(pack2h(in[0], in[1]), pack2h(in[2], in[3]), pack2h(in[4], in[5]), pack2h(in[6], in[7]));
}

// 4 consecutive elements (8 bytes of bf16 / fp16, 16 bytes of f32)
__device__ __forceinline__ void st4(bf16* p, const float* v) { *reinterpret_cast<uint2*>(p) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3])); }
__device__ __forceinline__ void st4(float* p, const float* v) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void ld4(const bf16* p, float* v) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u); v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
}
__device__ __forceinline__ void st4(f16* p, const float* v) { *reinterpret_cast<uint2*>(p) = make_uint2(pack2h(v[0], v[1]), pack2h(v[2], v[3])); }
__device__ __forceinline__ void ld4(const f16* p, float* v) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  unpack2<f16>(u.x, v[0], v[1]); unpack2<f16>(u.y, v[2], v[3]);
}
__device__ __forceinline__ void ld4(const float* p, float* v) { const float4 u = *reinterpret_cast<const float4*>(p); v[0] = u.x; v[1] = u.y; v[2] = u.z; v[3] = u.w; }

// ---- activations -----------------------------------------------------------------------
enum { ACT_NONE = 0, ACT_GELU = 1, ACT_ELU = 2 };

__device__ __forceinline__ float gelu_f(float x) {   // tanh form (reference modules.py:18-29)
  const float k = 0.7978845608028654f;
  float u = k * (x + 0.044715f * x * x * x);
  return 0.5f * x * (1.f + tanhf(u));
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float k = 0.7978845608028654f;
  float x2 = x * x;
  float u = k * (x + 0.044715f * x * x2);
  float t = tanhf(u);
  float du = k * (1.f + 3.f * 0.044715f * x2);
  return 0.5f * (1.f + t) + 0.5f * x * (1.f - t * t) * du;
}
__device__ __forceinline__ float elu_f(float x) { return x > 0.f ? x : expm1f(x); }
// ELU for bf16-stored results: v_exp_f32 based (abs error ~1e-7 near 0, far below bf16 resolution)
__device__ __forceinline__ float elu_fast(float x) { return x > 0.f ? x : __expf(x) - 1.f; }
// branch-free ELU for MFMA epilogues: v_min, v_mul, v_exp, v_add, v_cmp, v_cndmask (exp2 of a non-positive argument needs no range fix-up)
__device__ __forceinline__ float elu_bf(float x) {
  const float e = __builtin_amdgcn_exp2f(fminf(x, 0.f) * 1.4426950408889634f) - 1.f;
  return x > 0.f ? x : e;
}
// ELU in four VALU instructions: max(x, min(exp(x), 1) - 1) -- for x > 0 the clamped exponential is 1 and the maximum is x, for x <= 0
// exp(x) - 1 >= x.  fmed3(e, 0, 1) folds into the CLAMP output modifier of v_exp_f32 (v_mul, v_exp clamp, v_add, v_max).
__device__ __forceinline__ float elu_c(float x) {
  const float e = __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(x * 1.4426950408889634f), 0.f, 1.f);
  return fmaxf(x, e - 1.f);
}
__device__ __forceinline__ float apply_act_fast(float x, int act) {
  return act == 2 ? elu_fast(x) : (act == 1 ? x * 0.5f * (1.f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x))) : x);
}
__device__ __forceinline__ float apply_act(float x, int act) {
  return act == ACT_GELU ? gelu_f(x) : (act == ACT_ELU ? elu_f(x) : x);
}

// ---- unary activations of stj_unary_fwd / _bwd (csrc/util.hip) and of the kernels that fuse them ----
enum { U_GELU = 1, U_ELU = 2, U_TANHS = 3 };
// The op is a template parameter (no per-element selection) and the bf16 kernels use a v_exp_f32 based tanh (abs error ~2e-7,
// two decades below bf16 resolution); the f32 parity mode keeps libm tanhf / expm1f.
template <bool FAST> __device__ __forceinline__ float tanh_sel(float u) {
  if constexpr (FAST) {
    const float e = __builtin_amdgcn_exp2f(fminf(u, 15.f) * 2.885390081777927f);       // e^(2u), clamped: tanh(15) == 1 in f32
    return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
  } else return tanhf(u);
}
template <int OP, bool FAST> __device__ __forceinline__ float unary_f(float x, float p0) {
  if constexpr (OP == U_GELU) {
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    return 0.5f * x * (1.f + tanh_sel<FAST>(u));
  } else if constexpr (OP == U_ELU) return FAST ? elu_bf(x) : elu_f(x);
  else return tanh_sel<FAST>(x) * p0;
}
template <int OP, bool FAST> __device__ __forceinline__ float unary_g(float dy, float s, float p0) {
  if constexpr (OP == U_GELU) {
    const float k = 0.7978845608028654f, x2 = s * s;
    const float t = tanh_sel<FAST>(k * (s + 0.044715f * s * x2));
    return dy * (0.5f * (1.f + t) + 0.5f * s * (1.f - t * t) * (k * (1.f + 3.f * 0.044715f * x2)));
  } else if constexpr (OP == U_ELU) return s > 0.f ? dy : dy * (s + 1.f);
  else { const float t = s / p0; return dy * p0 * (1.f - t * t); }
}

// ---- wave reductions (64 lanes) ----------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- MFMA tile engine ---------------------------------------------------------------------
// LDS operand tiles are "K-contiguous": A as [m][k], B as [n][k], leading dim `ld` elements.
// Frag<T>: per-lane operand registers for one 16x16xKSTEP MFMA.
template <typename T> struct Mma;
template <> struct Mma<float> {
  // one "step" = 16 k's = 4 x v_mfma_f32_16x16x4_f32.  Lane (row=lane&15, g=lane>>4) supplies k = k0+4g+j to the
  // j-th MFMA; A and B use the same k mapping, so the contraction is merely visited in a permuted order.
  static constexpr int KSTEP = 16;
  typedef f32x4 Frag;
  __device__ static __forceinline__ Frag load(const float* tile, int ld, int row0, int k0, int lane) {
    return *reinterpret_cast<const f32x4*>(tile + (row0 + (lane & 15)) * ld + k0 + (lane >> 4) * 4);
  }
  // generic strided read: element (row,k) at tile[row*sr + k*sk]
  __device__ static __forceinline__ Frag load_strided(const float* tile, int sr, int sk, int row0, int k0, int lane) {
    const float* p = tile + (row0 + (lane & 15)) * sr + (k0 + (lane >> 4) * 4) * sk;
    Frag f;
#pragma unroll
    for (int j = 0; j < 4; ++j) f[j] = p[j * sk];
    return f;
  }
  __device__ static __forceinline__ Frag load_tr(const float* tile, int ldt, int row0, int k0, int lane) {
    return load_strided(tile, 1, ldt, row0, k0, lane);
  }
  __device__ static __forceinline__ Frag from_global(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
  static constexpr int LANE_K = 4;   // consecutive k elements per lane per step
  __device__ static __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], c, 0, 0, 0);
    return c;
  }
};
// 16-bit storage types share the operand layout (8 consecutive k per lane); only the MFMA opcode differs
template <typename T> struct Mma16 {
  static constexpr int KSTEP = 32;
  typedef s16x8 Frag;
  // elements (row = lane&15, k = k0 + (lane>>4)*8 .. +8), 16-byte aligned ds_read_b128
  __device__ static __forceinline__ Frag load(const T* tile, int ld, int row0, int k0, int lane) {
    return *reinterpret_cast<const s16x8*>(tile + (row0 + (lane & 15)) * ld + k0 + (lane >> 4) * 8);
  }
  __device__ static __forceinline__ Frag load_strided(const T* tile, int sr, int sk, int row0, int k0, int lane) {
    const uint16_t* p = reinterpret_cast<const uint16_t*>(tile) + (row0 + (lane & 15)) * sr + (k0 + (lane >> 4) * 8) * sk;
    s16x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (short)p[j * sk];
    return f;
  }
  // fragment of an operand stored UN-transposed as [k][rows] (rows contiguous, row stride ldt elements, 8-byte aligned):
  // two ds_read_b64_tr_b16 (gfx950 LDS transpose read; semantics probed in tools/probes/tr16_probe.hip): lane p of a
  // 16-lane group reads 4 consecutive rows 4*(p%4).. of k-row (p/4); the hardware hands lane l row (l&15), k-rows 0..3.
  __device__ static __forceinline__ Frag load_tr(const T* tile, int ldt, int row0, int k0, int lane) {
    typedef __attribute__((ext_vector_type(4))) short s4;
    const int g = lane >> 4, p = lane & 15;
    const T* a = tile + (k0 + 8 * g + (p >> 2)) * ldt + row0 + 4 * (p & 3);
    const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(a));
    const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(a + 4 * ldt));
    return (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  }
  __device__ static __forceinline__ Frag from_global(const T* p) { return *reinterpret_cast<const s16x8*>(p); }
  static constexpr int LANE_K = 8;
};
template <> struct Mma<bf16> : Mma16<bf16> {
  __device__ static __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct Mma<f16> : Mma16<f16> {
  __device__ static __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
};

// sum over the 16 lanes of a DPP row (the lanes of equal lane >> 4); every lane gets the total.  quad_perm xor 1, xor 2, then
// row_half_mirror / row_mirror (the quads / halves already hold equal values)
__device__ __forceinline__ float row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));
  return v;
}

// accumulate a (FM*16) x (FN*16) wave tile over kk in [0,KT) from K-contiguous LDS tiles
template <typename T, int FM, int FN>
__device__ __forceinline__ void mma_tile(const T* As, int lda, const T* Bs, int ldb, int KT, int lane, f32x4 (&acc)[FM][FN]) {
  for (int k0 = 0; k0 < KT; k0 += Mma<T>::KSTEP) {
    typename Mma<T>::Frag a[FM], b[FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) a[i] = Mma<T>::load(As, lda, i * 16, k0, lane);
#pragma unroll
    for (int j = 0; j < FN; ++j) b[j] = Mma<T>::load(Bs, ldb, j * 16, k0, lane);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = Mma<T>::mma(a[i], b[j], acc[i][j]);
  }
}

// LDS row padding (elements) that keeps 16-byte alignment of rows
template <typename T> struct LdsPad;
template <> struct LdsPad<float> { static constexpr int P = 4; };
// 16 elements = 32 bytes: rows of (multiple of 64 B) + 32 B put consecutive rows 2 (mod 4) 16-byte slots apart, the only
// row strides for which the four 16-lane groups of a ds_read_b128 fragment read (rows = lane&15, +16 B per lane>>4) are
// conflict-free on gfx950 (MI355X_MICROARCH.md LDS table; +16-byte padding measured 48 % conflict cycles in the conv kernels)
template <> struct LdsPad<bf16> { static constexpr int P = 16; };
template <> struct LdsPad<f16> { static constexpr int P = 16; };

// ---- zero-padded clamped bilinear sampling (reference occu_metric.py:345-409 + tfa_image.py:87-173) ----
struct Bil {
  int y0, x0; float ay, ax; bool gy, gx;   // floor indices in the padded image, alphas, alpha-differentiable flags
};
__device__ __forceinline__ Bil bil_setup(float qx, float qy, int Hp, int Wp) {
  Bil r;
  float fy = fminf(fmaxf(0.f, floorf(qy)), (float)(Hp - 2));
  float fx = fminf(fmaxf(0.f, floorf(qx)), (float)(Wp - 2));
  float ay = qy - fy, ax = qx - fx;
  // TF clip gradients: max(0,a) passes a's gradient only for a > 0 (ties go to the constant); min(.,1) passes it for a <= 1
  r.gy = ay > 0.f && ay <= 1.f;
  r.gx = ax > 0.f && ax <= 1.f;
  r.ay = fminf(fmaxf(ay, 0.f), 1.f);
  r.ax = fminf(fmaxf(ax, 0.f), 1.f);
  r.y0 = (int)fy; r.x0 = (int)fx;
  return r;
}
// value of the zero-padded image at padded coords (y,x); img is [H][W] with element stride `es`
__device__ __forceinline__ float pad_at(const float* img, int H, int W, int es, int y, int x) {
  return (y >= 1 && y <= H && x >= 1 && x <= W) ? img[((y - 1) * W + (x - 1)) * es] : 0.f;
}


// ---- host-side error plumbing -------------------------------------------------------------
enum { STJ_OK = 0, STJ_EINVAL = -1, STJ_ELAUNCH = -2, STJ_EUNSUPPORTED = -3 };
enum { STJ_F32 = 0, STJ_BF16 = 1, STJ_F16 = 2 };
static inline bool stj_is16(int dtype) { return dtype == STJ_BF16 || dtype == STJ_F16; }
static inline bool stj_dtype_ok(int dtype) { return dtype == STJ_F32 || dtype == STJ_BF16 || dtype == STJ_F16; }
void stj_set_error(const char* fmt, ...);
int stj_check_launch(const char* what);

// A launcher's once-per-process state that is really once per DEVICE (hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the current
// device's copy of the code object; the CU count is the current device's): one slot per device ordinal, read / written for the current one.
template <typename V> struct PerDevice {
  V v[64] = {};
  V& ref() { int d = 0; if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) d = 0; return v[d]; }
  operator V() { return ref(); }
  PerDevice& operator=(V x) { ref() = x; return *this; }
};
