// Chained-operand MFMA helpers shared by the fused attention kernels (xattn_fused.hip, fgattn.hip): the accumulator fragments of one
// transposed product D[m = output column][n = token] are the B operand of the next (see swin_fused.hip for the scheme), plus the
// 48-deep head contraction (32 + 16 columns) used by the 42- and 48-wide attention heads.
#pragma once
#include "common.h"

namespace chain {
// ---- the tail of the 48-wide head contraction (32 + 16) for the 16-bit types ------------------------------------------------------
// The 16 tail columns go through the SAME 16x16x32 MFMA as the main part, on operands whose upper four k slots are zero.  (A
// v_mfma_f32_16x16x16_bf16 accumulating into the result of a v_mfma_f32_16x16x32_bf16 a few instructions earlier read stale
// values of the first two accumulator registers: hipcc (ROCm 7.2) puts no wait states between the two opcodes.  Found on the
// GPU as a timing-dependent error of S = K q^T that any extra instruction in front of it cured.)
typedef __attribute__((ext_vector_type(4))) short s16x4v;
__device__ __forceinline__ s16x8 zext8(s16x4v v) { return (s16x8){v[0], v[1], v[2], v[3], 0, 0, 0, 0}; }
// accumulator fragment (m = 4g + r) -> tail B operand (k slot e < 4: column 4g + e; upper slots zero)
template <typename T> __device__ __forceinline__ s16x8 pack4(const f32x4& d) {
  typedef __attribute__((ext_vector_type(4))) uint32_t u4;
  const u4 w = {pack2<T>(d[0], d[1]), pack2<T>(d[2], d[3]), 0u, 0u};
  return __builtin_bit_cast(s16x8, w);
}
// tail A operand, image [rows][ld] (k contiguous): row = row0 + (lane & 15), k = k0 + 4g .. +3
template <typename T> __device__ __forceinline__ s16x8 ldA16(const T* t, int ld, int row0, int k0, int lane) {
  return zext8(*reinterpret_cast<const s16x4v*>(t + (row0 + (lane & 15)) * ld + k0 + (lane >> 4) * 4));
}
// tail A operand, image [k][ldt] (rows contiguous): one ds_read_b64_tr_b16
template <typename T> __device__ __forceinline__ s16x8 ldA16_tr(const T* t, int ldt, int row0, int k0, int lane) {
  const int g = lane >> 4, p = lane & 15;
  const T* a = t + (k0 + 4 * g + (p >> 2)) * ldt + row0 + 4 * (p & 3);
  return zext8(__builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4v*)(a)));
}

// ---- chained-operand fragments (same conventions as swin_fused.hip) ----------------------------------------------------------
template <typename T> struct Ch;
template <> struct Ch<float> {
  static constexpr int ND = 1;
  typedef f32x4 Frag;
  __device__ static __forceinline__ Frag from_acc(const f32x4* d) { return d[0]; }
  __device__ static __forceinline__ Frag ldA(const float* t, int ld, int row0, int k0, int lane) { return Mma<float>::load(t, ld, row0, k0, lane); }
  __device__ static __forceinline__ Frag ldA_tr(const float* t, int ldt, int row0, int k0, int lane) { return Mma<float>::load_tr(t, ldt, row0, k0, lane); }
  // chain-ordered B fragment of a global row (k = k0 + 4g .. +3)
  __device__ static __forceinline__ Frag ldB_row(const float* rowp, int k0, int lane) { return *reinterpret_cast<const f32x4*>(rowp + k0 + (lane >> 4) * 4); }
};
template <typename T> struct Ch16 {
  static constexpr int ND = 2;
  typedef s16x8 Frag;
  typedef __attribute__((ext_vector_type(4))) uint32_t u4;
  __device__ static __forceinline__ Frag from_acc(const f32x4* d) {
    const u4 w = {pack2<T>(d[0][0], d[0][1]), pack2<T>(d[0][2], d[0][3]), pack2<T>(d[1][0], d[1][1]), pack2<T>(d[1][2], d[1][3])};
    return __builtin_bit_cast(s16x8, w);
  }
  __device__ static __forceinline__ Frag ldA(const T* t, int ld, int row0, int k0, int lane) {
    const T* p = t + (row0 + (lane & 15)) * ld + k0 + (lane >> 4) * 4;
    const uint2 lo = *reinterpret_cast<const uint2*>(p), hi = *reinterpret_cast<const uint2*>(p + 16);
    const u4 w = {lo.x, lo.y, hi.x, hi.y};
    return __builtin_bit_cast(s16x8, w);
  }
  __device__ static __forceinline__ Frag ldA_tr(const T* t, int ldt, int row0, int k0, int lane) {
    const int g = lane >> 4, p = lane & 15;
    const T* a = t + (k0 + 4 * g + (p >> 2)) * ldt + row0 + 4 * (p & 3);
    const s16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4v*)(a));
    const s16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4v*)(a + 16 * ldt));
    return (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  }
  // chain-ordered B fragment of a global row: k = k0 + 4g + e (e < 4), k0 + 16 + 4g + (e - 4)
  __device__ static __forceinline__ Frag ldB_row(const T* rowp, int k0, int lane) {
    const T* p = rowp + k0 + (lane >> 4) * 4;
    const uint2 lo = *reinterpret_cast<const uint2*>(p), hi = *reinterpret_cast<const uint2*>(p + 16);
    const u4 w = {lo.x, lo.y, hi.x, hi.y};
    return __builtin_bit_cast(s16x8, w);
  }
};
template <> struct Ch<bf16> : Ch16<bf16> {};
template <> struct Ch<f16> : Ch16<f16> {};

// the 48-deep chained B operand of one head (q^T, O^T, dO^T, dq^T: k = head column, n = token)
template <typename T> struct HeadOp;
template <> struct HeadOp<float> {
  f32x4 s[3];
  __device__ __forceinline__ void from_acc(const f32x4* d) { s[0] = d[0]; s[1] = d[1]; s[2] = d[2]; }
  __device__ __forceinline__ void from_row(const float* p, int lane) {        // p: the token's row, at the head's first column
#pragma unroll
    for (int i = 0; i < 3; ++i) s[i] = *reinterpret_cast<const f32x4*>(p + 16 * i + (lane >> 4) * 4);
  }
};
template <typename T> struct HeadOp16 {
  s16x8 m, t;
  __device__ __forceinline__ void from_acc(const f32x4* d) { m = Ch<T>::from_acc(d); t = pack4<T>(d[2]); }
  __device__ __forceinline__ void from_row(const T* p, int lane) {
    m = Ch<T>::ldB_row(p, 0, lane);
    t = zext8(*reinterpret_cast<const s16x4v*>(p + 32 + (lane >> 4) * 4));
  }
};
template <> struct HeadOp<bf16> : HeadOp16<bf16> {};
template <> struct HeadOp<f16> : HeadOp16<f16> {};
// a += A[m = row0 + ..][k = head column] b, A from an image [rows][ld] (k contiguous)
template <typename T>
__device__ __forceinline__ f32x4 k48_rows(const T* t, int ld, int row0, const HeadOp<T>& b, int lane, f32x4 a) {
  if constexpr (sizeof(T) == 2) {
    a = Mma<T>::mma(Ch<T>::ldA(t, ld, row0, 0, lane), b.m, a);
    a = Mma<T>::mma(ldA16<T>(t, ld, row0, 32, lane), b.t, a);
  } else {
#pragma unroll
    for (int s = 0; s < 3; ++s) a = Mma<T>::mma(Ch<T>::ldA(t, ld, row0, 16 * s, lane), b.s[s], a);
  }
  return a;
}
// the same with A from an image [k = head column][ldt] (m contiguous)
template <typename T>
__device__ __forceinline__ f32x4 k48_tr(const T* t, int ldt, int row0, const HeadOp<T>& b, int lane, f32x4 a) {
  if constexpr (sizeof(T) == 2) {
    a = Mma<T>::mma(Ch<T>::ldA_tr(t, ldt, row0, 0, lane), b.m, a);
    a = Mma<T>::mma(ldA16_tr<T>(t, ldt, row0, 32, lane), b.t, a);
  } else {
#pragma unroll
    for (int s = 0; s < 3; ++s) a = Mma<T>::mma(Ch<T>::ldA_tr(t, ldt, row0, 16 * s, lane), b.s[s], a);
  }
  return a;
}
}  // namespace chain
