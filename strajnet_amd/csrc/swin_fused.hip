// Fused Swin-block kernels: the MLP half of SwinTransformerBlock as ONE kernel per direction.
//
//   forward   y = x + dp * ( gelu( LN(x) W1 + b1 ) W2 + b2 )                 reference modules.py:260 (second half of
//             SwinTransformerBlock.call), :40-46 (Mlp.call), :18-29 (Gelu), :137-151 (drop_path; dp = per-sample factor)
//   backward  dx = dy + LN'( (dp dy W2^T (.) gelu'(pre)) W1^T ),  LN gamma/beta gradients, and the operands of the two weight
//             gradients written once: h = gelu(pre), dpre, ln = LN(x), dys = dp dy   (dW1 = ln^T dpre, dW2 = h^T dys are
//             split-K GEMMs over the 32768 / 8192 rows: they need ~1000 rows per accumulating block, a fused block has 128)
//
// Layout of the work: a wave OWNS 16*RF token rows for the whole chain; nothing but the weights goes through LDS.
//   * x (and dy) are loaded straight from global memory as MFMA B fragments (n = row, 8 consecutive k per lane); LayerNorm
//     runs in those registers (a row lives in the 4 lanes (row, g = 0..3): two xor-shuffles per reduction).
//   * every product is computed TRANSPOSED, D[m = output column][n = row], with the weight tile as the A operand read from
//     LDS.  A lane then holds 4 consecutive output columns of its row -- and the accumulator fragments of one product ARE the
//     B operand of the next one (same n = row, k = the 4g+r columns the lane holds): GEMM -> GELU -> GEMM chains in registers
//     with no LDS round trip of the 4C-wide hidden tile.  For the 16-bit types one 32-deep MFMA k-step is fed by TWO
//     accumulator fragments, so lane (g, e) carries k = 4g+e (e < 4) or 16+4g+(e-4); the weight fragments are read with the
//     same permutation (two 8-byte LDS reads, or two ds_read_b64_tr_b16 for a [k][rows] image) -- a contraction does not care
//     about the order it is visited in.  (f32: one 16-deep k-step per accumulator fragment, natural order.)
//   * the hidden dimension is walked in chunks of HC columns: W1[:, chunk] and W2[chunk, :] are staged in LDS in their
//     natural Keras [in, out] layouts with coalesced 16-byte loads, shared by the 4 waves of the block.
#include "common.h"
#include "rng.h"

// ---- chained-operand fragments -------------------------------------------------------------------------------
template <typename T> struct Chain;
template <> struct Chain<float> {
  static constexpr int ND = 1;                       // accumulator fragments per MFMA k-step
  typedef f32x4 Frag;
  __device__ static __forceinline__ Frag from_acc(const f32x4* d) { return d[0]; }
  // A operand, image [rows][ld] (k contiguous)
  __device__ static __forceinline__ Frag ldA(const float* t, int ld, int row0, int k0, int lane) { return Mma<float>::load(t, ld, row0, k0, lane); }
  // A operand, image [k][ldt] (rows contiguous)
  __device__ static __forceinline__ Frag ldA_tr(const float* t, int ldt, int row0, int k0, int lane) { return Mma<float>::load_tr(t, ldt, row0, k0, lane); }
};
template <typename T> struct Chain16 {
  static constexpr int ND = 2;
  typedef s16x8 Frag;
  __device__ static __forceinline__ Frag from_acc(const f32x4* d) {
    typedef __attribute__((ext_vector_type(4))) uint32_t u4;
    const u4 w = {pack2<T>(d[0][0], d[0][1]), pack2<T>(d[0][2], d[0][3]), pack2<T>(d[1][0], d[1][1]), pack2<T>(d[1][2], d[1][3])};
    return __builtin_bit_cast(s16x8, w);
  }
  __device__ static __forceinline__ Frag ldA(const T* t, int ld, int row0, int k0, int lane) {
    const T* p = t + (row0 + (lane & 15)) * ld + k0 + (lane >> 4) * 4;
    const uint2 lo = *reinterpret_cast<const uint2*>(p), hi = *reinterpret_cast<const uint2*>(p + 16);
    typedef __attribute__((ext_vector_type(4))) uint32_t u4;
    const u4 w = {lo.x, lo.y, hi.x, hi.y};
    return __builtin_bit_cast(s16x8, w);
  }
  __device__ static __forceinline__ Frag ldA_tr(const T* t, int ldt, int row0, int k0, int lane) {
    typedef __attribute__((ext_vector_type(4))) short s4;
    const int g = lane >> 4, p = lane & 15;
    const T* a = t + (k0 + 4 * g + (p >> 2)) * ldt + row0 + 4 * (p & 3);
    const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(a));
    const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(a + 16 * ldt));
    return (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  }
};
template <> struct Chain<bf16> : Chain16<bf16> {};
template <> struct Chain<f16> : Chain16<f16> {};

// ---- GELU (tanh form, reference modules.py:18-29).  f32 parity mode: the exact tanhf form.  16-bit storage: the same function
// written as x * sigmoid(2u), u = sqrt(2/pi)(x + 0.044715 x^3), on v_exp_f32 / v_rcp_f32 (abs error ~1e-6, far below the storage
// rounding): a tanhf per hidden element would cost more VALU time than the MFMAs of the whole kernel.
template <typename T> __device__ __forceinline__ void gelu_both(float x, float& h, float& dh) {
  if constexpr (sizeof(T) == 4) { h = gelu_f(x); dh = gelu_grad_f(x); }
  else {
    const float k2 = 2.f * 0.7978845608028654f, a = 0.044715f;
    const float x2 = x * x;
    const float v = k2 * (x + a * x * x2);
    const float sg = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
    h = x * sg;
    dh = sg + h * (1.f - sg) * k2 * (1.f + 3.f * a * x2);
  }
}
template <typename T> __device__ __forceinline__ float gelu_fwd(float x) {
  if constexpr (sizeof(T) == 4) return gelu_f(x);
  else {
    const float k2 = 2.f * 0.7978845608028654f, a = 0.044715f;
    const float v = k2 * (x + a * x * x * x);
    return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
  }
}

// ---- geometry ---------------------------------------------------------------------------------------------------
template <typename T, int C> struct MlpCfg {
  static constexpr int KSTEP = Mma<T>::KSTEP;
  static constexpr int KS = C / KSTEP;                // k-steps over the model dimension
  static constexpr int NF = C / 16;                   // 16-column fragments of the model dimension
  static constexpr int RF = (C <= 192) ? 2 : 1;       // 16-row fragments per wave
  static constexpr int ROWS = 4 * RF * 16;            // rows per block
  // hidden columns per staged chunk: as many as keep the two images under ~78 KB (two blocks per CU)
  static constexpr int HC = sizeof(T) == 2 ? (C == 96 ? 192 : (C == 192 ? 96 : 32)) : (C == 96 ? 96 : (C == 192 ? 48 : 16));
  static constexpr int P1 = 4;                        // W1 image [C][HC + P1]: rows 8-byte aligned (tr reads, 8-byte chain reads)
  static constexpr int P2 = sizeof(T) == 2 ? 8 : 4;   // W2 image [HC][C + P2]: rows 16-byte aligned (16-byte fragment reads in backward)
  static constexpr int LD1 = HC + P1, LD2 = C + P2;
  static constexpr int LDS_BYTES = (C * LD1 + HC * LD2) * (int)sizeof(T);
  static_assert(C % KSTEP == 0 && HC % KSTEP == 0 && (4 * C) % HC == 0, "shape");
};

struct MlpArgs {
  const void* x; const float* gamma; const float* beta; const void* w1; const float* b1; const void* w2; const float* b2;
  void* y;
  const void* dy; void* dx; void* h; void* dpre; void* ln; void* dys; float* dgamma; float* dbeta; int nparts; long long pstride;
  long long M; float eps;
  const long long* rng; int site; float p_drop; long long rows_per_sample;
};

// stage W1[:, hc0 : hc0+HC] ([C][4C] global, row stride 4C) and W2[hc0 : hc0+HC, :] ([4C][C] global) into LDS
template <typename T, int C>
__device__ __forceinline__ void mlp_stage(T* W1s, T* W2s, const T* w1, const T* w2, int hc0, int tid) {
  typedef MlpCfg<T, C> G;
  constexpr int VN = Vec<T>::N;
  constexpr int CP1 = G::HC / VN;
  for (int q = tid; q < C * CP1; q += 256) {
    const int k = q / CP1, c = (q % CP1) * VN;
    const uint4 v = *reinterpret_cast<const uint4*>(w1 + (long long)k * (4 * C) + hc0 + c);
    T* d = W1s + k * G::LD1 + c;
    if constexpr (sizeof(T) == 2) { uint2* d2 = reinterpret_cast<uint2*>(d); d2[0] = make_uint2(v.x, v.y); d2[1] = make_uint2(v.z, v.w); }
    else *reinterpret_cast<uint4*>(d) = v;
  }
  constexpr int CP2 = C / VN;
  for (int q = tid; q < G::HC * CP2; q += 256) {
    const int k = q / CP2, c = (q % CP2) * VN;
    *reinterpret_cast<uint4*>(W2s + k * G::LD2 + c) = *reinterpret_cast<const uint4*>(w2 + (long long)(hc0 + k) * C + c);
  }
}

// rows of this wave as B fragments + LayerNorm statistics.  xa[i][ks]: row (m0 + 16 i + ln), k = ks*KSTEP + LANE_K*g ..
template <typename T, int C, int RF>
__device__ __forceinline__ void load_rows(typename Mma<T>::Frag (&xa)[RF][C / Mma<T>::KSTEP], const T* x, long long m0, long long M, int lane) {
  constexpr int KS = C / Mma<T>::KSTEP, LK = Mma<T>::LANE_K;
  const int g = lane >> 4, ln = lane & 15;
#pragma unroll
  for (int i = 0; i < RF; ++i) {
    const long long row = m0 + 16 * i + ln;
    const T* p = x + row * C + LK * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (row < M) xa[i][ks] = Mma<T>::from_global(p + ks * Mma<T>::KSTEP);
      else {
#pragma unroll
        for (int e = 0; e < LK; ++e) xa[i][ks][e] = 0;
      }
    }
  }
}
template <typename T> __device__ __forceinline__ void frag_unpack(const typename Mma<T>::Frag& f, float* v) {
  if constexpr (sizeof(T) == 4) { v[0] = f[0]; v[1] = f[1]; v[2] = f[2]; v[3] = f[3]; }
  else {
    typedef __attribute__((ext_vector_type(4))) uint32_t u4;
    const u4 w = __builtin_bit_cast(u4, f);
#pragma unroll
    for (int e = 0; e < 4; ++e) unpack2<T>(w[e], v[2 * e], v[2 * e + 1]);
  }
}
template <typename T> __device__ __forceinline__ typename Mma<T>::Frag frag_pack(const float* v) {
  if constexpr (sizeof(T) == 4) return (f32x4){v[0], v[1], v[2], v[3]};
  else {
    typedef __attribute__((ext_vector_type(4))) uint32_t u4;
    const u4 w = {pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]), pack2<T>(v[4], v[5]), pack2<T>(v[6], v[7])};
    return __builtin_bit_cast(s16x8, w);
  }
}
// in-register LayerNorm of the rows held as B fragments (biased variance, eps inside the sqrt: Keras).  Returns mean / rstd of the
// lane's row in mu[i], rs[i] (identical in the 4 lanes of a row).
template <typename T, int C, int RF>
__device__ __forceinline__ void ln_rows(typename Mma<T>::Frag (&xa)[RF][C / Mma<T>::KSTEP], const float* gamma, const float* beta, float eps,
                                        float (&mu)[RF], float (&rs)[RF], int lane) {
  constexpr int KS = C / Mma<T>::KSTEP, LK = Mma<T>::LANE_K;
  const int g = lane >> 4;
#pragma unroll
  for (int i = 0; i < RF; ++i) {
    float s = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      float v[LK];
      frag_unpack<T>(xa[i][ks], v);
#pragma unroll
      for (int e = 0; e < LK; ++e) s += v[e];
    }
    s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
    const float m = s * (1.f / C);
    float q = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      float v[LK];
      frag_unpack<T>(xa[i][ks], v);
#pragma unroll
      for (int e = 0; e < LK; ++e) { const float d = v[e] - m; q += d * d; }
    }
    q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
    const float r = rsqrtf(q * (1.f / C) + eps);
    mu[i] = m; rs[i] = r;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      float v[LK];
      frag_unpack<T>(xa[i][ks], v);
      const int c0 = ks * Mma<T>::KSTEP + LK * g;
#pragma unroll
      for (int e = 0; e < LK; ++e) v[e] = (v[e] - m) * r * gamma[c0 + e] + beta[c0 + e];
      xa[i][ks] = frag_pack<T>(v);
    }
  }
}

// =====================================================================================================================
// forward
// =====================================================================================================================
template <typename T, int C>
__global__ __launch_bounds__(256, (sizeof(T) == 2 && C <= 192) ? 2 : 1) void swin_mlp_fwd_kernel(MlpArgs p) {
  typedef MlpCfg<T, C> G;
  constexpr int KS = G::KS, NF = G::NF, RF = G::RF, ND = Chain<T>::ND, KSTEP = G::KSTEP;
  extern __shared__ __attribute__((aligned(16))) unsigned char mlp_smem[];
  T* W1s = reinterpret_cast<T*>(mlp_smem);
  T* W2s = W1s + C * G::LD1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, ln = lane & 15;
  const long long m0 = (long long)blockIdx.x * G::ROWS + wave * (RF * 16);
  const T* x = reinterpret_cast<const T*>(p.x);
  const T* w1 = reinterpret_cast<const T*>(p.w1);
  const T* w2 = reinterpret_cast<const T*>(p.w2);

  typename Mma<T>::Frag xa[RF][KS];
  load_rows<T, C, RF>(xa, x, m0, p.M, lane);
  float mu[RF], rs[RF];
  ln_rows<T, C, RF>(xa, p.gamma, p.beta, p.eps, mu, rs, lane);

  f32x4 acc2[RF][NF];
#pragma unroll
  for (int i = 0; i < RF; ++i)
#pragma unroll
    for (int f = 0; f < NF; ++f) acc2[i][f] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int hc0 = 0; hc0 < 4 * C; hc0 += G::HC) {
    __syncthreads();                                  // previous chunk's fragment reads are done
    mlp_stage<T, C>(W1s, W2s, w1, w2, hc0, tid);
    __syncthreads();
#pragma unroll 1
    for (int s = 0; s < G::HC / KSTEP; ++s) {
      f32x4 a1[RF][ND];
#pragma unroll
      for (int d = 0; d < ND; ++d) {
        const float4 bv = *reinterpret_cast<const float4*>(p.b1 + hc0 + s * KSTEP + 16 * d + 4 * g);
#pragma unroll
        for (int i = 0; i < RF; ++i) a1[i][d] = (f32x4){bv.x, bv.y, bv.z, bv.w};
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int d = 0; d < ND; ++d) {
          const typename Mma<T>::Frag wf = Mma<T>::load_tr(W1s, G::LD1, s * KSTEP + 16 * d, ks * KSTEP, lane);
#pragma unroll
          for (int i = 0; i < RF; ++i) a1[i][d] = Mma<T>::mma(wf, xa[i][ks], a1[i][d]);
        }
      typename Chain<T>::Frag hf[RF];
#pragma unroll
      for (int i = 0; i < RF; ++i) {
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
          for (int r = 0; r < 4; ++r) a1[i][d][r] = gelu_fwd<T>(a1[i][d][r]);
        hf[i] = Chain<T>::from_acc(a1[i]);
      }
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const typename Chain<T>::Frag wf = Chain<T>::ldA_tr(W2s, G::LD2, 16 * f, s * KSTEP, lane);
#pragma unroll
        for (int i = 0; i < RF; ++i) acc2[i][f] = Mma<T>::mma(wf, hf[i], acc2[i][f]);
      }
    }
  }

  // epilogue: y = x + dp * (acc + b2); the lane holds columns 16 f + 4 g .. +3 of row (m0 + 16 i + ln)
  T* y = reinterpret_cast<T*>(p.y);
#pragma unroll
  for (int i = 0; i < RF; ++i) {
    const long long row = m0 + 16 * i + ln;
    if (row >= p.M) continue;
    const float dp = drop_path_scale(p.rng, p.site, row / p.rows_per_sample, p.p_drop);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int col = 16 * f + 4 * g;
      const float4 bv = *reinterpret_cast<const float4*>(p.b2 + col);
      float xv[4];
      ld4(x + row * C + col, xv);
      const float v[4] = {xv[0] + dp * (acc2[i][f][0] + bv.x), xv[1] + dp * (acc2[i][f][1] + bv.y),
                          xv[2] + dp * (acc2[i][f][2] + bv.z), xv[3] + dp * (acc2[i][f][3] + bv.w)};
      st4(y + row * C + col, v);
    }
  }
}

// =====================================================================================================================
// backward
// =====================================================================================================================
template <typename T, int C>
__global__ __launch_bounds__(256, (sizeof(T) == 2 && C <= 96) ? 2 : 1) void swin_mlp_bwd_kernel(MlpArgs p) {
  typedef MlpCfg<T, C> G;
  constexpr int KS = G::KS, NF = G::NF, RF = G::RF, ND = Chain<T>::ND, KSTEP = G::KSTEP, LK = Mma<T>::LANE_K;
  extern __shared__ __attribute__((aligned(16))) unsigned char mlp_smem[];
  __shared__ float red[2][C];
  T* W1s = reinterpret_cast<T*>(mlp_smem);
  T* W2s = W1s + C * G::LD1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, ln = lane & 15;
  const long long m0 = (long long)blockIdx.x * G::ROWS + wave * (RF * 16);
  const T* x = reinterpret_cast<const T*>(p.x);
  const T* dy = reinterpret_cast<const T*>(p.dy);
  const T* w1 = reinterpret_cast<const T*>(p.w1);
  const T* w2 = reinterpret_cast<const T*>(p.w2);
  for (int c = tid; c < 2 * C; c += 256) (&red[0][0])[c] = 0.f;

  typename Mma<T>::Frag xa[RF][KS], da[RF][KS];
  load_rows<T, C, RF>(xa, x, m0, p.M, lane);
  load_rows<T, C, RF>(da, dy, m0, p.M, lane);
  float mu[RF], rs[RF], dp[RF];
  ln_rows<T, C, RF>(xa, p.gamma, p.beta, p.eps, mu, rs, lane);
  // hand-off operands of the weight gradients: ln = LN(x) and dys = dp * dy, written from the B fragments (16-byte rows segments)
  T* lnq = reinterpret_cast<T*>(p.ln);
  T* dysq = reinterpret_cast<T*>(p.dys);
#pragma unroll
  for (int i = 0; i < RF; ++i) {
    const long long row = m0 + 16 * i + ln;
    dp[i] = row < p.M ? drop_path_scale(p.rng, p.site, row / p.rows_per_sample, p.p_drop) : 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (dp[i] != 1.f) {                              // scale dy once; every later use (dh, dys) wants dp * dy
        float v[LK];
        frag_unpack<T>(da[i][ks], v);
#pragma unroll
        for (int e = 0; e < LK; ++e) v[e] *= dp[i];
        da[i][ks] = frag_pack<T>(v);
      }
      if (row < p.M) {
        const long long o = row * C + ks * KSTEP + LK * g;
        *reinterpret_cast<typename Mma<T>::Frag*>(lnq + o) = xa[i][ks];
        if (dysq) *reinterpret_cast<typename Mma<T>::Frag*>(dysq + o) = da[i][ks];
      }
    }
  }

  f32x4 acc[RF][NF];                                   // d LN(x)^T [c][row]
#pragma unroll
  for (int i = 0; i < RF; ++i)
#pragma unroll
    for (int f = 0; f < NF; ++f) acc[i][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
  T* hq = reinterpret_cast<T*>(p.h);
  T* dpq = reinterpret_cast<T*>(p.dpre);

  for (int hc0 = 0; hc0 < 4 * C; hc0 += G::HC) {
    __syncthreads();
    mlp_stage<T, C>(W1s, W2s, w1, w2, hc0, tid);
    __syncthreads();
#pragma unroll 1
    for (int s = 0; s < G::HC / KSTEP; ++s) {
      f32x4 a1[RF][ND], a3[RF][ND];                    // pre^T and dh^T, [hidden][row]
#pragma unroll
      for (int d = 0; d < ND; ++d) {
        const float4 bv = *reinterpret_cast<const float4*>(p.b1 + hc0 + s * KSTEP + 16 * d + 4 * g);
#pragma unroll
        for (int i = 0; i < RF; ++i) { a1[i][d] = (f32x4){bv.x, bv.y, bv.z, bv.w}; a3[i][d] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int d = 0; d < ND; ++d) {
          const typename Mma<T>::Frag wf = Mma<T>::load_tr(W1s, G::LD1, s * KSTEP + 16 * d, ks * KSTEP, lane);      // W1^T: m = hidden, k = c
          const typename Mma<T>::Frag vf = Mma<T>::load(W2s, G::LD2, s * KSTEP + 16 * d, ks * KSTEP, lane);         // W2 : m = hidden, k = out col
#pragma unroll
          for (int i = 0; i < RF; ++i) {
            a1[i][d] = Mma<T>::mma(wf, xa[i][ks], a1[i][d]);
            a3[i][d] = Mma<T>::mma(vf, da[i][ks], a3[i][d]);
          }
        }
      typename Chain<T>::Frag df[RF];
#pragma unroll
      for (int i = 0; i < RF; ++i) {
        const long long row = m0 + 16 * i + ln;
#pragma unroll
        for (int d = 0; d < ND; ++d) {
          float hv[4], gv[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float gd;
            gelu_both<T>(a1[i][d][r], hv[r], gd);
            gv[r] = a3[i][d][r] * gd;
            a3[i][d][r] = gv[r];
          }
          if (row < p.M) {
            const long long o = row * (4 * C) + hc0 + s * KSTEP + 16 * d + 4 * g;
            st4(hq + o, hv);
            st4(dpq + o, gv);
          }
        }
        df[i] = Chain<T>::from_acc(a3[i]);
      }
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const typename Chain<T>::Frag wf = Chain<T>::ldA(W1s, G::LD1, 16 * f, s * KSTEP, lane);                  // W1: m = c, k = hidden
#pragma unroll
        for (int i = 0; i < RF; ++i) acc[i][f] = Mma<T>::mma(wf, df[i], acc[i][f]);
      }
    }
  }

  // LayerNorm backward on the accumulator layout (lane: columns 16 f + 4 g .. +3 of row m0 + 16 i + ln) + the skip gradient
  T* dx = reinterpret_cast<T*>(p.dx);
  float dgs[NF][4], dbs[NF][4];
#pragma unroll
  for (int f = 0; f < NF; ++f)
#pragma unroll
    for (int r = 0; r < 4; ++r) { dgs[f][r] = 0.f; dbs[f][r] = 0.f; }
#pragma unroll
  for (int i = 0; i < RF; ++i) {
    const long long row = m0 + 16 * i + ln;
    const bool live = row < p.M;
    float xh[NF][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int col = 16 * f + 4 * g;
      float xv[4] = {0.f, 0.f, 0.f, 0.f};
      if (live) ld4(x + row * C + col, xv);
      const float4 gm = *reinterpret_cast<const float4*>(p.gamma + col);
      const float gmv[4] = {gm.x, gm.y, gm.z, gm.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        xh[f][r] = (xv[r] - mu[i]) * rs[i];
        const float d = live ? acc[i][f][r] : 0.f;
        dgs[f][r] += d * xh[f][r];
        dbs[f][r] += d;
        const float a = d * gmv[r];
        acc[i][f][r] = a;
        s1 += a; s2 += a * xh[f][r];
      }
    }
    s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
    s1 *= (1.f / C); s2 *= (1.f / C);
    if (live) {
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const int col = 16 * f + 4 * g;
        float dv[4];
        ld4(dy + row * C + col, dv);
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = dv[r] + rs[i] * (acc[i][f][r] - s1 - xh[f][r] * s2);
        st4(dx + row * C + col, v);
      }
    }
  }
  // gamma / beta gradients: lanes of equal g hold the same columns -> reduce over the 16 rows (ln), then over the waves in LDS,
  // then ONE global atomic per column and block into copy (block % nparts)
#pragma unroll
  for (int f = 0; f < NF; ++f)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float a = dgs[f][r], b = dbs[f][r];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
      if (ln == 0) { atomicAdd(&red[0][16 * f + 4 * g + r], a); atomicAdd(&red[1][16 * f + 4 * g + r], b); }
    }
  __syncthreads();
  const long long po = (long long)(blockIdx.x % p.nparts) * p.pstride;
  for (int c = tid; c < C; c += 256) { atomicAdd(p.dgamma + po + c, red[0][c]); atomicAdd(p.dbeta + po + c, red[1][c]); }
}

template <typename T, int C>
static int mlp_launch(bool bwd, const MlpArgs& a, hipStream_t st) {
  typedef MlpCfg<T, C> G;
  const void* fn = bwd ? (const void*)swin_mlp_bwd_kernel<T, C> : (const void*)swin_mlp_fwd_kernel<T, C>;
  static bool attr[2] = {false, false};
  if (!attr[bwd]) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES) != hipSuccess) {
      stj_set_error("swin_mlp: cannot reserve %d bytes of LDS", G::LDS_BYTES); return STJ_ELAUNCH;
    }
    attr[bwd] = true;
  }
  dim3 grid((unsigned)((a.M + G::ROWS - 1) / G::ROWS));
  if (bwd) hipLaunchKernelGGL((swin_mlp_bwd_kernel<T, C>), grid, dim3(256), G::LDS_BYTES, st, a);
  else hipLaunchKernelGGL((swin_mlp_fwd_kernel<T, C>), grid, dim3(256), G::LDS_BYTES, st, a);
  return stj_check_launch(bwd ? "stj_swin_mlp_bwd" : "stj_swin_mlp_fwd");
}
template <typename T>
static int mlp_dispatch(bool bwd, int C, const MlpArgs& a, hipStream_t st) {
  switch (C) {
    case 96: return mlp_launch<T, 96>(bwd, a, st);
    case 192: return mlp_launch<T, 192>(bwd, a, st);
    case 384: return mlp_launch<T, 384>(bwd, a, st);
    default: stj_set_error("swin_mlp: C must be 96, 192 or 384 (got %d)", C); return STJ_EUNSUPPORTED;
  }
}
static int mlp_any(bool bwd, int C, int dtype, const MlpArgs& a, hipStream_t st) {
  if (a.M <= 0) return STJ_OK;
  if (a.rows_per_sample < 16 || a.rows_per_sample % 16) { stj_set_error("swin_mlp: rows_per_sample must be a positive multiple of 16"); return STJ_EINVAL; }
  if (!(a.p_drop >= 0.f && a.p_drop < 1.f)) { stj_set_error("swin_mlp: need 0 <= p_drop < 1"); return STJ_EINVAL; }
  if (dtype == STJ_BF16) return mlp_dispatch<bf16>(bwd, C, a, st);
  if (dtype == STJ_F16) return mlp_dispatch<f16>(bwd, C, a, st);
  if (dtype == STJ_F32) return mlp_dispatch<float>(bwd, C, a, st);
  stj_set_error("swin_mlp: bad dtype %d", dtype);
  return STJ_EINVAL;
}

extern "C" int stj_swin_mlp_fwd(const void* x, const float* gamma, const float* beta, const void* w1, const float* b1, const void* w2,
                                const float* b2, void* y, long long M, int C, float eps, const long long* rng_state, int site,
                                float p_drop, long long rows_per_sample, int dtype, hipStream_t stream) {
  MlpArgs a = {};
  a.x = x; a.gamma = gamma; a.beta = beta; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.y = y; a.M = M; a.eps = eps;
  a.rng = rng_state; a.site = site; a.p_drop = p_drop; a.rows_per_sample = rows_per_sample;
  return mlp_any(false, C, dtype, a, stream);
}

extern "C" int stj_swin_mlp_bwd(const void* x, const void* dy, const float* gamma, const float* beta, const void* w1, const float* b1,
                                const void* w2, void* dx, void* h, void* dpre, void* ln, void* dys, float* dgamma, float* dbeta,
                                int nparts, long long part_stride, long long M, int C, float eps, const long long* rng_state, int site,
                                float p_drop, long long rows_per_sample, int dtype, hipStream_t stream) {
  if (nparts < 1) { stj_set_error("swin_mlp_bwd: nparts must be >= 1"); return STJ_EINVAL; }
  MlpArgs a = {};
  a.x = x; a.dy = dy; a.gamma = gamma; a.beta = beta; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.dx = dx; a.h = h; a.dpre = dpre; a.ln = ln;
  a.dys = dys; a.dgamma = dgamma; a.dbeta = dbeta; a.nparts = nparts; a.pstride = part_stride; a.M = M; a.eps = eps;
  a.rng = rng_state; a.site = site; a.p_drop = p_drop; a.rows_per_sample = rows_per_sample;
  return mlp_any(true, C, dtype, a, stream);
}
