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
#include <algorithm>
#include "rng.h"
#include <stdlib.h>

// tuning switches (A/B builds: -DSTJ_MLP_MINB=2 ...)
#ifndef STJ_MLP_MINB
#define STJ_MLP_MINB 1          // min resident blocks per CU the MLP kernels are compiled for (2 caps them at 256 registers)
#endif
#ifndef STJ_MLP_PREFETCH
#define STJ_MLP_PREFETCH 1      // next weight chunk's global loads issued before the current chunk's MFMAs
#endif
#ifndef STJ_MLP_HC96W
#define STJ_MLP_HC96W 192      // C = 96 MLP kernels at 32768 rows: hidden columns per weight chunk (96: -0.6 %; 384 = everything resident, 154 KB of LDS: -0.3 %, profiles/r06_zo_*, r06_zt_*)
#endif
#ifndef STJ_ATTN_MINB96
#define STJ_ATTN_MINB96 3       // swin_attn_fwd at C = 96 (16-bit) compiled for three waves per SIMD (158 registers, no spill; 40 KB of LDS per window: three windows per CU
#endif                          // instead of two): inference +0.4 %, cfg-512 +0.2 %, train step equal; four (128 registers, 24 spilled) loses 0.7 % (profiles/r06_zr_attn_occ3.txt)
#ifndef STJ_ATTN_HG192
#define STJ_ATTN_HG192 2
#endif
#ifndef STJ_MLP_HC96
#define STJ_MLP_HC96 96         // (96: 1076 vs 1070 scenes/s for 192) hidden columns per staged weight chunk of the MLP kernels (16-bit types), C = 96 / C = 192
#endif
#ifndef STJ_MLP_HC192
#define STJ_MLP_HC192 96
#endif
#ifndef STJ_MLP_HC192S
#define STJ_MLP_HC192S 64       // C = 192 in the eight-wave workgroups (256 registers per wave: 128 columns spill in backward)
#endif
#ifndef STJ_MLP_HC384
#define STJ_MLP_HC384 32        // C = 384 (the split kernels: 192 hidden columns per workgroup)
#endif
#ifndef STJ_ATTNB_HG96
#define STJ_ATTNB_HG96 1       // heads per pass of the attention backward kernel (16-bit types); 1: 1068 vs 1059 scenes/s for 3 (59 instead of 121 KB LDS per window)
#endif
#ifndef STJ_ATTNB_HG192
#define STJ_ATTNB_HG192 2
#endif
#ifndef STJ_ATTNB_MINB
#define STJ_ATTNB_MINB 1
#endif
#ifndef STJ_ATTN_MINB
#define STJ_ATTN_MINB 1
#endif
#ifndef STJ_ATTN_HG96
#define STJ_ATTN_HG96 1         // heads per pass of the attention kernel at C = 96 (16-bit types); 1: 1059 vs 1052 scenes/s for 3 (less LDS per window); at 131072 rows (round 6): inference 5813-5889 vs 5596-5665, cfg-512 773-774 vs 761 (profiles/r06_zn_attn_hg96.txt)
#endif

#ifdef STJ_STAMP   // A/B builds only (tools/build_variant.sh): per-workgroup cycle stamps of wave 0 at the kernels' phase boundaries
__device__ unsigned long long g_stamp[8 * 2048];
#define STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 2048) g_stamp[blockIdx.x * 8 + (i)] = (i) == 0 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int stj_dbg_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp), sizeof(g_stamp)); }
extern "C" int stj_dbg_clear() { void* p = nullptr; if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_stamp)) != hipSuccess) return 1; return (int)hipMemset(p, 0, sizeof(g_stamp)); }
// per-wave-0 accumulated cycles between TICK(k-1) and TICK(k) inside the chunk loop -> g_tick[block][k]
__device__ unsigned long long g_tick[8 * 2048];
#define TICK_DECL unsigned long long tk_prev = __builtin_amdgcn_s_memtime(), tk_acc[5] = {0, 0, 0, 0, 0}
#define TICK(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tk_acc[k] += t_ - tk_prev; tk_prev = t_; } while (0)
#define TICK_STORE do { if (threadIdx.x == 0 && blockIdx.x < 2048) for (int k_ = 0; k_ < 5; ++k_) g_tick[blockIdx.x * 8 + k_] = tk_acc[k_]; } while (0)
extern "C" int stj_dbg_ticks(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tick), sizeof(g_tick)); }
#else
#define STAMP(i) do {} while (0)
#define TICK_DECL
#define TICK(k) do {} while (0)
#define TICK_STORE do {} while (0)
#endif

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


// ---- SPLIT == 2: the slices of one unit (row block / window) hand their partial sums over INSIDE the launch ------------------------
// Every slice workgroup writes its accumulators as one slab (write-through 16-byte stores, fragment order: each store instruction of
// the workgroup is one contiguous 4 KB piece), drains them, and draws a ticket from the unit's counter; the workgroup that draws the
// last ticket adds the other slabs to its own accumulators (sc1 loads: the producers stored sc1, MI355X_MICROARCH.md "Workgroup
// dispatch, XCD placement & inter-workgroup visibility") and runs the epilogue.  No dispatch-order, residency or placement assumption:
// nobody waits for anybody.  With two slices the sum is own + other in either arrival order: bitwise reproducible.
// ws = [slabs [unit][slice][NFR][NTA] f32x4 (the same region holds the [slice][M][C] partial sums of the SPLIT == 1 launches)]
//      [FIX_CNT_BYTES of int counters, zeroed ONCE by the caller (the last arriver re-arms its counter)]
constexpr int FIX_CNT_BYTES = 16384;                 // up to 4096 units per launch
// Threads tid < NTA hold accumulators (the others only take part in the barriers).
template <int NFR, int NTA = 256>
__device__ __forceinline__ bool slice_combine(f32x4* acc, void* slabs, int* counters, int unit, int sp, int S, int tid, int* s_ticket) {
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  int* cnt = counters + unit;
  constexpr unsigned SLAB = (unsigned)NTA * NFR * 16u;        // bytes of one slab
  const bool active = tid < NTA;
  char* base = reinterpret_cast<char*>(slabs) + (long long)unit * S * SLAB;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, (unsigned)S * SLAB, 0x00020000);
  if (active) {
#pragma unroll
    for (int j = 0; j < NFR; ++j)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[j]), rs, (unsigned)sp * SLAB + (unsigned)(j * NTA + tid) * 16u, 0, 16 /* sc1 */);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) *s_ticket = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (*s_ticket != S - 1) return false;
  if (tid == 0) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int s = 0; s < S; ++s) {
    if (s == sp || !active) continue;
#pragma unroll
    for (int j = 0; j < NFR; ++j) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)s * SLAB + (unsigned)(j * NTA + tid) * 16u, 0, 16 /* sc1 */);
      acc[j] += __builtin_bit_cast(f32x4, v);
    }
  }
  return true;
}

// ---- geometry ---------------------------------------------------------------------------------------------------
// NW waves per workgroup in HS hidden groups of RG = NW / HS waves: wave (rg, hg) owns 16 RF rows and, of every staged chunk, the k-steps
// s = hg, hg + HS, ...; the hidden groups' sums meet in LDS behind the chunk loop.  Eight waves = two per SIMD: one wave's GELU / LayerNorm
// VALU work and LDS-read latencies run under the other's MFMAs (with four waves the chunk loop ran 75 cycles per MFMA: stamps).
template <typename T, int C, int RF_ = ((C <= 192) ? 2 : 1), int NW_ = 4, int HS_ = 1, int HCX = 0> struct MlpCfg {
  static constexpr int KSTEP = Mma<T>::KSTEP;
  static constexpr int KS = C / KSTEP;                // k-steps over the model dimension
  static constexpr int NF = C / 16;                   // 16-column fragments of the model dimension
  static constexpr int RF = RF_;                      // 16-row fragments per wave
  static constexpr int NW = NW_, HS = HS_, RG = NW / HS, NT = 64 * NW;
  static constexpr int ROWS = RG * RF * 16;           // rows per block
  // hidden columns per staged chunk: as many as keep the two images under ~78 KB (two blocks per CU); with two hidden groups an even number of k-steps
  static constexpr int HC = HCX ? HCX : sizeof(T) == 2 ? (C == 96 ? STJ_MLP_HC96 : (C == 192 ? (NW == 8 ? STJ_MLP_HC192S : STJ_MLP_HC192) : (HS == 2 ? 64 : STJ_MLP_HC384)))
                                           : (C == 96 ? 96 : (C == 192 ? 48 : 16));
  static constexpr int P1 = 4;                        // W1 image [C][HC + P1]: rows 8-byte aligned (tr reads, 8-byte chain reads)
  static constexpr int P2 = sizeof(T) == 2 ? 8 : 4;   // W2 image [HC][C + P2]: rows 16-byte aligned (16-byte fragment reads in backward)
  static constexpr int LD1 = HC + P1, LD2 = C + P2;
  static constexpr int LDS_BYTES = (C * LD1 + HC * LD2) * (int)sizeof(T);
  static_assert(C % KSTEP == 0 && HC % KSTEP == 0 && (4 * C) % HC == 0 && (HC / KSTEP) % HS == 0 && NW % HS == 0, "shape");
  static_assert(HS == 1 || RG * RF * NF * 1024 <= LDS_BYTES, "the hidden groups' sums meet in the weight images' LDS");
};

struct MlpArgs {
  const void* x; const float* gamma; const float* beta; const void* w1; const float* b1; const void* w2; const float* b2;
  void* y;
  const void* dy; void* dx; void* h; void* dpre; void* ln; void* dys; float* dgamma; float* dbeta; int nparts; long long pstride;
  long long M; float eps;
  const long long* rng; int site; float p_drop; long long rows_per_sample;
  int split; float* part; int* cnt;   // SPLIT kernels: workgroup = (row block, slice of the hidden dimension); partial sums [split][M][C] f32 | slabs + arrival counters
};

// Staging of W1[:, hc0 : hc0+HC] ([C][4C] global, row stride 4C) and W2[hc0 : hc0+HC, :] ([4C][C] global) into LDS, split into
// "global -> registers" (issued one chunk AHEAD: the loads fly while the current chunk's MFMAs run -- with a synchronous copy every
// chunk exposed a full L2 / HBM round trip per 16 KB in flight, and the weights are cold in the real step: 114 vs 53 us for the
// 8192 x 192 forward) and "registers -> LDS".
template <typename T, int C, typename G> struct MlpStage {
  static constexpr int VN = Vec<T>::N, NT = G::NT;
  static constexpr int CP1 = G::HC / VN, CP2 = C / VN;
  static constexpr int N1 = (C * CP1 + NT - 1) / NT, N2 = (G::HC * CP2 + NT - 1) / NT;
  uint4 r1[N1], r2[N2];
  __device__ __forceinline__ void issue(const T* w1, const T* w2, int hc0, int tid) {
#pragma unroll
    for (int i = 0; i < N1; ++i) {
      const int q = tid + i * NT;
      uint4 v = make_uint4(0, 0, 0, 0);          // (always store a selected value: a conditional store keeps the array in scratch memory)
      if (q < C * CP1) { const int k = q / CP1, c = (q % CP1) * VN; v = *reinterpret_cast<const uint4*>(w1 + (long long)k * (4 * C) + hc0 + c); }
      r1[i] = v;
    }
#pragma unroll
    for (int i = 0; i < N2; ++i) {
      const int q = tid + i * NT;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (q < G::HC * CP2) { const int k = q / CP2, c = (q % CP2) * VN; v = *reinterpret_cast<const uint4*>(w2 + (long long)(hc0 + k) * C + c); }
      r2[i] = v;
    }
  }
  __device__ __forceinline__ void commit(T* W1s, T* W2s, int tid) const {
#pragma unroll
    for (int i = 0; i < N1; ++i) {
      const int q = tid + i * NT;
      if (q < C * CP1) {
        T* d = W1s + (q / CP1) * G::LD1 + (q % CP1) * VN;
        if constexpr (sizeof(T) == 2) { uint2* d2 = reinterpret_cast<uint2*>(d); d2[0] = make_uint2(r1[i].x, r1[i].y); d2[1] = make_uint2(r1[i].z, r1[i].w); }
        else *reinterpret_cast<uint4*>(d) = r1[i];
      }
    }
#pragma unroll
    for (int i = 0; i < N2; ++i) {
      const int q = tid + i * NT;
      if (q < G::HC * CP2) *reinterpret_cast<uint4*>(W2s + (q / CP2) * G::LD2 + (q % CP2) * VN) = r2[i];
    }
  }
};

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

// 4 consecutive elements of a row as loaded (unpacked where they are used: the loads are issued a phase early)
template <typename T> struct Raw4 { typedef uint2 type; };
template <> struct Raw4<float> { typedef float4 type; };
template <typename T> __device__ __forceinline__ void raw4_unpack(const typename Raw4<T>::type& r, float* v) {
  if constexpr (sizeof(T) == 4) { v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w; }
  else { unpack2<T>(r.x, v[0], v[1]); unpack2<T>(r.y, v[2], v[3]); }
}

// =====================================================================================================================
// forward
// =====================================================================================================================
template <typename T, int C, int RFP, int SPLIT = 0, int NW = 4, int HS = 1, int HCX = 0>
__global__ __launch_bounds__(64 * NW, STJ_MLP_MINB) void swin_mlp_fwd_kernel(MlpArgs p) {
  typedef MlpCfg<T, C, RFP, NW, HS, HCX> G;
  constexpr int KS = G::KS, NF = G::NF, RF = G::RF, ND = Chain<T>::ND, KSTEP = G::KSTEP, NT = G::NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char mlp_smem[];
  T* W1s = reinterpret_cast<T*>(mlp_smem);
  T* W2s = W1s + C * G::LD1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, ln = lane & 15;
  const int rg = wave % G::RG, hg = wave / G::RG;          // row group, hidden group of this wave
  // SPLIT (the 2048-row C = 384 stage: 32 row blocks cannot fill 256 CUs, and each would stream all 2.4 MB of weights): workgroup =
  // (row block, slice sp of the hidden dimension); consecutive workgroups take consecutive slices, so slice sp of the weights is
  // read by the workgroups of ONE XCD (blockIdx % 8 with 8 slices) and stays in that L2.  Where its 29 us go at 2048 rows (parts
  // switched off one at a time): 6 weight chunks 13 us (LDS-read bound: with one 16-row fragment per wave every MFMA needs its own A
  // fragment from LDS), the 25 MB of f32 partial sums 8 us, LayerNorm 2, launch + row loads + first chunk 6; layer by layer: 48 us
  const int unit = SPLIT ? (int)blockIdx.x / p.split : (int)blockIdx.x, sp = SPLIT ? (int)blockIdx.x % p.split : 0;
  const int hs0 = SPLIT ? sp * (4 * C / p.split) : 0, hs1 = SPLIT ? hs0 + 4 * C / p.split : 4 * C;
  const long long m0 = (long long)unit * G::ROWS + rg * (RF * 16);
  const T* x = reinterpret_cast<const T*>(p.x);
  const T* w1 = reinterpret_cast<const T*>(p.w1);
  const T* w2 = reinterpret_cast<const T*>(p.w2);

  // FFN1 bias in LDS: a global load inside the chunk loop sits BEHIND the prefetched weight chunk in the in-order vmcnt queue, so waiting
  // for it waited for the whole prefetch -- every chunk paid a full memory round trip (found on the C = 384 split kernels: 6 chunks, 32 us)
  __shared__ float b1s[4 * C];
  STAMP(0); STAMP(1);
  // (the bias loads go LAST: their LDS writes wait for everything issued before them, so bias, first weight chunk and rows are ONE memory
  //  round trip; in front of the others they were a round trip of their own.  Not in the C = 384 split kernels: 36.5 -> 39.5 us there)
  if constexpr (SPLIT == 1) { for (int c = hs0 + tid; c < hs1; c += NT) b1s[c] = p.b1[c]; }
  MlpStage<T, C, G> stg;
  stg.issue(w1, w2, hs0, tid);                        // first weight chunk in flight under the row loads + LayerNorm
  typename Mma<T>::Frag xa[RF][KS];
  load_rows<T, C, RF>(xa, x, m0, p.M, lane);
  if constexpr (SPLIT != 1) { for (int c = hs0 + tid; c < hs1; c += NT) b1s[c] = p.b1[c]; }
  float mu[RF], rs[RF];
  ln_rows<T, C, RF>(xa, p.gamma, p.beta, p.eps, mu, rs, lane);
  STAMP(2);

  f32x4 acc2[RF][NF];
#pragma unroll
  for (int i = 0; i < RF; ++i)
#pragma unroll
    for (int f = 0; f < NF; ++f) acc2[i][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // the shortcut rows in the accumulator layout: fetched under the last chunk / in front of the hand-over (see swin_attn_bwd_kernel)
  constexpr bool TAILPF = RF == 1 && SPLIT != 1;
  typename Raw4<T>::type xraw[TAILPF ? NF : 1];
  auto tail_prefetch = [&]() __attribute__((always_inline)) {
    const long long row = m0 + ln < p.M ? m0 + ln : p.M - 1;
#pragma unroll
    for (int f = 0; f < (TAILPF ? NF : 1); ++f) xraw[f] = *reinterpret_cast<const typename Raw4<T>::type*>(x + row * C + 16 * f + 4 * g);
  };

  TICK_DECL;
  for (int hc0 = hs0; hc0 < hs1; hc0 += G::HC) {
    TICK(0);
    __syncthreads();                                  // previous chunk's fragment reads are done
    TICK(1);
    if (!STJ_MLP_PREFETCH && hc0 > hs0) stg.issue(w1, w2, hc0, tid);
    stg.commit(W1s, W2s, tid);
    TICK(2);
    __syncthreads();
    TICK(3);
    if (hc0 == hs0) STAMP(6);
    if (STJ_MLP_PREFETCH && hc0 + G::HC < hs1) stg.issue(w1, w2, hc0 + G::HC, tid);      // next chunk: global loads overlap this chunk's MFMAs
    if constexpr (TAILPF && HS == 1 && SPLIT == 0) { if (hc0 + G::HC >= hs1) tail_prefetch(); }
#pragma unroll 1
    for (int s = hg; s < G::HC / KSTEP; s += HS) {
      f32x4 a1[RF][ND];
#pragma unroll
      for (int d = 0; d < ND; ++d) {
        const float4 bv = *reinterpret_cast<const float4*>(b1s + hc0 + s * KSTEP + 16 * d + 4 * g);
#pragma unroll
        for (int i = 0; i < RF; ++i) a1[i][d] = (f32x4){bv.x, bv.y, bv.z, bv.w};
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int d = 0; d < ND; ++d) {
#ifdef STJ_ABL_NOLDS1      // ablation builds (tools/probes/swin_ablate.sh): the fragment address does not depend on the step, the read is hoisted
          const typename Mma<T>::Frag wf = Mma<T>::load_tr(W1s, G::LD1, 16 * d, ks * KSTEP, lane);
#else
          const typename Mma<T>::Frag wf = Mma<T>::load_tr(W1s, G::LD1, s * KSTEP + 16 * d, ks * KSTEP, lane);
#endif
#pragma unroll
          for (int i = 0; i < RF; ++i) a1[i][d] = Mma<T>::mma(wf, xa[i][ks], a1[i][d]);
        }
      typename Chain<T>::Frag hf[RF];
#pragma unroll
      for (int i = 0; i < RF; ++i) {
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
#ifndef STJ_ABL_NOGELU
          for (int r = 0; r < 4; ++r) a1[i][d][r] = gelu_fwd<T>(a1[i][d][r]);
#else
          for (int r = 0; r < 4; ++r) a1[i][d][r] = a1[i][d][r] * 0.5f;
#endif
        hf[i] = Chain<T>::from_acc(a1[i]);
      }
#pragma unroll
      for (int f = 0; f < NF; ++f) {
#ifdef STJ_ABL_NOLDS2
        const typename Chain<T>::Frag wf = Chain<T>::ldA_tr(W2s, G::LD2, 16 * f, 0, lane);
#else
        const typename Chain<T>::Frag wf = Chain<T>::ldA_tr(W2s, G::LD2, 16 * f, s * KSTEP, lane);
#endif
#pragma unroll
        for (int i = 0; i < RF; ++i) acc2[i][f] = Mma<T>::mma(wf, hf[i], acc2[i][f]);
      }
    }
    TICK(4);
  }
  TICK_STORE;

  STAMP(3);
  if constexpr (TAILPF && (HS > 1 || SPLIT == 2)) tail_prefetch();
  if constexpr (HS > 1) {         // the hidden groups' sums meet in LDS (over the weight images): group 0 carries on with the total
    __syncthreads();
    f32x4* rb = reinterpret_cast<f32x4*>(mlp_smem) + (rg * RF * NF) * 64 + lane;
    if (hg == 1) {
#pragma unroll
      for (int i = 0; i < RF; ++i)
#pragma unroll
        for (int f = 0; f < NF; ++f) rb[(i * NF + f) * 64] = acc2[i][f];
    }
    __syncthreads();
    if (hg == 0) {
#pragma unroll
      for (int i = 0; i < RF; ++i)
#pragma unroll
        for (int f = 0; f < NF; ++f) acc2[i][f] += rb[(i * NF + f) * 64];
    }
  }
  // epilogue: y = x + dp * (acc + b2); the lane holds columns 16 f + 4 g .. +3 of row (m0 + 16 i + ln)
  if constexpr (SPLIT == 1) {
    if (hg != 0) return;     // this slice's share of the sum over the hidden dimension; swin_split_fwd_epi_kernel finishes the rows
#pragma unroll
    for (int i = 0; i < RF; ++i) {
      const long long row = m0 + 16 * i + ln;
      if (row >= p.M) continue;
      float* pr = p.part + ((long long)sp * p.M + row) * C + 4 * g;
#pragma unroll
      for (int f = 0; f < NF; ++f) *reinterpret_cast<f32x4*>(pr + 16 * f) = acc2[i][f];
    }
    return;
  }
  if constexpr (SPLIT == 2) {     // the slices' sums meet in the workgroup that finishes last (slice_combine); it alone goes on
    __shared__ int ticket;
    if (!slice_combine<RF * NF, 64 * G::RG>(&acc2[0][0], p.part, p.cnt, unit, sp, p.split, tid, &ticket)) return;
  }
  STAMP(4);
  if (hg != 0) return;
  T* y = reinterpret_cast<T*>(p.y);
#pragma unroll
  for (int i = 0; i < RF; ++i) {
    const long long row = m0 + 16 * i + ln;
    if (row >= p.M) continue;
    const float dp = drop_path_scale(p.rng, p.site, row / p.rows_per_sample, p.p_drop);
    float xs[NF][4];              // the shortcut row, read before the first store: y may alias x, so a load left in the store loop
                                  // waits behind the previous store -- NF dependent round trips at the end of every wave
    float4 bs[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      if constexpr (TAILPF) raw4_unpack<T>(xraw[f], xs[f]);
      else ld4(x + row * C + 16 * f + 4 * g, xs[f]);
      bs[f] = *reinterpret_cast<const float4*>(p.b2 + 16 * f + 4 * g);
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int col = 16 * f + 4 * g;
      const float4 bv = bs[f];
      const float v[4] = {xs[f][0] + dp * (acc2[i][f][0] + bv.x), xs[f][1] + dp * (acc2[i][f][1] + bv.y),
                          xs[f][2] + dp * (acc2[i][f][2] + bv.z), xs[f][3] + dp * (acc2[i][f][3] + bv.w)};
      st4(y + row * C + col, v);
    }
  }
  STAMP(5);
}

// =====================================================================================================================
// backward
// =====================================================================================================================
template <typename T, int C, int RFP, int SPLIT = 0, int NW = 4, int HS = 1, int HCX = 0>
__global__ __launch_bounds__(64 * NW, STJ_MLP_MINB) void swin_mlp_bwd_kernel(MlpArgs p) {
  typedef MlpCfg<T, C, RFP, NW, HS, HCX> G;
  constexpr int KS = G::KS, NF = G::NF, RF = G::RF, ND = Chain<T>::ND, KSTEP = G::KSTEP, LK = Mma<T>::LANE_K, NT = G::NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char mlp_smem[];
  __shared__ float red[2][C];
  T* W1s = reinterpret_cast<T*>(mlp_smem);
  T* W2s = W1s + C * G::LD1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, ln = lane & 15;
  const int rg = wave % G::RG, hg = wave / G::RG;          // (see the forward kernel)
  const int unit = SPLIT ? (int)blockIdx.x / p.split : (int)blockIdx.x, sp = SPLIT ? (int)blockIdx.x % p.split : 0;     // see the forward kernel
  const int hs0 = SPLIT ? sp * (4 * C / p.split) : 0, hs1 = SPLIT ? hs0 + 4 * C / p.split : 4 * C;
  const long long m0 = (long long)unit * G::ROWS + rg * (RF * 16);
  const T* x = reinterpret_cast<const T*>(p.x);
  const T* dy = reinterpret_cast<const T*>(p.dy);
  const T* w1 = reinterpret_cast<const T*>(p.w1);
  const T* w2 = reinterpret_cast<const T*>(p.w2);
  for (int c = tid; c < 2 * C; c += NT) (&red[0][0])[c] = 0.f;
  __shared__ float b1s[4 * C];                         // (see the forward kernel)

  MlpStage<T, C, G> stg;
  stg.issue(w1, w2, hs0, tid);
  typename Mma<T>::Frag xa[RF][KS], da[RF][KS];
  load_rows<T, C, RF>(xa, x, m0, p.M, lane);
  load_rows<T, C, RF>(da, dy, m0, p.M, lane);
  for (int c = hs0 + tid; c < hs1; c += NT) b1s[c] = p.b1[c];      // (last: see the forward kernel)
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
      if (row < p.M && sp == 0 && hg == 0) {
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
  // the LayerNorm backward's operands in the accumulator layout (x and dy of the wave's rows): fetched under the last chunk (or in front of the
  // hand-over of the hidden groups / slices), not where the epilogue needs them (see swin_attn_bwd_kernel)
  constexpr bool TAILPF = RF == 1 && SPLIT != 1;
  typename Raw4<T>::type xraw[TAILPF ? NF : 1], dyraw[TAILPF ? NF : 1];
  auto tail_prefetch = [&]() __attribute__((always_inline)) {
    const long long row = m0 + ln < p.M ? m0 + ln : p.M - 1;
#pragma unroll
    for (int f = 0; f < (TAILPF ? NF : 1); ++f) {
      xraw[f] = *reinterpret_cast<const typename Raw4<T>::type*>(x + row * C + 16 * f + 4 * g);
      dyraw[f] = *reinterpret_cast<const typename Raw4<T>::type*>(dy + row * C + 16 * f + 4 * g);
    }
  };

  for (int hc0 = hs0; hc0 < hs1; hc0 += G::HC) {
    __syncthreads();
    if (!STJ_MLP_PREFETCH && hc0 > hs0) stg.issue(w1, w2, hc0, tid);
    stg.commit(W1s, W2s, tid);
    __syncthreads();
    if (STJ_MLP_PREFETCH && hc0 + G::HC < hs1) stg.issue(w1, w2, hc0 + G::HC, tid);
    if constexpr (TAILPF && HS == 1 && SPLIT == 0) { if (hc0 + G::HC >= hs1) tail_prefetch(); }
#pragma unroll 1
    for (int s = hg; s < G::HC / KSTEP; s += HS) {
      f32x4 a1[RF][ND], a3[RF][ND];                    // pre^T and dh^T, [hidden][row]
#pragma unroll
      for (int d = 0; d < ND; ++d) {
        const float4 bv = *reinterpret_cast<const float4*>(b1s + hc0 + s * KSTEP + 16 * d + 4 * g);
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

  if constexpr (TAILPF && (HS > 1 || SPLIT == 2)) tail_prefetch();
  if constexpr (HS > 1) {         // (see the forward kernel)
    __syncthreads();
    f32x4* rb = reinterpret_cast<f32x4*>(mlp_smem) + (rg * RF * NF) * 64 + lane;
    if (hg == 1) {
#pragma unroll
      for (int i = 0; i < RF; ++i)
#pragma unroll
        for (int f = 0; f < NF; ++f) rb[(i * NF + f) * 64] = acc[i][f];
    }
    __syncthreads();
    if (hg == 0) {
#pragma unroll
      for (int i = 0; i < RF; ++i)
#pragma unroll
        for (int f = 0; f < NF; ++f) acc[i][f] += rb[(i * NF + f) * 64];
    }
  }
  if constexpr (SPLIT == 1) {     // this slice's share of d LN(x); swin_split_bwd_epi_kernel sums the slices and runs the LayerNorm backward
    if (hg != 0) return;
#pragma unroll
    for (int i = 0; i < RF; ++i) {
      const long long row = m0 + 16 * i + ln;
      if (row >= p.M) continue;
      float* pr = p.part + ((long long)sp * p.M + row) * C + 4 * g;
#pragma unroll
      for (int f = 0; f < NF; ++f) *reinterpret_cast<f32x4*>(pr + 16 * f) = acc[i][f];
    }
    return;
  }
  if constexpr (SPLIT == 2) {     // (see the forward kernel)
    __shared__ int ticket;
    if (!slice_combine<RF * NF, 64 * G::RG>(&acc[0][0], p.part, p.cnt, unit, sp, p.split, tid, &ticket)) return;
  }
  // LayerNorm backward on the accumulator layout (lane: columns 16 f + 4 g .. +3 of row m0 + 16 i + ln) + the skip gradient
  T* dx = reinterpret_cast<T*>(p.dx);
  if (hg == 0) {                  // (the waves of the other hidden group only take part in the barrier below)
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
    float dyv[NF][4];             // read up front: dx may alias dy, so loads left in the store loop below are serialised behind each store
    float s1 = 0.f, s2 = 0.f;
    if constexpr (TAILPF) {
#pragma unroll
      for (int f = 0; f < NF; ++f) raw4_unpack<T>(dyraw[f], dyv[f]);
    } else if (live) {
#pragma unroll
      for (int f = 0; f < NF; ++f) ld4(dy + row * C + 16 * f + 4 * g, dyv[f]);
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int col = 16 * f + 4 * g;
      float xv[4] = {0.f, 0.f, 0.f, 0.f};
      if constexpr (TAILPF) raw4_unpack<T>(xraw[f], xv);
      else if (live) ld4(x + row * C + col, xv);
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
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = dyv[f][r] + rs[i] * (acc[i][f][r] - s1 - xh[f][r] * s2);
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
  }
  __syncthreads();
  const long long po = (long long)(blockIdx.x % p.nparts) * p.pstride;
  for (int c = tid; c < C; c += NT) { atomicAdd(p.dgamma + po + c, red[0][c]); atomicAdd(p.dbeta + po + c, red[1][c]); }
}

// ---- the second launch of the SPLIT kernels: sum the slices' partial sums and finish the rows -----------------------------------
// forward: y = x + dp * (sum_s part[s] + bias)   (both halves of the block: rows_per_sample rows share one DropPath factor)
template <typename T>
__global__ __launch_bounds__(256) void swin_split_fwd_epi_kernel(const T* x, const float* part, int S, const float* bias, T* y, long long M, int C,
                                                                 const long long* rng, int site, float p_drop, long long rows_per_sample) {
  const long long n4 = M * C / 4;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += gridDim.x * 256ll) {
    const long long row = i / (C / 4);
    const int col = (int)(i % (C / 4)) * 4;
    const float4 bv = *reinterpret_cast<const float4*>(bias + col);
    float a[4] = {bv.x, bv.y, bv.z, bv.w};
    for (int s = 0; s < S; ++s) {
      const float4 v = *reinterpret_cast<const float4*>(part + ((long long)s * M + row) * C + col);
      a[0] += v.x; a[1] += v.y; a[2] += v.z; a[3] += v.w;
    }
    const float dp = drop_path_scale(rng, site, row / rows_per_sample, p_drop);
    float xv[4];
    ld4(x + row * C + col, xv);
    const float v[4] = {xv[0] + dp * a[0], xv[1] + dp * a[1], xv[2] + dp * a[2], xv[3] + dp * a[3]};
    st4(y + row * C + col, v);
  }
}
template <typename T> __device__ __forceinline__ void ld2(const T* p, float& a, float& b) {
  if constexpr (sizeof(T) == 4) { const float2 v = *reinterpret_cast<const float2*>(p); a = v.x; b = v.y; }
  else unpack2<T>(*reinterpret_cast<const uint32_t*>(p), a, b);
}
template <typename T> __device__ __forceinline__ void st2(T* p, float a, float b) {
  if constexpr (sizeof(T) == 4) *reinterpret_cast<float2*>(p) = make_float2(a, b);
  else *reinterpret_cast<uint32_t*>(p) = pack2<T>(a, b);
}
// backward: d = sum_s part[s] = d LN(x); dx = dy + LN'(d) with the row statistics recomputed from x (or read from mean / rstd when given);
// gamma / beta gradients into copy (block % nparts).  A wave owns RPW rows at a time (all their loads in flight together: one row at a
// time ran 19 us for 2048 rows, a chain of load -> three wave reductions per row), a lane 2 adjacent columns of every 128.
template <typename T, int C, int RPW>
__global__ __launch_bounds__(256) void swin_split_bwd_epi_kernel(const T* x, const T* dy, const float* part, int S, const float* gamma, float eps,
                                                                 const float* mean, const float* rstd, T* dx, float* dgamma, float* dbeta,
                                                                 int nparts, long long pstride, long long M) {
  constexpr int NJ = C / 128;
  __shared__ float red[2][C];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int c = tid; c < 2 * C; c += 256) (&red[0][0])[c] = 0.f;
  __syncthreads();
  float dg[NJ][2], db[NJ][2], gm[NJ][2];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const float2 g2 = *reinterpret_cast<const float2*>(gamma + 128 * j + 2 * lane);
    gm[j][0] = g2.x; gm[j][1] = g2.y;
    dg[j][0] = dg[j][1] = db[j][0] = db[j][1] = 0.f;
  }
  for (long long r0 = (blockIdx.x * 4ll + wave) * RPW; r0 < M; r0 += gridDim.x * 4ll * RPW) {
    float xv[RPW][NJ][2], d[RPW][NJ][2], dv[RPW][NJ][2], s[RPW], q[RPW], mu[RPW], rs[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const long long row = r0 + i < M ? r0 + i : M - 1;            // (tail rows repeat the last row; their results are dropped)
      s[i] = 0.f;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const long long o = row * C + 128 * j + 2 * lane;
        ld2<T>(x + o, xv[i][j][0], xv[i][j][1]);
        ld2<T>(dy + o, dv[i][j][0], dv[i][j][1]);
        d[i][j][0] = d[i][j][1] = 0.f;
        s[i] += xv[i][j][0] + xv[i][j][1];
      }
    }
    for (int sl = 0; sl < S; ++sl)
#pragma unroll
      for (int i = 0; i < RPW; ++i) {
        const long long row = r0 + i < M ? r0 + i : M - 1;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const float2 v = *reinterpret_cast<const float2*>(part + ((long long)sl * M + row) * C + 128 * j + 2 * lane);
          d[i][j][0] += v.x; d[i][j][1] += v.y;
        }
      }
    if (mean != nullptr) {
#pragma unroll
      for (int i = 0; i < RPW; ++i) { const long long row = r0 + i < M ? r0 + i : M - 1; mu[i] = mean[row]; rs[i] = rstd[row]; }
    } else {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int i = 0; i < RPW; ++i) s[i] += __shfl_xor(s[i], o, 64);
#pragma unroll
      for (int i = 0; i < RPW; ++i) {
        mu[i] = s[i] * (1.f / C);
        q[i] = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) { const float a = xv[i][j][0] - mu[i], b = xv[i][j][1] - mu[i]; q[i] += a * a + b * b; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int i = 0; i < RPW; ++i) q[i] += __shfl_xor(q[i], o, 64);
#pragma unroll
      for (int i = 0; i < RPW; ++i) rs[i] = rsqrtf(q[i] * (1.f / C) + eps);
    }
    float s1[RPW], s2[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const bool live = r0 + i < M;
      s1[i] = s2[i] = 0.f;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float xh = (xv[i][j][e] - mu[i]) * rs[i];
          xv[i][j][e] = xh;
          const float dd = live ? d[i][j][e] : 0.f;
          dg[j][e] += dd * xh;
          db[j][e] += dd;
          const float t = dd * gm[j][e];
          d[i][j][e] = t;
          s1[i] += t; s2[i] += t * xh;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int i = 0; i < RPW; ++i) { s1[i] += __shfl_xor(s1[i], o, 64); s2[i] += __shfl_xor(s2[i], o, 64); }
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      if (r0 + i >= M) continue;
      const float a1 = s1[i] * (1.f / C), a2 = s2[i] * (1.f / C);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const long long o = (r0 + i) * C + 128 * j + 2 * lane;
        st2<T>(dx + o, dv[i][j][0] + rs[i] * (d[i][j][0] - a1 - xv[i][j][0] * a2), dv[i][j][1] + rs[i] * (d[i][j][1] - a1 - xv[i][j][1] * a2));
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int e = 0; e < 2; ++e) { atomicAdd(&red[0][128 * j + 2 * lane + e], dg[j][e]); atomicAdd(&red[1][128 * j + 2 * lane + e], db[j][e]); }
  __syncthreads();
  const long long po = (long long)(blockIdx.x % nparts) * pstride;
  for (int c = tid; c < C; c += 256) { atomicAdd(dgamma + po + c, red[0][c]); atomicAdd(dbeta + po + c, red[1][c]); }
}
template <typename T>
static int split_fwd_epi(const void* x, const float* part, int S, const float* bias, void* y, long long M, int C, const long long* rng, int site,
                         float p_drop, long long rows_per_sample, hipStream_t st) {
  const long long n4 = M * C / 4;
  const int grid = (int)((n4 + 255) / 256 > 4096 ? 4096 : (n4 + 255) / 256);
  hipLaunchKernelGGL(swin_split_fwd_epi_kernel<T>, dim3(grid), dim3(256), 0, st, (const T*)x, part, S, bias, (T*)y, M, C, rng, site, p_drop, rows_per_sample);
  return stj_check_launch("swin_split_fwd_epi");
}
template <typename T>
static int split_bwd_epi(const void* x, const void* dy, const float* part, int S, const float* gamma, float eps, const float* mean, const float* rstd,
                         void* dx, float* dgamma, float* dbeta, int nparts, long long pstride, long long M, hipStream_t st) {
  // 2 rows per wave in flight, 256 workgroups at 2048 rows: 16 us; (4 rows, 128 workgroups) 27 us, (1, 512) 23 us, (1 at a time, 128) 19 us
  // (at most one workgroup per CU, the row loop does the rest: cfg-512's 8192 rows as 1024 eight-row workgroups measured 741 scenes/s end
  //  to end, 512 workgroups 745, 256 workgroups 748)
  const int grid = (int)std::min<long long>((M + 7) / 8, 256);
  hipLaunchKernelGGL((swin_split_bwd_epi_kernel<T, 384, 2>), dim3(grid), dim3(256), 0, st, (const T*)x, (const T*)dy, part, S, gamma, eps, mean, rstd,
                     (T*)dx, dgamma, dbeta, nparts, pstride, M);
  return stj_check_launch("swin_split_bwd_epi");
}
constexpr int SPLIT_MAX = 8;         // slices of a split launch (the workspace holds SPLIT_MAX x M x C floats)
// slices per unit: enough for ~256 workgroups, no more -- every slice costs a [M][C] f32 partial-sum pass (cfg-512's 8192-row stage with
// 8 / 6 slices: 100 / 75 MB per kernel, the step 3 % slower than layer by layer; with 2: the 25 MB of the 2048-row stage)
// (cfg-512, 128 row blocks, scenes/s by hidden slices per block: 1: 631, 2: 671-674, 4: 663-665, 8: 651)
static int mlp_split_for(long long M) { const long long blocks = (M + 63) / 64; return blocks <= 32 ? 8 : (blocks <= 64 ? 4 : 2); }
static int attn_split_for(long long windows) { return windows <= 48 ? 6 : 2; }

template <typename T, int C, int RFP, int SPLIT = 0, int NW = 4, int HS = 1, int HCX = 0>
static int mlp_launch(bool bwd, const MlpArgs& a, hipStream_t st) {
  typedef MlpCfg<T, C, RFP, NW, HS, HCX> G;
  const void* fn = bwd ? (const void*)swin_mlp_bwd_kernel<T, C, RFP, SPLIT, NW, HS, HCX> : (const void*)swin_mlp_fwd_kernel<T, C, RFP, SPLIT, NW, HS, HCX>;
  static PerDevice<bool> attr[2];
  if (!attr[bwd]) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES) != hipSuccess) {
      stj_set_error("swin_mlp: cannot reserve %d bytes of LDS", G::LDS_BYTES); return STJ_ELAUNCH;
    }
    attr[bwd] = true;
  }
  dim3 grid((unsigned)((a.M + G::ROWS - 1) / G::ROWS) * (SPLIT ? a.split : 1));
  if (bwd) hipLaunchKernelGGL((swin_mlp_bwd_kernel<T, C, RFP, SPLIT, NW, HS, HCX>), grid, dim3(G::NT), G::LDS_BYTES, st, a);
  else hipLaunchKernelGGL((swin_mlp_fwd_kernel<T, C, RFP, SPLIT, NW, HS, HCX>), grid, dim3(G::NT), G::LDS_BYTES, st, a);
  return stj_check_launch(bwd ? "stj_swin_mlp_bwd" : "stj_swin_mlp_fwd");
}
template <typename T>
static int mlp_dispatch(bool bwd, int C, const MlpArgs& a, hipStream_t st) {
  // rows per block = 64 * RF: two row fragments per wave halve the weight-fragment LDS reads per MFMA, but the grid must still cover
  // the 256 CUs (32768 rows: 256 blocks of 128; 8192 rows: 128 blocks of 64)
  const bool two = a.M >= 256 * 128;
  switch (C) {
    case 96:         // 128-row blocks of eight waves (f32: of four waves with two row fragments each) | 64-row blocks of four
      // (Round 6, at >= 131072 rows -- cfg-512's stage 0, the B = 32 inference forward: 4 rounds of 128-row workgroups --: two / four 16-row
      //  fragments per wave, i.e. the staged weight chunks and their fragment reads serving 2 / 4 x the rows in 2 / 1 rounds (forward 158 / 250
      //  registers; backward 242, four fragments spill): cfg-512 776 / 778 against 779 scenes/s, inference inside its noise
      //  (profiles/r06_zb_mlp96_rf.txt): the kernel's time is its per-row-fragment latency chain, not staging or rounds.  Not kept.)
      // weight chunks of 192 hidden columns (two per block instead of four: half the barriers and commits) at 32768 rows: train step 1420 / 1413 /
      // 1414 / 1409 against 1411 / 1407 / 1406 / 1408 scenes/s with 96; at 131072 rows (B = 32 inference, cfg-512) 96 stays: inference 5744-5757
      // against 5790-5820, cfg-512 equal (profiles/r06_zo_mlp_hc96.txt)
      if constexpr (sizeof(T) == 2) {
        if (two && a.M < 4LL * 256 * 128) return mlp_launch<T, 96, 1, 0, 8, 1, STJ_MLP_HC96W>(bwd, a, st);
        return two ? mlp_launch<T, 96, 1, 0, 8>(bwd, a, st) : mlp_launch<T, 96, 1>(bwd, a, st);
      }
      else return two ? mlp_launch<T, 96, 2>(bwd, a, st) : mlp_launch<T, 96, 1>(bwd, a, st);
    case 192: {
      // 8192 rows (cfg-256's 32 x 32 stage at B = 8) are 128 row blocks of 64: with the workspace the hidden dimension is cut in two, 256
      // workgroups that each stage half of the weights and meet inside the launch (SPLIT == 2)
      if constexpr (sizeof(T) == 4) { if (two) return mlp_launch<T, 192, 2>(bwd, a, st); }
      if constexpr (sizeof(T) == 2) {
        if (two) return mlp_launch<T, 192, 1, 0, 8>(bwd, a, st);
        if (a.part != nullptr && (a.M + 63) / 64 <= FIX_CNT_BYTES / 4) {
          MlpArgs s = a;
          s.split = 2;
          return mlp_launch<T, 192, 1, 2, 8, 2>(bwd, s, st);       // 64-row blocks x 2 slices of the hidden dimension, eight waves: 2 hidden groups x 4 row groups
        }
      }
      if constexpr (sizeof(T) == 2) return mlp_launch<T, 192, 1, 0, 8, 2>(bwd, a, st);       // no workspace: one workgroup per row block, its two hidden groups meet in LDS
      else return mlp_launch<T, 192, 1>(bwd, a, st);
    }
    case 384: {
      // (a one-workgroup-per-row-block form without the workspace existed: 32 workgroups at B = 8, never taken by ops.py, 0.2 MB of code)
      if (a.part == nullptr) { stj_set_error("swin_mlp: C = 384 needs the workspace (stj_swin_split_workspace_bytes)"); return STJ_EINVAL; }
      MlpArgs s = a;                 // (row block, hidden slice) workgroups + the finishing launch
      s.split = mlp_split_for(a.M);
      if constexpr (sizeof(T) == 2) {       // two slices (cfg-512's 8192 rows): they meet inside the launch, no finishing launch
        // forward at >= 8192 rows (cfg-512): 128-row blocks of eight waves x 4 hidden slices + the finishing launch -- the weight slices are
        // staged once per 128 rows instead of once per 64 (1.18 MB per workgroup at ~60 GB/s per CU out of L2 is ~20 us of the launch):
        // 55 vs 64-67 us.  (Not the backward: its eight-wave form spills 163 registers, 128 vs 111-117 us.)
        // cfg-512 end to end, same box, alternating: 760-764 vs 754-758 scenes/s.
        if (!bwd && a.M >= 8192) {
          s.split = 4;
          const int rc = mlp_launch<T, 384, 1, 1, 8>(false, s, st);
          if (rc != STJ_OK) return rc;
          return split_fwd_epi<T>(a.x, a.part, s.split, a.b2, a.y, a.M, C, a.rng, a.site, a.p_drop, a.rows_per_sample, st);
        }
        if (s.split == 2 && (a.M + 63) / 64 <= FIX_CNT_BYTES / 4) return mlp_launch<T, 384, 1, 2>(bwd, s, st);
      }
      const int rc = mlp_launch<T, 384, 1, 1>(bwd, s, st);
      if (rc != STJ_OK) return rc;
      if (bwd) return split_bwd_epi<T>(a.x, a.dy, a.part, s.split, a.gamma, a.eps, nullptr, nullptr, a.dx, a.dgamma, a.dbeta, a.nparts, a.pstride, a.M, st);
      return split_fwd_epi<T>(a.x, a.part, s.split, a.b2, a.y, a.M, C, a.rng, a.site, a.p_drop, a.rows_per_sample, st);
    }
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

// bytes of the f32 workspace `ws` of the four stj_swin_* entry points at C = 384 (NULL: the one-workgroup-per-row-block kernels; other
// C: ignored): the slices' partial sums [SPLIT_MAX][M][C]
// partial-sum / slab region of the workspace; the arrival counters follow it
static long long split_region_bytes(long long M, int C) {
  const long long Mp = (M + 63) / 64 * 64;
  if (C == 384) return (long long)SPLIT_MAX * Mp * C * 4;
  if (C == 192 && M < 256 * 128) return 2 * Mp * C * 4;        // two slabs per 64-row unit
  return 0;
}
static int* split_counters(void* ws, long long M, int C) { return ws ? reinterpret_cast<int*>(reinterpret_cast<char*>(ws) + split_region_bytes(M, C)) : nullptr; }
extern "C" long long stj_swin_split_workspace_bytes(long long M, int C) {
  const long long r = split_region_bytes(M, C);
  return r ? r + FIX_CNT_BYTES : 0;
}

extern "C" int stj_swin_mlp_fwd(const void* x, const float* gamma, const float* beta, const void* w1, const float* b1, const void* w2,
                                const float* b2, void* y, long long M, int C, float eps, const long long* rng_state, int site,
                                float p_drop, long long rows_per_sample, int dtype, void* ws, hipStream_t stream) {
  MlpArgs a = {};
  a.part = reinterpret_cast<float*>(ws); a.cnt = split_counters(ws, M, C);
  a.x = x; a.gamma = gamma; a.beta = beta; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.y = y; a.M = M; a.eps = eps;
  a.rng = rng_state; a.site = site; a.p_drop = p_drop; a.rows_per_sample = rows_per_sample;
  return mlp_any(false, C, dtype, a, stream);
}

extern "C" int stj_swin_mlp_bwd(const void* x, const void* dy, const float* gamma, const float* beta, const void* w1, const float* b1,
                                const void* w2, void* dx, void* h, void* dpre, void* ln, void* dys, float* dgamma, float* dbeta,
                                int nparts, long long part_stride, long long M, int C, float eps, const long long* rng_state, int site,
                                float p_drop, long long rows_per_sample, int dtype, void* ws, hipStream_t stream) {
  if (nparts < 1) { stj_set_error("swin_mlp_bwd: nparts must be >= 1"); return STJ_EINVAL; }
  MlpArgs a = {};
  a.part = reinterpret_cast<float*>(ws); a.cnt = split_counters(ws, M, C);
  a.x = x; a.dy = dy; a.gamma = gamma; a.beta = beta; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.dx = dx; a.h = h; a.dpre = dpre; a.ln = ln;
  a.dys = dys; a.dgamma = dgamma; a.dbeta = dbeta; a.nparts = nparts; a.pstride = part_stride; a.M = M; a.eps = eps;
  a.rng = rng_state; a.site = site; a.p_drop = p_drop; a.rows_per_sample = rows_per_sample;
  return mlp_any(true, C, dtype, a, stream);
}

// =====================================================================================================================
// Fused attention half of SwinTransformerBlock, forward:
//     y = x + dp * ( proj( window_attention( LN(x) Wqkv + bqkv ) ) + bproj )
// reference modules.py:225-258 (norm1, roll, window_partition, attn, window_reverse, roll, drop_path + shortcut),
// :103-134 (WindowAttention.call), :189-216 (shift mask), :49-63 (partition / reverse) -- SURVEY.md K2.
// One workgroup = one 8x8 window (64 tokens), 4 waves, a wave OWNS 16 of the window's tokens (queries) end to end:
//   phase 1  LayerNorm in registers (rows are B fragments loaded straight from global, gathered through the shift / window
//            index arithmetic), q|k|v^T = W^T LN(x)^T on the accumulator layout, written token-major into an LDS tile
//            (the only activation data that goes through LDS: every wave needs all 64 keys / values of the window);
//   phase 2  per head: S^T = K Q^T (+ relative-position bias, -100 shift mask), softmax over the keys held by the lane and its
//            3 partner lanes (two xor-shuffles), O^T = V^T P^T with P^T chained in registers (Chain<T>, see the MLP kernels)
//            and V^T fragments by LDS transpose reads of the token-major tile;
//   phase 3  out^T += Wproj^T O^T, again chained in registers; epilogue bias + DropPath + shortcut, stores in token order.
// Heads are processed in groups of HG (LDS budget): a pass stages the q/k/v weight columns of its heads and the matching
// rows of Wproj.  Training additionally writes what backward needs: qkv [M,3C], a = attention output [M,C], ln = LN(x) [M,C],
// mean / rstd [M].
// =====================================================================================================================
template <typename T, int C, int HGP = 0> struct AttnCfg {
  static constexpr int KSTEP = Mma<T>::KSTEP;
  static constexpr int KS = C / KSTEP, NF = C / 16, HEADS = C / 32;
  static constexpr int HG = HGP ? HGP : (sizeof(T) == 2 ? (C == 96 ? STJ_ATTN_HG96 : (C == 192 ? STJ_ATTN_HG192 : 1)) : 1);     // heads per pass
  static constexpr int GC = 32 * HG;                                   // q (= k = v) columns per pass
  static constexpr int LDT = 3 * GC + (sizeof(T) == 2 ? 16 : 8);       // token-major q|k|v tile [64][LDT]
  static constexpr int LDW = 3 * GC + 4;                               // Wqkv slice image [C][LDW] ([k = c][q seg | k seg | v seg])
  static constexpr int LDP = C + 4;                                    // Wproj slice image [GC][LDP] ([k = c of the group][oc])
  // f32 at C = 384: the Wqkv slice of one head is 150 KB; it is staged in KH = 2 halves of the contraction (model) dimension, the q|k|v
  // accumulators of the wave's 16 tokens kept across the two (round 6: the f32 parity mode then runs the fused kernel at every width)
  static constexpr int KH = (sizeof(T) == 4 && C == 384) ? 2 : 1;
  static constexpr int TILE = 64 * LDT, WQ = (C / KH) * LDW, WP = GC * LDP;
  static constexpr int LDS_BYTES = (TILE + WQ + WP) * (int)sizeof(T) + HG * 225 * 4 + 2 * 64 * 4;
  static_assert(HEADS % HG == 0, "head groups");
};

struct AttnArgs {
  const void* x; const float* gamma; const float* beta; const void* wqkv; const float* bqkv; const float* table;
  const void* wproj; const float* bproj; void* y;
  void* qkv; void* a; void* ln; float* mean; float* rstd;         // training hand-offs (all NULL for inference)
  int B, res, shift; float eps;
  const long long* rng; int site; float p_drop;
  int split; float* part; int* cnt;   // SPLIT kernel: workgroup = (window, slice of the heads); partial sums [split][M][C] f32 in token order | slabs + arrival counters
};

// weight slices of one head group, global -> registers (issued ahead) -> LDS
template <typename T, int C, int HGP = 0> struct AttnStage {
  typedef AttnCfg<T, C, HGP> G;
  static constexpr int VN = Vec<T>::N, GC = G::GC;
  static constexpr int CPS = GC / VN, CPP = C / VN;
  static constexpr int NQ = (C * 3 * CPS + 255) / 256, NP = (GC * CPP + 255) / 256;
  uint4 rq[NQ], rp[NP];
  __device__ __forceinline__ void issue(const T* wq, const T* wp, int h0, int tid) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int q = tid + i * 256;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (q < C * 3 * CPS) {
        const int k = q / (3 * CPS), r = q % (3 * CPS), seg = r / CPS, c = (r % CPS) * VN;
        v = *reinterpret_cast<const uint4*>(wq + (long long)k * (3 * C) + seg * C + 32 * h0 + c);
      }
      rq[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int q = tid + i * 256;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (q < GC * CPP) v = *reinterpret_cast<const uint4*>(wp + (long long)(32 * h0 + q / CPP) * C + (q % CPP) * VN);
      rp[i] = v;
    }
  }
  __device__ __forceinline__ void commit(T* Wqs, T* Wps, int tid) const {
    auto put = [](T* d, const uint4& v) {
      if constexpr (sizeof(T) == 2) { uint2* d2 = reinterpret_cast<uint2*>(d); d2[0] = make_uint2(v.x, v.y); d2[1] = make_uint2(v.z, v.w); }
      else *reinterpret_cast<uint4*>(d) = v;
    };
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int q = tid + i * 256;
      if (q < C * 3 * CPS) { const int k = q / (3 * CPS), r = q % (3 * CPS); put(Wqs + k * G::LDW + (r / CPS) * GC + (r % CPS) * VN, rq[i]); }
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int q = tid + i * 256;
      if (q < GC * CPP) put(Wps + (q / CPP) * G::LDP + (q % CPP) * VN, rp[i]);
    }
  }
};

template <typename T, int C, int SPLIT = 0, int HGP = 0>
__global__ __launch_bounds__(256, (C == 96 && sizeof(T) == 2) ? STJ_ATTN_MINB96 : STJ_ATTN_MINB) void swin_attn_fwd_kernel(AttnArgs p) {
  typedef AttnCfg<T, C, HGP> G;
  constexpr int KS = G::KS, NF = G::NF, HG = G::HG, GC = G::GC, KSTEP = G::KSTEP, ND = Chain<T>::ND, LK = Mma<T>::LANE_K;
  constexpr int VN = Vec<T>::N;
  extern __shared__ __attribute__((aligned(16))) unsigned char at_smem[];
  T* tile = reinterpret_cast<T*>(at_smem);
  T* Wqs = tile + G::TILE;
  T* Wps = Wqs + G::WQ;
  float* tbl = reinterpret_cast<float*>(Wps + G::WP);               // [HG][225]
  int* tok = reinterpret_cast<int*>(tbl + HG * 225);                // [64] token index of window slot t (shift + partition)
  int* lab = tok + 64;                                              // [64] shift-mask region label

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g = lane >> 4, ln = lane & 15;
  const T* wq = reinterpret_cast<const T*>(p.wqkv);
  const T* wp = reinterpret_cast<const T*>(p.wproj);
  // SPLIT (C = 384: 32 windows at B = 8 cannot fill the chip): workgroup = (window, slice sp of the heads)
  const int unit = SPLIT ? (int)blockIdx.x / p.split : (int)blockIdx.x, sp = SPLIT ? (int)blockIdx.x % p.split : 0;
  const int hb0 = SPLIT ? sp * (G::HEADS / p.split) : 0, hb1 = SPLIT ? hb0 + G::HEADS / p.split : G::HEADS;
  __shared__ float bqs[3 * C];                       // qkv bias in LDS (a global load inside the head loop would wait for the prefetched weights)
  AttnStage<T, C, HGP> stg;
  if constexpr (G::KH == 1) stg.issue(wq, wp, hb0, tid);      // first head group's weights in flight under the row gather + LayerNorm
  const int nwx = p.res / 8, nW = nwx * nwx;
  const int win = unit % nW, b = unit / nW;
  const int wy = win / nwx, wx = win % nwx;
  const long long N = (long long)p.res * p.res;
  if (tid < 64) {
    const int t = tid;
    int ry = wy * 8 + (t >> 3), rx = wx * 8 + (t & 7);
    int sy = ry + p.shift; if (sy >= p.res) sy -= p.res;
    int sx = rx + p.shift; if (sx >= p.res) sx -= p.res;
    tok[t] = sy * p.res + sx;
    const int ly = ry < p.res - 8 ? 0 : (ry < p.res - p.shift ? 1 : 2);
    const int lx = rx < p.res - 8 ? 0 : (rx < p.res - p.shift ? 1 : 2);
    lab[t] = ly * 3 + lx;
  }
  __syncthreads();
  const T* x = reinterpret_cast<const T*>(p.x) + (long long)b * N * C;
  const long long myrow = (long long)b * N + tok[16 * wv + ln];      // global row (token) this lane's query lives at
  // ---- rows of this wave as B fragments (gathered), LayerNorm in registers
  typename Mma<T>::Frag xa[1][KS];
  {
    const T* px = x + (long long)tok[16 * wv + ln] * C + LK * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xa[0][ks] = Mma<T>::from_global(px + ks * KSTEP);
  }
  for (int c = tid; c < 3 * C; c += 256) bqs[c] = p.bqkv[c];          // (behind the weight and row loads: one memory round trip for the three)
  float mu[1], rs[1];
  ln_rows<T, C, 1>(xa, p.gamma, p.beta, p.eps, mu, rs, lane);
  if (p.ln && sp == 0) {
    T* lq = reinterpret_cast<T*>(p.ln) + myrow * C + LK * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) *reinterpret_cast<typename Mma<T>::Frag*>(lq + ks * KSTEP) = xa[0][ks];
    if (g == 0) { p.mean[myrow] = mu[0]; p.rstd[myrow] = rs[0]; }
  }

  f32x4 acco[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) acco[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float scale = 0.17677669529663687f;       // 32^-1/2
  // the shortcut row in the accumulator layout: fetched under the last pass / in front of the hand-over (see swin_attn_bwd_kernel)
  constexpr bool TAILPF = SPLIT != 1;
  typename Raw4<T>::type xraw[TAILPF ? NF : 1];
  auto tail_prefetch = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int f = 0; f < (TAILPF ? NF : 1); ++f)
      xraw[f] = *reinterpret_cast<const typename Raw4<T>::type*>(reinterpret_cast<const T*>(p.x) + myrow * C + 16 * f + 4 * g);
  };

  for (int h0 = hb0; h0 < hb1; h0 += HG) {
    // ---- weight slices of this head group (Wqkv[:, seg*C + 32 h0 .. +GC] for seg = q,k,v; Wproj[32 h0 .. +GC, :]): registers -> LDS
    if constexpr (G::KH == 1) {
    stg.commit(Wqs, Wps, tid);
    for (int q = tid; q < HG * 225; q += 256) tbl[q] = p.table[(q % 225) * G::HEADS + h0 + q / 225];
    __syncthreads();
    if (h0 + HG < hb1) stg.issue(wq, wp, h0 + HG, tid);            // next group's slices fly while this group computes
    // ---- phase 1: q|k|v of this group for the wave's 16 tokens -> tile
#pragma unroll 1
    for (int f = 0; f < 3 * GC / 16; ++f) {
      const int seg = (16 * f) / GC, within = (16 * f) % GC;
      const float4 bv = *reinterpret_cast<const float4*>(bqs + seg * C + 32 * h0 + within + 4 * g);
      f32x4 a = (f32x4){bv.x, bv.y, bv.z, bv.w};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) a = Mma<T>::mma(Mma<T>::load_tr(Wqs, G::LDW, 16 * f, ks * KSTEP, lane), xa[0][ks], a);
      const float v[4] = {a[0], a[1], a[2], a[3]};
      st4(tile + (16 * wv + ln) * G::LDT + 16 * f + 4 * g, v);
    }
    } else {
      // KH halves of the contraction dimension (see AttnCfg): the Wproj slice and the table first, then per half: rows [kh C / KH, +C / KH) of
      // the Wqkv slice -> LDS (straight copies: this is the parity mode's path), the q|k|v accumulators carried across the halves
      constexpr int CPS = GC / VN, CPP = C / VN, CH = C / G::KH, NQF = 3 * GC / 16;
      for (int q = tid; q < GC * CPP; q += 256)
        *reinterpret_cast<uint4*>(Wps + (q / CPP) * G::LDP + (q % CPP) * VN) = *reinterpret_cast<const uint4*>(wp + (long long)(32 * h0 + q / CPP) * C + (q % CPP) * VN);
      for (int q = tid; q < HG * 225; q += 256) tbl[q] = p.table[(q % 225) * G::HEADS + h0 + q / 225];
      f32x4 qa[NQF];
#pragma unroll
      for (int f = 0; f < NQF; ++f) {
        const int seg = (16 * f) / GC, within = (16 * f) % GC;
        const float4 bv = *reinterpret_cast<const float4*>(bqs + seg * C + 32 * h0 + within + 4 * g);
        qa[f] = (f32x4){bv.x, bv.y, bv.z, bv.w};
      }
#pragma unroll 1
      for (int kh = 0; kh < G::KH; ++kh) {
        if (kh > 0) __syncthreads();                 // the previous half's readers are done
        for (int q = tid; q < CH * 3 * CPS; q += 256) {
          const int k = q / (3 * CPS), r = q % (3 * CPS), seg = r / CPS, c = (r % CPS) * VN;
          *reinterpret_cast<uint4*>(Wqs + k * G::LDW + seg * GC + c) = *reinterpret_cast<const uint4*>(wq + (long long)(kh * CH + k) * (3 * C) + seg * C + 32 * h0 + c);
        }
        __syncthreads();
#pragma unroll
        for (int f = 0; f < NQF; ++f)
#pragma unroll
          for (int ks = 0; ks < KS / G::KH; ++ks) {
            // (xa indexed by a runtime half: both candidates are registers, picked by a select)
            const typename Mma<T>::Frag xb = kh == 0 ? xa[0][ks] : xa[0][KS / G::KH + ks];
            qa[f] = Mma<T>::mma(Mma<T>::load_tr(Wqs, G::LDW, 16 * f, ks * KSTEP, lane), xb, qa[f]);
          }
      }
#pragma unroll
      for (int f = 0; f < NQF; ++f) {
        const float v[4] = {qa[f][0], qa[f][1], qa[f][2], qa[f][3]};
        st4(tile + (16 * wv + ln) * G::LDT + 16 * f + 4 * g, v);
      }
    }
    __syncthreads();
    if (p.qkv) {                                     // coalesced copy of the tile to qkv[M, 3C] (rows in original token order)
      constexpr int CPS = GC / VN;
      T* qo = reinterpret_cast<T*>(p.qkv) + (long long)b * N * 3 * C;
      for (int q = tid; q < 64 * 3 * CPS; q += 256) {
        const int t = q / (3 * CPS), r = q % (3 * CPS), seg = r / CPS, c = (r % CPS) * VN;
        *reinterpret_cast<uint4*>(qo + (long long)tok[t] * 3 * C + seg * C + 32 * h0 + c) =
            *reinterpret_cast<const uint4*>(tile + t * G::LDT + seg * GC + c);
      }
    }
    // ---- phase 2: attention of the wave's 16 queries, head by head
    f32x4 of[2 * HG];
    const int qi = 16 * wv + ln;
#pragma unroll
    for (int hh = 0; hh < HG; ++hh) {
      f32x4 st[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        st[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k0 = 0; k0 < 32; k0 += KSTEP)
          st[j] = Mma<T>::mma(Mma<T>::load(tile + GC + 32 * hh, G::LDT, 16 * j, k0, lane), Mma<T>::load(tile + 32 * hh, G::LDT, 16 * wv, k0, lane), st[j]);
      }
      float m = -INFINITY;
      const int mylab = lab[qi];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = 16 * j + 4 * g + r;
          float v = st[j][r] * scale + tbl[hh * 225 + ((qi >> 3) - (key >> 3) + 7) * 15 + ((qi & 7) - (key & 7) + 7)];
          if (p.shift > 0 && lab[key] != mylab) v += -100.0f;
          st[j][r] = v;
          m = fmaxf(m, v);
        }
      m = fmaxf(m, __shfl_xor(m, 16, 64)); m = fmaxf(m, __shfl_xor(m, 32, 64));
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float e = __expf(st[j][r] - m); st[j][r] = e; sum += e; }
      sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);
      const float inv = 1.f / sum;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) st[j][r] *= inv;
      of[2 * hh] = (f32x4){0.f, 0.f, 0.f, 0.f}; of[2 * hh + 1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 64 / KSTEP; ++s) {
        const typename Chain<T>::Frag pf = Chain<T>::from_acc(&st[s * ND]);
#pragma unroll
        for (int jd = 0; jd < 2; ++jd)
          of[2 * hh + jd] = Mma<T>::mma(Chain<T>::ldA_tr(tile + 2 * GC + 32 * hh, G::LDT, 16 * jd, s * KSTEP, lane), pf, of[2 * hh + jd]);
      }
      if (p.a) {
        T* ao = reinterpret_cast<T*>(p.a) + myrow * C + 32 * (h0 + hh) + 4 * g;
#pragma unroll
        for (int jd = 0; jd < 2; ++jd) { const float v[4] = {of[2 * hh + jd][0], of[2 * hh + jd][1], of[2 * hh + jd][2], of[2 * hh + jd][3]}; st4(ao + 16 * jd, v); }
      }
    }
    if constexpr (TAILPF && SPLIT == 0) { if (h0 + HG >= hb1) tail_prefetch(); }
    // ---- phase 3: out^T += Wproj[32 h0 .. +GC, :]^T O^T, O^T chained from the accumulators
#pragma unroll
    for (int kk = 0; kk < GC / KSTEP; ++kk) {
      const typename Chain<T>::Frag ofr = Chain<T>::from_acc(&of[kk * ND]);
#pragma unroll
      for (int f = 0; f < NF; ++f) acco[f] = Mma<T>::mma(Chain<T>::ldA_tr(Wps, G::LDP, 16 * f, kk * KSTEP, lane), ofr, acco[f]);
    }
    __syncthreads();                                  // tile / weight images are overwritten by the next pass
  }

  // ---- epilogue: y = x + dp * (out + bproj)
  if constexpr (SPLIT == 1) {     // this head slice's share of the projection; swin_split_fwd_epi_kernel finishes the rows
    float* pr = p.part + ((long long)sp * p.B * N + myrow) * C + 4 * g;
#pragma unroll
    for (int f = 0; f < NF; ++f) *reinterpret_cast<f32x4*>(pr + 16 * f) = acco[f];
    return;
  }
  if constexpr (SPLIT == 2) {     // the head slices' shares meet in the workgroup that finishes last (slice_combine); it alone goes on
    __shared__ int ticket;
    tail_prefetch();
    if (!slice_combine<NF>(acco, p.part, p.cnt, unit, sp, p.split, tid, &ticket)) return;
  }
  const float dp = drop_path_scale(p.rng, p.site, b, p.p_drop);
  T* y = reinterpret_cast<T*>(p.y) + myrow * C;
  const T* xr = reinterpret_cast<const T*>(p.x) + myrow * C;
  float xs[NF][4];                // shortcut row read before the first store (y may alias x: see swin_mlp_fwd_kernel)
  float4 bs[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    if constexpr (TAILPF) raw4_unpack<T>(xraw[f], xs[f]);
    else ld4(xr + 16 * f + 4 * g, xs[f]);
    bs[f] = *reinterpret_cast<const float4*>(p.bproj + 16 * f + 4 * g);
  }
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const int col = 16 * f + 4 * g;
    const float4 bv = bs[f];
    const float v[4] = {xs[f][0] + dp * (acco[f][0] + bv.x), xs[f][1] + dp * (acco[f][1] + bv.y), xs[f][2] + dp * (acco[f][2] + bv.z), xs[f][3] + dp * (acco[f][3] + bv.w)};
    st4(y + col, v);
  }
}

template <typename T, int C, int SPLIT = 0, int HGP = 0>
static int attn_launch(const AttnArgs& a, hipStream_t st) {
  typedef AttnCfg<T, C, HGP> G;
  static PerDevice<bool> attr;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)swin_attn_fwd_kernel<T, C, SPLIT, HGP>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES) != hipSuccess) {
      stj_set_error("swin_attn: cannot reserve %d bytes of LDS", G::LDS_BYTES); return STJ_ELAUNCH;
    }
    attr = true;
  }
  const int nW = (a.res / 8) * (a.res / 8);
  hipLaunchKernelGGL((swin_attn_fwd_kernel<T, C, SPLIT, HGP>), dim3((unsigned)(a.B * nW * (SPLIT ? a.split : 1))), dim3(256), G::LDS_BYTES, st, a);
  return stj_check_launch("stj_swin_attn_fwd");
}
// C = 384 (the 16 x 16 stage: 32 windows at B = 8): (window, head slice) workgroups + the finishing launch; 16-bit types with a workspace
// (the f32 weight slice of a head does not fit LDS next to the tile: that mode keeps the layer-by-layer path)
template <typename T>
static int attn_split384(const AttnArgs& a, hipStream_t st) {
  if constexpr (sizeof(T) == 2) {
    if (a.part == nullptr) { stj_set_error("swin_attn: C = 384 needs the workspace (stj_swin_split_workspace_bytes)"); return STJ_EINVAL; }
    AttnArgs s = a;
    const long long N = (long long)a.res * a.res;
    s.split = attn_split_for(a.B * (N / 64));          // 6 slices of 2 heads (32 windows at B = 8: 192 workgroups) or 2 of 6
    if (s.split == 2 && a.B * (N / 64) <= FIX_CNT_BYTES / 4) return attn_launch<T, 384, 2>(s, st);      // two slices meet inside the launch
    const int rc = attn_launch<T, 384, 1>(s, st);
    if (rc != STJ_OK) return rc;
    return split_fwd_epi<T>(a.x, a.part, s.split, a.bproj, a.y, a.B * N, 384, a.rng, a.site, a.p_drop, N, st);
  } else {
    // f32 (parity mode): (window, slice of 2 heads) workgroups, the Wqkv slice of a head staged in two halves of the model dimension
    // (AttnCfg::KH), partial sums + the finishing launch
    if (a.part == nullptr) { stj_set_error("swin_attn: C = 384 needs the workspace (stj_swin_split_workspace_bytes)"); return STJ_EINVAL; }
    AttnArgs s = a;
    const long long N = (long long)a.res * a.res;
    s.split = 6;
    const int rc = attn_launch<T, 384, 1>(s, st);
    if (rc != STJ_OK) return rc;
    return split_fwd_epi<T>(a.x, a.part, s.split, a.bproj, a.y, a.B * N, 384, a.rng, a.site, a.p_drop, N, st);
  }
}
template <typename T>
static int attn_dispatch(int C, const AttnArgs& a, hipStream_t st) {
  switch (C) {
    case 96: return attn_launch<T, 96>(a, st);
    case 192: {
      // fewer than 256 windows (cfg-256's 32 x 32 stage at B = 8 has 128): two workgroups per window, three heads each, one head per pass
      // (65 KB of LDS instead of 127), meeting inside the launch (SPLIT == 2)
      const long long nwin = (long long)a.B * (a.res / 8) * (a.res / 8);
      if constexpr (sizeof(T) == 2) {
        if (a.part != nullptr && nwin < 256) { AttnArgs s = a; s.split = 2; return attn_launch<T, 192, 2, 1>(s, st); }
      }
      return attn_launch<T, 192>(a, st);
    }
    case 384: return attn_split384<T>(a, st);
    default: stj_set_error("swin_attn: C must be 96, 192 or 384 (got %d)", C); return STJ_EUNSUPPORTED;
  }
}

extern "C" int stj_swin_attn_fwd(const void* x, const float* gamma, const float* beta, const void* wqkv, const float* bqkv,
                                 const float* table, const void* wproj, const float* bproj, void* y, void* qkv, void* a, void* ln,
                                 float* mean, float* rstd, int B, int res, int C, int shift, float eps, const long long* rng_state,
                                 int site, float p_drop, int dtype, void* ws, hipStream_t stream) {
  if (B <= 0) return STJ_OK;
  if (res % 8 != 0 || shift < 0 || shift >= 8) { stj_set_error("swin_attn: res %% 8 != 0 or bad shift"); return STJ_EINVAL; }
  if (!(p_drop >= 0.f && p_drop < 1.f)) { stj_set_error("swin_attn: need 0 <= p_drop < 1"); return STJ_EINVAL; }
  if ((qkv || a || ln) && !(qkv && a && ln && mean && rstd)) { stj_set_error("swin_attn: training outputs come all or none"); return STJ_EINVAL; }
  AttnArgs p = {};
  p.x = x; p.gamma = gamma; p.beta = beta; p.wqkv = wqkv; p.bqkv = bqkv; p.table = table; p.wproj = wproj; p.bproj = bproj; p.y = y;
  p.qkv = qkv; p.a = a; p.ln = ln; p.mean = mean; p.rstd = rstd; p.B = B; p.res = res; p.shift = shift; p.eps = eps;
  p.rng = rng_state; p.site = site; p.p_drop = p_drop; p.part = reinterpret_cast<float*>(ws); p.cnt = split_counters(ws, (long long)B * res * res, C);
  if (dtype == STJ_BF16) return attn_dispatch<bf16>(C, p, stream);
  if (dtype == STJ_F16) return attn_dispatch<f16>(C, p, stream);
  if (dtype == STJ_F32) return attn_dispatch<float>(C, p, stream);
  stj_set_error("swin_attn: bad dtype %d", dtype);
  return STJ_EINVAL;
}

// =====================================================================================================================
// Fused attention half of SwinTransformerBlock, backward (tape.gradient of the forward kernel above):
//     dys = dp * dy ; da = dys Wproj^T ; (dq, dk, dv) = window-attention backward ; dLN = dqkv Wqkv^T ; dx = dy + LN'(dLN)
// plus the relative-position-bias-table and LN gamma / beta gradients.  One workgroup per window, a wave owns 16 of its tokens
// (as queries, and -- for dK / dV -- as keys).  What goes through LDS: the saved q|k|v tile of the current head group (the
// gradients dq|dk|dv overwrite it in place, head by head), per head the P and dS tiles (the sums over QUERIES of dV = P^T dO and
// dK = dS^T Q cross the waves) and a 64 x 32 dO tile; weights in K-chunks as MFMA A operands.  Everything else chains in registers
// exactly as in the forward kernel.  The weight gradients stay split-K GEMMs on what this kernel writes once: dqkv [M,3C] and
// dys [M,C] (next to the forward's saved a = attention output and ln = LN(x)).
// =====================================================================================================================
template <typename T, int C, int HGP = 0> struct AttnBCfg {
  static constexpr int KSTEP = Mma<T>::KSTEP;
  static constexpr int KS = C / KSTEP, NF = C / 16, HEADS = C / 32;
  static constexpr int HG = HGP ? HGP : (sizeof(T) == 2 ? (C == 96 ? STJ_ATTNB_HG96 : (C == 192 ? STJ_ATTNB_HG192 : 1)) : 1);    // heads per pass
  static constexpr int GC = 32 * HG;
  static constexpr int PADK = sizeof(T) == 2 ? 16 : 8;                 // pad of k-contiguous images read with 16-byte fragments
  static constexpr int LDT = 3 * GC + PADK;                            // q|k|v (then dq|dk|dv) tile [64][LDT]
  static constexpr int KW = 3 * GC;                                    // columns of the weight buffer
  static constexpr int LDW = KW + PADK;                                // weight image [C][LDW] (rows = c, k contiguous)
  static constexpr int KC1 = C <= KW ? C : 96;                         // K-chunk of the proj product
  static constexpr int LDP = 64 + (sizeof(T) == 2 ? 8 : 4);            // P / dS tiles [64][LDP]
  static constexpr int LDO = 32 + (sizeof(T) == 2 ? 8 : 4);            // dO tile of one head [64][LDO]
  // f32 at C = 384: the Wqkv slice of a head (rows = the C outputs of d LN(x)) is staged in KH = 2 halves of its ROWS (see AttnCfg::KH)
  static constexpr int KH = (sizeof(T) == 4 && C == 384) ? 2 : 1;
  static constexpr int TILE = 64 * LDT, WB = (C / KH) * LDW, PT = 64 * LDP, DO = 64 * LDO;
  static constexpr int LDS_BYTES = (TILE + WB + 2 * PT + DO) * (int)sizeof(T) + 225 * 4 + 2 * 64 * 4 + 2 * C * 4;
  static_assert(HEADS % HG == 0 && C % KC1 == 0, "shape");
};

struct AttnBArgs {
  const void* x; const void* dy; const void* qkv; const float* mean; const float* rstd; const float* gamma;
  const void* wqkv; const void* wproj; const float* table;
  void* dx; void* dqkv; void* dys; float* dtable; int tparts; float* dgamma; float* dbeta; int nparts; long long pstride;
  int B, res, shift;
  const long long* rng; int site; float p_drop;
  int split; float* part; int* cnt;   // SPLIT kernel: workgroup = (window, slice of the heads); partial d LN(x) [split][M][C] f32 in token order | slabs + arrival counters
};

// staging geometry of the backward kernel (chunks per thread).  (The first version copied each 16-byte piece load -> store in a loop:
// 20 dependent round trips per 80 KB chunk, cold; now all loads of a chunk are in flight at once, issued a phase ahead.)
template <typename T, int C, int HGP = 0> struct AttnBStage {
  typedef AttnBCfg<T, C, HGP> G;
  static constexpr int VN = Vec<T>::N, GC = G::GC;
  static constexpr int CPS = GC / VN, CP1 = G::KC1 / VN;
  static constexpr int CH = C / G::KH;                // rows of the Wqkv slice staged at a time
  static constexpr int NWQ = (CH * 3 * CPS + 255) / 256, NWP = (C * CP1 + 255) / 256;
  static constexpr int NT = (64 * 3 * CPS + 255) / 256;
};

template <typename T, int C, int NSPLIT = 1, int HGP = 0, bool FIX = false>
__global__ __launch_bounds__(256, STJ_ATTNB_MINB) void swin_attn_bwd_kernel(AttnBArgs p) {
  typedef AttnBCfg<T, C, HGP> G;
  constexpr int KS = G::KS, NF = G::NF, HG = G::HG, GC = G::GC, KSTEP = G::KSTEP, ND = Chain<T>::ND, LK = Mma<T>::LANE_K;
  constexpr int VN = Vec<T>::N;
  // NSPLIT > 1 (C = 384: 32 windows at B = 8 cannot fill the chip): workgroup = (window, slice sp of the heads); it needs da only for
  // its NH heads (their 32 NH rows of Wproj), and leaves its share of d LN(x) to swin_split_bwd_epi_kernel
  constexpr bool SPLIT = NSPLIT > 1;
  constexpr int NH = G::HEADS / NSPLIT;               // heads of this workgroup
  constexpr int DAF = SPLIT ? 2 * NH : NF;            // 16-column fragments of da it computes
  constexpr int PR = SPLIT ? 32 * NH : C;             // rows of Wproj it stages
  static_assert(G::HEADS % NSPLIT == 0 && NH % HG == 0, "head slices");
  const int unit = SPLIT ? (int)blockIdx.x / NSPLIT : (int)blockIdx.x, sp = SPLIT ? (int)blockIdx.x % NSPLIT : 0;
  const int hb0 = SPLIT ? sp * NH : 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char ab_smem[];
  T* tile = reinterpret_cast<T*>(ab_smem);
  T* Wb = tile + G::TILE;
  T* Pt = Wb + G::WB;
  T* dSt = Pt + G::PT;
  T* dOt = dSt + G::PT;
  float* tbl = reinterpret_cast<float*>(dOt + G::DO);               // [225] bias table of the current head, then its gradient bins
  int* tok = reinterpret_cast<int*>(tbl + 225);
  int* lab = tok + 64;
  float* red = reinterpret_cast<float*>(lab + 64);                  // [2][C] gamma / beta partial sums of the block

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g = lane >> 4, ln = lane & 15;
  const int nwx = p.res / 8, nW = nwx * nwx;
  const int win = unit % nW, b = unit / nW;
  const int wy = win / nwx, wx = win % nwx;
  const long long N = (long long)p.res * p.res;
  if (tid < 64) {
    const int t = tid;
    int ry = wy * 8 + (t >> 3), rx = wx * 8 + (t & 7);
    int sy = ry + p.shift; if (sy >= p.res) sy -= p.res;
    int sx = rx + p.shift; if (sx >= p.res) sx -= p.res;
    tok[t] = sy * p.res + sx;
    const int ly = ry < p.res - 8 ? 0 : (ry < p.res - p.shift ? 1 : 2);
    const int lx = rx < p.res - 8 ? 0 : (rx < p.res - p.shift ? 1 : 2);
    lab[t] = ly * 3 + lx;
  }
  STAMP(0); STAMP(1);
  for (int c = tid; c < 2 * C; c += 256) red[c] = 0.f;
  // bias tables of the workgroup's heads in LDS up front: a global load inside the head loop sits behind the prefetched weight group in the
  // in-order vmcnt queue and made every head wait for the whole prefetch
  // (loaded BEHIND the first weight chunk and the dy rows, below: one memory round trip for the three instead of two)
  __shared__ float tbv[NH * 225];
  __syncthreads();
  const long long myrow = (long long)b * N + tok[16 * wv + ln];
  const T* wq = reinterpret_cast<const T*>(p.wqkv);
  const T* wp = reinterpret_cast<const T*>(p.wproj);
  const float scale = 0.17677669529663687f;
  const T* qkvb = reinterpret_cast<const T*>(p.qkv) + (long long)b * N * 3 * C;
  T* dqkvb = reinterpret_cast<T*>(p.dqkv) + (long long)b * N * 3 * C;
  // staging registers: weight chunk / q|k|v tile, global -> registers (all loads of a chunk in flight at once, issued a phase ahead) -> LDS
  typedef AttnBStage<T, C, HGP> SG;
  constexpr int NWP = (PR * SG::CP1 + 255) / 256;
  uint4 s_wp[NWP], s_wq[SG::NWQ], s_t[SG::NT];
  auto issue_wp = [&](int k0) __attribute__((always_inline)) {                      // Wproj[rows of the workgroup's heads, k0 .. k0+KC1]
#pragma unroll
    for (int i = 0; i < NWP; ++i) {
      const int q = tid + i * 256;
      uint4 v = make_uint4(0, 0, 0, 0);        // unconditional store of a selected value: a conditional store keeps the array in scratch
      if (q < PR * SG::CP1) v = *reinterpret_cast<const uint4*>(wp + (long long)(32 * hb0 + q / SG::CP1) * C + k0 + (q % SG::CP1) * VN);
      s_wp[i] = v;
    }
  };
  auto commit_wp = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NWP; ++i) {
      const int q = tid + i * 256;
      if (q < PR * SG::CP1) *reinterpret_cast<uint4*>(Wb + (q / SG::CP1) * G::LDW + (q % SG::CP1) * VN) = s_wp[i];
    }
  };
  auto issue_group = [&](int h0) __attribute__((always_inline)) {                   // Wqkv[:, q|k|v columns of the head group] and the group's q|k|v tile
#pragma unroll
    for (int i = 0; i < SG::NWQ; ++i) {
      const int q = tid + i * 256;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (q < SG::CH * 3 * SG::CPS) {
        const int r0 = q / (3 * SG::CPS), r = q % (3 * SG::CPS);
        v = *reinterpret_cast<const uint4*>(wq + (long long)r0 * (3 * C) + (r / SG::CPS) * C + 32 * h0 + (r % SG::CPS) * VN);
      }
      s_wq[i] = v;
    }
#pragma unroll
    for (int i = 0; i < SG::NT; ++i) {
      const int q = tid + i * 256;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (q < 64 * 3 * SG::CPS) {
        const int tt = q / (3 * SG::CPS), r = q % (3 * SG::CPS);
        v = *reinterpret_cast<const uint4*>(qkvb + (long long)tok[tt] * 3 * C + (r / SG::CPS) * C + 32 * h0 + (r % SG::CPS) * VN);
      }
      s_t[i] = v;
    }
  };
  auto commit_group = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < SG::NWQ; ++i) {
      const int q = tid + i * 256;
      if (q < SG::CH * 3 * SG::CPS) { const int r0 = q / (3 * SG::CPS), r = q % (3 * SG::CPS); *reinterpret_cast<uint4*>(Wb + r0 * G::LDW + (r / SG::CPS) * GC + (r % SG::CPS) * VN) = s_wq[i]; }
    }
#pragma unroll
    for (int i = 0; i < SG::NT; ++i) {
      const int q = tid + i * 256;
      if (q < 64 * 3 * SG::CPS) { const int tt = q / (3 * SG::CPS), r = q % (3 * SG::CPS); *reinterpret_cast<uint4*>(tile + tt * G::LDT + (r / SG::CPS) * GC + (r % SG::CPS) * VN) = s_t[i]; }
    }
  };
  issue_wp(0);                                       // first Wproj chunk in flight under the dy loads

  // ---- rows of dy (this wave's 16 tokens) as B fragments, scaled by the DropPath factor of the sample; dys for the proj weight gradient
  typename Mma<T>::Frag dya[KS];
  {
    const T* pd = reinterpret_cast<const T*>(p.dy) + myrow * C + LK * g;
    const float dp = drop_path_scale(p.rng, p.site, b, p.p_drop);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      dya[ks] = Mma<T>::from_global(pd + ks * KSTEP);
      if (dp != 1.f) {
        float v[LK];
        frag_unpack<T>(dya[ks], v);
#pragma unroll
        for (int e = 0; e < LK; ++e) v[e] *= dp;
        dya[ks] = frag_pack<T>(v);
      }
      if (p.dys && sp == 0) *reinterpret_cast<typename Mma<T>::Frag*>(reinterpret_cast<T*>(p.dys) + myrow * C + ks * KSTEP + LK * g) = dya[ks];
    }
  }
  for (int q = tid; q < NH * 225; q += 256) tbv[q] = p.table[(q % 225) * G::HEADS + hb0 + q / 225];      // (readers are behind the proj loop's barriers)
  // ---- phase 1: da^T[c][tok] = sum_oc Wproj[c][oc] dys^T[oc][tok]  (A = Wproj rows, k = oc contiguous: the natural layout)
  f32x4 da[DAF];
#pragma unroll
  for (int f = 0; f < DAF; ++f) da[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k0 = 0; k0 < C; k0 += G::KC1) {
    __syncthreads();
    if (k0 > 0) issue_wp(k0);
    commit_wp();
    __syncthreads();
    if (k0 + G::KC1 >= C) issue_group(hb0);           // first head group: in flight under the proj MFMAs
#pragma unroll
    for (int kk = 0; kk < G::KC1 / KSTEP; ++kk)
#pragma unroll
      for (int f = 0; f < DAF; ++f)
        da[f] = Mma<T>::mma(Mma<T>::load(Wb, G::LDW, 16 * f, kk * KSTEP, lane), dya[k0 / KSTEP + kk], da[f]);
  }

  STAMP(2);
  f32x4 dln[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) dln[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int qi = 16 * wv + ln;
  const int mylab = lab[qi];
  TICK_DECL;
  // the LayerNorm backward's operands (x row, dy row, statistics): fetched under the LAST pass's dLN product -- loaded where they are used
  // the epilogue was two dependent memory round trips on an otherwise finished workgroup (17.6 of its 79 kcycles at C = 96: stamps)
  constexpr bool TAILPF = !SPLIT || FIX;       // (with slices meeting inside the launch: in front of the hand-over, by every slice)
  typename Raw4<T>::type xraw[TAILPF ? NF : 1], dyraw[TAILPF ? NF : 1];
  float mu_pf = 0.f, rs_pf = 0.f;

#pragma unroll
  for (int hl = 0; hl < NH; hl += HG) {            // unrolled: the da[] fragments of a head are picked by a compile-time index
    const int h0 = hb0 + hl;
    __syncthreads();                                  // previous pass done with the tile and the weight buffer
    // q|k|v of this head group -> tile (token-major, gathered); Wqkv[:, group columns] -> weight buffer (rows = c): fetched during
    // the previous phase, committed here; the next group's go in flight right away
    TICK(0);
    commit_group();
    __syncthreads();
    if (hl + HG < NH) issue_group(h0 + HG);
    TICK(1);
#pragma unroll
    for (int hh = 0; hh < HG; ++hh) {
      const int h = h0 + hh, hd = hl + hh;             // head; its index among the da[] fragments
      if (hh > 0) __syncthreads();                    // previous head's P / dS / dO / table readers are done
      // (reading tbv in place -- no copy, one barrier per head less: 1400 / 1398 / 1401 against 1399 / 1402 / 1399 scenes/s, cfg-512 780.8 / 780.2
      //  against 779.8 / 779.7: nothing, profiles/r06_zd_attnb_tbv.txt)
      for (int q = tid; q < 225; q += 256) tbl[q] = tbv[hd * 225 + q];
      {   // dO of this head (this wave's 16 queries) -> LDS, token-major
#pragma unroll
        for (int jd = 0; jd < 2; ++jd) {
          const f32x4 v4 = da[2 * hd + jd];
          const float v[4] = {v4[0], v4[1], v4[2], v4[3]};
          st4(dOt + qi * G::LDO + 16 * jd + 4 * g, v);
        }
      }
      __syncthreads();                                // table staged
      // S^T = K Q^T, P^T = softmax over keys (as in the forward kernel)
      f32x4 st[4], dpt[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        st[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k0 = 0; k0 < 32; k0 += KSTEP)
          st[j] = Mma<T>::mma(Mma<T>::load(tile + GC + 32 * hh, G::LDT, 16 * j, k0, lane), Mma<T>::load(tile + 32 * hh, G::LDT, 16 * wv, k0, lane), st[j]);
      }
      float m = -INFINITY;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = 16 * j + 4 * g + r;
          float v = st[j][r] * scale + tbl[((qi >> 3) - (key >> 3) + 7) * 15 + ((qi & 7) - (key & 7) + 7)];
          if (p.shift > 0 && lab[key] != mylab) v += -100.0f;
          st[j][r] = v;
          m = fmaxf(m, v);
        }
      m = fmaxf(m, __shfl_xor(m, 16, 64)); m = fmaxf(m, __shfl_xor(m, 32, 64));
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float e = __expf(st[j][r] - m); st[j][r] = e; sum += e; }
      sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);
      const float inv = 1.f / sum;
      // dP^T[key][q] = sum_d V[key][d] dO^T[d][q]   (B = dO^T chained from the da accumulators of this head)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        dpt[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 32 / KSTEP; ++kk)
          dpt[j] = Mma<T>::mma(Chain<T>::ldA(tile + 2 * GC + 32 * hh, G::LDT, 16 * j, kk * KSTEP, lane), Chain<T>::from_acc(&da[2 * hd + kk * ND]), dpt[j]);
      }
      float dsum = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) { st[j][r] *= inv; dsum += st[j][r] * dpt[j][r]; }
      dsum += __shfl_xor(dsum, 16, 64); dsum += __shfl_xor(dsum, 32, 64);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int r = 0; r < 4; ++r) dpt[j][r] = st[j][r] * (dpt[j][r] - dsum);         // dS^T
        const float pv[4] = {st[j][0], st[j][1], st[j][2], st[j][3]}, dv_[4] = {dpt[j][0], dpt[j][1], dpt[j][2], dpt[j][3]};
        st4(Pt + qi * G::LDP + 16 * j + 4 * g, pv);                                    // P[q][key], dS[q][key]: row q, 4 consecutive keys
        st4(dSt + qi * G::LDP + 16 * j + 4 * g, dv_);
      }
      // dQ^T[d][q] = scale sum_key K^T[d][key] dS^T[key][q]   (A = K^T by transposed reads of the token-major tile; B chained)
      f32x4 dq[2];
#pragma unroll
      for (int jd = 0; jd < 2; ++jd) {
        dq[jd] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 64 / KSTEP; ++s)
          dq[jd] = Mma<T>::mma(Chain<T>::ldA_tr(tile + GC + 32 * hh, G::LDT, 16 * jd, s * KSTEP, lane), Chain<T>::from_acc(&dpt[s * ND]), dq[jd]);
      }
      __syncthreads();                                // P, dS, dO of all 64 queries are in LDS
      // the sums over queries: this wave's 16 KEYS.  dV^T[d][key] = sum_q dO^T[d][q] P[q][key];  dK^T[d][key] = scale sum_q Q^T[d][q] dS[q][key]
      f32x4 dv[2], dk[2];
#pragma unroll
      for (int jd = 0; jd < 2; ++jd) {
        dv[jd] = (f32x4){0.f, 0.f, 0.f, 0.f}; dk[jd] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k0 = 0; k0 < 64; k0 += KSTEP) {
          dv[jd] = Mma<T>::mma(Mma<T>::load_tr(dOt, G::LDO, 16 * jd, k0, lane), Mma<T>::load_tr(Pt, G::LDP, 16 * wv, k0, lane), dv[jd]);
          dk[jd] = Mma<T>::mma(Mma<T>::load_tr(tile + 32 * hh, G::LDT, 16 * jd, k0, lane), Mma<T>::load_tr(dSt, G::LDP, 16 * wv, k0, lane), dk[jd]);
        }
      }
      // relative-position-bias gradient of this head (one wave): bin (dy, dx) collects dS[(ry,rx)][(ry-dy, rx-dx)]; lane l owns the
      // 8 x 8 block (query row l / 8, key row l % 8) of the dS tile, whose 15 diagonals are the dx bins of ONE dy
      float dacc[15];
      const bool tw = wv == (h & 3);
      if (tw) {
        const int ry = lane >> 3, ky = lane & 7;
#pragma unroll
        for (int d = 0; d < 15; ++d) dacc[d] = 0.f;
#pragma unroll
        for (int rx = 0; rx < 8; ++rx) {
          const T* src = dSt + (ry * 8 + rx) * G::LDP + ky * 8;
          float v[8];
          if constexpr (VN == 8) ld16(src, v);
          else { ld16(src, v); ld16(src + 4, v + 4); }
#pragma unroll
          for (int kx = 0; kx < 8; ++kx) dacc[rx - kx + 7] += v[kx];
        }
      }
      __syncthreads();                                // every wave is done with q_h, k_h, v_h, P, dS, dO and the bias table
      {   // dq | dk | dv of this head overwrite q | k | v in the tile (rows = this wave's tokens), 4 consecutive d per lane
#pragma unroll
        for (int jd = 0; jd < 2; ++jd) {
          const float a0[4] = {dq[jd][0] * scale, dq[jd][1] * scale, dq[jd][2] * scale, dq[jd][3] * scale};
          const float a1[4] = {dk[jd][0] * scale, dk[jd][1] * scale, dk[jd][2] * scale, dk[jd][3] * scale};
          const float a2[4] = {dv[jd][0], dv[jd][1], dv[jd][2], dv[jd][3]};
          T* row = tile + qi * G::LDT + 32 * hh + 16 * jd + 4 * g;
          st4(row, a0); st4(row + GC, a1); st4(row + 2 * GC, a2);
        }
      }
      for (int q = tid; q < 225; q += 256) tbl[q] = 0.f;
      __syncthreads();
      if (tw) {
        const int ry = lane >> 3, ky = lane & 7;
#pragma unroll
        for (int d = 0; d < 15; ++d) atomicAdd(&tbl[(ry - ky + 7) * 15 + d], dacc[d]);
      }
      __syncthreads();
      for (int bin = tid; bin < 225; bin += 256)
        atomicAdd(p.dtable + (long long)(unit % p.tparts) * 225 * G::HEADS + bin * G::HEADS + h, tbl[bin]);
    }
    TICK(2);
    __syncthreads();                                  // dq | dk | dv of the whole group are in the tile
    if constexpr (TAILPF && !FIX) {
      if (hl + HG >= NH) {
        const T* xr = reinterpret_cast<const T*>(p.x) + myrow * C + 4 * g;
        const T* dyr = reinterpret_cast<const T*>(p.dy) + myrow * C + 4 * g;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          xraw[f] = *reinterpret_cast<const typename Raw4<T>::type*>(xr + 16 * f);
          dyraw[f] = *reinterpret_cast<const typename Raw4<T>::type*>(dyr + 16 * f);
        }
        mu_pf = p.mean[myrow]; rs_pf = p.rstd[myrow];
      }
    }
    {   // copy-out for the qkv weight gradient (rows in original token order), then dLN^T += Wqkv[:, group] dqkv_group^T
      constexpr int CPS = GC / VN;
      for (int q = tid; q < 64 * 3 * CPS; q += 256) {
        const int t = q / (3 * CPS), r = q % (3 * CPS), seg = r / CPS, c = (r % CPS) * VN;
        *reinterpret_cast<uint4*>(dqkvb + (long long)tok[t] * 3 * C + seg * C + 32 * h0 + c) = *reinterpret_cast<const uint4*>(tile + t * G::LDT + seg * GC + c);
      }
    }
    if constexpr (G::KH == 1) {
#pragma unroll 1
    for (int kk = 0; kk < G::KW / KSTEP; ++kk) {
      const typename Mma<T>::Frag bf = Mma<T>::load(tile, G::LDT, 16 * wv, kk * KSTEP, lane);
#pragma unroll
      for (int f = 0; f < NF; ++f) dln[f] = Mma<T>::mma(Mma<T>::load(Wb, G::LDW, 16 * f, kk * KSTEP, lane), bf, dln[f]);
    }
    } else {
      // the weight buffer holds rows [0, C / 2) of the slice (commit_group); rows [C / 2, C) replace them for the second half of d LN(x)
      static_assert(G::KH == 2 && NF % 2 == 0, "two halves");
#pragma unroll 1
      for (int kk = 0; kk < G::KW / KSTEP; ++kk) {
        const typename Mma<T>::Frag bf = Mma<T>::load(tile, G::LDT, 16 * wv, kk * KSTEP, lane);
#pragma unroll
        for (int f = 0; f < NF / 2; ++f) dln[f] = Mma<T>::mma(Mma<T>::load(Wb, G::LDW, 16 * f, kk * KSTEP, lane), bf, dln[f]);
      }
      __syncthreads();
      for (int q = tid; q < SG::CH * 3 * SG::CPS; q += 256) {
        const int r0 = q / (3 * SG::CPS), r = q % (3 * SG::CPS);
        *reinterpret_cast<uint4*>(Wb + r0 * G::LDW + (r / SG::CPS) * GC + (r % SG::CPS) * VN) =
            *reinterpret_cast<const uint4*>(wq + (long long)(SG::CH + r0) * (3 * C) + (r / SG::CPS) * C + 32 * h0 + (r % SG::CPS) * VN);
      }
      __syncthreads();
#pragma unroll 1
      for (int kk = 0; kk < G::KW / KSTEP; ++kk) {
        const typename Mma<T>::Frag bf = Mma<T>::load(tile, G::LDT, 16 * wv, kk * KSTEP, lane);
#pragma unroll
        for (int f = 0; f < NF / 2; ++f) dln[NF / 2 + f] = Mma<T>::mma(Mma<T>::load(Wb, G::LDW, 16 * f, kk * KSTEP, lane), bf, dln[NF / 2 + f]);
      }
    }
    TICK(3);
  }
  TICK_STORE;
  STAMP(3);

  if constexpr (SPLIT && !FIX) {  // this head slice's share of d LN(x)
    float* pr = p.part + ((long long)sp * p.B * N + myrow) * C + 4 * g;
#pragma unroll
    for (int f = 0; f < NF; ++f) *reinterpret_cast<f32x4*>(pr + 16 * f) = dln[f];
    return;
  }
  if constexpr (SPLIT && FIX) {   // the head slices' shares meet in the workgroup that finishes last (slice_combine); it alone goes on
    __shared__ int ticket;
    {
      const T* xr = reinterpret_cast<const T*>(p.x) + myrow * C + 4 * g;
      const T* dyr = reinterpret_cast<const T*>(p.dy) + myrow * C + 4 * g;
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        xraw[f] = *reinterpret_cast<const typename Raw4<T>::type*>(xr + 16 * f);
        dyraw[f] = *reinterpret_cast<const typename Raw4<T>::type*>(dyr + 16 * f);
      }
      mu_pf = p.mean[myrow]; rs_pf = p.rstd[myrow];
    }
    if (!slice_combine<NF>(dln, p.part, p.cnt, unit, sp, NSPLIT, tid, &ticket)) return;
  }
  // ---- LayerNorm backward on the accumulator layout + the shortcut gradient; gamma / beta partial sums
  {
    const float mu = TAILPF ? mu_pf : p.mean[myrow], rs = TAILPF ? rs_pf : p.rstd[myrow];
    const T* xr = reinterpret_cast<const T*>(p.x) + myrow * C;
    const T* dyr = reinterpret_cast<const T*>(p.dy) + myrow * C;
    T* dxr = reinterpret_cast<T*>(p.dx) + myrow * C;
    float xh[NF][4];
    float dyv[NF][4];             // read up front: dx may alias dy, so loads left in the store loop below are serialised behind each store
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      if constexpr (TAILPF) raw4_unpack<T>(dyraw[f], dyv[f]);
      else ld4(dyr + 16 * f + 4 * g, dyv[f]);
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int col = 16 * f + 4 * g;
      float xv[4];
      if constexpr (TAILPF) raw4_unpack<T>(xraw[f], xv);
      else ld4(xr + col, xv);
      const float4 gm = *reinterpret_cast<const float4*>(p.gamma + col);
      const float gmv[4] = {gm.x, gm.y, gm.z, gm.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        xh[f][r] = (xv[r] - mu) * rs;
        const float d = dln[f][r];
        const float a = row16_sum(d * xh[f][r]), bsum = row16_sum(d);         // over the wave's 16 rows (DPP, no LDS crossbar)
        if (ln == 0) { atomicAdd(&red[col + r], a); atomicAdd(&red[C + col + r], bsum); }
        const float t = d * gmv[r];
        dln[f][r] = t;
        s1 += t; s2 += t * xh[f][r];
      }
    }
    s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
    s1 *= (1.f / C); s2 *= (1.f / C);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int col = 16 * f + 4 * g;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = dyv[f][r] + rs * (dln[f][r] - s1 - xh[f][r] * s2);
      st4(dxr + col, v);
    }
  }
  __syncthreads();
  const long long po = (long long)(blockIdx.x % p.nparts) * p.pstride;
  for (int c = tid; c < C; c += 256) { atomicAdd(p.dgamma + po + c, red[c]); atomicAdd(p.dbeta + po + c, red[C + c]); }
  STAMP(5);
}

template <typename T, int C, int NSPLIT = 1, int HGP = 0, bool FIX = false>
static int attnb_launch(const AttnBArgs& a, hipStream_t st) {
  typedef AttnBCfg<T, C, HGP> G;
  static PerDevice<bool> attr;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)swin_attn_bwd_kernel<T, C, NSPLIT, HGP, FIX>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES) != hipSuccess) {
      stj_set_error("swin_attn_bwd: cannot reserve %d bytes of LDS", G::LDS_BYTES); return STJ_ELAUNCH;
    }
    attr = true;
  }
  const int nW = (a.res / 8) * (a.res / 8);
  hipLaunchKernelGGL((swin_attn_bwd_kernel<T, C, NSPLIT, HGP, FIX>), dim3((unsigned)(a.B * nW * NSPLIT)), dim3(256), G::LDS_BYTES, st, a);
  return stj_check_launch("stj_swin_attn_bwd");
}
template <typename T>
static int attnb_split384(const AttnBArgs& a, hipStream_t st) {          // see attn_split384
  if constexpr (sizeof(T) == 2) {
    if (a.part == nullptr) { stj_set_error("swin_attn_bwd: C = 384 needs the workspace (stj_swin_split_workspace_bytes)"); return STJ_EINVAL; }
    const long long N = (long long)a.res * a.res;
    const int split = attn_split_for(a.B * (N / 64));
    if (split == 2) {
      if (a.B * (N / 64) > FIX_CNT_BYTES / 4) { stj_set_error("swin_attn_bwd: more than %d windows in one launch", FIX_CNT_BYTES / 4); return STJ_EUNSUPPORTED; }
      return attnb_launch<T, 384, 2, 0, true>(a, st);
    }
    const int rc = attnb_launch<T, 384, 6>(a, st);
    if (rc != STJ_OK) return rc;
    return split_bwd_epi<T>(a.x, a.dy, a.part, split, a.gamma, 0.f, a.mean, a.rstd, a.dx, a.dgamma, a.dbeta, a.nparts, a.pstride, a.B * N, st);
  } else {
    // f32 (parity mode): six slices of two heads, the Wqkv slice of a head staged in two halves of its rows (AttnBCfg::KH), partial sums +
    // the finishing launch
    if (a.part == nullptr) { stj_set_error("swin_attn_bwd: C = 384 needs the workspace (stj_swin_split_workspace_bytes)"); return STJ_EINVAL; }
    const long long N = (long long)a.res * a.res;
    const int rc = attnb_launch<T, 384, 6>(a, st);
    if (rc != STJ_OK) return rc;
    return split_bwd_epi<T>(a.x, a.dy, a.part, 6, a.gamma, 0.f, a.mean, a.rstd, a.dx, a.dgamma, a.dbeta, a.nparts, a.pstride, a.B * N, st);
  }
}

template <typename T>
static int attnb_192(const AttnBArgs& a, hipStream_t st) {              // see attn_dispatch
  if constexpr (sizeof(T) == 2) {
    if (a.part != nullptr && (long long)a.B * (a.res / 8) * (a.res / 8) < 256) return attnb_launch<T, 192, 2, 1, true>(a, st);
  }
  return attnb_launch<T, 192>(a, st);
}

extern "C" int stj_swin_attn_bwd(const void* x, const void* dy, const void* qkv, const float* mean, const float* rstd, const float* gamma,
                                 const void* wqkv, const void* wproj, const float* table, void* dx, void* dqkv, void* dys,
                                 float* dtable, int tparts, float* dgamma, float* dbeta, int nparts, long long part_stride,
                                 int B, int res, int C, int shift, const long long* rng_state, int site, float p_drop, int dtype,
                                 void* ws, hipStream_t stream) {
  if (B <= 0) return STJ_OK;
  if (res % 8 != 0 || shift < 0 || shift >= 8) { stj_set_error("swin_attn_bwd: res %% 8 != 0 or bad shift"); return STJ_EINVAL; }
  if (tparts < 1 || nparts < 1) { stj_set_error("swin_attn_bwd: tparts / nparts must be >= 1"); return STJ_EINVAL; }
  if (!(p_drop >= 0.f && p_drop < 1.f)) { stj_set_error("swin_attn_bwd: need 0 <= p_drop < 1"); return STJ_EINVAL; }
  AttnBArgs p = {};
  p.x = x; p.dy = dy; p.qkv = qkv; p.mean = mean; p.rstd = rstd; p.gamma = gamma; p.wqkv = wqkv; p.wproj = wproj; p.table = table;
  p.dx = dx; p.dqkv = dqkv; p.dys = dys; p.dtable = dtable; p.tparts = tparts; p.dgamma = dgamma; p.dbeta = dbeta; p.nparts = nparts;
  p.pstride = part_stride; p.B = B; p.res = res; p.shift = shift; p.rng = rng_state; p.site = site; p.p_drop = p_drop;
  p.part = reinterpret_cast<float*>(ws); p.cnt = split_counters(ws, (long long)B * res * res, C);
#define STJ_AB(TT) (C == 96 ? attnb_launch<TT, 96>(p, stream) : (C == 192 ? attnb_192<TT>(p, stream) : (C == 384 ? attnb_split384<TT>(p, stream) : (stj_set_error("swin_attn_bwd: C must be 96, 192 or 384 (got %d)", C), (int)STJ_EUNSUPPORTED))))
  if (dtype == STJ_BF16) return STJ_AB(bf16);
  if (dtype == STJ_F16) return STJ_AB(f16);
  if (dtype == STJ_F32) return STJ_AB(float);
#undef STJ_AB
  stj_set_error("swin_attn_bwd: bad dtype %d", dtype);
  return STJ_EINVAL;
}
