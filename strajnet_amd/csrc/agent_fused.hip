// Fused agent branch: trajNet's TrajEncoder for every agent of a batch as ONE launch per direction, and the 64-agent
// interaction block (Cross_Attention + FFN + the segment LayerNorms) as ONE launch per direction.
//   reference trajNet.py:29-48  TrajEncoder: Conv1D(5 -> 64, k 1) + ELU; tfa MultiHeadAttention(head_size 64, 4 heads, output 320, dropout .1)
//                               over the 11 time steps with the mask (x != 0) (x) (x != 0); GlobalMaxPooling1D; Dense(3 -> 64, no bias) on the
//                               step-0 type one-hot; concat -> Dense(384 -> 384) + ELU
//   reference trajNet.py:65-87  Cross_Attention: tfa MHA(head_size 64, 6 heads, output 384, dropout .1) -> LayerNorm(1e-3) -> Dense(1536, elu)
//                               -> Dropout -> Dense(384) -> Dropout -> LayerNorm(1e-3)
//   reference trajNet.py:125-187 TrajNet.call: masks, segment embedding, masked concat, the cross attention, residuals, obs_norm | occ_norm
// Layer by layer this branch was 24 launches forward and ~45 backward of 4-15 us each, on a chain the cross-attention (forward) and the
// FG-MSA backward (a replayed hipGraph serialises the two chains) wait for.
//
// Work layout.  FLOPs are irrelevant here (3.5 GFLOP per 8 scenes); what the kernels are built around is the weight stream: 0.56 MB
// (encoder) / 3.5 MB (interaction block) of 16-bit weights per workgroup.  Activations live in LDS tiles [token][channel]; every Dense layer
// is computed transposed, D[m = output column][n = token] = sum_k W^T[m][k] X[n][k], with the WEIGHT fragment (MFMA A operand) loaded
// straight from global memory / L2 into registers -- one 16-byte load per lane and k-step, the next column tile's fragments in flight
// under the current tile's MFMAs -- and the activation fragment (B operand) read from the LDS tile.  The waves of a workgroup split the
// output columns, so a weight byte is read once per workgroup.  That needs the weights K-contiguous per output column: stj_agent_pack
// writes the transposed copies [N][K] once per step (the forward's layout); the backward's input-gradient products contract over the
// OUTPUT columns and read the natural Keras layouts [K][N] in place.  Weight gradients are not computed here: the backward kernels write
// each layer's dY once and the caller queues dW += X^T dY on the grouped stream-K launch (csrc/wgrad_sk.hip); only the tiny ones
// (5 x 64, 3 x 64, 2 x 384, LayerNorm gamma / beta) are accumulated in place with atomics.
#include "common.h"
#include "rng.h"
#include "chain48.h"

namespace agf {
using namespace chain;

constexpr int TN = 11, TP = 16;                 // time steps of a track, padded to one MFMA tile
constexpr int NF = 64;                          // node features
constexpr int EH = 4, ED = 64, EHD = EH * ED;   // TrajEncoder attention: heads, head size
constexpr int EO = 320;                         // its output width
constexpr int CB = 384;                         // agent embedding width
constexpr int IH = 6, IDH = 64;                 // interaction attention: heads, head size (6 x 64 = 384)
constexpr int FF = 1536;                        // FFN hidden width
constexpr int NA = 64;                          // agents per scene
// transposed weight copies ([N][K], K contiguous) in the pack, element offsets
constexpr long long P_EQKV = 0;                                // [3 x 256][64]   node_attention query | key | value kernels, row = (m, h, o)
constexpr long long P_EWO = P_EQKV + 3LL * EHD * NF;           // [320][256]      node_attention projection
constexpr long long P_EWS = P_EWO + (long long)EO * EHD;       // [384][384]      sublayer
constexpr long long P_IQKV = P_EWS + (long long)CB * CB;       // [3 x 384][384]  cross_attention/mha query | key | value
constexpr long long P_IWO = P_IQKV + 3LL * CB * CB;            // [384][384]      cross_attention/mha projection
constexpr long long P_IW1 = P_IWO + (long long)CB * CB;        // [1536][384]     FFN1
constexpr long long P_IW2 = P_IW1 + (long long)FF * CB;        // [384][1536]     FFN2
constexpr long long P_TOTAL = P_IW2 + (long long)CB * FF;

template <typename T> __device__ __forceinline__ float exp_t(float x) {
  if constexpr (sizeof(T) == 4) return expf(x); else return __expf(x);
}
template <typename T> __device__ __forceinline__ float elu_t(float x) {
  if constexpr (sizeof(T) == 4) return elu_f(x); else return elu_bf(x);
}
template <typename T> __device__ __forceinline__ float rnd(float x) { T t; stf(&t, x); return ldf(&t); }      // value as the storage type holds it
__device__ __forceinline__ bool keep1(const long long* rng, int site, long long idx, float p) {
  bool k[4];
  keep4(rng, site, idx >> 2, p, k);
  return k[idx & 3];
}
__device__ __forceinline__ void keep_scale4(const long long* rng, int site, long long idx, float p, float sc, float (&f)[4]) {   // idx % 4 == 0
  bool k[4];
  keep4(rng, site, idx >> 2, p, k);
#pragma unroll
  for (int e = 0; e < 4; ++e) f[e] = k[e] ? sc : 0.f;
}
template <typename T> __device__ __forceinline__ void st_acc(T* p, const f32x4& v) { const float t[4] = {v[0], v[1], v[2], v[3]}; st4(p, t); }

// ---- the Dense building block --------------------------------------------------------------------------------------------------------
// The calling wave computes, for its column tiles ct = w0, w0 + nw, ... < nct,
//     acc[mt][r] = sum_{k < K} Wrow(16 ct + 4 g + r)[k] * X[16 mt + (lane & 15)][xk0 + k]          (g = lane >> 4)
// wp(row, s) -> pointer to the KSTEP weights of k-step s of weight row `row` (global memory, 16-byte aligned); X: LDS tile, row stride ldx.
// K is walked in chunks of KC (the weight fragments of a chunk live in registers; the next unit's are in flight while this one multiplies).
template <typename T, int MT, int K, int KC, typename WP, typename EP>
__device__ __forceinline__ void gemm_cols(WP wp, const T* X, int ldx, int xk0, int nct, int w0, int nw, int lane, EP ep) {
  typedef Mma<T> M;
  constexpr int KS = KC / M::KSTEP, NKC = K / KC;
  static_assert(K % KC == 0 && KC % M::KSTEP == 0, "chunking");
  if (w0 >= nct) return;
  const int ln = lane & 15, g = lane >> 4;
  const int nu = ((nct - w0 + nw - 1) / nw) * NKC;
  typename M::Frag a[KS], an[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) a[s] = M::from_global(wp(w0 * 16 + ln, s) + M::LANE_K * g);
  f32x4 acc[MT];
#pragma unroll 1
  for (int u = 0; u < nu; ++u) {
    const int un = u + 1 < nu ? u + 1 : u;                    // (past the end: the last unit again -- staging arrays are written unconditionally)
    const int ctn = w0 + (un / NKC) * nw, kcn = un % NKC;
#pragma unroll
    for (int s = 0; s < KS; ++s) an[s] = M::from_global(wp(ctn * 16 + ln, kcn * KS + s) + M::LANE_K * g);
    const int kc = u % NKC;
    if (kc == 0) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // the activation fragments one k-step ahead, and no further (left alone the scheduler hoists every LDS read of the unit: 192 registers)
    typename M::Frag bf[MT], bn[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) bf[mt] = M::load(X, ldx, mt * 16, xk0 + kc * KS * M::KSTEP, lane);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int sn = s + 1 < KS ? s + 1 : s;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) bn[mt] = M::load(X, ldx, mt * 16, xk0 + (kc * KS + sn) * M::KSTEP, lane);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt] = M::mma(a[s], bf[mt], acc[mt]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) bf[mt] = bn[mt];
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kc == NKC - 1) ep(w0 + (u / NKC) * nw, acc);
#pragma unroll
    for (int s = 0; s < KS; ++s) a[s] = an[s];
  }
}
// the same for a FIXED set of NCT column tiles (w0 + j nw) whose accumulators the caller keeps across several calls (K chunk by K chunk)
template <typename T, int MT, int KC, int NCT, typename WP>
__device__ __forceinline__ void gemm_cols_acc(WP wp, const T* X, int ldx, int xk0, int w0, int nw, int lane, f32x4 (&acc)[NCT][MT]) {
  typedef Mma<T> M;
  constexpr int KS = KC / M::KSTEP;
  const int ln = lane & 15, g = lane >> 4;
  typename M::Frag a[KS], an[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) a[s] = M::from_global(wp(w0 * 16 + ln, s) + M::LANE_K * g);
#pragma unroll
  for (int j = 0; j < NCT; ++j) {
    const int jn = j + 1 < NCT ? j + 1 : j;
#pragma unroll
    for (int s = 0; s < KS; ++s) an[s] = M::from_global(wp((w0 + jn * nw) * 16 + ln, s) + M::LANE_K * g);
    typename M::Frag bf[MT], bn[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) bf[mt] = M::load(X, ldx, mt * 16, xk0, lane);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int sn = s + 1 < KS ? s + 1 : s;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) bn[mt] = M::load(X, ldx, mt * 16, xk0 + sn * M::KSTEP, lane);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[j][mt] = M::mma(a[s], bf[mt], acc[j][mt]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) bf[mt] = bn[mt];
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) a[s] = an[s];
  }
}

// cooperative 16-byte copies between an LDS tile and global rows
template <typename T>
__device__ __forceinline__ void rows_to_global(const T* tile, int ld, T* dst, long long dld, int rows, int cols, int tid, int nthr) {
  constexpr int V = Vec<T>::N;
  const int vpr = cols / V;
  for (int i = tid; i < rows * vpr; i += nthr) {
    const int r = i / vpr, c = (i % vpr) * V;
    *reinterpret_cast<uint4*>(dst + (long long)r * dld + c) = *reinterpret_cast<const uint4*>(tile + r * ld + c);
  }
}

// =====================================================================================================================================
// weight pack: transposed copies in the activation dtype
// =====================================================================================================================================
struct PackJob { const float* src; long long dst; int K, N, hs; };       // hs > 0: src is a tfa kernel [H][K][hs] (N = H hs, row n = (h, o)); 0: [K][N]
constexpr int NPACK = 11;
struct PackArgs { PackJob j[NPACK]; void* out; };
template <typename T>
__global__ __launch_bounds__(256) void agent_pack_kernel(PackArgs p) {
  PackJob jb = p.j[0];
#pragma unroll
  for (int i = 1; i < NPACK; ++i)
    if ((int)blockIdx.y == i) jb = p.j[i];
  const long long tot = (long long)jb.K * jb.N;
  T* dst = reinterpret_cast<T*>(p.out) + jb.dst;
  for (long long e = blockIdx.x * 256ll + threadIdx.x; e < tot; e += gridDim.x * 256ll) {
    const int n = (int)(e / jb.K), k = (int)(e % jb.K);
    const float v = jb.hs ? jb.src[((long long)(n / jb.hs) * jb.K + k) * jb.hs + n % jb.hs] : jb.src[(long long)k * jb.N + n];
    stf(dst + e, v);
  }
}

// =====================================================================================================================================
// TrajEncoder
// =====================================================================================================================================
template <typename T> struct EG {
  static constexpr int AG = sizeof(T) == 2 ? 2 : 1;       // agents per workgroup (LDS: the q|k|v tile is 784 elements wide)
  static constexpr int R = AG * TP;                        // token rows of a tile: 16 per agent, 11 real
  static constexpr int PAD = LdsPad<T>::P;
  static constexpr int LDN = NF + PAD, LDQ = 3 * EHD + PAD, LDA = EHD + PAD, LDC = CB + PAD, LDO = EO + PAD;
  static constexpr int KP = (Mma<T>::KSTEP > 16 ? Mma<T>::KSTEP : 16) + PAD;       // per-wave 16 x 16 tiles, padded to one k-step
  static constexpr int KCW = sizeof(T) == 2 ? 384 : 192;   // weight k-chunk held in registers (12 fragments)
};

struct EncArgs {
  const float* obs; const float* occ; int n_obs, n_occ, B;
  const void* pack;
  const float* wn; const float* bn; const float* wv3; const float* bo; const float* bs;      // f32 masters: Conv1D kernel [5][64] + bias, vector_feature [3][64], biases
  void* enc; int* cmi;
  void* s_nodes; void* s_qkv; void* s_att; unsigned short* s_pmask; void* s_cat;                // training: what backward reads (NULL: not written)
  const long long* rng; int site; float p_drop;
  // backward
  const void* d_enc; int d_enc_f32; const void* wq; const void* wk; const void* wv; const void* wo; const void* ws;      // natural-layout weights, activation dtype
  void* dpre_s; void* dout; void* dqkv;                                                            // dY of the sublayer / projection / q|k|v layers
  float* dwn; float* dbn; float* dwv3;
};

// the tile's agents: global agent index a0 + ag; track pointer of agent i = (b, a)
__device__ __forceinline__ const float* track(const EncArgs& p, int i) {
  const int A = p.n_obs + p.n_occ, b = i / A, a = i % A;
  return a < p.n_obs ? p.obs + ((long long)b * p.n_obs + a) * TN * 8 : p.occ + ((long long)b * p.n_occ + (a - p.n_obs)) * TN * 8;
}

// S^T = K Q^T of one (agent, head) from the q|k|v tile, softmax over the 11 keys with the tfa mask, dropout factors.
// Lane (i = lane & 15: query, j = 4 g + r: key).  pr: probabilities as the storage type holds them; f: keep / (1 - p) (1 without dropout).
template <typename T>
__device__ __forceinline__ void enc_probs(const T* QKV, int ldq, int row0, int h, const int* vt, long long agent, const long long* rng, int site,
                                          float p_drop, int lane, float (&pr)[4], float (&f)[4]) {
  typedef Mma<T> M;
  const int ln = lane & 15, g = lane >> 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < ED / M::KSTEP; ++ks)
    s = M::mma(M::load(QKV, ldq, row0, EHD + h * ED + ks * M::KSTEP, lane), M::load(QKV, ldq, row0, h * ED + ks * M::KSTEP, lane), s);
  const bool qv = vt[row0 + ln] != 0;
  float x[4], m = -INFINITY;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int j = 4 * g + r;
    float v = s[r] * 0.125f;                                   // tfa: query /= sqrt(head_size = 64)
    if (!(qv && vt[row0 + j] != 0)) v = v + (-10e9f);          // f32 add, as the reference (logits += -10e9 (1 - mask))
    x[r] = j < TN ? v : -INFINITY;
    m = fmaxf(m, x[r]);
  }
  m = fmaxf(m, __shfl_xor(m, 16, 64)); m = fmaxf(m, __shfl_xor(m, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) { x[r] = 4 * g + r < TN ? exp_t<T>(x[r] - m) : 0.f; sum += x[r]; }
  sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.f / sum;
  const bool drop = rng != nullptr && p_drop > 0.f;
  const float dsc = drop ? 1.f / (1.f - p_drop) : 1.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int j = 4 * g + r;
    pr[r] = rnd<T>(x[r] * inv);
    f[r] = 1.f;
    if (drop && ln < TN && j < TN) f[r] = keep1(rng, site, ((agent * EH + h) * TN + ln) * TN + j, p_drop) ? dsc : 0.f;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void agent_enc_fwd_kernel(EncArgs p) {
  typedef EG<T> E;
  typedef Mma<T> M;
  constexpr int AG = E::AG, R = E::R;
  extern __shared__ __attribute__((aligned(16))) unsigned char ag_smem[];
  float* x8 = reinterpret_cast<float*>(ag_smem);                  // [R][8] raw track rows (padded steps zero)
  int* vt = reinterpret_cast<int*>(x8 + R * 8);                   // [R] step valid
  T* Xn = reinterpret_cast<T*>(vt + R);                           // [R][LDN] nodes
  T* QKV = Xn + R * E::LDN;                                       // [R + 16][LDQ] (16 spare rows: transposed reads of a 32-deep k-step run past an agent's 16)
  T* ATT = QKV + (R + 16) * E::LDQ;                               // [R][LDA]
  T* CAT = ATT + R * E::LDA;                                      // [16][LDC] pooled | vector, one row per agent
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, ln = lane & 15, g = lane >> 4;
  const int a0 = blockIdx.x * AG;                                 // first agent (global index b * A + a) of this tile
  const T* pk = reinterpret_cast<const T*>(p.pack);
  const bool train = p.s_qkv != nullptr;

  for (int i = tid; i < R * 8; i += 256) {               // (values as the storage type holds them: the layer-by-layer path's stj_agent_prep casts the tracks)
    const int row = i >> 3, ag = row / TP, t = row % TP;
    x8[i] = t < TN ? rnd<T>(track(p, a0 + ag)[t * 8 + (i & 7)]) : 0.f;
  }
  for (int i = tid; i < 16 * E::LDQ; i += 256) stf(QKV + R * E::LDQ + i, 0.f);
  for (int i = tid; i < 16 * E::LDC; i += 256) stf(CAT + i, 0.f);
  __syncthreads();
  if (tid < R) vt[tid] = (tid % TP) < TN && x8[tid * 8] != 0.f;
  // nodes = ELU(x[:, :5] Wn + bn)   (Conv1D kernel size 1)
  for (int i = tid; i < R * NF; i += 256) {
    const int row = i / NF, c = i % NF;
    float v = 0.f;
    if (row % TP < TN) {
      v = p.bn[c];
#pragma unroll
      for (int k = 0; k < 5; ++k) v += x8[row * 8 + k] * rnd<T>(p.wn[k * NF + c]);
      v = elu_t<T>(v);
    }
    stf(Xn + row * E::LDN + c, v);
  }
  __syncthreads();
  if (tid < AG) {
    int any = 0;
    for (int t = 0; t < TN; ++t) any |= vt[tid * TP + t];
    p.cmi[a0 + tid] = any;
  }
  if (train) {
    for (int ag = 0; ag < AG; ++ag)
      rows_to_global(Xn + ag * TP * E::LDN, E::LDN, reinterpret_cast<T*>(p.s_nodes) + (long long)(a0 + ag) * TN * NF, NF, TN, NF, tid, 256);
  }
  // q | k | v = nodes W   [R][768]
  gemm_cols<T, AG, NF, NF>([&](int row, int s) { return pk + P_EQKV + (long long)row * NF + s * M::KSTEP; }, Xn, E::LDN, 0, 3 * EHD / 16, wv, 4, lane,
                           [&](int ct, f32x4 (&acc)[AG]) {
#pragma unroll
                             for (int mt = 0; mt < AG; ++mt) st_acc(QKV + (mt * 16 + ln) * E::LDQ + ct * 16 + 4 * g, acc[mt]);
                           });
  __syncthreads();
  if (train) {
    for (int ag = 0; ag < AG; ++ag)
      rows_to_global(QKV + ag * TP * E::LDQ, E::LDQ, reinterpret_cast<T*>(p.s_qkv) + (long long)(a0 + ag) * TN * 3 * EHD, 3 * EHD, TN, 3 * EHD, tid, 256);
  }
  // attention over the 11 steps, one (agent, head) per pass of a wave
  for (int u = wv; u < AG * EH; u += 4) {
    const int ag = u / EH, h = u % EH, row0 = ag * TP;
    float pr[4], f[4];
    enc_probs<T>(QKV, E::LDQ, row0, h, vt, a0 + ag, p.rng, p.site, p.p_drop, lane, pr, f);
    f32x4 st[2];
#pragma unroll
    for (int r = 0; r < 4; ++r) st[0][r] = pr[r] * f[r];
    st[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const typename Ch<T>::Frag pf = Ch<T>::from_acc(st);          // P^T chained as the B operand: O^T = V^T P^T
#pragma unroll
    for (int ct = 0; ct < ED / 16; ++ct) {
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
      o = M::mma(Ch<T>::ldA_tr(QKV, E::LDQ, 2 * EHD + h * ED + ct * 16, row0, lane), pf, o);
      if (ln >= TN) o = (f32x4){0.f, 0.f, 0.f, 0.f};             // padded query rows
      st_acc(ATT + (row0 + ln) * E::LDA + h * ED + ct * 16 + 4 * g, o);
    }
  }
  __syncthreads();
  if (train) {
    for (int ag = 0; ag < AG; ++ag)
      rows_to_global(ATT + ag * TP * E::LDA, E::LDA, reinterpret_cast<T*>(p.s_att) + (long long)(a0 + ag) * TN * EHD, EHD, TN, EHD, tid, 256);
  }
  // out = att Wo + bo  [R][320], max over the 11 steps straight from the accumulators (lane & 15 = step)
  gemm_cols<T, AG, EHD, EHD>([&](int row, int s) { return pk + P_EWO + (long long)row * EHD + s * M::KSTEP; }, ATT, E::LDA, 0, EO / 16, wv, 4, lane,
                             [&](int ct, f32x4 (&acc)[AG]) {
                               const float4 b4 = *reinterpret_cast<const float4*>(p.bo + ct * 16 + 4 * g);
                               const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                               for (int mt = 0; mt < AG; ++mt)
#pragma unroll
                                 for (int r = 0; r < 4; ++r) {
                                   const float v = rnd<T>(acc[mt][r] + bb[r]);
                                   float m = ln < TN ? v : -INFINITY;
                                   m = fmaxf(m, __shfl_xor(m, 1, 64)); m = fmaxf(m, __shfl_xor(m, 2, 64));
                                   m = fmaxf(m, __shfl_xor(m, 4, 64)); m = fmaxf(m, __shfl_xor(m, 8, 64));
                                   const unsigned long long bal = __ballot(ln < TN && v == m);
                                   if (ln == r) {
                                     const int col = ct * 16 + 4 * g + r;
                                     stf(CAT + mt * E::LDC + col, m);
                                     if (train) p.s_pmask[(long long)(a0 + mt) * EO + col] = (unsigned short)((bal >> (16 * g)) & 0xffffull);      // the steps that tie for the maximum
                                   }
                                 }
                             });
  // vector = type one-hot of step 0 (x[0, 5:8]) W
  for (int i = tid; i < AG * NF; i += 256) {
    const int ag = i / NF, c = i % NF;
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) v += x8[(ag * TP) * 8 + 5 + k] * rnd<T>(p.wv3[k * NF + c]);
    stf(CAT + ag * E::LDC + EO + c, v);
  }
  __syncthreads();
  if (train) rows_to_global(CAT, E::LDC, reinterpret_cast<T*>(p.s_cat) + (long long)a0 * CB, CB, AG, CB, tid, 256);
  // enc = ELU(cat Ws + bs)
  gemm_cols<T, 1, CB, E::KCW>([&](int row, int s) { return pk + P_EWS + (long long)row * CB + s * M::KSTEP; }, CAT, E::LDC, 0, CB / 16, wv, 4, lane,
                              [&](int ct, f32x4 (&acc)[1]) {
                                if (ln < AG) {
                                  const float4 b4 = *reinterpret_cast<const float4*>(p.bs + ct * 16 + 4 * g);
                                  const float v[4] = {elu_t<T>(acc[0][0] + b4.x), elu_t<T>(acc[0][1] + b4.y), elu_t<T>(acc[0][2] + b4.z), elu_t<T>(acc[0][3] + b4.w)};
                                  st4(reinterpret_cast<T*>(p.enc) + (long long)(a0 + ln) * CB + ct * 16 + 4 * g, v);
                                }
                              });
}
template <typename T> static size_t enc_fwd_lds() {
  typedef EG<T> E;
  return (size_t)E::R * 8 * 4 + E::R * 4 + sizeof(T) * ((size_t)E::R * E::LDN + (E::R + 16) * E::LDQ + E::R * E::LDA + 16 * E::LDC);
}

// ---- backward ---------------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void agent_enc_bwd_kernel(EncArgs p) {
  typedef EG<T> E;
  typedef Mma<T> M;
  constexpr int AG = E::AG, R = E::R;
  extern __shared__ __attribute__((aligned(16))) unsigned char ag_smem[];
  float* x8 = reinterpret_cast<float*>(ag_smem);
  int* vt = reinterpret_cast<int*>(x8 + R * 8);
  T* Xn = reinterpret_cast<T*>(vt + R);                           // [R][LDN] nodes (saved)
  T* QKV = Xn + R * E::LDN;                                       // [R + 16][LDQ] q|k|v (saved; loaded once the region's first tenants are dead), overwritten head by head with dq|dk|dv
  T* DS = QKV;                                                    //   first: [16][LDC] dpre of the sublayer (rows >= AG zero)
  T* DCAT = DS + 16 * E::LDC;                                     //          [16][LDC]
  T* DOUT = DCAT + 16 * E::LDC;                                   //          [R][LDO]
  static_assert(2 * 16 * E::LDC + R * E::LDO <= (R + 16) * E::LDQ, "the early tiles fit the q|k|v region");
  T* DATT = QKV + (R + 16) * E::LDQ;                              // [R + 16][LDA]
  T* WT = DATT + (R + 16) * E::LDA;                               // per wave: dS [16][KP], dS^T, Pd^T
  float* dnf = reinterpret_cast<float*>(WT);                      // at the end: [R][NF] d(pre-activation of the node features), f32
  static_assert((size_t)R * NF * 4 <= sizeof(T) * 4 * 3 * 16 * E::KP, "dnf fits the per-wave tiles");
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, ln = lane & 15, g = lane >> 4;
  const int a0 = blockIdx.x * AG;
  constexpr int V = Vec<T>::N;

  for (int i = tid; i < R * 8; i += 256) {
    const int row = i >> 3, ag = row / TP, t = row % TP;
    x8[i] = t < TN ? rnd<T>(track(p, a0 + ag)[t * 8 + (i & 7)]) : 0.f;
  }
  for (int i = tid; i < (R + 16) * E::LDA; i += 256) stf(DATT + i, 0.f);
  for (int i = tid; i < R * E::LDN; i += 256) stf(Xn + i, 0.f);
  for (int i = tid; i < 16 * E::LDC; i += 256) stf(DS + i, 0.f);
  __syncthreads();
  if (tid < R) vt[tid] = (tid % TP) < TN && x8[tid * 8] != 0.f;
  for (int i = tid; i < AG * TN * (NF / V); i += 256) {
    const int r = i / (NF / V), c = (i % (NF / V)) * V, ag = r / TN, t = r % TN;
    *reinterpret_cast<uint4*>(Xn + (ag * TP + t) * E::LDN + c) =
        *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(p.s_nodes) + ((long long)(a0 + ag) * TN + t) * NF + c);
  }
  // dpre_s = d_enc * ELU'(enc)      (ELU' from the output: y > 0 ? 1 : y + 1)
  for (int i = tid; i < AG * (CB / V); i += 256) {
    const int ag = i / (CB / V), c = (i % (CB / V)) * V;
    float d[V], y[V];
    if (p.d_enc_f32) {                 // d_enc_f32 slabs [B A][384] f32 (the interaction kernels' shares), added in a fixed order
#pragma unroll
      for (int e = 0; e < V; ++e) d[e] = 0.f;
      for (int sl = 0; sl < p.d_enc_f32; ++sl) {
        float t4[V];
#pragma unroll
        for (int e = 0; e < V; e += 4) ld4(reinterpret_cast<const float*>(p.d_enc) + ((long long)sl * p.B * (p.n_obs + p.n_occ) + a0 + ag) * CB + c + e, t4 + e);
#pragma unroll
        for (int e = 0; e < V; ++e) d[e] += t4[e];
      }
    } else ld16(reinterpret_cast<const T*>(p.d_enc) + (long long)(a0 + ag) * CB + c, d);
    ld16(reinterpret_cast<const T*>(p.enc) + (long long)(a0 + ag) * CB + c, y);
#pragma unroll
    for (int e = 0; e < V; ++e) d[e] *= y[e] > 0.f ? 1.f : y[e] + 1.f;
    st16(DS + ag * E::LDC + c, d);
    st16(reinterpret_cast<T*>(p.dpre_s) + (long long)(a0 + ag) * CB + c, d);
  }
  __syncthreads();
  // dcat = dpre_s Ws^T
  gemm_cols<T, 1, CB, E::KCW>([&](int row, int s) { return reinterpret_cast<const T*>(p.ws) + (long long)row * CB + s * M::KSTEP; }, DS, E::LDC, 0, CB / 16,
                              wv, 4, lane, [&](int ct, f32x4 (&acc)[1]) { st_acc(DCAT + ln * E::LDC + ct * 16 + 4 * g, acc[0]); });
  __syncthreads();
  // dout[(agent, t)][col] = dcat[agent][col] / ties on the steps that held the maximum (TF reduce_max gradient)
  for (int i = tid; i < R * EO; i += 256) {
    const int row = i / EO, col = i % EO, ag = row / TP, t = row % TP;
    float v = 0.f;
    if (t < TN) {
      const unsigned m = p.s_pmask[(long long)(a0 + ag) * EO + col];
      if ((m >> t) & 1u) v = ldf(DCAT + ag * E::LDC + col) / (float)__popc(m);
      stf(reinterpret_cast<T*>(p.dout) + ((long long)(a0 + ag) * TN + t) * EO + col, v);
    }
    stf(DOUT + row * E::LDO + col, v);
  }
  // dWv3 += x[0, 5:8]^T dcat[:, 320:]
  if (tid < 3 * NF) {
    const int k = tid / NF, c = tid % NF;
    float s = 0.f;
    for (int ag = 0; ag < AG; ++ag) s += x8[(ag * TP) * 8 + 5 + k] * ldf(DCAT + ag * E::LDC + EO + c);
    atomicAdd(p.dwv3 + k * NF + c, s);
  }
  __syncthreads();
  // datt = dout Wo^T
  gemm_cols<T, AG, EO, EO>([&](int row, int s) { return reinterpret_cast<const T*>(p.wo) + (long long)row * EO + s * M::KSTEP; }, DOUT, E::LDO, 0, EHD / 16,
                           wv, 4, lane, [&](int ct, f32x4 (&acc)[AG]) {
#pragma unroll
                             for (int mt = 0; mt < AG; ++mt) st_acc(DATT + (mt * 16 + ln) * E::LDA + ct * 16 + 4 * g, acc[mt]);
                           });
  __syncthreads();
  // the region's early tenants (dpre_s, dcat, dout) are dead: bring in the saved q|k|v (padded rows zero)
  for (int i = tid; i < (R + 16) * E::LDQ / V; i += 256) *reinterpret_cast<uint4*>(QKV + i * V) = make_uint4(0, 0, 0, 0);
  for (int i = tid; i < 4 * 3 * 16 * E::KP / V; i += 256) *reinterpret_cast<uint4*>(WT + i * V) = make_uint4(0, 0, 0, 0);
  __syncthreads();
  for (int i = tid; i < AG * TN * (3 * EHD / V); i += 256) {
    const int r = i / (3 * EHD / V), c = (i % (3 * EHD / V)) * V, ag = r / TN, t = r % TN;
    *reinterpret_cast<uint4*>(QKV + (ag * TP + t) * E::LDQ + c) =
        *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(p.s_qkv) + ((long long)(a0 + ag) * TN + t) * 3 * EHD + c);
  }
  __syncthreads();
  // attention backward, one (agent, head) per pass of a wave; probabilities recomputed
  T* dS = WT + wv * 3 * 16 * E::KP;
  T* dST = dS + 16 * E::KP;
  T* PdT = dST + 16 * E::KP;
  for (int u = wv; u < AG * EH; u += 4) {
    const int ag = u / EH, h = u % EH, row0 = ag * TP;
    float pr[4], f[4];
    enc_probs<T>(QKV, E::LDQ, row0, h, vt, a0 + ag, p.rng, p.site, p.p_drop, lane, pr, f);
    // dPd^T[j][i] = sum_c V[j][c] dO[i][c]
    f32x4 dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < ED / M::KSTEP; ++ks)
      dp = M::mma(M::load(QKV, E::LDQ, row0, 2 * EHD + h * ED + ks * M::KSTEP, lane), M::load(DATT, E::LDA, row0, h * ED + ks * M::KSTEP, lane), dp);
    float t = 0.f, dpv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { dpv[r] = (4 * g + r < TN && ln < TN) ? dp[r] * f[r] : 0.f; t += pr[r] * dpv[r]; }
    t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = 4 * g + r;
      const bool in = j < TN && ln < TN;
      const float ds = in ? pr[r] * (dpv[r] - t) * 0.125f : 0.f;       // the 1 / sqrt(64) of the logits folded in
      const float pd = in ? pr[r] * f[r] : 0.f;
      stf(dS + ln * E::KP + j, ds);
      stf(dST + j * E::KP + ln, ds);
      stf(PdT + j * E::KP + ln, pd);
    }
    // dQ^T[c][i] = sum_j K[j][c] dS[i][j] ; dK^T[c][j] = sum_i Q[i][c] dS[i][j] ; dV^T[c][j] = sum_i dO[i][c] Pd[i][j]
    f32x4 dq[ED / 16], dk[ED / 16], dv[ED / 16];
    const typename M::Frag bs = M::load(dS, E::KP, 0, 0, lane), bst = M::load(dST, E::KP, 0, 0, lane), bpt = M::load(PdT, E::KP, 0, 0, lane);
#pragma unroll
    for (int ct = 0; ct < ED / 16; ++ct) {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      dq[ct] = M::mma(M::load_tr(QKV, E::LDQ, EHD + h * ED + ct * 16, row0, lane), bs, z);
      dk[ct] = M::mma(M::load_tr(QKV, E::LDQ, h * ED + ct * 16, row0, lane), bst, z);
      dv[ct] = M::mma(M::load_tr(DATT, E::LDA, h * ED + ct * 16, row0, lane), bpt, z);
    }
    // q | k | v of this (agent, head) are dead: their slots take the gradients (rows >= 11 stay zero: dS / Pd are zero there)
#pragma unroll
    for (int ct = 0; ct < ED / 16; ++ct) {
      T* row = QKV + (row0 + ln) * E::LDQ + h * ED + ct * 16 + 4 * g;
      st_acc(row, dq[ct]); st_acc(row + EHD, dk[ct]); st_acc(row + 2 * EHD, dv[ct]);
    }
  }
  __syncthreads();
  for (int ag = 0; ag < AG; ++ag)
    rows_to_global(QKV + ag * TP * E::LDQ, E::LDQ, reinterpret_cast<T*>(p.dqkv) + (long long)(a0 + ag) * TN * 3 * EHD, 3 * EHD, TN, 3 * EHD, tid, 256);
  // dnodes[tok][i] = sum_{m,h,o} dqkv[tok][(m,h,o)] W_m[h][i][o]  -> d(pre-activation) = dnodes * ELU'(nodes)
  gemm_cols<T, AG, 3 * EHD, E::KCW>(
      [&](int row, int s) {
        const int kk = s * M::KSTEP, m = kk / EHD, h = (kk % EHD) / ED, o = kk % ED;
        const T* w = reinterpret_cast<const T*>(m == 0 ? p.wq : (m == 1 ? p.wk : p.wv));
        return w + ((long long)h * NF + row) * ED + o;
      },
      QKV, E::LDQ, 0, NF / 16, wv, 4, lane, [&](int ct, f32x4 (&acc)[AG]) {
#pragma unroll
        for (int mt = 0; mt < AG; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = mt * 16 + ln, c = ct * 16 + 4 * g + r;
            const float y = ldf(Xn + row * E::LDN + c);
            dnf[row * NF + c] = (ln < TN) ? acc[mt][r] * (y > 0.f ? 1.f : y + 1.f) : 0.f;
          }
      });
  __syncthreads();
  // dWn += x[:, :5]^T dpre ; dbn += column sums   (5 x 64 + 64 values per workgroup)
  for (int i = tid; i < 6 * NF; i += 256) {
    const int k = i / NF, c = i % NF;
    float s = 0.f;
    for (int row = 0; row < R; ++row) s += (k < 5 ? x8[row * 8 + k] : 1.f) * dnf[row * NF + c];
    atomicAdd(k < 5 ? p.dwn + k * NF + c : p.dbn + c, s);
  }
}
template <typename T> static size_t enc_bwd_lds() {
  typedef EG<T> E;
  return (size_t)E::R * 8 * 4 + E::R * 4 + sizeof(T) * ((size_t)E::R * E::LDN + (E::R + 16) * E::LDQ + (E::R + 16) * E::LDA + 4 * 3 * 16 * E::KP);
}


// =====================================================================================================================================
// interaction block: Cross_Attention over the 64 agents of a scene + FFN + residual + obs_norm | occ_norm (trajNet.py:65-87,135-187)
// 16-bit storage types only (the parity mode keeps the layer-by-layer chain for this block).
// A first version ran ONE workgroup per scene through the whole block: correct, and 211 us forward / 265 us backward at B = 8 -- eight
// CUs each streaming all 3.5 MB of weights behind ~45 dependent fetch -> multiply units per wave (profiles/r06_b_agent_fused_prof_v1.txt).
// The block is therefore cut where its weights can be spread: (scene, head) workgroups for the attention (0.2 MB of weights each) whose
// output-projection partials meet in an f32 accumulator (atomics), (scene, hidden chunk) workgroups for the FFN (0.6 MB each) whose FFN2
// partials meet the same way, and a row-wise tail; three launches per direction, no workgroup waits for another.
// =====================================================================================================================================
template <typename T> struct IGeo {
  static constexpr int PAD = LdsPad<T>::P;
  static constexpr int LD = CB + PAD;                 // full-width tiles [64][LD]
  static constexpr int LDH = IDH + PAD;               // per-head tiles [64][LDH]
  static constexpr int XR = NA + 16;                  // rows of the projection input: 64 agents + a tile holding the two segment-embedding rows
  static constexpr int NW = 8;
  static constexpr int NC = FF / CB;                  // hidden chunks (workgroups per scene in the FFN kernels)
};

struct IntArgs {
  const void* enc; const int* cmi; int n_obs, B;
  const void* pack; const void* seg;                 // seg_embed kernel [2][384], activation dtype
  const float* bo; const float* g1; const float* be1; const float* b1; const float* b2; const float* g2; const float* be2;
  const float* go; const float* beo; const float* gc; const float* bec;                      // obs_norm | occ_norm
  void* key;
  float* v1acc; float* u2acc; void* n1;              // workspaces: slabs [6][B 64][384] / [4][B 64][384] f32 (per head / hidden chunk partial sums), n1 [B 64][384] (= s_n1 in training)
  void* s_concat; void* s_qin; void* s_q; void* s_k; void* s_v; void* s_att; void* s_v1; void* s_h; void* s_u2; void* s_out;
  const long long* rng; int site_a, site_1, site_2; float p_drop;
  // backward
  const void* dkey; const void* wq; const void* wk; const void* wv; const void* wo; const void* w1; const void* w2;
  float* d_enc;                                      // slabs [7][B 64][384] f32: the residual's share (tail kernel) + one per head
  float* dn1acc;                                     // slabs [4][B 64][384] f32
  void* dq; void* dk; void* dv; void* dv1; void* dpre1; void* dz2;
  float* dseg; float* dg1; float* dbe1; float* dg2; float* dbe2; float* dgo; float* dbeo; float* dgc; float* dbec;
};

// row-wise work: a wave owns a row, a lane the columns 128 j + 2 lane + e (j < 3, e < 2)
#define AGF_ROW(j, e) for (int j = 0; j < 3; ++j) for (int e = 0; e < 2; ++e)
__device__ __forceinline__ int rcol(int j, int lane, int e) { return 128 * j + 2 * lane + e; }
__device__ __forceinline__ void row_stats(const float (&v)[3][2], float eps, float& mu, float& rs) {
  float s = 0.f;
#pragma unroll
  AGF_ROW(j, e) s += v[j][e];
  mu = wave_sum(s) * (1.f / CB);
  float q = 0.f;
#pragma unroll
  AGF_ROW(j, e) { const float d = v[j][e] - mu; q += d * d; }
  rs = rsqrtf(wave_sum(q) * (1.f / CB) + eps);
}
// LayerNorm backward of one row: xh = (x - mu) rs given, t = dy * gamma -> dx
__device__ __forceinline__ void row_ln_bwd(const float (&xh)[3][2], const float (&t)[3][2], float rs, float (&dx)[3][2]) {
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  AGF_ROW(j, e) { s1 += t[j][e]; s2 += t[j][e] * xh[j][e]; }
  s1 = wave_sum(s1) * (1.f / CB); s2 = wave_sum(s2) * (1.f / CB);
#pragma unroll
  AGF_ROW(j, e) dx[j][e] = rs * (t[j][e] - s1 - xh[j][e] * s2);
}

// logits of (head tile of the queries qt) x all 64 keys from the per-head tiles; softmax with the tfa mask; dropout factors.
// Lane: query = 16 qt + (lane & 15); st[jt][r]: key 16 jt + 4 g + r.  On return st = probabilities, f = keep / (1 - p).
template <typename T>
__device__ __forceinline__ void int_probs(const T* HQ, const T* HK, int ldh, int qt, const int* kval, long long bh, const long long* rng, int site,
                                          float p_drop, int lane, f32x4 (&st)[4], float (&f)[4][4]) {
  typedef Mma<T> M;
  const int ln = lane & 15, g = lane >> 4;
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) {
    st[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < IDH / M::KSTEP; ++ks)
      st[jt] = M::mma(M::load(HK, ldh, jt * 16, ks * M::KSTEP, lane), M::load(HQ, ldh, qt * 16, ks * M::KSTEP, lane), st[jt]);
  }
  const bool qv = kval[qt * 16 + ln] != 0;
  float m = -INFINITY;
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = st[jt][r] * 0.125f;                            // tfa: query /= sqrt(head_size = 64)
      if (!(qv && kval[16 * jt + 4 * g + r] != 0)) v = v + (-10e9f);
      st[jt][r] = v;
      m = fmaxf(m, v);
    }
  m = fmaxf(m, __shfl_xor(m, 16, 64)); m = fmaxf(m, __shfl_xor(m, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float e = exp_t<T>(st[jt][r] - m); st[jt][r] = e; sum += e; }
  sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.f / sum;
  const bool drop = rng != nullptr && p_drop > 0.f;
  const float dsc = drop ? 1.f / (1.f - p_drop) : 1.f;
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { st[jt][r] = rnd<T>(st[jt][r] * inv); f[jt][r] = 1.f; }
    if (drop) keep_scale4(rng, site, (bh * NA + qt * 16 + ln) * NA + 16 * jt + 4 * g, p_drop, dsc, f[jt]);
  }
}

// ---- forward 1: workgroup = (scene, head): concat / embed, q | k | v of the head, attention, the head's share of the output projection
template <typename T>
__global__ __launch_bounds__(512) void agent_int_attn_fwd_kernel(IntArgs p) {
  typedef IGeo<T> G;
  typedef Mma<T> M;
  constexpr int LD = G::LD, LDH = G::LDH, NW = G::NW, V = Vec<T>::N;
  extern __shared__ __attribute__((aligned(16))) unsigned char ag_smem[];
  T* XC = reinterpret_cast<T*>(ag_smem);                          // [80][LD] concat rows + the two segment rows
  T* HQ = XC + G::XR * LD;                                        // [64][LDH] x 3
  T* HK = HQ + NA * LDH;
  T* HV = HK + NA * LDH;
  T* OH = HV + NA * LDH;                                          // [64][LDH] attention output of the head
  int* kval = reinterpret_cast<int*>(OH + NA * LDH);              // [64]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, ln = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / IH, h = blockIdx.x % IH;
  const long long r0 = (long long)b * NA;                         // first row of the scene in the [B 64][.] tensors
  const T* pk = reinterpret_cast<const T*>(p.pack);
  const T* enc = reinterpret_cast<const T*>(p.enc) + r0 * CB;
  const bool train = p.s_q != nullptr, lead = train && h == 0;

  if (tid < NA) kval[tid] = p.cmi[r0 + tid];
  for (int i = tid; i < G::XR * (CB / V); i += 512) {
    const int row = i / (CB / V), c = (i % (CB / V)) * V;
    float v[V];
    if (row < NA) {
      ld16(enc + (long long)row * CB + c, v);
      const float cm = p.cmi[r0 + row] ? 1.f : 0.f;
#pragma unroll
      for (int e = 0; e < V; ++e) v[e] *= cm;
      st16(XC + row * LD + c, v);
      if (lead) {
        st16(reinterpret_cast<T*>(p.s_concat) + (r0 + row) * CB + c, v);
        float em[V];
        ld16(reinterpret_cast<const T*>(p.seg) + (row < p.n_obs ? 0 : CB) + c, em);
#pragma unroll
        for (int e = 0; e < V; ++e) em[e] += rnd<T>(v[e]);
        st16(reinterpret_cast<T*>(p.s_qin) + (r0 + row) * CB + c, em);
      }
    } else {
#pragma unroll
      for (int e = 0; e < V; ++e) v[e] = 0.f;
      if (row < NA + 2) ld16(reinterpret_cast<const T*>(p.seg) + (row - NA) * CB + c, v);
      st16(XC + row * LD + c, v);
    }
  }
  __syncthreads();
  // q | k | v of the head: 12 column tiles; q = (concat + embed) Wq -- the embedding's two distinct rows ride as a fifth token tile and are added per segment
  gemm_cols<T, 5, CB, CB>(
      [&](int row, int s) { return pk + P_IQKV + ((long long)((row >> 6) * CB + h * IDH + (row & 63))) * CB + s * M::KSTEP; }, XC, LD, 0, 3 * IDH / 16, wv, NW,
      lane, [&](int ct, f32x4 (&acc)[5]) {
        const int m = ct >> 2, c0 = (ct & 3) * 16 + 4 * g;
        T* HT = HQ + m * (NA * LDH);                         // HQ | HK | HV are adjacent (a select between LDS pointers put the staging arrays in scratch)
        T* sv = reinterpret_cast<T*>(m == 0 ? p.s_q : (m == 1 ? p.s_k : p.s_v));
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const int tok = mt * 16 + ln, src = (lane & 48) | (tok < p.n_obs ? 0 : 1);
          f32x4 q = acc[mt];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = __shfl(acc[4][r], src, 64);
            if (m == 0) q[r] += e;
          }
          st_acc(HT + tok * LDH + c0, q);
          if (train) st_acc(sv + (r0 + tok) * CB + h * IDH + c0, q);
        }
      });
  __syncthreads();
  {
    const int qt = wv & 3, half = wv >> 2;                     // the two waves of a query tile share its probabilities, each takes half of the head columns
    f32x4 st[4];
    float f[4][4];
    int_probs<T>(HQ, HK, LDH, qt, kval, (long long)b * IH + h, p.rng, p.site_a, p.p_drop, lane, st, f);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) st[jt][r] *= f[jt][r];
    f32x4 o[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int s = 0; s < NA / M::KSTEP; ++s) {
      const typename Ch<T>::Frag pf = Ch<T>::from_acc(&st[s * Ch<T>::ND]);       // O^T = V^T P^T, P^T chained from the accumulators
#pragma unroll
      for (int jd = 0; jd < 2; ++jd) o[jd] = M::mma(Ch<T>::ldA_tr(HV, LDH, 16 * (2 * half + jd), s * M::KSTEP, lane), pf, o[jd]);
    }
#pragma unroll
    for (int jd = 0; jd < 2; ++jd) {
      const int c0 = 16 * (2 * half + jd) + 4 * g;
      st_acc(OH + (qt * 16 + ln) * LDH + c0, o[jd]);
      if (train) st_acc(reinterpret_cast<T*>(p.s_att) + (r0 + qt * 16 + ln) * CB + h * IDH + c0, o[jd]);
    }
  }
  __syncthreads();
  // the head's share of v1 = att Wo: K = its 64 columns; the six heads meet in the f32 accumulator
  gemm_cols<T, 4, IDH, IDH>([&](int row, int s) { return pk + P_IWO + (long long)row * CB + h * IDH + s * M::KSTEP; }, OH, LDH, 0, CB / 16, wv, NW, lane,
                            [&](int ct, f32x4 (&acc)[4]) {
#pragma unroll
                              for (int mt = 0; mt < 4; ++mt)      // the head's slab (plain 16-byte stores; the FFN kernel adds the six in a fixed order)
                                *reinterpret_cast<f32x4*>(p.v1acc + ((long long)h * p.B * NA + r0 + mt * 16 + ln) * CB + ct * 16 + 4 * g) = acc[mt];
                            });
}
template <typename T> static size_t int_attn_fwd_lds() {
  typedef IGeo<T> G;
  return sizeof(T) * ((size_t)G::XR * G::LD + 4 * NA * G::LDH) + NA * 4;
}

// ---- forward 2: workgroup = (scene, hidden chunk): v1 = acc + bo, n1 = LayerNorm(v1); the chunk's hidden columns; its share of FFN2
template <typename T>
__global__ __launch_bounds__(512) void agent_int_ffn_fwd_kernel(IntArgs p) {
  typedef IGeo<T> G;
  typedef Mma<T> M;
  constexpr int LD = G::LD, NW = G::NW;
  extern __shared__ __attribute__((aligned(16))) unsigned char ag_smem[];
  T* N1 = reinterpret_cast<T*>(ag_smem);                          // [64][LD]
  T* HC = N1 + NA * LD;                                           // [64][LD] hidden chunk
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, ln = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / G::NC, c = blockIdx.x % G::NC;
  const long long r0 = (long long)b * NA;
  const T* pk = reinterpret_cast<const T*>(p.pack);
  const bool train = p.s_h != nullptr, lead = c == 0;
  const bool drop = p.rng != nullptr && p.p_drop > 0.f;
  const float dsc = drop ? 1.f / (1.f - p.p_drop) : 1.f;
  for (int row = wv; row < NA; row += NW) {
    float v[3][2], mu, rs;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int cc = rcol(j, lane, 0);
      float2 a = make_float2(p.bo[cc], p.bo[cc + 1]);
#pragma unroll
      for (int hh = 0; hh < IH; ++hh) {
        const float2 t = *reinterpret_cast<const float2*>(p.v1acc + ((long long)hh * p.B * NA + r0 + row) * CB + cc);
        a.x += t.x; a.y += t.y;
      }
      v[j][0] = rnd<T>(a.x); v[j][1] = rnd<T>(a.y);
      if (lead && train) *reinterpret_cast<uint32_t*>(reinterpret_cast<T*>(p.s_v1) + (r0 + row) * CB + cc) = pack2<T>(v[j][0], v[j][1]);
    }
    row_stats(v, 1e-3f, mu, rs);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int cc = rcol(j, lane, 0);
      const uint32_t w = pack2<T>((v[j][0] - mu) * rs * p.g1[cc] + p.be1[cc], (v[j][1] - mu) * rs * p.g1[cc + 1] + p.be1[cc + 1]);
      *reinterpret_cast<uint32_t*>(N1 + row * LD + cc) = w;
      if (lead && p.n1) *reinterpret_cast<uint32_t*>(reinterpret_cast<T*>(p.n1) + (r0 + row) * CB + cc) = w;
    }
  }
  __syncthreads();
  gemm_cols<T, 4, CB, CB>([&](int row, int s) { return pk + P_IW1 + ((long long)(c * CB + row)) * CB + s * M::KSTEP; }, N1, LD, 0, CB / 16, wv, NW, lane,
                          [&](int ct, f32x4 (&acc)[4]) {
                            const int col = c * CB + ct * 16 + 4 * g;
                            const float4 b4 = *reinterpret_cast<const float4*>(p.b1 + col);
                            const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                            for (int mt = 0; mt < 4; ++mt) {
                              const int tok = mt * 16 + ln;
                              float f[4] = {1.f, 1.f, 1.f, 1.f};
                              if (drop) keep_scale4(p.rng, p.site_1, (r0 + tok) * FF + col, p.p_drop, dsc, f);
                              f32x4 hv;
#pragma unroll
                              for (int r = 0; r < 4; ++r) hv[r] = rnd<T>(elu_t<T>(acc[mt][r] + bb[r])) * f[r];
                              st_acc(HC + tok * LD + ct * 16 + 4 * g, hv);
                              if (train) st_acc(reinterpret_cast<T*>(p.s_h) + (r0 + tok) * FF + col, hv);
                            }
                          });
  __syncthreads();
  gemm_cols<T, 4, CB, CB>([&](int row, int s) { return pk + P_IW2 + (long long)row * FF + c * CB + s * M::KSTEP; }, HC, LD, 0, CB / 16, wv, NW, lane,
                          [&](int ct, f32x4 (&acc)[4]) {
#pragma unroll
                            for (int mt = 0; mt < 4; ++mt)
                              *reinterpret_cast<f32x4*>(p.u2acc + ((long long)c * p.B * NA + r0 + mt * 16 + ln) * CB + ct * 16 + 4 * g) = acc[mt];
                          });
}
template <typename T> static size_t int_ffn_lds() { return sizeof(T) * (size_t)2 * NA * IGeo<T>::LD; }

// ---- forward 3 (row-wise, a wave per agent row): u2 = dropout(acc + b2); value = LayerNorm(u2); out = enc + value + embed; key = obs_norm | occ_norm (out)
template <typename T>
__global__ __launch_bounds__(256) void agent_int_out_fwd_kernel(IntArgs p) {
  const int lane = threadIdx.x & 63;
  const long long row = blockIdx.x * 4ll + (threadIdx.x >> 6);
  if (row >= (long long)p.B * NA) return;
  const bool ob = (int)(row % NA) < p.n_obs;
  const bool drop = p.rng != nullptr && p.p_drop > 0.f;
  const float dsc = drop ? 1.f / (1.f - p.p_drop) : 1.f;
  float v[3][2], mu, rs;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int c = rcol(j, lane, 0);
    float2 a = make_float2(p.b2[c], p.b2[c + 1]);
#pragma unroll
    for (int cc = 0; cc < FF / CB; ++cc) {
      const float2 t = *reinterpret_cast<const float2*>(p.u2acc + ((long long)cc * p.B * NA + row) * CB + c);
      a.x += t.x; a.y += t.y;
    }
    float f0 = 1.f, f1 = 1.f;
    if (drop) {
      bool k4[4];
      const long long idx = row * CB + c;
      keep4(p.rng, p.site_2, idx >> 2, p.p_drop, k4);
      f0 = k4[idx & 3] ? dsc : 0.f; f1 = k4[(idx & 3) + 1] ? dsc : 0.f;
    }
    v[j][0] = rnd<T>(rnd<T>(a.x) * f0); v[j][1] = rnd<T>(rnd<T>(a.y) * f1);
    if (p.s_u2) *reinterpret_cast<uint32_t*>(reinterpret_cast<T*>(p.s_u2) + row * CB + c) = pack2<T>(v[j][0], v[j][1]);
  }
  row_stats(v, 1e-3f, mu, rs);
#pragma unroll
  AGF_ROW(j, e) {
    const int c = rcol(j, lane, e);
    const float val = rnd<T>((v[j][e] - mu) * rs * p.g2[c] + p.be2[c]);
    const float t = rnd<T>(ldf(reinterpret_cast<const T*>(p.enc) + row * CB + c) + val);                      // (enc + value) rounded, then + embed: the order of the two adds
    v[j][e] = rnd<T>(t + ldf(reinterpret_cast<const T*>(p.seg) + (ob ? 0 : CB) + c));
    if (p.s_out) stf(reinterpret_cast<T*>(p.s_out) + row * CB + c, v[j][e]);
  }
  row_stats(v, 1e-3f, mu, rs);
  const float* gm = ob ? p.go : p.gc;
  const float* bt = ob ? p.beo : p.bec;
#pragma unroll
  AGF_ROW(j, e) { const int c = rcol(j, lane, e); stf(reinterpret_cast<T*>(p.key) + row * CB + c, (v[j][e] - mu) * rs * gm[c] + bt[c]); }
}

// ---- backward 1 (row-wise; a workgroup = 4 waves x 4 rows of ONE segment): obs_norm | occ_norm backward -> dout (d_enc's residual share, d_embed);
// LayerNorm2 backward; dropout2 -> dz2
template <typename T>
__global__ __launch_bounds__(256) void agent_int_out_bwd_kernel(IntArgs p) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const bool drop = p.rng != nullptr && p.p_drop > 0.f;
  const float dsc = drop ? 1.f / (1.f - p.p_drop) : 1.f;
  float ag[2][3][2], ab[2][3][2], ae[2][3][2], ag2[3][2], ab2[3][2];          // gamma / beta of obs | occ, d_embed per segment, gamma2 / beta2
#pragma unroll
  AGF_ROW(j, e) { ag[0][j][e] = ag[1][j][e] = ab[0][j][e] = ab[1][j][e] = ae[0][j][e] = ae[1][j][e] = ag2[j][e] = ab2[j][e] = 0.f; }
  const long long nrows = (long long)p.B * NA;
  for (long long row = blockIdx.x * 16ll + wv; row < nrows && row < (blockIdx.x + 1) * 16ll; row += 4) {
    const int sg = (int)(row % NA) < p.n_obs ? 0 : 1;
    const float* gm = sg ? p.gc : p.go;
    float x[3][2], xh[3][2], t[3][2], d[3][2], mu, rs;
#pragma unroll
    AGF_ROW(j, e) x[j][e] = ldf(reinterpret_cast<const T*>(p.s_out) + row * CB + rcol(j, lane, e));
    row_stats(x, 1e-3f, mu, rs);
#pragma unroll
    AGF_ROW(j, e) {
      const int c = rcol(j, lane, e);
      const float dy = ldf(reinterpret_cast<const T*>(p.dkey) + row * CB + c);
      xh[j][e] = (x[j][e] - mu) * rs;
      ag[sg][j][e] += dy * xh[j][e]; ab[sg][j][e] += dy;
      t[j][e] = dy * gm[c];
    }
    row_ln_bwd(xh, t, rs, d);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      d[j][0] = rnd<T>(d[j][0]); d[j][1] = rnd<T>(d[j][1]);
      ae[sg][j][0] += d[j][0]; ae[sg][j][1] += d[j][1];
      *reinterpret_cast<float2*>(p.d_enc + row * CB + rcol(j, lane, 0)) = make_float2(d[j][0], d[j][1]);   // the residual's share; the attention kernel adds its own
    }
    // value = LayerNorm2(u2)
#pragma unroll
    AGF_ROW(j, e) x[j][e] = ldf(reinterpret_cast<const T*>(p.s_u2) + row * CB + rcol(j, lane, e));
    row_stats(x, 1e-3f, mu, rs);
#pragma unroll
    AGF_ROW(j, e) {
      const int c = rcol(j, lane, e);
      xh[j][e] = (x[j][e] - mu) * rs;
      ag2[j][e] += d[j][e] * xh[j][e]; ab2[j][e] += d[j][e];
      t[j][e] = d[j][e] * p.g2[c];
    }
    float du[3][2];
    row_ln_bwd(xh, t, rs, du);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int c = rcol(j, lane, 0);
      if (drop) {                                              // dropout2: the two columns of a lane share a draw group (c even; group of 4)
        bool k4[4];
        const long long idx = row * CB + c;
        keep4(p.rng, p.site_2, idx >> 2, p.p_drop, k4);
        du[j][0] *= k4[idx & 3] ? dsc : 0.f; du[j][1] *= k4[(idx & 3) + 1] ? dsc : 0.f;
      }
      *reinterpret_cast<uint32_t*>(reinterpret_cast<T*>(p.dz2) + row * CB + c) = pack2<T>(du[j][0], du[j][1]);
    }
  }
  // the workgroup's 16 rows -> LDS -> one atomic per column and array
  __shared__ float red[4][8][CB];
#pragma unroll
  AGF_ROW(j, e) {
    const int c = rcol(j, lane, e);
    red[wv][0][c] = ag[0][j][e]; red[wv][1][c] = ab[0][j][e]; red[wv][2][c] = ag[1][j][e]; red[wv][3][c] = ab[1][j][e];
    red[wv][4][c] = ae[0][j][e]; red[wv][5][c] = ae[1][j][e]; red[wv][6][c] = ag2[j][e]; red[wv][7][c] = ab2[j][e];
  }
  __syncthreads();
  float* const dst[8] = {p.dgo, p.dbeo, p.dgc, p.dbec, p.dseg, p.dseg + CB, p.dg2, p.dbe2};
  for (int i = threadIdx.x; i < 8 * CB; i += 256) {
    const int a = i / CB, c = i % CB;
    const float s = red[0][a][c] + red[1][a][c] + red[2][a][c] + red[3][a][c];
    if (s != 0.f) atomicAdd(dst[a] + c, s);
  }
}

// ---- backward 2: workgroup = (scene, hidden chunk): dh = dz2 W2^T ; dpre = dh * keep * ELU' ; its share of dn1 = dpre W1^T
template <typename T>
__global__ __launch_bounds__(512) void agent_int_ffn_bwd_kernel(IntArgs p) {
  typedef IGeo<T> G;
  typedef Mma<T> M;
  constexpr int LD = G::LD, NW = G::NW, V = Vec<T>::N;
  extern __shared__ __attribute__((aligned(16))) unsigned char ag_smem[];
  T* DZ = reinterpret_cast<T*>(ag_smem);                          // [64][LD]
  T* DP = DZ + NA * LD;                                           // [64][LD]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, ln = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / G::NC, c = blockIdx.x % G::NC;
  const long long r0 = (long long)b * NA;
  const bool drop = p.rng != nullptr && p.p_drop > 0.f;
  const float dsc = drop ? 1.f / (1.f - p.p_drop) : 1.f;
  for (int i = tid; i < NA * (CB / V); i += 512) {
    const int row = i / (CB / V), cc = (i % (CB / V)) * V;
    *reinterpret_cast<uint4*>(DZ + row * LD + cc) = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(p.dz2) + (r0 + row) * CB + cc);
  }
  __syncthreads();
  gemm_cols<T, 4, CB, CB>([&](int row, int s) { return reinterpret_cast<const T*>(p.w2) + ((long long)(c * CB + row)) * CB + s * M::KSTEP; }, DZ, LD, 0, CB / 16,
                          wv, NW, lane, [&](int ct, f32x4 (&acc)[4]) {
                            const int col = c * CB + ct * 16 + 4 * g;
#pragma unroll
                            for (int mt = 0; mt < 4; ++mt) {
                              const int tok = mt * 16 + ln;
                              float hv[4], f[4] = {1.f, 1.f, 1.f, 1.f};
                              ld4(reinterpret_cast<const T*>(p.s_h) + (r0 + tok) * FF + col, hv);      // dropout(elu(pre)): elu(pre) / (1 - p) where kept
                              if (drop) keep_scale4(p.rng, p.site_1, (r0 + tok) * FF + col, p.p_drop, dsc, f);
                              f32x4 dp;
#pragma unroll
                              for (int r = 0; r < 4; ++r) {
                                const float y = drop ? hv[r] * (1.f - p.p_drop) : hv[r];                  // elu(pre)
                                dp[r] = acc[mt][r] * f[r] * (y > 0.f ? 1.f : y + 1.f);
                              }
                              st_acc(DP + tok * LD + ct * 16 + 4 * g, dp);
                              st_acc(reinterpret_cast<T*>(p.dpre1) + (r0 + tok) * FF + col, dp);
                            }
                          });
  __syncthreads();
  gemm_cols<T, 4, CB, CB>([&](int row, int s) { return reinterpret_cast<const T*>(p.w1) + (long long)row * FF + c * CB + s * M::KSTEP; }, DP, LD, 0, CB / 16, wv, NW,
                          lane, [&](int ct, f32x4 (&acc)[4]) {
#pragma unroll
                            for (int mt = 0; mt < 4; ++mt)
                              *reinterpret_cast<f32x4*>(p.dn1acc + ((long long)c * p.B * NA + r0 + mt * 16 + ln) * CB + ct * 16 + 4 * g) = acc[mt];
                          });
}

// ---- backward 3: workgroup = (scene, head): LayerNorm1 backward (every head for itself; head 0 publishes dv1 and the parameter gradients),
// the head's datt = dv1 Wo[h]^T, attention backward, dq | dk | dv, and the head's share of d(qin), d(concat) -> d_enc, d_embed
template <typename T>
__global__ __launch_bounds__(512) void agent_int_attn_bwd_kernel(IntArgs p) {
  typedef IGeo<T> G;
  typedef Mma<T> M;
  constexpr int LD = G::LD, LDH = G::LDH, NW = G::NW, V = Vec<T>::N;
  extern __shared__ __attribute__((aligned(16))) unsigned char ag_smem[];
  T* DV1 = reinterpret_cast<T*>(ag_smem);                         // [64][LD]
  T* HQ = DV1 + NA * LD;                                          // [64][LDH] x 5: q, k, v, dS^T, Pd^T
  T* HK = HQ + NA * LDH; T* HV = HK + NA * LDH; T* DST = HV + NA * LDH; T* PDT = DST + NA * LDH;
  T* DAT = PDT + NA * LDH;                                        // [64][LDH] datt of the head
  T* DQh = DAT + NA * LDH;                                        // [64][LDH] x 3
  T* DKh = DQh + NA * LDH; T* DVh = DKh + NA * LDH;
  int* kval = reinterpret_cast<int*>(DVh + NA * LDH);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, ln = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / IH, h = blockIdx.x % IH;
  const long long r0 = (long long)b * NA;
  const bool lead = h == 0;
  if (tid < NA) kval[tid] = p.cmi[r0 + tid];
  for (int i = tid; i < 3 * NA * (IDH / V); i += 512) {
    const int m = i / (NA * (IDH / V)), rem = i % (NA * (IDH / V)), row = rem / (IDH / V), c = (rem % (IDH / V)) * V;
    const T* src = reinterpret_cast<const T*>(m == 0 ? p.s_q : (m == 1 ? p.s_k : p.s_v)) + (r0 + row) * CB + h * IDH + c;
    *reinterpret_cast<uint4*>((m == 0 ? HQ : (m == 1 ? HK : HV)) + row * LDH + c) = *reinterpret_cast<const uint4*>(src);
  }
  {
    float ag1[3][2], ab1[3][2];
#pragma unroll
    AGF_ROW(j, e) ag1[j][e] = ab1[j][e] = 0.f;
    for (int row = wv; row < NA; row += NW) {
      float x[3][2], xh[3][2], t[3][2], d[3][2], mu, rs;
#pragma unroll
      AGF_ROW(j, e) x[j][e] = ldf(reinterpret_cast<const T*>(p.s_v1) + (r0 + row) * CB + rcol(j, lane, e));
      row_stats(x, 1e-3f, mu, rs);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int c = rcol(j, lane, 0);
        float2 dy = make_float2(0.f, 0.f);
#pragma unroll
        for (int cc = 0; cc < FF / CB; ++cc) {
          const float2 t2 = *reinterpret_cast<const float2*>(p.dn1acc + ((long long)cc * p.B * NA + r0 + row) * CB + c);
          dy.x += t2.x; dy.y += t2.y;
        }
        xh[j][0] = (x[j][0] - mu) * rs; xh[j][1] = (x[j][1] - mu) * rs;
        ag1[j][0] += dy.x * xh[j][0]; ab1[j][0] += dy.x; ag1[j][1] += dy.y * xh[j][1]; ab1[j][1] += dy.y;
        t[j][0] = dy.x * p.g1[c]; t[j][1] = dy.y * p.g1[c + 1];
      }
      row_ln_bwd(xh, t, rs, d);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int c = rcol(j, lane, 0);
        const uint32_t w = pack2<T>(d[j][0], d[j][1]);
        *reinterpret_cast<uint32_t*>(DV1 + row * LD + c) = w;
        if (lead) *reinterpret_cast<uint32_t*>(reinterpret_cast<T*>(p.dv1) + (r0 + row) * CB + c) = w;
      }
    }
    if (lead) {
#pragma unroll
      AGF_ROW(j, e) { const int c = rcol(j, lane, e); atomicAdd(p.dg1 + c, ag1[j][e]); atomicAdd(p.dbe1 + c, ab1[j][e]); }
    }
  }
  __syncthreads();
  // datt of the head = dv1 Wo[h]^T  (4 column tiles)
  gemm_cols<T, 4, CB, CB>([&](int row, int s) { return reinterpret_cast<const T*>(p.wo) + ((long long)(h * IDH + row)) * CB + s * M::KSTEP; }, DV1, LD, 0, IDH / 16,
                          wv, NW, lane, [&](int ct, f32x4 (&acc)[4]) {
#pragma unroll
                            for (int mt = 0; mt < 4; ++mt) st_acc(DAT + (mt * 16 + ln) * LDH + ct * 16 + 4 * g, acc[mt]);
                          });
  __syncthreads();
  {
    const int qt = wv & 3, half = wv >> 2;
    f32x4 st[4];
    float f[4][4];
    int_probs<T>(HQ, HK, LDH, qt, kval, (long long)b * IH + h, p.rng, p.site_a, p.p_drop, lane, st, f);
    // dPd^T[key][query] = sum_c V[key][c] dO[query][c]
    f32x4 ds[4];
    float t = 0.f;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      f32x4 dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < IDH / M::KSTEP; ++ks)
        dp = M::mma(M::load(HV, LDH, jt * 16, ks * M::KSTEP, lane), M::load(DAT, LDH, qt * 16, ks * M::KSTEP, lane), dp);
#pragma unroll
      for (int r = 0; r < 4; ++r) { ds[jt][r] = dp[r] * f[jt][r]; t += st[jt][r] * ds[jt][r]; }
    }
    t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) ds[jt][r] = st[jt][r] * (ds[jt][r] - t) * 0.125f;
    // dS^T and Pd^T tiles [key][query] for the products that contract over the queries (the two waves of a query tile write half each)
#pragma unroll
    for (int jt = 2 * half; jt < 2 * half + 2; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = 16 * jt + 4 * g + r;
        stf(DST + key * LDH + qt * 16 + ln, ds[jt][r]);
        stf(PDT + key * LDH + qt * 16 + ln, st[jt][r] * f[jt][r]);
      }
    // dQ^T[c][query] = sum_key K[key][c] dS[query][key], dS chained from the registers (half of the head columns per wave)
    f32x4 dq[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int s = 0; s < NA / M::KSTEP; ++s) {
      const typename Ch<T>::Frag sf = Ch<T>::from_acc(&ds[s * Ch<T>::ND]);
#pragma unroll
      for (int jd = 0; jd < 2; ++jd) dq[jd] = M::mma(Ch<T>::ldA_tr(HK, LDH, 16 * (2 * half + jd), s * M::KSTEP, lane), sf, dq[jd]);
    }
#pragma unroll
    for (int jd = 0; jd < 2; ++jd) {
      const int col = 16 * (2 * half + jd) + 4 * g;
      st_acc(DQh + (qt * 16 + ln) * LDH + col, dq[jd]);
      st_acc(reinterpret_cast<T*>(p.dq) + (r0 + qt * 16 + ln) * CB + h * IDH + col, dq[jd]);
    }
  }
  __syncthreads();
  // dK^T[c][key] = sum_query Q[query][c] dS[query][key] ; dV^T[c][key] = sum_query dO[query][c] Pd[query][key]: 32 (matrix, c tile, key tile) units
  for (int u = wv; u < 32; u += NW) {
    const int mv = u >> 4, ct = (u >> 2) & 3, jt = u & 3;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NA / M::KSTEP; ++ks)
      acc = M::mma(M::load_tr(mv == 0 ? HQ : DAT, LDH, ct * 16, ks * M::KSTEP, lane), M::load(mv == 0 ? DST : PDT, LDH, jt * 16, ks * M::KSTEP, lane), acc);
    const int key = jt * 16 + ln, col = ct * 16 + 4 * g;
    st_acc((mv == 0 ? DKh : DVh) + key * LDH + col, acc);
    st_acc(reinterpret_cast<T*>(mv == 0 ? p.dk : p.dv) + (r0 + key) * CB + h * IDH + col, acc);
  }
  __syncthreads();
  // the head's share of d(qin) = dq_h Wq[h]^T and d(concat) = dk_h Wk[h]^T + dv_h Wv[h]^T
  f32x4 dqi[3][4], dco[3][4];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) dqi[j][mt] = dco[j][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  gemm_cols_acc<T, 4, IDH, 3>([&](int row, int s) { return reinterpret_cast<const T*>(p.wq) + ((long long)h * CB + row) * IDH + s * M::KSTEP; }, DQh, LDH, 0, wv, NW,
                              lane, dqi);
  gemm_cols_acc<T, 4, IDH, 3>([&](int row, int s) { return reinterpret_cast<const T*>(p.wk) + ((long long)h * CB + row) * IDH + s * M::KSTEP; }, DKh, LDH, 0, wv, NW,
                              lane, dco);
  gemm_cols_acc<T, 4, IDH, 3>([&](int row, int s) { return reinterpret_cast<const T*>(p.wv) + ((long long)h * CB + row) * IDH + s * M::KSTEP; }, DVh, LDH, 0, wv, NW,
                              lane, dco);
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int col = (wv + j * NW) * 16 + 4 * g;
    float e0[4] = {0.f, 0.f, 0.f, 0.f}, e1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int tok = mt * 16 + ln;
      const float cm = kval[tok] != 0 ? 1.f : 0.f;
      f32x4 de;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float q = dqi[j][mt][r];
        if (tok < p.n_obs) e0[r] += q; else e1[r] += q;
        de[r] = cm * (q + dco[j][mt][r]);
      }
      *reinterpret_cast<f32x4*>(p.d_enc + ((long long)(1 + h) * p.B * NA + r0 + tok) * CB + col) = de;      // slab 1 + h (slab 0: the residual's share)
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { e0[r] += __shfl_xor(e0[r], o, 64); e1[r] += __shfl_xor(e1[r], o, 64); }
      if (ln == 0) { atomicAdd(p.dseg + col + r, e0[r]); atomicAdd(p.dseg + CB + col + r, e1[r]); }
    }
  }
}
template <typename T> static size_t int_attn_bwd_lds() {
  typedef IGeo<T> G;
  return sizeof(T) * ((size_t)NA * G::LD + 9 * NA * G::LDH) + NA * 4;
}

}  // namespace agf

// ---- C ABI ---------------------------------------------------------------------------------------------------------------------------
#include "agent_fused_abi.h"

extern "C" long long stj_agent_pack_workspace_bytes(int dtype) { return agf::P_TOTAL * (dtype == STJ_F32 ? 4 : 2); }

extern "C" int stj_agent_pack(const stj_agent_weights* w, void* out, int dtype, hipStream_t stream) {
  if (!w || !out) { stj_set_error("stj_agent_pack: null pointer"); return STJ_EINVAL; }
  using namespace agf;
  PackArgs a = {};
  a.j[0] = {w->e_wq, P_EQKV, NF, EHD, ED};
  a.j[1] = {w->e_wk, P_EQKV + (long long)EHD * NF, NF, EHD, ED};
  a.j[2] = {w->e_wv, P_EQKV + 2LL * EHD * NF, NF, EHD, ED};
  a.j[3] = {w->e_wo, P_EWO, EHD, EO, 0};
  a.j[4] = {w->e_ws, P_EWS, CB, CB, 0};
  a.j[5] = {w->i_wq, P_IQKV, CB, CB, IDH};
  a.j[6] = {w->i_wk, P_IQKV + (long long)CB * CB, CB, CB, IDH};
  a.j[7] = {w->i_wv, P_IQKV + 2LL * CB * CB, CB, CB, IDH};
  a.j[8] = {w->i_wo, P_IWO, CB, CB, 0};
  a.j[9] = {w->i_w1, P_IW1, CB, FF, 0};
  a.j[10] = {w->i_w2, P_IW2, FF, CB, 0};
  for (int i = 0; i < NPACK; ++i)
    if (!a.j[i].src) { stj_set_error("stj_agent_pack: weight %d is NULL", i); return STJ_EINVAL; }
  a.out = out;
  const dim3 grid(64, NPACK);
  if (dtype == STJ_BF16) hipLaunchKernelGGL(agent_pack_kernel<bf16>, grid, dim3(256), 0, stream, a);
  else if (dtype == STJ_F16) hipLaunchKernelGGL(agent_pack_kernel<f16>, grid, dim3(256), 0, stream, a);
  else if (dtype == STJ_F32) hipLaunchKernelGGL(agent_pack_kernel<float>, grid, dim3(256), 0, stream, a);
  else { stj_set_error("stj_agent_pack: bad dtype %d", dtype); return STJ_EINVAL; }
  return stj_check_launch("stj_agent_pack");
}

extern "C" int stj_agent_enc_supported(int n_obs, int n_occ, int Tn, int dtype) {
  return Tn == agf::TN && n_obs >= 0 && n_occ >= 0 && (n_obs + n_occ) % 2 == 0 && n_obs + n_occ > 0 && stj_dtype_ok(dtype);
}

template <typename T, bool BWD> static int enc_launch(const agf::EncArgs& a, hipStream_t stream) {
  using namespace agf;
  static PerDevice<int> attr_set;
  const size_t lds = BWD ? enc_bwd_lds<T>() : enc_fwd_lds<T>();
  const void* fn = BWD ? (const void*)agent_enc_bwd_kernel<T> : (const void*)agent_enc_fwd_kernel<T>;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)agent_enc_fwd_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)enc_fwd_lds<T>()) != hipSuccess ||
        hipFuncSetAttribute((const void*)agent_enc_bwd_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)enc_bwd_lds<T>()) != hipSuccess) {
      stj_set_error("stj_agent_enc: cannot reserve %zu bytes of LDS", lds);
      return STJ_ELAUNCH;
    }
    attr_set = 1;
  }
  (void)fn;
  const int nag = a.B * (a.n_obs + a.n_occ);
  const dim3 grid(nag / EG<T>::AG);
  if (BWD) hipLaunchKernelGGL(agent_enc_bwd_kernel<T>, grid, dim3(256), lds, stream, a);
  else hipLaunchKernelGGL(agent_enc_fwd_kernel<T>, grid, dim3(256), lds, stream, a);
  return stj_check_launch(BWD ? "stj_agent_enc_bwd" : "stj_agent_enc_fwd");
}

static int enc_args(const stj_agent_enc_args* s, agf::EncArgs& a, bool bwd) {
  if (!s || !s->obs || !s->occ || !s->enc || !s->cmi) { stj_set_error("stj_agent_enc: null pointer"); return STJ_EINVAL; }
  if (s->B <= 0) return 1;
  if (!stj_agent_enc_supported(s->n_obs, s->n_occ, agf::TN, s->dtype)) { stj_set_error("stj_agent_enc: geometry / dtype not supported"); return STJ_EUNSUPPORTED; }
  if (!(s->p_drop >= 0.f && s->p_drop < 1.f)) { stj_set_error("stj_agent_enc: need 0 <= p_drop < 1"); return STJ_EINVAL; }
  a = {};
  a.obs = s->obs; a.occ = s->occ; a.n_obs = s->n_obs; a.n_occ = s->n_occ; a.B = s->B; a.pack = s->pack;
  a.wn = s->wn; a.bn = s->bn; a.wv3 = s->wv3; a.bo = s->bo; a.bs = s->bs; a.enc = s->enc; a.cmi = s->cmi;
  a.s_nodes = s->s_nodes; a.s_qkv = s->s_qkv; a.s_att = s->s_att; a.s_pmask = (unsigned short*)s->s_pmask; a.s_cat = s->s_cat;
  a.rng = s->rng_state; a.site = s->site; a.p_drop = s->p_drop;
  a.d_enc = s->d_enc; a.d_enc_f32 = s->d_enc_f32; a.wq = s->wq; a.wk = s->wk; a.wv = s->wv; a.wo = s->wo; a.ws = s->ws;
  a.dpre_s = s->dpre_s; a.dout = s->dout; a.dqkv = s->dqkv; a.dwn = s->dwn; a.dbn = s->dbn; a.dwv3 = s->dwv3;
  if (!bwd) {
    if (!s->pack || !s->wn || !s->bn || !s->wv3 || !s->bo || !s->bs) { stj_set_error("stj_agent_enc_fwd: null weight pointer"); return STJ_EINVAL; }
    const bool any = s->s_nodes || s->s_qkv || s->s_att || s->s_pmask || s->s_cat, all = s->s_nodes && s->s_qkv && s->s_att && s->s_pmask && s->s_cat;
    if (any && !all) { stj_set_error("stj_agent_enc_fwd: the five saved tensors go together"); return STJ_EINVAL; }
  } else if (!s->d_enc || !s->wq || !s->wk || !s->wv || !s->wo || !s->ws || !s->dpre_s || !s->dout || !s->dqkv || !s->dwn || !s->dbn || !s->dwv3 ||
             !s->s_nodes || !s->s_qkv || !s->s_pmask) {
    stj_set_error("stj_agent_enc_bwd: null pointer"); return STJ_EINVAL;
  }
  return 0;
}

extern "C" int stj_agent_enc_fwd(const stj_agent_enc_args* s, hipStream_t stream) {
  agf::EncArgs a;
  const int rc = enc_args(s, a, false);
  if (rc) return rc > 0 ? STJ_OK : rc;
  if (s->dtype == STJ_BF16) return enc_launch<bf16, false>(a, stream);
  if (s->dtype == STJ_F16) return enc_launch<f16, false>(a, stream);
  return enc_launch<float, false>(a, stream);
}
extern "C" int stj_agent_enc_bwd(const stj_agent_enc_args* s, hipStream_t stream) {
  agf::EncArgs a;
  const int rc = enc_args(s, a, true);
  if (rc) return rc > 0 ? STJ_OK : rc;
  if (s->dtype == STJ_BF16) return enc_launch<bf16, true>(a, stream);
  if (s->dtype == STJ_F16) return enc_launch<f16, true>(a, stream);
  return enc_launch<float, true>(a, stream);
}

extern "C" int stj_agent_int_supported(int n_obs, int n_occ, int dtype) { return n_obs >= 0 && n_occ >= 0 && n_obs + n_occ == agf::NA && stj_is16(dtype); }

template <typename T, bool BWD> static int int_launch(const agf::IntArgs& a, hipStream_t stream) {
  using namespace agf;
  static PerDevice<int> attr_set;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)agent_int_attn_fwd_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)int_attn_fwd_lds<T>()) != hipSuccess ||
        hipFuncSetAttribute((const void*)agent_int_ffn_fwd_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)int_ffn_lds<T>()) != hipSuccess ||
        hipFuncSetAttribute((const void*)agent_int_ffn_bwd_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)int_ffn_lds<T>()) != hipSuccess ||
        hipFuncSetAttribute((const void*)agent_int_attn_bwd_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)int_attn_bwd_lds<T>()) != hipSuccess) {
      stj_set_error("stj_agent_int: cannot reserve %zu bytes of LDS", int_attn_bwd_lds<T>());
      return STJ_ELAUNCH;
    }
    attr_set = 1;
  }
  const int rows = a.B * NA;
  if (!BWD) {
    hipLaunchKernelGGL(agent_int_attn_fwd_kernel<T>, dim3(a.B * IH), dim3(512), int_attn_fwd_lds<T>(), stream, a);
    hipLaunchKernelGGL(agent_int_ffn_fwd_kernel<T>, dim3(a.B * IGeo<T>::NC), dim3(512), int_ffn_lds<T>(), stream, a);
    hipLaunchKernelGGL(agent_int_out_fwd_kernel<T>, dim3((rows + 3) / 4), dim3(256), 0, stream, a);
  } else {
    hipLaunchKernelGGL(agent_int_out_bwd_kernel<T>, dim3((rows + 15) / 16), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(agent_int_ffn_bwd_kernel<T>, dim3(a.B * IGeo<T>::NC), dim3(512), int_ffn_lds<T>(), stream, a);
    hipLaunchKernelGGL(agent_int_attn_bwd_kernel<T>, dim3(a.B * IH), dim3(512), int_attn_bwd_lds<T>(), stream, a);
  }
  return stj_check_launch(BWD ? "stj_agent_int_bwd" : "stj_agent_int_fwd");
}

static int int_args(const stj_agent_int_args* s, agf::IntArgs& a, bool bwd) {
  if (!s || !s->enc || !s->cmi) { stj_set_error("stj_agent_int: null pointer"); return STJ_EINVAL; }
  if (s->B <= 0) return 1;
  if (!stj_agent_int_supported(s->n_obs, s->n_occ, s->dtype)) { stj_set_error("stj_agent_int: 64 agents per scene and a 16-bit dtype only"); return STJ_EUNSUPPORTED; }
  if (!(s->p_drop >= 0.f && s->p_drop < 1.f)) { stj_set_error("stj_agent_int: need 0 <= p_drop < 1"); return STJ_EINVAL; }
  a = {};
  a.enc = s->enc; a.cmi = s->cmi; a.n_obs = s->n_obs; a.B = s->B; a.pack = s->pack; a.seg = s->seg;
  a.bo = s->bo; a.g1 = s->g1; a.be1 = s->be1; a.b1 = s->b1; a.b2 = s->b2; a.g2 = s->g2; a.be2 = s->be2;
  a.go = s->g_obs; a.beo = s->b_obs; a.gc = s->g_occ; a.bec = s->b_occ; a.key = s->key;
  a.v1acc = s->ws_v1; a.u2acc = s->ws_u2; a.n1 = s->s_n1; a.dn1acc = s->ws_dn1;
  a.s_concat = s->s_concat; a.s_qin = s->s_qin; a.s_q = s->s_q; a.s_k = s->s_k; a.s_v = s->s_v; a.s_att = s->s_att; a.s_v1 = s->s_v1;
  a.s_h = s->s_h; a.s_u2 = s->s_u2; a.s_out = s->s_out;
  a.rng = s->rng_state; a.site_a = s->site_a; a.site_1 = s->site_1; a.site_2 = s->site_2; a.p_drop = s->p_drop;
  a.dkey = s->dkey; a.wq = s->wq; a.wk = s->wk; a.wv = s->wv; a.wo = s->wo; a.w1 = s->w1; a.w2 = s->w2;
  a.d_enc = s->d_enc; a.dq = s->dq; a.dk = s->dk; a.dv = s->dv; a.dv1 = s->dv1; a.dpre1 = s->dpre1; a.dz2 = s->dz2;
  a.dseg = s->dseg; a.dg1 = s->dg1; a.dbe1 = s->dbe1; a.dg2 = s->dg2; a.dbe2 = s->dbe2; a.dgo = s->dg_obs; a.dbeo = s->db_obs; a.dgc = s->dg_occ; a.dbec = s->db_occ;
  const void* saves[] = {s->s_concat, s->s_qin, s->s_q, s->s_k, s->s_v, s->s_att, s->s_v1, s->s_n1, s->s_h, s->s_u2, s->s_out};
  int nset = 0;
  for (const void* q : saves) nset += q != nullptr;
  if (!s->seg || !s->g1 || !s->g2 || !s->g_obs || !s->g_occ) { stj_set_error("stj_agent_int: null parameter pointer"); return STJ_EINVAL; }
  if (!bwd) {
    if (!s->pack || !s->key || !s->bo || !s->be1 || !s->b1 || !s->b2 || !s->be2 || !s->b_obs || !s->b_occ || !s->ws_v1 || !s->ws_u2) { stj_set_error("stj_agent_int_fwd: null pointer"); return STJ_EINVAL; }
    if (nset != 0 && nset != 11) { stj_set_error("stj_agent_int_fwd: the eleven saved tensors go together"); return STJ_EINVAL; }
  } else {
    const void* need[] = {s->dkey, s->wq, s->wk, s->wv, s->wo, s->w1, s->w2, s->d_enc, s->dq, s->dk, s->dv, s->dv1, s->dpre1, s->dz2, s->dseg, s->dg1, s->dbe1,
                          s->dg2, s->dbe2, s->dg_obs, s->db_obs, s->dg_occ, s->db_occ, s->ws_dn1};
    for (const void* q : need)
      if (!q) { stj_set_error("stj_agent_int_bwd: null pointer"); return STJ_EINVAL; }
    if (nset != 11) { stj_set_error("stj_agent_int_bwd: needs the eleven saved tensors"); return STJ_EINVAL; }
  }
  return 0;
}
extern "C" int stj_agent_int_fwd(const stj_agent_int_args* s, hipStream_t stream) {
  agf::IntArgs a;
  const int rc = int_args(s, a, false);
  if (rc) return rc > 0 ? STJ_OK : rc;
  return s->dtype == STJ_BF16 ? int_launch<bf16, false>(a, stream) : int_launch<f16, false>(a, stream);
}
extern "C" int stj_agent_int_bwd(const stj_agent_int_args* s, hipStream_t stream) {
  agf::IntArgs a;
  const int rc = int_args(s, a, true);
  if (rc) return rc > 0 ? STJ_OK : rc;
  return s->dtype == STJ_BF16 ? int_launch<bf16, true>(a, stream) : int_launch<f16, true>(a, stream);
}
