// Fused FG-MSA attention core: reference FG_MSA.py:138-178 -- per sample b and group (head) g of 48 channels
//     S = 48^-1/2 q k^T + bias ,   bias[q, k] = sample(warp_attn_rel_table[:, :, g])(drow - off1[k], dcol - off0[k])      (:150-172)
//     a = softmax(S) v                                                                                                   (:173-176)
// as ONE kernel per direction: the [B, 8, HW, HW] logits / bias / probabilities never exist in HBM (the layer-by-layer path wrote
// S and the bias in f32 and P in the activation type: 3 x 17 MB per direction at B = 8, in 4 + 5 launches).  16-bit storage types;
// the f32 parity mode keeps the layer-by-layer kernels (its K / V / P tiles would not fit LDS in the backward kernel).
//
// Work layout (the scheme of xattn_fused.hip): one workgroup = 64 queries of one (b, g); a wave owns 16 queries; K and V of the group
// ([HW][48], HW = 64 or 256) sit in LDS next to the group's 31 x 31 table and the keys' offsets; S^T = K q^T on the accumulator layout
// (a lane holds 4 consecutive keys of its query, the 3 partner lanes the rest), the bilinear bias is sampled per (query, key) pair
// from the LDS table (zero pad + clamp rules of occu_metric.sample, common.h bil_setup), softmax by two xor-shuffles, O^T = V^T P^T
// with P^T chained in registers.  Backward recomputes S / P, chains dP^T = V dO^T and dq^T = K^T dS^T, sends the bias gradient to an
// LDS copy of the table gradient and (reduced over the wave's 16 queries by shuffles) to the keys' offset gradients, and -- the sums
// over QUERIES cross the waves -- computes dV = P^T dO, dK = dS^T q for the keys each wave owns from [query][.] tiles in LDS; per-tile
// partial sums are reduced by a second small kernel.
#include "common.h"
#include "chain48.h"

namespace fga {
using namespace chain;
constexpr int D = 48, TOK = 64, LDK = D + 4, LDT = D + 4;

struct Args {
  const void* q; const void* k; const void* v; const void* off; const float* table;
  void* a; float* lse;
  const void* da; void* dq; float* dkp; float* dvp; float* dtable; float* doff;
  int B, G, Hh, Ww;
  float scale;
};

// bilinear sample of the zero-padded table at the (query, key) displacement minus the key's offset; optionally its gradient pieces
struct BiasPt { Bil c; float tl, tr, bl, br; };
__device__ __forceinline__ float bias_at(const float* tbl, int TH, int TW, int qi, int qj, int ki, int kj, float off0, float off1, BiasPt& s) {
  const float x = (float)(qi - ki) - off1 + 1.f;
  const float y = (float)(qj - kj) - off0 + 1.f;
  s.c = bil_setup(x, y, TH + 2, TW + 2);
  s.tl = pad_at(tbl, TH, TW, 1, s.c.y0, s.c.x0); s.tr = pad_at(tbl, TH, TW, 1, s.c.y0, s.c.x0 + 1);
  s.bl = pad_at(tbl, TH, TW, 1, s.c.y0 + 1, s.c.x0); s.br = pad_at(tbl, TH, TW, 1, s.c.y0 + 1, s.c.x0 + 1);
  const float top = s.c.ax * (s.tr - s.tl) + s.tl, bot = s.c.ax * (s.br - s.bl) + s.bl;
  return s.c.ay * (bot - top) + top;
}

// K / V rows of group g: global [B, HW, C] (48 contiguous channels at column 48 g) -> LDS tiles [HW][LDK]
template <typename T>
__device__ __forceinline__ void load_kv(const T* k, const T* v, T* Kt, T* Vt, long long row0, int HW, int C, int g, int tid) {
  constexpr int VN = 8, CPR = D / VN;            // 16-byte pieces per row
  for (int i = tid; i < 2 * HW * CPR; i += 256) {
    const int t = i / (HW * CPR), rem = i % (HW * CPR), r = rem / CPR, c = (rem % CPR) * VN;
    const uint4 w = *reinterpret_cast<const uint4*>((t ? v : k) + (row0 + r) * C + D * g + c);
    uint2* d = reinterpret_cast<uint2*>((t ? Vt : Kt) + r * LDK + c);          // rows are 8-byte aligned (104 bytes)
    d[0] = make_uint2(w.x, w.y); d[1] = make_uint2(w.z, w.w);
  }
  for (int i = tid; i < 2 * HW * (LDK - D); i += 256) {
    const int t = i / (HW * (LDK - D)), rem = i % (HW * (LDK - D));
    stf((t ? Vt : Kt) + (rem / (LDK - D)) * LDK + D + rem % (LDK - D), 0.f);
  }
}

// the 8 logits (scaled product + sampled bias) a lane holds of key fragments 2s, 2s+1: v[e], keys 32 s + 16 (e / 4) + 4 g4 + e % 4
template <typename T, int NKF, bool GRAD>
__device__ __forceinline__ void logits8(float (&v)[8], BiasPt (&bp)[GRAD ? 8 : 1], const T* Kt, const HeadOp<T>& qop, const float* tbl, const float* offs,
                                        int TH, int TW, int s, int qi, int qj, float scale, int lane) {
  constexpr int Ww = NKF == 4 ? 8 : 16;          // square maps only (check()): 8 x 8 or 16 x 16
  const int g4 = lane >> 4;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  const f32x4 sa = k48_rows<T>(Kt, LDK, 32 * s, qop, lane, z), sb = k48_rows<T>(Kt, LDK, 32 * s + 16, qop, lane, z);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int key = 32 * s + 16 * (e >> 2) + 4 * g4 + (e & 3);
    const float2 o = *reinterpret_cast<const float2*>(offs + 2 * key);
    v[e] = (e < 4 ? sa[e & 3] : sb[e & 3]) * scale + bias_at(tbl, TH, TW, qi, qj, key / Ww, key % Ww, o.x, o.y, bp[GRAD ? e : 0]);
  }
}

template <typename T, int NKF> struct Lds {
  static constexpr int HW = 16 * NKF;
  static constexpr int LDP = HW + 8;
  static int fwd_bytes(int tt) { return 2 * HW * LDK * (int)sizeof(T) + (tt + 2 * HW) * 4; }
  static int bwd_bytes(int tt) { return (2 * HW * LDK + 2 * TOK * LDP + 2 * TOK * LDT) * (int)sizeof(T) + (2 * tt + 4 * HW) * 4; }
};

// =====================================================================================================================
// Forward: the keys are visited 32 at a time with a running maximum / sum (the logits of a query never exist all at once: 16 x 4 live
// accumulator registers per lane instead of 256 x 4 / 4).
template <typename T, int NKF>
__global__ __launch_bounds__(256, 2) void fgattn_fwd_kernel(Args p) {
  constexpr int HW = 16 * NKF, Ww = NKF == 4 ? 8 : 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char fg_smem[];
  const int TH = 2 * p.Hh - 1, TW = 2 * p.Ww - 1, TT = TH * TW, C = p.G * D;
  T* Kt = reinterpret_cast<T*>(fg_smem);
  T* Vt = Kt + HW * LDK;
  float* tbl = reinterpret_cast<float*>(Vt + HW * LDK);
  float* offs = tbl + TT;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g4 = lane >> 4, ln = lane & 15;
  constexpr int tiles = HW / TOK;
  const int tile = blockIdx.x % tiles, g = (blockIdx.x / tiles) % p.G, b = blockIdx.x / (tiles * p.G);
  const long long row0 = (long long)b * HW;
  const T* q = reinterpret_cast<const T*>(p.q);
  load_kv<T>(reinterpret_cast<const T*>(p.k), reinterpret_cast<const T*>(p.v), Kt, Vt, row0, HW, C, g, tid);
  for (int i = tid; i < TT; i += 256) tbl[i] = p.table[i * p.G + g];
  {
    const T* o = reinterpret_cast<const T*>(p.off) + ((long long)b * p.G + g) * HW * 2;
    for (int i = tid; i < 2 * HW; i += 256) offs[i] = ldf(o + i);
  }
  const int tok = tile * TOK + wv * 16 + ln;
  HeadOp<T> qop;
  qop.from_row(q + (row0 + tok) * C + D * g, lane);
  __syncthreads();
  f32x4 o[3];
#pragma unroll
  for (int jd = 0; jd < 3; ++jd) o[jd] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, l = 0.f;
#pragma unroll 1
  for (int s = 0; s < NKF / 2; ++s) {
    float v[8];
    BiasPt bp[1];
    logits8<T, NKF, false>(v, bp, Kt, qop, tbl, offs, TH, TW, s, tok / Ww, tok % Ww, p.scale, lane);
    float mx = v[0];
#pragma unroll
    for (int e = 1; e < 8; ++e) mx = fmaxf(mx, v[e]);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64)); mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mn = fmaxf(m, mx), alpha = __expf(m - mn);
    m = mn;
    l *= alpha;
    f32x4 e2[2];
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float x = __expf(v[e] - mn); e2[e >> 2][e & 3] = x; l += x; }
    const typename Ch<T>::Frag pf = Ch<T>::from_acc(e2);
#pragma unroll
    for (int jd = 0; jd < 3; ++jd) {
      o[jd] *= alpha;
      o[jd] = Mma<T>::mma(Ch<T>::ldA_tr(Vt, LDK, 16 * jd, 32 * s, lane), pf, o[jd]);
    }
  }
  l += __shfl_xor(l, 16, 64); l += __shfl_xor(l, 32, 64);
  const float inv = 1.f / l;
  T* a = reinterpret_cast<T*>(p.a) + (row0 + tok) * C + D * g + 4 * g4;
#pragma unroll
  for (int jd = 0; jd < 3; ++jd) { const float w[4] = {o[jd][0] * inv, o[jd][1] * inv, o[jd][2] * inv, o[jd][3] * inv}; st4(a + 16 * jd, w); }
  if (p.lse != nullptr && g4 == 0) p.lse[((long long)b * p.G + g) * HW + tok] = m + __logf(l);
}

// =====================================================================================================================
// Backward: P = exp(logit - lse) from the forward's log-sum-exp, the softmax's row term sum_k P dP = dO . O from the forward output.
template <typename T, int NKF>
__global__ __launch_bounds__(256, 1) void fgattn_bwd_kernel(Args p) {
  constexpr int HW = 16 * NKF, LDP = HW + 8, MW = NKF / 4, Ww = NKF == 4 ? 8 : 16;       // MW: 16-key fragments each wave owns for dK / dV
  extern __shared__ __attribute__((aligned(16))) unsigned char fg_smem[];
  const int TH = 2 * p.Hh - 1, TW = 2 * p.Ww - 1, TT = TH * TW, C = p.G * D;
  T* Kt = reinterpret_cast<T*>(fg_smem);
  T* Vt = Kt + HW * LDK;
  T* PT = Vt + HW * LDK;                   // P   [64 queries][LDP]
  T* ST = PT + TOK * LDP;                  // dS  [64 queries][LDP]   (times 48^-1/2)
  T* OT = ST + TOK * LDP;                  // dO  [64 queries][LDT]
  T* QT = OT + TOK * LDT;                  // q   [64 queries][LDT]
  float* tbl = reinterpret_cast<float*>(QT + TOK * LDT);
  float* dtb = tbl + TT;
  float* offs = dtb + TT;                  // [HW][2]
  float* doffs = offs + 2 * HW;            // [HW][2]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g4 = lane >> 4, ln = lane & 15;
  constexpr int tiles = HW / TOK;
  const int tile = blockIdx.x % tiles, g = (blockIdx.x / tiles) % p.G, b = blockIdx.x / (tiles * p.G);
  const long long row0 = (long long)b * HW;
  load_kv<T>(reinterpret_cast<const T*>(p.k), reinterpret_cast<const T*>(p.v), Kt, Vt, row0, HW, C, g, tid);
  for (int i = tid; i < TT; i += 256) { tbl[i] = p.table[i * p.G + g]; dtb[i] = 0.f; }
  {
    const T* o = reinterpret_cast<const T*>(p.off) + ((long long)b * p.G + g) * HW * 2;
    for (int i = tid; i < 2 * HW; i += 256) { offs[i] = ldf(o + i); doffs[i] = 0.f; }
  }
  const int tok = tile * TOK + wv * 16 + ln;
  const int qi = tok / Ww, qj = tok % Ww;
  const T* qrow = reinterpret_cast<const T*>(p.q) + (row0 + tok) * C + D * g;
  const T* drow = reinterpret_cast<const T*>(p.da) + (row0 + tok) * C + D * g;
  const T* arow = reinterpret_cast<const T*>(p.a) + (row0 + tok) * C + D * g;
  HeadOp<T> qop, dop;
  qop.from_row(qrow, lane);
  dop.from_row(drow, lane);
  float dsum = 0.f;
  {   // dO and q of the wave's queries -> tiles [query][head column]; dO . O
    T* orow = OT + (16 * wv + ln) * LDT + 4 * g4;
    T* qtr = QT + (16 * wv + ln) * LDT + 4 * g4;
#pragma unroll
    for (int jd = 0; jd < 3; ++jd) {
      float a4[4], b4[4], c4[4];
      ld4(drow + 16 * jd + 4 * g4, a4); st4(orow + 16 * jd, a4);
      ld4(qrow + 16 * jd + 4 * g4, b4); st4(qtr + 16 * jd, b4);
      ld4(arow + 16 * jd + 4 * g4, c4);
#pragma unroll
      for (int r = 0; r < 4; ++r) dsum += a4[r] * c4[r];
    }
  }
  dsum += __shfl_xor(dsum, 16, 64); dsum += __shfl_xor(dsum, 32, 64);
  const float lse = p.lse[((long long)b * p.G + g) * HW + tok];
  __syncthreads();
  f32x4 dq[3];
#pragma unroll
  for (int jd = 0; jd < 3; ++jd) dq[jd] = (f32x4){0.f, 0.f, 0.f, 0.f};
  T* prow = PT + (16 * wv + ln) * LDP + 4 * g4;
  T* srow = ST + (16 * wv + ln) * LDP + 4 * g4;
#pragma unroll 1
  for (int s = 0; s < NKF / 2; ++s) {
    float v[8];
    BiasPt bp[8];
    logits8<T, NKF, true>(v, bp, Kt, qop, tbl, offs, TH, TW, s, qi, qj, p.scale, lane);
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const f32x4 dPa = k48_rows<T>(Vt, LDK, 32 * s, dop, lane, z), dPb = k48_rows<T>(Vt, LDK, 32 * s + 16, dop, lane, z);
    f32x4 ds2[2];
    float pv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int key = 32 * s + 16 * (e >> 2) + 4 * g4 + (e & 3);
      const float pr = __expf(v[e] - lse);
      pv[e] = pr;
      const float go = pr * ((e < 4 ? dPa[e & 3] : dPb[e & 3]) - dsum);        // d(logit) = d(bias): the bias enters the logit with weight 1
      ds2[e >> 2][e & 3] = go * p.scale;
      // bias backward (the gradient rules of stj_fg_bias_bwd: TF's clip gradients, zero outside the padded table)
      const BiasPt& c = bp[e];
      const float top = c.c.ax * (c.tr - c.tl) + c.tl, bot = c.c.ax * (c.br - c.bl) + c.bl;
      float d0 = c.c.gy ? -go * (bot - top) : 0.f;
      float d1 = c.c.gx ? -go * (c.c.ay * (c.br - c.bl) + (1.f - c.c.ay) * (c.tr - c.tl)) : 0.f;
      if (go != 0.f) {
        const float w[4] = {(1.f - c.c.ay) * (1.f - c.c.ax), (1.f - c.c.ay) * c.c.ax, c.c.ay * (1.f - c.c.ax), c.c.ay * c.c.ax};
        const int yy[4] = {c.c.y0, c.c.y0, c.c.y0 + 1, c.c.y0 + 1}, xx[4] = {c.c.x0, c.c.x0 + 1, c.c.x0, c.c.x0 + 1};
#pragma unroll
        for (int t = 0; t < 4; ++t)
          if (yy[t] >= 1 && yy[t] <= TH && xx[t] >= 1 && xx[t] <= TW && w[t] != 0.f) atomicAdd(&dtb[(yy[t] - 1) * TW + (xx[t] - 1)], go * w[t]);
      }
      // the key's offset gradient: sum over the wave's 16 queries (lanes of equal g4), one LDS add per wave
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) { d0 += __shfl_xor(d0, o, 64); d1 += __shfl_xor(d1, o, 64); }
      if (ln == 0) { atomicAdd(&doffs[2 * key], d0); atomicAdd(&doffs[2 * key + 1], d1); }
    }
    st4(prow + 32 * s, pv); st4(prow + 32 * s + 16, pv + 4);
    { const float x0[4] = {ds2[0][0], ds2[0][1], ds2[0][2], ds2[0][3]}, x1[4] = {ds2[1][0], ds2[1][1], ds2[1][2], ds2[1][3]};
      st4(srow + 32 * s, x0); st4(srow + 32 * s + 16, x1); }
    const typename Ch<T>::Frag sf = Ch<T>::from_acc(ds2);            // dq^T += K^T dS^T (dS carries the 48^-1/2)
#pragma unroll
    for (int jd = 0; jd < 3; ++jd) dq[jd] = Mma<T>::mma(Ch<T>::ldA_tr(Kt, LDK, 16 * jd, 32 * s, lane), sf, dq[jd]);
  }
  {
    T* dqo = reinterpret_cast<T*>(p.dq) + (row0 + tok) * C + D * g + 4 * g4;
#pragma unroll
    for (int jd = 0; jd < 3; ++jd) { const float w[4] = {dq[jd][0], dq[jd][1], dq[jd][2], dq[jd][3]}; st4(dqo + 16 * jd, w); }
  }
  __syncthreads();
  {   // dV = P^T dO, dK = dS^T q for the MW x 16 keys this wave owns: m = key, n = head column, k = the tile's 64 queries
    f32x4 dv[MW][3], dk[MW][3];
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
      for (int jd = 0; jd < 3; ++jd) { dv[m][jd] = (f32x4){0.f, 0.f, 0.f, 0.f}; dk[m][jd] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int s = 0; s < TOK / 32; ++s) {
      typename Mma<T>::Frag bo[3], bq[3];
#pragma unroll
      for (int jd = 0; jd < 3; ++jd) { bo[jd] = Mma<T>::load_tr(OT, LDT, 16 * jd, 32 * s, lane); bq[jd] = Mma<T>::load_tr(QT, LDT, 16 * jd, 32 * s, lane); }
#pragma unroll
      for (int m = 0; m < MW; ++m) {
        const typename Mma<T>::Frag ap = Mma<T>::load_tr(PT, LDP, 16 * (MW * wv + m), 32 * s, lane);
        const typename Mma<T>::Frag as = Mma<T>::load_tr(ST, LDP, 16 * (MW * wv + m), 32 * s, lane);
#pragma unroll
        for (int jd = 0; jd < 3; ++jd) { dv[m][jd] = Mma<T>::mma(ap, bo[jd], dv[m][jd]); dk[m][jd] = Mma<T>::mma(as, bq[jd], dk[m][jd]); }
      }
    }
    const long long pb = ((((long long)b * p.G + g) * tiles + tile) * HW) * D;
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
      for (int jd = 0; jd < 3; ++jd)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long long o = pb + (long long)(16 * (MW * wv + m) + 4 * g4 + r) * D + 16 * jd + ln;
          p.dvp[o] = dv[m][jd][r];
          p.dkp[o] = dk[m][jd][r];
        }
  }
  // table / offset gradients of this tile
  for (int i = tid; i < TT; i += 256)
    if (dtb[i] != 0.f) atomicAdd(p.dtable + i * p.G + g, dtb[i]);
  float* dof = p.doff + ((long long)b * p.G + g) * HW * 2;
  for (int i = tid; i < 2 * HW; i += 256) {
    if (tiles == 1) dof[i] = doffs[i];
    else atomicAdd(dof + i, doffs[i]);
  }
}

// dk / dv [B, HW, C] (activation dtype) = sum over the query tiles of the partials [B*G][tiles][HW][48] f32
template <typename T>
__global__ __launch_bounds__(256) void fgattn_dkv_reduce_kernel(const float* dkp, const float* dvp, T* dk, T* dv, int B, int G, int HW, int tiles) {
  const long long total = (long long)B * HW * G * D;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < 2 * total; i += gridDim.x * 256ll) {
    const bool second = i >= total;
    const long long j = second ? i - total : i;
    const int d = (int)(j % D), g = (int)((j / D) % G), key = (int)((j / ((long long)D * G)) % HW);
    const long long b = j / ((long long)D * G * HW);
    const float* src = (second ? dvp : dkp) + ((b * G + g) * tiles * HW + key) * D + d;
    float a = 0.f;
    for (int t = 0; t < tiles; ++t) a += src[(long long)t * HW * D];
    stf((second ? dv : dk) + j, a);
  }
}

template <typename T, int NKF> static int launch(bool bwd, const Args& a, hipStream_t st) {
  const int tt = (2 * a.Hh - 1) * (2 * a.Ww - 1);
  const int lds = bwd ? Lds<T, NKF>::bwd_bytes(tt) : Lds<T, NKF>::fwd_bytes(tt);
  const void* fn = bwd ? (const void*)fgattn_bwd_kernel<T, NKF> : (const void*)fgattn_fwd_kernel<T, NKF>;
  if (lds > 160 * 1024 || hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
    stj_set_error("fg_attn: cannot reserve %d bytes of LDS", lds); return STJ_ELAUNCH;
  }
  const dim3 grid((unsigned)(a.B * a.G * (16 * NKF / TOK)));
  if (bwd) hipLaunchKernelGGL((fgattn_bwd_kernel<T, NKF>), grid, dim3(256), lds, st, a);
  else hipLaunchKernelGGL((fgattn_fwd_kernel<T, NKF>), grid, dim3(256), lds, st, a);
  return stj_check_launch(bwd ? "stj_fg_attn_bwd" : "stj_fg_attn_fwd");
}
template <typename T> static int dispatch(bool bwd, const Args& a, hipStream_t st) {
  if (a.Hh == 16 && a.Ww == 16) return launch<T, 16>(bwd, a, st);
  if (a.Hh == 8 && a.Ww == 8) return launch<T, 4>(bwd, a, st);
  stj_set_error("fg_attn: the map must be 8 x 8 or 16 x 16 (got %d x %d)", a.Hh, a.Ww);
  return STJ_EUNSUPPORTED;
}
static int check(int B, int G, int Hh, int Ww, int dtype) {
  if (B <= 0) return 1;
  if (!stj_is16(dtype)) { stj_set_error("fg_attn: 16-bit storage types only (f32 keeps the layer-by-layer kernels)"); return STJ_EUNSUPPORTED; }
  if (G <= 0 || Hh <= 0 || Ww <= 0) { stj_set_error("fg_attn: bad geometry"); return STJ_EINVAL; }
  return 0;
}
}  // namespace fga

// a [B,HW,G*48] = softmax(scale q k^T + sampled bias) v per sample and group; q, k, v [B,HW,G*48], off [B,G,HW,2] (activation dtype),
// table f32 [2Hh-1, 2Ww-1, G].  lse f32 [B,G,HW] (the rows' log-sum-exp, for the backward) or NULL.  Hh = Ww in {8, 16}; dtype STJ_BF16 / STJ_F16.
extern "C" int stj_fg_attn_fwd(const void* q, const void* k, const void* v, const void* off, const float* table, void* a, float* lse, int B, int G,
                               int Hh, int Ww, float scale, int dtype, hipStream_t stream) {
  const int c = fga::check(B, G, Hh, Ww, dtype);
  if (c) return c > 0 ? STJ_OK : c;
  fga::Args p = {};
  p.q = q; p.k = k; p.v = v; p.off = off; p.table = table; p.a = a; p.lse = lse; p.B = B; p.G = G; p.Hh = Hh; p.Ww = Ww; p.scale = scale;
  return dtype == STJ_BF16 ? fga::dispatch<bf16>(false, p, stream) : fga::dispatch<f16>(false, p, stream);
}
// bytes of each of the two partial-sum workspaces (dkp, dvp) of stj_fg_attn_bwd
extern "C" long long stj_fg_attn_bwd_workspace_bytes(int B, int G, int Hh, int Ww) {
  const long long HW = (long long)Hh * Ww;
  return (long long)B * G * (HW / fga::TOK) * HW * fga::D * 4;
}
// Backward: a, lse as the forward wrote them, da [B,HW,G*48] -> dq, dk, dv (same shape, written; dk / dv via the f32 per-tile partials
// dkp / dvp and a second launch), dtable f32 [2Hh-1,2Ww-1,G] "+=", doff f32 [B,G,HW,2]: written when Hh = 8, "+=" (caller zeroes it)
// when Hh = 16.
extern "C" int stj_fg_attn_bwd(const void* q, const void* k, const void* v, const void* off, const float* table, const void* a, const float* lse,
                               const void* da, void* dq, void* dk, void* dv, float* dkp, float* dvp, float* dtable, float* doff, int B, int G,
                               int Hh, int Ww, float scale, int dtype, hipStream_t stream) {
  const int c = fga::check(B, G, Hh, Ww, dtype);
  if (c) return c > 0 ? STJ_OK : c;
  fga::Args p = {};
  p.q = q; p.k = k; p.v = v; p.off = off; p.table = table; p.a = const_cast<void*>(a); p.lse = const_cast<float*>(lse); p.da = da; p.dq = dq;
  p.dkp = dkp; p.dvp = dvp; p.dtable = dtable; p.doff = doff;
  p.B = B; p.G = G; p.Hh = Hh; p.Ww = Ww; p.scale = scale;
  int rc = dtype == STJ_BF16 ? fga::dispatch<bf16>(true, p, stream) : fga::dispatch<f16>(true, p, stream);
  if (rc != STJ_OK) return rc;
  const int HW = Hh * Ww, tiles = HW / fga::TOK;
  const long long n = 2ll * B * HW * G * fga::D;
  const int grid = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
  if (dtype == STJ_BF16) hipLaunchKernelGGL(fga::fgattn_dkv_reduce_kernel<bf16>, dim3(grid), dim3(256), 0, stream, dkp, dvp, (bf16*)dk, (bf16*)dv, B, G, HW, tiles);
  else hipLaunchKernelGGL(fga::fgattn_dkv_reduce_kernel<f16>, dim3(grid), dim3(256), 0, stream, dkp, dvp, (f16*)dk, (f16*)dv, B, G, HW, tiles);
  return stj_check_launch("stj_fg_attn_dkv_reduce");
}
