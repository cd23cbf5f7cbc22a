// Fused FG-MSA attention core: reference FG_MSA.py:138-178 -- per sample b and group (head) g of 48 channels
//     S = 48^-1/2 q k^T + bias ,   bias[q, k] = sample(warp_attn_rel_table[:, :, g])(drow - off1[k], dcol - off0[k])      (:150-172)
//     a = softmax(S) v                                                                                                   (:173-176)
// as ONE kernel per direction: the [B, 8, HW, HW] logits / bias / probabilities never exist in HBM (the layer-by-layer path wrote
// S and the bias in f32 and P in the activation type: 3 x 17 MB per direction at B = 8, in 4 + 5 launches).  All three storage types
// run the same code (round 5: the f32 parity mode too -- exact-f32 MFMA through Mma<float> / chain48.h, 16-deep k-steps; its backward
// takes 16 instead of 64 queries per workgroup so that K, V, P and dS fit the 160 KB of LDS: the oracle gate covers this kernel).
//
// Work layout (the chained-operand scheme of xattn_fused.hip): a workgroup = 32 (forward) or 64 (backward) queries of one (b, g); a wave
// = 16 queries against a quarter of the keys (Split<>; 4 waves per SIMD at B = 8).  K and V of the group ([HW][48], HW = 64 or 256) sit
// in LDS next to the group's 31 x 31 table and the keys' offsets.  S^T = K q^T lands on the accumulator layout (a lane holds 4
// consecutive keys of its query, the 3 partner lanes the rest); the bilinear bias is sampled per (query, key) pair from the LDS table
// (zero pad + clamp rules of occu_metric.sample: common.h bil_setup); the keys are visited 32 at a time with a running maximum / sum
// (the logits of a query never exist all at once); O^T = V^T P^T with P^T chained in registers; the key slices merge through LDS.
// Backward: P = exp(logit - lse) from the forward's log-sum-exp, sum_k P dP = dO . O from the forward output; dP^T = V dO^T and
// dq^T = K^T dS^T are chained; the bias gradient goes to an LDS copy of the table gradient (neighbouring lanes share cells: see there)
// and, summed over the wave's 16 queries by DPP, to the keys' offset gradients; the sums over QUERIES cross the waves: dV = P^T dO,
// dK = dS^T q for the keys each wave owns come from [query][.] tiles in LDS, as per-tile f32 partials that a second small kernel sums.
// Measured alone at B = 8, 16 x 16, bf16: forward 19 us, backward 47 + 10 us (layer by layer: ~55 + ~100 us in 4 + 5 launches).
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
template <typename T, int NT>
__device__ __forceinline__ void load_kv(const T* k, const T* v, T* Kt, T* Vt, long long row0, int HW, int C, int g, int tid) {
  constexpr int VN = 16 / (int)sizeof(T), CPR = D / VN;            // 16-byte pieces per row
  for (int i = tid; i < 2 * HW * CPR; i += NT) {
    const int t = i / (HW * CPR), rem = i % (HW * CPR), r = rem / CPR, c = (rem % CPR) * VN;
    const uint4 w = *reinterpret_cast<const uint4*>((t ? v : k) + (row0 + r) * C + D * g + c);
    uint2* d = reinterpret_cast<uint2*>((t ? Vt : Kt) + r * LDK + c);          // rows are 8-byte aligned (104 bytes in the 16-bit types)
    d[0] = make_uint2(w.x, w.y); d[1] = make_uint2(w.z, w.w);
  }
  for (int i = tid; i < 2 * HW * (LDK - D); i += NT) {
    const int t = i / (HW * (LDK - D)), rem = i % (HW * (LDK - D));
    stf((t ? Vt : Kt) + (rem / (LDK - D)) * LDK + D + rem % (LDK - D), 0.f);
  }
}

// the 8 logits (scaled product + sampled bias) a lane holds of key fragments 2s, 2s+1: v[e], keys 32 s + 16 (e / 4) + 4 g4 + e % 4
template <typename T, int NKF>
__device__ __forceinline__ void logits8(float (&v)[8], const T* Kt, const HeadOp<T>& qop, const float* tbl, const float* offs,
                                        int TH, int TW, int s, int qi, int qj, float scale, int lane) {
  constexpr int Ww = NKF == 4 ? 8 : 16;          // square maps only (check()): 8 x 8 or 16 x 16
  const int g4 = lane >> 4;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  const f32x4 sa = k48_rows<T>(Kt, LDK, 32 * s, qop, lane, z), sb = k48_rows<T>(Kt, LDK, 32 * s + 16, qop, lane, z);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int key = 32 * s + 16 * (e >> 2) + 4 * g4 + (e & 3);
    const float2 o = *reinterpret_cast<const float2*>(offs + 2 * key);
    BiasPt bp;
    v[e] = (e < 4 ? sa[e & 3] : sb[e & 3]) * scale + bias_at(tbl, TH, TW, qi, qj, key / Ww, key % Ww, o.x, o.y, bp);
  }
}

// The waves of a workgroup: QG groups of 16 queries x KS slices of the keys (a wave = 16 queries against HW / KS keys).  One wave per
// 16 queries against all 256 keys left the chip at one wave per SIMD (8 x 8 x 16 waves at B = 8) with every LDS / transcendental
// latency exposed: 35 us forward, 111 us backward; the key slices of one query group merge through LDS.
template <int NKF, bool BWD, typename T = bf16> struct Split {
  static constexpr int KS = NKF == 16 ? 4 : 1;
  // (f32 backward on the 16 x 16 map: ONE query group, or P / dS [queries][256 keys] do not fit LDS next to K and V)
  static constexpr int QG = NKF == 16 ? (BWD ? (sizeof(T) == 4 ? 1 : 4) : 2) : 4;
  static constexpr int NT = 64 * QG * KS;
  static constexpr int TQ = 16 * QG;               // queries per workgroup
  static constexpr int IT = NKF / 2 / KS;          // 32-key steps per wave
};
template <typename T, int NKF> struct Lds {
  static constexpr int HW = 16 * NKF;
  static constexpr int LDP = HW + 8;
  static constexpr int TQB = Split<NKF, true, T>::TQ;
  static int fwd_bytes(int tt) { return 2 * HW * LDK * (int)sizeof(T) + (tt + 2 * HW) * 4; }
  static int bwd_bytes(int tt) { return (2 * HW * LDK + 2 * TQB * LDP + 2 * TQB * LDT) * (int)sizeof(T) + (2 * tt + 4 * HW) * 4; }
};

// =====================================================================================================================
// Forward: the keys are visited 32 at a time with a running maximum / sum (the logits of a query never exist all at once).
template <typename T, int NKF>
__global__ __launch_bounds__((Split<NKF, false, T>::NT)) void fgattn_fwd_kernel(Args p) {
  typedef Split<NKF, false, T> W;
  constexpr int ND = Ch<T>::ND;
  constexpr int HW = 16 * NKF, Ww = NKF == 4 ? 8 : 16, NT = W::NT, QG = W::QG, KS = W::KS;
  extern __shared__ __attribute__((aligned(16))) unsigned char fg_smem[];
  const int TH = 2 * p.Hh - 1, TW = 2 * p.Ww - 1, TT = TH * TW, C = p.G * D;
  T* Kt = reinterpret_cast<T*>(fg_smem);
  T* Vt = Kt + HW * LDK;
  float* tbl = reinterpret_cast<float*>(Vt + HW * LDK);
  float* offs = tbl + TT;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g4 = lane >> 4, ln = lane & 15;
  const int qg = wv % QG, kq = wv / QG;
  constexpr int tiles = HW / W::TQ;
  const int tile = blockIdx.x % tiles, g = (blockIdx.x / tiles) % p.G, b = blockIdx.x / (tiles * p.G);
  const long long row0 = (long long)b * HW;
  const T* q = reinterpret_cast<const T*>(p.q);
  load_kv<T, NT>(reinterpret_cast<const T*>(p.k), reinterpret_cast<const T*>(p.v), Kt, Vt, row0, HW, C, g, tid);
  for (int i = tid; i < TT; i += NT) tbl[i] = p.table[i * p.G + g];
  {
    const T* o = reinterpret_cast<const T*>(p.off) + ((long long)b * p.G + g) * HW * 2;
    for (int i = tid; i < 2 * HW; i += NT) offs[i] = ldf(o + i);
  }
  const int tok = tile * W::TQ + qg * 16 + ln;
  HeadOp<T> qop;
  qop.from_row(q + (row0 + tok) * C + D * g, lane);
  __syncthreads();
  f32x4 o[3];
#pragma unroll
  for (int jd = 0; jd < 3; ++jd) o[jd] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, l = 0.f;
#pragma unroll 1
  for (int s = kq * W::IT; s < (kq + 1) * W::IT; ++s) {
    float v[8];
    logits8<T, NKF>(v, Kt, qop, tbl, offs, TH, TW, s, tok / Ww, tok % Ww, p.scale, lane);
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
#pragma unroll
    for (int jd = 0; jd < 3; ++jd) o[jd] *= alpha;
#pragma unroll
    for (int hf = 0; hf < 2 / ND; ++hf) {            // one 32-deep k-step (16-bit) or two 16-deep ones (f32) over the 32 keys
      const typename Ch<T>::Frag pf = Ch<T>::from_acc(&e2[hf * ND]);
#pragma unroll
      for (int jd = 0; jd < 3; ++jd) o[jd] = Mma<T>::mma(Ch<T>::ldA_tr(Vt, LDK, 16 * jd, 32 * s + 16 * hf, lane), pf, o[jd]);
    }
  }
  l += __shfl_xor(l, 16, 64); l += __shfl_xor(l, 32, 64);
  if constexpr (KS > 1) {          // merge the key slices of a query group through LDS (over the K / V tiles, which are done with)
    __syncthreads();
    float* mb = reinterpret_cast<float*>(fg_smem);                       // [KS][QG][14][64]
    float* my = mb + ((kq * QG + qg) * 14) * 64 + lane;
    my[0] = m; my[64] = l;
#pragma unroll
    for (int jd = 0; jd < 3; ++jd)
#pragma unroll
      for (int r = 0; r < 4; ++r) my[(2 + 4 * jd + r) * 64] = o[jd][r];
    __syncthreads();
    if (kq != 0) return;
    float M = m;
#pragma unroll
    for (int i = 1; i < KS; ++i) M = fmaxf(M, mb[((i * QG + qg) * 14) * 64 + lane]);
    l = 0.f;
#pragma unroll
    for (int jd = 0; jd < 3; ++jd) o[jd] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      const float* src = mb + ((i * QG + qg) * 14) * 64 + lane;
      const float w = __expf(src[0] - M);
      l += src[64] * w;
#pragma unroll
      for (int jd = 0; jd < 3; ++jd)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[jd][r] += src[(2 + 4 * jd + r) * 64] * w;
    }
    m = M;
  }
  const float inv = 1.f / l;
  T* a = reinterpret_cast<T*>(p.a) + (row0 + tok) * C + D * g + 4 * g4;
#pragma unroll
  for (int jd = 0; jd < 3; ++jd) { const float w[4] = {o[jd][0] * inv, o[jd][1] * inv, o[jd][2] * inv, o[jd][3] * inv}; st4(a + 16 * jd, w); }
  if (p.lse != nullptr && g4 == 0) p.lse[((long long)b * p.G + g) * HW + tok] = m + __logf(l);
}

// =====================================================================================================================
// Backward: P = exp(logit - lse) from the forward's log-sum-exp, the softmax's row term sum_k P dP = dO . O from the forward output.
template <typename T, int NKF>
__global__ __launch_bounds__((Split<NKF, true, T>::NT)) void fgattn_bwd_kernel(Args p) {
  typedef Split<NKF, true, T> W;
  constexpr int HW = 16 * NKF, LDP = HW + 8, Ww = NKF == 4 ? 8 : 16, NT = W::NT, QG = W::QG, KS = W::KS, TQ = W::TQ;
  constexpr int ND = Ch<T>::ND, KSTEP = Mma<T>::KSTEP;
  constexpr int MW = NKF / (QG * KS);          // 16-key fragments each wave owns for dK / dV
  static_assert(MW >= 1 && MW * QG * KS == NKF && TQ % KSTEP == 0, "the dK / dV contraction runs over the workgroup's queries, every key owned once");
  extern __shared__ __attribute__((aligned(16))) unsigned char fg_smem[];
  const int TH = 2 * p.Hh - 1, TW = 2 * p.Ww - 1, TT = TH * TW, C = p.G * D;
  T* Kt = reinterpret_cast<T*>(fg_smem);
  T* Vt = Kt + HW * LDK;
  T* PT = Vt + HW * LDK;                   // P   [64 queries][LDP]
  T* ST = PT + TQ * LDP;                   // dS  [64 queries][LDP]   (times 48^-1/2)
  T* OT = ST + TQ * LDP;                   // dO  [64 queries][LDT]
  T* QT = OT + TQ * LDT;                   // q   [64 queries][LDT]
  float* tbl = reinterpret_cast<float*>(QT + TQ * LDT);
  float* dtb = tbl + TT;
  float* offs = dtb + TT;                  // [HW][2]
  float* doffs = offs + 2 * HW;            // [HW][2]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g4 = lane >> 4, ln = lane & 15;
  const int qg = wv % QG, kq = wv / QG;
  constexpr int tiles = HW / TQ;
  const int tile = blockIdx.x % tiles, g = (blockIdx.x / tiles) % p.G, b = blockIdx.x / (tiles * p.G);
  const long long row0 = (long long)b * HW;
  load_kv<T, NT>(reinterpret_cast<const T*>(p.k), reinterpret_cast<const T*>(p.v), Kt, Vt, row0, HW, C, g, tid);
  for (int i = tid; i < TT; i += NT) { tbl[i] = p.table[i * p.G + g]; dtb[i] = 0.f; }
  {
    const T* o = reinterpret_cast<const T*>(p.off) + ((long long)b * p.G + g) * HW * 2;
    for (int i = tid; i < 2 * HW; i += NT) { offs[i] = ldf(o + i); doffs[i] = 0.f; }
  }
  const int tok = tile * TQ + qg * 16 + ln;
  const int qi = tok / Ww, qj = tok % Ww;
  const T* qrow = reinterpret_cast<const T*>(p.q) + (row0 + tok) * C + D * g;
  const T* drow = reinterpret_cast<const T*>(p.da) + (row0 + tok) * C + D * g;
  const T* arow = reinterpret_cast<const T*>(p.a) + (row0 + tok) * C + D * g;
  HeadOp<T> qop, dop;
  qop.from_row(qrow, lane);
  dop.from_row(drow, lane);
  float dsum = 0.f;
  {   // dO . O; dO and q of the group's queries -> tiles [query][head column] (by the group's first key slice)
    T* orow = OT + (16 * qg + ln) * LDT + 4 * g4;
    T* qtr = QT + (16 * qg + ln) * LDT + 4 * g4;
#pragma unroll
    for (int jd = 0; jd < 3; ++jd) {
      float a4[4], b4[4], c4[4];
      ld4(drow + 16 * jd + 4 * g4, a4);
      ld4(arow + 16 * jd + 4 * g4, c4);
      if (kq == 0) { ld4(qrow + 16 * jd + 4 * g4, b4); st4(orow + 16 * jd, a4); st4(qtr + 16 * jd, b4); }
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
  T* prow = PT + (16 * qg + ln) * LDP + 4 * g4;
  T* srow = ST + (16 * qg + ln) * LDP + 4 * g4;
#pragma unroll 1
  for (int s = kq * W::IT; s < (kq + 1) * W::IT; ++s) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const f32x4 sa = k48_rows<T>(Kt, LDK, 32 * s, qop, lane, z), sb = k48_rows<T>(Kt, LDK, 32 * s + 16, qop, lane, z);
    const f32x4 dPa = k48_rows<T>(Vt, LDK, 32 * s, dop, lane, z), dPb = k48_rows<T>(Vt, LDK, 32 * s + 16, dop, lane, z);
    f32x4 ds2[2];
    float pv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int key = 32 * s + 16 * (e >> 2) + 4 * g4 + (e & 3);
      const float2 ok = *reinterpret_cast<const float2*>(offs + 2 * key);
      BiasPt c;
      const float v = (e < 4 ? sa[e & 3] : sb[e & 3]) * p.scale + bias_at(tbl, TH, TW, qi, qj, key / Ww, key % Ww, ok.x, ok.y, c);
      const float pr = __expf(v - lse);
      pv[e] = pr;
      const float go = pr * ((e < 4 ? dPa[e & 3] : dPb[e & 3]) - dsum);        // d(logit) = d(bias): the bias enters the logit with weight 1
      ds2[e >> 2][e & 3] = go * p.scale;
      // bias backward (the gradient rules of stj_fg_bias_bwd: TF's clip gradients, zero outside the padded table)
      const float top = c.c.ax * (c.tr - c.tl) + c.tl, bot = c.c.ax * (c.br - c.bl) + c.bl;
      float d0 = c.c.gy ? -go * (bot - top) : 0.f;
      float d1 = c.c.gx ? -go * (c.c.ay * (c.br - c.bl) + (1.f - c.c.ay) * (c.tr - c.tl)) : 0.f;
      {
        // table gradient: go (1 - ay) goes to table row y0, go ay to row y0 + 1, each split (1 - ax, ax) over columns x0, x0 + 1.  The 16
        // queries of a wave run along one map row, so for one key lane ln's row y0 + 1 IS lane ln + 1's row y0 (same x0): the upper half
        // is handed to the next lane inside the DPP row and one LDS atomic covers both -- LDS float atomics retire about one lane per
        // clock, and with four of them per (query, key) pair they were 34 of this kernel's 64 us.  (Lanes whose sample point was
        // clamped carry weight 0 for every cell inside the table and drop out.)
        const float hi = go * c.c.ay;
        const int y0p = __builtin_amdgcn_update_dpp(-9, c.c.y0, 0x111, 0xF, 0xF, false), x0p = __builtin_amdgcn_update_dpp(-9, c.c.x0, 0x111, 0xF, 0xF, false);
        const int y0n = __builtin_amdgcn_update_dpp(-9, c.c.y0, 0x101, 0xF, 0xF, false), x0n = __builtin_amdgcn_update_dpp(-9, c.c.x0, 0x101, 0xF, 0xF, false);
        const float hip = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(hi), 0x111, 0xF, 0xF, false));
        const bool absorb = ln > 0 && y0p + 1 == c.c.y0 && x0p == c.c.x0;          // this lane adds the previous lane's upper half
        const bool given = ln < 15 && c.c.y0 + 1 == y0n && c.c.x0 == x0n;          // the next lane adds this lane's upper half
        const float lo = go * (1.f - c.c.ay) + (absorb ? hip : 0.f);
        const bool xa = c.c.x0 >= 1 && c.c.x0 <= TW, xb = c.c.x0 + 1 >= 1 && c.c.x0 + 1 <= TW;
        if (c.c.y0 >= 1 && c.c.y0 <= TH && lo != 0.f) {
          float* row = dtb + (c.c.y0 - 1) * TW + (c.c.x0 - 1);
          if (xa && c.c.ax != 1.f) atomicAdd(row, lo * (1.f - c.c.ax));
          if (xb && c.c.ax != 0.f) atomicAdd(row + 1, lo * c.c.ax);
        }
        if (!given && c.c.y0 + 1 >= 1 && c.c.y0 + 1 <= TH && hi != 0.f) {
          float* row = dtb + c.c.y0 * TW + (c.c.x0 - 1);
          if (xa && c.c.ax != 1.f) atomicAdd(row, hi * (1.f - c.c.ax));
          if (xb && c.c.ax != 0.f) atomicAdd(row + 1, hi * c.c.ax);
        }
      }
      // the key's offset gradient: sum over the wave's 16 queries (the lanes of equal g4 = a DPP row), one LDS add per wave
      d0 = row16_sum(d0); d1 = row16_sum(d1);
      if (ln == 0) { atomicAdd(&doffs[2 * key], d0); atomicAdd(&doffs[2 * key + 1], d1); }
    }
    st4(prow + 32 * s, pv); st4(prow + 32 * s + 16, pv + 4);
    { const float x0[4] = {ds2[0][0], ds2[0][1], ds2[0][2], ds2[0][3]}, x1[4] = {ds2[1][0], ds2[1][1], ds2[1][2], ds2[1][3]};
      st4(srow + 32 * s, x0); st4(srow + 32 * s + 16, x1); }
#pragma unroll
    for (int hf = 0; hf < 2 / ND; ++hf) {
      const typename Ch<T>::Frag sf = Ch<T>::from_acc(&ds2[hf * ND]);            // dq^T += K^T dS^T (dS carries the 48^-1/2)
#pragma unroll
      for (int jd = 0; jd < 3; ++jd) dq[jd] = Mma<T>::mma(Ch<T>::ldA_tr(Kt, LDK, 16 * jd, 32 * s + 16 * hf, lane), sf, dq[jd]);
    }
  }
  __syncthreads();
  float* mb = reinterpret_cast<float*>(fg_smem);                       // dq of the key slices [KS][QG][12][64], over the K / V tiles
  if constexpr (KS > 1) {
    float* my = mb + ((kq * QG + qg) * 12) * 64 + lane;
#pragma unroll
    for (int jd = 0; jd < 3; ++jd)
#pragma unroll
      for (int r = 0; r < 4; ++r) my[(4 * jd + r) * 64] = dq[jd][r];
  }
  {   // dV = P^T dO, dK = dS^T q for the MW x 16 keys this wave owns: m = key, n = head column, k = the tile's 64 queries
    f32x4 dv[MW][3], dk[MW][3];
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
      for (int jd = 0; jd < 3; ++jd) { dv[m][jd] = (f32x4){0.f, 0.f, 0.f, 0.f}; dk[m][jd] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int s = 0; s < TQ / KSTEP; ++s) {
      typename Mma<T>::Frag bo[3], bq[3];
#pragma unroll
      for (int jd = 0; jd < 3; ++jd) { bo[jd] = Mma<T>::load_tr(OT, LDT, 16 * jd, KSTEP * s, lane); bq[jd] = Mma<T>::load_tr(QT, LDT, 16 * jd, KSTEP * s, lane); }
#pragma unroll
      for (int m = 0; m < MW; ++m) {
        const typename Mma<T>::Frag ap = Mma<T>::load_tr(PT, LDP, 16 * (MW * wv + m), KSTEP * s, lane);
        const typename Mma<T>::Frag as = Mma<T>::load_tr(ST, LDP, 16 * (MW * wv + m), KSTEP * s, lane);
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
  if constexpr (KS > 1) {
    __syncthreads();
    if (kq == 0) {
#pragma unroll
      for (int i = 1; i < KS; ++i) {
        const float* src = mb + ((i * QG + qg) * 12) * 64 + lane;
#pragma unroll
        for (int jd = 0; jd < 3; ++jd)
#pragma unroll
          for (int r = 0; r < 4; ++r) dq[jd][r] += src[(4 * jd + r) * 64];
      }
    }
  }
  if (kq == 0) {
    T* dqo = reinterpret_cast<T*>(p.dq) + (row0 + tok) * C + D * g + 4 * g4;
#pragma unroll
    for (int jd = 0; jd < 3; ++jd) { const float w[4] = {dq[jd][0], dq[jd][1], dq[jd][2], dq[jd][3]}; st4(dqo + 16 * jd, w); }
  }
  // table / offset gradients of this tile
  for (int i = tid; i < TT; i += NT)
    if (dtb[i] != 0.f) atomicAdd(p.dtable + i * p.G + g, dtb[i]);
  float* dof = p.doff + ((long long)b * p.G + g) * HW * 2;
  for (int i = tid; i < 2 * HW; i += NT) {
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
  static PerDevice<int> reserved[2];           // per (T, NKF) instantiation and direction: the attribute is set once per size, not per launch
  if (lds > 160 * 1024) { stj_set_error("fg_attn: %d bytes of LDS", lds); return STJ_ELAUNCH; }
  if (lds > reserved[bwd]) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
      stj_set_error("fg_attn: cannot reserve %d bytes of LDS", lds); return STJ_ELAUNCH;
    }
    reserved[bwd] = lds;
  }
  if (bwd) hipLaunchKernelGGL((fgattn_bwd_kernel<T, NKF>), dim3((unsigned)(a.B * a.G * (16 * NKF / Split<NKF, true, T>::TQ))), dim3(Split<NKF, true, T>::NT), lds, st, a);
  else hipLaunchKernelGGL((fgattn_fwd_kernel<T, NKF>), dim3((unsigned)(a.B * a.G * (16 * NKF / Split<NKF, false, T>::TQ))), dim3(Split<NKF, false, T>::NT), lds, st, a);
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
  if (!stj_dtype_ok(dtype)) { stj_set_error("fg_attn: bad dtype %d", dtype); return STJ_EINVAL; }
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
  return dtype == STJ_BF16 ? fga::dispatch<bf16>(false, p, stream) : (dtype == STJ_F16 ? fga::dispatch<f16>(false, p, stream) : fga::dispatch<float>(false, p, stream));
}
// bytes of each of the two partial-sum workspaces (dkp, dvp) of stj_fg_attn_bwd
extern "C" long long stj_fg_attn_bwd_workspace_bytes(int B, int G, int Hh, int Ww) {
  const long long HW = (long long)Hh * Ww;
  return (long long)B * G * (HW / 16) * HW * fga::D * 4;         // sized for the smallest query tile any dtype uses (f32 on the 16 x 16 map: 16)
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
  int rc = dtype == STJ_BF16 ? fga::dispatch<bf16>(true, p, stream) : (dtype == STJ_F16 ? fga::dispatch<f16>(true, p, stream) : fga::dispatch<float>(true, p, stream));
  if (rc != STJ_OK) return rc;
  const int HW = Hh * Ww;
  const int tq = Hh == 16 ? (dtype == STJ_F32 ? fga::Split<16, true, float>::TQ : fga::Split<16, true, bf16>::TQ) : fga::Split<4, true, bf16>::TQ;
  const int tiles = HW / tq;
  const long long n = 2ll * B * HW * G * fga::D;
  const int grid = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
  if (dtype == STJ_BF16) hipLaunchKernelGGL(fga::fgattn_dkv_reduce_kernel<bf16>, dim3(grid), dim3(256), 0, stream, dkp, dvp, (bf16*)dk, (bf16*)dv, B, G, HW, tiles);
  else if (dtype == STJ_F16) hipLaunchKernelGGL(fga::fgattn_dkv_reduce_kernel<f16>, dim3(grid), dim3(256), 0, stream, dkp, dvp, (f16*)dk, (f16*)dv, B, G, HW, tiles);
  else hipLaunchKernelGGL(fga::fgattn_dkv_reduce_kernel<float>, dim3(grid), dim3(256), 0, stream, dkp, dvp, (float*)dk, (float*)dv, B, G, HW, tiles);
  return stj_check_launch("stj_fg_attn_dkv_reduce");
}
