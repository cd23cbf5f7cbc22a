// Fused FG-MSA offset head: the chain between the query projection and the offsets as ONE launch per direction.
//   reference FG_MSA.py:84-92    conv_offset = Conv2D(3x3, groups = 8, SAME) -> LayerNorm(1e-3) -> gelu -> Conv2D(1x1, 48 -> 2 per group, no bias)
//   reference FG_MSA.py:109-123  offset = tanh(conv_offset(q)) * offset_range            (offset_range = H / 2)
// Layer by layer this was im2col + grouped GEMM + LayerNorm + gelu + the offset kernel forward (5 launches of 11-26 us on the inference
// forward's critical chain) and offset + gelu' + LayerNorm' + GEMM + col2im backward.
//
// Work layout.  One workgroup owns R consecutive image rows of one scene (R x W = 16 or 32 pixels: W = 8 | 16 | 32, all 384 channels); its
// eight waves are the eight groups.  The R + 2 input rows the 3x3 windows touch live in an LDS halo tile [R + 2][W + 2][384] (zero rows /
// columns outside the image); the
// grouped conv is an implicit GEMM per group, D[m = output channel 0..47][n = pixel] = sum_k W^T[m][k] X[n][k] with k = tap * 48 + input
// channel: the WEIGHT fragments (MFMA A operand) come straight from global memory / L2 in fragment order (stj_fgoff_pack writes them so:
// one contiguous kilobyte per wave load), all of a wave's fragments in flight before the halo is staged; the activation fragment
// (B operand) of a k-step is 16 bytes of ONE tap's channels of a shifted pixel, read in place from the halo tile -- no im2col.
// LayerNorm, gelu, the 48 -> 2 product and tanh are a row-wise tail on the conv's output tile in LDS (lane = 6 channels, 8 lanes = one group).
// The backward kernel runs the tail's backward for the R + 2 rows its transposed conv needs (recomputing the two neighbours' is cheaper than a
// second launch and a round trip), keeps the conv output's gradient in the halo tile, and runs the same implicit GEMM with the
// tap-mirrored, channel-transposed weight pack.  The conv's weight gradient stays the caller's GEMM (cols^T dc on the grouped stream-K
// launch): in training the forward writes the im2col matrix from its halo tile, and the backward writes dc.
// Roundings follow the layer-by-layer chain (each layer's output in the storage dtype), so both paths agree to the summation order.
#include "common.h"
#include "fgoff_fused_abi.h"

namespace fgo {
constexpr int G = 8, GC = 48, C = G * GC, KK = 9 * GC;      // groups, channels per group, 384, 432 = contraction length per group

template <typename T> struct FG {
  typedef Mma<T> M;
  static constexpr int KSTEP = M::KSTEP, LK = M::LANE_K;
  static constexpr int NKS = (KK + KSTEP - 1) / KSTEP;               // 14 (16-bit: 448, the tail zero) | 27 (f32: 432)
  // k-steps whose weight fragments a wave holds at once: forward 16-bit all 14 (168 registers, nothing else is live); backward two halves,
  // double buffered (the row-wise backward in front of the conv needs ~90 registers of its own); f32 9 of 27, not double buffered
  static constexpr int CHF = sizeof(T) == 2 ? NKS : 9, CHB = sizeof(T) == 2 ? NKS / 2 : 9;
  static constexpr bool DBB = sizeof(T) == 2;
  static constexpr int LD = C + LdsPad<T>::P;
  static constexpr long long DIR = (long long)G * NKS * 3 * 64 * LK;   // elements of one direction's pack
};

template <typename T> __device__ __forceinline__ float rnd(float x) { T t; stf(&t, x); return ldf(&t); }      // value as the storage type holds it

// ---- weight pack: MFMA A fragments in the order the waves load them -----------------------------------------------------------------
// element ((((dir * 8 + g) * NKS + s) * 3 + mt) * 64 + lane) * LK + j  =  weight of row m = 16 mt + (lane & 15), k = s KSTEP + (lane >> 4) LK + j
//   dir 0 (forward):  row = output channel g 48 + m, k = tap 48 + ic        -> w[tap][ic][g 48 + m]
//   dir 1 (backward): row = input channel ic = m,    k = tap' 48 + oc       -> w[8 - tap'][m][g 48 + oc]     (taps mirrored)
template <typename T>
__global__ __launch_bounds__(256) void fgoff_pack_kernel(const float* __restrict__ w, T* __restrict__ out) {
  typedef FG<T> F;
  const long long tot = 2 * F::DIR;
  for (long long e = blockIdx.x * 256ll + threadIdx.x; e < tot; e += gridDim.x * 256ll) {
    long long t = e;
    const int j = (int)(t % F::LK); t /= F::LK;
    const int lane = (int)(t % 64); t /= 64;
    const int mt = (int)(t % 3); t /= 3;
    const int s = (int)(t % F::NKS); t /= F::NKS;
    const int g = (int)(t % G); const int dir = (int)(t / G);
    const int m = mt * 16 + (lane & 15), k = s * F::KSTEP + (lane >> 4) * F::LK + j;
    float v = 0.f;
    if (k < KK) {
      const int tap = k / GC, cc = k % GC;
      v = dir == 0 ? w[((long long)tap * GC + cc) * C + g * GC + m] : w[((long long)(8 - tap) * GC + m) * C + g * GC + cc];
    }
    stf(out + e, v);
  }
}

struct Args {
  int B, H, W; float scale, eps;
  const void* q; const void* pack; const float* bias; const float* gamma; const float* beta; const void* w1;
  void* off; void* cols; void* c; float* mean; float* rstd;
  const void* doff; void* dc; void* dq; float* d_w1; float* d_gamma; float* d_beta; float* d_bias;
};

template <typename T, int W, int R> constexpr size_t lds_bytes(bool bwd) {
  return (size_t)((R + 2) * (W + 2) + R * W) * FG<T>::LD * sizeof(T) + (bwd ? (3 * C + 2 * GC) * sizeof(float) : 0);
}

// Weight fragments of the calling wave (= group) in chunks of CH k-steps.  Chunk 0 is issued by the kernel before anything else (load_w), so
// its latency hides under the halo staging / the row-wise backward; DB: the next chunk's loads are issued in front of this chunk's MFMAs.
template <typename T, int CH>
__device__ __forceinline__ void load_w(const T* pk, int ch, int lane, typename Mma<T>::Frag (&wa)[CH][3]) {
  typedef FG<T> F;
#pragma unroll
  for (int s = 0; s < CH; ++s)
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) wa[s][mt] = Mma<T>::from_global(pk + ((long long)((ch * CH + s) * 3 + mt) * 64 + lane) * F::LK);
}
// acc[mt][nt] = the group's 48 x (R W) output tile, token n = 16 nt + (lane & 15) = pixel (n / W, n % W) of the workgroup's rows; pk: the
// group's fragments of the direction; wa[0]: chunk 0, already loaded
template <typename T, int W, int R, int CH, bool DB>
__device__ __forceinline__ void conv_rows(const T* halo, const T* pk, int g, int lane, typename Mma<T>::Frag (&wa)[DB ? 2 : 1][CH][3], f32x4 (&acc)[3][R * W / 16]) {
  typedef FG<T> F; typedef Mma<T> M;
  constexpr int WP = W + 2, NT = R * W / 16, NCH = F::NKS / CH;
  static_assert(F::NKS % CH == 0, "chunks");
  const int ln = lane & 15, gq = lane >> 4;
  int tok[NT];                                     // halo position of the lane's pixel, per token tile
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) { const int n = nt * 16 + ln; tok[nt] = (n / W) * WP + n % W; }
#pragma unroll
  for (int mt = 0; mt < 3; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    if (DB) { if (ch + 1 < NCH) load_w<T, CH>(pk, ch + 1, lane, wa[(ch + 1) & (DB ? 1 : 0)]); }
    else if (ch > 0) load_w<T, CH>(pk, ch, lane, wa[0]);
#pragma unroll
    for (int s = 0; s < CH; ++s) {
      const int k0 = (ch * CH + s) * F::KSTEP + gq * F::LK;
      int tap = k0 / GC;
      const int cc = k0 - tap * GC;                 // (the zero tail of the 16-bit pack, k0 >= 432: any finite data will do -- tap 8's)
      tap = tap > 8 ? 8 : tap;
      const int dy = tap / 3, dx = tap - 3 * dy;
      const T* bp = halo + ((dy * WP + dx) * F::LD + g * GC + cc);
      typename M::Frag bf[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bf[nt] = *reinterpret_cast<const typename M::Frag*>(bp + tok[nt] * F::LD);
#pragma unroll
      for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = M::mma(wa[DB ? (ch & 1) : 0][s][mt], bf[nt], acc[mt][nt]);
    }
  }
}
// accumulators (+ per-channel f32 addend) -> [pixel][384] tile in the storage dtype
template <typename T, int NT>
__device__ __forceinline__ void acc_to_tile(const f32x4 (&acc)[3][NT], const float* add, T* ct, int g, int lane) {
  const int ln = lane & 15, gq = lane >> 4;
#pragma unroll
  for (int mt = 0; mt < 3; ++mt) {
    const int ch = g * GC + mt * 16 + 4 * gq;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (add) bv = *reinterpret_cast<const float4*>(add + ch);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const float v[4] = {acc[mt][nt][0] + bv.x, acc[mt][nt][1] + bv.y, acc[mt][nt][2] + bv.z, acc[mt][nt][3] + bv.w};
      st4(ct + (nt * 16 + ln) * FG<T>::LD + ch, v);
    }
  }
}
template <typename T>
__device__ __forceinline__ void tile_to_global(const T* ct, T* dst, int rows, int tid) {        // [rows][384] contiguous
  constexpr int V = Vec<T>::N, VPR = C / V;
  for (int i = tid; i < rows * VPR; i += 512) {
    const int r = i / VPR, c = (i % VPR) * V;
    *reinterpret_cast<uint4*>(dst + (long long)r * C + c) = *reinterpret_cast<const uint4*>(ct + r * FG<T>::LD + c);
  }
}

// =====================================================================================================================================
// forward
// =====================================================================================================================================
template <typename T, int W, int R>
__global__ __launch_bounds__(512) void fgoff_fwd_kernel(Args a) {
  typedef FG<T> F; typedef Mma<T> M;
  constexpr int NT = R * W / 16, TOK = R * W, WP = W + 2, LD = F::LD, V = Vec<T>::N, VPR = C / V;
  constexpr bool FAST = sizeof(T) == 2;
  extern __shared__ __align__(16) unsigned char fgo_smem[];
  T* halo = reinterpret_cast<T*>(fgo_smem);         // [R + 2][WP][LD]
  T* ct = halo + (R + 2) * WP * LD;                  // [R W][LD]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int rb = a.H / R, b = blockIdx.x / rb, row = (blockIdx.x % rb) * R;
  const T* pk = reinterpret_cast<const T*>(a.pack) + (long long)wv * F::NKS * 3 * 64 * F::LK;
  typename M::Frag wa[1][F::CHF][3];
  load_w<T, F::CHF>(pk, 0, lane, wa[0]);
  // halo tile
  const T* q = reinterpret_cast<const T*>(a.q);
  for (int i = tid; i < (R + 2) * WP * VPR; i += 512) {
    const int cv = i % VPR, t = i / VPR, px = t % WP, r = t / WP;
    const int sy = row - 1 + r, sx = px - 1;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (sy >= 0 && sy < a.H && sx >= 0 && sx < W) v = *reinterpret_cast<const uint4*>(q + ((long long)(b * a.H + sy) * W + sx) * C + cv * V);
    *reinterpret_cast<uint4*>(halo + (r * WP + px) * LD + cv * V) = v;
  }
  __syncthreads();
  const long long m0 = (long long)(b * a.H + row) * W;
  if (a.cols) {                                      // im2col rows of these image rows: [pixel][group][tap][48]
    constexpr int VPT = GC / V;
    T* cols = reinterpret_cast<T*>(a.cols) + m0 * G * KK;
    for (int i = tid; i < TOK * G * 9 * VPT; i += 512) {
      const int v = i % VPT; int t = i / VPT;
      const int tap = t % 9; t /= 9;
      const int g = t % G, p = t / G;
      *reinterpret_cast<uint4*>(cols + (long long)i * V) =
          *reinterpret_cast<const uint4*>(halo + ((p / W + tap / 3) * WP + p % W + tap % 3) * LD + g * GC + v * V);
    }
  }
  f32x4 acc[3][NT];
  conv_rows<T, W, R, F::CHF, false>(halo, pk, wv, lane, wa, acc);
  acc_to_tile<T, NT>(acc, a.bias, ct, wv, lane);
  __syncthreads();
  if (a.c) tile_to_global<T>(ct, reinterpret_cast<T*>(a.c) + m0 * C, TOK, tid);
  // row-wise tail: LayerNorm -> gelu -> 48 -> 2 -> tanh * scale.  lane = 6 channels, 8 lanes = one group
  const int c0 = lane * 6, g = lane >> 3, i0 = (lane & 7) * 6;
  float gam[6], bet[6], w1[6][2];
  const T* W1 = reinterpret_cast<const T*>(a.w1);
#pragma unroll
  for (int e = 0; e < 6; ++e) { gam[e] = a.gamma[c0 + e]; bet[e] = a.beta[c0 + e]; w1[e][0] = ldf(W1 + 2 * (i0 + e)); w1[e][1] = ldf(W1 + 2 * (i0 + e) + 1); }
  T* off = reinterpret_cast<T*>(a.off);
  const int HW = a.H * W;
  for (int p = wv; p < TOK; p += 8) {                // (rows are consecutive: pixel p of the tile is pixel row W + p of the scene)
    float x[6], s = 0.f;
#pragma unroll
    for (int e = 0; e < 6; ++e) { x[e] = ldf(ct + p * LD + c0 + e); s += x[e]; }
    const float mu = wave_sum(s) * (1.f / C);
    float qq = 0.f;
#pragma unroll
    for (int e = 0; e < 6; ++e) { const float d = x[e] - mu; qq += d * d; }
    const float rs = rsqrtf(wave_sum(qq) * (1.f / C) + a.eps);
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int e = 0; e < 6; ++e) {
      const float y = rnd<T>((x[e] - mu) * rs * gam[e] + bet[e]);
      const float act = rnd<T>(unary_f<U_GELU, FAST>(y, 0.f));
      a0 += act * w1[e][0]; a1 += act * w1[e][1];
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) { a0 += __shfl_xor(a0, o, 64); a1 += __shfl_xor(a1, o, 64); }
    if ((lane & 7) == 0) {
      T* op = off + ((long long)(b * G + g) * HW + row * W + p) * 2;
      stf(op, tanhf(a0) * a.scale); stf(op + 1, tanhf(a1) * a.scale);
    }
    if (a.mean && lane == 0) { a.mean[m0 + p] = mu; a.rstd[m0 + p] = rs; }
  }
}

// =====================================================================================================================================
// backward
// =====================================================================================================================================
template <typename T, int W, int R>
__global__ __launch_bounds__(512) void fgoff_bwd_kernel(Args a) {
  typedef FG<T> F; typedef Mma<T> M;
  constexpr int NT = R * W / 16, TOK = R * W, WP = W + 2, LD = F::LD, V = Vec<T>::N;
  constexpr bool FAST = sizeof(T) == 2;
  extern __shared__ __align__(16) unsigned char fgo_smem[];
  T* halo = reinterpret_cast<T*>(fgo_smem);         // [R + 2][WP][LD]  gradient of the conv output, rows row-1 .. row+R
  T* ct = halo + (R + 2) * WP * LD;                  // [R W][LD]
  float* red = reinterpret_cast<float*>(ct + TOK * LD);    // [3][384] d_gamma | d_beta | d_bias, [48][2] d_w1
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int rb = a.H / R, b = blockIdx.x / rb, row = (blockIdx.x % rb) * R;
  const T* pk = reinterpret_cast<const T*>(a.pack) + F::DIR + (long long)wv * F::NKS * 3 * 64 * F::LK;
  typename M::Frag wa[F::DBB ? 2 : 1][F::CHB][3];
  load_w<T, F::CHB>(pk, 0, lane, wa[0]);
  for (int i = tid; i < (R + 2) * WP * LD / V; i += 512) *reinterpret_cast<uint4*>(halo + i * V) = make_uint4(0, 0, 0, 0);
  for (int i = tid; i < 3 * C + 2 * GC; i += 512) red[i] = 0.f;
  __syncthreads();
  // row-wise tail backward for the tokens of the R + 2 rows
  const int c0 = lane * 6, g = lane >> 3, i0 = (lane & 7) * 6;
  float gam[6], bet[6], w1[6][2];
  const T* W1 = reinterpret_cast<const T*>(a.w1);
#pragma unroll
  for (int e = 0; e < 6; ++e) { gam[e] = a.gamma[c0 + e]; bet[e] = a.beta[c0 + e]; w1[e][0] = ldf(W1 + 2 * (i0 + e)); w1[e][1] = ldf(W1 + 2 * (i0 + e) + 1); }
  float dg[6], db[6], dbi[6], dw[6][2];
#pragma unroll
  for (int e = 0; e < 6; ++e) dg[e] = db[e] = dbi[e] = dw[e][0] = dw[e][1] = 0.f;
  const T* cs = reinterpret_cast<const T*>(a.c);
  const T* off = reinterpret_cast<const T*>(a.off);
  const T* doff = reinterpret_cast<const T*>(a.doff);
  T* dc = reinterpret_cast<T*>(a.dc);
  const int HW = a.H * W;
  const float inv_scale = 1.f / a.scale;
  for (int t = wv; t < (R + 2) * W; t += 8) {
    const int r = t / W, p = t % W, sy = row - 1 + r;
    if (sy < 0 || sy >= a.H) continue;               // (wave-uniform)
    const long long m = (long long)(b * a.H + sy) * W + p;
    const bool mine = r >= 1 && r <= R;              // the workgroup's own rows: the others are recomputed for the conv's halo only
    const float own = mine ? 1.f : 0.f;
    const float mu = a.mean[m], rs = a.rstd[m];
    const long long oi = ((long long)(b * G + g) * HW + sy * W + p) * 2;
    const float o0 = ldf(off + oi), o1 = ldf(off + oi + 1);
    const float ds0 = ldf(doff + oi) * (a.scale - o0 * o0 * inv_scale), ds1 = ldf(doff + oi + 1) * (a.scale - o1 * o1 * inv_scale);
    float xh[6], dxh[6], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < 6; ++e) {
      xh[e] = (ldf(cs + m * C + c0 + e) - mu) * rs;
      const float y = rnd<T>(xh[e] * gam[e] + bet[e]);
      const float act = rnd<T>(unary_f<U_GELU, FAST>(y, 0.f));
      const float da = rnd<T>(ds0 * w1[e][0] + ds1 * w1[e][1]);
      const float dy = rnd<T>(unary_g<U_GELU, FAST>(da, y, 0.f));
      dw[e][0] += own * act * ds0; dw[e][1] += own * act * ds1;
      dg[e] += own * dy * xh[e]; db[e] += own * dy;
      dxh[e] = dy * gam[e];
      s1 += dxh[e]; s2 += dxh[e] * xh[e];
    }
    s1 = wave_sum(s1) * (1.f / C); s2 = wave_sum(s2) * (1.f / C);
#pragma unroll
    for (int e = 0; e < 6; ++e) {
      const float v = rnd<T>(rs * (dxh[e] - s1 - xh[e] * s2));
      stf(halo + (r * WP + p + 1) * LD + c0 + e, v);
      dbi[e] += own * v;
      if (mine) stf(dc + m * C + c0 + e, v);
    }
  }
  // parameter gradients: lanes -> LDS -> one atomic per entry and workgroup
#pragma unroll
  for (int e = 0; e < 6; ++e) {
    atomicAdd(red + c0 + e, dg[e]); atomicAdd(red + C + c0 + e, db[e]); atomicAdd(red + 2 * C + c0 + e, dbi[e]);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float v = dw[e][j];
      v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
      if (lane < 8) atomicAdd(red + 3 * C + 2 * (i0 + e) + j, v);
    }
  }
  __syncthreads();
  for (int i = tid; i < C; i += 512) {
    atomicAdd(a.d_gamma + i, red[i]); atomicAdd(a.d_beta + i, red[C + i]); atomicAdd(a.d_bias + i, red[2 * C + i]);
  }
  if (tid < 2 * GC) atomicAdd(a.d_w1 + tid, red[3 * C + tid]);
  // transposed conv of the gradient tile
  f32x4 acc[3][NT];
  conv_rows<T, W, R, F::CHB, F::DBB>(halo, pk, wv, lane, wa, acc);
  acc_to_tile<T, NT>(acc, nullptr, ct, wv, lane);
  __syncthreads();
  tile_to_global<T>(ct, reinterpret_cast<T*>(a.dq) + (long long)(b * a.H + row) * W * C, TOK, tid);
}
}  // namespace fgo

// =====================================================================================================================================
// C ABI
// =====================================================================================================================================
// rows per workgroup: W = 8 -> 2 (one MFMA token tile), W = 32 -> 1; W = 16 -> 2 (half the weight stream per pixel, 4 halo rows per 2
// instead of 3 per 1) when that still leaves a workgroup per CU, else 1 -- measured, forward / forward + backward: B = 32 23 / 84 us with
// two rows against 28 / 113 us with one; B = 8 (64 against 128 workgroups) 18 / 69 against 13 / 62 us.  (f32: two rows' LDS tiles do not fit.)
static int fgoff_rows(int H, int W, int dtype, int B) {
  if (W == 8) return H % 2 == 0 ? 2 : 0;
  if (W == 16) return (stj_is16(dtype) && H % 2 == 0 && (long long)B * (H / 2) >= 256) ? 2 : 1;
  if (W == 32) return stj_is16(dtype) ? 1 : 0;
  return 0;
}
extern "C" int stj_fgoff_supported(int H, int W, int C, int G, int dtype) {
  if (!stj_dtype_ok(dtype) || C != fgo::C || G != fgo::G || H < 1) return 0;
  return fgoff_rows(H, W, dtype, 1) > 0;
}
static long long fgoff_pack_elems(int dtype) { return 2 * (stj_is16(dtype) ? fgo::FG<bf16>::DIR : fgo::FG<float>::DIR); }
extern "C" long long stj_fgoff_pack_workspace_bytes(int dtype) {
  if (!stj_dtype_ok(dtype)) return 0;
  return fgoff_pack_elems(dtype) * (stj_is16(dtype) ? 2 : 4);
}
extern "C" int stj_fgoff_pack(const float* w, void* out, int dtype, hipStream_t stream) {
  if (!w || !out) { stj_set_error("stj_fgoff_pack: null pointer"); return STJ_EINVAL; }
  if (!stj_dtype_ok(dtype)) { stj_set_error("stj_fgoff_pack: bad dtype %d", dtype); return STJ_EINVAL; }
  if ((uintptr_t)out & 15) { stj_set_error("stj_fgoff_pack: out must be 16-byte aligned"); return STJ_EINVAL; }
  const int grid = (int)((fgoff_pack_elems(dtype) + 255) / 256);
  if (dtype == STJ_BF16) hipLaunchKernelGGL(fgo::fgoff_pack_kernel<bf16>, dim3(grid), dim3(256), 0, stream, w, (bf16*)out);
  else if (dtype == STJ_F16) hipLaunchKernelGGL(fgo::fgoff_pack_kernel<f16>, dim3(grid), dim3(256), 0, stream, w, (f16*)out);
  else hipLaunchKernelGGL(fgo::fgoff_pack_kernel<float>, dim3(grid), dim3(256), 0, stream, w, (float*)out);
  return stj_check_launch("stj_fgoff_pack");
}

template <typename T, int W, int R, bool BWD> static int fgoff_launch(const fgo::Args& a, hipStream_t stream) {
  using namespace fgo;
  static PerDevice<int> attr_set;
  constexpr size_t lds = lds_bytes<T, W, R>(BWD);
  if (!attr_set) {
    const void* fn = BWD ? (const void*)fgoff_bwd_kernel<T, W, R> : (const void*)fgoff_fwd_kernel<T, W, R>;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      stj_set_error("stj_fgoff: cannot reserve %zu bytes of LDS", lds);
      return STJ_ELAUNCH;
    }
    attr_set = 1;
  }
  if (BWD) hipLaunchKernelGGL((fgoff_bwd_kernel<T, W, R>), dim3(a.B * (a.H / R)), dim3(512), lds, stream, a);
  else hipLaunchKernelGGL((fgoff_fwd_kernel<T, W, R>), dim3(a.B * (a.H / R)), dim3(512), lds, stream, a);
  return stj_check_launch(BWD ? "stj_fgoff_bwd" : "stj_fgoff_fwd");
}
template <bool BWD> static int fgoff_run(const stj_fgoff_args* s, hipStream_t stream) {
  const char* who = BWD ? "stj_fgoff_bwd" : "stj_fgoff_fwd";
  if (!s) { stj_set_error("%s: null argument block", who); return STJ_EINVAL; }
  if (s->B <= 0) return STJ_OK;
  if (!stj_fgoff_supported(s->H, s->W, fgo::C, fgo::G, s->dtype)) { stj_set_error("%s: geometry / dtype not supported (H=%d W=%d dtype=%d)", who, s->H, s->W, s->dtype); return STJ_EUNSUPPORTED; }
  if (!s->pack || !s->gamma || !s->beta || !s->w1 || !s->off) { stj_set_error("%s: null pointer", who); return STJ_EINVAL; }
  if (!(s->scale > 0.f)) { stj_set_error("%s: need scale > 0", who); return STJ_EINVAL; }
  if (!BWD) {
    if (!s->q || !s->bias) { stj_set_error("%s: null pointer", who); return STJ_EINVAL; }
    const bool any = s->cols || s->c || s->mean || s->rstd, all = s->cols && s->c && s->mean && s->rstd;
    if (any && !all) { stj_set_error("%s: the four saved tensors go together", who); return STJ_EINVAL; }
    if (((uintptr_t)s->q | (uintptr_t)s->pack | (uintptr_t)s->cols | (uintptr_t)s->c | (uintptr_t)s->bias) & 15) { stj_set_error("%s: pointers must be 16-byte aligned", who); return STJ_EINVAL; }
  } else {
    if (!s->c || !s->mean || !s->rstd || !s->doff || !s->dc || !s->dq || !s->d_w1 || !s->d_gamma || !s->d_beta || !s->d_bias) { stj_set_error("%s: null pointer", who); return STJ_EINVAL; }
    if (((uintptr_t)s->pack | (uintptr_t)s->dq) & 15) { stj_set_error("%s: pointers must be 16-byte aligned", who); return STJ_EINVAL; }
  }
  fgo::Args a = {};
  a.B = s->B; a.H = s->H; a.W = s->W; a.scale = s->scale; a.eps = s->eps;
  a.q = s->q; a.pack = s->pack; a.bias = s->bias; a.gamma = s->gamma; a.beta = s->beta; a.w1 = s->w1;
  a.off = s->off; a.cols = s->cols; a.c = s->c; a.mean = s->mean; a.rstd = s->rstd;
  a.doff = s->doff; a.dc = s->dc; a.dq = s->dq; a.d_w1 = s->d_w1; a.d_gamma = s->d_gamma; a.d_beta = s->d_beta; a.d_bias = s->d_bias;
  const int R = fgoff_rows(s->H, s->W, s->dtype, s->B);
#define FGOFF_GO(WW, RR)                                                              \
  if (s->W == WW && R == RR) {                                                        \
    if (s->dtype == STJ_BF16) return fgoff_launch<bf16, WW, RR, BWD>(a, stream);      \
    if (s->dtype == STJ_F16) return fgoff_launch<f16, WW, RR, BWD>(a, stream);        \
  }
  FGOFF_GO(8, 2) FGOFF_GO(16, 1) FGOFF_GO(16, 2) FGOFF_GO(32, 1)
#undef FGOFF_GO
  if (s->W == 8) return fgoff_launch<float, 8, 2, BWD>(a, stream);
  return fgoff_launch<float, 16, 1, BWD>(a, stream);
}
extern "C" int stj_fgoff_fwd(const stj_fgoff_args* s, hipStream_t stream) { return fgoff_run<false>(s, stream); }
extern "C" int stj_fgoff_bwd(const stj_fgoff_args* s, hipStream_t stream) { return fgoff_run<true>(s, stream); }
