// Fused (shifted-)window multi-head self-attention, forward and backward: one wavefront per
// (scene, window, head).  Replaces roll -> window_partition -> per-head softmax(q k^T*scale + relpos
// bias [+ shift mask]) v -> window_reverse -> roll of the reference (modules.py:49-63 window_partition/
// reverse, :103-134 WindowAttention.call, :189-216 shift mask, :229-255 roll/partition plumbing).
// The cyclic shift and the window partition are pure index arithmetic on the token map: q/k/v are
// gathered straight from the [B, res*res, 3C] qkv tensor in original token order and the result is
// written back in original token order (no rolled / partitioned tensors are materialised).
// window = 8x8 tokens (64), head_dim = 32:  S = Q K^T is a 64x64x32 MFMA tile, O = P V a 64x32x64 one;
// softmax row reductions are 16-lane xor shuffles on the MFMA accumulator layout.
#include "common.h"

#define WS 8
#define WN 64
#define HD 32

template <typename T> struct WinCfg;
template <> struct WinCfg<bf16> { static constexpr int WPB = 4; static constexpr int WPB_BWD = 1; };   // bwd: 46 KB of LDS per wave -> 3 resident waves per CU (2 with WPB_BWD = 2)
template <> struct WinCfg<f16> { static constexpr int WPB = 4; static constexpr int WPB_BWD = 1; };
template <> struct WinCfg<float> { static constexpr int WPB = 2; static constexpr int WPB_BWD = 2; };

struct WinGeom {
  int B, res, heads, shift;
  long long items;
  int nparts;      // backward: number of partial dtable copies the blocks spread their atomics over
};

__device__ __forceinline__ int win_token(int res, int shift, int wy, int wx, int t) {
  int ry = wy * WS + (t >> 3), rx = wx * WS + (t & 7);
  int sy = ry + shift; if (sy >= res) sy -= res;
  int sx = rx + shift; if (sx >= res) sx -= res;
  return sy * res + sx;
}
__device__ __forceinline__ int win_label(int res, int shift, int wy, int wx, int t) {
  int ry = wy * WS + (t >> 3), rx = wx * WS + (t & 7);
  int ly = ry < res - WS ? 0 : (ry < res - shift ? 1 : 2);
  int lx = rx < res - WS ? 0 : (rx < res - shift ? 1 : 2);
  return ly * 3 + lx;
}

// load a [64][32] operand (which = 0 q, 1 k, 2 v of qkv; or a plain [B,N,C] tensor when stride3 = 1) into LDS rows
template <typename T>
__device__ __forceinline__ void win_load(T* dst, int ld, const T* src, long long tokstride, int coff, const int* tok, int lane) {
  constexpr int VN = Vec<T>::N;
  constexpr int CPR = HD / VN;               // 16-byte chunks per token row
  for (int i = lane; i < WN * CPR; i += 64) {
    int t = i / CPR, c = (i % CPR) * VN;
    *reinterpret_cast<uint4*>(dst + t * ld + c) =
        *reinterpret_cast<const uint4*>(src + (long long)tok[t] * tokstride + coff + c);
  }
}

// scores -> probabilities in the accumulator layout.  s[i][j][r]: row = i*16+(lane>>4)*4+r, col = j*16+(lane&15)
__device__ __forceinline__ void win_softmax(f32x4 (&s)[4][4], const float* tbl, const int* lab, int use_mask, float scale, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = i * 16 + (lane >> 4) * 4 + r;
      float m = -INFINITY;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = j * 16 + (lane & 15);
        float v = s[i][j][r] * scale + tbl[((row >> 3) - (col >> 3) + WS - 1) * (2 * WS - 1) + ((row & 7) - (col & 7) + WS - 1)];
        if (use_mask && lab[row] != lab[col]) v += -100.0f;
        s[i][j][r] = v;
        m = fmaxf(m, v);
      }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) { float e = __expf(s[i][j][r] - m); s[i][j][r] = e; sum += e; }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
      const float inv = 1.f / sum;
#pragma unroll
      for (int j = 0; j < 4; ++j) s[i][j][r] *= inv;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(64 * WinCfg<T>::WPB) void win_attn_fwd_kernel(const T* qkv, const float* table, T* out, WinGeom g) {
  constexpr int WPB = WinCfg<T>::WPB;
  constexpr int LQ = HD + LdsPad<T>::P, LP = WN + LdsPad<T>::P;
  // per wave: Q|K region (reused for P), Vt, table, token ids, labels
  constexpr int QK_ELEMS = (2 * WN * LQ > WN * LP) ? 2 * WN * LQ : WN * LP;
  __shared__ __attribute__((aligned(16))) T s_qk[WPB][QK_ELEMS];
  __shared__ __attribute__((aligned(16))) T s_vt[WPB][WN * LQ];
  __shared__ float s_tbl[WPB][(2 * WS - 1) * (2 * WS - 1)];
  __shared__ int s_tok[WPB][WN];
  __shared__ int s_lab[WPB][WN];

  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  long long item = blockIdx.x * (long long)WPB + w;
  const bool live = item < g.items;
  if (!live) item = g.items - 1;
  const int C = g.heads * HD;
  const int nwx = g.res / WS, nW = nwx * nwx;
  const int h = (int)(item % g.heads);
  const long long bw = item / g.heads;
  const int win = (int)(bw % nW);
  const int b = (int)(bw / nW);
  const int wy = win / nwx, wx = win % nwx;

  s_tok[w][lane] = win_token(g.res, g.shift, wy, wx, lane);
  s_lab[w][lane] = win_label(g.res, g.shift, wy, wx, lane);
  for (int i = lane; i < (2 * WS - 1) * (2 * WS - 1); i += 64) s_tbl[w][i] = table[i * g.heads + h];
  __syncthreads();

  T* Qs = s_qk[w];
  T* Ks = s_qk[w] + WN * LQ;
  T* Vt = s_vt[w];
  const T* base = qkv + (long long)b * g.res * g.res * 3 * C;
  win_load<T>(Qs, LQ, base, 3 * C, 0 * C + h * HD, s_tok[w], lane);
  win_load<T>(Ks, LQ, base, 3 * C, 1 * C + h * HD, s_tok[w], lane);
  win_load<T>(Vt, LQ, base, 3 * C, 2 * C + h * HD, s_tok[w], lane);     // V token-major [key][d]; transposed in the fragment read
  __syncthreads();

  f32x4 s[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  mma_tile<T, 4, 4>(Qs, LQ, Ks, LQ, HD, lane, s);
  win_softmax(s, s_tbl[w], s_lab[w], g.shift > 0, rsqrtf((float)HD), lane);
  __syncthreads();                       // all Q/K fragment reads done before P overwrites the region
  T* Ps = s_qk[w];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) stf(Ps + (i * 16 + (lane >> 4) * 4 + r) * LP + j * 16 + (lane & 15), s[i][j][r]);
  __syncthreads();

  f32x4 o[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) o[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // O^T[m = d][n = query] = sum_key V[key][d] P[query][key]: V^T fragments by LDS transpose read, P fragments plain; the lane
  // then owns 4 consecutive d of query (16 i + lane&15) -> one vector store per fragment
  for (int k0 = 0; k0 < WN; k0 += Mma<T>::KSTEP) {
    typename Mma<T>::Frag pf[4], vf[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) pf[i] = Mma<T>::load(Ps, LP, i * 16, k0, lane);
#pragma unroll
    for (int j = 0; j < 2; ++j) vf[j] = Mma<T>::load_tr(Vt, LQ, j * 16, k0, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) o[i][j] = Mma<T>::mma(vf[j], pf[i], o[i][j]);
  }
  if (live) {
    T* ob = out + (long long)b * g.res * g.res * C + h * HD + (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      T* dst = ob + (long long)s_tok[w][i * 16 + (lane & 15)] * C;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float v[4] = {o[i][j][0], o[i][j][1], o[i][j][2], o[i][j][3]};
        st4(dst + j * 16, v);
      }
    }
  }
}

// Backward.  dqkv [B,N,3C] gets dq,dk,dv;  dtable [nparts][225,heads] (f32) accumulated atomically: block i adds into copy
// i % nparts and the caller sums the copies.  With ONE copy the 1536 blocks of a 64x64 stage queue 512 same-address atomics on
// each of the 675 table entries -- 34 of the kernel's 80 us (ablation, tools/probes/win_bwd_ablate.py).
template <typename T>
__global__ __launch_bounds__(64 * WinCfg<T>::WPB_BWD) void win_attn_bwd_kernel(const T* qkv, const float* table, const T* dout,
                                                                           T* dqkv, float* dtable, WinGeom g) {
  constexpr int WPB = WinCfg<T>::WPB_BWD;
  constexpr int LQ = HD + LdsPad<T>::P, LP = WN + LdsPad<T>::P;
  __shared__ __attribute__((aligned(16))) T s_q[WPB][WN * LQ];
  __shared__ __attribute__((aligned(16))) T s_k[WPB][WN * LQ];
  // V and P share one region: V is dead once dP = dO V^T is in registers, and only then is P (dV = P^T dO) written -- by the
  // same wave, in program order.  39 KB of LDS per wave instead of 45 KB: 4 resident waves per CU.
  __shared__ __attribute__((aligned(16))) T s_vp[WPB][WN * LP];
  __shared__ __attribute__((aligned(16))) T s_do[WPB][WN * LQ];
  __shared__ __attribute__((aligned(16))) T s_ds[WPB][WN * LP];
  T (*s_v)[WN * LP] = s_vp;
  T (*s_p)[WN * LP] = s_vp;
  __shared__ float s_tbl[WPB][(2 * WS - 1) * (2 * WS - 1)];
  __shared__ int s_tok[WPB][WN];
  __shared__ int s_lab[WPB][WN];

  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  long long item = blockIdx.x * (long long)WPB + w;
  const bool live = item < g.items;
  if (!live) item = g.items - 1;
  const int C = g.heads * HD;
  const int nwx = g.res / WS, nW = nwx * nwx;
  const int h = (int)(item % g.heads);
  const long long bw = item / g.heads;
  const int win = (int)(bw % nW);
  const int b = (int)(bw / nW);
  const int wy = win / nwx, wx = win % nwx;
  const float scale = rsqrtf((float)HD);

  s_tok[w][lane] = win_token(g.res, g.shift, wy, wx, lane);
  s_lab[w][lane] = win_label(g.res, g.shift, wy, wx, lane);
  for (int i = lane; i < (2 * WS - 1) * (2 * WS - 1); i += 64) s_tbl[w][i] = table[i * g.heads + h];
  __syncthreads();
  const T* base = qkv + (long long)b * g.res * g.res * 3 * C;
  win_load<T>(s_q[w], LQ, base, 3 * C, 0 * C + h * HD, s_tok[w], lane);
  win_load<T>(s_k[w], LQ, base, 3 * C, 1 * C + h * HD, s_tok[w], lane);
  win_load<T>(s_v[w], LQ, base, 3 * C, 2 * C + h * HD, s_tok[w], lane);
  win_load<T>(s_do[w], LQ, dout + (long long)b * g.res * g.res * C, C, h * HD, s_tok[w], lane);
  __syncthreads();

  f32x4 p[4][4], dp[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) { p[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  mma_tile<T, 4, 4>(s_q[w], LQ, s_k[w], LQ, HD, lane, p);
  win_softmax(p, s_tbl[w], s_lab[w], g.shift > 0, scale, lane);
  mma_tile<T, 4, 4>(s_do[w], LQ, s_v[w], LQ, HD, lane, dp);        // dP = dO V^T
  // dS = P * (dP - rowsum(dP*P)); stash P and dS (as T) for the three remaining products
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float d = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) d += dp[i][j][r] * p[i][j][r];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
      const int row = i * 16 + (lane >> 4) * 4 + r;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = j * 16 + (lane & 15);
        const float ds = p[i][j][r] * (dp[i][j][r] - d);
        stf(s_p[w] + row * LP + col, p[i][j][r]);
        stf(s_ds[w] + row * LP + col, ds);
      }
    }
  }
  __syncthreads();
  T* dbase = dqkv + (long long)b * g.res * g.res * 3 * C;
  f32x4 acc[4][2];
  auto zero = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  // operands are swapped in the MFMA (A = the d-side operand), so a lane ends up with 4 CONSECUTIVE d of one token:
  // out^T[m = d 16 j + 4 (lane>>4) + r][n = token 16 i + (lane&15)] -> one 8/16-byte global store per fragment
  auto store = [&](int which, float mul) {
    if (!live) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      T* dst = dbase + (long long)s_tok[w][i * 16 + (lane & 15)] * 3 * C + which * C + h * HD + (lane >> 4) * 4;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float v[4] = {acc[i][j][0] * mul, acc[i][j][1] * mul, acc[i][j][2] * mul, acc[i][j][3] * mul};
        st4(dst + j * 16, v);
      }
    }
  };
  // out[tok][d] = sum_k Aop(tok,k) * Bop(k,d).  ATR: Aop is stored [k][tok] (read with the LDS transpose read), else [tok][k];
  // Bop is always stored [k][d] (q, k, v, dO tiles are token-major) -> transpose read.  (v0 assembled every fragment from 8
  // ds_read_u16: 288 2-byte LDS reads per lane for the three products.)
  auto prod = [&](const T* Am, int lda, bool atr, const T* Bm) {
    for (int k0 = 0; k0 < WN; k0 += Mma<T>::KSTEP) {
      typename Mma<T>::Frag a[4], bb[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = atr ? Mma<T>::load_tr(Am, lda, i * 16, k0, lane) : Mma<T>::load(Am, lda, i * 16, k0, lane);
#pragma unroll
      for (int j = 0; j < 2; ++j) bb[j] = Mma<T>::load_tr(Bm, LQ, j * 16, k0, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = Mma<T>::mma(bb[j], a[i], acc[i][j]);
    }
  };
  // dV[key][d] = sum_q P[q][key] dO[q][d]             A(tok=key, k=q) = s_p [q][key]  -> transposed read
  zero(); prod(s_p[w], LP, true, s_do[w]); store(2, 1.f);
  // dQ[q][d] = scale * sum_key dS[q][key] K[key][d]   A(tok=q, k=key) = s_ds [q][key] -> plain read
  zero(); prod(s_ds[w], LP, false, s_k[w]); store(0, scale);
  // dK[key][d] = scale * sum_q dS[q][key] Q[q][d]     A(tok=key, k=q) = s_ds [q][key] -> transposed read
  zero(); prod(s_ds[w], LP, true, s_q[w]); store(1, scale);
  // relative-position-bias gradient: bin (dy, dx) collects dS[(ry,rx)][(ry-dy, rx-dx)] over the window.  Lane l owns the 8 x 8
  // block (query row ry = l / 8, key row ky = l % 8) of the stashed dS tile: all of it has dy = ry - ky, and its 15 diagonals are
  // the dx bins -- eight 16-byte LDS reads and 64 adds in registers, then 15 LDS adds into the (now dead) bias-table slot and one
  // global atomic per bin.  (v0: 4096 LDS atomics per window; v1: <= 64 dependent 2-byte reads per bin in a rolled loop, a
  // quarter of the kernel's time.)
  {
    const int ry = lane >> 3, ky = lane & 7;
    float dacc[2 * WS - 1];
#pragma unroll
    for (int d = 0; d < 2 * WS - 1; ++d) dacc[d] = 0.f;
#pragma unroll
    for (int rx = 0; rx < WS; ++rx) {
      const T* src = s_ds[w] + (ry * WS + rx) * LP + ky * WS;
      float v[WS];
      if constexpr (Vec<T>::N == 8) ld16(src, v);
      else { ld16(src, v); ld16(src + 4, v + 4); }
#pragma unroll
      for (int kx = 0; kx < WS; ++kx) dacc[rx - kx + WS - 1] += v[kx];
    }
    for (int i = lane; i < (2 * WS - 1) * (2 * WS - 1); i += 64) s_tbl[w][i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int d = 0; d < 2 * WS - 1; ++d) atomicAdd(&s_tbl[w][(ry - ky + WS - 1) * (2 * WS - 1) + d], dacc[d]);
    __syncthreads();
    if (live)
      for (int bin = lane; bin < (2 * WS - 1) * (2 * WS - 1); bin += 64)
        atomicAdd(dtable + (long long)(blockIdx.x % g.nparts) * (2 * WS - 1) * (2 * WS - 1) * g.heads + bin * g.heads + h, s_tbl[w][bin]);
  }
}

extern "C" int stj_win_attn_fwd(const void* qkv, const float* table, void* out, int B, int res, int heads, int shift,
                                int dtype, hipStream_t stream) {
  if (res % WS != 0 || shift < 0 || shift >= WS) { stj_set_error("win_attn: res %% 8 != 0 or bad shift"); return STJ_EINVAL; }
  if (((uintptr_t)qkv | (uintptr_t)out) & 15) { stj_set_error("win_attn: unaligned pointers"); return STJ_EINVAL; }
  WinGeom g; g.B = B; g.res = res; g.heads = heads; g.shift = shift; g.nparts = 1;
  g.items = (long long)B * (res / WS) * (res / WS) * heads;
  if (g.items <= 0) return STJ_OK;
  if (dtype == STJ_BF16) {
    constexpr int W = WinCfg<bf16>::WPB;
    hipLaunchKernelGGL(win_attn_fwd_kernel<bf16>, dim3((unsigned)((g.items + W - 1) / W)), dim3(64 * W), 0, stream, (const bf16*)qkv, table, (bf16*)out, g);
  } else if (dtype == STJ_F16) {
    constexpr int W = WinCfg<f16>::WPB;
    hipLaunchKernelGGL(win_attn_fwd_kernel<f16>, dim3((unsigned)((g.items + W - 1) / W)), dim3(64 * W), 0, stream, (const f16*)qkv, table, (f16*)out, g);
  } else {
    constexpr int W = WinCfg<float>::WPB;
    hipLaunchKernelGGL(win_attn_fwd_kernel<float>, dim3((unsigned)((g.items + W - 1) / W)), dim3(64 * W), 0, stream, (const float*)qkv, table, (float*)out, g);
  }
  return stj_check_launch("stj_win_attn_fwd");
}

extern "C" int stj_win_attn_bwd(const void* qkv, const float* table, const void* dout, void* dqkv, float* dtable, int nparts,
                                int B, int res, int heads, int shift, int dtype, hipStream_t stream) {
  if (res % WS != 0 || shift < 0 || shift >= WS) { stj_set_error("win_attn: res %% 8 != 0 or bad shift"); return STJ_EINVAL; }
  if (nparts < 1) { stj_set_error("win_attn_bwd: nparts must be >= 1"); return STJ_EINVAL; }
  WinGeom g; g.B = B; g.res = res; g.heads = heads; g.shift = shift; g.nparts = nparts;
  g.items = (long long)B * (res / WS) * (res / WS) * heads;
  if (g.items <= 0) return STJ_OK;
  if (dtype == STJ_BF16) {
    constexpr int W = WinCfg<bf16>::WPB_BWD;
    hipLaunchKernelGGL(win_attn_bwd_kernel<bf16>, dim3((unsigned)((g.items + W - 1) / W)), dim3(64 * W), 0, stream, (const bf16*)qkv, table, (const bf16*)dout, (bf16*)dqkv, dtable, g);
  } else if (dtype == STJ_F16) {
    constexpr int W = WinCfg<f16>::WPB_BWD;
    hipLaunchKernelGGL(win_attn_bwd_kernel<f16>, dim3((unsigned)((g.items + W - 1) / W)), dim3(64 * W), 0, stream, (const f16*)qkv, table, (const f16*)dout, (f16*)dqkv, dtable, g);
  } else {
    constexpr int W = WinCfg<float>::WPB_BWD;
    hipLaunchKernelGGL(win_attn_bwd_kernel<float>, dim3((unsigned)((g.items + W - 1) / W)), dim3(64 * W), 0, stream, (const float*)qkv, table, (const float*)dout, (float*)dqkv, dtable, g);
  }
  return stj_check_launch("stj_win_attn_bwd");
}
