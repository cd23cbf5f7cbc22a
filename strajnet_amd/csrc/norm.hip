// LayerNorm forward / backward, one wavefront per row, reductions by wave shuffles.
// Keras LayerNormalization semantics (biased variance, eps inside sqrt): reference modules.py:179,184,272,433,
// 517,557 (eps 1e-5) and FG_MSA.py:52, trajNet.py:72-73,110-111,206-207 (eps 1e-3).
// Optional fused PatchMerging gather (reference modules.py:282-287): logical row (b,i,j) of width 4*C0 is
// the concat [x(2i,2j), x(2i+1,2j), x(2i,2j+1), x(2i+1,2j+1)] of a [B,res,res,C0] map -- no concat copy.
// Parameter groups: rows are cut into runs of `group_rows`; run c uses gamma/beta number (c % ngroups), found at
// gamma + g * gstride.  This batches the 8 per-waypoint LayerNorms of the time-separated cross-attentions
// (reference trajNet.py:206-207,257: 8 Cross_AttentionT layers with their own norm1/norm2) into one launch.
#include "common.h"

__device__ __forceinline__ long long ln_src(long long row, int c, int C, int gres, int C0) {
  if (gres == 0) return row * C + c;
  const int half = gres >> 1;
  const int j = (int)(row % half); const long long t = row / half;
  const int i = (int)(t % half); const long long b = t / half;
  const int q = c / C0, cc = c - q * C0;
  const int di = q & 1, dj = q >> 1;
  return ((b * gres + 2 * i + di) * gres + 2 * j + dj) * C0 + cc;
}

struct LnGroups { long long group_rows; int ngroups; long long gstride; int S; long long L;   // S sub-runs of L rows per run (bwd)
                  int nparts; long long pstride; };    // bwd: dgamma / dbeta copies the blocks spread their atomics over

template <typename T, int NPL>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* x, const float* gamma, const float* beta, T* y,
                                                     float* mean, float* rstd, long long rows, int C, float eps,
                                                     int gres, int C0, LnGroups G, const T* res) {
  const int lane = threadIdx.x & 63;
  const long long wave = blockIdx.x * 4ll + (threadIdx.x >> 6);
  for (long long row = wave; row < rows; row += gridDim.x * 4ll) {
    const long long goff = G.ngroups > 1 ? ((row / G.group_rows) % G.ngroups) * G.gstride : 0;
    float v[NPL];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const int c = lane + 64 * j;
      v[j] = c < C ? ldf(x + ln_src(row, c, C, gres, C0)) : 0.f;
      s += v[j];
    }
    const float mu = wave_sum(s) / C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const int c = lane + 64 * j;
      const float d = c < C ? v[j] - mu : 0.f;
      q += d * d;
    }
    const float rs = rsqrtf(wave_sum(q) / C + eps);
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const int c = lane + 64 * j;
      if (c < C) stf(y + row * C + c, (v[j] - mu) * rs * gamma[goff + c] + beta[goff + c] + (res ? ldf(res + row * C + c) : 0.f));
    }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
  }
}

// grid = ngroups * nb blocks: block (g, sub) walks the row runs c with c % ngroups == g, so its per-lane dgamma/dbeta
// partials belong to one parameter group; one atomic per (block, channel) at the end (nb bounds the contention).
#define LNB_WAVES 16
template <typename T, int NPL>
__global__ __launch_bounds__(64 * LNB_WAVES) void ln_bwd_kernel(const T* dy, const T* x, const float* gamma, const float* mean,
                                                     const float* rstd, T* dx, float* dgamma, float* dbeta,
                                                     long long rows, int C, int gres, int C0, LnGroups G, int nb, const T* dres) {
  __shared__ float red[2][64 * NPL];     // block-level dgamma/dbeta partials (LDS atomics), then ONE global atomic per channel
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int c = threadIdx.x; c < 2 * 64 * NPL; c += 64 * LNB_WAVES) (&red[0][0])[c] = 0.f;
  __syncthreads();
  const int g = blockIdx.x % G.ngroups, sub = blockIdx.x / G.ngroups;
  const long long goff = (long long)g * G.gstride;
  const long long nruns = (rows + G.group_rows - 1) / G.group_rows;
  const long long nr_g = (nruns - g + G.ngroups - 1) / G.ngroups;      // runs of this group
  float ag[NPL], ab[NPL];
#pragma unroll
  for (int j = 0; j < NPL; ++j) { ag[j] = 0.f; ab[j] = 0.f; }
  for (long long jj = sub; jj < nr_g * G.S; jj += nb) {
    const long long run = g + (long long)G.ngroups * (jj / G.S);
    const long long part = jj % G.S;
    const long long chunk_begin = run * G.group_rows + part * G.L;
    const long long rend = min(rows, run * G.group_rows + min(G.group_rows, (part + 1) * G.L));
    // RI rows in flight per wave: all loads of the RI rows are issued before any reduction (memory-level parallelism;
    // one row per iteration was latency-bound at ~1 us per row)
    constexpr int RI = NPL <= 3 ? 4 : (NPL <= 6 ? 2 : 1);
    for (long long row0 = chunk_begin + w * RI; row0 < rend; row0 += LNB_WAVES * RI) {
      float dv[RI][NPL], xv[RI][NPL], mu[RI], rs[RI];
#pragma unroll
      for (int u = 0; u < RI; ++u) {
        const long long row = row0 + u;
        const bool ok = row < rend;
        mu[u] = ok ? mean[row] : 0.f; rs[u] = ok ? rstd[row] : 0.f;
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
          const int c = lane + 64 * j;
          const bool in = ok && c < C;
          dv[u][j] = in ? ldf(dy + row * C + c) : 0.f;
          xv[u][j] = in ? ldf(x + ln_src(row, c, C, gres, C0)) : 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < RI; ++u) {
        const long long row = row0 + u;
        if (row >= rend) continue;
        float xh[NPL], gg[NPL];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
          const int c = lane + 64 * j;
          if (c < C) {
            xh[j] = (xv[u][j] - mu[u]) * rs[u];
            gg[j] = dv[u][j] * gamma[goff + c];
            ag[j] += dv[u][j] * xh[j];
            ab[j] += dv[u][j];
          } else { xh[j] = 0.f; gg[j] = 0.f; }
          s1 += gg[j];
          s2 += gg[j] * xh[j];
        }
        s1 = wave_sum(s1) / C;
        s2 = wave_sum(s2) / C;
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
          const int c = lane + 64 * j;
          if (c < C) stf(dx + ln_src(row, c, C, gres, C0), rs[u] * (gg[j] - s1 - xh[j] * s2) + (dres ? ldf(dres + row * C + c) : 0.f));
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NPL; ++j) { atomicAdd(&red[0][lane + 64 * j], ag[j]); atomicAdd(&red[1][lane + 64 * j], ab[j]); }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 64 * LNB_WAVES) {
    atomicAdd(dgamma + (blockIdx.x / G.ngroups % G.nparts) * G.pstride + goff + c, red[0][c]);
    atomicAdd(dbeta + (blockIdx.x / G.ngroups % G.nparts) * G.pstride + goff + c, red[1][c]);
  }
}

// =====================================================================================================
// v2 (16-byte vector) kernels.  v1 above reads 2-byte elements (lane = column), one row per wave at a time: backward ran at
// 0.5-1 TB/s, spilled for C >= 384 (1024-thread blocks cap a wave at 128 VGPRs) and cost ~40 us even for 128 rows.
// v2: a row is covered by LPR lanes x NCHK 16-byte chunks (C = 96: 12 of 16 lanes busy, 4 rows per wave pass; 384: 48 of 64
// lanes), U passes are in flight per wave, row reductions are xor-shuffles inside the LPR-lane group, 256-thread blocks.
// =====================================================================================================
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// source offset of the VN-element chunk starting at logical column c (a chunk never straddles PatchMerging quadrants: C0 % VN == 0)
template <typename T, int LPR, int NCHK>
__global__ __launch_bounds__(256) void ln_fwd2_kernel(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      T* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd,
                                                      long long rows, int C, float eps, int gres, int C0, LnGroups G,
                                                      const T* __restrict__ res) {
  constexpr int VN = Vec<T>::N, RPW = 64 / LPR;
  constexpr int U = NCHK == 1 ? 4 : (NCHK == 2 ? 2 : 1);
  const int lane = threadIdx.x & 63, sl = lane % LPR, rsub = lane / LPR;
  const long long wave = blockIdx.x * 4ll + (threadIdx.x >> 6), nwaves = gridDim.x * 4ll;
  const float invC = 1.f / C;
  for (long long row0 = wave * (U * RPW); row0 < rows; row0 += nwaves * (U * RPW)) {
    float v[U][NCHK][VN];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long row = row0 + u * RPW + rsub;
#pragma unroll
      for (int k = 0; k < NCHK; ++k) {
        const int c = (sl + LPR * k) * VN;
        if (row < rows && c < C) ld16(x + ln_src(row, c, C, gres, C0), v[u][k]);
        else {
#pragma unroll
          for (int e = 0; e < VN; ++e) v[u][k][e] = 0.f;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long row = row0 + u * RPW + rsub;
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < NCHK; ++k)
#pragma unroll
        for (int e = 0; e < VN; ++e) s += v[u][k][e];
      const float mu = group_sum<LPR>(s) * invC;
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < NCHK; ++k) {
        const int c = (sl + LPR * k) * VN;
#pragma unroll
        for (int e = 0; e < VN; ++e) { const float d = c < C ? v[u][k][e] - mu : 0.f; q += d * d; }
      }
      const float rs = rsqrtf(group_sum<LPR>(q) * invC + eps);
      if (row >= rows) continue;
      const long long goff = G.ngroups > 1 ? ((row / G.group_rows) % G.ngroups) * G.gstride : 0;
#pragma unroll
      for (int k = 0; k < NCHK; ++k) {
        const int c = (sl + LPR * k) * VN;
        if (c >= C) continue;
        float o[VN];
#pragma unroll
        for (int e = 0; e < VN; e += 4) {
          const float4 gv = *reinterpret_cast<const float4*>(gamma + goff + c + e), bv = *reinterpret_cast<const float4*>(beta + goff + c + e);
          o[e] = (v[u][k][e] - mu) * rs * gv.x + bv.x; o[e + 1] = (v[u][k][e + 1] - mu) * rs * gv.y + bv.y;
          o[e + 2] = (v[u][k][e + 2] - mu) * rs * gv.z + bv.z; o[e + 3] = (v[u][k][e + 3] - mu) * rs * gv.w + bv.w;
        }
        if (res) {                       // y = LN(x) + res: the sum a caller would otherwise make in a pass of its own
          float r[VN];
          ld16(res + row * C + c, r);
#pragma unroll
          for (int e = 0; e < VN; ++e) o[e] += r[e];
        }
        st16(y + row * C + c, o);
      }
      if (sl == 0) { mean[row] = mu; rstd[row] = rs; }
    }
  }
}

template <typename T, int LPR, int NCHK>
__global__ __launch_bounds__(256) void ln_bwd2_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ gamma,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd, T* __restrict__ dx,
                                                      float* dgamma, float* dbeta, long long rows, int C, int gres, int C0, LnGroups G, int nb,
                                                      const T* __restrict__ dres) {
  constexpr int VN = Vec<T>::N, RPW = 64 / LPR;
  constexpr int U = NCHK == 1 ? 4 : (NCHK == 2 ? 2 : 1);
  __shared__ float red[2][LPR * NCHK * VN];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, sl = lane % LPR, rsub = lane / LPR;
  for (int c = threadIdx.x; c < 2 * LPR * NCHK * VN; c += 256) (&red[0][0])[c] = 0.f;
  __syncthreads();
  const int g = blockIdx.x % G.ngroups, sub = blockIdx.x / G.ngroups;
  const long long goff = (long long)g * G.gstride;
  const long long nruns = (rows + G.group_rows - 1) / G.group_rows;
  const long long nr_g = (nruns - g + G.ngroups - 1) / G.ngroups;
  const float invC = 1.f / C;
  float gam[NCHK][VN], ag[NCHK][VN], ab[NCHK][VN];
#pragma unroll
  for (int k = 0; k < NCHK; ++k) {
    const int c = (sl + LPR * k) * VN;
#pragma unroll
    for (int e = 0; e < VN; ++e) { gam[k][e] = c < C ? gamma[goff + c + e] : 0.f; ag[k][e] = 0.f; ab[k][e] = 0.f; }
  }
  for (long long jj = sub; jj < nr_g * G.S; jj += nb) {
    const long long run = g + (long long)G.ngroups * (jj / G.S);
    const long long part = jj % G.S;
    const long long chunk_begin = run * G.group_rows + part * G.L;
    const long long rend = min(rows, run * G.group_rows + min(G.group_rows, (part + 1) * G.L));
    for (long long row0 = chunk_begin + (long long)w * (U * RPW); row0 < rend; row0 += 4 * (U * RPW)) {
      float dv[U][NCHK][VN], xv[U][NCHK][VN], mu[U], rs[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long row = row0 + u * RPW + rsub;
        const bool ok = row < rend;
        mu[u] = ok ? mean[row] : 0.f; rs[u] = ok ? rstd[row] : 0.f;
#pragma unroll
        for (int k = 0; k < NCHK; ++k) {
          const int c = (sl + LPR * k) * VN;
          if (ok && c < C) { ld16(dy + row * C + c, dv[u][k]); ld16(x + ln_src(row, c, C, gres, C0), xv[u][k]); }
          else {
#pragma unroll
            for (int e = 0; e < VN; ++e) { dv[u][k][e] = 0.f; xv[u][k][e] = 0.f; }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long row = row0 + u * RPW + rsub;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < NCHK; ++k) {
          const int c = (sl + LPR * k) * VN;
#pragma unroll
          for (int e = 0; e < VN; ++e) {
            const float xh = c < C ? (xv[u][k][e] - mu[u]) * rs[u] : 0.f;      // rs = 0 on rows past the end
            const float gg = dv[u][k][e] * gam[k][e];
            ag[k][e] += dv[u][k][e] * xh;
            ab[k][e] += dv[u][k][e];
            xv[u][k][e] = xh; dv[u][k][e] = gg;
            s1 += gg; s2 += gg * xh;
          }
        }
        s1 = group_sum<LPR>(s1) * invC;
        s2 = group_sum<LPR>(s2) * invC;
        if (row >= rend) continue;
#pragma unroll
        for (int k = 0; k < NCHK; ++k) {
          const int c = (sl + LPR * k) * VN;
          if (c >= C) continue;
          float o[VN];
#pragma unroll
          for (int e = 0; e < VN; ++e) o[e] = rs[u] * (dv[u][k][e] - s1 - xv[u][k][e] * s2);
          if (dres) {              // gradient arriving over the residual connection that bypasses this LayerNorm (fused add)
            float rr[VN];
            ld16(dres + row * C + c, rr);
#pragma unroll
            for (int e = 0; e < VN; ++e) o[e] += rr[e];
          }
          st16(dx + ln_src(row, c, C, gres, C0), o);
        }
      }
    }
  }
  // dgamma / dbeta: sum the RPW row sub-groups of the wave, then the 4 waves through LDS, then one atomic per column and block
#pragma unroll
  for (int k = 0; k < NCHK; ++k)
#pragma unroll
    for (int e = 0; e < VN; ++e) {
      float a = ag[k][e], b = ab[k][e];
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
      if (rsub == 0) { atomicAdd(&red[0][(sl + LPR * k) * VN + e], a); atomicAdd(&red[1][(sl + LPR * k) * VN + e], b); }
    }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    atomicAdd(dgamma + (blockIdx.x / G.ngroups % G.nparts) * G.pstride + goff + c, red[0][c]);
    atomicAdd(dbeta + (blockIdx.x / G.ngroups % G.nparts) * G.pstride + goff + c, red[1][c]);
  }
}

// LPR lanes x NCHK chunks cover a row of C / VN 16-byte chunks
template <typename T, int LPR, int NCHK>
static void ln2_launch(bool fwd, int grid, hipStream_t stream, const void* x, const float* gamma, const float* beta, void* y, float* mean,
                       float* rstd, const void* dy, void* dx, float* dgamma, float* dbeta, long long rows, int C, float eps, int gres, int C0,
                       LnGroups G, int nb, const void* dres) {
  if (fwd) hipLaunchKernelGGL((ln_fwd2_kernel<T, LPR, NCHK>), dim3(grid), dim3(256), 0, stream, (const T*)x, gamma, beta, (T*)y, mean, rstd, rows, C, eps, gres, C0, G, (const T*)dres);
  else hipLaunchKernelGGL((ln_bwd2_kernel<T, LPR, NCHK>), dim3(grid), dim3(256), 0, stream, (const T*)dy, (const T*)x, gamma, mean, rstd, (T*)dx, dgamma, dbeta, rows, C, gres, C0, G, nb, (const T*)dres);
}

#define LN_DISPATCH(NPLV)                                                                                         \
  if (fwd) hipLaunchKernelGGL((ln_fwd_kernel<T, NPLV>), dim3(grid), dim3(256), 0, stream, (const T*)x, gamma, beta, \
                              (T*)y, mean, rstd, rows, C, eps, gres, C0, G, (const T*)dres);                       \
  else hipLaunchKernelGGL((ln_bwd_kernel<T, NPLV>), dim3(grid), dim3(64 * LNB_WAVES), 0, stream, (const T*)dy, (const T*)x,    \
                          gamma, mean, rstd, (T*)dx, dgamma, dbeta, rows, C, gres, C0, G, nb, (const T*)dres);

template <typename T>
static int ln_launch(bool fwd, const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                     const void* dy, void* dx, float* dgamma, float* dbeta, long long rows, int C, float eps, int gres,
                     int C0, LnGroups G, hipStream_t stream, const void* dres = nullptr) {
  const int npl = (C + 63) / 64;
  int grid, nb = 1;
  {
    // v2 path: whole 16-byte chunks, aligned rows, gather quadrants a multiple of the chunk, parameter sets 16-byte aligned
    constexpr int VN = Vec<T>::N;
    const bool v2 = true;          // (the scalar-row kernels below serve the widths / alignments this form does not take)
    const int chunks = C / VN;
    const bool al = ((uintptr_t)x % 16 == 0) && (!fwd || ((uintptr_t)y % 16 == 0 && (uintptr_t)dres % 16 == 0)) && (fwd || ((uintptr_t)dy % 16 == 0 && (uintptr_t)dx % 16 == 0 && (uintptr_t)dres % 16 == 0)) &&
                    ((uintptr_t)gamma % 16 == 0) && (fwd ? (uintptr_t)beta % 16 == 0 : true) && (G.gstride % 4 == 0);
    if (v2 && C % VN == 0 && (gres == 0 || C0 % VN == 0) && chunks <= 192 && al) {
      LnGroups G2 = G;
      if (fwd) {
        const int lpr = chunks <= 16 ? 16 : (chunks <= 32 ? 32 : 64);
        const int rows_per_wave = (64 / lpr) * (chunks <= 64 ? 4 : (chunks <= 128 ? 2 : 1));
        long long want = (rows + 4ll * rows_per_wave - 1) / (4ll * rows_per_wave);
        grid = (int)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
      } else {
        if (G2.ngroups <= 1) { G2.ngroups = 1; G2.gstride = 0; G2.group_rows = rows; }
        const long long nruns = (rows + G2.group_rows - 1) / G2.group_rows;
        const long long nr = (nruns + G2.ngroups - 1) / G2.ngroups;
        // one pass of a block = 4 waves x U passes x (64 / LPR) rows; at most 256 blocks share the dgamma/dbeta atomics
        // (measured on 32768 x 96: 16.4 us with 256 blocks, 21.6 us with 512)
        const int lpr2 = chunks <= 16 ? 16 : (chunks <= 32 ? 32 : 64);
        const int rows_per_pass = 4 * (chunks <= 64 ? 4 : (chunks <= 128 ? 2 : 1)) * (64 / lpr2);
        long long nbt = (rows / G2.ngroups + rows_per_pass - 1) / rows_per_pass;
        if (nbt > 256 / G2.ngroups) nbt = 256 / G2.ngroups;
        if (nbt < 1) nbt = 1;
        long long S = (nbt + nr - 1) / nr;
        if (S < 1) S = 1;
        G2.S = (int)S;
        G2.L = (G2.group_rows + S - 1) / S;
        nb = (int)(nbt < nr * S ? nbt : nr * S);
        grid = G2.ngroups * nb;
      }
#define LN2(LPRV, NCHKV) ln2_launch<T, LPRV, NCHKV>(fwd, grid, stream, x, gamma, beta, y, mean, rstd, dy, dx, dgamma, dbeta, rows, C, eps, gres, C0, G2, nb, dres)
      if (chunks <= 16) LN2(16, 1);
      else if (chunks <= 32) LN2(32, 1);
      else if (chunks <= 64) LN2(64, 1);
      else if (chunks <= 128) LN2(64, 2);
      else LN2(64, 3);
#undef LN2
      return stj_check_launch("stj_layernorm(v2)");
    }
  }
  if (fwd) {
    long long want = (rows + 3) / 4;
    grid = (int)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
  } else {
    if (G.ngroups <= 1) { G.ngroups = 1; G.gstride = 0; G.group_rows = rows; }
    // <= ~64 blocks (of 16 waves) in total share the dgamma/dbeta global atomics (same-address atomics serialise: 256
    // contending blocks cost ~25 us per launch); every run is cut into S sub-runs of L rows
    const long long nruns = (rows + G.group_rows - 1) / G.group_rows;
    const long long nr = (nruns + G.ngroups - 1) / G.ngroups;
    long long nbt = (rows / G.ngroups + 63) / 64;
    if (nbt > 256 / G.ngroups) nbt = 256 / G.ngroups;
    if (nbt < 1) nbt = 1;
    long long S = (nbt + nr - 1) / nr;
    if (S < 1) S = 1;
    G.S = (int)S;
    G.L = (G.group_rows + S - 1) / S;
    nb = (int)(nbt < nr * S ? nbt : nr * S);
    grid = G.ngroups * nb;
  }
  // ONE register width (24 values per lane, C <= 1536): this form serves the widths and alignments the chunked kernels refuse, not a
  // timed path, and each instantiation is 50-100 KB of code (six widths x three storage types x forward / backward were 2 MB of the library)
  if (npl <= 24) { LN_DISPATCH(24) }
  else { stj_set_error("layernorm: C=%d > 1536 unsupported", C); return STJ_EUNSUPPORTED; }
  return stj_check_launch("stj_layernorm");
}

// =====================================================================================================
// Two LayerNorm backward passes in one launch: the stem's  y = LN2(LN1(x1) [+ add])  (modules.py:437-446 + :578 / :590; the forward is
// csrc/patch_embed.hip).  d2 = dLN2(dy) at x2 = LN1(x1) [+ add]  (= the gradient of `add`, written when asked for);  dx1 = dLN1(d2) at x1.
// d2 is rounded to the storage type before it is used, exactly as the two separate launches hand it over.  Rows are covered as in
// ln_bwd2_kernel (LPR lanes x one 16-byte chunk, U rows per lane group in flight); plain [rows, C] tensors, one parameter set per norm.
// =====================================================================================================
template <typename T> __device__ __forceinline__ float ln_rnd(float x) { T t; stf(&t, x); return ldf(&t); }
template <typename T, int LPR>
__global__ __launch_bounds__(256) void ln_bwd_chain_kernel(const T* __restrict__ dy, const T* __restrict__ x2, const float* __restrict__ gamma2,
                                                           const float* __restrict__ mean2, const float* __restrict__ rstd2,
                                                           const T* __restrict__ x1, const float* __restrict__ gamma1,
                                                           const float* __restrict__ mean1, const float* __restrict__ rstd1,
                                                           T* __restrict__ d2out, T* __restrict__ dx1, float* dg2, float* db2, float* dg1, float* db1,
                                                           long long rows, int C, int np2, long long ps2, int np1, long long ps1) {
  constexpr int VN = Vec<T>::N, RPW = 64 / LPR, U = 2;
  __shared__ float red[4][LPR * VN];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, sl = lane % LPR, rsub = lane / LPR;
  for (int c = threadIdx.x; c < 4 * LPR * VN; c += 256) (&red[0][0])[c] = 0.f;
  __syncthreads();
  const int c0 = sl * VN;
  const bool col = c0 < C;
  const float invC = 1.f / C;
  float g2[VN], g1[VN], a2[VN], b2[VN], a1[VN], b1[VN];
#pragma unroll
  for (int e = 0; e < VN; ++e) { g2[e] = col ? gamma2[c0 + e] : 0.f; g1[e] = col ? gamma1[c0 + e] : 0.f; a2[e] = b2[e] = a1[e] = b1[e] = 0.f; }
  for (long long row0 = ((long long)blockIdx.x * 4 + w) * (U * RPW); row0 < rows; row0 += (long long)gridDim.x * 4 * (U * RPW)) {
    float dv[U][VN], xa[U][VN], xb[U][VN], m2[U], r2[U], m1[U], r1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long row = row0 + u * RPW + rsub;
      const bool ok = row < rows;
      m2[u] = ok ? mean2[row] : 0.f; r2[u] = ok ? rstd2[row] : 0.f; m1[u] = ok ? mean1[row] : 0.f; r1[u] = ok ? rstd1[row] : 0.f;
      if (ok && col) { ld16(dy + row * C + c0, dv[u]); ld16(x2 + row * C + c0, xa[u]); ld16(x1 + row * C + c0, xb[u]); }
      else {
#pragma unroll
        for (int e = 0; e < VN; ++e) { dv[u][e] = 0.f; xa[u][e] = 0.f; xb[u][e] = 0.f; }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long row = row0 + u * RPW + rsub;
      const bool ok = row < rows;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int e = 0; e < VN; ++e) {
        const float xh = col ? (xa[u][e] - m2[u]) * r2[u] : 0.f;              // r = 0 on rows past the end
        const float gg = dv[u][e] * g2[e];
        a2[e] += dv[u][e] * xh; b2[e] += dv[u][e];
        xa[u][e] = xh; dv[u][e] = gg;
        s1 += gg; s2 += gg * xh;
      }
      s1 = group_sum<LPR>(s1) * invC; s2 = group_sum<LPR>(s2) * invC;
      float d[VN];
#pragma unroll
      for (int e = 0; e < VN; ++e) d[e] = ln_rnd<T>(r2[u] * (dv[u][e] - s1 - xa[u][e] * s2));      // as stored by a launch of its own
      if (ok && col && d2out) st16(d2out + row * C + c0, d);
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int e = 0; e < VN; ++e) {
        const float dd = (ok && col) ? d[e] : 0.f;
        const float xh = col ? (xb[u][e] - m1[u]) * r1[u] : 0.f;
        const float gg = dd * g1[e];
        a1[e] += dd * xh; b1[e] += dd;
        xb[u][e] = xh; d[e] = gg;
        t1 += gg; t2 += gg * xh;
      }
      t1 = group_sum<LPR>(t1) * invC; t2 = group_sum<LPR>(t2) * invC;
      if (ok && col) {
        float o[VN];
#pragma unroll
        for (int e = 0; e < VN; ++e) o[e] = r1[u] * (d[e] - t1 - xb[u][e] * t2);
        st16(dx1 + row * C + c0, o);
      }
    }
  }
  // the four parameter gradients: row sub-groups of the wave, then the waves through LDS, then one atomic per column and block
#pragma unroll
  for (int e = 0; e < VN; ++e) {
    float v[4] = {a2[e], b2[e], a1[e], b1[e]};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) v[k] += __shfl_xor(v[k], o, 64);
      if (rsub == 0) atomicAdd(&red[k][c0 + e], v[k]);
    }
  }
  __syncthreads();
  const long long o2 = (long long)(blockIdx.x % np2) * ps2, o1 = (long long)(blockIdx.x % np1) * ps1;
  for (int c = threadIdx.x; c < C; c += 256) {
    atomicAdd(dg2 + o2 + c, red[0][c]); atomicAdd(db2 + o2 + c, red[1][c]);
    atomicAdd(dg1 + o1 + c, red[2][c]); atomicAdd(db1 + o1 + c, red[3][c]);
  }
}
template <typename T>
static int ln_chain_launch(const void* dy, const void* x2, const float* gamma2, const float* mean2, const float* rstd2, const void* x1,
                           const float* gamma1, const float* mean1, const float* rstd1, void* d2, void* dx1, float* dg2, float* db2, float* dg1,
                           float* db1, long long rows, int C, int np2, long long ps2, int np1, long long ps1, hipStream_t st) {
  constexpr int VN = Vec<T>::N;
  const int chunks = C / VN;
  const int lpr = chunks <= 16 ? 16 : (chunks <= 32 ? 32 : 64);
  const long long per_pass = 4ll * 2 * (64 / lpr);                 // rows of one pass of a workgroup
  long long nb = (rows + per_pass - 1) / per_pass;
  if (nb > 256) nb = 256;                                          // (as stj_layernorm_bwd: more workgroups only share the parameter-gradient atomics)
#define LNC(L) hipLaunchKernelGGL((ln_bwd_chain_kernel<T, L>), dim3((unsigned)nb), dim3(256), 0, st, (const T*)dy, (const T*)x2, gamma2, mean2, rstd2, \
                                  (const T*)x1, gamma1, mean1, rstd1, (T*)d2, (T*)dx1, dg2, db2, dg1, db1, rows, C, np2, ps2, np1, ps1)
  if (lpr == 16) LNC(16); else if (lpr == 32) LNC(32); else LNC(64);
#undef LNC
  return stj_check_launch("stj_layernorm_bwd_chain");
}
extern "C" int stj_layernorm_bwd_chain_supported(int C, int dtype) {
  const int vn = dtype == STJ_F32 ? 4 : 8;
  return stj_dtype_ok(dtype) && C > 0 && C % vn == 0 && C / vn <= 64;
}
extern "C" int stj_layernorm_bwd_chain(const void* dy, const void* x2, const float* gamma2, const float* mean2, const float* rstd2, const void* x1,
                                       const float* gamma1, const float* mean1, const float* rstd1, void* d2, void* dx1, float* dgamma2,
                                       float* dbeta2, float* dgamma1, float* dbeta1, long long rows, int C, int nparts2, long long part_stride2,
                                       int nparts1, long long part_stride1, int dtype, hipStream_t stream) {
  if (rows <= 0) return STJ_OK;
  if (!stj_layernorm_bwd_chain_supported(C, dtype)) { stj_set_error("layernorm_bwd_chain: C = %d / dtype %d not built (C a multiple of the 16-byte vector, at most 64 vectors)", C, dtype); return STJ_EUNSUPPORTED; }
  if (!dy || !x2 || !x1 || !dx1 || !gamma2 || !gamma1 || !mean2 || !rstd2 || !mean1 || !rstd1 || !dgamma2 || !dbeta2 || !dgamma1 || !dbeta1 || nparts2 < 1 ||
      nparts1 < 1 || (nparts2 > 1 && part_stride2 < C) || (nparts1 > 1 && part_stride1 < C) ||
      (((uintptr_t)dy | (uintptr_t)x2 | (uintptr_t)x1 | (uintptr_t)dx1 | (uintptr_t)d2) & 15)) {
    stj_set_error("layernorm_bwd_chain: bad arguments (null / unaligned pointer, nparts / part_stride)"); return STJ_EINVAL;
  }
  if (dtype == STJ_BF16) return ln_chain_launch<bf16>(dy, x2, gamma2, mean2, rstd2, x1, gamma1, mean1, rstd1, d2, dx1, dgamma2, dbeta2, dgamma1, dbeta1, rows, C, nparts2, part_stride2, nparts1, part_stride1, stream);
  if (dtype == STJ_F16) return ln_chain_launch<f16>(dy, x2, gamma2, mean2, rstd2, x1, gamma1, mean1, rstd1, d2, dx1, dgamma2, dbeta2, dgamma1, dbeta1, rows, C, nparts2, part_stride2, nparts1, part_stride1, stream);
  return ln_chain_launch<float>(dy, x2, gamma2, mean2, rstd2, x1, gamma1, mean1, rstd1, d2, dx1, dgamma2, dbeta2, dgamma1, dbeta1, rows, C, nparts2, part_stride2, nparts1, part_stride1, stream);
}

// group_rows/ngroups/gstride: parameter groups (ngroups <= 1: one gamma/beta for all rows).
extern "C" int stj_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                 long long rows, int C, float eps, int gather_res, int C0, long long group_rows, int ngroups,
                                 long long gstride, int dtype, hipStream_t stream) {
  if (rows <= 0) return STJ_OK;
  if (gather_res && (C != 4 * C0 || (gather_res & 1))) { stj_set_error("layernorm: bad gather geometry"); return STJ_EINVAL; }
  if (ngroups > 1 && group_rows <= 0) { stj_set_error("layernorm: bad group_rows"); return STJ_EINVAL; }
  LnGroups G; G.group_rows = group_rows > 0 ? group_rows : rows; G.ngroups = ngroups > 1 ? ngroups : 1; G.gstride = gstride; G.S = 1; G.L = G.group_rows; G.nparts = 1; G.pstride = 0;
  if (dtype == STJ_BF16) return ln_launch<bf16>(true, x, gamma, beta, y, mean, rstd, nullptr, nullptr, nullptr, nullptr, rows, C, eps, gather_res, C0, G, stream);
  if (dtype == STJ_F16) return ln_launch<f16>(true, x, gamma, beta, y, mean, rstd, nullptr, nullptr, nullptr, nullptr, rows, C, eps, gather_res, C0, G, stream);
  return ln_launch<float>(true, x, gamma, beta, y, mean, rstd, nullptr, nullptr, nullptr, nullptr, rows, C, eps, gather_res, C0, G, stream);
}
// y = LayerNorm(x) * gamma + beta + res (res [rows, C], same type): LayerNorm followed by a sum with another branch
// (the stem's vec + maps before all_patch_norm's input, modules.py:589; Cross_AttentionT's output + query, trajNet.py:305-317).
extern "C" int stj_layernorm_res_fwd(const void* x, const float* gamma, const float* beta, const void* res, void* y, float* mean,
                                     float* rstd, long long rows, int C, float eps, long long group_rows, int ngroups,
                                     long long gstride, int dtype, hipStream_t stream) {
  if (rows <= 0) return STJ_OK;
  if (ngroups > 1 && group_rows <= 0) { stj_set_error("layernorm: bad group_rows"); return STJ_EINVAL; }
  LnGroups G; G.group_rows = group_rows > 0 ? group_rows : rows; G.ngroups = ngroups > 1 ? ngroups : 1; G.gstride = gstride; G.S = 1; G.L = G.group_rows; G.nparts = 1; G.pstride = 0;
  if (dtype == STJ_BF16) return ln_launch<bf16>(true, x, gamma, beta, y, mean, rstd, nullptr, nullptr, nullptr, nullptr, rows, C, eps, 0, 0, G, stream, res);
  if (dtype == STJ_F16) return ln_launch<f16>(true, x, gamma, beta, y, mean, rstd, nullptr, nullptr, nullptr, nullptr, rows, C, eps, 0, 0, G, stream, res);
  return ln_launch<float>(true, x, gamma, beta, y, mean, rstd, nullptr, nullptr, nullptr, nullptr, rows, C, eps, 0, 0, G, stream, res);
}
extern "C" int stj_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                 void* dx, float* dgamma, float* dbeta, long long rows, int C, int gather_res, int C0,
                                 long long group_rows, int ngroups, long long gstride, const void* dres, int nparts, long long part_stride,
                                 int dtype, hipStream_t stream) {
  if (rows <= 0) return STJ_OK;
  if (nparts < 1 || (nparts > 1 && part_stride < C)) { stj_set_error("layernorm_bwd: bad nparts / part_stride"); return STJ_EINVAL; }
  if (dres && gather_res) { stj_set_error("layernorm_bwd: dres with the PatchMerging gather is not supported"); return STJ_EINVAL; }
  LnGroups G; G.group_rows = group_rows > 0 ? group_rows : rows; G.ngroups = ngroups > 1 ? ngroups : 1; G.gstride = gstride; G.S = 1; G.L = G.group_rows; G.nparts = nparts; G.pstride = part_stride;
  if (dtype == STJ_BF16) return ln_launch<bf16>(false, x, gamma, nullptr, nullptr, const_cast<float*>(mean), const_cast<float*>(rstd), dy, dx, dgamma, dbeta, rows, C, 0.f, gather_res, C0, G, stream, dres);
  if (dtype == STJ_F16) return ln_launch<f16>(false, x, gamma, nullptr, nullptr, const_cast<float*>(mean), const_cast<float*>(rstd), dy, dx, dgamma, dbeta, rows, C, 0.f, gather_res, C0, G, stream, dres);
  return ln_launch<float>(false, x, gamma, nullptr, nullptr, const_cast<float*>(mean), const_cast<float*>(rstd), dy, dx, dgamma, dbeta, rows, C, 0.f, gather_res, C0, G, stream, dres);
}
