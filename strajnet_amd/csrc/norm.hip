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

struct LnGroups { long long group_rows; int ngroups; long long gstride; int S; long long L; };   // S sub-runs of L rows per run (bwd)

template <typename T, int NPL>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* x, const float* gamma, const float* beta, T* y,
                                                     float* mean, float* rstd, long long rows, int C, float eps,
                                                     int gres, int C0, LnGroups G) {
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
      if (c < C) stf(y + row * C + c, (v[j] - mu) * rs * gamma[goff + c] + beta[goff + c]);
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
                                                     long long rows, int C, int gres, int C0, LnGroups G, int nb) {
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
          if (c < C) stf(dx + ln_src(row, c, C, gres, C0), rs[u] * (gg[j] - s1 - xh[j] * s2));
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NPL; ++j) { atomicAdd(&red[0][lane + 64 * j], ag[j]); atomicAdd(&red[1][lane + 64 * j], ab[j]); }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 64 * LNB_WAVES) {
    atomicAdd(dgamma + goff + c, red[0][c]);
    atomicAdd(dbeta + goff + c, red[1][c]);
  }
}

#define LN_DISPATCH(NPLV)                                                                                         \
  if (fwd) hipLaunchKernelGGL((ln_fwd_kernel<T, NPLV>), dim3(grid), dim3(256), 0, stream, (const T*)x, gamma, beta, \
                              (T*)y, mean, rstd, rows, C, eps, gres, C0, G);                                       \
  else hipLaunchKernelGGL((ln_bwd_kernel<T, NPLV>), dim3(grid), dim3(64 * LNB_WAVES), 0, stream, (const T*)dy, (const T*)x,    \
                          gamma, mean, rstd, (T*)dx, dgamma, dbeta, rows, C, gres, C0, G, nb);

template <typename T>
static int ln_launch(bool fwd, const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                     const void* dy, void* dx, float* dgamma, float* dbeta, long long rows, int C, float eps, int gres,
                     int C0, LnGroups G, hipStream_t stream) {
  const int npl = (C + 63) / 64;
  int grid, nb = 1;
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
  if (npl <= 2) { LN_DISPATCH(2) }
  else if (npl <= 3) { LN_DISPATCH(3) }
  else if (npl <= 6) { LN_DISPATCH(6) }
  else if (npl <= 8) { LN_DISPATCH(8) }
  else if (npl <= 12) { LN_DISPATCH(12) }
  else if (npl <= 24) { LN_DISPATCH(24) }
  else { stj_set_error("layernorm: C=%d > 1536 unsupported", C); return STJ_EUNSUPPORTED; }
  return stj_check_launch("stj_layernorm");
}

// group_rows/ngroups/gstride: parameter groups (ngroups <= 1: one gamma/beta for all rows).
extern "C" int stj_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                 long long rows, int C, float eps, int gather_res, int C0, long long group_rows, int ngroups,
                                 long long gstride, int dtype, hipStream_t stream) {
  if (rows <= 0) return STJ_OK;
  if (gather_res && (C != 4 * C0 || (gather_res & 1))) { stj_set_error("layernorm: bad gather geometry"); return STJ_EINVAL; }
  if (ngroups > 1 && group_rows <= 0) { stj_set_error("layernorm: bad group_rows"); return STJ_EINVAL; }
  LnGroups G; G.group_rows = group_rows > 0 ? group_rows : rows; G.ngroups = ngroups > 1 ? ngroups : 1; G.gstride = gstride; G.S = 1; G.L = G.group_rows;
  if (dtype == STJ_BF16) return ln_launch<bf16>(true, x, gamma, beta, y, mean, rstd, nullptr, nullptr, nullptr, nullptr, rows, C, eps, gather_res, C0, G, stream);
  return ln_launch<float>(true, x, gamma, beta, y, mean, rstd, nullptr, nullptr, nullptr, nullptr, rows, C, eps, gather_res, C0, G, stream);
}
extern "C" int stj_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                 void* dx, float* dgamma, float* dbeta, long long rows, int C, int gather_res, int C0,
                                 long long group_rows, int ngroups, long long gstride, int dtype, hipStream_t stream) {
  if (rows <= 0) return STJ_OK;
  LnGroups G; G.group_rows = group_rows > 0 ? group_rows : rows; G.ngroups = ngroups > 1 ? ngroups : 1; G.gstride = gstride; G.S = 1; G.L = G.group_rows;
  if (dtype == STJ_BF16) return ln_launch<bf16>(false, x, gamma, nullptr, nullptr, const_cast<float*>(mean), const_cast<float*>(rstd), dy, dx, dgamma, dbeta, rows, C, 0.f, gather_res, C0, G, stream);
  return ln_launch<float>(false, x, gamma, nullptr, nullptr, const_cast<float*>(mean), const_cast<float*>(rstd), dy, dx, dgamma, dbeta, rows, C, 0.f, gather_res, C0, G, stream);
}
