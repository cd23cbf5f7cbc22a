// Batched strided GEMM on MFMA with fused epilogue, plus column-sum (bias gradient).
//   C[z][m,n] = epi( alpha * sum_k A[z](m,k) * B[z](k,n) )      epi: +bias[n] -> act -> +res[m,n]
// Serves every Dense / 1x1-conv / tfa-MHA projection of the hot path and their dgrad / wgrad
// (reference: Keras Dense modules.py:36-37,76-79,270; FG_MSA.py:54-64; trajNet.py:32-36,71-76,195-210;
//  time-collapsed Conv3D modules.py:693-717 -- SURVEY.md K2-K8,K10).
// Operands may be given in either orientation through element strides (one of the two strides of
// each operand must be 1); tiles are staged into K-contiguous LDS images ([m][k], [n][k]) and fed
// to v_mfma_f32_16x16x32_bf16 / v_mfma_f32_16x16x4_f32.  Weight gradients use split-K with f32
// atomic accumulation straight into the flat gradient buffer.
#include "common.h"

struct GemmArgs {
  const void* A; const void* B; void* C; const float* bias; const void* res;
  int M, N, K, nb1, nb2;     // batch index z = z1 * nb2 + z2, each operand has a stride per level
  long long sAb1, sAb2, sAm, sAk, sBb1, sBb2, sBk, sBn, sCb1, sCb2, ldc, sBias1, sBias2, sRes1, sRes2, ldres;
  int act, c_f32, accumulate, splitk;
  int vecA, vecB;   // 16-byte vector path allowed for the contiguous dimension
  float alpha;
};

// stage a [ROWS][BK] K-contiguous LDS tile from a strided global operand.
//   element (r,k) at g[r*sR + k*sK];  rows valid: r < nr, k valid: k < nk.
template <typename T, int ROWS, int BK>
__device__ __forceinline__ void stage_tile(T* lds, int ld, const T* g, long long sR, long long sK, int nr, int nk,
                                           int vec, int tid) {
  constexpr int VN = Vec<T>::N;
  if (sK == 1) {                                // K contiguous: 16-byte chunks along k
    constexpr int CPR = BK / VN;
    for (int i = tid; i < ROWS * CPR; i += 256) {
      int r = i / CPR, k = (i % CPR) * VN;
      T* dst = lds + r * ld + k;
      if (r < nr && k + VN <= nk && vec) {
        *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(g + (long long)r * sR + k);
      } else {
#pragma unroll
        for (int e = 0; e < VN; ++e) {
          T z; if constexpr (sizeof(T) == 2) z.v = 0; else z = 0.f;
          dst[e] = (r < nr && k + e < nk) ? g[(long long)r * sR + k + e] : z;
        }
      }
    }
  } else if (sR == 1) {                         // row contiguous (transposed operand): chunks along rows
    constexpr int CPK = ROWS / VN;
    for (int i = tid; i < BK * CPK; i += 256) {
      int k = i / CPK, r = (i % CPK) * VN;
      __attribute__((aligned(16))) T tmp[VN];
      if (k < nk && r + VN <= nr && vec) {
        *reinterpret_cast<uint4*>(tmp) = *reinterpret_cast<const uint4*>(g + (long long)k * sK + r);
      } else {
#pragma unroll
        for (int e = 0; e < VN; ++e) {
          T z; if constexpr (sizeof(T) == 2) z.v = 0; else z = 0.f;
          tmp[e] = (k < nk && r + e < nr) ? g[(long long)k * sK + r + e] : z;
        }
      }
#pragma unroll
      for (int e = 0; e < VN; ++e) lds[(r + e) * ld + k] = tmp[e];
    }
  } else {                                      // general strides
    for (int i = tid; i < ROWS * BK; i += 256) {
      int r = i / BK, k = i % BK;
      T z; if constexpr (sizeof(T) == 2) z.v = 0; else z = 0.f;
      lds[r * ld + k] = (r < nr && k < nk) ? g[(long long)r * sR + (long long)k * sK] : z;
    }
  }
}

template <typename T, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs p) {
  constexpr int BK = 128 / sizeof(T);
  constexpr int LD = BK + LdsPad<T>::P;
  constexpr int FM = BM / (16 * WM), FN = BN / (16 * WN);
  __shared__ __attribute__((aligned(16))) T As[BM * LD];
  __shared__ __attribute__((aligned(16))) T Bs[BN * LD];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int z = blockIdx.y / p.splitk, ks = blockIdx.y % p.splitk;
  const int m0 = tm * BM, n0 = tn * BN;

  const int ktiles = (p.K + BK - 1) / BK;
  const int kt_per = (ktiles + p.splitk - 1) / p.splitk;
  const int kt0 = ks * kt_per, kt1 = min(ktiles, kt0 + kt_per);

  const long long z1 = z / p.nb2, z2 = z % p.nb2;
  const T* A = reinterpret_cast<const T*>(p.A) + z1 * p.sAb1 + z2 * p.sAb2 + (long long)m0 * p.sAm;
  const T* B = reinterpret_cast<const T*>(p.B) + z1 * p.sBb1 + z2 * p.sBb2 + (long long)n0 * p.sBn;

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int kt = kt0; kt < kt1; ++kt) {
    const int k0 = kt * BK;
    stage_tile<T, BM, BK>(As, LD, A + (long long)k0 * p.sAk, p.sAm, p.sAk, p.M - m0, p.K - k0, p.vecA, tid);
    stage_tile<T, BN, BK>(Bs, LD, B + (long long)k0 * p.sBk, p.sBn, p.sBk, p.N - n0, p.K - k0, p.vecB, tid);
    __syncthreads();
    mma_tile<T, FM, FN>(As + (wm * FM * 16) * LD, LD, Bs + (wn * FN * 16) * LD, LD, BK, lane, acc);
    __syncthreads();
  }
  if (kt0 >= kt1 && !(p.accumulate == 0 && ks == 0)) return;

  // epilogue
  const float* bias = p.bias ? p.bias + z1 * p.sBias1 + z2 * p.sBias2 : nullptr;
  const T* res = p.res ? reinterpret_cast<const T*>(p.res) + z1 * p.sRes1 + z2 * p.sRes2 : nullptr;
  float* Cf = reinterpret_cast<float*>(p.C) + z1 * p.sCb1 + z2 * p.sCb2;
  T* Ct = reinterpret_cast<T*>(p.C) + z1 * p.sCb1 + z2 * p.sCb2;
#pragma unroll
  for (int i = 0; i < FM; ++i) {
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int col = n0 + (wn * FN + j) * 16 + (lane & 15);
      if (col >= p.N) continue;
      const float bv = (bias && ks == 0) ? bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + (wm * FM + i) * 16 + (lane >> 4) * 4 + r;
        if (row >= p.M) continue;
        float v = acc[i][j][r] * p.alpha + bv;
        v = apply_act(v, p.act);
        if (res) v += ldf(res + (long long)row * p.ldres + col);
        const long long o = (long long)row * p.ldc + col;
        if (p.accumulate) atomicAdd(Cf + o, v);
        else if (p.c_f32) Cf[o] = v;
        else stf(Ct + o, v);
      }
    }
  }
}

template <typename T>
static int launch_gemm(const GemmArgs& p, hipStream_t st) {
  const long long tiles128 = (long long)((p.M + 127) / 128) * ((p.N + 127) / 128) * p.nb1 * p.nb2 * p.splitk;
  dim3 blk(256);
  if (p.N > 64 && p.M > 64 && tiles128 >= 128) {
    dim3 grid(((p.M + 127) / 128) * ((p.N + 127) / 128), p.nb1 * p.nb2 * p.splitk);
    hipLaunchKernelGGL((gemm_kernel<T, 128, 128, 2, 2>), grid, blk, 0, st, p);
  } else if (p.N <= 64 && p.M >= 4096) {
    dim3 grid(((p.M + 127) / 128) * ((p.N + 63) / 64), p.nb1 * p.nb2 * p.splitk);
    hipLaunchKernelGGL((gemm_kernel<T, 128, 64, 4, 1>), grid, blk, 0, st, p);
  } else {
    dim3 grid(((p.M + 63) / 64) * ((p.N + 63) / 64), p.nb1 * p.nb2 * p.splitk);
    hipLaunchKernelGGL((gemm_kernel<T, 64, 64, 2, 2>), grid, blk, 0, st, p);
  }
  return stj_check_launch("stj_gemm");
}

extern "C" int stj_gemm(const void* A, const void* B, void* C, const float* bias, const void* res,
                        int M, int N, int K, int nb1, int nb2,
                        long long sAb1, long long sAb2, long long sAm, long long sAk,
                        long long sBb1, long long sBb2, long long sBk, long long sBn,
                        long long sCb1, long long sCb2, long long ldc,
                        long long sBias1, long long sBias2, long long sRes1, long long sRes2, long long ldres,
                        int act, float alpha, int dtype, int c_f32, int accumulate, int splitk, hipStream_t stream) {
  if (M <= 0 || N <= 0 || nb1 <= 0 || nb2 <= 0) return STJ_OK;
  if (K < 0 || splitk < 1) { stj_set_error("stj_gemm: bad K/splitk"); return STJ_EINVAL; }
  if (accumulate && !c_f32) { stj_set_error("stj_gemm: accumulate requires f32 output"); return STJ_EINVAL; }
  if (splitk > 1 && !accumulate) { stj_set_error("stj_gemm: splitk>1 requires accumulate"); return STJ_EINVAL; }
  if (splitk > 1 && (act != ACT_NONE || res)) { stj_set_error("stj_gemm: splitk with nonlinear epilogue"); return STJ_EINVAL; }
  if ((long long)nb1 * nb2 * splitk > 65535) { stj_set_error("stj_gemm: batch*splitk too large"); return STJ_EINVAL; }
  GemmArgs p;
  p.A = A; p.B = B; p.C = C; p.bias = bias; p.res = res;
  p.M = M; p.N = N; p.K = K; p.nb1 = nb1; p.nb2 = nb2;
  p.sAb1 = sAb1; p.sAb2 = sAb2; p.sAm = sAm; p.sAk = sAk;
  p.sBb1 = sBb1; p.sBb2 = sBb2; p.sBk = sBk; p.sBn = sBn;
  p.sCb1 = sCb1; p.sCb2 = sCb2; p.ldc = ldc;
  p.sBias1 = sBias1; p.sBias2 = sBias2; p.sRes1 = sRes1; p.sRes2 = sRes2; p.ldres = ldres;
  p.act = act; p.c_f32 = c_f32; p.accumulate = accumulate; p.splitk = splitk; p.alpha = alpha;
  const long long es = dtype == STJ_BF16 ? 2 : 4;
  auto al = [&](const void* ptr, long long s0, long long s1, long long s2) {
    return ((uintptr_t)ptr % 16 == 0) && ((s0 * es) % 16 == 0) && ((s1 * es) % 16 == 0) && ((s2 * es) % 16 == 0);
  };
  p.vecA = (sAk == 1) ? al(A, sAm, sAb1, sAb2) : (sAm == 1 ? al(A, sAk, sAb1, sAb2) : 0);
  p.vecB = (sBk == 1) ? al(B, sBn, sBb1, sBb2) : (sBn == 1 ? al(B, sBk, sBb1, sBb2) : 0);
  if (dtype == STJ_BF16) return launch_gemm<bf16>(p, stream);
  if (dtype == STJ_F32) return launch_gemm<float>(p, stream);
  stj_set_error("stj_gemm: bad dtype %d", dtype);
  return STJ_EINVAL;
}

// ---- column sums: out[n] += sum_m X[m, n]  (bias gradients; out is f32, accumulated atomically) ----
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* X, float* out, int M, int N, long long ld, int rows_per_block) {
  __shared__ float part[4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  for (int c0 = blockIdx.y * 64; c0 < N; c0 += gridDim.y * 64) {
    const int c = c0 + cx;
    float s = 0.f;
    if (c < N)
      for (int r = r0 + ry; r < r1; r += 4) s += ldf(X + (long long)r * ld + c);
    part[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && c < N) atomicAdd(out + c, part[0][cx] + part[1][cx] + part[2][cx] + part[3][cx]);
    __syncthreads();
  }
}

extern "C" int stj_colsum(const void* X, float* out, int M, int N, long long ld, int dtype, hipStream_t stream) {
  if (M <= 0 || N <= 0) return STJ_OK;
  const int rpb = 256;
  dim3 grid((M + rpb - 1) / rpb, min(8, (N + 63) / 64));
  if (dtype == STJ_BF16) hipLaunchKernelGGL(colsum_kernel<bf16>, grid, dim3(256), 0, stream, (const bf16*)X, out, M, N, ld, rpb);
  else hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, stream, (const float*)X, out, M, N, ld, rpb);
  return stj_check_launch("stj_colsum");
}
