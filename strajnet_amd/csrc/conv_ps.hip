// Wide-channel layers of the decoder (Cin = 192, 384: upconv_2_0 / upconv_3_0) -- "phase-per-wave, streamed weights".
// (reference op: UpSampling3D(1,2,2) -> Conv2D 3x3 SAME + bias -> ELU, modules.py:746-748,732-735; tap algebra in conv.hip)
//
// The weight-stationary kernels of conv_ws.hip keep all 4 tap matrices of a phase in VGPRs, which stops at Cin = 128.  The generic
// kernel of conv.hip that served the two wide layers re-staged 8 (phase,tap) weight slots through LDS three times per 32-channel
// chunk (6 barriers per chunk, 0.64 LDS fragment reads per MFMA, 16-byte row padding with bank conflicts) and ran at 11 % of the
// MFMA peak.  Here, per 32-channel chunk:
//   * wave w owns output phase (a,b) = (w>>1, w&1) of an 8x16 low-res tile and ALL 8 tile rows: 8 x FN accumulator fragments;
//   * its 4 tap matrices of the chunk (4 x FN fragments) come straight from global / L2 into VGPRs, prefetched one chunk ahead --
//     weights never touch LDS (the 4 waves need disjoint matrices, so an LDS copy would be shared by nobody);
//   * the input halo chunk (10 x 18 pixels x 32 channels) is double buffered in LDS, prefetched through registers one chunk ahead;
//     a halo row fragment is read once and feeds the r = 0 tap of tile row j and the r = 1 tap of row j - 1:
//     18 ds_read_b128 per 32 FN MFMAs (0.14 reads per MFMA at FN = 4), one barrier per chunk;
//   * bias + ELU epilogue through an LDS stage that aliases the halo buffers, 16-byte coalesced stores.
#include "common.h"
#include <stdlib.h>

#define PS_TW 16
#define PS_HW (PS_TW + 2)
#define PS_KC 32
#define PS_RES_BATCH 4      // skip-sum items whose operands are in flight together (item by item: 5691 / 5889 / 5678 against 5729 / 5912 / 5711 scenes/s, B = 32 inference; 8: 5788 / 5731 / 5716; profiles/r06_zi_ps_res_batch.txt)

// TH = rows of the low-res tile.  16 (round 6): the workgroup's weight stream -- every workgroup reads all 16 tap matrices of its cout block,
// 393 KB at Cin = 384, and with 8-row tiles that was 302 of the 400 MB the 384 -> 192 launch pulls out of L2, the level this kernel saturates
// (~6 TB/s) -- is amortised over twice the pixels; the epilogue goes through the LDS stage in two halves so that two workgroups still share a CU.
template <typename T, int FN, int MINB, int TH>
__global__ __launch_bounds__(256, MINB) void upconv_fwd_ps_kernel(const T* __restrict__ X, const T* __restrict__ Wf,
                                                                  const float* __restrict__ bias, T* __restrict__ Y,
                                                                  const T* __restrict__ R1, T* __restrict__ Y2, const T* __restrict__ R2,
                                                                  int F, int Hi, int Wi, int Cin, int Cout) {
  constexpr int LDK = PS_KC + 16;                  // 96-byte pixel stride: 2 (mod 4) 16-byte slots, conflict-free b128 fragment reads
  constexpr int CT = FN * 16, LDO = CT + 8;
  constexpr int PS_TH = TH, PS_HH = TH + 2;
  constexpr int HPIX = PS_HH * PS_HW;              // 180 / 324
  constexpr int NCH = (HPIX * 4 + 255) / 256;      // 16-byte chunks per thread per halo chunk (3)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* halo0 = reinterpret_cast<T*>(smem_raw);
  T* halo1 = halo0 + HPIX * LDK;
  T* ostage = reinterpret_cast<T*>(smem_raw);      // aliases the halo buffers after the last chunk

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int a = w >> 1, b = w & 1;
  const int g = lane >> 4, ln = lane & 15;
  const int tiles_x = (Wi + PS_TW - 1) / PS_TW, tiles_y = (Hi + PS_TH - 1) / PS_TH;
  int bid = blockIdx.x;
  const int tx0 = (bid % tiles_x) * PS_TW; bid /= tiles_x;
  const int ty0 = (bid % tiles_y) * PS_TH; const int f = bid / tiles_y;
  const int n0 = blockIdx.y * CT;
  const int nch = Cin / PS_KC;
  const T* Xf = X + (long long)f * Hi * Wi * Cin;

  f32x4 acc[PS_TH][FN];
#pragma unroll
  for (int j = 0; j < PS_TH; ++j)
#pragma unroll
    for (int n = 0; n < FN; ++n) acc[j][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // per-thread halo chunk geometry (does not depend on the channel chunk)
  int hoff[NCH]; bool hin[NCH]; int hlds[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int q = tid + i * 256;
    const int px = q >> 2, ch = (q & 3) * 8;
    const int gy = ty0 + px / PS_HW - 1, gx = tx0 + px % PS_HW - 1;
    hin[i] = q < HPIX * 4 && gy >= 0 && gy < Hi && gx >= 0 && gx < Wi;
    hoff[i] = hin[i] ? (gy * Wi + gx) * Cin + ch : 0;      // out-of-image pixels load pixel 0 and are zeroed on the way to LDS
    hlds[i] = q < HPIX * 4 ? px * LDK + ch : -1;
  }
  long long woff[FN];
#pragma unroll
  for (int n = 0; n < FN; ++n) woff[n] = ((long long)(a * 8 + b * 4) * Cout + n0 + n * 16 + ln) * Cin + g * 8;    // Cout % CT == 0 (dispatch)
  const long long wtap = (long long)Cout * Cin;

  uint4 pre[NCH];
  auto ld_halo = [&](int c0) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      pre[i] = *reinterpret_cast<const uint4*>(Xf + hoff[i] + c0);      // unconditional (no branch, no vmcnt(0) per load)
    }
  };
  auto st_halo = [&](T* h) {
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      if (hlds[i] >= 0) *reinterpret_cast<uint4*>(h + hlds[i]) = hin[i] ? pre[i] : make_uint4(0, 0, 0, 0);
  };
  // the chunk's tap matrices of tap column s: [0] = tap row 0, [1] = tap row 1 -- half a chunk's weights at a time (with 16-row tiles the
  // accumulators take 128 registers: a whole chunk in flight next to a whole chunk in use spilled)
  auto ld_w = [&](int c0, int sh, s16x8 (&wv)[2][FN]) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int n = 0; n < FN; ++n) wv[r][n] = *reinterpret_cast<const s16x8*>(Wf + woff[n] + (2 * r + sh) * wtap + c0);
  };
  auto compute = [&](const T* h, int sh, const s16x8 (&wv)[2][FN]) {
#pragma unroll
    for (int hr = 0; hr <= PS_TH; ++hr) {
      const s16x8 xb = *reinterpret_cast<const s16x8*>(h + ((hr + a) * PS_HW + ln + b + sh) * LDK + g * 8);
      if (hr < PS_TH) {
#pragma unroll
        for (int n = 0; n < FN; ++n) acc[hr][n] = Mma<T>::mma(wv[0][n], xb, acc[hr][n]);                 // r = 0: D[cout][pixel]
      }
      if (hr >= 1) {
#pragma unroll
        for (int n = 0; n < FN; ++n) acc[hr - 1][n] = Mma<T>::mma(wv[1][n], xb, acc[hr - 1][n]);         // r = 1
      }
    }
  };

  s16x8 wA[2][FN], wB[2][FN];
  ld_w(0, 0, wA);
  ld_halo(0);
  st_halo(halo0);
  __syncthreads();
#pragma unroll 1
  for (int c = 0; c < nch; ++c) {
    const T* hc = halo0 + (c & 1) * (HPIX * LDK);
    T* hn = halo0 + ((c + 1) & 1) * (HPIX * LDK);
    const int cn = min(c + 1, nch - 1) * PS_KC;    // (the last prefetch re-reads the last chunk: unconditional loads keep the register arrays out of scratch)
    ld_w(c * PS_KC, 1, wB);
    ld_halo(cn);
    compute(hc, 0, wA);
    ld_w(cn, 0, wA);
    compute(hc, 1, wB);
    st_halo(hn);
    __syncthreads();
  }
  __syncthreads();                                 // (halo stores of the dangling prefetch are done: the stage may overwrite them)

  // ---- bias + ELU into the stage (8 tile rows at a time): lane holds couts n*16 + 4g .. +3 of output pixel (2j + a, 2 ln + b) ----
  float bv[FN][4];
#pragma unroll
  for (int n = 0; n < FN; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = n0 + n * 16 + g * 4 + r;
      bv[n][r] = co < Cout ? bias[co] : 0.f;
    }
#pragma unroll
  for (int hj = 0; hj < PS_TH; hj += 8) {
    if (hj) __syncthreads();                         // the previous half has left the stage
#pragma unroll
    for (int n = 0; n < FN; ++n)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t p0 = pack2<T>(elu_bf(acc[hj + j][n][0] + bv[n][0]), elu_bf(acc[hj + j][n][1] + bv[n][1]));
        const uint32_t p1 = pack2<T>(elu_bf(acc[hj + j][n][2] + bv[n][2]), elu_bf(acc[hj + j][n][3] + bv[n][3]));
        *reinterpret_cast<uint2*>(ostage + ((2 * j + a) * (2 * PS_TW) + 2 * ln + b) * LDO + n * 16 + g * 4) = make_uint2(p0, p1);
      }
    __syncthreads();
    constexpr int SEG = CT / 8;
    const int Ho = 2 * Hi, Wo = 2 * Wi;
    T* Yf = Y + (long long)f * Ho * Wo * Cout;
    constexpr int NIT = 2 * 8 * 2 * PS_TW * SEG / 256;          // 16-byte items per thread and half (8 at FN = 2)
    if (R1 != nullptr) {
      // decoder skips (reference modules.py:750-765): Y = ELU(conv) + R1 and, optionally, Y2 = Y + R2 -- each sum rounded to the storage type
      // exactly as a separate elementwise add of the stored tensors would round it.  The skip operands of PS_RES_BATCH items at a time are
      // fetched before any of them is used: item by item every item was two dependent memory round trips (R1 -> store Y -> R2 -> store Y2) on
      // a workgroup that has nothing else left to do
      static_assert(NIT % PS_RES_BATCH == 0, "whole batches");
#pragma unroll 1
      for (int q0 = tid; q0 < 256 * NIT; q0 += 256 * PS_RES_BATCH) {
        uint4 r1v[PS_RES_BATCH], r2v[PS_RES_BATCH];
        long long fo[PS_RES_BATCH];
        bool ok[PS_RES_BATCH];
#pragma unroll
        for (int k = 0; k < PS_RES_BATCH; ++k) {
          const int q = q0 + 256 * k, sg = q % SEG, p = q / SEG;
          const int oy = 2 * (ty0 + hj) + p / (2 * PS_TW), ox = 2 * tx0 + p % (2 * PS_TW), co = n0 + sg * 8;
          ok[k] = oy < Ho && ox < Wo && co < Cout;
          fo[k] = ok[k] ? (long long)f * Ho * Wo * Cout + ((long long)oy * Wo + ox) * Cout + co : 0;
          r1v[k] = *reinterpret_cast<const uint4*>(R1 + fo[k]);
          r2v[k] = R2 != nullptr ? *reinterpret_cast<const uint4*>(R2 + fo[k]) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < PS_RES_BATCH; ++k) {
          const int q = q0 + 256 * k, sg = q % SEG, p = q / SEG;
          if (!ok[k]) continue;
          float u[8], r[8];
          ld16<T>(ostage + p * LDO + sg * 8, u);
          ld16<T>(reinterpret_cast<const T*>(&r1v[k]), r);
#pragma unroll
          for (int e = 0; e < 8; ++e) u[e] += r[e];
          __attribute__((aligned(16))) T rounded[8];
          st16<T>(rounded, u);
          *reinterpret_cast<uint4*>(Y + fo[k]) = *reinterpret_cast<const uint4*>(rounded);
          if (Y2 != nullptr) {
            ld16<T>(rounded, u);                   // the rounded sum, as a separate add would read it back
            ld16<T>(reinterpret_cast<const T*>(&r2v[k]), r);
#pragma unroll
            for (int e = 0; e < 8; ++e) u[e] += r[e];
            st16<T>(Y2 + fo[k], u);
          }
        }
      }
    } else {
      for (int q = tid; q < 2 * 8 * 2 * PS_TW * SEG; q += 256) {
        const int sg = q % SEG, p = q / SEG;
        const int oy = 2 * (ty0 + hj) + p / (2 * PS_TW), ox = 2 * tx0 + p % (2 * PS_TW), co = n0 + sg * 8;
        if (oy < Ho && ox < Wo && co < Cout)
          *reinterpret_cast<uint4*>(Yf + ((long long)oy * Wo + ox) * Cout + co) = *reinterpret_cast<const uint4*>(ostage + p * LDO + sg * 8);
      }
    }
  }
}

template <typename T, int FN, int MINB, int TH>
static bool fwd_ps_launch(const void* X, const void* Wf, const float* bias, void* Y, const void* R1, void* Y2, const void* R2, int F, int Hi, int Wi, int Cin, int Cout, hipStream_t st) {
  constexpr int LDK = PS_KC + 16, CT = FN * 16, LDO = CT + 8;
  const size_t lds_h = (size_t)2 * (TH + 2) * PS_HW * LDK * 2, lds_o = (size_t)2 * 8 * 2 * PS_TW * LDO * 2;
  const size_t lds = lds_h > lds_o ? lds_h : lds_o;
  static PerDevice<bool> attr_set;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)upconv_fwd_ps_kernel<T, FN, MINB, TH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
    attr_set = true;
  }
  const int tiles = ((Wi + PS_TW - 1) / PS_TW) * ((Hi + TH - 1) / TH) * F;
  hipLaunchKernelGGL((upconv_fwd_ps_kernel<T, FN, MINB, TH>), dim3(tiles, (Cout + CT - 1) / CT), dim3(256), lds, st, (const T*)X, (const T*)Wf, bias,
                     (T*)Y, (const T*)R1, (T*)Y2, (const T*)R2, F, Hi, Wi, Cin, Cout);
  return true;
}

template <typename T>
static bool fwd_ps_try_t(const void* X, const void* Wf, const float* bias, void* Y, const void* R1, void* Y2, const void* R2, int F, int Hi, int Wi, int Cin, int Cout, hipStream_t st) {
  // measured (tools/bench_conv.py, us for 384->192 @16^2 / 192->128 @32^2, F = 64; generic kernel 113 / 140): FN = 2, two workgroups
  // per CU: 64 / 93;  FN = 3: 84 / 137;  FN = 4 (one wave per SIMD): 105 / 128.  At FN = 2 the kernel moves ~6 TB/s from L2 (each
  // workgroup streams 32 KB of weights + 11.5 KB of halo per 4.2 MFLOP chunk), which is where the 64x64 GEMM tiles saturate too.
  if (Cout % 32) return false;
  // round 6 (half-chunk weight staging: 152 registers at 8-row tiles, 253 at 16): 8-row tiles 54.8 / 83.9 us, 16-row tiles 55.0 / 69.5 us;
  // end to end (alternating same-box runs, 16 vs 8 rows) inference 5533 / 5535 / 5414 vs 5515 / 5387 / 5349, training 1377 / 1373 / 1372 vs 1368 / 1374 / 1375
  if (Hi % 16 == 0) return fwd_ps_launch<T, 2, 2, 16>(X, Wf, bias, Y, R1, Y2, R2, F, Hi, Wi, Cin, Cout, st);
  return fwd_ps_launch<T, 2, 2, 8>(X, Wf, bias, Y, R1, Y2, R2, F, Hi, Wi, Cin, Cout, st);
}
// true when this kernel took the problem: 16-bit, ELU, Cin a multiple of 32 above the weight-stationary range
bool upconv_fwd_ps_try(const void* X, const void* Wf, const float* bias, void* Y, int F, int Hi, int Wi, int Cin, int Cout, int act,
                       int dtype, hipStream_t st) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("STJ_NO_WS"); on = !(e && atoi(e)); }      // STJ_NO_WS=1: generic conv kernels everywhere
  if (!on || act != ACT_ELU || Cin % PS_KC || Cin <= 128 || Cout % 32) return false;
  return dtype == STJ_F16 ? fwd_ps_try_t<f16>(X, Wf, bias, Y, nullptr, nullptr, nullptr, F, Hi, Wi, Cin, Cout, st)
                          : fwd_ps_try_t<bf16>(X, Wf, bias, Y, nullptr, nullptr, nullptr, F, Hi, Wi, Cin, Cout, st);
}
// the same with the skip sums in the epilogue: Y = ELU(conv) + R1, Y2 = Y + R2 (Y2 / R2 may be null)
bool upconv_fwd_ps_res_try(const void* X, const void* Wf, const float* bias, void* Y, const void* R1, void* Y2, const void* R2, int F, int Hi,
                           int Wi, int Cin, int Cout, int dtype, hipStream_t st) {
  if (Cin % PS_KC || Cin <= 128 || Cout % 32 || R1 == nullptr || ((Y2 == nullptr) != (R2 == nullptr))) return false;
  return dtype == STJ_F16 ? fwd_ps_try_t<f16>(X, Wf, bias, Y, R1, Y2, R2, F, Hi, Wi, Cin, Cout, st)
                          : fwd_ps_try_t<bf16>(X, Wf, bias, Y, R1, Y2, R2, F, Hi, Wi, Cin, Cout, st);
}
