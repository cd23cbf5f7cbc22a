// Weight-stationary, persistent forward kernel for the folded nearest-2x-upsample 3x3 convolution (bf16).
// (reference op: UpSampling3D(1,2,2) -> Conv2D 3x3 SAME + bias -> ELU, modules.py:746-748,732-735; algebra in conv.hip)
//
// Why: the first kernel (conv.hip) re-staged all 16 effective-tap weight matrices through LDS for every 8x16 pixel
// tile and ran at ~8 % MFMA utilisation (rocprof PMC: 57 % of wave cycles waiting, half of all LDS cycles bank
// conflicts; profiles/r01_b_*).  Here
//   * one wavefront owns one output phase (a,b); its 4 tap matrices Weff[a,b,r,s] (CoutTile x Cin) sit in VGPRs for the
//     whole kernel as MFMA *A* fragments (v_mfma_f32_16x16x32_bf16: D[cout][pixel] += W[cout][k] * X[k][pixel]);
//   * workgroups are persistent (one per CU) and walk over pixel tiles; the only LDS traffic of the main loop is the
//     B fragment (16 pixels x 32 channels = one ds_read_b128 per lane) which feeds NF MFMAs;
//   * the input halo (10 x 18 low-res pixels x Cin) is double buffered: the next tile's global loads are issued before
//     the MFMA loop and written to the other LDS buffer after it;
//   * results (bias + ELU applied) are staged in LDS and leave as full rows of 8-byte segments, coalesced.
#include "common.h"
#include <stdlib.h>

#define WS_TH 8
#define WS_TW 16
#define WS_HW (WS_TW + 2)
#define WS_HH (WS_TH + 2)

template <int KS, int NF, int NW>
__global__ __launch_bounds__(256 * NW, NW) void upconv_fwd_ws_kernel(const bf16* __restrict__ X, const bf16* __restrict__ Wf,
                                                               const float* __restrict__ bias, bf16* __restrict__ Y,
                                                               int F, int Hi, int Wi, int Cout, int act, int ntiles) {
  constexpr int CIN = KS * 32;
  constexpr int LDK = CIN + 8;                     // halo pixel stride (elements)
  constexpr int CT = NF * 16;                      // cout tile of this workgroup
  constexpr int LDO = CT + 4;                      // output-stage pixel stride (elements), 8-byte aligned
  constexpr int HPIX = WS_HH * WS_HW;              // 180 halo pixels
  constexpr int CPP = CIN / 8;                     // 16-byte chunks per halo pixel
  constexpr int NT = 256 * NW;                     // threads: NW waves per output phase, each owning WS_TH/NW tile rows
  constexpr int NCH = (HPIX * CPP + NT - 1) / NT;  // chunks per thread
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16* halo0 = reinterpret_cast<bf16*>(smem_raw);
  bf16* halo1 = halo0 + HPIX * LDK;
  bf16* ostage = halo1 + HPIX * LDK;               // [16][32][LDO]

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int a = (w & 3) >> 1, b = w & 1;           // this wave's output phase
  const int row0 = (w >> 2) * (WS_TH / NW);        // first tile row of this wave
  const int g = lane >> 4, ln = lane & 15;
  const int n0 = blockIdx.y * CT;
  const int tiles_x = (Wi + WS_TW - 1) / WS_TW, tiles_y = (Hi + WS_TH - 1) / WS_TH;
  const int Ho = 2 * Hi, Wo = 2 * Wi;

  // ---- stationary weights: wf[tap][n][ks] = Weff[a,b,r,s][n0 + n*16 + ln][ks*32 + g*8 .. +8] ----
  s16x8 wf[4][NF][KS];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int pt = a * 8 + b * 4 + t;
#pragma unroll
    for (int n = 0; n < NF; ++n) {
      const int co = n0 + n * 16 + ln;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (co < Cout) wf[t][n][ks] = *reinterpret_cast<const s16x8*>(Wf + ((long long)pt * Cout + co) * CIN + ks * 32 + g * 8);
        else wf[t][n][ks] = (s16x8){0, 0, 0, 0, 0, 0, 0, 0};
      }
    }
  }
  float bv[NF][4];
#pragma unroll
  for (int n = 0; n < NF; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = n0 + n * 16 + g * 4 + r;
      bv[n][r] = co < Cout ? bias[co] : 0.f;
    }

  auto tile_coords = [&](int tile, int& f, int& ty0, int& tx0) {
    const int tx = tile % tiles_x; const int t2 = tile / tiles_x;
    ty0 = (t2 % tiles_y) * WS_TH; f = t2 / tiles_y; tx0 = tx * WS_TW;
  };
  uint4 pre[NCH];
  auto prefetch = [&](int tile) {                  // global -> registers (asynchronous until first use)
    int f, ty0, tx0;
    tile_coords(tile, f, ty0, tx0);
    const bf16* Xf = X + (long long)f * Hi * Wi * CIN;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int q = tid + i * NT;
      const int px = q / CPP, ch = (q % CPP) * 8;
      const int gy = ty0 + px / WS_HW - 1, gx = tx0 + px % WS_HW - 1;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (q < HPIX * CPP && gy >= 0 && gy < Hi && gx >= 0 && gx < Wi)
        v = *reinterpret_cast<const uint4*>(Xf + ((long long)gy * Wi + gx) * CIN + ch);
      pre[i] = v;
    }
  };
  auto commit = [&](bf16* halo) {                  // registers -> LDS
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int q = tid + i * NT;
      if (q < HPIX * CPP) *reinterpret_cast<uint4*>(halo + (q / CPP) * LDK + (q % CPP) * 8) = pre[i];
    }
  };

  int tile = blockIdx.x;
  if (tile < ntiles) { prefetch(tile); commit(halo0); }
  __syncthreads();
  int buf = 0;
  for (; tile < ntiles; tile += gridDim.x) {
    const bf16* halo = buf ? halo1 : halo0;
    const int next = tile + gridDim.x;
    if (next < ntiles) prefetch(next);
    int f, ty0, tx0;
    tile_coords(tile, f, ty0, tx0);

    // ---- MFMA main loop: two tile rows at a time ----
#pragma unroll NW == 1 ? 4 : 1
    for (int mf = row0; mf < row0 + WS_TH / NW; mf += 2) {
      f32x4 acc[2][NF];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NF; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            s16x8 xb[2];
#pragma unroll
            for (int m = 0; m < 2; ++m)
              xb[m] = *reinterpret_cast<const s16x8*>(halo + ((mf + m + a + r) * WS_HW + ln + b + s) * LDK + ks * 32 + g * 8);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
              for (int n = 0; n < NF; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[r * 2 + s][n][ks]),
                                                                    __builtin_bit_cast(bf16x8_t, xb[m]), acc[m][n], 0, 0, 0);
          }
      // epilogue into the LDS stage: lane holds couts g*4..g*4+3 of pixel (2(mf+m)+a, 2 ln + b)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NF; ++n) {
          const uint32_t p0 = pack2bf(apply_act_fast(acc[m][n][0] + bv[n][0], act), apply_act_fast(acc[m][n][1] + bv[n][1], act));
          const uint32_t p1 = pack2bf(apply_act_fast(acc[m][n][2] + bv[n][2], act), apply_act_fast(acc[m][n][3] + bv[n][3], act));
          *reinterpret_cast<uint2*>(ostage + ((2 * (mf + m) + a) * (2 * WS_TW) + 2 * ln + b) * LDO + n * 16 + g * 4) = make_uint2(p0, p1);
        }
    }
    if (next < ntiles) commit(buf ? halo0 : halo1);
    __syncthreads();
    // ---- coalesced store of the 16 x 32 x CT output tile (8-byte segments) ----
    {
      constexpr int SEG = CT / 4;                  // 8-byte segments per pixel
      bf16* Yf = Y + (long long)f * Ho * Wo * Cout;
      for (int q = tid; q < 2 * WS_TH * 2 * WS_TW * SEG; q += NT) {
        const int sg = q % SEG, p = q / SEG;
        const int hr = p / (2 * WS_TW), hc = p % (2 * WS_TW);
        const int oy = 2 * ty0 + hr, ox = 2 * tx0 + hc, co = n0 + sg * 4;
        if (oy < Ho && ox < Wo && co < Cout)
          *reinterpret_cast<uint2*>(Yf + ((long long)oy * Wo + ox) * Cout + co) = *reinterpret_cast<const uint2*>(ostage + p * LDO + sg * 4);
      }
    }
    __syncthreads();
    buf ^= 1;
  }
}

template <int KS, int NF, int NW>
static bool ws_launch(const void* X, const void* Wf, const float* bias, void* Y, int F, int Hi, int Wi, int Cout, int act, hipStream_t st) {
  constexpr int CIN = KS * 32, LDK = CIN + 8, CT = NF * 16, LDO = CT + 4;
  const size_t lds = (size_t)(2 * WS_HH * WS_HW * LDK + 2 * WS_TH * 2 * WS_TW * LDO) * 2;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)upconv_fwd_ws_kernel<KS, NF, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
    attr_set = true;
  }
  const int ntiles = ((Wi + WS_TW - 1) / WS_TW) * ((Hi + WS_TH - 1) / WS_TH) * F;
  const int ct = (Cout + CT - 1) / CT;
  int nblk = 256 / ct;
  if (nblk > ntiles) nblk = ntiles;
  if (nblk < 1) nblk = 1;
  hipLaunchKernelGGL((upconv_fwd_ws_kernel<KS, NF, NW>), dim3(nblk, ct), dim3(256 * NW), lds, st, (const bf16*)X, (const bf16*)Wf, bias, (bf16*)Y,
                     F, Hi, Wi, Cout, act, ntiles);
  return true;
}

// returns true when the weight-stationary kernel handles this shape (bf16, Cin in {96,128}, Cout multiple of 4)
bool upconv_fwd_ws_try(const void* X, const void* Wf, const float* bias, void* Y, int F, int Hi, int Wi, int Cin, int Cout, int act,
                       hipStream_t st) {
  if (Cout % 4) return false;
  static int variant = -1;
  if (variant < 0) { const char* e = getenv("STJ_WS_VARIANT"); variant = e ? atoi(e) : 1; }
  // Cin=96: the 2-waves-per-SIMD variant spills (144 weight VGPRs + prefetch); one wave per SIMD measured faster (386 vs 432 us)
  if (Cin == 96) return variant == 2 ? ws_launch<3, 3, 2>(X, Wf, bias, Y, F, Hi, Wi, Cout, act, st) : ws_launch<3, 3, 1>(X, Wf, bias, Y, F, Hi, Wi, Cout, act, st);
  if (Cin == 128) return variant == 0 ? ws_launch<4, 3, 1>(X, Wf, bias, Y, F, Hi, Wi, Cout, act, st) : ws_launch<4, 2, 2>(X, Wf, bias, Y, F, Hi, Wi, Cout, act, st);
  return false;
}
