// Decoder convolutions of the FPN / pyramid decoder (reference modules.py:630-772, Pyramid3DDecoder):
//   UpSampling3D(1,2,2) nearest -> Conv2D 3x3 SAME + bias -> ELU   (modules.py:746-748, 732-735)
// as an implicit GEMM on MFMA with the 2x nearest upsample FOLDED into the weights: output pixel
// (2i+a, 2j+b) only ever sees the 2x2 low-res neighbourhood X[i+a-1+r, j+b-1+s], r,s in {0,1}, with
// summed taps  Weff[a,b,r,s] = sum_{dy in R(a,r)} sum_{dx in R(b,s)} W[dy,dx]
//   R(0,0)={-1}  R(0,1)={0,+1}  R(1,0)={-1,0}  R(1,1)={+1}
// (exact algebra, 2.25x fewer MACs than the materialised upsample; SURVEY.md App. C KAT vi).
// Zero padding at the upsampled border == zero padding of the low-res map, so halo pixels outside the
// low-res map are staged as zeros.
//   forward : Y[f,2i+a,2j+b,:] = ELU( sum_{r,s} X[f,i+a-1+r,j+b-1+s,:] . Weff[a,b,r,s] + bias )
//   dgrad   : dX[f,i,j,:]      = sum_{u,v in -1..2} dP[f,2i+u,2j+v,:] . Weff[a,b,r,s]^T,  a=u&1, r=(2-a-u)/2 (same for v)
//   wgrad   : dWeff[a,b,r,s]   = sum_{f,i,j} X[f,i+a-1+r,j+b-1+s,:]^T dP[f,2i+a,2j+b,:]  (split over pixel strips, f32 atomics)
// Also here: the 48->2 output convolutions writing straight into the [B,H,W,32] result layout
// (modules.py:767-770 + transpose :838), patch im2col for the 4x4/stride-4 patch embeds (modules.py:430-431),
// and 3x3 im2col / col2im for the small grouped conv of FG-MSA (FG_MSA.py:51).
#include "common.h"
#include <stdlib.h>

#define TILE_H 8
#define TILE_W 16
#define KC 32

// ---------------------------------------------------------------------------------------------------
// weight preparation / gradient fold-back
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void tap_range(int a, int r, int& lo, int& hi) {   // indices dy+1 in [lo,hi]
  if (a == 0) { lo = r == 0 ? 0 : 1; hi = r == 0 ? 0 : 2; }
  else { lo = r == 0 ? 0 : 2; hi = r == 0 ? 1 : 2; }
}
__device__ __forceinline__ int tap_of(int a, int dyi) {  // which r of phase a uses tap index dyi (0..2)
  return a == 0 ? (dyi == 0 ? 0 : 1) : (dyi == 2 ? 1 : 0);
}

// W [3][3][Cin][Cout] f32 (Keras HWIO) -> Wf [16][Cout][Cin] (forward B operand, K=cin contiguous)
//                                         Wd [16][Cin][Cout] (dgrad  B operand, K=cout contiguous, indexed (u+1)*4+(v+1))
// LDS row padding of the generic (v1) conv kernels: 16 bytes.  (The 32-byte LdsPad that makes b128 fragment reads conflict-free
// costs these kernels a resident block per CU: measured 0.31 -> 0.47 ms on the two small decoder layers.)
template <typename T> struct ConvPad { static constexpr int P = 16 / sizeof(T); };

template <typename T>
__global__ __launch_bounds__(256) void upconv_prep_kernel(const float* W, T* Wf, T* Wd, int Cin, int Cout) {
  const int total = 16 * Cout * Cin;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const int ci = idx % Cin, co = (idx / Cin) % Cout, pt = idx / (Cin * Cout);
    const int a = pt >> 3, b = (pt >> 2) & 1, r = (pt >> 1) & 1, s = pt & 1;
    int ylo, yhi, xlo, xhi;
    tap_range(a, r, ylo, yhi);
    tap_range(b, s, xlo, xhi);
    float acc = 0.f;
    for (int y = ylo; y <= yhi; ++y)
      for (int x = xlo; x <= xhi; ++x) acc += W[((y * 3 + x) * Cin + ci) * Cout + co];
    stf(Wf + ((long long)pt * Cout + co) * Cin + ci, acc);
    const int u = 2 - a - 2 * r, v = 2 - b - 2 * s;
    stf(Wd + ((long long)((u + 1) * 4 + (v + 1)) * Cin + ci) * Cout + co, acc);
  }
}
// dW[dy][dx][ci][co] += sum_{a,b} dWeff[a,b,r(a,dy),s(b,dx)][co][ci]
__global__ __launch_bounds__(256) void upconv_fold_kernel(const float* dWeff, float* dW, int Cin, int Cout) {
  const int total = 9 * Cin * Cout;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const int co = idx % Cout, ci = (idx / Cout) % Cin, t = idx / (Cin * Cout);
    const int dyi = t / 3, dxi = t % 3;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int pt = a * 8 + b * 4 + tap_of(a, dyi) * 2 + tap_of(b, dxi);
        acc += dWeff[((long long)pt * Cout + co) * Cin + ci];
      }
    dW[idx] += acc;
  }
}

// ---------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------
template <typename T, int FN>
__global__ __launch_bounds__(256, 2) void upconv_fwd_kernel(const T* X, const T* Wf, const float* bias, T* Y,
                                                         int F, int Hi, int Wi, int Cin, int Cout, int act) {
  constexpr int BN = FN * 16;
  constexpr int LDK = KC + ConvPad<T>::P;
  constexpr int VN = Vec<T>::N, CPR = KC / VN;
  constexpr int HH = TILE_H + 2, HW = TILE_W + 2;
  __shared__ __attribute__((aligned(16))) T halo[HH * HW * LDK];
  __shared__ __attribute__((aligned(16))) T Bs[8 * BN * LDK];   // (phase,tap) slots of one dy group

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int tiles_x = (Wi + TILE_W - 1) / TILE_W, tiles_y = (Hi + TILE_H - 1) / TILE_H;
  int bid = blockIdx.x;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y; const int f = bid / tiles_y;
  const int ty0 = ty * TILE_H, tx0 = tx * TILE_W, n0 = blockIdx.y * BN;
  const int mf0 = 2 * w;

  f32x4 acc[4][2][FN];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < FN; ++n) acc[p][m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const T* Xf = X + (long long)f * Hi * Wi * Cin;
  const uint4 z4 = make_uint4(0, 0, 0, 0);
  for (int c0 = 0; c0 < Cin; c0 += KC) {
    for (int i = tid; i < HH * HW * CPR; i += 256) {
      const int px = i / CPR, ch = (i % CPR) * VN;
      const int gy = ty0 + px / HW - 1, gx = tx0 + px % HW - 1;
      uint4 v = z4;
      if (gy >= 0 && gy < Hi && gx >= 0 && gx < Wi && c0 + ch < Cin)
        v = *reinterpret_cast<const uint4*>(Xf + ((long long)gy * Wi + gx) * Cin + c0 + ch);
      *reinterpret_cast<uint4*>(halo + px * LDK + ch) = v;
    }
#pragma unroll
    for (int dyi = 0; dyi < 3; ++dyi) {
      // (a,r) pairs with a + r == dyi: dyi=0 -> (0,0); dyi=1 -> (0,1),(1,0); dyi=2 -> (1,1).  slot = al*4 + b*2 + s
      const int npair = dyi == 1 ? 2 : 1;
      __syncthreads();               // halo staged (dyi==0) / previous group's B reads finished
      for (int i = tid; i < npair * 4 * BN * CPR; i += 256) {
        const int ch = (i % CPR) * VN, n = (i / CPR) % BN, slot = i / (CPR * BN);
        const int al = slot >> 2, b = (slot >> 1) & 1, s2 = slot & 1;
        const int a = dyi == 0 ? 0 : (dyi == 2 ? 1 : al);
        const int r = dyi - a;
        const int pt = a * 8 + b * 4 + r * 2 + s2;
        uint4 v = z4;
        if (n0 + n < Cout && c0 + ch < Cin)
          v = *reinterpret_cast<const uint4*>(Wf + ((long long)pt * Cout + n0 + n) * Cin + c0 + ch);
        *reinterpret_cast<uint4*>(Bs + (slot * BN + n) * LDK + ch) = v;
      }
      __syncthreads();
      for (int k0 = 0; k0 < KC; k0 += Mma<T>::KSTEP) {
#pragma unroll
        for (int dxi = 0; dxi < 3; ++dxi) {
          typename Mma<T>::Frag af[2];
#pragma unroll
          for (int m = 0; m < 2; ++m) af[m] = Mma<T>::load(halo + ((mf0 + m + dyi) * HW + dxi) * LDK, LDK, 0, k0, lane);
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            const int r = dyi - a;
            if (r < 0 || r > 1) continue;
            const int al = dyi == 1 ? a : 0;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
              const int s2 = dxi - b;
              if (s2 < 0 || s2 > 1) continue;
              const int slot = al * 4 + b * 2 + s2;
#pragma unroll
              for (int n = 0; n < FN; ++n) {
                typename Mma<T>::Frag bf = Mma<T>::load(Bs + (slot * BN + n * 16) * LDK, LDK, 0, k0, lane);
#pragma unroll
                for (int m = 0; m < 2; ++m) acc[a * 2 + b][m][n] = Mma<T>::mma(bf, af[m], acc[a * 2 + b][m][n]);   // D[m = cout][n = pixel]
              }
            }
          }
        }
      }
    }
    __syncthreads();
  }
  // epilogue: weights are the MFMA A operand, so a lane holds 4 CONSECUTIVE couts (n0 + 16 n + 4 (lane>>4) ..) of pixel
  // (row mf0 + m, column lane & 15) -> one 8/16-byte store per fragment (v0 of this kernel stored single elements and
  // picked the activation per element at run time: ~5000 instructions per tile)
  const int Ho = 2 * Hi, Wo = 2 * Wi;
  T* Yf = Y + (long long)f * Ho * Wo * Cout;
  const int g4 = (lane >> 4) * 4, px = tx0 + (lane & 15);
  const bool elu = act == ACT_ELU;
#pragma unroll
  for (int n = 0; n < FN; ++n) {
    const int col = n0 + n * 16 + g4;
    if (col >= Cout) continue;                       // Cout % 4 == 0 (checked by the launcher)
    const float4 bv = *reinterpret_cast<const float4*>(bias + col);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int a = p >> 1, b = p & 1;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int py = ty0 + mf0 + m;
        if (py >= Hi || px >= Wi) continue;
        float v[4] = {acc[p][m][n][0] + bv.x, acc[p][m][n][1] + bv.y, acc[p][m][n][2] + bv.z, acc[p][m][n][3] + bv.w};
        if (elu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = elu_bf(v[e]);
        } else if (act != ACT_NONE) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], act);
        }
        st4(Yf + ((long long)(2 * py + a) * Wo + 2 * px + b) * Cout + col, v);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// dgrad
// ---------------------------------------------------------------------------------------------------
template <typename T, int FN>
__global__ __launch_bounds__(256) void upconv_dgrad_kernel(const T* dP, const T* Wd, T* dX, const T* Xelu, int F, int Hi, int Wi, int Cin, int Cout) {
  constexpr int BN = FN * 16;
  constexpr int LDK = KC + ConvPad<T>::P;
  constexpr int VN = Vec<T>::N, CPR = KC / VN;
  constexpr int HH = 2 * TILE_H + 2, HW = 2 * TILE_W + 2;
  __shared__ __attribute__((aligned(16))) T halo[HH * HW * LDK];
  __shared__ __attribute__((aligned(16))) T Bs[4 * BN * LDK];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int tiles_x = (Wi + TILE_W - 1) / TILE_W, tiles_y = (Hi + TILE_H - 1) / TILE_H;
  int bid = blockIdx.x;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y; const int f = bid / tiles_y;
  const int ty0 = ty * TILE_H, tx0 = tx * TILE_W, n0 = blockIdx.y * BN;
  const int mf0 = 2 * w;
  const int Ho = 2 * Hi, Wo = 2 * Wi;

  f32x4 acc[2][FN];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < FN; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const T* Pf = dP + (long long)f * Ho * Wo * Cout;
  const uint4 z4 = make_uint4(0, 0, 0, 0);
  for (int c0 = 0; c0 < Cout; c0 += KC) {
    for (int i = tid; i < HH * HW * CPR; i += 256) {
      const int px = i / CPR, ch = (i % CPR) * VN;
      const int gy = 2 * ty0 + px / HW - 1, gx = 2 * tx0 + px % HW - 1;
      uint4 v = z4;
      if (gy >= 0 && gy < Ho && gx >= 0 && gx < Wo && c0 + ch < Cout)
        v = *reinterpret_cast<const uint4*>(Pf + ((long long)gy * Wo + gx) * Cout + c0 + ch);
      *reinterpret_cast<uint4*>(halo + px * LDK + ch) = v;
    }
    for (int ui = 0; ui < 4; ++ui) {
      __syncthreads();                 // previous group's B reads finished (and halo staged on ui == 0)
      for (int i = tid; i < 4 * BN * CPR; i += 256) {
        const int ch = (i % CPR) * VN, n = (i / CPR) % BN, vi = i / (CPR * BN);
        uint4 v = z4;
        if (n0 + n < Cin && c0 + ch < Cout)
          v = *reinterpret_cast<const uint4*>(Wd + ((long long)(ui * 4 + vi) * Cin + n0 + n) * Cout + c0 + ch);
        *reinterpret_cast<uint4*>(Bs + (vi * BN + n) * LDK + ch) = v;
      }
      __syncthreads();
      for (int k0 = 0; k0 < KC; k0 += Mma<T>::KSTEP) {
#pragma unroll
        for (int vi = 0; vi < 4; ++vi) {
          typename Mma<T>::Frag af[2];
#pragma unroll
          for (int m = 0; m < 2; ++m)
            af[m] = Mma<T>::load(halo + ((2 * (mf0 + m) + ui) * HW + vi) * LDK, 2 * LDK, 0, k0, lane);
#pragma unroll
          for (int n = 0; n < FN; ++n) {
            typename Mma<T>::Frag bf = Mma<T>::load(Bs + (vi * BN + n * 16) * LDK, LDK, 0, k0, lane);
#pragma unroll
            for (int m = 0; m < 2; ++m) acc[m][n] = Mma<T>::mma(bf, af[m], acc[m][n]);   // D[m = cin][n = pixel]
          }
        }
      }
    }
    __syncthreads();
  }
  // lane = 4 consecutive cins (n0 + 16 n + 4 (lane>>4) ..) of pixel (row mf0 + m, column lane & 15)
  T* Xf = dX + (long long)f * Hi * Wi * Cin;
  const int g4 = (lane >> 4) * 4, px = tx0 + (lane & 15);
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int py = ty0 + mf0 + m;
    if (py >= Hi || px >= Wi) continue;
#pragma unroll
    for (int n = 0; n < FN; ++n) {
      const int col = n0 + n * 16 + g4;
      if (col >= Cin) continue;                      // Cin % 4 == 0 (checked by the launcher)
      const long long o = ((long long)py * Wi + px) * Cin + col;
      float v[4] = {acc[m][n][0], acc[m][n][1], acc[m][n][2], acc[m][n][3]};
      if (Xelu) {
        float xv[4];
        ld4(Xelu + (long long)f * Hi * Wi * Cin + o, xv);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= xv[e] > 0.f ? 1.f : xv[e] + 1.f;
      }
      st4(Xf + o, v);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// dgrad of the wide layers (192 <- 128, 384 <- 192; 16-bit, whole tiles), pipelined.  The kernel above stages one quarter of the tap
// matrices at a time, synchronously: 16 dependent global -> LDS -> barrier -> 32 MFMA rounds per tile, each exposing a full memory
// latency (PMC: 153 us per launch, MFMA busy 12 %, 72 % of the wave cycles waiting).  Here a K-chunk (32 couts) of the dP halo AND of
// all 16 tap matrices (16 x 64 x 32) is resident at once (131 KB LDS, one persistent workgroup per CU): one barrier pair per 128 MFMAs
// per wave, and the next halo chunk (of the next tile of a group of four that share the weight chunk) and the next weight chunk are in
// flight in 26 registers per lane meanwhile.
// All loads, LDS writes and stores are unconditional (clamped halo addresses, zeroed at the commit through a pinned mask: conv_ws.hip
// has the story), so the commit's wait counts only what it needs.
// ---------------------------------------------------------------------------------------------------
template <typename T, int FN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void upconv_dgrad_pf_kernel(const T* __restrict__ dP, const T* __restrict__ Wd, T* __restrict__ dX,
                                                                 const T* __restrict__ Xelu, int F, int Hi, int Wi, int Cin, int Cout, int ntiles, int nseq) {
  static_assert(sizeof(T) == 2, "16-bit operands");
  constexpr int BN = FN * 16, LDK = KC + 8, CPR = KC / 8;
  constexpr int HH = 2 * TILE_H + 2, HW = 2 * TILE_W + 2, HPIX = HH * HW;
  constexpr int NH = (HPIX * CPR + 255) / 256, NB = 16 * BN * CPR / 256;
  static_assert(16 * BN * CPR % 256 == 0, "tap matrices split evenly over the threads");
  extern __shared__ __attribute__((aligned(16))) unsigned char pf_smem[];
  T* halo = reinterpret_cast<T*>(pf_smem);             // [HPIX][LDK]
  T* Bs = halo + HPIX * LDK;                           // [16 taps][BN][LDK]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // Work map: workgroups are dealt to the 8 XCDs round-robin by linear id and every XCD has its own L2.  The ct cin tiles of one tile
  // sequence read the same dP halo, so they get adjacent slots of the SAME XCD (bxs = this workgroup's tile sequence, nseq of them)
  const int ct = Cin / BN;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int bxs = (slot / ct) * 8 + xcd, byc = slot % ct;
  if (bxs >= nseq) return;
  const int n0 = byc * BN, mf0 = 2 * w;
  const int tiles_x = Wi / TILE_W, tiles_y = Hi / TILE_H;
  const int Ho = 2 * Hi, Wo = 2 * Wi;
  // chunk geometry: with CPR = 4 a thread's chunks are pixel (tid >> 2) + 64 i, channels 8 (tid & 3) .. of the halo, and cin row
  // (tid >> 2) of tap j of the tap matrices: everything but the halo pixel's (row, column) is a compile-time stride from one base
  static_assert(CPR == 4 && BN == 64, "chunk geometry below");
  const int p0 = tid >> 2, ch0 = (tid & 3) * 8;
  const bool dup_last = tid >= HPIX * CPR - (NH - 1) * 256;       // surplus chunk of the last round: the previous chunk again
  const int bg0 = (n0 + p0) * Cout + ch0, bl0 = p0 * LDK + ch0, hl0 = p0 * LDK + ch0;
  uint4 hb[NH], bb[NB];
  int hmsk = 0;
  auto fetch = [&](int tile, int c0) {
    const int tx = tile % tiles_x, t2 = tile / tiles_x;
    const int ty0 = (t2 % tiles_y) * TILE_H, f = t2 / tiles_y, tx0 = tx * TILE_W;
    const T* Pf = dP + (long long)f * Ho * Wo * Cout + c0 + ch0;
    int m = 0;
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      const int px = p0 + 64 * ((i == NH - 1 && dup_last) ? i - 1 : i);
      const int gy = 2 * ty0 + px / HW - 1, gx = 2 * tx0 + px % HW - 1;
      const int cy = min(max(gy, 0), Ho - 1), cx = min(max(gx, 0), Wo - 1);
      m |= (cy == gy && cx == gx) ? (1 << i) : 0;
      hb[i] = *reinterpret_cast<const uint4*>(Pf + (cy * Wo + cx) * Cout);
    }
    hmsk = m;
  };
  auto fetch_w = [&](int c0) {
    const T* Wc = Wd + bg0 + c0;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const uint4 t = *reinterpret_cast<const uint4*>(Wc + (long long)j * Cin * Cout);
      bb[j] = make_uint4(t.x, t.y, t.z, t.w);        // (component-wise: a whole-struct copy of the array element leaves bb[] in scratch)
    }
  };
  auto commit = [&]() {
    int m = hmsk;
    asm volatile("" : "+v"(m));                        // the zeroing (and the wait for the loads) stays below the barrier
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      const bool in = (m >> i) & 1;
      const int o = hl0 + 64 * LDK * ((i == NH - 1 && dup_last) ? i - 1 : i);
      *reinterpret_cast<uint4*>(halo + o) = make_uint4(in ? hb[i].x : 0u, in ? hb[i].y : 0u, in ? hb[i].z : 0u, in ? hb[i].w : 0u);
    }
  };
  auto commit_w = [&]() {
#pragma unroll
    for (int j = 0; j < NB; ++j) *reinterpret_cast<uint4*>(Bs + bl0 + j * BN * LDK) = make_uint4(bb[j].x, bb[j].y, bb[j].z, bb[j].w);
  };
  // TG tiles of this workgroup share every weight chunk: with one tile per chunk the 65 KB of tap matrices were re-read from L2 for
  // each of 4 x 512 x 3 (chunk, tile, cin tile) steps -- 0.64 GB per launch of the 192 <- 128 layer at the L2 -> L1 ceiling (~6.5 TB/s)
  // that the 64 x 64 GEMM tiles hit as well.  The TG accumulator sets live in AGPRs (one wave per SIMD: 512 registers).
  constexpr int TG = 4;                                // measured 76 / 107 us; TG = 2: 82 / 112; TG = 1: 91 / 109
  const int last_tile = ntiles - 1;
  const int nchunk = Cout / KC;
  auto tile_of = [&](int k) { return min(bxs + k * nseq, last_tile); };
  const int my_tiles = (bxs < ntiles) ? (ntiles - 1 - bxs) / nseq + 1 : 0;
  fetch(tile_of(0), 0);
  fetch_w(0);
  for (int k0 = 0; k0 < my_tiles; k0 += TG) {
    f32x4 acc[TG][2][FN];
#pragma unroll
    for (int t = 0; t < TG; ++t)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < FN; ++n) acc[t][m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int ci = 0; ci < nchunk; ++ci) {
      const int c0 = ci * KC;
#pragma unroll
      for (int t = 0; t < TG; ++t) {
        __syncthreads();                               // the previous step's fragment reads are finished
        commit();
        if (t == 0) commit_w();
        __syncthreads();
        // next step: the next tile of the group at this chunk, or the group's first tile at the next chunk, or the next group
        const bool last_t = t == TG - 1;
        const bool more = ci + 1 < nchunk;
        const int nk = !last_t ? k0 + t + 1 : (more ? k0 : k0 + TG);
        const int nc = !last_t ? c0 : (more ? c0 + KC : 0);
        fetch(tile_of(nk), nc);                        // in flight during the 128 MFMAs below (past the end: fetched, never used)
        if (t == 0) fetch_w(more ? c0 + KC : 0);       // the NEXT chunk's tap matrices: four steps to arrive
        // one wave per SIMD: nothing else hides the LDS latency, so the fragments of tap q + 1 are read before the MFMAs of tap q
        // are issued (left to the compiler the reads came two MFMAs ahead, with an lgkmcnt wait in front of every second MFMA).
        // (Two / three / four taps ahead, round 6: 68.9 / 70.2 / 71.8 us against 67.6 at 384 -> 192 and 94.5 / 97.0 / 99.7 against 93.6 at
        //  192 -> 128: one tap ahead is the depth, profiles/r06_zl_pf_depth.txt.)
        typedef typename Mma<T>::Frag Frag;
        Frag af[2][2], bf[2][FN];
        auto load_tap = [&](int q, Frag (&a)[2], Frag (&b)[FN]) {
          const int ui = q >> 2, vi = q & 3;
#pragma unroll
          for (int m = 0; m < 2; ++m) a[m] = Mma<T>::load(halo + ((2 * (mf0 + m) + ui) * HW + vi) * LDK, 2 * LDK, 0, 0, lane);
#pragma unroll
          for (int n = 0; n < FN; ++n) b[n] = Mma<T>::load(Bs + (q * BN + n * 16) * LDK, LDK, 0, 0, lane);
        };
        load_tap(0, af[0], bf[0]);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          if (q + 1 < 16) load_tap(q + 1, af[(q + 1) & 1], bf[(q + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);           // (the machine scheduler sinks the reads back to their uses otherwise)
#pragma unroll
          for (int n = 0; n < FN; ++n)
#pragma unroll
            for (int m = 0; m < 2; ++m) acc[t][m][n] = Mma<T>::mma(bf[q & 1][n], af[q & 1][m], acc[t][m][n]);   // D[m = cin][n = pixel]
        }
      }
    }
    // lane = 4 consecutive cins (n0 + 16 n + 4 (lane >> 4) ..) of pixel (row mf0 + m, column lane & 15)
#pragma unroll
    for (int t = 0; t < TG; ++t) {
      if (k0 + t >= my_tiles) break;                   // uniform: the group's tail
      const int tile = bxs + (k0 + t) * nseq;
      const int tx = tile % tiles_x, t2 = tile / tiles_x;
      const int ty0 = (t2 % tiles_y) * TILE_H, f = t2 / tiles_y, tx0 = tx * TILE_W;
      const long long fo = (long long)f * Hi * Wi * Cin;
      const int g4 = (lane >> 4) * 4, px = tx0 + (lane & 15);
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int py = ty0 + mf0 + m;
#pragma unroll
        for (int n = 0; n < FN; ++n) {
          const long long o = fo + ((long long)py * Wi + px) * Cin + n0 + n * 16 + g4;
          float v[4] = {acc[t][m][n][0], acc[t][m][n][1], acc[t][m][n][2], acc[t][m][n][3]};
          if (Xelu) {                                  // uniform
            float xv[4];
            ld4(Xelu + o, xv);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= xv[e] > 0.f ? 1.f : xv[e] + 1.f;
          }
          st4(dX + o, v);
        }
      }
    }
  }
}
template <typename T>
static bool upconv_dgrad_pf_try(const void* dP, const void* Wd, void* dX, const void* Xelu, int F, int Hi, int Wi, int Cin, int Cout, hipStream_t st) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("STJ_NO_WS"); on = !(e && atoi(e)); }      // STJ_NO_WS=1: generic conv kernels everywhere
  if (!on || Hi % TILE_H || Wi % TILE_W || Cin % 64 || Cout % KC || Cin <= 128) return false;
  constexpr int FN = 4, BN = FN * 16, LDK = KC + 8;
  const size_t lds = (size_t)((2 * TILE_H + 2) * (2 * TILE_W + 2) * LDK + 16 * BN * LDK) * sizeof(T);
  static PerDevice<bool> attr_set;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)upconv_dgrad_pf_kernel<T, FN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
    attr_set = true;
  }
  const int ntiles = (Wi / TILE_W) * (Hi / TILE_H) * F, ct = Cin / BN;
  // tile sequences: a multiple of 8 (whole XCD rounds of the work map) with all their workgroups resident at once (one per CU, 32 CUs
  // per XCD -- with 85 sequences x 3 cin tiles five XCDs got 33 workgroups and the kernel took two rounds)
  int nblk = 256 / ct / 8 * 8;
  if (nblk < 8) nblk = 8;
  if (nblk > ntiles) nblk = ntiles;
  // ... and no more of them than the step count needs: a sequence runs ceil(tiles / 4) groups of four tile slots, so 64 sequences of
  // 8 tiles take as long as 80 of 6-7 and leave a quarter of the L2 / LDS traffic's contenders out (96 -> 93 us, 70 -> 68 us)
  auto slots = [&](int n) { return ((ntiles + n - 1) / n + 3) / 4 * 4; };
  while (nblk > 8 && slots(nblk - 8) == slots(nblk)) nblk -= 8;
  const int nseq8 = (nblk + 7) / 8 * 8;                // (ntiles < 8: padded, the surplus workgroups exit at once)
  hipLaunchKernelGGL((upconv_dgrad_pf_kernel<T, FN>), dim3(nseq8 * ct), dim3(256), lds, st, (const T*)dP, (const T*)Wd, (T*)dX, (const T*)Xelu, F, Hi, Wi,
                     Cin, Cout, ntiles, nblk);
  return true;
}

// ---------------------------------------------------------------------------------------------------
// wgrad: block = (pixel strip, phase, cout-tile x cin-tile); wave = tap (r,s); K = pixels in chunks of one
// low-res row segment of 32 columns.
// ---------------------------------------------------------------------------------------------------
#define WG_W 32
template <typename T, int FO, int FI>
__global__ __launch_bounds__(256) void upconv_wgrad_kernel(const T* X, const T* dP, float* dWeff, float* dbias, int db_parts, int F, int Hi, int Wi,
                                                           int Cin, int Cout, int chunks_per_block) {
  constexpr int BO = FO * 16, BI = FI * 16;
  constexpr int LDO = BO + ConvPad<T>::P, LDI = BI + ConvPad<T>::P;
  constexpr int VN = Vec<T>::N;
  constexpr int XW = WG_W + 2;
  __shared__ __attribute__((aligned(16))) T dYs[WG_W * LDO];
  __shared__ __attribute__((aligned(16))) T Xs[2 * XW * LDI];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int a = blockIdx.y >> 1, b = blockIdx.y & 1;
  const int r = w >> 1, s = w & 1;
  const int cin_tiles = (Cin + BI - 1) / BI;
  const int co0 = (blockIdx.z / cin_tiles) * BO, ci0 = (blockIdx.z % cin_tiles) * BI;
  const int segs = (Wi + WG_W - 1) / WG_W;
  const long long nchunks = (long long)F * Hi * segs;
  const long long c_begin = (long long)blockIdx.x * chunks_per_block;
  const long long c_end = min(nchunks, c_begin + chunks_per_block);
  const int Ho = 2 * Hi, Wo = 2 * Wi;

  f32x4 acc[FO][FI];
#pragma unroll
  for (int m = 0; m < FO; ++m)
#pragma unroll
    for (int n = 0; n < FI; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const uint4 z4 = make_uint4(0, 0, 0, 0);
  const bool do_db = dbias != nullptr && ci0 == 0 && w == 0 && lane < BO;   // fused bias gradient (sum of dP over pixels)
  float dbacc = 0.f;

  for (long long c = c_begin; c < c_end; ++c) {
    const int seg = (int)(c % segs); long long t = c / segs;
    const int i = (int)(t % Hi); const int f = (int)(t / Hi);
    const int j0 = seg * WG_W;
    // dP of this phase: pixels (2i+a, 2(j0+cj)+b), cj in [0,32)
    for (int q = tid; q < WG_W * (BO / VN); q += 256) {
      const int cj = q / (BO / VN), ch = (q % (BO / VN)) * VN;
      uint4 v = z4;
      if (j0 + cj < Wi && co0 + ch < Cout)
        v = *reinterpret_cast<const uint4*>(dP + (((long long)f * Ho + 2 * i + a) * Wo + 2 * (j0 + cj) + b) * Cout + co0 + ch);
      *reinterpret_cast<uint4*>(dYs + cj * LDO + ch) = v;
    }
    // X rows i+a-1+{0,1}, cols j0+b-1 .. j0+b-1+33
    for (int q = tid; q < 2 * XW * (BI / VN); q += 256) {
      const int ch = (q % (BI / VN)) * VN; const int p = q / (BI / VN);
      const int rr = p / XW, cc = p % XW;
      const int gy = i + a - 1 + rr, gx = j0 + b - 1 + cc;
      uint4 v = z4;
      if (gy >= 0 && gy < Hi && gx >= 0 && gx < Wi && ci0 + ch < Cin)
        v = *reinterpret_cast<const uint4*>(X + (((long long)f * Hi + gy) * Wi + gx) * Cin + ci0 + ch);
      *reinterpret_cast<uint4*>(Xs + p * LDI + ch) = v;
    }
    __syncthreads();
    if (do_db) {
      float sdb = 0.f;
      for (int q = 0; q < WG_W; ++q) sdb += ldf(dYs + q * LDO + lane);
      dbacc += sdb;
    }
    for (int k0 = 0; k0 < WG_W; k0 += Mma<T>::KSTEP) {
      typename Mma<T>::Frag af[FO], bf[FI];
#pragma unroll
      for (int m = 0; m < FO; ++m) af[m] = Mma<T>::load_strided(dYs, 1, LDO, m * 16, k0, lane);
#pragma unroll
      for (int n = 0; n < FI; ++n) bf[n] = Mma<T>::load_strided(Xs + (r * XW + s) * LDI, 1, LDI, n * 16, k0, lane);
#pragma unroll
      for (int m = 0; m < FO; ++m)
#pragma unroll
        for (int n = 0; n < FI; ++n) acc[m][n] = Mma<T>::mma(af[m], bf[n], acc[m][n]);
    }
    __syncthreads();
  }
  if (do_db && co0 + lane < Cout) atomicAdd(dbias + (long long)(blockIdx.x % db_parts) * Cout + co0 + lane, dbacc);
  const int pt = a * 8 + b * 4 + r * 2 + s;
#pragma unroll
  for (int m = 0; m < FO; ++m)
#pragma unroll
    for (int n = 0; n < FI; ++n) {
      const int ci = ci0 + n * 16 + (lane & 15);
      if (ci >= Cin) continue;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int co = co0 + m * 16 + (lane >> 4) * 4 + rg;
        if (co < Cout) atomicAdd(dWeff + ((long long)pt * Cout + co) * Cin + ci, acc[m][n][rg]);
      }
    }
}

// ---------------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------------
extern "C" int stj_upconv_prep(const float* W, void* Wf, void* Wd, int Cin, int Cout, int dtype, hipStream_t stream) {
  const int g = min(2048, (16 * Cin * Cout + 255) / 256);
  if (dtype == STJ_BF16) hipLaunchKernelGGL(upconv_prep_kernel<bf16>, dim3(g), dim3(256), 0, stream, W, (bf16*)Wf, (bf16*)Wd, Cin, Cout);
  else if (dtype == STJ_F16) hipLaunchKernelGGL(upconv_prep_kernel<f16>, dim3(g), dim3(256), 0, stream, W, (f16*)Wf, (f16*)Wd, Cin, Cout);
  else hipLaunchKernelGGL(upconv_prep_kernel<float>, dim3(g), dim3(256), 0, stream, W, (float*)Wf, (float*)Wd, Cin, Cout);
  return stj_check_launch("stj_upconv_prep");
}
extern "C" int stj_upconv_fold(const float* dWeff, float* dW, int Cin, int Cout, hipStream_t stream) {
  const int g = min(2048, (9 * Cin * Cout + 255) / 256);
  hipLaunchKernelGGL(upconv_fold_kernel, dim3(g), dim3(256), 0, stream, dWeff, dW, Cin, Cout);
  return stj_check_launch("stj_upconv_fold");
}

static int upconv_check(int F, int Hi, int Wi, int Cin, int Cout, int dtype) {
  const int vn = stj_is16(dtype) ? 8 : 4;
  if (F <= 0 || Hi <= 0 || Wi <= 0) { stj_set_error("upconv: empty problem"); return STJ_EINVAL; }
  if (Cin % vn || Cout % vn) { stj_set_error("upconv: channels must be multiples of %d (Cin=%d Cout=%d)", vn, Cin, Cout); return STJ_EINVAL; }
  return STJ_OK;
}

template <typename T>
static int upconv_fwd_launch(const void* X, const void* Wf, const float* bias, void* Y, int F, int Hi, int Wi, int Cin, int Cout, int act, hipStream_t st) {
  const int tiles = ((Wi + TILE_W - 1) / TILE_W) * ((Hi + TILE_H - 1) / TILE_H) * F;
  if (Cout % 64 == 0 || Cout > 96) {
    hipLaunchKernelGGL((upconv_fwd_kernel<T, 4>), dim3(tiles, (Cout + 63) / 64), dim3(256), 0, st, (const T*)X, (const T*)Wf, bias, (T*)Y, F, Hi, Wi, Cin, Cout, act);
  } else {
    hipLaunchKernelGGL((upconv_fwd_kernel<T, 3>), dim3(tiles, (Cout + 47) / 48), dim3(256), 0, st, (const T*)X, (const T*)Wf, bias, (T*)Y, F, Hi, Wi, Cin, Cout, act);
  }
  return stj_check_launch("stj_upconv_fwd");
}
bool upconv_fwd_ws_try(const void* X, const void* Wf, const float* bias, void* Y, int F, int Hi, int Wi, int Cin, int Cout, int act,
                       int dtype, hipStream_t st);   // conv_ws.hip
bool upconv_fwd_ps_try(const void* X, const void* Wf, const float* bias, void* Y, int F, int Hi, int Wi, int Cin, int Cout, int act,
                       int dtype, hipStream_t st);   // conv_ps.hip
bool upconv_wgrad_tr_try(const void* X, const void* dP, float* dWeff, float* dbias, int db_parts, int F, int Hi, int Wi, int Cin, int Cout, int wg_budget, hipStream_t st);
bool outconv_fwd_mfma_try(const void* X, const float* W, const float* bias, float* Y, int F, int Hh, int Ww, int C, int Tn,
                          long long y_bs, long long y_ts, long long y_ps, int dtype, hipStream_t st);
bool outconv_bwd_mfma_try(const void* X, const float* W, const float* dY, void* dX, float* dW, float* db, int F, int Hh, int Ww, int C,
                          int Tn, long long y_bs, long long y_ts, long long y_ps, int elu_in, void* ws, long long ws_bytes, hipStream_t st);
long long outconv_bwd_ws_bytes();
bool upconv_dgrad_ws_try(const void* dP, const void* Wd, void* dX, const void* Xelu, int F, int Hi, int Wi, int Cin, int Cout, hipStream_t st);
static bool ws_enabled() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("STJ_NO_WS"); v = (e && e[0] == '1') ? 0 : 1; }
  return v == 1;
}
extern "C" int stj_upconv_fwd(const void* X, const void* Wf, const float* bias, void* Y, int F, int Hi, int Wi, int Cin,
                              int Cout, int act, int dtype, hipStream_t stream) {
  int e = upconv_check(F, Hi, Wi, Cin, Cout, dtype);
  if (e) return e;
  if (stj_is16(dtype) && ws_enabled() && upconv_fwd_ws_try(X, Wf, bias, Y, F, Hi, Wi, Cin, Cout, act, dtype, stream))
    return stj_check_launch("stj_upconv_fwd(ws)");
  if (stj_is16(dtype) && ws_enabled() && upconv_fwd_ps_try(X, Wf, bias, Y, F, Hi, Wi, Cin, Cout, act, dtype, stream))
    return stj_check_launch("stj_upconv_fwd(ps)");
  if (dtype == STJ_F16) return upconv_fwd_launch<f16>(X, Wf, bias, Y, F, Hi, Wi, Cin, Cout, act, stream);
  return dtype == STJ_BF16 ? upconv_fwd_launch<bf16>(X, Wf, bias, Y, F, Hi, Wi, Cin, Cout, act, stream)
                           : upconv_fwd_launch<float>(X, Wf, bias, Y, F, Hi, Wi, Cin, Cout, act, stream);
}

// Up-conv with the decoder's skip sums in its epilogue: Y = ELU(conv(up(X)) + bias) + R1 and (optional) Y2 = Y + R2, all [F,2Hi,2Wi,Cout]
// in the activation dtype, each sum rounded like a separate elementwise add.  Covered: the 16-bit wide layers (Cin = 192, 384: the two
// levels that take skips, modules.py:750-765); STJ_EUNSUPPORTED otherwise (run stj_upconv_fwd and add).
bool upconv_fwd_ps_res_try(const void* X, const void* Wf, const float* bias, void* Y, const void* R1, void* Y2, const void* R2, int F, int Hi,
                           int Wi, int Cin, int Cout, int dtype, hipStream_t st);
extern "C" int stj_upconv_fwd_res(const void* X, const void* Wf, const float* bias, void* Y, const void* R1, void* Y2, const void* R2, int F,
                                  int Hi, int Wi, int Cin, int Cout, int dtype, hipStream_t stream) {
  int e = upconv_check(F, Hi, Wi, Cin, Cout, dtype);
  if (e) return e;
  if (stj_is16(dtype) && ws_enabled() && upconv_fwd_ps_res_try(X, Wf, bias, Y, R1, Y2, R2, F, Hi, Wi, Cin, Cout, dtype, stream))
    return stj_check_launch("stj_upconv_fwd_res");
  stj_set_error("upconv_fwd_res: shape / dtype not covered (Cin=%d Cout=%d dtype=%d)", Cin, Cout, dtype);
  return STJ_EUNSUPPORTED;
}

template <typename T>
static int upconv_dgrad_launch(const void* dP, const void* Wd, void* dX, const void* Xelu, int F, int Hi, int Wi, int Cin, int Cout, hipStream_t st) {
  const int tiles = ((Wi + TILE_W - 1) / TILE_W) * ((Hi + TILE_H - 1) / TILE_H) * F;
  if (Cin % 96 == 0 && Cin % 64 != 0) {
    hipLaunchKernelGGL((upconv_dgrad_kernel<T, 6>), dim3(tiles, Cin / 96), dim3(256), 0, st, (const T*)dP, (const T*)Wd, (T*)dX, (const T*)Xelu, F, Hi, Wi, Cin, Cout);
  } else {
    hipLaunchKernelGGL((upconv_dgrad_kernel<T, 4>), dim3(tiles, (Cin + 63) / 64), dim3(256), 0, st, (const T*)dP, (const T*)Wd, (T*)dX, (const T*)Xelu, F, Hi, Wi, Cin, Cout);
  }
  return stj_check_launch("stj_upconv_dgrad");
}
// Xelu (nullable): the layer input X when it is itself an ELU output; dX is then multiplied by ELU'(x) = (x > 0 ? 1 : x + 1).
extern "C" int stj_upconv_dgrad(const void* dP, const void* Wd, void* dX, const void* Xelu, int F, int Hi, int Wi, int Cin, int Cout,
                                int dtype, hipStream_t stream) {
  int e = upconv_check(F, Hi, Wi, Cin, Cout, dtype);
  if (e) return e;
  if (dtype == STJ_BF16 && ws_enabled() && upconv_dgrad_ws_try(dP, Wd, dX, Xelu, F, Hi, Wi, Cin, Cout, stream))
    return stj_check_launch("stj_upconv_dgrad(ws)");
  if (dtype == STJ_BF16 && upconv_dgrad_pf_try<bf16>(dP, Wd, dX, Xelu, F, Hi, Wi, Cin, Cout, stream)) return stj_check_launch("stj_upconv_dgrad(pf)");
  if (dtype == STJ_F16 && upconv_dgrad_pf_try<f16>(dP, Wd, dX, Xelu, F, Hi, Wi, Cin, Cout, stream)) return stj_check_launch("stj_upconv_dgrad(pf)");
  if (dtype == STJ_F16) return upconv_dgrad_launch<f16>(dP, Wd, dX, Xelu, F, Hi, Wi, Cin, Cout, stream);
  return dtype == STJ_BF16 ? upconv_dgrad_launch<bf16>(dP, Wd, dX, Xelu, F, Hi, Wi, Cin, Cout, stream)
                           : upconv_dgrad_launch<float>(dP, Wd, dX, Xelu, F, Hi, Wi, Cin, Cout, stream);
}

template <typename T>
static int upconv_wgrad_launch(const void* X, const void* dP, float* dWeff, float* dbias, int db_parts, int F, int Hi, int Wi, int Cin, int Cout, hipStream_t st) {
  const long long nchunks = (long long)F * Hi * ((Wi + WG_W - 1) / WG_W);
  if (Cout <= 48 && Cin <= 96) {
    const int tiles = 1;
    int strips = (int)min(nchunks, (long long)(1024 / (4 * tiles)));
    const int cpb = (int)((nchunks + strips - 1) / strips);
    strips = (int)((nchunks + cpb - 1) / cpb);
    hipLaunchKernelGGL((upconv_wgrad_kernel<T, 3, 6>), dim3(strips, 4, tiles), dim3(256), 0, st, (const T*)X, (const T*)dP, dWeff, dbias, db_parts, F, Hi, Wi, Cin, Cout, cpb);
  } else {
    const int tiles = ((Cout + 63) / 64) * ((Cin + 63) / 64);
    int strips = (int)min(nchunks, (long long)max(1, 1024 / (4 * tiles)));
    const int cpb = (int)((nchunks + strips - 1) / strips);
    strips = (int)((nchunks + cpb - 1) / cpb);
    hipLaunchKernelGGL((upconv_wgrad_kernel<T, 4, 4>), dim3(strips, 4, tiles), dim3(256), 0, st, (const T*)X, (const T*)dP, dWeff, dbias, db_parts, F, Hi, Wi, Cin, Cout, cpb);
  }
  return stj_check_launch("stj_upconv_wgrad");
}
// dWeff: f32 [16][Cout][Cin] scratch, must be zero on entry (caller memsets); fold with stj_upconv_fold afterwards.
// dbias (optional): f32 [db_parts][Cout], "+=": the sum over all pixels of dP (the conv bias gradient); workgroup i adds into copy
// i % db_parts and the caller sums the copies (one copy = ~1000 same-address atomics per channel at the end of the kernel).
// wg_budget: workgroups the two large layers (>= 64x64 inputs) may occupy; 0 = 128, half the CUs (a step whose branches run
// concurrently leaves the other half to them), 256 = one per CU (the kernel alone on the GPU).
extern "C" int stj_upconv_wgrad(const void* X, const void* dP, float* dWeff, float* dbias, int db_parts, int F, int Hi, int Wi, int Cin,
                                int Cout, int wg_budget, int dtype, hipStream_t stream) {
  int e = upconv_check(F, Hi, Wi, Cin, Cout, dtype);
  if (e) return e;
  if (dbias && db_parts < 1) { stj_set_error("upconv_wgrad: db_parts must be >= 1"); return STJ_EINVAL; }
  if (wg_budget < 0 || wg_budget > 4096) { stj_set_error("upconv_wgrad: wg_budget %d out of range", wg_budget); return STJ_EINVAL; }
  if (db_parts < 1) db_parts = 1;
  if (dtype == STJ_BF16 && ws_enabled() && upconv_wgrad_tr_try(X, dP, dWeff, dbias, db_parts, F, Hi, Wi, Cin, Cout, wg_budget, stream))
    return stj_check_launch("stj_upconv_wgrad(tr)");
  if (dtype == STJ_F16) return upconv_wgrad_launch<f16>(X, dP, dWeff, dbias, db_parts, F, Hi, Wi, Cin, Cout, stream);
  return dtype == STJ_BF16 ? upconv_wgrad_launch<bf16>(X, dP, dWeff, dbias, db_parts, F, Hi, Wi, Cin, Cout, stream)
                           : upconv_wgrad_launch<float>(X, dP, dWeff, dbias, db_parts, F, Hi, Wi, Cin, Cout, stream);
}

// ---------------------------------------------------------------------------------------------------
// 3x3 SAME conv C->2 (no activation), HBM-streaming: each block stages an 18x18 halo of all C channels in LDS
// (f32), one thread per output pixel.  Output is written with arbitrary pixel/frame strides so the two heads
// land directly in the [B,H,W,8*4] result (channel 4t + {0,1} / {2,3}).  Y is f32 (model output).
// ---------------------------------------------------------------------------------------------------
#define OC_T 16
template <typename T>
__global__ __launch_bounds__(256) void outconv_fwd_kernel(const T* X, const float* W, const float* bias, float* Y, int Hh, int Ww,
                                                          int C, int Tn, long long y_bstride, long long y_tstride, long long y_pstride) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int LDC = C + 1;
  float* halo = smem;                       // [18*18][C+1]
  float* Ws = smem + 18 * 18 * LDC;         // [9][C][2]
  const int tid = threadIdx.x;
  const int tiles_x = Ww / OC_T, tiles_y = Hh / OC_T;
  int bid = blockIdx.x;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y; const int f = bid / tiles_y;
  const T* Xf = X + (long long)f * Hh * Ww * C;
  {
    constexpr int VN = Vec<T>::N;
    const int cpp = C / VN;                     // 16-byte chunks per pixel (C % VN == 0 checked on the host)
    for (int i = tid; i < 18 * 18 * cpp; i += 256) {
      const int c = (i % cpp) * VN, p = i / cpp;
      const int gy = ty * OC_T + p / 18 - 1, gx = tx * OC_T + p % 18 - 1;
      float v[VN];
      if (gy >= 0 && gy < Hh && gx >= 0 && gx < Ww) ld16(Xf + ((long long)gy * Ww + gx) * C + c, v);
      else {
#pragma unroll
        for (int e = 0; e < VN; ++e) v[e] = 0.f;
      }
#pragma unroll
      for (int e = 0; e < VN; ++e) halo[p * LDC + c + e] = v[e];
    }
  }
  for (int i = tid; i < 9 * C * 2; i += 256) Ws[i] = W[i];
  __syncthreads();
  const int py = tid / OC_T, px = tid % OC_T;
  float a0 = bias[0], a1 = bias[1];
  for (int t = 0; t < 9; ++t) {
    const float* h = halo + ((py + t / 3) * 18 + px + t % 3) * LDC;
    const float* wv = Ws + t * C * 2;
    for (int c = 0; c < C; ++c) { const float x = h[c]; a0 += x * wv[2 * c]; a1 += x * wv[2 * c + 1]; }
  }
  const int b = f / Tn, tt = f % Tn;
  float* dst = Y + b * y_bstride + tt * y_tstride + ((long long)(ty * OC_T + py) * Ww + tx * OC_T + px) * y_pstride;
  dst[0] = a0; dst[1] = a1;
}

// backward of the output conv: dX [F,H,W,C] (type T) and dW [3][3][C][2], db [2] (f32 atomics).
// dY is read with the same strides as Y was written.
template <typename T>
__global__ __launch_bounds__(256) void outconv_bwd_kernel(const T* X, const float* W, const float* dY, T* dX, float* dW, float* db,
                                                          int F, int Hh, int Ww, int C, int Tn, long long y_bstride, long long y_tstride,
                                                          long long y_pstride, int elu_in) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int LDC = C + 1;
  float* halo = smem;                       // X halo [18*18][C+1]
  float* Ws = halo + 18 * 18 * LDC;         // [9][C][2]
  float* dys = Ws + 9 * C * 2;              // dY halo [18*18][2]
  const int tid = threadIdx.x;
  const int tiles_x = Ww / OC_T, tiles_y = Hh / OC_T;
  const int ntiles = F * tiles_x * tiles_y;
  for (int i = tid; i < 9 * C * 2; i += 256) Ws[i] = W[i];
  // per-thread dW partials for items i = tid, tid+256, ... (< 9*C <= 4*256)
  float w0[4] = {0.f, 0.f, 0.f, 0.f}, w1[4] = {0.f, 0.f, 0.f, 0.f};
  float b0 = 0.f, b1 = 0.f;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int bid = tile;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y; const int f = bid / tiles_y;
    const int b = f / Tn, tt = f % Tn;
    const T* Xf = X + (long long)f * Hh * Ww * C;
    const float* dYf = dY + b * y_bstride + tt * y_tstride;
    __syncthreads();
    {
      constexpr int VN = Vec<T>::N;
      const int cpp = C / VN;
      for (int i = tid; i < 18 * 18 * cpp; i += 256) {
        const int c = (i % cpp) * VN, p = i / cpp;
        const int gy = ty * OC_T + p / 18 - 1, gx = tx * OC_T + p % 18 - 1;
        float v[VN];
        if (gy >= 0 && gy < Hh && gx >= 0 && gx < Ww) ld16(Xf + ((long long)gy * Ww + gx) * C + c, v);
        else {
#pragma unroll
          for (int e = 0; e < VN; ++e) v[e] = 0.f;
        }
#pragma unroll
        for (int e = 0; e < VN; ++e) halo[p * LDC + c + e] = v[e];
      }
    }
    for (int i = tid; i < 18 * 18; i += 256) {
      const int gy = ty * OC_T + i / 18 - 1, gx = tx * OC_T + i % 18 - 1;
      const bool ok = gy >= 0 && gy < Hh && gx >= 0 && gx < Ww;
      const float* src = dYf + ((long long)gy * Ww + gx) * y_pstride;
      dys[2 * i] = ok ? src[0] : 0.f;
      dys[2 * i + 1] = ok ? src[1] : 0.f;
    }
    __syncthreads();
    // dX[p][c] = sum_t sum_o dY[p - off(t)][o] W[t][c][o]   (halo index of p - off(t): (py+2-t/3, px+2-t%3))
    {
      const int py = tid / OC_T, px = tid % OC_T;
      T* dst = dX + (long long)f * Hh * Ww * C + ((long long)(ty * OC_T + py) * Ww + tx * OC_T + px) * C;
      float g0[9], g1[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int hp = (py + 2 - t / 3) * 18 + px + 2 - t % 3;
        g0[t] = dys[2 * hp]; g1[t] = dys[2 * hp + 1];
      }
      constexpr int VN = Vec<T>::N;
      for (int c0 = 0; c0 < C; c0 += VN) {
        float o[VN];
#pragma unroll
        for (int e = 0; e < VN; ++e) {
          float acc = 0.f;
#pragma unroll
          for (int t = 0; t < 9; ++t) acc += g0[t] * Ws[(t * C + c0 + e) * 2] + g1[t] * Ws[(t * C + c0 + e) * 2 + 1];
          if (elu_in) { const float xv = halo[((py + 1) * 18 + px + 1) * LDC + c0 + e]; acc *= xv > 0.f ? 1.f : xv + 1.f; }
          o[e] = acc;
        }
        st16(dst + c0, o);
      }
      const int hp = (py + 1) * 18 + px + 1;
      b0 += dys[2 * hp]; b1 += dys[2 * hp + 1];
    }
    // dW[t][c][o] += sum_p X[p + off(t)][c] dY[p][o] ; threads over (t,c)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int i = tid + it * 256;
      if (i < 9 * C) {
        const int c = i % C, t = i / C;
        float s0 = 0.f, s1 = 0.f;
        for (int p = 0; p < OC_T * OC_T; ++p) {
          const int py = p / OC_T, px = p % OC_T;
          const float x = halo[((py + t / 3) * 18 + px + t % 3) * LDC + c];
          const int hp = (py + 1) * 18 + px + 1;
          s0 += x * dys[2 * hp]; s1 += x * dys[2 * hp + 1];
        }
        w0[it] += s0; w1[it] += s1;
      }
    }
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int i = tid + it * 256;
    if (i < 9 * C) { atomicAdd(dW + i * 2, w0[it]); atomicAdd(dW + i * 2 + 1, w1[it]); }
  }
  b0 = wave_sum(b0); b1 = wave_sum(b1);
  if ((tid & 63) == 0) { atomicAdd(db, b0); atomicAdd(db + 1, b1); }
}

extern "C" int stj_outconv_fwd(const void* X, const float* W, const float* bias, float* Y, int F, int Hh, int Ww, int C, int Tn,
                               long long y_bstride, long long y_tstride, long long y_pstride, int dtype, hipStream_t stream) {
  if (Hh % OC_T || Ww % OC_T || C % 8) { stj_set_error("outconv: H,W must be multiples of 16 and C of 8"); return STJ_EINVAL; }
  if (stj_is16(dtype) && ws_enabled() && outconv_fwd_mfma_try(X, W, bias, Y, F, Hh, Ww, C, Tn, y_bstride, y_tstride, y_pstride, dtype, stream))
    return stj_check_launch("stj_outconv_fwd(mfma)");
  const size_t lds = (size_t)(18 * 18 * (C + 1) + 9 * C * 2) * 4;
  if (lds > 160 * 1024) { stj_set_error("outconv: C=%d too large for LDS", C); return STJ_EUNSUPPORTED; }
  const int grid = F * (Hh / OC_T) * (Ww / OC_T);
  if (dtype == STJ_BF16) {
    hipFuncSetAttribute((const void*)outconv_fwd_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(outconv_fwd_kernel<bf16>, dim3(grid), dim3(256), lds, stream, (const bf16*)X, W, bias, Y, Hh, Ww, C, Tn, y_bstride, y_tstride, y_pstride);
  } else if (dtype == STJ_F16) {
    hipFuncSetAttribute((const void*)outconv_fwd_kernel<f16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(outconv_fwd_kernel<f16>, dim3(grid), dim3(256), lds, stream, (const f16*)X, W, bias, Y, Hh, Ww, C, Tn, y_bstride, y_tstride, y_pstride);
  } else {
    hipFuncSetAttribute((const void*)outconv_fwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(outconv_fwd_kernel<float>, dim3(grid), dim3(256), lds, stream, (const float*)X, W, bias, Y, Hh, Ww, C, Tn, y_bstride, y_tstride, y_pstride);
  }
  return stj_check_launch("stj_outconv_fwd");
}
// Both heads of the model output in one launch: Y [B,H,W,4*Tn] f32, channel 4 t + 2 head + o; X0 / X1 the two 48-channel decoder
// branches, frames ordered f = b*Tn + t (t_major = 0) or t*B + b (t_major = 1).  16-bit dtypes, C = 48, Tn = 8 (the hot-path shape);
// STJ_EUNSUPPORTED otherwise: call stj_outconv_fwd once per head instead.
bool outconv_pair_fwd_try(const void* X0, const void* X1, const float* W0, const float* W1, const float* b0, const float* b1, float* Y,
                          int B, int Tn, int Hh, int Ww, int C, int t_major, int dtype, hipStream_t st);
extern "C" int stj_outconv_pair_fwd(const void* X0, const void* X1, const float* W0, const float* W1, const float* bias0,
                                    const float* bias1, float* Y, int B, int Tn, int Hh, int Ww, int C, int t_major, int dtype,
                                    hipStream_t stream) {
  if (B <= 0 || Tn <= 0) { stj_set_error("outconv_pair: empty problem"); return STJ_EINVAL; }
  if (stj_is16(dtype) && ws_enabled() && outconv_pair_fwd_try(X0, X1, W0, W1, bias0, bias1, Y, B, Tn, Hh, Ww, C, t_major, dtype, stream))
    return stj_check_launch("stj_outconv_pair_fwd");
  stj_set_error("outconv_pair: shape / dtype not covered by the paired kernel");
  return STJ_EUNSUPPORTED;
}
// Inference form of the last decoder level + the two output heads (no y48 tensor): stj_upconv_fwd_head writes the per-pixel projection
// Z [F, 2 Hi, 2 Wi, 24] of ELU(up-conv) onto the head's 9 x 2 (tap, output) weights, stj_outconv_pair_gather sums the 9 neighbours of both
// branches into [B, H, W, 32].  16-bit dtypes, the 96 -> 48 level, whole 8 x 16 tiles; STJ_EUNSUPPORTED otherwise (run stj_upconv_fwd +
// stj_outconv_pair_fwd).
bool upconv_fwd_head_try(const void* X, const void* Wf, const float* bias, const float* Wh, void* Z, int F, int Hi, int Wi, int Cin, int Cout, int dtype,
                         hipStream_t st);
bool outconv_pair_gather_try(const void* Z0, const void* Z1, const float* b0, const float* b1, float* Y, int B, int Tn, int Hh, int Ww, int t_major,
                             int dtype, hipStream_t st);
extern "C" int stj_upconv_fwd_head(const void* X, const void* Wf, const float* bias, const float* Whead, void* Z, int F, int Hi, int Wi, int Cin,
                                   int Cout, int dtype, hipStream_t stream) {
  int e = upconv_check(F, Hi, Wi, Cin, Cout, dtype);
  if (e) return e;
  if (stj_is16(dtype) && ws_enabled() && upconv_fwd_head_try(X, Wf, bias, Whead, Z, F, Hi, Wi, Cin, Cout, dtype, stream))
    return stj_check_launch("stj_upconv_fwd_head");
  stj_set_error("upconv_fwd_head: shape / dtype not covered (96 -> 48, whole 8 x 16 tiles, 16-bit)");
  return STJ_EUNSUPPORTED;
}
extern "C" int stj_outconv_pair_gather(const void* Z0, const void* Z1, const float* bias0, const float* bias1, float* Y, int B, int Tn, int Hh,
                                       int Ww, int t_major, int dtype, hipStream_t stream) {
  if (B <= 0 || Tn <= 0) { stj_set_error("outconv_pair_gather: empty problem"); return STJ_EINVAL; }
  if (stj_is16(dtype) && outconv_pair_gather_try(Z0, Z1, bias0, bias1, Y, B, Tn, Hh, Ww, t_major, dtype, stream))
    return stj_check_launch("stj_outconv_pair_gather");
  stj_set_error("outconv_pair_gather: shape / dtype not covered");
  return STJ_EUNSUPPORTED;
}
// ws: caller-owned scratch of stj_outconv_bwd_workspace_bytes() bytes (need not be zeroed; per-block dW/db partials of the bf16
// path live there between its two kernels); without it the slower generic kernel runs.
extern "C" long long stj_outconv_bwd_workspace_bytes() { return outconv_bwd_ws_bytes(); }
// elu_in != 0: X is the output of an ELU (the producing up-conv); dX is then multiplied by ELU'(x) = (x > 0 ? 1 : x + 1), i.e. the
// gradient w.r.t. the producer's PRE-activation is returned and the producer skips its own ELU' pass.
extern "C" int stj_outconv_bwd(const void* X, const float* W, const float* dY, void* dX, float* dW, float* db, int F, int Hh, int Ww,
                               int C, int Tn, long long y_bstride, long long y_tstride, long long y_pstride, int elu_in, void* ws,
                               long long ws_bytes, int dtype, hipStream_t stream) {
  if (Hh % OC_T || Ww % OC_T || C % 8) { stj_set_error("outconv: H,W must be multiples of 16 and C of 8"); return STJ_EINVAL; }
  if (dW == nullptr || db == nullptr) { stj_set_error("stj_outconv_bwd: dW / db must not be NULL"); return STJ_EINVAL; }
  if (dtype == STJ_BF16 && ws_enabled() && outconv_bwd_mfma_try(X, W, dY, dX, dW, db, F, Hh, Ww, C, Tn, y_bstride, y_tstride, y_pstride, elu_in, ws, ws_bytes, stream))
    return stj_check_launch("stj_outconv_bwd(mfma)");
  const size_t lds = (size_t)(18 * 18 * (C + 1) + 9 * C * 2 + 18 * 18 * 2) * 4;
  if (lds > 160 * 1024 || 9 * C > 1024) { stj_set_error("outconv: C=%d too large", C); return STJ_EUNSUPPORTED; }
  const int grid = min(1024, F * (Hh / OC_T) * (Ww / OC_T));
  if (dtype == STJ_BF16) {
    hipFuncSetAttribute((const void*)outconv_bwd_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(outconv_bwd_kernel<bf16>, dim3(grid), dim3(256), lds, stream, (const bf16*)X, W, dY, (bf16*)dX, dW, db, F, Hh, Ww, C, Tn, y_bstride, y_tstride, y_pstride, elu_in);
  } else if (dtype == STJ_F16) {
    hipFuncSetAttribute((const void*)outconv_bwd_kernel<f16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(outconv_bwd_kernel<f16>, dim3(grid), dim3(256), lds, stream, (const f16*)X, W, dY, (f16*)dX, dW, db, F, Hh, Ww, C, Tn, y_bstride, y_tstride, y_pstride, elu_in);
  } else {
    hipFuncSetAttribute((const void*)outconv_bwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(outconv_bwd_kernel<float>, dim3(grid), dim3(256), lds, stream, (const float*)X, W, dY, (float*)dX, dW, db, F, Hh, Ww, C, Tn, y_bstride, y_tstride, y_pstride, elu_in);
  }
  return stj_check_launch("stj_outconv_bwd");
}

// ---------------------------------------------------------------------------------------------------
// im2col helpers (tiny tensors)
// ---------------------------------------------------------------------------------------------------
// patch embed: src f32 [B,H,W,*] read as src[((b*H+y)*W+x)*pix_stride + c*ch_stride], dst T [B*(H/4)*(W/4), 16*Cin]
// with k = (dy*4+dx)*Cin + c  (matches Keras kernel [4,4,Cin,Cout] flattened).  Also performs the f32->T cast and the
// stride-2 pick of the vehicle channel ogm[...,0] (reference modules.py:572).
template <typename T>
__global__ __launch_bounds__(256) void im2col_patch_kernel(const float* src, T* dst, int B, int H, int W, int Cin, long long pix_stride, int ch_stride) {
  const int K = 16 * Cin, Ph = H / 4, Pw = W / 4;
  const long long total = (long long)B * Ph * Pw * K;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) {
    const int k = (int)(i % K); long long m = i / K;
    const int c = k % Cin, d = k / Cin, dy = d >> 2, dx = d & 3;
    const int pj = (int)(m % Pw); long long t = m / Pw;
    const int pi = (int)(t % Ph); const long long b = t / Ph;
    stf(dst + i, src[((b * H + 4 * pi + dy) * W + 4 * pj + dx) * pix_stride + (long long)c * ch_stride]);
  }
}
extern "C" int stj_im2col_patch(const float* src, void* dst, int B, int H, int W, int Cin, long long pix_stride, int ch_stride, int dtype, hipStream_t stream) {
  const long long total = (long long)B * (H / 4) * (W / 4) * 16 * Cin;
  if (total <= 0) return STJ_OK;
  const int g = (int)min(8192ll, (total + 255) / 256);
  if (dtype == STJ_BF16) hipLaunchKernelGGL(im2col_patch_kernel<bf16>, dim3(g), dim3(256), 0, stream, src, (bf16*)dst, B, H, W, Cin, pix_stride, ch_stride);
  else if (dtype == STJ_F16) hipLaunchKernelGGL(im2col_patch_kernel<f16>, dim3(g), dim3(256), 0, stream, src, (f16*)dst, B, H, W, Cin, pix_stride, ch_stride);
  else hipLaunchKernelGGL(im2col_patch_kernel<float>, dim3(g), dim3(256), 0, stream, src, (float*)dst, B, H, W, Cin, pix_stride, ch_stride);
  return stj_check_launch("stj_im2col_patch");
}

// grouped 3x3 SAME: x [N,H,W,G*Cg] -> cols [N*H*W, G, 9*Cg] with k = (dy*3+dx)*Cg + c ; col2im is the adjoint.
template <typename T>
__global__ __launch_bounds__(256) void im2col3_kernel(const T* x, T* cols, int N, int H, int W, int G, int Cg) {
  const int K = 9 * Cg; const int C = G * Cg;
  const long long total = (long long)N * H * W * G * K;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) {
    const int k = (int)(i % K); long long t = i / K;
    const int g = (int)(t % G); t /= G;
    const int xw = (int)(t % W); t /= W;
    const int yh = (int)(t % H); const long long n = t / H;
    const int c = k % Cg, d = k / Cg;
    const int sy = yh + d / 3 - 1, sx = xw + d % 3 - 1;
    T z; if constexpr (sizeof(T) == 2) z.v = 0; else z = 0.f;
    cols[i] = (sy >= 0 && sy < H && sx >= 0 && sx < W) ? x[((n * H + sy) * W + sx) * C + g * Cg + c] : z;
  }
}
// the same, 16 bytes (VN channels of one tap) per thread: Cg % VN == 0, 16-byte aligned tensors.  (The element-wise kernel above spends its
// time on 64-bit index arithmetic per 2-byte element: 28 us for the 14 MB of FG-MSA's offset conv at B = 8, on the forward critical chain.)
template <typename T>
__global__ __launch_bounds__(256) void im2col3_vec_kernel(const T* x, T* cols, int N, int H, int W, int G, int Cg) {
  constexpr int VN = Vec<T>::N;
  const int KV = 9 * Cg / VN, CgV = Cg / VN; const int C = G * Cg;
  const int total = N * H * W * G * KV;                       // (launcher: fits 31 bits)
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int kv = i % KV; int t = i / KV;
    const int g = t % G; t /= G;
    const int xw = t % W; t /= W;
    const int yh = t % H; const int n = t / H;
    const int cv = kv % CgV, d = kv / CgV;
    const int sy = yh + d / 3 - 1, sx = xw + d % 3 - 1;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (sy >= 0 && sy < H && sx >= 0 && sx < W) v = *reinterpret_cast<const uint4*>(x + ((long long)(n * H + sy) * W + sx) * C + g * Cg + cv * VN);
    *reinterpret_cast<uint4*>(cols + (long long)i * VN) = v;
  }
}
template <typename T>
__global__ __launch_bounds__(256) void col2im3_vec_kernel(const T* dcols, T* dx, int N, int H, int W, int G, int Cg) {
  constexpr int VN = Vec<T>::N;
  const int K = 9 * Cg; const int C = G * Cg, CV = C / VN;
  const int total = N * H * W * CV;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int ch = (i % CV) * VN; int t = i / CV;
    const int xw = t % W; t /= W;
    const int yh = t % H; const int n = t / H;
    const int g = ch / Cg, c = ch % Cg;
    float acc[VN];
#pragma unroll
    for (int e = 0; e < VN; ++e) acc[e] = 0.f;
#pragma unroll
    for (int d = 0; d < 9; ++d) {
      const int oy = yh - (d / 3 - 1), ox = xw - (d % 3 - 1);      // output pixel whose tap d reads (yh,xw)
      if (oy >= 0 && oy < H && ox >= 0 && ox < W) {
        float v[VN];
        ld16(dcols + ((((long long)(n * H + oy) * W + ox) * G + g) * K) + d * Cg + c, v);
#pragma unroll
        for (int e = 0; e < VN; ++e) acc[e] += v[e];
      }
    }
    st16(dx + (long long)i * VN, acc);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void col2im3_kernel(const T* dcols, T* dx, int N, int H, int W, int G, int Cg) {
  const int K = 9 * Cg; const int C = G * Cg;
  const long long total = (long long)N * H * W * C;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) {
    const int ch = (int)(i % C); long long t = i / C;
    const int xw = (int)(t % W); t /= W;
    const int yh = (int)(t % H); const long long n = t / H;
    const int g = ch / Cg, c = ch % Cg;
    float acc = 0.f;
#pragma unroll
    for (int d = 0; d < 9; ++d) {
      const int oy = yh - (d / 3 - 1), ox = xw - (d % 3 - 1);      // output pixel whose tap d reads (yh,xw)
      if (oy >= 0 && oy < H && ox >= 0 && ox < W) acc += ldf(dcols + ((((n * H + oy) * W + ox) * G + g) * (long long)K) + d * Cg + c);
    }
    stf(dx + i, acc);
  }
}
extern "C" int stj_im2col3(const void* x, void* cols, int N, int H, int W, int G, int Cg, int dtype, hipStream_t stream) {
  const long long total = (long long)N * H * W * G * 9 * Cg;
  if (total <= 0) return STJ_OK;
  const int vn = dtype == STJ_F32 ? 4 : 8;
  if (Cg % vn == 0 && total < (1ll << 31) && !((((uintptr_t)x) | ((uintptr_t)cols)) & 15)) {
    const int gv = (int)min(8192ll, (total / vn + 255) / 256);
    if (dtype == STJ_BF16) hipLaunchKernelGGL(im2col3_vec_kernel<bf16>, dim3(gv), dim3(256), 0, stream, (const bf16*)x, (bf16*)cols, N, H, W, G, Cg);
    else if (dtype == STJ_F16) hipLaunchKernelGGL(im2col3_vec_kernel<f16>, dim3(gv), dim3(256), 0, stream, (const f16*)x, (f16*)cols, N, H, W, G, Cg);
    else hipLaunchKernelGGL(im2col3_vec_kernel<float>, dim3(gv), dim3(256), 0, stream, (const float*)x, (float*)cols, N, H, W, G, Cg);
    return stj_check_launch("stj_im2col3");
  }
  const int g = (int)min(8192ll, (total + 255) / 256);
  if (dtype == STJ_BF16) hipLaunchKernelGGL(im2col3_kernel<bf16>, dim3(g), dim3(256), 0, stream, (const bf16*)x, (bf16*)cols, N, H, W, G, Cg);
  else if (dtype == STJ_F16) hipLaunchKernelGGL(im2col3_kernel<f16>, dim3(g), dim3(256), 0, stream, (const f16*)x, (f16*)cols, N, H, W, G, Cg);
  else hipLaunchKernelGGL(im2col3_kernel<float>, dim3(g), dim3(256), 0, stream, (const float*)x, (float*)cols, N, H, W, G, Cg);
  return stj_check_launch("stj_im2col3");
}
extern "C" int stj_col2im3(const void* dcols, void* dx, int N, int H, int W, int G, int Cg, int dtype, hipStream_t stream) {
  const long long total = (long long)N * H * W * G * Cg;
  if (total <= 0) return STJ_OK;
  const int vn = dtype == STJ_F32 ? 4 : 8;
  if (Cg % vn == 0 && total * 9 < (1ll << 31) && !((((uintptr_t)dcols) | ((uintptr_t)dx)) & 15)) {
    const int gv = (int)min(8192ll, (total / vn + 255) / 256);
    if (dtype == STJ_BF16) hipLaunchKernelGGL(col2im3_vec_kernel<bf16>, dim3(gv), dim3(256), 0, stream, (const bf16*)dcols, (bf16*)dx, N, H, W, G, Cg);
    else if (dtype == STJ_F16) hipLaunchKernelGGL(col2im3_vec_kernel<f16>, dim3(gv), dim3(256), 0, stream, (const f16*)dcols, (f16*)dx, N, H, W, G, Cg);
    else hipLaunchKernelGGL(col2im3_vec_kernel<float>, dim3(gv), dim3(256), 0, stream, (const float*)dcols, (float*)dx, N, H, W, G, Cg);
    return stj_check_launch("stj_col2im3");
  }
  const int g = (int)min(8192ll, (total + 255) / 256);
  if (dtype == STJ_BF16) hipLaunchKernelGGL(col2im3_kernel<bf16>, dim3(g), dim3(256), 0, stream, (const bf16*)dcols, (bf16*)dx, N, H, W, G, Cg);
  else if (dtype == STJ_F16) hipLaunchKernelGGL(col2im3_kernel<f16>, dim3(g), dim3(256), 0, stream, (const f16*)dcols, (f16*)dx, N, H, W, G, Cg);
  else hipLaunchKernelGGL(col2im3_kernel<float>, dim3(g), dim3(256), 0, stream, (const float*)dcols, (float*)dx, N, H, W, G, Cg);
  return stj_check_launch("stj_col2im3");
}
