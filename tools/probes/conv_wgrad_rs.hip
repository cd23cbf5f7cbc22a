// EXPERIMENT (round 4), NOT part of the library: a row-streaming LDS-DMA form of the up-conv weight gradient.  Correct (it passed
// tests/test_timed_kernels_gpu.py::test_upconv_ws_bench_shapes_bf16 and tools/probes/wrs_check.py when wired into stj_upconv_wgrad), but
// SLOWER than the chunked kernel it was meant to replace: 277 us vs 208 us on the 96 -> 48 layer at F = 64, 128 x 128 (of which ~44 us is
// the flush: 16 tap matrices per workgroup instead of 8).  What was measured on the way: units in flight 3 -> 5:
// no change; unpadded LDS rows (no padding lanes in the DMA stream, bank conflicts back): no change; a per-workgroup start row against
// HBM channel lockstep: no change; padding chunks fetched from the neighbouring data chunk instead of a zero page: 6x slower; the
// bias-gradient dots behind every fragment read instead of behind the MFMAs: +20 us.  The stream is bound by the per-unit dependent chain
// (counted wait -> barrier -> 36 transpose reads -> 36 MFMAs per wave at 2 waves per SIMD), not by HBM or the DMA depth.
// Row-streaming weight gradient of the decoder's up-convs (nearest-2x upsample folded into 2x2 taps), 16-bit activations.
//   dWeff[a,b,r,s][co][ci] = sum_{f,i,j} dP[f, 2i+a, 2j+b, co] * X[f, i+a-1+r, j+b-1+s, ci]        (16 tap matrices [Cout][Cin])
// (reference: tape.gradient (train.py:223) of Conv2D(3x3) o UpSampling2D(2), modules.py:746-748; the fold to 2x2 taps: SURVEY App. C)
//
// The chunked kernels (conv_ws.hip upconv_wgrad_tr / tr4) stage a chunk in registers, write it to LDS, meet at a barrier and only
// then feed the MFMAs: the phases of a chunk add up (HBM 0.44, MFMA 0.31 of peak on the 96 -> 48 layer, the step's dominant launch),
// every chunk re-reads an X halo row, and the two row parities of a strip are two workgroups reading X twice.
// Here the image is STREAMED ROW BY ROW through LDS rings by LDS-DMA (global_load_lds_dwordx4, counted vmcnt waits, one raw barrier per
// row -- the scheme of wgrad_sk.hip):
//   * a workgroup owns a strip (frame, 32 low-res columns, RS rows) and both row parities a: unit n of the strip brings X row n
//     (34 pixels with the column halo) and the two hi-res dP rows 2(n-1), 2(n-1)+1; low-res row t is computed from X rows t-1, t, t+1
//     (still in the ring: no vertical halo re-read) and dP(t);
//   * 8 waves = (column parity b, tap r, tap s), each with both a: 2 x FO x FI accumulator fragments per wave;
//   * both operands keep their [pixel][channel] layout and become MFMA fragments by ds_read_b64_tr_b16 with the k-permutation
//     (pixels 4g.., 16+4g.. per 16-lane group) and 8 * odd-dword pixel strides: conflict-free;
//   * image borders: out-of-range rows / columns are fetched from a zero page, so the inner loop has no border case;
//   * D = 5 units (110 KB) in flight per CU; workgroups are persistent over their strips and flush their accumulators ONCE (f32 atomics
//     into dWeff; the bias gradient as packed dot products with ones of the dP fragments).
#include "common.h"

namespace wrs {

constexpr int CB = 32;                 // low-res columns per strip
constexpr int XPX = CB + 2;            // X pixels per row image (column halo)
constexpr int D = 5;                   // units in flight
constexpr int NX = D + 3, ND = D + 1;  // ring slots

__device__ uint4 zero_page[4];         // 64 zero bytes: the source of everything outside the image / in the LDS row padding

template <int FO, int FI>
struct Geo {
  static constexpr int BO = FO * 16, BI = FI * 16;
  static constexpr int LDO = BO + 8;                                  // consecutive low-res pixels lie 2 hi-res pixels = LDO dwords apart
  static constexpr int LDI = (BI / 16) % 2 ? BI : BI + 16;            // consecutive pixels lie LDI / 2 dwords apart
  static_assert((LDO / 8) % 2 == 1 && (LDI / 16) % 2 == 1 && LDO % 8 == 0, "conflict-free transpose-read strides");
  static constexpr int XCH = LDI / 8, DCH = LDO / 8;                  // 16-byte chunks per pixel (incl. padding)
  static constexpr int X_CHUNKS = XPX * XCH, D_CHUNKS = 2 * 2 * CB * DCH;
  static constexpr int X_INS = (X_CHUNKS + 63) / 64, D_INS = (D_CHUNKS + 63) / 64;      // DMA instructions per unit
  static constexpr int X_SLOT = X_INS * 1024, D_SLOT = D_INS * 1024;                    // bytes
  static constexpr int INS = X_INS + D_INS;
  static constexpr int SLOTS = (INS + 7) / 8;                          // per wave (waves >= INS - 8 (SLOTS - 1) issue one fewer)
  static constexpr int LDS = NX * X_SLOT + ND * D_SLOT;
};

struct Args {
  const void* X; const void* dP; float* dWeff; float* dbias;
  int db_parts, F, Hi, Wi, Cin, Cout, ncols, cin_tiles, ntiles;
};

__device__ __forceinline__ void glds16(const char* g, uint32_t lds_byte) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(lds_byte) : "memory");
}
typedef __attribute__((ext_vector_type(4))) short s4;
__device__ __forceinline__ s4 read_tr4(uint32_t b) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(uintptr_t)b);
}
__device__ __forceinline__ s16x8 join(const s4& lo, const s4& hi) { return (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}; }
// sum of the 4 bf16 of one transpose read: two packed dot products with ones.  (As asm statements: through __builtin_amdgcn_fdot2_f32_bf16
// this compiler emitted the dot product of dword 0 for every dword of the fragment -- visible in the ISA, caught by the column-indexed
// bias-gradient probe in tools/probes/wrs_check.py.)
__device__ __forceinline__ float half_sum_bf16(float c, const s4& h) {
  typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
  const u32x2 w = __builtin_bit_cast(u32x2, h);
  const uint32_t ones = 0x3f803f80u;
  // (the trailing s_nop: a DOT result needs 3 wait states before a different VALU opcode may read it, and hipcc pads nothing it cannot see
  //  inside an asm statement -- without it the last product was lost)
  asm("v_dot2c_f32_bf16 %0, %1, %2\n\tv_dot2c_f32_bf16 %0, %1, %3\n\ts_nop 3" : "+v"(c) : "v"(ones), "v"(w.x), "v"(w.y));
  return c;
}

// s_waitcnt vmcnt(n * PER): at most the DMA instructions of n younger units of this wave outstanding
template <int PER>
__device__ __forceinline__ void wait_units(int n) {
  static_assert(D <= 5 && 4 * PER < 64, "vmcnt immediates");
  if (n >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * PER) : "memory");
  else if (n == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PER) : "memory");
  else if (n == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
  else if (n == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int FO, int FI>
__global__ __launch_bounds__(512, 2) void upconv_wgrad_rs_kernel(Args p) {
  typedef Geo<FO, FI> G;
  extern __shared__ __attribute__((aligned(1024))) unsigned char ring[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = w >> 2, r = (w >> 1) & 1, s = w & 1;
  const int g = lane >> 4, pl = lane & 15;
  const uint32_t ring0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)ring);
  const uint32_t xring = ring0, dring = ring0 + NX * G::X_SLOT;
  // the X ring starts zeroed: the first two units of the workgroup multiply (all-zero) dP rows with slots nothing has been written to yet
  for (int i = tid; i < NX * G::X_SLOT / 16; i += 512) reinterpret_cast<uint4*>(ring)[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  const char* zp = reinterpret_cast<const char*>(zero_page);
  const int Hi = p.Hi, Wi = p.Wi, Cin = p.Cin, Cout = p.Cout;
  const int Ho = 2 * Hi, Wo = 2 * Wi;
  const int ncb = Wi / CB;

  // Work: "columns" (channel tile, frame, 32-column block), each Hi rows; a workgroup takes columns [c0, c1) and walks each as TWO strips,
  // rows [rot, Hi) then [0, rot), with a workgroup-specific rot: all columns start on addresses that are equal modulo any power-of-two
  // channel interleave (a frame is a multiple of 1 MB), so workgroups that all began at row 0 would sweep the same few HBM channels in
  // lockstep for the whole launch.
  const int c0 = (int)((long long)p.ncols * blockIdx.x / gridDim.x), c1 = (int)((long long)p.ncols * (blockIdx.x + 1) / gridDim.x);
  if (c0 >= c1) return;
  const int rot = 1 + (int)((blockIdx.x * 2654435761u >> 7) % (unsigned)(Hi - 1));
  const int nst = 2 * (c1 - c0);                            // strips
  const int Q = (c1 - c0) * (Hi + 4);                       // units: every strip brings its rows plus one X row above and below

  // ---- this lane's DMA chunks (geometry independent of the unit): slot q of wave w is instruction i = w + 8 q
  //   i < X_INS: X-row chunk c = 64 i + lane -> pixel c / XCH (column c0 - 1 + pixel), 16-byte chunk c % XCH (>= BI / 8: padding)
  //   else     : dP chunk c = 64 (i - X_INS) + lane -> hi-res row c / (2 CB DCH), pixel, chunk (>= BO / 8: padding)
  int voff[G::SLOTS];           // byte offset from the unit's base address (negative for the left halo pixel)
  int xpx[G::SLOTS];            // X slots: pixel index (for the column-border test); -1 otherwise
  unsigned pad = 0;             // bit q: slot q of this lane is LDS row padding / beyond the image
#pragma unroll
  for (int q = 0; q < G::SLOTS; ++q) {
    const int i = w + 8 * q;
    voff[q] = 0; xpx[q] = -1;
    if (i < G::X_INS) {
      const int c = 64 * i + lane, px = c / G::XCH, cc = c % G::XCH;
      // (padding chunks come from the zero page.  Re-reading the pixel's last data chunk instead -- duplicate addresses next to real ones
      //  inside one DMA instruction -- measured 6x SLOWER: 1416 vs 233 us for the whole stream.)
      if (c < G::X_CHUNKS && cc < G::BI / 8) { voff[q] = ((px - 1) * Cin + cc * 8) * 2; xpx[q] = px; }
      else pad |= 1u << q;
    } else if (i < G::INS) {
      const int c = 64 * (i - G::X_INS) + lane, hr = c / (2 * CB * G::DCH), rem = c % (2 * CB * G::DCH), px = rem / G::DCH, cc = rem % G::DCH;
      if (c < G::D_CHUNKS && cc < G::BO / 8) voff[q] = ((hr * Wo + px) * Cout + cc * 8) * 2;
      else pad |= 1u << q;
    }
  }

  // ---- strip decode (scalar)
  struct Strip { const char* xb; const char* db; int i0, n, c0, co0, ci0, tile; };
  auto decode = [&](int k) {           // strip k of this workgroup
    Strip st;
    int t = c0 + (k >> 1);
    const int cb = t % ncb; t /= ncb;
    const int f = t % p.F; const int tile = t / p.F;
    st.tile = tile;
    st.co0 = (tile / p.cin_tiles) * G::BO; st.ci0 = (tile % p.cin_tiles) * G::BI;
    st.i0 = (k & 1) ? 0 : rot; st.n = (k & 1) ? rot : Hi - rot;
    st.c0 = cb * CB;
    st.xb = reinterpret_cast<const char*>(p.X) + (((long long)f * Hi * Wi + st.c0) * Cin + st.ci0) * 2;            // + row * Wi * Cin * 2
    st.db = reinterpret_cast<const char*>(p.dP) + (((long long)f * Ho * Wo + 2 * st.c0) * Cout + st.co0) * 2;       // + hi-res row * Wo * Cout * 2
    return st;
  };

  // ---- loader: unit lq (0 .. Q) -> X row i0 + t, dP rows 2 (i0 + t - 1) .., t = lq % UPS - 1
  int lq = 0;
  int lk = 0;
  Strip L = decode(0);
  int lt = -1;                              // t of unit lq
  unsigned colmask = 0;                     // per slot: bit q set = this lane's X chunk lies outside the image columns for strip L
  auto set_colmask = [&]() {
    colmask = 0;
#pragma unroll
    for (int q = 0; q < G::SLOTS; ++q)
      if (xpx[q] >= 0) { const int col = L.c0 - 1 + xpx[q]; if (col < 0 || col >= Wi) colmask |= 1u << q; }
  };
  set_colmask();
  auto issue = [&]() {
    const int row = L.i0 + lt;                                          // X row of this unit
    const bool xrow_ok = row >= 0 && row < Hi;
    const bool drow_ok = lt >= 1;                                       // dP of low-res row i0 + lt - 1 (units -1 and 0 carry none)
    const char* xb = L.xb + (long long)row * Wi * Cin * 2;
    const char* db = L.db + (long long)(2 * (L.i0 + lt - 1)) * Wo * Cout * 2;
    const uint32_t xdst = xring + (uint32_t)(lq % NX) * G::X_SLOT, ddst = dring + (uint32_t)(lq % ND) * G::D_SLOT;
#pragma unroll
    for (int q = 0; q < G::SLOTS; ++q) {
      const int i = w + 8 * q;
      if (i >= G::INS) continue;                                        // (wave-uniform: the last slot exists for the low waves only)
      const bool isx = i < G::X_INS;
      const bool ok = !((pad >> q) & 1) && (isx ? (xrow_ok && !((colmask >> q) & 1)) : drow_ok);
      const char* src = ok ? (isx ? xb : db) + voff[q] : zp;
      glds16(src, __builtin_amdgcn_readfirstlane(isx ? xdst + i * 1024 : ddst + (i - G::X_INS) * 1024));
    }
    ++lq; ++lt;
    if (lt > L.n && lq < Q) { L = decode(++lk); lt = -1; set_colmask(); }
  };

  // ---- consumer
  f32x4 acc[2][FO][FI];
  float csum[2][FO];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int m = 0; m < FO; ++m) {
      csum[a][m] = 0.f;
#pragma unroll
      for (int n = 0; n < FI; ++n) acc[a][m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  int ck = 0;
  Strip S = decode(0);
  const bool db_wave = p.dbias != nullptr && r == 0 && s == 0;       // waves 0 and 4: each hi-res dP pixel once
  // per-lane fragment addresses inside a slot: low-res pixel j = 4 g + pl / 4 (+ 16), channels 4 (pl % 4) .. of a 16-channel block
  const int j = 4 * g + (pl >> 2), kch = 4 * (pl & 3);
  const uint32_t aD = ((2 * j + b) * G::LDO + kch) * 2;               // + a * (2 CB) * LDO * 2 (hi-res row a); + 16 pixels: + 32 LDO * 2
  const uint32_t aX = ((j + b + s) * G::LDI + kch) * 2;               // slot pixel 0 = column c0 - 1; + 16 pixels: + 16 LDI * 2
  constexpr int per_hi = G::SLOTS, per_lo = G::SLOTS - 1;             // DMA instructions per unit of a wave
  const bool hi_wave = w + 8 * (G::SLOTS - 1) < G::INS;

  auto flush = [&](const Strip& st) {
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int pt = a * 8 + b * 4 + r * 2 + s;
      float* Wt = p.dWeff + ((long long)pt * Cout + st.co0) * Cin + st.ci0;
#pragma unroll
      for (int m = 0; m < FO; ++m)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int co = m * 16 + 4 * g + rg;
#pragma unroll
          for (int n = 0; n < FI; ++n) {
            atomicAdd(Wt + (long long)co * Cin + n * 16 + pl, acc[a][m][n][rg]);
            acc[a][m][n][rg] = 0.f;
          }
        }
      if (db_wave && st.ci0 == 0) {
        float* dbp = p.dbias + (long long)((blockIdx.x * 2 + b) % p.db_parts) * Cout + st.co0;
#pragma unroll
        for (int m = 0; m < FO; ++m) {
          float v = csum[a][m];
          v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
          if (g == 0) atomicAdd(dbp + m * 16 + pl, v);
        }
      }
#pragma unroll
      for (int m = 0; m < FO; ++m) csum[a][m] = 0.f;
    }
  };

#pragma unroll 1
  for (int q = 0; q < D; ++q)
    if (lq < Q) issue();

  int ct = -1;                              // t of unit c
#pragma unroll 1
  for (int c = 0; c < Q; ++c) {
    // unit c landed (this wave's share): at most the younger units' instructions may be outstanding
    const int ahead = lq - c - 1;
    if (hi_wave) wait_units<per_hi>(ahead); else wait_units<per_lo>(ahead);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();           // everybody's share of unit c landed; everybody is through with unit c - 1's compute
    asm volatile("" ::: "memory");
    if (lq < Q) issue();                    // unit c + D into the slots of units c - 3 (X) / c - 1 (dP)

    {                                       // low-res row t = ct - 1 of the strip: X units c - 2, c - 1, c; dP of unit c.
      // Units -1 and 0 of a strip (ct < 1) complete no row: their dP slot was filled from the zero page, so their MFMAs add nothing --
      // cheaper than branching around them (accumulators updated under a branch cost the compiler a second copy: 170 registers of scratch).
      const uint32_t dsl = dring + (uint32_t)(c % ND) * G::D_SLOT;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        // X row t + a - 1 + r = unit c - 2 + a + r
        const uint32_t xsl = xring + (uint32_t)((c + NX - 2 + a + r) % NX) * G::X_SLOT + aX;
        const uint32_t dsa = dsl + aD + a * (2 * CB * G::LDO * 2);
        s16x8 af[FO];
        s4 alo[FO], ahi[FO];
#pragma unroll
        for (int m = 0; m < FO; ++m) {
          alo[m] = read_tr4(dsa + m * 32); ahi[m] = read_tr4(dsa + m * 32 + 32 * G::LDO * 2);
          af[m] = join(alo[m], ahi[m]);
        }
#pragma unroll
        for (int n = 0; n < FI; ++n) {
          const s16x8 bf = join(read_tr4(xsl + n * 32), read_tr4(xsl + n * 32 + 16 * G::LDI * 2));
#pragma unroll
          for (int m = 0; m < FO; ++m)
            acc[a][m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[m]), __builtin_bit_cast(bf16x8_t, bf), acc[a][m][n], 0, 0, 0);
        }
        if (db_wave) {        // (after the MFMAs: behind each fragment read it serialised the reads' latencies)
#pragma unroll
          for (int m = 0; m < FO; ++m) csum[a][m] = half_sum_bf16(half_sum_bf16(csum[a][m], alo[m]), ahi[m]);
        }
        __builtin_amdgcn_sched_barrier(0);        // (keeps the second row parity's fragment reads behind this one's MFMAs: 256 registers)
      }
    }
    ++ct;
    if (ct > S.n) {                         // strip done
      ct = -1;
      if (c + 1 < Q) {
        const Strip nx = decode(++ck);
        if (nx.tile != S.tile) {            // leaving a channel tile: flush (drain the DMA queue first, as in wgrad_sk.hip)
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          flush(S);
        }
        S = nx;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  flush(S);
}

}  // namespace wrs

template <int FO, int FI>
static bool wgrad_rs_launch(const void* X, const void* dP, float* dWeff, float* dbias, int db_parts, int F, int Hi, int Wi, int Cin, int Cout,
                            int wg_budget, hipStream_t st) {
  typedef wrs::Geo<FO, FI> G;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)wrs::upconv_wgrad_rs_kernel<FO, FI>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS) != hipSuccess) return false;
    attr_set = true;
  }
  wrs::Args a;
  a.X = X; a.dP = dP; a.dWeff = dWeff; a.dbias = dbias; a.db_parts = db_parts; a.F = F; a.Hi = Hi; a.Wi = Wi; a.Cin = Cin; a.Cout = Cout;
  a.cin_tiles = Cin / G::BI; a.ntiles = a.cin_tiles * (Cout / G::BO);
  a.ncols = a.ntiles * F * (Wi / wrs::CB);
  int grid = wg_budget > 0 ? wg_budget : 256;
  if (grid > a.ncols) grid = a.ncols;
  hipLaunchKernelGGL((wrs::upconv_wgrad_rs_kernel<FO, FI>), dim3(grid), dim3(512), G::LDS, st, a);
  return true;
}

// true when handled: bf16, Wi % 32 == 0, Hi % 16 == 0, channels an exact multiple of the tile
bool upconv_wgrad_rs_try(const void* X, const void* dP, float* dWeff, float* dbias, int db_parts, int F, int Hi, int Wi, int Cin, int Cout,
                         int wg_budget, hipStream_t st) {
  if (Wi % wrs::CB || Hi % 16 || (long long)F * Hi * Wi < 65536) return false;
  if (Cout == 48 && Cin == 96) return wgrad_rs_launch<3, 6>(X, dP, dWeff, dbias, db_parts, F, Hi, Wi, Cin, Cout, wg_budget, st);
  return false;
}
