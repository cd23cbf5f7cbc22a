// Grouped stream-K weight gradients:  dW[z] += X[z]^T dY[z]  (+ db[z] += 1^T dY[z])  for a LIST of Dense layers in ONE launch.
// (reference: tape.gradient (train.py:223) of every Keras Dense / 1x1 conv / tfa-MHA projection of the hot path --
//  modules.py:36-37,76-83,270-272, trajNet.py:71-77,195-211, FG_MSA.py:54-64)
//
// Why not the tile GEMM (gemm.hip): a weight gradient contracts over the ROWS (2048 .. 32768 tokens) into a small [Cin, Cout]
// result.  As 64 x 64 output tiles x split-K it re-stages every k-slab of X and dY once per tile it meets (4-13 x from L2) and ends
// every one of its ~768 workgroups in 4096 f32 atomics: 3.1 M atomics per launch, ~10 us of the chip's atomic throughput
// (measured 315 G atomics/s whatever the contention, tools/probes/atomic_probe.hip) under launches whose operands stream in 5 us.
//
// Here a workgroup owns a FULL-WIDTH tile: a 96-column slice of one operand (the "narrow" one) x a 384-column slice of the
// other (the "wide" one) -- for the Swin widths (96 -> 288 / 96 / 384, 384 -> 96) that is the whole problem, so every row of X and
// dY is read from HBM exactly once.  4 waves as 2 x 2, a wave owns 48 x 192 outputs = 36 accumulator fragments (144 registers).
// The rows are streamed in 32-row slabs (one MFMA k-step) through a 4-deep LDS ring filled by LDS-DMA (global_load_lds_dwordx4,
// no staging registers, three slabs = 96 KB per CU in flight behind counted vmcnt waits and one raw s_barrier per slab).
// Both operands sit in LDS un-transposed ([row][column], as in memory) and become MFMA fragments through the LDS transpose read
// (ds_read_b64_tr_b16); the image row strides (112 / 400 elements = 56 / 200 dwords = 8 * odd) put the 8 k-rows a 32-lane group
// reads in one pass on 8 disjoint 8-bank groups: conflict-free.
//
// Work = units of (job, batch z, tile, 32-row slab), linearised; the launch's G workgroups take EQUAL CONTIGUOUS RANGES of units
// (stream-K): a workgroup keeps its accumulators while it stays inside one tile and flushes them with f32 atomics when it leaves
// it.  A problem is therefore touched by about (its share of the units) x G workgroups: ~G + tiles flushes per launch in total
// instead of 768 per problem, and any mix of shapes balances.  The bias gradient rides along as packed dot products of the dY
// fragments with ones (v_dot2_f32_bf16 / _f16), kept per lane and reduced at the flush.
#include "common.h"
#include <stdlib.h>

namespace wsk {

constexpr int TW_ = 384;                       // tile: wide columns
constexpr int KU = 32;                         // rows per unit (one 16x16x32 MFMA k-step)
constexpr int LDW = TW_ + 16;                  // LDS row stride of the wide image (elements): 200 dwords = 8 * odd
constexpr int CHW = LDW / 8;                   // 16-byte chunks per wide image row: 50
constexpr int MAXJ = 28;                       // jobs per launch (descriptor lives in the kernarg segment: 4 KB)
// Two tile geometries (round 5).  TN = 96: 4 waves as 2 x 2 (the Swin widths of the 64 x 64 stage are whole problems).  TN = 192: 8 waves
// as 4 x 2 on the same 48 x 192 wave tile, for launches whose problems are all at least 192 wide on both sides (the C = 192 / 384 stages,
// cfg-512's 8192-row C = 384 stage): a 32-row slab of 192 + 384 columns feeds twice the MFMAs of one of 96 + 384 -- 128 instead of 77
// FLOP per byte of the L2 -> LDS stream those launches are bound by (one launch of cfg-512's 24 wide jobs: 2.26 GB streamed in 613 us).
template <int TN> struct Geo {
  static constexpr int TN_ = TN;
  static constexpr int NWV = TN == 96 ? 4 : 8;               // waves per workgroup
  static constexpr int LDN = TN == 96 ? 112 : 240;           // LDS row stride of the narrow image (elements): 56 / 120 dwords = 8 * odd
  static constexpr int CHN = LDN / 8;                        // 16-byte chunks per narrow image row (the pad chunks re-read a valid one)
  static constexpr int IMG_N = KU * LDN * 2;                 // 7168 / 15360 bytes = 7 / 15 x 1 KB DMA instructions
  static constexpr int STAGE = KU * (LDN + LDW) * 2;         // 32768 / 40960 bytes = 32 / 40 DMA instructions: 8 / 5 per wave
  static constexpr int NS = TN == 96 ? 4 : 3;                // ring depth (96 / 80 KB in flight)
  static constexpr int SLOTS = STAGE / 1024 / NWV;           // DMA instructions per wave and stage
  static_assert(IMG_N % 1024 == 0 && STAGE % (1024 * NWV) == 0 && LDN % 16 == 0 && (LDN / 16) % 2 == 1 && LDN >= TN + 8, "whole DMA instructions, conflict-free stride");
};

struct Job {
  const void* nar; const void* wid; float* C; float* cs;      // narrow / wide operand, dW, db (or null)
  long long nb1s, nb2s, wb1s, wb2s, cb1s, cb2s, sb1s, sb2s;   // batch strides (elements): narrow, wide, C, cs
  int ld_nar, ld_wid, ldc;                                     // row strides (elements)
  int n_nar, n_wid;                                            // widths
  int t_nar, t_wid;                                            // tiles
  int ku;                                                      // units per tile = rows / 32
  int nb2;                                                     // z = z1 * nb2 + z2
  int ustart;                                                  // first unit of this job in its list
};
struct Desc {
  Job job[MAXJ];               // [0, n0): X is the narrow operand; [n0, n): dY is (the "swapped" orientation)
  int n0, n, total0, total1;   // units of the two lists
  int g0;                      // workgroups [0, g0) serve list 0, the rest list 1
};
static_assert(sizeof(Desc) <= 4096, "kernarg segment");

struct Seg {                // one (job, z, tile) visit: units k0 .. k1 of that tile
  const char* pn; const char* pw;      // operand addresses of unit 0 of the tile (batch and tile column offsets applied)
  float* C; float* cs;                 // tile origin in dW; db slice of the tile (null: none / not this tile's duty)
  int stepn, stepw;                    // bytes per unit
  int ldnb, ldwb;                      // row strides in bytes
  int vn, vw;                          // valid widths of this tile
  int ldc;
  int k0, k1;
};

typedef const __attribute__((address_space(4))) Desc* KD;
struct KL { KD kd; int jb, je; };       // a job list

// unit u of list kl -> its segment, cut at uend
template <bool SWAP, int TN_>
__device__ __forceinline__ Seg decode(const KL& kl, int u, int uend) {
  KD kd = kl.kd;
  int j = kl.jb;
  while (j + 1 < kl.je && u >= kd->job[j + 1].ustart) ++j;
  const int local = u - kd->job[j].ustart;
  const int ku = kd->job[j].ku, t_nar = kd->job[j].t_nar, t_wid = kd->job[j].t_wid, nb2 = kd->job[j].nb2;
  const int k = local % ku;
  int t = local / ku;
  const int tn = t % t_nar; t /= t_nar;
  const int tw = t % t_wid;
  const int z = t / t_wid;
  const long long z1 = z / nb2, z2 = z % nb2;
  Seg s;
  const int ld_nar = kd->job[j].ld_nar, ld_wid = kd->job[j].ld_wid;
  s.ldnb = ld_nar * 2; s.ldwb = ld_wid * 2;
  s.stepn = KU * s.ldnb; s.stepw = KU * s.ldwb;
  s.pn = reinterpret_cast<const char*>(kd->job[j].nar) + (z1 * kd->job[j].nb1s + z2 * kd->job[j].nb2s + tn * TN_) * 2;
  s.pw = reinterpret_cast<const char*>(kd->job[j].wid) + (z1 * kd->job[j].wb1s + z2 * kd->job[j].wb2s + tw * TW_) * 2;
  s.vn = min(TN_, kd->job[j].n_nar - tn * TN_);
  s.vw = min(TW_, kd->job[j].n_wid - tw * TW_);
  s.ldc = kd->job[j].ldc;
  float* C = kd->job[j].C + z1 * kd->job[j].cb1s + z2 * kd->job[j].cb2s;
  float* cs = kd->job[j].cs;
  if (SWAP) {                            // dW rows run along the wide operand (X); db along the narrow one: the tw == 0 tiles add it
    s.C = C + (long long)tw * TW_ * s.ldc + tn * TN_;
    s.cs = (cs && tw == 0) ? cs + z1 * kd->job[j].sb1s + z2 * kd->job[j].sb2s + tn * TN_ : nullptr;
  } else {                               // dW rows run along the narrow operand (X); db along the wide one: the tn == 0 tiles add it
    s.C = C + (long long)tn * TN_ * s.ldc + tw * TW_;
    s.cs = (cs && tn == 0) ? cs + z1 * kd->job[j].sb1s + z2 * kd->job[j].sb2s + tw * TW_ : nullptr;
  }
  s.k0 = k;
  s.k1 = min(ku, k + (uend - u));
  return s;
}

// LDS-DMA: 64 lanes x 16 bytes from per-lane global addresses to 1 KB of LDS at the wave-uniform byte address `lds_byte` (M0).
// Inline asm on purpose: hipcc keeps no count of it, so the counted vmcnt waits below are the only ones (with the builtin it puts a
// vmcnt(0) in front of every LDS read that follows, which serialises the ring).
__device__ __forceinline__ void glds16(const char* g, uint32_t lds_byte) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(lds_byte) : "memory");
}

template <typename T> __device__ __forceinline__ float dot_ones(float c, uint32_t pair);
template <> __device__ __forceinline__ float dot_ones<bf16>(float c, uint32_t pair) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, pair), __builtin_bit_cast(bf16x2_t, 0x3f803f80u), c, false);
}
template <> __device__ __forceinline__ float dot_ones<f16>(float c, uint32_t pair) {
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, pair), __builtin_bit_cast(f16x2_t, 0x3c003c00u), c, false);
}
template <typename T> __device__ __forceinline__ float frag_sum(float c, const s16x8& f) {
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
  const u32x4 w = __builtin_bit_cast(u32x4, f);
#pragma unroll
  for (int e = 0; e < 4; ++e) c = dot_ones<T>(c, w[e]);
  return c;
}

typedef __attribute__((ext_vector_type(4))) short s4;
__device__ __forceinline__ s16x8 read_tr(uint32_t lds_byte_lo, uint32_t lds_byte_hi) {
  const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(uintptr_t)lds_byte_lo);
  const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(uintptr_t)lds_byte_hi);
  return (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// One workgroup's range [u0, u1) of the units of job list `jl` (all of one orientation).
//   SWAP = false: X is the narrow operand; MFMA A = narrow fragment, so D rows = cin, D columns (lane & 15) = cout: the flush adds
//                 64-byte row segments of dW.
//   SWAP = true : dY is the narrow operand; the MFMA operands are exchanged (A = wide = X), D rows = cin again.
// (Flushing D^T instead -- 16-byte pieces -- measured 4x slower: the atomic units retire 64-byte segments, tools/probes/atomic_probe.hip.)
template <typename T, bool SWAP, int TN>
__device__ __forceinline__ void body(const KL kl, const int u0, const int u1, unsigned char* ring) {
  typedef Geo<TN> GE;
  constexpr int NWV = GE::NWV, LDN = GE::LDN, CHN = GE::CHN, IMG_N = GE::IMG_N, STAGE = GE::STAGE, NS = GE::NS, SLOTS = GE::SLOTS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const uint32_t ring0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)ring);

  // DMA slot geometry of this lane: instruction i = wave + NWV s of a stage covers chunks 64 i .. 64 i + 63 of the stage image
  int srow[SLOTS], scc[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int i = wave + NWV * s;
    const bool nar = i < IMG_N / 1024;
    const int c = (nar ? i : i - IMG_N / 1024) * 64 + lane;
    srow[s] = nar ? c / CHN : c / CHW;
    scc[s] = nar ? c % CHN : c % CHW;
  }

  // ---- loader state ----
  Seg L = decode<SWAP, TN>(kl, u0, u1);
  int lu = u0, lk = L.k0;
  int voff[SLOTS];
  auto set_voff = [&]() {
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const bool nar = (wave + NWV * s) < IMG_N / 1024;
      const int nvc = (nar ? L.vn : L.vw) >> 3;
      // pad / out-of-tile chunks re-read a valid one (fetching them from a zero page instead measured 6 % slower grouped, 3.5x slower on
      // one-problem launches)
      voff[s] = srow[s] * (nar ? L.ldnb : L.ldwb) + min(scc[s], nvc - 1) * 16;
    }
  };
  set_voff();
  auto issue = [&]() {          // DMA of unit lu into ring slot (lu - u0) % NS
    const uint32_t dst = ring0 + (uint32_t)((lu - u0) % NS) * STAGE + wave * 1024;
    const char* bn = L.pn + (long long)lk * L.stepn;
    const char* bw = L.pw + (long long)lk * L.stepw;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const bool nar = (wave + NWV * s) < IMG_N / 1024;
      glds16((nar ? bn : bw) + voff[s], dst + s * (NWV * 1024));
    }
    ++lu; ++lk;
    if (lk == L.k1 && lu < u1) { L = decode<SWAP, TN>(kl, lu, u1); lk = L.k0; set_voff(); }
  };

  // ---- consumer state ----
  Seg S = decode<SWAP, TN>(kl, u0, u1);
  int ck = S.k0;
  constexpr int FA = SWAP ? 12 : 3, FB = SWAP ? 3 : 12;       // D fragments: rows x columns
  constexpr int NCS = SWAP ? 3 : 12;                           // dY fragments of a wave = D column fragments either way
  f32x4 acc[FA][FB];
  float csum[NCS];
#pragma unroll
  for (int a = 0; a < FA; ++a)
#pragma unroll
    for (int b = 0; b < FB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < NCS; ++j) csum[j] = 0.f;

  const int g = lane >> 4, p = lane & 15;
  const int kr = 4 * g + (p >> 2), c4 = 4 * (p & 3);
  const uint32_t aN = ring0 + (kr * LDN + wm * 48 + c4) * 2;
  const uint32_t aW = ring0 + IMG_N + (kr * LDW + wn * 192 + c4) * 2;
  // D row / column origin of this wave inside the tile; the dY duty (bias gradient) falls to the waves with row origin 0
  const int row0 = SWAP ? wn * 192 : wm * 48, col0 = SWAP ? wm * 48 : wn * 192;
  const bool cs_wave = SWAP ? wn == 0 : wm == 0;

  // prologue: NS - 1 units in flight
#pragma unroll 1
  for (int q = 0; q < NS - 1; ++q)
    if (lu < u1) issue();
  int landed = 0;             // units after cu whose DMA is known to have landed (after the drain of a flush)

#pragma unroll 1
  for (int cu = u0; cu < u1; ++cu) {
    // unit cu's DMA (this wave's share) must have landed: at most the younger units' instructions may be outstanding
    if (landed > 0) --landed;
    else {
      const int ahead = lu - cu - 1;
      if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * SLOTS) : "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SLOTS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();       // everybody's share landed; everybody is done reading the slot of unit cu - 1
    asm volatile("" ::: "memory");
    if (lu < u1) issue();               // refill the slot of unit cu - 1 with unit cu + NS - 1

    const uint32_t st = (uint32_t)((cu - u0) % NS) * STAGE;
    s16x8 an[3], bw[12];
#pragma unroll
    for (int i = 0; i < 3; ++i) an[i] = read_tr(aN + st + i * 32, aN + st + i * 32 + 16 * LDN * 2);
#pragma unroll
    for (int j = 0; j < 12; ++j) bw[j] = read_tr(aW + st + j * 32, aW + st + j * 32 + 16 * LDW * 2);
#pragma unroll
    for (int a = 0; a < FA; ++a)
#pragma unroll
      for (int b = 0; b < FB; ++b) acc[a][b] = SWAP ? Mma<T>::mma(bw[a], an[b], acc[a][b]) : Mma<T>::mma(an[a], bw[b], acc[a][b]);
    if (S.cs && cs_wave) {
#pragma unroll
      for (int j = 0; j < NCS; ++j) csum[j] = frag_sum<T>(csum[j], SWAP ? an[j] : bw[j]);
    }
    ++ck;
    if (ck == S.k1) {
      // ---- leave the tile: flush.  First drain the DMA queue (the slabs in flight were issued long ago), so that the counted waits
      // of the following units never have to look past 144 atomics per lane.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      landed = lu - cu - 1;
      const int ldc = S.ldc;
      const int vr = SWAP ? S.vw : S.vn, vc = SWAP ? S.vn : S.vw;      // valid D rows / columns
      float* Cl = S.C + (long long)(row0 + 4 * g) * ldc + col0 + p;
#pragma unroll
      for (int a = 0; a < FA; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool rok = row0 + 16 * a + 4 * g + r < vr;
#pragma unroll
          for (int b = 0; b < FB; ++b)
            if (rok && col0 + 16 * b + p < vc) atomicAdd(Cl + (long long)(16 * a + r) * ldc + 16 * b, acc[a][b][r]);
        }
      if (S.cs && cs_wave) {
#pragma unroll
        for (int j = 0; j < NCS; ++j) {
          float v = csum[j];
          v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
          if (g == 0 && col0 + 16 * j + p < vc) atomicAdd(S.cs + col0 + 16 * j + p, v);
        }
      }
#pragma unroll
      for (int a = 0; a < FA; ++a)
#pragma unroll
        for (int b = 0; b < FB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < NCS; ++j) csum[j] = 0.f;
      if (cu + 1 < u1) { S = decode<SWAP, TN>(kl, cu + 1, u1); ck = S.k0; }
    }
  }
}

// Workgroups [0, g0) take the X-narrow jobs, the others the dY-narrow ones: one launch, two instruction streams.
template <typename T, int TN>
__global__ __launch_bounds__(64 * Geo<TN>::NWV, TN == 96 ? 2 : 1) void wgrad_sk_kernel(Desc dsc) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char ring[];
  (void)dsc;
  KD kd = (KD)__builtin_amdgcn_kernarg_segment_ptr();
  const int g0 = kd->g0, w = blockIdx.x;
  if (w < g0) {
    const int total = kd->total0;
    const int u0 = (int)((long long)total * w / g0), u1 = (int)((long long)total * (w + 1) / g0);
    if (u0 < u1) body<T, false, TN>(KL{kd, 0, kd->n0}, u0, u1, ring);
  } else {
    const int total = kd->total1, g1 = gridDim.x - g0, w1 = w - g0;
    const int u0 = (int)((long long)total * w1 / g1), u1 = (int)((long long)total * (w1 + 1) / g1);
    if (u0 < u1) body<T, true, TN>(KL{kd, kd->n0, kd->n}, u0, u1, ring);
  }
}

}  // namespace wsk

// ---- C ABI ---------------------------------------------------------------------------------------------------------------
struct stj_wgrad_job {
  const void* x; const void* dy; float* dw; float* db;
  int rows, cin, cout, nb1, nb2;
  long long ldx, lddy, lddw;
  long long sx1, sx2, sdy1, sdy2, sdw1, sdw2, sdb1, sdb2;
};

static bool wsk_supported(const stj_wgrad_job& j, int dtype) {
  if (!stj_is16(dtype)) return false;
  if (j.rows <= 0 || j.rows % wsk::KU || j.cin <= 0 || j.cout <= 0 || j.cin % 8 || j.cout % 8 || j.nb1 <= 0 || j.nb2 <= 0) return false;
  if (((uintptr_t)j.x | (uintptr_t)j.dy) % 16) return false;
  if (j.ldx % 8 || j.lddy % 8 || j.sx1 % 8 || j.sx2 % 8 || j.sdy1 % 8 || j.sdy2 % 8) return false;
  if (j.ldx < j.cin || j.lddy < j.cout || j.lddw < j.cout) return false;
  if (j.ldx > (1 << 24) || j.lddy > (1 << 24) || j.lddw > (1 << 24)) return false;
  return true;
}

extern "C" int stj_wgrad_job_supported(const stj_wgrad_job* job, int dtype) { return job && wsk_supported(*job, dtype) ? 1 : 0; }

// dW[z] += X[z]^T dY[z], db[z] += column sums of dY[z] for every job, as ONE stream-K launch per 24 jobs on `wg_budget` workgroups
// (<= 0: one per CU).  Every job must satisfy stj_wgrad_job_supported (16-bit activations, rows % 32 == 0, widths and strides
// multiples of 8 elements, 16-byte aligned operands); others are the caller's to send through stj_gemm.
extern "C" int stj_wgrad_group(const stj_wgrad_job* jobs, int njobs, int dtype, int wg_budget, hipStream_t stream) {
  if (njobs <= 0) return STJ_OK;
  if (!jobs) { stj_set_error("stj_wgrad_group: NULL jobs"); return STJ_EINVAL; }
  for (int i = 0; i < njobs; ++i)
    if (!wsk_supported(jobs[i], dtype)) { stj_set_error("stj_wgrad_group: job %d is not supported (see stj_wgrad_job_supported)", i); return STJ_EUNSUPPORTED; }
  static PerDevice<int> attr_set;
  constexpr int lds96 = wsk::Geo<96>::NS * wsk::Geo<96>::STAGE, lds192 = wsk::Geo<192>::NS * wsk::Geo<192>::STAGE;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)wsk::wgrad_sk_kernel<bf16, 96>, hipFuncAttributeMaxDynamicSharedMemorySize, lds96) != hipSuccess ||
        hipFuncSetAttribute((const void*)wsk::wgrad_sk_kernel<f16, 96>, hipFuncAttributeMaxDynamicSharedMemorySize, lds96) != hipSuccess ||
        hipFuncSetAttribute((const void*)wsk::wgrad_sk_kernel<bf16, 192>, hipFuncAttributeMaxDynamicSharedMemorySize, lds192) != hipSuccess ||
        hipFuncSetAttribute((const void*)wsk::wgrad_sk_kernel<f16, 192>, hipFuncAttributeMaxDynamicSharedMemorySize, lds192) != hipSuccess) {
      stj_set_error("stj_wgrad_group: cannot reserve %d bytes of LDS", lds96 > lds192 ? lds96 : lds192);
      return STJ_ELAUNCH;
    }
    attr_set = 1;
  }
  static PerDevice<int> ncu;
  if (!ncu) {
    int dev = 0; hipDeviceProp_t pr;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) ncu = 256;
    else ncu = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
  }
  const int budget = wg_budget > 0 ? wg_budget : (int)ncu;
  auto tiles = [](int w, int t) { return (w + t - 1) / t; };
  for (int j0 = 0; j0 < njobs; j0 += wsk::MAXJ) {
    const int n = njobs - j0 < wsk::MAXJ ? njobs - j0 : wsk::MAXJ;
    // tile geometry of this launch: 192-column narrow slices (8 waves) when every problem is at least 192 wide on both sides
    // AND the launch is long: at least 32 of the larger units per workgroup.  (Measured, hot loop, 256 workgroups, 96 / 192-column slices:
    // cfg-512's 8192-row C = 384 stage, 24 jobs, 345 / 261 us; its 32768-row C = 192 stage 144 / 126 us; cfg-256's 2048-row C = 384 stage
    // -- 12 units per workgroup -- 59 / 99 us, its 8192-row C = 192 stage 64 / 86 us: tools/probes/wgrad_sk_budget.py.)
    int TN = 192;
    long long units192 = 0;
    for (int i = 0; i < n; ++i) {
      const stj_wgrad_job& j = jobs[j0 + i];
      if (j.cin < 192 || j.cout < 192) TN = 96;
      const long long nt_ns = (long long)tiles(j.cout, wsk::TW_) * tiles(j.cin, 192), nt_sw = (long long)tiles(j.cin, wsk::TW_) * tiles(j.cout, 192);
      units192 += (long long)j.nb1 * j.nb2 * (j.rows / wsk::KU) * (nt_ns < nt_sw ? nt_ns : nt_sw);
    }
    if (units192 < 32ll * budget) TN = 96;
#ifdef STJ_WSK_FORCE_TN
    TN = STJ_WSK_FORCE_TN;
#endif
    // orientation of a job: which operand is cut into TN-column slices.  Cost = operand columns staged per row of the problem.
    auto swapped = [&](const stj_wgrad_job& j) {
      const long long cost_ns = (long long)tiles(j.cout, wsk::TW_) * j.cin + (long long)tiles(j.cin, TN) * j.cout;      // X narrow
      const long long cost_sw = (long long)tiles(j.cin, wsk::TW_) * j.cout + (long long)tiles(j.cout, TN) * j.cin;      // dY narrow
      const long long nt_ns = (long long)tiles(j.cout, wsk::TW_) * tiles(j.cin, TN), nt_sw = (long long)tiles(j.cin, wsk::TW_) * tiles(j.cout, TN);
      return cost_sw < cost_ns || (cost_sw == cost_ns && nt_sw < nt_ns);
    };
    wsk::Desc d;
    long long total[2] = {0, 0}, cost[2] = {0, 0};
    int k = 0;
    for (int pass = 0; pass < 2; ++pass) {
      if (pass == 1) d.n0 = k;
      for (int i = 0; i < n; ++i) {
        const stj_wgrad_job& j = jobs[j0 + i];
        const bool sw = swapped(j);
        if ((int)sw != pass) continue;
        wsk::Job& o = d.job[k++];
        if (sw) {
          o.nar = j.dy; o.wid = j.x; o.ld_nar = (int)j.lddy; o.ld_wid = (int)j.ldx; o.n_nar = j.cout; o.n_wid = j.cin;
          o.nb1s = j.sdy1; o.nb2s = j.sdy2; o.wb1s = j.sx1; o.wb2s = j.sx2;
        } else {
          o.nar = j.x; o.wid = j.dy; o.ld_nar = (int)j.ldx; o.ld_wid = (int)j.lddy; o.n_nar = j.cin; o.n_wid = j.cout;
          o.nb1s = j.sx1; o.nb2s = j.sx2; o.wb1s = j.sdy1; o.wb2s = j.sdy2;
        }
        o.C = j.dw; o.cs = j.db; o.ldc = (int)j.lddw;
        o.cb1s = j.sdw1; o.cb2s = j.sdw2; o.sb1s = j.sdb1; o.sb2s = j.sdb2;
        o.t_nar = tiles(o.n_nar, TN); o.t_wid = tiles(o.n_wid, wsk::TW_);
        o.ku = j.rows / wsk::KU; o.nb2 = j.nb2;
        o.ustart = (int)total[pass];
        const long long units = (long long)j.nb1 * j.nb2 * o.t_nar * o.t_wid * o.ku;
        total[pass] += units;
        // bytes a unit moves ~ the valid columns of its two operands (averaged over the job's tiles)
        cost[pass] += (long long)j.nb1 * j.nb2 * o.ku * ((long long)o.t_wid * o.n_nar + (long long)o.t_nar * o.n_wid);
        if (total[pass] > (1LL << 30)) { stj_set_error("stj_wgrad_group: too many units"); return STJ_EINVAL; }
      }
    }
    d.n = k; d.total0 = (int)total[0]; d.total1 = (int)total[1];
#ifndef STJ_WSK_MINU
#define STJ_WSK_MINU 32
#endif
    // at least 32 slabs (1024 rows) per workgroup: every visit of a tile ends in up to 36864 f32 atomics, and short launches are faster on
    // fewer workgroups (the encoder's stage flushes, 8 problems each: 2048 rows 71 us on 256 workgroups, 55 on 192; 8192 rows 62 / 53.5;
    // the 32768-row stage and cfg-512's launches have > 32 slabs per workgroup anyway: tools/probes/wgrad_sk_budget.py)
    long long G = (total[0] + total[1]) / STJ_WSK_MINU;
    if (G > budget) G = budget;
    if (G < 1) G = 1;
    // workgroups per orientation in proportion to the bytes it streams (at least one where there is work)
    long long g0 = cost[0] + cost[1] > 0 ? (G * cost[0] + (cost[0] + cost[1]) / 2) / (cost[0] + cost[1]) : G;
    if (total[0] > 0 && g0 < 1) g0 = 1;
    if (total[1] > 0 && g0 > G - 1) g0 = G - 1;
    if (total[0] == 0) g0 = 0;
    if (total[1] == 0) g0 = G;
    if (g0 < 0 || (total[0] > 0 && g0 == 0)) { g0 = total[0] > 0 ? 1 : 0; if (G < g0 + (total[1] > 0 ? 1 : 0)) G = g0 + 1; }
    d.g0 = (int)g0;
    if (TN == 96) {
      if (dtype == STJ_BF16) hipLaunchKernelGGL((wsk::wgrad_sk_kernel<bf16, 96>), dim3((unsigned)G), dim3(256), lds96, stream, d);
      else hipLaunchKernelGGL((wsk::wgrad_sk_kernel<f16, 96>), dim3((unsigned)G), dim3(256), lds96, stream, d);
    } else {
      if (dtype == STJ_BF16) hipLaunchKernelGGL((wsk::wgrad_sk_kernel<bf16, 192>), dim3((unsigned)G), dim3(512), lds192, stream, d);
      else hipLaunchKernelGGL((wsk::wgrad_sk_kernel<f16, 192>), dim3((unsigned)G), dim3(512), lds192, stream, d);
    }
    int e = stj_check_launch("stj_wgrad_group");
    if (e) return e;
  }
  return STJ_OK;
}
