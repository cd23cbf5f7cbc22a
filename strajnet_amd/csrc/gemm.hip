// Batched strided GEMM on MFMA with fused epilogue, plus column-sum (bias gradient).
//   C[z][m,n] = epi( alpha * sum_k A[z](m,k) * B[z](k,n) )      epi: +bias[n] -> act -> +res[m,n]
// Serves every Dense / 1x1-conv / tfa-MHA projection of the hot path and their dgrad / wgrad, the Q K^T / P V
// products of the global attentions, and the time-collapsed Conv3D skips
// (reference: Keras Dense modules.py:36-37,76-79,270; FG_MSA.py:54-64,147,176; trajNet.py:32-36,71-76,195-210;
//  Conv3D(8,1,1) modules.py:693-717 -- SURVEY.md K2-K8,K10).
//
// Operand orientation is given by element strides.  A K-contiguous operand is staged into an LDS image [rows][BK]
// and read as 16-byte MFMA fragments (ds_read_b128); a row-contiguous ("transposed") operand -- x^T and dY in every
// weight gradient, W in every forward Dense with Keras [in,out] layout, V in P V -- is staged UN-transposed as
// [BK][rows] with coalesced 16-byte global reads, and the transposition happens in the fragment read
// (two ds_read_b64_tr_b16 LDS transpose reads per fragment).  The first version transposed
// while writing to LDS and lost >10x to 32-way bank conflicts on 2-byte scatter writes (profiles/r01_a_*).
// The epilogue goes through LDS so that global stores (and residual loads) are full 16-byte row segments.
// Weight gradients use split-K with f32 atomic accumulation straight into the flat gradient buffer.
#include "common.h"
#include <stdlib.h>

struct GemmArgs {
  const void* A; const void* B; void* C; const float* bias; const void* res; float* colsum;
  int M, N, K, nb1, nb2;     // batch index z = z1 * nb2 + z2, each operand has a stride per level
  long long sAb1, sAb2, sAm, sAk, sBb1, sBb2, sBk, sBn, sCb1, sCb2, ldc, sBias1, sBias2, sRes1, sRes2, ldres;
  int act, c_f32, accumulate, splitk;
  int vecA, vecB, vecC, vecR;   // 16-byte vector paths legal for A / B staging, C stores, residual loads
  float alpha;
  // K segments: the contraction runs over nkb segments of K elements; segment s of A / B starts sAkb / sBkb elements further.
  // (sum over the 8 waypoint weight sets of a shared input's gradient = ONE GEMM with K' = 8 K instead of 8 atomically
  // accumulated ones)
  int nkb; long long sAkb, sBkb;
};

template <typename T> struct PadT;            // row padding of the un-transposed image: bank-conflict-free strided reads
template <> struct PadT<bf16> { static constexpr int P = 4; };   // rows 8-byte aligned for ds_read_b64_tr_b16
template <> struct PadT<f16> { static constexpr int P = 4; };
template <> struct PadT<float> { static constexpr int P = 4; };

template <typename T> __device__ __forceinline__ T zero_of() {
  T z;
  if constexpr (sizeof(T) == 2) z.v = 0; else z = 0.f;
  return z;
}

// ---- operand staging, split into "global -> registers" (issued one k-tile ahead, so the loads fly while the MFMAs of
// the current tile run) and "registers -> LDS".  A thread owns NCH 16-byte chunks of the ROWS x BK tile.
//   RC = false: K-contiguous operand, image [ROWS][LD], chunk = VN consecutive k of one row
//   RC = true : row-contiguous operand kept un-transposed, image [BK][LDT], chunk = VN consecutive rows of one k
template <typename T, int ROWS, int BK, bool RC>
struct Stage {
  static constexpr int VN = Vec<T>::N;
  static constexpr int NCH = ROWS * BK / VN / 256;
  static constexpr int CPR = RC ? ROWS / VN : BK / VN;      // chunks along the contiguous axis
  static_assert(ROWS * BK / VN % 256 == 0, "tile must split evenly over 256 threads");

  __device__ static __forceinline__ void load(uint4 (&reg)[NCH], const T* g, long long sR, long long sK, int nr, int nk, int vec, int tid) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int idx = tid + i * 256;
      const int a = idx / CPR, c = (idx % CPR) * VN;         // KC: (row a, k c)   RC: (k a, row c)
      const int r = RC ? c : a, k = RC ? a : c;
      const bool full = RC ? (k < nk && r + VN <= nr) : (r < nr && k + VN <= nk);
      if (full && vec) {
        reg[i] = *reinterpret_cast<const uint4*>(g + (long long)r * sR + (long long)k * sK);
      } else {
        __attribute__((aligned(16))) T tmp[VN];
#pragma unroll
        for (int e = 0; e < VN; ++e) {
          const int rr = RC ? r + e : r, kk = RC ? k : k + e;
          tmp[e] = (rr < nr && kk < nk) ? g[(long long)rr * sR + (long long)kk * sK] : zero_of<T>();
        }
        reg[i] = *reinterpret_cast<const uint4*>(tmp);
      }
    }
  }
  // 16-byte-vector operands only (the launcher checks vecA / vecB and K % VN == 0): unconditional loads from a clamped address, zeroed
  // by a select -- no scalar fallback, no branch per chunk (with 6 chunks per operand the general form above spilled 400 bytes / lane)
  // The zeroing of out-of-range chunks happens in commit_m, from the mask returned here (bit i = chunk i in range).  Written as
  // `reg = in ? load : 0` the compiler sinks every load under its own exec-masked branch and rebuilds the value from phis: a
  // `s_waitcnt vmcnt(0)` directly behind the 5th of 12 loads of each k-tile (gemm_deepk TT 64x64), i.e. one exposed memory latency
  // per k-tile before the MFMAs start.
  __device__ static __forceinline__ int load_vec(uint4 (&reg)[NCH], const T* g, long long sR, long long sK, int nr, int nk, int tid) {
    int msk = 0;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int idx = tid + i * 256;
      const int a = idx / CPR, c = (idx % CPR) * VN;
      const int r = RC ? c : a, k = RC ? a : c;
      const bool in = RC ? (k < nk && r + VN <= nr) : (r < nr && k + VN <= nk);
      reg[i] = *reinterpret_cast<const uint4*>(g + (in ? (long long)r * sR + (long long)k * sK : 0));
      msk |= in ? (1 << i) : 0;
    }
    return msk;
  }
  template <int LDX>
  __device__ static __forceinline__ void commit_m(T* lds, const uint4 (&reg)[NCH], int msk, int tid) {
    asm volatile("" : "+v"(msk));        // keeps the selects below (and with them the wait for the loads) at the commit
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int idx = tid + i * 256;
      const int a = idx / CPR, c = (idx % CPR) * VN;
      T* dst = lds + a * LDX + c;
      const bool in = (msk >> i) & 1;
      const uint4 v = make_uint4(in ? reg[i].x : 0u, in ? reg[i].y : 0u, in ? reg[i].z : 0u, in ? reg[i].w : 0u);
      if constexpr (RC && sizeof(T) == 2) {                  // rows only 8-byte aligned (LDT = ROWS + 4)
        uint2* d = reinterpret_cast<uint2*>(dst);
        d[0] = make_uint2(v.x, v.y); d[1] = make_uint2(v.z, v.w);
      } else {
        *reinterpret_cast<uint4*>(dst) = v;
      }
    }
  }
  template <int LDX>
  __device__ static __forceinline__ void commit(T* lds, const uint4 (&reg)[NCH], int tid) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int idx = tid + i * 256;
      const int a = idx / CPR, c = (idx % CPR) * VN;
      T* dst = lds + a * LDX + c;
      if constexpr (RC && sizeof(T) == 2) {                  // rows only 8-byte aligned (LDT = ROWS + 4)
        uint2* d = reinterpret_cast<uint2*>(dst);
        d[0] = make_uint2(reg[i].x, reg[i].y); d[1] = make_uint2(reg[i].z, reg[i].w);
      } else {
        *reinterpret_cast<uint4*>(dst) = reg[i];
      }
    }
  }
};

// LDS bytes of one tile configuration (operand stage and epilogue stage share the buffer)
template <typename T, int BM, int BN, int WM, int WN, bool TA, bool TB, int BKB = 128>
struct GemmSmem {
  static constexpr int BK = BKB / sizeof(T);        // k-tile: BKB bytes of K per row (128 = the default 64 16-bit / 32 f32 elements)
  static constexpr int LD = BK + LdsPad<T>::P;
  static constexpr int LDTA = BM + PadT<T>::P, LDTB = BN + PadT<T>::P;
  static constexpr int FM = BM / (16 * WM);
  static constexpr int A_ELEMS = TA ? BK * LDTA : BM * LD;
  static constexpr int B_ELEMS = ((TB ? BK * LDTB : BN * LD) + 7) / 8 * 8;
  static constexpr int A_ELEMS_AL = (A_ELEMS + 7) / 8 * 8;
  static constexpr int EP_ROWS = BM < 64 ? BM : (BM % 64 == 0 ? 64 : FM * 16);
  static constexpr int LDC = BN + 4;
  static constexpr int STAGE_BYTES = (A_ELEMS_AL + B_ELEMS) * (int)sizeof(T);
  static constexpr int EPI_BYTES = EP_ROWS * LDC * 4;
  static constexpr int BYTES = STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES;
};

// One workgroup's share of a GEMM: tile / batch / k-slice (bxi, byi) of a (gdx, gdy) grid.  `smem` has GemmSmem<...>::BYTES bytes,
// `bias_s` BN floats.  (A function, not the kernel, so that gemm_group_kernel below can run tiles of SEVERAL problems in one launch.)
template <typename T, int BM, int BN, int WM, int WN, bool TA, bool TB, int BKB = 128>
__device__ __forceinline__ void gemm_body(const GemmArgs& p, const int bxi, const int byi, const int gdx, const int gdy,
                                          unsigned char* smem, float* bias_s) {
  constexpr int BK = BKB / sizeof(T);
  constexpr int LD = BK + LdsPad<T>::P;
  constexpr int LDTA = BM + PadT<T>::P, LDTB = BN + PadT<T>::P;
  constexpr int FM = BM / (16 * WM), FN = BN / (16 * WN);
  constexpr int A_ELEMS_AL = GemmSmem<T, BM, BN, WM, WN, TA, TB, BKB>::A_ELEMS_AL;
  constexpr int EP_ROWS = BM < 64 ? BM : (BM % 64 == 0 ? 64 : FM * 16);      // a wave's rows must lie inside one epilogue pass
  constexpr int LDC = BN + 4;
  T* As = reinterpret_cast<T*>(smem);
  T* Bs = As + A_ELEMS_AL;
  float* Cs = reinterpret_cast<float*>(smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_n = (p.N + BN - 1) / BN;
  // Work map.  Workgroups are dealt to the 8 XCDs round-robin by linear id, and each XCD has its own L2.  For split-K GEMMs
  // (weight gradients: 64x64 output tiles, every k-slice of X and dY is needed by ALL tiles) the tiles of one k-slice are
  // therefore given consecutive slots of the SAME XCD, so a slice is fetched into one L2 once instead of by every XCD.
  int bx = bxi, by = byi;
  if (p.splitk > 1 && (gdy & 7) == 0) {
    const int lin = bxi + gdx * byi;
    const int xcd = lin & 7, slot = lin >> 3;
    bx = slot % gdx;
    by = (slot / gdx) * 8 + xcd;
  }
  const int tm = bx / tiles_n, tn = bx % tiles_n;
  const int z = by / p.splitk, ks = by % p.splitk;
  const int m0 = tm * BM, n0 = tn * BN;

  const int ktiles_seg = (p.K + BK - 1) / BK;       // k-tiles per K segment
  const int ktiles = ktiles_seg * p.nkb;
  const int kt_per = (ktiles + p.splitk - 1) / p.splitk;
  const int kt0 = ks * kt_per, kt1 = min(ktiles, kt0 + kt_per);
  if (kt0 >= kt1 && (p.accumulate || ks != 0)) return;

  const long long z1 = z / p.nb2, z2 = z % p.nb2;
  const T* A = reinterpret_cast<const T*>(p.A) + z1 * p.sAb1 + z2 * p.sAb2 + (long long)m0 * p.sAm;
  const T* B = reinterpret_cast<const T*>(p.B) + z1 * p.sBb1 + z2 * p.sBb2 + (long long)n0 * p.sBn;

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool do_cs = p.colsum != nullptr && tm == 0;
  float cs = 0.f;
  if (tid < BN) bias_s[tid] = (p.bias && ks == 0 && n0 + tid < p.N) ? p.bias[z1 * p.sBias1 + z2 * p.sBias2 + n0 + tid] : 0.f;

  typedef Stage<T, BM, BK, TA> SA;
  typedef Stage<T, BN, BK, TB> SB;
  uint4 ra[SA::NCH], rb[SB::NCH];
  const long long a_sR = TA ? 1 : p.sAm, b_sR = TB ? 1 : p.sBn;
  int ma = 0, mb = 0;
  auto issue = [&](int kt) {
    const int seg = kt / ktiles_seg, k1 = (kt - seg * ktiles_seg) * BK;
    if constexpr (BKB != 128) {
      ma = SA::load_vec(ra, A + seg * p.sAkb + (long long)k1 * p.sAk, a_sR, p.sAk, p.M - m0, p.K - k1, tid);
      mb = SB::load_vec(rb, B + seg * p.sBkb + (long long)k1 * p.sBk, b_sR, p.sBk, p.N - n0, p.K - k1, tid);
    } else {
      SA::load(ra, A + seg * p.sAkb + (long long)k1 * p.sAk, a_sR, p.sAk, p.M - m0, p.K - k1, p.vecA, tid);
      SB::load(rb, B + seg * p.sBkb + (long long)k1 * p.sBk, b_sR, p.sBk, p.N - n0, p.K - k1, p.vecB, tid);
    }
  };
  if (kt0 < kt1) issue(kt0);
  for (int kt = kt0; kt < kt1; ++kt) {
    if constexpr (BKB != 128) {
      SA::template commit_m<TA ? LDTA : LD>(As, ra, ma, tid);
      SB::template commit_m<TB ? LDTB : LD>(Bs, rb, mb, tid);
    } else {
      SA::template commit<TA ? LDTA : LD>(As, ra, tid);
      SB::template commit<TB ? LDTB : LD>(Bs, rb, tid);
    }
    __syncthreads();
    // next tile's global loads overlap this tile's MFMAs (deep k-tiles: unconditionally -- the last tile is fetched once more and
    // dropped -- so that the loads are straight-line code)
    if constexpr (BKB != 128) issue(kt + 1 < kt1 ? kt + 1 : kt);
    else if (kt + 1 < kt1) issue(kt + 1);
    if (do_cs && tid < BN) {           // fused bias gradient: column sums of the B (= dY) tile over this block's k range
      float s = 0.f;
      if constexpr (TB) { for (int k = 0; k < BK; ++k) s += ldf(Bs + k * LDTB + tid); }
      else { for (int k = 0; k < BK; ++k) s += ldf(Bs + tid * LD + k); }
      cs += s;
    }
    for (int kk = 0; kk < BK; kk += Mma<T>::KSTEP) {
      typename Mma<T>::Frag a[FM], b[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        if constexpr (TA) a[i] = Mma<T>::load_tr(As, LDTA, (wm * FM + i) * 16, kk, lane);
        else a[i] = Mma<T>::load(As, LD, (wm * FM + i) * 16, kk, lane);
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        if constexpr (TB) b[j] = Mma<T>::load_tr(Bs, LDTB, (wn * FN + j) * 16, kk, lane);
        else b[j] = Mma<T>::load(Bs, LD, (wn * FN + j) * 16, kk, lane);
      }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = Mma<T>::mma(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }

  if (do_cs && tid < BN && n0 + tid < p.N) atomicAdd(p.colsum + z1 * p.sBias1 + z2 * p.sBias2 + n0 + tid, cs);

  // ---- epilogue through LDS: EP_ROWS rows per pass, full-row 16-byte global accesses ----
  const T* res = p.res ? reinterpret_cast<const T*>(p.res) + z1 * p.sRes1 + z2 * p.sRes2 : nullptr;
  float* Cf = reinterpret_cast<float*>(p.C) + z1 * p.sCb1 + z2 * p.sCb2;
  T* Ct = reinterpret_cast<T*>(p.C) + z1 * p.sCb1 + z2 * p.sCb2;
  constexpr int WROWS = FM * 16;                 // rows owned by one wave
  for (int r0 = 0; r0 < BM; r0 += EP_ROWS) {
    if (wm * WROWS >= r0 && wm * WROWS < r0 + EP_ROWS) {
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            Cs[(wm * WROWS - r0 + i * 16 + (lane >> 4) * 4 + r) * LDC + (wn * FN + j) * 16 + (lane & 15)] = acc[i][j][r] * p.alpha;
    }
    __syncthreads();
    if (p.accumulate) {
      for (int i = tid; i < EP_ROWS * BN; i += 256) {
        const int r = i / BN, c = i % BN;
        const int row = m0 + r0 + r, col = n0 + c;
        if (row < p.M && col < p.N) {
          float v = Cs[r * LDC + c];
          v += bias_s[c];
          atomicAdd(Cf + (long long)row * p.ldc + col, v);
        }
      }
    } else if (p.c_f32) {
      constexpr int CH = BN / 4;
      for (int i = tid; i < EP_ROWS * CH; i += 256) {
        const int r = i / CH, c = (i % CH) * 4;
        const int row = m0 + r0 + r, col = n0 + c;
        if (row >= p.M || col >= p.N) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = Cs[r * LDC + c + e] + bias_s[c + e];
        if (p.act == ACT_ELU) {          // uniform branch, hoisted out of the per-element work
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = elu_f(v[e]);
        } else if (p.act == ACT_GELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_f(v[e]);
        }
        float* dst = Cf + (long long)row * p.ldc + col;
        if (p.vecC && col + 4 <= p.N && !res) {
          *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          for (int e = 0; e < 4 && col + e < p.N; ++e) dst[e] = v[e] + (res ? ldf(res + (long long)row * p.ldres + col + e) : 0.f);
        }
      }
    } else {
      constexpr int VN = Vec<T>::N;
      constexpr int CH = BN / VN;
      for (int i = tid; i < EP_ROWS * CH; i += 256) {
        const int r = i / CH, c = (i % CH) * VN;
        const int row = m0 + r0 + r, col = n0 + c;
        if (row >= p.M || col >= p.N) continue;
        float v[VN];
#pragma unroll
        for (int e = 0; e < VN; ++e) v[e] = Cs[r * LDC + c + e] + bias_s[c + e];
        if (p.act == ACT_ELU) {          // uniform branch, hoisted out of the per-element work
#pragma unroll
          for (int e = 0; e < VN; ++e) v[e] = sizeof(T) == 2 ? elu_bf(v[e]) : elu_f(v[e]);
        } else if (p.act == ACT_GELU) {
#pragma unroll
          for (int e = 0; e < VN; ++e) v[e] = gelu_f(v[e]);
        }
        T* dst = Ct + (long long)row * p.ldc + col;
        if (p.vecC && col + VN <= p.N) {
          if (res) {
            float rr[VN];
            if (p.vecR) ld16(res + (long long)row * p.ldres + col, rr);
            else {
#pragma unroll
              for (int e = 0; e < VN; ++e) rr[e] = ldf(res + (long long)row * p.ldres + col + e);
            }
#pragma unroll
            for (int e = 0; e < VN; ++e) v[e] += rr[e];
          }
          st16(dst, v);
        } else {
          for (int e = 0; e < VN && col + e < p.N; ++e) stf(dst + e, v[e] + (res ? ldf(res + (long long)row * p.ldres + col + e) : 0.f));
        }
      }
    }
    __syncthreads();
  }
}

template <typename T, int BM, int BN, int WM, int WN, bool TA, bool TB>
__global__ __launch_bounds__(256, (BM == 64 && BN == 64 && sizeof(T) == 2) ? ((!TA && !TB) ? 4 : 5) : 1) void gemm_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[GemmSmem<T, BM, BN, WM, WN, TA, TB>::BYTES];
  __shared__ float bias_s[BN];            // bias of this column tile: ONE coalesced load instead of per-element global loads in the epilogue
  gemm_body<T, BM, BN, WM, WN, TA, TB>(p, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y, smem, bias_s);
}

// Deep k-tile variant (16-bit types, plain GEMMs): 192 elements of K per barrier pair instead of 64.  The Dense layers of the small
// stages (2048 .. 16384 rows, K = 192 .. 1536) are bound by the dependent chain load -> LDS -> barrier -> MFMA -> barrier of each k-tile
// (~2 us per link whatever its size, tools/bench_gemm_pmc.py); a K = 384 layer is 2 links instead of 6.
template <typename T, int BM, int BN, bool TA, bool TB>
__global__ __launch_bounds__(256, BM == 32 ? 5 : 3) void gemm_deepk_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[GemmSmem<T, BM, BN, 2, 2, TA, TB, 384>::BYTES];
  __shared__ float bias_s[BN];
  gemm_body<T, BM, BN, 2, 2, TA, TB, 384>(p, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y, smem, bias_s);
}

// ---- grouped launch: up to GG_MAX independent GEMMs (same dtype, 64x64 or 32x32 tiles, any operand orientation) in ONE kernel ----
// The small layers of the hot path (agent encoder, FG-MSA, the cross-attentions, the 16x16 Swin stage) form dependent chains of
// 5-15 us launches in which the input gradient and the weight gradient of a Dense layer, the dP / dV and dQ / dK products of an
// attention, or the q / k / v projections are independent of each other but cost a link of the chain each.  A group puts the tiles
// of such problems behind one another in one grid: problem i owns workgroups [start[i], start[i+1]) (starts are multiples of 8 so
// that the XCD-aware work map of split-K problems still sees its own XCD).
#define GG_MAX 4
struct GemmGroup {
  GemmArgs p[GG_MAX];
  int start[GG_MAX + 1], gx[GG_MAX], gy[GG_MAX], cfg[GG_MAX];      // cfg = (32x32 tiles ? 4 : 0) + 2 TA + TB (+ 8: deep k-tile, groups of one only)
  int n;
};
template <typename T>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? 5 : 1) void gemm_group_kernel(GemmGroup g) {
  constexpr int B0 = GemmSmem<T, 64, 64, 2, 2, false, false>::BYTES, B1 = GemmSmem<T, 64, 64, 2, 2, true, true>::BYTES;
  constexpr int B2 = GemmSmem<T, 64, 64, 2, 2, true, false>::BYTES, B3 = GemmSmem<T, 64, 64, 2, 2, false, true>::BYTES;
  constexpr int BA = B0 > B1 ? B0 : B1, BB = B2 > B3 ? B2 : B3;
  __shared__ __attribute__((aligned(16))) unsigned char smem[BA > BB ? BA : BB];
  __shared__ float bias_s[64];
  // The descriptor is read straight from the kernarg segment with scalar loads: indexing the by-value parameter `g` with a
  // run-time i would make the compiler copy all of it to scratch (1.1 KB per lane, every p.field a scratch load).
  typedef const __attribute__((address_space(4))) GemmGroup* KG;
  KG kg = (KG)__builtin_amdgcn_kernarg_segment_ptr();
  (void)g;
  const int bid = blockIdx.x;
  const int n = kg->n;
  int i = 0;
  while (i + 1 < n && bid >= kg->start[i + 1]) ++i;
  const int l = bid - kg->start[i];
  const int gx = kg->gx[i], gy = kg->gy[i];
  if (l >= gx * gy) return;                      // padding up to the next multiple of 8
  typedef const __attribute__((address_space(4))) uint32_t* KW;
  KW src = (KW)(&kg->p[i]);
  union { GemmArgs a; uint32_t w[sizeof(GemmArgs) / 4]; } u;
#pragma unroll
  for (int j = 0; j < (int)(sizeof(GemmArgs) / 4); ++j) u.w[j] = src[j];
  const GemmArgs& p = u.a;
  const int bx = l % gx, by = l / gx;
  switch (kg->cfg[i]) {
    case 0: gemm_body<T, 64, 64, 2, 2, false, false>(p, bx, by, gx, gy, smem, bias_s); break;
    case 1: gemm_body<T, 64, 64, 2, 2, false, true>(p, bx, by, gx, gy, smem, bias_s); break;
    case 2: gemm_body<T, 64, 64, 2, 2, true, false>(p, bx, by, gx, gy, smem, bias_s); break;
    case 3: gemm_body<T, 64, 64, 2, 2, true, true>(p, bx, by, gx, gy, smem, bias_s); break;
    case 4: gemm_body<T, 32, 32, 2, 2, false, false>(p, bx, by, gx, gy, smem, bias_s); break;
    case 5: gemm_body<T, 32, 32, 2, 2, false, true>(p, bx, by, gx, gy, smem, bias_s); break;
    case 6: gemm_body<T, 32, 32, 2, 2, true, false>(p, bx, by, gx, gy, smem, bias_s); break;
    default: gemm_body<T, 32, 32, 2, 2, true, true>(p, bx, by, gx, gy, smem, bias_s); break;
  }
}

// =====================================================================================================
// Row-streaming linear kernel (bf16 / fp16): Y[M,N] = epi(X[M,K] W) for the token-major Dense layers with M >> N,K
// (Swin qkv/proj/fc1/fc2 and their input gradients at 32768 / 8192 tokens, the per-waypoint 1x1 skips).
// The generic tile kernel above spends its time on per-tile fixed costs there (two staged operands, 2-4 barriers and an
// LDS epilogue for 12 MFMAs per wave).  Here the WEIGHT tile (NC output columns x all of K) is staged in LDS once per
// block and is the MFMA A operand (m = output column); the activations never touch LDS: a lane loads its MFMA B
// fragment (row = lane&15, 8 consecutive k) straight from global memory with one 16-byte load per k-step, for the whole K
// up front.  D[m = column 4g+r][n = row] leaves each lane with 4 consecutive output columns of one row -> 8-byte stores
// (+bias, ELU, residual) directly from the accumulators.  One barrier per block; 128 rows x NC columns per block.
// =====================================================================================================
template <typename T, int KS, int NC, bool TB>
__global__ __launch_bounds__(256, 3) void linear_rs_kernel(GemmArgs p) {
  constexpr int K = KS * 32;
  constexpr int LDW = TB ? NC + 4 : K + 16;       // TB: image [K][NC+4] (n contiguous, as stored); else [NC][K+16]
  constexpr int NJ = NC / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char rs_smem[];
  T* Ws = reinterpret_cast<T*>(rs_smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, ln = lane & 15;
  const int z = blockIdx.z;
  const long long z1 = z / p.nb2, z2 = z % p.nb2;
  const int n0 = blockIdx.y * NC;
  const int m0 = blockIdx.x * 128 + wave * 32;
  const T* A = reinterpret_cast<const T*>(p.A) + z1 * p.sAb1 + z2 * p.sAb2;
  const T* B = reinterpret_cast<const T*>(p.B) + z1 * p.sBb1 + z2 * p.sBb2;

  // activations: B fragments of this wave's 32 rows, all k-steps (loads in flight while the weight tile is staged)
  s16x8 xa[2][KS];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = m0 + 16 * i + ln;
    const T* ap = A + (long long)row * p.sAm + 8 * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (row < p.M) xa[i][ks] = *reinterpret_cast<const s16x8*>(ap + ks * 32);
      else xa[i][ks] = (s16x8){0, 0, 0, 0, 0, 0, 0, 0};
    }
  }
  // weight tile -> LDS
  if constexpr (TB) {           // W[k][n], n contiguous: chunk = 8 consecutive n of one k
    constexpr int CPR = NC / 8;
    for (int q = tid; q < K * CPR; q += 256) {
      const int k = q / CPR, c = (q % CPR) * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (n0 + c < p.N) v = *reinterpret_cast<const uint4*>(B + (long long)k * p.sBk + n0 + c);     // N % 8 == 0 (dispatch)
      uint2* d = reinterpret_cast<uint2*>(Ws + k * LDW + c);
      d[0] = make_uint2(v.x, v.y); d[1] = make_uint2(v.z, v.w);
    }
  } else {                      // W^T[n][k], k contiguous
    constexpr int CPR = K / 8;
    for (int q = tid; q < NC * CPR; q += 256) {
      const int n = q / CPR, c = (q % CPR) * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (n0 + n < p.N) v = *reinterpret_cast<const uint4*>(B + (long long)(n0 + n) * p.sBn + c);
      *reinterpret_cast<uint4*>(Ws + n * LDW + c) = v;
    }
  }
  __syncthreads();
  if (m0 >= p.M) return;

  f32x4 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      s16x8 wf;
      if constexpr (TB) wf = Mma<T>::load_tr(Ws, LDW, j * 16, ks * 32, lane);
      else wf = Mma<T>::load(Ws, LDW, j * 16, ks * 32, lane);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i][j] = Mma<T>::mma(wf, xa[i][ks], acc[i][j]);
    }

  // epilogue from the accumulators: lane = 4 consecutive columns (n0 + 16 j + 4 g ..) of row (m0 + 16 i + ln)
  const float* bias = p.bias ? p.bias + z1 * p.sBias1 + z2 * p.sBias2 : nullptr;
  const T* res = p.res ? reinterpret_cast<const T*>(p.res) + z1 * p.sRes1 + z2 * p.sRes2 : nullptr;
  T* C = reinterpret_cast<T*>(p.C) + z1 * p.sCb1 + z2 * p.sCb2;
  const bool elu = p.act == ACT_ELU;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int col = n0 + 16 * j + 4 * g;
    if (col >= p.N) continue;                    // N % 4 == 0 (dispatch)
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) bv = *reinterpret_cast<const float4*>(bias + col);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = m0 + 16 * i + ln;
      if (row >= p.M) continue;
      float v0 = acc[i][j][0] * p.alpha + bv.x, v1 = acc[i][j][1] * p.alpha + bv.y;
      float v2 = acc[i][j][2] * p.alpha + bv.z, v3 = acc[i][j][3] * p.alpha + bv.w;
      if (elu) { v0 = elu_bf(v0); v1 = elu_bf(v1); v2 = elu_bf(v2); v3 = elu_bf(v3); }
      if (res) {
        const uint2 rv = *reinterpret_cast<const uint2*>(res + (long long)row * p.ldres + col);
        float r0, r1, r2, r3;
        unpack2<T>(rv.x, r0, r1); unpack2<T>(rv.y, r2, r3);
        v0 += r0; v1 += r1; v2 += r2; v3 += r3;
      }
      *reinterpret_cast<uint2*>(C + (long long)row * p.ldc + col) = make_uint2(pack2<T>(v0, v1), pack2<T>(v2, v3));
    }
  }
}

template <typename T, int KS, int NC, bool TB>
static bool rs_launch2(const GemmArgs& p, hipStream_t st) {
  constexpr int K = KS * 32;
  constexpr size_t lds = (size_t)(TB ? K * (NC + 4) : NC * (K + 16)) * 2;
  static PerDevice<bool> attr_set;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)linear_rs_kernel<T, KS, NC, TB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
    attr_set = true;
  }
  dim3 grid((p.M + 127) / 128, (p.N + NC - 1) / NC, p.nb1 * p.nb2);
  hipLaunchKernelGGL((linear_rs_kernel<T, KS, NC, TB>), grid, dim3(256), lds, st, p);
  return true;
}
template <typename T, int KS, int NC>
static bool rs_launch(const GemmArgs& p, bool tb, hipStream_t st) {
  return tb ? rs_launch2<T, KS, NC, true>(p, st) : rs_launch2<T, KS, NC, false>(p, st);
}
// true when the row-streaming kernel took the problem
template <typename T>
static bool linear_rs_try(const GemmArgs& p, bool ta, bool tb, hipStream_t st) {
  static int enabled = -1;
  const int rs_min_m = 16384;      // measured: wins at 32768 tokens (1.3-1.7x), loses at <= 8192 (too few 128-row blocks for 256 CUs)
  if (enabled < 0) { const char* e = getenv("STJ_NO_RS"); enabled = !(e && atoi(e)); }
  if (!enabled || ta || p.c_f32 || p.accumulate || p.splitk != 1 || p.colsum || p.nkb != 1) return false;
  if (p.M < rs_min_m || p.K % 32 || p.N % 8 || p.sAk != 1 || !p.vecA || !p.vecB || !p.vecC) return false;
  if (p.act != ACT_NONE && p.act != ACT_ELU) return false;
  if (p.res && (!p.vecR)) return false;
  if (p.bias && (((uintptr_t)p.bias) % 16 || p.sBias1 % 4 || p.sBias2 % 4)) return false;
  if (!tb && p.sBk != 1) return false;
  if ((long long)p.nb1 * p.nb2 > 65535) return false;
  switch (p.K) {
    case 96: return rs_launch<T, 3, 128>(p, tb, st);
    case 128: return rs_launch<T, 4, 128>(p, tb, st);
    case 192: return rs_launch<T, 6, 128>(p, tb, st);
    case 288: return rs_launch<T, 9, 64>(p, tb, st);
    case 384: return rs_launch<T, 12, 64>(p, tb, st);
    default: return false;
  }
}

template <typename T, int BM, int BN, int WM, int WN>
static void launch_tile(const GemmArgs& p, bool ta, bool tb, hipStream_t st) {
  dim3 grid(((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN), p.nb1 * p.nb2 * p.splitk), blk(256);
  if (ta && tb) hipLaunchKernelGGL((gemm_kernel<T, BM, BN, WM, WN, true, true>), grid, blk, 0, st, p);
  else if (ta) hipLaunchKernelGGL((gemm_kernel<T, BM, BN, WM, WN, true, false>), grid, blk, 0, st, p);
  else if (tb) hipLaunchKernelGGL((gemm_kernel<T, BM, BN, WM, WN, false, true>), grid, blk, 0, st, p);
  else hipLaunchKernelGGL((gemm_kernel<T, BM, BN, WM, WN, false, false>), grid, blk, 0, st, p);
}

// ---- group recording (stj_gemm_group_begin / _end): the state lives in a CALLER-OWNED host buffer (the library keeps none) ----
struct GroupState { unsigned magic; int rec_dtype, rec_blocks; GemmGroup grp; };
static constexpr unsigned GROUP_MAGIC = 0x53544a47u;     // "STJG"

static int group_flush(GroupState* gs, hipStream_t st) {
  GemmGroup& g_grp = gs->grp;
  if (g_grp.n == 0) return STJ_OK;
  GemmGroup g = g_grp;
  const int dtype = gs->rec_dtype;
  g_grp.n = 0; gs->rec_blocks = 0; gs->rec_dtype = -1;
  if (g.n == 1) {                                   // a group of one is an ordinary launch
    const int c = g.cfg[0] & 7, deep1 = g.cfg[0] & 8;
    dim3 grid(g.gx[0], g.gy[0]);
    if (deep1) {
#define STJ_ONE_DEEP(T) \
      switch (c) { \
        case 0: hipLaunchKernelGGL((gemm_deepk_kernel<T, 64, 64, false, false>), grid, dim3(256), 0, st, g.p[0]); break; \
        case 1: hipLaunchKernelGGL((gemm_deepk_kernel<T, 64, 64, false, true>), grid, dim3(256), 0, st, g.p[0]); break; \
        case 2: hipLaunchKernelGGL((gemm_deepk_kernel<T, 64, 64, true, false>), grid, dim3(256), 0, st, g.p[0]); break; \
        case 3: hipLaunchKernelGGL((gemm_deepk_kernel<T, 64, 64, true, true>), grid, dim3(256), 0, st, g.p[0]); break; \
        case 4: hipLaunchKernelGGL((gemm_deepk_kernel<T, 32, 32, false, false>), grid, dim3(256), 0, st, g.p[0]); break; \
        case 5: hipLaunchKernelGGL((gemm_deepk_kernel<T, 32, 32, false, true>), grid, dim3(256), 0, st, g.p[0]); break; \
        case 6: hipLaunchKernelGGL((gemm_deepk_kernel<T, 32, 32, true, false>), grid, dim3(256), 0, st, g.p[0]); break; \
        default: hipLaunchKernelGGL((gemm_deepk_kernel<T, 32, 32, true, true>), grid, dim3(256), 0, st, g.p[0]); break; \
      }
      if (dtype == STJ_BF16) { STJ_ONE_DEEP(bf16) } else { STJ_ONE_DEEP(f16) }
#undef STJ_ONE_DEEP
      return stj_check_launch("stj_gemm(group of 1, deep k)");
    }
#define STJ_ONE(T) \
    switch (c) { \
      case 0: hipLaunchKernelGGL((gemm_kernel<T, 64, 64, 2, 2, false, false>), grid, dim3(256), 0, st, g.p[0]); break; \
      case 1: hipLaunchKernelGGL((gemm_kernel<T, 64, 64, 2, 2, false, true>), grid, dim3(256), 0, st, g.p[0]); break; \
      case 2: hipLaunchKernelGGL((gemm_kernel<T, 64, 64, 2, 2, true, false>), grid, dim3(256), 0, st, g.p[0]); break; \
      case 3: hipLaunchKernelGGL((gemm_kernel<T, 64, 64, 2, 2, true, true>), grid, dim3(256), 0, st, g.p[0]); break; \
      case 4: hipLaunchKernelGGL((gemm_kernel<T, 32, 32, 2, 2, false, false>), grid, dim3(256), 0, st, g.p[0]); break; \
      case 5: hipLaunchKernelGGL((gemm_kernel<T, 32, 32, 2, 2, false, true>), grid, dim3(256), 0, st, g.p[0]); break; \
      case 6: hipLaunchKernelGGL((gemm_kernel<T, 32, 32, 2, 2, true, false>), grid, dim3(256), 0, st, g.p[0]); break; \
      default: hipLaunchKernelGGL((gemm_kernel<T, 32, 32, 2, 2, true, true>), grid, dim3(256), 0, st, g.p[0]); break; \
    }
    if (dtype == STJ_BF16) { STJ_ONE(bf16) } else if (dtype == STJ_F16) { STJ_ONE(f16) } else { STJ_ONE(float) }
#undef STJ_ONE
    return stj_check_launch("stj_gemm(group of 1)");
  }
  for (int i = 0; i < g.n; ++i) g.cfg[i] &= 7;      // deep k-tile bodies are not in the group kernel (measured: its occupancy drops 4 -> 3, 916 -> 908 scenes/s)
  const int total = g.start[g.n];
  if (dtype == STJ_BF16) hipLaunchKernelGGL(gemm_group_kernel<bf16>, dim3(total), dim3(256), 0, st, g);
  else if (dtype == STJ_F16) hipLaunchKernelGGL(gemm_group_kernel<f16>, dim3(total), dim3(256), 0, st, g);
  else hipLaunchKernelGGL(gemm_group_kernel<float>, dim3(total), dim3(256), 0, st, g);
  return stj_check_launch("stj_gemm(group)");
}

template <typename T>
static int launch_gemm(GemmArgs& p, bool ta, bool tb, int dtype, GroupState* gs, hipStream_t st) {
  if (gs) {
    GemmGroup& g_grp = gs->grp;
    int& g_rec_dtype = gs->rec_dtype;
    int& g_rec_blocks = gs->rec_blocks;
    // recorded, not launched: 64x64 tiles, or 32x32 when that leaves most CUs idle; split-K as for a plain launch
    const long long nb = (long long)p.nb1 * p.nb2;
    const long long tiles64 = (long long)((p.M + 63) / 64) * ((p.N + 63) / 64) * nb;
    const bool small = !p.accumulate && tiles64 < 256 && p.M >= 32 && p.N >= 32;
    const int bm = small ? 32 : 64;
    const long long tiles = (long long)((p.M + bm - 1) / bm) * ((p.N + bm - 1) / bm);
    constexpr int BK = 128 / sizeof(T);
    if (p.splitk == 0) {
      const int ktiles = ((p.K + BK - 1) / BK) * p.nkb;
      long long s2 = (768 + tiles * nb - 1) / (tiles * nb);
      if (s2 > 96) s2 = 96;
      if (s2 > ktiles / 2) s2 = ktiles / 2;
      if (s2 < 1) s2 = 1;
      if (s2 >= 8 && nb == 1) s2 = s2 / 8 * 8;
      p.splitk = (int)s2;
    }
    const long long gy = nb * p.splitk;
    if (tiles * gy > (1 << 20)) { stj_set_error("stj_gemm: problem too large for a group"); return STJ_EINVAL; }
    if (g_grp.n == GG_MAX || (g_grp.n > 0 && g_rec_dtype != dtype)) { int e = group_flush(gs, st); if (e) return e; }
    const int i = g_grp.n++;
    g_rec_dtype = dtype;
    g_grp.p[i] = p;
    g_grp.gx[i] = (int)tiles; g_grp.gy[i] = (int)gy;
    const bool deep = sizeof(T) == 2 && !p.accumulate && p.splitk == 1 && p.K > 128 && p.vecA && p.vecB && p.K % 8 == 0 && p.M % 8 == 0 &&
                      p.N % 8 == 0;
    g_grp.cfg[i] = (deep ? 8 : 0) + (small ? 4 : 0) + (ta ? 2 : 0) + (tb ? 1 : 0);
    g_grp.start[i] = g_rec_blocks;
    g_rec_blocks += (int)((tiles * gy + 7) / 8 * 8);
    g_grp.start[i + 1] = g_rec_blocks;
    return STJ_OK;
  }
  if constexpr (sizeof(T) == 2) {
    if (linear_rs_try<T>(p, ta, tb, st)) return stj_check_launch("stj_gemm(rs)");
  }
  constexpr int BK = 128 / sizeof(T);
  const long long nb = (long long)p.nb1 * p.nb2;
  const long long tiles64 = (long long)((p.M + 63) / 64) * ((p.N + 63) / 64) * nb;
  const int ktiles = ((p.K + BK - 1) / BK) * p.nkb;
  // Tile choice.  The hot-path GEMMs are skinny (K <= 1536, mostly 96..384) and latency / memory-parallelism bound, not
  // MFMA bound: measured on every Dense shape of the model, 64x64 tiles (4x the workgroups in flight) beat 128x128 by
  // 1.3-1.8x (profiles/r01_c_gemm_tiles.txt), so 64x64 is the default; 128x128 only pays for genuinely large problems.
  // (a 128x128 configuration existed for problems beyond 2e11 FLOP with K >= 1024: no product of this model -- cfg-512 and batch-32
  // inference included -- reached it, and its 12 instantiations were 0.3 MB of the library)
  int cfg = 2;   // 2: 64x64, 3: 32x32
  // Few rows (the 16x16 Swin stage, the 64-agent encoder): 64x64 tiles leave most of the 256 CUs idle and a plain GEMM cannot
  // split K.  32x32 tiles quadruple the workgroups.
  if (!p.accumulate && tiles64 < 256 && p.M >= 32 && p.N >= 32) cfg = 3;
  // Weight gradients (split-K, f32 atomics): measured (tools/bench_gemm.py, STJ_WGRAD_CFG) 96x128 / 128x96 / 96x96 tiles -- exact covers
  // of the 96 * 2^i Swin widths, 3x the MFMAs per barrier pair -- and 128x128 tiles against the 64x64 default: slower on every
  // shape but two (e.g. [192x576] over 8192 rows 32.8 vs 18.8 us, [96x384] over 32768 rows 31.8 vs 28.7 us; 128x128: 38-47 us), 828
  // vs 860 scenes/s end to end: these launches are bound by how many workgroups are in flight, not by what one of them does.
  if (p.splitk == 0) {            // auto split-K (accumulating GEMMs only): aim at ~2 blocks per CU
    const long long tiles = tiles64;
    const int tgt = 768, cap = 96;      // (swept repeatedly: within noise around these)
    long long s = (tgt + tiles - 1) / tiles;
    if (s > cap) s = cap;                 // bound same-address atomic contention
    if (s > ktiles / 2) s = ktiles / 2;
    if (s < 1) s = 1;
    if (s * nb > 65535) s = 65535 / nb;
    if (s >= 8 && nb == 1) s = s / 8 * 8;      // multiple of 8: enables the XCD-aware work map
    p.splitk = (int)s;
  }
  if constexpr (sizeof(T) == 2) {
    if ((cfg == 2 || cfg == 3) && p.K > 128 && p.vecA && p.vecB && p.K % 8 == 0 && p.M % 8 == 0 && p.N % 8 == 0) {
      dim3 grid(cfg == 2 ? (unsigned)(((p.M + 63) / 64) * ((p.N + 63) / 64)) : (unsigned)(((p.M + 31) / 32) * ((p.N + 31) / 32)), p.nb1 * p.nb2 * p.splitk), blk(256);
#define STJ_DEEP(BMN) \
      do { \
        if (ta && tb) hipLaunchKernelGGL((gemm_deepk_kernel<T, BMN, BMN, true, true>), grid, blk, 0, st, p); \
        else if (ta) hipLaunchKernelGGL((gemm_deepk_kernel<T, BMN, BMN, true, false>), grid, blk, 0, st, p); \
        else if (tb) hipLaunchKernelGGL((gemm_deepk_kernel<T, BMN, BMN, false, true>), grid, blk, 0, st, p); \
        else hipLaunchKernelGGL((gemm_deepk_kernel<T, BMN, BMN, false, false>), grid, blk, 0, st, p); \
      } while (0)
      if (cfg == 2) STJ_DEEP(64); else STJ_DEEP(32);
#undef STJ_DEEP
      return stj_check_launch("stj_gemm(deep k)");
    }
  }
  if (cfg == 3) launch_tile<T, 32, 32, 2, 2>(p, ta, tb, st);
  else launch_tile<T, 64, 64, 2, 2>(p, ta, tb, st);
  return stj_check_launch("stj_gemm");
}

// colsum (optional, f32, accumulate GEMMs only): colsum[z][n] += sum_k B[z](k,n)  -- the bias gradient that goes with
// a weight gradient dW = x^T dY; addressed with the bias batch strides.
extern "C" int stj_gemm(const void* A, const void* B, void* C, const float* bias, const void* res, float* colsum,
                        int M, int N, int K, int nb1, int nb2,
                        long long sAb1, long long sAb2, long long sAm, long long sAk,
                        long long sBb1, long long sBb2, long long sBk, long long sBn,
                        long long sCb1, long long sCb2, long long ldc,
                        long long sBias1, long long sBias2, long long sRes1, long long sRes2, long long ldres,
                        int act, float alpha, int dtype, int c_f32, int accumulate, int splitk,
                        int nkb, long long sAkb, long long sBkb, void* group, hipStream_t stream) {
  if (M <= 0 || N <= 0 || nb1 <= 0 || nb2 <= 0) return STJ_OK;
  GroupState* gs = reinterpret_cast<GroupState*>(group);
  if (gs && gs->magic != GROUP_MAGIC) { stj_set_error("stj_gemm: group was not initialised by stj_gemm_group_begin"); return STJ_EINVAL; }
  if (K < 0 || splitk < 0 || nkb < 1) { stj_set_error("stj_gemm: bad K/splitk/nkb"); return STJ_EINVAL; }
  if (accumulate && !c_f32) { stj_set_error("stj_gemm: accumulate requires f32 output"); return STJ_EINVAL; }
  if (splitk != 1 && !accumulate) { stj_set_error("stj_gemm: splitk != 1 requires accumulate"); return STJ_EINVAL; }
  if (colsum && (!accumulate || bias)) { stj_set_error("stj_gemm: colsum needs accumulate and no bias"); return STJ_EINVAL; }
  if (accumulate && (act != ACT_NONE || res)) { stj_set_error("stj_gemm: accumulate with nonlinear epilogue"); return STJ_EINVAL; }
  if ((long long)nb1 * nb2 * (splitk > 0 ? splitk : 1) > 65535) { stj_set_error("stj_gemm: batch*splitk too large"); return STJ_EINVAL; }
  GemmArgs p;
  p.A = A; p.B = B; p.C = C; p.bias = bias; p.res = res; p.colsum = colsum;
  p.M = M; p.N = N; p.K = K; p.nb1 = nb1; p.nb2 = nb2;
  p.sAb1 = sAb1; p.sAb2 = sAb2; p.sAm = sAm; p.sAk = sAk;
  p.sBb1 = sBb1; p.sBb2 = sBb2; p.sBk = sBk; p.sBn = sBn;
  p.sCb1 = sCb1; p.sCb2 = sCb2; p.ldc = ldc;
  p.sBias1 = sBias1; p.sBias2 = sBias2; p.sRes1 = sRes1; p.sRes2 = sRes2; p.ldres = ldres;
  p.act = act; p.c_f32 = c_f32; p.accumulate = accumulate; p.splitk = splitk; p.alpha = alpha;
  p.nkb = nkb; p.sAkb = sAkb; p.sBkb = sBkb;
  const long long es = stj_is16(dtype) ? 2 : 4;
  auto al = [&](const void* ptr, long long esz, long long s0, long long s1, long long s2) {
    return ((uintptr_t)ptr % 16 == 0) && ((s0 * esz) % 16 == 0) && ((s1 * esz) % 16 == 0) && ((s2 * esz) % 16 == 0);
  };
  const bool ta = (sAm == 1 && sAk != 1), tb = (sBn == 1 && sBk != 1);
  p.vecA = ta ? al(A, es, sAk, sAb1, sAb2) : (sAk == 1 ? al(A, es, sAm, sAb1, sAb2) : 0);
  p.vecB = tb ? al(B, es, sBk, sBb1, sBb2) : (sBk == 1 ? al(B, es, sBn, sBb1, sBb2) : 0);
  if (nkb > 1) { p.vecA = p.vecA && (sAkb * es) % 16 == 0; p.vecB = p.vecB && (sBkb * es) % 16 == 0; }
  p.vecC = al(C, c_f32 ? 4 : es, ldc, sCb1, sCb2);
  p.vecR = res ? al(res, es, ldres, sRes1, sRes2) : 0;
  if (dtype == STJ_BF16) return launch_gemm<bf16>(p, ta, tb, dtype, gs, stream);
  if (dtype == STJ_F16) return launch_gemm<f16>(p, ta, tb, dtype, gs, stream);
  if (dtype == STJ_F32) return launch_gemm<float>(p, ta, tb, dtype, gs, stream);
  stj_set_error("stj_gemm: bad dtype %d", dtype);
  return STJ_EINVAL;
}

// Group launch: stj_gemm calls that are handed a group (their last pointer argument) are RECORDED into it (arguments validated, their
// `stream` only used if the group fills up and has to be flushed early) and launched by stj_gemm_group_end as one kernel per GG_MAX
// problems of equal dtype.  The problems of a group must be independent of each other (no output of one is an operand of another).
// `group` is caller-owned HOST memory of stj_gemm_group_workspace_bytes() bytes: the library itself holds no state.
extern "C" long long stj_gemm_group_workspace_bytes(void) { return (long long)sizeof(GroupState); }
extern "C" int stj_gemm_group_begin(void* group) {
  if (!group) { stj_set_error("stj_gemm_group_begin: NULL group"); return STJ_EINVAL; }
  GroupState* gs = reinterpret_cast<GroupState*>(group);
  gs->magic = GROUP_MAGIC; gs->grp.n = 0; gs->rec_blocks = 0; gs->rec_dtype = -1;
  return STJ_OK;
}
extern "C" int stj_gemm_group_end(void* group, hipStream_t stream) {
  GroupState* gs = reinterpret_cast<GroupState*>(group);
  if (!gs || gs->magic != GROUP_MAGIC) { stj_set_error("stj_gemm_group_end: not a group"); return STJ_EINVAL; }
  return group_flush(gs, stream);
}

// ---- column sums: out[n] += sum_m X[m, n]  (bias gradients; out is f32, accumulated atomically) ----
// Threads are laid out as (16-byte column vectors) x (row lanes); each block streams a strip of rows with
// coalesced 16-byte loads, reduces over its row lanes in LDS, then issues one atomic per column.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* X, float* out, int M, int N, long long ld, int rows_per_block, int vec) {
  constexpr int VN = Vec<T>::N;
  __shared__ float part[256 * VN];
  const int cv = (N + VN - 1) / VN;                 // column vectors
  const int cvb = cv < 256 ? cv : 256;              // column vectors handled per pass
  const int rl = 256 / cvb;                         // row lanes
  const int tx = threadIdx.x % cvb, ty = threadIdx.x / cvb;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
  for (int c0 = 0; c0 < cv; c0 += cvb) {
    const int c = (c0 + tx) * VN;
    float s[VN];
#pragma unroll
    for (int e = 0; e < VN; ++e) s[e] = 0.f;
    if (ty < rl && c < N) {
      if (vec && c + VN <= N) {
        for (int r = r0 + ty; r < r1; r += rl) {
          float v[VN];
          ld16(X + (long long)r * ld + c, v);
#pragma unroll
          for (int e = 0; e < VN; ++e) s[e] += v[e];
        }
      } else {
        for (int r = r0 + ty; r < r1; r += rl)
          for (int e = 0; e < VN && c + e < N; ++e) s[e] += ldf(X + (long long)r * ld + c + e);
      }
    }
#pragma unroll
    for (int e = 0; e < VN; ++e) part[threadIdx.x * VN + e] = s[e];
    __syncthreads();
    if (ty == 0 && c < N) {
      for (int e = 0; e < VN && c + e < N; ++e) {
        float t = 0.f;
        for (int q = 0; q < rl; ++q) t += part[(q * cvb + tx) * VN + e];
        atomicAdd(out + c + e, t);
      }
    }
    __syncthreads();
  }
}

extern "C" int stj_colsum(const void* X, float* out, int M, int N, long long ld, int dtype, hipStream_t stream) {
  if (M <= 0 || N <= 0) return STJ_OK;
  if (!stj_dtype_ok(dtype)) { stj_set_error("stj_colsum: bad dtype %d", dtype); return STJ_EINVAL; }
  const long long es = stj_is16(dtype) ? 2 : 4;
  int rpb = (M + 511) / 512;
  if (rpb < 64) rpb = 64;
  const int vec = ((uintptr_t)X % 16 == 0) && ((ld * es) % 16 == 0);
  dim3 grid((M + rpb - 1) / rpb);
  if (dtype == STJ_BF16) hipLaunchKernelGGL(colsum_kernel<bf16>, grid, dim3(256), 0, stream, (const bf16*)X, out, M, N, ld, rpb, vec);
  else if (dtype == STJ_F16) hipLaunchKernelGGL(colsum_kernel<f16>, grid, dim3(256), 0, stream, (const f16*)X, out, M, N, ld, rpb, vec);
  else hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, stream, (const float*)X, out, M, N, ld, rpb, vec);
  return stj_check_launch("stj_colsum");
}
