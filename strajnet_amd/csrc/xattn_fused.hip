// Fused Cross_AttentionT block: the 8 time-separated cross-attentions of TrajNetCrossAttention as ONE kernel per
// direction.  Reference trajNet.py:189-234 (Cross_AttentionT: tfa MultiHeadAttention(head_size 42, 3 heads, output 128, dropout .1)
// -> LayerNorm(1e-3) -> Dense(512, elu) -> Dropout -> Dense(384) -> Dropout -> LayerNorm(1e-3)) and :305-317 (the loop over the 8
// waypoints and `+ query`).  Per waypoint set z and scene b:
//     q = query Wq[z] / sqrt(42)            [HW, 3 x 42]          k, v = key Wk[z], key Wv[z]   [64, 3 x 42]  (projected by the caller)
//     P = softmax(q k^T + -1e10 (1 - mask)) ; Pd = dropout(P) ; O = Pd v
//     v1 = O Wo[z] + bo ; n1 = LN(v1) ; hd = dropout(elu(n1 W1[z] + b1)) ; u2 = dropout(hd W2[z] + b2) ; y = LN(u2) + query
//
// Work layout (same scheme as the fused Swin kernels, swin_fused.hip): one workgroup = 64 tokens of one (z, b), a wave OWNS 16
// tokens for the whole chain.  Every product is computed transposed, D[m = output column][n = token], so the accumulator
// fragments of one product are the B operand of the next (Chain<T>): q -> S -> P -> O -> v1 -> LN -> hidden -> u2 never leave
// registers.  Only weights and the 64 projected keys / values of the scene go through LDS.  The weights of a set are streamed as
// CHUNKS that stj_xattn_pack laid out ahead of time exactly as the LDS images the fragment reads want (zero-padded 42 -> 48 head
// columns, padded row strides): staging a chunk is a flat 16-byte copy global -> registers (issued one chunk ahead) -> LDS,
// double buffered, one barrier per chunk.  blockIdx % Z = z: the 8 sets sit on the 8 XCDs, a set's 0.7 MB stream stays in one L2.
#include "common.h"
#include "rng.h"
#include "chain48.h"

namespace xat {
using namespace chain;
constexpr int CB = 384, NH = 3, HS = 42, HP = 48, O1 = 128, F1 = 512, NKEY = 64, TOK = 64;
constexpr int cmax(int a, int b) { return a > b ? a : b; }

template <typename T> struct Geo {
  static constexpr int ES = (int)sizeof(T);
  static constexpr int KSTEP = Mma<T>::KSTEP;
  static constexpr int KS = CB / KSTEP;                    // k-steps over the 384 query channels
  static constexpr int LDQ = HP + 4;                       // Wq head image [in][LDQ] (k = in rows, m = head column contiguous)
  static constexpr int KQ = ES == 2 ? 192 : 96;            // `in` rows per Wq chunk
  static constexpr int NQC = CB / KQ;                      // chunks per head
  static constexpr int QCH = KQ * LDQ;                     // elements per Wq chunk
  // row strides of the images read with ds_read_b64_tr_b16 (4 k-rows x 32 bytes per 16-lane group): stride = 8 dwords (mod 64) keeps the
  // four rows on distinct banks; C + 8 elements (stride 4 mod 64 for C = 128, 384) measured 2.2 us per FFN chunk, i.e. LDS-conflict bound
  static constexpr int LDO = O1 + (ES == 2 ? 16 : 4);      // Wo head image [48][LDO] (k = head column rows, m = out contiguous)
  static constexpr int OCH = HP * LDO;
  static constexpr int HC = KSTEP;                         // hidden columns per FFN chunk (one MFMA k-step)
  static constexpr int NFC = F1 / HC;
  static constexpr int LD1 = HC + 4;                       // W1 slice image [128][LD1]
  static constexpr int LD2 = CB + (ES == 2 ? 16 : 4);      // W2 slice image [HC][LD2]
  static constexpr int FCH = O1 * LD1 + HC * LD2;
  static constexpr int OFF_Q = 0, OFF_O = NH * NQC * QCH, OFF_F = OFF_O + NH * OCH;
  static constexpr int STREAM = OFF_F + NFC * FCH;         // elements per weight set
  static constexpr int BUF = cmax(QCH, cmax(OCH, FCH));
  // the forward kernel stages every chunk as a FIXED-size copy of BUFE elements (whole 4 KB pieces: 256 threads x 16 bytes), reading
  // past the end of the shorter chunks into what follows them in the stream (stj_xattn_pack pads the tail): no size cases, no guards
  static constexpr int BUFE = (BUF * ES + 4095) / 4096 * 4096 / ES;
  static constexpr int NCH = NH * NQC + NH + NFC;          // chunks per set: Wq (head, k-part), Wo (head), FFN slices
  static constexpr int FI = NH * NQC + NH;                 // index of the first FFN chunk
  __host__ __device__ static constexpr long long chunk_off(int i) {
    return i < NH * NQC ? (long long)i * QCH : (i < FI ? (long long)OFF_O + (long long)(i - NH * NQC) * OCH : (long long)OFF_F + (long long)(i - FI) * FCH);
  }
  static constexpr int LDK = HP + 4;                       // per-head K / V tiles [64][LDK]
  static constexpr int KS1 = O1 / KSTEP;                   // k-steps over the 128 projection outputs
  static constexpr int QS = NH * HP;                       // row stride of the saved q / O / dq tensors (3 x 48, pads zero)
  static_assert((QCH * ES) % 16 == 0 && (OCH * ES) % 16 == 0 && (FCH * ES) % 16 == 0 && (O1 * LD1 * ES) % 16 == 0, "16-byte chunks");
};

template <typename T> __device__ __forceinline__ float elu_t(float x) {
  if constexpr (sizeof(T) == 4) return elu_f(x); else return elu_bf(x);
}
template <typename T> __device__ __forceinline__ float exp_t(float x) {
  if constexpr (sizeof(T) == 4) return expf(x); else return __expf(x);
}

// ---- weight-chunk staging: flat 16-byte copy global -> registers (one chunk ahead) -> LDS ---------------------------------------
template <typename T> struct Stage {
  static constexpr int NR = (Geo<T>::BUF * (int)sizeof(T) / 16 + 255) / 256;
  uint4 r[NR];
  template <int ELEMS> __device__ __forceinline__ void issue(const T* src, int tid) {
    constexpr int NV = ELEMS * (int)sizeof(T) / 16;
    const uint4* s = reinterpret_cast<const uint4*>(src);
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      uint4 v = make_uint4(0, 0, 0, 0);          // (always store a selected value: a conditional store keeps the array in scratch)
      if (i * 256 < NV) {
        const int q = tid + i * 256;
        if ((i + 1) * 256 <= NV || q < NV) v = s[q];
      }
      r[i] = v;
    }
  }
  template <int ELEMS> __device__ __forceinline__ void commit(T* dst, int tid) const {
    constexpr int NV = ELEMS * (int)sizeof(T) / 16;
    uint4* d = reinterpret_cast<uint4*>(dst);
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      if (i * 256 < NV) {
        const int q = tid + i * 256;
        if ((i + 1) * 256 <= NV || q < NV) d[q] = r[i];
      }
    }
  }
};

// fixed-size variant (forward kernel): BUFE elements per chunk, every load unconditional, TWO chunks in flight (two of these)
template <typename T> struct StageF {
  static constexpr int NR = Geo<T>::BUFE * (int)sizeof(T) / 16 / 256;
  uint4 r[NR];
  __device__ __forceinline__ void issue(const T* src, int tid) {
    const uint4* s = reinterpret_cast<const uint4*>(src) + tid;
#pragma unroll
    for (int i = 0; i < NR; ++i) { const uint4 t = s[i * 256]; r[i] = make_uint4(t.x, t.y, t.z, t.w); }   // (component-wise: whole-struct copies keep r[] in scratch)
  }
  __device__ __forceinline__ void commit(T* dst, int tid) const {
    uint4* d = reinterpret_cast<uint4*>(dst) + tid;
#pragma unroll
    for (int i = 0; i < NR; ++i) d[i * 256] = make_uint4(r[i].x, r[i].y, r[i].z, r[i].w);
  }
};

// the 64 projected keys / values of head h of (z, b): global [.., 64, 3 x 42] -> registers -> per-head tiles [64][LDK] (pad columns stay zero)
template <typename T> struct KvStage {
  static constexpr int DW = HS * (int)sizeof(T) / 4;            // dwords per row and head
  static constexpr int RS = NH * HS * (int)sizeof(T) / 4;       // row stride in dwords
  static constexpr int N = (2 * NKEY * DW + 255) / 256;
  uint32_t r[N];
  __device__ __forceinline__ void issue(const T* k, const T* v, long long row0, int h, int tid) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int idx = tid + i * 256;
      uint32_t w = 0;
      if (idx < 2 * NKEY * DW) {
        const int t = idx / (NKEY * DW), rem = idx % (NKEY * DW), rr = rem / DW, c = rem % DW;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(t ? v : k) + (row0 + rr) * RS + h * DW + c;
        w = *src;
      }
      r[i] = w;
    }
  }
  __device__ __forceinline__ void commit(T* Kt, T* Vt, int tid) const {
    constexpr int LDW = Geo<T>::LDK * (int)sizeof(T) / 4;       // tile row stride in dwords
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int idx = tid + i * 256;
      if (idx < 2 * NKEY * DW) {
        const int t = idx / (NKEY * DW), rem = idx % (NKEY * DW), rr = rem / DW, c = rem % DW;
        reinterpret_cast<uint32_t*>(t ? Vt : Kt)[rr * LDW + c] = r[i];
      }
    }
  }
};

struct Args {
  const void* query; const void* k; const void* v; const int* kvalid; const void* pack;
  const float* bo; const float* g1; const float* be1; const float* b1; const float* b2; const float* g2; const float* be2;
  long long zstride;
  void* y; void* sq; void* so; void* sv1; void* su2;
  int Z, B, HW;
  const long long* rng; int site_a, site_1, site_2; float p_drop;
  // backward
  const void* dy; void* dquery; float* dkp; float* dvp;            // dkp / dvp: per-tile partial sums [Z][B][HW/64][64][3 x 48] f32
  void* hd; void* dpre; void* du2; void* n1; void* dv1; void* dq;    // operands of the weight-gradient GEMMs, written once
  float* dg1; float* dbe1; float* dbo; float* dg2; float* dbe2;     // "+=" LayerNorm gamma / beta and projection-bias gradients (set 0; set z at + z * zstride)
};

// dropout keep decisions of the 4 consecutive draws starting at `idx` (idx % 4 == 0), as scale factors
__device__ __forceinline__ void keep_scale(const long long* rng, int site, long long idx, float p, float sc, float (&f)[4]) {
  bool k[4];
  keep4(rng, site, idx >> 2, p, k);
#pragma unroll
  for (int e = 0; e < 4; ++e) f[e] = k[e] ? sc : 0.f;
}

// D[jd] += A^T-image [k = 64 keys][rows = head column] times the chained key-major fragments st[4]   (O^T = V^T P^T, dq^T = K^T dS^T)
template <typename T>
__device__ __forceinline__ void keys_contract(const T* tile, const f32x4 (&st)[4], f32x4 (&o)[3], int lane) {
  typedef Geo<T> G;
#pragma unroll
  for (int s = 0; s < NKEY / G::KSTEP; ++s) {
    const typename Ch<T>::Frag pf = Ch<T>::from_acc(&st[s * Ch<T>::ND]);
#pragma unroll
    for (int jd = 0; jd < 3; ++jd) o[jd] = Mma<T>::mma(Ch<T>::ldA_tr(tile, G::LDK, 16 * jd, s * G::KSTEP, lane), pf, o[jd]);
  }
}
// softmax over the 64 keys of the lane's token (fragments st[j][r]: key 16 j + 4 g + r; partner lanes g' hold the other keys)
template <typename T>
__device__ __forceinline__ void softmax_keys(f32x4 (&st)[4], const int* kval, float scale, int g) {
  float m = -INFINITY;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = st[j][r] * scale;
      if (!kval[16 * j + 4 * g + r]) v = v + (-10e9f);          // f32 add, as the reference (tfa: logits += -10e9 * (1 - mask))
      st[j][r] = v;
      m = fmaxf(m, v);
    }
  m = fmaxf(m, __shfl_xor(m, 16, 64)); m = fmaxf(m, __shfl_xor(m, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float e = exp_t<T>(st[j][r] - m); st[j][r] = e; sum += e; }
  sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.f / sum;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) st[j][r] *= inv;
}

// =====================================================================================================================
// forward
// =====================================================================================================================
template <typename T>
__global__ __launch_bounds__(256, 1) void xattn_fwd_kernel(Args p) {
  typedef Geo<T> G;
  constexpr int KSTEP = G::KSTEP, KS = G::KS, ND = Ch<T>::ND, LK = Mma<T>::LANE_K, KS1 = G::KS1;
  extern __shared__ __attribute__((aligned(16))) unsigned char xa_smem[];
  T* Kt = reinterpret_cast<T*>(xa_smem);
  T* Vt = Kt + NKEY * G::LDK;
  T* buf0 = Vt + NKEY * G::LDK;
  int* kval = reinterpret_cast<int*>(buf0 + 2 * G::BUFE);
  float* b1s = reinterpret_cast<float*>(kval + NKEY);         // FFN1 bias in LDS: a global load inside the chunk loop would wait for the weight prefetch issued just before it
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g = lane >> 4, ln = lane & 15;
  const int z = blockIdx.x % p.Z, rest = blockIdx.x / p.Z, tiles = p.HW / TOK, tile = rest % tiles, b = rest / tiles;
  const int tok = tile * TOK + wv * 16 + ln;
  const long long R = (long long)p.B * p.HW;
  const long long zrow = (long long)z * R + (long long)b * p.HW + tok;          // row of this lane's token in the [Z, B, HW, .] tensors
  const long long kvrow0 = ((long long)z * p.B + b) * NKEY;
  const T* wz = reinterpret_cast<const T*>(p.pack) + (long long)z * G::STREAM;
  const long long zo = (long long)z * p.zstride;
  const bool train = p.rng != nullptr && p.p_drop > 0.f;
  const float dsc = train ? 1.0f / (1.0f - p.p_drop) : 1.f;

  // weight chunks: two in flight in registers (st[i & 1] holds chunk i), committed to LDS buffer i & 1 when chunk i - 2 has been consumed
  // (two named objects and lambdas that take one by reference: an array indexed by the chunk parity stayed in scratch memory)
  StageF<T> sA, sB;
  T* const bufA = buf0;
  T* const bufB = buf0 + G::BUFE;
  static_assert(G::NQC % 2 == 0 && (NH * G::NQC) % 2 == 0 && G::NFC % 2 == 0, "chunk parities are fixed per phase");
  sA.issue(wz + G::chunk_off(0), tid);
  sB.issue(wz + G::chunk_off(1), tid);
  KvStage<T> kvs;
  kvs.issue(reinterpret_cast<const T*>(p.k), reinterpret_cast<const T*>(p.v), kvrow0, 0, tid);
  typename Mma<T>::Frag xa[KS];
  {
    const T* px = reinterpret_cast<const T*>(p.query) + zrow * CB + LK * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xa[ks] = Mma<T>::from_global(px + ks * KSTEP);
  }
  for (int i = tid; i < 2 * NKEY * (G::LDK - HS); i += 256) {           // pad columns of the K / V tiles: zero, once
    const int t = i / (NKEY * (G::LDK - HS)), rem = i % (NKEY * (G::LDK - HS));
    stf((t ? Vt : Kt) + (rem / (G::LDK - HS)) * G::LDK + HS + rem % (G::LDK - HS), 0.f);
  }
  if (tid < NKEY) kval[tid] = p.kvalid ? p.kvalid[(long long)b * NKEY + tid] : 1;
  for (int i = tid; i < F1; i += 256) b1s[i] = p.b1[zo + i];

  const float scale = 0.15430334996209191f;        // 42^-1/2 (tfa: query /= sqrt(head_size))
  f32x4 of[NH][3];
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    f32x4 qf[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) qf[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto q_step = [&](int c, StageF<T>& sg, T* W) __attribute__((always_inline)) {
      const int gi = h * G::NQC + c;
      sg.commit(W, tid);
      // every wave is past the previous head's attention once it has passed the barrier of chunk 0: the tiles may be rewritten now
      if (c == 1) kvs.commit(Kt, Vt, tid);
      __syncthreads();
      sg.issue(wz + G::chunk_off(gi + 2), tid);
      if (c == 1 && h + 1 < NH) kvs.issue(reinterpret_cast<const T*>(p.k), reinterpret_cast<const T*>(p.v), kvrow0, h + 1, tid);
#pragma unroll
      for (int kk = 0; kk < G::KQ / KSTEP; ++kk)
#pragma unroll
        for (int j = 0; j < 3; ++j)
          qf[j] = Mma<T>::mma(Mma<T>::load_tr(W, G::LDQ, 16 * j, kk * KSTEP, lane), xa[c * (G::KQ / KSTEP) + kk], qf[j]);
    };
#pragma unroll
    for (int c = 0; c < G::NQC; c += 2) { q_step(c, sA, bufA); q_step(c + 1, sB, bufB); }
    if (p.sq) {
      T* sq = reinterpret_cast<T*>(p.sq) + zrow * G::QS + HP * h + 4 * g;
#pragma unroll
      for (int j = 0; j < 3; ++j) { const float v[4] = {qf[j][0], qf[j][1], qf[j][2], qf[j][3]}; st4(sq + 16 * j, v); }
    }
    // ---- attention of this head for the wave's 16 tokens
    f32x4 st[4];
    {
      HeadOp<T> qop;
      qop.from_acc(qf);
#pragma unroll
      for (int j = 0; j < 4; ++j) st[j] = k48_rows<T>(Kt, G::LDK, 16 * j, qop, lane, (f32x4){0.f, 0.f, 0.f, 0.f});
    }
    softmax_keys<T>(st, kval, scale, g);
    if (train) {
      const long long base = ((((long long)z * p.B + b) * NH + h) * p.HW + tok) * NKEY + 4 * g;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float f[4];
        keep_scale(p.rng, p.site_a, base + 16 * j, p.p_drop, dsc, f);
#pragma unroll
        for (int r = 0; r < 4; ++r) st[j][r] *= f[r];
      }
    }
#pragma unroll
    for (int jd = 0; jd < 3; ++jd) of[h][jd] = (f32x4){0.f, 0.f, 0.f, 0.f};
    keys_contract<T>(Vt, st, of[h], lane);
    if (p.so) {
      T* so = reinterpret_cast<T*>(p.so) + zrow * G::QS + HP * h + 4 * g;
#pragma unroll
      for (int jd = 0; jd < 3; ++jd) { const float v[4] = {of[h][jd][0], of[h][jd][1], of[h][jd][2], of[h][jd][3]}; st4(so + 16 * jd, v); }
    }
  }

  // ---- v1^T = Wo^T O^T + bo  (head by head: one Wo chunk per head)
  f32x4 o1[O1 / 16];
#pragma unroll
  for (int f = 0; f < O1 / 16; ++f) {
    const float4 bv = *reinterpret_cast<const float4*>(p.bo + zo + 16 * f + 4 * g);
    o1[f] = (f32x4){bv.x, bv.y, bv.z, bv.w};
  }
  auto o_step = [&](int h, StageF<T>& sg, T* W) __attribute__((always_inline)) {
    const int gi = NH * G::NQC + h;
    sg.commit(W, tid);
    __syncthreads();
    sg.issue(wz + G::chunk_off(gi + 2), tid);
    HeadOp<T> oop;
    oop.from_acc(of[h]);
#pragma unroll
    for (int f = 0; f < O1 / 16; ++f) o1[f] = k48_tr<T>(W, G::LDO, 16 * f, oop, lane, o1[f]);
  };
  static_assert(NH == 3, "three Wo chunks: A, B, A");
  o_step(0, sA, bufA); o_step(1, sB, bufB); o_step(2, sA, bufA);
  if (p.sv1) {
    T* s1 = reinterpret_cast<T*>(p.sv1) + zrow * O1 + 4 * g;
#pragma unroll
    for (int f = 0; f < O1 / 16; ++f) { const float v[4] = {o1[f][0], o1[f][1], o1[f][2], o1[f][3]}; st4(s1 + 16 * f, v); }
  }
  // ---- LayerNorm(1e-3) over the 128 columns of the lane's token (accumulator layout: 32 values here, the rest in the 3 partner lanes)
  {
    float s = 0.f;
#pragma unroll
    for (int f = 0; f < O1 / 16; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) s += o1[f][r];
    s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
    const float mean = s * (1.f / O1);
    float q = 0.f;
#pragma unroll
    for (int f = 0; f < O1 / 16; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float d = o1[f][r] - mean; q += d * d; }
    q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
    const float rstd = rsqrtf(q * (1.f / O1) + 1e-3f);
#pragma unroll
    for (int f = 0; f < O1 / 16; ++f) {
      const float4 gm = *reinterpret_cast<const float4*>(p.g1 + zo + 16 * f + 4 * g);
      const float4 bt = *reinterpret_cast<const float4*>(p.be1 + zo + 16 * f + 4 * g);
      o1[f][0] = (o1[f][0] - mean) * rstd * gm.x + bt.x; o1[f][1] = (o1[f][1] - mean) * rstd * gm.y + bt.y;
      o1[f][2] = (o1[f][2] - mean) * rstd * gm.z + bt.z; o1[f][3] = (o1[f][3] - mean) * rstd * gm.w + bt.w;
    }
  }
  typename Ch<T>::Frag n1c[KS1];
#pragma unroll
  for (int s = 0; s < KS1; ++s) n1c[s] = Ch<T>::from_acc(&o1[s * ND]);

  // ---- FFN: u2^T = W2^T dropout(elu(W1^T n1^T + b1)), hidden chunk by hidden chunk (the 512-wide tile never exists)
  f32x4 o2[CB / 16];
#pragma unroll
  for (int f = 0; f < CB / 16; ++f) o2[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto ffn_step = [&](int c, StageF<T>& sg, T* W1s) __attribute__((always_inline)) {
    T* W2s = W1s + O1 * G::LD1;
    sg.commit(W1s, tid);
    __syncthreads();
    // (unconditional: a staging array written under a condition stays in scratch memory; past the end the last chunk is simply fetched again)
    sg.issue(wz + G::OFF_F + (long long)(c + 2 < G::NFC ? c + 2 : G::NFC - 1) * G::FCH, tid);
    f32x4 a1[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) {
      const float4 bv = *reinterpret_cast<const float4*>(b1s + c * G::HC + 16 * d + 4 * g);
      a1[d] = (f32x4){bv.x, bv.y, bv.z, bv.w};
    }
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
      for (int d = 0; d < ND; ++d) a1[d] = Mma<T>::mma(Ch<T>::ldA_tr(W1s, G::LD1, 16 * d, ks * KSTEP, lane), n1c[ks], a1[d]);
#pragma unroll
    for (int d = 0; d < ND; ++d) {
      float f[4] = {1.f, 1.f, 1.f, 1.f};
      if (train) keep_scale(p.rng, p.site_1, zrow * F1 + c * G::HC + 16 * d + 4 * g, p.p_drop, dsc, f);
#pragma unroll
      for (int r = 0; r < 4; ++r) a1[d][r] = elu_t<T>(a1[d][r]) * f[r];
    }
    const typename Ch<T>::Frag hf = Ch<T>::from_acc(a1);
#pragma unroll
    for (int f = 0; f < CB / 16; ++f) o2[f] = Mma<T>::mma(Ch<T>::ldA_tr(W2s, G::LD2, 16 * f, 0, lane), hf, o2[f]);
  };
  static_assert(G::FI % 2 == 1, "first FFN chunk on the B stage");
#pragma unroll 1
  for (int c = 0; c < G::NFC; c += 2) {
    ffn_step(c, sB, bufB);              // (the first FFN chunk has odd index FI)
    ffn_step(c + 1, sA, bufA);
  }

  // ---- epilogue: u2 = dropout(acc + b2) ; y = LN(u2) + query
  float s = 0.f;
#pragma unroll
  for (int f = 0; f < CB / 16; ++f) {
    const int col = 16 * f + 4 * g;
    const float4 bv = *reinterpret_cast<const float4*>(p.b2 + zo + col);
    float fk[4] = {1.f, 1.f, 1.f, 1.f};
    if (train) keep_scale(p.rng, p.site_2, zrow * CB + col, p.p_drop, dsc, fk);
    o2[f][0] = (o2[f][0] + bv.x) * fk[0]; o2[f][1] = (o2[f][1] + bv.y) * fk[1];
    o2[f][2] = (o2[f][2] + bv.z) * fk[2]; o2[f][3] = (o2[f][3] + bv.w) * fk[3];
    s += o2[f][0] + o2[f][1] + o2[f][2] + o2[f][3];
  }
  if (p.su2) {
    T* s2 = reinterpret_cast<T*>(p.su2) + zrow * CB + 4 * g;
#pragma unroll
    for (int f = 0; f < CB / 16; ++f) { const float v[4] = {o2[f][0], o2[f][1], o2[f][2], o2[f][3]}; st4(s2 + 16 * f, v); }
  }
  s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
  const float mean = s * (1.f / CB);
  float q = 0.f;
#pragma unroll
  for (int f = 0; f < CB / 16; ++f)
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float d = o2[f][r] - mean; q += d * d; }
  q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
  const float rstd = rsqrtf(q * (1.f / CB) + 1e-3f);
  T* y = reinterpret_cast<T*>(p.y) + zrow * CB;
  const T* xr = reinterpret_cast<const T*>(p.query) + zrow * CB;
  float xs[CB / 16][4];           // the query row, read before the first store: y may alias it, and a load left in the store loop waits
                                  // behind the previous store (24 dependent round trips at the end of every wave)
  float4 gms[CB / 16], bts[CB / 16];
#pragma unroll
  for (int f = 0; f < CB / 16; ++f) {
    ld4(xr + 16 * f + 4 * g, xs[f]);
    gms[f] = *reinterpret_cast<const float4*>(p.g2 + zo + 16 * f + 4 * g);
    bts[f] = *reinterpret_cast<const float4*>(p.be2 + zo + 16 * f + 4 * g);
  }
#pragma unroll
  for (int f = 0; f < CB / 16; ++f) {
    const int col = 16 * f + 4 * g;
    const float4 gm = gms[f], bt = bts[f];
    const float v[4] = {(o2[f][0] - mean) * rstd * gm.x + bt.x + xs[f][0], (o2[f][1] - mean) * rstd * gm.y + bt.y + xs[f][1],
                        (o2[f][2] - mean) * rstd * gm.z + bt.z + xs[f][2], (o2[f][3] - mean) * rstd * gm.w + bt.w + xs[f][3]};
    st4(y + col, v);
  }
}

// =====================================================================================================================
// weight packing: the stream of LDS images one set's chunks are staged from (see the header)
// =====================================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void xattn_pack_kernel(const float* wq, const float* wo, const float* w1, const float* w2, long long zstride,
                                                         T* out) {
  typedef Geo<T> G;
  const int z = blockIdx.y;
  const float* q0 = wq + z * zstride; const float* o0 = wo + z * zstride; const float* a0 = w1 + z * zstride; const float* c0 = w2 + z * zstride;
  T* dst = out + (long long)z * G::STREAM;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < G::STREAM; e += gridDim.x * 256) {
    float val = 0.f;
    if (e < G::OFF_O) {
      const int hc = e / G::QCH, w = e % G::QCH, r = w / G::LDQ, col = w % G::LDQ, h = hc / G::NQC, i = (hc % G::NQC) * G::KQ + r;
      if (col < HS) val = q0[((long long)h * CB + i) * HS + col];
    } else if (e < G::OFF_F) {
      const int idx = e - G::OFF_O, h = idx / G::OCH, w = idx % G::OCH, r = w / G::LDO, col = w % G::LDO;
      if (r < HS && col < O1) val = o0[((long long)h * HS + r) * O1 + col];
    } else {
      const int idx = e - G::OFF_F, c = idx / G::FCH;
      int w = idx % G::FCH;
      if (w < O1 * G::LD1) {
        const int r = w / G::LD1, col = w % G::LD1;
        if (col < G::HC) val = a0[(long long)r * F1 + c * G::HC + col];
      } else {
        w -= O1 * G::LD1;
        const int r = w / G::LD2, col = w % G::LD2;
        if (col < CB) val = c0[(long long)(c * G::HC + r) * CB + col];
      }
    }
    stf(dst + e, val);
  }
}


// =====================================================================================================================
// backward (tape.gradient of the forward kernel).  Same ownership: a wave = 16 tokens of a (z, b) tile.  Reads dy, query, the
// projected k / v, the forward's saves (q, v1 = pre-LN1, u2 = pre-LN2; O is only an operand of the dWo GEMM) and the same packed
// weight stream; re-derives the three dropout masks.  Writes dquery, per-tile partial sums of dk / dv (the sums over TOKENS cross
// the waves: Pd, dS, dO and q go through LDS tiles [token][.] and each wave then owns 16 KEYS), and -- once -- the operands of the
// weight-gradient GEMMs the caller launches: hd [.,512], dpre [.,512], du2 [.,384], n1 [.,128], dv1 [.,128], dq [.,144]:
//     dW2 = hd^T du2 (+ db2 = colsum du2), dW1 = n1^T dpre (+ db1), dWo[h] = O_h^T dv1, dWq[h] = query^T dq_h.
// LayerNorm gamma / beta and the projection-bias gradients are reduced here (rows -> shuffles -> LDS -> one atomic per column and
// workgroup).  Chunk order: FFN slices, Wo heads, then per head the Wq chunks (dquery += Wq dq).
// =====================================================================================================================
template <typename T> struct BGeo {
  typedef Geo<T> G;
  static constexpr int NBUF = sizeof(T) == 2 ? 2 : 1;               // f32: one weight buffer (LDS budget), two barriers per chunk
  static constexpr int LDP = NKEY + (sizeof(T) == 2 ? 8 : 4);       // Pd / dS tiles [64 tokens][LDP] (k = token rows, m = key contiguous)
  static constexpr int LDT = HP + 4;                                // dO / q tiles [64 tokens][LDT]
  static constexpr int NRED = 3 * O1 + 2 * CB + F1;             // LayerNorm / bias gradient sums + the FFN1 bias copy
  static constexpr int LDS_BYTES = (2 * NKEY * G::LDK + NBUF * G::BUF + 2 * TOK * LDP + 2 * TOK * LDT) * (int)sizeof(T) + NKEY * 4 + NRED * 4;
};
template <typename T> __device__ __forceinline__ void unpackB(const typename Mma<T>::Frag& f, float* v) {
  if constexpr (sizeof(T) == 4) { v[0] = f[0]; v[1] = f[1]; v[2] = f[2]; v[3] = f[3]; }
  else {
    typedef __attribute__((ext_vector_type(4))) uint32_t u4;
    const u4 w = __builtin_bit_cast(u4, f);
#pragma unroll
    for (int e = 0; e < 4; ++e) unpack2<T>(w[e], v[2 * e], v[2 * e + 1]);
  }
}
template <typename T> __device__ __forceinline__ typename Mma<T>::Frag packB(const float* v) {
  if constexpr (sizeof(T) == 4) return (f32x4){v[0], v[1], v[2], v[3]};
  else {
    typedef __attribute__((ext_vector_type(4))) uint32_t u4;
    const u4 w = {pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]), pack2<T>(v[4], v[5]), pack2<T>(v[6], v[7])};
    return __builtin_bit_cast(s16x8, w);
  }
}
// column of element e of chain fragment s (k-step s) held by lane group g
template <typename T> __device__ __forceinline__ int chain_col(int s, int g, int e) {
  if constexpr (sizeof(T) == 4) return 16 * s + 4 * g + e;
  else return 32 * s + (e < 4 ? 4 * g + e : 16 + 4 * g + (e - 4));
}
// sum over the 16 token rows of the wave (lanes of equal g), then ONE LDS add by the lane of row 0
__device__ __forceinline__ void rows_to_lds(float v, float* dst, int ln) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if (ln == 0) atomicAdd(dst, v);
}

template <typename T>
__global__ __launch_bounds__(256, 1) void xattn_bwd_kernel(Args p) {
  typedef Geo<T> G;
  typedef BGeo<T> BG;
  constexpr int KSTEP = G::KSTEP, KS = G::KS, ND = Ch<T>::ND, LK = Mma<T>::LANE_K, KS1 = G::KS1, NBUF = BG::NBUF, CL = 4 * ND;
  extern __shared__ __attribute__((aligned(16))) unsigned char xa_smem[];
  T* Kt = reinterpret_cast<T*>(xa_smem);
  T* Vt = Kt + NKEY * G::LDK;
  T* buf0 = Vt + NKEY * G::LDK;
  T* PT = buf0 + NBUF * G::BUF;
  T* ST = PT + TOK * BG::LDP;
  T* OT = ST + TOK * BG::LDP;
  T* QT = OT + TOK * BG::LDT;
  int* kval = reinterpret_cast<int*>(QT + TOK * BG::LDT);
  float* red = reinterpret_cast<float*>(kval + NKEY);          // [dg1 128 | dbe1 128 | dbo 128 | dg2 384 | dbe2 384]
  float* b1s = red + 3 * O1 + 2 * CB;                          // FFN1 bias (LDS copy: no global load inside the chunk loop)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g = lane >> 4, ln = lane & 15;
  const int z = blockIdx.x % p.Z, rest = blockIdx.x / p.Z, tiles = p.HW / TOK, tile = rest % tiles, b = rest / tiles;
  const int tok = tile * TOK + wv * 16 + ln;
  const long long R = (long long)p.B * p.HW;
  const long long zrow = (long long)z * R + (long long)b * p.HW + tok;
  const long long kvrow0 = ((long long)z * p.B + b) * NKEY;
  const T* wz = reinterpret_cast<const T*>(p.pack) + (long long)z * G::STREAM;
  const long long zo = (long long)z * p.zstride;
  const bool train = p.rng != nullptr && p.p_drop > 0.f;
  const float dsc = train ? 1.0f / (1.0f - p.p_drop) : 1.f;
  const float scale = 0.15430334996209191f;

  Stage<T> stg;
  stg.template issue<G::FCH>(wz + G::OFF_F, tid);
  KvStage<T> kvs;
  kvs.issue(reinterpret_cast<const T*>(p.k), reinterpret_cast<const T*>(p.v), kvrow0, 0, tid);
  for (int i = tid; i < BG::NRED - F1; i += 256) red[i] = 0.f;
  for (int i = tid; i < F1; i += 256) b1s[i] = p.b1[zo + i];
  for (int i = tid; i < 2 * NKEY * (G::LDK - HS); i += 256) {
    const int t = i / (NKEY * (G::LDK - HS)), rem = i % (NKEY * (G::LDK - HS));
    stf((t ? Vt : Kt) + (rem / (G::LDK - HS)) * G::LDK + HS + rem % (G::LDK - HS), 0.f);
  }
  if (tid < NKEY) kval[tid] = p.kvalid ? p.kvalid[(long long)b * NKEY + tid] : 1;
  __syncthreads();                                   // red[] zeroed before the first LDS adds

  // ---- LayerNorm-2 backward on the token's row held as natural B fragments (k = 384 output columns); du2 = d(FFN2 output + b2)
  typename Mma<T>::Frag db[KS];                      // dy on the way in, the masked du2 (the B operand of dhd^T = W2 du2^T) on the way out
  {
    typename Mma<T>::Frag ub[KS];
    const T* up = reinterpret_cast<const T*>(p.su2) + zrow * CB + LK * g;
    const T* dp = reinterpret_cast<const T*>(p.dy) + zrow * CB + LK * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { ub[ks] = Mma<T>::from_global(up + ks * KSTEP); db[ks] = Mma<T>::from_global(dp + ks * KSTEP); }
    float s = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      float v[LK];
      unpackB<T>(ub[ks], v);
#pragma unroll
      for (int e = 0; e < LK; ++e) s += v[e];
    }
    s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
    const float mean = s * (1.f / CB);
    float q = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      float v[LK];
      unpackB<T>(ub[ks], v);
#pragma unroll
      for (int e = 0; e < LK; ++e) { const float d = v[e] - mean; q += d * d; }
    }
    q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
    const float rstd = rsqrtf(q * (1.f / CB) + 1e-3f);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      float u[LK], d[LK];
      unpackB<T>(ub[ks], u); unpackB<T>(db[ks], d);
      const float* gm = p.g2 + zo + ks * KSTEP + LK * g;
#pragma unroll
      for (int e = 0; e < LK; ++e) { const float a = d[e] * gm[e]; s1 += a; s2 += a * (u[e] - mean) * rstd; }
    }
    s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
    s1 *= (1.f / CB); s2 *= (1.f / CB);
    T* du2o = reinterpret_cast<T*>(p.du2) + zrow * CB + LK * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      float u[LK], d[LK], o[LK];
      unpackB<T>(ub[ks], u); unpackB<T>(db[ks], d);
      const int c0 = ks * KSTEP + LK * g;
      const float* gm = p.g2 + zo + c0;
#pragma unroll
      for (int q4 = 0; q4 < LK / 4; ++q4) {
        float fk[4] = {1.f, 1.f, 1.f, 1.f};
        if (train) keep_scale(p.rng, p.site_2, zrow * CB + c0 + 4 * q4, p.p_drop, dsc, fk);
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
          const int e = 4 * q4 + e4;
          const float xh = (u[e] - mean) * rstd;
          o[e] = rstd * (d[e] * gm[e] - s1 - xh * s2) * fk[e4];
          rows_to_lds(d[e] * xh, red + 3 * O1 + c0 + e, ln);
          rows_to_lds(d[e], red + 3 * O1 + CB + c0 + e, ln);
        }
      }
      db[ks] = packB<T>(o);
      *reinterpret_cast<typename Mma<T>::Frag*>(du2o + ks * KSTEP) = db[ks];
    }
  }

  // ---- n1 = LN1(v1) recomputed from the saved v1, held as chained B fragments; written once for dW1 = n1^T dpre
  typename Ch<T>::Frag n1c[KS1];
  float mean1, rstd1;
  {
    const T* v1p = reinterpret_cast<const T*>(p.sv1) + zrow * O1;
    float s = 0.f;
#pragma unroll
    for (int sI = 0; sI < KS1; ++sI) {
      n1c[sI] = Ch<T>::ldB_row(v1p, sI * KSTEP, lane);
      float v[CL];
      unpackB<T>(n1c[sI], v);
#pragma unroll
      for (int e = 0; e < CL; ++e) s += v[e];
    }
    s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
    mean1 = s * (1.f / O1);
    float q = 0.f;
#pragma unroll
    for (int sI = 0; sI < KS1; ++sI) {
      float v[CL];
      unpackB<T>(n1c[sI], v);
#pragma unroll
      for (int e = 0; e < CL; ++e) { const float d = v[e] - mean1; q += d * d; }
    }
    q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
    rstd1 = rsqrtf(q * (1.f / O1) + 1e-3f);
    T* n1o = reinterpret_cast<T*>(p.n1) + zrow * O1;
#pragma unroll
    for (int sI = 0; sI < KS1; ++sI) {
      float v[CL];
      unpackB<T>(n1c[sI], v);
#pragma unroll
      for (int e = 0; e < CL; ++e) { const int col = chain_col<T>(sI, g, e); v[e] = (v[e] - mean1) * rstd1 * p.g1[zo + col] + p.be1[zo + col]; }
      n1c[sI] = packB<T>(v);
#pragma unroll
      for (int h2 = 0; h2 < ND; ++h2) st4(n1o + chain_col<T>(sI, g, 4 * h2), v + 4 * h2);
    }
  }

  // ---- FFN backward, hidden chunk by hidden chunk: pre^T recomputed, dhd^T = W2 du2^T, dpre chained into dn1^T += W1 dpre^T
  f32x4 dn1[O1 / 16];
#pragma unroll
  for (int f = 0; f < O1 / 16; ++f) dn1[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int cur = 0;
  T* hdo = reinterpret_cast<T*>(p.hd) + zrow * F1 + 4 * g;
  T* dpo = reinterpret_cast<T*>(p.dpre) + zrow * F1 + 4 * g;
#pragma unroll 1
  for (int c = 0; c < G::NFC; ++c) {
    T* W1s = buf0 + cur * G::BUF;
    T* W2s = W1s + O1 * G::LD1;
    if (NBUF == 1) __syncthreads();
    stg.template commit<G::FCH>(W1s, tid);
    if (c == 0) kvs.commit(Kt, Vt, tid);             // head 0's keys / values (first used in the attention phase, many barriers later)
    __syncthreads();
    if (c + 1 < G::NFC) stg.template issue<G::FCH>(wz + G::OFF_F + (long long)(c + 1) * G::FCH, tid);
    else stg.template issue<G::OCH>(wz + G::OFF_O, tid);
    f32x4 a1[ND], a3[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) {
      const float4 bv = *reinterpret_cast<const float4*>(b1s + c * G::HC + 16 * d + 4 * g);
      a1[d] = (f32x4){bv.x, bv.y, bv.z, bv.w};
      a3[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
      for (int d = 0; d < ND; ++d) a1[d] = Mma<T>::mma(Ch<T>::ldA_tr(W1s, G::LD1, 16 * d, ks * KSTEP, lane), n1c[ks], a1[d]);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int d = 0; d < ND; ++d) a3[d] = Mma<T>::mma(Mma<T>::load(W2s, G::LD2, 16 * d, ks * KSTEP, lane), db[ks], a3[d]);
#pragma unroll
    for (int d = 0; d < ND; ++d) {
      float fk[4] = {1.f, 1.f, 1.f, 1.f};
      if (train) keep_scale(p.rng, p.site_1, zrow * F1 + c * G::HC + 16 * d + 4 * g, p.p_drop, dsc, fk);
      float hv[4], gv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pre = a1[d][r], h = elu_t<T>(pre);
        hv[r] = h * fk[r];
        gv[r] = a3[d][r] * fk[r] * (pre > 0.f ? 1.f : h + 1.f);          // ELU'(x) = exp(x) = elu(x) + 1 for x <= 0
        a3[d][r] = gv[r];
      }
      st4(hdo + c * G::HC + 16 * d, hv);
      st4(dpo + c * G::HC + 16 * d, gv);
    }
    const typename Ch<T>::Frag df = Ch<T>::from_acc(a3);
#pragma unroll
    for (int f = 0; f < O1 / 16; ++f) dn1[f] = Mma<T>::mma(Ch<T>::ldA(W1s, G::LD1, 16 * f, 0, lane), df, dn1[f]);
    cur ^= (NBUF - 1);
  }

  // ---- LayerNorm-1 backward on the accumulator layout; dv1 = d(O Wo + bo): written once (dWo = O^T dv1), chained into dO^T = Wo dv1^T
  typename Ch<T>::Frag dv1c[KS1];
  {
    const T* v1p = reinterpret_cast<const T*>(p.sv1) + zrow * O1 + 4 * g;
    float xh[O1 / 16][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int f = 0; f < O1 / 16; ++f) {
      float xv[4];
      ld4(v1p + 16 * f, xv);
      const float4 gm = *reinterpret_cast<const float4*>(p.g1 + zo + 16 * f + 4 * g);
      const float gmv[4] = {gm.x, gm.y, gm.z, gm.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        xh[f][r] = (xv[r] - mean1) * rstd1;
        const float d = dn1[f][r];
        rows_to_lds(d * xh[f][r], red + 16 * f + 4 * g + r, ln);
        rows_to_lds(d, red + O1 + 16 * f + 4 * g + r, ln);
        const float a = d * gmv[r];
        dn1[f][r] = a;
        s1 += a; s2 += a * xh[f][r];
      }
    }
    s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
    s1 *= (1.f / O1); s2 *= (1.f / O1);
    T* dv1o = reinterpret_cast<T*>(p.dv1) + zrow * O1 + 4 * g;
#pragma unroll
    for (int f = 0; f < O1 / 16; ++f) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = rstd1 * (dn1[f][r] - s1 - xh[f][r] * s2);
        dn1[f][r] = v[r];
        rows_to_lds(v[r], red + 2 * O1 + 16 * f + 4 * g + r, ln);
      }
      st4(dv1o + 16 * f, v);
    }
#pragma unroll
    for (int sI = 0; sI < KS1; ++sI) dv1c[sI] = Ch<T>::from_acc(&dn1[sI * ND]);
  }

  // ---- dO^T = Wo dv1^T, head by head (one Wo chunk per head)
  f32x4 dO[NH][3];
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    T* W = buf0 + cur * G::BUF;
    if (NBUF == 1) __syncthreads();
    stg.template commit<G::OCH>(W, tid);
    __syncthreads();
    if (h + 1 < NH) stg.template issue<G::OCH>(wz + G::OFF_O + (h + 1) * G::OCH, tid);
    else stg.template issue<G::QCH>(wz + G::OFF_Q, tid);
#pragma unroll
    for (int jd = 0; jd < 3; ++jd) {
      dO[h][jd] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int sI = 0; sI < KS1; ++sI) dO[h][jd] = Mma<T>::mma(Ch<T>::ldA(W, G::LDO, 16 * jd, sI * KSTEP, lane), dv1c[sI], dO[h][jd]);
    }
    cur ^= (NBUF - 1);
  }

  // ---- attention backward head by head, then dquery^T += Wq dq^T
  f32x4 dqy[CB / 16];
#pragma unroll
  for (int f = 0; f < CB / 16; ++f) dqy[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const long long pbase = ((((long long)z * p.B + b) * tiles + tile) * NKEY + 16 * wv + 4 * g) * G::QS + ln;
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    if (h + 1 < NH) kvs.issue(reinterpret_cast<const T*>(p.k), reinterpret_cast<const T*>(p.v), kvrow0, h + 1, tid);
    HeadOp<T> qop;
    qop.from_row(reinterpret_cast<const T*>(p.sq) + zrow * G::QS + HP * h, lane);
    f32x4 st[4], dP[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) st[j] = k48_rows<T>(Kt, G::LDK, 16 * j, qop, lane, (f32x4){0.f, 0.f, 0.f, 0.f});
    softmax_keys<T>(st, kval, scale, g);
    {
      HeadOp<T> dop;
      dop.from_acc(dO[h]);
#pragma unroll
      for (int j = 0; j < 4; ++j) dP[j] = k48_rows<T>(Vt, G::LDK, 16 * j, dop, lane, (f32x4){0.f, 0.f, 0.f, 0.f});
    }
    T* prow = PT + (16 * wv + ln) * BG::LDP + 4 * g;
    T* srow = ST + (16 * wv + ln) * BG::LDP + 4 * g;
    float t = 0.f;
    const long long base = ((((long long)z * p.B + b) * NH + h) * p.HW + tok) * NKEY + 4 * g;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float fk[4] = {1.f, 1.f, 1.f, 1.f};
      if (train) keep_scale(p.rng, p.site_a, base + 16 * j, p.p_drop, dsc, fk);
      float pd[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { pd[r] = st[j][r] * fk[r]; dP[j][r] *= fk[r]; t += st[j][r] * dP[j][r]; }
      st4(prow + 16 * j, pd);                          // Pd -> tile [token][key]
    }
    t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { ds[r] = st[j][r] * (dP[j][r] - t) * scale; dP[j][r] = ds[r]; }
      st4(srow + 16 * j, ds);                          // dS (already times 42^-1/2) -> tile [token][key]
    }
    {   // dO and q of the wave's tokens -> tiles [token][head column]
      T* orow = OT + (16 * wv + ln) * BG::LDT + 4 * g;
      T* qrow = QT + (16 * wv + ln) * BG::LDT + 4 * g;
      const T* qsrc = reinterpret_cast<const T*>(p.sq) + zrow * G::QS + HP * h + 4 * g;
#pragma unroll
      for (int jd = 0; jd < 3; ++jd) {
        const float v[4] = {dO[h][jd][0], dO[h][jd][1], dO[h][jd][2], dO[h][jd][3]};
        st4(orow + 16 * jd, v);
        float qv[4];
        ld4(qsrc + 16 * jd, qv);
        st4(qrow + 16 * jd, qv);
      }
    }
    f32x4 dq[3];
#pragma unroll
    for (int jd = 0; jd < 3; ++jd) dq[jd] = (f32x4){0.f, 0.f, 0.f, 0.f};
    keys_contract<T>(Kt, dP, dq, lane);               // dq^T = K^T dS^T
    {
      T* dqo = reinterpret_cast<T*>(p.dq) + zrow * G::QS + HP * h + 4 * g;
#pragma unroll
      for (int jd = 0; jd < 3; ++jd) { const float v[4] = {dq[jd][0], dq[jd][1], dq[jd][2], dq[jd][3]}; st4(dqo + 16 * jd, v); }
    }
    __syncthreads();                                  // tiles complete; every wave is done with this head's K / V tiles
    if (h + 1 < NH) kvs.commit(Kt, Vt, tid);          // (made visible by the chunk barriers below)
    {   // dV = Pd^T dO, dK = dS^T q for the 16 keys this wave owns: m = key, n = head column, k = the workgroup's 64 tokens
      f32x4 dv[3], dk[3];
#pragma unroll
      for (int jd = 0; jd < 3; ++jd) { dv[jd] = (f32x4){0.f, 0.f, 0.f, 0.f}; dk[jd] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int sI = 0; sI < TOK / KSTEP; ++sI) {
        const typename Mma<T>::Frag ap = Mma<T>::load_tr(PT, BG::LDP, 16 * wv, sI * KSTEP, lane);
        const typename Mma<T>::Frag as = Mma<T>::load_tr(ST, BG::LDP, 16 * wv, sI * KSTEP, lane);
#pragma unroll
        for (int jd = 0; jd < 3; ++jd) {
          dv[jd] = Mma<T>::mma(ap, Mma<T>::load_tr(OT, BG::LDT, 16 * jd, sI * KSTEP, lane), dv[jd]);
          dk[jd] = Mma<T>::mma(as, Mma<T>::load_tr(QT, BG::LDT, 16 * jd, sI * KSTEP, lane), dk[jd]);
        }
      }
#pragma unroll
      for (int jd = 0; jd < 3; ++jd)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          p.dvp[pbase + (long long)r * G::QS + HP * h + 16 * jd] = dv[jd][r];
          p.dkp[pbase + (long long)r * G::QS + HP * h + 16 * jd] = dk[jd][r];
        }
    }
    HeadOp<T> dqop;
    dqop.from_acc(dq);
#pragma unroll
    for (int c = 0; c < G::NQC; ++c) {
      T* W = buf0 + cur * G::BUF;
      if (NBUF == 1) __syncthreads();
      stg.template commit<G::QCH>(W, tid);
      __syncthreads();
      if (c + 1 < G::NQC) stg.template issue<G::QCH>(wz + G::OFF_Q + (h * G::NQC + c + 1) * G::QCH, tid);
      else if (h + 1 < NH) stg.template issue<G::QCH>(wz + G::OFF_Q + (h + 1) * G::NQC * G::QCH, tid);
#pragma unroll
      for (int fr = 0; fr < G::KQ / 16; ++fr)
        dqy[c * (G::KQ / 16) + fr] = k48_rows<T>(W, G::LDQ, 16 * fr, dqop, lane, dqy[c * (G::KQ / 16) + fr]);
      cur ^= (NBUF - 1);
    }
  }

  // ---- dquery = dy (the `+ query` residual) + the attention path
  {
    const T* dyr = reinterpret_cast<const T*>(p.dy) + zrow * CB + 4 * g;
    T* dqr = reinterpret_cast<T*>(p.dquery) + zrow * CB + 4 * g;
    float dv[CB / 16][4];         // read before the first store (dquery may alias dy: see xattn_fwd_kernel)
#pragma unroll
    for (int f = 0; f < CB / 16; ++f) ld4(dyr + 16 * f, dv[f]);
#pragma unroll
    for (int f = 0; f < CB / 16; ++f) {
      const float v[4] = {dv[f][0] + dqy[f][0], dv[f][1] + dqy[f][1], dv[f][2] + dqy[f][2], dv[f][3] + dqy[f][3]};
      st4(dqr + 16 * f, v);
    }
  }
  __syncthreads();
  for (int i = tid; i < O1; i += 256) {
    atomicAdd(p.dg1 + zo + i, red[i]); atomicAdd(p.dbe1 + zo + i, red[O1 + i]); atomicAdd(p.dbo + zo + i, red[2 * O1 + i]);
  }
  for (int i = tid; i < CB; i += 256) { atomicAdd(p.dg2 + zo + i, red[3 * O1 + i]); atomicAdd(p.dbe2 + zo + i, red[3 * O1 + CB + i]); }
}

// dk / dv [Z*B, 64, 3 x 42] (activation dtype) = sum over the token tiles of the per-tile partials [Z*B][tiles][64][3 x 48] f32
template <typename T>
__global__ __launch_bounds__(256) void xattn_dkv_reduce_kernel(const float* dkp, const float* dvp, T* dk, T* dv, long long zb, int tiles) {
  const long long total = zb * NKEY * NH * HS;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < 2 * total; i += gridDim.x * 256ll) {
    const bool second = i >= total;
    const long long j = second ? i - total : i;
    const int d = (int)(j % HS), h = (int)((j / HS) % NH), key = (int)((j / (HS * NH)) % NKEY);
    const long long s = j / ((long long)HS * NH * NKEY);
    const float* src = (second ? dvp : dkp) + ((s * tiles) * NKEY + key) * (NH * HP) + HP * h + d;
    float a = 0.f;
    for (int t = 0; t < tiles; ++t) a += src[(long long)t * NKEY * NH * HP];
    stf((second ? dv : dk) + j, a);
  }
}

template <typename T> static int launch_bwd(const Args& a, hipStream_t st) {
  static PerDevice<bool> attr;
  const int lds = BGeo<T>::LDS_BYTES;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)xattn_bwd_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
      stj_set_error("xattn_bwd: cannot reserve %d bytes of LDS", lds); return STJ_ELAUNCH;
    }
    attr = true;
  }
  hipLaunchKernelGGL((xattn_bwd_kernel<T>), dim3((unsigned)(a.Z * a.B * (a.HW / TOK))), dim3(256), lds, st, a);
  return stj_check_launch("stj_xattn_bwd");
}

template <typename T> static int lds_bytes_fwd() { return (2 * NKEY * Geo<T>::LDK + 2 * Geo<T>::BUFE) * (int)sizeof(T) + NKEY * 4 + F1 * 4; }

template <typename T> static int launch_fwd(const Args& a, hipStream_t st) {
  static PerDevice<bool> attr;
  const int lds = lds_bytes_fwd<T>();
  if (!attr) {
    if (hipFuncSetAttribute((const void*)xattn_fwd_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
      stj_set_error("xattn_fwd: cannot reserve %d bytes of LDS", lds); return STJ_ELAUNCH;
    }
    attr = true;
  }
  hipLaunchKernelGGL((xattn_fwd_kernel<T>), dim3((unsigned)(a.Z * a.B * (a.HW / TOK))), dim3(256), lds, st, a);
  return stj_check_launch("stj_xattn_fwd");
}
}  // namespace xat

// bytes of one set's packed weight stream (stj_xattn_pack writes Z of them back to back)
extern "C" long long stj_xattn_pack_workspace_bytes(int dtype) {
  return stj_is16(dtype) ? (long long)xat::Geo<bf16>::STREAM * 2 : (long long)xat::Geo<float>::STREAM * 4;
}
// bytes the caller adds ONCE behind the Z streams: the kernels copy fixed-size pieces and read this far past the last chunk
extern "C" long long stj_xattn_pack_tail_workspace_bytes(int dtype) {
  return stj_is16(dtype) ? (long long)xat::Geo<bf16>::BUFE * 2 : (long long)xat::Geo<float>::BUFE * 4;
}
// wq [3,384,42], wo [3,42,128], w1 [128,512], w2 [512,384] of set 0 (f32 masters; set z lies zstride elements further) -> pack
extern "C" int stj_xattn_pack(const float* wq, const float* wo, const float* w1, const float* w2, long long zstride, int Z, void* pack, int dtype,
                              hipStream_t stream) {
  if (Z <= 0) return STJ_OK;
  if (!stj_dtype_ok(dtype)) { stj_set_error("xattn_pack: bad dtype %d", dtype); return STJ_EINVAL; }
  if (((uintptr_t)pack) & 15) { stj_set_error("xattn_pack: pack must be 16-byte aligned"); return STJ_EINVAL; }
  const dim3 grid(128, (unsigned)Z);
  if (dtype == STJ_BF16) hipLaunchKernelGGL(xat::xattn_pack_kernel<bf16>, grid, dim3(256), 0, stream, wq, wo, w1, w2, zstride, (bf16*)pack);
  else if (dtype == STJ_F16) hipLaunchKernelGGL(xat::xattn_pack_kernel<f16>, grid, dim3(256), 0, stream, wq, wo, w1, w2, zstride, (f16*)pack);
  else hipLaunchKernelGGL(xat::xattn_pack_kernel<float>, grid, dim3(256), 0, stream, wq, wo, w1, w2, zstride, (float*)pack);
  return stj_check_launch("stj_xattn_pack");
}

// y [Z,B,HW,384] = LN(FFN(LN(MHA(query, k, v)))) + query for Z weight sets.  query [Z,B,HW,384]; k, v [Z,B,64,126] (projected keys /
// values); kvalid [B,64] int32 or NULL; pack from stj_xattn_pack; bo / g1 / be1 [128], b1 [512], b2 / g2 / be2 [384]: f32 vectors
// of set 0 (set z at + z * zstride).  Training (rng != NULL): the three dropout sites draw from the Philox stream with the
// unfused layouts ([Z,B,3,HW,64], [Z,B*HW,512], [Z,B*HW,384]); sq / so [Z,B,HW,144], sv1 [.,128], su2 [.,384] (all or none)
// receive what the backward kernel reads.
extern "C" int stj_xattn_fwd(const void* query, const void* k, const void* v, const int* kvalid, const void* pack, const float* bo,
                             const float* g1, const float* be1, const float* b1, const float* b2, const float* g2, const float* be2,
                             long long zstride, void* y, void* sq, void* so, void* sv1, void* su2, int Z, int B, int HW,
                             const long long* rng_state, int site_a, int site_1, int site_2, float p_drop, int dtype, hipStream_t stream) {
  if (Z <= 0 || B <= 0) return STJ_OK;
  if (HW <= 0 || HW % xat::TOK) { stj_set_error("xattn: HW must be a multiple of %d (got %d)", xat::TOK, HW); return STJ_EINVAL; }
  if (!(p_drop >= 0.f && p_drop < 1.f)) { stj_set_error("xattn: need 0 <= p_drop < 1"); return STJ_EINVAL; }
  if ((sq || so || sv1 || su2) && !(sq && so && sv1 && su2)) { stj_set_error("xattn: training outputs come all or none"); return STJ_EINVAL; }
  if ((((uintptr_t)query) | ((uintptr_t)pack) | ((uintptr_t)y)) & 15) { stj_set_error("xattn: query / pack / y must be 16-byte aligned"); return STJ_EINVAL; }
  xat::Args a = {};
  a.query = query; a.k = k; a.v = v; a.kvalid = kvalid; a.pack = pack; a.bo = bo; a.g1 = g1; a.be1 = be1; a.b1 = b1; a.b2 = b2; a.g2 = g2;
  a.be2 = be2; a.zstride = zstride; a.y = y; a.sq = sq; a.so = so; a.sv1 = sv1; a.su2 = su2; a.Z = Z; a.B = B; a.HW = HW;
  a.rng = rng_state; a.site_a = site_a; a.site_1 = site_1; a.site_2 = site_2; a.p_drop = p_drop;
  if (dtype == STJ_BF16) return xat::launch_fwd<bf16>(a, stream);
  if (dtype == STJ_F16) return xat::launch_fwd<f16>(a, stream);
  if (dtype == STJ_F32) return xat::launch_fwd<float>(a, stream);
  stj_set_error("xattn: bad dtype %d", dtype);
  return STJ_EINVAL;
}

// floats of the dk / dv partial-sum workspace of stj_xattn_bwd (each of dkp, dvp)
extern "C" long long stj_xattn_bwd_workspace_bytes(int Z, int B, int HW) {
  return (long long)Z * B * (HW / xat::TOK) * xat::NKEY * xat::NH * xat::HP * 4;
}
// Backward of stj_xattn_fwd (see the kernel header).  dy, dquery [Z,B,HW,384]; sq, so, sv1, su2: the forward's saves; dkp / dvp:
// workspaces of stj_xattn_bwd_workspace_bytes (written, then reduced into dk / dv [Z,B,64,126] by a second small launch);
// hd, dpre [.,512], du2 [.,384], n1, dv1 [.,128], dq [.,144]: written (operands of the caller's weight-gradient GEMMs);
// dg1 / dbe1 / dbo [128], dg2 / dbe2 [384]: "+=" into set z at + z * zstride.
extern "C" int stj_xattn_bwd(const void* dy, const void* query, const void* k, const void* v, const int* kvalid, const void* pack,
                             const float* g1, const float* be1, const float* b1, const float* g2, long long zstride, const void* sq,
                             const void* sv1, const void* su2, void* dquery, void* dk, void* dv, float* dkp, float* dvp, void* hd,
                             void* dpre, void* du2, void* n1, void* dv1, void* dq, float* dg1, float* dbe1, float* dbo, float* dg2,
                             float* dbe2, int Z, int B, int HW, const long long* rng_state, int site_a, int site_1, int site_2,
                             float p_drop, int dtype, hipStream_t stream) {
  if (Z <= 0 || B <= 0) return STJ_OK;
  if (HW <= 0 || HW % xat::TOK) { stj_set_error("xattn: HW must be a multiple of %d (got %d)", xat::TOK, HW); return STJ_EINVAL; }
  if (!(p_drop >= 0.f && p_drop < 1.f)) { stj_set_error("xattn: need 0 <= p_drop < 1"); return STJ_EINVAL; }
  if ((((uintptr_t)query) | ((uintptr_t)pack) | ((uintptr_t)dy) | ((uintptr_t)dquery) | ((uintptr_t)su2) | ((uintptr_t)du2)) & 15) {
    stj_set_error("xattn_bwd: dy / query / pack / dquery / su2 / du2 must be 16-byte aligned"); return STJ_EINVAL;
  }
  xat::Args a = {};
  a.dy = dy; a.query = query; a.k = k; a.v = v; a.kvalid = kvalid; a.pack = pack; a.g1 = g1; a.be1 = be1; a.b1 = b1; a.g2 = g2;
  a.zstride = zstride; a.sq = const_cast<void*>(sq); a.sv1 = const_cast<void*>(sv1); a.su2 = const_cast<void*>(su2);
  a.dquery = dquery; a.dkp = dkp; a.dvp = dvp; a.hd = hd; a.dpre = dpre; a.du2 = du2; a.n1 = n1; a.dv1 = dv1; a.dq = dq;
  a.dg1 = dg1; a.dbe1 = dbe1; a.dbo = dbo; a.dg2 = dg2; a.dbe2 = dbe2; a.Z = Z; a.B = B; a.HW = HW;
  a.rng = rng_state; a.site_a = site_a; a.site_1 = site_1; a.site_2 = site_2; a.p_drop = p_drop;
  int rc;
  if (dtype == STJ_BF16) rc = xat::launch_bwd<bf16>(a, stream);
  else if (dtype == STJ_F16) rc = xat::launch_bwd<f16>(a, stream);
  else if (dtype == STJ_F32) rc = xat::launch_bwd<float>(a, stream);
  else { stj_set_error("xattn: bad dtype %d", dtype); return STJ_EINVAL; }
  if (rc != STJ_OK) return rc;
  const long long zb = (long long)Z * B, n = 2 * zb * xat::NKEY * xat::NH * xat::HS;
  const int grid = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
  const int tiles = HW / xat::TOK;
  if (dtype == STJ_BF16) hipLaunchKernelGGL(xat::xattn_dkv_reduce_kernel<bf16>, dim3(grid), dim3(256), 0, stream, dkp, dvp, (bf16*)dk, (bf16*)dv, zb, tiles);
  else if (dtype == STJ_F16) hipLaunchKernelGGL(xat::xattn_dkv_reduce_kernel<f16>, dim3(grid), dim3(256), 0, stream, dkp, dvp, (f16*)dk, (f16*)dv, zb, tiles);
  else hipLaunchKernelGGL(xat::xattn_dkv_reduce_kernel<float>, dim3(grid), dim3(256), 0, stream, dkp, dvp, (float*)dk, (float*)dv, zb, tiles);
  return stj_check_launch("stj_xattn_dkv_reduce");
}
