// Row softmax (fwd/bwd) for the global attentions (tfa MultiHeadAttention restatement and FG-MSA), and the
// FG-MSA bilinearly-sampled relative-position bias (fwd/bwd).
//   tfa MHA logits/mask/softmax: tensorflow_addons MultiHeadAttention as called at reference trajNet.py:33,42
//   (self-attn), :71,80 and :195,225 (cross) -- logits += -10e9*(1-mask) in f32, softmax over keys.
//   FG-MSA: reference FG_MSA.py:147-174 (attn*scale + bias -> softmax) and :150-172 (rpe bias through
//   occu_metric.sample = zero-pad + clamped bilinear, occu_metric.py:345-409 / tfa_image.py:87-173).
// Q K^T, P V and their gradients run on the MFMA batched GEMM (gemm.hip); these kernels are the HBM-bound
// row passes in between.  One wavefront per row, reductions by wave shuffles.
#include "common.h"

// S: f32 logits [rows][Nk] (already scaled); rows = batch*H*Nq.  P (type T) out.
template <typename T>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* S, T* P, const int* qvalid, const int* kvalid,
                                                          const float* bias, long long rows, int H, int Nq, int Nk) {
  const int lane = threadIdx.x & 63;
  for (long long row = blockIdx.x * 4ll + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4ll) {
    const int q = (int)(row % Nq);
    const long long b = row / ((long long)Nq * H);
    const bool qv = qvalid ? qvalid[b * Nq + q] != 0 : true;
    float v[4];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = lane + 64 * j;
      if (k < Nk) {
        float x = S[row * Nk + k];
        if (bias) x += bias[row * Nk + k];
        const bool ok = qv && (kvalid ? kvalid[b * Nk + k] != 0 : true);
        if (!ok) x = x + (-10e9f);            // f32 add, exactly as the reference (absorbs the logit)
        v[j] = x;
        m = fmaxf(m, x);
      } else v[j] = -INFINITY;
    }
    m = wave_max(m);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = (lane + 64 * j < Nk) ? expf(v[j] - m) : 0.f; s += v[j]; }
    s = wave_sum(s);
    const float inv = 1.f / s;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int k = lane + 64 * j; if (k < Nk) stf(P + row * Nk + k, v[j] * inv); }
  }
}

// dS = P * (dP - sum_k P dP)
template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const T* P, const float* dP, T* dS, long long rows, int Nk) {
  const int lane = threadIdx.x & 63;
  for (long long row = blockIdx.x * 4ll + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4ll) {
    float p[4], d[4];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = lane + 64 * j;
      p[j] = k < Nk ? ldf(P + row * Nk + k) : 0.f;
      d[j] = k < Nk ? dP[row * Nk + k] : 0.f;
      s += p[j] * d[j];
    }
    s = wave_sum(s);
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int k = lane + 64 * j; if (k < Nk) stf(dS + row * Nk + k, p[j] * (d[j] - s)); }
  }
}

extern "C" int stj_softmax_fwd(const float* S, void* P, const int* qvalid, const int* kvalid, const float* bias,
                               long long batch, int H, int Nq, int Nk, int dtype, hipStream_t stream) {
  if (Nk > 256 || Nk <= 0) { stj_set_error("softmax: Nk=%d unsupported (1..256)", Nk); return STJ_EUNSUPPORTED; }
  const long long rows = batch * H * Nq;
  if (rows <= 0) return STJ_OK;
  const int grid = (int)((rows + 3) / 4 > 8192 ? 8192 : (rows + 3) / 4);
  if (dtype == STJ_BF16) hipLaunchKernelGGL(softmax_fwd_kernel<bf16>, dim3(grid), dim3(256), 0, stream, S, (bf16*)P, qvalid, kvalid, bias, rows, H, Nq, Nk);
  else if (dtype == STJ_F16) hipLaunchKernelGGL(softmax_fwd_kernel<f16>, dim3(grid), dim3(256), 0, stream, S, (f16*)P, qvalid, kvalid, bias, rows, H, Nq, Nk);
  else hipLaunchKernelGGL(softmax_fwd_kernel<float>, dim3(grid), dim3(256), 0, stream, S, (float*)P, qvalid, kvalid, bias, rows, H, Nq, Nk);
  return stj_check_launch("stj_softmax_fwd");
}
extern "C" int stj_softmax_bwd(const void* P, const float* dP, void* dS, long long rows, int Nk, int dtype, hipStream_t stream) {
  if (Nk > 256 || Nk <= 0) { stj_set_error("softmax: Nk=%d unsupported (1..256)", Nk); return STJ_EUNSUPPORTED; }
  if (rows <= 0) return STJ_OK;
  const int grid = (int)((rows + 3) / 4 > 8192 ? 8192 : (rows + 3) / 4);
  if (dtype == STJ_BF16) hipLaunchKernelGGL(softmax_bwd_kernel<bf16>, dim3(grid), dim3(256), 0, stream, (const bf16*)P, dP, (bf16*)dS, rows, Nk);
  else if (dtype == STJ_F16) hipLaunchKernelGGL(softmax_bwd_kernel<f16>, dim3(grid), dim3(256), 0, stream, (const f16*)P, dP, (f16*)dS, rows, Nk);
  else hipLaunchKernelGGL(softmax_bwd_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)P, dP, (float*)dS, rows, Nk);
  return stj_check_launch("stj_softmax_bwd");
}

// ---- tiny attention: at most 16 queries / keys per (batch element, head) -----------------------------------------------------------
// The TrajEncoder's self-attention over the 11 time steps of an agent (trajNet.py:33,42: 512 agents x 4 heads x 11 x 11 at B = 8) as ONE
// launch per direction: layer by layer it was Q K^T, softmax, dropout, P V -- four dependent launches (76 us, the 11 x 11 products as
// batched GEMM tiles of 64 x 64) in the chain the cross-attention waits on, and five + the dropout in backward.  One workgroup per batch
// element: q, k, v (and dO) rows in LDS, every product as plain FMAs, the probabilities rounded to the storage type where the layer-by-layer
// path stores them, the dropout draws of the [Bt,H,N,N] tensor re-derived (rng.h).  Backward recomputes P.
struct SmallAttnArgs {
  const void* q; const void* k; const void* v; const int* qvalid; const int* kvalid; void* o;
  const void* dO; void* dq; void* dk; void* dv;
  int N, H, d; float scale; const long long* rng; int site; float p_drop;
};
#include "rng.h"
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void small_attn_kernel(SmallAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sa_smem[];
  const int N = a.N, H = a.H, d = a.d, HD = H * d, NN = N * N, tid = threadIdx.x;
  const int LD = HD + 2;                                 // row stride of the LDS tiles: rows of different keys on different banks
  float* P = reinterpret_cast<float*>(sa_smem);         // [H][N][N] probabilities (values of the storage type)
  float* Pd = P + H * NN;                                // after dropout
  float* dS = Pd + H * NN;                               // backward: dP, then dS
  float* ksc = dS + H * NN;                              // keep / (1 - p) of every coefficient
  T* qs = reinterpret_cast<T*>(ksc + H * NN);
  T* ks = qs + N * LD;
  T* vs = ks + N * LD;
  T* gs = vs + N * LD;                                   // backward: dO
  const long long bt = blockIdx.x, base = bt * N * HD;
  const T* q = reinterpret_cast<const T*>(a.q) + base;
  const T* k = reinterpret_cast<const T*>(a.k) + base;
  const T* v = reinterpret_cast<const T*>(a.v) + base;
  for (int i = tid; i < N * HD; i += 256) {
    const int o = (i / HD) * LD + i % HD;
    qs[o] = q[i]; ks[o] = k[i]; vs[o] = v[i];
    if (BWD) gs[o] = reinterpret_cast<const T*>(a.dO)[base + i];
  }
  __syncthreads();
  const bool drop = a.rng != nullptr && a.p_drop > 0.f;
  const float dsc = drop ? 1.f / (1.f - a.p_drop) : 1.f;
  for (int e = tid; e < H * NN; e += 256) {
    const int h = e / NN, i = (e / N) % N, j = e % N;
    const T* qr = qs + i * LD + h * d;
    const T* kr = ks + j * LD + h * d;
    float acc = 0.f;
    for (int c = 0; c < d; ++c) acc += ldf(qr + c) * ldf(kr + c);
    float x = acc * a.scale;
    const bool ok = (a.qvalid ? a.qvalid[bt * N + i] != 0 : true) && (a.kvalid ? a.kvalid[bt * N + j] != 0 : true);
    if (!ok) x = x + (-10e9f);                           // f32 add, exactly as the reference (absorbs the logit)
    P[e] = x;
    float f = 1.f;
    if (drop) {
      const long long idx = (bt * H + h) * NN + i * N + j;
      bool k4[4];
      keep4(a.rng, a.site, idx >> 2, a.p_drop, k4);
      f = k4[idx & 3] ? dsc : 0.f;
    }
    ksc[e] = f;
    if (BWD) {                                           // dPd = dO V^T
      const T* gr = gs + i * LD + h * d;
      const T* vr = vs + j * LD + h * d;
      float g = 0.f;
      for (int c = 0; c < d; ++c) g += ldf(gr + c) * ldf(vr + c);
      dS[e] = g * f;                                     // dropout backward on the f32 product
    }
  }
  __syncthreads();
  for (int r = tid; r < H * N; r += 256) {               // softmax of row r over its N keys; dS = P (dP - sum P dP)
    float* row = P + r * N;
    float m = -INFINITY;
    for (int j = 0; j < N; ++j) m = fmaxf(m, row[j]);
    float sum = 0.f;
    for (int j = 0; j < N; ++j) { const float x = expf(row[j] - m); row[j] = x; sum += x; }
    const float inv = 1.f / sum;
    float t = 0.f;
    for (int j = 0; j < N; ++j) {
      T pr;
      stf(&pr, row[j] * inv);
      row[j] = ldf(&pr);
      T pd;
      stf(&pd, row[j] * ksc[r * N + j]);
      Pd[r * N + j] = ldf(&pd);
      if (BWD) t += row[j] * dS[r * N + j];
    }
    if (BWD)
      for (int j = 0; j < N; ++j) { T s_; stf(&s_, row[j] * (dS[r * N + j] - t)); dS[r * N + j] = ldf(&s_); }
  }
  __syncthreads();
  if (!BWD) {
    T* o = reinterpret_cast<T*>(a.o) + base;
    for (int e = tid; e < N * HD; e += 256) {
      const int i = e / HD, hc = e % HD, h = hc / d;
      const float* pr = Pd + (h * N + i) * N;
      float acc = 0.f;
      for (int j = 0; j < N; ++j) acc += pr[j] * ldf(vs + j * LD + hc);
      stf(o + e, acc);
    }
  } else {
    T* dq = reinterpret_cast<T*>(a.dq) + base;
    T* dk = reinterpret_cast<T*>(a.dk) + base;
    T* dv = reinterpret_cast<T*>(a.dv) + base;
    for (int e = tid; e < N * HD; e += 256) {
      const int i = e / HD, hc = e % HD, h = hc / d;
      float aq = 0.f, ak = 0.f, av = 0.f;
      for (int j = 0; j < N; ++j) {
        aq += dS[(h * N + i) * N + j] * ldf(ks + j * LD + hc);          // dq[i] = scale sum_j dS[i][j] k[j]
        ak += dS[(h * N + j) * N + i] * ldf(qs + j * LD + hc);          // dk[i] = scale sum_j dS[j][i] q[j]
        av += Pd[(h * N + j) * N + i] * ldf(gs + j * LD + hc);          // dv[i] = sum_j Pd[j][i] dO[j]
      }
      stf(dq + e, aq * a.scale); stf(dk + e, ak * a.scale); stf(dv + e, av);
    }
  }
}
static size_t small_attn_lds(int N, int H, int d, int dtype, bool bwd) {
  return (size_t)4 * H * N * N * 4 + (size_t)(bwd ? 4 : 3) * N * (H * d + 2) * (dtype == STJ_F32 ? 4 : 2);
}
// 1 when stj_small_attn_* take this geometry (else use stj_gemm + stj_softmax_* + stj_dropout)
extern "C" int stj_small_attn_supported(int N, int H, int d, int dtype) {
  return N >= 1 && N <= 16 && H >= 1 && H <= 16 && d >= 1 && small_attn_lds(N, H, d, dtype, true) <= 64 * 1024;
}
template <bool BWD> static int small_attn_launch(const SmallAttnArgs& a, long long Bt, int dtype, hipStream_t stream) {
  if (Bt <= 0) return STJ_OK;
  if (!stj_small_attn_supported(a.N, a.H, a.d, dtype)) { stj_set_error("small_attn: N = %d, H = %d, d = %d not supported", a.N, a.H, a.d); return STJ_EUNSUPPORTED; }
  if (!(a.p_drop >= 0.f && a.p_drop < 1.f)) { stj_set_error("small_attn: need 0 <= p_drop < 1"); return STJ_EINVAL; }
  const size_t lds = small_attn_lds(a.N, a.H, a.d, dtype, BWD);
  if (dtype == STJ_BF16) hipLaunchKernelGGL((small_attn_kernel<bf16, BWD>), dim3((unsigned)Bt), dim3(256), lds, stream, a);
  else if (dtype == STJ_F16) hipLaunchKernelGGL((small_attn_kernel<f16, BWD>), dim3((unsigned)Bt), dim3(256), lds, stream, a);
  else if (dtype == STJ_F32) hipLaunchKernelGGL((small_attn_kernel<float, BWD>), dim3((unsigned)Bt), dim3(256), lds, stream, a);
  else { stj_set_error("small_attn: bad dtype %d", dtype); return STJ_EINVAL; }
  return stj_check_launch(BWD ? "stj_small_attn_bwd" : "stj_small_attn_fwd");
}
// o [Bt,N,H*d] = dropout(softmax(scale q k^T + mask)) v per batch element and head; q, k, v [Bt,N,H*d]; qvalid / kvalid int32 [Bt,N] or NULL
// (tfa MultiHeadAttention: masked logits += -10e9); dropout on the coefficients [Bt,H,N,N] at (rng_state, site) drawn as stj_dropout draws it.
extern "C" int stj_small_attn_fwd(const void* q, const void* k, const void* v, const int* qvalid, const int* kvalid, void* o, long long Bt, int N,
                                  int H, int d, float scale, const long long* rng_state, int site, float p_drop, int dtype, hipStream_t stream) {
  SmallAttnArgs a = {};
  a.q = q; a.k = k; a.v = v; a.qvalid = qvalid; a.kvalid = kvalid; a.o = o; a.N = N; a.H = H; a.d = d; a.scale = scale;
  a.rng = rng_state; a.site = site; a.p_drop = p_drop;
  return small_attn_launch<false>(a, Bt, dtype, stream);
}
extern "C" int stj_small_attn_bwd(const void* q, const void* k, const void* v, const int* qvalid, const int* kvalid, const void* dO, void* dq,
                                  void* dk, void* dv, long long Bt, int N, int H, int d, float scale, const long long* rng_state, int site,
                                  float p_drop, int dtype, hipStream_t stream) {
  SmallAttnArgs a = {};
  a.q = q; a.k = k; a.v = v; a.qvalid = qvalid; a.kvalid = kvalid; a.dO = dO; a.dq = dq; a.dk = dk; a.dv = dv; a.N = N; a.H = H; a.d = d;
  a.scale = scale; a.rng = rng_state; a.site = site; a.p_drop = p_drop;
  return small_attn_launch<true>(a, Bt, dtype, stream);
}

// ---- FG-MSA sampled relative-position bias -------------------------------------------------------------------
// bias[b,g,q,k] = sample(table_g)(x = drow - off1[k], y = dcol - off0[k]),  q=(iq,jq), k=(ik,jk), drow=iq-ik,
// dcol=jq-jk (reference FG_MSA.py:155-166 with the 'xy' meshgrid of :96-100): x indexes the
// table's width axis, y its height axis; zero pad 1 + clamped floor/alpha as in tfa_image.py:124-139.
template <typename T>
__global__ __launch_bounds__(256) void fg_bias_fwd_kernel(const T* off, const float* table, float* bias, int B, int G, int Hh, int Ww) {
  const int HW = Hh * Ww, TH = 2 * Hh - 1, TW = 2 * Ww - 1;
  const long long total = (long long)B * G * HW * HW;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) {
    const int k = (int)(i % HW); long long t = i / HW;
    const int q = (int)(t % HW); t /= HW;
    const int g = (int)(t % G); const int b = (int)(t / G);
    const T* o = off + (((long long)b * G + g) * HW + k) * 2;
    const float off0 = ldf(o), off1 = ldf(o + 1);
    const float x = (float)(q / Ww - k / Ww) - off1 + 1.f;
    const float y = (float)(q % Ww - k % Ww) - off0 + 1.f;
    Bil c = bil_setup(x, y, TH + 2, TW + 2);
    const float* img = table + g;
    const float tl = pad_at(img, TH, TW, G, c.y0, c.x0), tr = pad_at(img, TH, TW, G, c.y0, c.x0 + 1);
    const float bl = pad_at(img, TH, TW, G, c.y0 + 1, c.x0), br = pad_at(img, TH, TW, G, c.y0 + 1, c.x0 + 1);
    const float top = c.ax * (tr - tl) + tl, bot = c.ax * (br - bl) + bl;
    bias[i] = c.ay * (bot - top) + top;
  }
}

// dbias (type T, = dS of the attention) -> dtable (f32 atomics), doff [B,G,HW,2] (f32, written).
// One workgroup per (b,g); thread = key k, loop over queries q: dbias[q][k] is read coalesced across threads, the
// per-key offset gradient accumulates in registers (no cross-lane reduction), the table gradient in an LDS copy of the
// (2H-1)x(2W-1) table that is flushed with one global atomic per entry.  (v1 swept q across lanes with stride-HW reads:
// 0.94 ms at B=8.)
template <typename T>
__global__ __launch_bounds__(256) void fg_bias_bwd_kernel(const T* off, const float* table, const T* dbias, float* dtable,
                                                          float* doff, int B, int G, int Hh, int Ww, int QS) {
  extern __shared__ float fg_lds[];
  const int HW = Hh * Ww, TH = 2 * Hh - 1, TW = 2 * Ww - 1;
  float* tbl = fg_lds;                 // table_g values
  float* dtb = fg_lds + TH * TW;       // gradient accumulator
  // block = (b, g, query slice): B*G alone is 64 blocks on 256 CUs; the QS slices of the query range add their offset
  // gradients with f32 atomics (doff is zeroed by the caller)
  const int bg = blockIdx.x / QS, qs = blockIdx.x % QS, g = bg % G;
  const int q_lo = (int)((long long)HW * qs / QS), q_hi = (int)((long long)HW * (qs + 1) / QS);
  for (int i = threadIdx.x; i < TH * TW; i += 256) { tbl[i] = table[i * G + g]; dtb[i] = 0.f; }
  __syncthreads();
  for (int k = threadIdx.x; k < HW; k += 256) {
    const long long ok = ((long long)bg * HW + k) * 2;
    const float off0 = ldf(off + ok), off1 = ldf(off + ok + 1);
    const int ik = k / Ww, jk = k % Ww;
    float d0 = 0.f, d1 = 0.f;
    for (int q = q_lo; q < q_hi; ++q) {
      const float go = ldf(dbias + ((long long)bg * HW + q) * HW + k);
      const float x = (float)(q / Ww - ik) - off1 + 1.f;
      const float y = (float)(q % Ww - jk) - off0 + 1.f;
      Bil c = bil_setup(x, y, TH + 2, TW + 2);
      const float tl = pad_at(tbl, TH, TW, 1, c.y0, c.x0), tr = pad_at(tbl, TH, TW, 1, c.y0, c.x0 + 1);
      const float bl = pad_at(tbl, TH, TW, 1, c.y0 + 1, c.x0), br = pad_at(tbl, TH, TW, 1, c.y0 + 1, c.x0 + 1);
      const float top = c.ax * (tr - tl) + tl, bot = c.ax * (br - bl) + bl;
      if (c.gy) d0 -= go * (bot - top);
      if (c.gx) d1 -= go * (c.ay * (br - bl) + (1.f - c.ay) * (tr - tl));
      if (go != 0.f) {
        const float w[4] = {(1.f - c.ay) * (1.f - c.ax), (1.f - c.ay) * c.ax, c.ay * (1.f - c.ax), c.ay * c.ax};
        const int yy[4] = {c.y0, c.y0, c.y0 + 1, c.y0 + 1}, xx[4] = {c.x0, c.x0 + 1, c.x0, c.x0 + 1};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (yy[e] >= 1 && yy[e] <= TH && xx[e] >= 1 && xx[e] <= TW && w[e] != 0.f)
            atomicAdd(&dtb[(yy[e] - 1) * TW + (xx[e] - 1)], go * w[e]);
      }
    }
    if (QS == 1) { doff[ok] = d0; doff[ok + 1] = d1; }
    else { atomicAdd(doff + ok, d0); atomicAdd(doff + ok + 1, d1); }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < TH * TW; i += 256)
    if (dtb[i] != 0.f) atomicAdd(dtable + i * G + g, dtb[i]);
}

// ---- FG-MSA offset head ------------------------------------------------------------------------------------------
// off[b,g,hw,:] = tanh(o[b,hw,g,:] . W1) * scale            (1x1 conv gc -> 2 without bias, FG_MSA.py:136-142)
// fh [b,g,hw,:] = off[b,g,hw,:] . W2 + b2                   (1x1 conv 2 -> C2, FG_MSA.py:143-146; optional)
// Both products have one dimension of 2: as GEMM launches they are tile padding (2 of 64 columns live) plus a regrouped copy of o,
// a tanh kernel and, backwards, two split-K weight gradients over 16384 rows for 864 parameters.  Here one workgroup owns the G
// groups of 16 pixels of one sample (16 G rows), reads o in the layout the offset conv wrote it ([B,HW,G,gc]) and writes both
// results.  With `zmajor` the second result is written group-major, [G,B,HW,C2], and `qres` [B,HW,C2] is added to every group:
// that IS the decoder query of modules.py:827-831 (query = q broadcast over the waypoints + flow_hidden), ready for the batched
// cross-attention.
__device__ __forceinline__ long long fgo_fh_row(int zmajor, long long b, int g, int hw, int B, int G, int HW) {
  return zmajor ? ((long long)g * B + b) * HW + hw : (b * G + g) * HW + hw;
}
template <typename T>
__global__ __launch_bounds__(256) void fg_offset_fwd_kernel(const T* __restrict__ o, const T* __restrict__ W1, const T* __restrict__ W2,
                                                            const float* __restrict__ b2, const T* __restrict__ qres, T* off,
                                                            T* __restrict__ fh, int B, int HW, int G, int gc, int C2, float scale, int zmajor) {
  extern __shared__ float fo_lds[];
  float* w2 = fo_lds;                  // [2][C2]
  float* bb = w2 + 2 * C2;             // [C2]
  float* soff = bb + C2;               // [16 G][2]
  const int t = threadIdx.x, RB = 16 * G;
  const int nhb = HW / 16;
  const long long b = blockIdx.x / nhb; const int hw0 = (blockIdx.x % nhb) * 16;
  if (fh) {
    for (int i = t; i < 2 * C2; i += 256) w2[i] = ldf(W2 + i);
    for (int i = t; i < C2; i += 256) bb[i] = b2 ? b2[i] : 0.f;
  }
  if (!o) {                                               // second half only: off is an input
    for (int i = t; i < RB * 2; i += 256) {
      const int lr = i >> 1, g = lr >> 4, hw = hw0 + (lr & 15);
      soff[i] = ldf(off + ((b * G + g) * HW + hw) * 2 + (i & 1));
    }
  } else
  for (int lr = t >> 1; lr < RB; lr += 128) {             // two lanes per row
    const int part = t & 1, g = lr >> 4, hw = hw0 + (lr & 15);
    const T* op = o + ((b * HW + hw) * G + g) * gc;
    float a0 = 0.f, a1 = 0.f;
    for (int i = part; i < gc; i += 2) {
      const float x = ldf(op + i);
      a0 += x * ldf(W1 + 2 * i); a1 += x * ldf(W1 + 2 * i + 1);
    }
    a0 += __shfl_xor(a0, 1); a1 += __shfl_xor(a1, 1);
    T* q = off + ((b * G + g) * HW + hw) * 2 + part;
    stf(q, tanhf(part ? a1 : a0) * scale);
    soff[lr * 2 + part] = ldf(q);              // fh is the product of the ROUNDED offsets (what the backward and the bias sampler see)
  }
  if (!fh) return;
  __syncthreads();
  constexpr int VN = Vec<T>::N;
  const int NV = C2 / VN;
  for (int i = t; i < RB * NV; i += 256) {
    const int lr = i / NV, c = (i % NV) * VN, g = lr >> 4, hw = hw0 + (lr & 15);
    const float o0 = soff[lr * 2], o1 = soff[lr * 2 + 1];
    float v[VN];
#pragma unroll
    for (int e = 0; e < VN; ++e) v[e] = o0 * w2[c + e] + o1 * w2[C2 + c + e] + bb[c + e];
    if (qres) {
      float r[VN];
      ld16(qres + (b * HW + hw) * C2 + c, r);
#pragma unroll
      for (int e = 0; e < VN; ++e) v[e] += r[e];
    }
    st16(fh + fgo_fh_row(zmajor, b, g, hw, B, G, HW) * C2 + c, v);
  }
}

// Either half runs alone: o == null takes off as an input (second half), fh == null stops after off.  The model uses the halves
// separately -- the query needs the attention output that the offsets feed -- but through the same two kernels.
// Backward of the pair: doff (gradient of off from the bias sampler, may be null) and dfh (may be null) ->
// dO [B,HW,G,gc], dq [B,HW,C2] (sum of dfh over the groups, when qres was given), dW1 [gc,2], dW2 [2,C2], db2 [C2] (f32, one
// atomic per entry and workgroup).  Same row ownership as the forward: (A) lanes over 16-byte column groups of dfh keep the
// column sums for dW2 / db2 and the per-pixel sum over the groups (dq) in registers and add their slice of dfh . W2^T to the
// row's offset gradient in LDS; (B) tanh'; (C) lanes over the gc input channels.  dO == null stops after (A) and writes the
// offset gradient to doff_out [B,G,HW,2] instead (second half alone).
template <typename T>
__global__ __launch_bounds__(256) void fg_offset_bwd_kernel(const T* __restrict__ o, const T* __restrict__ off, const T* __restrict__ W1,
                                                            const T* __restrict__ W2, const T* __restrict__ doff, const T* __restrict__ dfh,
                                                            T* __restrict__ dO, T* __restrict__ dq, T* __restrict__ doff_out,
                                                            float* __restrict__ dW1,
                                                            float* __restrict__ dW2, float* __restrict__ db2, int B, int HW, int G,
                                                            int gc, int C2, float scale, int zmajor) {
  constexpr int VN = Vec<T>::N;
  extern __shared__ float fo_lds[];
  const int t = threadIdx.x, RB = 16 * G;
  float* w2 = fo_lds;                  // [2][C2]
  float* scol = w2 + 2 * C2;           // [3][C2]: dW2 rows, db2
  float* soff = scol + 3 * C2;         // [RB][2]
  float* sdo = soff + RB * 2;          // [RB][2]  offset gradient, then pre-activation gradient
  float* sw1 = sdo + RB * 2;           // [gc][2]
  float* sdw1 = sw1 + gc * 2;          // [gc][2]
  const int nhb = HW / 16;
  const long long b = blockIdx.x / nhb; const int hw0 = (blockIdx.x % nhb) * 16;
  for (int i = t; i < 2 * C2; i += 256) w2[i] = dfh ? ldf(W2 + i) : 0.f;
  for (int i = t; i < 3 * C2; i += 256) scol[i] = 0.f;
  for (int i = t; i < RB * 2; i += 256) {
    const int lr = i >> 1, g = lr >> 4, hw = hw0 + (lr & 15);
    const long long at = ((b * G + g) * HW + hw) * 2 + (i & 1);
    soff[i] = ldf(off + at); sdo[i] = doff ? ldf(doff + at) : 0.f;
  }
  for (int i = t; i < gc * 2; i += 256) { sw1[i] = W1 ? ldf(W1 + i) : 0.f; sdw1[i] = 0.f; }
  __syncthreads();
  if (dfh) {
    const int NV = C2 / VN, nrg = 256 / NV;
    const int cv = t % NV, rg = t / NV, c = cv * VN;
    if (rg < nrg) {
      float cs[VN], a0[VN], a1[VN], wa[VN], wb[VN];
#pragma unroll
      for (int e = 0; e < VN; ++e) { cs[e] = a0[e] = a1[e] = 0.f; wa[e] = w2[c + e]; wb[e] = w2[C2 + c + e]; }
      for (int hwl = rg; hwl < 16; hwl += nrg) {
        float qs[VN];
#pragma unroll
        for (int e = 0; e < VN; ++e) qs[e] = 0.f;
        for (int g = 0; g < G; ++g) {
          const int lr = g * 16 + hwl;
          float v[VN];
          ld16(dfh + fgo_fh_row(zmajor, b, g, hw0 + hwl, B, G, HW) * C2 + c, v);
          const float o0 = soff[lr * 2], o1 = soff[lr * 2 + 1];
          float p0 = 0.f, p1 = 0.f;
#pragma unroll
          for (int e = 0; e < VN; ++e) {
            cs[e] += v[e]; a0[e] += o0 * v[e]; a1[e] += o1 * v[e]; qs[e] += v[e];
            p0 += v[e] * wa[e]; p1 += v[e] * wb[e];
          }
          atomicAdd(&sdo[lr * 2], p0); atomicAdd(&sdo[lr * 2 + 1], p1);
        }
        if (dq) st16(dq + (b * HW + hw0 + hwl) * C2 + c, qs);
      }
#pragma unroll
      for (int e = 0; e < VN; ++e) {
        atomicAdd(&scol[c + e], a0[e]); atomicAdd(&scol[C2 + c + e], a1[e]); atomicAdd(&scol[2 * C2 + c + e], cs[e]);
      }
    }
    __syncthreads();
    for (int i = t; i < 2 * C2; i += 256) atomicAdd(dW2 + i, scol[i]);
    if (db2) for (int i = t; i < C2; i += 256) atomicAdd(db2 + i, scol[2 * C2 + i]);
  }
  if (!dO) {                                              // second half only: hand the offset gradient on
    if (doff_out)
      for (int i = t; i < RB * 2; i += 256) {
        const int lr = i >> 1, g = lr >> 4, hw = hw0 + (lr & 15);
        stf(doff_out + ((b * G + g) * HW + hw) * 2 + (i & 1), sdo[i]);
      }
    return;
  }
  // (B) d tanh(u)*s = s - off^2 / s
  if (t < RB * 2) { const float ov = soff[t]; sdo[t] *= scale - ov * ov / scale; }
  __syncthreads();
  // (C) lanes over input channels
  {
    const int nrg = 256 / gc, i = t % gc, rg = t / gc;
    if (rg < nrg) {
      const float wa = sw1[2 * i], wb = sw1[2 * i + 1];
      float a0 = 0.f, a1 = 0.f;
      for (int lr = rg; lr < RB; lr += nrg) {
        const int g = lr >> 4, hw = hw0 + (lr & 15);
        const long long at = ((b * HW + hw) * G + g) * gc + i;
        const float d0 = sdo[lr * 2], d1 = sdo[lr * 2 + 1];
        const float x = ldf(o + at);
        a0 += x * d0; a1 += x * d1;
        stf(dO + at, d0 * wa + d1 * wb);
      }
      atomicAdd(&sdw1[2 * i], a0); atomicAdd(&sdw1[2 * i + 1], a1);
    }
    __syncthreads();
    for (int j = t; j < gc * 2; j += 256) atomicAdd(dW1 + j, sdw1[j]);
  }
}

static int fgo_check(const char* who, int B, int HW, int G, int gc, int C2, int dtype, const void* p16a, const void* p16b) {
  const int vn = dtype == STJ_F32 ? 4 : 8;
  if (B < 1 || HW % 16 || G < 1 || G > 8 || gc < 1 || gc > 256 || C2 % vn || C2 / vn > 256) {
    stj_set_error("%s: unsupported geometry B=%d HW=%d G=%d gc=%d C2=%d", who, B, HW, G, gc, C2); return STJ_EINVAL;
  }
  if ((((uintptr_t)p16a) | ((uintptr_t)p16b)) & 15) { stj_set_error("%s: fh / qres pointers must be 16-byte aligned", who); return STJ_EINVAL; }
  return STJ_OK;
}
extern "C" int stj_fg_offset_fwd(const void* o, const void* W1, const void* W2, const float* b2, const void* qres, void* off, void* fh,
                                 int B, int HW, int G, int gc, int C2, float scale, int zmajor, int dtype, hipStream_t stream) {
  if (int e = fgo_check("stj_fg_offset_fwd", B, HW, G, gc, C2, dtype, fh, qres)) return e;
  const size_t lds = (size_t)(3 * C2 + 32 * G) * sizeof(float);
  const int grid = B * (HW / 16);
#define FGO_FWD(TT) hipLaunchKernelGGL(fg_offset_fwd_kernel<TT>, dim3(grid), dim3(256), lds, stream, (const TT*)o, (const TT*)W1, (const TT*)W2, b2, (const TT*)qres, (TT*)off, (TT*)fh, B, HW, G, gc, C2, scale, zmajor)
  if (dtype == STJ_BF16) FGO_FWD(bf16); else if (dtype == STJ_F16) FGO_FWD(f16); else FGO_FWD(float);
#undef FGO_FWD
  return stj_check_launch("stj_fg_offset_fwd");
}
extern "C" int stj_fg_offset_bwd(const void* o, const void* off, const void* W1, const void* W2, const void* doff, const void* dfh,
                                 void* dO, void* dq, void* doff_out, float* dW1, float* dW2, float* db2, int B, int HW, int G, int gc,
                                 int C2, float scale, int zmajor, int dtype, hipStream_t stream) {
  if (int e = fgo_check("stj_fg_offset_bwd", B, HW, G, gc, C2, dtype, dfh, dq)) return e;
  const size_t lds = (size_t)(5 * C2 + 64 * G + 4 * gc) * sizeof(float);
  const int grid = B * (HW / 16);
#define FGO_BWD(TT) hipLaunchKernelGGL(fg_offset_bwd_kernel<TT>, dim3(grid), dim3(256), lds, stream, (const TT*)o, (const TT*)off, (const TT*)W1, (const TT*)W2, (const TT*)doff, (const TT*)dfh, (TT*)dO, (TT*)dq, (TT*)doff_out, dW1, dW2, db2, B, HW, G, gc, C2, scale, zmajor)
  if (dtype == STJ_BF16) FGO_BWD(bf16); else if (dtype == STJ_F16) FGO_BWD(f16); else FGO_BWD(float);
#undef FGO_BWD
  return stj_check_launch("stj_fg_offset_bwd");
}

extern "C" int stj_fg_bias_fwd(const void* off, const float* table, float* bias, int B, int G, int Hh, int Ww, int dtype, hipStream_t stream) {
  const long long total = (long long)B * G * Hh * Ww * Hh * Ww;
  if (total <= 0) return STJ_OK;
  const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  if (dtype == STJ_BF16) hipLaunchKernelGGL(fg_bias_fwd_kernel<bf16>, dim3(grid), dim3(256), 0, stream, (const bf16*)off, table, bias, B, G, Hh, Ww);
  else if (dtype == STJ_F16) hipLaunchKernelGGL(fg_bias_fwd_kernel<f16>, dim3(grid), dim3(256), 0, stream, (const f16*)off, table, bias, B, G, Hh, Ww);
  else hipLaunchKernelGGL(fg_bias_fwd_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)off, table, bias, B, G, Hh, Ww);
  return stj_check_launch("stj_fg_bias_fwd");
}
extern "C" int stj_fg_bias_bwd(const void* off, const float* table, const void* dbias, float* dtable, float* doff,
                               int B, int G, int Hh, int Ww, int dtype, hipStream_t stream) {
  if (B * G <= 0) return STJ_OK;
  // doff (f32 [B,G,HW,2]) MUST BE ZERO on entry: query slices accumulate into it
  int QS = 512 / (B * G);
  if (QS < 1) QS = 1;
  if (QS > 8) QS = 8;
  const int grid = B * G * QS;
  const size_t lds = (size_t)(2 * Hh - 1) * (2 * Ww - 1) * 2 * sizeof(float);
  if (dtype == STJ_BF16) hipLaunchKernelGGL(fg_bias_bwd_kernel<bf16>, dim3(grid), dim3(256), lds, stream, (const bf16*)off, table, (const bf16*)dbias, dtable, doff, B, G, Hh, Ww, QS);
  else if (dtype == STJ_F16) hipLaunchKernelGGL(fg_bias_bwd_kernel<f16>, dim3(grid), dim3(256), lds, stream, (const f16*)off, table, (const f16*)dbias, dtable, doff, B, G, Hh, Ww, QS);
  else hipLaunchKernelGGL(fg_bias_bwd_kernel<float>, dim3(grid), dim3(256), lds, stream, (const float*)off, table, (const float*)dbias, dtable, doff, B, G, Hh, Ww, QS);
  return stj_check_launch("stj_fg_bias_bwd");
}
