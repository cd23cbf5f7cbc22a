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
