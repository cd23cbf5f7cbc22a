// OGMFlow loss, fused: one streaming pass over the [B,H,W,32] logits + ground truth for the forward sums,
// one for d(loss)/d(logits).  Restates OGMFlow_loss.__call__ (reference loss.py:50-170).  With the flags
// train.py:195-196 uses (use_focal_loss=False, use_pred=False, no_use_warp=False):
//   observed_xe / occluded_xe : sigmoid cross entropy sums (loss.py:173-229)
//   flow                      : masked L1 (loss.py:273-295)
//   flow_warp_xe              : XE(labels=true_all, logits = clip(sig(gt_obs)+sig(gt_occ),0,1) * warp(origin, id+pred_flow))
//                               (loss.py:144-158,231-250 -- the quirk of SURVEY App. D-8 is reproduced)
//   use_gt gate               : res_k = [PR-AUC(true_all, warp(origin, id+gt_flow)*true_all) > 0]  (loss.py:127-137),
//                               Keras AUC(num_thresholds=100, curve='PR', summation 'interpolation') restated below.
// The constructor defaults (use_focal_loss=True) and use_pred=True are template variants of the same two kernels:
//   focal (loss.py:183-190,212-219): obs/occ pixel term = XE + tfa sigmoid_focal_crossentropy(from_logits, alpha .25, gamma 2);
//   focal warp (loss.py:244-245)   : focal on PROBABILITIES q (Keras backend BCE, q clipped to [1e-7, 1-1e-7]) summed over
//                                    pixels + Keras BinaryCrossentropy = per-sample MEAN over pixels, summed over the batch;
//   use_pred (loss.py:253-268)     : q = clip(sig(pred_obs)+sig(pred_occ),0,1) * warp, and only the per-sample-mean BCE
//                                    survives (loss.py:265 overwrites xe_sum); gradients reach obs, occ and flow logits.
// Channel slicing of the logits follows train.py:105-123: 4k+0 obs, 4k+1 occ, 4k+2..3 flow (dx,dy).
#include "common.h"

#define NWP 8
#define LOSS_PARTS 32
// per-waypoint accumulator slots
enum { S_OBS = 0, S_OCC = 1, S_L1 = 2, S_EX = 3, S_WARP = 4, S_N = 5 };

__device__ __forceinline__ float xe_logits(float z, float x) {   // tf.nn.sigmoid_cross_entropy_with_logits
  return fmaxf(x, 0.f) - x * z + log1pf(__expf(-fabsf(x)));      // v_exp_f32 (1 ulp on a value in (0, 1]); log1p stays exact for small arguments
}
__device__ __forceinline__ float sigmoidf(float x) { return __frcp_rn(1.f + __expf(-x)); }
// XE + tfa focal term on a logit x with label y; *d = derivative w.r.t. x when d != NULL
__device__ __forceinline__ float xe_focal_logits(float y, float x, float* d) {
  const float ce = xe_logits(y, x), p = sigmoidf(x);
  const float pt = y * p + (1.f - y) * (1.f - p), at = y * 0.25f + (1.f - y) * 0.75f, om = 1.f - pt;
  if (d) {
    const float dce = p - y, dpt = (2.f * y - 1.f) * p * (1.f - p);
    *d = dce + at * (om * om * dce - 2.f * om * dpt * ce);
  }
  return ce + at * om * om * ce;
}
// Keras backend binary_crossentropy(from_logits=False): clip to [eps, 1-eps], -(y log(q+eps) + (1-y) log(1-q+eps));
// tf.clip_by_value passes the gradient for eps <= q <= 1-eps (bounds included)
__device__ __forceinline__ float bce_prob(float y, float q, float* d) {
  const float eps = 1e-7f, hi = 1.f - 1e-7f;
  const float qc = fminf(fmaxf(q, eps), hi);
  if (d) *d = (q >= eps && q <= hi) ? (1.f - y) / (1.f - qc + eps) - y / (qc + eps) : 0.f;
  return -(y * logf(qc + eps) + (1.f - y) * logf(1.f - qc + eps));
}
// tfa focal term on a probability q (pred_prob = q unclipped, ce = bce_prob)
__device__ __forceinline__ float focal_prob(float y, float q, float* d) {
  float dce;
  const float ce = bce_prob(y, q, &dce);
  const float pt = y * q + (1.f - y) * (1.f - q), at = y * 0.25f + (1.f - y) * 0.75f, om = 1.f - pt;
  if (d) *d = at * (om * om * dce - 2.f * om * (2.f * y - 1.f) * ce);
  return at * om * om * ce;
}
// the warp-consistency pixel term on the joint probability q with label ta; inv_hw = 1 / (H*W)
template <bool FOCAL, bool PRED>
__device__ __forceinline__ float warp_term(float ta, float q, float inv_hw, float* d) {
  if (PRED) {
    const float v = bce_prob(ta, q, d);
    if (d) *d *= inv_hw;
    return v * inv_hw;
  }
  if (FOCAL) {
    float d0, d1;
    const float v = focal_prob(ta, q, &d0) + bce_prob(ta, q, &d1) * inv_hw;
    if (d) *d = d0 + d1 * inv_hw;
    return v;
  }
  if (d) *d = sigmoidf(q) - ta;
  return xe_logits(ta, q);
}

// bilinear sample of a single-channel [H][W] image at (x,y) (sample(): pad 1, warp+1); optionally d/dx, d/dy
__device__ __forceinline__ float warp_sample(const float* img, int H, int W, float x, float y, float* ddx, float* ddy) {
  Bil c = bil_setup(x + 1.f, y + 1.f, H + 2, W + 2);
  const float tl = pad_at(img, H, W, 1, c.y0, c.x0), tr = pad_at(img, H, W, 1, c.y0, c.x0 + 1);
  const float bl = pad_at(img, H, W, 1, c.y0 + 1, c.x0), br = pad_at(img, H, W, 1, c.y0 + 1, c.x0 + 1);
  const float top = c.ax * (tr - tl) + tl, bot = c.ax * (br - bl) + bl;
  if (ddx) *ddx = c.gx ? (c.ay * (br - bl) + (1.f - c.ay) * (tr - tl)) : 0.f;
  if (ddy) *ddy = c.gy ? (bot - top) : 0.f;
  return c.ay * (bot - top) + top;
}

// Keras AUC bucket of a prediction: the number of thresholds strictly below it, thresholds t0 = -1e-7, t_i = i/99 (i = 1..98),
// t_99 = 1 + 1e-7 as float32 (tf.keras.metrics.AUC(num_thresholds=100))
__device__ __forceinline__ int auc_bucket(float pred) {
  int bk = 0;
  if (pred > -1e-7f) {
    bk = 1;
    int j = (int)(pred * 99.f);
    j = j < 0 ? 0 : (j > 98 ? 98 : j);
    // count i in 1..98 with t_i < pred, robust to rounding of pred*99
    int cnt = j;
    if (cnt >= 1 && !((float)((double)cnt / 99.0) < pred)) cnt -= 1;
    else if (cnt < 98 && ((float)((double)(cnt + 1) / 99.0) < pred)) cnt += 1;
    bk += cnt;
    if (pred > (float)(1.0 + 1e-7)) bk += 1;
  }
  return bk;
}

// ---- AUC gate ----------------------------------------------------------------------------------------
// hist [NWP][2][101] (int): bucket = #thresholds strictly below pred; class 1 = label true.
// cnt (optional): int[NWP], += the pixels of waypoint k with a non-zero true flow (the flow term's denominator: loss_coef_kernel)
__global__ __launch_bounds__(256) void auc_hist_kernel(const float* gt_obs, const float* gt_occ, const float* gt_flow,
                                                       const float* origin, int* hist, int* cnt, int B, int H, int W) {
  __shared__ int sh[2 * 101 + 1];
  const int k = blockIdx.y;
  for (int i = threadIdx.x; i < 203; i += 256) sh[i] = 0;
  __syncthreads();
  int nex = 0;
  const long long npix = (long long)B * H * W;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < npix; i += gridDim.x * 256ll) {
    const int x = (int)(i % W); long long t = i / W;
    const int y = (int)(t % H); const long long b = t / H;
    const long long g = ((b * NWP + k) * H + y) * W + x;
    const float ta = fminf(fmaxf(gt_obs[g] + gt_occ[g], 0.f), 1.f);
    const float* img = origin + (b * NWP + k) * (long long)H * W;
    const float fx = gt_flow[2 * g], fy = gt_flow[2 * g + 1];
    nex += (fx != 0.f || fy != 0.f) ? 1 : 0;
    const float wp = warp_sample(img, H, W, (float)x + fx, (float)y + fy, nullptr, nullptr);
    const float pred = wp * ta;
    const int bk = auc_bucket(pred);
    atomicAdd(&sh[(ta != 0.f ? 101 : 0) + bk], 1);
  }
  if (cnt) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nex += __shfl_xor(nex, o, 64);
    if ((threadIdx.x & 63) == 0 && nex) atomicAdd(&sh[202], nex);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 202; i += 256)
    if (sh[i]) atomicAdd(hist + k * 202 + i, sh[i]);
  if (cnt && threadIdx.x == 0 && sh[202]) atomicAdd(cnt + k, sh[202]);
}
// one 128-thread block per waypoint, thread i = threshold i: Keras interpolate_pr_auc from the histogram; gate[k] = auc > 0;
// auc_out optional.  (v0 ran the whole recurrence in ONE thread per waypoint with 1.6 KB of f64 scratch arrays: 82 us.)
__global__ __launch_bounds__(128) void auc_gate_kernel(const int* hist, float* gate, float* auc_out) {   // gate may be NULL
  __shared__ int hn[101], hp[101];
  __shared__ double tp[100], pp[100], part[128];
  const int k = blockIdx.x, i = threadIdx.x;
  if (i <= 100) { hn[i] = hist[k * 202 + i]; hp[i] = hist[k * 202 + 101 + i]; }
  __syncthreads();
  double totp = 0, totn = 0;
  for (int j = 0; j <= 100; ++j) { totp += hp[j]; totn += hn[j]; }
  if (i < 100) {                       // positive at threshold i <=> bucket > i
    double cp = 0, cn = 0;
    for (int j = 0; j <= i; ++j) { cp += hp[j]; cn += hn[j]; }
    tp[i] = totp - cp;
    pp[i] = tp[i] + (totn - cn);
  }
  __syncthreads();
  double term = 0;
  if (i < 99) {
    const double dtp = tp[i] - tp[i + 1], dp = pp[i] - pp[i + 1];
    const double den = dp > 0 ? dp : 0;
    const double slope = den != 0 ? dtp / den : 0;
    const double icpt = tp[i + 1] - slope * pp[i + 1];
    double ratio = 1.0;
    if (pp[i] > 0 && pp[i + 1] > 0) ratio = pp[i] / pp[i + 1];
    const double d2 = totp > 0 ? totp : 0;    // tp + fn = all positives
    term = d2 != 0 ? slope * (dtp + icpt * log(ratio)) / d2 : 0;
  }
  part[i] = term;
  __syncthreads();
  if (i == 0) {
    double auc = 0;
    for (int j = 0; j < 99; ++j) auc += part[j];          // same summation order as the serial recurrence
    if (gate) gate[k] = ((1.0 - auc) < 1.0) ? 1.f : 0.f;
    if (auc_out) auc_out[k] = (float)auc;
  }
}

// ---- forward sums -------------------------------------------------------------------------------------
// One thread per (pixel, waypoint) -- thread index = pixel * 8 + k: a lane loads ITS float4 of the pixel's 128-byte logit line (fully
// coalesced) and carries 5 accumulators; the ground-truth reads of the 8 lanes of a waypoint within a wave cover 8 consecutive pixels
// (32-byte sectors).  The first version had one thread per pixel with all 8 waypoints (40 accumulators + 32 logits: 154 VGPRs, three
// waves per SIMD for a kernel that waits on four gathers per waypoint, and every 16-byte logit load touching 64 different cache lines
// per wave): 85 us for 151 MB.
template <bool FOCAL, bool PRED>
__global__ __launch_bounds__(256) void loss_fwd_kernel(const float* logits, const float* gt_obs, const float* gt_occ,
                                                       const float* gt_flow, const float* origin, float* sums,
                                                       int B, int H, int W, int use_warp) {
  __shared__ float red[4][NWP * S_N];
  const float inv_hw = 1.f / ((float)H * (float)W);
  float acc[S_N];
#pragma unroll
  for (int i = 0; i < S_N; ++i) acc[i] = 0.f;
  const long long nitem = (long long)B * H * W * NWP;
  const int k = threadIdx.x & 7;                       // 256 % 8 == 0 and the grid stride is a multiple of 8: k is fixed per thread
  for (long long it = blockIdx.x * 256ll + threadIdx.x; it < nitem; it += gridDim.x * 256ll) {
    const long long i = it >> 3;
    const int x = (int)(i % W); long long t = i / W;
    const int y = (int)(t % H); const long long b = t / H;
    const float4 lg = reinterpret_cast<const float4*>(logits)[it];
    const long long g = ((b * NWP + k) * H + y) * W + x;
    const float to = gt_obs[g], tc = gt_occ[g];
    const float2 fl = reinterpret_cast<const float2*>(gt_flow)[g];
    const float fx = fl.x, fy = fl.y;
    acc[S_OBS] += FOCAL ? xe_focal_logits(to, lg.x, nullptr) : xe_logits(to, lg.x);
    acc[S_OCC] += FOCAL ? xe_focal_logits(tc, lg.y, nullptr) : xe_logits(tc, lg.y);
    const float ex = (fx != 0.f || fy != 0.f) ? 1.f : 0.f;
    acc[S_L1] += (fabsf(fx - lg.z) + fabsf(fy - lg.w)) * ex;
    acc[S_EX] += ex;
    if (use_warp) {
      const float* img = origin + (b * NWP + k) * (long long)H * W;
      const float wp = warp_sample(img, H, W, (float)x + lg.z, (float)y + lg.w, nullptr, nullptr);
      const float sg = PRED ? fminf(fmaxf(sigmoidf(lg.x) + sigmoidf(lg.y), 0.f), 1.f)
                            : fminf(fmaxf(sigmoidf(to) + sigmoidf(tc), 0.f), 1.f);
      const float ta = fminf(fmaxf(to + tc, 0.f), 1.f);
      acc[S_WARP] += warp_term<FOCAL, PRED>(ta, sg * wp, inv_hw, nullptr);
    }
  }
  // lanes with equal (lane & 7) hold the same waypoint: reduce over lane bits 3..5, then the 4 waves through LDS
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < S_N; ++i) {
    float v = acc[i];
    v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
    if (lane < 8) red[w][lane * S_N + i] = v;
  }
  __syncthreads();
  // LOSS_PARTS copies of the 40 accumulators: 2048 blocks on one copy queue 2048 same-address atomics per slot (~70 of this
  // kernel's 92 us); loss_finalize_kernel folds the copies
  if (threadIdx.x < NWP * S_N)
    atomicAdd(sums + (blockIdx.x % LOSS_PARTS) * (NWP * S_N) + threadIdx.x, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

struct LossCfg { float ogm_w, occ_w, fow, replica; int use_warp; };

// loss[5] = observed_xe, occluded_xe, flow, flow_warp_xe, their sum ; coef [NWP][4] per-waypoint backward coefficients (before upstream grads)
__global__ void loss_finalize_kernel(const float* sums_parts, int nparts, const float* gate, float* loss, float* coef, float npix, LossCfg c) {
  // 240 threads: six per accumulator, each over every sixth copy (a serial walk of 128 copies by 40 threads was 36 us of dependent loads
  // on the step's critical path); the six partial sums are added in a fixed order
  __shared__ float sums[NWP * S_N], part[6][NWP * S_N];
  if (threadIdx.x < 6 * NWP * S_N) {
    const int slot = threadIdx.x % (NWP * S_N), j = threadIdx.x / (NWP * S_N);
    float a = 0.f;
#pragma unroll 8
    for (int q = j; q < nparts; q += 6) a += sums_parts[q * (NWP * S_N) + slot];
    part[j][slot] = a;
  }
  __syncthreads();
  if (threadIdx.x < NWP * S_N) {
    const int t = threadIdx.x;
    sums[t] = ((((part[0][t] + part[1][t]) + part[2][t]) + part[3][t]) + part[4][t]) + part[5][t];
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  float so = 0.f, sc = 0.f, sf = 0.f, sw = 0.f, fc = 0.f;
  for (int k = 0; k < NWP; ++k) fc += gate[k];
  for (int k = 0; k < NWP; ++k) {
    const float* s = sums + k * S_N;
    so += c.ogm_w * s[S_OBS] / (npix * c.replica);
    sc += c.occ_w * s[S_OCC] / (npix * c.replica);
    const float den = s[S_EX] * c.replica / 2.f;
    const float fl = den != 0.f ? s[S_L1] / den : 0.f;
    sf += gate[k] * fl;
    sw += gate[k] * c.fow * s[S_WARP] / (npix * c.replica);
    coef[k * 4 + 0] = c.ogm_w / (npix * c.replica) / NWP;
    coef[k * 4 + 1] = c.occ_w / (npix * c.replica) / NWP;
    coef[k * 4 + 2] = den != 0.f ? gate[k] / fc / den : 0.f;
    coef[k * 4 + 3] = c.use_warp ? gate[k] / fc * c.fow / (npix * c.replica) : 0.f;
  }
  loss[0] = so / NWP;
  loss[1] = sc / NWP;
  loss[2] = sf / fc;
  loss[3] = c.use_warp ? sw / fc : 0.f;
  loss[4] = loss[0] + loss[1] + loss[2] + loss[3];      // the training objective (train.py:221), in the order a host-side sum adds them
}

// dlogits[B,H,W,32] = sum_j up[j] * d loss_j / d logits
template <bool FOCAL, bool PRED>
__global__ __launch_bounds__(256) void loss_bwd_kernel(const float* logits, const float* gt_obs, const float* gt_occ,
                                                       const float* gt_flow, const float* origin, const float* coef,
                                                       const float* up, float* dlogits, int B, int H, int W, int use_warp, int up1) {
  // one thread per (pixel, waypoint), as in loss_fwd_kernel: one coalesced float4 in, one out
  const int k = threadIdx.x & 7;
  // up1: ONE upstream gradient for all four terms (the step differentiates their sum)
  const float c0 = coef[4 * k] * up[0], c1 = coef[4 * k + 1] * up[up1 ? 0 : 1], c2 = coef[4 * k + 2] * up[up1 ? 0 : 2], c3 = coef[4 * k + 3] * up[up1 ? 0 : 3];
  const long long nitem = (long long)B * H * W * NWP;
  const float inv_hw = 1.f / ((float)H * (float)W);
  for (long long it = blockIdx.x * 256ll + threadIdx.x; it < nitem; it += gridDim.x * 256ll) {
    const long long i = it >> 3;
    const int x = (int)(i % W); long long t = i / W;
    const int y = (int)(t % H); const long long b = t / H;
    const float4 lg = reinterpret_cast<const float4*>(logits)[it];
    const long long g = ((b * NWP + k) * H + y) * W + x;
    const float to = gt_obs[g], tc = gt_occ[g];
    const float2 fl = reinterpret_cast<const float2*>(gt_flow)[g];
    const float fx = fl.x, fy = fl.y;
    float g0, g1;
    if (FOCAL) { xe_focal_logits(to, lg.x, &g0); xe_focal_logits(tc, lg.y, &g1); }
    else { g0 = sigmoidf(lg.x) - to; g1 = sigmoidf(lg.y) - tc; }
    g0 *= c0; g1 *= c1;
    const float ex = (fx != 0.f || fy != 0.f) ? 1.f : 0.f;
    const float d0 = fx - lg.z, d1 = fy - lg.w;
    float g2 = -c2 * ex * (d0 > 0.f ? 1.f : (d0 < 0.f ? -1.f : 0.f));
    float g3 = -c2 * ex * (d1 > 0.f ? 1.f : (d1 < 0.f ? -1.f : 0.f));
    if (use_warp && c3 != 0.f) {
      const float* img = origin + (b * NWP + k) * (long long)H * W;
      float ddx, ddy;
      const float wp = warp_sample(img, H, W, (float)x + lg.z, (float)y + lg.w, &ddx, &ddy);
      const float sa = sigmoidf(PRED ? lg.x : to), sb = sigmoidf(PRED ? lg.y : tc);
      const float ssum = sa + sb;
      const float sg = fminf(fmaxf(ssum, 0.f), 1.f);
      const float ta = fminf(fmaxf(to + tc, 0.f), 1.f);
      float dq;
      warp_term<FOCAL, PRED>(ta, sg * wp, inv_hw, &dq);
      dq *= c3;
      g2 += dq * sg * ddx;
      g3 += dq * sg * ddy;
      if (PRED && ssum <= 1.f) {          // clip_by_value passes the gradient inside [0, 1] (sum of two sigmoids > 0)
        g0 += dq * wp * sa * (1.f - sa);
        g1 += dq * wp * sb * (1.f - sb);
      }
    }
    reinterpret_cast<float4*>(dlogits)[it] = make_float4(g0, g1, g2, g3);
  }
}

// ---- forward sums AND d(total)/d(logits) in one pass --------------------------------------------------
// Every backward coefficient (loss_finalize_kernel: coef[k][0..3]) depends on the GROUND TRUTH alone -- the weights, the pixel count, the
// AUC gate and, for the flow term, the number of pixels with a non-zero true flow (loss.py:279-291) -- so a step that differentiates
// the sum of the four terms with a unit upstream gradient (train.py:221-223) does not need the forward sums before it can write
// d/dlogits: loss_coef (below, issued with the gate on the loss-preparation stream, under the model's forward pass) produces the
// coefficients, and this kernel is loss_fwd_kernel + loss_bwd_kernel on ONE read of the logits and the ground truth.  In the captured
// train step the two passes and the finalize launch between them ran alone on the machine between the last forward and the first
// backward kernel (69 + 8 + 59 us of a 5.9 ms step, profiles/r06_p_timeline_concurrent.txt).
#define LOSS_PARTS_FUSED 128
#ifndef LOSS_FB_MAXG
#define LOSS_FB_MAXG 16384
#endif
template <bool FOCAL, bool PRED>
__global__ __launch_bounds__(256) void loss_fwd_bwd_kernel(const float* logits, const float* gt_obs, const float* gt_occ,
                                                           const float* gt_flow, const float* origin, const float* coef,
                                                           float* sums, float* dlogits, int B, int H, int W, int use_warp) {
  __shared__ float red[4][NWP * S_N];
  const float inv_hw = 1.f / ((float)H * (float)W);
  float acc[S_N];
#pragma unroll
  for (int i = 0; i < S_N; ++i) acc[i] = 0.f;
  const long long nitem = (long long)B * H * W * NWP;
  const int k = threadIdx.x & 7;
  const float c0 = coef[4 * k], c1 = coef[4 * k + 1], c2 = coef[4 * k + 2], c3 = coef[4 * k + 3];
  for (long long it = blockIdx.x * 256ll + threadIdx.x; it < nitem; it += gridDim.x * 256ll) {
    const long long i = it >> 3;
    const int x = (int)(i % W); long long t = i / W;
    const int y = (int)(t % H); const long long b = t / H;
    const float4 lg = reinterpret_cast<const float4*>(logits)[it];
    const long long g = ((b * NWP + k) * H + y) * W + x;
    const float to = gt_obs[g], tc = gt_occ[g];
    const float2 fl = reinterpret_cast<const float2*>(gt_flow)[g];
    const float fx = fl.x, fy = fl.y;
    float g0, g1;
    if (FOCAL) { acc[S_OBS] += xe_focal_logits(to, lg.x, &g0); acc[S_OCC] += xe_focal_logits(tc, lg.y, &g1); }
    else {
      acc[S_OBS] += xe_logits(to, lg.x); acc[S_OCC] += xe_logits(tc, lg.y);
      g0 = sigmoidf(lg.x) - to; g1 = sigmoidf(lg.y) - tc;
    }
    g0 *= c0; g1 *= c1;
    const float ex = (fx != 0.f || fy != 0.f) ? 1.f : 0.f;
    const float d0 = fx - lg.z, d1 = fy - lg.w;
    acc[S_L1] += (fabsf(d0) + fabsf(d1)) * ex;
    acc[S_EX] += ex;
    float g2 = -c2 * ex * (d0 > 0.f ? 1.f : (d0 < 0.f ? -1.f : 0.f));
    float g3 = -c2 * ex * (d1 > 0.f ? 1.f : (d1 < 0.f ? -1.f : 0.f));
    if (use_warp) {
      const float* img = origin + (b * NWP + k) * (long long)H * W;
      float ddx, ddy;
      const float wp = warp_sample(img, H, W, (float)x + lg.z, (float)y + lg.w, &ddx, &ddy);
      const float sa = sigmoidf(PRED ? lg.x : to), sb = sigmoidf(PRED ? lg.y : tc);
      const float ssum = sa + sb;
      const float sg = fminf(fmaxf(ssum, 0.f), 1.f);
      const float ta = fminf(fmaxf(to + tc, 0.f), 1.f);
      float dq;
      acc[S_WARP] += warp_term<FOCAL, PRED>(ta, sg * wp, inv_hw, &dq);
      if (c3 != 0.f) {
        dq *= c3;
        g2 += dq * sg * ddx;
        g3 += dq * sg * ddy;
        if (PRED && ssum <= 1.f) {
          g0 += dq * wp * sa * (1.f - sa);
          g1 += dq * wp * sb * (1.f - sb);
        }
      }
    }
    reinterpret_cast<float4*>(dlogits)[it] = make_float4(g0, g1, g2, g3);
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < S_N; ++i) {
    float v = acc[i];
    v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
    if (lane < 8) red[w][lane * S_N + i] = v;
  }
  __syncthreads();
  if (threadIdx.x < NWP * S_N)
    atomicAdd(sums + (blockIdx.x % LOSS_PARTS_FUSED) * (NWP * S_N) + threadIdx.x, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// pixels with a non-zero true flow, per waypoint (the denominator of the flow term, loss.py:279-291): cnt int[8], zero on entry
template <bool VEC4>
__global__ __launch_bounds__(256) void loss_flow_count_kernel(const float* gt_flow, int* cnt, long long plane) {
  __shared__ int red[4];
  const int k = blockIdx.y % NWP;
  const float* fp = gt_flow + (long long)blockIdx.y * plane * 2;
  int n = 0;
  if (VEC4) {       // two pixels per 16-byte load
    const float4* f = reinterpret_cast<const float4*>(fp);
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < plane / 2; i += gridDim.x * 256ll) {
      const float4 v = f[i];
      n += ((v.x != 0.f || v.y != 0.f) ? 1 : 0) + ((v.z != 0.f || v.w != 0.f) ? 1 : 0);
    }
  } else {
    const float2* f = reinterpret_cast<const float2*>(fp);
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < plane; i += gridDim.x * 256ll) {
      const float2 v = f[i];
      n += (v.x != 0.f || v.y != 0.f) ? 1 : 0;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) { const int s = red[0] + red[1] + red[2] + red[3]; if (s) atomicAdd(cnt + k, s); }
}
// the coefficients loss_finalize_kernel derives from the forward sums, from the count instead (bit-identical: the float sum of ones is exact)
__global__ void loss_coef_kernel(const int* cnt, const float* gate, float* coef, float npix, LossCfg c) {
  if (threadIdx.x != 0) return;
  float fc = 0.f;
  for (int k = 0; k < NWP; ++k) fc += gate[k];
  for (int k = 0; k < NWP; ++k) {
    const float den = (float)cnt[k] * c.replica / 2.f;
    coef[k * 4 + 0] = c.ogm_w / (npix * c.replica) / NWP;
    coef[k * 4 + 1] = c.occ_w / (npix * c.replica) / NWP;
    coef[k * 4 + 2] = den != 0.f ? gate[k] / fc / den : 0.f;
    coef[k * 4 + 3] = c.use_warp ? gate[k] / fc * c.fow / (npix * c.replica) : 0.f;
  }
}

// gate: f32[8] out.  hist: int[8*202] scratch, MUST BE ZERO on entry (no memset node here: hipMemsetAsync inside a captured
// hipGraph replayed with stale contents on ROCm 7.2 -- tools/probes/dbg_graph2.py).  auc_out optional f32[8].
extern "C" int stj_loss_auc_gate(const float* gt_obs, const float* gt_occ, const float* gt_flow, const float* origin,
                                 int* hist, float* gate, float* auc_out, int B, int H, int W, hipStream_t stream) {
  const long long npix = (long long)B * H * W;
  const int gx = (int)min(256ll, (npix + 255) / 256);
  hipLaunchKernelGGL(auc_hist_kernel, dim3(gx, NWP), dim3(256), 0, stream, gt_obs, gt_occ, gt_flow, origin, hist, (int*)nullptr, B, H, W);
  hipLaunchKernelGGL(auc_gate_kernel, dim3(NWP), dim3(128), 0, stream, hist, gate, auc_out);
  return stj_check_launch("stj_loss_auc_gate");
}
// sums: f32[32*40] scratch (32 copies of the 40 accumulators), MUST BE ZERO on entry; loss f32[4]; coef f32[32]
extern "C" int stj_loss_fwd(const float* logits, const float* gt_obs, const float* gt_occ, const float* gt_flow, const float* origin,
                            const float* gate, float* sums, float* loss, float* coef, int B, int H, int W, float ogm_w, float occ_w,
                            float flow_origin_w, float replica, int flags, hipStream_t stream) {
  if (((uintptr_t)logits) & 15) { stj_set_error("loss: logits must be 16-byte aligned"); return STJ_EINVAL; }
  if (((uintptr_t)gt_flow) & 7) { stj_set_error("loss: gt_flow must be 8-byte aligned"); return STJ_EINVAL; }
  if (flags & ~7) { stj_set_error("loss: unknown flag bits %d", flags); return STJ_EINVAL; }
  const long long npix = (long long)B * H * W;
  const int gx = (int)min(4096ll, (npix * NWP + 255) / 256);
  const int use_warp = flags & 1, focal = (flags >> 1) & 1, pred = (flags >> 2) & 1;
#define LOSS_FWD(FO, PR) hipLaunchKernelGGL((loss_fwd_kernel<FO, PR>), dim3(gx), dim3(256), 0, stream, logits, gt_obs, gt_occ, gt_flow, origin, sums, B, H, W, use_warp)
  if (focal && pred) LOSS_FWD(true, true); else if (focal) LOSS_FWD(true, false); else if (pred) LOSS_FWD(false, true); else LOSS_FWD(false, false);
#undef LOSS_FWD
  LossCfg c; c.ogm_w = ogm_w; c.occ_w = occ_w; c.fow = flow_origin_w; c.replica = replica; c.use_warp = use_warp;
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, stream, sums, LOSS_PARTS, gate, loss, coef, (float)npix, c);
  return stj_check_launch("stj_loss_fwd");
}
extern "C" int stj_loss_bwd(const float* logits, const float* gt_obs, const float* gt_occ, const float* gt_flow, const float* origin,
                            const float* coef, const float* upstream, float* dlogits, int B, int H, int W, int flags, hipStream_t stream) {
  if (flags & ~15) { stj_set_error("loss: unknown flag bits %d", flags); return STJ_EINVAL; }
  if ((((uintptr_t)logits) | ((uintptr_t)dlogits)) & 15) { stj_set_error("loss: logits / dlogits must be 16-byte aligned"); return STJ_EINVAL; }
  if (((uintptr_t)gt_flow) & 7) { stj_set_error("loss: gt_flow must be 8-byte aligned"); return STJ_EINVAL; }
  const long long npix = (long long)B * H * W;
  const int gx = (int)min(8192ll, (npix * NWP + 255) / 256);
  const int use_warp = flags & 1, focal = (flags >> 1) & 1, pred = (flags >> 2) & 1;
#define LOSS_BWD(FO, PR) hipLaunchKernelGGL((loss_bwd_kernel<FO, PR>), dim3(gx), dim3(256), 0, stream, logits, gt_obs, gt_occ, gt_flow, origin, coef, upstream, dlogits, B, H, W, use_warp, (flags >> 3) & 1)
  if (focal && pred) LOSS_BWD(true, true); else if (focal) LOSS_BWD(true, false); else if (pred) LOSS_BWD(false, true); else LOSS_BWD(false, false);
#undef LOSS_BWD
  return stj_check_launch("stj_loss_bwd");
}

// coef f32[32]: the backward coefficients from the ground truth alone (see loss_fwd_bwd_kernel); cnt int[8] scratch, MUST BE ZERO on entry
extern "C" int stj_loss_coef(const float* gt_flow, const float* gate, int* cnt, float* coef, int B, int H, int W, float ogm_w, float occ_w,
                             float flow_origin_w, float replica, int flags, hipStream_t stream) {
  if (((uintptr_t)gt_flow) & 7) { stj_set_error("loss: gt_flow must be 8-byte aligned"); return STJ_EINVAL; }
  if (flags & ~7) { stj_set_error("loss: unknown flag bits %d", flags); return STJ_EINVAL; }
  const long long plane = (long long)H * W;
  if (B > 0 && plane > 0) {
    // 16 workgroups per (sample, waypoint) plane at 256 x 256: 8 pixels per thread; one atomic per workgroup (128 per counter at B = 8)
    const int gx = (int)min(16ll, (plane / 2 + 255) / 256);
    if (plane % 2 == 0 && (((uintptr_t)gt_flow) & 15) == 0)
      hipLaunchKernelGGL(loss_flow_count_kernel<true>, dim3(gx, B * NWP), dim3(256), 0, stream, gt_flow, cnt, plane);
    else
      hipLaunchKernelGGL(loss_flow_count_kernel<false>, dim3(gx, B * NWP), dim3(256), 0, stream, gt_flow, cnt, plane);
  }
  LossCfg c; c.ogm_w = ogm_w; c.occ_w = occ_w; c.fow = flow_origin_w; c.replica = replica; c.use_warp = flags & 1;
  hipLaunchKernelGGL(loss_coef_kernel, dim3(1), dim3(64), 0, stream, cnt, gate, coef, (float)((long long)B * H * W), c);
  return stj_check_launch("stj_loss_coef");
}
// stj_loss_auc_gate + stj_loss_coef on ONE pass over the ground truth (the histogram kernel reads gt_flow anyway): hist int[8*202 + 8]
// scratch, MUST BE ZERO on entry (the last 8: the flow counts)
extern "C" int stj_loss_gate_coef(const float* gt_obs, const float* gt_occ, const float* gt_flow, const float* origin, int* hist, float* gate,
                                  float* auc_out, float* coef, int B, int H, int W, float ogm_w, float occ_w, float flow_origin_w,
                                  float replica, int flags, hipStream_t stream) {
  if (flags & ~7) { stj_set_error("loss: unknown flag bits %d", flags); return STJ_EINVAL; }
  const long long npix = (long long)B * H * W;
  const int gx = (int)min(256ll, (npix + 255) / 256);
  int* cnt = hist + NWP * 202;
  if (gx > 0) hipLaunchKernelGGL(auc_hist_kernel, dim3(gx, NWP), dim3(256), 0, stream, gt_obs, gt_occ, gt_flow, origin, hist, cnt, B, H, W);
  hipLaunchKernelGGL(auc_gate_kernel, dim3(NWP), dim3(128), 0, stream, hist, gate, auc_out);
  LossCfg c; c.ogm_w = ogm_w; c.occ_w = occ_w; c.fow = flow_origin_w; c.replica = replica; c.use_warp = flags & 1;
  hipLaunchKernelGGL(loss_coef_kernel, dim3(1), dim3(64), 0, stream, cnt, gate, coef, (float)npix, c);
  return stj_check_launch("stj_loss_gate_coef");
}
// stj_loss_fwd and stj_loss_bwd with a unit upstream gradient on the sum of the four terms, as one pass: coef_in = stj_loss_coef's output;
// sums f32[128*40] scratch, MUST BE ZERO on entry; loss f32[5] and coef_out f32[32] as stj_loss_fwd writes them (coef_out == coef_in)
extern "C" int stj_loss_fwd_bwd(const float* logits, const float* gt_obs, const float* gt_occ, const float* gt_flow, const float* origin,
                                const float* gate, const float* coef_in, float* sums, float* loss, float* coef_out, float* dlogits,
                                int B, int H, int W, float ogm_w, float occ_w, float flow_origin_w, float replica, int flags,
                                hipStream_t stream) {
  if ((((uintptr_t)logits) | ((uintptr_t)dlogits)) & 15) { stj_set_error("loss: logits / dlogits must be 16-byte aligned"); return STJ_EINVAL; }
  if (((uintptr_t)gt_flow) & 7) { stj_set_error("loss: gt_flow must be 8-byte aligned"); return STJ_EINVAL; }
  if (flags & ~7) { stj_set_error("loss: unknown flag bits %d", flags); return STJ_EINVAL; }
  const long long npix = (long long)B * H * W;
  // one (pixel, waypoint) item per thread up to 16384 workgroups: waves that load, store and end (the streaming probe)
  const int gx = (int)min((long long)LOSS_FB_MAXG, (npix * NWP + 255) / 256);
  const int use_warp = flags & 1, focal = (flags >> 1) & 1, pred = (flags >> 2) & 1;
  if (gx > 0) {
#define LOSS_FB(FO, PR) hipLaunchKernelGGL((loss_fwd_bwd_kernel<FO, PR>), dim3(gx), dim3(256), 0, stream, logits, gt_obs, gt_occ, gt_flow, origin, coef_in, sums, dlogits, B, H, W, use_warp)
    if (focal && pred) LOSS_FB(true, true); else if (focal) LOSS_FB(true, false); else if (pred) LOSS_FB(false, true); else LOSS_FB(false, false);
#undef LOSS_FB
  }
  if (loss == nullptr) return stj_check_launch("stj_loss_fwd_bwd");        // the caller finishes with stj_loss_finalize (e.g. on another stream)
  LossCfg c; c.ogm_w = ogm_w; c.occ_w = occ_w; c.fow = flow_origin_w; c.replica = replica; c.use_warp = use_warp;
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, stream, sums, LOSS_PARTS_FUSED, gate, loss, coef_out, (float)npix, c);
  return stj_check_launch("stj_loss_fwd_bwd");
}
// the last launch of stj_loss_fwd_bwd on its own (loss == NULL there): nothing on the backward path reads what it writes, so a caller may
// issue it on a side stream behind the pass; sums = the pass's f32[128*40]
extern "C" int stj_loss_finalize(const float* sums, const float* gate, float* loss, float* coef_out, int B, int H, int W, float ogm_w,
                                 float occ_w, float flow_origin_w, float replica, int flags, hipStream_t stream) {
  if (flags & ~7) { stj_set_error("loss: unknown flag bits %d", flags); return STJ_EINVAL; }
  LossCfg c; c.ogm_w = ogm_w; c.occ_w = occ_w; c.fow = flow_origin_w; c.replica = replica; c.use_warp = flags & 1;
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, stream, sums, LOSS_PARTS_FUSED, gate, loss, coef_out, (float)((long long)B * H * W), c);
  return stj_check_launch("stj_loss_finalize");
}


// =====================================================================================================
// Evaluation metrics on device (reference occu_metric.py:26-140 compute_occupancy_flow_metrics, called every train and
// validation step: train.py:243-249,280-282): per waypoint k
//   observed / occluded PR-AUC(true, pred) and soft IoU (occu_metric.py:152-201), flow EPE (:204-252),
//   flow-warped occupancy: warped = sample(flow_origin_k, identity + pred_flow_k) (:255-317), grounded = clip(pred_obs +
//   pred_occ, 0, 1) * warped; AUC(y_true = grounded, y_pred = true_all) and IoU(grounded, true_all) -- the reference passes
//   the prediction in the y_true slot (:120-126); Keras casts y_true to bool (grounded != 0).
// One streaming pass accumulates 3 histograms + 10 sums per waypoint; stj_metrics_finalize turns them into the 7 means.
// pred is the model output [B,H,W,32]; pred_is_logits: occupancy channels are logits (sigmoid applied here, as
// _apply_sigmoid_to_occupancy_logits train.py:142-154 does) or already probabilities.
// =====================================================================================================
enum { M_IO = 0, M_TO, M_PO, M_IC, M_TC, M_PC, M_EPE, M_EX, M_IW, M_TW, M_PW, M_N };
__global__ __launch_bounds__(256) void metrics_kernel(const float* pred, const float* gt_obs, const float* gt_occ, const float* gt_flow,
                                                      const float* origin, int* hist, float* sums, int B, int H, int W,
                                                      int pred_is_logits, int use_warp) {
  __shared__ int sh[3 * 202];
  __shared__ float red[4][M_N];
  const int k = blockIdx.y;
  for (int i = threadIdx.x; i < 3 * 202; i += 256) sh[i] = 0;
  __syncthreads();
  float acc[M_N];
#pragma unroll
  for (int i = 0; i < M_N; ++i) acc[i] = 0.f;
  const long long npix = (long long)B * H * W;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < npix; i += gridDim.x * 256ll) {
    const int x = (int)(i % W); long long t = i / W;
    const int y = (int)(t % H); const long long b = t / H;
    const float4 q = reinterpret_cast<const float4*>(pred + i * 32)[k];
    const float po = pred_is_logits ? sigmoidf(q.x) : q.x, pc = pred_is_logits ? sigmoidf(q.y) : q.y;
    const long long g = ((b * NWP + k) * H + y) * W + x;
    const float to = gt_obs[g], tc = gt_occ[g];
    const float fx = gt_flow[2 * g], fy = gt_flow[2 * g + 1];
    atomicAdd(&sh[0 * 202 + (to != 0.f ? 101 : 0) + auc_bucket(po)], 1);
    atomicAdd(&sh[1 * 202 + (tc != 0.f ? 101 : 0) + auc_bucket(pc)], 1);
    acc[M_IO] += po * to; acc[M_TO] += to; acc[M_PO] += po;
    acc[M_IC] += pc * tc; acc[M_TC] += tc; acc[M_PC] += pc;
    const float ex = (fx != 0.f || fy != 0.f) ? 1.f : 0.f;
    const float dx = (fx - q.z) * ex, dy = (fy - q.w) * ex;
    acc[M_EPE] += sqrtf(dx * dx + dy * dy);
    acc[M_EX] += ex;
    if (use_warp) {
      const float* img = origin + (b * NWP + k) * (long long)H * W;
      const float wp = warp_sample(img, H, W, (float)x + q.z, (float)y + q.w, nullptr, nullptr);
      const float ta = fminf(fmaxf(to + tc, 0.f), 1.f);
      const float fg = fminf(fmaxf(po + pc, 0.f), 1.f) * wp;
      atomicAdd(&sh[2 * 202 + (fg != 0.f ? 101 : 0) + auc_bucket(ta)], 1);     // y_true = grounded prediction, y_pred = true_all
      acc[M_IW] += ta * fg; acc[M_TW] += fg; acc[M_PW] += ta;
    }
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < M_N; ++i) {
    const float s = wave_sum(acc[i]);
    if (lane == 0) red[w][i] = s;
  }
  __syncthreads();
  if (threadIdx.x < M_N) atomicAdd(sums + k * M_N + threadIdx.x, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
  for (int i = threadIdx.x; i < 3 * 202; i += 256)
    if (sh[i]) atomicAdd(hist + (k * 3 + i / 202) * 202 + i % 202, sh[i]);
}
// out[7] = observed_auc, occluded_auc, observed_iou, occluded_iou, flow_epe, flow_warped_occupancy_auc, flow_warped_occupancy_iou
__global__ void metrics_finalize_kernel(const float* auc, const float* sums, float* out, int use_warp) {
  if (threadIdx.x != 0) return;
  float m[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < NWP; ++k) {
    const float* s = sums + k * M_N;
    auto iou = [](float i, float t, float p) { const float d = p + t - i; return d != 0.f ? i / d : 0.f; };   // divide_no_nan
    m[0] += auc[k * 3 + 0]; m[1] += auc[k * 3 + 1];
    m[2] += iou(s[M_IO], s[M_TO], s[M_PO]); m[3] += iou(s[M_IC], s[M_TC], s[M_PC]);
    m[4] += s[M_EX] != 0.f ? s[M_EPE] / s[M_EX] : 0.f;
    if (use_warp) { m[5] += auc[k * 3 + 2]; m[6] += iou(s[M_IW], s[M_TW], s[M_PW]); }
  }
  for (int i = 0; i < 7; ++i) out[i] = m[i] / NWP;
}
// hist: int[8*3*202], sums: f32[8*11], auc: f32[24] scratch -- hist and sums MUST BE ZERO on entry.  out: f32[7].
extern "C" int stj_metrics(const float* pred, const float* gt_obs, const float* gt_occ, const float* gt_flow, const float* origin,
                           int* hist, float* sums, float* auc, float* out, int B, int H, int W, int pred_is_logits, int use_warp,
                           hipStream_t stream) {
  if (((uintptr_t)pred) & 15) { stj_set_error("metrics: pred must be 16-byte aligned"); return STJ_EINVAL; }
  const long long npix = (long long)B * H * W;
  if (npix <= 0) return STJ_OK;
  const int gx = (int)min(256ll, (npix + 255) / 256);
  hipLaunchKernelGGL(metrics_kernel, dim3(gx, NWP), dim3(256), 0, stream, pred, gt_obs, gt_occ, gt_flow, origin, hist, sums, B, H, W, pred_is_logits, use_warp);
  hipLaunchKernelGGL(auc_gate_kernel, dim3(NWP * 3), dim3(128), 0, stream, hist, (float*)nullptr, auc);
  hipLaunchKernelGGL(metrics_finalize_kernel, dim3(1), dim3(64), 0, stream, auc, sums, out, use_warp);
  return stj_check_launch("stj_metrics");
}
