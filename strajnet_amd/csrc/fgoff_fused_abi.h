// C-ABI argument block of the fused FG-MSA offset head (fgoff_fused.hip); the same declaration is in include/strajnet_hip.h.
#pragma once
#include <hip/hip_runtime_api.h>
extern "C" {
typedef struct stj_fgoff_args {
  int B, H, W, dtype;                        // q is [B][H][W][384]; W = 8 | 16 | 32 (stj_fgoff_supported)
  float scale, eps;                          // offset range (H / 2, FG_MSA.py:139) and the LayerNorm epsilon (1e-3)
  const void* q;                             // forward input [B][H][W][384], activation dtype
  const void* pack;                          // stj_fgoff_pack output
  const float* bias;                         // conv_offset_0/bias [384]                 (f32 masters)
  const float* gamma; const float* beta;     // conv_norm gamma / beta [384]
  const void* w1;                            // conv_offset_proj/kernel [48][2], ACTIVATION dtype
  void* off;                                 // [B][8][H W][2]   (forward: out; backward: in)
  void* cols;                                // training: im2col of q, [B H W][8][432] (the conv's weight gradient reads it); NULL: not written
  void* c; float* mean; float* rstd;         // training: conv output + bias [B H W][384] and its LayerNorm statistics [B H W]; NULL: inference
  /* backward only */
  const void* doff;                          // gradient of off [B][8][H W][2]
  void* dc;                                  // written: gradient of the conv output [B H W][384] (dW = cols^T dc is the caller's GEMM)
  void* dq;                                  // written: gradient of q [B][H][W][384]
  float* d_w1; float* d_gamma; float* d_beta; float* d_bias;      // += (atomics): [48][2], [384], [384], [384] f32
} stj_fgoff_args;
}
