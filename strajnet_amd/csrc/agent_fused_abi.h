// C-ABI argument blocks of the fused agent-branch kernels (agent_fused.hip); the same declarations are in include/strajnet_hip.h.
#pragma once
#include <hip/hip_runtime_api.h>
extern "C" {
typedef struct stj_agent_weights {          // f32 masters (views of the flat parameter buffer), Keras / tfa layouts
  const float* e_wq; const float* e_wk; const float* e_wv;     // traj_encoder/node_attention query | key | value kernels [4][64][64]
  const float* e_wo;                                            // .../projection_kernel [4][64][320]
  const float* e_ws;                                            // traj_encoder/sublayer/kernel [384][384]
  const float* i_wq; const float* i_wk; const float* i_wv;     // cross_attention/mha query | key | value kernels [6][384][64]
  const float* i_wo;                                            // .../projection_kernel [6][64][384]
  const float* i_w1; const float* i_w2;                         // FFN1/kernel [384][1536], FFN2/kernel [1536][384]
} stj_agent_weights;

typedef struct stj_agent_enc_args {
  const float* obs; const float* occ;        // tracks [B][n_obs][11][8], [B][n_occ][11][8] f32
  int n_obs, n_occ, B, dtype;
  const void* pack;                          // stj_agent_pack output
  const float* wn; const float* bn;          // node_feature kernel [1][5][64] + bias (f32 masters)
  const float* wv3;                          // vector_feature kernel [3][64]
  const float* bo; const float* bs;          // node_attention projection_bias [320], sublayer bias [384]
  void* enc;                                 // [B (n_obs + n_occ)][384] (forward: out; backward: in)
  int* cmi;                                  // [B (n_obs + n_occ)] agent has a valid step (forward: out)
  void* s_nodes; void* s_qkv; void* s_att; void* s_pmask; void* s_cat;     // saved for backward ([rows 11][64], [rows 11][768], [rows 11][256], uint16 [agents][320], [agents][384]); all NULL: inference
  const long long* rng_state; int site; float p_drop;                    // attention dropout, drawn as stj_dropout draws [agents][4][11][11]
  /* backward only */
  const void* d_enc; int d_enc_f32;          // [agents][384] in the activation dtype (d_enc_f32 = 0), or d_enc_f32 f32 slabs [d_enc_f32][agents][384] to be added
  const void* wq; const void* wk; const void* wv; const void* wo; const void* ws;      // the kernels in the activation dtype, natural layouts
  void* dpre_s; void* dout; void* dqkv;      // written: dY of sublayer [agents][384], projection [rows 11][320], q|k|v [rows 11][768]
  float* dwn; float* dbn; float* dwv3;       // += (atomics)
} stj_agent_enc_args;

typedef struct stj_agent_int_args {
  const void* enc; const int* cmi;           // stj_agent_enc_fwd outputs [B 64][384], [B 64]
  int n_obs, n_occ, B, dtype;                // n_obs + n_occ == 64; 16-bit dtypes
  const void* pack;
  const void* seg;                           // seg_embed kernel [2][384], ACTIVATION dtype
  const float* bo; const float* g1; const float* be1; const float* b1; const float* b2; const float* g2; const float* be2;     // cross_attention: projection_bias, norm1, FFN1 / FFN2 bias, norm2 (f32 masters)
  const float* g_obs; const float* b_obs; const float* g_occ; const float* b_occ;                                             // obs_norm | occ_norm
  void* key;                                 // out [B 64][384]
  float* ws_v1; float* ws_u2;                // forward workspaces: [6][B 64][384] and [4][B 64][384] f32 (per head / per hidden chunk partial sums, added in a fixed order: no atomics)
  void* s_concat; void* s_qin; void* s_q; void* s_k; void* s_v; void* s_att; void* s_v1; void* s_n1; void* s_h; void* s_u2; void* s_out;   // saved for backward ([B 64][384]; s_h [B 64][1536]); all NULL: inference
  const long long* rng_state; int site_a, site_1, site_2; float p_drop;      // dropout sites: coefficients [B][6][64][64], after FFN1 [B 64][1536], after FFN2 [B 64][384]
  /* backward only */
  const void* dkey;
  const void* wq; const void* wk; const void* wv; const void* wo; const void* w1; const void* w2;      // natural layouts, activation dtype
  float* d_enc;                              // written: gradient of enc as 7 slabs [7][B 64][384] f32 (stj_agent_enc_bwd with d_enc_f32 = 7 adds them)
  float* ws_dn1;                             // backward workspace [4][B 64][384] f32
  void* dq; void* dk; void* dv; void* dv1; void* dpre1; void* dz2;          // written: dY of the q / k / v projections, the output projection, FFN1, FFN2
  float* dseg; float* dg1; float* dbe1; float* dg2; float* dbe2; float* dg_obs; float* db_obs; float* dg_occ; float* db_occ;      // += (atomics)
} stj_agent_int_args;
}
