// Counter-based random stream shared by the dropout kernels (rng.hip) and the fused Swin kernels (swin_fused.hip):
// Philox-4x32-10, key = (seed, step), counter = (draw / 4, site).  See rng.hip for the semantics of a "site".
#pragma once
#include "common.h"

__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
    c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
    k.x += 0x9E3779B9u; k.y += 0xBB67AE85u;
  }
  return c;
}
// the four keep decisions of draw group `grp` (draws 4*grp .. 4*grp+3) of a site
__device__ __forceinline__ void keep4(const long long* state, int site, long long grp, float p, bool k[4]) {
  const uint2 key = make_uint2((uint32_t)state[0] ^ (uint32_t)((unsigned long long)state[0] >> 32) * 0x9E3779B9u, (uint32_t)state[1]);
  const uint4 r = philox4x32_10(make_uint4((uint32_t)grp, (uint32_t)((unsigned long long)grp >> 32), (uint32_t)site, 0x53544a4eu), key);
  const float s = 1.0f / 16777216.0f;
  k[0] = (float)(r.x >> 8) * s >= p; k[1] = (float)(r.y >> 8) * s >= p;
  k[2] = (float)(r.z >> 8) * s >= p; k[3] = (float)(r.w >> 8) * s >= p;
}


// DropPath factor of sample `b` at site `site` (reference modules.py:137-151: x / keep * floor(keep + U)); p = 1 - keep.
__device__ __forceinline__ float drop_path_scale(const long long* state, int site, long long b, float p) {
  if (state == nullptr || p <= 0.f) return 1.f;
  bool k[4];
  keep4(state, site, b >> 2, p, k);
  return k[b & 3] ? 1.0f / (1.0f - p) : 0.f;
}
