// philox.hpp — the counter-based generator of the device-RNG paths (replay sampling,
// the acting head's epsilon-greedy draws): stateless, keyed by (seed, call counter, lane).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mirl {

// Philox4x32-10 (Salmon et al., SC'11) for the device-RNG mode.
__device__ __forceinline__ void philox_round(uint32_t c[4], uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
  uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
  uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ double philox_u53(uint64_t seed, uint64_t call, uint32_t lane) {
  uint32_t c[4] = {lane, (uint32_t)call, (uint32_t)(call >> 32), 0x52544D45u};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  for (int r = 0; r < 10; ++r) { philox_round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
  // 53-bit uniform in [0,1) like MT19937's genrand_res53
  return ((double)(c[0] >> 5) * 67108864.0 + (double)(c[1] >> 6)) * (1.0 / 9007199254740992.0);
}

// the raw 4 x 32 bits of one counter value (acting head: a uniform and a random action per env)
__device__ __forceinline__ void philox_4x32(uint64_t seed, uint64_t call, uint32_t lane, uint32_t out[4]) {
  uint32_t c[4] = {lane, (uint32_t)call, (uint32_t)(call >> 32), 0x52544D45u};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  for (int r = 0; r < 10; ++r) { philox_round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
  out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

}  // namespace mirl
