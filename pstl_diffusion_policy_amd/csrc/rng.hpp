// rng.hpp -- counter-based normal noise for the reverse diffusion (Philox4x32-7 + Box-Muller).
//
// The reference draws its noise from torch's global generator (nusc_train.py:563,584), which cannot be reproduced
// bit for bit on another device; parity tests therefore pass the noise in.  In production ("PSTL_FLAG_RNG") the
// kernels draw it themselves: the four values of outputs f0..f0+3 of GLOBAL row R at reverse step i are a pure function
// of (seed, R, f0/4, i), so results do not depend on tiling, launch segmentation or how scenes are sharded over GPUs.
#pragma once
#include <stdint.h>

namespace pstl {

struct u32x4 {
  uint32_t x, y, z, w;
};

// Rounds of the Philox4x32 bijection.  Round 6: 7 instead of 10 -- Philox4x32-7 is the variant Salmon et al. (SC'11, "Parallel
// random numbers: as easy as 1, 2, 3", table 2) report as passing every test of BigCrush; 10 is their default with a margin.  The
// rounds are a third of what the denoiser kernel issues beside its MFMAs per noise quad (csrc/chain2_kernels.hip: 52 -> 40
// single-instruction steps), i.e. ~3 % of the dominant launch; the stream is this library's own (the reference draws from torch's
// generator, which no other device reproduces), every consumer goes through this one constant, and tests/test_gpu_noise_quality.py
// holds the drawn values to the moments and independence a sampler needs.
#ifndef PSTL_PHILOX_ROUNDS
#define PSTL_PHILOX_ROUNDS 7      // (-DPSTL_PHILOX_ROUNDS=10: the A/B build of tools/dbg/build_full_variant.sh)
#endif
constexpr int kPhiloxRounds = PSTL_PHILOX_ROUNDS;

__host__ __device__ inline u32x4 philox4x32(u32x4 c, uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < kPhiloxRounds; ++r) {
    const uint64_t p0 = (uint64_t)M0 * c.x, p1 = (uint64_t)M1 * c.z;
    u32x4 n;
    n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
    n.y = (uint32_t)p1;
    n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
    n.w = (uint32_t)p0;
    c = n;
    k0 += W0;
    k1 += W1;
  }
  return c;
}

// A 64-bit / 32-bit run-time parameter kept in device memory (pstl_dyn, include/pstl_hip.h), read as a wave-UNIFORM value:
// a plain load through a global pointer that may alias the kernel's stores is a vector load (the seed would sit in vector
// registers and every Philox round key would be vector arithmetic); readfirstlane puts it where a by-value argument is.
__device__ inline uint64_t uniform_u64(const void* p) {
  const uint32_t* q = static_cast<const uint32_t*>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane(q[0]), hi = __builtin_amdgcn_readfirstlane(q[1]);
  return ((uint64_t)hi << 32) | lo;
}
__device__ inline float uniform_f32(const void* p) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(*static_cast<const uint32_t*>(p)));
}

// four independent N(0,1) values for (global row, quad = f0/4, step)
// (contraction is switched off inside: the function is compiled into two translation units with different
// -ffp-contract settings and must give the same bits in both)
__device__ inline void normal4(uint64_t seed, int64_t row, int quad, int step, float* out) {
#pragma clang fp contract(off)
  const uint64_t e = (uint64_t)row * 10u + (uint64_t)quad;
  const u32x4 r = philox4x32(u32x4{(uint32_t)e, (uint32_t)(e >> 32), (uint32_t)step, 0x5053544Cu}, (uint32_t)seed,
                                (uint32_t)(seed >> 32));
  const float k = 2.3283064365386963e-10f;  // 2^-32
  const float u0 = ((float)r.x + 1.0f) * k, u1 = (float)r.y * k;   // u0 in (0,1], u1 in [0,1]
  const float u2 = ((float)r.z + 1.0f) * k, u3 = (float)r.w * k;
  // single hardware instructions only (v_log_f32 = log2, v_sqrt_f32, v_sin/v_cos): library expansions of log/sqrt
  // depend on the translation unit's contraction setting, and both units must draw identical bits.
  // -2 ln(u) = (-2 ln 2) log2(u)
  const float c2 = -1.3862943611198906f;
  const float r0 = __builtin_amdgcn_sqrtf(c2 * __builtin_amdgcn_logf(u0));
  const float r1 = __builtin_amdgcn_sqrtf(c2 * __builtin_amdgcn_logf(u2));
  // v_sin_f32 / v_cos_f32 take their argument in revolutions: exactly the uniform variate
  out[0] = r0 * __builtin_amdgcn_cosf(u1);
  out[1] = r0 * __builtin_amdgcn_sinf(u1);
  out[2] = r1 * __builtin_amdgcn_cosf(u3);
  out[3] = r1 * __builtin_amdgcn_sinf(u3);
}

}  // namespace pstl
