// split3.hpp — what csrc/gemm3.hip and csrc/conv3.hip share: the exact three-way bf16 split of f32 values, the
// bf16 MFMA wrapper and the LDS-only barrier (see gemm3.hip's header for the method).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mirl {

typedef __bf16 g3_bf16x8 __attribute__((ext_vector_type(8)));
typedef float g3_f32x16 __attribute__((ext_vector_type(16)));
typedef float g3_f32x4 __attribute__((ext_vector_type(4)));

constexpr int G3_PITCH = 48;                 // bytes per LDS row: 16 bf16 + 16 B pad
constexpr int G3_PLANE = 256 * G3_PITCH;     // one part of one operand
constexpr int G3_STAGE = 6 * G3_PLANE;
constexpr int G3_LDS = 2 * G3_STAGE;
constexpr int G3_EPITCH = 68;               // floats per row of a wave's 64 x 64 epilogue transpose (8 x 17 408 B <= G3_LDS)

__device__ __forceinline__ unsigned g3_pk(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

// x[0..3] (four consecutive k of one row) -> three 8-byte groups of bf16 parts
__device__ __forceinline__ void g3_split4(const float (&x)[4], uint2& h, uint2& m, uint2& l) {
  unsigned ph[2], pm[2], pl[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const float x0 = x[2 * e], x1 = x[2 * e + 1];
    ph[e] = g3_pk(x0, x1);
    const float r0 = x0 - __uint_as_float(ph[e] << 16), r1 = x1 - __uint_as_float(ph[e] & 0xffff0000u);
    pm[e] = g3_pk(r0, r1);
    const float s0 = r0 - __uint_as_float(pm[e] << 16), s1 = r1 - __uint_as_float(pm[e] & 0xffff0000u);
    pl[e] = g3_pk(s0, s1);
  }
  h = make_uint2(ph[0], ph[1]); m = make_uint2(pm[0], pm[1]); l = make_uint2(pl[0], pl[1]);
}

__device__ __forceinline__ g3_f32x16 g3_mfma(g3_bf16x8 a, g3_bf16x8 b, g3_f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ void g3_barrier() {
  // LDS writes of this wave done, then the workgroup barrier; global loads stay in flight (no vmcnt wait)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

}  // namespace mirl
