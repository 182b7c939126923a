// conv3.hip — forward of the middle conv layers (NHWC, f32 in / out) as an implicit GEMM on the bf16 matrix pipe
// with f32 results: the method of csrc/gemm3.hip (exact three-way bf16 split of both operands while their tiles
// are staged, six part products accumulated in f32) with a convolution's addressing and a 64-column tile.
//
// What it replaces: `F.relu(conv(x))` of rltime/models/torch/modules/cnn.py:47-49 for the Atari models' layers 2 and 3
// (configs/models/cnn_*.json: 32 -> 64 filters k 4 s 2 on 20x20, 64 -> 64 k 3 s 1 on 9x9), which ran as MIOpen's f32
// implicit GEMM (119 TFLOP/s = 76 % of the f32 MFMA peak, 9.1 ms per learner step for the four passes) plus a separate
// in-place bias + ReLU pass (1.3 ms per step).  Here: rows m = (n, oh, ow), columns = the F <= 64 filters,
// K = KH * KW * C in (kh, kw, c) order, which is how both operands sit in memory: an NHWC activation row holds KW * C
// contiguous floats per kh, a channels_last weight holds [F][KH][KW][C].  Bias and ReLU ride in the epilogue.
//
// Tiling: workgroup = 256 rows x 64 filters, 8 waves of 32 rows x 64 columns (2 MFMA tiles, 32 accumulator
// registers), K in steps of 16 floats; the A row pointer jumps by (W - KW) * C floats after every KW * C (a multiple of
// 16).  LDS: two stages of [A 256 rows | B 64 rows] x 3 bf16 parts = 61 440 bytes (two workgroups per CU); the epilogue
// transposes each wave's block through 4 608 bytes of it into 16-byte row stores.
//
// What bounds it (round 6, profiles/r06_conv3_fwd_register_A_experiment.jsonl): with only F <= 64 columns the split of
// the A tile (4 096 floats per K-step, ~6 VALU operations each) is amortised over 96 MFMAs — ~520 VALU cycles per SIMD
// beside 768 MFMA cycles, from the same waves — and the two together hold the clock at ~1.5 GHz.  A variant whose A
// operand never touches LDS (fragments loaded straight from global memory, split in registers, 32 floats = whole 128-byte
// lines per iteration) measured the SAME 124-132 TFLOP/s and was removed: not LDS traffic, not L1 line splitting, not the
// barrier.  (131 TFLOP/s is 0.83 of the f32-input MFMA peak: an unsplit f32 kernel could not be faster either.)
#include "common.hpp"
#include "split3.hpp"
#include <stdlib.h>

namespace mirl {

// LDS rows are 32 bytes (16 bf16 of one part) with NO padding; the 16-byte half a k-group lands in is XOR-ed with
// bit 4 of the row, which makes both the 32-row ds_read_b128 fragments (served in 16-lane groups {0-3,12-15,20-27},
// {4-11,16-19,28-31} per half wave) and the 8-byte split writes (16 contiguous lanes = 4 rows x 4 k-quads) conflict
// free — and two stages fit in 61 440 bytes, so TWO workgroups share a CU (4 waves per SIMD).
constexpr int C3_PITCH = 32;
constexpr int C3_APL = 256 * C3_PITCH;              // one part of the A tile
constexpr int C3_BPL = 64 * C3_PITCH;               // one part of the B tile
constexpr int C3_STAGE = 3 * C3_APL + 3 * C3_BPL;   // 30 720
constexpr int C3_LDS = 2 * C3_STAGE;
constexpr int C3_EPITCH = 36;                       // floats per row of a wave's 32 x 32 epilogue transpose (8 x 4 608 B)
__device__ __forceinline__ int c3_wr_off(int row, int kq) { return row * C3_PITCH + ((kq ^ (((row >> 4) & 1) << 1)) << 3); }
__device__ __forceinline__ int c3_rd_off(int row, int khalf) { return row * C3_PITCH + ((khalf ^ ((row >> 4) & 1)) << 4); }

struct C3Args {
  const float* x; const float* w; const float* bias; float* y;
  int64_t M;              // N * OH * OW output positions
  int F, K;               // filters, KH * KW * C
  int H, W, C, OH, OW, S;
  int seg_steps;          // K-steps per contiguous (kw, c) run = KW * C / 16
  int64_t seg_jump;       // floats from the end of one run to the start of the next kh's = (W - KW) * C
  int relu;
};

__device__ __forceinline__ void c3_store3(char* planes, int plane_bytes, int off, const float (&v)[4]) {
  uint2 a, b, c;
  g3_split4(v, a, b, c);
  *reinterpret_cast<uint2*>(planes + off) = a;
  *reinterpret_cast<uint2*>(planes + plane_bytes + off) = b;
  *reinterpret_cast<uint2*>(planes + 2 * plane_bytes + off) = c;
}

// 12 MFMAs of one K-step on this wave's 32 x 64 block
__device__ __forceinline__ void c3_compute(const char* stage, g3_f32x16 (&acc)[2], int a_off, int b_off) {
  const char* pa = stage + a_off;
  const char* pb = stage + 3 * C3_APL + b_off;
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0};          // smallest products first: (lo,hi) (hi,lo) (mid,mid) (mid,hi) (hi,mid) (hi,hi)
  constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
  g3_bf16x8 a[3], b[3][2];
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    a[p] = *reinterpret_cast<const g3_bf16x8*>(pa + p * C3_APL);
#pragma unroll
    for (int j = 0; j < 2; ++j) b[p][j] = *reinterpret_cast<const g3_bf16x8*>(pb + p * C3_BPL + j * 32 * C3_PITCH);   // row + 32: same swizzle bit
  }
#pragma unroll
  for (int c = 0; c < 6; ++c)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[j] = g3_mfma(a[PA[c]], b[PB[c]][j], acc[j]);
}

__global__ void __launch_bounds__(512)
k_conv3_fwd(C3Args g) {
  extern __shared__ __attribute__((aligned(16))) char c3_lds[];
  const int t = threadIdx.x, lane = t & 63;
  const int wu = __builtin_amdgcn_readfirstlane(t >> 6);
  const int64_t m0 = (int64_t)blockIdx.x * 256;
  const int nk = g.K / 16;

  // A: rows r and r + 128 of the tile, k quad kq; B: row t >> 2 of the 64 filters (threads 0..255)
  const int r = t >> 2, kq = t & 3;
  const float* pa[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    int64_t m = m0 + r + 128 * h; if (m > g.M - 1) m = g.M - 1;
    const int64_t ohw = (int64_t)g.OH * g.OW;
    const int64_t n = m / ohw;
    const int rem = (int)(m - n * ohw);
    const int oh = rem / g.OW, ow = rem - oh * g.OW;
    pa[h] = g.x + ((n * g.H + (int64_t)g.S * oh) * g.W + (int64_t)g.S * ow) * g.C + kq * 4;
  }
  const bool has_b = t < 256;
  const int fb = r < g.F ? r : g.F - 1;
  const float* pb = g.w + (int64_t)fb * g.K + kq * 4;
  const int a_lds = c3_wr_off(r, kq), b_lds = c3_wr_off(r, kq);        // rows r and r + 128 share swizzle bit 4
  int seg_pos = 0;

  float va[2][4], vb[4];
  auto load = [&]() {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float4 q = *reinterpret_cast<const float4*>(pa[h]);
      va[h][0] = q.x; va[h][1] = q.y; va[h][2] = q.z; va[h][3] = q.w;
      pa[h] += 16;
    }
    if (++seg_pos == g.seg_steps) { seg_pos = 0; pa[0] += g.seg_jump; pa[1] += g.seg_jump; }
    if (has_b) {
      const float4 q = *reinterpret_cast<const float4*>(pb);
      vb[0] = q.x; vb[1] = q.y; vb[2] = q.z; vb[3] = q.w;
      pb += 16;
    }
  };
  auto store = [&](char* stage) {
    c3_store3(stage, C3_APL, a_lds, va[0]);
    c3_store3(stage, C3_APL, a_lds + 128 * C3_PITCH, va[1]);
    if (has_b) c3_store3(stage + 3 * C3_APL, C3_BPL, b_lds, vb);
  };

  g3_f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[j][q] = 0.0f;

  const int a_off = c3_rd_off(wu * 32 + (lane & 31), lane >> 5);
  const int b_off = c3_rd_off(lane & 31, lane >> 5);
  const bool stage_first = (wu >> 2) & 1;        // waves w and w + 4 share a SIMD and take opposite orders

  load();
  store(c3_lds);
  if (nk > 1) load();
  g3_barrier();
  for (int k = 0; k < nk; ++k) {
    const char* cur = c3_lds + (k & 1) * C3_STAGE;
    char* nxt = c3_lds + ((k + 1) & 1) * C3_STAGE;
    if (stage_first) {
      if (k + 1 < nk) store(nxt);
      if (k + 2 < nk) load();
    }
    c3_compute(cur, acc, a_off, b_off);
    if (!stage_first) {
      if (k + 1 < nk) store(nxt);
      if (k + 2 < nk) load();
    }
    g3_barrier();
  }

  // epilogue: transpose this wave's 32 x 64 block through LDS one 32-column half at a time, then + bias, ReLU and
  // 16-byte row stores (an accumulator lane owns ONE column: straight stores would be 32 four-byte instructions)
  float* tl = reinterpret_cast<float*>(c3_lds) + wu * (32 * C3_EPITCH);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
#pragma unroll
    for (int q = 0; q < 16; ++q)
      tl[((q & 3) + 8 * (q >> 2) + 4 * (lane >> 5)) * C3_EPITCH + (lane & 31)] = acc[j][q];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int c4 = lane & 7, col = j * 32 + c4 * 4;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.bias && col < g.F) bv = *reinterpret_cast<const float4*>(g.bias + col);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int rl = q * 8 + (lane >> 3);
      const int64_t m = m0 + wu * 32 + rl;
      float4 v = *reinterpret_cast<const float4*>(tl + rl * C3_EPITCH + c4 * 4);
      if (m < g.M && col < g.F) {
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        if (g.relu) { v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f; }
        *reinterpret_cast<float4*>(g.y + m * g.F + col) = v;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this half read before the next one overwrites it
  }
}

}  // namespace mirl

extern "C" int mirl_conv3_fwd_supported(int32_t C, int32_t F, int32_t KH, int32_t KW, int32_t S, int32_t H, int32_t W) {
  if (C < 4 || (C % 4) || F < 4 || F > 64 || (F % 4) || KH < 1 || KW < 1 || S < 1 || H < KH || W < KW) return 0;
  if ((KW * C) % 16) return 0;                    // a (kw, c) run is a whole number of K-steps
  if (KH * KW * C < 32) return 0;
  return 1;
}

extern "C" int mirl_conv3_fwd(int64_t N, int32_t H, int32_t W, int32_t C, int32_t F, int32_t KH, int32_t KW, int32_t S,
                              const float* x, const float* w, const float* bias, int32_t relu, float* y, void* stream) {
  using namespace mirl;
  if (N <= 0 || !x || !w || !y) return fail(MIRL_ERR_ARG, "bad conv3_fwd arguments");
  if (!mirl_conv3_fwd_supported(C, F, KH, KW, S, H, W)) return fail(MIRL_ERR_ARG, "conv3_fwd: unsupported layer shape");
  if (((uintptr_t)x % 16) || ((uintptr_t)w % 16) || ((uintptr_t)y % 16) || (bias && ((uintptr_t)bias % 16)))
    return fail(MIRL_ERR_ARG, "conv3_fwd: pointers must be 16-byte aligned");
  const int OH = (H - KH) / S + 1, OW = (W - KW) / S + 1;
  const int64_t M = N * OH * OW;
  if (M >= (1LL << 31) * 256) return fail(MIRL_ERR_ARG, "conv3_fwd: too many output positions for one launch");
  C3Args g;
  g.x = x; g.w = w; g.bias = bias; g.y = y; g.M = M; g.F = F; g.K = KH * KW * C;
  g.H = H; g.W = W; g.C = C; g.OH = OH; g.OW = OW; g.S = S;
  g.seg_steps = KW * C / 16; g.seg_jump = (int64_t)(W - KW) * C; g.relu = relu ? 1 : 0;
  hipStream_t st = (hipStream_t)stream;
  static bool attr = false;
  if (!attr) { MIRL_HIP(hipFuncSetAttribute((const void*)k_conv3_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, C3_LDS)); attr = true; }
  ProfScope ps("k_conv3_fwd", 4.0 * ((double)N * H * W * C + (double)M * F + (double)F * g.K), st, 2.0 * (double)M * F * g.K);
  hipLaunchKernelGGL(k_conv3_fwd, dim3((unsigned)((M + 255) / 256)), dim3(512), C3_LDS, st, g);
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}
