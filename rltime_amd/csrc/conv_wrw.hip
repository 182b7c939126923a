// conv_wrw.hip — weight gradient of the middle conv layers (NHWC, f32 operands) on the bf16 matrix pipe with f32 results.
//
// What it replaces: MIOpen's igemm_wrw kernels behind autograd for conv layers 2 and 3 of the Atari models
// (rltime/models/torch/modules/cnn.py:43-50; configs/models/cnn_*.json: 32 -> 64 filters k 4 s 2 on 20 x 20,
// 64 -> 64 k 3 s 1 on 9 x 9): 1.83 + 1.44 ms of the learner step at BASELINE configs[3] on the f32 MFMA pipe.
//
//   dW[f][(kh, kw, c)] = sum over (n, oh, ow) of g[n][oh][ow][f] * x[n][S oh + kh][S ow + kw][c]
//
// GEMM view: D[64 filters][T taps] = G^T [64][P] . Patch [P][T] with the reduction over the P output positions, tap
// index T = (kh KW + kw) C + c — the order a channels_last weight is stored in, and the order in which an NHWC
// activation holds a window row, so Patch[p][T] = x_frame[base(p) + (kh W + kw) C + c]: no im2col, no transposition.
// Both operands are f32: csrc/gemm3.hip's exact three-way bf16 split of both (six part products, f32 accumulation),
// on v_mfma_f32_16x16x32_bf16 — rows = 16 filters, columns = 16 taps, K = 32 positions per instruction.
//
//   * a workgroup (8 waves) takes FPI frames per LDS fill.  Both operands are split ONCE while they are staged:
//     x as three bf16 planes in its own NHWC order with C + 2 halfwords per pixel (the four lane quarters of a B read
//     are 8 positions = 8 S pixels apart: with C halfwords per pixel that is a multiple of 128 bytes and all four hit
//     the same banks); g TRANSPOSED to [filter][position] so that an A fragment (8 consecutive positions of one filter)
//     is one 16-byte read.  Positions of the FPI frames form one flat list padded
//     with zero g to a multiple of 32; a table (built once) gives every position's window origin in the x planes.
//   * wave w owns the tap tiles [w TPW, (w + 1) TPW) for ALL 64 filters and ALL positions: 4 x TPW accumulator tiles
//     (TPW <= 5: 80 VGPRs) that live in registers for the whole launch — no cross-wave reduction; per K-step it reads
//     the 12 A fragments (4 filter tiles x 3 parts) once and, per tap tile, 3 x 8 two-byte B values (the 8 positions'
//     window origins + the lane's tap offset), then 24 MFMAs; the next tile's B reads are issued before them.
//   * each workgroup writes one [64][T] slab at the end; k_conv_wrw_reduce sums the slabs in index order — fixed
//     partition, fixed order, no float atomics: bit-identical reruns.
#include "common.hpp"
#include "split3.hpp"
#include <stdlib.h>

namespace mirl {

typedef float cw_f4 __attribute__((ext_vector_type(4)));

constexpr int CW_F = 64;            // filters (4 MFMA row tiles)
constexpr int CW_MAXT = 5;          // tap tiles per wave

struct CwArgs {
  const float* x; const float* g; float* partial;
  int N, H, W, C, KH, KW, S, OH, OW;
  int T;              // taps = KH * KW * C
  int NT;             // tap tiles = T / 16
  int FPI;            // frames per LDS fill
  int KP;             // positions per fill, padded to a multiple of 32
  int Cp;             // halfwords per pixel in the x planes: C + 2 (see k_conv_wrw_b3)
  int xplane;         // halfwords per x plane (FPI * H * W * Cp, padded)
  int gpitch;         // halfwords per filter row of a g plane (KP + 8)
};

template <int TPW>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_conv_wrw_b3(CwArgs a) {
  extern __shared__ __align__(16) char cw_lds[];
  uint16_t* xs = reinterpret_cast<uint16_t*>(cw_lds);                       // [3][xplane]
  uint16_t* gs = xs + 3 * a.xplane;                                         // [3][64][gpitch]
  int* tbl = reinterpret_cast<int*>(gs + 3 * CW_F * a.gpitch);              // [KP] window origins (halfword index in a plane)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kq = lane >> 4;
  const int OHW = a.OH * a.OW, fx = a.H * a.W * a.C, gplane = CW_F * a.gpitch;
  const int ksteps = a.KP >> 5;
  for (int p = tid; p < a.KP; p += 512) {
    const int f = p / OHW, r = p - f * OHW, oh = r / a.OW, ow = r - oh * a.OW;
    tbl[p] = f < a.FPI ? ((f * a.H + a.S * oh) * a.W + a.S * ow) * a.Cp : 0;
  }
  // frames a short last unit does not bring keep whatever the planes held: zero them once (g is zero there, but 0 x NaN is not)
  for (int o = tid; o < 3 * a.xplane / 8; o += 512) reinterpret_cast<uint4*>(xs)[o] = make_uint4(0u, 0u, 0u, 0u);
  // this wave's tap tiles and the lane's tap offset inside each
  const int nt0 = wave * TPW;
  int toff[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    int t = (nt0 + i) * 16 + j; if (t > a.T - 1) t = a.T - 1;
    const int c = t % a.C, kk = t / a.C, kw = kk % a.KW, kh = kk / a.KW;
    toff[i] = (kh * a.W + kw) * a.Cp + c;
  }
  g3_f32x4 acc[4][TPW];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int i = 0; i < TPW; ++i) acc[m][i] = g3_f32x4{0.f, 0.f, 0.f, 0.f};

  const int units = (a.N + a.FPI - 1) / a.FPI;
  auto put_x = [&](const cw_f4& v, int o) {                                 // 4 consecutive channels of one pixel of the unit's x block
    const float q[4] = {v.x, v.y, v.z, v.w};
    uint2 h, m, l;
    g3_split4(q, h, m, l);
    const int pix = (o * 4) / a.C, c0 = o * 4 - pix * a.C;
    uint32_t* d = reinterpret_cast<uint32_t*>(xs + pix * a.Cp + c0);          // 4-byte aligned: Cp and c0 are even
    d[0] = h.x; d[1] = h.y;
    d[a.xplane / 2] = m.x; d[a.xplane / 2 + 1] = m.y;
    d[a.xplane] = l.x; d[a.xplane + 1] = l.y;
  };
  // g transposed to [filter][position]: one item = 8 consecutive positions of one filter — eight loads (a wave's lanes =
  // 64 consecutive filters: 256 contiguous bytes each), one split, ONE 16-byte store per part (a float4-of-filters item
  // scattered 12 two-byte stores over rows 832 bytes apart: 8-way bank conflicts, as long as the MFMAs of the unit)
  auto load_g8 = [&](float (&q)[8], const float* gu, int item, int valid_pos) {
    const int f = item & 63, p0 = (item >> 6) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int p = p0 + e;
      const float t = gu[(int64_t)(p < valid_pos ? p : valid_pos - 1) * CW_F + f];
      q[e] = p < valid_pos ? t : 0.f;
    }
  };
  auto put_g8 = [&](const float (&q)[8], int item) {
    const int f = item & 63, p0 = (item >> 6) * 8;
    const float q0[4] = {q[0], q[1], q[2], q[3]}, q1[4] = {q[4], q[5], q[6], q[7]};
    uint2 h0, m0, l0, h1, m1, l1;
    g3_split4(q0, h0, m0, l0);
    g3_split4(q1, h1, m1, l1);
    uint16_t* d = gs + f * a.gpitch + p0;
    *reinterpret_cast<uint4*>(d) = make_uint4(h0.x, h0.y, h1.x, h1.y);
    *reinterpret_cast<uint4*>(d + gplane) = make_uint4(m0.x, m0.y, m1.x, m1.y);
    *reinterpret_cast<uint4*>(d + 2 * gplane) = make_uint4(l0.x, l0.y, l1.x, l1.y);
  };
  for (int u0 = blockIdx.x; u0 < units; u0 += gridDim.x) {
    const int n0 = u0 * a.FPI;
    const int frames = a.N - n0 < a.FPI ? a.N - n0 : a.FPI;
    const int xv = frames * fx / 4;
    __syncthreads();                              // table built / every wave done with the previous unit
    {
      // both operands split once while they are staged.  EVERY global load of the fill is issued before the first split
      // (x: coalesced 16-byte vectors; g: see put_g8): with one workgroup per CU nothing else covers a load's latency, so
      // the fill pays it once, not once per pass
      const cw_f4* x4 = reinterpret_cast<const cw_f4*>(a.x + (int64_t)n0 * fx);
      const float* gu = a.g + (int64_t)n0 * OHW * CW_F;
      const int items = (a.KP >> 3) * CW_F, valid_pos = frames * OHW;
      constexpr int LX = 7, LG = 2;
      cw_f4 v[LX];
      float q[LG][8];
#pragma unroll
      for (int k = 0; k < LX; ++k) { const int o = tid + k * 512; v[k] = x4[o < xv ? o : xv - 1]; }
#pragma unroll
      for (int k = 0; k < LG; ++k) { const int it = tid + k * 512; load_g8(q[k], gu, it < items ? it : items - 1, valid_pos); }
#pragma unroll
      for (int k = 0; k < LX; ++k) { const int o = tid + k * 512; if (o < xv) put_x(v[k], o); }
#pragma unroll
      for (int k = 0; k < LG; ++k) { const int it = tid + k * 512; if (it < items) put_g8(q[k], it); }
      // fills larger than the vectors held above
      for (int o0 = tid + LX * 512; o0 < xv; o0 += 512) put_x(x4[o0], o0);
      for (int it = tid + LG * 512; it < items; it += 512) { float r[8]; load_g8(r, gu, it, valid_pos); put_g8(r, it); }
    }
    __syncthreads();
    for (int ks = 0; ks < ksteps; ++ks) {
      const int p0 = ks * 32 + kq * 8;
      int org[8];
      {
        const int4 t0 = *reinterpret_cast<const int4*>(tbl + p0), t1 = *reinterpret_cast<const int4*>(tbl + p0 + 4);
        org[0] = t0.x; org[1] = t0.y; org[2] = t0.z; org[3] = t0.w; org[4] = t1.x; org[5] = t1.y; org[6] = t1.z; org[7] = t1.w;
      }
      g3_bf16x8 af[4][3];
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          af[m][p] = *reinterpret_cast<const g3_bf16x8*>(gs + p * gplane + (16 * m + j) * a.gpitch + p0);
      uint32_t raw[2][3][8];
      auto read_b = [&](uint32_t (&r)[3][8], int i) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int e = 0; e < 8; ++e) r[p][e] = xs[p * a.xplane + org[e] + toff[i]];
      };
      read_b(raw[0], 0);
#pragma unroll
      for (int i = 0; i < TPW; ++i) {
        if (i + 1 < TPW) read_b(raw[(i + 1) & 1], i + 1);
        __builtin_amdgcn_sched_barrier(0);
        g3_bf16x8 bf[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          const uint32_t* r = raw[i & 1][p];
          bf[p] = __builtin_bit_cast(g3_bf16x8, make_uint4(r[0] | (r[1] << 16), r[2] | (r[3] << 16), r[4] | (r[5] << 16), r[6] | (r[7] << 16)));
        }
        // smallest products first: (lo,hi) (hi,lo) (mid,mid) (mid,hi) (hi,mid) (hi,hi); the four filter tiles are independent chains
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
        constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int c = 0; c < 6; ++c)
#pragma unroll
          for (int m = 0; m < 4; ++m)
            acc[m][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m][PA[c]], bf[PB[c]], acc[m][i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  // the workgroup's slab: accumulator tile (filter tile m, tap tile nt0 + i): row 4 kq + r <-> filter 16 m + 4 kq + r,
  // column j <-> tap 16 (nt0 + i) + j
  float* out = a.partial + (int64_t)blockIdx.x * CW_F * a.T;
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int t = (nt0 + i) * 16 + j;
    if (nt0 + i < a.NT) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(int64_t)(16 * m + 4 * kq + r) * a.T + t] = acc[m][i][r];
    }
  }
}

// dw[i] = sum over the slabs of partial[s][i] in slab order, i over 64 * T (four floats per thread)
__global__ void __launch_bounds__(256)
k_conv_wrw_reduce(const cw_f4* __restrict__ partial, int parts, int64_t n4, cw_f4* __restrict__ dw) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  cw_f4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  int p = 0;
  for (; p + 4 <= parts; p += 4) {
    s0 += partial[(int64_t)p * n4 + i]; s1 += partial[(int64_t)(p + 1) * n4 + i];
    s2 += partial[(int64_t)(p + 2) * n4 + i]; s3 += partial[(int64_t)(p + 3) * n4 + i];
  }
  for (; p < parts; ++p) s0 += partial[(int64_t)p * n4 + i];
  dw[i] = (s0 + s1) + (s2 + s3);
}

static int cw_geometry(int64_t N, int H, int W, int C, int F, int KH, int KW, int S, CwArgs* a, size_t* lds, unsigned* grid) {
  if (F != CW_F || C < 4 || (C % 4) || KH < 1 || KW < 1 || S < 1 || H < KH || W < KW) return 0;
  const int T = KH * KW * C;
  if (T % 16) return 0;
  const int NT = T / 16;
  if (NT > 8 * CW_MAXT) return 0;
  const int OH = (H - KH) / S + 1, OW = (W - KW) / S + 1, OHW = OH * OW;
  if ((OHW * CW_F) % 4) return 0;
  // frames per fill: the count (<= 4) that wastes the fewest padded positions among those whose fill fits the LDS
  // (ties: the larger count — fewer fills)
  int best = 0;
  size_t best_lds = 0;
  double best_waste = 0.0;
  for (int fpi = 1; fpi <= 4; ++fpi) {
    const int KP = (fpi * OHW + 31) / 32 * 32;
    const int xplane = (fpi * H * W * (C + 2) + 7) / 8 * 8;
    const size_t need = (size_t)3 * xplane * 2 + (size_t)3 * CW_F * (KP + 8) * 2 + (size_t)KP * 4;
    if (need > 150 * 1024) break;
    const double waste = (double)KP / (fpi * OHW);
    if (!best || waste <= best_waste + 1e-9) { best = fpi; best_lds = need; best_waste = waste; }
  }
  if (!best) return 0;
  if (a) {
    a->N = (int)N; a->H = H; a->W = W; a->C = C; a->KH = KH; a->KW = KW; a->S = S; a->OH = OH; a->OW = OW;
    a->T = T; a->NT = NT; a->FPI = best; a->KP = (best * OHW + 31) / 32 * 32;
    a->Cp = C + 2; a->xplane = (best * H * W * (C + 2) + 7) / 8 * 8; a->gpitch = a->KP + 8;
    const int64_t units = (N + best - 1) / best;
    *grid = (unsigned)(units < 256 ? units : 256);
    *lds = best_lds;
  }
  return 1;
}

}  // namespace mirl

extern "C" int mirl_conv_wrw_b3_supported(int32_t C, int32_t F, int32_t KH, int32_t KW, int32_t S, int32_t H, int32_t W) {
  return mirl::cw_geometry(1, H, W, C, F, KH, KW, S, nullptr, nullptr, nullptr);
}

extern "C" int mirl_conv_wrw_b3_scratch_bytes(int32_t C, int32_t F, int32_t KH, int32_t KW, int64_t* bytes) {
  if (!bytes || F != mirl::CW_F || C < 1 || KH < 1 || KW < 1) return mirl::fail(MIRL_ERR_ARG, "conv_wrw_b3_scratch_bytes: bad arguments");
  *bytes = (int64_t)256 * F * KH * KW * C * (int64_t)sizeof(float);       // one [64][T] slab per workgroup
  return MIRL_OK;
}

extern "C" int mirl_conv_wrw_b3(int64_t N, int32_t H, int32_t W, int32_t C, int32_t F, int32_t KH, int32_t KW, int32_t S,
                                const float* x, const float* g, void* scratch, int64_t scratch_bytes, float* dw, void* stream) {
  using namespace mirl;
  CwArgs a;
  size_t lds = 0;
  unsigned grid = 0;
  if (N <= 0 || N >= (1LL << 30) || !x || !g || !scratch || !dw) return fail(MIRL_ERR_ARG, "bad conv_wrw_b3 arguments");
  if (!cw_geometry(N, H, W, C, F, KH, KW, S, &a, &lds, &grid)) return fail(MIRL_ERR_ARG, "conv_wrw_b3: unsupported layer shape");
  if (((uintptr_t)x % 16) || ((uintptr_t)g % 16) || ((uintptr_t)scratch % 16) || ((uintptr_t)dw % 16))
    return fail(MIRL_ERR_ARG, "conv_wrw_b3: pointers must be 16-byte aligned");
  if (scratch_bytes < (int64_t)grid * CW_F * a.T * (int64_t)sizeof(float)) return fail(MIRL_ERR_ARG, "conv_wrw_b3: scratch too small");
  a.x = x; a.g = g; a.partial = (float*)scratch;
  hipStream_t st = (hipStream_t)stream;
  const int tpw = (a.NT + 7) / 8;
  const void* fns[CW_MAXT] = {(const void*)k_conv_wrw_b3<1>, (const void*)k_conv_wrw_b3<2>, (const void*)k_conv_wrw_b3<3>,
                              (const void*)k_conv_wrw_b3<4>, (const void*)k_conv_wrw_b3<5>};
  static bool attr[CW_MAXT] = {false, false, false, false, false};
  const void* fn = fns[tpw - 1];
  if (!attr[tpw - 1]) { MIRL_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); attr[tpw - 1] = true; }
  {
    ProfScope ps("k_conv_wrw_b3", 4.0 * (double)N * ((double)H * W * C + (double)a.OH * a.OW * CW_F), st,
                 2.0 * (double)N * a.OH * a.OW * CW_F * a.T);
    void* kargs[] = {(void*)&a};
    MIRL_HIP(hipLaunchKernel(fn, dim3(grid), dim3(512), kargs, lds, st));
  }
  ProfScope ps("k_conv_wrw_reduce", (double)(grid + 1) * CW_F * a.T * 4.0, st);
  const int64_t n4 = (int64_t)CW_F * a.T / 4;
  hipLaunchKernelGGL(k_conv_wrw_reduce, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const cw_f4*>(scratch),
                     (int)grid, n4, reinterpret_cast<cw_f4*>(dw));
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}
