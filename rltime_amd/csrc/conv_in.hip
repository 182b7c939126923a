// conv_in.hip — the network's input layer straight from the replay's uint8 frames.
//
// Reference: rltime/models/torch/modules/cnn.py:44-49 — `x.float() * scale`, then
// `F.relu(conv(x))` for every conv layer; the first layer of every Atari config
// is Conv2d(4 stacked frames -> 32 filters, kernel 8, stride 4) on 84x84 frames
// (rltime/configs/models/cnn_*.json).  Round 1/2 ran it as three passes: u8 NCHW
// -> f32 NHWC conversion (csrc/convert.hip, 5 B per pixel of HBM traffic), a
// MIOpen implicit-GEMM convolution that re-reads those 4 B pixels (measured
// 3.2 ms per 20 480-frame call, ~25 % of the f32 MFMA rate — the K = 256, N = 32
// shape is far from what its tiles are tuned for) and an in-place bias + ReLU
// pass.  This kernel does the layer in ONE pass from the uint8 planes:
//
//   HBM traffic   1 B per input pixel + 4 B per output element (28 KB in, 51 KB
//                 out per frame) — 3.3 GB for a 41 472-frame block, 0.6 ms at the
//                 achievable HBM rate, so the kernel is bound by the f32 MFMA pipe:
//                 2 * 256 * 32 flop per output position = 6.55 MFLOP per frame,
//                 1.7 ms per 41 472 frames at the 157 TFLOP/s f32 MFMA peak.
//   work split    persistent workgroups (4 waves, 2 per CU), each staging FPI whole
//                 frames (4 planes, raw uint8) in LDS with 16 B per lane and then
//                 walking their 16-position output tiles, one tile per wave at a time.
//   contraction   v_mfma_f32_16x16x4_f32 with the FILTERS as rows and the output
//                 POSITIONS as columns: one instruction sums the 4 input planes
//                 (k = plane) for one (kh, kw) tap; 64 taps x 2 filter halves = 128
//                 MFMAs per tile on two independent accumulator chains (dependent
//                 issue distance 64 cycles > the 40-cycle MFMA latency).
//   operands      the 2 x 64 weight operands of a lane are loop-invariant and live
//                 in registers for the whole kernel (packed once per launch by
//                 k_conv1_pack_w into the lane order, times `scale`); the position
//                 operand is float(u8), read as 8 consecutive bytes per (plane, kh)
//                 row from LDS.  x*(scale*w) instead of (x*scale)*w: one rounding of
//                 the reference's product moved, 1e-7 relative.
//   LDS layout    plane pitch padded to 16 (mod 64) dwords, so the four planes a
//                 ds_read touches land on disjoint bank groups.
//   epilogue      accumulator rows are 4 consecutive filters: + bias, ReLU, two
//                 16 B stores per lane; a wave writes 2 KiB of contiguous NHWC rows.
//
// The layer's input needs no gradient; its weight gradient (k_conv1_u8_wrw, further
// down in this file) reads the same uint8 frames.  Autograd wiring:
// rltime_amd/models/torch/fused.py:_ConvU8BiasReLU.
#include "common.hpp"
#include "split3.hpp"
#include <stdlib.h>
#include <unordered_map>

namespace mirl {

typedef float cv_f4 __attribute__((ext_vector_type(4)));

constexpr int C1_PLANES = 4, C1_K = 8, C1_S = 4, C1_F = 32, C1_TAPS = C1_K * C1_K;
constexpr int C1_WPK = 2 * C1_TAPS * 64;      // packed weight image: [half][tap][lane]
constexpr int C1_LD = 7;                      // staging loads in flight per lane (one 84x84 frame = 6.9 x 256 vectors)

// wpk[(m*64 + tap)*64 + lane] = w[f][c][kh][kw] with the MFMA A-operand lane map
// (row i = lane & 15, k = lane >> 4 = input plane) and the row -> filter
// permutation f = (i>>2)*8 + m*4 + (i&3), which makes the 4 accumulator rows a
// lane owns 4 CONSECUTIVE filters (and the two halves 8 consecutive ones).
__global__ void __launch_bounds__(256)
k_conv1_pack_w(const float* __restrict__ w, int64_t so, int64_t sc, int64_t sh, int64_t sw, float scale, float* __restrict__ wpk) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= C1_WPK) return;
  const int lane = t & 63, tap = (t >> 6) & 63, m = t >> 12;
  const int i = lane & 15, c = lane >> 4, kh = tap >> 3, kw = tap & 7;
  const int f = (i >> 2) * 8 + m * 4 + (i & 3);
  wpk[t] = w[f * so + c * sc + kh * sh + kw * sw] * scale;   // x*scale*w summed as x*(scale*w): one rounding moved
}

__device__ __forceinline__ float c1_byte(uint32_t v, int b) { return (float)((v >> (8 * b)) & 0xffu); }

// x: uint8 [N][4][H][W]; y: float [N][OH][OW][32] (NHWC memory of the logical
// (N, 32, OH, OW) tensor).  Work unit u = (frame group, part): a group is FPI
// consecutive frames, `split` parts share a group's tiles (split > 1 only for
// small N, to spread few frames over the chip).  NTS: non-temporal output stores.
// DBG: timing experiments only (tools/conv_in_probe.py): bit 0 no u8->f32
// conversion, bit 1 no output stores, bit 2 no refill after the first unit.
//
// VALU work next to the MFMA chain is NOT free here (both are issued through the
// SIMD's one VALU port; measured: the 64 conversions + 64 scale multiplies per tile
// cost 11 % of the kernel), so the per-tile scalar work is kept minimal: the scale
// is folded into the packed weights, the position -> (row, column) split is one
// multiply-high, the frame-of-tile split is a compare, and the LDS words of the
// NEXT tile are requested before the current tile's chain so their latency hides
// under it.
#define C1_TILE_ADDR(tt_, f_, p_, base_)                                                        \
  {                                                                                             \
    f_ = (FPI > 1 && tt_ >= tiles) ? 1 : 0;                                                     \
    p_ = (tt_ - f_ * tiles) * 16 + j;                                                           \
    const int pc_ = p_ < OHW ? p_ : OHW - 1;                                                    \
    const int oh_ = OW == 1 ? pc_ : (int)__umulhi((unsigned)pc_, ow_magic);                     \
    base_ = c1_lds + (f_ * C1_PLANES + kq) * pitch + (oh_ * C1_S) * W + (pc_ - oh_ * OW) * C1_S; \
  }
#define C1_TILE_READ(base_, px_)                                                                \
  _Pragma("unroll") for (int kh = 0; kh < C1_K; ++kh) {                                         \
    px_[2 * kh] = *reinterpret_cast<const uint32_t*>(base_ + kh * W);                           \
    px_[2 * kh + 1] = *reinterpret_cast<const uint32_t*>(base_ + kh * W + 4);                   \
  }
// one tile: 128 MFMAs on two accumulator chains, then + bias, ReLU, two 16 B stores
#define C1_TILE_COMPUTE(px_, f_, p_)                                                            \
  {                                                                                             \
    cv_f4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};                                 \
    float vv_[C1_TAPS];                                                                         \
    /* tap s = kh*8 + kw: word kh*2 + (kw>>2), byte kw&3 */                                     \
    _Pragma("unroll") for (int s = 0; s < C1_TAPS; ++s)                                         \
      vv_[s] = (DBG & 1) ? __uint_as_float(px_[s >> 2] & 0x3fffffffu) : c1_byte(px_[s >> 2], s & 3); \
    if (BULK) __builtin_amdgcn_sched_barrier(0);   /* all conversions before the chain */       \
    _Pragma("unroll") for (int s = 0; s < C1_TAPS; ++s) {                                       \
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wr0[s], vv_[s], a0, 0, 0, 0);                   \
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wr1[s], vv_[s], a1, 0, 0, 0);                   \
    }                                                                                           \
    if ((DBG & 2) ? (a0.x == 12345.678f) : (p_ < OHW)) {                                        \
      cv_f4 o0 = a0 + b0, o1 = a1 + b1;                                                         \
      o0.x = o0.x > 0.f ? o0.x : 0.f; o0.y = o0.y > 0.f ? o0.y : 0.f; o0.z = o0.z > 0.f ? o0.z : 0.f; o0.w = o0.w > 0.f ? o0.w : 0.f; \
      o1.x = o1.x > 0.f ? o1.x : 0.f; o1.y = o1.y > 0.f ? o1.y : 0.f; o1.z = o1.z > 0.f ? o1.z : 0.f; o1.w = o1.w > 0.f ? o1.w : 0.f; \
      cv_f4* dst = reinterpret_cast<cv_f4*>(y + ((n0 + f_) * (int64_t)OHW + p_) * C1_F + kq * 8); \
      if (NTS) { __builtin_nontemporal_store(o0, dst); __builtin_nontemporal_store(o1, dst + 1); } \
      else { dst[0] = o0; dst[1] = o1; }                                                        \
    }                                                                                           \
  }

template <int FPI, int NTS, int DBG, int BULK = 0>
__global__ void __launch_bounds__(256, 2)
k_conv1_u8_fwd(int N, int H, int W, int OH, int OW, unsigned ow_magic, int pitch, int split, const uint8_t* __restrict__ x,
               const float* __restrict__ wpk, const float* __restrict__ bias, float* __restrict__ y) {
  extern __shared__ __align__(16) uint8_t c1_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kq = lane >> 4;
  float wr0[C1_TAPS], wr1[C1_TAPS];
#pragma unroll
  for (int s = 0; s < C1_TAPS; ++s) { wr0[s] = wpk[s * 64 + lane]; wr1[s] = wpk[(C1_TAPS + s) * 64 + lane]; }
  const cv_f4 b0 = *reinterpret_cast<const cv_f4*>(bias + kq * 8), b1 = *reinterpret_cast<const cv_f4*>(bias + kq * 8 + 4);
  const int HW = H * W, OHW = OH * OW, tiles = (OHW + 15) >> 4, hw16 = HW >> 4;
  const int groups = (N + FPI - 1) / FPI, units = groups * split, step = 4 * split;
  bool first = true;
  for (int u = blockIdx.x; u < units; u += gridDim.x) {
    const int group = split == 1 ? u : u / split, part = u - group * split;
    const int n0 = group * FPI;
    const int frames = N - n0 < FPI ? N - n0 : FPI;
    if (!first && !(DBG & 4)) __syncthreads();      // every wave is done reading the previous frames
    if (first || !(DBG & 4)) {
      // the unit's frames x 4 planes are one contiguous run of 16 B vectors in HBM; only
      // the LDS side has the padded plane pitch.  C1_LD loads in flight per lane.
      const int vecs = frames * C1_PLANES * hw16;
      const uint4* s4 = reinterpret_cast<const uint4*>(x + (int64_t)n0 * (C1_PLANES * HW));
      for (int o0 = tid; o0 < vecs; o0 += 256 * C1_LD) {
        uint4 v[C1_LD];
#pragma unroll
        for (int k = 0; k < C1_LD; ++k) { const int o = o0 + k * 256; v[k] = s4[o < vecs ? o : vecs - 1]; }
#pragma unroll
        for (int k = 0; k < C1_LD; ++k) {
          const int o = o0 + k * 256;
          if (o < vecs) { const int pl = o / hw16; *reinterpret_cast<uint4*>(c1_lds + pl * pitch + (o - pl * hw16) * 16) = v[k]; }
        }
      }
      __syncthreads();
    }
    first = false;
    // this wave's tiles: tt = part*4 + wave, + step, ... < frames * tiles; two tiles per
    // trip with ping-pong pixel registers, the next tile's LDS reads issued ahead
    const int tend = frames * tiles;
    int tt = part * 4 + wave;
    if (tt >= tend) continue;
    uint32_t pxa[2 * C1_K], pxb[2 * C1_K];
    int fa, pa, fb = 0, pb = 0;
    const uint8_t* base;
    C1_TILE_ADDR(tt, fa, pa, base);
    C1_TILE_READ(base, pxa);
    for (;;) {
      const bool more_b = tt + step < tend;
      if (more_b) { C1_TILE_ADDR(tt + step, fb, pb, base); C1_TILE_READ(base, pxb); }
      __builtin_amdgcn_sched_barrier(0);            // LDS words of the next tile in flight before this chain
      C1_TILE_COMPUTE(pxa, fa, pa);
      if (!more_b) break;
      tt += 2 * step;
      const bool more_a = tt < tend;
      if (more_a) { C1_TILE_ADDR(tt, fa, pa, base); C1_TILE_READ(base, pxa); }
      __builtin_amdgcn_sched_barrier(0);
      C1_TILE_COMPUTE(pxb, fb, pb);
      if (!more_a) break;
    }
  }
}
#undef C1_TILE_ADDR
#undef C1_TILE_READ
#undef C1_TILE_COMPUTE

// ---------------------------------------------------------------------------
// The same forward on the bf16 matrix pipe, still an f32 result.  A uint8 pixel is EXACT in bf16 (8 significand
// bits), so only the weights need the three-way split of csrc/gemm3.hip (w * scale = hi + mid + lo exactly) and
// the product x * w is the sum of THREE bf16 products, each exact in the MFMA's f32 accumulator (8 x 8 bits).
// v_mfma_f32_16x16x32_bf16 with filters as rows and positions as columns, as above, takes k = 32 = (plane = lane >> 4,
// kw = 0..7): the 8 bytes a lane reads per (plane, kh) row ARE its 8 k-elements, so one instruction covers one
// kh row of the 8 x 8 patch over all four planes — 8 kh x 2 filter halves x 3 weight parts = 48 MFMAs of 16 cycles
// per tile against 128 of 32 cycles on the f32 pipe.  hi and mid parts of a lane's weights live in registers
// (2 x 16 x 4 VGPRs), the lo parts in LDS (16 KB, one ds_read_b128 per MFMA); the three parts accumulate on
// separate chains and are added smallest first.  Byte -> bf16 is v_cvt_f32_ubyteN + one v_perm_b32 per pair (the
// upper half of the f32 is the exact bf16).  Staging, tile walk and epilogue are the f32 kernel's.
typedef __bf16 cv_bf8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned c1_pk_bf(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

// wpk3[((part*2 + m)*8 + kh)*64 + lane] = 8 bf16 (kw = 0..7) of part `part` of w[f][c][kh][kw] * scale with the
// A-operand lane map row i = lane & 15 -> filter f = (i>>2)*8 + m*4 + (i&3) (as k_conv1_pack_w), k group lane >> 4 = c
__global__ void __launch_bounds__(256)
k_conv1_pack_w3(const float* __restrict__ w, int64_t so, int64_t sc, int64_t sh, int64_t sw, float scale, uint4* __restrict__ wpk3) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= 2 * C1_K * 64) return;
  const int lane = t & 63, kh = (t >> 6) & 7, m = t >> 9;
  const int i = lane & 15, c = lane >> 4;
  const int f = (i >> 2) * 8 + m * 4 + (i & 3);
  unsigned ph[4], pm[4], pl[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float x0 = w[f * so + c * sc + kh * sh + (2 * e) * sw] * scale, x1 = w[f * so + c * sc + kh * sh + (2 * e + 1) * sw] * scale;
    ph[e] = c1_pk_bf(x0, x1);
    const float r0 = x0 - __uint_as_float(ph[e] << 16), r1 = x1 - __uint_as_float(ph[e] & 0xffff0000u);
    pm[e] = c1_pk_bf(r0, r1);
    const float s0 = r0 - __uint_as_float(pm[e] << 16), s1 = r1 - __uint_as_float(pm[e] & 0xffff0000u);
    pl[e] = c1_pk_bf(s0, s1);
  }
  const int o = (m * 8 + kh) * 64 + lane;
  wpk3[o] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
  wpk3[16 * 64 + o] = make_uint4(pm[0], pm[1], pm[2], pm[3]);
  wpk3[32 * 64 + o] = make_uint4(pl[0], pl[1], pl[2], pl[3]);
}

__device__ __forceinline__ cv_bf8 c1_as_bf8(uint4 v) { return __builtin_bit_cast(cv_bf8, v); }

// bytes (2e, 2e+1) of v -> two bf16 in one dword
__device__ __forceinline__ unsigned c1_bytes_bf(uint32_t v, int e) {
  const float a = (float)((v >> (16 * e)) & 0xffu), b = (float)((v >> (16 * e + 8)) & 0xffu);
  return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}

#define C1_TILE_ADDR(tt_, f_, p_, base_)                                                        \
  {                                                                                             \
    f_ = (FPI > 1 && tt_ >= tiles) ? 1 : 0;                                                     \
    p_ = (tt_ - f_ * tiles) * 16 + j;                                                           \
    const int pc_ = p_ < OHW ? p_ : OHW - 1;                                                    \
    const int oh_ = OW == 1 ? pc_ : (int)__umulhi((unsigned)pc_, ow_magic);                     \
    base_ = c1_lds + (f_ * C1_PLANES + kq) * pitch + (oh_ * C1_S) * W + (pc_ - oh_ * OW) * C1_S; \
  }
#define C1_TILE_READ(base_, px_)                                                                \
  _Pragma("unroll") for (int kh = 0; kh < C1_K; ++kh) {                                         \
    px_[2 * kh] = *reinterpret_cast<const uint32_t*>(base_ + kh * W);                           \
    px_[2 * kh + 1] = *reinterpret_cast<const uint32_t*>(base_ + kh * W + 4);                   \
  }
#define C1B_TILE_COMPUTE(px_, f_, p_)                                                           \
  {                                                                                             \
    cv_f4 h0 = {0.f, 0.f, 0.f, 0.f}, m0 = h0, l0 = h0, h1 = h0, m1 = h0, l1 = h0;               \
    _Pragma("unroll") for (int kh = 0; kh < C1_K; ++kh) {                                       \
      const cv_bf8 b_ = c1_as_bf8(make_uint4(c1_bytes_bf(px_[2 * kh], 0), c1_bytes_bf(px_[2 * kh], 1),       \
                                             c1_bytes_bf(px_[2 * kh + 1], 0), c1_bytes_bf(px_[2 * kh + 1], 1))); \
      l0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c1_as_bf8(wlo[kh * 64 + lane]), b_, l0, 0, 0, 0);          \
      l1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c1_as_bf8(wlo[(C1_K + kh) * 64 + lane]), b_, l1, 0, 0, 0); \
      m0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm0[kh], b_, m0, 0, 0, 0);                   \
      m1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm1[kh], b_, m1, 0, 0, 0);                   \
      h0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh0[kh], b_, h0, 0, 0, 0);                   \
      h1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh1[kh], b_, h1, 0, 0, 0);                   \
    }                                                                                           \
    if (p_ < OHW) {                                                                             \
      cv_f4 o0 = ((l0 + m0) + h0) + b0, o1 = ((l1 + m1) + h1) + b1;                             \
      o0.x = o0.x > 0.f ? o0.x : 0.f; o0.y = o0.y > 0.f ? o0.y : 0.f; o0.z = o0.z > 0.f ? o0.z : 0.f; o0.w = o0.w > 0.f ? o0.w : 0.f; \
      o1.x = o1.x > 0.f ? o1.x : 0.f; o1.y = o1.y > 0.f ? o1.y : 0.f; o1.z = o1.z > 0.f ? o1.z : 0.f; o1.w = o1.w > 0.f ? o1.w : 0.f; \
      cv_f4* dst = reinterpret_cast<cv_f4*>(y + ((n0 + f_) * (int64_t)OHW + p_) * C1_F + kq * 8); \
      if (NTS) { __builtin_nontemporal_store(o0, dst); __builtin_nontemporal_store(o1, dst + 1); } \
      else { dst[0] = o0; dst[1] = o1; }                                                        \
    }                                                                                           \
  }

template <int FPI, int NTS>
__global__ void __launch_bounds__(256, 2)
k_conv1_u8_fwd_bf(int N, int H, int W, int OH, int OW, unsigned ow_magic, int pitch, int split, const uint8_t* __restrict__ x,
                  const uint4* __restrict__ wpk3, const float* __restrict__ bias, float* __restrict__ y) {
  extern __shared__ __align__(16) uint8_t c1_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kq = lane >> 4;
  cv_bf8 wh0[C1_K], wh1[C1_K], wm0[C1_K], wm1[C1_K];
#pragma unroll
  for (int kh = 0; kh < C1_K; ++kh) {
    wh0[kh] = c1_as_bf8(wpk3[kh * 64 + lane]);             wh1[kh] = c1_as_bf8(wpk3[(C1_K + kh) * 64 + lane]);
    wm0[kh] = c1_as_bf8(wpk3[(16 + kh) * 64 + lane]);      wm1[kh] = c1_as_bf8(wpk3[(16 + C1_K + kh) * 64 + lane]);
  }
  // lo parts: 16 KB behind the frame area
  uint4* wlo = reinterpret_cast<uint4*>(c1_lds + FPI * C1_PLANES * pitch);
  for (int o = tid; o < 16 * 64; o += 256) wlo[o] = wpk3[32 * 64 + o];
  const cv_f4 b0 = *reinterpret_cast<const cv_f4*>(bias + kq * 8), b1 = *reinterpret_cast<const cv_f4*>(bias + kq * 8 + 4);
  const int HW = H * W, OHW = OH * OW, tiles = (OHW + 15) >> 4, hw16 = HW >> 4;
  const int groups = (N + FPI - 1) / FPI, units = groups * split, step = 4 * split;
  bool first = true;
  for (int u = blockIdx.x; u < units; u += gridDim.x) {
    const int group = split == 1 ? u : u / split, part = u - group * split;
    const int n0 = group * FPI;
    const int frames = N - n0 < FPI ? N - n0 : FPI;
    if (!first) __syncthreads();      // every wave is done reading the previous frames
    first = false;
    {
      const int vecs = frames * C1_PLANES * hw16;
      const uint4* s4 = reinterpret_cast<const uint4*>(x + (int64_t)n0 * (C1_PLANES * HW));
      for (int o0 = tid; o0 < vecs; o0 += 256 * C1_LD) {
        uint4 v[C1_LD];
#pragma unroll
        for (int k = 0; k < C1_LD; ++k) { const int o = o0 + k * 256; v[k] = s4[o < vecs ? o : vecs - 1]; }
#pragma unroll
        for (int k = 0; k < C1_LD; ++k) {
          const int o = o0 + k * 256;
          if (o < vecs) { const int pl = o / hw16; *reinterpret_cast<uint4*>(c1_lds + pl * pitch + (o - pl * hw16) * 16) = v[k]; }
        }
      }
      __syncthreads();
    }
    const int tend = frames * tiles;
    int tt = part * 4 + wave;
    if (tt >= tend) continue;
    uint32_t pxa[2 * C1_K], pxb[2 * C1_K];
    int fa, pa, fb = 0, pb = 0;
    const uint8_t* base;
    C1_TILE_ADDR(tt, fa, pa, base);
    C1_TILE_READ(base, pxa);
    for (;;) {
      const bool more_b = tt + step < tend;
      if (more_b) { C1_TILE_ADDR(tt + step, fb, pb, base); C1_TILE_READ(base, pxb); }
      __builtin_amdgcn_sched_barrier(0);            // LDS words of the next tile in flight before this chain
      C1B_TILE_COMPUTE(pxa, fa, pa);
      if (!more_b) break;
      tt += 2 * step;
      const bool more_a = tt < tend;
      if (more_a) { C1_TILE_ADDR(tt, fa, pa, base); C1_TILE_READ(base, pxa); }
      __builtin_amdgcn_sched_barrier(0);
      C1B_TILE_COMPUTE(pxb, fb, pb);
      if (!more_a) break;
    }
  }
}
#undef C1_TILE_ADDR
#undef C1_TILE_READ
#undef C1B_TILE_COMPUTE

// ---------------------------------------------------------------------------
// Weight gradient of the same layer from the same uint8 frames:
//   dW[f][c][kh][kw] = scale * sum over (n, oh, ow) of g[n][oh][ow][f] * float(x[n][c][4 oh + kh][4 ow + kw])
// (what autograd derives for cnn.py:44-49; the layer's input needs no gradient).
// GEMM view: D[32 filters][256 taps] = G^T [32][P] x Patch [P][256] with the
// reduction over the P = N * OH * OW output positions.  One v_mfma_f32_16x16x4_f32
// takes 4 positions (k), 16 filters (rows) and 16 taps (columns); a wave keeps the
// WHOLE 32 x 256 result in registers (2 x 16 accumulator tiles = 128 VGPRs, all
// independent: no dependent-issue stalls) and streams positions through it:
//   A operand  g[pos][2j], g[pos][2j+1] — one 8 B load per lane per k-step, a wave
//              load covers 4 consecutive 128 B rows of g (row i <-> filter 2i + half);
//              the next block's loads are issued before the current block's chain
//   B operand  one byte per (lane, tap group) from the frame staged in LDS as in the
//              forward: lane (tap j, position q) reads x[c][4 oh_q + kh][4 ow_q + kw]
//              for the 16 groups t -> c = t>>2, kh = 2 (t&3) + (j>>3), kw = j&7, so that
//              group*16 + j IS the flattened (c, kh, kw) index; with the frame shape a
//              template constant the 16 offsets fold into the ds_read immediates
// Work split as in the forward (persistent workgroups, FPI frames per LDS fill, the 4
// waves interleave blocks of 4 k-steps).  Partials: the 4 waves of a workgroup add
// up in LDS in wave order, the workgroup writes one 32 KB slab, k_conv1_wrw_reduce
// sums the slabs in index order — fixed partition, fixed order, no float atomics.
// Algorithmic bytes: 28 224 B of pixels + 51 200 B of g per frame (as the forward);
// 6.55 MFLOP per frame on the f32 MFMA pipe.
constexpr int C1_WU = 4;                      // k-steps (4 positions each) per block
constexpr int C1_DW = C1_F * C1_PLANES * C1_TAPS;   // 8192 weight-gradient elements

#define C1_WRW_LOAD(bb_, gv_, po_, f_)                                                          \
  {                                                                                             \
    f_ = (FPI > 1 && bb_ >= blocks) ? 1 : 0;                                                    \
    const int ks0_ = (bb_ - f_ * blocks) * C1_WU;                                               \
    const float* gf_ = g + (int64_t)(n0 + f_) * OHW * C1_F + 2 * j;                            \
    _Pragma("unroll") for (int w = 0; w < C1_WU; ++w) {                                         \
      const int p_ = (ks0_ + w) * 4 + q;                                                        \
      const bool ok_ = p_ < OHW;                                                                \
      const int pc_ = ok_ ? p_ : OHW - 1;                                                       \
      float2 t_ = *reinterpret_cast<const float2*>(gf_ + pc_ * C1_F);                          \
      if (MASK) {     /* g = dy where the forward output is positive (the layer's ReLU), its column sums = the bias gradient */ \
        const float2 y_ = *reinterpret_cast<const float2*>(yv + (gf_ - g) + pc_ * C1_F);       \
        t_.x = y_.x > 0.f ? t_.x : 0.f; t_.y = y_.y > 0.f ? t_.y : 0.f;                         \
      }                                                                                         \
      gv_[w].x = ok_ ? t_.x : 0.f; gv_[w].y = ok_ ? t_.y : 0.f;                                 \
      if (MASK) { dbx += gv_[w].x; dby += gv_[w].y; }                                           \
      const int oh_ = OW == 1 ? pc_ : (int)__umulhi((unsigned)pc_, ow_magic);                   \
      po_[w] = (oh_ * C1_S) * Wd + (pc_ - oh_ * OW) * C1_S;                                     \
    }                                                                                           \
  }
#define C1_WRW_COMPUTE(gv_, po_, f_)                                                            \
  {                                                                                             \
    const uint8_t* fl_ = c1_lds + f_ * C1_PLANES * Pd + tap_off;                                \
    _Pragma("unroll") for (int w = 0; w < C1_WU; ++w) {                                         \
      const uint8_t* pb_ = fl_ + po_[w];                                                        \
      float vv_[16];                                                                            \
      _Pragma("unroll") for (int t = 0; t < 16; ++t) vv_[t] = (float)pb_[(t >> 2) * Pd + (t & 3) * 2 * Wd]; \
      if (BULK) __builtin_amdgcn_sched_barrier(0);   /* the k-step's 16 conversions before its 32 MFMAs */ \
      _Pragma("unroll") for (int t = 0; t < 16; ++t) {                                          \
        acc0[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(gv_[w].x, vv_[t], acc0[t], 0, 0, 0);     \
        acc1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(gv_[w].y, vv_[t], acc1[t], 0, 0, 0);     \
      }                                                                                         \
    }                                                                                           \
  }

// WC / PC: frame width and LDS plane pitch as compile-time constants (0 = runtime)
// MASK: g is the gradient w.r.t. the layer's OUTPUT (after its ReLU) and yv the forward output: the ReLU mask is applied
// while the operand is loaded and the bias gradient (column sums of the masked gradient) comes out of the same pass as
// 32 more floats per slab — the separate mask + bias-gradient pass over the (frames, 20, 20, 32) block (three 2.2 GB
// streams at config D) is not needed for this layer, whose input takes no gradient.
template <int FPI, int WC, int PC, int BULK = 0, bool MASK = false>
__global__ void __launch_bounds__(256, 2)
k_conv1_u8_wrw(int N, int H, int W, int OH, int OW, unsigned ow_magic, int pitch, const uint8_t* __restrict__ x,
               const float* __restrict__ g, float* __restrict__ partial, const float* __restrict__ yv = nullptr, int slab = C1_DW) {
  float dbx = 0.f, dby = 0.f;
  extern __shared__ __align__(16) uint8_t c1_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, q = lane >> 4;
  const int Wd = WC ? WC : W, Pd = PC ? PC : pitch;
  const int HW = H * Wd, OHW = OH * OW, hw16 = HW >> 4;
  const int ksteps = (OHW + 3) >> 2, blocks = (ksteps + C1_WU - 1) / C1_WU;
  const int tap_off = (j >> 3) * Wd + (j & 7);
  cv_f4 acc0[16], acc1[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) { acc0[t] = cv_f4{0.f, 0.f, 0.f, 0.f}; acc1[t] = cv_f4{0.f, 0.f, 0.f, 0.f}; }
  const int units = (N + FPI - 1) / FPI;
  bool first = true;
  for (int u = blockIdx.x; u < units; u += gridDim.x) {
    const int n0 = u * FPI;
    const int frames = N - n0 < FPI ? N - n0 : FPI;
    if (!first) __syncthreads();
    first = false;
    {
      const int vecs = frames * C1_PLANES * hw16;
      const uint4* s4 = reinterpret_cast<const uint4*>(x + (int64_t)n0 * (C1_PLANES * HW));
      for (int o0 = tid; o0 < vecs; o0 += 256 * C1_LD) {
        uint4 v[C1_LD];
#pragma unroll
        for (int k = 0; k < C1_LD; ++k) { const int o = o0 + k * 256; v[k] = s4[o < vecs ? o : vecs - 1]; }
#pragma unroll
        for (int k = 0; k < C1_LD; ++k) {
          const int o = o0 + k * 256;
          if (o < vecs) { const int pl = o / hw16; *reinterpret_cast<uint4*>(c1_lds + pl * Pd + (o - pl * hw16) * 16) = v[k]; }
        }
      }
    }
    __syncthreads();
    const int bend = frames * blocks;
    int bb = wave;
    if (bb >= bend) continue;
    float2 ga[C1_WU], gb[C1_WU];
    int pa[C1_WU], pb[C1_WU], fa, fb = 0;
    C1_WRW_LOAD(bb, ga, pa, fa);
    for (;;) {
      const bool more_b = bb + 4 < bend;
      if (more_b) C1_WRW_LOAD(bb + 4, gb, pb, fb);
      __builtin_amdgcn_sched_barrier(0);            // next block's g loads in flight before this chain
      C1_WRW_COMPUTE(ga, pa, fa);
      if (!more_b) break;
      bb += 8;
      const bool more_a = bb < bend;
      if (more_a) C1_WRW_LOAD(bb, ga, pa, fa);
      __builtin_amdgcn_sched_barrier(0);
      C1_WRW_COMPUTE(gb, pb, fb);
      if (!more_a) break;
    }
  }
  // workgroup partial: waves add up in LDS in wave order.  Accumulator tile (half m, tap
  // group t): row (lane>>4)*4 + r <-> filter 2*row + m, column lane&15 <-> tap t*16 + column
  __syncthreads();
  float* red = reinterpret_cast<float*>(c1_lds);
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i0 = (2 * (q * 4 + r)) * (C1_PLANES * C1_TAPS) + t * 16 + j, i1 = i0 + C1_PLANES * C1_TAPS;
          red[i0] = (w ? red[i0] : 0.f) + acc0[t][r];
          red[i1] = (w ? red[i1] : 0.f) + acc1[t][r];
        }
      }
    }
    __syncthreads();
  }
  if (MASK) {
    // bias gradient: lane (j, q) summed filters 2j, 2j + 1 over its positions; the four q in a fixed butterfly, the
    // four waves in wave order behind the slab's 8192 weight-gradient sums
    dbx += __shfl_xor(dbx, 16); dbx += __shfl_xor(dbx, 32);
    dby += __shfl_xor(dby, 16); dby += __shfl_xor(dby, 32);
    for (int w = 0; w < 4; ++w) {
      if (wave == w && q == 0) {
        red[C1_DW + 2 * j] = (w ? red[C1_DW + 2 * j] : 0.f) + dbx;
        red[C1_DW + 2 * j + 1] = (w ? red[C1_DW + 2 * j + 1] : 0.f) + dby;
      }
      __syncthreads();
    }
  }
  float* out = partial + (int64_t)blockIdx.x * slab;
  for (int k = tid; k < (MASK ? C1_DW + C1_F : C1_DW); k += 256) out[k] = red[k];
}
#undef C1_WRW_LOAD
#undef C1_WRW_COMPUTE

// ---------------------------------------------------------------------------------------------------------------
// The same weight gradient on the bf16 matrix pipe, still an f32 result (the forward's argument turned around): a uint8
// pixel is EXACT in bf16, so only g needs the three-way split of csrc/gemm3.hip (g = hi + mid + lo exactly) and
// g * x is the sum of THREE bf16 products, each exact in the MFMA's f32 accumulator: v_mfma_f32_16x16x32_bf16, rows =
// 16 filters, columns = 16 taps, K = 32 output positions per instruction — 2.5 PFLOP/s / 3 against the f32 pipe's 157.
//   * positions are walked in OCTETS of 8 consecutive ow of one output row (a row of OW positions = ceil(OW / 8) octets, the
//     tail octet's positions beyond OW carry g = 0): lane quarter kq of a K-step owns octet 4 ks + kq, so a lane's 8
//     A values are g[oh][ow0 .. ow0 + 7][filter] (8 strided loads, split in registers) and its 8 B values for tap
//     (c, kh, kw) are the pixels x[c][4 oh + kh][4 (ow0 + e) + kw], e < 8: a stride-4 run of one frame row;
//   * the frame is staged ONCE per workgroup as bf16 (two bytes per pixel, converted while staging: 24 VALU ops per 16
//     pixels instead of 12 per fragment), so a B fragment is eight 2-byte LDS reads and no conversion; 56.7 KB per
//     84 x 84 x 4 frame: two workgroups per CU;
//   * per K-step a wave issues 96 MFMAs (16 tap groups x 2 filter halves x 3 parts of g) on 32 independent accumulator
//     tiles (the WHOLE 32 x 256 result stays in registers as in the f32 kernel), smallest part first;
//   * partials, the masked form and the bias gradient exactly as above (same slabs, same k_conv1_wrw_reduce).
template <bool MASK>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_conv1_u8_wrw_b3(int N, int H, int W, int OH, int OW, int Pd, const uint8_t* __restrict__ x, const float* __restrict__ g,
                  float* __restrict__ partial, const float* __restrict__ yv, int slab) {
  extern __shared__ __align__(16) uint8_t c1_lds[];
  uint16_t* px = reinterpret_cast<uint16_t*>(c1_lds);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kq = lane >> 4;
  const int HW = H * W, OHW = OH * OW, hw16 = HW >> 4;
  const int no = (OW + 7) >> 3, octets = OH * no, ksteps = (octets + 3) >> 2;
  const int tap_off = (j >> 3) * W + (j & 7);
  float dbs[2] = {0.f, 0.f};
  g3_f32x4 acc0[16], acc1[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) { acc0[t] = g3_f32x4{0.f, 0.f, 0.f, 0.f}; acc1[t] = g3_f32x4{0.f, 0.f, 0.f, 0.f}; }
  // the slack behind each plane is read (tail octets) but never staged: zero it once — any finite value would do, times g = 0
  for (int i = tid; i < C1_PLANES * (Pd - HW); i += 256) { const int pl = i / (Pd - HW); px[pl * Pd + HW + (i - pl * (Pd - HW))] = 0; }
  bool first = true;
  for (int n = blockIdx.x; n < N; n += gridDim.x) {
    if (!first) __syncthreads();                  // every wave is done reading the previous frame
    first = false;
    {
      // the frame's 4 planes are one contiguous run of 16 B vectors in HBM; 16 pixels -> 16 bf16 = two 16 B LDS stores
      const int vecs = C1_PLANES * hw16;
      const uint4* s4 = reinterpret_cast<const uint4*>(x + (int64_t)n * (C1_PLANES * HW));
      for (int o0 = tid; o0 < vecs; o0 += 256 * C1_LD) {
        uint4 v[C1_LD];
#pragma unroll
        for (int k = 0; k < C1_LD; ++k) { const int o = o0 + k * 256; v[k] = s4[o < vecs ? o : vecs - 1]; }
#pragma unroll
        for (int k = 0; k < C1_LD; ++k) {
          const int o = o0 + k * 256;
          if (o < vecs) {
            const int pl = o / hw16;
            const uint32_t wv[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
            uint32_t h[8];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
              // float(byte) has at most 8 significant bits: its top 16 bits ARE the bf16 value
              const uint32_t f0 = __float_as_uint(c1_byte(wv[d], 0)), f1 = __float_as_uint(c1_byte(wv[d], 1));
              const uint32_t f2 = __float_as_uint(c1_byte(wv[d], 2)), f3 = __float_as_uint(c1_byte(wv[d], 3));
              h[2 * d] = (f0 >> 16) | (f1 & 0xffff0000u);
              h[2 * d + 1] = (f2 >> 16) | (f3 & 0xffff0000u);
            }
            uint4* d4 = reinterpret_cast<uint4*>(px + pl * Pd + (o - pl * hw16) * 16);
            d4[0] = make_uint4(h[0], h[1], h[2], h[3]);
            d4[1] = make_uint4(h[4], h[5], h[6], h[7]);
          }
        }
      }
    }
    __syncthreads();
    const float* gf = g + (int64_t)n * OHW * C1_F + j;
    // the NEXT K-step's g (and y) values are requested before this K-step's split and MFMAs: with two waves per SIMD
    // nothing else covers a global load's latency
    float gn[2][8], yn[2][8];
    auto request = [&](int ks) {
      const int o = ks * 4 + kq;
      const int ohr = o / no, ow0 = (o - ohr * no) * 8;
      const int oh = ohr < OH ? ohr : OH - 1;
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int ow = ow0 + e;
          const int64_t off = (int64_t)(oh * OW + (ow < OW ? ow : OW - 1)) * C1_F + 16 * m;
          gn[m][e] = gf[off];
          if (MASK) yn[m][e] = yv[(gf - g) + off];
        }
    };
    if (wave < ksteps) request(wave);
    for (int ks = wave; ks < ksteps; ks += 4) {
      // this lane's octet: 8 consecutive ow of output row oh
      const int o = ks * 4 + kq;
      const int ohr = o / no, ow0 = (o - ohr * no) * 8;
      const bool row_ok = ohr < OH;
      const int oh = row_ok ? ohr : OH - 1;
      float gc[2][8];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float t = gn[m][e];
          if (MASK) t = yn[m][e] > 0.f ? t : 0.f;                 // the layer's ReLU
          gc[m][e] = (row_ok && ow0 + e < OW) ? t : 0.f;
        }
      if (ks + 4 < ksteps) request(ks + 4);
      uint4 ap[2][3];                               // [filter half][part] = 8 bf16 along k
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        float gv[2][4];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          gv[e >> 2][e & 3] = gc[m][e];
          if (MASK) dbs[m] += gc[m][e];
        }
        uint2 h0, m0, l0, h1, m1, l1;
        g3_split4(gv[0], h0, m0, l0);
        g3_split4(gv[1], h1, m1, l1);
        ap[m][0] = make_uint4(h0.x, h0.y, h1.x, h1.y);
        ap[m][1] = make_uint4(m0.x, m0.y, m1.x, m1.y);
        ap[m][2] = make_uint4(l0.x, l0.y, l1.x, l1.y);
      }
      const uint16_t* pb = px + (oh * C1_S) * W + ow0 * C1_S + tap_off;
      // tap group t: plane t >> 2, rows kh = 2 (t & 3) + (j >> 3) (in tap_off), 8 positions 4 pixels apart.  The NEXT
      // group's eight halfwords are requested before this group's six MFMAs (left alone the compiler reads, waits,
      // multiplies, reads ...: the LDS latency of every group exposed)
      uint32_t raw[2][8];
#pragma unroll
      for (int e = 0; e < 8; ++e) raw[0][e] = pb[4 * e];
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        if (t + 1 < 16) {
          const uint16_t* pt = pb + ((t + 1) >> 2) * Pd + ((t + 1) & 3) * 2 * W;
#pragma unroll
          for (int e = 0; e < 8; ++e) raw[(t + 1) & 1][e] = pt[4 * e];
        }
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t* rw = raw[t & 1];
        const g3_bf16x8 bv = __builtin_bit_cast(g3_bf16x8, make_uint4(rw[0] | (rw[1] << 16), rw[2] | (rw[3] << 16),
                                                                     rw[4] | (rw[5] << 16), rw[6] | (rw[7] << 16)));
#pragma unroll
        for (int p = 2; p >= 0; --p) {              // smallest part of g first
          acc0[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(g3_bf16x8, ap[0][p]), bv, acc0[t], 0, 0, 0);
          acc1[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(g3_bf16x8, ap[1][p]), bv, acc1[t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  // workgroup partial: waves add up in LDS in wave order.  Accumulator tile (half m, tap group t): row 4 kq + r <-> filter
  // 16 m + 4 kq + r, column j <-> tap t * 16 + j
  __syncthreads();
  float* red = reinterpret_cast<float*>(c1_lds);
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i0 = (4 * kq + r) * (C1_PLANES * C1_TAPS) + t * 16 + j, i1 = i0 + 16 * C1_PLANES * C1_TAPS;
          red[i0] = (w ? red[i0] : 0.f) + acc0[t][r];
          red[i1] = (w ? red[i1] : 0.f) + acc1[t][r];
        }
      }
    }
    __syncthreads();
  }
  if (MASK) {
    // bias gradient: lane (j, kq) summed filters j and 16 + j over its octets; the four kq in a fixed butterfly, the four
    // waves in wave order behind the slab's 8192 weight-gradient sums
#pragma unroll
    for (int m = 0; m < 2; ++m) { dbs[m] += __shfl_xor(dbs[m], 16); dbs[m] += __shfl_xor(dbs[m], 32); }
    for (int w = 0; w < 4; ++w) {
      if (wave == w && kq == 0) {
        red[C1_DW + j] = (w ? red[C1_DW + j] : 0.f) + dbs[0];
        red[C1_DW + 16 + j] = (w ? red[C1_DW + 16 + j] : 0.f) + dbs[1];
      }
      __syncthreads();
    }
  }
  float* out = partial + (int64_t)blockIdx.x * slab;
  for (int k = tid; k < (MASK ? C1_DW + C1_F : C1_DW); k += 256) out[k] = red[k];
}

// LDS halfwords per plane of the bf16 frame: the frame, the over-read of a row's tail octet (its positions beyond OW carry
// g = 0 but are still read), rounded to 16 bytes
static int c1b_pitch(int HW, int OW) { return (HW + 4 * (8 * ((OW + 7) / 8) - OW) + 8 + 7) / 8 * 8; }

// dw[f][c][kh][kw] (element strides so, sc, sh, sw) = scale * sum of the slabs in slab order
__global__ void __launch_bounds__(256)
k_conv1_wrw_reduce(const float* __restrict__ partial, int parts, float scale, float* __restrict__ dw, int64_t so, int64_t sc,
                   int64_t sh, int64_t sw, int slab = C1_DW, float* __restrict__ db = nullptr) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= C1_DW + (db ? C1_F : 0)) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int p = 0;
  for (; p + 4 <= parts; p += 4) {
    s0 += partial[(int64_t)p * slab + t]; s1 += partial[(int64_t)(p + 1) * slab + t];
    s2 += partial[(int64_t)(p + 2) * slab + t]; s3 += partial[(int64_t)(p + 3) * slab + t];
  }
  for (; p < parts; ++p) s0 += partial[(int64_t)p * slab + t];
  if (t >= C1_DW) { db[t - C1_DW] = (s0 + s1) + (s2 + s3); return; }      // the bias gradient takes no input scale
  const int f = t >> 8, tap = t & 255;
  dw[f * so + (tap >> 6) * sc + ((tap >> 3) & 7) * sh + (tap & 7) * sw] = ((s0 + s1) + (s2 + s3)) * scale;
}

// LDS plane pitch in bytes: >= HW, a multiple of 16 B, and 16 (mod 64) in dwords
static int c1_pitch(int HW) {
  int dw = (HW + 3) / 4;
  dw += ((16 - dw % 64) + 64) % 64;
  return dw * 4;
}

}  // namespace mirl

static int g_conv1_wrw_bf16 = -1;  // -1: MIRL_CONV1_WRW_BF16 (default on); 0 / 1: set by mirl_conv1_wrw_bf16_set (in-process A/B tests)
extern "C" int mirl_conv1_wrw_bf16_set(int32_t mode) {
  g_conv1_wrw_bf16 = mode < 0 ? -1 : (mode ? 1 : 0);
  return MIRL_OK;
}

namespace mirl {
// the weight gradient on the bf16 pipe (k_conv1_u8_wrw_b3) when the bf16 frame fits the LDS, else / when switched off: false
template <bool MASK>
static int c1_wrw_b3(int64_t N, int32_t H, int32_t W, const uint8_t* x, const float* g, const float* y, float* scratch, int slab,
                     unsigned* grid_out, hipStream_t st, bool* ran) {
  static const int env = getenv("MIRL_CONV1_WRW_BF16") ? atoi(getenv("MIRL_CONV1_WRW_BF16")) : 1;
  *ran = false;
  if (!(g_conv1_wrw_bf16 >= 0 ? g_conv1_wrw_bf16 : env)) return MIRL_OK;
  const int OH = (H - C1_K) / C1_S + 1, OW = (W - C1_K) / C1_S + 1, HW = H * W, Pd = c1b_pitch(HW, OW);
  size_t lds = (size_t)C1_PLANES * Pd * 2;
  if (lds > 80 * 1024) return MIRL_OK;                             // two workgroups per CU or the f32-pipe kernel
  if (lds < (size_t)slab * 4) lds = (size_t)slab * 4;              // the workgroup's partial is reduced there
  const unsigned grid = (unsigned)(N < 512 ? N : 512);
  static bool attr[2] = {false, false};
  const void* fn = (const void*)k_conv1_u8_wrw_b3<MASK>;
  if (!attr[MASK]) { MIRL_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024)); attr[MASK] = true; }
  ProfScope ps("k_conv1_u8_wrw_b3", (double)N * (C1_PLANES * HW + (MASK ? 2.0 : 1.0) * OH * OW * C1_F * 4), st,
               (double)N * OH * OW * 2.0 * C1_PLANES * C1_K * C1_K * C1_F);
  hipLaunchKernelGGL((k_conv1_u8_wrw_b3<MASK>), dim3(grid), dim3(256), lds, st, (int)N, H, W, OH, OW, Pd, x, g, scratch, y, slab);
  MIRL_LAUNCH_CHECK();
  *grid_out = grid;
  *ran = true;
  return MIRL_OK;
}
}  // namespace mirl

static int g_conv1_bf16 = -1;      // -1: MIRL_CONV1_BF16 (default on); 0 / 1: set by mirl_conv1_bf16_set (in-process A/B tests)
extern "C" int mirl_conv1_bf16_set(int32_t mode) {
  g_conv1_bf16 = mode < 0 ? -1 : (mode ? 1 : 0);
  return MIRL_OK;
}

extern "C" int mirl_conv1_u8_supported(int32_t C, int32_t H, int32_t W, int32_t F, int32_t K, int32_t S) {
  using namespace mirl;
  if (C != C1_PLANES || F != C1_F || K != C1_K || S != C1_S) return 0;
  if (H < C1_K || W < C1_K || (W % 4) != 0 || ((H * W) % 16) != 0) return 0;
  return C1_PLANES * c1_pitch(H * W) <= 64 * 1024 ? 1 : 0;
}

// flags: bit 0 = plain (cached) output stores instead of non-temporal ones; bit 3 = skip the weight
// packing (wpk was packed by an earlier call with the same weights and scale); bit 2 = byte->float
// conversions interleaved with the MFMA chain instead of hoisted in front of it; bits 8.. =
// frames per LDS fill override (1 or 2), bits 16.. = split override; bits 24-26 =
// timing-experiment variants (see the kernel).
extern "C" int mirl_conv1_u8_fwd_ex(int64_t N, int32_t H, int32_t W, const uint8_t* x, const float* weight, int64_t ws_o,
                                    int64_t ws_c, int64_t ws_h, int64_t ws_w, const float* bias, float scale, float* wpk,
                                    float* y, int32_t flags, void* stream) {
  using namespace mirl;
  if (N <= 0 || N >= (1LL << 30) || !x || !weight || !bias || !wpk || !y) return fail(MIRL_ERR_ARG, "bad conv1_u8_fwd arguments");
  if (!mirl_conv1_u8_supported(C1_PLANES, H, W, C1_F, C1_K, C1_S)) return fail(MIRL_ERR_ARG, "conv1_u8_fwd: unsupported frame shape");
  if (((uintptr_t)x % 16) || ((uintptr_t)y % 16) || ((uintptr_t)bias % 16) || ((uintptr_t)wpk % 16))
    return fail(MIRL_ERR_ARG, "conv1_u8_fwd: pointers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int OH = (H - C1_K) / C1_S + 1, OW = (W - C1_K) / C1_S + 1, HW = H * W, pitch = c1_pitch(HW);
  const int tiles = (OH * OW + 15) / 16;
  static const int bf_env0 = getenv("MIRL_CONV1_BF16") ? atoi(getenv("MIRL_CONV1_BF16")) : 1;
  const int bf_env = g_conv1_bf16 >= 0 ? g_conv1_bf16 : bf_env0;
  const int dbg0 = (flags >> 24) & 7;
  const bool bf = bf_env && !dbg0 && !(flags & 32) && !(flags & 4);      // bit 5: force the f32-MFMA kernel (probe / A-B)
  // which kernel's operand order a scratch block holds is remembered per pointer (host side, one caller thread per the
  // library's contract): a `packed` call whose kernel choice differs from the packing call's is refused instead of
  // reading a mismatched layout
  static std::unordered_map<const void*, int> packed_as;
  if (flags & 8) {
    auto it = packed_as.find((const void*)wpk);
    if (it == packed_as.end() || it->second != (bf ? 2 : 1))
      return fail(MIRL_ERR_STATE, "conv1_u8_fwd: flags bit 3 (weights already packed) but this scratch block was not packed by the same kernel variant");
  } else {
    packed_as[(const void*)wpk] = bf ? 2 : 1;
  }
  if (!(flags & 8)) {                                     // bit 3: wpk already holds these weights packed (acting steps between updates)
    ProfScope ps("k_conv1_pack_w", 2.0 * C1_WPK * 4, st);
    if (bf) hipLaunchKernelGGL(k_conv1_pack_w3, dim3(4), dim3(256), 0, st, weight, ws_o, ws_c, ws_h, ws_w, scale, (uint4*)wpk);
    else hipLaunchKernelGGL(k_conv1_pack_w, dim3((C1_WPK + 255) / 256), dim3(256), 0, st, weight, ws_o, ws_c, ws_h, ws_w, scale, wpk);
    MIRL_LAUNCH_CHECK();
  }
  int fpi = (flags >> 8) & 0xff, split = (flags >> 16) & 0xff;
  if (fpi != 1 && fpi != 2) fpi = (N >= 1024 && 2 * C1_PLANES * pitch <= 64 * 1024) ? 2 : 1;
  if (fpi == 2 && 2 * C1_PLANES * pitch > 64 * 1024) fpi = 1;
  const int max_split = (tiles + 3) / 4;
  if (split <= 0) split = N >= 512 ? 1 : (int)((512 + N - 1) / N);
  if (split > max_split) split = max_split;
  if (fpi == 2) split = 1;
  const int64_t units = (N + fpi - 1) / fpi * split;
  const unsigned grid = (unsigned)(units < 512 ? units : 512);
  const size_t lds = (size_t)fpi * C1_PLANES * pitch + (bf ? 16 * 1024 : 0);
  ProfScope ps("k_conv1_u8_fwd", (double)N * (C1_PLANES * HW + (double)OH * OW * C1_F * 4), st,
               (double)N * OH * OW * 2.0 * C1_PLANES * C1_K * C1_K * C1_F);
  const bool nts = !(flags & 1);
  if (bf) {
    const uint4* w3 = (const uint4*)wpk;
#define C1B_LAUNCH(FPI_, NTS_) \
  hipLaunchKernelGGL((k_conv1_u8_fwd_bf<FPI_, NTS_>), dim3(grid), dim3(256), lds, st, (int)N, H, W, OH, OW, ow_magic_bf, pitch, split, x, w3, bias, y)
    const unsigned ow_magic_bf = OW > 1 ? (unsigned)(((1ULL << 32) + OW - 1) / OW) : 0u;
    static bool attr_set = false;
    if (!attr_set) {                                        // 2 frames + 16 KB of weight parts = 74 KB of dynamic LDS
      MIRL_HIP(hipFuncSetAttribute((const void*)k_conv1_u8_fwd_bf<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
      MIRL_HIP(hipFuncSetAttribute((const void*)k_conv1_u8_fwd_bf<2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
      MIRL_HIP(hipFuncSetAttribute((const void*)k_conv1_u8_fwd_bf<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
      MIRL_HIP(hipFuncSetAttribute((const void*)k_conv1_u8_fwd_bf<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
      attr_set = true;
    }
    if (fpi == 2) { if (nts) C1B_LAUNCH(2, 1); else C1B_LAUNCH(2, 0); }
    else          { if (nts) C1B_LAUNCH(1, 1); else C1B_LAUNCH(1, 0); }
#undef C1B_LAUNCH
    MIRL_LAUNCH_CHECK();
    return MIRL_OK;
  }
  const int dbg = (flags >> 24) & 7;
  const unsigned ow_magic = OW > 1 ? (unsigned)(((1ULL << 32) + OW - 1) / OW) : 0u;   // exact n / OW for n < 2^16
#define C1_LAUNCH(FPI_, NTS_, DBG_) \
  hipLaunchKernelGGL((k_conv1_u8_fwd<FPI_, NTS_, DBG_>), dim3(grid), dim3(256), lds, st, (int)N, H, W, OH, OW, ow_magic, pitch, split, x, wpk, bias, y)
  if (dbg) {                                             // timing experiments, fpi 2 + nt stores only
    if (fpi != 2 || !nts) return fail(MIRL_ERR_ARG, "conv1_u8_fwd: debug variants exist for fpi 2 with nt stores only");
    switch (dbg) {
      case 1: C1_LAUNCH(2, 1, 1); break;  case 2: C1_LAUNCH(2, 1, 2); break;
      case 4: C1_LAUNCH(2, 1, 4); break;  case 7: C1_LAUNCH(2, 1, 7); break;
      default: return fail(MIRL_ERR_ARG, "conv1_u8_fwd: unknown debug variant");
    }
  } else if (fpi == 2 && nts && !(flags & 4)) {
    // all 64 byte->float conversions of a tile in front of its MFMA chain: 2.20 vs 2.41 ms per
    // 41 472 frames (conversions interleaved with the chain delay MFMA issue; 250 VGPRs, still
    // 2 waves per SIMD).  Bit 2 of flags keeps the interleaved variant for the probe.
    hipLaunchKernelGGL((k_conv1_u8_fwd<2, 1, 0, 1>), dim3(grid), dim3(256), lds, st, (int)N, H, W, OH, OW, ow_magic, pitch, split, x, wpk, bias, y);
  } else if (fpi == 2) {
    if (nts) C1_LAUNCH(2, 1, 0); else C1_LAUNCH(2, 0, 0);
  } else {
    if (nts) C1_LAUNCH(1, 1, 0); else C1_LAUNCH(1, 0, 0);
  }
#undef C1_LAUNCH
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}

extern "C" int mirl_conv1_u8_fwd(int64_t N, int32_t H, int32_t W, const uint8_t* x, const float* weight, int64_t ws_o,
                                 int64_t ws_c, int64_t ws_h, int64_t ws_w, const float* bias, float scale, float* wpk,
                                 float* y, void* stream) {
  static const int flags = getenv("MIRL_CONV1_FLAGS") ? atoi(getenv("MIRL_CONV1_FLAGS")) : 0;
  return mirl_conv1_u8_fwd_ex(N, H, W, x, weight, ws_o, ws_c, ws_h, ws_w, bias, scale, wpk, y, flags, stream);
}

// floats of `wpk` scratch mirl_conv1_u8_fwd[_ex] needs (the bf16-pipe kernel packs three 16 KB weight parts: more than the
// f32 kernel's 8192 floats of rounds 2 — callers size the block from this query, not from a constant)
extern "C" int mirl_conv1_u8_wpk_floats(int64_t* out) {
  if (!out) return mirl::fail(MIRL_ERR_ARG, "null out");
  *out = 12288;
  return MIRL_OK;
}

extern "C" int mirl_conv1_u8_wrw_scratch_floats(int64_t* out) {
  if (!out) return mirl::fail(MIRL_ERR_ARG, "null out");
  *out = (int64_t)512 * (mirl::C1_DW + mirl::C1_F);      // per-workgroup slabs: 8192 weight-gradient sums (+ 32 bias-gradient sums, masked form)
  return MIRL_OK;
}

// flags bit 0: conversions interleaved with the MFMAs (the first version) instead of hoisted per k-step
extern "C" int mirl_conv1_u8_wrw_ex(int64_t N, int32_t H, int32_t W, const uint8_t* x, const float* g, float scale,
                                    float* scratch, float* dw, int64_t ws_o, int64_t ws_c, int64_t ws_h, int64_t ws_w,
                                    int32_t flags, void* stream) {
  using namespace mirl;
  if (N <= 0 || N >= (1LL << 30) || !x || !g || !scratch || !dw) return fail(MIRL_ERR_ARG, "bad conv1_u8_wrw arguments");
  if (!mirl_conv1_u8_supported(C1_PLANES, H, W, C1_F, C1_K, C1_S)) return fail(MIRL_ERR_ARG, "conv1_u8_wrw: unsupported frame shape");
  if (((uintptr_t)x % 16) || ((uintptr_t)g % 16) || ((uintptr_t)scratch % 16))
    return fail(MIRL_ERR_ARG, "conv1_u8_wrw: pointers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int OH = (H - C1_K) / C1_S + 1, OW = (W - C1_K) / C1_S + 1, HW = H * W, pitch = c1_pitch(HW);
  const int fpi = (N >= 1024 && 2 * C1_PLANES * pitch <= 64 * 1024) ? 2 : 1;
  const int64_t units = (N + fpi - 1) / fpi;
  const unsigned grid = (unsigned)(units < 512 ? units : 512);
  size_t lds = (size_t)fpi * C1_PLANES * pitch;
  if (lds < (size_t)C1_DW * 4) lds = (size_t)C1_DW * 4;          // the workgroup's 32 KB partial is reduced there
  const unsigned ow_magic = OW > 1 ? (unsigned)(((1ULL << 32) + OW - 1) / OW) : 0u;
  bool b3 = false;
  unsigned b3_grid = 0;
  if (!(flags & 2)) {                                               // flags bit 1: force the f32-pipe kernel
    if (int rc = c1_wrw_b3<false>(N, H, W, x, g, nullptr, scratch, (int)C1_DW, &b3_grid, st, &b3)) return rc;
  }
  if (b3) {
    ProfScope ps("k_conv1_wrw_reduce", (double)b3_grid * C1_DW * 4, st);
    hipLaunchKernelGGL(k_conv1_wrw_reduce, dim3((C1_DW + 255) / 256), dim3(256), 0, st, scratch, (int)b3_grid, scale, dw, ws_o, ws_c, ws_h, ws_w,
                       (int)C1_DW, (float*)nullptr);
    MIRL_LAUNCH_CHECK();
    return MIRL_OK;
  }
  {
    ProfScope ps("k_conv1_u8_wrw", (double)N * (C1_PLANES * HW + (double)OH * OW * C1_F * 4), st,
                 (double)N * OH * OW * 2.0 * C1_PLANES * C1_K * C1_K * C1_F);
#define C1_WLAUNCH(FPI_, WC_, PC_) \
  hipLaunchKernelGGL((k_conv1_u8_wrw<FPI_, WC_, PC_>), dim3(grid), dim3(256), lds, st, (int)N, H, W, OH, OW, ow_magic, pitch, x, g, scratch, (const float*)nullptr, (int)C1_DW)
    const bool atari = W == 84 && pitch == 7232;
    if (fpi == 2 && atari && !(flags & 1))
      hipLaunchKernelGGL((k_conv1_u8_wrw<2, 84, 7232, 1>), dim3(grid), dim3(256), lds, st, (int)N, H, W, OH, OW, ow_magic, pitch, x, g, scratch, (const float*)nullptr, (int)C1_DW);
    else if (fpi == 2) { if (atari) C1_WLAUNCH(2, 84, 7232); else C1_WLAUNCH(2, 0, 0); }
    else               { if (atari) C1_WLAUNCH(1, 84, 7232); else C1_WLAUNCH(1, 0, 0); }
#undef C1_WLAUNCH
    MIRL_LAUNCH_CHECK();
  }
  ProfScope ps("k_conv1_wrw_reduce", (double)grid * C1_DW * 4, st);
  hipLaunchKernelGGL(k_conv1_wrw_reduce, dim3((C1_DW + 255) / 256), dim3(256), 0, st, scratch, (int)grid, scale, dw, ws_o, ws_c, ws_h, ws_w,
                     (int)C1_DW, (float*)nullptr);
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}

// Weight AND bias gradient from the gradient w.r.t. the layer's output: dy masked by the forward output y > 0 while it is
// loaded (the layer's ReLU, cnn.py:47-49), db[f] = the masked gradient's column sums from the same pass.
extern "C" int mirl_conv1_u8_wrw_masked(int64_t N, int32_t H, int32_t W, const uint8_t* x, const float* dy, const float* y, float scale,
                                        float* scratch, float* dw, int64_t ws_o, int64_t ws_c, int64_t ws_h, int64_t ws_w,
                                        float* db, void* stream) {
  using namespace mirl;
  if (N <= 0 || N >= (1LL << 30) || !x || !dy || !y || !scratch || !dw || !db) return fail(MIRL_ERR_ARG, "bad conv1_u8_wrw_masked arguments");
  if (!mirl_conv1_u8_supported(C1_PLANES, H, W, C1_F, C1_K, C1_S)) return fail(MIRL_ERR_ARG, "conv1_u8_wrw: unsupported frame shape");
  if (((uintptr_t)x % 16) || ((uintptr_t)dy % 16) || ((uintptr_t)y % 16) || ((uintptr_t)scratch % 16))
    return fail(MIRL_ERR_ARG, "conv1_u8_wrw: pointers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int OH = (H - C1_K) / C1_S + 1, OW = (W - C1_K) / C1_S + 1, HW = H * W, pitch = c1_pitch(HW);
  const int fpi = (N >= 1024 && 2 * C1_PLANES * pitch <= 64 * 1024) ? 2 : 1;
  const int64_t units = (N + fpi - 1) / fpi;
  const unsigned grid = (unsigned)(units < 512 ? units : 512);
  const int slab = C1_DW + C1_F;
  size_t lds = (size_t)fpi * C1_PLANES * pitch;
  if (lds < (size_t)slab * 4) lds = (size_t)slab * 4;
  const unsigned ow_magic = OW > 1 ? (unsigned)(((1ULL << 32) + OW - 1) / OW) : 0u;
  bool b3 = false;
  unsigned b3_grid = 0;
  if (int rc = c1_wrw_b3<true>(N, H, W, x, dy, y, scratch, slab, &b3_grid, st, &b3)) return rc;
  if (b3) {
    ProfScope ps("k_conv1_wrw_reduce", (double)b3_grid * slab * 4, st);
    hipLaunchKernelGGL(k_conv1_wrw_reduce, dim3((slab + 255) / 256), dim3(256), 0, st, scratch, (int)b3_grid, scale, dw, ws_o, ws_c, ws_h, ws_w, slab, db);
    MIRL_LAUNCH_CHECK();
    return MIRL_OK;
  }
  {
    ProfScope ps("k_conv1_u8_wrw", (double)N * (C1_PLANES * HW + 2.0 * OH * OW * C1_F * 4), st,
                 (double)N * OH * OW * 2.0 * C1_PLANES * C1_K * C1_K * C1_F);
#define C1_MLAUNCH(FPI_, WC_, PC_, BULK_) \
  hipLaunchKernelGGL((k_conv1_u8_wrw<FPI_, WC_, PC_, BULK_, true>), dim3(grid), dim3(256), lds, st, (int)N, H, W, OH, OW, ow_magic, pitch, x, dy, scratch, y, slab)
    const bool atari = W == 84 && pitch == 7232;
    if (fpi == 2 && atari) C1_MLAUNCH(2, 84, 7232, 1);
    else if (fpi == 2) C1_MLAUNCH(2, 0, 0, 0);
    else if (atari) C1_MLAUNCH(1, 84, 7232, 0);
    else C1_MLAUNCH(1, 0, 0, 0);
#undef C1_MLAUNCH
    MIRL_LAUNCH_CHECK();
  }
  ProfScope ps("k_conv1_wrw_reduce", (double)grid * slab * 4, st);
  hipLaunchKernelGGL(k_conv1_wrw_reduce, dim3((slab + 255) / 256), dim3(256), 0, st, scratch, (int)grid, scale, dw, ws_o, ws_c, ws_h, ws_w, slab, db);
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}

extern "C" int mirl_conv1_u8_wrw(int64_t N, int32_t H, int32_t W, const uint8_t* x, const float* g, float scale,
                                 float* scratch, float* dw, int64_t ws_o, int64_t ws_c, int64_t ws_h, int64_t ws_w,
                                 void* stream) {
  static const int flags = getenv("MIRL_CONV1_WRW_FLAGS") ? atoi(getenv("MIRL_CONV1_WRW_FLAGS")) : 0;   // hoisted conversions: 2.60 vs 2.66 ms per 42 496 frames (profiles/r03)
  return mirl_conv1_u8_wrw_ex(N, H, W, x, g, scale, scratch, dw, ws_o, ws_c, ws_h, ws_w, flags, stream);
}
