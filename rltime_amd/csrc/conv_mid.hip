// conv_mid.hip — data gradient of the second conv layer on the f32 MFMA pipe.
//
// Reference: the backward autograd derives for `F.relu(conv(x))`
// (rltime/models/torch/modules/cnn.py:47-49) at the second layer of every Atari
// model (configs/models/cnn_*.json: Conv2d(32 -> 64, kernel 4, stride 2)).  The
// library's kernel for it (igemm_bwd_gtcx35_nhwc_fp32 ... bt128x32x32) was the
// slowest contraction of the step relative to its size: 5.4 ms for 42 496 frames
// = 42 TFLOP/s, 27 % of the f32 MFMA peak (profiles/r02_rocprofv3_kernel_stats_summary.txt).
//
// With stride 2 and kernel 4 every input pixel (ih, iw) = (2u + ph, 2v + pw) receives
// exactly 2 x 2 taps, kh = ph + 2a, kw = pw + 2b, from the output positions (u - a, v - b):
//   dx[2u+ph][2v+pw][c] = sum over a, b in {0,1}, f < 64 of g[u-a][v-b][f] * W[f][c][ph+2a][pw+2b]
// i.e. four independent GEMMs, one per parity class (ph, pw), each with K = 4*64 = 256
// and N = 32 — the very shape of the input layer's forward (conv_in.hip), and the same
// plan: v_mfma_f32_16x16x4_f32 with the 32 channels as rows (2 halves), 16 pixels of
// one class as columns, the 2 x 64 weight operands of a lane resident in registers.
//   * a workgroup is 4 waves = the 4 parity classes, so each wave loads ITS class's
//     8192 weights once (k_conv2_pack_w puts them in lane order) and all four stay
//     balanced; persistent workgroups stream FPI = 2 frames per LDS fill.
//   * the g tile of a frame (OH x OW x 64 floats) is staged in LDS inside a one-position
//     ZERO border (rows shared between the stacked frames), so the out-of-range taps of
//     edge pixels need no masks; position pitch 68 floats keeps the 16 B reads of 16
//     neighbouring positions on disjoint banks.
//   * per 16-pixel tile a lane reads 16 x 16 B (its quarter of the 64 filters for the four
//     taps) and issues 128 MFMAs on two independent chains; no conversion work at all.
//   * epilogue: two 16 B stores per lane (8 consecutive channels of one pixel).
// Traffic: g 20.7 KB in + dx 51.2 KB out per frame; 4 classes x (OH+1)(OW+1) pixels x
// 2*256*32 flop = 6.55 MFLOP per frame (19 % of it on the zero border — the price of
// mask-free edges).
#include "common.hpp"
#include "split3.hpp"
#include <stdlib.h>

namespace mirl {

typedef float cm_f4 __attribute__((ext_vector_type(4)));

constexpr int C2_C = 32, C2_F = 64, C2_K = 4, C2_S = 2;
constexpr int C2_STEPS = 64;                  // MFMA steps per tile and channel half: 4 taps x 16 filters per lane quarter
constexpr int C2_WPK = 4 * 2 * C2_STEPS * 64; // packed weights: [class][half][step][lane] = all 32768 weights
constexpr int C2_PP = 68;                     // LDS floats per position (64 filters + 4 pad)
constexpr int C2_LD = 6;                      // staging loads in flight per lane

// wpk[((cls*2 + m)*64 + s)*64 + lane] = W[f][c][kh][kw] with lane = (i = lane&15, kq = lane>>4):
// f = kq*16 + (s&15), tap a = s>>5, b = (s>>4)&1, kh = ph + 2a, kw = pw + 2b (cls = ph*2 + pw),
// c = (i>>2)*8 + m*4 + (i&3) (the 4 accumulator rows of a lane = 4 consecutive channels).
__global__ void __launch_bounds__(256)
k_conv2_pack_w(const float* __restrict__ w, int64_t so, int64_t sc, int64_t sh, int64_t sw, float* __restrict__ wpk) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= C2_WPK) return;
  const int lane = t & 63, s = (t >> 6) & 63, m = (t >> 12) & 1, cls = t >> 13;
  const int i = lane & 15, kq = lane >> 4;
  const int f = kq * 16 + (s & 15), kh = (cls >> 1) + 2 * (s >> 5), kw = (cls & 1) + 2 * ((s >> 4) & 1);
  const int c = (i >> 2) * 8 + m * 4 + (i & 3);
  wpk[t] = w[f * so + c * sc + kh * sh + kw * sw];
}

// g: float [N][OH][OW][64]; dx: float [N][2 OH + 2][2 OW + 2][32] (both NHWC memory).
template <int FPI>
__global__ void __launch_bounds__(256, 2)
k_conv2_bwd_data(int N, int OH, int OW, unsigned v_magic, const float* __restrict__ g,
                 const float* __restrict__ wpk, float* __restrict__ dx) {
  extern __shared__ __align__(16) float c2_lds[];
  const int tid = threadIdx.x, lane = tid & 63, cls = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kq = lane >> 4;
  const int ph = cls >> 1, pw = cls & 1;
  float wr0[C2_STEPS], wr1[C2_STEPS];
#pragma unroll
  for (int s = 0; s < C2_STEPS; ++s) {
    wr0[s] = wpk[((cls * 2 + 0) * C2_STEPS + s) * 64 + lane];
    wr1[s] = wpk[((cls * 2 + 1) * C2_STEPS + s) * 64 + lane];
  }
  const int PW = OW + 2, rows = FPI * (OH + 1) + 1;
  const int U = OH + 1, V = OW + 1, UV = U * V, IH = 2 * OH + 2, IW = 2 * OW + 2;
  const int pos16 = OH * OW * 16;               // 16 B vectors per frame of g
  // zero everything once: the border stays zero, the fills only write interiors
  for (int o = tid; o < rows * PW * (C2_PP / 4); o += 256) reinterpret_cast<cm_f4*>(c2_lds)[o] = cm_f4{0.f, 0.f, 0.f, 0.f};
  const int units = (N + FPI - 1) / FPI;
  for (int u0 = blockIdx.x; u0 < units; u0 += gridDim.x) {
    const int n0 = u0 * FPI;
    const int frames = N - n0 < FPI ? N - n0 : FPI;
    __syncthreads();                            // zeroing / every wave done with the previous frames
    {
      const int vecs = frames * pos16;
      const cm_f4* s4 = reinterpret_cast<const cm_f4*>(g + (int64_t)n0 * OH * OW * C2_F);
      for (int o0 = tid; o0 < vecs; o0 += 256 * C2_LD) {
        cm_f4 v[C2_LD];
#pragma unroll
        for (int k = 0; k < C2_LD; ++k) { const int o = o0 + k * 256; v[k] = s4[o < vecs ? o : vecs - 1]; }
#pragma unroll
        for (int k = 0; k < C2_LD; ++k) {
          const int o = o0 + k * 256;
          if (o < vecs) {
            const int f = o >= pos16 ? 1 : 0, r = o - f * pos16, pos = r >> 4, sub = r & 15;
            const int oh2 = pos / OW, ow2 = pos - oh2 * OW;
            *reinterpret_cast<cm_f4*>(c2_lds + ((f * (OH + 1) + 1 + oh2) * PW + 1 + ow2) * C2_PP + sub * 4) = v[k];
          }
        }
      }
    }
    __syncthreads();
    const int pend = frames * UV;
    for (int p0 = 0; p0 < pend; p0 += 16) {
      const int p = p0 + j, pc = p < pend ? p : pend - 1;
      const int f = (FPI > 1 && pc >= UV) ? 1 : 0, r = pc - f * UV;
      const int u = V == 1 ? r : (int)__umulhi((unsigned)r, v_magic), v = r - u * V;
      // position (u - a, v - b) of frame f inside the bordered grid, this lane's filter quarter
      const float* base = c2_lds + ((f * (OH + 1) + 1 + u) * PW + 1 + v) * C2_PP + kq * 16;
      cm_f4 gq[16];
#pragma unroll
      for (int ab = 0; ab < 4; ++ab) {
        const float* pb = base - ((ab >> 1) * PW + (ab & 1)) * C2_PP;
#pragma unroll
        for (int e = 0; e < 4; ++e) gq[ab * 4 + e] = *reinterpret_cast<const cm_f4*>(pb + e * 4);
      }
      cm_f4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < C2_STEPS; ++s) {
        const float val = gq[s >> 2][s & 3];    // tap s>>4, filter kq*16 + (s&15)
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wr0[s], val, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wr1[s], val, a1, 0, 0, 0);
      }
      if (p < pend) {
        cm_f4* dst = reinterpret_cast<cm_f4*>(dx + ((((int64_t)(n0 + f) * IH + 2 * u + ph) * IW) + 2 * v + pw) * C2_C + kq * 8);
        dst[0] = a0; dst[1] = a1;
      }
    }
  }
}

static size_t c2_lds_bytes(int fpi, int OH, int OW) { return (size_t)(fpi * (OH + 1) + 1) * (OW + 2) * C2_PP * 4; }

// ---------------------------------------------------------------------------------------------------------------
// The same four parity-class GEMMs on the bf16 matrix pipe with f32 results: csrc/gemm3.hip's exact three-way bf16
// split (six part products, f32 accumulation) on v_mfma_f32_16x16x32_bf16.  2.65 x the f32 pipe's rate per product,
// and the 6.55 MFLOP per frame were what bound the f32-pipe kernel above (0.64 of that pipe's peak).
//   * a workgroup is 8 waves = 4 parity classes x 2 channel halves: wave (cls, m) owns channels 16 m .. 16 m + 15
//     (the MFMA's 16 rows) of class cls and keeps ITS 16 x 256 weights, split once by k_conv2_pack_w3b, in registers
//     for the whole launch: 8 K-steps x 3 parts x 8 bf16 per lane = 96 VGPRs;
//   * the g tile of FPI frames is split ONCE while it is staged, into three bf16 parts laid out for the fragment reads:
//     a 16-byte chunk = 8 filters of one position; chunk (K-step half sb, lane quarter kq) of position pos sits at
//     (kq & 1) * HP + (2 sb + (kq >> 1)) * S + pos * 16, i.e. consecutive positions are consecutive 16-byte units.  A
//     ds_read_b128 is served in four groups of 16 lanes that MIX two lane quarters ({0-3,12-15,20-27}, ...,
//     MI355X_MICROARCH.md LDS table): with HP a multiple of 256 bytes the 16 positions of a group are 16 consecutive
//     units = all 64 banks whichever quarter a lane is in (a [position][64 filters] layout with a padded pitch has 2-way
//     conflicts on 7 of a group's 16 lanes).  Same one-position zero border as above; 93.7 KB for two
//     9 x 9 frames: one workgroup per CU;
//   * per 16-pixel tile a wave reads 8 K-steps x 3 parts x 16 B of g (its filter eighth of the tap's position) and
//     issues 48 MFMAs; K order = (tap, filter), a K-step = 32 filters of one tap;
//   * epilogue: one 16 B store per lane (4 consecutive channels of one pixel).
constexpr int C2B_SLOTS = 4 * 2 * 8;          // (class, channel half, K-step)
constexpr int C2B_PF = 6;                     // 16-byte vectors of the NEXT unit a thread holds while this one is multiplied
constexpr int C2B_WPK_BYTES = C2B_SLOTS * 3 * 64 * 16;

// bytes between the four chunk planes of a half (16 more than a multiple of 256: the 8-byte staging writes of one position's
// chunks then spread over the banks) and per half plane (a multiple of 256: see above)
__host__ __device__ inline int c2b_chunk_stride(int npos) { return ((npos + 15) / 16) * 256 + 16; }
__host__ __device__ inline int c2b_half_plane(int npos) { return (4 * c2b_chunk_stride(npos) + 255) / 256 * 256; }

// wpk3[((slot * 3 + part) * 64 + lane)] = 8 bf16: channel 16 m + (lane & 15), filters 32 (s & 1) + 8 (lane >> 4) .. + 7
// of tap s >> 1 (a = tap >> 1, b = tap & 1: kh = ph + 2 a, kw = pw + 2 b), slot = (cls * 2 + m) * 8 + s
__global__ void __launch_bounds__(256)
k_conv2_pack_w3b(const float* __restrict__ w, int64_t so, int64_t sc, int64_t sh, int64_t sw, uint4* __restrict__ wpk3) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= C2B_SLOTS * 64) return;
  const int lane = t & 63, slot = t >> 6, st = slot & 7, m = (slot >> 3) & 1, cls = slot >> 4;
  const int c = 16 * m + (lane & 15), tap = st >> 1;
  const int kh = (cls >> 1) + 2 * (tap >> 1), kw = (cls & 1) + 2 * (tap & 1);
  const int f0 = 32 * (st & 1) + 8 * (lane >> 4);
  float x[2][4];
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e >> 2][e & 3] = w[(f0 + e) * so + c * sc + kh * sh + kw * sw];
  uint2 h0, m0, l0, h1, m1, l1;
  g3_split4(x[0], h0, m0, l0);
  g3_split4(x[1], h1, m1, l1);
  wpk3[(slot * 3 + 0) * 64 + lane] = make_uint4(h0.x, h0.y, h1.x, h1.y);
  wpk3[(slot * 3 + 1) * 64 + lane] = make_uint4(m0.x, m0.y, m1.x, m1.y);
  wpk3[(slot * 3 + 2) * 64 + lane] = make_uint4(l0.x, l0.y, l1.x, l1.y);
}

template <int FPI>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_conv2_bwd_data_b3(int N, int OH, int OW, unsigned v_magic, const float* __restrict__ g,
                    const uint4* __restrict__ wpk3, float* __restrict__ dx) {
  extern __shared__ __align__(16) char c2b_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cls = wave >> 1, m = wave & 1;
  const int j = lane & 15, kq = lane >> 4;
  const int ph = cls >> 1, pw = cls & 1;
  g3_bf16x8 wr[8][3];
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int p = 0; p < 3; ++p)
      wr[s][p] = __builtin_bit_cast(g3_bf16x8, wpk3[((((cls * 2 + m) * 8 + s) * 3) + p) * 64 + lane]);
  const int PW = OW + 2, rows = FPI * (OH + 1) + 1;
  const int U = OH + 1, V = OW + 1, UV = U * V, IH = 2 * OH + 2, IW = 2 * OW + 2;
  const int pos16 = OH * OW * 16;               // 16 B vectors per frame of g
  const int S = c2b_chunk_stride(rows * PW), HP = c2b_half_plane(rows * PW), plane = 2 * HP;
  // zero all three parts once: the border stays zero, the fills only write interiors
  for (int o = tid; o < 3 * plane / 16; o += 512) reinterpret_cast<uint4*>(c2b_lds)[o] = make_uint4(0u, 0u, 0u, 0u);
  const int units = (N + FPI - 1) / FPI;
  // the next unit's g vectors are requested BEFORE this unit's MFMAs and land in registers while they run (one
  // workgroup per CU: nobody else would cover the load latency); units too large for C2B_PF vectors per thread are
  // loaded in place instead (pf == false)
  constexpr int PF = C2B_PF;
  const bool pf = FPI * pos16 <= 512 * PF;
  cm_f4 nv[PF];
  auto request = [&](int unit) {
    const int nn0 = unit * FPI;
    const int vecs = (N - nn0 < FPI ? N - nn0 : FPI) * pos16;
    const cm_f4* s4 = reinterpret_cast<const cm_f4*>(g + (int64_t)nn0 * OH * OW * C2_F);
#pragma unroll
    for (int k = 0; k < PF; ++k) { const int o = tid + k * 512; nv[k] = s4[o < vecs ? o : vecs - 1]; }
  };
  auto put = [&](const cm_f4& val, int o) {
    const int f = o >= pos16 ? 1 : 0, r = o - f * pos16, pos = r >> 4, sub = r & 15;
    const int oh2 = pos / OW, ow2 = pos - oh2 * OW;
    const float x[4] = {val.x, val.y, val.z, val.w};
    uint2 hh, mm, ll;
    g3_split4(x, hh, mm, ll);
    const int chunk = sub >> 1, kqw = chunk & 3;          // 8 filters: K-step half chunk >> 2, lane quarter kqw
    char* d = c2b_lds + (kqw & 1) * HP + (2 * (chunk >> 2) + (kqw >> 1)) * S + ((f * (OH + 1) + 1 + oh2) * PW + 1 + ow2) * 16 + (sub & 1) * 8;
    *reinterpret_cast<uint2*>(d) = hh;
    *reinterpret_cast<uint2*>(d + plane) = mm;
    *reinterpret_cast<uint2*>(d + 2 * plane) = ll;
  };
  if (pf && (int)blockIdx.x < units) request(blockIdx.x);
  for (int u0 = blockIdx.x; u0 < units; u0 += gridDim.x) {
    const int n0 = u0 * FPI;
    const int frames = N - n0 < FPI ? N - n0 : FPI;
    __syncthreads();                            // zeroing / every wave done with the previous frames
    {
      const int vecs = frames * pos16;
      if (pf) {
#pragma unroll
        for (int k = 0; k < PF; ++k) { const int o = tid + k * 512; if (o < vecs) put(nv[k], o); }
      } else {
        const cm_f4* s4 = reinterpret_cast<const cm_f4*>(g + (int64_t)n0 * OH * OW * C2_F);
        constexpr int LD = 4;
        for (int o0 = tid; o0 < vecs; o0 += 512 * LD) {
          cm_f4 v[LD];
#pragma unroll
          for (int k = 0; k < LD; ++k) { const int o = o0 + k * 512; v[k] = s4[o < vecs ? o : vecs - 1]; }
#pragma unroll
          for (int k = 0; k < LD; ++k) { const int o = o0 + k * 512; if (o < vecs) put(v[k], o); }
        }
      }
    }
    if (pf && u0 + (int)gridDim.x < units) request(u0 + gridDim.x);
    g3_barrier();                               // LDS writes visible; the requests just issued stay in flight (no vmcnt wait)
    const int pend = frames * UV;
    // Software pipeline over half tiles (4 K-steps = 12 fragments = 48 VGPRs): while the 24 MFMAs of one half run, the
    // other half's ds_read_b128 are in flight — left to itself the compiler reads each fragment right before its MFMA
    // (s_waitcnt lgkmcnt(0) in front of every second MFMA).  sched_barrier keeps the phases apart.
    int f, u, v;
    auto locate = [&](int p0) -> const char* {
      const int p = p0 + j, pc = p < pend ? p : pend - 1;
      f = (FPI > 1 && pc >= UV) ? 1 : 0;
      const int r = pc - f * UV;
      u = V == 1 ? r : (int)__umulhi((unsigned)r, v_magic);
      v = r - u * V;
      // position (u - a, v - b) of frame f inside the bordered grid, this lane's filter eighth of a 32-filter K-step
      return c2b_lds + (kq & 1) * HP + (kq >> 1) * S + ((f * (OH + 1) + 1 + u) * PW + 1 + v) * 16;
    };
    auto load_half = [&](g3_bf16x8 (&b)[4][3], const char* base, int hs) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int s = hs * 4 + q;               // K-step: tap s >> 1 = (a, b), filters 32 (s & 1) ..
        const char* pb = base - (((s >> 2) & 1) * PW + ((s >> 1) & 1)) * 16 + (s & 1) * 2 * S;
        b[q][0] = *reinterpret_cast<const g3_bf16x8*>(pb);
        b[q][1] = *reinterpret_cast<const g3_bf16x8*>(pb + plane);
        b[q][2] = *reinterpret_cast<const g3_bf16x8*>(pb + 2 * plane);
      }
    };
    // (one accumulator chain: a second independent chain measured no faster — the workgroup's phases, not the MFMA
    //  dependency, bound this kernel: see the note at the end of the file)
    auto mfma_half = [&](g3_f32x4 acc, const g3_bf16x8 (&b)[4][3], int hs) -> g3_f32x4 {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int s = hs * 4 + q;
        // smallest products first: (lo,hi) (hi,lo) (mid,mid) (mid,hi) (hi,mid) (hi,hi)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[s][2], b[q][0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[s][0], b[q][2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[s][1], b[q][1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[s][1], b[q][0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[s][0], b[q][1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[s][0], b[q][0], acc, 0, 0, 0);
      }
      return acc;
    };
    g3_bf16x8 b0[4][3], b1[4][3];
    const char* base = locate(0);
    load_half(b0, base, 0);
    for (int p0 = 0; p0 < pend; p0 += 16) {
      load_half(b1, base, 1);
      float* dst = dx + ((((int64_t)(n0 + f) * IH + 2 * u + ph) * IW) + 2 * v + pw) * C2_C + 16 * m + 4 * kq;
      const bool live = p0 + j < pend;
      __builtin_amdgcn_sched_barrier(0);
      g3_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = mfma_half(acc, b0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (p0 + 16 < pend) {                      // workgroup-uniform
        base = locate(p0 + 16);
        load_half(b0, base, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      acc = mfma_half(acc, b1, 1);
      __builtin_amdgcn_sched_barrier(0);
      if (live) *reinterpret_cast<g3_f32x4*>(dst) = acc;
    }
  }
}

static size_t c2b_lds_bytes(int fpi, int OH, int OW) { return (size_t)3 * 2 * c2b_half_plane((fpi * (OH + 1) + 1) * (OW + 2)); }

// ---------------------------------------------------------------------------------------------------------------
// Layer 3's data gradient (64 -> 64 filters, kernel 3, stride 1: autograd of cnn.py:47-49 at the Atari models' third conv
// layer; MIOpen's igemm_bwd ran it at 2.1 ms per learner step on the f32 pipe) with the machinery above:
//   dx[y][x][c] = sum over kh, kw < 3, f < 64 of g[y - kh][x - kw][f] * W[f][c][kh][kw]        (g zero outside the frame)
// One GEMM per frame — rows = 16 of the 64 channels, columns = 16 pixels, K = 9 taps x 64 filters = 18 K-steps of 32 — but
// 16 channels x 576 are 216 VGPRs as three bf16 parts, so the 8 waves are 4 channel tiles x 2 K HALVES (9 K-steps, 108
// VGPRs of weights each): a wave multiplies its half for ALL pixel tiles of the fill (<= 11 accumulator tiles), then the
// upper halves leave their partial tiles in LDS (the g planes are dead by then) and the lower halves add and store.
// g is staged exactly as above (three bf16 parts, the conflict-free chunk layout) inside a TWO-position zero border.
constexpr int C3B_C = 64, C3B_F = 64, C3B_K = 3;
constexpr int C3B_HS = 9;                      // K-steps per K half (18 in all: tap s >> 1, filters 32 (s & 1) ..)
constexpr int C3B_SLOTS = 4 * 2 * C3B_HS;      // (channel tile, K half, step)
constexpr int C3B_WPK_BYTES = C3B_SLOTS * 3 * 64 * 16;
constexpr int C3B_MAXT = 11;                   // pixel tiles per fill: two 9 x 9 frames

__global__ void __launch_bounds__(256)
k_conv3_pack_w3b(const float* __restrict__ w, int64_t so, int64_t sc, int64_t sh, int64_t sw, uint4* __restrict__ wpk3) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= C3B_SLOTS * 64) return;
  const int lane = t & 63, slot = t >> 6, q = slot % C3B_HS, kh2 = (slot / C3B_HS) & 1, ct = slot / (2 * C3B_HS);
  const int st = kh2 * C3B_HS + q, tap = st >> 1, kh = tap / C3B_K, kw = tap - kh * C3B_K;
  const int c = 16 * ct + (lane & 15), f0 = 32 * (st & 1) + 8 * (lane >> 4);
  float x[2][4];
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e >> 2][e & 3] = w[(f0 + e) * so + c * sc + kh * sh + kw * sw];
  uint2 h0, m0, l0, h1, m1, l1;
  g3_split4(x[0], h0, m0, l0);
  g3_split4(x[1], h1, m1, l1);
  wpk3[(slot * 3 + 0) * 64 + lane] = make_uint4(h0.x, h0.y, h1.x, h1.y);
  wpk3[(slot * 3 + 1) * 64 + lane] = make_uint4(m0.x, m0.y, m1.x, m1.y);
  wpk3[(slot * 3 + 2) * 64 + lane] = make_uint4(l0.x, l0.y, l1.x, l1.y);
}

// g: float [N][OH][OW][64]; dx: float [N][OH + 2][OW + 2][64] (NHWC memory)
template <int FPI>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_conv3_bwd_data_b3(int N, int OH, int OW, unsigned iw_magic, const float* __restrict__ g, const uint4* __restrict__ wpk3,
                    float* __restrict__ dx) {
  extern __shared__ __align__(16) char c3b_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = wave >> 1, kh2 = wave & 1;
  const int j = lane & 15, kq = lane >> 4;
  g3_bf16x8 wr[C3B_HS][3];
#pragma unroll
  for (int q = 0; q < C3B_HS; ++q)
#pragma unroll
    for (int p = 0; p < 3; ++p)
      wr[q][p] = __builtin_bit_cast(g3_bf16x8, wpk3[((((ct * 2 + kh2) * C3B_HS + q) * 3) + p) * 64 + lane]);
  const int IH = OH + 2, IW = OW + 2, IHW = IH * IW, PW = OW + 4, rows = FPI * (OH + 2) + 2;
  const int pos16 = OH * OW * 16;               // 16 B vectors per frame of g
  const int S = c2b_chunk_stride(rows * PW), HP = c2b_half_plane(rows * PW), plane = 2 * HP;
  // this wave's nine K-steps: byte offset of the tap's position shift and of the 32-filter half
  int tsh[C3B_HS], fsh[C3B_HS];
#pragma unroll
  for (int q = 0; q < C3B_HS; ++q) {
    const int st = kh2 * C3B_HS + q, tap = st >> 1, kh = tap / C3B_K, kw = tap - kh * C3B_K;
    tsh[q] = (kh * PW + kw) * 16;
    fsh[q] = (st & 1) * 2 * S;
  }
  for (int o = tid; o < 3 * plane / 16; o += 512) reinterpret_cast<uint4*>(c3b_lds)[o] = make_uint4(0u, 0u, 0u, 0u);
  const int units = (N + FPI - 1) / FPI;
  bool first = true;
  for (int u0 = blockIdx.x; u0 < units; u0 += gridDim.x) {
    const int n0 = u0 * FPI;
    const int frames = N - n0 < FPI ? N - n0 : FPI;
    __syncthreads();                            // zeroing / the lower halves are done with the exchange blocks
    if (!first) {
      // the exchange blocks of the previous fill overwrote part of the planes, border included: zero that part again
      for (int o = tid; o < 4 * C3B_MAXT * 64; o += 512) reinterpret_cast<uint4*>(c3b_lds)[o] = make_uint4(0u, 0u, 0u, 0u);
      __syncthreads();
    }
    first = false;
    {
      const int vecs = frames * pos16;
      const cm_f4* s4 = reinterpret_cast<const cm_f4*>(g + (int64_t)n0 * OH * OW * C3B_F);
      constexpr int LD = 4;
      for (int o0 = tid; o0 < vecs; o0 += 512 * LD) {
        cm_f4 v[LD];
#pragma unroll
        for (int k = 0; k < LD; ++k) { const int o = o0 + k * 512; v[k] = s4[o < vecs ? o : vecs - 1]; }
#pragma unroll
        for (int k = 0; k < LD; ++k) {
          const int o = o0 + k * 512;
          if (o < vecs) {
            const int f = o / pos16, r = o - f * pos16, pos = r >> 4, sub = r & 15;
            const int oh2 = pos / OW, ow2 = pos - oh2 * OW;
            const float x[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
            uint2 hh, mm, ll;
            g3_split4(x, hh, mm, ll);
            const int chunk = sub >> 1, kqw = chunk & 3;
            char* d = c3b_lds + (kqw & 1) * HP + (2 * (chunk >> 2) + (kqw >> 1)) * S + ((f * (OH + 2) + 2 + oh2) * PW + 2 + ow2) * 16 + (sub & 1) * 8;
            *reinterpret_cast<uint2*>(d) = hh;
            *reinterpret_cast<uint2*>(d + plane) = mm;
            *reinterpret_cast<uint2*>(d + 2 * plane) = ll;
          }
        }
      }
    }
    __syncthreads();
    const int pend = frames * IHW;
    g3_f32x4 acc[C3B_MAXT];
#pragma unroll
    for (int t = 0; t < C3B_MAXT; ++t) {
      acc[t] = g3_f32x4{0.f, 0.f, 0.f, 0.f};
      if (t * 16 < pend) {                       // workgroup-uniform
        const int p = t * 16 + j, pc = p < pend ? p : pend - 1;
        const int f = pc / IHW, r = pc - f * IHW;
        const int y = (int)__umulhi((unsigned)r, iw_magic), x = r - y * IW;
        const char* base = c3b_lds + (kq & 1) * HP + (kq >> 1) * S + ((f * (OH + 2) + 2 + y) * PW + 2 + x) * 16;
        g3_bf16x8 b[C3B_HS][3];
#pragma unroll
        for (int q = 0; q < C3B_HS; ++q) {
          const char* pb = base - tsh[q] + fsh[q];
          b[q][0] = *reinterpret_cast<const g3_bf16x8*>(pb);
          b[q][1] = *reinterpret_cast<const g3_bf16x8*>(pb + plane);
          b[q][2] = *reinterpret_cast<const g3_bf16x8*>(pb + 2 * plane);
        }
        g3_f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < C3B_HS; ++q) {
          // smallest products first: (lo,hi) (hi,lo) (mid,mid) (mid,hi) (hi,mid) (hi,hi)
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[q][2], b[q][0], a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[q][0], b[q][2], a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[q][1], b[q][1], a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[q][1], b[q][0], a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[q][0], b[q][1], a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[q][0], b[q][0], a, 0, 0, 0);
        }
        acc[t] = a;
      }
    }
    // the two K halves of a channel tile meet in LDS: block (ct, t) = one 16-byte vector per lane
    __syncthreads();                            // every wave is done reading the planes
    g3_f32x4* xb = reinterpret_cast<g3_f32x4*>(c3b_lds) + (ct * C3B_MAXT) * 64 + lane;
    if (kh2 == 1) {
#pragma unroll
      for (int t = 0; t < C3B_MAXT; ++t) if (t * 16 < pend) xb[t * 64] = acc[t];
    }
    __syncthreads();
    if (kh2 == 0) {
#pragma unroll
      for (int t = 0; t < C3B_MAXT; ++t) {
        const int p = t * 16 + j;
        if (p < pend) {
          const int f = p / IHW, r = p - f * IHW;
          *reinterpret_cast<g3_f32x4*>(dx + ((int64_t)(n0 + f) * IHW + r) * C3B_C + 16 * ct + 4 * kq) = acc[t] + xb[t * 64];
        }
      }
    }
  }
}

// the planes, and at least the exchange blocks that reuse their space (4 channel tiles x C3B_MAXT pixel tiles x 1 KB)
static size_t c3b_lds_bytes(int fpi, int OH, int OW) {
  const size_t planes = (size_t)3 * 2 * c2b_half_plane((fpi * (OH + 2) + 2) * (OW + 4)), xchg = (size_t)4 * C3B_MAXT * 1024;
  return planes > xchg ? planes : xchg;
}

}  // namespace mirl

extern "C" int mirl_conv2_bwd_data_supported(int32_t C, int32_t F, int32_t K, int32_t S, int32_t IH, int32_t IW, int32_t OH, int32_t OW) {
  using namespace mirl;
  if (C != C2_C || F != C2_F || K != C2_K || S != C2_S || OH < 1 || OW < 1) return 0;
  if (IH != 2 * OH + 2 || IW != 2 * OW + 2) return 0;      // no forward rows / columns left uncovered
  return c2_lds_bytes(1, OH, OW) <= 64 * 1024 ? 1 : 0;
}

extern "C" int mirl_conv2_bwd_data(int64_t N, int32_t OH, int32_t OW, const float* g, const float* weight, int64_t ws_o,
                                   int64_t ws_c, int64_t ws_h, int64_t ws_w, float* wpk, float* dx, void* stream) {
  using namespace mirl;
  if (N <= 0 || N >= (1LL << 30) || !g || !weight || !wpk || !dx) return fail(MIRL_ERR_ARG, "bad conv2_bwd_data arguments");
  if (!mirl_conv2_bwd_data_supported(C2_C, C2_F, C2_K, C2_S, 2 * OH + 2, 2 * OW + 2, OH, OW))
    return fail(MIRL_ERR_ARG, "conv2_bwd_data: unsupported shape");
  if (((uintptr_t)g % 16) || ((uintptr_t)dx % 16) || ((uintptr_t)wpk % 16))
    return fail(MIRL_ERR_ARG, "conv2_bwd_data: pointers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  {
    ProfScope ps("k_conv2_pack_w", 2.0 * C2_WPK * 4, st);
    hipLaunchKernelGGL(k_conv2_pack_w, dim3((C2_WPK + 255) / 256), dim3(256), 0, st, weight, ws_o, ws_c, ws_h, ws_w, wpk);
    MIRL_LAUNCH_CHECK();
  }
  const int fpi = (N >= 1024 && c2_lds_bytes(2, OH, OW) <= 64 * 1024) ? 2 : 1;
  const int64_t units = (N + fpi - 1) / fpi;
  const unsigned grid = (unsigned)(units < 512 ? units : 512);
  const int V = OW + 1;
  const unsigned v_magic = V > 1 ? (unsigned)(((1ULL << 32) + V - 1) / V) : 0u;
  // flop: the convolution's own 2 * K*K*C*F per output position (the zero border the kernel multiplies instead of masking
  // edges is issued work, not algorithmic work)
  ProfScope ps("k_conv2_bwd_data", (double)N * ((double)OH * OW * C2_F * 4 + (double)(2 * OH + 2) * (2 * OW + 2) * C2_C * 4), st,
               (double)N * OH * OW * 2.0 * C2_K * C2_K * C2_C * C2_F);
  if (fpi == 2) hipLaunchKernelGGL((k_conv2_bwd_data<2>), dim3(grid), dim3(256), c2_lds_bytes(2, OH, OW), st, (int)N, OH, OW, v_magic, g, wpk, dx);
  else          hipLaunchKernelGGL((k_conv2_bwd_data<1>), dim3(grid), dim3(256), c2_lds_bytes(1, OH, OW), st, (int)N, OH, OW, v_magic, g, wpk, dx);
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}

extern "C" int mirl_conv2_bwd_data_wpk_floats(int64_t* floats) {
  if (!floats) return mirl::fail(MIRL_ERR_ARG, "conv2_bwd_data_wpk_floats: null out");
  *floats = mirl::C2B_WPK_BYTES / 4 > mirl::C2_WPK ? mirl::C2B_WPK_BYTES / 4 : mirl::C2_WPK;
  return MIRL_OK;
}

// pipe: 0 = f32 MFMA (mirl_conv2_bwd_data above), 1 = bf16 MFMA with the exact three-way split (f32 results)
extern "C" int mirl_conv2_bwd_data_ex(int64_t N, int32_t OH, int32_t OW, const float* g, const float* weight, int64_t ws_o,
                                      int64_t ws_c, int64_t ws_h, int64_t ws_w, float* wpk, int64_t wpk_floats, float* dx, int32_t pipe, void* stream) {
  using namespace mirl;
  if (wpk_floats < (pipe == 1 ? C2B_WPK_BYTES / 4 : C2_WPK)) return fail(MIRL_ERR_ARG, "conv2_bwd_data_ex: wpk smaller than mirl_conv2_bwd_data_wpk_floats()");
  if (pipe == 0) return mirl_conv2_bwd_data(N, OH, OW, g, weight, ws_o, ws_c, ws_h, ws_w, wpk, dx, stream);
  if (pipe != 1) return fail(MIRL_ERR_ARG, "conv2_bwd_data_ex: pipe is 0 (f32) or 1 (split bf16)");
  if (N <= 0 || N >= (1LL << 30) || !g || !weight || !wpk || !dx) return fail(MIRL_ERR_ARG, "bad conv2_bwd_data arguments");
  if (!mirl_conv2_bwd_data_supported(C2_C, C2_F, C2_K, C2_S, 2 * OH + 2, 2 * OW + 2, OH, OW) || c2b_lds_bytes(1, OH, OW) > 160 * 1024)
    return fail(MIRL_ERR_ARG, "conv2_bwd_data: unsupported shape");
  if (((uintptr_t)g % 16) || ((uintptr_t)dx % 16) || ((uintptr_t)wpk % 16))
    return fail(MIRL_ERR_ARG, "conv2_bwd_data: pointers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  {
    ProfScope ps("k_conv2_pack_w3b", 4.0 * C2_WPK + C2B_WPK_BYTES, st);
    hipLaunchKernelGGL(k_conv2_pack_w3b, dim3((C2B_SLOTS * 64 + 255) / 256), dim3(256), 0, st, weight, ws_o, ws_c, ws_h, ws_w,
                       reinterpret_cast<uint4*>(wpk));
    MIRL_LAUNCH_CHECK();
  }
  const int fpi = (N >= 512 && c2b_lds_bytes(2, OH, OW) <= 160 * 1024) ? 2 : 1;
  const size_t lds = c2b_lds_bytes(fpi, OH, OW);
  const int64_t units = (N + fpi - 1) / fpi;
  const unsigned grid = (unsigned)(units < 256 ? units : 256);
  const int V = OW + 1;
  const unsigned v_magic = V > 1 ? (unsigned)(((1ULL << 32) + V - 1) / V) : 0u;
  static bool attr[2] = {false, false};
  const void* fn = fpi == 2 ? (const void*)k_conv2_bwd_data_b3<2> : (const void*)k_conv2_bwd_data_b3<1>;
  if (!attr[fpi - 1]) { MIRL_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr[fpi - 1] = true; }
  ProfScope ps("k_conv2_bwd_data_b3", (double)N * ((double)OH * OW * C2_F * 4 + (double)(2 * OH + 2) * (2 * OW + 2) * C2_C * 4), st,
               (double)N * OH * OW * 2.0 * C2_K * C2_K * C2_C * C2_F);
  if (fpi == 2) hipLaunchKernelGGL((k_conv2_bwd_data_b3<2>), dim3(grid), dim3(512), lds, st, (int)N, OH, OW, v_magic, g, reinterpret_cast<const uint4*>(wpk), dx);
  else          hipLaunchKernelGGL((k_conv2_bwd_data_b3<1>), dim3(grid), dim3(512), lds, st, (int)N, OH, OW, v_magic, g, reinterpret_cast<const uint4*>(wpk), dx);
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}

// Measured (MI355X, 40 960 frames of 9 x 9 x 64): 1.46-1.50 ms against 2.10-2.13 ms for the f32-pipe kernel above = 0.35 of
// the bf16/6 peak.  Switching pieces off one at a time (timing experiments, not kept in the code): MFMAs 0.52 ms, dx stores
// 0.26, the split + LDS writes of the staging 0.25, everything else (g loads, LDS fragment reads, barriers) 0.83 — the
// eight waves of the ONE workgroup a CU holds (93.7 KB of LDS, 247 VGPRs) go through stage -> read -> multiply -> store
// in lockstep, so the phases add up instead of overlapping; conflict-free fragment reads, half-tile software pipelining
// and a second accumulator chain each moved the total by less than 0.05 ms.  Next step: two staging buffers (FPI = 1) so
// the next frame's split runs under this frame's MFMAs.

extern "C" int mirl_conv3_bwd_data_supported(int32_t C, int32_t F, int32_t K, int32_t S, int32_t IH, int32_t IW, int32_t OH, int32_t OW) {
  using namespace mirl;
  if (C != C3B_C || F != C3B_F || K != C3B_K || S != 1 || OH < 1 || OW < 1) return 0;
  if (IH != OH + 2 || IW != OW + 2) return 0;                 // no forward rows / columns left uncovered
  if (IH * IW > 16 * C3B_MAXT) return 0;                      // a frame's pixel tiles fit the accumulator array
  return c3b_lds_bytes(1, OH, OW) <= 150 * 1024 ? 1 : 0;
}

extern "C" int mirl_conv3_bwd_data_wpk_floats(int64_t* floats) {
  if (!floats) return mirl::fail(MIRL_ERR_ARG, "conv3_bwd_data_wpk_floats: null out");
  *floats = mirl::C3B_WPK_BYTES / 4;
  return MIRL_OK;
}

extern "C" int mirl_conv3_bwd_data(int64_t N, int32_t OH, int32_t OW, const float* g, const float* weight, int64_t ws_o,
                                   int64_t ws_c, int64_t ws_h, int64_t ws_w, float* wpk, int64_t wpk_floats, float* dx, void* stream) {
  using namespace mirl;
  if (N <= 0 || N >= (1LL << 30) || !g || !weight || !wpk || !dx) return fail(MIRL_ERR_ARG, "bad conv3_bwd_data arguments");
  if (wpk_floats < C3B_WPK_BYTES / 4) return fail(MIRL_ERR_ARG, "conv3_bwd_data: wpk smaller than mirl_conv3_bwd_data_wpk_floats()");
  if (!mirl_conv3_bwd_data_supported(C3B_C, C3B_F, C3B_K, 1, OH + 2, OW + 2, OH, OW)) return fail(MIRL_ERR_ARG, "conv3_bwd_data: unsupported shape");
  if (((uintptr_t)g % 16) || ((uintptr_t)dx % 16) || ((uintptr_t)wpk % 16))
    return fail(MIRL_ERR_ARG, "conv3_bwd_data: pointers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  {
    ProfScope ps("k_conv3_pack_w3b", 4.0 * C3B_F * C3B_C * 9 + C3B_WPK_BYTES, st);
    hipLaunchKernelGGL(k_conv3_pack_w3b, dim3((C3B_SLOTS * 64 + 255) / 256), dim3(256), 0, st, weight, ws_o, ws_c, ws_h, ws_w,
                       reinterpret_cast<uint4*>(wpk));
    MIRL_LAUNCH_CHECK();
  }
  const int IHW = (OH + 2) * (OW + 2);
  const int fpi = (N >= 512 && 2 * IHW <= 16 * C3B_MAXT && c3b_lds_bytes(2, OH, OW) <= 150 * 1024) ? 2 : 1;
  const size_t lds = c3b_lds_bytes(fpi, OH, OW);
  const int64_t units = (N + fpi - 1) / fpi;
  const unsigned grid = (unsigned)(units < 256 ? units : 256);
  const int IW = OW + 2;
  const unsigned iw_magic = (unsigned)(((1ULL << 32) + IW - 1) / IW);
  static bool attr[2] = {false, false};
  const void* fn = fpi == 2 ? (const void*)k_conv3_bwd_data_b3<2> : (const void*)k_conv3_bwd_data_b3<1>;
  if (!attr[fpi - 1]) { MIRL_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); attr[fpi - 1] = true; }
  ProfScope ps("k_conv3_bwd_data_b3", (double)N * 4.0 * C3B_F * ((double)OH * OW + (double)IHW), st,
               (double)N * OH * OW * 2.0 * C3B_K * C3B_K * C3B_C * C3B_F);
  if (fpi == 2) hipLaunchKernelGGL((k_conv3_bwd_data_b3<2>), dim3(grid), dim3(512), lds, st, (int)N, OH, OW, iw_magic, g, reinterpret_cast<const uint4*>(wpk), dx);
  else          hipLaunchKernelGGL((k_conv3_bwd_data_b3<1>), dim3(grid), dim3(512), lds, st, (int)N, OH, OW, iw_magic, g, reinterpret_cast<const uint4*>(wpk), dx);
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}
