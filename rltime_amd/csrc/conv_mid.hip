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
