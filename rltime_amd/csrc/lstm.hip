// lstm.hip — fused LSTM-cell pointwise kernels for the recurrent core.
//
// The reference runs T sequential torch.nn.LSTMCell steps with a per-step
// state reset on `initials` (rltime/models/torch/modules/lstm.py:83-116); in
// PyTorch that is ~12 small launches per step forward and ~25 backward.  Here
// one step is one recurrent GEMM (rocBLAS, through torch) plus ONE of these
// kernels; the input projection of all T steps and the weight gradient are
// single large GEMMs outside the loop (rltime_amd/models/torch/lstm_seq.py).
//
// Gate order i, f, g, o as torch.nn.LSTMCell.  `gates` holds pre-activations on
// entry and the activated gates on exit of the forward kernel (kept for the
// backward); the backward overwrites them with d(loss)/d(pre-activation).
// Resets: h_in(t) = h(t-1) * keep(t), c_in(t) = c(t-1) * keep(t), keep = 1 - initials.
// HBM-bound elementwise work: 4H+H floats in, 4H+4H floats out per row.
#include "common.hpp"

namespace mirl {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// one lane per (b, j), j < H
__global__ void __launch_bounds__(256)
k_lstm_cell_fwd(int B, int H, float* __restrict__ gates, const float* __restrict__ c_in,
                const float* __restrict__ keep_next, float* __restrict__ h_out, float* __restrict__ c_out,
                float* __restrict__ h_next, float* __restrict__ c_next) {
  int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)B * H) return;
  int b = (int)(idx / H), j = (int)(idx - (int64_t)b * H);
  float* g = gates + (int64_t)b * 4 * H;
  float i = sigmoidf_(g[j]), f = sigmoidf_(g[H + j]), gg = tanhf(g[2 * H + j]), o = sigmoidf_(g[3 * H + j]);
  float c = f * c_in[idx] + i * gg;
  float h = o * tanhf(c);
  g[j] = i; g[H + j] = f; g[2 * H + j] = gg; g[3 * H + j] = o;
  if (h_out) h_out[idx] = h;
  if (c_out) c_out[idx] = c;
  float k = keep_next ? keep_next[b] : 1.0f;
  h_next[idx] = h * k;
  c_next[idx] = c * k;
}

// gates: activated (in) -> d pre-activation (out).  dh_rec / dc_rec are the
// gradients flowing back from step t+1 w.r.t. its (masked) inputs; dc_rec is
// updated in place to the gradient w.r.t. this step's c_in.
__global__ void __launch_bounds__(256)
k_lstm_cell_bwd(int B, int H, float* __restrict__ gates, const float* __restrict__ c_t,
                const float* __restrict__ c_in, const float* __restrict__ d_out, const float* __restrict__ dh_rec,
                float* __restrict__ dc_rec, const float* __restrict__ keep_next, int first) {
  int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)B * H) return;
  int b = (int)(idx / H), j = (int)(idx - (int64_t)b * H);
  float* g = gates + (int64_t)b * 4 * H;
  float i = g[j], f = g[H + j], gg = g[2 * H + j], o = g[3 * H + j];
  float k = keep_next ? keep_next[b] : 1.0f;
  float dh = (d_out ? d_out[idx] : 0.0f) + (first ? 0.0f : dh_rec[idx] * k);
  float tc = tanhf(c_t[idx]);
  float dc = (first ? 0.0f : dc_rec[idx] * k) + dh * o * (1.0f - tc * tc);
  g[j] = dc * gg * i * (1.0f - i);
  g[H + j] = dc * c_in[idx] * f * (1.0f - f);
  g[2 * H + j] = dc * i * (1.0f - gg * gg);
  g[3 * H + j] = dh * tc * o * (1.0f - o);
  dc_rec[idx] = dc * f;
}

// ---------------------------------------------------------------------------
// One LSTM timestep in ONE launch: the recurrent GEMM on f32 MFMA with the cell in
// its epilogue.  gates[b][g*H + j] += sum_k h_in[b][k] * W_hh[g*H + j][k]
// is a real dense contraction (512 x 512 x 2048 per step at config D), so it runs on
// v_mfma_f32_16x16x4_f32 (exact f32: a k-ordered fmaf chain); what rocBLAS + the
// pointwise kernel did in two launches, with the 4 MB gate tensor written and
// re-read in between, stays inside one workgroup:
//   * workgroup = 8 waves = a (32 batch x 16 hidden x 4 gates) tile; wave (m, g)
//     owns the 16x16 block of batch half m and gate g.  Grid (H/16, B/32) = 512
//     workgroups at B = H = 512: 16 waves per CU = 4 per SIMD, so one wave's global /
//     LDS latencies, barriers and epilogue hide behind the others' MFMAs (a first
//     version with 32x32x2 tiles had ONE wave per SIMD and measured 21 us per step);
//   * K streamed in chunks of 64 through double-buffered LDS (global -> registers for
//     chunk c+1 while chunk c feeds the MFMAs); rows padded to 68 floats: 16-byte
//     aligned b128 stores and conflict-free operand reads (bank = 4 row + k);
//   * all 16 operand pairs of a chunk are requested from LDS before the first MFMA;
//   * epilogue: the four gate blocks meet in LDS, every lane activates ONE element
//     (sigmoid / tanh), updates c, h, applies the next step's reset mask and writes
//     the activated gates for the backward pass.
// Operand maps (cdna_hip_programming.md section 3): lane l supplies A[i=l&15][k=l>>4],
// B[k=l>>4][j=l&15]; D: col = l&15, row = (l>>4)*4 + r, r in [0,4).
typedef float ls_f4 __attribute__((ext_vector_type(4)));
#define LS_KC 64
#define LS_LD 68
#define LS_ROWS 96            // 32 rows of h_in + 4 x 16 rows of W_hh

__global__ void __launch_bounds__(512)
k_lstm_step_fwd(int B, int H, const float* __restrict__ h_in, const float* __restrict__ w, float* __restrict__ gates,
                const float* __restrict__ c_in, const float* __restrict__ keep_next, float* __restrict__ h_out,
                float* __restrict__ c_out, float* __restrict__ h_next, float* __restrict__ c_next) {
  __shared__ __attribute__((aligned(16))) float sT[2][LS_ROWS * LS_LD];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave & 1, wg = wave >> 1;
  const int j0 = blockIdx.x * 16, b0 = blockIdx.y * 32;
  // loader: 96 rows x 16 quads = 1536 float4, three per lane
  const float* src[3];
  int dst[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int idx = tid + 512 * i, row = idx >> 4, kq = (idx & 15) * 4;
    dst[i] = row * LS_LD + kq;
    if (row < 32) src[i] = h_in + (int64_t)(b0 + row) * H + kq;
    else { const int g = (row - 32) >> 4, j = (row - 32) & 15; src[i] = w + ((int64_t)g * H + j0 + j) * H + kq; }
  }
  ls_f4 reg[3];
  ls_f4 acc = {0.f, 0.f, 0.f, 0.f};
  const int chunks = H / LS_KC;
#pragma unroll
  for (int i = 0; i < 3; ++i) reg[i] = *(const ls_f4*)(src[i]);
#pragma unroll
  for (int i = 0; i < 3; ++i) *(ls_f4*)&sT[0][dst[i]] = reg[i];
  __syncthreads();
  const int oa = (16 * wm + (lane & 15)) * LS_LD + (lane >> 4);
  const int ob = (32 + 16 * wg + (lane & 15)) * LS_LD + (lane >> 4);
  for (int c = 0; c < chunks; ++c) {
    if (c + 1 < chunks) {
#pragma unroll
      for (int i = 0; i < 3; ++i) reg[i] = *(const ls_f4*)(src[i] + (c + 1) * LS_KC);
    }
    const float* t = sT[c & 1];
    float av[LS_KC / 4], bv[LS_KC / 4];
#pragma unroll
    for (int kk = 0; kk < LS_KC / 4; ++kk) { av[kk] = t[oa + 4 * kk]; bv[kk] = t[ob + 4 * kk]; }
    __builtin_amdgcn_sched_barrier(0);        // keep the LDS reads ahead of the MFMA chain
#pragma unroll
    for (int kk = 0; kk < LS_KC / 4; ++kk)
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk], bv[kk], acc, 0, 0, 0);
    if (c + 1 < chunks) {
#pragma unroll
      for (int i = 0; i < 3; ++i) *(ls_f4*)&sT[(c + 1) & 1][dst[i]] = reg[i];
    }
    __syncthreads();
  }
  // the four gate blocks meet in LDS (buffer 0 is free after the last barrier)
  float* G = sT[0];                                    // [gate][32 rows][17]
#pragma unroll
  for (int r = 0; r < 4; ++r)
    G[(wg * 32 + 16 * wm + (lane >> 4) * 4 + r) * 17 + (lane & 15)] = acc[r];
  __syncthreads();
  {
    const int row = tid >> 4, col = tid & 15;
    const int b = b0 + row, j = j0 + col;
    float* g = gates + (int64_t)b * 4 * H + j;
    const float pi = G[(0 * 32 + row) * 17 + col] + g[0];
    const float pf = G[(1 * 32 + row) * 17 + col] + g[H];
    const float pg = G[(2 * 32 + row) * 17 + col] + g[2 * H];
    const float po = G[(3 * 32 + row) * 17 + col] + g[3 * H];
    const float i = sigmoidf_(pi), f = sigmoidf_(pf), gg = tanhf(pg), o = sigmoidf_(po);
    const int64_t idx = (int64_t)b * H + j;
    const float cc = f * c_in[idx] + i * gg;
    const float hh = o * tanhf(cc);
    g[0] = i; g[H] = f; g[2 * H] = gg; g[3 * H] = o;
    if (h_out) h_out[idx] = hh;
    if (c_out) c_out[idx] = cc;
    const float k = keep_next ? keep_next[b] : 1.0f;
    h_next[idx] = hh * k;
    c_next[idx] = cc * k;
  }
}

}  // namespace mirl

using namespace mirl;

extern "C" int mirl_lstm_step_fwd(int32_t B, int32_t H, const float* h_in, const float* w_hh, float* gates, const float* c_in,
                                  const float* keep_next, float* h_out, float* c_out, float* h_next, float* c_next, void* stream) {
  if (B <= 0 || H <= 0 || !h_in || !w_hh || !gates || !c_in || !h_next || !c_next) return fail(MIRL_ERR_ARG, "bad lstm_step_fwd arguments");
  if ((B % 32) || (H % 64)) return fail(MIRL_ERR_ARG, "lstm_step_fwd needs a batch that is a multiple of 32 and a hidden size that is a multiple of 64");
  if (((uintptr_t)h_in % 16) || ((uintptr_t)w_hh % 16)) return fail(MIRL_ERR_ARG, "lstm_step_fwd needs 16-byte aligned h_in / w_hh");
  // algorithmic bytes: h_in + W_hh + c_in read, gates read + written, h, c, h_next, c_next written
  ProfScope ps("k_lstm_step_fwd", 4.0 * ((double)B * H * 2 + 4.0 * H * H + 8.0 * B * H + 4.0 * B * H), (hipStream_t)stream);
  hipLaunchKernelGGL(k_lstm_step_fwd, dim3(H / 16, B / 32), dim3(512), 0, (hipStream_t)stream, (int)B, (int)H, h_in, w_hh, gates, c_in,
                     keep_next, h_out, c_out, h_next, c_next);
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}

extern "C" int mirl_lstm_cell_fwd(int32_t B, int32_t H, float* gates, const float* c_in, const float* keep_next,
                                  float* h_out, float* c_out, float* h_next, float* c_next, void* stream) {
  if (B <= 0 || H <= 0 || !gates || !c_in || !h_next || !c_next) return fail(MIRL_ERR_ARG, "bad lstm_cell_fwd arguments");
  int64_t n = (int64_t)B * H;
  ProfScope ps("k_lstm_cell_fwd", (double)n * 4.0 * (4 + 1 + 4 + 2 + (h_out ? 1 : 0) + (c_out ? 1 : 0)), (hipStream_t)stream);
  hipLaunchKernelGGL(k_lstm_cell_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (int)B, (int)H, gates, c_in,
                     keep_next, h_out, c_out, h_next, c_next);
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}

extern "C" int mirl_lstm_cell_bwd(int32_t B, int32_t H, float* gates, const float* c_t, const float* c_in, const float* d_out,
                                  const float* dh_rec, float* dc_rec, const float* keep_next, int32_t first, void* stream) {
  if (B <= 0 || H <= 0 || !gates || !c_t || !c_in || !dc_rec || (!first && !dh_rec)) return fail(MIRL_ERR_ARG, "bad lstm_cell_bwd arguments");
  int64_t n = (int64_t)B * H;
  ProfScope ps("k_lstm_cell_bwd", (double)n * 4.0 * (4 + 2 + (d_out ? 1 : 0) + (first ? 0 : 2) + 4 + 1), (hipStream_t)stream);
  hipLaunchKernelGGL(k_lstm_cell_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (int)B, (int)H, gates, c_t,
                     c_in, d_out, dh_rec, dc_rec, keep_next, (int)first);
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}
