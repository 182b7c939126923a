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

}  // namespace mirl

using namespace mirl;

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
