// optim.hip — the learner step's tail as two launches: global gradient norm -> clip -> Adam.
//
// reference: rltime/training/torch_trainer.py:177-199 (train_batch: backward, clip_grad_norm_, optimizer.step) with
// torch.optim.Adam (amsgrad off, no weight decay) behind it.  As PyTorch issues it that tail is ~15 launches for the
// norm and the clip (per-tensor norms, stack, norm, + 1e-6, reciprocal, x clip, clamp, for-each scale) plus the
// optimizer's own (a for-each Adam: ~10 multi-tensor launches; with device-side step counters — needed under graph
// capture — another ~2 x #parameters single-tensor ones).  Every one of them has the ~5 us floor of this part, and for
// the T = 1 configs (DQN B = 256: ~1 ms per learner step) that was a quarter of the step.
//
//   k_adam_sqsum   one workgroup per 4096-element chunk of the gradients: sum of squares (f32 per lane over its 16
//                  elements, f64 across the workgroup) -> partial[chunk]; a tensor's first workgroup also latches that
//                  tensor's step + 1 (torch keeps one counter per parameter: a parameter that skipped steps has its own).
//   k_adam_update  every workgroup adds the partials in the same fixed order (f64) -> norm -> coef = min(clip /
//                  (norm + 1e-6), 1) as clip_grad_norm_ computes it -> scales its chunk of the gradient (written back:
//                  the reference leaves clipped gradients in .grad), updates exp_avg, exp_avg_sq and the parameter
//                  with the bias corrections evaluated in f64 from its tensor's latched step (what the default, host-side Adam
//                  does; a `capturable` torch Adam evaluates them in f32 on the device).
// Tensor pointers travel by value in the kernel arguments (<= 32 tensors per launch), so a captured graph replays them
// as they were at capture time — the .grad tensors of a captured backward have fixed addresses.  HBM-bound: 7 streams
// of 4 bytes per parameter (g, m, v, p read; m, v, p written; + g written when the clip bites).
#include "common.hpp"

namespace mirl {

#define ADAM_MAXT 32
#define ADAM_CHUNK 4096

struct AdamTensors {
  float* p[ADAM_MAXT]; float* g[ADAM_MAXT]; float* m[ADAM_MAXT]; float* v[ADAM_MAXT]; float* step[ADAM_MAXT];
  long long n[ADAM_MAXT];
  int chunk0[ADAM_MAXT + 1];       // first chunk of tensor t within this launch
  int count;
  int chunk_base;                  // this launch's first chunk in the step-wide partial array
  int tensor_base;                 // ... and its first tensor
};

struct AdamHyper {
  double lr; const float* lr_dev; double beta1, beta2, eps, clip;
  int total_chunks;
  double* partial;                 // [total_chunks] sums of squares, then [tensors] the latched steps
  float* norm_out;                 // [2]: norm, norm * coef (may be NULL)
};

__device__ __forceinline__ int adam_find(const AdamTensors& t, int b) {
  int k = 0;
  for (int i = 1; i < t.count; ++i) k = (b >= t.chunk0[i]) ? i : k;
  return k;
}

__global__ void __launch_bounds__(256)
k_adam_sqsum(AdamTensors t, AdamHyper h) {
  __shared__ double red[256];
  const int b = blockIdx.x, k = adam_find(t, b), tid = threadIdx.x;
  const long long n = t.n[k], base = (long long)(b - t.chunk0[k]) * ADAM_CHUNK;
  const float* __restrict__ g = t.g[k];
  float s = 0.f;
  const bool vec = !((uintptr_t)g & 15) && base + ADAM_CHUNK <= n;
  if (vec) {
    const float4* g4 = (const float4*)(g + base);
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float4 x = g4[tid + 256 * i]; s += x.x * x.x; s += x.y * x.y; s += x.z * x.z; s += x.w * x.w; }
  } else {
    for (long long i = base + tid; i < base + ADAM_CHUNK && i < n; i += 256) { const float x = g[i]; s += x * x; }
  }
  red[tid] = (double)s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) { if (tid < w) red[tid] += red[tid + w]; __syncthreads(); }
  if (tid == 0) {
    h.partial[t.chunk_base + b] = red[0];
    if (b == t.chunk0[k]) h.partial[h.total_chunks + t.tensor_base + k] = (double)t.step[k][0] + 1.0;
  }
}

__global__ void __launch_bounds__(256)
k_adam_update(AdamTensors t, AdamHyper h) {
  __shared__ double red[256];
  const int b = blockIdx.x, k = adam_find(t, b), tid = threadIdx.x;
  double acc = 0.0;
  for (int i = tid; i < h.total_chunks; i += 256) acc += h.partial[i];
  red[tid] = acc;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) { if (tid < w) red[tid] += red[tid + w]; __syncthreads(); }
  const float norm = (float)sqrt(red[0]);
  float coef = 1.f;
  if (h.clip > 0.0) { coef = (float)h.clip / (norm + 1e-6f); coef = coef > 1.f ? 1.f : coef; }
  const double step = h.partial[h.total_chunks + t.tensor_base + k];
  const double lr = h.lr_dev ? (double)h.lr_dev[0] : h.lr;
  const double bc1 = 1.0 - pow(h.beta1, step), bc2 = 1.0 - pow(h.beta2, step);
  const float step_size = (float)(lr / bc1), bc2_sqrt = (float)sqrt(bc2), eps = (float)h.eps;
  const float b2 = (float)h.beta2, w1 = (float)(1.0 - h.beta1), w2 = (float)(1.0 - h.beta2);
  const long long n = t.n[k], base = (long long)(b - t.chunk0[k]) * ADAM_CHUNK;
  float* __restrict__ p = t.p[k]; float* __restrict__ g = t.g[k]; float* __restrict__ m = t.m[k]; float* __restrict__ v = t.v[k];
  const bool scale = h.clip > 0.0;          // the reference multiplies by coef even when it is 1 (x * 1.0f is exact)
  // one element exactly as the for-each Adam evaluates it: lerp, mul + addcmul, sqrt / bc2_sqrt + eps, addcdiv
#define ADAM_ELEM(P, G, M, V)                                         \
  { float g_ = (G); if (scale) g_ *= coef;                            \
    const float m_ = (M) + w1 * (g_ - (M));                           \
    const float v_ = (V) * b2 + (w2 * g_) * g_;                       \
    const float den_ = sqrtf(v_) / bc2_sqrt + eps;                    \
    (G) = g_; (M) = m_; (V) = v_; (P) = (P) - step_size * (m_ / den_); }
  const bool vec = !(((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) && base + ADAM_CHUNK <= n;
  if (vec) {
    float4* p4 = (float4*)(p + base); float4* g4 = (float4*)(g + base); float4* m4 = (float4*)(m + base); float4* v4 = (float4*)(v + base);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = tid + 256 * i;
      float4 pp = p4[j], gg = g4[j], mm = m4[j], vv = v4[j];
      ADAM_ELEM(pp.x, gg.x, mm.x, vv.x) ADAM_ELEM(pp.y, gg.y, mm.y, vv.y) ADAM_ELEM(pp.z, gg.z, mm.z, vv.z) ADAM_ELEM(pp.w, gg.w, mm.w, vv.w)
      p4[j] = pp; m4[j] = mm; v4[j] = vv;
      if (coef != 1.f) g4[j] = gg;
    }
  } else {
    for (long long i = base + tid; i < base + ADAM_CHUNK && i < n; i += 256) {
      float pp = p[i], gg = g[i], mm = m[i], vv = v[i];
      ADAM_ELEM(pp, gg, mm, vv)
      p[i] = pp; m[i] = mm; v[i] = vv;
      if (coef != 1.f) g[i] = gg;
    }
  }
#undef ADAM_ELEM
  if (tid == 0 && b == t.chunk0[k]) t.step[k][0] = (float)step;
  if (tid == 0 && t.chunk_base + b == 0 && h.norm_out) { h.norm_out[0] = norm; h.norm_out[1] = norm * coef; }
}

static int64_t adam_chunks(int32_t count, const int64_t* numel) {
  int64_t c = 0;
  for (int i = 0; i < count; ++i) c += (numel[i] + ADAM_CHUNK - 1) / ADAM_CHUNK;
  return c;
}

}  // namespace mirl

extern "C" int mirl_adam_clip_workspace_bytes(int32_t count, const int64_t* numel, int64_t* bytes) {
  using namespace mirl;
  if (count < 1 || !numel || !bytes) return fail(MIRL_ERR_ARG, "bad adam_clip_workspace_bytes arguments");
  for (int i = 0; i < count; ++i) if (numel[i] < 1) return fail(MIRL_ERR_ARG, "adam_clip: empty tensor");
  *bytes = (adam_chunks(count, numel) + count) * (int64_t)sizeof(double);
  return MIRL_OK;
}

extern "C" int mirl_adam_clip_step(int32_t count, float* const* param, float* const* grad, float* const* exp_avg,
                                   float* const* exp_avg_sq, float* const* step, const int64_t* numel, double lr,
                                   const float* lr_dev, double beta1, double beta2, double eps, double clip, void* workspace,
                                   int64_t workspace_bytes, float* norm_out, void* stream) {
  using namespace mirl;
  if (count < 1 || !param || !grad || !exp_avg || !exp_avg_sq || !step || !numel || !workspace)
    return fail(MIRL_ERR_ARG, "bad adam_clip_step arguments");
  if (!(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0) || (!lr_dev && !(lr >= 0.0)))
    return fail(MIRL_ERR_ARG, "adam_clip_step: 0 <= beta < 1, eps >= 0, lr >= 0");
  for (int i = 0; i < count; ++i)
    if (!param[i] || !grad[i] || !exp_avg[i] || !exp_avg_sq[i] || !step[i] || numel[i] < 1)
      return fail(MIRL_ERR_ARG, "adam_clip_step: null tensor / empty tensor");
  const int64_t total = adam_chunks(count, numel);
  if (total >= (1LL << 31)) return fail(MIRL_ERR_ARG, "adam_clip_step: too many elements");
  if (workspace_bytes < (total + count) * (int64_t)sizeof(double) || ((uintptr_t)workspace & 7))
    return fail(MIRL_ERR_ARG, "adam_clip_step: workspace smaller than mirl_adam_clip_workspace_bytes / misaligned");
  hipStream_t st = (hipStream_t)stream;
  AdamHyper h;
  h.lr = lr; h.lr_dev = lr_dev; h.beta1 = beta1; h.beta2 = beta2; h.eps = eps; h.clip = clip;
  h.total_chunks = (int)total; h.partial = (double*)workspace; h.norm_out = norm_out;
  double elems = 0.0;
  for (int i = 0; i < count; ++i) elems += (double)numel[i];
  for (int pass = 0; pass < 2; ++pass) {
    ProfScope ps(pass == 0 ? "k_adam_sqsum" : "k_adam_update", elems * 4.0 * (pass == 0 ? 1.0 : 7.0), st);
    int chunk_base = 0;
    for (int t0 = 0; t0 < count; t0 += ADAM_MAXT) {
      AdamTensors t;
      t.count = count - t0 < ADAM_MAXT ? count - t0 : ADAM_MAXT;
      t.chunk_base = chunk_base;
      t.tensor_base = t0;
      int c = 0;
      for (int i = 0; i < t.count; ++i) {
        t.p[i] = param[t0 + i]; t.g[i] = grad[t0 + i]; t.m[i] = exp_avg[t0 + i]; t.v[i] = exp_avg_sq[t0 + i];
        t.step[i] = step[t0 + i]; t.n[i] = numel[t0 + i];
        t.chunk0[i] = c;
        c += (int)((numel[t0 + i] + ADAM_CHUNK - 1) / ADAM_CHUNK);
      }
      t.chunk0[t.count] = c;
      if (pass == 0) hipLaunchKernelGGL(k_adam_sqsum, dim3((unsigned)c), dim3(256), 0, st, t, h);
      else hipLaunchKernelGGL(k_adam_update, dim3((unsigned)c), dim3(256), 0, st, t, h);
      MIRL_LAUNCH_CHECK();
      chunk_base += c;
    }
  }
  return MIRL_OK;
}
