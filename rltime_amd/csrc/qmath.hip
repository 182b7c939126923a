// qmath.hip — fused Q-learning target and loss kernels for gfx950.
//
// All four kernels are single-pass and HBM-bound: each network output element is
// read once, nothing of size (M, N', N) is ever materialised (the reference's
// IQN loss builds ~6 such temporaries, training/torch/iqn.py:85-104).
//
//   k_target_dqn   dqn.py:52-71 (double-Q select) + torch_trainer.py:124-147
//   k_target_iqn   iqn.py:36-52 (argmax of the quantile mean) + the same tail
//   k_loss_dqn     dqn.py:141-161 forward + analytic backward
//   k_loss_iqn     iqn.py:77-120 pairwise quantile-Huber forward + backward
//
// Floating point, fp32 like the reference (the h^-1 of value rescaling in fp64
// as torch_trainer.py:59-61 does).  Parity bar: 1e-4 against oracle/qmath.py.
#include "common.hpp"
#include "vfscale.hpp"

namespace mirl {

__global__ void __launch_bounds__(256)
k_target_dqn(int64_t M, int A, const float* __restrict__ qt, const float* __restrict__ qs,
             const float* __restrict__ returns, const float* __restrict__ nsteps, const float* __restrict__ masks,
             float gamma, double vf_eps, float* __restrict__ out) {
  int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  const float* s = qs + m * A;
  int best = 0; float bv = s[0];
  for (int a = 1; a < A; ++a) { float v = s[a]; if (v > bv) { bv = v; best = a; } }   // first maximum, like torch.argmax
  float v = qt[m * A + best];
  out[m] = finish_target(v, returns[m], powf(gamma, nsteps[m]), masks[m], vf_eps);
}

// One wavefront per transition, no block-level barrier.
// Phase 1: the (Ns x A) selection block is read with fully coalesced loads
// (element e = lane + 64k, all loads issued back to back) into the wave's own
// LDS strip.  Phase 2: lane a < A sums its action column from LDS in quantile
// order (deterministic), mean, then a first-maximum argmax over A.
// Phase 3: lanes < Nt pick the target quantile of that action and finish.
__global__ void __launch_bounds__(256)
k_target_iqn(int64_t M, int Nt, int Ns, int A, const float* __restrict__ zt, const float* __restrict__ zs,
             const float* __restrict__ returns, const float* __restrict__ nsteps, const float* __restrict__ masks,
             float gamma, double vf_eps, float* __restrict__ out) {
  extern __shared__ float lds[];                 // [4 waves][Ns*A + A]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t m = (int64_t)blockIdx.x * 4 + wave;
  if (m >= M) return;                            // whole wave exits together; no __syncthreads below
  const int NA = Ns * A;
  float* strip = lds + wave * (NA + A);
  float* acc = strip + NA;
  const float* s = zs + m * (int64_t)NA;
  for (int e = lane; e < NA; e += 64) strip[e] = s[e];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  for (int a = lane; a < A; a += 64) {
    float sum = 0.f;
    for (int n = 0; n < Ns; ++n) sum += strip[n * A + a];
    acc[a] = sum / (float)Ns;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  int best = 0; float bv = acc[0];
  for (int a = 1; a < A; ++a) { float v = acc[a]; if (v > bv) { bv = v; best = a; } }
  const float ret = returns[m], mask = masks[m], disc = powf(gamma, nsteps[m]);
  const float* t = zt + m * (int64_t)Nt * A;
  for (int i = lane; i < Nt; i += 64) out[m * Nt + i] = finish_target(t[i * A + best], ret, disc, mask, vf_eps);
}

// dqn.py:105-111 and its derivative
__device__ __forceinline__ void huber_pair(float e, float kappa, float& val, float& grad) {
  float a = fabsf(e);
  if (a <= kappa) { val = 0.5f * e * e; grad = e; }
  else { val = kappa * (a - 0.5f * kappa); grad = e > 0.f ? kappa : -kappa; }
}

__global__ void __launch_bounds__(256)
k_loss_dqn(int64_t M, int A, const float* __restrict__ q, const int64_t* __restrict__ actions,
           const float* __restrict__ targets, const float* __restrict__ weights, float kappa, int mode,
           float row_scale, float* __restrict__ row_loss, float* __restrict__ dq, float* __restrict__ td_out) {
  int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  int a = (int)actions[m];
  float td = q[m * A + a] - targets[m];               // dqn.py:151
  float val, grad;
  if (mode == 1) { val = td * td; grad = 2.f * td; } else huber_pair(td, kappa, val, grad);
  float w = weights ? weights[m] : 1.f;
  row_loss[m] = val * w;
  td_out[m] = td;                                      // dqn.py:167 signed td report
  float g = grad * w * row_scale;
  for (int k = 0; k < A; ++k) dq[m * A + k] = (k == a) ? g : 0.f;
}

// One wavefront per transition, register/shuffle only (no LDS, no barrier) when
// N and Nt fit one wavefront (the shipped configs: N = Nt = 32); otherwise the
// generic LDS kernel below.  Lane l owns online quantile j = l % N and walks the
// targets of part p = l / N (64/N parts split the Nt targets); targets are
// broadcast with a wave shuffle.  Per pair (i, j):
//   td = y_i - theta_j ; rho = |tau_j - 1{td<0}| * huber(td) / kappa
//   row = mean_i sum_j rho ; report = mean_ij |td| ; d row / d theta_j = -(1/Nt) sum_i |..| huber'(td)/kappa
__global__ void __launch_bounds__(256)
k_loss_iqn_wave(int64_t M, int N, int Nt, int A, const float* __restrict__ z, const float* __restrict__ taus,
                const int64_t* __restrict__ actions, const float* __restrict__ targets, const float* __restrict__ weights,
                float kappa, float row_scale, float* __restrict__ row_loss, float* __restrict__ dz,
                float* __restrict__ abs_td) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t m = (int64_t)blockIdx.x * 4 + wave;
  if (m >= M) return;
  const int parts = 64 / N;                      // >= 1 because N <= 64
  const int j = lane % N, part = lane / N;
  const bool active = part < parts;
  const int a = (int)actions[m];
  const float w = weights ? weights[m] : 1.f;
  const float yv = lane < Nt ? targets[m * Nt + lane] : 0.f;
  const float theta = z[(m * N + j) * (int64_t)A + a], tau = taus[m * N + j];
  const int per = (Nt + parts - 1) / parts;
  const int i0 = part * per, i1 = (i0 + per < Nt) ? i0 + per : Nt;
  float lj = 0.f, aj = 0.f, gj = 0.f;
  for (int i = 0; i < per; ++i) {
    const int ii = i0 + i;                       // uniform trip count keeps the shuffle convergent
    const float y = __shfl(yv, ii < Nt ? ii : 0);
    if (active && ii < i1) {
      float td = y - theta;                      // iqn.py:85-86
      float val, grad;
      huber_pair(td, kappa, val, grad);
      float pen = fabsf(tau - (td < 0.f ? 1.f : 0.f));   // iqn.py:98-99 (indicator detached)
      lj += pen * val / kappa;                   // iqn.py:100
      gj -= pen * grad / kappa;                  // d td / d theta = -1
      aj += fabsf(td);
    }
  }
  // combine the parts of one quantile (lanes j, j+N, j+2N, ...), then all quantiles
  for (int o = N; o < 64; o <<= 1) gj += __shfl_xor(gj, o);
  float loss_sum = lj, abs_sum = aj;
  for (int o = 32; o > 0; o >>= 1) { loss_sum += __shfl_xor(loss_sum, o); abs_sum += __shfl_xor(abs_sum, o); }
  if (lane == 0) {
    row_loss[m] = (loss_sum / (float)Nt) * w;    // iqn.py:102-104, then importance weight (:118)
    abs_td[m] = abs_sum / ((float)Nt * (float)N);  // iqn.py:112
  }
  const float gq = gj * (w * row_scale / (float)Nt);
  // dense gradient row: coalesced (N*A contiguous floats), zero off the acted action
  float* g = dz + m * (int64_t)N * A;
  const int NA = N * A;
  for (int e0 = 0; e0 < NA; e0 += 64) {
    const int e = e0 + lane;
    const int jj = (e < NA ? e : 0) / A;
    const float gv = __shfl(gq, jj);             // lane jj (part 0) holds quantile jj's gradient
    if (e < NA) g[e] = (e - jj * A == a) ? gv : 0.f;
  }
}

// Generic shapes (N or Nt > 64): lane j strides over quantiles, targets and the
// per-quantile gradients go through the wave's LDS strip.
__global__ void __launch_bounds__(256)
k_loss_iqn(int64_t M, int N, int Nt, int A, const float* __restrict__ z, const float* __restrict__ taus,
           const int64_t* __restrict__ actions, const float* __restrict__ targets, const float* __restrict__ weights,
           float kappa, float row_scale, float* __restrict__ row_loss, float* __restrict__ dz,
           float* __restrict__ abs_td) {
  extern __shared__ float lds[];                 // [4 waves][Nt targets + N grads]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t m = (int64_t)blockIdx.x * 4 + wave;
  float* y = lds + wave * (Nt + N);
  float* gsh = y + Nt;
  const bool live = m < M;
  if (live) for (int i = lane; i < Nt; i += 64) y[i] = targets[m * Nt + i];
  __syncthreads();
  const int a = live ? (int)actions[m] : 0;
  const float w = (live && weights) ? weights[m] : 1.f;
  const float* zr = z + (live ? m : 0) * (int64_t)N * A;
  float loss_sum = 0.f, abs_sum = 0.f;
  if (live) {
    for (int j = lane; j < N; j += 64) {
      const float theta = zr[j * A + a], tau = taus[m * N + j];
      float lj = 0.f, aj = 0.f, gj = 0.f;
      for (int i = 0; i < Nt; ++i) {
        float td = y[i] - theta;
        float val, grad;
        huber_pair(td, kappa, val, grad);
        float pen = fabsf(tau - (td < 0.f ? 1.f : 0.f));
        lj += pen * val / kappa;
        gj -= pen * grad / kappa;
        aj += fabsf(td);
      }
      loss_sum += lj; abs_sum += aj;
      gsh[j] = gj * (w * row_scale / (float)Nt);
    }
  }
  for (int o = 32; o > 0; o >>= 1) { loss_sum += __shfl_xor(loss_sum, o); abs_sum += __shfl_xor(abs_sum, o); }
  __syncthreads();                                   // gsh[] written by lane j, read by every lane below
  if (!live) return;
  if (lane == 0) {
    row_loss[m] = (loss_sum / (float)Nt) * w;
    abs_td[m] = abs_sum / ((float)Nt * (float)N);
  }
  float* g = dz + m * (int64_t)N * A;
  for (int e = lane; e < N * A; e += 64) { int j = e / A, k = e - j * A; g[e] = (k == a) ? gsh[j] : 0.f; }
}

}  // namespace mirl

using namespace mirl;

extern "C" int mirl_q_target_dqn(int64_t M, int32_t A, const float* q_target, const float* q_select, const float* returns,
                                 const float* nsteps, const float* masks, double gamma, double vf_eps, float* targets, void* stream) {
  if (M <= 0 || A <= 0 || !q_target || !q_select || !returns || !nsteps || !masks || !targets) return fail(MIRL_ERR_ARG, "bad q_target_dqn arguments");
  ProfScope ps("k_target_dqn", (double)M * (2.0 * A * 4 + 16), (hipStream_t)stream);
  hipLaunchKernelGGL(k_target_dqn, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, M, (int)A, q_target, q_select,
                     returns, nsteps, masks, (float)gamma, vf_eps, targets);
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}

extern "C" int mirl_q_target_iqn(int64_t M, int32_t Nt, int32_t Ns, int32_t A, const float* z_target, const float* z_select,
                                 const float* returns, const float* nsteps, const float* masks, double gamma, double vf_eps,
                                 float* targets, void* stream) {
  if (M <= 0 || A <= 0 || Nt <= 0 || Ns <= 0 || !z_target || !z_select || !returns || !nsteps || !masks || !targets) return fail(MIRL_ERR_ARG, "bad q_target_iqn arguments");
  size_t lds = sizeof(float) * 4 * ((size_t)Ns * A + A);
  if (lds > 64 * 1024) return fail(MIRL_ERR_ARG, "q_target_iqn: Ns*A too large for the LDS strip");
  ProfScope ps("k_target_iqn", (double)M * (((double)Nt + Ns) * A * 4 + Nt * 4 + 12), (hipStream_t)stream);
  hipLaunchKernelGGL(k_target_iqn, dim3((unsigned)((M + 3) / 4)), dim3(256), lds, (hipStream_t)stream, M, (int)Nt, (int)Ns, (int)A,
                     z_target, z_select, returns, nsteps, masks, (float)gamma, vf_eps, targets);
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}

extern "C" int mirl_loss_dqn(int64_t M, int32_t A, const float* q, const int64_t* actions, const float* targets, const float* weights,
                             double kappa, int32_t mode, double row_scale, float* row_loss, float* dq, float* td, void* stream) {
  if (M <= 0 || A <= 0 || !q || !actions || !targets || !row_loss || !dq || !td) return fail(MIRL_ERR_ARG, "bad loss_dqn arguments");
  ProfScope ps("k_loss_dqn", (double)M * (2.0 * A * 4 + 8 + 4 + (weights ? 4 : 0) + 8), (hipStream_t)stream);
  hipLaunchKernelGGL(k_loss_dqn, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, M, (int)A, q, actions, targets, weights,
                     (float)kappa, (int)mode, (float)row_scale, row_loss, dq, td);
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}

extern "C" int mirl_loss_iqn(int64_t M, int32_t N, int32_t Nt, int32_t A, const float* z, const float* taus, const int64_t* actions,
                             const float* targets, const float* weights, double kappa, double row_scale, float* row_loss, float* dz,
                             float* abs_td, void* stream) {
  if (M <= 0 || A <= 0 || N <= 0 || Nt <= 0 || !z || !taus || !actions || !targets || !row_loss || !dz || !abs_td) return fail(MIRL_ERR_ARG, "bad loss_iqn arguments");
  ProfScope ps("k_loss_iqn", (double)M * (2.0 * N * A * 4 + N * 4 + Nt * 4 + 8 + (weights ? 4 : 0) + 8), (hipStream_t)stream);
  if (N <= 64 && Nt <= 64 && (64 % N) == 0) {
    hipLaunchKernelGGL(k_loss_iqn_wave, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, M, (int)N, (int)Nt, (int)A, z, taus,
                       actions, targets, weights, (float)kappa, (float)row_scale, row_loss, dz, abs_td);
  } else {
    size_t lds = sizeof(float) * 4 * (size_t)(Nt + N);
    hipLaunchKernelGGL(k_loss_iqn, dim3((unsigned)((M + 3) / 4)), dim3(256), lds, (hipStream_t)stream, M, (int)N, (int)Nt, (int)A, z, taus,
                       actions, targets, weights, (float)kappa, (float)row_scale, row_loss, dz, abs_td);
  }
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}
