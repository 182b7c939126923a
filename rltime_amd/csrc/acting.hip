// acting.hip — per-vector-step bookkeeping of the device-resident actor.
//
// k_episode_track: the episode statistics the reference keeps in
// PolicyTrainer._track_rewards (rltime/training/policy_trainer.py:93-131: running
// reward / length per env on the RAW rewards, reported when `done`), plus the action
// histogram of _format_action_hist (:75-91), as ONE launch per vector step instead of
// a handful of tiny tensor ops.  Finished episodes are written to a row of a ring
// (reward, length; length 0 = no episode ended for that env this step) that the host
// reads back asynchronously — the acting loop never synchronises.
#include "common.hpp"

namespace mirl {

__global__ void __launch_bounds__(256)
k_episode_track(int E, int A, const float* __restrict__ rewards, const uint8_t* __restrict__ dones,
                const int32_t* __restrict__ actions, float* __restrict__ ep_reward, int32_t* __restrict__ ep_len,
                float* __restrict__ out_reward, int32_t* __restrict__ out_len, int32_t* __restrict__ action_counts) {
  int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  float r = ep_reward[e] + rewards[e];
  int n = ep_len[e] + 1;
  if (dones[e]) { out_reward[e] = r; out_len[e] = n; r = 0.f; n = 0; }
  else { out_reward[e] = 0.f; out_len[e] = 0; }
  ep_reward[e] = r; ep_len[e] = n;
  if (action_counts && actions) { int a = actions[e]; if (a >= 0 && a < A) atomicAdd(action_counts + a, 1); }
}

// The actor's head after the network: dueling combine (policies/torch/dqn.py:74-87:
// V + A - mean_a A), mean over the IQN quantile samples (policies/torch/iqn.py:
// _actor_predict_postprocess), greedy action (dqn.py:140-141 argmax) and the
// epsilon-greedy remap (exploration/epsilon_greedy.py:74-100 with the Ape-X
// per-actor exponent) — ~11 tiny PyTorch launches per vector step in one.
// One wave per env: lane n holds quantile row n (strided loop when N > 64), the
// per-action sums over n are wave shuffle reductions.
__global__ void __launch_bounds__(256)
k_actor_head(int E, int N, int A, const float* __restrict__ adv, const float* __restrict__ val, int Q,
             const double* __restrict__ eps, const double* __restrict__ expo, double eps_min,
             const float* __restrict__ u, const int64_t* __restrict__ rnd,
             int32_t* __restrict__ actions, float* __restrict__ qvalues, float* __restrict__ eps_used) {
  const int lane = threadIdx.x & 63;
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= E) return;
  float best = 0.f; int arg = 0;
  for (int a0 = 0; a0 < A; a0 += 8) {                 // 8 actions per sweep: bounded registers for any A
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    for (int n = lane; n < N; n += 64) {
      const float* row = adv + ((int64_t)e * N + n) * A;
      float off = 0.f;
      if (val) {                                        // dueling: V + A - mean_a A
        float m = 0.f;
        for (int a = 0; a < A; ++a) m = m + row[a];
        off = val[((int64_t)e * N + n) * Q] - m / (float)A;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) if (a0 + k < A) acc[k] = acc[k] + (row[a0 + k] + off);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float s = acc[k];
      for (int o = 32; o > 0; o >>= 1) s = s + __shfl_xor(s, o);
      if (a0 + k < A) {
        const float q = s / (float)N;
        if (lane == 0) qvalues[(int64_t)e * A + a0 + k] = q;
        if ((a0 + k == 0) || q > best) { best = q; arg = a0 + k; }     // first maximum, like argmax
      }
    }
  }
  if (lane == 0) {
    int act = arg;
    if (eps) {                                          // epsilon_greedy.py:74-100
      double pe = pow(*eps, expo ? expo[e] : 1.0);
      float per = (float)(pe > eps_min ? pe : eps_min);
      if (u[e] < per) act = (int)rnd[e];
      if (eps_used) eps_used[e] = per;
    }
    actions[e] = act;
  }
}

}  // namespace mirl

extern "C" int mirl_actor_head(int32_t E, int32_t N, int32_t A, const float* adv, const float* val, int32_t Q,
                               const double* eps, const double* expo, double eps_min, const float* u, const int64_t* rnd,
                               int32_t* actions, float* qvalues, float* eps_used, void* stream) {
  if (E <= 0 || N <= 0 || A <= 0 || !adv || !actions || !qvalues || (eps && (!u || !rnd)) || (val && Q <= 0))
    return mirl::fail(MIRL_ERR_ARG, "bad actor_head arguments");
  mirl::ProfScope ps("k_actor_head", 0.0, (hipStream_t)stream);
  hipLaunchKernelGGL(mirl::k_actor_head, dim3((E + 3) / 4), dim3(256), 0, (hipStream_t)stream, (int)E, (int)N, (int)A, adv, val, (int)Q,
                     eps, expo, eps_min, u, rnd, actions, qvalues, eps_used);
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}

extern "C" int mirl_episode_track(int32_t E, int32_t A, const float* rewards, const uint8_t* dones, const int32_t* actions,
                                  float* ep_reward, int32_t* ep_len, float* out_reward, int32_t* out_len,
                                  int32_t* action_counts, void* stream) {
  if (E <= 0 || !rewards || !dones || !ep_reward || !ep_len || !out_reward || !out_len) return mirl::fail(MIRL_ERR_ARG, "bad episode_track arguments");
  mirl::ProfScope ps("k_episode_track", 0.0, (hipStream_t)stream);
  hipLaunchKernelGGL(mirl::k_episode_track, dim3((E + 255) / 256), dim3(256), 0, (hipStream_t)stream, (int)E, (int)A, rewards, dones,
                     actions, ep_reward, ep_len, out_reward, out_len, action_counts);
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}
