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

}  // namespace mirl

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
