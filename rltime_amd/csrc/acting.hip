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
#include "philox.hpp"

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
             int32_t* __restrict__ actions, float* __restrict__ qvalues, float* __restrict__ eps_used,
             uint64_t rng_seed, const uint64_t* __restrict__ rng_step, int adv_pitch) {
  const int lane = threadIdx.x & 63;
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= E) return;
  float best = 0.f; int arg = 0;
  for (int a0 = 0; a0 < A; a0 += 8) {                 // 8 actions per sweep: bounded registers for any A
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    for (int n = lane; n < N; n += 64) {
      const float* row = adv + ((int64_t)e * N + n) * adv_pitch;
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
      if (rng_step) {
        // the draws made here instead of two torch launches: one Philox4x32-10 block per (step, env)
        uint32_t r[4];
        philox_4x32(rng_seed, *rng_step, (uint32_t)e, r);
        const float uf = (float)(r[0] >> 8) * (1.0f / 16777216.0f);          // 24-bit uniform in [0, 1) like torch.rand
        if (uf < per) act = (int)(((uint64_t)r[1] * (uint64_t)A) >> 32);     // unbiased enough for A << 2^32
      } else if (u[e] < per) act = (int)rnd[e];
      if (eps_used) eps_used[e] = per;
    }
    actions[e] = act;
  }
}

// Everything between the environment step and the policy forward of the device-resident
// actor, ONE launch per vector step (reference: Actor.get_samples, acting/actor.py:124-131
// -> make_input_state -> LSTM.get_state, modules/lstm.py:131-161, and the episode
// statistics of PolicyTrainer._track_rewards, policy_trainer.py:93-131):
//   * recurrent carry reset: h_in = h * (1 - done), c_in = c * (1 - done); h_in goes
//     straight into the tail of the LSTM GEMM's input rows [features | h_in] (row pitch
//     xh_pitch floats), c_in into the cell kernel's input;
//   * the transition's stored recurrent state [h_in | c_in] (what mirl_replay_ingest takes
//     as `state`) and `initials` = done;
//   * reward clipping (np.sign, policy_trainer.py:252-254) AFTER the raw reward went into
//     the episode statistics; dones as uint8;
//   * episode reward / length accumulators and the action histogram (k_episode_track);
//   * the step counter the acting head's Philox draws are keyed with.
struct ActorPreArgs {
  int H, A;
  const int32_t* actions; const float* h; const float* c;
  float* xh_tail; int64_t xh_pitch; float* c_in; float* state_pack; float* initials; float* rewards_out; uint8_t* dones_out;
  int clip;
  float* ep_reward; int32_t* ep_len; float* out_reward; int32_t* out_len; int32_t* action_counts;
  uint64_t* rng_step; uint64_t step;
};

// env e's share of the pre-step, by the 256 threads of one workgroup: (rr, dn) = the env's raw reward / done
__device__ __forceinline__ void actor_pre_env(const ActorPreArgs& p, int e, float rr, int dn) {
  const int H = p.H;
  const float keep = dn ? 0.0f : 1.0f;
  for (int j = threadIdx.x; j < H; j += 256) {
    const float hm = p.h[(int64_t)e * H + j] * keep, cm = p.c[(int64_t)e * H + j] * keep;
    p.xh_tail[(int64_t)e * p.xh_pitch + j] = hm;
    p.c_in[(int64_t)e * H + j] = cm;
    p.state_pack[(int64_t)e * 2 * H + j] = hm;
    p.state_pack[(int64_t)e * 2 * H + H + j] = cm;
  }
  if (threadIdx.x == 0) {
    p.initials[e] = dn ? 1.0f : 0.0f;
    p.dones_out[e] = (uint8_t)dn;
    p.rewards_out[e] = p.clip ? (rr > 0.f ? 1.f : (rr < 0.f ? -1.f : 0.f)) : rr;
    if (p.ep_reward) {
      float r = p.ep_reward[e] + rr;
      int n = p.ep_len[e] + 1;
      if (dn) { p.out_reward[e] = r; p.out_len[e] = n; r = 0.f; n = 0; }
      else { p.out_reward[e] = 0.f; p.out_len[e] = 0; }
      p.ep_reward[e] = r; p.ep_len[e] = n;
      if (p.action_counts && p.actions) { const int a = p.actions[e]; if (a >= 0 && a < p.A) atomicAdd(p.action_counts + a, 1); }
    }
    // step == MIRL_STEP_ADVANCE: the counter advances on the device (a captured rollout replays with the same argument)
    if (e == 0 && p.rng_step) *p.rng_step = p.step == ~0ull ? *p.rng_step + 1ull : p.step;
  }
}

__global__ void __launch_bounds__(256)
k_actor_pre(ActorPreArgs p, const float* __restrict__ rewards_raw, const uint8_t* __restrict__ dones) {
  const int e = blockIdx.x;
  actor_pre_env(p, e, rewards_raw[e], dones[e] ? 1 : 0);
}

// The frame-stack wrapper's shift on the device (env_wrappers/common.py:141-178 under an
// auto-resetting vector env): out[e] = [in[e][1..P-1] or zeros when done[e], newest[e]].
__global__ void __launch_bounds__(256)
k_stack_shift(int P, int plane_q, const uint4* __restrict__ in, uint4* __restrict__ out, const uint4* __restrict__ newest,
              const uint8_t* __restrict__ dones) {
  const int e = blockIdx.y;
  const bool reset = dones[e] != 0;
  const int64_t base = (int64_t)e * P * plane_q;
  for (int c = blockIdx.x * 256 + threadIdx.x; c < P * plane_q; c += gridDim.x * 256) {
    const int p = c / plane_q;
    uint4 v;
    if (p == P - 1) v = newest[(int64_t)e * plane_q + (c - p * plane_q)];
    else if (reset) v = uint4{0u, 0u, 0u, 0u};
    else v = in[base + c + plane_q];
    out[base + c] = v;
  }
}

// One step of the synthetic Atari-shaped vector env (acting/synthetic_env.py) with nothing decided on the host:
// step number t = clock[slot] + 1 (the env's device step counter: a pair of words read / written alternately), observation
// = frame batch t % pool_n of a pre-generated pool copied into the env's static output block, reward in {-1, 0, 1} with
// cumulative probabilities (p_neg, p_nonpos) and done with probability p_done from ONE Philox4x32-10 block per
// (seed, t, env).  Capturable: every argument is the same for every replay.
// PRE: the actor's pre-step (k_actor_pre) for env e rides in the workgroup that draws the env's reward / done — every one of
// its threads makes the same Philox draw (no exchange), so env.step and the pre-step are ONE launch.
template <bool PRE>
__global__ void __launch_bounds__(256)
k_synth_env_step(int row_q, const uint4* __restrict__ pool, int pool_n, int64_t pool_stride_q, const uint64_t* __restrict__ clock_in,
                 uint64_t* __restrict__ clock_out, uint64_t seed, float p_neg, float p_nonpos, float p_done,
                 uint4* __restrict__ obs, float* __restrict__ rewards, uint8_t* __restrict__ dones, ActorPreArgs pre) {
  const int e = blockIdx.y;
  const uint64_t t = *clock_in + 1ull;                 // every workgroup reads the word nobody writes in this launch
  const uint4* src = pool + (int64_t)(t % (uint64_t)pool_n) * pool_stride_q + (int64_t)e * row_q;
  uint4* dst = obs + (int64_t)e * row_q;
  for (int c = blockIdx.x * 256 + threadIdx.x; c < row_q; c += gridDim.x * 256) dst[c] = src[c];
  if (blockIdx.x == 0 && (PRE || threadIdx.x == 0)) {
    uint32_t r[4];
    philox_4x32(seed ^ 0xE17ull, t, (uint32_t)e, r);
    const float u0 = (float)(r[0] >> 8) * (1.0f / 16777216.0f), u1 = (float)(r[1] >> 8) * (1.0f / 16777216.0f);
    const float rr = u0 < p_neg ? -1.0f : (u0 < p_nonpos ? 0.0f : 1.0f);
    const int dn = u1 < p_done ? 1 : 0;
    if (threadIdx.x == 0) {
      rewards[e] = rr;
      dones[e] = (uint8_t)dn;
      if (e == 0) *clock_out = t;                      // the OTHER word of the pair: the next launch reads it
    }
    if (PRE) actor_pre_env(pre, e, rr, dn);
  }
}

}  // namespace mirl

extern "C" int mirl_synth_env_step(int32_t E, int64_t frame_bytes, const uint8_t* pool, int32_t pool_n, uint64_t* clock, int32_t slot,
                                   uint64_t seed, float p_neg, float p_nonpos, float p_done, uint8_t* obs, float* rewards,
                                   uint8_t* dones, void* stream) {
  if (E <= 0 || frame_bytes <= 0 || (frame_bytes % 16) || pool_n <= 0 || !pool || !clock || !obs || !rewards || !dones ||
      (slot != 0 && slot != 1) || ((uintptr_t)pool % 16) || ((uintptr_t)obs % 16) || ((uintptr_t)clock % 16))
    return mirl::fail(MIRL_ERR_ARG, "bad synth_env_step arguments (16-byte aligned frame rows, slot 0 | 1)");
  const int row_q = (int)(frame_bytes / 16);
  int gx = (row_q + 255) / 256; if (gx > 8) gx = 8;
  mirl::ProfScope ps("k_synth_env_step", 2.0 * (double)E * (double)frame_bytes, (hipStream_t)stream);
  // the step counter is a PAIR of words: this launch reads clock[slot] and writes clock[slot ^ 1] — no workgroup can see
  // the new value, no atomics (a shared arrival counter cost 18 of the kernel's 25 us at 256 envs)
  hipLaunchKernelGGL(mirl::k_synth_env_step<false>, dim3(gx, E), dim3(256), 0, (hipStream_t)stream, row_q, (const uint4*)pool, (int)pool_n,
                     (int64_t)E * row_q, (const uint64_t*)(clock + slot), clock + (slot ^ 1), seed, p_neg, p_nonpos, p_done,
                     (uint4*)obs, rewards, dones, mirl::ActorPreArgs{});
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}

static int fill_pre(mirl::ActorPreArgs& p, int32_t E, int32_t H, int32_t A, const int32_t* actions, const float* h, const float* c,
                    float* xh_tail, int64_t xh_pitch, float* c_in, float* state_pack, float* initials, float* rewards_out,
                    uint8_t* dones_out, int32_t clip_rewards, float* ep_reward, int32_t* ep_len, float* out_reward, int32_t* out_len,
                    int32_t* action_counts, uint64_t* rng_step, uint64_t step) {
  // H == 0: a policy without a recurrent layer (nothing to mask or store; the state pointers may be NULL)
  if (E <= 0 || H < 0 || (H > 0 && (!h || !c || !xh_tail || !c_in || !state_pack)) || !initials || !rewards_out ||
      !dones_out || (ep_reward && (!ep_len || !out_reward || !out_len)))
    return mirl::fail(MIRL_ERR_ARG, "bad actor_pre arguments");
  p.H = H; p.A = A; p.actions = actions; p.h = h; p.c = c; p.xh_tail = xh_tail; p.xh_pitch = xh_pitch; p.c_in = c_in;
  p.state_pack = state_pack; p.initials = initials; p.rewards_out = rewards_out; p.dones_out = dones_out; p.clip = clip_rewards;
  p.ep_reward = ep_reward; p.ep_len = ep_len; p.out_reward = out_reward; p.out_len = out_len; p.action_counts = action_counts;
  p.rng_step = rng_step; p.step = step;
  return MIRL_OK;
}

extern "C" int mirl_synth_env_step_pre(int32_t E, int64_t frame_bytes, const uint8_t* pool, int32_t pool_n, uint64_t* clock, int32_t slot,
                                       uint64_t seed, float p_neg, float p_nonpos, float p_done, uint8_t* obs, float* rewards,
                                       uint8_t* dones, int32_t H, int32_t A, const int32_t* actions, const float* h, const float* c,
                                       float* xh_tail, int64_t xh_pitch, float* c_in, float* state_pack, float* initials,
                                       float* rewards_out, uint8_t* dones_out, int32_t clip_rewards, float* ep_reward, int32_t* ep_len,
                                       float* out_reward, int32_t* out_len, int32_t* action_counts, uint64_t* rng_step, uint64_t step,
                                       void* stream) {
  if (E <= 0 || frame_bytes <= 0 || (frame_bytes % 16) || pool_n <= 0 || !pool || !clock || !obs || !rewards || !dones ||
      (slot != 0 && slot != 1) || ((uintptr_t)pool % 16) || ((uintptr_t)obs % 16) || ((uintptr_t)clock % 16))
    return mirl::fail(MIRL_ERR_ARG, "bad synth_env_step arguments (16-byte aligned frame rows, slot 0 | 1)");
  mirl::ActorPreArgs p;
  int rc = fill_pre(p, E, H, A, actions, h, c, xh_tail, xh_pitch, c_in, state_pack, initials, rewards_out, dones_out, clip_rewards,
                    ep_reward, ep_len, out_reward, out_len, action_counts, rng_step, step);
  if (rc) return rc;
  const int row_q = (int)(frame_bytes / 16);
  int gx = (row_q + 255) / 256; if (gx > 8) gx = 8;
  mirl::ProfScope ps("k_synth_env_step_pre", 2.0 * (double)E * (double)frame_bytes + (double)E * H * 4.0 * 6.0, (hipStream_t)stream);
  hipLaunchKernelGGL(mirl::k_synth_env_step<true>, dim3(gx, E), dim3(256), 0, (hipStream_t)stream, row_q, (const uint4*)pool, (int)pool_n,
                     (int64_t)E * row_q, (const uint64_t*)(clock + slot), clock + (slot ^ 1), seed, p_neg, p_nonpos, p_done,
                     (uint4*)obs, rewards, dones, p);
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}

extern "C" int mirl_stack_shift(int32_t E, int32_t P, int32_t plane_bytes, const uint8_t* in, uint8_t* out, const uint8_t* newest,
                                const uint8_t* dones, void* stream) {
  if (E <= 0 || P <= 1 || plane_bytes <= 0 || (plane_bytes % 16) || !in || !out || !newest || !dones || in == out ||
      ((uintptr_t)in % 16) || ((uintptr_t)out % 16) || ((uintptr_t)newest % 16))
    return mirl::fail(MIRL_ERR_ARG, "bad stack_shift arguments (16-byte aligned planes, out != in)");
  const int pq = plane_bytes / 16;
  int gx = (P * pq + 255) / 256; if (gx > 8) gx = 8;
  hipLaunchKernelGGL(mirl::k_stack_shift, dim3(gx, E), dim3(256), 0, (hipStream_t)stream, (int)P, pq, (const uint4*)in, (uint4*)out,
                     (const uint4*)newest, dones);
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}

extern "C" int mirl_actor_pre(int32_t E, int32_t H, int32_t A, const float* rewards_raw, const uint8_t* dones,
                              const int32_t* actions, const float* h, const float* c, float* xh_tail, int64_t xh_pitch,
                              float* c_in, float* state_pack, float* initials, float* rewards_out, uint8_t* dones_out,
                              int32_t clip_rewards, float* ep_reward, int32_t* ep_len, float* out_reward, int32_t* out_len,
                              int32_t* action_counts, uint64_t* rng_step, uint64_t step, void* stream) {
  if (!rewards_raw || !dones) return mirl::fail(MIRL_ERR_ARG, "bad actor_pre arguments");
  mirl::ActorPreArgs p;
  int rc = fill_pre(p, E, H, A, actions, h, c, xh_tail, xh_pitch, c_in, state_pack, initials, rewards_out, dones_out, clip_rewards,
                    ep_reward, ep_len, out_reward, out_len, action_counts, rng_step, step);
  if (rc) return rc;
  mirl::ProfScope ps("k_actor_pre", (double)E * H * 4.0 * 6.0, (hipStream_t)stream);
  hipLaunchKernelGGL(mirl::k_actor_pre, dim3(E), dim3(256), 0, (hipStream_t)stream, p, rewards_raw, dones);
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}

extern "C" int mirl_actor_head_rng(int32_t E, int32_t N, int32_t A, const float* adv, int32_t adv_pitch, const float* val, int32_t Q,
                                   const double* eps, const double* expo, double eps_min, uint64_t rng_seed,
                                   const uint64_t* rng_step, int32_t* actions, float* qvalues, float* eps_used, void* stream) {
  if (E <= 0 || N <= 0 || A <= 0 || adv_pitch < A || !adv || !actions || !qvalues || (eps && !rng_step) || (val && Q <= 0))
    return mirl::fail(MIRL_ERR_ARG, "bad actor_head_rng arguments");
  mirl::ProfScope ps("k_actor_head", 0.0, (hipStream_t)stream);
  hipLaunchKernelGGL(mirl::k_actor_head, dim3((E + 3) / 4), dim3(256), 0, (hipStream_t)stream, (int)E, (int)N, (int)A, adv, val, (int)Q,
                     eps, expo, eps_min, (const float*)nullptr, (const int64_t*)nullptr, actions, qvalues, eps_used, rng_seed, rng_step, (int)adv_pitch);
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}

extern "C" int mirl_actor_head(int32_t E, int32_t N, int32_t A, const float* adv, const float* val, int32_t Q,
                               const double* eps, const double* expo, double eps_min, const float* u, const int64_t* rnd,
                               int32_t* actions, float* qvalues, float* eps_used, void* stream) {
  if (E <= 0 || N <= 0 || A <= 0 || !adv || !actions || !qvalues || (eps && (!u || !rnd)) || (val && Q <= 0))
    return mirl::fail(MIRL_ERR_ARG, "bad actor_head arguments");
  mirl::ProfScope ps("k_actor_head", 0.0, (hipStream_t)stream);
  hipLaunchKernelGGL(mirl::k_actor_head, dim3((E + 3) / 4), dim3(256), 0, (hipStream_t)stream, (int)E, (int)N, (int)A, adv, val, (int)Q,
                     eps, expo, eps_min, u, rnd, actions, qvalues, eps_used, (uint64_t)0, (const uint64_t*)nullptr, (int)A);
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
