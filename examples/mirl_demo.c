/* mirl_demo.c — the C-ABI of librltime_hip used from plain C (no Python, no torch):
 * create a prioritized replay shard, ingest vector steps from device buffers,
 * sample with the on-device RNG, gather a time-major sequence batch, feed losses
 * back, and verify on the host that every gathered frame is the one the replay
 * semantics call for (state of transition o == next_state of o-1).
 *
 *   gcc -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I include examples/mirl_demo.c -L rltime_amd -lrltime_hip -L/opt/rocm/lib -lamdhip64 -o mirl_demo
 */
#include <hip/hip_runtime_api.h>
#include <limits.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mirl.h"

#define CK(x) do { int _r = (x); if (_r < 0) { fprintf(stderr, "%s -> %d: %s\n", #x, _r, mirl_last_error()); return 1; } } while (0)
#define HK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(_e)); return 1; } } while (0)

enum { E = 4, F = 64, T = 6, P = 2, N = 2, B = 8, STEPS = 60 };

int main(void) {
  if (mirl_device_count() <= 0) { fprintf(stderr, "no HIP device\n"); return 2; }
  mirl_replay_config cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.size = 160; cfg.num_envs = E; cfg.frame_bytes = F; cfg.nstep_train = T; cfg.prefix_steps = P;
  cfg.nstep_target = N; cfg.gamma = 0.99; cfg.mode = MIRL_MODE_PER; cfg.train_frequency = 0;
  cfg.overlap = INT_MIN; cfg.alpha = 0.9; cfg.beta = 0.6; cfg.eps = 1e-6; cfg.max_weight_factor = 0.9;
  mirl_replay* h = NULL;
  CK(mirl_replay_create(&cfg, &h));

  uint8_t *d_frames, *d_dones; int32_t* d_actions; float* d_rewards;
  HK(hipMalloc((void**)&d_frames, E * F)); HK(hipMalloc((void**)&d_dones, E));
  HK(hipMalloc((void**)&d_actions, E * 4)); HK(hipMalloc((void**)&d_rewards, E * 4));
  uint8_t hf[E * F], hd[E]; int32_t ha[E]; float hr[E];
  for (int s = 0; s < STEPS; ++s) {
    for (int e = 0; e < E; ++e) {
      memset(hf + e * F, 0, F);
      hf[e * F] = (uint8_t)e; hf[e * F + 1] = (uint8_t)s;     /* identity: (env, per-env offset) */
      hd[e] = 0; ha[e] = s % 3; hr[e] = 1.0f;
    }
    HK(hipMemcpy(d_frames, hf, sizeof(hf), hipMemcpyHostToDevice)); HK(hipMemcpy(d_dones, hd, sizeof(hd), hipMemcpyHostToDevice));
    HK(hipMemcpy(d_actions, ha, sizeof(ha), hipMemcpyHostToDevice)); HK(hipMemcpy(d_rewards, hr, sizeof(hr), hipMemcpyHostToDevice));
    mirl_ingest in;
    memset(&in, 0, sizeof(in));
    in.count = E; in.frames = d_frames; in.actions = d_actions; in.rewards = d_rewards; in.dones = d_dones;
    CK(mirl_replay_ingest(h, &in, NULL));
    HK(hipDeviceSynchronize());                                /* the demo reuses its staging buffers */
  }

  int32_t rows = 0, overlapped = 0;
  CK(mirl_replay_state_rows(h, &rows, &overlapped));
  const int L = T + P;
  int32_t *d_slot, *d_env; int64_t *d_start, *d_lstart, *d_act, *d_lidx; float *d_w, *d_ret, *d_ns, *d_mk, *d_bw; uint8_t* d_out;
  HK(hipMalloc((void**)&d_slot, B * 4)); HK(hipMalloc((void**)&d_env, B * 4)); HK(hipMalloc((void**)&d_start, B * 8));
  HK(hipMalloc((void**)&d_lstart, B * 8)); HK(hipMalloc((void**)&d_w, B * 4)); HK(hipMalloc((void**)&d_out, (size_t)rows * B * F));
  HK(hipMalloc((void**)&d_ret, L * B * 4)); HK(hipMalloc((void**)&d_ns, L * B * 4)); HK(hipMalloc((void**)&d_mk, L * B * 4));
  HK(hipMalloc((void**)&d_act, L * B * 8)); HK(hipMalloc((void**)&d_bw, L * B * 4)); HK(hipMalloc((void**)&d_lidx, L * B * 16));
  int rc = mirl_replay_sample(h, B, 0.5, NULL, 1234, d_slot, d_env, d_start, d_lstart, d_w, NULL, NULL);
  CK(rc);
  if (rc == MIRL_NEED_MORE) { fprintf(stderr, "need more samples\n"); return 1; }
  mirl_batch out;
  memset(&out, 0, sizeof(out));
  out.frames = d_out; out.returns = d_ret; out.nsteps = d_ns; out.masks = d_mk; out.actions = d_act;
  out.weights = d_bw; out.loss_indices = d_lidx;
  CK(mirl_replay_gather(h, B, d_env, d_start, d_lstart, d_w, &out, NULL));
  float* d_loss; HK(hipMalloc((void**)&d_loss, T * B * 4));
  HK(hipMemset(d_loss, 0x3f, T * B * 4));                       /* some positive floats */
  CK(mirl_replay_update_losses(h, (int64_t)T * B, d_lidx + (size_t)P * B * 2, d_loss, NULL));
  HK(hipDeviceSynchronize());

  int32_t env[B]; int64_t start[B]; float ret[L * B];
  uint8_t* got = (uint8_t*)malloc((size_t)rows * B * F);
  HK(hipMemcpy(env, d_env, sizeof(env), hipMemcpyDeviceToHost)); HK(hipMemcpy(start, d_start, sizeof(start), hipMemcpyDeviceToHost));
  HK(hipMemcpy(got, d_out, (size_t)rows * B * F, hipMemcpyDeviceToHost)); HK(hipMemcpy(ret, d_ret, sizeof(ret), hipMemcpyDeviceToHost));
  int bad = 0;
  for (int r = 0; r < rows; ++r)
    for (int b = 0; b < B; ++b) {
      const uint8_t* fr = got + ((size_t)r * B + b) * F;
      int64_t want = start[b] + r - 1;                          /* overlapped layout: rows [0, L+n) */
      if (fr[0] != (uint8_t)env[b] || fr[1] != (uint8_t)want) ++bad;
    }
  for (int i = 0; i < L * B; ++i) if (ret[i] != 1.0f + 0.99f) ++bad;   /* 2-step return of reward 1 (f64 sum -> f32) */
  int64_t total = 0, active = 0;
  CK(mirl_replay_stats(h, &total, &active, NULL, NULL, NULL));
  printf("rows=%d overlapped=%d total=%lld active=%lld mismatches=%d\n", rows, overlapped, (long long)total, (long long)active, bad);
  CK(mirl_replay_destroy(h));
  free(got);
  puts(bad ? "FAIL" : "OK");
  return bad ? 1 : 0;
}
