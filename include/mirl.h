/* mirl.h — C-ABI of librltime_hip.so: the MI355X (gfx950) backend for rltime's
 * Q-learning hot path.
 *
 * The reference (opherlieber/rltime) is pure Python and has no FFI; the
 * boundary this library sits behind is the reference's Python plugin API
 * (SURVEY.md section 8b).  Each entry point below names the reference
 * interface it replaces (paths relative to the reference root).  The Python
 * mirror of those classes lives in rltime_amd/ and binds this header with
 * ctypes (INTEGRATION.md shows the stub a reference maintainer would add).
 *
 * Conventions
 *   - every function returns 0 (MIRL_OK) or a negative error code; the text of
 *     the last error on the calling thread is mirl_last_error().  Nothing
 *     aborts.  MIRL_NEED_MORE (1) is the one non-error positive status: it is
 *     the reference's `get_train_data(...) -> None` ("feed more samples").
 *   - pointers are DEVICE pointers unless the name ends in `_host`.
 *   - all device work is enqueued on the caller's `stream` (a hipStream_t
 *     passed as void*; NULL = the null stream) and nothing synchronises the
 *     host: an ingest / sample / gather / update_losses chain is fully
 *     asynchronous.
 *   - one caller thread per handle (the reference's History is
 *     single-threaded: history.py, cyclic_array.py:8).
 */
#ifndef MIRL_H
#define MIRL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIRL_OK 0
#define MIRL_NEED_MORE 1
#define MIRL_ERR_ARG (-1)
#define MIRL_ERR_HIP (-2)
#define MIRL_ERR_STATE (-3)   /* violates a reference assert (e.g. buffer full but no batch) */
#define MIRL_ERR_NOGPU (-4)

#define MIRL_MODE_UNIFORM 0
#define MIRL_MODE_PER 1

typedef struct mirl_replay mirl_replay;

/* Constructor arguments of
 *   History.__init__                        rltime/history/history.py:17-59
 *   ReplayHistoryBuffer.__init__            rltime/history/replay_history.py:14-60
 *   PrioritizedReplayHistoryBuffer.__init__ rltime/history/prioritized_replay_history.py:41-134
 * plus the shapes of one stored transition (the reference learns those from the
 * first sample dict; a device-resident SoA store needs them up front).        */
typedef struct mirl_replay_config {
  int64_t size;             /* replay_history.py:14  `size` (transitions, all envs) */
  int32_t num_envs;         /* envs owned by this shard, ids env_base .. env_base+num_envs-1 */
  int32_t env_base;
  int32_t frame_bytes;      /* bytes of state["x"] (u8 observation, e.g. 4*84*84)   */
  int32_t extra_f32;        /* floats of the tuple-observation extra features (0 = none) */
  int32_t state_f32;        /* floats of recurrent state per transition (hx|cx per layer) */
  int32_t has_initials;     /* store the per-state `initials` flag (lstm.py:131-161) */
  int32_t policy_f32;       /* floats of policy_output kept besides the action (qvalues), 0 = drop */
  int32_t nstep_train;      /* history.py:25  */
  int32_t prefix_steps;     /* history.py:30  */
  int32_t nstep_target;     /* history.py:22  */
  double  gamma;            /* multi_step_trainer.py:70-74 discount closure */
  int32_t mode;             /* MIRL_MODE_* */
  int32_t train_frequency;  /* replay_history.py:21 (0 = free running) */
  int32_t avoid_episode_crossing; /* replay_history.py:40 */
  int32_t overlap;          /* prioritized_replay_history.py:56; INT32_MIN = default nstep_train/2 */
  double  alpha, beta, eps, max_weight_factor; /* :47-75 */
  int32_t beta_anneal_mode; /* 0 = off, 1 = anneal to 1.0 (True), 2 = anneal to beta_anneal_to */
  double  beta_anneal_to;
  int32_t global_importance_scaling; /* :76 */
  int32_t env_ring_slack;   /* extra ring slots per env beyond ceil(size/num_envs)+1 */
  int32_t device;           /* HIP device ordinal */
  /* Acting-time priority initialisation — NOT in the reference, which lists it as
   * missing (prioritized_replay_history.py:33-36: new samples enter at the constant
   * max loss 1.0).  When set (needs policy_f32 = number of actions), every ingested
   * transition j of an env makes transition j - n's TD error available from STORED
   * data: delta = Q(s_{j-n})[a_{j-n}] - h(R_n + gamma^n h^-1(max_a Q(s_j)) mask), the
   * n-step quantities exactly as History._update_nstep builds them and the target
   * exactly as torch_trainer.py:124-147; it is written through the update_losses
   * path (same |.|+eps, same fan-out to the overlapped sequences) right after the
   * vector step was added.  acting_vf_eps <= 0: no value rescaling.               */
  int32_t acting_priority_init;
  double  acting_vf_eps;
  /* Frame de-duplication in storage (the reference notes stacking support in the
   * buffer as planned but unsupported, history.py:56-59).  stack_planes = P > 1:
   * state["x"] is a stack of P equal planes, newest LAST, produced by the frame-
   * stack wrapper's shift contract (env_wrappers/common.py:141-178: every step rolls
   * the stack by one plane; a reset zero-fills all but the newest).  The ring then
   * keeps ONE plane per transition (the newest) plus the number of real planes in
   * its stack, P x less HBM per transition; the gather rebuilds the stacks while it
   * writes the time-major batch, bit-identical to storing them whole.  Ingest
   * verifies the contract against the stored planes; a frame that violates it makes
   * the next host call fail with MIRL_ERR_STATE.  Needs frame_bytes % (16 P) == 0. */
  int32_t stack_planes;
} mirl_replay_config;

/* One vector step handed to History.update: `count` transitions, transition k
 * belongs to env `env_ids_host[k]` (NULL = env_base+k, i.e. the synchronous
 * actor's env-minor order, actor.py:132-145).  An env may appear at most once
 * per call.  All payload arrays are device arrays with leading dimension
 * `count`; optional ones may be NULL when the config says 0.                  */
typedef struct mirl_ingest {
  int32_t count;
  const int32_t* env_ids_host;
  const uint8_t* frames;    /* [count][frame_bytes]   next_state["x"]            */
  const float*   extra;     /* [count][extra_f32]                                */
  const float*   state;     /* [count][state_f32]     next_state recurrent state */
  const float*   initials;  /* [count]                                           */
  const int32_t* actions;   /* [count]                policy_output["actions"]   */
  const float*   policy;    /* [count][policy_f32]    policy_output["qvalues"]   */
  const float*   rewards;   /* [count]                                           */
  const uint8_t* dones;     /* [count]                                           */
  /* De-duplicated storage (stack_planes = P > 1) only.  newest_plane_only = 1: `frames` holds ONE
   * plane per transition — the newest plane of its stack, frames_stride bytes apart (0 = packed;
   * the actor passes the last plane of its (count, P, h, w) observation block with frames_stride
   * = frame_bytes) — and the number of real planes of a stack follows from the shift contract
   * itself: 1 when this transition ended an episode (the auto-resetting env returned the new
   * episode's first stack with it) or for an unprimed env's first transition, else one more than the
   * predecessor's (capped at P).  No plane is compared against the ring.
   * 0 (default): `frames` holds whole stacks and the contract is VERIFIED against the stored
   * planes (debug / untrusted producers).                                                        */
  int32_t newest_plane_only;
  int64_t frames_stride;
} mirl_ingest;

/* Output of one get_train_data call (history.py:203-286 _make_train_batch),
 * time-major.  R = rows of the stacked state block:
 *   overlapped layout (nstep_target < L, no episode-crossing shift): R = L + n,
 *     states = rows [0, L), target_states = rows [n, L+n)   (history.py:245-265)
 *   separate layout: R = 2L, states = rows [0, L), target_states = rows [L, 2L)
 * with L = prefix_steps + nstep_train.  mirl_replay_state_rows() tells which.
 * dtypes are the ones the trainer sees after make_tensor
 * (models/torch/utils.py:95-123): u8 stays, everything else float32; the two
 * integer fields the trainer re-casts with .long() are emitted as int64.      */
typedef struct mirl_batch {
  uint8_t* frames;        /* [R][B][frame_bytes]                       */
  float*   extra;         /* [R][B][extra_f32]      or NULL            */
  float*   state;         /* [R][B][state_f32]      or NULL            */
  float*   initials;      /* [R][B]                 or NULL            */
  float*   returns;       /* [L][B]   n-step discounted return          */
  float*   nsteps;        /* [L][B]                                     */
  float*   masks;         /* [L][B]   target_masks                      */
  int64_t* actions;       /* [L][B]                                     */
  float*   policy;        /* [L][B][policy_f32]     or NULL            */
  float*   weights;       /* [L][B]   importance weights (PER) or NULL  */
  int64_t* loss_indices;  /* [L][B][2] (env_id, env offset); (-1,-1) on prefix rows; PER or NULL */
} mirl_batch;

const char* mirl_last_error(void);
int mirl_device_count(void);

/* ---- replay handle -------------------------------------------------------- */
int mirl_replay_create(const mirl_replay_config* cfg, mirl_replay** out);
int mirl_replay_destroy(mirl_replay* h);

/* History.update (history.py:123-176) + _sample_added hooks
 * (replay_history.py:77-91, prioritized_replay_history.py:136-172) and the
 * per-env split of Actor.get_samples (actor.py:132-145), as one batched
 * device write per vector step.                                               */
int mirl_replay_ingest(mirl_replay* h, const mirl_ingest* in, void* stream);
/* The same History.update, split so that a WHOLE acting rollout (actor.py:97-149: `iters` vector steps of env
 * step -> policy forward -> History.update) can be captured into one HIP graph: everything History.update decides on
 * the host is data-independent (ring heads, global-FIFO eviction, sequence activation / deactivation, free list), so
 *   mirl_replay_ingest_plan     advances the host bookkeeping by `steps` vector steps of `count` transitions
 *                               (env_ids_host as in mirl_ingest; NULL = env_base .. env_base + count - 1) and copies
 *                               their op lists, stream-ordered, into the shard's rollout-plan buffer (device address
 *                               fixed for the shard's lifetime; sized at the first call for max(steps, 64) steps —
 *                               later calls with more steps return MIRL_ERR_ARG);
 *   mirl_replay_ingest_planned  enqueues the device side of step `step` of the CURRENT plan (frames / state / scalars
 *                               into the rings, table / env / leaf ops, tree fix: one launch).  Its arguments do not
 *                               change from rollout to rollout, so the call can sit in a captured graph; `in->count`
 *                               must equal the plan's, env_ids_host / newest_plane_only are ignored / refused.
 * Shards with de-duplicated frame storage or acting_priority_init return MIRL_ERR_ARG (they need the multi-kernel
 * ingest); callers keep mirl_replay_ingest for those.                                                                */
int mirl_replay_ingest_plan(mirl_replay* h, int32_t steps, int32_t count, const int32_t* env_ids_host, void* stream);
int mirl_replay_ingest_planned(mirl_replay* h, int32_t step, const mirl_ingest* in, void* stream);

/* De-duplicated storage fed in the newest-plane form (mirl_ingest.newest_plane_only): the stack of an
 * env's FIRST transition reaches back to the observation its reset returned, which is not a
 * transition.  This call (once, before the first ingest) stores that observation's newest plane —
 * newest_planes[e] at `stride` bytes per env — as "transition -1" of every env.  Without it the
 * first stack of an env is taken to be a reset stack (one real plane).                              */
int mirl_replay_prime_stack(mirl_replay* h, const uint8_t* newest_planes, int64_t stride, void* stream);

/* mirl_replay_ingest issues ONE fused kernel per call (frame / state / q-value rows, scalars,
 * plan, tree fix) unless the shard de-duplicates frame stacks or initialises priorities at
 * acting time; 0 selects the separate kernels of rounds 1-2 for every shard (A/B tests; the
 * environment variable MIRL_INGEST_FUSED=0 does the same).                                  */
int mirl_ingest_fused_set(int32_t on);

/* needed_feed_count (replay_history.py:62-75).  *out = -1 for None.           */
int mirl_replay_needed_feed_count(mirl_replay* h, int32_t mbatch, int32_t num_envs, int64_t* out);

/* Overwrite the train quota (replay_history.py:60 `train_quota`): used when a
 * pre-filled or restored buffer should start from a steady-state balance
 * instead of owing `size * train_frequency` training.                        */
int mirl_replay_set_train_quota(mirl_replay* h, int64_t quota);

/* The sampling half of get_train_data (replay_history.py:173-184 quota,
 * :93-140 uniform choice, prioritized_replay_history.py:232-241 stratified
 * descent, :286-329 + :347-354 importance weights).
 *   rng_host: uniform mode — int64[mbatch] results of np.random.choice(total, mbatch)
 *             PER mode     — double[mbatch] results of random.random()
 *             NULL         — draw on the device (Philox4x32-10, key = seed, step counter)
 * Outputs (device): slot[mbatch] (PER tree index, or the flat choice),
 * env[mbatch] (LOCAL env index, 0-based), start[mbatch] (absolute per-env offset of
 * the first transition of the window, prefix included), weight[mbatch]
 * (normalised importance weight; 1.0 in uniform mode).
 * Returns MIRL_NEED_MORE for the reference's `None`.                          */
int mirl_replay_sample(mirl_replay* h, int32_t mbatch, double train_progress,
                       const void* rng_host, uint64_t seed,
                       int32_t* slot, int32_t* env, int64_t* start, int64_t* loss_start,
                       float* weight, double* stats, void* stream);
/* `loss_start` (device, mbatch int64, may be NULL): per-env offset of the first
 * TRAINED transition whose loss the sequence's priority tracks.  It equals
 * start + prefix_steps except when avoid_episode_crossing shifted the window:
 * the reference keeps reporting losses against the unshifted sequence
 * (prioritized_replay_history.py:335-338 uses base_offset, :320 shifts only the
 * sampled range).  Pass it on to mirl_replay_gather.                           */
/* `stats` (device, 2 doubles, may be NULL; PER only): [0] = sum of priorities of
 * this shard's tree, [1] = the un-normalised batch-max weight that `weight` was
 * divided by.  With the active-sequence count (mirl_replay_stats) this is what
 * ranks exchange to turn shard-local importance weights into global ones
 * (rltime_amd/parallel.py).                                                   */
/* Exact global sampling over env-sharded replays (no reference counterpart; SURVEY.md
 * section 8(e)2).  mirl_replay_tree_root writes {sum of priorities, active sequences}
 * of this shard to 2 device doubles; the ranks exchange them (rltime_amd/parallel.py)
 * and every rank calls mirl_replay_sample_global with the same table shard_totals
 * [world][2], the same seed and — because every rank makes the same sequence of calls —
 * the same step counter: the mbatch_global strata of the global mass are drawn with
 * one shared Philox stream and each goes to the shard whose cumulative range contains
 * it, exactly as ONE tree over the concatenated shards would sample.  A rank gets a
 * data-dependent number of strata, written in stratum order to the first rows of its
 * `rows`-row outputs; the rest is padding (slot -1, env -1: mirl_replay_gather gives
 * such rows no loss index, the caller gives them weight 0).  weight_raw (double) is
 * (p / P_g * N_g)^-beta, un-normalised; stats (4 device doubles): P_g, the local max
 * raw weight, strata kept, strata dropped because they exceeded `rows`.
 * Quota / NEED_MORE behave like mirl_replay_sample with mbatch_local.             */
/* Host bookkeeping only (no device work, no side effect): *ready = 1 when a sample call
 * for `mbatch` would NOT return MIRL_NEED_MORE (replay_history.py:110-114,
 * prioritized_replay_history.py:295-299).  Multi-rank callers agree on it BEFORE
 * entering any collective; a rank that then skips the call uses mirl_replay_sample_skip,
 * which has exactly the side effects of a NEED_MORE return (quota charged,
 * replay_history.py:176-181; device-RNG step counter advanced).                      */
int mirl_replay_sample_ready(mirl_replay* h, int32_t mbatch, int32_t* ready);
int mirl_replay_sample_skip(mirl_replay* h, int32_t mbatch);
int mirl_replay_tree_root(mirl_replay* h, double* root_dev, void* stream);
int mirl_replay_sample_global(mirl_replay* h, int32_t mbatch_local, int32_t mbatch_global, int32_t rows,
                              int32_t rank, int32_t world, const double* shard_totals,
                              double train_progress, uint64_t seed, int32_t* slot, int32_t* env,
                              int64_t* start, int64_t* loss_start, double* weight_raw,
                              int32_t* stratum, double* stats, void* stream);
/* uniform mode helper for the host RNG path: np.random.choice's `a` argument. */
int mirl_replay_uniform_total(mirl_replay* h, int64_t* total);

/* Rows of the state block for this handle's configuration and whether it is
 * the overlapped layout.                                                      */
int mirl_replay_state_rows(mirl_replay* h, int32_t* rows, int32_t* overlapped);

/* The assembly half of get_train_data: _make_sample_range + _update_nstep
 * (history.py:71-108,178-201), _make_train_batch + StateStore.stack
 * (history.py:203-286, general/backend.py:112-153).                           */
int mirl_replay_gather(mirl_replay* h, int32_t mbatch, const int32_t* env,
                       const int64_t* start, const int64_t* loss_start, const float* weight,
                       const mirl_batch* out, void* stream);

/* update_losses (prioritized_replay_history.py:243-279) + _recalc_weighted_priority
 * (:174-208).  indices = int64[count][2] (env_id, offset) exactly as emitted in
 * mirl_batch.loss_indices; rows with env_id < 0 are ignored.                   */
int mirl_replay_update_losses(mirl_replay* h, int64_t count, const int64_t* indices,
                              const float* losses, void* stream);

/* Snapshot / resume of a whole shard (the reference
 * checkpoints weights only, policy_trainer.py:170-185): host bookkeeping (ring
 * heads, global FIFO, free list, quota), every device array and the priority
 * trees, streamed to / from `path` through a pinned staging buffer.  Restoring
 * needs a handle created with the same mirl_replay_config; afterwards sampling,
 * gathering and priority updates continue bit-identically.  Both synchronise.   */
int mirl_replay_save(mirl_replay* h, const char* path_host);
int mirl_replay_load(mirl_replay* h, const char* path_host);

/* Per-kernel timing of the dominant kernel (the frame gather) with HIP events
 * on the launch stream: enable, run, then read {launches, total ms}.  Reading
 * synchronises the device.                                                    */
int mirl_replay_profile(mirl_replay* h, int32_t enable, int64_t* launches, double* total_ms);

/* Per-kernel timing of EVERY librltime_hip launch (any entry point of this
 * header) with HIP event pairs on the launch stream, accounted per kernel name
 * together with the algorithmic bytes of each launch, for
 * the per-kernel roofline table bench.py prints.  level 0 = off (default),
 * 2 = on.  collect() synchronises the device and folds the finished event pairs
 * into the per-kernel totals; get(i) reads entry i < *n_kernels.                 */
int mirl_profile_set(int32_t level);
int mirl_profile_collect(int32_t* n_kernels);
int mirl_profile_get(int32_t i, char* name_host, int32_t name_cap, int64_t* calls,
                     double* total_ms, double* algorithmic_bytes);
int mirl_profile_reset(void);
/* matrix-pipe work (flop, summed over the entry's launches) the call sites state next to the bytes: 2 M N K of a
 * GEMM / implicit GEMM, the recurrent products of an LSTM sweep; 0 for streaming and latency kernels.          */
int mirl_profile_get_flop(int32_t i, double* flop);

/* ---- introspection / test hooks (host results; these DO synchronise) ------- */
int mirl_replay_stats(mirl_replay* h, int64_t* total_items, int64_t* active_sequences,
                      int64_t* train_quota, int64_t* tree_capacity, int64_t* n_slots);
int mirl_replay_env_meta(mirl_replay* h, int64_t* first_host, int64_t* count_host);
int mirl_replay_free_slots(mirl_replay* h, int32_t* slots_host, int64_t* n);
int mirl_replay_slot_table(mirl_replay* h, int32_t* slot_env_host, int64_t* slot_base_host);
int mirl_replay_tree_nodes(mirl_replay* h, double* value_host, uint8_t* kind_host, double* min_host);
/* overwrite leaves [0, n) (value, kind 0/1/2) and rebuild every inner node     */
int mirl_replay_tree_set_leaves(mirl_replay* h, int64_t n, const double* value_host,
                                const uint8_t* kind_host, void* stream);
/* stratified descent only: idx_host[mbatch] for the given uniforms             */
int mirl_replay_tree_find(mirl_replay* h, int32_t mbatch, const double* uniforms_host,
                          int64_t* idx_host, void* stream);
int mirl_replay_losses_peek(mirl_replay* h, int32_t env_local, int64_t offset, int32_t n, float* out_host);

/* ---- target / loss arithmetic (stateless) ---------------------------------
 * DQN: _get_bootstrap_target_value (training/torch/dqn.py:52-71) + calc_target_values
 * tail (torch_trainer.py:124-147, _vf_scale/_vf_unscale :46-78).
 * q_target, q_select: [M][A]; returns/nsteps/masks: [M]; targets out: [M].
 * vf_eps <= 0 disables value rescaling.                                        */
int mirl_q_target_dqn(int64_t M, int32_t A, const float* q_target, const float* q_select,
                      const float* returns, const float* nsteps, const float* masks,
                      double gamma, double vf_eps, float* targets, void* stream);
/* IQN: training/torch/iqn.py:36-52 + the same tail.  z_target [M][Nt][A],
 * z_select [M][Ns][A] -> targets [M][Nt].                                      */
int mirl_q_target_iqn(int64_t M, int32_t Nt, int32_t Ns, int32_t A,
                      const float* z_target, const float* z_select,
                      const float* returns, const float* nsteps, const float* masks,
                      double gamma, double vf_eps, float* targets, void* stream);

/* DQN._compute_grads loss (training/torch/dqn.py:141-161; _calc_loss :98-114,
 * importance weights :83-96, aggregation :120-130), forward and analytic
 * backward in one pass.  row_scale = d(loss)/d(row loss) of the requested
 * aggregation.  mode: 0 = huber, 1 = mse.  weights may be NULL.
 * Outputs: row_loss[M] (weighted per-row loss; loss = row_scale * sum),
 * dq[M][A] = d loss / d q, td[M] = signed td report.                           */
int mirl_loss_dqn(int64_t M, int32_t A, const float* q, const int64_t* actions,
                  const float* targets, const float* weights, double kappa, int32_t mode,
                  double row_scale, float* row_loss, float* dq, float* td, void* stream);
/* IQN._compute_grads loss (training/torch/iqn.py:77-120).  z [M][N][A],
 * taus [M][N], targets [M][Nt].  Outputs: row_loss[M], dz[M][N][A],
 * abs_td[M] (mean |td| report, iqn.py:112).                                    */
int mirl_loss_iqn(int64_t M, int32_t N, int32_t Nt, int32_t A, const float* z,
                  const float* taus, const int64_t* actions, const float* targets,
                  const float* weights, double kappa, double row_scale,
                  float* row_loss, float* dz, float* abs_td, void* stream);

/* ---- recurrent core: fused LSTM-cell pointwise step --------------------------
 * Replaces the per-step elementwise chain of torch.nn.LSTMCell inside
 * rltime/models/torch/modules/lstm.py:83-116 (time loop with state reset on
 * `initials`).  gates [B][4H] (i,f,g,o): pre-activations in, activated gates
 * out; c_in [B][H] is the (already masked) cell input; h_next/c_next receive
 * h*keep_next / c*keep_next (keep_next [B] = 1 - initials of step t+1, NULL = 1)
 * i.e. the masked inputs of the following step.  h_out / c_out may be NULL.     */
int mirl_lstm_cell_fwd(int32_t B, int32_t H, float* gates, const float* c_in, const float* keep_next,
                       float* h_out, float* c_out, float* h_next, float* c_next, void* stream);
/* A whole forward sweep of the recurrent layer (all T steps of all B sequences) in ONE
 * persistent launch (csrc/lstm_seq.hip; replaces the T-step loop of
 * rltime/models/torch/modules/lstm.py:83-116): W_hh stays resident in LDS, the cell runs
 * in the MFMA accumulators' lanes, workgroups exchange h(t) through write-through stores
 * and per-row-tile arrival counters (no grid barrier).  gx [T][B][4H] holds the input
 * projection x W_ih^T + b_ih + b_hh on entry; with save_gates it holds the activated
 * gates (i, f, g, o) on exit, as mirl_lstm_cell_fwd leaves them.  h0 / c0 [B][H] are the
 * states before step 0 (the reset mask keep[0] is applied inside), keep [T][B] = 1 -
 * initials.  Optional outputs: out [T][B][H] = h(t); c_all [T][B][H] = c(t); hm / cm
 * [T+1][B][H] = the masked inputs of every step (row T = final state); h_last / c_last
 * [B][H] = that final state alone.  At least one of (hm, cm) / (h_last, c_last).
 * workspace: mirl_lstm_seq_workspace_bytes(B, H) bytes, 256-byte aligned, private to the
 * call until the stream has passed it.  Shapes: mirl_lstm_seq_supported (B multiple of
 * 16, H in {128, 256, 512}).  A workgroup that waits for a peer longer than ~seconds
 * gives up; the next call then returns MIRL_ERR_STATE (mirl_lstm_seq_status reads the
 * flag without clearing it).                                                           */
int mirl_lstm_seq_supported(int32_t T, int32_t B, int32_t H);
int mirl_lstm_seq_workspace_bytes(int32_t B, int32_t H, int64_t* bytes);
int mirl_lstm_seq_fwd(int32_t T, int32_t B, int32_t H, float* gx, const float* w_hh, const float* h0,
                      const float* c0, const float* keep, float* out, float* c_all, float* hm, float* cm,
                      float* h_last, float* c_last, int32_t save_gates, void* workspace, void* stream);
int mirl_lstm_seq_status(int32_t* status);
/* Launch geometry of the forward sweep for (B, H): the workgroups that must all be resident at once, their
 * dynamic LDS, the device's compute units and how many such workgroups one unit holds.  Two sweeps issued on
 * two streams only make progress together when 2 x workgroups <= compute_units x per_compute_unit.        */
int mirl_lstm_seq_fwd_grid(int32_t B, int32_t H, int32_t* workgroups, int64_t* lds_bytes, int32_t* compute_units,
                           int32_t* per_compute_unit);
/* The backward sweep of the same layer in ONE persistent launch (replaces the T-step loop of
 * mirl_lstm_cell_bwd + one recurrent GEMM per step): gates [T][B][4H] holds the activated gates
 * on entry and d loss / d pre-activation on exit; c_all / cm as mirl_lstm_seq_fwd wrote them,
 * d_out [T][B][H] = gradient w.r.t. the layer's outputs (NULL = zero).  The recurrent contraction
 * is split by column owner and its 16 x 16 partial blocks are summed in a fixed order, so reruns
 * are bit-identical.  Shapes: mirl_lstm_seq_bwd_supported (H = 512, B multiple of 16).          */
int mirl_lstm_seq_bwd_supported(int32_t T, int32_t B, int32_t H);
int mirl_lstm_seq_bwd_workspace_bytes(int32_t B, int32_t H, int64_t* bytes);
int mirl_lstm_seq_bwd(int32_t T, int32_t B, int32_t H, float* gates, const float* w_hh, const float* c_all,
                      const float* cm, const float* d_out, const float* keep, void* workspace, void* stream);
/* Backward of one step: gates holds the activated gates on entry and
 * d loss / d pre-activation on exit; d_out [B][H] = grad of this step's output h
 * (NULL = 0); dh_rec / dc_rec = grads w.r.t. the next step's masked inputs
 * (ignored when first != 0, i.e. for the last timestep); dc_rec is overwritten
 * with the grad w.r.t. this step's c_in.                                        */
int mirl_lstm_cell_bwd(int32_t B, int32_t H, float* gates, const float* c_t, const float* c_in,
                       const float* d_out, const float* dh_rec, float* dc_rec, const float* keep_next,
                       int32_t first, void* stream);

/* The CNN's input conversion (rltime/models/torch/modules/cnn.py:44-45,
 * `x.float() * scale`) fused with the NCHW -> NHWC layout change: src uint8
 * [N][C][HW] -> dst float32 [N][HW][C] = src * scale, one pass.                  */
int mirl_frames_to_f32_nhwc(int64_t N, int32_t C, int32_t HW, const uint8_t* src, float scale,
                            float* dst, void* stream);
/* the same with explicit launch shape (tuning probe): per_wg = 1024-pixel tiles per
 * workgroup (0 = heuristic), flags bit 0 = cached loads, bit 1 = cached stores
 * (default: non-temporal both ways).                                              */
int mirl_frames_to_f32_nhwc_ex(int64_t N, int32_t C, int32_t HW, const uint8_t* src, float scale,
                               float* dst, int32_t per_wg, int32_t flags, void* stream);

/* ---- the network's input layer straight from uint8 frames (csrc/conv_in.hip).
 * Replaces, for the first conv layer of the reference's Atari models
 * (rltime/models/torch/modules/cnn.py:44-49 with configs/models/cnn_*.json:
 * Conv2d(4 -> 32, kernel 8, stride 4)), the chain `x.float() * scale` ->
 * conv -> + bias -> ReLU by ONE kernel (default: on the bf16 matrix pipe with an f32 result —
 * a uint8 pixel is exact in bf16 and weight * scale is split exactly into three bf16 parts, so
 * every product is exact and only the f32 summation order differs; flags bit 5 of the _ex form
 * or MIRL_CONV1_BF16=0 selects the f32-MFMA kernel):
 *   x       uint8 [N][4][H][W] (the replay's gathered rows, as stored)
 *   weight  float, logical [32][4][8][8] with element strides ws_o, ws_c, ws_h, ws_w
 *   y       float [N][OH][OW][32] = relu(conv(x * scale, weight) + bias), the NHWC
 *           memory of the logical (N, 32, OH, OW) tensor; OH = (H-8)/4+1, OW likewise
 *   wpk     scratch, 12288 floats (the weights re-ordered into MFMA operand order,
 *           rewritten by every call)
 * Products are float(x)*scale times weight as in the reference; only the order of
 * the 256-term sum differs (fp32 tolerance 1e-4, tests/test_conv_in_gpu.py).
 * mirl_conv1_u8_supported() says whether a layer shape is covered (C = 4, F = 32,
 * K = 8, S = 4, W % 4 == 0, H*W % 16 == 0, 4 padded planes <= 64 KiB of LDS);
 * anything else returns MIRL_ERR_ARG and the caller keeps the generic path
 * (mirl_frames_to_f32_nhwc + library convolution).                              */
int mirl_conv1_u8_supported(int32_t C, int32_t H, int32_t W, int32_t F, int32_t K, int32_t S);
/* floats the `wpk` scratch block of mirl_conv1_u8_fwd[_ex] must hold (12288 since the bf16-pipe kernel; 8192 before):
 * size the block from this query.  A call with flags bit 3 (weights already packed) on a block that was packed by the
 * OTHER kernel variant (f32 / bf16 pipe) returns MIRL_ERR_STATE instead of reading a mismatched layout.               */
int mirl_conv1_u8_wpk_floats(int64_t* out);
/* process-wide choice of the forward's matrix pipe: 1 = bf16 (exact split), 0 = f32 MFMA, < 0 = back to the
 * MIRL_CONV1_BF16 environment default.  For in-process A/B runs (tests/test_network_ab_gpu.py).              */
int mirl_conv1_bf16_set(int32_t mode);
/* Same switch for the layer's WEIGHT gradient (mirl_conv1_u8_wrw / _wrw_ex / _wrw_masked): 1 = bf16 MFMA with the exact
 * three-way split of the gradient operand (pixels are exact in bf16; the default, MIRL_CONV1_WRW_BF16), 0 = f32 MFMA,
 * negative = back to the environment's choice.  In-process A/B tests.                                                  */
int mirl_conv1_wrw_bf16_set(int32_t mode);
int mirl_conv1_u8_fwd(int64_t N, int32_t H, int32_t W, const uint8_t* x, const float* weight,
                      int64_t ws_o, int64_t ws_c, int64_t ws_h, int64_t ws_w, const float* bias,
                      float scale, float* wpk, float* y, void* stream);
/* the same with explicit launch shape (tuning probe): flags bit 0 = cached output
 * stores (default non-temporal), bit 2 = conversions interleaved with the MFMA chain (default:
 * hoisted in front of it), bits 8-15 = frames per LDS fill (1 | 2, 0 = heuristic),
 * bits 16-23 = workgroups sharing one frame's tiles (0 = heuristic), bit 3 = wpk already holds
 * these weights packed by an earlier call with the same kernel choice, bit 5 = f32-MFMA kernel.  */
int mirl_conv1_u8_fwd_ex(int64_t N, int32_t H, int32_t W, const uint8_t* x, const float* weight,
                         int64_t ws_o, int64_t ws_c, int64_t ws_h, int64_t ws_w, const float* bias,
                         float scale, float* wpk, float* y, int32_t flags, void* stream);

/* Weight gradient of the same layer from the same uint8 frames (what autograd derives
 * for cnn.py:44-49; the layer's input needs no gradient):
 *   dw[f][c][kh][kw] = scale * sum over (n, oh, ow) of g[n][oh][ow][f] * float(x[n][c][4 oh + kh][4 ow + kw])
 *   g        float [N][OH][OW][32], the gradient w.r.t. the conv output AFTER the ReLU
 *            mask (mirl_relu_bwd_bias_rows), NHWC memory like y
 *   dw       float, logical [32][4][8][8] written with element strides ws_o, ws_c, ws_h, ws_w
 *   scratch  *out of mirl_conv1_u8_wrw_scratch_floats() floats (per-workgroup partial slabs; summed
 *            in slab order by a second kernel: fixed partition and order, no float atomics,
 *            bit-identical reruns)
 * Same shape coverage as mirl_conv1_u8_fwd.                                             */
int mirl_conv1_u8_wrw_scratch_floats(int64_t* out);
int mirl_conv1_u8_wrw(int64_t N, int32_t H, int32_t W, const uint8_t* x, const float* g, float scale,
                      float* scratch, float* dw, int64_t ws_o, int64_t ws_c, int64_t ws_h, int64_t ws_w,
                      void* stream);
/* the same with an explicit variant (tuning probe): flags bit 0 = byte->float conversions
 * interleaved with the MFMAs instead of hoisted in front of each k-step's 32 MFMAs.      */
int mirl_conv1_u8_wrw_ex(int64_t N, int32_t H, int32_t W, const uint8_t* x, const float* g, float scale,
                         float* scratch, float* dw, int64_t ws_o, int64_t ws_c, int64_t ws_h, int64_t ws_w,
                         int32_t flags, void* stream);
/* The same weight gradient straight from the gradient w.r.t. the layer's OUTPUT: dy is masked by the forward output
 * (y > 0: the layer's ReLU, cnn.py:47-49) while it is loaded and db[32] receives the masked gradient's column sums (the
 * bias gradient) from the same pass — no separate mask / bias-gradient pass over the (N, OH, OW, 32) block.  dy and y in
 * the forward's NHWC layout; scratch as for mirl_conv1_u8_wrw.                                                          */
int mirl_conv1_u8_wrw_masked(int64_t N, int32_t H, int32_t W, const uint8_t* x, const float* dy, const float* y, float scale,
                             float* scratch, float* dw, int64_t ws_o, int64_t ws_c, int64_t ws_h, int64_t ws_w,
                             float* db, void* stream);

/* ---- data gradient of the second conv layer (csrc/conv_mid.hip).  For the backward
 * autograd derives for `F.relu(conv(x))` (rltime/models/torch/modules/cnn.py:47-49) at
 * Conv2d(32 -> 64, kernel 4, stride 2) (configs/models/cnn_*.json), input IH x IW =
 * (2 OH + 2) x (2 OW + 2):
 *   dx[n][ih][iw][c] = sum over f, kh, kw with ih = 2 oh + kh, iw = 2 ow + kw of
 *                      g[n][oh][ow][f] * weight[f][c][kh][kw]
 *   g       float [N][OH][OW][64]  (gradient w.r.t. the conv output after the ReLU mask; NHWC)
 *   weight  float, logical [64][32][4][4], element strides ws_o, ws_c, ws_h, ws_w
 *   dx      float [N][IH][IW][32]  (NHWC), every element written
 *   wpk     scratch, 32768 floats (weights in MFMA operand order, rewritten by every call)
 * f32 MFMA, four parity-class GEMMs with K = 256, N = 32; fp32 tolerance 1e-4
 * (tests/test_conv_mid_gpu.py; bit-exact on integer-valued operands).
 * mirl_conv2_bwd_data_supported() gates the shape; otherwise MIRL_ERR_ARG and the caller
 * keeps the library path.                                                           */
int mirl_conv2_bwd_data_supported(int32_t C, int32_t F, int32_t K, int32_t S, int32_t IH, int32_t IW,
                                  int32_t OH, int32_t OW);
int mirl_conv2_bwd_data(int64_t N, int32_t OH, int32_t OW, const float* g, const float* weight,
                        int64_t ws_o, int64_t ws_c, int64_t ws_h, int64_t ws_w, float* wpk, float* dx,
                        void* stream);
/* The same gradient on either matrix pipe: pipe 0 = f32 MFMA (the call above), pipe 1 = bf16 MFMA with mirl_gemm3's exact
 * three-way split of both operands (six part products, f32 accumulation: f32 results, 2.65 x the f32 pipe's rate; bit-exact
 * on small-integer operands, tests/test_conv_mid_gpu.py).  wpk: wpk_floats >= *floats of mirl_conv2_bwd_data_wpk_floats() (checked), either pipe. */
int mirl_conv2_bwd_data_wpk_floats(int64_t* floats);
int mirl_conv2_bwd_data_ex(int64_t N, int32_t OH, int32_t OW, const float* g, const float* weight, int64_t ws_o,
                           int64_t ws_c, int64_t ws_h, int64_t ws_w, float* wpk, int64_t wpk_floats, float* dx, int32_t pipe,
                           void* stream);
/* Data gradient of the THIRD conv layer (64 -> 64 filters, kernel 3, stride 1; autograd of cnn.py:47-49) on the bf16 pipe,
 * same method and layouts: g float [N][OH][OW][64], dx float [N][OH + 2][OW + 2][64] (NHWC memory), weight (64, 64, 3, 3) by
 * element strides.  wpk: wpk_floats >= *floats of mirl_conv3_bwd_data_wpk_floats() (checked).  mirl_conv3_bwd_data_supported() gates it (no
 * uncovered input rows / columns, (OH + 2)(OW + 2) <= 176).  Replaces MIOpen's igemm_bwd for this layer.                */
int mirl_conv3_bwd_data_supported(int32_t C, int32_t F, int32_t K, int32_t S, int32_t IH, int32_t IW, int32_t OH, int32_t OW);
int mirl_conv3_bwd_data_wpk_floats(int64_t* floats);
int mirl_conv3_bwd_data(int64_t N, int32_t OH, int32_t OW, const float* g, const float* weight, int64_t ws_o,
                        int64_t ws_c, int64_t ws_h, int64_t ws_w, float* wpk, int64_t wpk_floats, float* dx, void* stream);

/* ---- f32 GEMMs of the wide layers on the bf16 matrix pipe (csrc/gemm3.hip).  Replaces the
 * library f32 GEMMs behind the nn.Linear layers of the reference's recurrent IQN model
 * (rltime/policies/torch/dqn.py:50-112 dueling head, rltime/policies/torch/iqn.py:82-102
 * quantile layer, rltime/models/torch/modules/lstm.py:60-81 input projection) and the data /
 * weight gradients autograd derives for them.  Each f32 operand element is split exactly into
 * three bf16 parts (hi + mid + lo = x) while its tile is staged; six of the nine part products
 * are accumulated in f32 (the three dropped ones are < 2^-23 |a||b| per term), so the result
 * is an f32 GEMM: tests/test_gemm3_gpu.py holds it to the float64 product at least as tightly
 * as the library's f32 GEMM.  Non-finite inputs give NaN.  Row-major, strides in floats:
 *   layout 0 "NT": C[M][N] = A[M][K] . B[N][K]^T (+ bias[N], ReLU if relu)      nn.Linear forward
 *   layout 1 "NN": C[M][N] = A[M][K] . B[K][N]                                   data gradient
 *   layout 2 "TN": C[M][N] = A[K][M]^T . B[K][N]                                 weight gradient
 * K % 16 == 0.  k-contiguous operands (A in NT/NN, B in NT): 16-byte aligned, ld % 4 == 0.
 * TN splits K over workgroups into partial tiles in `workspace`
 * (mirl_gemm3_workspace_bytes) that a second kernel sums in a fixed order (deterministic);
 * no bias / ReLU there, C 16-byte aligned, ldc % 4 == 0, N % 4 == 0.
 * mirl_gemm3_supported() gates layout and shape; otherwise MIRL_ERR_ARG.                  */
int mirl_gemm3_supported(int32_t layout, int64_t M, int64_t N, int64_t K);
int mirl_gemm3_workspace_bytes(int32_t layout, int64_t M, int64_t N, int64_t K, int64_t* bytes);
int mirl_gemm3(int32_t layout, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
               const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, int32_t relu,
               void* workspace, int64_t workspace_bytes, void* stream);
/* layout NT with the IQN feature product in the epilogue (rltime/policies/torch/iqn.py:82-102:
 * relu(linear(phi)) times the state's features repeated over its quantile rows):
 *   C[r][c] = f(A B^T + bias)[r][c] * mul[r >> group_shift][c],  f = ReLU if relu
 * and, if `pre` is given, pre[r][c] = f(...)[r][c] (what the backward needs).  Rows of one state are
 * 2^group_shift consecutive rows.  Removes the separate multiply pass over the (rows, N) embedding. */
/* Plain NT products with too few 256 x 256 tiles to fill the chip take a 256 x 128 tile (k_gemm3_mid; the acting batch's
 * hidden layer at 256 envs: 8 192 x 1 024 x 512 in 55 us against 71 with the big tile and 73 on the library).  Process-wide
 * switch for in-process A/B runs: -1 = the MIRL_GEMM3_MID environment default (on), 0 = always the big tile, 1 = on.        */
int mirl_gemm3_mid_set(int32_t mode);
int mirl_gemm3_nt_mul(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B,
                      int64_t ldb, float* C, int64_t ldc, const float* bias, int32_t relu,
                      const float* mul, int64_t ldmul, int32_t group_shift, float* pre, int64_t ldpre,
                      void* stream);
/* The B operand of an NT / NN product split ONCE: a weight matrix multiplies with thousands of row tiles per launch and
 * with several launches per optimizer step, each of which would otherwise split the same tile again.
 *   mirl_gemm3_presplit   rows x K floats (element (r, k) at W[r * row_stride + k * k_stride]: (ld, 1) for the [N][K]
 *                         weight of an NT product, (1, ld) for the [K][N] operand of an NN product) -> `planes`,
 *                         mirl_gemm3_presplit_bytes(rows, K) bytes: per row K / 16 blocks of [hi 16 | mid 16 | lo 16] bf16
 *   mirl_gemm3_ps[_mul]   C = A[M][K] . planes^T (+ bias, ReLU[, x multiplier rows as mirl_gemm3_nt_mul]): the same tiles,
 *                         the same six part products in the same order as mirl_gemm3 — bit-identical results
 *                         (tests/test_gemm3_gpu.py) — minus the B half of the split work in the loop.                  */
/* A wide layer and the NARROW layer that follows it in one pass over the activation: next to C = relu(A . B^T + bias)
 * (NT form of mirl_gemm3; C may be NULL when the activation itself is not needed again — the no-grad passes) the launch
 * produces out2[row][o] = bias2[o] + sum_n C[row][n] * w2[o][n] for o < O <= 8 output units: the dueling head's advantage
 * and value outputs (policies/torch/dqn.py:101-112,50-66) from the epilogue of the joint hidden layer's product, instead
 * of two library GEMMs re-reading the (rows, 1024) activation.  w2 is float [8][N] (rows >= O zero-filled), bias2 [O] or
 * NULL, out2 rows ldo floats apart; workspace = mirl_gemm3_head_workspace_bytes(M, N) bytes, 16-byte aligned (per-row
 * partial sums of every 64-column block, reduced in a fixed order by a second small launch: deterministic).            */
int mirl_gemm3_head_workspace_bytes(int64_t M, int64_t N, int64_t* bytes);
int mirl_gemm3_nt_head(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                       float* C, int64_t ldc, const float* bias, int32_t relu, const float* w2, int32_t O,
                       const float* bias2, float* out2, int64_t ldo, void* workspace, int64_t workspace_bytes, void* stream);
/* The data gradient d = A[M][K] . B[K][N] (NN form) of the layer that FOLLOWS the IQN feature product
 * out[r] = x[r >> 5] * emb[r] (policies/torch/iqn.py:84,102; 32 quantile rows per state), with that product's backward in
 * the epilogue — d itself never reaches HBM (at the benchmark shape 2.7 GB written and read back by mirl_iqn_mul_bwd):
 *   d_pre[r][c] = emb[r][c] > 0 ? d[r][c] * x[r >> 5][c] : 0        (gradient of the embedding layer's pre-activation)
 *   dx[m][c]    = sum over the state's 32 rows of d[r][c] * emb[r][c]   (fixed order)
 *   db_partial[p][c], p < mirl_gemm3_nn_qp_partial_rows(M): column sums of d_pre over 128-row blocks; the caller adds
 *                 the rows (the embedding layer's bias gradient)
 * M % 32 == 0, N % 4 == 0, all rows 16-byte aligned.  Same six part products as mirl_gemm3.                        */
int mirl_gemm3_nn_qp_partial_rows(int64_t M, int64_t* rows);
int mirl_gemm3_nn_qp(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                     const float* emb, int64_t ldemb, const float* x, int64_t ldx, float* d_pre, int64_t ldd,
                     float* dx, int64_t lddx, float* db_partial, void* stream);

/* ---- forward of the middle conv layers on the bf16 matrix pipe, f32 result (csrc/conv3.hip).  Replaces
 * `F.relu(conv(x))` of rltime/models/torch/modules/cnn.py:47-49 for NHWC activations (the Atari models' layers 2
 * and 3: configs/models/cnn_*.json) — the library's f32 implicit GEMM plus a separate bias + ReLU pass — by ONE
 * implicit GEMM with the exact three-way bf16 split of mirl_gemm3 (six part products accumulated in f32):
 *   x     float [N][H][W][C]            (NHWC memory of the logical (N, C, H, W) tensor)
 *   w     float [F][KH][KW][C]          (channels_last memory of the logical (F, C, KH, KW) weight)
 *   y     float [N][OH][OW][F] = f(conv(x, w, stride S, no padding) + bias), f = ReLU if relu
 * mirl_conv3_fwd_supported(): C % 4 == 0, F % 4 == 0, F <= 64, (KW * C) % 16 == 0; otherwise MIRL_ERR_ARG and
 * the caller keeps the library path.  fp32 tolerance as mirl_gemm3 (tests/test_conv3_gpu.py; bit-exact on
 * small-integer operands).                                                                                   */
int mirl_conv3_fwd_supported(int32_t C, int32_t F, int32_t KH, int32_t KW, int32_t S, int32_t H, int32_t W);
int mirl_conv3_fwd(int64_t N, int32_t H, int32_t W, int32_t C, int32_t F, int32_t KH, int32_t KW, int32_t S,
                   const float* x, const float* w, const float* bias, int32_t relu, float* y, void* stream);

/* ---- weight gradient of the middle conv layers on the bf16 matrix pipe, f32 results (csrc/conv_wrw.hip).
 * Replaces MIOpen's igemm_wrw kernels behind autograd for conv layers 2 and 3
 * (rltime/models/torch/modules/cnn.py:43-50):
 *   dw[f][kh][kw][c] = sum over (n, oh, ow) of g[n][oh][ow][f] * x[n][S oh + kh][S ow + kw][c]
 *   x float [N][H][W][C], g float [N][OH][OW][F] (NHWC memory), dw float [F][KH][KW][C] (channels_last weight memory);
 *   no padding, no dilation.  Both operands split exactly into three bf16 parts, six part products, f32 accumulation
 *   (mirl_gemm3's method): bit-exact on small-integer operands, fp32 tolerance otherwise (tests/test_conv_wrw_gpu.py).
 * mirl_conv_wrw_b3_supported(): F == 64, C % 4 == 0, (KH * KW * C) % 16 == 0 and <= 640, one frame's operands inside
 * the LDS; otherwise 0 and the caller keeps the library path.  scratch: mirl_conv_wrw_b3_scratch_bytes() bytes, 16-byte
 * aligned (one [F][KH*KW*C] slab per workgroup, summed in a fixed order: bit-identical reruns).                      */
int mirl_conv_wrw_b3_supported(int32_t C, int32_t F, int32_t KH, int32_t KW, int32_t S, int32_t H, int32_t W);
int mirl_conv_wrw_b3_scratch_bytes(int32_t C, int32_t F, int32_t KH, int32_t KW, int64_t* bytes);
int mirl_conv_wrw_b3(int64_t N, int32_t H, int32_t W, int32_t C, int32_t F, int32_t KH, int32_t KW, int32_t S,
                     const float* x, const float* g, void* scratch, int64_t scratch_bytes, float* dw, void* stream);

/* ---- non-contraction glue around the network's GEMMs / convolutions (csrc/nnops.hip).
 * All tensors row-major (rows, C), channel / feature index fastest (NHWC
 * activations, (M, features) matrices).  One HBM pass each; column sums are
 * two-stage and deterministic.
 *
 * y <- relu(y + bias) in place: the `F.relu(conv(x))` epilogue of
 * rltime/models/torch/modules/cnn.py:47-49 after a bias-less convolution.        */
int mirl_bias_relu_rows(int64_t rows, int32_t C, float* y, const float* bias, void* stream);
/* number of partial-sum blocks the two-stage column sums use for `rows` rows
 * (the caller allocates partial[blocks][C])                                        */
int mirl_colsum_blocks(int64_t rows, int32_t C, int32_t* blocks);
/* backward of relu(. + bias): g = dy * (y > 0) and db[c] = sum_r g[r][c] in one read
 * of dy and y.  C = 4 * 2^k <= 1024.                                               */
int mirl_relu_bwd_bias_rows(int64_t rows, int32_t C, const float* dy, const float* y, float* g,
                            float* db, float* partial, int32_t blocks, void* stream);
/* IQN cosine embedding (rltime/policies/torch/iqn.py:78-81): phi[r][i] =
 * cos(tau[r] * freq[i]), freq = embedding_range * pi (float32), D % 4 == 0.       */
int mirl_cos_embed(int64_t rows, int32_t D, const float* tau, const float* freq, float* phi, void* stream);
/* The same features with the quantile fractions drawn inside the kernel (Philox4x32-10 keyed by
 * (seed, *step, row), 24-bit uniforms in [0, 1) as torch.rand draws them): the device actor's
 * IQN forward without a torch.rand launch per vector step.  tau_out (rows floats) may be NULL.  */
int mirl_cos_embed_rng(int64_t rows, int32_t D, uint64_t seed, const uint64_t* step, const float* freq,
                       float* phi, float* tau_out, void* stream);
/* IQN feature product (iqn.py:84,102): out[m*N+n][c] = x[m][c] * emb[m*N+n][c]; out may be emb
 * itself (no-grad passes: one read-modify-write stream instead of two streams).          */
int mirl_iqn_mul_fwd(int64_t M, int32_t N, int32_t C, const float* x, const float* emb, float* out, void* stream);
/* its backward fused with the ReLU mask and bias gradient of the embedding layer
 * (emb = relu(pre)):  d_pre = emb > 0 ? g * x[m] : 0,  dx[m] = sum_n g * emb,
 * db[c] = sum_r d_pre[r][c].                                                       */
int mirl_iqn_mul_bwd(int64_t M, int32_t N, int32_t C, const float* g, const float* emb, const float* x,
                     float* d_pre, float* dx, float* db, float* partial, int32_t blocks, void* stream);

/* backward of the dueling head's two output layers fused with the ReLU mask and bias
 * gradient of the joint hidden activation `both` (M, H1+Hv):
 *   g = (both > 0) * [ga @ wo | gv @ wq],  db = column sums of g
 * ga (M, A), gv (M, Q), wo (A, H1), wq (Q, Hv) row-major; A, Q <= 16.
 * (rltime/policies/torch/dqn.py:50-66,74-112 — the autograd of out_layer /
 * value_layer over the last FC layer and the value-hidden layer.)                 */
int mirl_dueling_tail_bwd(int64_t M, int32_t H1, int32_t Hv, int32_t A, int32_t Q, const float* ga,
                          const float* gv, const float* wo, const float* wq, const float* both,
                          float* g, float* db, float* partial, int32_t blocks, void* stream);
/* the same pass with the two output layers' weight gradients riding along (`both` is in registers anyway):
 *   dwj[k][c] = sum_r ga[r][k] * both[r][c]        (c <  H1, k < A):  d loss / d out_layer.weight[k][c]
 *             = sum_r gv[r][k] * both[r][c]        (c >= H1, k < Q):  d loss / d value_layer.weight[k][c - H1]
 * dwj is KW x (H1+Hv) floats with KW = max(A, Q) <= 8 (rows k >= A resp. k >= Q of a branch are zero);
 * partial_w is scratch of blocks * KW * (H1+Hv) floats; fixed-order (deterministic) block sums.       */
int mirl_dueling_tail_bwd_w(int64_t M, int32_t H1, int32_t Hv, int32_t A, int32_t Q, const float* ga,
                            const float* gv, const float* wo, const float* wq, const float* both, float* g,
                            float* db, float* partial, int32_t blocks, float* dwj, float* partial_w,
                            void* stream);

/* ---- device-resident actor bookkeeping (csrc/acting.hip) -----------------------
 * Episode statistics on the RAW rewards and the action histogram
 * (rltime/training/policy_trainer.py:75-131), one launch per vector step:
 * ep_reward / ep_len [E] are the running accumulators; out_reward / out_len [E] get
 * (reward, length) of the episodes that ended at this step (length 0 = none);
 * action_counts [A] accumulates (may be NULL).  The host reads the out rows back
 * asynchronously.                                                                  */
/* The acting head after the network, one launch per vector step: adv (E*N, A) and the
 * optional dueling value val (E*N, Q; only column 0 is used) -> q[e][a] = mean over
 * the N quantile rows of (val + adv - mean_a adv) (policies/torch/dqn.py:74-87,
 * iqn.py actor post-processing), greedy action (first maximum), epsilon-greedy remap
 * (exploration/epsilon_greedy.py:74-100): per-env epsilon = max(eps ** expo[e], eps_min),
 * action := rnd[e] when u[e] < epsilon.  eps NULL = no exploration; val NULL = plain
 * (non-dueling) outputs.  eps / expo are device doubles (eps one value).           */
int mirl_actor_head(int32_t E, int32_t N, int32_t A, const float* adv, const float* val, int32_t Q,
                    const double* eps, const double* expo, double eps_min, const float* u, const int64_t* rnd,
                    int32_t* actions, float* qvalues, float* eps_used, void* stream);
/* The same head with its epsilon-greedy draws made inside the kernel (one Philox4x32-10
 * block per (step, env), keyed by rng_seed and the device word *rng_step that
 * mirl_actor_pre advances): no torch.rand / torch.randint launches per vector step.
 * adv rows are adv_pitch floats apart and val rows Q floats apart, so both may be column
 * blocks of one (rows, A + q) output GEMM (val = adv + A, Q = adv_pitch).                */
int mirl_actor_head_rng(int32_t E, int32_t N, int32_t A, const float* adv, int32_t adv_pitch, const float* val, int32_t Q,
                        const double* eps, const double* expo, double eps_min, uint64_t rng_seed,
                        const uint64_t* rng_step, int32_t* actions, float* qvalues, float* eps_used, void* stream);
/* The frame-stack wrapper's shift for a device-resident vector env (env_wrappers/common.py:141-178
 * with auto-reset): out[e] = [in[e] planes 1..P-1 — or zeros where dones[e] —, newest[e]].  One launch
 * (the synthetic env's torch expression of it is four).                                              */
int mirl_stack_shift(int32_t E, int32_t P, int32_t plane_bytes, const uint8_t* in, uint8_t* out,
                     const uint8_t* newest, const uint8_t* dones, void* stream);
/* Everything between env.step and the policy forward of the device-resident actor in ONE
 * launch (acting/actor.py:124-131 -> modules/lstm.py:131-161 state reset on `done`;
 * policy_trainer.py:93-131 episode statistics; :252-254 reward sign clipping): h_in = h *
 * (1 - done) -> xh_tail rows (pitch xh_pitch floats: the tail of the LSTM GEMM's
 * [features | h_in] input), c_in = c * (1 - done), state_pack [E][2H] = [h_in | c_in] and
 * initials = done (the transition's stored recurrent state), rewards_out (clipped when
 * clip_rewards), dones_out (uint8), the mirl_episode_track accumulators (optional) and
 * *rng_step = step — or, with step = MIRL_STEP_ADVANCE, *rng_step + 1: the counter then lives on
 * the device alone and the call can be replayed from a captured graph.  H = 0: a policy
 * without a recurrent layer (h, c, xh_tail, c_in, state_pack may then be NULL).               */
#define MIRL_STEP_ADVANCE (~0ull)
int mirl_actor_pre(int32_t E, int32_t H, int32_t A, const float* rewards_raw, const uint8_t* dones,
                   const int32_t* actions, const float* h, const float* c, float* xh_tail, int64_t xh_pitch,
                   float* c_in, float* state_pack, float* initials, float* rewards_out, uint8_t* dones_out,
                   int32_t clip_rewards, float* ep_reward, int32_t* ep_len, float* out_reward, int32_t* out_len,
                   int32_t* action_counts, uint64_t* rng_step, uint64_t step, void* stream);
/* One step of the synthetic Atari-shaped vector env of the benchmark (i.i.d. uint8 frames from a
 * pre-generated pool, rewards in {-1, 0, 1}, done with a fixed probability) decided entirely on the device:
 * step t = clock[slot] + 1, obs [E][frame_bytes] <- pool[t % pool_n] (pool [pool_n][E][frame_bytes]), reward / done from one
 * Philox4x32-10 block per (seed, t, env) with cumulative reward probabilities p_neg <= p_nonpos; the launch stores
 * clock[slot ^ 1] = t, so consecutive steps alternate `slot` (0, 1, 0, ...: host-tracked parity, fixed per position of a
 * captured rollout).  `clock` is a 16-byte aligned block of two 64-bit words.  No host-side state otherwise.          */
int mirl_synth_env_step(int32_t E, int64_t frame_bytes, const uint8_t* pool, int32_t pool_n, uint64_t* clock, int32_t slot,
                        uint64_t seed, float p_neg, float p_nonpos, float p_done, uint8_t* obs, float* rewards,
                        uint8_t* dones, void* stream);
/* ---- the policy network of one acting vector step at acting batch sizes (csrc/actnet.hip) -------------------------
 * Replaces, for CNN -> LSTM -> [quantile layer] -> FC -> dueling head policies, the library calls the actor's forward
 * (acting/actor.py:108-122 -> policies/torch/dqn.py:132-148, iqn.py:67-106) was made of: six launches per vector step
 * instead of fifteen, f32 MFMA throughout.
 *
 * mirl_act_conv_fwd: conv layer 2 (Ci 32, 4x4, stride 2) or 3 (Ci 64, 3x3, stride 1) of the Atari stack with 64 output
 * channels, bias + ReLU (models/torch/modules/cnn.py:43-50).  x (frames, Hi, Wi, Ci) NHWC f32; w_taps (64, k*k*Ci) =
 * the conv weight permuted to (Co, kh, kw, Ci); y: frame f at y + f * y_frame_pitch floats, pixel p / channel c at
 * [p * 64 + c] (a pitch larger than Ho*Wo*64 lets layer 3 write the head of the LSTM product's input rows).          */
int mirl_act_conv_supported(int32_t layer, int32_t Ci, int32_t Co, int32_t k, int32_t stride, int32_t Hi, int32_t Wi);
int mirl_act_conv_fwd(int32_t layer, int64_t frames, int32_t Hi, int32_t Wi, const float* x, const float* w_taps,
                      const float* bias, float* y, int64_t y_frame_pitch, void* stream);
/* One LSTMCell step for E <= 64 rows (models/torch/modules/lstm.py:83-116 at timesteps = 1, gate order i, f, g, o):
 * gates = xh (E rows of K = F + H floats, xh_pitch apart: [features | h_in]) x w^T (4H, K: [W_ih | W_hh]) + bias
 * (4H: b_ih + b_hh); c_out = f * c_in + i * g; h_out = o * tanh(c_out).  One launch, the gates never reach HBM:
 * (H / 8) column blocks x KB slices of K, the last slice of a block to arrive adds the shares and runs the cell.
 * `workspace` (mirl_act_lstm_workspace_bytes, 16-byte aligned) must be ZERO before the first call; every launch
 * leaves its counters at zero again, so the call can sit in a captured graph.                                       */
int mirl_act_lstm_supported(int32_t E, int32_t H, int32_t K);
int mirl_act_lstm_workspace_bytes(int32_t E, int32_t H, int32_t K, int64_t* bytes);
int mirl_act_lstm_fwd(int32_t E, int32_t H, int32_t K, const float* xh, int64_t xh_pitch, const float* w, const float* bias,
                      const float* c_in, float* h_out, float* c_out, void* workspace, void* stream);
/* The quantile layer's product over R = E * N rows (policies/torch/iqn.py:67-106): tau = taus[m] or a Philox draw keyed
 * (seed, *step, m) exactly as mirl_cos_embed_rng; x[m] = relu(cos(freq * tau) . wq^T + bq) * h[m / N] (wq (H, D),
 * D in {16, 32, 48, 64}; x (R, H)); tau_out (R) optional.                                                           */
int mirl_act_embed(int32_t E, int32_t N, int32_t H, int32_t D, const float* h, const float* freq, const float* taus,
                   uint64_t seed, const uint64_t* step, const float* wq, const float* bq, float* x, float* tau_out, void* stream);
/* The head's hidden layers over R rows of x (R, H): hid = relu(x . wfc^T + bfc) (wfc (HID, H): the last FC layer's rows,
 * then the dueling value-hidden layer's, dqn.py:50-66); part[p][m][o] = the share of out[m][o] = hid[m] . wout[o] (wout
 * (NO, HID)) that hidden columns 64p .. 64p+63 contribute — p < `parts`, rows `pitch` floats apart
 * (mirl_act_head_parts).  The (R, HID) activations stay in registers / LDS.                                         */
int mirl_act_head_supported(int32_t E, int32_t N, int32_t H, int32_t D, int32_t HID, int32_t NO);
int mirl_act_head_parts(int32_t HID, int32_t NO, int32_t* parts, int32_t* pitch);
int mirl_act_head_hidden(int32_t R, int32_t H, int32_t HID, int32_t NO, const float* x, const float* wfc, const float* bfc,
                         const float* wout, float* part, void* stream);
/* mirl_actor_head_rng over those shares: out[m][o] = bout[o] + sum_p part[p][m][o]; columns 0 .. A-1 = advantages,
 * column A = the dueling value when has_val; then V + A - mean_a A, mean over the N rows, first maximum,
 * epsilon-greedy with the same Philox draws (policies/torch/dqn.py:74-87, exploration/epsilon_greedy.py:74-100).   */
int mirl_act_head_select(int32_t E, int32_t N, int32_t A, int32_t parts, int32_t pitch, const float* part, const float* bout,
                         int32_t has_val, const double* eps, const double* expo, double eps_min, uint64_t rng_seed,
                         const uint64_t* rng_step, int32_t* actions, float* qvalues, void* stream);
/* mirl_synth_env_step and mirl_actor_pre as ONE launch: the workgroup that draws env e's reward / done runs env e's
 * pre-step on them (arguments: those of the two calls; rewards / dones still receive the env's raw outputs).      */
int mirl_synth_env_step_pre(int32_t E, int64_t frame_bytes, const uint8_t* pool, int32_t pool_n, uint64_t* clock, int32_t slot,
                            uint64_t seed, float p_neg, float p_nonpos, float p_done, uint8_t* obs, float* rewards,
                            uint8_t* dones, int32_t H, int32_t A, const int32_t* actions, const float* h, const float* c,
                            float* xh_tail, int64_t xh_pitch, float* c_in, float* state_pack, float* initials,
                            float* rewards_out, uint8_t* dones_out, int32_t clip_rewards, float* ep_reward, int32_t* ep_len,
                            float* out_reward, int32_t* out_len, int32_t* action_counts, uint64_t* rng_step, uint64_t step,
                            void* stream);
int mirl_episode_track(int32_t E, int32_t A, const float* rewards, const uint8_t* dones,
                       const int32_t* actions, float* ep_reward, int32_t* ep_len,
                       float* out_reward, int32_t* out_len, int32_t* action_counts, void* stream);

/* ---- device copy micro-benchmark used by bench.py for the measured HBM peak */
int mirl_copy_bytes(void* dst, const void* src, int64_t bytes, void* stream);
/* nt = 0: grid-stride copy with cached accesses; 1: one 16 B chunk per lane, non-temporal (bench.py reports the better one). */
int mirl_copy_bytes_ex(void* dst, const void* src, int64_t bytes, int32_t nt, void* stream);

/* ---- host-only hooks (no GPU needed; used by the CPU test-suite) ----------
 * A bookkeeping-only replay (rings, FIFO, free list, activation) that records
 * the device plan it would issue.                                             */
typedef struct mirl_book mirl_book;
int mirl_book_create(const mirl_replay_config* cfg, mirl_book** out);
int mirl_book_destroy(mirl_book* b);
int mirl_book_ingest(mirl_book* b, int32_t count, const int32_t* env_ids_host);
int mirl_book_stats(mirl_book* b, int64_t* total_items, int64_t* active_sequences,
                    int64_t* train_quota, int64_t* tree_capacity, int64_t* n_slots);
int mirl_book_env_meta(mirl_book* b, int64_t* first_host, int64_t* count_host);
int mirl_book_free_slots(mirl_book* b, int32_t* slots_host, int64_t* n);
int mirl_book_slot_table(mirl_book* b, int32_t* slot_env_host, int64_t* slot_base_host);
int mirl_book_needed_feed_count(mirl_book* b, int32_t mbatch, int32_t num_envs, int64_t* out);
int mirl_book_charge_quota(mirl_book* b, int32_t mbatch);
int mirl_book_uniform_total(mirl_book* b, int64_t* total);
int mirl_book_uniform_map(mirl_book* b, int32_t mbatch, const int64_t* picks_host,
                          int32_t* env_host, int64_t* start_host);
/* np_emul.h on the host: build the tagged heap from leaves, run descents, and
 * evaluate one sequence priority — compared with the golden tree fixtures.     */
int mirl_emul_build_tree(int64_t capacity, const double* leaf_value, const uint8_t* leaf_kind,
                         double* node_value, uint8_t* node_kind);
int mirl_emul_find(int64_t capacity, const double* node_value, const uint8_t* node_kind,
                   int32_t mbatch, const double* uniforms, int64_t* idx);
int mirl_emul_seq_priority(int32_t nstep_train, double alpha, double max_weight_factor,
                           const float* loss_slots, double* value, uint8_t* kind);

/* ---- the learner step's tail: global gradient norm -> clip -> Adam, in two launches (csrc/optim.hip) ----------------
 * replaces rltime/training/torch_trainer.py:177-199 (clip_grad_norm_ + torch.optim.Adam.step(): amsgrad off, no weight
 * decay) over `count` float32 tensors of numel[i] elements each; param / grad / exp_avg / exp_avg_sq [count] are HOST
 * arrays of device pointers (element i of all four has the same memory layout), step [count] of device pointers to one
 * float each (the per-parameter step counters of a capturable torch Adam: each advances by 1 and its tensor's bias
 * corrections use its new value, in float64).  coef = min(clip / (norm + 1e-6), 1) scales the gradients IN PLACE before the update
 * (clip <= 0: no scaling); lr_dev (a device float) overrides lr when not NULL.  workspace: mirl_adam_clip_workspace_bytes,
 * 8-byte aligned, private to the call until the stream has passed it.  norm_out (device, may be NULL) receives
 * [norm, norm * coef].  Graph-capturable: the pointers are baked into the launches.                                  */
int mirl_adam_clip_workspace_bytes(int32_t count, const int64_t* numel, int64_t* bytes);
int mirl_adam_clip_step(int32_t count, float* const* param, float* const* grad, float* const* exp_avg,
                        float* const* exp_avg_sq, float* const* step, const int64_t* numel, double lr,
                        const float* lr_dev, double beta1, double beta2, double eps, double clip, void* workspace,
                        int64_t workspace_bytes, float* norm_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MIRL_H */
