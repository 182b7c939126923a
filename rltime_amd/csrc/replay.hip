// replay.hip — device-resident replay shard for gfx950 (MI355X).
//
// Storage is struct-of-arrays in HBM, one ring of C slots per env:
//     frames  u8  [E][C][Fp]   next_state["x"] of every transition (Fp = F padded to 16 B)
//     extra   f32 [E][C][X]    tuple-observation extra features
//     state   f32 [E][C][S]    stored recurrent state (hx|cx per layer)
//     initials/rewards f32, actions i32, dones u8, policy f32 [E][C][A]
//     loss f32, prio_index i32, stamp u64          (prioritized replay only)
// Transition `off` of env `e` lives in slot off % C.  Its *state* is the
// next_state of transition off-1 (history.py:167; the first ever transition
// of an env reuses its own, :163), so one extra slot per env keeps the
// predecessor of the oldest live transition alive and nothing is stored twice.
//
// Kernels (all HBM- or latency-bound; no MFMA — there is no contraction here):
//   k_scatter_rows    ingest: [K][row] dense -> ring slots            (a1, a16)
//   k_ingest_scalars  ingest: per-transition scalars + PER init       (a1, a2)
//   k_plan_apply      ingest: host plan -> tables, leaves, dirty list (a2)
//   k_tree_fix        re-sum ancestors of dirty leaves, level by level (a3)
//   k_per_sample      LDS-staged stratified descent + IS weights      (a3, a4, a8)
//   k_uniform_sample  flat choice -> (env, start) (+ episode refine)  (a5)
//   k_gather_rows(_v1) time-major state-block gather, 16 B/lane non-temporal (a7)
//   k_gather_scalars  n-step scan + per-step scalars                  (a6, a7)
//   k_loss_stamp / k_loss_write / k_recalc_flagged                    (a9)
// Row labels are SURVEY.md section 8(a).  Also here: mirl_replay_save / _load
// (snapshot) and the host-only test hooks (mirl_book_*, mirl_emul_*).
#include "common.hpp"
#include "book.hpp"
#include "np_emul.h"
#include "philox.hpp"
#include "vfscale.hpp"

#include <cstring>
#include <cmath>
#include <cstdlib>

namespace mirl {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct Dev {                 // passed by value to kernels
  int32_t E, F, Fp, X, S, A, has_init, per;
  int32_t T, P, N, L, gap, env_base, log2cap, avoid_xing;
  int64_t C, cap, n_slots;
  uint8_t* frames; float* extra; float* state; float* initials;
  int32_t* actions; float* policy; float* rewards; uint8_t* dones;
  float* loss; int32_t* prio_index; unsigned long long* stamp;
  double* tv; uint8_t* tk; double* tmin;
  int32_t* slot_env; int64_t* slot_base; uint8_t* flag;
  int32_t* dirty; int32_t* dirty_count;
  int64_t* first; int64_t* count;
  int32_t planes, plane_bytes;   // frame de-dup: P planes per stack, bytes of one plane (0 = stacks stored whole)
  uint8_t* depth;            // [E][C] real planes in the stack of each transition's next_state (de-dup)
  int32_t* bad;              // host-coherent flag: a sampled leaf was inactive (reference: assert, prioritized_replay_history.py:306)
  const double* gpow;        // gamma ** k, k < N, computed by the host libm like Python's float.__pow__
  double alpha, mwf, eps;
};

__device__ __forceinline__ int64_t slot_of(const Dev& d, int32_t e, int64_t off) {
  return (int64_t)e * d.C + off % d.C;
}

// ---------------------------------------------------------------------------
// ingest
// ---------------------------------------------------------------------------
// Copy K dense rows into ring slots.  One block column per row, 16 B per lane
// when the geometry allows, else a byte loop.
__global__ void __launch_bounds__(256)
k_scatter_rows(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
               const int32_t* __restrict__ s_env, const int64_t* __restrict__ s_off,
               int64_t C, int32_t row_bytes, int64_t dst_stride, int vec, int64_t src_stride) {
  const int k = blockIdx.y;
  const int64_t slot = (int64_t)s_env[k] * C + s_off[k] % C;
  const uint8_t* s = src + (int64_t)k * src_stride;
  uint8_t* t = dst + slot * dst_stride;
  if (vec) {
    const int n = row_bytes >> 4;
    const u32x4* s4 = (const u32x4*)s;
    u32x4* t4 = (u32x4*)t;
    for (int c = blockIdx.x * 256 + threadIdx.x; c < n; c += gridDim.x * 256) t4[c] = s4[c];
  } else {
    for (int c = blockIdx.x * 256 + threadIdx.x; c < row_bytes; c += gridDim.x * 256) t[c] = s[c];
  }
}

__global__ void __launch_bounds__(256)
k_ingest_scalars(Dev d, int K, const int32_t* __restrict__ s_env, const int64_t* __restrict__ s_off,
                 const float* __restrict__ initials, const int32_t* __restrict__ actions,
                 const float* __restrict__ rewards, const uint8_t* __restrict__ dones) {
  int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  int64_t sl = slot_of(d, s_env[k], s_off[k]);
  if (d.has_init) d.initials[sl] = initials[k];
  d.actions[sl] = actions[k];
  d.rewards[sl] = rewards[k];
  d.dones[sl] = dones[k] ? 1 : 0;
  if (d.per) {
    d.loss[sl] = MIRL_LOSS_FRESH;     // sample['loss'] = self._max_loss (python 1.0), prioritized_replay_history.py:141
    d.prio_index[sl] = -1;
    d.stamp[sl] = 0ull;
  }
}

// ---------------------------------------------------------------------------
// frame de-duplication (mirl_replay_config.stack_planes)
// ---------------------------------------------------------------------------
// One workgroup per ingested transition: how many planes of its stack are real.
// Plane P-1 is the new one; plane P-1-j must equal the newest plane stored with
// the env's transition off-j (the wrapper rolled the stack), for j = 1.. until
// the first mismatch; everything older must then be the zero fill of a reset
// (env_wrappers/common.py:175-178).  Anything else violates the contract.
__global__ void __launch_bounds__(256)
k_dedup_depth(Dev d, const uint8_t* __restrict__ frames, const int32_t* __restrict__ s_env,
              const int64_t* __restrict__ s_off) {
  const int k = blockIdx.x;
  const int32_t e = s_env[k];
  const int64_t off = s_off[k];
  const u32x4* stack = (const u32x4*)(frames + (int64_t)k * d.F);
  const int nq = d.plane_bytes >> 4;
  int depth = 1;
  bool chain = true;
  int violation = 0;
  for (int j = 1; j < d.planes; ++j) {
    const u32x4* mine = stack + (int64_t)(d.planes - 1 - j) * nq;
    int same = 1, zero = 1;
    const bool have = off - j >= 0;
    // the env's first transitions have no stored predecessor: their older planes (the
    // reset observation's) become VIRTUAL predecessors in the ring slots -1, -2, ...
    // (mod C), which no live transition can occupy before they are out of reach
    u32x4* theirs = (u32x4*)(d.frames + ((int64_t)e * d.C + ((off - j) % d.C + d.C) % d.C) * (int64_t)d.plane_bytes);
    if (!have && !chain) same = 0;               // nothing stored to match and the chain is already cut: must be zero fill
    for (int q = threadIdx.x; q < nq; q += 256) {
      u32x4 a = mine[q];
      if (a.x | a.y | a.z | a.w) zero = 0;
      if (have) { u32x4 b = theirs[q]; if ((a.x ^ b.x) | (a.y ^ b.y) | (a.z ^ b.z) | (a.w ^ b.w)) same = 0; }
      else if (chain) theirs[q] = a;             // virtual predecessor (only while it continues this stack's chain:
    }                                            //  a cut chain must not overwrite an older stack's virtual planes)
    same = __syncthreads_and(same);
    zero = __syncthreads_and(zero);
    if (chain && same) ++depth;
    else { chain = false; if (!zero) violation = 1; }
  }
  if (threadIdx.x == 0) {
    d.depth[slot_of(d, e, off)] = (uint8_t)depth;
    if (violation) *(volatile int32_t*)d.bad = 2;
  }
}

struct LossGet {
  const float* loss; int64_t C; int64_t ring0; int64_t off;
  __device__ float operator()(int t) const { return loss[ring0 + (off + t) % C]; }
};

__device__ __forceinline__ TV priority_of(const Dev& d, int32_t e, int64_t base) {
  LossGet g{d.loss, d.C, (int64_t)e * d.C, base};
  PrioParams p{d.T, d.alpha, d.mwf};
  return seq_priority(g, p);
}

// Apply one ingest plan (book.hpp): table writes, env mirrors, leaf ops.
__global__ void __launch_bounds__(256)
k_plan_apply(Dev d, int n_table, const TableOp* __restrict__ tops, int n_env, const EnvOp* __restrict__ eops,
             int n_leaf, const LeafOp* __restrict__ lops) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n_table) d.prio_index[slot_of(d, tops[i].env, tops[i].off)] = tops[i].value;
  if (i < n_env) { d.first[eops[i].env] = eops[i].first; d.count[eops[i].env] = eops[i].count; }
  if (i < n_leaf) {
    LeafOp op = lops[i];
    int64_t leaf = d.cap + op.slot;
    if (op.activate) {
      d.slot_env[op.slot] = op.env; d.slot_base[op.slot] = op.base;
      TV p = priority_of(d, op.env, op.base);
      d.tv[leaf] = p.v; d.tk[leaf] = p.k;
      if (d.tmin) d.tmin[leaf] = p.v;
    } else {                         // prioritized_replay_history.py:223-227
      d.slot_env[op.slot] = -1; d.slot_base[op.slot] = -1;
      d.tv[leaf] = 0.0; d.tk[leaf] = KW;
      if (d.tmin) d.tmin[leaf] = INFINITY;
    }
    int at = atomicAdd(d.dirty_count, 1);
    d.dirty[at] = op.slot;
  }
}

// Re-derive every ancestor of the dirty leaves from its two children, one tree
// level per barrier (segment_tree.py:91-97; the result depends only on the
// leaves, so the batch order is irrelevant).  Single workgroup: the tree is
// latency-bound, not bandwidth-bound.
__device__ __forceinline__ void tree_fix_body(const Dev& d) {
  const int n = *d.dirty_count;
  for (int lvl = d.log2cap - 1; lvl >= 0; --lvl) {
    for (int i = threadIdx.x; i < n; i += 1024) {
      int64_t node = (d.cap + d.dirty[i]) >> (d.log2cap - lvl);
      TV a{d.tv[2 * node], d.tk[2 * node]}, b{d.tv[2 * node + 1], d.tk[2 * node + 1]};
      TV r = tadd(a, b);
      d.tv[node] = r.v; d.tk[node] = r.k;
      if (d.tmin) { double x = d.tmin[2 * node], y = d.tmin[2 * node + 1]; d.tmin[node] = y < x ? y : x; }
    }
    __syncthreads();
  }
  __syncthreads();
  if (threadIdx.x == 0) *d.dirty_count = 0;
}

__global__ void __launch_bounds__(1024)
k_tree_fix(Dev d) { tree_fix_body(d); }

// The whole device side of one History.update call (history.py:123-176 + the _sample_added /
// _sample_removed hooks) in ONE launch — rounds 1-2 issued k_scatter_rows x 2-4,
// k_ingest_scalars, k_plan_apply and k_tree_fix per vector step.  Grid (bx, K + 1):
//   y < K   copy workgroups of transition y: frame row, extra / recurrent-state / q-value rows
//           into their ring slots (16 B per lane when the geometry allows);
//   y == K  ONE bookkeeping workgroup: per-transition scalars (so that the freshly ingested
//           losses exist before a leaf activation reads them), the plan's table / env /
//           leaf ops, and — when a leaf changed — the tree fix, in this order behind
//           workgroup barriers.  Copy and bookkeeping workgroups touch disjoint arrays.
struct IngestSrc {
  const uint8_t* frames; const float* extra; const float* state; const float* policy;
  const float* initials; const int32_t* actions; const float* rewards; const uint8_t* dones;
  int vec_frames, vec_extra, vec_state, vec_policy;
  int64_t frames_stride;      // bytes between the frame rows of consecutive transitions in `frames`
  int32_t row_bytes, slot_bytes;   // bytes copied per transition / ring slot pitch (de-dup: one plane)
};

__device__ __forceinline__ void copy_row(const uint8_t* s, uint8_t* t, int row_bytes, int vec, int part, int parts) {
  if (!row_bytes) return;
  if (vec) {
    const int n = row_bytes >> 4;
    const u32x4* s4 = (const u32x4*)s;
    u32x4* t4 = (u32x4*)t;
    for (int c = part * 1024 + threadIdx.x; c < n; c += parts * 1024) t4[c] = s4[c];
  } else {
    for (int c = part * 1024 + threadIdx.x; c < row_bytes; c += parts * 1024) t[c] = s[c];
  }
}

__device__ __forceinline__ void
ingest_fused_body(const Dev& d, int K, const IngestSrc& in, const int32_t* __restrict__ s_env, const int64_t* __restrict__ s_off,
                  int n_table, const TableOp* __restrict__ tops, int n_env, const EnvOp* __restrict__ eops,
                  int n_leaf, const LeafOp* __restrict__ lops) {
  const int k = blockIdx.y;
  if (k < K) {
    const int64_t slot = (int64_t)s_env[k] * d.C + s_off[k] % d.C;
    const int part = blockIdx.x, parts = gridDim.x;
    copy_row(in.frames + (int64_t)k * in.frames_stride, d.frames + slot * (int64_t)in.slot_bytes, in.row_bytes, in.vec_frames, part, parts);
    if (d.X) copy_row((const uint8_t*)(in.extra + (int64_t)k * d.X), (uint8_t*)(d.extra + slot * (int64_t)d.X), d.X * 4, in.vec_extra, part, parts);
    if (d.S) copy_row((const uint8_t*)(in.state + (int64_t)k * d.S), (uint8_t*)(d.state + slot * (int64_t)d.S), d.S * 4, in.vec_state, part, parts);
    if (d.A) copy_row((const uint8_t*)(in.policy + (int64_t)k * d.A), (uint8_t*)(d.policy + slot * (int64_t)d.A), d.A * 4, in.vec_policy, part, parts);
    return;
  }
  if (blockIdx.x) return;
  for (int i = threadIdx.x; i < K; i += 1024) {
    const int64_t sl = slot_of(d, s_env[i], s_off[i]);
    if (d.has_init) d.initials[sl] = in.initials[i];
    d.actions[sl] = in.actions[i];
    d.rewards[sl] = in.rewards[i];
    d.dones[sl] = in.dones[i] ? 1 : 0;
    if (d.per) { d.loss[sl] = MIRL_LOSS_FRESH; d.prio_index[sl] = -1; d.stamp[sl] = 0ull; }
    if (d.planes) {
      // newest-plane form of de-duplicated storage: the stack of transition `off` has one real
      // plane after a reset (an unprimed env's first transition, or this transition ended an
      // episode: env_wrappers/common.py:175-178 zero-fills the rest), else one more than its predecessor's
      const int64_t off = s_off[i];
      int dep = 1;
      // off == 0: the predecessor is the reset observation mirl_replay_prime_stack put into ring
      // slot -1 (depth 0 there = nothing primed: the first stack is taken as a reset stack)
      const int64_t prev = (int64_t)s_env[i] * d.C + ((off - 1) % d.C + d.C) % d.C;
      // (an auto-resetting env returns the NEW episode's first stack with the transition that has done = 1)
      if (!in.dones[i] && (off > 0 || d.depth[prev] > 0)) { dep = (int)d.depth[prev] + 1; if (dep > d.planes) dep = d.planes; }
      d.depth[sl] = (uint8_t)dep;
    }
  }
  __threadfence_block();
  __syncthreads();
  for (int i = threadIdx.x; i < n_table; i += 1024) d.prio_index[slot_of(d, tops[i].env, tops[i].off)] = tops[i].value;
  for (int i = threadIdx.x; i < n_env; i += 1024) { d.first[eops[i].env] = eops[i].first; d.count[eops[i].env] = eops[i].count; }
  for (int i = threadIdx.x; i < n_leaf; i += 1024) {
    const LeafOp op = lops[i];
    const int64_t leaf = d.cap + op.slot;
    if (op.activate) {
      d.slot_env[op.slot] = op.env; d.slot_base[op.slot] = op.base;
      TV p = priority_of(d, op.env, op.base);
      d.tv[leaf] = p.v; d.tk[leaf] = p.k;
      if (d.tmin) d.tmin[leaf] = p.v;
    } else {
      d.slot_env[op.slot] = -1; d.slot_base[op.slot] = -1;
      d.tv[leaf] = 0.0; d.tk[leaf] = KW;
      if (d.tmin) d.tmin[leaf] = INFINITY;
    }
    const int at = atomicAdd(d.dirty_count, 1);
    d.dirty[at] = op.slot;
  }
  if (n_leaf) {                                   // uniform over the workgroup
    __threadfence_block();
    __syncthreads();
    tree_fix_body(d);
  }
}

__global__ void __launch_bounds__(1024)
k_ingest_fused(Dev d, int K, IngestSrc in, const int32_t* __restrict__ s_env, const int64_t* __restrict__ s_off,
               int n_table, const TableOp* __restrict__ tops, int n_env, const EnvOp* __restrict__ eops,
               int n_leaf, const LeafOp* __restrict__ lops) {
  ingest_fused_body(d, K, in, s_env, s_off, n_table, tops, n_env, eops, n_leaf, lops);
}

// The same launch with its op lists taken from step `step` of a ROLLOUT PLAN that already sits in device memory
// (mirl_replay_ingest_plan): no per-step host argument, so a whole acting rollout — env, policy, ingest x the
// quota's vector steps — can be captured into ONE HIP graph and replayed with one launch per learner step.
struct PlanHdr { int32_t n_table, n_env, n_leaf, pad; int64_t o_env, o_off, o_tab, o_eop, o_lop; };

__global__ void __launch_bounds__(1024)
k_ingest_fused_planned(Dev d, int K, IngestSrc in, const char* __restrict__ plan, int step) {
  const PlanHdr hd = reinterpret_cast<const PlanHdr*>(plan)[step];
  ingest_fused_body(d, K, in, reinterpret_cast<const int32_t*>(plan + hd.o_env), reinterpret_cast<const int64_t*>(plan + hd.o_off),
                    hd.n_table, reinterpret_cast<const TableOp*>(plan + hd.o_tab), hd.n_env, reinterpret_cast<const EnvOp*>(plan + hd.o_eop),
                    hd.n_leaf, reinterpret_cast<const LeafOp*>(plan + hd.o_lop));
}

// full rebuild of one level (test hook mirl_replay_tree_set_leaves)
__global__ void k_tree_level(Dev d, int64_t lo, int64_t hi) {
  int64_t node = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (node >= hi) return;
  TV a{d.tv[2 * node], d.tk[2 * node]}, b{d.tv[2 * node + 1], d.tk[2 * node + 1]};
  TV r = tadd(a, b);
  d.tv[node] = r.v; d.tk[node] = r.k;
  if (d.tmin) { double x = d.tmin[2 * node], y = d.tmin[2 * node + 1]; d.tmin[node] = y < x ? y : x; }
}
__global__ void k_set_leaves(Dev d, int64_t n, const double* v, const uint8_t* k) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  d.tv[d.cap + i] = v[i]; d.tk[d.cap + i] = k[i];
  if (d.tmin) d.tmin[d.cap + i] = (v[i] == 0.0) ? INFINITY : v[i];
}

// ---------------------------------------------------------------------------
// sampling
// ---------------------------------------------------------------------------
// Philox4x32-10 for the device-RNG mode: philox.hpp

// _refine_sample_range (replay_history.py:142-171) on absolute offsets.
__device__ __forceinline__ int64_t refine_start(const Dev& d, int32_t e, int64_t start) {
  if (!d.avoid_xing) return start;
  const int amount = d.L;
  const int64_t first = d.first[e], cnt = d.count[e];
  for (int k = 0; k < amount - 1; ++k) {
    if (d.dones[slot_of(d, e, start + k)]) {
      if (2 * k < amount) {                 // index < amount / 2
        int64_t s = start - (amount - k - 1);
        start = s > first ? s : first;      // max(pos - (...), 0)
      } else {
        int64_t s = start + k + 1, hi = cnt - amount;
        start = s < hi ? s : hi;            // min(pos + k + 1, len - amount)
      }
      break;
    }
  }
  return start;
}

#define MIRL_LDS_NODES 2048
// One workgroup: the top 11 levels of the heap are staged in LDS, every lane
// runs one stratified query (prioritized_replay_history.py:232-241) with the
// exact promotion arithmetic of np_emul.h, then the importance weights
// (:327, :347-354) with a wavefront-shuffle max.
__global__ void __launch_bounds__(1024)
k_per_sample(Dev d, int B, const double* __restrict__ uniforms, uint64_t seed, uint64_t call,
             double active, double beta, int global_scale,
             int32_t* __restrict__ slot_out, int32_t* __restrict__ env_out, int64_t* __restrict__ start_out,
             int64_t* __restrict__ base_out, float* __restrict__ w_out, double* __restrict__ w_tmp,
             double* __restrict__ stats) {
  __shared__ double s_tv[MIRL_LDS_NODES];
  __shared__ uint8_t s_tk[MIRL_LDS_NODES];
  __shared__ double s_red[16];
  const int64_t staged = 2 * d.cap < MIRL_LDS_NODES ? 2 * d.cap : MIRL_LDS_NODES;
  for (int i = threadIdx.x; i < staged; i += 1024) { s_tv[i] = d.tv[i]; s_tk[i] = d.tk[i]; }
  __syncthreads();
  const TV total{s_tv[1], s_tk[1]};
  double local_max = 0.0;
  for (int i = threadIdx.x; i < B; i += 1024) {
    double u = uniforms ? uniforms[i] : philox_u53(seed, call, (uint32_t)i);
    TV mass = stratum_mass(total, B, i, u);
    int64_t pos = 1;
    while (pos < d.cap) {
      int64_t c = 2 * pos;
      TV left;
      if (c < staged) { left.v = s_tv[c]; left.k = s_tk[c]; } else { left.v = d.tv[c]; left.k = d.tk[c]; }
      if (tgreater(left, mass)) pos = c; else { mass = tsub(mass, left); pos = c + 1; }
    }
    int32_t idx = (int32_t)(pos - d.cap);
    slot_out[i] = idx;
    int32_t e = d.slot_env[idx];
    const int64_t base = d.slot_base[idx];
    int64_t start = base - d.P;
    if (e >= 0) start = refine_start(d, e, start);
    else *(volatile int32_t*)d.bad = 1;               // surfaced as MIRL_ERR_STATE by the next host call
    env_out[i] = e; start_out[i] = start;
    if (base_out) base_out[i] = base;                  // losses stay attached to the unshifted sequence
    // weight = ((leaf / p_sum) * total_items) ** (-beta)
    double w = e >= 0 ? pow((d.tv[pos] / total.v) * active, -beta) : 0.0;   // an inactive row trains with weight 0
    w_tmp[i] = w;
    local_max = w > local_max ? w : local_max;
  }
  // batch max: wavefront shuffle reduction, then 16 partials through LDS
  for (int o = 32; o > 0; o >>= 1) { double x = __shfl_xor(local_max, o); local_max = x > local_max ? x : local_max; }
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = local_max;
  __syncthreads();
  double top = s_red[0];
  for (int k = 1; k < 16; ++k) top = s_red[k] > top ? s_red[k] : top;
  if (global_scale) top = pow((d.tmin[1] / total.v) * active, -beta);   // :350-351
  for (int i = threadIdx.x; i < B; i += 1024) w_out[i] = (float)(w_tmp[i] / top);
  if (stats && threadIdx.x == 0) { stats[0] = total.v; stats[1] = top; }
}

// ---------------------------------------------------------------------------
// Exact global stratified sampling over env-sharded replays.
// Every rank runs this kernel with the same Philox key / step counter and the same
// table of shard totals; the B_g strata of the GLOBAL mass [0, sum_q P_q) are assigned
// to the shard whose cumulative range [C_r, C_r + P_r) contains them — what one tree
// over the concatenation of the shards' leaves would sample — so a rank draws a
// data-dependent number of sequences (about B_g P_r / P_g).  Its batch has a fixed
// B_pad rows; unused rows are padding (slot -1, env -1 -> weight 0, no loss index).
// Cross-shard arithmetic is float64; the local descent uses the shard tree's own
// tagged arithmetic with a float64 mass.  out: w_raw = (p/P_g N_g)^-beta (double),
// stats = {P_g, local max raw weight, strata owned, strata dropped (B_pad overflow)}.
__global__ void __launch_bounds__(1024)
k_per_sample_global(Dev d, int Bg, int Bpad, int rank, int R, const double* __restrict__ shard, uint64_t seed, uint64_t call,
                    double beta, int32_t* __restrict__ slot_out, int32_t* __restrict__ env_out, int64_t* __restrict__ start_out,
                    int64_t* __restrict__ base_out, double* __restrict__ w_raw, int32_t* __restrict__ stratum_out,
                    double* __restrict__ stats) {
  __shared__ double s_tv[MIRL_LDS_NODES];
  __shared__ uint8_t s_tk[MIRL_LDS_NODES];
  __shared__ double s_red[16];
  __shared__ int s_cnt[2][16];
  const int64_t staged = 2 * d.cap < MIRL_LDS_NODES ? 2 * d.cap : MIRL_LDS_NODES;
  for (int i = threadIdx.x; i < staged; i += 1024) { s_tv[i] = d.tv[i]; s_tk[i] = d.tk[i]; }
  double Pg = 0.0, Ng = 0.0, Cr = 0.0;
  for (int q = 0; q < R; ++q) { if (q < rank) Cr += shard[2 * q]; Pg += shard[2 * q]; Ng += shard[2 * q + 1]; }
  const double Pr = shard[2 * rank], seg = Pg / (double)Bg;
  const double hi_edge = rank == R - 1 ? INFINITY : Cr + Pr;
  // pass 1: how many strata lie below this shard's range / inside it
  int below = 0, mine = 0;
  for (int i = threadIdx.x; i < Bg; i += 1024) {
    const double mass = (philox_u53(seed, call, (uint32_t)i) + (double)i) * seg;
    below += mass < Cr;
    mine += mass >= Cr && mass < hi_edge;
  }
  for (int o = 32; o > 0; o >>= 1) { below += __shfl_xor(below, o); mine += __shfl_xor(mine, o); }
  if ((threadIdx.x & 63) == 0) { s_cnt[0][threadIdx.x >> 6] = below; s_cnt[1][threadIdx.x >> 6] = mine; }
  __syncthreads();
  below = 0; mine = 0;
  for (int k = 0; k < 16; ++k) { below += s_cnt[0][k]; mine += s_cnt[1][k]; }
  // pass 2: owned strata, in stratum order, to rows [0, mine)
  double local_max = 0.0;
  for (int i = threadIdx.x; i < Bg; i += 1024) {
    const double mass = (philox_u53(seed, call, (uint32_t)i) + (double)i) * seg;
    if (!(mass >= Cr && mass < hi_edge)) continue;
    const int p = i - below;
    if (p >= Bpad) continue;                                    // counted as dropped below
    TV m{mass - Cr, K64};
    int64_t pos = 1;
    while (pos < d.cap) {
      int64_t c = 2 * pos;
      TV left;
      if (c < staged) { left.v = s_tv[c]; left.k = s_tk[c]; } else { left.v = d.tv[c]; left.k = d.tk[c]; }
      if (tgreater(left, m)) pos = c; else { m = tsub(m, left); pos = c + 1; }
    }
    const int32_t idx = (int32_t)(pos - d.cap);
    const int32_t e = d.slot_env[idx];
    const int64_t base = d.slot_base[idx];
    int64_t start = base - d.P;
    if (e >= 0) start = refine_start(d, e, start);
    slot_out[p] = idx; env_out[p] = e; start_out[p] = start; stratum_out[p] = i;
    if (base_out) base_out[p] = base;
    const double w = e >= 0 ? pow((d.tv[pos] / Pg) * Ng, -beta) : 0.0;
    w_raw[p] = w;
    local_max = w > local_max ? w : local_max;
  }
  const int kept = mine < Bpad ? mine : Bpad;
  for (int p = kept + threadIdx.x; p < Bpad; p += 1024) {       // padding rows
    slot_out[p] = -1; env_out[p] = -1; start_out[p] = 0; stratum_out[p] = -1; w_raw[p] = 0.0;
    if (base_out) base_out[p] = 0;
  }
  for (int o = 32; o > 0; o >>= 1) { double x = __shfl_xor(local_max, o); local_max = x > local_max ? x : local_max; }
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = local_max;
  __syncthreads();
  if (threadIdx.x == 0) {
    double top = s_red[0];
    for (int k = 1; k < 16; ++k) top = s_red[k] > top ? s_red[k] : top;
    stats[0] = Pg; stats[1] = top; stats[2] = (double)kept; stats[3] = (double)(mine - kept);
  }
}

__global__ void k_tree_root(Dev d, double active, double* __restrict__ out) { out[0] = d.tv[1]; out[1] = active; }

// test hook: descent only
__global__ void k_tree_find(Dev d, int B, const double* __restrict__ uniforms, int64_t* __restrict__ idx) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  TV total{d.tv[1], d.tk[1]};
  idx[i] = tagged_descend(d.tv, d.tk, d.cap, stratum_mass(total, B, i, uniforms[i]));
}

// Uniform replay (replay_history.py:118-134): flat choice -> (env, start) by a
// binary search over the per-env cumulative availability the host uploaded.
__global__ void __launch_bounds__(256)
k_uniform_sample(Dev d, int B, const int64_t* __restrict__ picks, uint64_t seed, uint64_t call,
                 int n_cum, const int64_t* __restrict__ cum, const int32_t* __restrict__ cum_env,
                 int32_t* __restrict__ slot_out, int32_t* __restrict__ env_out,
                 int64_t* __restrict__ start_out, int64_t* __restrict__ base_out, float* __restrict__ w_out) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B) return;
  const int64_t total = cum[n_cum - 1];
  int64_t p;
  if (picks) p = picks[i];
  else { p = (int64_t)(philox_u53(seed, call, (uint32_t)i) * (double)total); if (p >= total) p = total - 1; }
  int lo = 0, hi = n_cum - 1;
  while (lo < hi) { int mid = (lo + hi) >> 1; if (p < cum[mid]) hi = mid; else lo = mid + 1; }
  int32_t e = cum_env[lo];
  int64_t before = lo ? cum[lo - 1] : 0;
  int64_t start = refine_start(d, e, d.first[e] + (p - before));
  slot_out[i] = (int32_t)p; env_out[i] = e; start_out[i] = start; w_out[i] = 1.0f;
  if (base_out) base_out[i] = start + d.P;
}

// ---------------------------------------------------------------------------
// gather
// ---------------------------------------------------------------------------
// Source ring offset of state-block row r for the window starting at `start`.
__device__ __forceinline__ int64_t row_src_off(const Dev& d, int overlapped, int r, int32_t e, int64_t start) {
  int64_t o;
  if (overlapped || r < d.L) {
    o = start + r;                       // state of transition o == next_state of o-1
    return o > 0 ? o - 1 : 0;            // history.py:163 first-ever sample
  }
  o = start + (r - d.L);                 // target state of transition o
  int64_t avail = d.count[e] - o;        // history.py:83 end = min(index+n, len)
  int64_t ns = avail < d.N ? avail : d.N;
  return o + ns - 1;
}

// out[r][b][:] = ring[env[b]][src(r, b)][:] for one leaf of the state pytree.
// One workgroup per (r, b) row: the source is a contiguous ring slot, the
// destination a contiguous row of the time-major batch, 16 B per lane, four
// independent loads in flight per lane before the first store.
template <int NT>
__global__ void __launch_bounds__(256)
k_gather_rows(Dev d, const uint8_t* __restrict__ ring, uint8_t* __restrict__ out,
              const int32_t* __restrict__ env, const int64_t* __restrict__ start,
              int B, int overlapped, int32_t row_bytes, int64_t ring_stride, int vec) {
  const int64_t rb = blockIdx.x;
  const int r = (int)(rb / B), b = (int)(rb % B);
  int32_t e = env[b];
  if (e < 0 || e >= d.E) e = 0;          // inactive slot (reference would assert); stay in bounds
  const int64_t src_off = row_src_off(d, overlapped, r, e, start[b]);
  const uint8_t* s = ring + ((int64_t)e * d.C + src_off % d.C) * ring_stride;
  uint8_t* t = out + rb * (int64_t)row_bytes;
  if (vec) {
    const int n = row_bytes >> 4;
    const u32x4* s4 = (const u32x4*)s;
    u32x4* t4 = (u32x4*)t;
    int c = threadIdx.x;
    for (; c + 768 < n; c += 1024) {
      u32x4 v0, v1, v2, v3;
      if (NT) {
        v0 = __builtin_nontemporal_load(s4 + c); v1 = __builtin_nontemporal_load(s4 + c + 256);
        v2 = __builtin_nontemporal_load(s4 + c + 512); v3 = __builtin_nontemporal_load(s4 + c + 768);
        __builtin_nontemporal_store(v0, t4 + c); __builtin_nontemporal_store(v1, t4 + c + 256);
        __builtin_nontemporal_store(v2, t4 + c + 512); __builtin_nontemporal_store(v3, t4 + c + 768);
      } else {
        v0 = s4[c]; v1 = s4[c + 256]; v2 = s4[c + 512]; v3 = s4[c + 768];
        t4[c] = v0; t4[c + 256] = v1; t4[c + 512] = v2; t4[c + 768] = v3;
      }
    }
    for (; c < n; c += 256) {
      if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(s4 + c), t4 + c);
      else t4[c] = s4[c];
    }
  } else {
    for (int c = threadIdx.x; c < row_bytes; c += 256) t[c] = s[c];
  }
}

// Default launch shape for 16-byte-aligned rows (MIRL_GATHER_VARIANT=1): 512
// lanes per row and every load of the row issued before its first store.
// Measured on MI355X at B=512, L+n=122, 1M-transition replay (profiles/):
// 0.638 ms (5.53 TB/s) vs 0.654 ms for the 256-lane kernel above; a persistent
// 2048..8192-workgroup variant and a source-contiguous block order were slower
// (0.66-0.72 ms) at this replay size and were dropped.
template <int NTL>
__global__ void __launch_bounds__(512)
k_gather_rows_v1(Dev d, const uint8_t* __restrict__ ring, uint8_t* __restrict__ out,
                 const int32_t* __restrict__ env, const int64_t* __restrict__ start,
                 int B, int overlapped, int32_t row_bytes, int64_t ring_stride, int order) {
  int64_t rb = blockIdx.x;
  const int R = gridDim.x / B;
  int r, b;
  if (order) { b = (int)(rb / R); r = (int)(rb % R); rb = (int64_t)r * B + b; }   // source-contiguous order
  else { r = (int)(rb / B); b = (int)(rb % B); }
  int32_t e = env[b];
  if (e < 0 || e >= d.E) e = 0;
  const int64_t src_off = row_src_off(d, overlapped, r, e, start[b]);
  const u32x4* s4 = (const u32x4*)(ring + ((int64_t)e * d.C + src_off % d.C) * ring_stride);
  u32x4* t4 = (u32x4*)(out + rb * (int64_t)row_bytes);
  const int n = row_bytes >> 4;
  int c = threadIdx.x;
  for (; c + 1536 < n; c += 2048) {
    u32x4 v0, v1, v2, v3;
    if (NTL) {
      v0 = __builtin_nontemporal_load(s4 + c); v1 = __builtin_nontemporal_load(s4 + c + 512);
      v2 = __builtin_nontemporal_load(s4 + c + 1024); v3 = __builtin_nontemporal_load(s4 + c + 1536);
    } else { v0 = s4[c]; v1 = s4[c + 512]; v2 = s4[c + 1024]; v3 = s4[c + 1536]; }
    __builtin_nontemporal_store(v0, t4 + c); __builtin_nontemporal_store(v1, t4 + c + 512);
    __builtin_nontemporal_store(v2, t4 + c + 1024); __builtin_nontemporal_store(v3, t4 + c + 1536);
  }
  u32x4 w0, w1, w2; bool h0 = c < n, h1 = c + 512 < n, h2 = c + 1024 < n;
  if (h0) w0 = NTL ? __builtin_nontemporal_load(s4 + c) : s4[c];
  if (h1) w1 = NTL ? __builtin_nontemporal_load(s4 + c + 512) : s4[c + 512];
  if (h2) w2 = NTL ? __builtin_nontemporal_load(s4 + c + 1024) : s4[c + 1024];
  if (h0) __builtin_nontemporal_store(w0, t4 + c);
  if (h1) __builtin_nontemporal_store(w1, t4 + c + 512);
  if (h2) __builtin_nontemporal_store(w2, t4 + c + 1024);
}

// out[r][b] = the P-plane stack of ring transition src(r, b), rebuilt from the
// newest planes of that transition and its P-1 predecessors (zero fill beyond the
// recorded depth).  Same launch shape as the whole-frame gather: 512 lanes per
// 28 KB row, 16 B per lane, all loads of the row issued before the first store.
__global__ void __launch_bounds__(512)
k_gather_rows_dedup(Dev d, uint8_t* __restrict__ out, const int32_t* __restrict__ env,
                    const int64_t* __restrict__ start, int B, int overlapped) {
  const int64_t rb = blockIdx.x;
  const int r = (int)(rb / B), b = (int)(rb % B);
  int32_t e = env[b];
  if (e < 0 || e >= d.E) e = 0;
  const int64_t o = row_src_off(d, overlapped, r, e, start[b]);
  const int dep = d.depth[slot_of(d, e, o)];
  const int nq = d.plane_bytes >> 4, n = nq * d.planes;
  u32x4* t4 = (u32x4*)(out + rb * (int64_t)d.F);
  const uint8_t* ring0 = d.frames + (int64_t)e * d.C * d.plane_bytes;
  u32x4 v[4]; int c[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    c[k] = threadIdx.x + k * 512;
    v[k] = u32x4{0u, 0u, 0u, 0u};
    if (c[k] < n) {
      const int p = c[k] / nq, q = c[k] - p * nq, back = d.planes - 1 - p;
      if (back < dep) v[k] = __builtin_nontemporal_load((const u32x4*)(ring0 + (((o - back) % d.C + d.C) % d.C) * (int64_t)d.plane_bytes) + q);
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) if (c[k] < n) __builtin_nontemporal_store(v[k], t4 + c[k]);
  for (int cc = threadIdx.x + 2048; cc < n; cc += 512) {       // stacks larger than 32 KB
    const int p = cc / nq, q = cc - p * nq, back = d.planes - 1 - p;
    u32x4 w = u32x4{0u, 0u, 0u, 0u};
    if (back < dep) w = __builtin_nontemporal_load((const u32x4*)(ring0 + (((o - back) % d.C + d.C) % d.C) * (int64_t)d.plane_bytes) + q);
    __builtin_nontemporal_store(w, t4 + cc);
  }
}

// The same rebuild with the planes of a run of consecutive rows staged ONCE in LDS:
// rows r .. r+RC-1 of one sequence come from consecutive transitions, whose stacks
// share all but one plane, so a workgroup loads RC + P - 1 planes (instead of RC * P)
// and writes RC stacks: (RC + P - 1) / RC read amplification instead of P.  Chunks
// whose source transitions are not one short run (layout seams, truncated n-step
// targets at the ring end) take the direct path of k_gather_rows_dedup.
#define MIRL_DD_MAX_PLANES 22
__global__ void __launch_bounds__(512)
k_gather_rows_dedup_lds(Dev d, uint8_t* __restrict__ out, const int32_t* __restrict__ env,
                        const int64_t* __restrict__ start, int B, int R, int overlapped, int RC, int lds_planes) {
  extern __shared__ u32x4 s_planes[];
  const int chunks = (R + RC - 1) / RC;
  const int ch = blockIdx.x / B, b = blockIdx.x - ch * B;
  (void)chunks;
  const int r0 = ch * RC, r1 = r0 + RC < R ? r0 + RC : R;
  int32_t e = env[b];
  if (e < 0 || e >= d.E) e = 0;
  const int64_t st = start[b];
  int64_t lo = INT64_MAX, hi = INT64_MIN;
  for (int r = r0; r < r1; ++r) { int64_t o = row_src_off(d, overlapped, r, e, st); lo = o < lo ? o : lo; hi = o > hi ? o : hi; }
  const int nq = d.plane_bytes >> 4, n = nq * d.planes;
  const uint8_t* ring0 = d.frames + (int64_t)e * d.C * d.plane_bytes;
  const int64_t first = lo - (d.planes - 1);
  const int span = (int)(hi - first + 1);
  const bool staged = span <= lds_planes;
  if (staged) {
    for (int c = threadIdx.x; c < span * nq; c += 512) {
      const int p = c / nq, q = c - p * nq;
      s_planes[c] = __builtin_nontemporal_load((const u32x4*)(ring0 + (((first + p) % d.C + d.C) % d.C) * (int64_t)d.plane_bytes) + q);
    }
    __syncthreads();
  }
  for (int r = r0; r < r1; ++r) {
    const int64_t o = row_src_off(d, overlapped, r, e, st);
    const int dep = d.depth[slot_of(d, e, o)];
    u32x4* t4 = (u32x4*)(out + ((int64_t)r * B + b) * (int64_t)d.F);
    for (int c = threadIdx.x; c < n; c += 512) {
      const int p = c / nq, q = c - p * nq, back = d.planes - 1 - p;
      u32x4 v = u32x4{0u, 0u, 0u, 0u};
      if (back < dep) {
        if (staged) v = s_planes[(int)(o - back - first) * nq + q];
        else v = __builtin_nontemporal_load((const u32x4*)(ring0 + (((o - back) % d.C + d.C) % d.C) * (int64_t)d.plane_bytes) + q);
      }
      __builtin_nontemporal_store(v, t4 + c);
    }
  }
}

// Per-step scalars of the batch.  One lane per (t, b):
//   _update_nstep (history.py:71-108): forward scan over <= n rewards / dones,
//   return accumulated in float64 with gamma**k from the host libm (the Python
//   float arithmetic of multi_step_trainer.py:72-73), two roundings per term;
//   actions, stored policy outputs, importance weights, loss indices
//   (prioritized_replay_history.py:329-338) and the `initials` rows.
__global__ void __launch_bounds__(256)
k_gather_scalars(Dev d, int B, int R, int overlapped, const int32_t* __restrict__ env,
                 const int64_t* __restrict__ start, const int64_t* __restrict__ loss_start,
                 const float* __restrict__ weight, mirl_batch o) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int rows = R > d.L ? R : d.L;
  if (i >= (int64_t)rows * B) return;
  const int r = (int)(i / B), b = (int)(i % B);
  int32_t e = env[b];
  const bool inactive = e < 0 || e >= d.E;        // rows stay in bounds but report no loss index
  if (inactive) e = 0;
  const int64_t s0 = start[b];
  if (r < R && o.initials) {
    int64_t src = row_src_off(d, overlapped, r, e, s0);
    o.initials[i] = d.initials[slot_of(d, e, src)];
  }
  if (r >= d.L) return;
  const int64_t off = s0 + r;
  const int64_t sl = slot_of(d, e, off);
  double ret = (double)d.rewards[sl];              // history.py:146 float(reward)
  int mask = d.dones[sl] ? 0 : 1;                  // :147
  int ns = 1;
  const int64_t cnt = d.count[e];
  for (int k = 1; k < d.N && off + k < cnt; ++k) {
    int64_t sj = slot_of(d, e, off + k);
    if (mask) {                                    // :87-90
      double term = d.gpow[k] * (double)d.rewards[sj];
      ret = ret + term;
    }
    ++ns;                                          // :98
    if (d.dones[sj]) mask = 0;                     // :104-108
  }
  o.returns[i] = (float)ret;
  o.nsteps[i] = (float)ns;
  o.masks[i] = (float)mask;
  o.actions[i] = (int64_t)d.actions[sl];
  if (o.policy) for (int a = 0; a < d.A; ++a) o.policy[i * d.A + a] = d.policy[sl * d.A + a];
  if (o.weights) o.weights[i] = weight[b];
  if (o.loss_indices) {
    if (r < d.P || inactive) { o.loss_indices[2 * i] = -1; o.loss_indices[2 * i + 1] = -1; }
    else {
      o.loss_indices[2 * i] = (int64_t)(e + d.env_base);
      o.loss_indices[2 * i + 1] = loss_start ? loss_start[b] + (r - d.P) : off;
    }
  }
}

// ---------------------------------------------------------------------------
// update_losses (prioritized_replay_history.py:243-279)
// ---------------------------------------------------------------------------
// The reference walks (index, loss) pairs in order, so when a transition occurs
// twice the LAST pair wins.  Pass 1 elects that winner per transition with a
// 64-bit atomicMax of (call epoch << 32 | pair index); pass 2 lets only the
// winner write, and every pair marks the overlapped sequences it touches.
__global__ void __launch_bounds__(256)
k_loss_stamp(Dev d, int64_t n, const int64_t* __restrict__ idx, uint64_t epoch) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int64_t env_id = idx[2 * i], off = idx[2 * i + 1];
  if (env_id < 0) return;
  int32_t e = (int32_t)(env_id - d.env_base);
  if (e < 0 || e >= d.E || off < d.first[e] || off >= d.count[e]) return;   // :254 evicted meanwhile
  atomicMax(&d.stamp[slot_of(d, e, off)], (unsigned long long)((epoch << 32) | (uint64_t)i));
}
__global__ void __launch_bounds__(256)
k_loss_write(Dev d, int64_t n, const int64_t* __restrict__ idx, const float* __restrict__ losses, uint64_t epoch) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int64_t env_id = idx[2 * i], off = idx[2 * i + 1];
  if (env_id < 0) return;
  int32_t e = (int32_t)(env_id - d.env_base);
  if (e < 0 || e >= d.E) return;
  const int64_t first = d.first[e];
  if (off < first || off >= d.count[e]) return;
  int64_t sl = slot_of(d, e, off);
  if (d.stamp[sl] == (unsigned long long)((epoch << 32) | (uint64_t)i))
    d.loss[sl] = fabsf(losses[i]) + (float)d.eps;          // :263 abs(np.float32) + eps -> float32
  int64_t base = off - off % d.gap;                        // :267-274
  while (base + d.T > off && base >= first) {
    int32_t s = d.prio_index[slot_of(d, e, base)];
    if (s >= 0) d.flag[s] = 1;
    base -= d.gap;
  }
}
__global__ void __launch_bounds__(256)
k_recalc_flagged(Dev d) {
  int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= d.n_slots || !d.flag[s]) return;
  d.flag[s] = 0;
  int32_t e = d.slot_env[s];
  if (e < 0) return;
  TV p = priority_of(d, e, d.slot_base[s]);
  d.tv[d.cap + s] = p.v; d.tk[d.cap + s] = p.k;
  if (d.tmin) d.tmin[d.cap + s] = p.v;
  int at = atomicAdd(d.dirty_count, 1);
  d.dirty[at] = (int32_t)s;
}

// Same as k_recalc_flagged for 8 <= T <= 128 (one NumPy pairwise block): one
// WAVE per sequence instead of one lane walking T strided losses.  The T loss
// slots are read coalesced into LDS; lanes 0..7 keep NumPy's eight interleaved
// accumulators r0..r7 (np_pairwise_block: r_j = a[j] + a[8+j] + a[16+j] + ...),
// three shuffle steps combine them in NumPy's association
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), lane 0 adds the n%8 tail; the max is a
// wave shuffle reduction.  Values and kinds are those of np_emul.h::seq_priority.
template <class T_>
__device__ __forceinline__ void wave_pairwise(const float* sl, int n, int lane, bool as_f64_fresh, T_& sum, T_& mx) {
  auto g = [&](int t) -> T_ { float l = sl[t]; return (as_f64_fresh && l < 0.0f) ? (T_)1.0 : (T_)l; };
  T_ m = g(lane < n ? lane : 0);
  for (int t = lane + 64; t < n; t += 64) { T_ x = g(t); m = x > m ? x : m; }
  for (int o = 32; o > 0; o >>= 1) { T_ x = __shfl_xor(m, o); m = x > m ? x : m; }
  mx = m;
  const int j = lane & 7, full = n - (n % 8);
  T_ r = g(j);
  for (int i = 8; i < full; i += 8) r = r + g(i + j);
  r = r + __shfl_down(r, 1);          // lanes 0,2,4,6: r0+r1, r2+r3, r4+r5, r6+r7
  r = r + __shfl_down(r, 2);          // lanes 0,4
  r = r + __shfl_down(r, 4);          // lane 0
  for (int i = full; i < n; ++i) r = r + g(i);
  sum = r;                            // valid on lane 0
}

__global__ void __launch_bounds__(256)
k_recalc_flagged_wave(Dev d) {
  __shared__ float s_loss[4][128];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t s = (int64_t)blockIdx.x * 4 + w;
  bool live = s < d.n_slots && d.flag[s];
  int32_t e = live ? d.slot_env[s] : -1;
  float* sl = s_loss[w];
  const int n = d.T;
  if (live && e >= 0) {
    const int64_t base = d.slot_base[s], ring0 = (int64_t)e * d.C;
    for (int t = lane; t < n; t += 64) sl[t] = d.loss[ring0 + (base + t) % d.C];
  }
  __syncthreads();
  if (!live) return;
  if (lane == 0) d.flag[s] = 0;
  if (e < 0) return;
  bool fresh = false;
  for (int t = lane; t < n; t += 64) fresh |= sl[t] < 0.0f;
  fresh = __any(fresh);
  TV p;
  if (fresh) {                         // a Python-float 1.0 in the list -> float64 array
    double sum, mx;
    wave_pairwise<double>(sl, n, lane, true, sum, mx);
    double mean = sum / (double)n;
    double a = d.mwf * mx, b = (1.0 - d.mwf) * mean;
    double mixed = a + b;
    p.v = pow(mixed, d.alpha); p.k = K64;
  } else {
    float sum, mx;
    wave_pairwise<float>(sl, n, lane, false, sum, mx);
    float mean = (float)((double)sum / (double)n);
    float a = (float)d.mwf * mx, b = (float)(1.0 - d.mwf) * mean;
    float mixed = a + b;
    p.v = (double)(float)pow((double)mixed, (double)(float)d.alpha); p.k = K32;
  }
  if (lane == 0) {
    d.tv[d.cap + s] = p.v; d.tk[d.cap + s] = p.k;
    if (d.tmin) d.tmin[d.cap + s] = p.v;
    int at = atomicAdd(d.dirty_count, 1);
    d.dirty[at] = (int32_t)s;
  }
}

// Acting-time priority initialisation (mirl_replay_config.acting_priority_init).
// One lane per ingested transition j: the TD error of transition t = j - n from
// stored rewards / dones / actions / q-values, emitted as an update_losses row.
__global__ void __launch_bounds__(256)
k_acting_td(Dev d, int K, const int32_t* __restrict__ s_env, const int64_t* __restrict__ s_off, double vf_eps,
            int64_t* __restrict__ idx, float* __restrict__ loss) {
  int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  const int32_t e = s_env[k];
  const int64_t j = s_off[k], t = j - d.N;
  idx[2 * k] = -1; idx[2 * k + 1] = -1; loss[k] = 0.f;
  if (t < d.first[e]) return;
  const int64_t st = slot_of(d, e, t);
  double ret = (double)d.rewards[st];                  // History.update / _update_nstep (history.py:71-108,146-147)
  int mask = d.dones[st] ? 0 : 1;
  for (int q = 1; q < d.N; ++q) {
    const int64_t sq = slot_of(d, e, t + q);
    if (mask) { double term = d.gpow[q] * (double)d.rewards[sq]; ret = ret + term; }
    if (d.dones[sq]) mask = 0;
  }
  const float* qj = d.policy + slot_of(d, e, j) * d.A;
  float v = qj[0];
  for (int a = 1; a < d.A; ++a) v = qj[a] > v ? qj[a] : v;
  const float y = finish_target(v, (float)ret, (float)d.gpow[d.N], (float)mask, vf_eps);
  const float chosen = d.policy[st * d.A + d.actions[st]];
  idx[2 * k] = (int64_t)(e + d.env_base); idx[2 * k + 1] = t;
  loss[k] = chosen - y;
}

__global__ void k_copy16(u32x4* __restrict__ dst, const u32x4* __restrict__ src, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// same copy with non-temporal access, one 16 B element per lane and a full-size
// grid: the device-copy ceiling the gather is compared against
__global__ void __launch_bounds__(512) k_copy16_nt(u32x4* __restrict__ dst, const u32x4* __restrict__ src, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * 2048 + threadIdx.x;
  u32x4 v0, v1, v2, v3;
  bool h0 = i < n, h1 = i + 512 < n, h2 = i + 1024 < n, h3 = i + 1536 < n;
  if (h0) v0 = __builtin_nontemporal_load(src + i);
  if (h1) v1 = __builtin_nontemporal_load(src + i + 512);
  if (h2) v2 = __builtin_nontemporal_load(src + i + 1024);
  if (h3) v3 = __builtin_nontemporal_load(src + i + 1536);
  if (h0) __builtin_nontemporal_store(v0, dst + i);
  if (h1) __builtin_nontemporal_store(v1, dst + i + 512);
  if (h2) __builtin_nontemporal_store(v2, dst + i + 1024);
  if (h3) __builtin_nontemporal_store(v3, dst + i + 1536);
}

}  // namespace mirl

// ===========================================================================
// C-ABI
// ===========================================================================
using namespace mirl;

namespace mirl {
std::string& last_error_ref() { static thread_local std::string e; return e; }
Profiler& profiler() { static Profiler p; return p; }
}

extern "C" int mirl_profile_set(int32_t level) { profiler().level = level; return MIRL_OK; }
extern "C" int mirl_profile_collect(int32_t* n_kernels) { profiler().collect(); if (n_kernels) *n_kernels = (int32_t)profiler().entries.size(); return MIRL_OK; }
extern "C" int mirl_profile_reset(void) { profiler().reset(); return MIRL_OK; }
extern "C" int mirl_profile_get(int32_t i, char* name_host, int32_t name_cap, int64_t* calls, double* total_ms, double* algorithmic_bytes) {
  Profiler& p = profiler();
  if (i < 0 || (size_t)i >= p.entries.size() || !name_host || name_cap <= 0) return fail(MIRL_ERR_ARG, "bad profile_get arguments");
  const ProfEntry& e = p.entries[(size_t)i];
  snprintf(name_host, (size_t)name_cap, "%s", e.name.c_str());
  if (calls) *calls = e.calls;
  if (total_ms) *total_ms = e.ms;
  if (algorithmic_bytes) *algorithmic_bytes = e.bytes;
  return MIRL_OK;
}
extern "C" int mirl_profile_get_flop(int32_t i, double* flop) {
  Profiler& p = profiler();
  if (i < 0 || (size_t)i >= p.entries.size() || !flop) return fail(MIRL_ERR_ARG, "bad profile_get_flop arguments");
  *flop = p.entries[(size_t)i].flop;
  return MIRL_OK;
}

struct mirl_replay {
  Book book;
  Dev d;
  StagingRing staging;
  Plan plan;
  int overlapped = 0, rows = 0;
  uint64_t epoch = 1, sample_calls = 0;
  double* w_tmp = nullptr; int64_t w_tmp_cap = 0;
  std::vector<void*> allocs;
  double* gpow_dev = nullptr;
  int32_t* bad_host = nullptr;
  int64_t* td_idx = nullptr; float* td_loss = nullptr; int td_cap = 0;   // acting-time priority scratch
  int gather_nt = 0;
  int gather_variant = 1, gather_order = 0;
  int recalc_wave = 1;
  int dedup_lds = 1;
  int prof = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
  // rollout plan (mirl_replay_ingest_plan / _planned): ONE device buffer at a fixed address — captured graphs hold it
  char* roll_plan = nullptr; size_t roll_cap = 0; int roll_steps = 0, roll_count = 0, roll_cap_steps = 0, roll_cap_count = 0;
  bool book_broken = false;            // a rollout plan failed after the book had moved: host and device disagree
  std::vector<char> roll_host;
};

extern "C" const char* mirl_last_error(void) { return last_error_ref().c_str(); }

extern "C" int mirl_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

template <class T>
static int dev_alloc(mirl_replay* h, T** p, size_t count, int fill_byte = 0) {
  size_t bytes = count * sizeof(T);
  if (!bytes) { *p = nullptr; return MIRL_OK; }
  hipError_t e = hipMalloc((void**)p, bytes);
  if (e != hipSuccess) return fail(MIRL_ERR_HIP, std::string("hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e));
  h->allocs.push_back((void*)*p);
  MIRL_HIP(hipMemset(*p, fill_byte, bytes));
  return MIRL_OK;
}

extern "C" int mirl_replay_destroy(mirl_replay* h) {
  if (!h) return MIRL_OK;
  (void)hipDeviceSynchronize();
  h->staging.destroy();
  if (h->bad_host) (void)hipHostFree(h->bad_host);
  for (void* p : h->allocs) (void)hipFree(p);
  delete h;
  return MIRL_OK;
}

extern "C" int mirl_replay_create(const mirl_replay_config* cfg, mirl_replay** out) {
  if (!cfg || !out) return fail(MIRL_ERR_ARG, "null argument");
  if (mirl_device_count() <= 0) return fail(MIRL_ERR_NOGPU, "no HIP device visible: librltime_hip needs an AMD GPU (there is no CPU fallback)");
  if (cfg->frame_bytes <= 0) return fail(MIRL_ERR_ARG, "frame_bytes must be > 0");
  if (cfg->stack_planes > 1 && (cfg->frame_bytes % (16 * cfg->stack_planes)))
    return fail(MIRL_ERR_ARG, "stack_planes: frame_bytes must be a multiple of 16 * stack_planes");
  if (cfg->acting_priority_init && (cfg->mode != MIRL_MODE_PER || cfg->policy_f32 <= 0))
    return fail(MIRL_ERR_ARG, "acting_priority_init needs prioritized replay and stored q-values (policy_f32 = number of actions)");
  MIRL_HIP(hipSetDevice(cfg->device));
  mirl_replay* h = new mirl_replay();
  int rc = h->book.init(*cfg);
  if (rc) { last_error_ref() = h->book.err; delete h; return rc; }
  rc = h->staging.init();
  if (rc) { delete h; return rc; }
  Book& bk = h->book;
  Dev& d = h->d;
  memset(&d, 0, sizeof(d));
  d.E = bk.E; d.F = cfg->frame_bytes; d.Fp = (int32_t)align_up((size_t)cfg->frame_bytes, 16);
  if (cfg->stack_planes > 1) {
    d.planes = cfg->stack_planes; d.plane_bytes = cfg->frame_bytes / cfg->stack_planes;
    d.Fp = d.plane_bytes;                                  // the ring keeps one plane per transition
  }
  d.X = cfg->extra_f32; d.S = cfg->state_f32; d.A = cfg->policy_f32; d.has_init = cfg->has_initials;
  d.per = bk.per; d.T = bk.T; d.P = bk.P; d.N = bk.N; d.L = bk.L; d.gap = bk.gap; d.env_base = cfg->env_base;
  d.avoid_xing = cfg->avoid_episode_crossing; d.C = bk.C; d.cap = bk.tree_cap; d.n_slots = bk.n_slots;
  d.alpha = cfg->alpha; d.mwf = cfg->max_weight_factor; d.eps = cfg->eps;
  int lg = 0; while ((1LL << lg) < d.cap) ++lg; d.log2cap = lg;
  const size_t slots = (size_t)d.E * (size_t)d.C;
#define TRY(x) do { int _r = (x); if (_r) { mirl_replay_destroy(h); return _r; } } while (0)
  TRY(dev_alloc(h, &d.frames, slots * (size_t)d.Fp));
  TRY(dev_alloc(h, &d.extra, slots * (size_t)d.X));
  TRY(dev_alloc(h, &d.state, slots * (size_t)d.S));
  if (d.has_init) TRY(dev_alloc(h, &d.initials, slots));
  TRY(dev_alloc(h, &d.actions, slots));
  TRY(dev_alloc(h, &d.policy, slots * (size_t)d.A));
  TRY(dev_alloc(h, &d.rewards, slots));
  TRY(dev_alloc(h, &d.dones, slots));
  if (d.planes) TRY(dev_alloc(h, &d.depth, slots));
  TRY(dev_alloc(h, &d.first, (size_t)d.E));
  TRY(dev_alloc(h, &d.count, (size_t)d.E));
  {
    std::vector<double> g((size_t)d.N + 1);                                        // [N] = gamma ** n for the acting-time TD
    for (int k = 0; k <= d.N; ++k) g[(size_t)k] = pow(cfg->gamma, (double)k);  // float.__pow__ -> libm pow
    TRY(dev_alloc(h, &h->gpow_dev, (size_t)d.N + 1));
    hipError_t e = hipMemcpy(h->gpow_dev, g.data(), g.size() * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) { mirl_replay_destroy(h); return fail(MIRL_ERR_HIP, hipGetErrorString(e)); }
    d.gpow = h->gpow_dev;
  }
  {
    hipError_t e = hipHostMalloc((void**)&h->bad_host, sizeof(int32_t), hipHostMallocDefault);
    if (e != hipSuccess) { mirl_replay_destroy(h); return fail(MIRL_ERR_HIP, hipGetErrorString(e)); }
    *h->bad_host = 0;
    d.bad = h->bad_host;
  }
  if (d.per) {
    TRY(dev_alloc(h, &d.loss, slots));
    TRY(dev_alloc(h, &d.prio_index, slots, 0xFF));
    TRY(dev_alloc(h, &d.stamp, slots));
    TRY(dev_alloc(h, &d.tv, (size_t)(2 * d.cap)));          // neutral 0.0 (python float) == all-zero bytes, kind KW == 0
    TRY(dev_alloc(h, &d.tk, (size_t)(2 * d.cap)));
    if (cfg->global_importance_scaling) {
      TRY(dev_alloc(h, &d.tmin, (size_t)(2 * d.cap)));
      std::vector<double> inf((size_t)(2 * d.cap), INFINITY);
      hipError_t e = hipMemcpy(d.tmin, inf.data(), inf.size() * sizeof(double), hipMemcpyHostToDevice);
      if (e != hipSuccess) { mirl_replay_destroy(h); return fail(MIRL_ERR_HIP, hipGetErrorString(e)); }
    }
    TRY(dev_alloc(h, &d.slot_env, (size_t)d.n_slots, 0xFF));
    TRY(dev_alloc(h, &d.slot_base, (size_t)d.n_slots, 0xFF));
    TRY(dev_alloc(h, &d.flag, (size_t)d.n_slots));
    TRY(dev_alloc(h, &d.dirty, (size_t)d.n_slots + 16));
    TRY(dev_alloc(h, &d.dirty_count, 1));
  }
#undef TRY
  // history.py:245-246: overlapped stacking iff nstep_target < L and every
  // sampled step has its full n-step; the latter can only fail when windows
  // are shifted to the ring end by avoid_episode_crossing.
  h->overlapped = (bk.N < bk.L && !cfg->avoid_episode_crossing) ? 1 : 0;
  h->rows = h->overlapped ? bk.L + bk.N : 2 * bk.L;
  const char* nt = getenv("MIRL_GATHER_NT");
  h->gather_nt = nt ? atoi(nt) : 1;
  if (const char* v = getenv("MIRL_GATHER_VARIANT")) h->gather_variant = atoi(v);
  if (const char* v = getenv("MIRL_GATHER_ORDER")) h->gather_order = atoi(v);
  if (const char* v = getenv("MIRL_RECALC_WAVE")) h->recalc_wave = atoi(v);
  if (const char* v = getenv("MIRL_DEDUP_LDS")) h->dedup_lds = atoi(v);
  MIRL_HIP(hipDeviceSynchronize());
  *out = h;
  return MIRL_OK;
}

static int scatter(mirl_replay* h, const void* src, void* ring, const int32_t* s_env, const int64_t* s_off,
                   int K, int32_t row_bytes, int64_t stride, hipStream_t st, int64_t src_stride = 0) {
  if (!row_bytes || !src) return MIRL_OK;
  if (!src_stride) src_stride = row_bytes;
  int vec = (row_bytes % 16 == 0) && (stride % 16 == 0) && (((uintptr_t)src) % 16 == 0) && (src_stride % 16 == 0);
  int chunks = vec ? row_bytes / 16 : row_bytes;
  int gx = (chunks + 255) / 256; if (gx > 64) gx = 64; if (gx < 1) gx = 1;
  ProfScope ps(row_bytes >= 8192 ? "k_scatter_rows(frames)" : "k_scatter_rows(small rows)", 2.0 * K * row_bytes, st);
  hipLaunchKernelGGL(k_scatter_rows, dim3(gx, K), dim3(256), 0, st, (const uint8_t*)src, (uint8_t*)ring, s_env, s_off,
                     h->d.C, row_bytes, stride, vec, src_stride);
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}

static int update_losses_impl(mirl_replay* h, int64_t count, const int64_t* indices, const float* losses, hipStream_t st);

// Reset observation of every env as "transition -1" of a de-duplicated shard (newest-plane ingest).
__global__ void __launch_bounds__(256)
k_prime_stack(Dev d, const uint8_t* __restrict__ planes, int64_t stride) {
  const int e = blockIdx.x;
  const int64_t slot = (int64_t)e * d.C + (d.C - 1);
  const u32x4* s4 = (const u32x4*)(planes + (int64_t)e * stride);
  u32x4* t4 = (u32x4*)(d.frames + slot * (int64_t)d.plane_bytes);
  for (int q = threadIdx.x; q < (d.plane_bytes >> 4); q += 256) t4[q] = s4[q];
  if (threadIdx.x == 0) { d.depth[slot] = 1; d.dones[slot] = 0; }
}

extern "C" int mirl_replay_prime_stack(mirl_replay* h, const uint8_t* newest_planes, int64_t stride, void* stream) {
  if (!h || !newest_planes) return fail(MIRL_ERR_ARG, "bad prime_stack arguments");
  Dev& d = h->d;
  if (!d.planes) return fail(MIRL_ERR_ARG, "prime_stack: the shard does not de-duplicate frame stacks");
  if (h->book.total_items() != 0) return fail(MIRL_ERR_STATE, "prime_stack: only before the first transition");
  if (stride <= 0) stride = d.plane_bytes;
  if (((uintptr_t)newest_planes % 16) || (stride % 16)) return fail(MIRL_ERR_ARG, "prime_stack: 16-byte aligned planes");
  hipLaunchKernelGGL(k_prime_stack, dim3(d.E), dim3(256), 0, (hipStream_t)stream, d, newest_planes, stride);
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}

static IngestSrc make_ingest_src(const Dev& d, const mirl_ingest* in) {
  const bool planes = d.planes != 0;
  const int64_t f_stride = in->frames_stride > 0 ? in->frames_stride : (planes ? d.plane_bytes : d.F);
  const int32_t f_row = planes ? d.plane_bytes : d.F, f_slot = planes ? d.plane_bytes : d.Fp;
  auto vec_ok = [](const void* src, const void* dst, int64_t row_bytes, int64_t src_stride, int64_t dst_stride) {
    return (int)(row_bytes && row_bytes % 16 == 0 && src_stride % 16 == 0 && dst_stride % 16 == 0 &&
                 ((uintptr_t)src) % 16 == 0 && ((uintptr_t)dst) % 16 == 0);
  };
  return IngestSrc{in->frames, in->extra, in->state, in->policy, in->initials, in->actions, in->rewards, in->dones,
                   vec_ok(in->frames, d.frames, f_row, f_stride, f_slot), vec_ok(in->extra, d.extra, d.X * 4, d.X * 4, d.X * 4),
                   vec_ok(in->state, d.state, d.S * 4, d.S * 4, d.S * 4), vec_ok(in->policy, d.policy, d.A * 4, d.A * 4, d.A * 4),
                   f_stride, f_row, f_slot};
}

static int g_ingest_fused = -1;     // -1: take MIRL_INGEST_FUSED (default on) at the first ingest
extern "C" int mirl_ingest_fused_set(int32_t on) { g_ingest_fused = on ? 1 : 0; return MIRL_OK; }

extern "C" int mirl_replay_ingest(mirl_replay* h, const mirl_ingest* in, void* stream) {
  if (h && h->book_broken) return fail(MIRL_ERR_STATE, "this shard's host bookkeeping ran ahead of its device rows (an ingest_plan failed half-way); the shard is unusable");
  if (!h || !in || in->count <= 0) return fail(MIRL_ERR_ARG, "bad ingest arguments");
  hipStream_t st = (hipStream_t)stream;
  Dev& d = h->d;
  const int K = in->count;
  if (!in->frames || !in->actions || !in->rewards || !in->dones) return fail(MIRL_ERR_ARG, "frames/actions/rewards/dones are required");
  if (K > 65535) return fail(MIRL_ERR_ARG, "at most 65535 transitions per ingest call (split the vector step)");
  if ((d.X && !in->extra) || (d.S && !in->state) || (d.has_init && !in->initials) || (d.A && !in->policy))
    return fail(MIRL_ERR_ARG, "a configured payload array is NULL");
  // every argument check sits AHEAD of the host bookkeeping: a refused call must leave the book, the staging
  // ring and the device rings exactly where they were
  if (d.planes && in->newest_plane_only && (h->book.cfg.acting_priority_init && d.per))
    return fail(MIRL_ERR_ARG, "newest_plane_only ingest cannot be combined with acting_priority_init");
  if (d.planes && !in->newest_plane_only && (((uintptr_t)in->frames) % 16))
    return fail(MIRL_ERR_ARG, "stack_planes: the frames array must be 16-byte aligned");
  int rc = h->book.ingest(K, in->env_ids_host, h->plan);
  if (rc) { last_error_ref() = h->book.err; return rc; }
  Plan& p = h->plan;
  // pack the plan: [sample_env i32][sample_off i64][table][env][leaf]
  size_t o_env = 0, o_off = align_up(o_env + sizeof(int32_t) * K, 16), o_tab = align_up(o_off + sizeof(int64_t) * K, 16);
  size_t o_eop = align_up(o_tab + sizeof(TableOp) * p.table_ops.size(), 16);
  size_t o_lop = align_up(o_eop + sizeof(EnvOp) * p.env_ops.size(), 16);
  size_t total = align_up(o_lop + sizeof(LeafOp) * p.leaf_ops.size(), 16);
  char *hb, *db;
  rc = h->staging.acquire(total, &hb, &db); if (rc) return rc;
  memcpy(hb + o_env, p.sample_env.data(), sizeof(int32_t) * K);
  memcpy(hb + o_off, p.sample_off.data(), sizeof(int64_t) * K);
  if (!p.table_ops.empty()) memcpy(hb + o_tab, p.table_ops.data(), sizeof(TableOp) * p.table_ops.size());
  if (!p.env_ops.empty()) memcpy(hb + o_eop, p.env_ops.data(), sizeof(EnvOp) * p.env_ops.size());
  if (!p.leaf_ops.empty()) memcpy(hb + o_lop, p.leaf_ops.data(), sizeof(LeafOp) * p.leaf_ops.size());
  rc = h->staging.upload(total, st); if (rc) return rc;
  const int32_t* s_env = (const int32_t*)(db + o_env);
  const int64_t* s_off = (const int64_t*)(db + o_off);
  if (g_ingest_fused < 0) g_ingest_fused = (getenv("MIRL_INGEST_FUSED") && atoi(getenv("MIRL_INGEST_FUSED")) == 0) ? 0 : 1;
  if ((g_ingest_fused && !d.planes && !(h->book.cfg.acting_priority_init && d.per)) || (d.planes && in->newest_plane_only)) {
    const IngestSrc src = make_ingest_src(d, in);
    const int32_t f_row = src.row_bytes;
    const int nt = d.per ? (int)p.table_ops.size() : 0, ne = (int)p.env_ops.size(), nl = d.per ? (int)p.leaf_ops.size() : 0;
    // 16 KB per copy workgroup: a (4, 84, 84) frame row takes 2
    int parts = (int)((f_row + 16383) / 16384); if (parts < 1) parts = 1; if (parts > 8) parts = 8;
    {
      ProfScope ps("k_ingest_fused", 2.0 * K * ((double)f_row + 4.0 * (d.X + d.S + d.A) + 13 + (d.per ? 16 : 0)), st);
      hipLaunchKernelGGL(k_ingest_fused, dim3(parts, K + 1), dim3(1024), 0, st, d, K, src, s_env, s_off, nt, (const TableOp*)(db + o_tab),
                         ne, (const EnvOp*)(db + o_eop), nl, (const LeafOp*)(db + o_lop));
    }
    MIRL_LAUNCH_CHECK();
    return h->staging.mark(st);
  }
  if (d.planes) {
    // de-dup: verify the stack-shift contract against the stored planes, record the
    // depth, keep the newest plane only
    {
      ProfScope ps("k_dedup_depth", (double)K * (2.0 * d.F - d.plane_bytes), st);
      hipLaunchKernelGGL(k_dedup_depth, dim3(K), dim3(256), 0, st, d, in->frames, s_env, s_off);
    }
    MIRL_LAUNCH_CHECK();
    rc = scatter(h, in->frames + (size_t)(d.planes - 1) * d.plane_bytes, d.frames, s_env, s_off, K, d.plane_bytes, d.plane_bytes, st, d.F);
    if (rc) return rc;
  } else {
    rc = scatter(h, in->frames, d.frames, s_env, s_off, K, d.F, d.Fp, st); if (rc) return rc;
  }
  rc = scatter(h, in->extra, d.extra, s_env, s_off, K, d.X * 4, (int64_t)d.X * 4, st); if (rc) return rc;
  rc = scatter(h, in->state, d.state, s_env, s_off, K, d.S * 4, (int64_t)d.S * 4, st); if (rc) return rc;
  rc = scatter(h, in->policy, d.policy, s_env, s_off, K, d.A * 4, (int64_t)d.A * 4, st); if (rc) return rc;
  {
    ProfScope ps("k_ingest_scalars", 2.0 * K * (13 + (d.per ? 16 : 0)), st);
    hipLaunchKernelGGL(k_ingest_scalars, dim3((K + 255) / 256), dim3(256), 0, st, d, K, s_env, s_off,
                       in->initials, in->actions, in->rewards, in->dones);
  }
  MIRL_LAUNCH_CHECK();
  int nt = (int)p.table_ops.size(), ne = (int)p.env_ops.size(), nl = d.per ? (int)p.leaf_ops.size() : 0;
  if (!d.per) nt = 0;
  int nmax = nt > ne ? nt : ne; nmax = nl > nmax ? nl : nmax;
  if (nmax) {
    ProfScope ps("k_plan_apply", 0.0, st);
    hipLaunchKernelGGL(k_plan_apply, dim3((nmax + 255) / 256), dim3(256), 0, st, d, nt, (const TableOp*)(db + o_tab),
                       ne, (const EnvOp*)(db + o_eop), nl, (const LeafOp*)(db + o_lop));
    MIRL_LAUNCH_CHECK();
  }
  if (nl) { ProfScope ps("k_tree_fix(ingest)", 0.0, st); hipLaunchKernelGGL(k_tree_fix, dim3(1), dim3(1024), 0, st, d); MIRL_LAUNCH_CHECK(); }
  if (h->book.cfg.acting_priority_init && d.per) {
    // TD errors of the transitions whose n-step target just became available, from
    // stored q-values, through the update_losses path (see mirl_replay_config)
    if (h->td_cap < K) {
      int cap = 256; while (cap < K) cap *= 2;
      int64_t* pi = nullptr; float* pl = nullptr;
      MIRL_HIP(hipMalloc((void**)&pi, sizeof(int64_t) * 2 * (size_t)cap));
      MIRL_HIP(hipMalloc((void**)&pl, sizeof(float) * (size_t)cap));
      h->allocs.push_back(pi); h->allocs.push_back(pl);
      h->td_idx = pi; h->td_loss = pl; h->td_cap = cap;
    }
    {
      ProfScope ps("k_acting_td", (double)K * (d.N * 5.0 + 2.0 * d.A * 4 + 28), st);
      hipLaunchKernelGGL(k_acting_td, dim3((K + 255) / 256), dim3(256), 0, st, d, K, s_env, s_off, h->book.cfg.acting_vf_eps,
                         h->td_idx, h->td_loss);
    }
    MIRL_LAUNCH_CHECK();
    rc = update_losses_impl(h, K, h->td_idx, h->td_loss, st); if (rc) return rc;
  }
  return h->staging.mark(st);
}

// ---- rollout plan: the host bookkeeping of `steps` vector steps ahead of their device side ------------------
// Everything the ingest decides on the host is data-independent (ring heads, global-FIFO eviction, sequence
// activation / deactivation, the FIFO free list: history.py:123-176, prioritized_replay_history.py:136-172,210-230),
// so the plans of a whole acting rollout can be made before its first kernel runs.  mirl_replay_ingest_plan advances
// the book by `steps` vector steps of `count` transitions, packs their op lists behind a table of PlanHdr into
// pinned memory and copies them (stream-ordered) into the shard's rollout-plan buffer, whose device address never
// changes; mirl_replay_ingest_planned enqueues the device side of ONE of those steps with no host argument that
// varies from rollout to rollout — the call a HIP graph of the whole rollout captures (acting/fast_step.py).
static bool planned_ingest_ok(mirl_replay* h) {
  if (g_ingest_fused < 0) g_ingest_fused = (getenv("MIRL_INGEST_FUSED") && atoi(getenv("MIRL_INGEST_FUSED")) == 0) ? 0 : 1;
  return g_ingest_fused && !h->d.planes && !(h->book.cfg.acting_priority_init && h->d.per);
}

extern "C" int mirl_replay_ingest_plan(mirl_replay* h, int32_t steps, int32_t count, const int32_t* env_ids_host, void* stream) {
  if (!h || steps <= 0 || steps > 4096 || count <= 0 || count > 65535) return fail(MIRL_ERR_ARG, "bad ingest_plan arguments");
  if (h->book_broken) return fail(MIRL_ERR_STATE, "this shard's host bookkeeping ran ahead of its device rows (an earlier ingest_plan failed half-way); the shard is unusable");
  if (!planned_ingest_ok(h)) return fail(MIRL_ERR_ARG, "planned ingest needs the fused ingest kernel (no de-duplicated storage, no acting_priority_init)");
  hipStream_t st = (hipStream_t)stream;
  const int K = count;
  // Transactional: EVERYTHING that can refuse the call — arguments, buffer sizes, the staging block — is settled before
  // the book moves, so a refused call (MIRL_ERR_ARG / MIRL_ERR_HIP from here) leaves the book, the rings and the plan
  // buffer exactly where they were and the caller may fall back to per-step ingest.  A failure AFTER the book has moved
  // marks the shard broken (every later ingest / sample call returns MIRL_ERR_STATE): the host would otherwise count
  // transitions whose device rows were never written.
  // worst case per transition: table ops <= 2 (one deactivation, one activation), leaf ops <= 2, env ops <= 2
  auto per_step_bytes = [](int k) -> size_t {
    return align_up(sizeof(int32_t) * (size_t)k, 16) + align_up(sizeof(int64_t) * (size_t)k, 16) +
           align_up(2 * (size_t)k * sizeof(TableOp), 16) + align_up(2 * (size_t)k * sizeof(EnvOp), 16) +
           align_up(2 * (size_t)k * sizeof(LeafOp), 16) + 64;
  };
  if (h->roll_plan && (steps > h->roll_cap_steps || K > h->roll_cap_count))
    return fail(MIRL_ERR_ARG, "ingest_plan: more steps / transitions per step than the rollout-plan buffer was sized for at its first call");
  for (int32_t k = 0; env_ids_host && k < K; ++k) {
    const int32_t e = env_ids_host[k] - h->book.cfg.env_base;
    if (e < 0 || e >= h->book.E) return fail(MIRL_ERR_ARG, "env id outside this shard");
    for (int32_t j = 0; j < k; ++j) if (env_ids_host[j] == env_ids_host[k]) return fail(MIRL_ERR_ARG, "an env may appear once per ingest call (split the vector steps)");
  }
  if (!env_ids_host && K > h->book.E) return fail(MIRL_ERR_ARG, "env id outside this shard");
  if (!h->roll_plan) {
    // sized once (the address is baked into captured graphs): at least 64 steps of this call's transition count
    const int cap_steps = steps > 64 ? steps : 64;
    const size_t cap = align_up(sizeof(PlanHdr) * (size_t)cap_steps, 256) + per_step_bytes(K) * (size_t)cap_steps;
    char* buf = nullptr;
    MIRL_HIP(hipMalloc((void**)&buf, cap));
    h->roll_plan = buf; h->roll_cap = cap;
    h->allocs.push_back(h->roll_plan);
    h->roll_cap_steps = cap_steps; h->roll_cap_count = K;
  }
  const size_t hdr_bytes = align_up(sizeof(PlanHdr) * (size_t)h->roll_cap_steps, 256);
  const size_t worst = hdr_bytes + per_step_bytes(K) * (size_t)steps;      // <= roll_cap by construction
  char *hb, *db;
  int rc = h->staging.acquire(worst, &hb, &db); if (rc) return rc;
  std::vector<char>& buf = h->roll_host;
  buf.assign(hdr_bytes, 0);
  auto put = [&buf](const void* src, size_t bytes) -> int64_t {
    const size_t at = align_up(buf.size(), 16);
    buf.resize(at + bytes);
    if (bytes) memcpy(buf.data() + at, src, bytes);
    return (int64_t)at;
  };
  // whatever failed, the caller sees MIRL_ERR_STATE from here on (never MIRL_ERR_ARG, which means "refused before
  // anything moved, fall back to per-step ingest"): the message of the original failure stays in mirl_last_error()
  auto broken = [h](int) -> int { h->book_broken = true; return MIRL_ERR_STATE; };
  for (int s = 0; s < steps; ++s) {
    rc = h->book.ingest(K, env_ids_host, h->plan);
    if (rc) { last_error_ref() = h->book.err; return broken(rc); }     // a reference assert (ring overflow, no free index): fatal anyway
    const Plan& p = h->plan;
    PlanHdr hd;
    hd.n_table = h->d.per ? (int32_t)p.table_ops.size() : 0;
    hd.n_env = (int32_t)p.env_ops.size();
    hd.n_leaf = h->d.per ? (int32_t)p.leaf_ops.size() : 0;
    hd.pad = 0;
    hd.o_env = put(p.sample_env.data(), sizeof(int32_t) * K);
    hd.o_off = put(p.sample_off.data(), sizeof(int64_t) * K);
    hd.o_tab = put(p.table_ops.data(), sizeof(TableOp) * p.table_ops.size());
    hd.o_eop = put(p.env_ops.data(), sizeof(EnvOp) * p.env_ops.size());
    hd.o_lop = put(p.leaf_ops.data(), sizeof(LeafOp) * p.leaf_ops.size());
    memcpy(buf.data() + sizeof(PlanHdr) * (size_t)s, &hd, sizeof(PlanHdr));
  }
  const size_t total = align_up(buf.size(), 16);
  if (total > worst || total > h->roll_cap) return broken(fail(MIRL_ERR_STATE, "ingest_plan: the rollout's op lists exceed their worst-case bound (library bug)"));
  memcpy(hb, buf.data(), buf.size());
  rc = h->staging.upload(total, st); if (rc) return broken(rc);
  if (hipMemcpyAsync(h->roll_plan, db, total, hipMemcpyDeviceToDevice, st) != hipSuccess)     // stream order: after the last rollout that read it
    return broken(fail(MIRL_ERR_HIP, "ingest_plan: copy into the rollout-plan buffer failed"));
  h->roll_steps = steps; h->roll_count = K;
  rc = h->staging.mark(st);
  return rc ? broken(rc) : MIRL_OK;
}

extern "C" int mirl_replay_ingest_planned(mirl_replay* h, int32_t step, const mirl_ingest* in, void* stream) {
  if (h && h->book_broken) return fail(MIRL_ERR_STATE, "this shard's host bookkeeping ran ahead of its device rows (an ingest_plan failed half-way); the shard is unusable");
  if (!h || !in || in->count <= 0 || step < 0) return fail(MIRL_ERR_ARG, "bad ingest_planned arguments");
  if (!h->roll_plan || step >= h->roll_cap_steps) return fail(MIRL_ERR_STATE, "ingest_planned: no rollout plan covers this step (call mirl_replay_ingest_plan first)");
  if (!planned_ingest_ok(h) || in->newest_plane_only) return fail(MIRL_ERR_ARG, "planned ingest needs the fused ingest kernel (no de-duplicated storage, no acting_priority_init)");
  Dev& d = h->d;
  const int K = in->count;
  if (K != h->roll_count) return fail(MIRL_ERR_ARG, "ingest_planned: transition count differs from the plan's");
  if (!in->frames || !in->actions || !in->rewards || !in->dones) return fail(MIRL_ERR_ARG, "frames/actions/rewards/dones are required");
  if ((d.X && !in->extra) || (d.S && !in->state) || (d.has_init && !in->initials) || (d.A && !in->policy))
    return fail(MIRL_ERR_ARG, "a configured payload array is NULL");
  hipStream_t st = (hipStream_t)stream;
  const IngestSrc src = make_ingest_src(d, in);
  int parts = (int)((src.row_bytes + 16383) / 16384); if (parts < 1) parts = 1; if (parts > 8) parts = 8;
  ProfScope ps("k_ingest_fused", 2.0 * K * ((double)src.row_bytes + 4.0 * (d.X + d.S + d.A) + 13 + (d.per ? 16 : 0)), st);
  hipLaunchKernelGGL(k_ingest_fused_planned, dim3(parts, K + 1), dim3(1024), 0, st, d, K, src, (const char*)h->roll_plan, (int)step);
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}

extern "C" int mirl_replay_needed_feed_count(mirl_replay* h, int32_t mbatch, int32_t num_envs, int64_t* out) {
  if (!h || !out) return fail(MIRL_ERR_ARG, "null argument");
  *out = h->book.needed_feed_count(mbatch, num_envs);
  return MIRL_OK;
}

extern "C" int mirl_replay_set_train_quota(mirl_replay* h, int64_t quota) {
  if (!h) return fail(MIRL_ERR_ARG, "null handle");
  h->book.quota = quota;
  return MIRL_OK;
}

extern "C" int mirl_replay_uniform_total(mirl_replay* h, int64_t* total) {
  if (!h || !total) return fail(MIRL_ERR_ARG, "null argument");
  *total = h->book.uniform_total();
  return MIRL_OK;
}

extern "C" int mirl_replay_state_rows(mirl_replay* h, int32_t* rows, int32_t* overlapped) {
  if (!h) return fail(MIRL_ERR_ARG, "null argument");
  if (rows) *rows = h->rows;
  if (overlapped) *overlapped = h->overlapped;
  return MIRL_OK;
}

static double anneal_beta(const mirl_replay_config& c, double progress) {
  // general/utils.py:85-103 anneal_value(beta, progress, beta_anneal, 1.0)
  if (progress > 1.0) progress = 1.0;
  if (c.beta_anneal_mode == 0) return c.beta;
  double target = c.beta_anneal_mode == 1 ? 1.0 : c.beta_anneal_to;
  return c.beta + (target - c.beta) * progress;
}

extern "C" int mirl_replay_sample(mirl_replay* h, int32_t B, double train_progress, const void* rng_host, uint64_t seed,
                                  int32_t* slot, int32_t* env, int64_t* start, int64_t* loss_start, float* weight,
                                  double* stats, void* stream) {
  if (h && h->book_broken) return fail(MIRL_ERR_STATE, "this shard's host bookkeeping ran ahead of its device rows (an ingest_plan failed half-way); the shard is unusable");
  if (!h || B <= 0 || !slot || !env || !start || !weight) return fail(MIRL_ERR_ARG, "bad sample arguments");
  hipStream_t st = (hipStream_t)stream;
  Book& bk = h->book;
  Dev& d = h->d;
  if (h->bad_host && *h->bad_host == 2) {
    *h->bad_host = 0;
    return fail(MIRL_ERR_STATE, "stack_planes: an ingested frame violates the frame-stack shift contract (env_wrappers/common.py:141-178): its older planes are neither the previous frames nor a reset's zero fill");
  }
  if (h->bad_host && *h->bad_host) {
    *h->bad_host = 0;
    return fail(MIRL_ERR_STATE, "an earlier sample call drew an inactive tree leaf (prioritized_replay_history.py:306 asserts base_sample is not None); those rows were given weight 0 and no loss index");
  }
  int rc = bk.charge_quota(B);                       // replay_history.py:176-181 (even when None is returned)
  if (rc) { last_error_ref() = bk.err; return rc; }
  ++h->sample_calls;
  if (d.per) {
    if (train_progress < 0) return fail(MIRL_ERR_ARG, "train_progress must be >= 0 (general/utils.py:97)");
    if (bk.active < B) {                             // prioritized_replay_history.py:295-299
      if (bk.total_items() >= bk.cfg.size) return fail(MIRL_ERR_STATE, "buffer full but fewer sequences than mbatch (prioritized_replay_history.py:298)");
      return MIRL_NEED_MORE;
    }
    const double* u_dev = nullptr;
    if (rng_host) {
      char *hb, *db;
      rc = h->staging.acquire(sizeof(double) * B, &hb, &db); if (rc) return rc;
      memcpy(hb, rng_host, sizeof(double) * B);
      rc = h->staging.upload(sizeof(double) * B, st); if (rc) return rc;
      u_dev = (const double*)db;
    }
    if (h->w_tmp_cap < B) {
      double* p = nullptr;
      MIRL_HIP(hipMalloc((void**)&p, sizeof(double) * (size_t)B));
      h->allocs.push_back(p); h->w_tmp = p; h->w_tmp_cap = B;
    }
    double beta = anneal_beta(bk.cfg, train_progress);
    {
      ProfScope ps("k_per_sample", (double)B * d.log2cap * 9.0, st);
      hipLaunchKernelGGL(k_per_sample, dim3(1), dim3(1024), 0, st, d, (int)B, u_dev, seed, h->sample_calls,
                         (double)bk.active, beta, bk.cfg.global_importance_scaling, slot, env, start, loss_start, weight, h->w_tmp, stats);
    }
    MIRL_LAUNCH_CHECK();
    if (rng_host) return h->staging.mark(st);
    return MIRL_OK;
  }
  // uniform
  std::vector<int64_t> cum; std::vector<int32_t> envs;
  int64_t tot = 0;
  for (int32_t e : bk.env_order) { int64_t a = (bk.count[e] - bk.first[e]) - (bk.L + bk.N - 1); if (a > 0) { tot += a; cum.push_back(tot); envs.push_back(e); } }
  if (tot < B) {                                       // replay_history.py:110-114
    if (bk.total_items() >= bk.cfg.size) return fail(MIRL_ERR_STATE, "buffer full but not enough start positions (replay_history.py:113)");
    return MIRL_NEED_MORE;
  }
  size_t nc = cum.size();
  size_t o_cum = 0, o_env = align_up(o_cum + sizeof(int64_t) * nc, 16), o_pick = align_up(o_env + sizeof(int32_t) * nc, 16);
  size_t total = o_pick + (rng_host ? sizeof(int64_t) * (size_t)B : 0);
  char *hb, *db;
  rc = h->staging.acquire(total, &hb, &db); if (rc) return rc;
  memcpy(hb + o_cum, cum.data(), sizeof(int64_t) * nc);
  memcpy(hb + o_env, envs.data(), sizeof(int32_t) * nc);
  if (rng_host) {
    const int64_t* picks = (const int64_t*)rng_host;
    for (int i = 0; i < B; ++i) if (picks[i] < 0 || picks[i] >= tot) return fail(MIRL_ERR_ARG, "uniform pick out of range");
    memcpy(hb + o_pick, picks, sizeof(int64_t) * (size_t)B);
  }
  rc = h->staging.upload(total, st); if (rc) return rc;
  {
    ProfScope ps("k_uniform_sample", 0.0, st);
    hipLaunchKernelGGL(k_uniform_sample, dim3((B + 255) / 256), dim3(256), 0, st, d, (int)B,
                       rng_host ? (const int64_t*)(db + o_pick) : (const int64_t*)nullptr, seed, h->sample_calls,
                       (int)nc, (const int64_t*)(db + o_cum), (const int32_t*)(db + o_env), slot, env, start, loss_start, weight);
  }
  MIRL_LAUNCH_CHECK();
  return h->staging.mark(st);
}

extern "C" int mirl_replay_sample_ready(mirl_replay* h, int32_t mbatch, int32_t* ready) {
  if (!h || mbatch <= 0 || !ready) return fail(MIRL_ERR_ARG, "bad sample_ready arguments");
  const Book& bk = h->book;
  *ready = bk.per ? (bk.active >= mbatch) : (bk.uniform_total() >= mbatch);
  return MIRL_OK;
}

extern "C" int mirl_replay_sample_skip(mirl_replay* h, int32_t mbatch) {
  if (!h || mbatch <= 0) return fail(MIRL_ERR_ARG, "bad sample_skip arguments");
  int rc = h->book.charge_quota(mbatch);              // replay_history.py:176-181: charged even when None is returned
  if (rc) { last_error_ref() = h->book.err; return rc; }
  ++h->sample_calls;                                  // the Philox step counter advances like in a NEED_MORE return
  return MIRL_OK;
}

extern "C" int mirl_replay_tree_root(mirl_replay* h, double* root_dev, void* stream) {
  if (!h || !h->d.per || !root_dev) return fail(MIRL_ERR_ARG, "bad tree_root arguments");
  hipLaunchKernelGGL(k_tree_root, dim3(1), dim3(1), 0, (hipStream_t)stream, h->d, (double)h->book.active, root_dev);
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}

extern "C" int mirl_replay_sample_global(mirl_replay* h, int32_t mbatch_local, int32_t mbatch_global, int32_t rows, int32_t rank,
                                         int32_t world, const double* shard_totals, double train_progress, uint64_t seed,
                                         int32_t* slot, int32_t* env, int64_t* start, int64_t* loss_start, double* weight_raw,
                                         int32_t* stratum, double* stats, void* stream) {
  if (!h || !h->d.per || mbatch_local <= 0 || mbatch_global <= 0 || rows <= 0 || rank < 0 || rank >= world || !shard_totals ||
      !slot || !env || !start || !weight_raw || !stratum || !stats)
    return fail(MIRL_ERR_ARG, "bad sample_global arguments");
  if (train_progress < 0) return fail(MIRL_ERR_ARG, "train_progress must be >= 0 (general/utils.py:97)");
  hipStream_t st = (hipStream_t)stream;
  Book& bk = h->book;
  if (h->bad_host && *h->bad_host) { *h->bad_host = 0; return fail(MIRL_ERR_STATE, "an earlier call flagged an invalid sample or frame (see mirl_replay_sample)"); }
  int rc = bk.charge_quota(mbatch_local);
  if (rc) { last_error_ref() = bk.err; return rc; }
  ++h->sample_calls;
  if (bk.active < mbatch_local) {
    if (bk.total_items() >= bk.cfg.size) return fail(MIRL_ERR_STATE, "buffer full but fewer sequences than mbatch (prioritized_replay_history.py:298)");
    return MIRL_NEED_MORE;
  }
  const double beta = anneal_beta(bk.cfg, train_progress);
  ProfScope ps("k_per_sample_global", (double)mbatch_global * h->d.log2cap * 9.0 / world, st);
  hipLaunchKernelGGL(k_per_sample_global, dim3(1), dim3(1024), 0, st, h->d, (int)mbatch_global, (int)rows, (int)rank, (int)world,
                     shard_totals, seed, h->sample_calls, beta, slot, env, start, loss_start, weight_raw, stratum, stats);
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}

static int gather_leaf(mirl_replay* h, const void* ring, void* out, const int32_t* env, const int64_t* start,
                       int B, int32_t row_bytes, int64_t ring_stride, hipStream_t st) {
  if (!row_bytes || !out) return MIRL_OK;
  int vec = (row_bytes % 16 == 0) && (ring_stride % 16 == 0) && (((uintptr_t)out) % 16 == 0);
  int64_t blocks = (int64_t)h->rows * B;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  const bool timed = h->prof && ring == (const void*)h->d.frames;
  if (timed) { MIRL_HIP(hipEventCreate(&e0)); MIRL_HIP(hipEventCreate(&e1)); MIRL_HIP(hipEventRecord(e0, st)); }
  {
  ProfScope ps(ring == (const void*)h->d.frames ? "k_gather_rows(frames)" : (ring == (const void*)h->d.state ? "k_gather_rows(recurrent state)" : "k_gather_rows(extra)"),
               2.0 * (double)blocks * row_bytes, st);
  if (vec && h->gather_variant == 1 && row_bytes >= 512 * 16) {    // small rows (recurrent state) keep the 256-lane shape
    if (h->gather_nt == 2)                                          // cached loads, non-temporal stores
      hipLaunchKernelGGL(k_gather_rows_v1<0>, dim3((unsigned)blocks), dim3(512), 0, st, h->d, (const uint8_t*)ring, (uint8_t*)out,
                         env, start, B, h->overlapped, row_bytes, ring_stride, h->gather_order);
    else
      hipLaunchKernelGGL(k_gather_rows_v1<1>, dim3((unsigned)blocks), dim3(512), 0, st, h->d, (const uint8_t*)ring, (uint8_t*)out,
                         env, start, B, h->overlapped, row_bytes, ring_stride, h->gather_order);
  }
  else if (h->gather_nt)
    hipLaunchKernelGGL(k_gather_rows<1>, dim3((unsigned)blocks), dim3(256), 0, st, h->d, (const uint8_t*)ring, (uint8_t*)out,
                       env, start, B, h->overlapped, row_bytes, ring_stride, vec);
  else
    hipLaunchKernelGGL(k_gather_rows<0>, dim3((unsigned)blocks), dim3(256), 0, st, h->d, (const uint8_t*)ring, (uint8_t*)out,
                       env, start, B, h->overlapped, row_bytes, ring_stride, vec);
  }
  MIRL_LAUNCH_CHECK();
  if (timed) { MIRL_HIP(hipEventRecord(e1, st)); h->prof_events.push_back(std::make_pair(e0, e1)); }
  return MIRL_OK;
}

extern "C" int mirl_replay_profile(mirl_replay* h, int32_t enable, int64_t* launches, double* total_ms) {
  if (!h) return fail(MIRL_ERR_ARG, "null handle");
  MIRL_HIP(hipDeviceSynchronize());
  double tot = 0.0; int64_t n = 0;
  for (auto& pr : h->prof_events) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) { tot += ms; ++n; }
    (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second);
  }
  h->prof_events.clear();
  if (launches) *launches = n;
  if (total_ms) *total_ms = tot;
  h->prof = enable;
  return MIRL_OK;
}

// ---- snapshot / resume ------------------------------------------------------------
namespace {
const uint64_t kSnapMagic = 0x4D49524C534E4150ull;   // "MIRLSNAP"
const uint32_t kSnapVersion = 2;

struct SnapIO {
  FILE* f = nullptr; char* pin = nullptr; size_t pin_bytes = 64u << 20; bool ok = true;
  bool open(const char* path, const char* mode) {
    f = fopen(path, mode);
    if (!f) return false;
    if (hipHostMalloc((void**)&pin, pin_bytes, hipHostMallocDefault) != hipSuccess) { fclose(f); f = nullptr; return false; }
    return true;
  }
  void close() { if (f) fclose(f); if (pin) (void)hipHostFree(pin); f = nullptr; pin = nullptr; }
  void put(const void* p, size_t n) { if (ok && n && fwrite(p, 1, n, f) != n) ok = false; }
  void get(void* p, size_t n) { if (ok && n && fread(p, 1, n, f) != n) ok = false; }
  template <class T> void put_vec(const std::vector<T>& v) { uint64_t n = v.size(); put(&n, 8); put(v.data(), n * sizeof(T)); }
  template <class T> void get_vec(std::vector<T>& v) { uint64_t n = 0; get(&n, 8); if (!ok || n > (1ull << 40)) { ok = false; return; } v.resize((size_t)n); get(v.data(), n * sizeof(T)); }
  void put_dev(const void* d, size_t n) {          // device -> file, section prefixed by its size
    uint64_t sz = d ? n : 0; put(&sz, 8);
    for (size_t at = 0; ok && at < sz; at += pin_bytes) {
      size_t c = sz - at < pin_bytes ? sz - at : pin_bytes;
      if (hipMemcpy(pin, (const char*)d + at, c, hipMemcpyDeviceToHost) != hipSuccess) { ok = false; break; }
      put(pin, c);
    }
  }
  void get_dev(void* d, size_t n) {                // file -> device; sizes must agree
    uint64_t sz = 0; get(&sz, 8);
    if (!ok || sz != (d ? n : 0)) { ok = false; return; }
    for (size_t at = 0; ok && at < sz; at += pin_bytes) {
      size_t c = sz - at < pin_bytes ? sz - at : pin_bytes;
      get(pin, c);
      if (ok && hipMemcpy((char*)d + at, pin, c, hipMemcpyHostToDevice) != hipSuccess) ok = false;
    }
  }
};

bool same_config(const mirl_replay_config& a, const mirl_replay_config& b) {
  return a.size == b.size && a.num_envs == b.num_envs && a.env_base == b.env_base && a.frame_bytes == b.frame_bytes &&
         a.extra_f32 == b.extra_f32 && a.state_f32 == b.state_f32 && a.has_initials == b.has_initials &&
         a.policy_f32 == b.policy_f32 && a.nstep_train == b.nstep_train && a.prefix_steps == b.prefix_steps &&
         a.nstep_target == b.nstep_target && a.gamma == b.gamma && a.mode == b.mode &&
         a.train_frequency == b.train_frequency && a.avoid_episode_crossing == b.avoid_episode_crossing &&
         a.overlap == b.overlap && a.alpha == b.alpha && a.beta == b.beta && a.eps == b.eps &&
         a.max_weight_factor == b.max_weight_factor && a.beta_anneal_mode == b.beta_anneal_mode &&
         a.beta_anneal_to == b.beta_anneal_to && a.global_importance_scaling == b.global_importance_scaling &&
         a.env_ring_slack == b.env_ring_slack && a.acting_priority_init == b.acting_priority_init &&
         a.acting_vf_eps == b.acting_vf_eps && a.stack_planes == b.stack_planes;
}

template <class F>
void snap_device_arrays(mirl_replay* h, F&& io) {
  Dev& d = h->d;
  const size_t slots = (size_t)d.E * (size_t)d.C;
  io(d.frames, slots * (size_t)d.Fp);
  io(d.extra, slots * (size_t)d.X * 4);
  io(d.state, slots * (size_t)d.S * 4);
  io(d.initials, d.has_init ? slots * 4 : 0);
  io(d.actions, slots * 4);
  io(d.policy, slots * (size_t)d.A * 4);
  io(d.rewards, slots * 4);
  io(d.dones, slots);
  io(d.depth, d.planes ? slots : 0);
  io(d.first, (size_t)d.E * 8);
  io(d.count, (size_t)d.E * 8);
  if (d.per) {
    io(d.loss, slots * 4);
    io(d.prio_index, slots * 4);
    io(d.stamp, slots * 8);
    io(d.tv, (size_t)(2 * d.cap) * 8);
    io(d.tk, (size_t)(2 * d.cap));
    io(d.tmin, d.tmin ? (size_t)(2 * d.cap) * 8 : 0);
    io(d.slot_env, (size_t)d.n_slots * 4);
    io(d.slot_base, (size_t)d.n_slots * 8);
  }
}
}  // namespace

extern "C" int mirl_replay_save(mirl_replay* h, const char* path) {
  if (!h || !path) return fail(MIRL_ERR_ARG, "null argument");
  MIRL_HIP(hipDeviceSynchronize());
  SnapIO io;
  if (!io.open(path, "wb")) return fail(MIRL_ERR_ARG, std::string("cannot open snapshot for writing: ") + path);
  Book& b = h->book;
  io.put(&kSnapMagic, 8); io.put(&kSnapVersion, 4);
  io.put(&b.cfg, sizeof(b.cfg));
  int64_t scal[4] = {b.quota, b.active, (int64_t)h->epoch, (int64_t)h->sample_calls};
  io.put(scal, sizeof(scal));
  io.put_vec(b.first); io.put_vec(b.count); io.put_vec(b.env_order); io.put_vec(b.env_seen);
  std::vector<int32_t> tmp;
  b.fifo.dump(tmp); io.put_vec(tmp);
  b.free_slots.dump(tmp); io.put_vec(tmp);
  io.put_vec(b.slot_env); io.put_vec(b.slot_base); io.put_vec(b.prio_index);
  snap_device_arrays(h, [&](const void* p, size_t n) { io.put_dev(p, n); });
  bool ok = io.ok;
  io.close();
  return ok ? MIRL_OK : fail(MIRL_ERR_HIP, "snapshot write failed");
}

extern "C" int mirl_replay_load(mirl_replay* h, const char* path) {
  if (!h || !path) return fail(MIRL_ERR_ARG, "null argument");
  MIRL_HIP(hipDeviceSynchronize());
  SnapIO io;
  if (!io.open(path, "rb")) return fail(MIRL_ERR_ARG, std::string("cannot open snapshot: ") + path);
  Book& b = h->book;
  uint64_t magic = 0; uint32_t ver = 0; mirl_replay_config cfg;
  io.get(&magic, 8); io.get(&ver, 4); io.get(&cfg, sizeof(cfg));
  if (!io.ok || magic != kSnapMagic || ver != kSnapVersion || !same_config(cfg, b.cfg)) {   // device ordinal may differ
    io.close();
    return fail(MIRL_ERR_ARG, "snapshot does not match this handle's configuration (or is not a snapshot)");
  }
  int64_t scal[4];
  io.get(scal, sizeof(scal));
  io.get_vec(b.first); io.get_vec(b.count); io.get_vec(b.env_order); io.get_vec(b.env_seen);
  std::vector<int32_t> tmp;
  io.get_vec(tmp); bool r1 = b.fifo.restore(tmp);
  io.get_vec(tmp); bool r2 = b.free_slots.restore(tmp);
  io.get_vec(b.slot_env); io.get_vec(b.slot_base); io.get_vec(b.prio_index);
  if (io.ok && r1 && r2) {
    b.quota = scal[0]; b.active = scal[1]; h->epoch = (uint64_t)scal[2]; h->sample_calls = (uint64_t)scal[3];
    snap_device_arrays(h, [&](void* p, size_t n) { io.get_dev(p, n); });
    if (h->d.per) (void)hipMemset(h->d.flag, 0, (size_t)h->d.n_slots), (void)hipMemset(h->d.dirty_count, 0, 4);
  }
  bool ok = io.ok && r1 && r2;
  io.close();
  MIRL_HIP(hipDeviceSynchronize());
  return ok ? MIRL_OK : fail(MIRL_ERR_STATE, "snapshot is truncated or inconsistent; the handle must be recreated");
}

extern "C" int mirl_replay_gather(mirl_replay* h, int32_t B, const int32_t* env, const int64_t* start, const int64_t* loss_start,
                                  const float* weight, const mirl_batch* out, void* stream) {
  if (!h || B <= 0 || !env || !start || !out) return fail(MIRL_ERR_ARG, "bad gather arguments");
  if (!out->frames || !out->returns || !out->nsteps || !out->masks || !out->actions) return fail(MIRL_ERR_ARG, "frames/returns/nsteps/masks/actions outputs are required");
  if (out->weights && !weight) return fail(MIRL_ERR_ARG, "weights output requested without a weight input");
  hipStream_t st = (hipStream_t)stream;
  Dev& d = h->d;
  int rc;
  if (d.planes) {
    if (((uintptr_t)out->frames) % 16) return fail(MIRL_ERR_ARG, "stack_planes: the frames output must be 16-byte aligned");
    const int64_t blocks = (int64_t)h->rows * B;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->prof) { MIRL_HIP(hipEventCreate(&e0)); MIRL_HIP(hipEventCreate(&e1)); MIRL_HIP(hipEventRecord(e0, st)); }
    {
      // algorithmic bytes: every output stack written once + every distinct plane of a
      // window read once (rows + P - 1 planes per sequence and state block)
      ProfScope ps("k_gather_rows_dedup(frames)", (double)blocks * d.F + (double)B * (h->rows + d.planes - 1) * d.plane_bytes, st);
      // LDS-staged variant: as many planes as fit in 64 KB; RC rows per workgroup
      static const int lds_kb = getenv("MIRL_DEDUP_LDS_KB") ? atoi(getenv("MIRL_DEDUP_LDS_KB")) : 64;
      int lds_planes = (int)((size_t)lds_kb * 1024 / d.plane_bytes); if (lds_planes > MIRL_DD_MAX_PLANES) lds_planes = MIRL_DD_MAX_PLANES;
      const int RC = lds_planes - (d.planes - 1);
      static bool lds_attr = false;
      if (!lds_attr && (size_t)lds_planes * d.plane_bytes > 65536) {
        MIRL_HIP(hipFuncSetAttribute((const void*)k_gather_rows_dedup_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        lds_attr = true;
      }
      if (h->dedup_lds && RC >= 2) {
        const int chunks = (h->rows + RC - 1) / RC;
        hipLaunchKernelGGL(k_gather_rows_dedup_lds, dim3((unsigned)(chunks * B)), dim3(512), (size_t)lds_planes * d.plane_bytes, st, d,
                           out->frames, env, start, B, h->rows, h->overlapped, RC, lds_planes);
      } else
      hipLaunchKernelGGL(k_gather_rows_dedup, dim3((unsigned)blocks), dim3(512), 0, st, d, out->frames, env, start, B, h->overlapped);
    }
    MIRL_LAUNCH_CHECK();
    if (h->prof) { MIRL_HIP(hipEventRecord(e1, st)); h->prof_events.push_back(std::make_pair(e0, e1)); }
  } else {
    rc = gather_leaf(h, d.frames, out->frames, env, start, B, d.F, d.Fp, st); if (rc) return rc;
  }
  rc = gather_leaf(h, d.extra, out->extra, env, start, B, d.X * 4, (int64_t)d.X * 4, st); if (rc) return rc;
  rc = gather_leaf(h, d.state, out->state, env, start, B, d.S * 4, (int64_t)d.S * 4, st); if (rc) return rc;
  mirl_batch o = *out;
  if (!d.has_init) o.initials = nullptr;
  if (!d.A) o.policy = nullptr;
  int rows = h->rows > d.L ? h->rows : d.L;
  int64_t n = (int64_t)rows * B;
  {
    ProfScope ps("k_gather_scalars", (double)d.L * B * (5.0 * d.N + 24.0 + (o.weights ? 4 : 0) + (o.loss_indices ? 16 : 0) + 8.0 * d.A) + (double)rows * B * 8.0, st);
    hipLaunchKernelGGL(k_gather_scalars, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d, (int)B, h->rows, h->overlapped,
                       env, start, loss_start, weight, o);
  }
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}

static int update_losses_impl(mirl_replay* h, int64_t count, const int64_t* indices, const float* losses, hipStream_t st);

extern "C" int mirl_replay_update_losses(mirl_replay* h, int64_t count, const int64_t* indices, const float* losses, void* stream) {
  if (!h) return fail(MIRL_ERR_ARG, "null handle");
  if (!h->d.per || count <= 0) return MIRL_OK;           // history.py:332-335 no-op for non-prioritized buffers
  if (!indices || !losses) return fail(MIRL_ERR_ARG, "null indices/losses");
  if (count >= (1LL << 32)) return fail(MIRL_ERR_ARG, "too many loss rows");
  return update_losses_impl(h, count, indices, losses, (hipStream_t)stream);
}

static int update_losses_impl(mirl_replay* h, int64_t count, const int64_t* indices, const float* losses, hipStream_t st) {
  Dev& d = h->d;
  uint64_t epoch = h->epoch++;
  unsigned g = (unsigned)((count + 255) / 256);
  { ProfScope ps("k_loss_stamp", (double)count * 24.0, st);
    hipLaunchKernelGGL(k_loss_stamp, dim3(g), dim3(256), 0, st, d, count, indices, epoch); }
  MIRL_LAUNCH_CHECK();
  { ProfScope ps("k_loss_write", (double)count * 32.0, st);
    hipLaunchKernelGGL(k_loss_write, dim3(g), dim3(256), 0, st, d, count, indices, losses, epoch); }
  MIRL_LAUNCH_CHECK();
  if (d.T >= 8 && d.T <= 128 && h->recalc_wave) {
    ProfScope ps("k_recalc_flagged_wave", (double)d.n_slots + (double)count * 4.0 * 2, st);
    hipLaunchKernelGGL(k_recalc_flagged_wave, dim3((unsigned)((d.n_slots + 3) / 4)), dim3(256), 0, st, d);
  } else {
    ProfScope ps("k_recalc_flagged", (double)d.n_slots + (double)count * 4.0 * 2, st);
    hipLaunchKernelGGL(k_recalc_flagged, dim3((unsigned)((d.n_slots + 255) / 256)), dim3(256), 0, st, d);
  }
  MIRL_LAUNCH_CHECK();
  { ProfScope ps("k_tree_fix(update_losses)", 0.0, st);
    hipLaunchKernelGGL(k_tree_fix, dim3(1), dim3(1024), 0, st, d); }
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}

// ---- introspection ---------------------------------------------------------
extern "C" int mirl_replay_stats(mirl_replay* h, int64_t* total_items, int64_t* active, int64_t* quota, int64_t* cap, int64_t* n_slots) {
  if (!h) return fail(MIRL_ERR_ARG, "null handle");
  if (total_items) *total_items = h->book.total_items();
  if (active) *active = h->book.active;
  if (quota) *quota = h->book.quota;
  if (cap) *cap = h->book.tree_cap;
  if (n_slots) *n_slots = h->book.n_slots;
  return MIRL_OK;
}
extern "C" int mirl_replay_env_meta(mirl_replay* h, int64_t* first_host, int64_t* count_host) {
  if (!h) return fail(MIRL_ERR_ARG, "null handle");
  // read back the DEVICE mirrors so tests see what the kernels see
  MIRL_HIP(hipDeviceSynchronize());
  if (first_host) MIRL_HIP(hipMemcpy(first_host, h->d.first, sizeof(int64_t) * h->d.E, hipMemcpyDeviceToHost));
  if (count_host) MIRL_HIP(hipMemcpy(count_host, h->d.count, sizeof(int64_t) * h->d.E, hipMemcpyDeviceToHost));
  return MIRL_OK;
}
extern "C" int mirl_replay_free_slots(mirl_replay* h, int32_t* slots_host, int64_t* n) {
  if (!h || !n) return fail(MIRL_ERR_ARG, "null argument");
  *n = h->book.free_slots.size();
  if (slots_host) for (int64_t i = 0; i < *n; ++i) slots_host[i] = h->book.free_slots.at(i);
  return MIRL_OK;
}
extern "C" int mirl_replay_slot_table(mirl_replay* h, int32_t* slot_env_host, int64_t* slot_base_host) {
  if (!h || !h->d.per) return fail(MIRL_ERR_ARG, "not a prioritized replay");
  MIRL_HIP(hipDeviceSynchronize());
  if (slot_env_host) MIRL_HIP(hipMemcpy(slot_env_host, h->d.slot_env, sizeof(int32_t) * h->d.n_slots, hipMemcpyDeviceToHost));
  if (slot_base_host) MIRL_HIP(hipMemcpy(slot_base_host, h->d.slot_base, sizeof(int64_t) * h->d.n_slots, hipMemcpyDeviceToHost));
  return MIRL_OK;
}
extern "C" int mirl_replay_tree_nodes(mirl_replay* h, double* value_host, uint8_t* kind_host, double* min_host) {
  if (!h || !h->d.per) return fail(MIRL_ERR_ARG, "not a prioritized replay");
  MIRL_HIP(hipDeviceSynchronize());
  size_t n = (size_t)(2 * h->d.cap);
  if (value_host) MIRL_HIP(hipMemcpy(value_host, h->d.tv, sizeof(double) * n, hipMemcpyDeviceToHost));
  if (kind_host) MIRL_HIP(hipMemcpy(kind_host, h->d.tk, n, hipMemcpyDeviceToHost));
  if (min_host && h->d.tmin) MIRL_HIP(hipMemcpy(min_host, h->d.tmin, sizeof(double) * n, hipMemcpyDeviceToHost));
  return MIRL_OK;
}
extern "C" int mirl_replay_tree_set_leaves(mirl_replay* h, int64_t n, const double* value_host, const uint8_t* kind_host, void* stream) {
  if (!h || !h->d.per || n < 0 || n > h->d.cap) return fail(MIRL_ERR_ARG, "bad tree_set_leaves arguments");
  hipStream_t st = (hipStream_t)stream;
  size_t o_k = align_up(sizeof(double) * (size_t)n, 16);
  char *hb, *db;
  int rc = h->staging.acquire(o_k + (size_t)n, &hb, &db); if (rc) return rc;
  memcpy(hb, value_host, sizeof(double) * (size_t)n);
  memcpy(hb + o_k, kind_host, (size_t)n);
  rc = h->staging.upload(o_k + (size_t)n, st); if (rc) return rc;
  if (n) { hipLaunchKernelGGL(k_set_leaves, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, h->d, n, (const double*)db, (const uint8_t*)(db + o_k)); MIRL_LAUNCH_CHECK(); }
  for (int lvl = h->d.log2cap - 1; lvl >= 0; --lvl) {
    int64_t lo = 1LL << lvl, hi = 2LL << lvl;
    hipLaunchKernelGGL(k_tree_level, dim3((unsigned)((hi - lo + 255) / 256)), dim3(256), 0, st, h->d, lo, hi);
    MIRL_LAUNCH_CHECK();
  }
  return h->staging.mark(st);
}
extern "C" int mirl_replay_tree_find(mirl_replay* h, int32_t B, const double* uniforms_host, int64_t* idx_host, void* stream) {
  if (!h || !h->d.per || B <= 0) return fail(MIRL_ERR_ARG, "bad tree_find arguments");
  hipStream_t st = (hipStream_t)stream;
  size_t o_i = align_up(sizeof(double) * (size_t)B, 16);
  char *hb, *db;
  int rc = h->staging.acquire(o_i + sizeof(int64_t) * (size_t)B, &hb, &db); if (rc) return rc;
  memcpy(hb, uniforms_host, sizeof(double) * (size_t)B);
  rc = h->staging.upload(sizeof(double) * (size_t)B, st); if (rc) return rc;
  hipLaunchKernelGGL(k_tree_find, dim3((B + 255) / 256), dim3(256), 0, st, h->d, (int)B, (const double*)db, (int64_t*)(db + o_i));
  MIRL_LAUNCH_CHECK();
  MIRL_HIP(hipMemcpyAsync(idx_host, db + o_i, sizeof(int64_t) * (size_t)B, hipMemcpyDeviceToHost, st));
  MIRL_HIP(hipStreamSynchronize(st));
  return h->staging.mark(st);
}
extern "C" int mirl_replay_losses_peek(mirl_replay* h, int32_t env_local, int64_t offset, int32_t n, float* out_host) {
  if (!h || !h->d.per || env_local < 0 || env_local >= h->d.E || n <= 0) return fail(MIRL_ERR_ARG, "bad losses_peek arguments");
  MIRL_HIP(hipDeviceSynchronize());
  for (int i = 0; i < n; ++i) {
    int64_t sl = (int64_t)env_local * h->d.C + (offset + i) % h->d.C;
    MIRL_HIP(hipMemcpy(out_host + i, h->d.loss + sl, sizeof(float), hipMemcpyDeviceToHost));
  }
  return MIRL_OK;
}

// experiment variants of the plain copy (tools/copy_probe.py: where do the ~10 % between this box's 5.7 TB/s and the
// guide's 6.29 TB/s float4 copy go?): PER 16-byte vectors per lane, all loads issued before the first store;
// NTL / NTS: non-temporal loads / stores
template <int PER, bool NTL, bool NTS>
__global__ void __launch_bounds__(256) k_copy16_v(u32x4* __restrict__ dst, const u32x4* __restrict__ src, int64_t n) {
  const int64_t base = (int64_t)blockIdx.x * (256 * PER) + threadIdx.x;
  u32x4 v[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int64_t i = base + (int64_t)k * 256;
    if (i < n) v[k] = NTL ? __builtin_nontemporal_load(src + i) : src[i];
  }
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int64_t i = base + (int64_t)k * 256;
    if (i < n) { if (NTS) __builtin_nontemporal_store(v[k], dst + i); else dst[i] = v[k]; }
  }
}

extern "C" int mirl_copy_bytes_ex(void* dst, const void* src, int64_t bytes, int32_t nt, void* stream) {
  if (!dst || !src || bytes <= 0 || (bytes % 16) || ((uintptr_t)dst % 16) || ((uintptr_t)src % 16)) return fail(MIRL_ERR_ARG, "copy needs 16-byte aligned pointers and size");
  const int64_t n = bytes / 16;
  hipStream_t cs = (hipStream_t)stream;
  if (nt >= 2) {
    const unsigned g8 = (unsigned)((n + 256 * 8 - 1) / (256 * 8)), g4 = (unsigned)((n + 256 * 4 - 1) / (256 * 4)), g2 = (unsigned)((n + 256 * 2 - 1) / (256 * 2));
    switch (nt) {
      case 2: hipLaunchKernelGGL((k_copy16_v<8, true, true>), dim3(g8), dim3(256), 0, cs, (u32x4*)dst, (const u32x4*)src, n); break;
      case 3: MIRL_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, cs)); return MIRL_OK;
      case 4: hipLaunchKernelGGL((k_copy16_v<4, false, true>), dim3(g4), dim3(256), 0, cs, (u32x4*)dst, (const u32x4*)src, n); break;
      case 5: hipLaunchKernelGGL((k_copy16_v<4, false, false>), dim3(g4), dim3(256), 0, cs, (u32x4*)dst, (const u32x4*)src, n); break;
      case 6: hipLaunchKernelGGL((k_copy16_v<2, true, true>), dim3(g2), dim3(256), 0, cs, (u32x4*)dst, (const u32x4*)src, n); break;
      case 7: hipLaunchKernelGGL((k_copy16_v<8, false, false>), dim3(g8), dim3(256), 0, cs, (u32x4*)dst, (const u32x4*)src, n); break;
      default: return fail(MIRL_ERR_ARG, "copy: unknown variant");
    }
    MIRL_LAUNCH_CHECK();
    return MIRL_OK;
  }
  if (nt) hipLaunchKernelGGL(k_copy16_nt, dim3((unsigned)((n + 2047) / 2048)), dim3(512), 0, (hipStream_t)stream, (u32x4*)dst, (const u32x4*)src, n);
  else hipLaunchKernelGGL(k_copy16, dim3(2048), dim3(256), 0, (hipStream_t)stream, (u32x4*)dst, (const u32x4*)src, n);
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}

extern "C" int mirl_copy_bytes(void* dst, const void* src, int64_t bytes, void* stream) {
  static int nt = getenv("MIRL_COPY_NT") ? atoi(getenv("MIRL_COPY_NT")) : 0;
  return mirl_copy_bytes_ex(dst, src, bytes, nt, stream);
}

// ---- host-only hooks ---------------------------------------------------------
struct mirl_book { Book book; Plan plan; };

extern "C" int mirl_book_create(const mirl_replay_config* cfg, mirl_book** out) {
  if (!cfg || !out) return fail(MIRL_ERR_ARG, "null argument");
  mirl_book* b = new mirl_book();
  int rc = b->book.init(*cfg);
  if (rc) { last_error_ref() = b->book.err; delete b; return rc; }
  *out = b;
  return MIRL_OK;
}
extern "C" int mirl_book_destroy(mirl_book* b) { delete b; return MIRL_OK; }
extern "C" int mirl_book_ingest(mirl_book* b, int32_t count, const int32_t* env_ids_host) {
  if (!b) return fail(MIRL_ERR_ARG, "null handle");
  int rc = b->book.ingest(count, env_ids_host, b->plan);
  if (rc) last_error_ref() = b->book.err;
  return rc;
}
extern "C" int mirl_book_stats(mirl_book* b, int64_t* total_items, int64_t* active, int64_t* quota, int64_t* cap, int64_t* n_slots) {
  if (!b) return fail(MIRL_ERR_ARG, "null handle");
  if (total_items) *total_items = b->book.total_items();
  if (active) *active = b->book.active;
  if (quota) *quota = b->book.quota;
  if (cap) *cap = b->book.tree_cap;
  if (n_slots) *n_slots = b->book.n_slots;
  return MIRL_OK;
}
extern "C" int mirl_book_env_meta(mirl_book* b, int64_t* first_host, int64_t* count_host) {
  if (!b) return fail(MIRL_ERR_ARG, "null handle");
  for (int e = 0; e < b->book.E; ++e) { if (first_host) first_host[e] = b->book.first[e]; if (count_host) count_host[e] = b->book.count[e]; }
  return MIRL_OK;
}
extern "C" int mirl_book_free_slots(mirl_book* b, int32_t* slots_host, int64_t* n) {
  if (!b || !n) return fail(MIRL_ERR_ARG, "null argument");
  *n = b->book.free_slots.size();
  if (slots_host) for (int64_t i = 0; i < *n; ++i) slots_host[i] = b->book.free_slots.at(i);
  return MIRL_OK;
}
extern "C" int mirl_book_slot_table(mirl_book* b, int32_t* slot_env_host, int64_t* slot_base_host) {
  if (!b) return fail(MIRL_ERR_ARG, "null handle");
  for (int64_t i = 0; i < b->book.n_slots; ++i) { if (slot_env_host) slot_env_host[i] = b->book.slot_env[(size_t)i]; if (slot_base_host) slot_base_host[i] = b->book.slot_base[(size_t)i]; }
  return MIRL_OK;
}
extern "C" int mirl_book_needed_feed_count(mirl_book* b, int32_t mbatch, int32_t num_envs, int64_t* out) {
  if (!b || !out) return fail(MIRL_ERR_ARG, "null argument");
  *out = b->book.needed_feed_count(mbatch, num_envs);
  return MIRL_OK;
}
extern "C" int mirl_book_charge_quota(mirl_book* b, int32_t mbatch) {
  if (!b) return fail(MIRL_ERR_ARG, "null handle");
  int rc = b->book.charge_quota(mbatch);
  if (rc) last_error_ref() = b->book.err;
  return rc;
}
extern "C" int mirl_book_uniform_total(mirl_book* b, int64_t* total) {
  if (!b || !total) return fail(MIRL_ERR_ARG, "null argument");
  *total = b->book.uniform_total();
  return MIRL_OK;
}
extern "C" int mirl_book_uniform_map(mirl_book* b, int32_t mbatch, const int64_t* picks_host, int32_t* env_host, int64_t* start_host) {
  if (!b) return fail(MIRL_ERR_ARG, "null handle");
  return b->book.uniform_map(mbatch, picks_host, env_host, start_host);
}

extern "C" int mirl_emul_build_tree(int64_t cap, const double* lv, const uint8_t* lk, double* nv, uint8_t* nk) {
  for (int64_t i = 0; i < cap; ++i) { nv[cap + i] = lv[i]; nk[cap + i] = lk[i]; }
  nv[0] = 0; nk[0] = 0;
  for (int64_t node = cap - 1; node >= 1; --node) {
    TV a{nv[2 * node], nk[2 * node]}, b{nv[2 * node + 1], nk[2 * node + 1]};
    TV r = tadd(a, b);
    nv[node] = r.v; nk[node] = r.k;
  }
  return MIRL_OK;
}
extern "C" int mirl_emul_find(int64_t cap, const double* nv, const uint8_t* nk, int32_t B, const double* u, int64_t* idx) {
  TV total{nv[1], nk[1]};
  for (int i = 0; i < B; ++i) idx[i] = tagged_descend(nv, nk, cap, stratum_mass(total, B, i, u[i]));
  return MIRL_OK;
}
extern "C" int mirl_emul_seq_priority(int32_t T, double alpha, double mwf, const float* loss_slots, double* value, uint8_t* kind) {
  PrioParams p{T, alpha, mwf};
  auto g = [&](int t) -> float { return loss_slots[t]; };
  TV r = seq_priority(g, p);
  *value = r.v; *kind = r.k;
  return MIRL_OK;
}
