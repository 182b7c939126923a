// book.hpp — host-side bookkeeping of one replay shard (pure C++, no HIP).
//
// Everything about a replay buffer that does NOT depend on device data is
// decided here, deterministically, from the sequence of ingested env ids:
// per-env ring heads, the global FIFO eviction order, the train quota, and for
// prioritized replay the sequence activation / deactivation and the FIFO free
// list of tree indices.  Each ingest produces a *plan* — flat op lists that one
// small kernel applies on the device — so the data path never round-trips to
// the host.
//
// Reference semantics followed (paths relative to the reference root):
//   rltime/history/history.py:123-176          History.update
//   rltime/history/replay_history.py:62-91     quota, global-FIFO eviction
//   rltime/history/replay_history.py:93-134    uniform choice -> (env, start)
//   rltime/history/prioritized_replay_history.py:97-172,210-230
#pragma once
#include <stdint.h>
#include <string>
#include <vector>
#include <climits>

#include "../../include/mirl.h"

namespace mirl {

struct LeafOp {      // final state of one tree leaf after a plan
  int32_t slot;
  int32_t activate;  // 1: bind to (env, base) and compute its priority; 0: clear (leaf := 0, min := +inf)
  int32_t env;       // local env index
  int32_t pad;
  int64_t base;      // absolute per-env offset of the sequence head
};
struct TableOp {     // prio_index[env][off % C] := value
  int32_t env;
  int32_t value;
  int64_t off;
};
struct EnvOp {       // device mirror of first/count for one env
  int32_t env;
  int32_t pad;
  int64_t first;
  int64_t count;
};
struct Plan {
  std::vector<int32_t> sample_env;   // local env per ingested transition
  std::vector<int64_t> sample_off;   // its absolute per-env offset
  std::vector<LeafOp> leaf_ops;      // coalesced: at most one per slot
  std::vector<TableOp> table_ops;
  std::vector<EnvOp> env_ops;        // coalesced: at most one per env
  void clear() { sample_env.clear(); sample_off.clear(); leaf_ops.clear(); table_ops.clear(); env_ops.clear(); }
};

template <class T>
class FifoRing {     // fixed-capacity FIFO with index access
 public:
  void init(int64_t cap) { buf_.assign((size_t)(cap > 0 ? cap : 1), T()); cap_ = cap > 0 ? cap : 1; head_ = 0; len_ = 0; }
  int64_t size() const { return len_; }
  bool push(T v) { if (len_ == cap_) return false; buf_[(size_t)((head_ + len_) % cap_)] = v; ++len_; return true; }
  T pop() { T v = buf_[(size_t)head_]; head_ = (head_ + 1) % cap_; --len_; return v; }
  T at(int64_t i) const { return buf_[(size_t)((head_ + i) % cap_)]; }
  // flat (FIFO-order) export / import for snapshots
  void dump(std::vector<T>& out) const { out.resize((size_t)len_); for (int64_t i = 0; i < len_; ++i) out[(size_t)i] = at(i); }
  bool restore(const std::vector<T>& in) { if ((int64_t)in.size() > cap_) return false; head_ = 0; len_ = 0; for (T v : in) push(v); return true; }
 private:
  std::vector<T> buf_;
  int64_t cap_ = 1, head_ = 0, len_ = 0;
};

class Book {
 public:
  mirl_replay_config cfg;
  int32_t E = 0, T = 1, P = 0, N = 1, L = 1, gap = 1, overlap = 0;
  int64_t C = 0;            // ring slots per env
  int64_t extra_keep = 0;   // slots behind the oldest live transition that must stay intact (frame de-dup)
  int64_t n_slots = 0;      // prioritized_replay_history.py:109 target_capacity
  int64_t tree_cap = 1;     // :110-112
  bool per = false;
  int64_t quota = 0;
  int64_t active = 0;
  std::vector<int64_t> first, count;       // per local env: first live offset / next offset
  std::vector<int32_t> env_order;          // local envs in first-seen order (dict order, replay_history.py:100)
  std::vector<uint8_t> env_seen;
  FifoRing<int32_t> fifo;                  // replay_history.py:55 linear_history (env of each live sample)
  FifoRing<int32_t> free_slots;            // prioritized_replay_history.py:123-125
  std::vector<int32_t> slot_env;           // :129 _index_data (env part), -1 = free
  std::vector<int64_t> slot_base;
  std::vector<int32_t> prio_index;         // [E][C] 'prioritization_index' of each live transition, -1 = none
  std::string err;

  int init(const mirl_replay_config& c) {
    cfg = c;
    E = c.num_envs; T = c.nstep_train; P = c.prefix_steps; N = c.nstep_target;
    if (E <= 0 || T <= 0 || P < 0 || N <= 0 || c.size <= 0) { err = "bad replay config"; return MIRL_ERR_ARG; }
    L = T + P;
    per = c.mode == MIRL_MODE_PER;
    int64_t per_env = (c.size + E - 1) / E;
    // de-duplicated frame storage rebuilds a stack from the P-1 predecessors' planes:
    // they must outlive their (logically evicted) transitions
    extra_keep = c.stack_planes > 1 ? c.stack_planes - 1 : 0;
    C = per_env + 1 + (c.env_ring_slack > 0 ? c.env_ring_slack : 0) + extra_keep;
    first.assign(E, 0); count.assign(E, 0);
    env_seen.assign(E, 0); env_order.clear();
    fifo.init(c.size);
    quota = 0; active = 0;
    if (per) {
      // prioritized_replay_history.py:97-105
      int64_t ov = c.overlap;
      if (c.overlap == INT32_MIN) ov = T / 2;
      else if (ov < 0) { ov = T + ov; if (ov < 0) { err = "overlap < -nstep_train"; return MIRL_ERR_ARG; } }
      if (ov >= T) { err = "Overlap must be < nstep_train"; return MIRL_ERR_ARG; }
      overlap = (int32_t)ov; gap = T - overlap;
      n_slots = c.size / gap;                       // :109 int(size / gap)
      tree_cap = 1; while (tree_cap < n_slots) tree_cap *= 2;   // :110-112
      free_slots.init(n_slots);
      for (int64_t i = 0; i < n_slots; ++i) free_slots.push((int32_t)i);
      slot_env.assign((size_t)n_slots, -1); slot_base.assign((size_t)n_slots, -1);
      prio_index.assign((size_t)(E * C), -1);
    }
    return MIRL_OK;
  }

  int64_t total_items() const { return fifo.size(); }

  // One call of History.update with `n` samples (history.py:132-175), in order.
  int ingest(int32_t n, const int32_t* env_ids /* global ids or NULL */, Plan& plan) {
    plan.clear();
    std::vector<int32_t> leaf_last;  // op index per slot (coalescing), lazily sized
    std::vector<int32_t> touched_env;
    std::vector<uint8_t> in_call(E, 0);
    for (int32_t k = 0; k < n; ++k) {
      int32_t e = env_ids ? env_ids[k] - cfg.env_base : k;
      if (e < 0 || e >= E) { err = "env id outside this shard"; return MIRL_ERR_ARG; }
      if (in_call[e]) { err = "an env may appear once per ingest call (split the vector steps)"; return MIRL_ERR_ARG; }
      in_call[e] = 1;
    }
    for (int32_t k = 0; k < n; ++k) {
      int32_t e = env_ids ? env_ids[k] - cfg.env_base : k;
      if (!env_seen[e]) { env_seen[e] = 1; env_order.push_back(e); }
      int64_t off = count[e]++;                       // history.py:171 append
      if (per) prio_index[(size_t)(e * C + off % C)] = -1;
      plan.sample_env.push_back(e);
      plan.sample_off.push_back(off);
      touched_env.push_back(e);
      // replay_history.py:79-87: evict the globally oldest transition
      if (fifo.size() >= cfg.size) {
        int32_t v = fifo.pop();
        if (per) {
          // prioritized_replay_history.py:216-227: the sample at ring position P
          int64_t probe = first[v] + P;
          if (probe >= count[v]) { err = "replay too small: ring shorter than prefix_steps at eviction"; return MIRL_ERR_STATE; }
          int32_t& pi = prio_index[(size_t)(v * C + probe % C)];
          if (probe % gap == 0 && pi >= 0) {
            int32_t slot = pi;
            pi = -1;
            plan.table_ops.push_back(TableOp{v, -1, probe});
            push_leaf(plan, leaf_last, LeafOp{slot, 0, v, 0, probe});
            free_slots.push(slot);
            slot_env[(size_t)slot] = -1; slot_base[(size_t)slot] = -1;
            --active;
          }
        }
        ++first[v];                                    // :229-230 / history.py:121
        touched_env.push_back(v);
      }
      fifo.push(e);                                    // replay_history.py:89
      if (cfg.train_frequency) quota += cfg.train_frequency;   // :90-91
      if (count[e] - first[e] + 1 + extra_keep > C) { err = "per-env ring overflow (raise env_ring_slack: envs are not fed in lock-step)"; return MIRL_ERR_STATE; }
      if (per) {
        // prioritized_replay_history.py:143-172
        int64_t f = first[e];
        int64_t base = off - T + 1 - N + 1;
        if (base >= 0 && base % gap == 0 && base >= f + P) {
          if (free_slots.size() == 0) { err = "no free prioritization index"; return MIRL_ERR_STATE; }
          int32_t slot = free_slots.pop();
          slot_env[(size_t)slot] = e; slot_base[(size_t)slot] = base;
          prio_index[(size_t)(e * C + base % C)] = slot;
          plan.table_ops.push_back(TableOp{e, slot, base});
          push_leaf(plan, leaf_last, LeafOp{slot, 1, e, 0, base});
          ++active;
        }
      }
    }
    // coalesced env mirror ops
    std::vector<uint8_t> done(E, 0);
    for (int32_t e : touched_env) if (!done[e]) { done[e] = 1; plan.env_ops.push_back(EnvOp{e, 0, first[e], count[e]}); }
    return MIRL_OK;
  }

  // replay_history.py:62-75
  int64_t needed_feed_count(int32_t /*mbatch*/, int32_t num_envs) const {
    if (!cfg.train_frequency) return 0;
    if (quota > 0) return -1;
    int64_t need = (int64_t)((double)(-quota) / (double)cfg.train_frequency);  // int(-q / f): truncation
    return need > num_envs ? need : num_envs;
  }

  // replay_history.py:176-181.  Returns MIRL_ERR_STATE when an assert would fire.
  int charge_quota(int32_t mbatch) {
    if (cfg.train_frequency) {
      quota -= (int64_t)mbatch * T;
      int64_t lim = 100LL * mbatch * T;
      if (!(quota < lim) || !(quota > -lim)) { err = "train/act ratio drifted by >100 batches (replay_history.py:179-181)"; return MIRL_ERR_STATE; }
    }
    return MIRL_OK;
  }

  // replay_history.py:98-107
  int64_t uniform_total() const {
    int64_t tot = 0;
    for (int32_t e : env_order) { int64_t a = (count[e] - first[e]) - (L + N - 1); if (a > 0) tot += a; }
    return tot;
  }

  // replay_history.py:120-134: flat choice -> (env, ring position) by a scan over
  // the envs in first-seen order; returned start is the ABSOLUTE offset.
  int uniform_map(int32_t mbatch, const int64_t* picks, int32_t* env_out, int64_t* start_out) const {
    std::vector<int64_t> cum; std::vector<int32_t> envs;
    int64_t tot = 0;
    for (int32_t e : env_order) { int64_t a = (count[e] - first[e]) - (L + N - 1); if (a > 0) { tot += a; cum.push_back(tot); envs.push_back(e); } }
    for (int32_t i = 0; i < mbatch; ++i) {
      int64_t p = picks[i];
      if (p < 0 || p >= tot) return MIRL_ERR_ARG;
      size_t lo = 0, hi = cum.size() - 1;
      while (lo < hi) { size_t mid = (lo + hi) / 2; if (p < cum[mid]) hi = mid; else lo = mid + 1; }
      int64_t before = lo ? cum[lo - 1] : 0;
      env_out[i] = envs[lo];
      start_out[i] = first[envs[lo]] + (p - before);
    }
    return MIRL_OK;
  }

 private:
  static void push_leaf(Plan& plan, std::vector<int32_t>& last, const LeafOp& op) {
    if ((size_t)op.slot >= last.size()) last.resize((size_t)op.slot + 1, -1);
    int32_t& at = last[(size_t)op.slot];
    if (at >= 0) plan.leaf_ops[(size_t)at] = op;   // later op on the same leaf wins
    else { at = (int32_t)plan.leaf_ops.size(); plan.leaf_ops.push_back(op); }
  }
};

}  // namespace mirl
