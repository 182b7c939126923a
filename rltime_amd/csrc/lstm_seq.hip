// lstm_seq.hip — the recurrent core as ONE persistent launch per sequence sweep
// (reference time loop rltime/models/torch/modules/lstm.py:83-116).
//
// The reference runs T sequential LSTMCell steps; rounds 1-2 ran a step as one
// rocBLAS GEMM (12.6 us: every batch tile re-reads the 4 MB W_hh) + one cell kernel
// (5.7 us) + two launch gaps = 19.8 us, 400 times per learner step.  Here a whole
// sweep (all T steps of all B sequences) is one launch:
//
//   * workgroup (cluster c, column group cg) owns 64 batch rows x (16 hidden units x
//     4 gates); its 64 x H slice of W_hh is loaded into LDS ONCE (132 KB at H = 512)
//     and stays there for all T steps — W_hh generates no L2 / HBM traffic in the loop;
//   * wave w of the workgroup owns the 16-row tile 4*rb + w: per step 4 gate tiles x
//     H/4 k-steps of v_mfma_f32_16x16x4_f32 (exact f32, 512 MFMAs = 6.8 us at H = 512),
//     A operand = its 16 x H rows of h(t-1) in registers (H/4 VGPRs), B operand =
//     16 B LDS reads (ds_read_b128, conflict-free: row pitch H+4 floats, k split as
//     k = (H/4) kg + 4 q + j so that a lane's four j are contiguous);
//   * the four accumulators of a lane are the i, f, g, o pre-activations of ONE
//     (row, hidden unit): the cell runs in registers, c(t) never leaves them;
//   * a step depends only on the H/16 waves (one per column group) that share a
//     16-row tile: they exchange h(t) through a 2-deep buffer in the MFMA operand
//     layout with 16 B write-through (sc1) stores and sc1 loads — no fence, no grid
//     barrier, nothing depends on where a workgroup runs (cdna_hip_programming.md
//     Guideline 16).  THE DATA IS THE FLAG: |h| <= 1, so bit 30 of its float (the top
//     exponent bit, set only for |x| >= 2, inf, nan) is free; the writer of step t
//     sets it to tag(t) = ((t >> 1) & 1) ^ 1 — the value alternates per buffer half
//     and differs from the zero fill — and a reader re-sweeps its 16 x H panel until
//     every word carries the expected tag, then clears it (x ^ tag: one VALU op per
//     register that is both the check value and the operand).  No drain of the store
//     queue, no atomic, no separate poll: one store -> load visibility hop per step
//     (a sweep that finds an old tag is followed by one agent-scope acquire, see below).
//     Only the initial state (arbitrary user values) goes through the classic form
//     R1: sc1 stores, `s_waitcnt vmcnt(0)`, one relaxed agent-scope arrival per wave
//     on the tile's counter, one polling lane.  blockIdx % clusters keeps a cluster
//     on one XCD when there are 8 clusters — speed only;
//   * every spin is bounded; a timeout sets a host-visible status word that turns
//     the next call into MIRL_ERR_STATE.  The grid never exceeds the CU count (one
//     workgroup per CU: 137 KB of LDS each), clusters loop over row blocks instead.
//
// Outputs are those of the per-step path (rltime_amd/models/torch/lstm_seq.py): out[t],
// the final (masked) state, and — for the backward pass — activated gates in place of
// the pre-activations, c(t), and the masked step inputs hm / cm.
#include "common.hpp"

namespace mirl {

typedef float sq_f4 __attribute__((ext_vector_type(4)));
typedef unsigned sq_u4 __attribute__((ext_vector_type(4)));

struct SeqFwdArgs {
  int T, B;
  float* gx;            // [T][B][4H] pre-activations (x W_ih^T + b); activated gates on exit when `save`
  const float* w;       // [4H][H] W_hh
  const float* h0;      // [B][H] state before step 0 (unmasked)
  const float* c0;
  const float* keep;    // [T][B] 1 - initials
  float* out;           // [T][B][H] h(t), or null
  float* c_all;         // [T][B][H] c(t), or null
  float* hm;            // [T+1][B][H] masked step inputs h(t-1) keep(t) (hm[T] = final h), or null
  float* cm;            // [T+1][B][H]
  float* h_last;        // [B][H] or null
  float* c_last;
  int save;             // write activated gates back into gx
};

// v_exp_f32 / v_rcp_f32 forms (1 ulp each; absolute error of the gates ~1e-7): the ocml
// expf / tanhf / division sequences cost ~1 us per step on the critical path of every wave
__device__ __forceinline__ float sq_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float sq_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f); }
__device__ __forceinline__ unsigned sq_tag(int t) { return (unsigned)(((t >> 1) & 1) ^ 1) << 30; }

#define SQ_SPIN_CAP (1u << 22)
#define SQ_TP 20          // floats per row of the per-wave transposition tile (16 + 4: 16-byte aligned rows)

template <int H>
__global__ void __launch_bounds__(256, 1)
k_lstm_seq_fwd(SeqFwdArgs a, unsigned* __restrict__ counters, float* X, int64_t x_half_floats, int nclusters, int* status) {
  constexpr int KQ = H / 16;        // k-steps of 4 MFMAs per gate tile
  constexpr int KG = H / 4;         // k stride between the four MFMA k-slots
  constexpr int PW = H + 4;         // LDS row pitch of the W slice
  constexpr int NCG = H / 16;       // column groups = waves sharing one 16-row tile
  extern __shared__ __attribute__((aligned(16))) char sq_smem[];
  float* Wl = (float*)sq_smem;                      // [64 gate columns][PW]
  float* Tl = Wl + 64 * PW;                         // 4 x [16][SQ_TP]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int cluster = blockIdx.x % nclusters, cg = blockIdx.x / nclusters;
  const int li = lane & 15, lk = lane >> 4;
  const int B = a.B, T = a.T;
  // the W_hh slice: column c = gate * 16 + hidden  <-  row gate * H + 16 cg + hidden
  for (int idx = tid; idx < 64 * (H / 4); idx += 256) {
    const int c = idx / (H / 4), k4 = idx - c * (H / 4);
    const sq_f4 v = *(const sq_f4*)(a.w + ((int64_t)((c >> 4) * H + 16 * cg + (c & 15))) * H + 4 * k4);
    *(sq_f4*)(Wl + c * PW + 4 * k4) = v;
  }
  __syncthreads();
  float* Tw = Tl + wave * 16 * SQ_TP;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(X, 0, (int)(3 * x_half_floats * 4), 0x00020000);
  // exchange layout: [tile][q][lane = row + 16 kg][j]; this lane's 16-byte chunk on the publishing side:
  // transposition-tile row pr = lane / 4, hidden quad pq = lane % 4  ->  k0 = 16 cg + 4 pq
  const int pr = lane >> 2, pq = lane & 3;
  const int pk0 = 16 * cg + 4 * pq;
  const int p_slot = (((pk0 % KG) >> 2) * 64 + (pr + 16 * (pk0 / KG))) * 4;     // float offset inside the tile's [KQ][64][4] block
  // the word lane l polls for producer l (column group l): last row, last quad of its 16 x 16 block
  const int poll_k0 = 16 * (lane % NCG) + 12;
  const int poll_slot = (((poll_k0 % KG) >> 2) * 64 + (15 + 16 * (poll_k0 / KG))) * 4;
  const int nrb = (B + 63) >> 6;
  const int b_off = (li * PW + KG * lk);             // + g * 16 * PW + 4 q
  for (int rb = cluster; rb < nrb; rb += nclusters) {
    const int tile = rb * 4 + wave;
    const int row0 = tile * 16;
    if (row0 >= B) continue;                        // wave-uniform (B is a multiple of 16)
    unsigned* cnt = counters + tile * 32;
    const int64_t x_tile = (int64_t)tile * KQ * 256;
    const int r0 = row0 + 4 * lk;                   // this lane's rows r0 .. r0+3, hidden unit 16 cg + li
    const int hcol = 16 * cg + li;
    float c_in[4], hmv[4];
    {
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const float k0 = a.keep[r0 + v];
        c_in[v] = a.c0[(int64_t)(r0 + v) * H + hcol] * k0;
        hmv[v] = a.h0[(int64_t)(r0 + v) * H + hcol] * k0;
      }
    }
    bool failed = false;
    for (int t = -1; t < T; ++t) {
      float hv[4], cv[4], gi[4], gf[4], gg[4], go[4];
      if (t >= 0) {
        // ---- this step's pre-activations and the next step's reset mask: requested before the wait
        float gxv[4][4], kn[4];
        const float* gxt = a.gx + ((int64_t)t * B + r0) * 4 * H + hcol;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
#pragma unroll
          for (int g = 0; g < 4; ++g) gxv[g][v] = gxt[(int64_t)v * 4 * H + g * H];
          kn[v] = (t + 1 < T) ? a.keep[(int64_t)(t + 1) * B + r0 + v] : 1.0f;
        }
        // ---- the step input: 16 rows x H in MFMA lane order (1 KiB per load).  Step 0: the
        // initial state, guarded by the tile's arrival counter; later steps: tagged data
        sq_f4 av[KQ];
        if (t == 0) {
          const unsigned want = (unsigned)NCG;
          int bad = 0;
          if (lane == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
              __builtin_amdgcn_s_sleep(1);
              if (++spins > SQ_SPIN_CAP) { bad = 1; break; }
            }
          }
          bad = __builtin_amdgcn_readfirstlane(bad);
          if (bad) { failed = true; break; }
          asm volatile("" ::: "memory");
        }
        {
          const unsigned tmask = t >= 1 ? sq_tag(t - 1) : 0u;
          const int xb = (int)(((t == 0 ? 2 * x_half_floats : (t & 1) * x_half_floats) + x_tile) * 4) + lane * 16;
          unsigned spins = 0;
          if (t >= 1) {
            // cheap pre-poll: lane l < NCG watches ONE word of producer l's block (a relaxed agent-scope
            // atomic load: 4 bytes per producer instead of 32 KB sweeps while the data is not there yet)
            const unsigned* probe = (const unsigned*)(X + (t & 1) * x_half_floats + x_tile + poll_slot);
            for (;;) {
              const unsigned wv = lane < NCG ? __hip_atomic_load(probe, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tmask;
              if (__all(((wv ^ tmask) & 0x40000000u) == 0u)) break;
              __builtin_amdgcn_s_sleep(1);
              if (++spins > SQ_SPIN_CAP) { failed = true; break; }
            }
            if (failed) break;
            asm volatile("" ::: "memory");
          }
          for (;;) {
            sq_u4 rv[KQ];
#pragma unroll
            for (int q = 0; q < KQ; ++q) rv[q] = __builtin_amdgcn_raw_buffer_load_b128(xr, xb + q * 1024, 0, 16);
            __builtin_amdgcn_sched_barrier(0);      // all KQ loads in flight before anything consumes them
            unsigned seen = 0;
#pragma unroll
            for (int q = 0; q < KQ; ++q) {
              const sq_u4 y = rv[q] ^ tmask;        // expected tag -> bit 30 clear: the operand itself
              seen |= (y[0] | y[1]) | (y[2] | y[3]);
              av[q] = __builtin_bit_cast(sq_f4, y);
            }
            if (t == 0 || __all((seen & 0x40000000u) == 0u)) break;
            // A failed sweep may leave lines without the expected tag in this CU's L1 (sc1 buffer loads
            // can allocate there) and a re-sweep would hit them for ever: one agent-scope acquire
            // (buffer_inv sc1) before every RE-sweep.  A stale line can delay a wave but never be
            // accepted: the previous content of a slot always carries the opposite tag.
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SQ_SPIN_CAP) { failed = true; break; }
          }
          if (failed) break;
        }
        __builtin_amdgcn_sched_barrier(0);
        sq_f4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = sq_f4{0.f, 0.f, 0.f, 0.f};
        sq_f4 bv[2][4];
#pragma unroll
        for (int g = 0; g < 4; ++g) bv[0][g] = *(const sq_f4*)(Wl + b_off + g * 16 * PW);
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
          if (q + 1 < KQ) {
#pragma unroll
            for (int g = 0; g < 4; ++g) bv[(q + 1) & 1][g] = *(const sq_f4*)(Wl + b_off + g * 16 * PW + 4 * (q + 1));
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
              acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q][j], bv[q & 1][g][j], acc[g], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);        // keep the four accumulator chains interleaved (dependent MFMAs 4 apart)
        }
        // ---- the cell, in registers: lane holds i, f, g, o of (rows r0..r0+3, hidden hcol)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          gi[v] = sq_sigmoid(acc[0][v] + gxv[0][v]);
          gf[v] = sq_sigmoid(acc[1][v] + gxv[1][v]);
          gg[v] = sq_tanh(acc[2][v] + gxv[2][v]);
          go[v] = sq_sigmoid(acc[3][v] + gxv[3][v]);
          cv[v] = gf[v] * c_in[v] + gi[v] * gg[v];
          hv[v] = go[v] * sq_tanh(cv[v]);
          c_in[v] = cv[v] * kn[v];
          hmv[v] = hv[v] * kn[v];
        }
      }
      // ---- publish the next step's input first (it is on every peer's critical path) ...
      if (t + 1 < T) {
#pragma unroll
        for (int v = 0; v < 4; ++v) Tw[(4 * lk + v) * SQ_TP + li] = hmv[v];
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const sq_f4 chunk = *(const sq_f4*)(Tw + pr * SQ_TP + 4 * pq);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (t < 0) {
          // initial state: arbitrary values, no tag — counter-guarded (form R1)
          const int off = (int)((2 * x_half_floats + x_tile + p_slot) * 4);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(sq_u4, chunk), xr, off, 0, 16);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (lane == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          const int off = (int)(((((t + 1) & 1) * x_half_floats) + x_tile + p_slot) * 4);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(sq_u4, chunk) | sq_tag(t), xr, off, 0, 16);
        }
      }
      // ---- ... then everything nobody waits for
      if (t < 0) {
        if (a.hm) {
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            a.hm[(int64_t)(r0 + v) * H + hcol] = hmv[v];
            a.cm[(int64_t)(r0 + v) * H + hcol] = c_in[v];
          }
        }
        continue;
      }
      const int64_t e0 = ((int64_t)t * B + r0) * H + hcol;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int64_t e = e0 + (int64_t)v * H;
        if (a.out) a.out[e] = hv[v];
        if (a.c_all) a.c_all[e] = cv[v];
        if (a.hm) { a.hm[e + (int64_t)B * H] = hmv[v]; a.cm[e + (int64_t)B * H] = c_in[v]; }
      }
      if (a.save) {
        float* gxt = a.gx + ((int64_t)t * B + r0) * 4 * H + hcol;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          gxt[(int64_t)v * 4 * H] = gi[v];
          gxt[(int64_t)v * 4 * H + H] = gf[v];
          gxt[(int64_t)v * 4 * H + 2 * H] = gg[v];
          gxt[(int64_t)v * 4 * H + 3 * H] = go[v];
        }
      }
      if (t == T - 1 && a.h_last) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          a.h_last[(int64_t)(r0 + v) * H + hcol] = hmv[v];
          a.c_last[(int64_t)(r0 + v) * H + hcol] = c_in[v];
        }
      }
    }
    if (failed) {
      if (lane == 0) __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
  }
}

// ---------------------------------------------------------------------------
// The same sweep for SMALL per-GPU batches (strong scaling: B = 64 per rank at 8 GPUs).  With
// 16 hidden units per wave a batch of 64 sequences is 4 row tiles x 32 column groups = 128
// waves: one eighth of the chip works and every step still costs a full 512-MFMA chain
// (8.7 us per step measured at B = 64).  Here a wave owns 4 hidden units x 4 gates = ONE
// 16-column MFMA tile: 128 MFMAs per step on four k-interleaved accumulators (1.7 us), four
// times as many waves.  Lane n of a 16-lane row holds gate n / 4 of hidden n % 4, so the four
// gates of a hidden unit sit in lanes n % 4 + {0, 4, 8, 12}: pre-activations are gathered with
// three cross-lane reads (ds_bpermute) per row.  Exchange layout, tags and the acquire-before-
// re-sweep rule are those of k_lstm_seq_fwd; a tile now has H / 4 producers.
template <int H>
__global__ void __launch_bounds__(256, 1)
k_lstm_seq_fwd_narrow(SeqFwdArgs a, unsigned* __restrict__ counters, float* X, int64_t x_half_floats, int nclusters, int* status) {
  constexpr int KQ = H / 16, KG = H / 4, PW = H + 4;
  constexpr int NP = H / 4;         // producers per row tile
  extern __shared__ __attribute__((aligned(16))) char sq_smem[];
  float* Wl = (float*)sq_smem;                      // [16 columns = gate * 4 + hid][PW]
  float* Tl = Wl + 16 * PW;                         // 4 x [16 rows][4]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int cluster = blockIdx.x % nclusters, cgn = blockIdx.x / nclusters;
  const int li = lane & 15, lk = lane >> 4;
  const int g0 = li >> 2, hid = li & 3;             // this lane's gate and hidden unit (4 cgn + hid)
  const int B = a.B, T = a.T;
  for (int idx = tid; idx < 16 * (H / 4); idx += 256) {
    const int c = idx / (H / 4), k4 = idx - c * (H / 4);
    const sq_f4 v = *(const sq_f4*)(a.w + ((int64_t)((c >> 2) * H + 4 * cgn + (c & 3))) * H + 4 * k4);
    *(sq_f4*)(Wl + c * PW + 4 * k4) = v;
  }
  __syncthreads();
  float* Tw = Tl + wave * 64;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(X, 0, (int)(3 * x_half_floats * 4), 0x00020000);
  const int pk0 = 4 * cgn;                           // this wave publishes hidden units pk0 .. pk0 + 3: lanes 0..15 = rows
  const int p_slot = (((pk0 % KG) >> 2) * 64 + ((lane & 15) + 16 * (pk0 / KG))) * 4;
  const int nrb = (B + 63) >> 6;
  const int b_off = (li * PW + KG * lk);
  const int src1 = (lane & 48) | ((li + 4) & 15), src2 = (lane & 48) | ((li + 8) & 15), src3 = (lane & 48) | ((li + 12) & 15);
  for (int rb = cluster; rb < nrb; rb += nclusters) {
    const int tile = rb * 4 + wave;
    const int row0 = tile * 16;
    if (row0 >= B) continue;
    unsigned* cnt = counters + tile * 32;
    const int64_t x_tile = (int64_t)tile * KQ * 256;
    const int r0 = row0 + 4 * lk;
    const int hcol = 4 * cgn + hid;
    float c_in[4], hmv[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const float k0 = a.keep[r0 + v];
      c_in[v] = a.c0[(int64_t)(r0 + v) * H + hcol] * k0;
      hmv[v] = a.h0[(int64_t)(r0 + v) * H + hcol] * k0;
    }
    bool failed = false;
    for (int t = -1; t < T; ++t) {
      float hv[4], cv[4], act[4];
      if (t >= 0) {
        float gxv[4], kn[4];
        const float* gxt = a.gx + ((int64_t)t * B + r0) * 4 * H + g0 * H + hcol;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          gxv[v] = gxt[(int64_t)v * 4 * H];
          kn[v] = (t + 1 < T) ? a.keep[(int64_t)(t + 1) * B + r0 + v] : 1.0f;
        }
        sq_f4 av[KQ];
        if (t == 0) {
          const unsigned want = (unsigned)NP;
          int bad = 0;
          if (lane == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
              __builtin_amdgcn_s_sleep(1);
              if (++spins > SQ_SPIN_CAP) { bad = 1; break; }
            }
          }
          bad = __builtin_amdgcn_readfirstlane(bad);
          if (bad) { failed = true; break; }
          asm volatile("" ::: "memory");
        }
        {
          const unsigned tmask = t >= 1 ? sq_tag(t - 1) : 0u;
          const int xb = (int)(((t == 0 ? 2 * x_half_floats : (t & 1) * x_half_floats) + x_tile) * 4) + lane * 16;
          unsigned spins = 0;
          if (t >= 1) {
            // pre-poll: one word of every producer's block (row 15 of its 4 hidden units), two loads per lane
            const unsigned* base = (const unsigned*)(X + (t & 1) * x_half_floats + x_tile);
            for (;;) {
              unsigned ok = 1u;
#pragma unroll
              for (int half = 0; half < (NP + 63) / 64; ++half) {
                const int pcg = lane + 64 * half;
                if (pcg < NP) {
                  const int k0 = 4 * pcg;
                  const unsigned wv = __hip_atomic_load(base + (((k0 % KG) >> 2) * 64 + (15 + 16 * (k0 / KG))) * 4, __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_AGENT);
                  ok &= (((wv ^ tmask) & 0x40000000u) == 0u) ? 1u : 0u;
                }
              }
              if (__all(ok != 0u)) break;
              __builtin_amdgcn_s_sleep(1);
              if (++spins > SQ_SPIN_CAP) { failed = true; break; }
            }
            if (failed) break;
            asm volatile("" ::: "memory");
          }
          for (;;) {
            sq_u4 rv[KQ];
#pragma unroll
            for (int q = 0; q < KQ; ++q) rv[q] = __builtin_amdgcn_raw_buffer_load_b128(xr, xb + q * 1024, 0, 16);
            __builtin_amdgcn_sched_barrier(0);
            unsigned seen = 0;
#pragma unroll
            for (int q = 0; q < KQ; ++q) {
              const sq_u4 y = rv[q] ^ tmask;
              seen |= (y[0] | y[1]) | (y[2] | y[3]);
              av[q] = __builtin_bit_cast(sq_f4, y);
            }
            if (t == 0 || __all((seen & 0x40000000u) == 0u)) break;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // see k_lstm_seq_fwd: stale tagged lines can sit in L1
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SQ_SPIN_CAP) { failed = true; break; }
          }
          if (failed) break;
        }
        __builtin_amdgcn_sched_barrier(0);
        sq_f4 acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = sq_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
          const sq_f4 bv = *(const sq_f4*)(Wl + b_off + 4 * q);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q][j], bv[j], acc[j], 0, 0, 0);
        }
        // ---- pre-activation of (row, own gate, own hidden), then the other three gates of that hidden unit
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const float pre = ((acc[0][v] + acc[1][v]) + (acc[2][v] + acc[3][v])) + gxv[v];
          const float mine = g0 == 2 ? sq_tanh(pre) : sq_sigmoid(pre);
          act[v] = mine;
          const float o1 = __shfl(mine, src1), o2 = __shfl(mine, src2), o3 = __shfl(mine, src3);   // gates g0+1, g0+2, g0+3 (mod 4)
          float gate[4];
          gate[g0] = mine; gate[(g0 + 1) & 3] = o1; gate[(g0 + 2) & 3] = o2; gate[(g0 + 3) & 3] = o3;
          cv[v] = gate[1] * c_in[v] + gate[0] * gate[2];
          hv[v] = gate[3] * sq_tanh(cv[v]);
          c_in[v] = cv[v] * kn[v];
          hmv[v] = hv[v] * kn[v];
        }
      }
      if (t + 1 < T) {
        if (g0 == 0) {
#pragma unroll
          for (int v = 0; v < 4; ++v) Tw[(4 * lk + v) * 4 + hid] = hmv[v];
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const sq_f4 chunk = *(const sq_f4*)(Tw + (lane & 15) * 4);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (t < 0) {
          const int off = (int)((2 * x_half_floats + x_tile + p_slot) * 4);
          if (lane < 16) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(sq_u4, chunk), xr, off, 0, 16);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (lane == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          const int off = (int)(((((t + 1) & 1) * x_half_floats) + x_tile + p_slot) * 4);
          if (lane < 16) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(sq_u4, chunk) | sq_tag(t), xr, off, 0, 16);
        }
      }
      const bool owner = g0 == 0;                    // one of the four lanes of a hidden unit writes its h / c
      if (t < 0) {
        if (a.hm && owner) {
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            a.hm[(int64_t)(r0 + v) * H + hcol] = hmv[v];
            a.cm[(int64_t)(r0 + v) * H + hcol] = c_in[v];
          }
        }
        continue;
      }
      const int64_t e0 = ((int64_t)t * B + r0) * H + hcol;
      if (owner) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int64_t e = e0 + (int64_t)v * H;
          if (a.out) a.out[e] = hv[v];
          if (a.c_all) a.c_all[e] = cv[v];
          if (a.hm) { a.hm[e + (int64_t)B * H] = hmv[v]; a.cm[e + (int64_t)B * H] = c_in[v]; }
        }
        if (t == T - 1 && a.h_last) {
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            a.h_last[(int64_t)(r0 + v) * H + hcol] = hmv[v];
            a.c_last[(int64_t)(r0 + v) * H + hcol] = c_in[v];
          }
        }
      }
      if (a.save) {
        float* gxs = a.gx + ((int64_t)t * B + r0) * 4 * H + g0 * H + hcol;
#pragma unroll
        for (int v = 0; v < 4; ++v) gxs[(int64_t)v * 4 * H] = act[v];
      }
    }
    if (failed) {
      if (lane == 0) __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
  }
}

// ---------------------------------------------------------------------------
// The backward sweep as one persistent launch (same ownership as the forward: wave =
// 16-row tile x 16 hidden units x 4 gates, W_hh slice resident in LDS for all T steps).
//
//   dh(t)[r][j] = d_out[t][r][j] + keep(t+1)[r] * sum_c dgates(t+1)[r][c] W_hh[c][j]
//
// The contraction over the 4H gate columns is split by OWNER: a wave multiplies ITS 64
// columns of dgates(t+1) — which it has just produced, no exchange on the operand side —
// with its 64 x H slice of W_hh into a partial dh for ALL H hidden units (K = 64, N = H:
// 512 MFMAs, 32 independent accumulator tiles), and hands the 16 x 16 block of hidden
// units 16 jt .. 16 jt + 15 to the wave that owns them.  Blocks travel in the accumulator
// layout itself (lane l holds rows 4 (l >> 4) + v of column l & 15): one 16 B write-through
// store per lane and block, no transposition on either side, and the receiving lane adds
// its 32 incoming float4 in source order — a fixed summation order, bit-identical reruns.
// Half the exchange traffic of gathering whole dgates rows (32 KB read + 32 KB written per
// wave and step instead of 128 KB read), and the dgates operand never leaves the CU.
// Gradients are unbounded, so there is no free tag bit: the classic form R1 of the guide —
// sc1 stores, `s_waitcnt vmcnt(0)`, one relaxed agent-scope arrival per wave and step on
// the tile's counter, one polling lane, ONE agent-scope acquire, then the loads.
struct SeqBwdArgs {
  int T, B;
  float* gates;          // [T][B][4H] activated gates in, d loss / d pre-activation out
  const float* w;        // [4H][H]
  const float* c_all;    // [T][B][H]
  const float* cm;       // [T+1][B][H] masked cell inputs (row t = c_in of step t)
  const float* d_out;    // [T][B][H]
  const float* keep;     // [T][B]
};

#define SQB_LP 36          // floats per (column, lane-column) row of the LDS weight slice: 32 + 4 (conflict-free b128 reads)

template <int H>
__global__ void __launch_bounds__(256, 1)
k_lstm_seq_bwd(SeqBwdArgs a, unsigned* __restrict__ counters, float* P, int64_t p_half_floats, int nclusters, int* status) {
  constexpr int NCG = H / 16;       // column groups = hidden-unit owners = 16-column blocks of a partial
  static_assert(NCG == 32, "the partial-block layout below is written for H = 512");
  extern __shared__ __attribute__((aligned(16))) char sq_smem[];
  float* Wl = (float*)sq_smem;                      // [c = gate * 16 + hid (64)][li (16)][SQB_LP]: W_hh[gate H + 16 cg + hid][16 jt + li] at jt
  float* Tl = Wl + 64 * 16 * SQB_LP;                // 4 x [gate][row][hid] transposition tiles
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int cluster = blockIdx.x % nclusters, cg = blockIdx.x / nclusters;
  const int li = lane & 15, lk = lane >> 4;
  const int B = a.B, T = a.T;
  for (int idx = tid; idx < 64 * H; idx += 256) {
    const int c = idx / H, j = idx - c * H;
    Wl[(c * 16 + (j & 15)) * SQB_LP + (j >> 4)] = a.w[((int64_t)((c >> 4) * H + 16 * cg + (c & 15))) * H + j];
  }
  __syncthreads();
  float* Tw = Tl + wave * 1024;
  const int tiles = B / 16;
  const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(P, 0, (int)(2 * p_half_floats * 4), 0x00020000);
  const int nrb = (B + 63) >> 6;
  for (int rb = cluster; rb < nrb; rb += nclusters) {
    const int tile = rb * 4 + wave;
    const int row0 = tile * 16;
    if (row0 >= B) continue;
    unsigned* cnt = counters + tile * 32;
    const int r0 = row0 + 4 * lk;
    const int hcol = 16 * cg + li;
    // partial blocks: [parity][tile][dst column group][src column group][lane][4]
    const int64_t p_tile = (int64_t)tile * NCG * NCG * 256;
    float dcr[4] = {0.f, 0.f, 0.f, 0.f};
    bool failed = false;
    for (int t = T - 1; t >= 0; --t) {
      const bool first = t == T - 1;
      // ---- this step's saved forward values, requested before the wait
      float gi[4], gf[4], gg[4], go[4], ct[4], cin[4], dout[4], kn[4];
      {
        const float* gt = a.gates + ((int64_t)t * B + r0) * 4 * H + hcol;
        const int64_t e0 = ((int64_t)t * B + r0) * H + hcol;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          gi[v] = gt[(int64_t)v * 4 * H]; gf[v] = gt[(int64_t)v * 4 * H + H];
          gg[v] = gt[(int64_t)v * 4 * H + 2 * H]; go[v] = gt[(int64_t)v * 4 * H + 3 * H];
          ct[v] = a.c_all[e0 + (int64_t)v * H];
          cin[v] = a.cm[e0 + (int64_t)v * H];
          dout[v] = a.d_out ? a.d_out[e0 + (int64_t)v * H] : 0.f;
          kn[v] = first ? 1.0f : a.keep[(int64_t)(t + 1) * B + r0 + v];
        }
      }
      // ---- dh_rec: the 32 partial blocks of this wave's (rows, hidden units) from step t + 1
      float dhr[4] = {0.f, 0.f, 0.f, 0.f};
      if (!first) {
        const unsigned want = (unsigned)NCG * (unsigned)(T - 1 - t);
        int bad = 0;
        if (lane == 0) {
          unsigned spins = 0;
          while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SQ_SPIN_CAP) { bad = 1; break; }
          }
        }
        bad = __builtin_amdgcn_readfirstlane(bad);
        if (bad) { failed = true; break; }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const int pb = (int)(((((t + 1) & 1) * p_half_floats) + p_tile + (int64_t)cg * NCG * 256) * 4) + lane * 16;
        sq_f4 part[NCG];
#pragma unroll
        for (int s = 0; s < NCG; ++s) part[s] = __builtin_bit_cast(sq_f4, __builtin_amdgcn_raw_buffer_load_b128(pr, pb + s * 1024, 0, 16));
#pragma unroll
        for (int s = 0; s < NCG; ++s) {
#pragma unroll
          for (int v = 0; v < 4; ++v) dhr[v] = dhr[v] + part[s][v];
        }
      }
      // ---- the cell's backward (csrc/lstm.hip k_lstm_cell_bwd), in registers
      float dgv[4][4];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const float dh = dout[v] + (first ? 0.0f : dhr[v] * kn[v]);
        const float tc = tanhf(ct[v]);
        const float dc = (first ? 0.0f : dcr[v] * kn[v]) + dh * go[v] * (1.0f - tc * tc);
        dgv[0][v] = dc * gg[v] * gi[v] * (1.0f - gi[v]);
        dgv[1][v] = dc * cin[v] * gf[v] * (1.0f - gf[v]);
        dgv[2][v] = dc * gi[v] * (1.0f - gg[v] * gg[v]);
        dgv[3][v] = dh * tc * go[v] * (1.0f - go[v]);
        dcr[v] = dc * gf[v];
      }
      if (t > 0) {
        // ---- partial dh(t-1 side) = dgates(t)[own 64 columns] x W slice: A operand via one LDS transposition
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
          for (int v = 0; v < 4; ++v) Tw[(g * 16 + 4 * lk + v) * 16 + li] = dgv[g][v];
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        sq_f4 av[4];                                  // A[row = li][k slot = lk] for k-steps s = 0..15: gate lk, hidden s
#pragma unroll
        for (int q = 0; q < 4; ++q) av[q] = *(const sq_f4*)(Tw + (lk * 16 + li) * 16 + 4 * q);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        sq_f4 acc[NCG];
#pragma unroll
        for (int jt = 0; jt < NCG; ++jt) acc[jt] = sq_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          sq_f4 bq[8];
          const float* brow = Wl + ((16 * lk + s) * 16 + li) * SQB_LP;
#pragma unroll
          for (int k = 0; k < 8; ++k) bq[k] = *(const sq_f4*)(brow + 4 * k);
#pragma unroll
          for (int jt = 0; jt < NCG; ++jt)
            acc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s >> 2][s & 3], bq[jt >> 2][jt & 3], acc[jt], 0, 0, 0);
        }
        // ---- hand block jt to the owner of hidden units 16 jt ..: accumulator layout, 16 B per lane
        const int ob = (int)((((t & 1) * p_half_floats) + p_tile + (int64_t)cg * 256) * 4) + lane * 16;
#pragma unroll
        for (int jt = 0; jt < NCG; ++jt)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(sq_u4, acc[jt]), pr, ob + jt * (NCG * 1024), 0, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      // ---- d loss / d pre-activation of this step, in place of the activated gates
      {
        float* gt = a.gates + ((int64_t)t * B + r0) * 4 * H + hcol;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          gt[(int64_t)v * 4 * H] = dgv[0][v]; gt[(int64_t)v * 4 * H + H] = dgv[1][v];
          gt[(int64_t)v * 4 * H + 2 * H] = dgv[2][v]; gt[(int64_t)v * 4 * H + 3 * H] = dgv[3][v];
        }
      }
    }
    if (failed) {
      if (lane == 0) __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
  }
  (void)tiles;
}

static int* g_seq_status = nullptr;       // pinned host word the kernels write on a spin timeout
static int g_seq_cus = 0;

static int seq_init() {
  if (g_seq_status) return MIRL_OK;
  int dev = 0;
  MIRL_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  MIRL_HIP(hipGetDeviceProperties(&prop, dev));
  g_seq_cus = prop.multiProcessorCount;
  int* p = nullptr;
  MIRL_HIP(hipHostMalloc((void**)&p, 64, hipHostMallocMapped));
  *p = 0;
  g_seq_status = p;
  return MIRL_OK;
}

template <int H>
static int seq_launch(const SeqFwdArgs& a, void* workspace, hipStream_t st) {
  constexpr int NCG = H / 16, KQ = H / 16;
  const int tiles = a.B / 16, nrb = (a.B + 63) / 64;
  // small batches: four times as many, four times narrower waves (k_lstm_seq_fwd_narrow) while they all
  // fit the chip at once (every wave a step waits for must be resident)
  static const int narrow_env = getenv("MIRL_LSTM_SEQ_NARROW") ? atoi(getenv("MIRL_LSTM_SEQ_NARROW")) : -1;
  const bool narrow = narrow_env >= 0 ? (narrow_env != 0 && nrb * (H / 4) <= g_seq_cus) : (nrb * (H / 4) <= g_seq_cus);
  if (narrow) {
    const size_t cnt_b = (size_t)tiles * 32 * sizeof(unsigned);
    const int64_t x_half_n = (int64_t)tiles * KQ * 256;
    unsigned* counters_n = (unsigned*)workspace;
    float* Xn = (float*)((char*)workspace + align_up(cnt_b, 256));
    MIRL_HIP(hipMemsetAsync(workspace, 0, align_up(cnt_b, 256) + (size_t)(2 * x_half_n) * sizeof(float), st));
    const size_t lds_n = (size_t)(16 * (H + 4) + 4 * 64) * sizeof(float);
    int* status_n = nullptr;
    MIRL_HIP(hipHostGetDevicePointer((void**)&status_n, g_seq_status, 0));
    hipLaunchKernelGGL(k_lstm_seq_fwd_narrow<H>, dim3((unsigned)(nrb * (H / 4))), dim3(256), lds_n, st, a, counters_n, Xn, x_half_n, nrb, status_n);
    MIRL_LAUNCH_CHECK();
    return MIRL_OK;
  }
  int ncl = g_seq_cus / NCG;
  if (ncl < 1) return fail(MIRL_ERR_ARG, "lstm_seq_fwd: fewer compute units than column groups");
  if (ncl > nrb) ncl = nrb;
  const size_t cnt_bytes = (size_t)tiles * 32 * sizeof(unsigned);
  const int64_t x_half = (int64_t)tiles * KQ * 256;
  // [arrival counters][exchange half 0][exchange half 1][initial-state buffer]; the counters and
  // both tagged halves start from zero (tag 0 = "nothing written yet") on every launch
  unsigned* counters = (unsigned*)workspace;
  float* X = (float*)((char*)workspace + align_up(cnt_bytes, 256));
  MIRL_HIP(hipMemsetAsync(workspace, 0, align_up(cnt_bytes, 256) + (size_t)(2 * x_half) * sizeof(float), st));
  const size_t lds = (size_t)(64 * (H + 4) + 4 * 16 * SQ_TP) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    MIRL_HIP(hipFuncSetAttribute((const void*)k_lstm_seq_fwd<H>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  int* status_dev = nullptr;
  MIRL_HIP(hipHostGetDevicePointer((void**)&status_dev, g_seq_status, 0));
  hipLaunchKernelGGL(k_lstm_seq_fwd<H>, dim3((unsigned)(ncl * NCG)), dim3(256), lds, st, a, counters, X, x_half, ncl, status_dev);
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}

template <int H>
static int seq_bwd_launch(const SeqBwdArgs& a, void* workspace, hipStream_t st) {
  constexpr int NCG = H / 16;
  const int tiles = a.B / 16, nrb = (a.B + 63) / 64;
  int ncl = g_seq_cus / NCG;
  if (ncl < 1) return fail(MIRL_ERR_ARG, "lstm_seq_bwd: fewer compute units than column groups");
  if (ncl > nrb) ncl = nrb;
  const size_t cnt_bytes = align_up((size_t)tiles * 32 * sizeof(unsigned), 256);
  const int64_t p_half = (int64_t)tiles * NCG * NCG * 256;
  MIRL_HIP(hipMemsetAsync(workspace, 0, cnt_bytes, st));
  const size_t lds = (size_t)(64 * 16 * SQB_LP + 4 * 1024) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    MIRL_HIP(hipFuncSetAttribute((const void*)k_lstm_seq_bwd<H>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  int* status_dev = nullptr;
  MIRL_HIP(hipHostGetDevicePointer((void**)&status_dev, g_seq_status, 0));
  hipLaunchKernelGGL(k_lstm_seq_bwd<H>, dim3((unsigned)(ncl * NCG)), dim3(256), lds, st, a, (unsigned*)workspace,
                     (float*)((char*)workspace + cnt_bytes), p_half, ncl, status_dev);
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}

}  // namespace mirl

using namespace mirl;

extern "C" int mirl_lstm_seq_bwd_supported(int32_t T, int32_t B, int32_t H) {
  // the partial blocks of a sweep must fit 32-bit buffer offsets: 2 x (B / 16) x 1 MiB
  return (T >= 1 && B >= 16 && B % 16 == 0 && H == 512 && (int64_t)(B / 16) * 2 * 1048576 < 2147483648LL) ? 1 : 0;
}

extern "C" int mirl_lstm_seq_bwd_workspace_bytes(int32_t B, int32_t H, int64_t* bytes) {
  if (!bytes || B <= 0 || H != 512 || B % 16) return fail(MIRL_ERR_ARG, "bad lstm_seq_bwd_workspace_bytes arguments");
  const int64_t tiles = B / 16, ncg = H / 16;
  *bytes = (int64_t)align_up((size_t)tiles * 32 * sizeof(unsigned), 256) + 2 * tiles * ncg * ncg * 256 * (int64_t)sizeof(float);
  return MIRL_OK;
}

extern "C" int mirl_lstm_seq_bwd(int32_t T, int32_t B, int32_t H, float* gates, const float* w_hh, const float* c_all,
                                 const float* cm, const float* d_out, const float* keep, void* workspace, void* stream) {
  if (!mirl_lstm_seq_bwd_supported(T, B, H)) return fail(MIRL_ERR_ARG, "lstm_seq_bwd: unsupported shape (B multiple of 16, H = 512)");
  if (!gates || !w_hh || !c_all || !cm || !keep || !workspace || ((uintptr_t)workspace % 256))
    return fail(MIRL_ERR_ARG, "bad lstm_seq_bwd arguments");
  int rc = seq_init();
  if (rc) return rc;
  if (*(volatile int*)g_seq_status) {
    *g_seq_status = 0;
    return fail(MIRL_ERR_STATE, "an earlier lstm_seq launch gave up waiting for a peer workgroup (results of that sweep are invalid)");
  }
  SeqBwdArgs a{T, B, gates, w_hh, c_all, cm, d_out, keep};
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps("k_lstm_seq_bwd", 4.0 * B * H * T * (8.0 + 3.0) + 4.0 * 4.0 * H * H, st, 2.0 * B * H * 4.0 * H * T);
  return seq_bwd_launch<512>(a, workspace, st);
}

extern "C" int mirl_lstm_seq_supported(int32_t T, int32_t B, int32_t H) {
  return (T >= 1 && B >= 16 && B % 16 == 0 && (H == 128 || H == 256 || H == 512)) ? 1 : 0;
}

extern "C" int mirl_lstm_seq_workspace_bytes(int32_t B, int32_t H, int64_t* bytes) {
  if (!bytes || B <= 0 || H <= 0 || B % 16 || H % 16) return fail(MIRL_ERR_ARG, "bad lstm_seq_workspace_bytes arguments");
  const int64_t tiles = B / 16;
  *bytes = (int64_t)align_up((size_t)tiles * 32 * sizeof(unsigned), 256) + 3 * tiles * (H / 16) * 256 * (int64_t)sizeof(float);
  return MIRL_OK;
}

extern "C" int mirl_lstm_seq_fwd(int32_t T, int32_t B, int32_t H, float* gx, const float* w_hh, const float* h0, const float* c0,
                                 const float* keep, float* out, float* c_all, float* hm, float* cm, float* h_last, float* c_last,
                                 int32_t save_gates, void* workspace, void* stream) {
  if (!mirl_lstm_seq_supported(T, B, H)) return fail(MIRL_ERR_ARG, "lstm_seq_fwd: unsupported shape (B multiple of 16, H in {128, 256, 512})");
  if (!gx || !w_hh || !h0 || !c0 || !keep || !workspace || (!hm != !cm) || (!h_last != !c_last) || (!hm && !h_last))
    return fail(MIRL_ERR_ARG, "bad lstm_seq_fwd arguments");
  if (((uintptr_t)w_hh % 16) || ((uintptr_t)workspace % 256)) return fail(MIRL_ERR_ARG, "lstm_seq_fwd needs a 16-byte aligned w_hh and a 256-byte aligned workspace");
  int rc = seq_init();
  if (rc) return rc;
  if (*(volatile int*)g_seq_status) {
    *g_seq_status = 0;
    return fail(MIRL_ERR_STATE, "an earlier lstm_seq_fwd launch gave up waiting for a peer workgroup — or its recurrent state became non-finite, which the tagged exchange cannot carry (results of that sweep are invalid)");
  }
  SeqFwdArgs a{T, B, gx, w_hh, h0, c0, keep, out, c_all, hm, cm, h_last, c_last, save_gates ? 1 : 0};
  hipStream_t st = (hipStream_t)stream;
  // per step: pre-activations read (+ activated gates written), h / c outputs; HBM-side algorithmic bytes of the sweep
  const double per_step = 4.0 * B * H * (4.0 + (save_gates ? 4.0 : 0.0) + (out ? 1.0 : 0.0) + (c_all ? 1.0 : 0.0) + (hm ? 2.0 : 0.0));
  ProfScope ps("k_lstm_seq_fwd", per_step * T + 4.0 * 4.0 * H * H, st, 2.0 * B * H * 4.0 * H * T);
  if (H == 512) return seq_launch<512>(a, workspace, st);
  if (H == 256) return seq_launch<256>(a, workspace, st);
  return seq_launch<128>(a, workspace, st);
}

// The forward sweep's launch geometry (what seq_launch picks): how many workgroups must be CO-RESIDENT for the sweep's
// exchange to make progress, their LDS, and how many of them one compute unit can hold.  Callers that want two sweeps in
// flight at once (two streams) use it to check that both grids fit the chip together.
extern "C" int mirl_lstm_seq_fwd_grid(int32_t B, int32_t H, int32_t* workgroups, int64_t* lds_bytes, int32_t* compute_units,
                                      int32_t* per_compute_unit) {
  if (!workgroups || !lds_bytes || !compute_units || !per_compute_unit || !mirl_lstm_seq_supported(2, B, H))
    return fail(MIRL_ERR_ARG, "bad lstm_seq_fwd_grid arguments");
  int rc = seq_init();
  if (rc) return rc;
  const int NCG = H / 16, nrb = (B + 63) / 64;
  static const int narrow_env = getenv("MIRL_LSTM_SEQ_NARROW") ? atoi(getenv("MIRL_LSTM_SEQ_NARROW")) : -1;
  const bool narrow = (narrow_env != 0) && (nrb * (H / 4) <= g_seq_cus);
  int64_t lds;
  if (narrow) {
    *workgroups = nrb * (H / 4);
    lds = (int64_t)(16 * (H + 4) + 4 * 64) * (int64_t)sizeof(float);
  } else {
    int ncl = g_seq_cus / NCG;
    if (ncl > nrb) ncl = nrb;
    *workgroups = ncl * NCG;
    lds = (int64_t)(64 * (H + 4) + 4 * 16 * SQ_TP) * (int64_t)sizeof(float);
  }
  *lds_bytes = lds;
  *compute_units = g_seq_cus;
  int64_t per = (160 * 1024) / (lds > 0 ? lds : 1);
  if (per > 8) per = 8;                       // 32 waves per compute unit, 4 per workgroup
  *per_compute_unit = (int32_t)per;
  return MIRL_OK;
}

extern "C" int mirl_lstm_seq_status(int32_t* status) {
  if (!status) return fail(MIRL_ERR_ARG, "bad lstm_seq_status arguments");
  *status = g_seq_status ? *(volatile int*)g_seq_status : 0;
  return MIRL_OK;
}
