// actnet.hip — the policy network of ONE acting vector step at acting batch sizes (E = 16 .. 256 envs).
//
// Reference: Actor.get_samples -> policy.actor_predict on the last input state (acting/actor.py:108-122,
// policies/torch/dqn.py:132-148, iqn.py:67-106): conv stack -> LSTMCell -> [quantile embedding x features] ->
// last FC (+ dueling value-hidden) -> outputs -> dueling combine / quantile mean / arg-max / epsilon-greedy.
// Rounds 2-4 ran that as 15 launches per vector step on the libraries (MIOpen: zero-fill + implicit GEMM +
// bias/ReLU per conv layer; hipBLASLt: four GEMMs whose M is 32 .. 8192 rows) — every one of them 4-16 us at
// E = 32 whatever it computes, and 40 vector steps per learner step.  Here the same network is six launches:
//
//   k_act_conv<L2>, k_act_conv<L3>  conv layers 2-3 as implicit GEMMs over NHWC rows, 16-row tiles (162 / 98
//                                   workgroups at E = 32), bias + ReLU in the epilogue, no zero-fill; layer 3
//                                   writes straight into the LSTM product's input rows [features | h]
//   k_act_lstm                      [features | h] x [W_ih | W_hh]^T + (b_ih + b_hh) with the CELL in the same launch:
//                                   256 workgroups = 64 column blocks (8 hidden units x 4 gates) x 4 slices of K, the
//                                   last slice of a block to arrive adds the shares and runs the cell (E <= 64)
//   k_act_embed                     quantile fractions (Philox) -> cos features -> embedding product + ReLU -> x features
//   k_act_hidden                    hidden layer(s) + ReLU -> the workgroup's share of the output layer; the
//                                   (rows, 1024) hidden activations never reach HBM
//   k_act_head_select               sums the output shares, dueling combine, mean over quantile rows, arg-max,
//                                   epsilon-greedy (the same Philox draws as k_actor_head)
//
// All products run on v_mfma_f32_16x16x4_f32 (exact f32 products, f32 accumulation): acting is latency-bound, not
// pipe-bound, at these sizes, and its results stay within the rounding of the library path it replaces
// (tests/test_actnet_gpu.py, tests/test_fast_acting_gpu.py).
//
// Operand trick used by every kernel: lane l = (r = l & 15, g = l >> 4) loads ONE float4 of its operand row,
// A[r][k0 + 4g .. 4g+3] resp. B[r][k0 + 4g .. 4g+3] (both operands K-contiguous: NHWC pixels, row-major weights),
// and issues four MFMAs on the components: MFMA q contracts k = k0 + 4g + q over the four lane groups, so 16 k
// are covered by one 16-byte load per operand row.  Maps (cdna_hip_programming.md section 3): lane l supplies
// A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15]; D: col = l & 15, row = 4 (l >> 4) + i.
#include "common.hpp"
#include "philox.hpp"
#include <stdlib.h>

namespace mirl {

typedef float an_f4 __attribute__((ext_vector_type(4)));

#define AN_MFMA4(acc, a, b)                                                   \
  do {                                                                        \
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32((a).x, (b).x, acc, 0, 0, 0);   \
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32((a).y, (b).y, acc, 0, 0, 0);   \
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32((a).z, (b).z, acc, 0, 0, 0);   \
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32((a).w, (b).w, acc, 0, 0, 0);   \
  } while (0)

__device__ __forceinline__ float an_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// ------------------------------------------------------------------------------------------------------------
// conv layers 2 / 3 (models/torch/modules/cnn.py:43-50: Conv2d + ReLU), NHWC in, NHWC out, 64 output channels.
// x (frames, Hi, Wi, CI); w (64, KH*KW*CI): tap-major, input channel fastest (= the conv weight in channels_last
// memory order); y: frame f at y + f * y_frame_pitch, pixel p, channel c at [p * 64 + c].
// Workgroup = 4 waves = 16*RT output pixels x 64 channels; wave w owns channels 16w .. 16w+15.
struct ActConvArgs {
  const float* x; const float* w; const float* bias; float* y;
  int64_t y_frame_pitch;
  int frames, Hi, Wi, Ho, Wo;
};

template <int CI, int KH, int KW, int S, int RT>
__global__ void __launch_bounds__(256)
k_act_conv(ActConvArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int HW = a.Ho * a.Wo;
  const int64_t M = (int64_t)a.frames * HW;
  const int64_t m0 = (int64_t)blockIdx.x * (16 * RT);
  constexpr int K = KH * KW * CI;
  const float* xp[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    int64_t m = m0 + 16 * t + r;
    if (m >= M) m = M - 1;                                  // clamped rows are computed and never stored
    const int f = (int)(m / HW), p = (int)(m - (int64_t)f * HW), oy = p / a.Wo, ox = p - oy * a.Wo;
    xp[t] = a.x + (((int64_t)f * a.Hi + oy * S) * a.Wi + ox * S) * CI + 4 * g;
  }
  const float* wp = a.w + (int64_t)(16 * wave + r) * K + 4 * g;
  an_f4 acc[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) acc[t] = an_f4{0.f, 0.f, 0.f, 0.f};
  // one kernel ROW (KW taps x CI channels = NS 16-wide K steps) per pass, double-buffered: the loads of row ky + 1 are
  // all issued before the first MFMA of row ky (sched_barrier keeps the compiler from sinking them back to their uses)
  constexpr int NS = KW * (CI / 16);
  an_f4 b[2][NS], av[2][NS][RT];
#define AN_CONV_LOAD(buf, ky)                                                                             \
  _Pragma("unroll") for (int kx = 0; kx < KW; ++kx)                                                       \
  _Pragma("unroll") for (int c = 0; c < CI / 16; ++c) {                                                   \
    b[buf][kx * (CI / 16) + c] = *(const an_f4*)(wp + ((ky) * KW + kx) * CI + 16 * c);                    \
    _Pragma("unroll") for (int t = 0; t < RT; ++t)                                                        \
      av[buf][kx * (CI / 16) + c][t] = *(const an_f4*)(xp[t] + ((ky) * a.Wi + kx) * CI + 16 * c);         \
  }
  // (RT = 2 — launches with thousands of workgroups — keeps ONE row in registers: 3 waves per SIMD hide the loads instead)
  constexpr bool DB = RT == 1;
  AN_CONV_LOAD(0, 0)
#pragma unroll
  for (int ky = 0; ky < KH; ++ky) {
    if (DB && ky + 1 < KH) { AN_CONV_LOAD((ky + 1) & 1, ky + 1) }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < NS; ++q)
#pragma unroll
      for (int t = 0; t < RT; ++t) AN_MFMA4(acc[t], av[DB ? (ky & 1) : 0][q][t], b[DB ? (ky & 1) : 0][q]);
    __builtin_amdgcn_sched_barrier(0);
    if (!DB && ky + 1 < KH) { AN_CONV_LOAD(0, ky + 1) }
  }
#undef AN_CONV_LOAD
  const int co = 16 * wave + r;
  const float bs = a.bias[co];
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t m = m0 + 16 * t + 4 * g + i;
      if (m < M) {
        const int64_t f = m / HW, p = m - f * HW;
        const float v = acc[t][i] + bs;
        a.y[f * a.y_frame_pitch + p * 64 + co] = v > 0.f ? v : 0.f;
      }
    }
}

// The same layers for HUNDREDS of frames (one GPU acting for all 256 envs): there the 16-row workgroups above re-read the
// layer's weights from L2 once per tile and the CU's L1 is what bounds them (measured 40 / 31 us at 256 frames against the
// 9 / 6 us the f32 matrix pipe needs).  Here the weights — 64 x K floats, 131 / 147 KB: they fit a CU's LDS — are staged ONCE
// per workgroup, one persistent workgroup per CU walks the pixel tiles, and a wave's K step is one 16-byte global load of
// its own pixel rows + two conflict-free 16-byte LDS reads per MFMA quad: the pipe, not the memory system, sets the time.
// Workgroup = 8 waves; wave w takes tile slot w >> 1 and channel half w & 1 (16 pixels x 32 channels).
template <int CI, int KH, int KW, int S>
__global__ void __launch_bounds__(512)
k_act_conv_wlds(ActConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float wl[];       // [64][K + 4]
  constexpr int K = KH * KW * CI, WP = K + 4, NS = CI / 16, TAPS = KH * KW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  {
    // every lane's share of the weights is requested before the first LDS write (one round trip, not PER of them)
    constexpr int PER = 64 * (K / 4) / 512;
    static_assert(PER * 512 == 64 * (K / 4), "weights must split evenly over the workgroup");
    an_f4 v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) v[k] = *(const an_f4*)(a.w + 4 * (int64_t)(tid + 512 * k));     // (col, q) = linear: rows of K floats
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int idx = tid + 512 * k, col = idx / (K / 4), q = idx - col * (K / 4);
      *(an_f4*)(wl + col * WP + 4 * q) = v[k];
    }
  }
  __syncthreads();
  const int HW = a.Ho * a.Wo;
  const int64_t M = (int64_t)a.frames * HW;
  const int64_t tiles = (M + 15) / 16;
  const int half = wave & 1;
  const float* wb = wl + (32 * half + r) * WP + 4 * g;                 // this lane's weight rows: columns 32 half + {r, 16 + r}
  const float bs0 = a.bias[32 * half + r], bs1 = a.bias[32 * half + 16 + r];
  for (int64_t tile = (int64_t)blockIdx.x * 4 + (wave >> 1); tile < tiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t m0 = tile * 16;
    int64_t m = m0 + r; if (m >= M) m = M - 1;
    const int f = (int)(m / HW), p = (int)(m - (int64_t)f * HW), oy = p / a.Wo, ox = p - oy * a.Wo;
    const float* xp = a.x + (((int64_t)f * a.Hi + oy * S) * a.Wi + ox * S) * CI + 4 * g;
    an_f4 acc0 = an_f4{0.f, 0.f, 0.f, 0.f}, acc1 = an_f4{0.f, 0.f, 0.f, 0.f};
    // the pixel rows of tap t + AHEAD are requested while tap t multiplies (a ring of AHEAD + 1 register buffers)
    constexpr int AHEAD = 3;
    an_f4 av[AHEAD + 1][NS];
#pragma unroll
    for (int t = 0; t < AHEAD && t < TAPS; ++t) {
      const int ky = t / KW, kx = t - ky * KW;
#pragma unroll
      for (int c = 0; c < NS; ++c) av[t][c] = *(const an_f4*)(xp + (ky * a.Wi + kx) * CI + 16 * c);
    }
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      if (t + AHEAD < TAPS) {
        const int ky = (t + AHEAD) / KW, kx = (t + AHEAD) - ky * KW;
#pragma unroll
        for (int c = 0; c < NS; ++c) av[(t + AHEAD) % (AHEAD + 1)][c] = *(const an_f4*)(xp + (ky * a.Wi + kx) * CI + 16 * c);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < NS; ++c) {
        const an_f4 b0 = *(const an_f4*)(wb + t * CI + 16 * c), b1 = *(const an_f4*)(wb + 16 * WP + t * CI + 16 * c);
        AN_MFMA4(acc0, av[t % (AHEAD + 1)][c], b0);
        AN_MFMA4(acc1, av[t % (AHEAD + 1)][c], b1);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t mm = m0 + 4 * g + i;
      if (mm < M) {
        const int64_t ff = mm / HW, pp = mm - ff * HW;
        float* y = a.y + ff * a.y_frame_pitch + pp * 64 + 32 * half + r;
        const float v0 = acc0[i] + bs0, v1 = acc1[i] + bs1;
        y[0] = v0 > 0.f ? v0 : 0.f;
        y[16] = v1 > 0.f ? v1 : 0.f;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// One LSTMCell step for E <= 64 envs (modules/lstm.py:83-116 at timesteps = 1; gate order i, f, g, o):
//   gates[e][q*H + j] = bias[q*H + j] + sum_k xh[e][k] * w[q*H + j][k],  k over [features | h_in] (K = F + H)
//   c = f * c_in + i * g,  h = o * tanh(c)
// At 32 rows this is a 30 MB weight stream against 0.5 GFLOP: what bounds it is how many CUs pull on the stream and how
// often the 467 KB of input rows are re-read, not the matrix pipe.  Workgroup (jb, kb) = hidden units 8 jb .. 8 jb + 7
// (two 16-column MFMA tiles whose columns are (gate q = r >> 2, unit r & 3): the four gates of a unit meet in one
// workgroup) x the kb-th of KB slices of K; (H / 8) x KB = 256 workgroups at H = 512, one per CU.  Its 8 waves take the
// slice's 16-wide K steps round-robin (neighbouring waves read neighbouring 64-byte pieces of a weight row); the input
// rows of a slice are read once per workgroup and feed both tiles.  Partial tiles meet in LDS, the workgroup's share goes
// to the workspace, and the LAST workgroup of a column block to arrive (one agent-scope counter per block) adds the KB
// shares and the bias and runs the cell — every weight byte is read once per step, the gates never exist in HBM, and
// there is no second launch.  The counter returns to zero, so the launch can be replayed from a captured graph.
struct ActLstmArgs {
  const float* xh; int64_t xh_pitch;
  const float* w; const float* bias; const float* c_in;
  float* h_out; float* c_out;
  float* shares; unsigned* arrived;      // workspace: [KB][H / 8][E][32] floats, [H / 8] counters (zero between launches)
  int E, H, K, KB, SPB;                  // SPB: K steps per slice
};

template <int RT>
__global__ void __launch_bounds__(512)
k_act_lstm(ActLstmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float red[];      // [8 waves][RT][2 tiles][16 rows][16 cols]
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int jb = blockIdx.x, kb = blockIdx.y, j0 = 8 * jb;
  const float* wp[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) wp[c] = a.w + ((int64_t)(r >> 2) * a.H + j0 + 4 * c + (r & 3)) * a.K + 4 * g;
  const float* xp[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    int row = 16 * t + r;
    if (row >= a.E) row = a.E - 1;                           // clamped rows are computed and never used
    xp[t] = a.xh + (int64_t)row * a.xh_pitch + 4 * g;
  }
  an_f4 acc[RT][2];
#pragma unroll
  for (int t = 0; t < RT; ++t) { acc[t][0] = an_f4{0.f, 0.f, 0.f, 0.f}; acc[t][1] = an_f4{0.f, 0.f, 0.f, 0.f}; }
  const int steps = a.K / 16;
  const int s0 = kb * a.SPB;
  const int s1 = s0 + a.SPB < steps ? s0 + a.SPB : steps;
  // wave w takes steps s0 + w, s0 + w + 8, ...; a pass = U of them, double-buffered: pass p + 1's loads are in flight while
  // pass p multiplies.  Straight-line passes (a conditional load would make the compiler wait for EVERY outstanding load
  // at the join): a step past the slice's end reads step s0 and multiplies by zeroed weights.
  constexpr int U = RT >= 4 ? 2 : 4;
  const int passes = ((a.SPB + 7) / 8 + U - 1) / U;
  an_f4 b[2][U][2], av[2][U][RT];
#define AN_LSTM_LOAD(buf, pass)                                                                  \
  _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                \
    int ss = s0 + wave + 8 * (U * (pass) + u);                                                   \
    if (ss >= s1) ss = s0;                                                                       \
    b[buf][u][0] = *(const an_f4*)(wp[0] + 16 * ss);                                             \
    b[buf][u][1] = *(const an_f4*)(wp[1] + 16 * ss);                                             \
    _Pragma("unroll") for (int t = 0; t < RT; ++t) av[buf][u][t] = *(const an_f4*)(xp[t] + 16 * ss); \
  }
#define AN_LSTM_MUL(buf, pass)                                                                   \
  _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                \
    const bool ok = s0 + wave + 8 * (U * (pass) + u) < s1;                                       \
    an_f4 b0 = b[buf][u][0], b1 = b[buf][u][1];                                                  \
    b0.x = ok ? b0.x : 0.f; b0.y = ok ? b0.y : 0.f; b0.z = ok ? b0.z : 0.f; b0.w = ok ? b0.w : 0.f; \
    b1.x = ok ? b1.x : 0.f; b1.y = ok ? b1.y : 0.f; b1.z = ok ? b1.z : 0.f; b1.w = ok ? b1.w : 0.f; \
    _Pragma("unroll") for (int t = 0; t < RT; ++t) {                                             \
      AN_MFMA4(acc[t][0], av[buf][u][t], b0);                                                    \
      AN_MFMA4(acc[t][1], av[buf][u][t], b1);                                                    \
    }                                                                                            \
  }
  AN_LSTM_LOAD(0, 0)
  for (int p = 0; p < passes; p += 2) {
    AN_LSTM_LOAD(1, p + 1)
    __builtin_amdgcn_sched_barrier(0);
    AN_LSTM_MUL(0, p)
    __builtin_amdgcn_sched_barrier(0);
    AN_LSTM_LOAD(0, p + 2)
    __builtin_amdgcn_sched_barrier(0);
    AN_LSTM_MUL(1, p + 1)
    __builtin_amdgcn_sched_barrier(0);
  }
#undef AN_LSTM_LOAD
#undef AN_LSTM_MUL
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i) red[(((wave * RT + t) * 2 + c) * 16 + 4 * g + i) * 16 + r] = acc[t][c][i];
  __syncthreads();
  // this workgroup's share: rows < E x 32 columns (column 16 c + r of the block)
  const int NJB = a.H / 8;
  float* mine = a.shares + ((int64_t)kb * NJB + jb) * a.E * 32;
  for (int idx = tid; idx < a.E * 32; idx += 512) {
    const int row = idx >> 5, col = idx & 31, t = row >> 4, rr = row & 15, c = col >> 4, cr = col & 15;
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) sum = sum + red[(((w * RT + t) * 2 + c) * 16 + rr) * 16 + cr];
    if (a.KB == 1) red[8 * RT * 512 + idx] = sum;          // single slice: the share stays in LDS
    else __hip_atomic_store(mine + idx, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through (sc1) store
  }
  if (a.KB > 1) {
    // No fence (an agent-scope release / acquire pair writes back and invalidates whole L2s on this 8-XCD part: measured
    // 2x the kernel).  The shares travel as write-through stores and sc1 loads; `s_waitcnt vmcnt(0)` holds the arrival
    // back until this wave's stores have reached the coherence point (the exchange form of csrc/lstm_seq.hip).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_last = __hip_atomic_fetch_add(a.arrived + jb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(a.KB - 1);
    __syncthreads();
    if (!s_last) return;
    if (tid == 0) __hip_atomic_store(a.arrived + jb, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch / replay
  } else {
    __syncthreads();
  }
  for (int idx = tid; idx < a.E * 8; idx += 512) {
    const int row = idx >> 3, u = idx & 7;
    float pre[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col = 16 * (u >> 2) + 4 * q + (u & 3);
      float sum = a.bias[q * a.H + j0 + u];
      if (a.KB == 1) sum = sum + red[8 * RT * 512 + row * 32 + col];
      else for (int k = 0; k < a.KB; ++k)
        sum = sum + __hip_atomic_load(a.shares + (((int64_t)k * NJB + jb) * a.E + row) * 32 + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pre[q] = sum;
    }
    const float gi = an_sigmoid(pre[0]), gf = an_sigmoid(pre[1]), gg = tanhf(pre[2]), go = an_sigmoid(pre[3]);
    const int64_t at = (int64_t)row * a.H + j0 + u;
    const float cc = gf * a.c_in[at] + gi * gg;
    a.c_out[at] = cc;
    a.h_out[at] = go * tanhf(cc);
  }
}

// ------------------------------------------------------------------------------------------------------------
// The quantile layer's product for R = E * N rows (policies/torch/iqn.py:67-106):
//   tau[m]    = given, or 24-bit uniform of Philox4x32-10(seed ^ 0x7A5, *step, m)     (iqn.py:76, as k_cos_embed_rng)
//   phi[m][i] = cos(freq[i] * tau[m]),  freq = pi * (1 .. D)                            (iqn.py:78-81)
//   x[m][c]   = relu(phi[m] . wq[c] + bq[c]) * h[m / N][c]                              (iqn.py:82-102)
// Workgroup = 16 rows x 128 columns; K = D <= 64 is four MFMA steps per tile: a latency kernel.
struct ActEmbedArgs {
  const float* h; const float* freq; const float* taus; const float* wq; const float* bq;
  float* x; float* tau_out;
  uint64_t seed; const uint64_t* step;
  int E, N, H, D;
  int groups;             // 16-row groups per workgroup (1 for small batches: a latency kernel there)
};

__global__ void __launch_bounds__(256)
k_act_embed(ActEmbedArgs a) {
  __shared__ __attribute__((aligned(16))) float phi[16 * 68];
  __shared__ float tau_s[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int R = a.E * a.N, PP = a.D + 4;
  const int nsD = a.D / 16;
  const int cbase = blockIdx.y * 128 + 32 * wave;
  const bool cols = cbase < a.H;                     // (a wave without columns still takes part in the barriers)
  const bool on1 = cbase + 16 < a.H;
  // this wave's weight fragments: loaded ONCE per workgroup and kept for all of its row groups (round 6: at the 256-env
  // acting batch's 8 192 rows a workgroup per 16 rows re-read the 128 KB of weights 2 048 times)
  an_f4 q0[4], q1[4];
#pragma unroll
  for (int sq = 0; sq < 4; ++sq) {
    const int sc = sq < nsD ? sq : 0;
    const int c0 = cols ? cbase : 0;
    q0[sq] = *(const an_f4*)(a.wq + (int64_t)(c0 + r) * a.D + 16 * sc + 4 * g);
    q1[sq] = *(const an_f4*)(a.wq + (int64_t)((cols && on1 ? cbase + 16 : c0) + r) * a.D + 16 * sc + 4 * g);
  }
  // The products run TRANSPOSED (A = 16 weight rows, B = 16 phi rows): a lane then owns FOUR CONSECUTIVE COLUMNS
  // cbase + 16 c + 4 g .. + 3 of ONE row (row0 + r) — bias, features and the result move as 16-byte vectors (round 6: with one
  // column of four rows per lane the 16.8 MB result of the 256-env batch went out in 4-byte stores at 0.8 TB/s)
  an_f4 bs[2] = {an_f4{0.f, 0.f, 0.f, 0.f}, an_f4{0.f, 0.f, 0.f, 0.f}};
  if (cols) { bs[0] = *(const an_f4*)(a.bq + cbase + 4 * g); if (on1) bs[1] = *(const an_f4*)(a.bq + cbase + 16 + 4 * g); }
  for (int rg = 0; rg < a.groups; ++rg) {
    const int row0 = (blockIdx.x * a.groups + rg) * 16;
    if (row0 >= R) break;                             // block-uniform
    if (rg) __syncthreads();                          // the previous group's phi rows are read
    if (tid < 16) {
      int m = row0 + tid; if (m >= R) m = R - 1;
      float t;
      if (a.taus) t = a.taus[m];
      else { uint32_t rn[4]; philox_4x32(a.seed ^ 0x7A5ull, *a.step, (uint32_t)m, rn); t = (float)(rn[0] >> 8) * (1.0f / 16777216.0f); }
      tau_s[tid] = t;
      if (a.tau_out && blockIdx.y == 0 && row0 + tid < R) a.tau_out[row0 + tid] = t;
    }
    __syncthreads();
    for (int idx = tid; idx < 16 * a.D; idx += 256) {
      const int rr = idx / a.D, i = idx - rr * a.D;
      phi[rr * PP + i] = cosf(a.freq[i] * tau_s[rr]);
    }
    __syncthreads();
    if (!cols) continue;
    an_f4 acc[2] = {an_f4{0.f, 0.f, 0.f, 0.f}, an_f4{0.f, 0.f, 0.f, 0.f}};
    an_f4 pa[4];
#pragma unroll
    for (int sq = 0; sq < 4; ++sq)
      pa[sq] = sq < nsD ? *(const an_f4*)(phi + r * PP + 16 * sq + 4 * g) : an_f4{0.f, 0.f, 0.f, 0.f};
    const int m = row0 + r;
    const int erow = (m < R ? m : R - 1) / a.N;       // env row of h (one 32-bit division, ahead of the products)
#pragma unroll
    for (int sq = 0; sq < 4; ++sq) { AN_MFMA4(acc[0], q0[sq], pa[sq]); AN_MFMA4(acc[1], q1[sq], pa[sq]); }
    if (m < R) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        if (c == 1 && !on1) break;
        const int col = cbase + 16 * c + 4 * g;
        const an_f4 hv = *(const an_f4*)(a.h + (int64_t)erow * a.H + col);
        an_f4 v = acc[c] + bs[c];
        v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
        *(an_f4*)(a.x + (int64_t)m * a.H + col) = v * hv;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// The acting head over the output shares, for ONE env by ONE wave (lane n = quantile row n): out[m][o] = bout[o] +
// sum_p part[p][m][o]; columns 0 .. A-1 are the advantages, column A (when has_val) the dueling value; rows are NOP = 8 CH
// floats apart.  Then exactly k_actor_head (acting.hip): V + A - mean_a A (dqn.py:74-87), mean over the N quantile rows
// (iqn.py actor post-processing), first maximum, epsilon-greedy with one Philox4x32-10 block per (step, env)
// (epsilon_greedy.py:74-100).  All of a row's shares are fetched as 16-byte loads before anything is added.  (Running
// this inside k_act_hidden, by the last column block of a row block to arrive, was built and measured: 104.6 vs 103.6 us
// per vector step at 32 envs — the serial tail costs what the launch saved; profiles/r05_acting_probe.jsonl.)
struct ActSelectArgs {
  const float* part; const float* bout;
  const double* eps; const double* expo; double eps_min;
  uint64_t rng_seed; const uint64_t* rng_step;
  int32_t* actions; float* qvalues;
  int E, N, A, P, has_val;
};

template <int CH>
__device__ __forceinline__ void act_select_env(const ActSelectArgs& a, int e, int lane) {
  constexpr int NOP = 8 * CH;
  const int R = a.E * a.N;
  float acc[NOP];
#pragma unroll
  for (int k = 0; k < NOP; ++k) acc[k] = 0.f;
  for (int n = lane; n < a.N; n += 64) {
    const int m = e * a.N + n;
    float o[NOP];
#pragma unroll
    for (int k = 0; k < NOP; ++k) o[k] = k < a.A + (a.has_val ? 1 : 0) ? a.bout[k] : 0.f;
    for (int p = 0; p < a.P; ++p) {
      const an_f4* src = (const an_f4*)(a.part + ((int64_t)p * R + m) * NOP);
#pragma unroll
      for (int c = 0; c < 2 * CH; ++c) {
        const an_f4 v = src[c];
        o[4 * c] = o[4 * c] + v.x; o[4 * c + 1] = o[4 * c + 1] + v.y; o[4 * c + 2] = o[4 * c + 2] + v.z; o[4 * c + 3] = o[4 * c + 3] + v.w;
      }
    }
    float off = 0.f;
    if (a.has_val) {                                    // dueling: V + A - mean_a A
      float mean = 0.f, v = 0.f;
#pragma unroll
      for (int k = 0; k < NOP; ++k) { if (k < a.A) mean = mean + o[k]; if (k == a.A) v = o[k]; }
      off = v - mean / (float)a.A;
    }
#pragma unroll
    for (int k = 0; k < NOP; ++k) if (k < a.A) acc[k] = acc[k] + (o[k] + off);
  }
  float best = 0.f; int arg = 0;
#pragma unroll
  for (int k = 0; k < NOP; ++k) {
    float s = acc[k];
    for (int d = 32; d > 0; d >>= 1) s = s + __shfl_xor(s, d);
    if (k < a.A) {
      const float q = s / (float)a.N;
      if (lane == 0) a.qvalues[(int64_t)e * a.A + k] = q;
      if (k == 0 || q > best) { best = q; arg = k; }     // first maximum, like argmax
    }
  }
  if (lane == 0) {
    int act = arg;
    if (a.eps) {
      const double pe = pow(*a.eps, a.expo ? a.expo[e] : 1.0);
      const float per = (float)(pe > a.eps_min ? pe : a.eps_min);
      uint32_t rn[4];
      philox_4x32(a.rng_seed, *a.rng_step, (uint32_t)e, rn);
      const float uf = (float)(rn[0] >> 8) * (1.0f / 16777216.0f);
      if (uf < per) act = (int)(((uint64_t)rn[1] * (uint64_t)a.A) >> 32);
    }
    a.actions[e] = act;
  }
}

template <int CH>
__global__ void __launch_bounds__(256)
k_act_head_select(ActSelectArgs a) {
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e < a.E) act_select_env<CH>(a, e, threadIdx.x & 63);
}

// ------------------------------------------------------------------------------------------------------------
// The head's hidden layers and its share of the output layer for R rows:
//   hid[m][j]      = relu(x[m] . wfc[j] + bfc[j]),  j < HID   (last FC layer, then the dueling value-hidden layer:
//                                                               dqn.py:50-66 — both read the same x)
//   part[cb][m][o] = sum over hidden columns 64 cb .. 64 cb + 63 of hid[m][j] * wout[o][j],  o < NO <= NOP <= 32
// (wout: the advantage rows over the FC columns and, when q-values are wanted, the value row over the value-hidden
// columns — block-diagonal).  x rows are xrow_div rows apart in units of rows: row m reads x[(m / xdiv)] (xdiv = 1 for
// the quantile product above; without a quantile layer x = h and xdiv = 1 too since N = 1).
// Workgroup = 2 x 2 waves = (32 WR) rows x 64 columns, wave = (16 WR) x 32; both operands straight from L2 in 16-byte
// pieces (the four waves share them through the CU's L1), K = H in passes of four 16-wide steps, fully unrolled and
// double-buffered.  At R = 1024, HID = 1024 (32 envs x 32 quantiles, q-values wanted) WR = 2 gives 256 workgroups =
// one wave per SIMD with 512 MFMAs each: the f32 matrix pipe's time for this product; at HID = 512 WR = 1 does.
// The output layer is one more MFMA per wave: the ReLU'd tile goes through LDS into operand layout and meets wout's rows.
struct ActHiddenArgs {
  const float* x; const float* wfc; const float* bfc; const float* wout;
  float* part;
  int R, H, HID, NO, NOP;
};

template <int WR, int HP>
__global__ void __launch_bounds__(256)
k_act_hidden(ActHiddenArgs a) {
  __shared__ __attribute__((aligned(16))) float hs[4 * 16 * WR * 36];       // per wave: (16 WR) x 32 hidden tile, pitch 36
  __shared__ float outs[4 * 16 * WR * 32];                                   // per wave: (16 WR) x NOP output shares
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;
  const int row0 = blockIdx.x * (32 * WR) + 16 * WR * wr;
  const int cbase = blockIdx.y * 64 + 32 * wc;
  const float* xp[WR];
#pragma unroll
  for (int t = 0; t < WR; ++t) { int m = row0 + 16 * t + r; if (m >= a.R) m = a.R - 1; xp[t] = a.x + (int64_t)m * a.H + 4 * g; }
  const bool on0 = cbase < a.HID, on1 = cbase + 16 < a.HID;
  const float* wp[2];
  wp[0] = a.wfc + (int64_t)((on0 ? cbase : 0) + r) * a.H + 4 * g;
  wp[1] = a.wfc + (int64_t)((on1 ? cbase + 16 : 0) + r) * a.H + 4 * g;
  an_f4 acc[WR][2];
#pragma unroll
  for (int t = 0; t < WR; ++t) { acc[t][0] = an_f4{0.f, 0.f, 0.f, 0.f}; acc[t][1] = an_f4{0.f, 0.f, 0.f, 0.f}; }
  an_f4 A[2][4][WR], B[2][4][2];
#define AN_HID_LOAD(buf, pass)                                                                   \
  _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                \
    B[buf][u][0] = *(const an_f4*)(wp[0] + 64 * (pass) + 16 * u);                                \
    B[buf][u][1] = *(const an_f4*)(wp[1] + 64 * (pass) + 16 * u);                                \
    _Pragma("unroll") for (int t = 0; t < WR; ++t) A[buf][u][t] = *(const an_f4*)(xp[t] + 64 * (pass) + 16 * u); \
  }
  AN_HID_LOAD(0, 0)
#pragma unroll
  for (int p = 0; p < HP; ++p) {
    if (p + 1 < HP) { AN_HID_LOAD((p + 1) & 1, p + 1) }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int t = 0; t < WR; ++t) {
        AN_MFMA4(acc[t][0], A[p & 1][u][t], B[p & 1][u][0]);
        AN_MFMA4(acc[t][1], A[p & 1][u][t], B[p & 1][u][1]);
      }
    __builtin_amdgcn_sched_barrier(0);
  }
#undef AN_HID_LOAD
  // bias + ReLU (columns past HID: zero), tile into LDS in operand layout
  float* hw = hs + wave * (16 * WR * 36);
  const float bf0 = on0 ? a.bfc[cbase + r] : 0.f, bf1 = on1 ? a.bfc[cbase + 16 + r] : 0.f;
#pragma unroll
  for (int t = 0; t < WR; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float v0 = acc[t][0][i] + bf0, v1 = acc[t][1][i] + bf1;
      hw[(16 * t + 4 * g + i) * 36 + r] = on0 ? (v0 > 0.f ? v0 : 0.f) : 0.f;
      hw[(16 * t + 4 * g + i) * 36 + 16 + r] = on1 ? (v1 > 0.f ? v1 : 0.f) : 0.f;
    }
  // this wave's share of the output layer: out[m][o] += sum over its 32 hidden columns of hid[m][j] * wout[o][j]
  // (the same wave wrote hw: LDS operations of one wave execute in order)
  float* ow = outs + wave * (16 * WR * 32);
  for (int ot = 0; ot < (a.NOP + 15) / 16; ++ot) {
    const int o = 16 * ot + r;
    an_f4 w0 = an_f4{0.f, 0.f, 0.f, 0.f}, w1 = an_f4{0.f, 0.f, 0.f, 0.f};
    if (o < a.NO) {
      if (on0) w0 = *(const an_f4*)(a.wout + (int64_t)o * a.HID + cbase + 4 * g);
      if (on1) w1 = *(const an_f4*)(a.wout + (int64_t)o * a.HID + cbase + 16 + 4 * g);
    }
#pragma unroll
    for (int t = 0; t < WR; ++t) {
      an_f4 oa = an_f4{0.f, 0.f, 0.f, 0.f};
      const an_f4 h0 = *(const an_f4*)(hw + (16 * t + r) * 36 + 4 * g), h1 = *(const an_f4*)(hw + (16 * t + r) * 36 + 16 + 4 * g);
      AN_MFMA4(oa, h0, w0);
      AN_MFMA4(oa, h1, w1);
#pragma unroll
      for (int i = 0; i < 4; ++i) if (o < a.NOP) ow[(16 * t + 4 * g + i) * 32 + o] = oa[i];
    }
  }
  __syncthreads();
  // the two column halves of a row half add up; one share per (column block, row, output)
  for (int idx = tid; idx < 32 * WR * a.NOP; idx += 256) {
    const int rr = idx / a.NOP, o = idx - rr * a.NOP;
    const int half = rr / (16 * WR), lr = rr - half * (16 * WR);
    const int m = blockIdx.x * (32 * WR) + rr;
    if (m < a.R && o < a.NO)
      a.part[((int64_t)blockIdx.y * a.R + m) * a.NOP + o] = outs[((2 * half) * 16 * WR + lr) * 32 + o] + outs[((2 * half + 1) * 16 * WR + lr) * 32 + o];
  }
}

}  // namespace mirl

using namespace mirl;

static bool an_al16(const void* p) { return ((uintptr_t)p % 16) == 0; }

extern "C" int mirl_act_conv_supported(int32_t layer, int32_t Ci, int32_t Co, int32_t k, int32_t stride, int32_t Hi, int32_t Wi) {
  if (Co != 64 || Hi < k || Wi < k) return 0;
  if (layer == 2) return Ci == 32 && k == 4 && stride == 2;
  if (layer == 3) return Ci == 64 && k == 3 && stride == 1;
  return 0;
}

extern "C" int mirl_act_conv_fwd(int32_t layer, int64_t frames, int32_t Hi, int32_t Wi, const float* x, const float* w_taps,
                                 const float* bias, float* y, int64_t y_frame_pitch, void* stream) {
  if ((layer != 2 && layer != 3) || frames <= 0 || frames > (1 << 20) || !x || !w_taps || !bias || !y)
    return fail(MIRL_ERR_ARG, "bad act_conv_fwd arguments");
  const int k = layer == 2 ? 4 : 3, s = layer == 2 ? 2 : 1, Ci = layer == 2 ? 32 : 64;
  if (!mirl_act_conv_supported(layer, Ci, 64, k, s, Hi, Wi)) return fail(MIRL_ERR_ARG, "act_conv_fwd: input smaller than the kernel");
  ActConvArgs a;
  a.x = x; a.w = w_taps; a.bias = bias; a.y = y; a.y_frame_pitch = y_frame_pitch;
  a.frames = (int)frames; a.Hi = Hi; a.Wi = Wi; a.Ho = (Hi - k) / s + 1; a.Wo = (Wi - k) / s + 1;
  if (y_frame_pitch < (int64_t)a.Ho * a.Wo * 64 || !an_al16(x) || !an_al16(w_taps))
    return fail(MIRL_ERR_ARG, "act_conv_fwd: 16-byte aligned x / w and a frame pitch >= Ho*Wo*64 are required");
  const int64_t M = frames * a.Ho * a.Wo;
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(layer == 2 ? "k_act_conv2" : "k_act_conv3", 4.0 * ((double)frames * Hi * Wi * Ci + (double)M * 64 + 64.0 * k * k * Ci), st,
               2.0 * (double)M * 64 * k * k * Ci);
  static const int wlds_env = getenv("MIRL_ACT_CONV_WLDS") ? atoi(getenv("MIRL_ACT_CONV_WLDS")) : 1;
  if (M >= 6144 && wlds_env) {
    // hundreds of frames: weights resident in LDS, one persistent workgroup per CU (k_act_conv_wlds)
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int64_t tiles = (M + 15) / 16;
    unsigned grid = (unsigned)((tiles + 3) / 4); if (grid > (unsigned)cus) grid = (unsigned)cus;
    const size_t lds = sizeof(float) * 64 * (size_t)(k * k * Ci + 4);
    static bool raised[2] = {false, false};
    if (layer == 2) {
      if (!raised[0]) { MIRL_HIP(hipFuncSetAttribute((const void*)k_act_conv_wlds<32, 4, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); raised[0] = true; }
      hipLaunchKernelGGL((k_act_conv_wlds<32, 4, 4, 2>), dim3(grid), dim3(512), lds, st, a);
    } else {
      if (!raised[1]) { MIRL_HIP(hipFuncSetAttribute((const void*)k_act_conv_wlds<64, 3, 3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); raised[1] = true; }
      hipLaunchKernelGGL((k_act_conv_wlds<64, 3, 3, 1>), dim3(grid), dim3(512), lds, st, a);
    }
    MIRL_LAUNCH_CHECK();
    return MIRL_OK;
  }
  const int rt = M > 8192 ? 2 : 1;                           // 16-row tiles while they are what fills the chip
  const unsigned grid = (unsigned)((M + 16 * rt - 1) / (16 * rt));
  if (layer == 2) {
    if (rt == 1) hipLaunchKernelGGL((k_act_conv<32, 4, 4, 2, 1>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_act_conv<32, 4, 4, 2, 2>), dim3(grid), dim3(256), 0, st, a);
  } else {
    if (rt == 1) hipLaunchKernelGGL((k_act_conv<64, 3, 3, 1, 1>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_act_conv<64, 3, 3, 1, 2>), dim3(grid), dim3(256), 0, st, a);
  }
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}

extern "C" int mirl_act_lstm_supported(int32_t E, int32_t H, int32_t K) {
  return E > 0 && E <= 64 && H > 0 && (H % 8) == 0 && K > 0 && (K % 16) == 0;
}

// K slices per column block: (H / 8) x KB workgroups ~ one per CU
static int act_lstm_kb(int H, int K) {
  int kb = 256 / (H / 8);
  if (kb > 16) kb = 16;
  if (kb > K / 16 / 8) kb = K / 16 / 8;
  return kb < 1 ? 1 : kb;
}

extern "C" int mirl_act_lstm_workspace_bytes(int32_t E, int32_t H, int32_t K, int64_t* bytes) {
  if (!bytes || !mirl_act_lstm_supported(E, H, K)) return fail(MIRL_ERR_ARG, "bad act_lstm_workspace_bytes arguments");
  const int kb = act_lstm_kb(H, K);
  *bytes = (int64_t)sizeof(float) * kb * (H / 8) * E * 32 + (int64_t)sizeof(unsigned) * (H / 8) + 256;
  return MIRL_OK;
}

extern "C" int mirl_act_lstm_fwd(int32_t E, int32_t H, int32_t K, const float* xh, int64_t xh_pitch, const float* w, const float* bias,
                                 const float* c_in, float* h_out, float* c_out, void* workspace, void* stream) {
  if (!xh || !w || !bias || !c_in || !h_out || !c_out || !workspace) return fail(MIRL_ERR_ARG, "bad act_lstm_fwd arguments");
  if (!mirl_act_lstm_supported(E, H, K)) return fail(MIRL_ERR_ARG, "act_lstm_fwd: E <= 64, H % 8 == 0 and K % 16 == 0 are required");
  if (xh_pitch < K || (xh_pitch % 4) || !an_al16(xh) || !an_al16(w) || !an_al16(workspace)) return fail(MIRL_ERR_ARG, "act_lstm_fwd: 16-byte aligned rows are required");
  ActLstmArgs a;
  a.xh = xh; a.xh_pitch = xh_pitch; a.w = w; a.bias = bias; a.c_in = c_in; a.h_out = h_out; a.c_out = c_out;
  a.E = E; a.H = H; a.K = K;
  a.KB = act_lstm_kb(H, K);
  a.SPB = (K / 16 + a.KB - 1) / a.KB;
  a.arrived = (unsigned*)workspace;                          // zero on first use (the caller allocates zeroed memory)
  a.shares = (float*)((char*)workspace + ((sizeof(unsigned) * (H / 8) + 255) / 256) * 256);
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps("k_act_lstm", 4.0 * (4.0 * H * K + (double)E * K + 3.0 * E * H), st, 2.0 * E * 4.0 * H * K);
  const dim3 grid(H / 8, a.KB);
  const int rt = E <= 16 ? 1 : (E <= 32 ? 2 : 4);
  const size_t lds = sizeof(float) * ((size_t)8 * rt * 512 + (a.KB == 1 ? (size_t)E * 32 : 0));
  if (rt == 1) hipLaunchKernelGGL(k_act_lstm<1>, grid, dim3(512), lds, st, a);
  else if (rt == 2) hipLaunchKernelGGL(k_act_lstm<2>, grid, dim3(512), lds, st, a);
  else {
    static bool raised = false;
    if (!raised) { MIRL_HIP(hipFuncSetAttribute((const void*)k_act_lstm<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)); raised = true; }
    hipLaunchKernelGGL(k_act_lstm<4>, grid, dim3(512), lds, st, a);
  }
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}

extern "C" int mirl_act_head_supported(int32_t E, int32_t N, int32_t H, int32_t D, int32_t HID, int32_t NO) {
  if (E <= 0 || N <= 0 || H <= 0 || HID <= 0 || NO <= 0 || NO > 32 || (int64_t)E * N > (1 << 24)) return 0;
  if ((HID % 16) || (D && (D % 16)) || D > 64) return 0;
  return H == 64 || H == 128 || H == 256 || H == 512 || H == 1024;
}

extern "C" int mirl_act_head_parts(int32_t HID, int32_t NO, int32_t* parts, int32_t* pitch) {
  if (!parts || !pitch || HID <= 0 || NO <= 0 || NO > 32) return fail(MIRL_ERR_ARG, "bad act_head_parts arguments");
  *parts = (HID + 63) / 64;
  *pitch = (NO + 7) / 8 * 8;
  return MIRL_OK;
}

extern "C" int mirl_act_embed(int32_t E, int32_t N, int32_t H, int32_t D, const float* h, const float* freq, const float* taus,
                              uint64_t seed, const uint64_t* step, const float* wq, const float* bq, float* x, float* tau_out, void* stream) {
  if (E <= 0 || N <= 0 || H <= 0 || (H % 16) || D <= 0 || (D % 16) || D > 64 || !h || !freq || !wq || !bq || !x || (!taus && !step) ||
      (int64_t)E * N > (1 << 24) || !an_al16(wq) || !an_al16(h) || !an_al16(bq) || !an_al16(x))
    return fail(MIRL_ERR_ARG, "bad act_embed arguments (H % 16, D in {16, 32, 48, 64}, taus or a step word, 16-byte aligned h / wq / bq / x)");
  ActEmbedArgs a;
  a.h = h; a.freq = freq; a.taus = taus; a.wq = wq; a.bq = bq; a.x = x; a.tau_out = tau_out; a.seed = seed; a.step = step;
  a.E = E; a.N = N; a.H = H; a.D = D;
  const int R = E * N;
  // enough workgroups to fill the chip twice, then more rows per workgroup (the weights stay in its registers)
  const int tiles = (R + 15) / 16, ycount = (H + 127) / 128;
  static const int groups_env = getenv("MIRL_ACT_EMBED_GROUPS") ? atoi(getenv("MIRL_ACT_EMBED_GROUPS")) : 0;
  int groups = groups_env > 0 ? groups_env : 1;
  if (groups_env <= 0) while (groups < 8 && (tiles / (2 * groups)) * ycount >= 512) groups *= 2;
  a.groups = groups;
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps("k_act_embed", 4.0 * ((double)R * H + (double)E * H + (double)H * D), st, 2.0 * (double)R * H * D);
  hipLaunchKernelGGL(k_act_embed, dim3((tiles + groups - 1) / groups, ycount), dim3(256), 0, st, a);
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}

template <int WR>
static void launch_hidden(const ActHiddenArgs& a, dim3 grid, hipStream_t st) {
  switch (a.H / 64) {
    case 1: hipLaunchKernelGGL((k_act_hidden<WR, 1>), grid, dim3(256), 0, st, a); break;
    case 2: hipLaunchKernelGGL((k_act_hidden<WR, 2>), grid, dim3(256), 0, st, a); break;
    case 4: hipLaunchKernelGGL((k_act_hidden<WR, 4>), grid, dim3(256), 0, st, a); break;
    case 8: hipLaunchKernelGGL((k_act_hidden<WR, 8>), grid, dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL((k_act_hidden<WR, 16>), grid, dim3(256), 0, st, a); break;
  }
}

static int fill_select(ActSelectArgs& q, int32_t E, int32_t N, int32_t A, int32_t parts, int32_t pitch, const float* part, const float* bout,
                       int32_t has_val, const double* eps, const double* expo, double eps_min, uint64_t rng_seed, const uint64_t* rng_step,
                       int32_t* actions, float* qvalues) {
  if (E <= 0 || N <= 0 || A <= 0 || A + (has_val ? 1 : 0) > pitch || pitch > 32 || (pitch % 8) || parts <= 0 || !part || !bout || !actions ||
      !qvalues || (eps && !rng_step) || !an_al16(part) || (int64_t)parts * E * N * pitch * 4 >= (1LL << 31))
    return fail(MIRL_ERR_ARG, "bad act_head_select arguments");
  q.part = part; q.bout = bout; q.eps = eps; q.expo = expo; q.eps_min = eps_min; q.rng_seed = rng_seed; q.rng_step = rng_step;
  q.actions = actions; q.qvalues = qvalues; q.E = E; q.N = N; q.A = A; q.P = parts; q.has_val = has_val ? 1 : 0;
  return MIRL_OK;
}

static int hidden_wr(int64_t R, int cbs) {
  // 32-row workgroups while they are what fills the chip, 64-row ones (half the weight traffic per product) beyond
  return ((R + 31) / 32 * cbs > 384) ? 2 : 1;
}

static int launch_hidden_any(ActHiddenArgs& a, hipStream_t st) {
  const int cbs = (a.HID + 63) / 64;
  const int wr = hidden_wr(a.R, cbs);
  const dim3 grid((unsigned)((a.R + 32 * wr - 1) / (32 * wr)), (unsigned)cbs);
  ProfScope ps("k_act_hidden", 4.0 * ((double)a.R * a.H + (double)a.HID * a.H + (double)a.NOP * a.R * cbs), st, 2.0 * (double)a.R * a.HID * a.H);
  if (wr == 1) launch_hidden<1>(a, grid, st); else launch_hidden<2>(a, grid, st);
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}

static int fill_hidden(ActHiddenArgs& a, int32_t R, int32_t H, int32_t HID, int32_t NO, const float* x, const float* wfc, const float* bfc,
                       const float* wout, float* part) {
  if (!x || !wfc || !bfc || !wout || !part || R <= 0) return fail(MIRL_ERR_ARG, "bad act_head_hidden arguments");
  if (!mirl_act_head_supported(R, 1, H, 0, HID, NO)) return fail(MIRL_ERR_ARG, "act_head_hidden: unsupported shape (H in {64 .. 1024} a power of two, HID % 16; NO <= 32)");
  if (!an_al16(x) || !an_al16(wfc) || !an_al16(wout) || !an_al16(part) || (HID % 4)) return fail(MIRL_ERR_ARG, "act_head_hidden: 16-byte aligned operands are required");
  a.x = x; a.wfc = wfc; a.bfc = bfc; a.wout = wout; a.part = part;
  a.R = R; a.H = H; a.HID = HID; a.NO = NO; a.NOP = (NO + 7) / 8 * 8;
  return MIRL_OK;
}

extern "C" int mirl_act_head_hidden(int32_t R, int32_t H, int32_t HID, int32_t NO, const float* x, const float* wfc, const float* bfc,
                                    const float* wout, float* part, void* stream) {
  ActHiddenArgs a;
  int rc = fill_hidden(a, R, H, HID, NO, x, wfc, bfc, wout, part); if (rc) return rc;
  return launch_hidden_any(a, (hipStream_t)stream);
}

extern "C" int mirl_act_head_select(int32_t E, int32_t N, int32_t A, int32_t parts, int32_t pitch, const float* part, const float* bout,
                                    int32_t has_val, const double* eps, const double* expo, double eps_min, uint64_t rng_seed,
                                    const uint64_t* rng_step, int32_t* actions, float* qvalues, void* stream) {
  ActSelectArgs q;
  int rc = fill_select(q, E, N, A, parts, pitch, part, bout, has_val, eps, expo, eps_min, rng_seed, rng_step, actions, qvalues); if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps("k_act_head_select", 0.0, st);
  const dim3 grid((E + 3) / 4);
  switch (pitch / 8) {
    case 1: hipLaunchKernelGGL(k_act_head_select<1>, grid, dim3(256), 0, st, q); break;
    case 2: hipLaunchKernelGGL(k_act_head_select<2>, grid, dim3(256), 0, st, q); break;
    case 3: hipLaunchKernelGGL(k_act_head_select<3>, grid, dim3(256), 0, st, q); break;
    default: hipLaunchKernelGGL(k_act_head_select<4>, grid, dim3(256), 0, st, q); break;
  }
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}
