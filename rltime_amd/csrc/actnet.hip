// actnet.hip — the policy network of ONE acting vector step at acting batch sizes (E = 16 .. 256 envs).
//
// Reference: Actor.get_samples -> policy.actor_predict on the last input state (acting/actor.py:108-122,
// policies/torch/dqn.py:132-148, iqn.py:67-106): conv stack -> LSTMCell -> [quantile embedding x features] ->
// last FC (+ dueling value-hidden) -> outputs -> dueling combine / quantile mean / arg-max / epsilon-greedy.
// Rounds 2-4 ran that as 15 launches per vector step on the libraries (MIOpen: zero-fill + implicit GEMM +
// bias/ReLU per conv layer; hipBLASLt: four GEMMs whose M is 32 .. 8192 rows) — every one of them 4-16 us at
// E = 32 whatever it computes, and 40 vector steps per learner step.  Here the same network is five launches:
//
//   k_act_conv<L2>, k_act_conv<L3>  conv layers 2-3 as implicit GEMMs over NHWC rows, 16-row tiles (162 / 98
//                                   workgroups at E = 32), bias + ReLU in the epilogue, no zero-fill; layer 3
//                                   writes straight into the LSTM product's input rows [features | h]
//   k_act_lstm                      [features | h] x [W_ih | W_hh]^T + (b_ih + b_hh) with the CELL in the epilogue:
//                                   a workgroup owns 4 hidden units x 4 gates (one 16-column MFMA tile), its 8 waves
//                                   split K; every weight byte is read exactly once per step (E <= 64)
//   k_act_head_hidden               quantile fractions (Philox) -> cos features -> embedding product + ReLU ->
//                                   x features (all in LDS) -> hidden layer(s) + ReLU -> the workgroup's share of
//                                   the output layer; the (rows, 512) activations never reach HBM
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

// ------------------------------------------------------------------------------------------------------------
// One LSTMCell step for E <= 64 envs (modules/lstm.py:83-116 at timesteps = 1; gate order i, f, g, o):
//   gates[e][q*H + j] = bias[q*H + j] + sum_k xh[e][k] * w[q*H + j][k],  k over [features | h_in] (K = F + H)
//   c = f * c_in + i * g,  h = o * tanh(c)
// Workgroup b = hidden units 4b .. 4b+3: the 16 columns of its MFMA tile are (gate q = r >> 2, unit r & 3), so the
// four gates of a unit meet in one workgroup.  Its 8 waves take the 16-wide K steps round-robin (wave w: steps w,
// w + 8, ...: neighbouring waves read neighbouring 64-byte pieces of a weight row), partial tiles meet in LDS.
// H / 4 workgroups x 8 waves = one wave per SIMD at H = 512; 12-15 loads of 1 KB in flight per wave.
struct ActLstmArgs {
  const float* xh; int64_t xh_pitch;
  const float* w; const float* bias; const float* c_in;
  float* h_out; float* c_out;
  int E, H, K;
};

template <int RT>
__global__ void __launch_bounds__(512)
k_act_lstm(ActLstmArgs a) {
  __shared__ float red[8 * RT * 256];                       // [wave][row tile][16 rows][16 cols]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int j0 = blockIdx.x * 4;
  const float* wp = a.w + ((int64_t)(r >> 2) * a.H + j0 + (r & 3)) * a.K + 4 * g;
  const float* xp[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    int row = 16 * t + r;
    if (row >= a.E) row = a.E - 1;
    xp[t] = a.xh + (int64_t)row * a.xh_pitch + 4 * g;
  }
  an_f4 acc[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) acc[t] = an_f4{0.f, 0.f, 0.f, 0.f};
  // wave w takes K steps w, w + 8, w + 16, ...; a pass = U of them, double-buffered: pass p + 1's loads (U weight pieces +
  // U * RT input pieces of 1 KB each) are in flight while pass p multiplies.  Steps past the end read step 0 and
  // multiply by zero weights (uniform passes, no tail loop).
  const int steps = a.K / 16;
  constexpr int U = RT >= 4 ? 4 : 8;
  const int passes = ((steps + 7) / 8 + U - 1) / U;
  an_f4 b[2][U], av[2][U][RT];
#define AN_LSTM_LOAD(buf, pass)                                                                  \
  _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                \
    int ss = wave + 8 * (U * (pass) + u);                                                        \
    if (ss >= steps) ss = 0;                                                                     \
    b[buf][u] = *(const an_f4*)(wp + 16 * ss);                                                   \
    _Pragma("unroll") for (int t = 0; t < RT; ++t) av[buf][u][t] = *(const an_f4*)(xp[t] + 16 * ss); \
  }
#define AN_LSTM_MUL(buf, pass)                                                                   \
  _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                \
    const bool ok = wave + 8 * (U * (pass) + u) < steps;                                         \
    an_f4 bb = b[buf][u];                                                                        \
    bb.x = ok ? bb.x : 0.f; bb.y = ok ? bb.y : 0.f; bb.z = ok ? bb.z : 0.f; bb.w = ok ? bb.w : 0.f; \
    _Pragma("unroll") for (int t = 0; t < RT; ++t) AN_MFMA4(acc[t], av[buf][u][t], bb);          \
  }
  // straight-line loop body (no conditional loads: a branch would make the compiler wait for EVERY outstanding load at
  // the join); a pass beyond the end loads step 0 and multiplies by zeros
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
    for (int i = 0; i < 4; ++i) red[((wave * RT + t) * 16 + 4 * g + i) * 16 + r] = acc[t][i];
  __syncthreads();
  for (int idx = threadIdx.x; idx < a.E * 4; idx += 512) {
    const int row = idx >> 2, u = idx & 3, t = row >> 4, rr = row & 15;
    float pre[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float sum = a.bias[q * a.H + j0 + u];
#pragma unroll
      for (int w = 0; w < 8; ++w) sum = sum + red[((w * RT + t) * 16 + rr) * 16 + q * 4 + u];
      pre[q] = sum;
    }
    const float gi = an_sigmoid(pre[0]), gf = an_sigmoid(pre[1]), gg = tanhf(pre[2]), go = an_sigmoid(pre[3]);
    const int64_t at = (int64_t)row * a.H + j0 + u;
    const float c = gf * a.c_in[at] + gi * gg;
    a.c_out[at] = c;
    a.h_out[at] = go * tanhf(c);
  }
}

// ------------------------------------------------------------------------------------------------------------
// The head's hidden layers for R = E * N rows (N quantile samples per env; N = 1 without a quantile layer):
//   tau[m]    = given, or 24-bit uniform of Philox4x32-10(seed ^ 0x7A5, *step, m)     (iqn.py:76, as k_cos_embed_rng)
//   phi[m][i] = cos(freq[i] * tau[m]),  freq = pi * (1 .. D)                            (iqn.py:78-81)
//   x[m][c]   = relu(phi[m] . wq[c] + bq[c]) * h[m / N][c]                              (iqn.py:82-102)
//   hid[m][j] = relu(x[m] . wfc[j] + bfc[j]),  j < HID   (last FC layer, then the dueling value-hidden layer:
//                                                         dqn.py:50-66 — both read the same x)
//   part[cb][m][o] = sum over the workgroup's 128 hidden columns of hid[m][j] * wout[o][j],  o < NO <= NOP <= 32
// (wout: the advantage rows over the FC columns and, when q-values are wanted, the value row over the value-hidden
// columns — block-diagonal).  Workgroup (rb, cb) = rows 16*RT*rb .., hidden columns 128*cb ..; x for its rows is
// rebuilt by each of the HID / 128 column workgroups (K = D is 8x shorter than K = H) and lives in LDS only.
struct ActHeadArgs {
  const float* h; const float* freq; const float* taus;
  const float* wq; const float* bq; const float* wfc; const float* bfc; const float* wout;
  float* part; float* tau_out;
  uint64_t seed; const uint64_t* step;
  int E, N, H, D, HID, NO, NOP;                            // NOP: floats per output-share row (NO rounded up to 8)
};

template <int RT>
__global__ void __launch_bounds__(256)
k_act_head_hidden(ActHeadArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int XP = a.H + 4, PP = a.D + 4;                     // row pitches: 16-byte aligned, conflict-free operand reads
  float* xs = lds;                                          // [16 RT][H + 4]
  float* phi = xs + 16 * RT * XP;                           // [16 RT][D + 4]
  float* outp = phi + 16 * RT * PP;                         // [4 waves][16 RT][NOP]
  float* tau_s = outp + 4 * 16 * RT * a.NOP;                // [16 RT]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int64_t R = (int64_t)a.E * a.N;
  const int64_t row0 = (int64_t)blockIdx.x * (16 * RT);
  const bool quant = a.freq != nullptr;
  if (quant) {
    if (tid < 16 * RT) {
      int64_t m = row0 + tid; if (m >= R) m = R - 1;
      float t;
      if (a.taus) t = a.taus[m];
      else { uint32_t rn[4]; philox_4x32(a.seed ^ 0x7A5ull, *a.step, (uint32_t)m, rn); t = (float)(rn[0] >> 8) * (1.0f / 16777216.0f); }
      tau_s[tid] = t;
      if (a.tau_out && blockIdx.y == 0 && row0 + tid < R) a.tau_out[row0 + tid] = t;
    }
    __syncthreads();
    for (int idx = tid; idx < 16 * RT * a.D; idx += 256) {
      const int rr = idx / a.D, i = idx - rr * a.D;
      phi[rr * PP + i] = cosf(a.freq[i] * tau_s[rr]);
    }
    __syncthreads();
    // embedding product: 16 RT rows x H columns x K = D <= 64; wave w takes column tiles w, w + 4, ... (H / 64 of them).
    // The phi rows stay in registers (zeros past D), the weights of tile j + 1 are in flight while tile j multiplies; a
    // tile index past the end repeats the last tile (same values stored twice: branch-free passes).
    const int nsD = a.D / 16, ntile = a.H / 64;
    an_f4 pa[4][RT];
#pragma unroll
    for (int sq = 0; sq < 4; ++sq)
#pragma unroll
      for (int t = 0; t < RT; ++t)
        pa[sq][t] = sq < nsD ? *(const an_f4*)(phi + (16 * t + r) * PP + 16 * sq + 4 * g) : an_f4{0.f, 0.f, 0.f, 0.f};
    an_f4 qb[2][4];
#define AN_EMB_LOAD(buf, j)                                                                      \
  {                                                                                              \
    const int ct = wave + 4 * ((j) < ntile ? (j) : ntile - 1);                                   \
    _Pragma("unroll") for (int sq = 0; sq < 4; ++sq)                                             \
      qb[buf][sq] = *(const an_f4*)(a.wq + (int64_t)(16 * ct + r) * a.D + 16 * (sq < nsD ? sq : 0) + 4 * g); \
  }
#define AN_EMB_MUL(buf, j)                                                                       \
  {                                                                                              \
    const int ct = wave + 4 * ((j) < ntile ? (j) : ntile - 1);                                   \
    const int col = 16 * ct + r;                                                                 \
    an_f4 acc[RT];                                                                               \
    _Pragma("unroll") for (int t = 0; t < RT; ++t) acc[t] = an_f4{0.f, 0.f, 0.f, 0.f};           \
    _Pragma("unroll") for (int sq = 0; sq < 4; ++sq)                                             \
      _Pragma("unroll") for (int t = 0; t < RT; ++t) AN_MFMA4(acc[t], pa[sq][t], qb[buf][sq]);   \
    const float bs = a.bq[col];                                                                  \
    _Pragma("unroll") for (int t = 0; t < RT; ++t)                                               \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                            \
        const int rr = 16 * t + 4 * g + i;                                                       \
        int64_t m = row0 + rr; if (m >= R) m = R - 1;                                            \
        float v = acc[t][i] + bs;                                                                \
        v = v > 0.f ? v : 0.f;                                                                   \
        xs[rr * XP + col] = v * a.h[(m / a.N) * a.H + col];                                      \
      }                                                                                          \
  }
    AN_EMB_LOAD(0, 0)
    for (int j = 0; j < ntile; j += 2) {
      AN_EMB_LOAD(1, j + 1)
      __builtin_amdgcn_sched_barrier(0);
      AN_EMB_MUL(0, j)
      __builtin_amdgcn_sched_barrier(0);
      AN_EMB_LOAD(0, j + 2)
      __builtin_amdgcn_sched_barrier(0);
      AN_EMB_MUL(1, j + 1)
      __builtin_amdgcn_sched_barrier(0);
    }
#undef AN_EMB_LOAD
#undef AN_EMB_MUL
  } else {
    for (int idx = tid; idx < 16 * RT * a.H; idx += 256) {
      const int rr = idx / a.H, c = idx - rr * a.H;
      int64_t m = row0 + rr; if (m >= R) m = R - 1;
      xs[rr * XP + c] = a.h[(m / a.N) * a.H + c];
    }
  }
  __syncthreads();
  // hidden layer(s): this workgroup's columns 128 cb + 32 wave + {0, 16}
  const int cbase = blockIdx.y * 128 + 32 * wave;
  an_f4 acc[RT][2];
#pragma unroll
  for (int t = 0; t < RT; ++t) { acc[t][0] = an_f4{0.f, 0.f, 0.f, 0.f}; acc[t][1] = an_f4{0.f, 0.f, 0.f, 0.f}; }
  const bool on0 = cbase < a.HID, on1 = cbase + 16 < a.HID;
  if (on0) {
    const float* w0 = a.wfc + (int64_t)(cbase + r) * a.H + 4 * g;
    const float* w1 = a.wfc + (int64_t)((on1 ? cbase + 16 : cbase) + r) * a.H + 4 * g;
    // K = H in passes of 4 steps (H % 64 == 0), weights double-buffered from L2 while the x rows come from LDS
    const int passes = a.H / 64;
    an_f4 b0[2][4], b1[2][4];
#define AN_HID_LOAD(buf, pass)                                                                   \
  {                                                                                              \
    const int pp = (pass) < passes ? (pass) : 0;                                                 \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                              \
      b0[buf][u] = *(const an_f4*)(w0 + 64 * pp + 16 * u);                                       \
      b1[buf][u] = *(const an_f4*)(w1 + 64 * pp + 16 * u);                                       \
    }                                                                                            \
  }
    // (a pass beyond the end multiplies ZERO x rows: the LDS operand is masked, so no VALU op waits on the global loads)
#define AN_HID_MUL(buf, pass)                                                                    \
  {                                                                                              \
    const bool ok = (pass) < passes;                                                             \
    const int pp = ok ? (pass) : 0;                                                              \
    _Pragma("unroll") for (int u = 0; u < 4; ++u)                                                \
      _Pragma("unroll") for (int t = 0; t < RT; ++t) {                                           \
        an_f4 av = *(const an_f4*)(xs + (16 * t + r) * XP + 64 * pp + 16 * u + 4 * g);           \
        av.x = ok ? av.x : 0.f; av.y = ok ? av.y : 0.f; av.z = ok ? av.z : 0.f; av.w = ok ? av.w : 0.f; \
        AN_MFMA4(acc[t][0], av, b0[buf][u]);                                                     \
        AN_MFMA4(acc[t][1], av, b1[buf][u]);                                                     \
      }                                                                                          \
  }
    AN_HID_LOAD(0, 0)
    for (int p = 0; p < passes; p += 2) {              // straight-line body, as in k_act_lstm
      AN_HID_LOAD(1, p + 1)
      __builtin_amdgcn_sched_barrier(0);
      AN_HID_MUL(0, p)
      __builtin_amdgcn_sched_barrier(0);
      AN_HID_LOAD(0, p + 2)
      __builtin_amdgcn_sched_barrier(0);
      AN_HID_MUL(1, p + 1)
      __builtin_amdgcn_sched_barrier(0);
    }
#undef AN_HID_LOAD
#undef AN_HID_MUL
  }
  // bias + ReLU, then this wave's share of the output layer: out[m][o] += sum_j hid[m][j] * wout[o][j]
  const int c0 = cbase + r, c1 = cbase + 16 + r;
  const float bf0 = on0 ? a.bfc[c0] : 0.f, bf1 = on1 ? a.bfc[c1] : 0.f;
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v0 = acc[t][0][i] + bf0, v1 = acc[t][1][i] + bf1;
      acc[t][0][i] = on0 ? (v0 > 0.f ? v0 : 0.f) : 0.f;
      acc[t][1][i] = on1 ? (v1 > 0.f ? v1 : 0.f) : 0.f;
    }
  for (int o = 0; o < a.NO; ++o) {
    const float wo0 = on0 ? a.wout[(int64_t)o * a.HID + c0] : 0.f, wo1 = on1 ? a.wout[(int64_t)o * a.HID + c1] : 0.f;
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float p = acc[t][0][i] * wo0 + acc[t][1][i] * wo1;
        p = p + __shfl_xor(p, 1); p = p + __shfl_xor(p, 2); p = p + __shfl_xor(p, 4); p = p + __shfl_xor(p, 8);
        if (r == 0) outp[(wave * 16 * RT + 16 * t + 4 * g + i) * a.NOP + o] = p;
      }
  }
  __syncthreads();
  for (int idx = tid; idx < 16 * RT * a.NOP; idx += 256) {
    const int rr = idx / a.NOP, o = idx - rr * a.NOP;
    const int64_t m = row0 + rr;
    if (m < R && o < a.NO) {
      const float s = ((outp[(0 * 16 * RT + rr) * a.NOP + o] + outp[(1 * 16 * RT + rr) * a.NOP + o]) +
                       outp[(2 * 16 * RT + rr) * a.NOP + o]) + outp[(3 * 16 * RT + rr) * a.NOP + o];
      a.part[((int64_t)blockIdx.y * R + m) * a.NOP + o] = s;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// The acting head over the output shares: out[m][o] = bout[o] + sum_p part[p][m][o]; columns 0 .. A-1 are the
// advantages, column A (when has_val) the dueling value; rows are NOP floats apart.  Then exactly k_actor_head (acting.hip): V + A - mean_a A
// (dqn.py:74-87), mean over the N quantile rows (iqn.py actor post-processing), first maximum, epsilon-greedy with
// one Philox4x32-10 block per (step, env) (epsilon_greedy.py:74-100).  One wave per env.
__global__ void __launch_bounds__(256)
k_act_head_select(int E, int N, int A, int P, int NOP, const float* __restrict__ part, const float* __restrict__ bout, int has_val,
                  const double* __restrict__ eps, const double* __restrict__ expo, double eps_min,
                  uint64_t rng_seed, const uint64_t* __restrict__ rng_step,
                  int32_t* __restrict__ actions, float* __restrict__ qvalues) {
  const int lane = threadIdx.x & 63;
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= E) return;
  const int64_t R = (int64_t)E * N;
  auto out = [&](int64_t m, int k) -> float {
    float v = bout[k];
    for (int p = 0; p < P; ++p) v = v + part[((int64_t)p * R + m) * NOP + k];
    return v;
  };
  float best = 0.f; int arg = 0;
  for (int a0 = 0; a0 < A; a0 += 8) {                 // 8 actions per sweep: bounded registers for any A
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    for (int n = lane; n < N; n += 64) {
      const int64_t m = (int64_t)e * N + n;
      float off = 0.f;
      if (has_val) {                                    // dueling: V + A - mean_a A
        float mean = 0.f;
        for (int k = 0; k < A; ++k) mean = mean + out(m, k);
        off = out(m, A) - mean / (float)A;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) if (a0 + k < A) acc[k] = acc[k] + (out(m, a0 + k) + off);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float s = acc[k];
      for (int d = 32; d > 0; d >>= 1) s = s + __shfl_xor(s, d);
      if (a0 + k < A) {
        const float q = s / (float)N;
        if (lane == 0) qvalues[(int64_t)e * A + a0 + k] = q;
        if ((a0 + k == 0) || q > best) { best = q; arg = a0 + k; }     // first maximum, like argmax
      }
    }
  }
  if (lane == 0) {
    int act = arg;
    if (eps) {
      const double pe = pow(*eps, expo ? expo[e] : 1.0);
      const float per = (float)(pe > eps_min ? pe : eps_min);
      uint32_t rn[4];
      philox_4x32(rng_seed, *rng_step, (uint32_t)e, rn);
      const float uf = (float)(rn[0] >> 8) * (1.0f / 16777216.0f);
      if (uf < per) act = (int)(((uint64_t)rn[1] * (uint64_t)A) >> 32);
    }
    actions[e] = act;
  }
}

template <int RT>
static int launch_head_hidden(const ActHeadArgs& a, size_t lds, dim3 grid, hipStream_t st) {
  static bool raised = false;
  if (!raised) {
    MIRL_HIP(hipFuncSetAttribute((const void*)k_act_head_hidden<RT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    raised = true;
  }
  hipLaunchKernelGGL(k_act_head_hidden<RT>, grid, dim3(256), lds, st, a);
  return MIRL_OK;
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
  const int rt = M > 8192 ? 2 : 1;                           // 16-row tiles while they are what fills the chip
  const unsigned grid = (unsigned)((M + 16 * rt - 1) / (16 * rt));
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(layer == 2 ? "k_act_conv2" : "k_act_conv3", 4.0 * ((double)frames * Hi * Wi * Ci + (double)M * 64 + 64.0 * k * k * Ci), st,
               2.0 * (double)M * 64 * k * k * Ci);
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
  return E > 0 && E <= 64 && H > 0 && (H % 4) == 0 && K > 0 && (K % 16) == 0;
}

extern "C" int mirl_act_lstm_fwd(int32_t E, int32_t H, int32_t K, const float* xh, int64_t xh_pitch, const float* w, const float* bias,
                                 const float* c_in, float* h_out, float* c_out, void* stream) {
  if (!xh || !w || !bias || !c_in || !h_out || !c_out) return fail(MIRL_ERR_ARG, "bad act_lstm_fwd arguments");
  if (!mirl_act_lstm_supported(E, H, K)) return fail(MIRL_ERR_ARG, "act_lstm_fwd: E <= 64, H % 4 == 0 and K % 16 == 0 are required");
  if (xh_pitch < K || (xh_pitch % 4) || !an_al16(xh) || !an_al16(w)) return fail(MIRL_ERR_ARG, "act_lstm_fwd: 16-byte aligned rows are required");
  ActLstmArgs a;
  a.xh = xh; a.xh_pitch = xh_pitch; a.w = w; a.bias = bias; a.c_in = c_in; a.h_out = h_out; a.c_out = c_out;
  a.E = E; a.H = H; a.K = K;
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps("k_act_lstm", 4.0 * (4.0 * H * K + (double)E * K + 3.0 * E * H), st, 2.0 * E * 4.0 * H * K);
  const dim3 grid(H / 4);
  if (E <= 16) hipLaunchKernelGGL(k_act_lstm<1>, grid, dim3(512), 0, st, a);
  else if (E <= 32) hipLaunchKernelGGL(k_act_lstm<2>, grid, dim3(512), 0, st, a);
  else hipLaunchKernelGGL(k_act_lstm<4>, grid, dim3(512), 0, st, a);
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}

extern "C" int mirl_act_head_supported(int32_t E, int32_t N, int32_t H, int32_t D, int32_t HID, int32_t NO) {
  if (E <= 0 || N <= 0 || H <= 0 || HID <= 0 || NO <= 0 || NO > 32) return 0;
  if ((H % 64) || (HID % 16) || (D && (D % 16)) || D > 64 || H > 1024) return 0;
  return 1;
}

extern "C" int mirl_act_head_parts(int32_t HID, int32_t NO, int32_t* parts, int32_t* pitch) {
  if (!parts || !pitch || HID <= 0 || NO <= 0 || NO > 32) return fail(MIRL_ERR_ARG, "bad act_head_parts arguments");
  *parts = (HID + 127) / 128;
  *pitch = (NO + 7) / 8 * 8;
  return MIRL_OK;
}

extern "C" int mirl_act_head_hidden(int32_t E, int32_t N, int32_t H, int32_t D, int32_t HID, int32_t NO, const float* h,
                                    const float* freq, const float* taus, uint64_t seed, const uint64_t* step, const float* wq,
                                    const float* bq, const float* wfc, const float* bfc, const float* wout, float* part,
                                    float* tau_out, void* stream) {
  if (!h || !wfc || !bfc || !wout || !part) return fail(MIRL_ERR_ARG, "bad act_head_hidden arguments");
  if (!mirl_act_head_supported(E, N, H, freq ? D : 0, HID, NO)) return fail(MIRL_ERR_ARG, "act_head_hidden: unsupported shape (H % 64, D in {16, 32, 48, 64}, HID % 16; NO <= 32)");
  if (freq && (!wq || !bq || D <= 0 || (!taus && !step))) return fail(MIRL_ERR_ARG, "act_head_hidden: a quantile layer needs wq / bq and taus or a step word");
  if (!an_al16(h) || !an_al16(wfc) || (freq && !an_al16(wq)) || !an_al16(part)) return fail(MIRL_ERR_ARG, "act_head_hidden: 16-byte aligned operands are required");
  ActHeadArgs a;
  a.h = h; a.freq = freq; a.taus = taus; a.wq = wq; a.bq = bq; a.wfc = wfc; a.bfc = bfc; a.wout = wout; a.part = part;
  a.tau_out = tau_out; a.seed = seed; a.step = step;
  a.E = E; a.N = N; a.H = H; a.D = freq ? D : 16; a.HID = HID; a.NO = NO; a.NOP = (NO + 7) / 8 * 8;
  const int64_t R = (int64_t)E * N;
  const int rt = R > 2048 ? 2 : 1;
  const size_t lds = sizeof(float) * ((size_t)16 * rt * (H + 4) + (size_t)16 * rt * (a.D + 4) + (size_t)4 * 16 * rt * a.NOP + 16 * rt);
  const dim3 grid((unsigned)((R + 16 * rt - 1) / (16 * rt)), (unsigned)((HID + 127) / 128));
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps("k_act_head_hidden", 4.0 * ((double)E * H + (double)HID * H + (freq ? (double)H * D : 0.0) + (double)a.NOP * R * grid.y), st,
               2.0 * (double)R * ((double)HID * H + (freq ? (double)grid.y * H * D : 0.0)));
  int rc = rt == 1 ? launch_head_hidden<1>(a, lds, grid, st) : launch_head_hidden<2>(a, lds, grid, st);
  if (rc) return rc;
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}

extern "C" int mirl_act_head_select(int32_t E, int32_t N, int32_t A, int32_t parts, int32_t pitch, const float* part, const float* bout, int32_t has_val,
                                    const double* eps, const double* expo, double eps_min, uint64_t rng_seed, const uint64_t* rng_step,
                                    int32_t* actions, float* qvalues, void* stream) {
  if (E <= 0 || N <= 0 || A <= 0 || A + (has_val ? 1 : 0) > pitch || parts <= 0 || !part || !bout || !actions || !qvalues || (eps && !rng_step))
    return fail(MIRL_ERR_ARG, "bad act_head_select arguments");
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps("k_act_head_select", 0.0, st);
  hipLaunchKernelGGL(k_act_head_select, dim3((E + 3) / 4), dim3(256), 0, st, (int)E, (int)N, (int)A, (int)parts, (int)pitch, part, bout, (int)has_val,
                     eps, expo, eps_min, rng_seed, rng_step, actions, qvalues);
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}
