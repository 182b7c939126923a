// nnops.hip — the non-contraction glue around the network's GEMMs / convolutions.
//
// The dense contractions of the policy (conv, LSTM, linear) stay on
// MIOpen / hipBLASLt (MFMA); what surrounds them in the reference
// (rltime/models/torch/modules/cnn.py:47-49 `F.relu(conv(x))`,
// rltime/policies/torch/iqn.py:67-106 cos-embedding * state features, the
// autograd passes PyTorch generates for them) is pure streaming work over
// multi-GB activations, one HBM pass per PyTorch op.  These kernels do each
// group in ONE pass:
//   k_bias_relu_rows       y <- relu(y + b)            in place, after a bias-less conv / GEMM
//   k_relu_bwd_bias_rows   g = dy * (y > 0), db += colsum(g)     (mask + bias-gradient in one read)
//   k_cos_embed            phi[r][i] = cos(tau[r] * (i+1) pi)    (iqn.py:78-81)
//   k_iqn_mul_bwd          backward of  out = x[m] * relu_emb[m*N+n]  fused with the ReLU mask
//                          of the embedding layer and its bias gradient
// All tensors are viewed as row-major (rows, C) with the channel / feature index
// fastest (NHWC activations, (M, features) matrices); every global access is
// 16 B per lane on a linear address stream.  Column sums are two-stage and
// deterministic (fixed partition, fixed order — no float atomics), so repeated
// runs give bit-identical gradients.
#include "common.hpp"
#include "philox.hpp"

namespace mirl {

typedef float nn_f4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256)
k_bias_relu_rows(nn_f4* __restrict__ y, const nn_f4* __restrict__ bias, int64_t n4, int CQ) {
  // 4 quads per lane, all loads before the first store
  int64_t i = ((int64_t)blockIdx.x * 256 * 4) + threadIdx.x;
  nn_f4 v[4]; bool h[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { h[k] = i + k * 256 < n4; if (h[k]) v[k] = y[i + k * 256]; }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (!h[k]) continue;
    const int64_t q = i + k * 256;
    nn_f4 b = bias[(CQ & (CQ - 1)) ? (int)(q % CQ) : (int)(q & (CQ - 1))];   // C/4 is a power of two for every shipped model
    nn_f4 r = v[k] + b;
    r.x = r.x > 0.f ? r.x : 0.f; r.y = r.y > 0.f ? r.y : 0.f; r.z = r.z > 0.f ? r.z : 0.f; r.w = r.w > 0.f ? r.w : 0.f;
    y[i + k * 256] = r;
  }
}

// generic C (not a multiple of 4): scalar
__global__ void __launch_bounds__(256)
k_bias_relu_rows_any(float* __restrict__ y, const float* __restrict__ bias, int64_t n, int C) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float r = y[i] + bias[i % C];
  y[i] = r > 0.f ? r : 0.f;
}

// Block b streams rows [b*rpb, (b+1)*rpb): the linear quad index advances by 256
// per iteration and 256 % CQ == 0, so a lane keeps its column quad for the whole
// sweep and accumulates its column sums in registers.
__global__ void __launch_bounds__(256)
k_relu_bwd_bias_rows(const nn_f4* __restrict__ dy, const nn_f4* __restrict__ y, nn_f4* __restrict__ g,
                     nn_f4* __restrict__ partial, int64_t rows, int CQ, int64_t rpb) {
  __shared__ nn_f4 s_acc[256];
  const int tid = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * rpb;
  int64_t r1 = r0 + rpb; if (r1 > rows) r1 = rows;
  const int64_t lo = r0 * CQ, hi = r1 * CQ;
  nn_f4 acc = {0.f, 0.f, 0.f, 0.f};
  int64_t i = lo + tid;
  for (; i + 768 < hi; i += 1024) {
    nn_f4 a0 = dy[i], a1 = dy[i + 256], a2 = dy[i + 512], a3 = dy[i + 768];
    nn_f4 b0 = y[i], b1 = y[i + 256], b2 = y[i + 512], b3 = y[i + 768];
#define MIRL_MASK(a, b) a.x = b.x > 0.f ? a.x : 0.f; a.y = b.y > 0.f ? a.y : 0.f; a.z = b.z > 0.f ? a.z : 0.f; a.w = b.w > 0.f ? a.w : 0.f;
    MIRL_MASK(a0, b0) MIRL_MASK(a1, b1) MIRL_MASK(a2, b2) MIRL_MASK(a3, b3)
    __builtin_nontemporal_store(a0, g + i); __builtin_nontemporal_store(a1, g + i + 256);
    __builtin_nontemporal_store(a2, g + i + 512); __builtin_nontemporal_store(a3, g + i + 768);
    acc = acc + a0; acc = acc + a1; acc = acc + a2; acc = acc + a3;
  }
  for (; i < hi; i += 256) {
    nn_f4 a0 = dy[i], b0 = y[i];
    MIRL_MASK(a0, b0)
    g[i] = a0;
    acc = acc + a0;
  }
  s_acc[tid] = acc;
  __syncthreads();
  if (tid < CQ) {
    nn_f4 t = s_acc[tid];
    for (int k = tid + CQ; k < 256; k += CQ) t = t + s_acc[k];
    partial[(int64_t)blockIdx.x * CQ + tid] = t;
  }
}

// db[c] = sum over the per-block partials in a FIXED order: one workgroup per 64
// columns, its 16 waves take interleaved blocks (b = w, w + 16, ...) with 64 coalesced
// columns per load and 8 loads in flight, then the 16 wave sums are added in wave order
// (round 2: 4 waves, 4 loads in flight — 59 us for 2048 x 512 partials, pure load latency).
__global__ void __launch_bounds__(1024)
k_colsum_partials(const float* __restrict__ partial, float* __restrict__ out, int blocks, int C) {
  __shared__ float s_part[16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (c < C) {
    int b = w;
    for (; b + 7 * 16 < blocks; b += 8 * 16) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = partial[(int64_t)(b + 16 * k) * C + c];
#pragma unroll
      for (int k = 0; k < 8; ++k) s = s + v[k];
    }
    for (; b < blocks; b += 16) s = s + partial[(int64_t)b * C + c];
  }
  s_part[w][lane] = s;
  __syncthreads();
  if (w == 0 && c < C) {
    float t = s_part[0][lane];
#pragma unroll
    for (int k = 1; k < 16; ++k) t = t + s_part[k][lane];
    out[c] = t;
  }
}

// phi[r][i] = cos(tau[r] * w[i]),  w[i] = (i+1) * pi rounded to f32 (the
// `embedding_range * np.pi` tensor of iqn.py:78).  One lane per 4 features.
__global__ void __launch_bounds__(256)
k_cos_embed(const float* __restrict__ tau, const nn_f4* __restrict__ w, nn_f4* __restrict__ phi, int64_t n4, int DQ) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float t = tau[i / DQ];
  const nn_f4 ww = w[i % DQ];
  nn_f4 r;
  r.x = cosf(ww.x * t); r.y = cosf(ww.y * t); r.z = cosf(ww.z * t); r.w = cosf(ww.w * t);
  phi[i] = r;
}

// The same features with tau drawn inside the kernel (Philox4x32-10 keyed by (seed, *step, row);
// 24-bit uniforms in [0, 1) like torch.rand): the acting path's quantile fractions without a
// torch.rand launch (and its two graph-RNG bookkeeping fills) per vector step.
__global__ void __launch_bounds__(256)
k_cos_embed_rng(uint64_t seed, const uint64_t* __restrict__ step, const nn_f4* __restrict__ w, nn_f4* __restrict__ phi,
                float* __restrict__ tau_out, int64_t n4, int DQ) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const int64_t row = i / DQ;
  uint32_t r[4];
  philox_4x32(seed ^ 0x7A5ull, *step, (uint32_t)row, r);
  const float t = (float)(r[0] >> 8) * (1.0f / 16777216.0f);
  const nn_f4 ww = w[i % DQ];
  nn_f4 o;
  o.x = cosf(ww.x * t); o.y = cosf(ww.y * t); o.z = cosf(ww.z * t); o.w = cosf(ww.w * t);
  phi[i] = o;
  if (tau_out && i % DQ == 0) tau_out[row] = t;
}

// out[(m*N+n)][c] = x[m][c] * emb[(m*N+n)][c]   (iqn.py:84,102 without the repeated copy of x).
// Same walk as the backward below: a workgroup takes whole groups m, lane (rl, cq)
// rows n = rl, rl + RL, ... of the group, so x[m] is read once per lane and no
// index division is needed.
// INPLACE: out == emb (no-grad passes: the embedding is not needed again) — a read-modify-write
// stream like k_bias_relu_rows (0.79 of the HBM peak on this part against 0.62 for two streams).
template <bool INPLACE>
__global__ void __launch_bounds__(256)
k_iqn_mul_fwd(const nn_f4* __restrict__ x, const nn_f4* emb, nn_f4* out,
              int64_t M, int N, int CQ, int64_t gpb) {
  const int tid = threadIdx.x, cq = tid % CQ, rl = tid / CQ, RL = 256 / CQ;
  const int64_t m0 = (int64_t)blockIdx.x * gpb;
  int64_t m1 = m0 + gpb; if (m1 > M) m1 = M;
  const int64_t st = (int64_t)RL * CQ;
  for (int64_t m = m0; m < m1; ++m) {
    const nn_f4 xv = x[m * CQ + cq];
    const int64_t base = m * N * (int64_t)CQ + cq;
    int n = rl;
    for (; n + 3 * RL < N; n += 4 * RL) {
      const int64_t i0 = base + (int64_t)n * CQ;
      nn_f4 e0 = emb[i0], e1 = emb[i0 + st], e2 = emb[i0 + 2 * st], e3 = emb[i0 + 3 * st];
      if (INPLACE) { out[i0] = e0 * xv; out[i0 + st] = e1 * xv; out[i0 + 2 * st] = e2 * xv; out[i0 + 3 * st] = e3 * xv; }
      else {
        __builtin_nontemporal_store(e0 * xv, out + i0); __builtin_nontemporal_store(e1 * xv, out + i0 + st);
        __builtin_nontemporal_store(e2 * xv, out + i0 + 2 * st); __builtin_nontemporal_store(e3 * xv, out + i0 + 3 * st);
      }
    }
    for (; n < N; n += RL) { const int64_t i0 = base + (int64_t)n * CQ; out[i0] = emb[i0] * xv; }
  }
}

// Backward of the product above fused with the ReLU mask of the embedding layer
// (emb = relu(pre)) and its bias gradient.  One workgroup walks whole groups m
// (the N quantile rows of one state, contiguous): lane (rl, cq) takes rows
// n = rl, rl + RL, ... of the group for column quad cq.
//   d_pre[r][c] = emb[r][c] > 0 ? g[r][c] * x[m][c] : 0
//   dx[m][c]    = sum_n g[r][c] * emb[r][c]          (fixed order: lanes, then LDS)
//   partial[b]  = column sums of d_pre over the block's groups
__global__ void __launch_bounds__(256)
k_iqn_mul_bwd(const nn_f4* __restrict__ g, const nn_f4* __restrict__ emb, const nn_f4* __restrict__ x,
              nn_f4* __restrict__ d_pre, nn_f4* __restrict__ dx, nn_f4* __restrict__ partial,
              int64_t M, int N, int CQ, int64_t gpb) {
  __shared__ nn_f4 s_dx[256];
  __shared__ nn_f4 s_db[256];
  const int tid = threadIdx.x, cq = tid % CQ, rl = tid / CQ, RL = 256 / CQ;
  const int64_t m0 = (int64_t)blockIdx.x * gpb;
  int64_t m1 = m0 + gpb; if (m1 > M) m1 = M;
  nn_f4 db = {0.f, 0.f, 0.f, 0.f};
  for (int64_t m = m0; m < m1; ++m) {
    const nn_f4 xv = x[m * CQ + cq];
    nn_f4 acc = {0.f, 0.f, 0.f, 0.f};
    const int64_t base = m * N * (int64_t)CQ + cq;
    int n = rl;
    for (; n + 3 * RL < N; n += 4 * RL) {
      const int64_t i0 = base + (int64_t)n * CQ, st = (int64_t)RL * CQ;
      nn_f4 g0 = g[i0], g1 = g[i0 + st], g2 = g[i0 + 2 * st], g3 = g[i0 + 3 * st];
      nn_f4 e0 = emb[i0], e1 = emb[i0 + st], e2 = emb[i0 + 2 * st], e3 = emb[i0 + 3 * st];
      acc = acc + g0 * e0; acc = acc + g1 * e1; acc = acc + g2 * e2; acc = acc + g3 * e3;
      nn_f4 p0 = g0 * xv, p1 = g1 * xv, p2 = g2 * xv, p3 = g3 * xv;
      MIRL_MASK(p0, e0) MIRL_MASK(p1, e1) MIRL_MASK(p2, e2) MIRL_MASK(p3, e3)
      __builtin_nontemporal_store(p0, d_pre + i0); __builtin_nontemporal_store(p1, d_pre + i0 + st);
      __builtin_nontemporal_store(p2, d_pre + i0 + 2 * st); __builtin_nontemporal_store(p3, d_pre + i0 + 3 * st);
      db = db + p0; db = db + p1; db = db + p2; db = db + p3;
    }
    for (; n < N; n += RL) {
      const int64_t i0 = base + (int64_t)n * CQ;
      nn_f4 g0 = g[i0], e0 = emb[i0];
      acc = acc + g0 * e0;
      nn_f4 p0 = g0 * xv;
      MIRL_MASK(p0, e0)
      d_pre[i0] = p0;
      db = db + p0;
    }
    if (RL > 1) {
      __syncthreads();                       // previous group's readers are done
      s_dx[tid] = acc;
      __syncthreads();
      if (rl == 0) {
        nn_f4 t = s_dx[cq];
        for (int k = 1; k < RL; ++k) t = t + s_dx[k * CQ + cq];
        dx[m * CQ + cq] = t;
      }
    } else {
      dx[m * CQ + cq] = acc;
    }
  }
  __syncthreads();
  s_db[tid] = db;
  __syncthreads();
  if (rl == 0) {
    nn_f4 t = s_db[cq];
    for (int k = 1; k < RL; ++k) t = t + s_db[k * CQ + cq];
    partial[(int64_t)blockIdx.x * CQ + cq] = t;
  }
}

// Backward of the dueling tail's two small output layers fused with the ReLU mask
// of the joint hidden activation and its bias gradient:
//   d_both[r][c] = sum_a ga[r][a] * wo[a][c]            (c <  H1: advantage branch, a < A)
//                = sum_q gv[r][q] * wq[q][c - H1]       (c >= H1: value branch,     q < Q)
//   g[r][c]      = both[r][c] > 0 ? d_both[r][c] : 0 ;  db[c] = sum_r g[r][c]
// PyTorch runs two K<=A GEMMs that WRITE the (M, H1+Hv) gradient (5.4 GB at the
// benchmark shape) and a mask pass that re-reads it; here the K<=16 dot products sit
// in registers, `both` is read once and g written once.  Lane = one column quad for
// the whole sweep (its 4 x A weights stay in registers), block = a row range.
#define MIRL_TAIL_MAXK 16
#define MIRL_TAIL_ROWS 16
__global__ void __launch_bounds__(256)
k_tail_bwd(const float* __restrict__ ga, const float* __restrict__ gv, const float* __restrict__ wo,
           const float* __restrict__ wq, const nn_f4* __restrict__ both, nn_f4* __restrict__ g,
           nn_f4* __restrict__ partial, int64_t rows, int H1, int Hv, int A, int Q, int64_t rpb) {
  __shared__ nn_f4 s_acc[256];
  __shared__ float s_ga[MIRL_TAIL_ROWS * MIRL_TAIL_MAXK];
  __shared__ float s_gv[MIRL_TAIL_ROWS * MIRL_TAIL_MAXK];
  const int C = H1 + Hv, CQ = C / 4, tid = threadIdx.x, cq = tid % CQ, rl = tid / CQ, RL = 256 / CQ;
  const int c0 = cq * 4;
  const bool adv = c0 < H1;
  const int K = adv ? A : Q;
  const float* w = adv ? wo + c0 : wq + (c0 - H1);
  const int ldw = adv ? H1 : Hv;
  const float* sg = adv ? s_ga : s_gv;
  nn_f4 wk[MIRL_TAIL_MAXK];
#pragma unroll
  for (int k = 0; k < MIRL_TAIL_MAXK; ++k)
    wk[k] = k < K ? nn_f4{w[(int64_t)k * ldw], w[(int64_t)k * ldw + 1], w[(int64_t)k * ldw + 2], w[(int64_t)k * ldw + 3]}
                  : nn_f4{0.f, 0.f, 0.f, 0.f};
  const int64_t r0 = (int64_t)blockIdx.x * rpb;
  int64_t r1 = r0 + rpb; if (r1 > rows) r1 = rows;
  nn_f4 acc = {0.f, 0.f, 0.f, 0.f};
  // chunks of 16 rows: their (16 x A) and (16 x Q) output-gradient rows are staged in
  // LDS with one coalesced load, every lane then reads them as LDS broadcasts
  for (int64_t base = r0; base < r1; base += MIRL_TAIL_ROWS) {
    const int nr = r1 - base < MIRL_TAIL_ROWS ? (int)(r1 - base) : MIRL_TAIL_ROWS;
    __syncthreads();
    for (int i = tid; i < nr * A; i += 256) s_ga[i] = ga[base * A + i];
    for (int i = tid; i < nr * Q; i += 256) s_gv[i] = gv[base * Q + i];
    __syncthreads();
    int u = rl;
    for (; u + 3 * RL < nr; u += 4 * RL) {               // four rows in flight per lane
      nn_f4 b[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) b[v] = both[(base + u + v * RL) * CQ + cq];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const float* gr = sg + (u + v * RL) * K;
        nn_f4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < MIRL_TAIL_MAXK; ++k) if (k < K) d = d + wk[k] * gr[k];
        MIRL_MASK(d, b[v])
        __builtin_nontemporal_store(d, g + (base + u + v * RL) * CQ + cq);
        acc = acc + d;
      }
    }
    for (; u < nr; u += RL) {
      const nn_f4 b = both[(base + u) * CQ + cq];
      const float* gr = sg + u * K;
      nn_f4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < MIRL_TAIL_MAXK; ++k) if (k < K) d = d + wk[k] * gr[k];
      MIRL_MASK(d, b)
      g[(base + u) * CQ + cq] = d;
      acc = acc + d;
    }
  }
  __syncthreads();
  s_acc[tid] = acc;
  __syncthreads();
  if (rl == 0) {
    nn_f4 t = s_acc[cq];
    for (int k = 1; k < RL; ++k) t = t + s_acc[k * CQ + cq];
    partial[(int64_t)blockIdx.x * CQ + cq] = t;
  }
}

// The same pass with the output layers' weight gradients riding along:
//   dwo[a][c] = sum_r ga[r][a] * both[r][c] (c < H1),  dwq[q][c - H1] = sum_r gv[r][q] * both[r][c] (c >= H1)
// `both` is in registers anyway, so the two (A | Q) x rows x H library GEMMs that re-read it (3.2 ms per step at the
// benchmark shape, 3x their HBM floor) become K more FMAs per element here.  K is padded to 8 with zero weights and
// zero output-gradient columns so every inner loop is unconditional; per-block partials [block][k][C], k < KW =
// max(A, Q) <= 8, are summed in a fixed order afterwards.
#define MIRL_TAIL_WGK 8
__global__ void __launch_bounds__(256)
k_tail_bwd_w(const float* __restrict__ ga, const float* __restrict__ gv, const float* __restrict__ wo,
             const float* __restrict__ wq, const nn_f4* __restrict__ both, nn_f4* __restrict__ g,
             nn_f4* __restrict__ partial, nn_f4* __restrict__ partial_w, int64_t rows, int H1, int Hv, int A, int Q,
             int KW, int64_t rpb) {
  __shared__ nn_f4 s_acc[256];
  __shared__ __attribute__((aligned(16))) float s_ga[MIRL_TAIL_ROWS * MIRL_TAIL_WGK];
  __shared__ __attribute__((aligned(16))) float s_gv[MIRL_TAIL_ROWS * MIRL_TAIL_WGK];
  const int C = H1 + Hv, CQ = C / 4, tid = threadIdx.x, cq = tid % CQ, rl = tid / CQ, RL = 256 / CQ;
  const int c0 = cq * 4;
  const bool adv = c0 < H1;
  const int K = adv ? A : Q;
  const float* w = adv ? wo + c0 : wq + (c0 - H1);
  const int ldw = adv ? H1 : Hv;
  const float* sg = adv ? s_ga : s_gv;
  nn_f4 wk[MIRL_TAIL_WGK], aw[MIRL_TAIL_WGK];
#pragma unroll
  for (int k = 0; k < MIRL_TAIL_WGK; ++k) {
    wk[k] = k < K ? nn_f4{w[(int64_t)k * ldw], w[(int64_t)k * ldw + 1], w[(int64_t)k * ldw + 2], w[(int64_t)k * ldw + 3]}
                  : nn_f4{0.f, 0.f, 0.f, 0.f};
    aw[k] = nn_f4{0.f, 0.f, 0.f, 0.f};
  }
  if (tid < MIRL_TAIL_ROWS * MIRL_TAIL_WGK) { s_ga[tid] = 0.f; s_gv[tid] = 0.f; }     // pad columns stay zero
  const int64_t r0 = (int64_t)blockIdx.x * rpb;
  int64_t r1 = r0 + rpb; if (r1 > rows) r1 = rows;
  nn_f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int64_t base = r0; base < r1; base += MIRL_TAIL_ROWS) {
    const int nr = r1 - base < MIRL_TAIL_ROWS ? (int)(r1 - base) : MIRL_TAIL_ROWS;
    __syncthreads();
    for (int i = tid; i < nr * A; i += 256) s_ga[(i / A) * MIRL_TAIL_WGK + i % A] = ga[base * A + i];
    for (int i = tid; i < nr * Q; i += 256) s_gv[(i / Q) * MIRL_TAIL_WGK + i % Q] = gv[base * Q + i];
    __syncthreads();
    for (int u = rl; u < nr; u += 2 * RL) {               // two rows in flight per lane
      const bool two = u + RL < nr;
      const nn_f4 b0 = both[(base + u) * CQ + cq];
      const nn_f4 b1 = two ? both[(base + u + RL) * CQ + cq] : nn_f4{0.f, 0.f, 0.f, 0.f};
      const nn_f4* g0 = reinterpret_cast<const nn_f4*>(sg + u * MIRL_TAIL_WGK);
      const nn_f4* g1 = reinterpret_cast<const nn_f4*>(sg + (two ? u + RL : u) * MIRL_TAIL_WGK);
      const nn_f4 ga0 = g0[0], gb0 = g0[1], ga1 = g1[0], gb1 = g1[1];
      const float gr0[8] = {ga0.x, ga0.y, ga0.z, ga0.w, gb0.x, gb0.y, gb0.z, gb0.w};
      const float gr1[8] = {ga1.x, ga1.y, ga1.z, ga1.w, gb1.x, gb1.y, gb1.z, gb1.w};
      nn_f4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < MIRL_TAIL_WGK; ++k) {
        d0 = d0 + wk[k] * gr0[k];
        d1 = d1 + wk[k] * gr1[k];
        aw[k] = aw[k] + b0 * gr0[k];
        aw[k] = aw[k] + b1 * gr1[k];                       // b1 is zero when there is no second row
      }
      MIRL_MASK(d0, b0)
      __builtin_nontemporal_store(d0, g + (base + u) * CQ + cq);
      acc = acc + d0;
      if (two) {
        MIRL_MASK(d1, b1)
        __builtin_nontemporal_store(d1, g + (base + u + RL) * CQ + cq);
        acc = acc + d1;
      }
    }
  }
  __syncthreads();
  s_acc[tid] = acc;
  __syncthreads();
  if (rl == 0) {
    nn_f4 t = s_acc[cq];
    for (int k = 1; k < RL; ++k) t = t + s_acc[k * CQ + cq];
    partial[(int64_t)blockIdx.x * CQ + cq] = t;
  }
#pragma unroll
  for (int k = 0; k < MIRL_TAIL_WGK; ++k) {
    if (k < KW) {                                          // block-uniform
      __syncthreads();
      s_acc[tid] = aw[k];                                  // lanes of the other branch hold zeros for k >= their K
      __syncthreads();
      if (rl == 0) {
        nn_f4 t = s_acc[cq];
        for (int j2 = 1; j2 < RL; ++j2) t = t + s_acc[j2 * CQ + cq];
        partial_w[((int64_t)blockIdx.x * KW + k) * CQ + cq] = t;
      }
    }
  }
}

static inline bool pow2_quads(int C) { int cq = C / 4; return (C % 4) == 0 && cq >= 1 && cq <= 256 && (256 % cq) == 0; }
static inline bool aligned16(const void* p) { return ((uintptr_t)p % 16) == 0; }

}  // namespace mirl

using namespace mirl;

extern "C" int mirl_bias_relu_rows(int64_t rows, int32_t C, float* y, const float* bias, void* stream) {
  if (rows <= 0 || C <= 0 || !y || !bias) return fail(MIRL_ERR_ARG, "bad bias_relu_rows arguments");
  hipStream_t st = (hipStream_t)stream;
  const int64_t n = rows * C;
  ProfScope ps("k_bias_relu_rows", 2.0 * n * 4, st);
  if ((C % 4) == 0 && aligned16(y) && aligned16(bias)) {
    const int64_t n4 = n / 4, blocks = (n4 + 1023) / 1024;
    if (blocks >= (1LL << 31)) return fail(MIRL_ERR_ARG, "bias_relu_rows: tensor too large for one launch");
    hipLaunchKernelGGL(k_bias_relu_rows, dim3((unsigned)blocks), dim3(256), 0, st, (nn_f4*)y, (const nn_f4*)bias, n4, C / 4);
  } else {
    const int64_t blocks = (n + 255) / 256;
    if (blocks >= (1LL << 31)) return fail(MIRL_ERR_ARG, "bias_relu_rows: tensor too large for one launch");
    hipLaunchKernelGGL(k_bias_relu_rows_any, dim3((unsigned)blocks), dim3(256), 0, st, y, bias, n, (int)C);
  }
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}

extern "C" int mirl_colsum_blocks(int64_t rows, int32_t C, int32_t* blocks) {
  if (!blocks) return fail(MIRL_ERR_ARG, "null argument");
  // enough workgroups to fill 256 CUs several times over, each with >= 64 rows
  int64_t b = (rows + 63) / 64;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  *blocks = (int32_t)b;
  (void)C;
  return MIRL_OK;
}

extern "C" int mirl_relu_bwd_bias_rows(int64_t rows, int32_t C, const float* dy, const float* y, float* g, float* db,
                                       float* partial, int32_t blocks, void* stream) {
  if (rows <= 0 || C <= 0 || !dy || !y || !g || !db || !partial || blocks <= 0) return fail(MIRL_ERR_ARG, "bad relu_bwd_bias_rows arguments");
  if (!pow2_quads(C) || !aligned16(dy) || !aligned16(y) || !aligned16(g) || !aligned16(partial))
    return fail(MIRL_ERR_ARG, "relu_bwd_bias_rows: C must be 4 * a power of two <= 1024 and pointers 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int64_t rpb = (rows + blocks - 1) / blocks;
  {
    ProfScope ps("k_relu_bwd_bias_rows", 3.0 * rows * C * 4, st);
    hipLaunchKernelGGL(k_relu_bwd_bias_rows, dim3((unsigned)blocks), dim3(256), 0, st, (const nn_f4*)dy, (const nn_f4*)y, (nn_f4*)g,
                       (nn_f4*)partial, rows, C / 4, rpb);
  }
  MIRL_LAUNCH_CHECK();
  {
    ProfScope ps("k_colsum_partials", (double)blocks * C * 4, st);
    hipLaunchKernelGGL(k_colsum_partials, dim3((unsigned)((C + 63) / 64)), dim3(1024), 0, st, partial, db, (int)blocks, (int)C);
  }
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}

extern "C" int mirl_cos_embed(int64_t rows, int32_t D, const float* tau, const float* freq, float* phi, void* stream) {
  if (rows <= 0 || D <= 0 || (D % 4) || !tau || !freq || !phi || !aligned16(freq) || !aligned16(phi))
    return fail(MIRL_ERR_ARG, "bad cos_embed arguments (embedding_dim must be a multiple of 4)");
  hipStream_t st = (hipStream_t)stream;
  const int64_t n4 = rows * (D / 4), blocks = (n4 + 255) / 256;
  if (blocks >= (1LL << 31)) return fail(MIRL_ERR_ARG, "cos_embed: tensor too large for one launch");
  ProfScope ps("k_cos_embed", (double)rows * (D * 4 + 4), st);
  hipLaunchKernelGGL(k_cos_embed, dim3((unsigned)blocks), dim3(256), 0, st, tau, (const nn_f4*)freq, (nn_f4*)phi, n4, D / 4);
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}

extern "C" int mirl_cos_embed_rng(int64_t rows, int32_t D, uint64_t seed, const uint64_t* step, const float* freq, float* phi,
                                  float* tau_out, void* stream) {
  if (rows <= 0 || rows >= (1LL << 32) || D <= 0 || (D % 4) || !step || !freq || !phi || !aligned16(freq) || !aligned16(phi))
    return fail(MIRL_ERR_ARG, "bad cos_embed_rng arguments (embedding_dim must be a multiple of 4)");
  hipStream_t st = (hipStream_t)stream;
  const int64_t n4 = rows * (D / 4), blocks = (n4 + 255) / 256;
  if (blocks >= (1LL << 31)) return fail(MIRL_ERR_ARG, "cos_embed_rng: tensor too large for one launch");
  ProfScope ps("k_cos_embed", (double)rows * (D * 4 + 4), st);
  hipLaunchKernelGGL(k_cos_embed_rng, dim3((unsigned)blocks), dim3(256), 0, st, seed, step, (const nn_f4*)freq, (nn_f4*)phi, tau_out, n4, D / 4);
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}

extern "C" int mirl_iqn_mul_fwd(int64_t M, int32_t N, int32_t C, const float* x, const float* emb, float* out, void* stream) {
  if (M <= 0 || N <= 0 || C <= 0 || !x || !emb || !out) return fail(MIRL_ERR_ARG, "bad iqn_mul_fwd arguments");
  if (!pow2_quads(C) || !aligned16(x) || !aligned16(emb) || !aligned16(out))
    return fail(MIRL_ERR_ARG, "iqn_mul_fwd: C must be 4 * a power of two <= 1024 and pointers 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  int64_t blocks = M < 4096 ? M : 4096;
  const int64_t gpb = (M + blocks - 1) / blocks;
  ProfScope ps("k_iqn_mul_fwd", (2.0 * M * N + M) * C * 4, st);
  if ((const void*)emb == (const void*)out)
    hipLaunchKernelGGL(k_iqn_mul_fwd<true>, dim3((unsigned)blocks), dim3(256), 0, st, (const nn_f4*)x, (const nn_f4*)emb, (nn_f4*)out, M, (int)N, C / 4, gpb);
  else
    hipLaunchKernelGGL(k_iqn_mul_fwd<false>, dim3((unsigned)blocks), dim3(256), 0, st, (const nn_f4*)x, (const nn_f4*)emb, (nn_f4*)out, M, (int)N, C / 4, gpb);
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}

extern "C" int mirl_iqn_mul_bwd(int64_t M, int32_t N, int32_t C, const float* g, const float* emb, const float* x,
                                float* d_pre, float* dx, float* db, float* partial, int32_t blocks, void* stream) {
  if (M <= 0 || N <= 0 || C <= 0 || !g || !emb || !x || !d_pre || !dx || !db || !partial || blocks <= 0)
    return fail(MIRL_ERR_ARG, "bad iqn_mul_bwd arguments");
  if (!pow2_quads(C) || !aligned16(g) || !aligned16(emb) || !aligned16(x) || !aligned16(d_pre) || !aligned16(dx) || !aligned16(partial))
    return fail(MIRL_ERR_ARG, "iqn_mul_bwd: C must be 4 * a power of two <= 1024 and pointers 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int64_t gpb = (M + blocks - 1) / blocks;
  {
    ProfScope ps("k_iqn_mul_bwd", (3.0 * M * N + 2.0 * M) * C * 4, st);
    hipLaunchKernelGGL(k_iqn_mul_bwd, dim3((unsigned)blocks), dim3(256), 0, st, (const nn_f4*)g, (const nn_f4*)emb, (const nn_f4*)x,
                       (nn_f4*)d_pre, (nn_f4*)dx, (nn_f4*)partial, M, (int)N, C / 4, gpb);
  }
  MIRL_LAUNCH_CHECK();
  {
    ProfScope ps("k_colsum_partials", (double)blocks * C * 4, st);
    hipLaunchKernelGGL(k_colsum_partials, dim3((unsigned)((C + 63) / 64)), dim3(1024), 0, st, partial, db, (int)blocks, (int)C);
  }
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}

static int tail_bwd_launch(int64_t M, int32_t H1, int32_t Hv, int32_t A, int32_t Q, const float* ga, const float* gv,
                           const float* wo, const float* wq, const float* both, float* g, float* db, float* partial,
                           int32_t blocks, float* dwj, float* partial_w, void* stream) {
  using namespace mirl;
  if (M <= 0 || H1 <= 0 || Hv <= 0 || A <= 0 || Q <= 0 || !ga || !gv || !wo || !wq || !both || !g || !db || !partial || blocks <= 0)
    return fail(MIRL_ERR_ARG, "bad dueling_tail_bwd arguments");
  if (A > MIRL_TAIL_MAXK || Q > MIRL_TAIL_MAXK || (H1 % 4) || !pow2_quads(H1 + Hv) || !aligned16(both) || !aligned16(g) || !aligned16(partial))
    return fail(MIRL_ERR_ARG, "dueling_tail_bwd: outputs per branch <= 16, H1 % 4 == 0, H1 + Hv = 4 * 2^k <= 1024, 16-byte aligned pointers");
  const bool wgrad = dwj != nullptr;
  const int KW = A > Q ? A : Q;
  if (wgrad && (KW > MIRL_TAIL_WGK || !partial_w || !aligned16(partial_w) || !aligned16(dwj)))
    return fail(MIRL_ERR_ARG, "dueling_tail_bwd_w: outputs per branch <= 8, 16-byte aligned buffers");
  hipStream_t st = (hipStream_t)stream;
  const int C = H1 + Hv;
  const int64_t rpb = (M + blocks - 1) / blocks;
  {
    ProfScope ps("k_tail_bwd", 2.0 * M * C * 4 + (double)M * (A + Q) * 4, st);
    if (wgrad)
      hipLaunchKernelGGL(k_tail_bwd_w, dim3((unsigned)blocks), dim3(256), 0, st, ga, gv, wo, wq, (const nn_f4*)both, (nn_f4*)g, (nn_f4*)partial,
                         (nn_f4*)partial_w, M, (int)H1, (int)Hv, (int)A, (int)Q, KW, rpb);
    else
      hipLaunchKernelGGL(k_tail_bwd, dim3((unsigned)blocks), dim3(256), 0, st, ga, gv, wo, wq, (const nn_f4*)both, (nn_f4*)g, (nn_f4*)partial,
                         M, (int)H1, (int)Hv, (int)A, (int)Q, rpb);
  }
  MIRL_LAUNCH_CHECK();
  {
    ProfScope ps("k_colsum_partials", (double)blocks * C * 4, st);
    hipLaunchKernelGGL(k_colsum_partials, dim3((unsigned)((C + 63) / 64)), dim3(1024), 0, st, partial, db, (int)blocks, (int)C);
  }
  MIRL_LAUNCH_CHECK();
  if (wgrad) {
    ProfScope ps("k_colsum_partials", (double)blocks * KW * C * 4, st);
    hipLaunchKernelGGL(k_colsum_partials, dim3((unsigned)((KW * C + 63) / 64)), dim3(1024), 0, st, partial_w, dwj, (int)blocks, (int)(KW * C));
    MIRL_LAUNCH_CHECK();
  }
  return MIRL_OK;
}

extern "C" int mirl_dueling_tail_bwd(int64_t M, int32_t H1, int32_t Hv, int32_t A, int32_t Q, const float* ga, const float* gv,
                                     const float* wo, const float* wq, const float* both, float* g, float* db, float* partial,
                                     int32_t blocks, void* stream) {
  return tail_bwd_launch(M, H1, Hv, A, Q, ga, gv, wo, wq, both, g, db, partial, blocks, nullptr, nullptr, stream);
}

extern "C" int mirl_dueling_tail_bwd_w(int64_t M, int32_t H1, int32_t Hv, int32_t A, int32_t Q, const float* ga, const float* gv,
                                       const float* wo, const float* wq, const float* both, float* g, float* db, float* partial,
                                       int32_t blocks, float* dwj, float* partial_w, void* stream) {
  if (!dwj) return mirl::fail(MIRL_ERR_ARG, "dueling_tail_bwd_w: null dwj");
  return tail_bwd_launch(M, H1, Hv, A, Q, ga, gv, wo, wq, both, g, db, partial, blocks, dwj, partial_w, stream);
}
