// gemm3.hip — the network's wide f32 GEMMs on the bf16 matrix pipe without giving up f32 results.
//
// What it replaces: the hipBLASLt f32 GEMMs behind the reference's nn.Linear layers of the recurrent IQN
// model (policies/torch/dqn.py:50-112 dueling head, iqn.py:82-102 quantile layer, modules/lstm.py:60-81
// input projection) and their data / weight gradients — 86 of the 136 ms learner step at BASELINE
// configs[3], already at 0.8-0.9 of the f32 MFMA peak (157 TFLOP/s = 1/16 of the bf16 rate).
//
// How: every f32 operand element is split EXACTLY into three bf16 parts while its tile is staged into LDS,
//     x = hi + mid + lo,   hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)
// (three 8-bit significands cover f32's 24 bits; both subtractions are exact), and six of the nine part
// products are accumulated in f32 by v_mfma_f32_32x32x16_bf16, smallest first:
//     a.b ~= lo.hi + hi.lo + mid.mid + mid.hi + hi.mid + hi.hi
// The three dropped products (mid.lo, lo.mid, lo.lo) are below 2^-23 |a||b| per term — the size of the
// rounding a plain f32 fma chain commits per product; tests/test_gemm3_gpu.py holds the result to the
// float64 product at least as tightly as the library's f32 GEMM on the same inputs.  Six bf16 MFMAs per
// 32x32x16 block of f32 work price the f32 product at 2.5 PFLOP/s / 6 = 417 TFLOP/s peak, 2.65x the f32 pipe.
// Non-finite inputs are not preserved (inf - inf in the split gives NaN); the callers' activations are finite.
//
// Tiling: one workgroup = 256 x 256 outputs, 8 waves (2 along M x 4 along N, 128 x 64 each = 4 x 2 MFMA tiles,
// 128 accumulator registers), K in steps of 16 floats.  LDS holds two stages of six bf16 planes
// [operand A|B][part][256 rows][16 k], 48-byte row pitch (ds_read_b128 of a 32-row fragment is conflict free),
// 147 456 bytes: one workgroup per CU, two waves per SIMD.  Per K-step a wave issues 48 MFMAs (1 536 cycles on its
// SIMD's matrix pipe, 3 072 for the pair) against 18 fragment reads, 2 x 16-byte global loads, ~100 VALU split ops
// and 12 LDS writes; the two waves of a SIMD run the stage/compute halves of the step in opposite order so one
// splits while the other multiplies.  Global loads for tile k+2 are in flight across the (raw) barrier.
//
// Operand forms (row-major, strides in floats): element (row, k) of an operand is either k-contiguous
// (base[row*ld + k]: activations / weights as stored) or k-strided (base[k*ld + row]: the transposed use in
// a weight gradient or an un-transposed weight in a data gradient).  layout 0 "NT": C = A[M][K] . B[N][K]^T (+bias,
// ReLU); 1 "NN": C = A[M][K] . B[K][N]; 2 "TN": C = A[K][M]^T . B[K][N], K split over workgroups into
// partial tiles that a second deterministic kernel sums.
//
// Epilogue: an accumulator lane owns one column of a 32 x 32 tile; each wave transposes its block through its own
// 17 KB of the idle staging LDS and stores 16 bytes per lane (256 contiguous bytes per row) — with bias, ReLU and,
// for the quantile layer (mirl_gemm3_nt_mul), the IQN feature product applied on those vectors.  Outputs whose
// rows are not 16-byte aligned take a scalar-store instantiation.
// Measured (MI355X, profiles/r03_gemm3_*): 200-226 TFLOP/s of f32 product against 142-145 for the library's f32
// kernels; the matrix pipe is 67 % busy at a power-limited 1.67 GHz.  Built, measured without gain and removed again
// (evidence under profiles/r04_gemm3_*): weights pre-split once per optimizer step, persistent workgroups for long K,
// all waves staging first / computing first.
#include "common.hpp"
#include "split3.hpp"
#include <stdlib.h>

namespace mirl {

struct G3Args {
  const float* A; const float* B; float* C; const float* bias;
  int64_t M, N, K;
  int64_t lda, ldb, ldc;
  int relu;
  int mt, nt;            // output tiles along M / N
  int splits;            // K chunks (1 = none); with splits > 1, C is [splits][M][N] partials (ldc = N)
  int steps_per_split;   // K-steps of 16 per chunk
  // epilogue extension (NT): C = f(A B^T + bias) * mul[row >> mul_shift][col]; `pre` (optional) receives f(...) itself
  const float* mul; int64_t ldmul; int mul_shift;
  float* pre; int64_t ldpre;
  int vec_ok;            // 16-byte stores possible: N % 4 == 0, every output / bias / multiplier row 16-byte aligned
  // epilogue extension EP == 2 (NT): the O <= 8 output units that FOLLOW this layer, out[row][o] = sum_n f(...)[row][n] * w2[o][n],
  // as partial sums per 64-column block: part[(col / 64)][row][8]; C may then be null (the activation itself is not stored)
  const float* w2; float* part;
  // epilogue extension EP == 3 (NN, a data gradient g W that feeds the IQN feature product's backward, iqn.py:84,102): with
  // d = A B (never stored), e = `pre` (the ReLU'd embedding, READ here), x = mul[row >> 5]:
  //   C[row][col] = e > 0 ? d * x : 0;   gsum[row >> 5][col] = sum over the group's 32 rows of d * e;
  //   part[2 it + wm][col] = this wave's 128-row column sum of C   (the embedding layer's bias gradient, summed by the caller)
  float* gsum; int64_t ldgsum;
  int prio;              // experiment switch (MIRL_GEMM3_PRIO): 1 = s_setprio 1 around every MFMA block, 2 = waves 4-7 raised once
};


// One operand's loader state: two rows per thread (r and r + 128 of the 256-row tile), four consecutive k each.
template <bool KC>
struct G3Loader {
  const float* p[2];
  int64_t step;          // pointer advance per K-step
  int64_t ld;
  int lds_off;           // byte offset of this thread's first row inside a plane
  __device__ __forceinline__ void init(const float* base, int64_t ld_, int64_t row0, int64_t rows, int64_t k0, int t) {
    ld = ld_;
    if (KC) {
      const int r = t >> 2, kq = t & 3;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int64_t row = row0 + r + 128 * h; if (row > rows - 1) row = rows - 1;
        p[h] = base + row * ld + k0 + kq * 4;
      }
      step = 16;
      lds_off = r * G3_PITCH + kq * 8;
    } else {
      const int kq = t >> 7, r = t & 127;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int64_t row = row0 + r + 128 * h; if (row > rows - 1) row = rows - 1;
        p[h] = base + (k0 + kq * 4) * ld + row;
      }
      step = 16 * ld;
      lds_off = r * G3_PITCH + kq * 8;
    }
  }
  __device__ __forceinline__ void load(float (&v)[2][4]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (KC) {
        const float4 q = *reinterpret_cast<const float4*>(p[h]);
        v[h][0] = q.x; v[h][1] = q.y; v[h][2] = q.z; v[h][3] = q.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[h][e] = p[h][e * ld];
      }
      p[h] += step;
    }
  }
  // split + write both rows into the three planes of this operand at `planes`
  __device__ __forceinline__ void store(char* planes, const float (&v)[2][4]) const {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      uint2 a, b, c;
      g3_split4(v[h], a, b, c);
      char* d = planes + lds_off + h * 128 * G3_PITCH;
      *reinterpret_cast<uint2*>(d) = a;
      *reinterpret_cast<uint2*>(d + G3_PLANE) = b;
      *reinterpret_cast<uint2*>(d + 2 * G3_PLANE) = c;
    }
  }
};

// 48 MFMAs of one K-step on this wave's 128 x 64 block: fragments of tile (i, j) are rows wm*128 + i*32 + (lane & 31)
// of A and rows wn*64 + j*32 + (lane & 31) of B, k = 8 (lane >> 5) .. +7.  (Issuing all 18 fragment reads before the
// first MFMA instead of per A half measured 3 % slower: 7.56 vs 7.17 ms at 1 310 720 x 1024 x 512.  Round 6: the
// second half's six reads requested under the first half's MFMAs — both halves' fragments live, 256 VGPRs — measured
// no faster on the plain forms and 3-15 % slower on the epilogue forms, profiles/r06_gemm3_pipelined_fragments_experiment.jsonl;
// the two waves of a SIMD already cover each other's fragment waits.  s_setprio 1 around this block: +1-2 %, kept.)
template <int TI>
__device__ __forceinline__ void g3_compute(const char* stage, g3_f32x16 (&acc)[TI][2], int a_off, int b_off) {
  const char* pa = stage + a_off;
  const char* pb = stage + 3 * G3_PLANE + b_off;
  // smallest products first: (lo,hi) (hi,lo) (mid,mid) (mid,hi) (hi,mid) (hi,hi)
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
  constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
  g3_bf16x8 b[3][2];
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int j = 0; j < 2; ++j) b[p][j] = *reinterpret_cast<const g3_bf16x8*>(pb + p * G3_PLANE + j * 32 * G3_PITCH);
  constexpr int IW = TI >= 2 ? 2 : 1;              // A tiles per fragment batch
#pragma unroll
  for (int ih = 0; ih < TI / IW; ++ih) {
    g3_bf16x8 a[3][IW];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int i = 0; i < IW; ++i) a[p][i] = *reinterpret_cast<const g3_bf16x8*>(pa + p * G3_PLANE + (ih * IW + i) * 32 * G3_PITCH);
#pragma unroll
    for (int c = 0; c < 6; ++c)
#pragma unroll
      for (int i = 0; i < IW; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[ih * IW + i][j] = g3_mfma(a[PA[c]][i], b[PB[c]][j], acc[ih * IW + i][j]);
  }
}


// NARROW: outputs at most 64 columns wide (the quantile layer's weight gradient, 512 x 64 over K = 1.3 M rows): the
// eight waves stack along M — 32 rows x 64 columns each, 12 MFMAs per K-step instead of 48 of which 36 multiplied
// columns that do not exist — so the product is bound by reading its K x M operand, not by the matrix pipe.
template <bool AKC, bool BKC, int EP, bool VEC, bool NARROW = false>
__global__ void __launch_bounds__(512)
k_gemm3(G3Args g) {
  constexpr int TI = NARROW ? 1 : 4;
  extern __shared__ __attribute__((aligned(16))) char g3_lds[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // workgroup -> (row tile, column tile, K chunk).  Workgroup b runs on XCD b % 8: the column tiles of one row
  // tile (and, with split K, the output tiles of one K chunk) are consecutive on ONE XCD and share its L2.
  // Without split K a workgroup walks the XCD's tile list e = l, l + L, ... (l = b / 8, L = workgroups per XCD;
  // element e = column tile e % nt of row tile 8 (e / nt) + xcd): L = list length is one tile per workgroup,
  // L = CUs per XCD is a persistent workgroup whose next tile's first loads fly during this tile's stores.
  const int b = blockIdx.x, xcd = b & 7, l = b >> 3, L = gridDim.x >> 3;
  const int64_t nk_all = g.K / 16;
  const int E = g.splits > 1 ? l + 1 : ((g.mt + 7) / 8) * g.nt;        // split K: exactly one pass
  int split = 0;
  int64_t ks0 = 0, nk = nk_all;

  const int wu = __builtin_amdgcn_readfirstlane(wave);
  const int wm = NARROW ? 0 : (wu & 1), wn = NARROW ? 0 : (wu >> 1);
  const int row_w = NARROW ? wu * 32 : wm * 128;                      // this wave's first row inside the tile
  const int a_off = (row_w + (lane & 31)) * G3_PITCH + (lane >> 5) * 16;
  const int b_off = (wn * 64 + (lane & 31)) * G3_PITCH + (lane >> 5) * 16;
  const bool stage_first = (wu >> 2) & 1;      // waves w and w + 4 share a SIMD and take opposite orders

  if (g.prio == 2 && wu >= 4) __builtin_amdgcn_s_setprio(1);
  // first tile of this workgroup
  int e = l, it = 0, jt = 0;
  if (g.splits > 1) {
    const int tiles = g.mt * g.nt, tile = l % tiles;
    split = xcd + 8 * (l / tiles);
    it = tile / g.nt; jt = tile % g.nt;
    ks0 = (int64_t)split * g.steps_per_split;
    nk = nk_all - ks0; if (nk > g.steps_per_split) nk = g.steps_per_split;
    if (nk < 0) nk = 0;
  } else {
    for (; e < E; e += L) { jt = e % g.nt; it = (e / g.nt) * 8 + xcd; if (it < g.mt) break; }
    if (e >= E) return;
  }

  G3Loader<AKC> la; G3Loader<BKC> lb;
  float va[2][4], vb[2][4];
  la.init(g.A, g.lda, (int64_t)it * 256, g.M, ks0 * 16, t);
  lb.init(g.B, g.ldb, (int64_t)jt * 256, g.N, ks0 * 16, t);
  if (nk > 0) { la.load(va); lb.load(vb); }

  g3_f32x16 acc[TI][2];
  while (true) {
    const int64_t m0 = (int64_t)it * 256, n0 = (int64_t)jt * 256;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    if (nk > 0) {
      la.store(g3_lds, va); lb.store(g3_lds + 3 * G3_PLANE, vb);
      if (nk > 1) { la.load(va); lb.load(vb); }
    }
    g3_barrier();
    for (int64_t k = 0; k < nk; ++k) {
      const char* cur = g3_lds + (k & 1) * G3_STAGE;
      char* nxt = g3_lds + ((k + 1) & 1) * G3_STAGE;
      // ONE copy of the MFMA block (two copies get two accumulator register assignments and the compiler
      // reconciles them with 128 moves per step); the small staging block sits before or after it
      if (stage_first) {
        if (k + 1 < nk) { la.store(nxt, va); lb.store(nxt + 3 * G3_PLANE, vb); }
        if (k + 2 < nk) { la.load(va); lb.load(vb); }
      }
      if (g.prio == 1) __builtin_amdgcn_s_setprio(1);
      g3_compute<TI>(cur, acc, a_off, b_off);
      if (g.prio == 1) __builtin_amdgcn_s_setprio(0);
      if (!stage_first) {
        if (k + 1 < nk) { la.store(nxt, va); lb.store(nxt + 3 * G3_PLANE, vb); }
        if (k + 2 < nk) { la.load(va); lb.load(vb); }
      }
      g3_barrier();
    }

    // next tile of this workgroup: its first K-step is requested before this tile's stores go out
    bool more = false;
    if (g.splits <= 1) {
      for (e += L; e < E; e += L) { jt = e % g.nt; it = (e / g.nt) * 8 + xcd; if (it < g.mt) break; }
      more = e < E;
      if (more) {
        la.init(g.A, g.lda, (int64_t)it * 256, g.M, 0, t);
        lb.init(g.B, g.ldb, (int64_t)jt * 256, g.N, 0, t);
        la.load(va); lb.load(vb);
      }
    }

    // epilogue: C/D register r of a 32x32 tile is row (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column lane & 31 — one column
    // per lane, so storing straight from the accumulators is 128 four-byte store instructions per wave, and the
    // store ISSUE (not bandwidth) was 13 us of a 90 us K = 512 tile.  Each wave transposes its block through its own
    // 17 KB of the (now idle) staging LDS in two 64-row halves and stores 16 bytes per lane, 256 contiguous bytes per row.
    float* C = g.C + (g.splits > 1 ? (int64_t)split * g.M * g.N : 0);
    if (VEC) {
      float* tl = reinterpret_cast<float*>(g3_lds) + wu * (64 * G3_EPITCH);
      constexpr int HALVES = NARROW ? 1 : 2, II = NARROW ? 1 : 2, QN = NARROW ? 8 : 16;     // NARROW: one 32-row block per wave
      float* w2l = reinterpret_cast<float*>(g3_lds) + 8 * (64 * G3_EPITCH);                 // EP 2: w2[8][256] of this tile's columns, 8 KB
      float bj[2] = {0.f, 0.f};
      g3_f32x4 sb = {0.f, 0.f, 0.f, 0.f};                                                   // EP 3: column sums of C over this wave's rows
      if (EP == 2) {
        // the following layer's weights for this tile's 256 columns -> the 8 KB of LDS behind the transposition areas
        // (one 16-byte vector per thread: o = t >> 6, columns 4 (t & 63) ..); visible after the barrier below
        const int64_t wc = n0 + (t & 63) * 4;
        float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
        if (wc < g.N) w = *reinterpret_cast<const float4*>(g.w2 + (int64_t)(t >> 6) * g.N + wc);
        *reinterpret_cast<float4*>(w2l + (t >> 6) * 256 + (t & 63) * 4) = w;
        // bias + ReLU go in BEFORE the transposition here (a lane owns columns wn 64 + 32 j + (lane & 31) of its tiles)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int64_t cj = n0 + wn * 64 + j * 32 + (lane & 31);
          bj[j] = (g.bias && cj < g.N) ? g.bias[cj] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              float v = acc[i][j][r] + bj[j];
              acc[i][j][r] = (g.relu && !(v > 0.f)) ? 0.f : v;
            }
        g3_barrier();
      }
#pragma unroll
      for (int h = 0; h < HALVES; ++h) {
#pragma unroll
        for (int ii = 0; ii < II; ++ii)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              tl[(ii * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * G3_EPITCH + j * 32 + (lane & 31)] = acc[II * h + ii][j][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int c4 = lane & 15;
        const int64_t col = n0 + wn * 64 + c4 * 4;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (EP != 2 && g.bias && col < g.N) {      // N % 4 == 0 on this path
          bv = *reinterpret_cast<const float4*>(g.bias + col);
        }
        if (EP == 2) {
          // out[row][o] += sum over this wave's 64 columns: lane = (rows rg + 16 i; columns 16 cq ..), two rows at a time
          // (2 x 8 running sums: the other half's 64 accumulator registers are still live), data rows and weight vectors
          // as 16-byte LDS reads (conflict-free: row pitch 68 floats puts the 16 lanes of a quarter on 64 different
          // banks; a weight vector is one address per quarter), then a two-step butterfly over the four column quarters.
          // 512 FMAs per lane and half: ~4 % of the tile's MFMA time.
          const int rg = lane & 15, cq = lane >> 4;
          const int64_t cb = (n0 >> 6) + wn;
#pragma unroll 1
          for (int ip = 0; ip < 2; ++ip) {
            float ps[2][8];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int o = 0; o < 8; ++o) ps[i][o] = 0.f;
#pragma unroll 1
            for (int c = 0; c < 4; ++c) {
              asm volatile("" ::: "memory");          // keep this column group's LDS reads inside its iteration (no hoisting:
              float4 dv[2];                           //  128 invariant weight vectors would not fit the register file)
#pragma unroll
              for (int i = 0; i < 2; ++i) dv[i] = *reinterpret_cast<const float4*>(tl + (rg + 16 * (2 * ip + i)) * G3_EPITCH + cq * 16 + c * 4);
#pragma unroll
              for (int o = 0; o < 8; ++o) {
                const float4 w = *reinterpret_cast<const float4*>(w2l + o * 256 + wn * 64 + cq * 16 + c * 4);
#pragma unroll
                for (int i = 0; i < 2; ++i)
                  ps[i][o] = fmaf(dv[i].w, w.w, fmaf(dv[i].z, w.z, fmaf(dv[i].y, w.y, fmaf(dv[i].x, w.x, ps[i][o]))));
              }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int o = 0; o < 8; ++o) {
                float v = ps[i][o];
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                ps[i][o] = v;
              }
            if (cq == 0) {
#pragma unroll
              for (int i = 0; i < 2; ++i) {
                const int64_t row = m0 + row_w + h * 64 + rg + 16 * (2 * ip + i);
                if (row < g.M) {
                  float* d = g.part + (cb * g.M + row) * 8;
                  *reinterpret_cast<float4*>(d) = make_float4(ps[i][0], ps[i][1], ps[i][2], ps[i][3]);
                  *reinterpret_cast<float4*>(d + 4) = make_float4(ps[i][4], ps[i][5], ps[i][6], ps[i][7]);
                }
              }
            }
          }
        }
        // IQN feature product (iqn.py:84,102): the row group r >> mul_shift (one state's quantile rows) shares one row of
        // `mul`.  A lane's 16 rows of this half are q * 4 + (lane >> 4): with groups of 2^s >= 32 rows (32 quantiles per
        // state, the shipped IQN configs) that is two multiplier vectors per lane and half, fetched before the first store
        // goes out instead of one dependent global load per output vector (measured: 1.19 -> 1.16 ms only — the K = 64
        // product is bound by its serialised load -> split -> MFMA -> store phases per tile).
        float4 mg0 = make_float4(0.f, 0.f, 0.f, 0.f), mg1 = mg0;
        const bool hoisted = (EP == 1 && g.mul_shift >= 5) || EP == 3;   // groups of >= 32 rows: at most two per 64-row half
        if ((EP == 1 || EP == 3) && hoisted) {
          // unconditional loads from clamped addresses (rows / columns beyond the edge are never stored)
          const int64_t rb = m0 + row_w + h * 64 + (lane >> 4);
          const int64_t r0 = rb < g.M ? rb : g.M - 1, r1 = rb + 32 < g.M ? rb + 32 : g.M - 1;
          const int64_t cc = col < g.N ? col : 0;
          mg0 = *reinterpret_cast<const float4*>(g.mul + (r0 >> g.mul_shift) * g.ldmul + cc);
          mg1 = *reinterpret_cast<const float4*>(g.mul + (r1 >> g.mul_shift) * g.ldmul + cc);
        }
        if (EP == 3) {
          // the feature product's backward on the data gradient while it is still in LDS: the two 32-row groups of this
          // half (one state's quantile rows each); the group's 8 embedding vectors of a lane are requested together
          const float4 mgs[2] = {mg0, mg1};
#pragma unroll
          for (int gq = 0; gq < 2; ++gq) {
            g3_f32x4 ev[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int64_t row = m0 + row_w + h * 64 + (gq * 8 + u) * 4 + (lane >> 4);
              const int64_t rr = row < g.M ? row : g.M - 1, cc = col < g.N ? col : 0;
              ev[u] = __builtin_nontemporal_load(reinterpret_cast<const g3_f32x4*>(g.pre + rr * g.ldpre + cc));
            }
            g3_f32x4 sg = {0.f, 0.f, 0.f, 0.f};
            const g3_f32x4 xm = {mgs[gq].x, mgs[gq].y, mgs[gq].z, mgs[gq].w};
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int rl = (gq * 8 + u) * 4 + (lane >> 4);
              const int64_t row = m0 + row_w + h * 64 + rl;
              const float4 t4 = *reinterpret_cast<const float4*>(tl + rl * G3_EPITCH + c4 * 4);
              const g3_f32x4 v = {t4.x, t4.y, t4.z, t4.w};
              if (row < g.M && col < g.N) {
                g3_f32x4 d = v * xm;
                d.x = ev[u].x > 0.f ? d.x : 0.f; d.y = ev[u].y > 0.f ? d.y : 0.f;
                d.z = ev[u].z > 0.f ? d.z : 0.f; d.w = ev[u].w > 0.f ? d.w : 0.f;
                __builtin_nontemporal_store(d, reinterpret_cast<g3_f32x4*>(C + row * g.ldc + col));
                sg += v * ev[u];
                sb += d;
              }
            }
            // the group's 32 rows sit in the four lane quarters (8 rows each): fixed-order butterfly
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float v = sg[e];
              v += __shfl_xor(v, 16);
              v += __shfl_xor(v, 32);
              sg[e] = v;
            }
            const int64_t row0 = m0 + row_w + h * 64 + gq * 32;
            if ((lane >> 4) == 0 && row0 < g.M && col < g.N)
              *reinterpret_cast<g3_f32x4*>(g.gsum + (row0 >> 5) * g.ldgsum + col) = sg;
          }
        }
#pragma unroll
        for (int q = 0; q < QN; ++q) {
          const int rl = q * 4 + (lane >> 4);
          const int64_t row = m0 + row_w + h * 64 + rl;
          if (EP == 3) break;                       // done above
          if (EP == 2 && !g.C) break;               // no-grad pass: only the following layer's outputs leave the kernel
          float4 v = *reinterpret_cast<const float4*>(tl + rl * G3_EPITCH + c4 * 4);
          if (row < g.M && col < g.N) {
            if (EP != 2) {
              v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
              if (g.relu) { v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f; }
            }
            if (EP == 1) {
              if (g.pre) __builtin_nontemporal_store(g3_f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<g3_f32x4*>(g.pre + row * g.ldpre + col));
              float4 m;
              if (hoisted) m = q < 8 ? mg0 : mg1;           // q is a compile-time constant of the unrolled loop
              else m = *reinterpret_cast<const float4*>(g.mul + (row >> g.mul_shift) * g.ldmul + col);
              v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
            }
            __builtin_nontemporal_store(g3_f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<g3_f32x4*>(C + row * g.ldc + col));
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this half read before the next one overwrites it
      }
      if (EP == 3) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = sb[e];
          v += __shfl_xor(v, 16);
          v += __shfl_xor(v, 32);
          sb[e] = v;
        }
        const int64_t col = n0 + wn * 64 + (lane & 15) * 4;
        if ((lane >> 4) == 0 && col < g.N)
          *reinterpret_cast<g3_f32x4*>(g.part + ((m0 >> 8) * 2 + wm) * g.N + col) = sb;      // (`it` already names the next tile)
      }
      if (more) g3_barrier();          // persistent walk: every wave's transpose done before the next tile is staged
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int64_t col = n0 + wn * 64 + j * 32 + (lane & 31);
        const bool col_ok = col < g.N;
        const float bv = (g.bias && col_ok) ? g.bias[col] : 0.0f;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
          const int64_t rb = m0 + row_w + i * 32 + 4 * (lane >> 5);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int64_t row = rb + (r & 3) + 8 * (r >> 2);
            if (!(col_ok && row < g.M)) continue;
            float v = acc[i][j][r] + bv;
            if (g.relu) v = v > 0.0f ? v : 0.0f;
            if (EP == 1) {
              if (g.pre) __builtin_nontemporal_store(v, g.pre + row * g.ldpre + col);
              v *= g.mul[(row >> g.mul_shift) * g.ldmul + col];
            }
            __builtin_nontemporal_store(v, C + row * g.ldc + col);
          }
        }
      }
    }
    if (!more) break;
  }
}


// ---- round 6: a 256 x 128 tile for products with too few 256 x 256 tiles to fill the chip -------------------------
// The acting batch's hidden layer at 256 envs (8 192 quantile rows x 1 024 x 512) is 128 tiles of 256 x 256 on 256
// compute units: half the chip idles and the library's f32 kernel wins (68.9 against 67 us, profiles/r05_acting_256_
// envs_learner_gemm_kernels_experiment.jsonl).  Same method, same staging and K loop, but a workgroup's eight waves
// (2 along M x 4 along N) own 128 rows x 32 columns each: 24 MFMAs per K-step against 12 + 3 fragment reads, 64
// accumulator registers, LDS two stages of [A 256 rows | B 128 rows] x 3 parts = 110 592 bytes.  NT form with bias /
// ReLU only (what the call site needs); 16-byte row stores through a per-wave 32 x 36 float transposition area.
constexpr int G3M_BROWS = 128;
constexpr int G3M_BPLANE = G3M_BROWS * G3_PITCH;
constexpr int G3M_STAGE = 3 * G3_PLANE + 3 * G3M_BPLANE;          // 55 296
constexpr int G3M_LDS = 2 * G3M_STAGE;
constexpr int G3M_EPITCH = 36;

__global__ void __launch_bounds__(512)
k_gemm3_mid(G3Args g) {
  extern __shared__ __attribute__((aligned(16))) char g3_lds[];
  const int t = threadIdx.x, lane = t & 63;
  const int wu = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wu & 1, wn = wu >> 1;
  const int b = blockIdx.x, xcd = b & 7, l = b >> 3;
  // column tiles of one row tile are consecutive on ONE XCD (they share the A rows in its L2), as in k_gemm3
  const int jt = l % g.nt, it = (l / g.nt) * 8 + xcd;
  if (it >= g.mt) return;
  const int64_t m0 = (int64_t)it * 256, n0 = (int64_t)jt * G3M_BROWS;
  const int64_t nk = g.K / 16;

  G3Loader<true> la;
  la.init(g.A, g.lda, m0, g.M, 0, t);
  // B: one row per thread (row t >> 2 of the 128, k quad t & 3)
  const int rb = t >> 2, kq = t & 3;
  int64_t brow = n0 + rb; if (brow > g.N - 1) brow = g.N - 1;
  const float* pb = g.B + brow * g.ldb + kq * 4;
  const int b_lds = rb * G3_PITCH + kq * 8;
  float va[2][4], vb[4];
  auto load_b = [&]() { const float4 q = *reinterpret_cast<const float4*>(pb); vb[0] = q.x; vb[1] = q.y; vb[2] = q.z; vb[3] = q.w; pb += 16; };
  auto store_b = [&](char* planes) {
    uint2 h, m, lo;
    g3_split4(vb, h, m, lo);
    *reinterpret_cast<uint2*>(planes + b_lds) = h;
    *reinterpret_cast<uint2*>(planes + G3M_BPLANE + b_lds) = m;
    *reinterpret_cast<uint2*>(planes + 2 * G3M_BPLANE + b_lds) = lo;
  };
  const int a_off = (wm * 128 + (lane & 31)) * G3_PITCH + (lane >> 5) * 16;
  const int b_off = (wn * 32 + (lane & 31)) * G3_PITCH + (lane >> 5) * 16;
  const bool stage_first = (wu >> 2) & 1;      // waves w and w + 4 share a SIMD and take opposite orders

  g3_f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0};          // smallest products first
  constexpr int PB[6] = {0, 2, 1, 0, 1, 0};

  if (nk > 0) { la.load(va); load_b(); la.store(g3_lds, va); store_b(g3_lds + 3 * G3_PLANE); }
  if (nk > 1) { la.load(va); load_b(); }
  g3_barrier();
  for (int64_t k = 0; k < nk; ++k) {
    const char* cur = g3_lds + (k & 1) * G3M_STAGE;
    char* nxt = g3_lds + ((k + 1) & 1) * G3M_STAGE;
    if (stage_first) {
      if (k + 1 < nk) { la.store(nxt, va); store_b(nxt + 3 * G3_PLANE); }
      if (k + 2 < nk) { la.load(va); load_b(); }
    }
    {
      const char* pa = cur + a_off;
      const char* pbf = cur + 3 * G3_PLANE + b_off;
      g3_bf16x8 bf[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) bf[p] = *reinterpret_cast<const g3_bf16x8*>(pbf + p * G3M_BPLANE);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ih = 0; ih < 2; ++ih) {
        g3_bf16x8 a[3][2];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int i = 0; i < 2; ++i) a[p][i] = *reinterpret_cast<const g3_bf16x8*>(pa + p * G3_PLANE + (ih * 2 + i) * 32 * G3_PITCH);
#pragma unroll
        for (int c = 0; c < 6; ++c)
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[ih * 2 + i] = g3_mfma(a[PA[c]][i], bf[PB[c]], acc[ih * 2 + i]);
      }
      __builtin_amdgcn_s_setprio(0);
    }
    if (!stage_first) {
      if (k + 1 < nk) { la.store(nxt, va); store_b(nxt + 3 * G3_PLANE); }
      if (k + 2 < nk) { la.load(va); load_b(); }
    }
    g3_barrier();
  }

  // epilogue: each 32 x 32 accumulator tile through this wave's 4 608-byte area, then + bias, ReLU, 16-byte row stores
  float* tl = reinterpret_cast<float*>(g3_lds) + wu * (32 * G3M_EPITCH);
  const int c4 = lane & 7;
  const int64_t col = n0 + wn * 32 + c4 * 4;
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (g.bias && col < g.N) bv = *reinterpret_cast<const float4*>(g.bias + col);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
      tl[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * G3M_EPITCH + (lane & 31)] = acc[i][r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int rl = q * 8 + (lane >> 3);
      const int64_t row = m0 + wm * 128 + i * 32 + rl;
      float4 v = *reinterpret_cast<const float4*>(tl + rl * G3M_EPITCH + c4 * 4);
      if (row < g.M && col < g.N) {
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        if (g.relu) { v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f; }
        *reinterpret_cast<float4*>(g.C + row * g.ldc + col) = v;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this tile read before the next one overwrites the area
  }
}

// out[i] = sum over splits of partial[s][i] (fixed order: deterministic), i over M*N, rows re-pitched to ldc
__global__ void __launch_bounds__(256)
k_gemm3_reduce(const float* __restrict__ partial, int splits, int64_t M, int64_t N, float* __restrict__ out, int64_t ldc) {
  const int64_t n4 = N >> 2;
  const int64_t total = M * n4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / n4, c4 = i - row * n4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < splits; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(partial + ((int64_t)k * M + row) * N + c4 * 4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(out + row * ldc + c4 * 4) = s;
  }
}

// out[row][o] = bias2[o] + sum over the 64-column blocks of part[cb][row][o], o < O <= 8 (fixed order: deterministic)
__global__ void __launch_bounds__(256)
k_g3_head_reduce(const float* __restrict__ part, int ncb, int64_t M, int O, const float* __restrict__ bias2,
                 float* __restrict__ out, int64_t ldo) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < M * 2; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i >> 1;
    const int half = (int)(i & 1);
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int cb = 0; cb < ncb; ++cb) {
      const float4 v = *reinterpret_cast<const float4*>(part + ((int64_t)cb * M + row) * 8 + half * 4);
      sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    }
    const float r[4] = {sum.x, sum.y, sum.z, sum.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int o = half * 4 + e;
      if (o < O) out[row * ldo + o] = r[e] + (bias2 ? bias2[o] : 0.f);
    }
  }
}

static int g_g3_mid_mode = -1;       // mirl_gemm3_mid_set: -1 = MIRL_GEMM3_MID (default on), 0 / 1 = forced for in-process A/B runs

static int g3_splits(int64_t M, int64_t N, int64_t K) {
  const int64_t tiles = ((M + 255) / 256) * ((N + 255) / 256);
  int64_t s = 8 * ((256 + 8 * tiles - 1) / (8 * tiles));      // tiles * splits >= 256 workgroups, splits % 8 == 0
  const int64_t nk = K / 16;
  while (s > 8 && nk / s < 64) s -= 8;                        // keep >= 64 K-steps per workgroup
  return (int)s;
}

}  // namespace mirl

extern "C" int mirl_gemm3_supported(int32_t layout, int64_t M, int64_t N, int64_t K) {
  if (layout < 0 || layout > 2) return 0;
  if (M < 1 || N < 1 || K < 16 || (K % 16)) return 0;
  if (M >= (1LL << 31) || N >= (1LL << 31)) return 0;
  if (layout == 2) {
    if (N % 4) return 0;                                      // the reduction works on float4 columns
    if (K / 16 < 8) return 0;
  }
  return 1;
}

extern "C" int mirl_gemm3_workspace_bytes(int32_t layout, int64_t M, int64_t N, int64_t K, int64_t* bytes) {
  using namespace mirl;
  if (!bytes) return fail(MIRL_ERR_ARG, "gemm3_workspace_bytes: null out");
  if (!mirl_gemm3_supported(layout, M, N, K)) return fail(MIRL_ERR_ARG, "gemm3: unsupported layout / shape");
  *bytes = layout == 2 ? (int64_t)g3_splits(M, N, K) * M * N * (int64_t)sizeof(float) : 0;
  return MIRL_OK;
}

static int g3_launch(int32_t layout, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B,
                     int64_t ldb, float* C, int64_t ldc, const float* bias, int32_t relu, void* workspace,
                     int64_t workspace_bytes, const float* mul, int64_t ldmul, int32_t mul_shift, float* pre, int64_t ldpre,
                     void* stream, const float* w2 = nullptr, float* part = nullptr,
                     float* gsum = nullptr, int64_t ldgsum = 0) {
  using namespace mirl;
  if (!mirl_gemm3_supported(layout, M, N, K)) return fail(MIRL_ERR_ARG, "gemm3: unsupported layout / shape");
  if (!A || !B || (!C && !w2)) return fail(MIRL_ERR_ARG, "gemm3: null operand");
  const bool akc = layout != 2, bkc = layout == 0;
  if (akc && ((lda % 4) || ((uintptr_t)A % 16) || lda < K)) return fail(MIRL_ERR_ARG, "gemm3: A must be 16-byte aligned with lda % 4 == 0");
  if (bkc && ((ldb % 4) || ((uintptr_t)B % 16) || ldb < K)) return fail(MIRL_ERR_ARG, "gemm3: B must be 16-byte aligned with ldb % 4 == 0");
  if (!akc && lda < M) return fail(MIRL_ERR_ARG, "gemm3: lda < M");
  if (!bkc && ldb < N) return fail(MIRL_ERR_ARG, "gemm3: ldb < N");
  if (ldc < N) return fail(MIRL_ERR_ARG, "gemm3: ldc < N");
  if (layout == 2 && (bias || relu)) return fail(MIRL_ERR_ARG, "gemm3: no epilogue with split K");
  hipStream_t st = (hipStream_t)stream;
  G3Args g;
  g.A = A; g.B = B; g.C = C; g.bias = bias; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.relu = relu ? 1 : 0;
  g.mul = mul; g.ldmul = ldmul; g.mul_shift = mul_shift; g.pre = pre; g.ldpre = ldpre;
  g.w2 = w2; g.part = part; g.gsum = gsum; g.ldgsum = ldgsum;
  static const int prio_env = getenv("MIRL_GEMM3_PRIO") ? atoi(getenv("MIRL_GEMM3_PRIO")) : 1;
  g.prio = prio_env;
  static const int vec_env = getenv("MIRL_GEMM3_VEC") ? atoi(getenv("MIRL_GEMM3_VEC")) : 1;
  g.vec_ok = vec_env && (N % 4 == 0) && (ldc % 4 == 0) && !((uintptr_t)C % 16) && (!w2 || !((uintptr_t)w2 % 16)) && (!bias || !((uintptr_t)bias % 16)) &&
             (!mul || ((ldmul % 4 == 0) && !((uintptr_t)mul % 16))) && (!pre || ((ldpre % 4 == 0) && !((uintptr_t)pre % 16)));
  g.mt = (int)((M + 255) / 256); g.nt = (int)((N + 255) / 256);
  g.splits = 1; g.steps_per_split = (int)(K / 16);
  static const int narrow_env = getenv("MIRL_GEMM3_NARROW") ? atoi(getenv("MIRL_GEMM3_NARROW")) : 1;
  unsigned grid = (unsigned)(8 * ((g.mt + 7) / 8) * g.nt);                 // one tile per workgroup
  if (K <= 128) {          // short K: the tile is mostly epilogue — let the next tile's loads fly during the stores
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const unsigned resident = (unsigned)(cus / 8 * 8);                        // one workgroup per CU (147 KB of LDS each)
    if (grid > resident) grid = resident;
  }
  if (layout == 2) {
    g.splits = g3_splits(M, N, K);
    const int64_t need = (int64_t)g.splits * M * N * (int64_t)sizeof(float);
    if (!workspace || workspace_bytes < need || ((uintptr_t)workspace % 16) || (ldc % 4) || ((uintptr_t)C % 16))
      return fail(MIRL_ERR_ARG, "gemm3: split-K workspace too small / misaligned");
    g.steps_per_split = (int)((K / 16 + g.splits - 1) / g.splits);
    g.C = (float*)workspace; g.ldc = N;
    grid = (unsigned)(g.mt * g.nt * g.splits);
  }
  const int vec = g.vec_ok ? 1 : 0;
  // too few 256 x 256 tiles for the chip: the 256 x 128 tile (plain NT with bias / ReLU; MIRL_GEMM3_MID=0 keeps the big tile)
  static const int mid_default = getenv("MIRL_GEMM3_MID") ? atoi(getenv("MIRL_GEMM3_MID")) : 1;
  const int mid_env = g_g3_mid_mode >= 0 ? g_g3_mid_mode : mid_default;
  if (mid_env && layout == 0 && !mul && !w2 && !gsum && vec && (N % 4) == 0 && (int64_t)g.mt * g.nt < 192 && N > 128 &&
      ((int64_t)g.mt * ((N + G3M_BROWS - 1) / G3M_BROWS) >= 192 || (int64_t)g.mt * g.nt < 24)) {
    // (between: neither tiling fills the chip — the callers' area gate, models/torch/gemm3.py _MIN_AREA, keeps those
    //  products on the library; tiny products take the smaller tile for its shorter tail)
    static bool mid_attr = false;
    if (!mid_attr) { MIRL_HIP(hipFuncSetAttribute((const void*)k_gemm3_mid, hipFuncAttributeMaxDynamicSharedMemorySize, G3M_LDS)); mid_attr = true; }
    g.nt = (int)((N + G3M_BROWS - 1) / G3M_BROWS);
    const unsigned mgrid = (unsigned)(8 * ((g.mt + 7) / 8) * g.nt);
    ProfScope ps("k_gemm3_nt_mid", 4.0 * ((double)M * K + (double)N * K + (double)M * N), st, 2.0 * (double)M * (double)N * (double)K);
    void* kargs[] = {(void*)&g};
    MIRL_HIP(hipLaunchKernel((const void*)k_gemm3_mid, dim3(mgrid), dim3(512), kargs, G3M_LDS, st));
    return MIRL_OK;
  }
  static bool attr[4][2] = {{false, false}, {false, false}, {false, false}, {false, false}};
  const void* fns[4][2] = {{(const void*)k_gemm3<true, true, 0, false>, (const void*)k_gemm3<true, true, 0, true>},
                           {(const void*)k_gemm3<true, false, 0, false>, (const void*)k_gemm3<true, false, 0, true>},
                           {(const void*)k_gemm3<false, false, 0, false>, (const void*)k_gemm3<false, false, 0, true>},
                           {(const void*)k_gemm3<true, true, 1, false>, (const void*)k_gemm3<true, true, 1, true>}};
  const int which = (mul && !gsum) ? 3 : layout;
  const void* fn = fns[which][vec];
  if (gsum) {
    if (!vec || layout != 1 || !mul || mul_shift != 5 || !pre || !part || (ldgsum % 4) || ((uintptr_t)gsum % 16) || ((uintptr_t)part % 16) || (M % 32))
      return fail(MIRL_ERR_ARG, "gemm3: the fused feature-product backward needs the NN form, groups of 32 rows and 16-byte aligned rows");
    static bool qp_attr = false;
    fn = (const void*)k_gemm3<true, false, 3, true>;
    if (!qp_attr) { MIRL_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS)); qp_attr = true; }
  } else if (w2) {
    if (!vec || layout != 0 || mul || !part) return fail(MIRL_ERR_ARG, "gemm3: the fused following layer needs the NT form with 16-byte aligned rows");
    static bool hd_attr = false;
    fn = (const void*)k_gemm3<true, true, 2, true>;
    if (!hd_attr) { MIRL_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS)); hd_attr = true; }
  } else if (layout == 2 && N <= 64 && narrow_env) {
    // weight gradient of a narrow layer: eight waves stacked along M, 12 MFMAs per K-step (k_gemm3 NARROW)
    static bool nr_attr[2] = {false, false};
    const void* nr[2] = {(const void*)k_gemm3<false, false, 0, false, true>, (const void*)k_gemm3<false, false, 0, true, true>};
    fn = nr[vec];
    if (!nr_attr[vec]) { MIRL_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS)); nr_attr[vec] = true; }
  } else
  if (!attr[which][vec]) { MIRL_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS)); attr[which][vec] = true; }
  {
    const double flop = 2.0 * (double)M * (double)N * (double)K;
    // HBM bytes: both operands read once, the result (and the pre-product embedding / the multiplier rows) written / read once
    double bytes = 4.0 * ((double)M * K + (double)N * K + (double)M * N * (layout == 2 ? (double)g.splits : 1.0));
    if (mul) bytes += 4.0 * ((pre ? (double)M * N : 0.0) + (double)(M >> mul_shift) * N * (gsum ? 2.0 : 1.0));
    if (w2) bytes += 4.0 * (8.0 * (double)N + 8.0 * (double)M * (double)((N + 63) / 64)) - (C ? 0.0 : 4.0 * (double)M * N);
    ProfScope ps(gsum ? "k_gemm3_nn_qp" : mul ? "k_gemm3_nt_mul" : w2 ? "k_gemm3_nt_head" : layout == 0 ? "k_gemm3_nt" : layout == 1 ? "k_gemm3_nn" : "k_gemm3_tn", bytes, st, flop);
    void* kargs[] = {(void*)&g};
    MIRL_HIP(hipLaunchKernel(fn, dim3(grid), dim3(512), kargs, G3_LDS, st));
  }
  if (layout == 2) {
    ProfScope ps("k_gemm3_reduce", (double)(g.splits + 1) * (double)M * (double)N * 4.0, st);
    const int64_t total = M * (N / 4);
    unsigned rg = (unsigned)((total + 255) / 256); if (rg > 4096) rg = 4096;
    hipLaunchKernelGGL(k_gemm3_reduce, dim3(rg), dim3(256), 0, st, (const float*)workspace, g.splits, M, N, C, ldc);
    MIRL_LAUNCH_CHECK();
  }
  return MIRL_OK;
}

extern "C" int mirl_gemm3(int32_t layout, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B,
                          int64_t ldb, float* C, int64_t ldc, const float* bias, int32_t relu, void* workspace,
                          int64_t workspace_bytes, void* stream) {
  return g3_launch(layout, M, N, K, A, lda, B, ldb, C, ldc, bias, relu, workspace, workspace_bytes, nullptr, 0, 0, nullptr, 0, stream);
}

extern "C" int mirl_gemm3_mid_set(int32_t mode) {
  if (mode < -1 || mode > 1) return mirl::fail(MIRL_ERR_ARG, "gemm3_mid_set: mode is -1 (environment default), 0 or 1");
  mirl::g_g3_mid_mode = mode;
  return MIRL_OK;
}

extern "C" int mirl_gemm3_nt_mul(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                                 float* C, int64_t ldc, const float* bias, int32_t relu, const float* mul, int64_t ldmul,
                                 int32_t group_shift, float* pre, int64_t ldpre, void* stream) {
  using namespace mirl;
  if (!mul || group_shift < 0 || group_shift > 30 || ldmul < N || (pre && ldpre < N))
    return fail(MIRL_ERR_ARG, "gemm3_nt_mul: bad multiplier / pre-activation arguments");
  return g3_launch(0, M, N, K, A, lda, B, ldb, C, ldc, bias, relu, nullptr, 0, mul, ldmul, group_shift, pre, ldpre, stream);
}

// ---- a data gradient g W whose consumer is the backward of the IQN feature product: that backward in the epilogue ----
extern "C" int mirl_gemm3_nn_qp_partial_rows(int64_t M, int64_t* rows) {
  if (!rows || M < 1) return mirl::fail(MIRL_ERR_ARG, "gemm3_nn_qp_partial_rows: M >= 1");
  *rows = 2 * ((M + 255) / 256);
  return MIRL_OK;
}

extern "C" int mirl_gemm3_nn_qp(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                                const float* emb, int64_t ldemb, const float* x, int64_t ldx, float* d_pre, int64_t ldd,
                                float* dx, int64_t lddx, float* db_partial, void* stream) {
  using namespace mirl;
  if (!emb || !x || !d_pre || !dx || !db_partial || ldemb < N || ldx < N || lddx < N || (ldemb % 4) || ((uintptr_t)emb % 16))
    return fail(MIRL_ERR_ARG, "gemm3_nn_qp: bad embedding / feature / output arguments");
  return g3_launch(1, M, N, K, A, lda, B, ldb, d_pre, ldd, nullptr, 0, nullptr, 0, x, ldx, 5, const_cast<float*>(emb), ldemb, stream,
                   nullptr, db_partial, dx, lddx);
}

// ---- a wide layer and the narrow one that follows it, in one pass over the activation ----------------------------
// out2[row][o] = bias2[o] + sum_n relu(A . B^T + bias)[row][n] * w2[o][n], o < O <= 8: the dueling head's advantage and
// value OUTPUT layers (policies/torch/dqn.py:101-112, 50-66: A + 1 units) computed in the epilogue of the joint hidden
// layer's product, so that the (rows, 1024) hidden activation is not read back by two more GEMMs — and, in the no-grad
// passes (C == NULL), not written at all.  w2 is [8][N] (rows >= O zero); `part` holds 8 floats per row and 64-column
// block: mirl_gemm3_head_workspace_bytes.
extern "C" int mirl_gemm3_head_workspace_bytes(int64_t M, int64_t N, int64_t* bytes) {
  if (!bytes || M < 1 || N < 1) return mirl::fail(MIRL_ERR_ARG, "bad gemm3_head_workspace_bytes arguments");
  *bytes = 8 * (int64_t)sizeof(float) * M * ((N + 255) / 256 * 4);
  return MIRL_OK;
}

extern "C" int mirl_gemm3_nt_head(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                                  float* C, int64_t ldc, const float* bias, int32_t relu, const float* w2, int32_t O,
                                  const float* bias2, float* out2, int64_t ldo, void* workspace, int64_t workspace_bytes,
                                  void* stream) {
  using namespace mirl;
  int64_t need = 0;
  if (O < 1 || O > 8 || !w2 || !out2 || ldo < O || !workspace || mirl_gemm3_head_workspace_bytes(M, N, &need) || workspace_bytes < need ||
      ((uintptr_t)workspace % 16) || (N % 4))
    return fail(MIRL_ERR_ARG, "bad gemm3_nt_head arguments (1 <= O <= 8, N % 4 == 0, 16-byte aligned workspace of head_workspace_bytes)");
  if (!C) ldc = N;
  int rc = g3_launch(0, M, N, K, A, lda, B, ldb, C, ldc, bias, relu, nullptr, 0, nullptr, 0, 0, nullptr, 0, stream, w2,
                     (float*)workspace);
  if (rc) return rc;
  const int ncb = (int)((N + 255) / 256 * 4);
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps("k_g3_head_reduce", 4.0 * (8.0 * ncb + O) * (double)M, st);
  unsigned grid = (unsigned)((2 * M + 255) / 256); if (grid > 16384) grid = 16384;
  hipLaunchKernelGGL(k_g3_head_reduce, dim3(grid), dim3(256), 0, st, (const float*)workspace, ncb, M, (int)O, bias2, out2, ldo);
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}
