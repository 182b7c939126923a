// np_emul.h — exact arithmetic model of the reference's priority tree.
//
// The reference keeps its sum-tree as a Python list whose slots are Python
// floats/ints, np.float32 or np.float64 scalars, and combines them with the
// plain `+`, `-`, `>` operators (rltime/history/data_structures/segment_tree.py:
// 87-97,116-142).  Under NumPy-2 promotion the rounding of every node therefore
// depends on the *kinds* of its two children.  To get
// bit-identical sampled indices we carry that kind next to every value:
//
//   KW  (0) weak Python scalar (float / int)  — value held in f64
//   K32 (1) np.float32                          — value is f32-representable
//   K64 (2) np.float64
//
//   KW  op KW  -> KW  (f64 arithmetic)        KW  op K32 -> K32 (KW cast to f32!)
//   K32 op K32 -> K32 (f32 arithmetic)        K64 op any -> K64 (f64 arithmetic)
//
// Everything here is a pure function usable from device code and — for the
// CPU-only unit tests of this header (tests/test_host_logic.py, which compare
// it with the golden tree fixtures) — from host code.  Compile with
// -ffp-contract=off: a fused multiply-add would change the roundings.
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define MIRL_HD __host__ __device__ __forceinline__
#else
#define MIRL_HD inline
#endif

namespace mirl {

enum : uint8_t { KW = 0, K32 = 1, K64 = 2 };

struct TV {      // tagged value
  double v;
  uint8_t k;
};

MIRL_HD uint8_t promote(uint8_t a, uint8_t b) {
  if (a == K64 || b == K64) return K64;
  if (a == K32 || b == K32) return K32;
  return KW;
}

// `a + b` (segment_tree.py:93-96 via operator.add)
MIRL_HD TV tadd(TV a, TV b) {
  TV r;
  r.k = promote(a.k, b.k);
  if (r.k == K32) {
    float x = (float)a.v, y = (float)b.v;
    r.v = (double)(x + y);
  } else {
    r.v = a.v + b.v;
  }
  return r;
}

// `a - b` (segment_tree.py:140, prefixsum -= value)
MIRL_HD TV tsub(TV a, TV b) {
  TV r;
  r.k = promote(a.k, b.k);
  if (r.k == K32) {
    float x = (float)a.v, y = (float)b.v;
    r.v = (double)(x - y);
  } else {
    r.v = a.v - b.v;
  }
  return r;
}

// `a > b` (segment_tree.py:137).  A weak Python scalar compared with an
// np.float32 is first cast to float32; every other pairing compares the exact
// values.
MIRL_HD bool tgreater(TV a, TV b) {
  if ((a.k == K32 && b.k == KW) || (a.k == KW && b.k == K32))
    return (float)a.v > (float)b.v;
  return a.v > b.v;
}

// mass_i = random.random() * seg + i * seg, seg = total / batch
// (prioritized_replay_history.py:235-238).  `u` and `i`/`batch` are weak.
MIRL_HD TV stratum_mass(TV total, int batch, int i, double u) {
  TV m;
  if (total.k == K32) {
    float seg = (float)total.v / (float)batch;
    float a = (float)u * seg;
    float b = (float)i * seg;
    m.v = (double)(a + b);
    m.k = K32;
  } else {
    double seg = total.v / (double)batch;
    double a = u * seg;
    double b = (double)i * seg;
    m.v = a + b;
    m.k = total.k;
  }
  return m;
}

// find_prefixsum_idx (segment_tree.py:116-142) on a (value, kind) heap.
MIRL_HD int64_t tagged_descend(const double* tv, const uint8_t* tk,
                               int64_t capacity, TV mass) {
  int64_t pos = 1;
  while (pos < capacity) {
    TV left;
    left.v = tv[2 * pos];
    left.k = tk[2 * pos];
    if (tgreater(left, mass)) {
      pos = 2 * pos;
    } else {
      mass = tsub(mass, left);
      pos = 2 * pos + 1;
    }
  }
  return pos - capacity;
}

// ---------------------------------------------------------------------------
// np.add.reduce over a contiguous 1-D array: NumPy's pairwise summation
// (numpy/_core/src/umath/loops_utils.h.src, @TYPE@_pairwise_sum): <8 elements
// sequential; <=128 eight interleaved accumulators combined as
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) then the tail; larger inputs split at
// n/2 rounded down to a multiple of 8.  Verified against np.mean for
// n in [2, 1000] in tests/test_host_logic.py.
template <class T, class Get>
MIRL_HD T np_pairwise_block(Get get, int lo, int n) {
  if (n < 8) {
    T r = (T)0;
    for (int i = 0; i < n; ++i) r = r + get(lo + i);
    return r;
  }
  T r0 = get(lo + 0), r1 = get(lo + 1), r2 = get(lo + 2), r3 = get(lo + 3);
  T r4 = get(lo + 4), r5 = get(lo + 5), r6 = get(lo + 6), r7 = get(lo + 7);
  int i = 8;
  for (; i < n - (n % 8); i += 8) {
    r0 = r0 + get(lo + i + 0);
    r1 = r1 + get(lo + i + 1);
    r2 = r2 + get(lo + i + 2);
    r3 = r3 + get(lo + i + 3);
    r4 = r4 + get(lo + i + 4);
    r5 = r5 + get(lo + i + 5);
    r6 = r6 + get(lo + i + 6);
    r7 = r7 + get(lo + i + 7);
  }
  T res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
  for (; i < n; ++i) res = res + get(lo + i);
  return res;
}

template <class T, class Get>
MIRL_HD T np_pairwise_sum(Get get, int lo, int n) {
  // iterative form of the recursion: collect <=128-element leaves left to
  // right on an explicit stack of (sum, depth) and merge equal depths —
  // identical association to the recursive definition because the split
  // point depends only on n.
  if (n <= 128) return np_pairwise_block<T>(get, lo, n);
  // explicit recursion stack (depth <= 24 covers n < 2^31)
  int stack_lo[32], stack_n[32];
  T acc[32];
  uint8_t state[32];  // 0 = need left, 1 = need right, 2 = done
  int sp = 0;
  stack_lo[0] = lo; stack_n[0] = n; state[0] = 0; acc[0] = (T)0;
  T ret = (T)0;
  while (sp >= 0) {
    int cn = stack_n[sp], clo = stack_lo[sp];
    if (cn <= 128) {
      ret = np_pairwise_block<T>(get, clo, cn);
      --sp;
      continue;
    }
    int n2 = cn / 2;
    n2 -= n2 % 8;
    if (state[sp] == 0) {
      state[sp] = 1;
      ++sp;
      stack_lo[sp] = clo; stack_n[sp] = n2; state[sp] = 0;
    } else if (state[sp] == 1) {
      acc[sp] = ret;  // left result
      state[sp] = 2;
      ++sp;
      stack_lo[sp] = clo + n2; stack_n[sp] = cn - n2; state[sp] = 0;
    } else {
      ret = acc[sp] + ret;
      --sp;
    }
  }
  return ret;
}

// ---------------------------------------------------------------------------
// Per-transition loss slot encoding: the reference stores `sample['loss']` as
// the Python float 1.0 until the first update_losses touches it
// (prioritized_replay_history.py:120,141) and as np.float32 afterwards
// (abs(np.float32)+eps, :263).  We keep one f32 per transition and encode the
// "still the weak Python 1.0" state as a negative sentinel.
#define MIRL_LOSS_FRESH (-1.0f)

struct PrioParams {
  int nstep_train;
  double alpha;              // Python float
  double max_weight_factor;  // Python float
};

// _recalc_weighted_priority (prioritized_replay_history.py:174-208).
// `get(t)` returns the stored f32 loss slot of step t of the sequence.
// pow() is the platform's; the reference's is glibc powf/pow — leaf VALUES are
// therefore compared with a 1-ulp tolerance, leaf KINDS exactly.
template <class Get>
MIRL_HD TV seq_priority(Get get, const PrioParams& p) {
  TV out;
  const int T = p.nstep_train;
  if (T == 1) {
    float l = get(0);
    if (l < 0.0f) {           // 1.0 ** alpha on Python floats
      out.v = 1.0;
      out.k = KW;
    } else {                  // np.float32 ** python float -> float32 pow
      out.v = (double)(float)pow((double)l, (double)(float)p.alpha);
      out.k = K32;
    }
    return out;
  }
  bool any_fresh = false;
  for (int t = 0; t < T; ++t) any_fresh |= (get(t) < 0.0f);
  if (any_fresh) {
    // np.max / np.mean over a list holding a Python float -> float64 array
    auto g = [&](int t) -> double {
      float l = get(t);
      return l < 0.0f ? 1.0 : (double)l;
    };
    double mx = g(0);
    for (int t = 1; t < T; ++t) { double x = g(t); mx = x > mx ? x : mx; }
    double mean = np_pairwise_sum<double>(g, 0, T) / (double)T;
    double w = p.max_weight_factor * mx;
    double m2 = (1.0 - p.max_weight_factor) * mean;
    double mixed = w + m2;
    out.v = pow(mixed, p.alpha);
    out.k = K64;
  } else {
    auto g = [&](int t) -> float { return get(t); };
    float mx = g(0);
    for (int t = 1; t < T; ++t) { float x = g(t); mx = x > mx ? x : mx; }
    float s = np_pairwise_sum<float>(g, 0, T);
    // _methods.py _mean: ret.dtype.type(ret / rcount) — f32 sum / intp count
    // is evaluated in f64 and rounded back to f32.
    float mean = (float)((double)s / (double)T);
    float w = (float)p.max_weight_factor * mx;
    float m2 = (float)(1.0 - p.max_weight_factor) * mean;
    float mixed = w + m2;
    out.v = (double)(float)pow((double)mixed, (double)(float)p.alpha);
    out.k = K32;
  }
  return out;
}

}  // namespace mirl
