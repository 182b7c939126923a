// conv_col.hip — the two data movements that turn the BACKWARD of a small NHWC convolution into the wide GEMMs
// csrc/gemm3.hip already runs (exact three-way bf16 split, f32 results):
//
//   weight gradient  dW[(ky,kx,c)][f]   = sum over rows of  col[row][(ky,kx,c)] * g[row][f]        "TN", K = rows
//   data gradient    dcol[row][(ky,kx,c)] = sum over f of   g[row][f] * W[f][(ky,kx,c)]            "NN", K = F
//                    dx[n][y][x][c]      = sum over the windows (oy, ox, ky, kx) that cover (y, x) of dcol
//
// with row = (n, oy, ox) one output position and col[row] the window it saw.  What it replaces: MIOpen's igemm_wrw /
// igemm_bwd kernels behind the autograd gradients of the reference's conv layers 2 and 3 (rltime/models/torch/modules/
// cnn.py:43-50) — 5.4 ms of the learner step at BASELINE configs[3] plus this library's own four-GEMM layer-2 data
// gradient (2.2 ms).  Both kernels here are pure HBM streams (no arithmetic besides the <= kh*kw additions of col2im,
// summed in a fixed (ky, kx) order: deterministic).
//
// NHWC throughout: x is [n][h][w][c], a window row of col is kh segments of kw*c CONTIGUOUS floats of x, so im2col moves
// 16-byte vectors (c % 4 == 0) and col2im reads 16 bytes per lane, c/4 consecutive lanes per input pixel.
#include "common.hpp"

namespace mirl {

typedef float cc_f4 __attribute__((ext_vector_type(4)));      // a native vector (the nontemporal builtins do not take HIP's float4 struct)

struct ColShape {
  int n, h, w, c, kh, kw, s, oh, ow;
  int kq;        // float4 per col row: kh*kw*c/4
  int segq;      // float4 per contiguous segment: kw*c/4
  int cq;        // c/4
};

__global__ void __launch_bounds__(256)
k_im2col_nhwc(ColShape d, const cc_f4* __restrict__ x, cc_f4* __restrict__ col, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const unsigned row = (unsigned)(i / d.kq), q = (unsigned)(i - (int64_t)row * d.kq);
    const unsigned ky = q / d.segq, rem = q - ky * d.segq;
    const unsigned f = row / (unsigned)(d.oh * d.ow), p = row - f * (unsigned)(d.oh * d.ow);
    const unsigned oy = p / d.ow, ox = p - oy * d.ow;
    const int64_t src = (((int64_t)f * d.h + oy * d.s + ky) * d.w + ox * d.s) * d.cq + rem;
    __builtin_nontemporal_store(x[src], col + i);
  }
}

// dx = col2im(dcol) [* (mask > 0)]: one thread per float4 of dx
__global__ void __launch_bounds__(256)
k_col2im_nhwc(ColShape d, const cc_f4* __restrict__ col, const cc_f4* __restrict__ mask, cc_f4* __restrict__ dx, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const unsigned pix = (unsigned)(i / d.cq), c4 = (unsigned)(i - (int64_t)pix * d.cq);
    const unsigned f = pix / (unsigned)(d.h * d.w), p = pix - f * (unsigned)(d.h * d.w);
    const int y = (int)(p / d.w), xx = (int)(p - (unsigned)y * d.w);
    cc_f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int ky = 0; ky < d.kh; ++ky) {
      const int ty = y - ky;
      if (ty < 0) break;
      const int oy = ty / d.s;
      if (oy * d.s != ty || oy >= d.oh) continue;
      for (int kx = 0; kx < d.kw; ++kx) {
        const int tx = xx - kx;
        if (tx < 0) break;
        const int ox = tx / d.s;
        if (ox * d.s != tx || ox >= d.ow) continue;
        acc += __builtin_nontemporal_load(col + (((int64_t)f * d.oh + oy) * d.ow + ox) * d.kq + (ky * d.kw + kx) * d.cq + c4);
      }
    }
    if (mask) {
      const cc_f4 m = mask[i];
      acc.x = m.x > 0.f ? acc.x : 0.f; acc.y = m.y > 0.f ? acc.y : 0.f;
      acc.z = m.z > 0.f ? acc.z : 0.f; acc.w = m.w > 0.f ? acc.w : 0.f;
    }
    dx[i] = acc;
  }
}

static int col_shape(int64_t n, int32_t h, int32_t w, int32_t c, int32_t kh, int32_t kw, int32_t s, ColShape* d) {
  if (n < 1 || h < 1 || w < 1 || c < 4 || (c % 4) || kh < 1 || kw < 1 || s < 1 || kh > h || kw > w)
    return fail(MIRL_ERR_ARG, "conv_col: need n, h, w >= 1, c a multiple of 4, 1 <= kh <= h, 1 <= kw <= w, stride >= 1");
  d->n = (int)n; d->h = h; d->w = w; d->c = c; d->kh = kh; d->kw = kw; d->s = s;
  d->oh = (h - kh) / s + 1; d->ow = (w - kw) / s + 1;
  d->kq = kh * kw * c / 4; d->segq = kw * c / 4; d->cq = c / 4;
  // 32-bit row / pixel indices inside the kernels
  if (n * (int64_t)d->oh * d->ow >= (1LL << 31) || n * (int64_t)h * w >= (1LL << 31))
    return fail(MIRL_ERR_ARG, "conv_col: more than 2^31 positions");
  return MIRL_OK;
}

}  // namespace mirl

extern "C" int mirl_im2col_nhwc(int64_t n, int32_t h, int32_t w, int32_t c, int32_t kh, int32_t kw, int32_t stride,
                                const float* x, float* col, void* stream) {
  using namespace mirl;
  ColShape d;
  if (int rc = col_shape(n, h, w, c, kh, kw, stride, &d)) return rc;
  if (!x || !col || ((uintptr_t)x % 16) || ((uintptr_t)col % 16)) return fail(MIRL_ERR_ARG, "im2col_nhwc: null / misaligned buffer");
  const int64_t total = n * (int64_t)d.oh * d.ow * d.kq;
  unsigned grid = (unsigned)((total + 255) / 256); if (grid > 16384) grid = 16384;
  ProfScope ps("k_im2col_nhwc", 16.0 * (double)total + 4.0 * (double)n * h * w * c, (hipStream_t)stream);
  hipLaunchKernelGGL(k_im2col_nhwc, dim3(grid), dim3(256), 0, (hipStream_t)stream, d, reinterpret_cast<const cc_f4*>(x),
                     reinterpret_cast<cc_f4*>(col), total);
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}

extern "C" int mirl_col2im_nhwc(int64_t n, int32_t h, int32_t w, int32_t c, int32_t kh, int32_t kw, int32_t stride,
                                const float* col, const float* relu_mask, float* dx, void* stream) {
  using namespace mirl;
  ColShape d;
  if (int rc = col_shape(n, h, w, c, kh, kw, stride, &d)) return rc;
  if (!dx || !col || ((uintptr_t)dx % 16) || ((uintptr_t)col % 16) || ((uintptr_t)relu_mask % 16))
    return fail(MIRL_ERR_ARG, "col2im_nhwc: null / misaligned buffer");
  const int64_t total = n * (int64_t)h * w * d.cq;
  unsigned grid = (unsigned)((total + 255) / 256); if (grid > 16384) grid = 16384;
  ProfScope ps("k_col2im_nhwc", 16.0 * (double)n * d.oh * d.ow * d.kq + (relu_mask ? 32.0 : 16.0) * (double)total, (hipStream_t)stream);
  hipLaunchKernelGGL(k_col2im_nhwc, dim3(grid), dim3(256), 0, (hipStream_t)stream, d, reinterpret_cast<const cc_f4*>(col),
                     reinterpret_cast<const cc_f4*>(relu_mask), reinterpret_cast<cc_f4*>(dx), total);
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}
