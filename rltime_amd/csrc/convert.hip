// convert.hip — the CNN's input conversion as one HBM pass.
#include "common.hpp"
#include <cstdlib>

// ---------------------------------------------------------------------------
// uint8 NCHW frames -> float32 NHWC * scale in one pass (the CNN's input
// conversion, rltime/models/torch/modules/cnn.py:44-45, fused with the layout
// change MIOpen's NHWC kernels want).  Stock PyTorch needs a u8 transpose pass
// plus a convert-and-scale pass.
//
// HBM-bound: per pixel C bytes in, 4*C bytes out (C = 4: 28 224 B in and
// 112 896 B out per 84x84x4 frame).  Every global access is 16 B per lane and
// every wave-level instruction covers one contiguous span:
//   load : a 1024-pixel tile of one frame = C plane segments of 1 KiB; lane l of
//          wave w reads bytes [16 l, 16 l + 16) of plane w's segment (one
//          global_load_dwordx4 per lane, non-temporal — the frame is not re-read);
//   LDS  : the C x 1 KiB tile, written as 16 B per lane, read back as bytes
//          ([plane][pixel]: consecutive lanes read consecutive bytes -> 16 dwords
//          per instruction, broadcast within a dword, no bank conflict);
//   store: instruction k of a wave writes the float4 of pixels 256 k + tid, i.e.
//          64 consecutive float4 = 1 KiB contiguous (round 1 wrote 64 B per lane
//          with a 64 B lane stride: every store instruction touched a quarter of
//          each of 64 lines, and the kernel ran at 52 % of the HBM peak).
namespace mirl {

typedef unsigned int cv_u32x4 __attribute__((ext_vector_type(4)));
typedef float cv_f32x4 __attribute__((ext_vector_type(4)));

#define MIRL_CV_TILE 1024

// One workgroup walks `per_wg` consecutive tiles of one frame.  The tile lives in
// a double-buffered LDS slab (one barrier per tile) and the next tile's 16 B are
// already in flight in a register while the current one is stored.
template <int NTL, int NTS>
__global__ void __launch_bounds__(256)
k_frames_to_f32_nhwc4(int64_t N, int HW, int tiles, int per_wg, const uint8_t* __restrict__ src, float scale,
                      float* __restrict__ dst) {
  __shared__ cv_u32x4 s_tile[2][4 * (MIRL_CV_TILE / 16)];
  const int splits = (tiles + per_wg - 1) / per_wg;
  const int64_t n = blockIdx.x / splits;
  const int t0 = (int)(blockIdx.x - n * splits) * per_wg;
  const int t1 = t0 + per_wg < tiles ? t0 + per_wg : tiles;
  const int tid = threadIdx.x;
  const int plane = tid >> 6, chunk = tid & 63;
  const uint8_t* g_plane = src + (n * 4 + plane) * (int64_t)HW;
  cv_u32x4 cur = {0u, 0u, 0u, 0u};
  if (t0 * MIRL_CV_TILE + chunk * 16 < HW) {
    const cv_u32x4* p = (const cv_u32x4*)(g_plane + t0 * MIRL_CV_TILE) + chunk;
    cur = NTL ? __builtin_nontemporal_load(p) : *p;
  }
  for (int t = t0; t < t1; ++t) {
    const int p_tile = t * MIRL_CV_TILE;
    const int left = HW - p_tile;                     // pixels of this tile (multiple of 16)
    cv_u32x4* slab = s_tile[t & 1];
    slab[plane * 64 + chunk] = cur;
    __syncthreads();
    if (t + 1 < t1 && p_tile + MIRL_CV_TILE + chunk * 16 < HW) {
      const cv_u32x4* p = (const cv_u32x4*)(g_plane + p_tile + MIRL_CV_TILE) + chunk;
      cur = NTL ? __builtin_nontemporal_load(p) : *p;
    }
    const uint8_t* sb = (const uint8_t*)slab;
    cv_f32x4* out = (cv_f32x4*)(dst + (n * (int64_t)HW + p_tile) * 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int pix = k * 256 + tid;
      if (pix < left) {
        cv_f32x4 v;
        v.x = (float)sb[pix] * scale;
        v.y = (float)sb[MIRL_CV_TILE + pix] * scale;
        v.z = (float)sb[2 * MIRL_CV_TILE + pix] * scale;
        v.w = (float)sb[3 * MIRL_CV_TILE + pix] * scale;
        if (NTS) __builtin_nontemporal_store(v, out + pix); else out[pix] = v;
      }
    }
  }
}

// generic shapes (C != 4 or a plane size that is not a multiple of 16): one
// lane per (frame, pixel), C byte loads (coalesced per plane across lanes) and
// C consecutive float stores.
__global__ void __launch_bounds__(256)
k_frames_to_f32_nhwc_any(int64_t N, int C, int HW, const uint8_t* __restrict__ src, float scale,
                         float* __restrict__ dst) {
  int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= N * HW) return;
  const int64_t n = idx / HW;
  const int p = (int)(idx - n * HW);
  const uint8_t* s = src + n * (int64_t)C * HW + p;
  float* d = dst + idx * C;
  for (int ch = 0; ch < C; ++ch) d[ch] = (float)s[(int64_t)ch * HW] * scale;
}
}  // namespace mirl

// per_wg: tiles per workgroup (0 = heuristic); flags bit 0: plain (cached) loads,
// bit 1: plain stores instead of non-temporal ones (tools/convert_probe.py sweeps them).
extern "C" int mirl_frames_to_f32_nhwc_ex(int64_t N, int32_t C, int32_t HW, const uint8_t* src, float scale, float* dst,
                                          int32_t per_wg, int32_t flags, void* stream) {
  if (N <= 0 || C <= 0 || HW <= 0 || !src || !dst) return mirl::fail(MIRL_ERR_ARG, "bad frames_to_f32_nhwc arguments");
  hipStream_t st = (hipStream_t)stream;
  mirl::ProfScope ps("k_frames_to_f32_nhwc", (double)N * C * HW * 5.0, st);
  if (C == 4 && (HW % 16) == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0) {
    const int tiles = (HW + MIRL_CV_TILE - 1) / MIRL_CV_TILE;
    // one 1024-pixel tile per workgroup: measured best at the config-D block
    // (62 464 frames: 1.50 ms = 73 % of the HBM peak with cached loads + non-temporal
    // stores; 7 tiles per workgroup 1.63 ms, non-temporal loads 1.56 ms —
    // profiles/r02_convert_variant_sweep.jsonl)
    if (per_wg <= 0) per_wg = 1;
    if (per_wg > tiles) per_wg = tiles;
    const int64_t blocks = N * ((tiles + per_wg - 1) / per_wg);
    if (blocks >= (1LL << 31)) return mirl::fail(MIRL_ERR_ARG, "frames_to_f32_nhwc: too many frames for one launch");
    const dim3 grid((unsigned)blocks), block(256);
    switch (flags & 3) {
      case 0: hipLaunchKernelGGL((mirl::k_frames_to_f32_nhwc4<1, 1>), grid, block, 0, st, N, (int)HW, tiles, (int)per_wg, src, scale, dst); break;
      case 1: hipLaunchKernelGGL((mirl::k_frames_to_f32_nhwc4<0, 1>), grid, block, 0, st, N, (int)HW, tiles, (int)per_wg, src, scale, dst); break;
      case 2: hipLaunchKernelGGL((mirl::k_frames_to_f32_nhwc4<1, 0>), grid, block, 0, st, N, (int)HW, tiles, (int)per_wg, src, scale, dst); break;
      default: hipLaunchKernelGGL((mirl::k_frames_to_f32_nhwc4<0, 0>), grid, block, 0, st, N, (int)HW, tiles, (int)per_wg, src, scale, dst); break;
    }
  } else {
    const int64_t n = N * HW;
    if ((n + 255) / 256 >= (1LL << 31)) return mirl::fail(MIRL_ERR_ARG, "frames_to_f32_nhwc: too many pixels for one launch");
    hipLaunchKernelGGL(mirl::k_frames_to_f32_nhwc_any, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, N, (int)C, (int)HW, src, scale, dst);
  }
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}

extern "C" int mirl_frames_to_f32_nhwc(int64_t N, int32_t C, int32_t HW, const uint8_t* src, float scale, float* dst, void* stream) {
  static const int per_wg = getenv("MIRL_CONVERT_PER_WG") ? atoi(getenv("MIRL_CONVERT_PER_WG")) : 0;
  static const int flags = getenv("MIRL_CONVERT_FLAGS") ? atoi(getenv("MIRL_CONVERT_FLAGS")) : 1;   // cached loads, nt stores
  return mirl_frames_to_f32_nhwc_ex(N, C, HW, src, scale, dst, per_wg, flags, stream);
}
