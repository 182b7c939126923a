// convert.hip — the CNN's input conversion as one HBM pass.
#include "common.hpp"

// ---------------------------------------------------------------------------
// uint8 NCHW frames -> float32 NHWC * scale in one pass (the CNN's input
// conversion, rltime/models/torch/modules/cnn.py:44-45, fused with the layout
// change MIOpen's NHWC kernels want).  Stock PyTorch needs a u8 transpose pass
// plus a convert-and-scale pass.  One lane handles 4 consecutive pixels of all C
// planes: 4-byte loads per plane, 16-byte stores when C == 4.
namespace mirl {
__global__ void __launch_bounds__(256)
k_frames_to_f32_nhwc(int64_t N, int C, int HW, const uint8_t* __restrict__ src, float scale, float* __restrict__ dst) {
  const int groups = (HW + 3) / 4;
  int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= N * groups) return;
  const int64_t n = idx / groups;
  const int p0 = (int)(idx - n * groups) * 4;
  const uint8_t* s = src + n * (int64_t)C * HW;
  float* d = dst + (n * HW + p0) * (int64_t)C;
  const int np = HW - p0 < 4 ? HW - p0 : 4;
  if (C == 4 && np == 4 && (HW & 3) == 0) {
    uint32_t a = *(const uint32_t*)(s + p0), b = *(const uint32_t*)(s + HW + p0);
    uint32_t c = *(const uint32_t*)(s + 2 * HW + p0), e = *(const uint32_t*)(s + 3 * HW + p0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float4 v;
      v.x = (float)((a >> (8 * k)) & 0xFF) * scale; v.y = (float)((b >> (8 * k)) & 0xFF) * scale;
      v.z = (float)((c >> (8 * k)) & 0xFF) * scale; v.w = (float)((e >> (8 * k)) & 0xFF) * scale;
      *(float4*)(d + 4 * k) = v;
    }
  } else {
    for (int k = 0; k < np; ++k)
      for (int ch = 0; ch < C; ++ch) d[k * C + ch] = (float)s[(int64_t)ch * HW + p0 + k] * scale;
  }
}
}  // namespace mirl

extern "C" int mirl_frames_to_f32_nhwc(int64_t N, int32_t C, int32_t HW, const uint8_t* src, float scale, float* dst, void* stream) {
  if (N <= 0 || C <= 0 || HW <= 0 || !src || !dst) return mirl::fail(MIRL_ERR_ARG, "bad frames_to_f32_nhwc arguments");
  int64_t n = N * ((HW + 3) / 4);
  hipLaunchKernelGGL(mirl::k_frames_to_f32_nhwc, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, N, (int)C, (int)HW, src, scale, dst);
  MIRL_LAUNCH_CHECK();
  return MIRL_OK;
}
