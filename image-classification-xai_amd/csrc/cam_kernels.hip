// K3: Grad-CAM channel-weighted reduce + bilinear up-sample for gfx950.
//
// gradcam: one 1024-lane workgroup per image.  A wave owns channels {wave, wave+16, ...}; for
// a channel its lanes hold the h*w gradient and activation values (pixel = lane + 64*k), the
// channel weight is a wave shuffle reduction (sum / hw), per-pixel partials stay in registers
// and the 16 wave partials meet in LDS (summed in wave order, then ReLU).  0.8 MB in per
// image at layer4 of ResNet-50: latency-, not bandwidth-bound.
#include "xai_common.h"

namespace {

constexpr int kWaves = 16;

template <int PPL>   // pixels per lane: h*w <= 64*PPL
__global__ __launch_bounds__(kWaves* kWave) void gradcam_kernel(const float* __restrict__ act, const float* __restrict__ grad, int C,
                                                                int hw, int relu, float* __restrict__ cam) {
  extern __shared__ __attribute__((aligned(16))) float part[];   // [kWaves][hw]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t img = static_cast<int64_t>(blockIdx.x) * C * hw;
  const float n = static_cast<float>(hw);
  float acc[PPL];
#pragma unroll
  for (int k = 0; k < PPL; ++k) acc[k] = 0.f;
  // UC channels per trip: 2*UC*PPL independent loads are in flight before the first shuffle
  // reduction (the kernel is latency-bound: one image is 0.8 MB)
  constexpr int UC = PPL <= 2 ? 8 : (PPL <= 4 ? 4 : 2);
  for (int c0 = wave; c0 < C; c0 += kWaves * UC) {
    float gv[UC][PPL], av[UC][PPL];
#pragma unroll
    for (int u = 0; u < UC; ++u) {
      const int c = c0 + u * kWaves;
      const float* g = grad + img + static_cast<int64_t>(c) * hw;
      const float* a = act + img + static_cast<int64_t>(c) * hw;
#pragma unroll
      for (int k = 0; k < PPL; ++k) {
        const int p = lane + 64 * k;
        const bool ok = c < C && p < hw;
        gv[u][k] = ok ? g[p] : 0.f;
        av[u][k] = ok ? a[p] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < UC; ++u) {                      // channel order per wave stays ascending
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < PPL; ++k) s += gv[u][k];
      const float w = wave_sum(s) / n;
#pragma unroll
      for (int k = 0; k < PPL; ++k) acc[k] += w * av[u][k];
    }
  }
#pragma unroll
  for (int k = 0; k < PPL; ++k) {
    const int p = lane + 64 * k;
    if (p < hw) part[wave * hw + p] = acc[k];
  }
  __syncthreads();
  for (int p = threadIdx.x; p < hw; p += kWaves * kWave) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) v += part[w * hw + p];
    cam[static_cast<int64_t>(blockIdx.x) * hw + p] = relu ? fmaxf(v, 0.f) : v;
  }
}

// one lane per output pixel; source taps follow ATen's area_pixel_compute_source_index
// (align_corners = False): src = max(scale * (dst + 0.5) - 0.5, 0), neighbour clamped.
__global__ __launch_bounds__(256) void bilinear_up_kernel(const float* __restrict__ src, int h, int w, int H, int W, float mult,
                                                          int take_abs, float* __restrict__ dst) {
  const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
  const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (ox >= W || oy >= H) return;
  const float sh = static_cast<float>(h) / static_cast<float>(H);
  const float sw = static_cast<float>(w) / static_cast<float>(W);
  const float fy = fmaxf(sh * (oy + 0.5f) - 0.5f, 0.f);
  const float fx = fmaxf(sw * (ox + 0.5f) - 0.5f, 0.f);
  const int y0 = static_cast<int>(fy), x0 = static_cast<int>(fx);
  const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
  const float ly1 = fy - y0, lx1 = fx - x0;
  const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
  const float* s = src + static_cast<int64_t>(blockIdx.z) * h * w;
  const float top = s[y0 * w + x0] * lx0 + s[y0 * w + x1] * lx1;
  const float bot = s[y1 * w + x0] * lx0 + s[y1 * w + x1] * lx1;
  float v = (top * ly0 + bot * ly1) * mult;
  if (take_abs) v = fabsf(v);
  dst[(static_cast<int64_t>(blockIdx.z) * H + oy) * W + ox] = v;
}

}  // namespace

XAI_EXPORT int xai_gradcam_f32(const float* act, const float* grad, int B, int C, int h, int w, int relu, float* cam,
                               xai_stream_t stream) {
  XAI_REQUIRE_PTR(act); XAI_REQUIRE_PTR(grad); XAI_REQUIRE_PTR(cam);
  XAI_REQUIRE(B > 0 && C > 0 && h > 0 && w > 0, XAI_E_SHAPE);
  const int hw = h * w;
  XAI_REQUIRE(hw <= 1024, XAI_E_UNSUPPORTED);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t lds = static_cast<size_t>(kWaves) * hw * sizeof(float);
  dim3 grid(B), block(kWaves * kWave);
#define XAI_CAM(P) hipLaunchKernelGGL(gradcam_kernel<P>, grid, block, lds, st, act, grad, C, hw, relu, cam)
  if (hw <= 64) XAI_CAM(1);
  else if (hw <= 128) XAI_CAM(2);
  else if (hw <= 256) XAI_CAM(4);
  else if (hw <= 512) XAI_CAM(8);
  else XAI_CAM(16);
#undef XAI_CAM
  return xai_launch_status();
}

XAI_EXPORT int xai_bilinear_up_f32(const float* src, int B, int h, int w, int H, int W, float scale, int take_abs, float* dst,
                                   xai_stream_t stream) {
  XAI_REQUIRE_PTR(src); XAI_REQUIRE_PTR(dst);
  XAI_REQUIRE(B > 0 && h > 0 && w > 0 && H > 0 && W > 0, XAI_E_SHAPE);
  XAI_REQUIRE(B <= 65535, XAI_E_UNSUPPORTED);
  dim3 grid((W + 63) / 64, (H + 3) / 4, B);
  hipLaunchKernelGGL(bilinear_up_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), src, h, w, H, W, scale, take_abs, dst);
  return xai_launch_status();
}
