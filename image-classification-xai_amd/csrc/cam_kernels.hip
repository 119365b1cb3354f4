// K3: Grad-CAM channel-weighted reduce + bilinear up-sample for gfx950.
//
// gradcam: a wave owns channels {wave, wave+16, ...} of its workgroup's channel slice; for a channel its
// lanes hold the h*w gradient and activation values (pixel = lane + 64*k), the channel weight is a wave
// shuffle reduction (sum / hw), per-pixel partials stay in registers and the 16 wave partials meet in LDS
// (summed in wave order).  One image is only 0.8 MB at layer4 of ResNet-50, but a single CU streams at
// ~26 GB/s, so with a caller-provided scratch the channels of one image are split over up to 16 workgroups
// (grid = slices x images) and a second tiny launch sums the slice partials in slice order and applies ReLU.
#include "xai_common.h"

namespace {

constexpr int kWaves = 16;

template <int PPL>   // pixels per lane: h*w <= 64*PPL
__global__ __launch_bounds__(kWaves* kWave) void gradcam_kernel(const float* __restrict__ act, const float* __restrict__ grad, int C,
                                                                int hw, int relu, int c_per_slice, float* __restrict__ cam) {
  extern __shared__ __attribute__((aligned(16))) float part[];   // [kWaves][hw]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t img = static_cast<int64_t>(blockIdx.y) * C * hw;
  const int c_begin = blockIdx.x * c_per_slice, c_end = min(C, c_begin + c_per_slice);
  const float n = static_cast<float>(hw);
  float acc[PPL];
#pragma unroll
  for (int k = 0; k < PPL; ++k) acc[k] = 0.f;
  // UC channels per trip: 2*UC*PPL independent loads are in flight before the first shuffle
  // reduction (the kernel is latency-bound: one image is 0.8 MB)
  constexpr int UC = PPL <= 2 ? 8 : (PPL <= 4 ? 4 : 2);
  for (int c0 = c_begin + wave; c0 < c_end; c0 += kWaves * UC) {
    float gv[UC][PPL], av[UC][PPL];
#pragma unroll
    for (int u = 0; u < UC; ++u) {
      const int c = c0 + u * kWaves;
      const float* g = grad + img + static_cast<int64_t>(c) * hw;
      const float* a = act + img + static_cast<int64_t>(c) * hw;
#pragma unroll
      for (int k = 0; k < PPL; ++k) {
        const int p = lane + 64 * k;
        const bool ok = c < c_end && p < hw;
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
    // one slice: final result; several: raw partial [image][slice][pixel] for gradcam_finish_kernel
    cam[(static_cast<int64_t>(blockIdx.y) * gridDim.x + blockIdx.x) * hw + p] = (relu && gridDim.x == 1) ? fmaxf(v, 0.f) : v;
  }
}

__global__ __launch_bounds__(256) void gradcam_finish_kernel(const float* __restrict__ part, int n_slices, int hw, int relu,
                                                             float* __restrict__ cam) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= hw) return;
  const float* src = part + static_cast<int64_t>(blockIdx.y) * n_slices * hw + p;
  float v = 0.f;
  for (int g = 0; g < n_slices; ++g) v += src[static_cast<int64_t>(g) * hw];
  cam[static_cast<int64_t>(blockIdx.y) * hw + p] = relu ? fmaxf(v, 0.f) : v;
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

namespace {
// channel slices per image: enough workgroups to spread one image over several CUs, >= 64 channels each
inline int cam_slices(int B, int C) {
  if (B >= 64) return 1;
  return static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(16, C / 64)));
}
}  // namespace

XAI_EXPORT size_t xai_gradcam_workspace_bytes(int B, int C, int h, int w) {
  if (B <= 0 || C <= 0 || h <= 0 || w <= 0) return 0;
  const int g = cam_slices(B, C);
  return g == 1 ? 0 : static_cast<size_t>(B) * g * h * w * sizeof(float);
}

XAI_EXPORT int xai_gradcam_f32(const float* act, const float* grad, int B, int C, int h, int w, int relu, float* cam, void* ws,
                               size_t ws_bytes, xai_stream_t stream) {
  XAI_REQUIRE_PTR(act); XAI_REQUIRE_PTR(grad); XAI_REQUIRE_PTR(cam);
  XAI_REQUIRE(B > 0 && C > 0 && h > 0 && w > 0, XAI_E_SHAPE);
  const int hw = h * w;
  XAI_REQUIRE(hw <= 1024 && B <= 65535, XAI_E_UNSUPPORTED);
  hipStream_t st = static_cast<hipStream_t>(stream);
  // without (enough) scratch the whole image goes to one workgroup: same result, one CU's bandwidth
  int slices = cam_slices(B, C);
  if (slices > 1 && (ws == nullptr || ws_bytes < xai_gradcam_workspace_bytes(B, C, h, w))) slices = 1;
  const int per = static_cast<int>(xai_ceil_div(C, slices));
  slices = static_cast<int>(xai_ceil_div(C, per));
  float* dst = slices == 1 ? cam : static_cast<float*>(ws);
  const size_t lds = static_cast<size_t>(kWaves) * hw * sizeof(float);
  dim3 grid(slices, B), block(kWaves * kWave);
#define XAI_CAM(P) hipLaunchKernelGGL(gradcam_kernel<P>, grid, block, lds, st, act, grad, C, hw, relu, per, dst)
  if (hw <= 64) XAI_CAM(1);
  else if (hw <= 128) XAI_CAM(2);
  else if (hw <= 256) XAI_CAM(4);
  else if (hw <= 512) XAI_CAM(8);
  else XAI_CAM(16);
#undef XAI_CAM
  if (slices > 1)
    hipLaunchKernelGGL(gradcam_finish_kernel, dim3((hw + 255) / 256, B), dim3(256), 0, st, static_cast<const float*>(ws), slices, hw,
                       relu, cam);
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
