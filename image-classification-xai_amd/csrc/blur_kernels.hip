// K7: zero-padded separable blur (the insertion substrate) for gfx950.
//
// One workgroup per 16x64 output tile of one channel plane: the (16+2r)x(64+2r) input halo is
// staged in LDS with zero fill, the horizontal pass writes a (16+2r)x64 LDS intermediate, the
// vertical pass writes the tile -- one HBM read and one HBM write per pixel, the 2r-row
// intermediate never leaves the CU.  Taps are applied in ascending order with separate
// multiply and add (file is built with -ffp-contract=off).
#include "xai_common.h"

namespace {

constexpr int TH = 16, TW = 64, kBlock = 256;

__global__ __launch_bounds__(kBlock) void blur_sep_kernel(const float* __restrict__ x, const float* __restrict__ k1d, int klen,
                                                          int H, int W, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int r = klen / 2;
  const int IH = TH + 2 * r, IW = TW + 2 * r;
  float* tin = lds;               // [IH][IW]
  float* tmid = lds + IH * IW;    // [IH][TW]
  const int64_t plane = static_cast<int64_t>(blockIdx.z) * H * W;
  const int y0 = blockIdx.y * TH, x0 = blockIdx.x * TW;
  for (int i = threadIdx.x; i < IH * IW; i += kBlock) {
    const int ly = i / IW, lx = i - ly * IW;
    const int gy = y0 + ly - r, gx = x0 + lx - r;
    tin[i] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? x[plane + static_cast<int64_t>(gy) * W + gx] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < IH * TW; i += kBlock) {
    const int ly = i / TW, lx = i - ly * TW;
    const float* row = tin + ly * IW + lx;
    float acc = 0.f;
    for (int j = 0; j < klen; ++j) acc += k1d[j] * row[j];
    tmid[i] = acc;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < TH * TW; i += kBlock) {
    const int ly = i / TW, lx = i - ly * TW;
    const int gy = y0 + ly, gx = x0 + lx;
    if (gy >= H || gx >= W) continue;
    const float* col = tmid + ly * TW + lx;
    float acc = 0.f;
    for (int j = 0; j < klen; ++j) acc += k1d[j] * col[j * TW];
    out[plane + static_cast<int64_t>(gy) * W + gx] = acc;
  }
}

// One 1-D zero-padded pass along W (axis 1) or H (axis 0) for kernels too long for the fused tile version
// (the growing-kernel search of the reference's MDA branch goes up to 101 taps): lane = output pixel, taps
// through the scalar cache, rows of neighbouring lanes coalesce.
__global__ __launch_bounds__(kBlock) void blur_1d_kernel(const float* __restrict__ x, const float* __restrict__ k1d, int klen, int axis,
                                                         int H, int W, float* __restrict__ out) {
  const int64_t plane = static_cast<int64_t>(blockIdx.y) * H * W;
  const int64_t p = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (p >= static_cast<int64_t>(H) * W) return;
  const int y = static_cast<int>(p / W), xx = static_cast<int>(p - static_cast<int64_t>(y) * W);
  const int r = klen / 2;
  float acc = 0.f;
  if (axis == 1) {
    for (int j = 0; j < klen; ++j) {
      const int gx = xx + j - r;
      acc += k1d[j] * ((gx >= 0 && gx < W) ? x[plane + static_cast<int64_t>(y) * W + gx] : 0.f);
    }
  } else {
    for (int j = 0; j < klen; ++j) {
      const int gy = y + j - r;
      acc += k1d[j] * ((gy >= 0 && gy < H) ? x[plane + static_cast<int64_t>(gy) * W + xx] : 0.f);
    }
  }
  out[plane + p] = acc;
}

}  // namespace

XAI_EXPORT int xai_blur_1d_f32(const float* x, const float* k1d, int klen, int axis, int B, int C, int H, int W, float* out,
                               xai_stream_t stream) {
  XAI_REQUIRE_PTR(x); XAI_REQUIRE_PTR(k1d); XAI_REQUIRE_PTR(out);
  XAI_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && klen > 0 && (klen & 1) && (axis == 0 || axis == 1) && x != out, XAI_E_SHAPE);
  XAI_REQUIRE(static_cast<int64_t>(B) * C <= 65535, XAI_E_UNSUPPORTED);
  dim3 grid(static_cast<unsigned>(xai_ceil_div(static_cast<int64_t>(H) * W, kBlock)), B * C);
  hipLaunchKernelGGL(blur_1d_kernel, grid, dim3(kBlock), 0, static_cast<hipStream_t>(stream), x, k1d, klen, axis, H, W, out);
  return xai_launch_status();
}

XAI_EXPORT int xai_blur_sep_f32(const float* x, const float* k1d, int klen, int B, int C, int H, int W, float* out,
                                xai_stream_t stream) {
  XAI_REQUIRE_PTR(x); XAI_REQUIRE_PTR(k1d); XAI_REQUIRE_PTR(out);
  XAI_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && klen > 0 && (klen & 1), XAI_E_SHAPE);
  XAI_REQUIRE(klen <= 63 && static_cast<int64_t>(B) * C <= 65535, XAI_E_UNSUPPORTED);
  const int r = klen / 2;
  const size_t lds = static_cast<size_t>((TH + 2 * r) * (TW + 2 * r) + (TH + 2 * r) * TW) * sizeof(float);
  dim3 grid((W + TW - 1) / TW, (H + TH - 1) / TH, B * C);
  hipLaunchKernelGGL(blur_sep_kernel, grid, dim3(kBlock), lds, static_cast<hipStream_t>(stream), x, k1d, klen, H, W, out);
  return xai_launch_status();
}
