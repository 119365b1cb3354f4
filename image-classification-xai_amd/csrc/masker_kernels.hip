// K11-K14: feature-map maskers of the RISE family (ViT-CX) for gfx950.
//
//   K11 up_rownorm     feature maps (R,h,w) -> bilinear (H,W) -> per-row min-max normalised masks, one launch:
//                      the map is stretched in LDS and its up-sampled values are computed twice (once for the
//                      row's min / max, once to write), so HBM sees R*H*W*4 bytes of writes and nothing else.
//   K12 rownorm        per-row min-max normalisation of stored rows (the cluster sums).
//   K13 cluster_sum    out[k] = sum of the member rows of cluster k, members in ascending row order (the order of
//                      the reference's `mask_clustering[label[i]] += mask[i]` loop, so the sums round identically).
//   K14 causal_apply   masked = x*m + (noise*scale)*(1-m);  plain = x + (noise*scale)*(1-m)  for N masks, written
//                      as the [2N][C][HW] stack the classifier consumes.
//   K16 masked_sums    weighted[p] = (sum_n w_n m_n[p]) / N and plain[p] = (sum_n m_n[p]) / N from ONE read of the mask
//                      stack (n ascending, fp32, product rounded before the add: bit-identical to two K2 launches).
// All element-wise, HBM-bound; built with -ffp-contract=off so every a*b+c rounds like the torch expression.
#include "xai_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxSrc = 4096;       // h*w floats of one feature map held in LDS
constexpr int kMaxStretch = 8192;   // h*W floats of its horizontally stretched copy (dynamic LDS, 32 KB)
constexpr int kMaxTaps = 1024;      // H vertical taps of 16 B behind it (16 KB): 64 KB per workgroup with `s`

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, kWave));
  return v;
}

// min / max over the workgroup; every lane gets the result.  `red` holds 2 * waves floats.
__device__ __forceinline__ void block_min_max(float& lo, float& hi, float* red) {
  lo = wave_min(lo);
  hi = wave_max(hi);
  const int wave = threadIdx.x >> 6, n_waves = kBlock >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    red[wave] = lo;
    red[n_waves + wave] = hi;
  }
  __syncthreads();
  lo = red[0];
  hi = red[n_waves];
  for (int i = 1; i < n_waves; ++i) {
    lo = fminf(lo, red[i]);
    hi = fmaxf(hi, red[n_waves + i]);
  }
}

// NaN-propagating variants are not needed: the reference's torch.min / torch.max propagate NaN, fminf / fmaxf drop it;
// feature maps and cluster sums are finite, and a NaN row would stay NaN in the reference only.

struct Tap {
  int i0, i1;
  float l0, l1;
};
__device__ __forceinline__ Tap make_tap(int o, int n_in, float ratio) {
  const float f = fmaxf(ratio * (o + 0.5f) - 0.5f, 0.f);
  Tap t;
  t.i0 = static_cast<int>(f);
  t.i1 = t.i0 + (t.i0 < n_in - 1 ? 1 : 0);
  t.l1 = f - t.i0;
  t.l0 = 1.f - t.l1;
  return t;
}

// Correctly rounded num / span from y = RN(1 / span) (Markstein): q = RN(num*y), r = num - q*span exactly (fma),
// RN(q + r*y) is the IEEE quotient -- 3 instructions per element instead of a full division sequence.
__device__ __forceinline__ float div_by(float num, float span, float y) {
  const float q = num * y;
  const float r = __builtin_fmaf(-q, span, num);
  return __builtin_fmaf(r, y, q);
}

// One workgroup per feature map.  The h x w source is first stretched horizontally into an h x W LDS tile
// (tmp[y][ox] = s[y][x0]*lx0 + s[y][x1]*lx1 -- exactly the `top` / `bot` terms of the bilinear formula), after which
// an output pixel is two LDS reads and v = tmp[y0][ox]*ly0 + tmp[y1][ox]*ly1; the H vertical taps sit in an LDS table.
// The map is walked twice: once for its min / max, once to write (v - lo) / (hi - lo).  VEC: a lane owns 4 consecutive
// pixels (W % 4 == 0), b128 LDS reads and stores, (row, column) tracked incrementally.  Plain stores on purpose: the
// masks are read again at once (cosine-similarity GEMM, cluster sums) and 154 MB fits the Infinity Cache.
// Mapping study: csrc/tune/tune_uprow.hip.
template <bool VEC>
__global__ __launch_bounds__(kBlock) void up_rownorm_kernel(const float* __restrict__ src, int h, int w, int H, int W,
                                                            float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float tmp[];   // [h][W], then H taps of 4 words
  __shared__ float s[kMaxSrc];
  __shared__ float red[2 * (kBlock / 64)];
  float4* taps = reinterpret_cast<float4*>(tmp + ((h * W + 3) & ~3));
  const float* row = src + static_cast<int64_t>(blockIdx.x) * h * w;
  for (int i = threadIdx.x; i < h * w; i += kBlock) s[i] = row[i];
  const float rh = static_cast<float>(h) / static_cast<float>(H);
  const float rw = static_cast<float>(w) / static_cast<float>(W);
  for (int oy = threadIdx.x; oy < H; oy += kBlock) {
    const Tap t = make_tap(oy, h, rh);
    taps[oy] = make_float4(__int_as_float(t.i0 * W), __int_as_float(t.i1 * W), t.l0, t.l1);
  }
  __syncthreads();
  for (int ox = threadIdx.x; ox < W; ox += kBlock) {
    const Tap tx = make_tap(ox, w, rw);
    for (int y = 0; y < h; ++y) tmp[y * W + ox] = s[y * w + tx.i0] * tx.l0 + s[y * w + tx.i1] * tx.l1;
  }
  __syncthreads();
  float lo = INFINITY, hi = -INFINITY;
  float* dst = out + static_cast<int64_t>(blockIdx.x) * H * W;
  if (VEC) {
    const int W4 = W / 4, Q = H * W4;
    const int d_row = kBlock / W4, d_col = kBlock % W4;
    int oy = threadIdx.x / W4, c = threadIdx.x % W4;
#pragma unroll 2
    for (int q = threadIdx.x; q < Q; q += kBlock) {
      const float4 t = taps[oy];
      const float4 a = ld4(tmp + __float_as_int(t.x) + 4 * c), b = ld4(tmp + __float_as_int(t.y) + 4 * c);
      const float vx = a.x * t.z + b.x * t.w, vy = a.y * t.z + b.y * t.w, vz = a.z * t.z + b.z * t.w, vw = a.w * t.z + b.w * t.w;
      lo = fminf(fminf(lo, vx), fminf(vy, fminf(vz, vw)));
      hi = fmaxf(fmaxf(hi, vx), fmaxf(vy, fmaxf(vz, vw)));
      oy += d_row; c += d_col;
      if (c >= W4) { c -= W4; ++oy; }
    }
    block_min_max(lo, hi, red);
    const float span = hi - lo, y = 1.f / span;
    oy = threadIdx.x / W4; c = threadIdx.x % W4;
#pragma unroll 2
    for (int q = threadIdx.x; q < Q; q += kBlock) {
      const float4 t = taps[oy];
      const float4 a = ld4(tmp + __float_as_int(t.x) + 4 * c), b = ld4(tmp + __float_as_int(t.y) + 4 * c);
      float4 v;
      v.x = div_by((a.x * t.z + b.x * t.w) - lo, span, y);
      v.y = div_by((a.y * t.z + b.y * t.w) - lo, span, y);
      v.z = div_by((a.z * t.z + b.z * t.w) - lo, span, y);
      v.w = div_by((a.w * t.z + b.w * t.w) - lo, span, y);
      st4(dst + 4 * static_cast<int64_t>(q), v);
      oy += d_row; c += d_col;
      if (c >= W4) { c -= W4; ++oy; }
    }
  } else {
    for (int oy = 0; oy < H; ++oy) {
      const float4 t = taps[oy];
      const float* t0 = tmp + __float_as_int(t.x);
      const float* t1 = tmp + __float_as_int(t.y);
      for (int ox = threadIdx.x; ox < W; ox += kBlock) {
        const float v = t0[ox] * t.z + t1[ox] * t.w;
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
      }
    }
    block_min_max(lo, hi, red);
    const float span = hi - lo, y = 1.f / span;
    for (int oy = 0; oy < H; ++oy) {
      const float4 t = taps[oy];
      const float* t0 = tmp + __float_as_int(t.x);
      const float* t1 = tmp + __float_as_int(t.y);
      for (int ox = threadIdx.x; ox < W; ox += kBlock) dst[oy * W + ox] = div_by((t0[ox] * t.z + t1[ox] * t.w) - lo, span, y);
    }
  }
}

// Every workgroup scans its whole row for min / max and normalises one slice of it, so a handful of rows still fills
// the chip.  The `slices` workgroups of a row re-read the same bytes, so they are placed on ONE XCD (workgroups go to
// XCDs round-robin by linear id, id % 8): the row is fetched into that XCD's L2 once and the other slices hit it.
// 1-D grid of 8 * slices * ceil(R / 8) workgroups; id -> (row = (id / (8*slices)) * 8 + id % 8, slice = (id / 8) % slices).
constexpr int kXcds = 8;
template <bool VEC>
__global__ __launch_bounds__(kBlock) void rownorm_kernel(const float* __restrict__ x, int R, int64_t P, int slices,
                                                         float* __restrict__ out) {
  __shared__ float red[2 * (kBlock / 64)];
  const int id = blockIdx.x;
  const int r = (id / (kXcds * slices)) * kXcds + id % kXcds;
  const int slice = (id / kXcds) % slices;
  if (r >= R) return;
  const float* row = x + static_cast<int64_t>(r) * P;
  float lo = INFINITY, hi = -INFINITY;
  if (VEC) {
    const int64_t P4 = P / 4;
#pragma unroll 4
    for (int64_t q = threadIdx.x; q < P4; q += kBlock) {
      const float4 v = ld4(row + 4 * q);
      lo = fminf(fminf(lo, v.x), fminf(v.y, fminf(v.z, v.w)));
      hi = fmaxf(fmaxf(hi, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
    }
  } else {
#pragma unroll 4
    for (int64_t p = threadIdx.x; p < P; p += kBlock) {
      const float v = row[p];
      lo = fminf(lo, v);
      hi = fmaxf(hi, v);
    }
  }
  block_min_max(lo, hi, red);
  const float span = hi - lo;
  float* dst = out + static_cast<int64_t>(r) * P;
  // `out` may alias `x`: all reads of the scan above are complete for THIS workgroup, but a neighbour slice may still be
  // scanning -- so in-place calls are launched with one slice per row (see the host function).
  if (VEC) {
    const int64_t P4 = P / 4;
    const int64_t per = (P4 + slices - 1) / slices;
    const int64_t q0 = slice * per, q1 = q0 + per < P4 ? q0 + per : P4;
    for (int64_t q = q0 + threadIdx.x; q < q1; q += kBlock) {
      float4 v = ld4(row + 4 * q);
      v.x = (v.x - lo) / span; v.y = (v.y - lo) / span; v.z = (v.z - lo) / span; v.w = (v.w - lo) / span;
      st4(dst + 4 * q, v);
    }
  } else {
    const int64_t per = (P + slices - 1) / slices;
    const int64_t p0 = slice * per, p1 = p0 + per < P ? p0 + per : P;
    for (int64_t p = p0 + threadIdx.x; p < p1; p += kBlock) dst[p] = (row[p] - lo) / span;
  }
}

// grid = (pixel tiles, clusters); a lane owns 4 consecutive pixels when the row length allows float4.
template <bool VEC>
__global__ __launch_bounds__(kBlock) void cluster_sum_kernel(const float* __restrict__ rows, const int32_t* __restrict__ members,
                                                             const int32_t* __restrict__ offs, int64_t P, float* __restrict__ out) {
  const int k = blockIdx.y;
  const int m0 = offs[k], m1 = offs[k + 1];
  if (VEC) {
    const int64_t p = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) * 4;
    if (p >= P) return;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int m = m0; m < m1; ++m) {
      const float4 v = ld4(rows + static_cast<int64_t>(members[m]) * P + p);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    st4(out + static_cast<int64_t>(k) * P + p, acc);
  } else {
    const int64_t p = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (p >= P) return;
    float acc = 0.f;
    for (int m = m0; m < m1; ++m) acc += rows[static_cast<int64_t>(members[m]) * P + p];
    out[static_cast<int64_t>(k) * P + p] = acc;
  }
}

// grid = (pixel tiles, masks); the lane's mask value and 1-m are shared by the C channels.
__global__ __launch_bounds__(kBlock) void causal_apply_kernel(const float* __restrict__ x, const float* __restrict__ masks,
                                                              const float* __restrict__ noise, int N, int C, int64_t HW,
                                                              float noise_scale, float* __restrict__ stack) {
  const int n = blockIdx.y;
  const int64_t p = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (p >= HW) return;
  const float m = masks[static_cast<int64_t>(n) * HW + p];
  const float inv = 1.f - m;
  for (int c = 0; c < C; ++c) {
    const int64_t e = (static_cast<int64_t>(n) * C + c) * HW + p;
    const float add = (__builtin_nontemporal_load(noise + e) * noise_scale) * inv;
    const float xv = x[static_cast<int64_t>(c) * HW + p];
    stack[e] = xv * m + add;
    stack[static_cast<int64_t>(N) * C * HW + e] = xv + add;
  }
}

// the same, 4 pixels per lane (HW % 4 == 0, 16-byte aligned planes): 16-byte loads and stores, identical per-element arithmetic
typedef float masker_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(kBlock) void causal_apply_kernel_v4(const float* __restrict__ x, const float* __restrict__ masks,
                                                                 const float* __restrict__ noise, int N, int C, int64_t HW,
                                                                 float noise_scale, float* __restrict__ stack) {
  const int n = blockIdx.y;
  const int64_t p = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) * 4;
  if (p >= HW) return;
  const float4 m = ld4(masks + static_cast<int64_t>(n) * HW + p);
  const float4 inv = make_float4(1.f - m.x, 1.f - m.y, 1.f - m.z, 1.f - m.w);
  for (int c = 0; c < C; ++c) {
    const int64_t e = (static_cast<int64_t>(n) * C + c) * HW + p;
    const masker_f4 nz = __builtin_nontemporal_load(reinterpret_cast<const masker_f4*>(noise + e));
    const float4 add = make_float4((nz.x * noise_scale) * inv.x, (nz.y * noise_scale) * inv.y, (nz.z * noise_scale) * inv.z, (nz.w * noise_scale) * inv.w);
    const float4 xv = ld4(x + static_cast<int64_t>(c) * HW + p);
    st4(stack + e, make_float4(xv.x * m.x + add.x, xv.y * m.y + add.y, xv.z * m.z + add.z, xv.w * m.w + add.w));
    st4(stack + static_cast<int64_t>(N) * C * HW + e, make_float4(xv.x + add.x, xv.y + add.y, xv.z + add.z, xv.w + add.w));
  }
}

// K16: lane = W consecutive positions p; walks the N rows in order with 8 independent loads in flight.
template <int W>
__global__ __launch_bounds__(kBlock) void masked_sums_kernel(const float* __restrict__ rows, const float* __restrict__ weights, int N,
                                                             int64_t P, float* __restrict__ out_weighted, float* __restrict__ out_plain) {
  constexpr int U = 8;
  const int64_t p = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) * W;
  if (p >= P) return;
  float aw[W], ap[W];
#pragma unroll
  for (int k = 0; k < W; ++k) aw[k] = ap[k] = 0.f;
  const float* r = rows + p;
  int n = 0;
  for (; n + U <= N; n += U) {
    float v[U][W];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if constexpr (W == 4) {
        const float4 t = ld4(r + static_cast<int64_t>(n + u) * P);
        v[u][0] = t.x; v[u][1] = t.y; v[u][2] = t.z; v[u][3] = t.w;
      } else {
        v[u][0] = r[static_cast<int64_t>(n + u) * P];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float w = weights[n + u];
#pragma unroll
      for (int k = 0; k < W; ++k) {
        aw[k] += v[u][k] * w;
        ap[k] += v[u][k];
      }
    }
  }
  for (; n < N; ++n) {
    const float w = weights[n];
#pragma unroll
    for (int k = 0; k < W; ++k) {
      const float t = r[static_cast<int64_t>(n) * P + k];
      aw[k] += t * w;
      ap[k] += t;
    }
  }
  const float denom = static_cast<float>(N);
#pragma unroll
  for (int k = 0; k < W; ++k) {
    out_weighted[p + k] = aw[k] / denom;
    out_plain[p + k] = ap[k] / denom;
  }
}

}  // namespace

XAI_EXPORT int xai_masked_sums_f32(const float* rows, const float* weights, int N, int64_t P, float* out_weighted, float* out_plain,
                                   xai_stream_t stream) {
  XAI_REQUIRE_PTR(rows); XAI_REQUIRE_PTR(weights); XAI_REQUIRE_PTR(out_weighted); XAI_REQUIRE_PTR(out_plain);
  XAI_REQUIRE(N > 0 && P > 0, XAI_E_SHAPE);
  if (P % 4 == 0 && xai_aligned16(rows)) {
    hipLaunchKernelGGL(masked_sums_kernel<4>, dim3(static_cast<unsigned>(xai_ceil_div(P, kBlock * 4))), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), rows, weights, N, P, out_weighted, out_plain);
  } else {
    hipLaunchKernelGGL(masked_sums_kernel<1>, dim3(static_cast<unsigned>(xai_ceil_div(P, kBlock))), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), rows, weights, N, P, out_weighted, out_plain);
  }
  return xai_launch_status();
}

XAI_EXPORT int xai_up_rownorm_f32(const float* src, int R, int h, int w, int H, int W, float* out, xai_stream_t stream) {
  XAI_REQUIRE_PTR(src); XAI_REQUIRE_PTR(out);
  XAI_REQUIRE(R > 0 && h > 0 && w > 0 && H > 0 && W > 0, XAI_E_SHAPE);
  XAI_REQUIRE(static_cast<int64_t>(h) * w <= kMaxSrc && static_cast<int64_t>(H) * W <= INT32_MAX &&
              static_cast<int64_t>(h) * W <= kMaxStretch, XAI_E_UNSUPPORTED);
  XAI_REQUIRE(H <= kMaxTaps, XAI_E_UNSUPPORTED);
  const size_t lds = (static_cast<size_t>((h * W + 3) & ~3) + 4 * static_cast<size_t>(H)) * sizeof(float);
  if (W % 4 == 0 && xai_aligned16(out))
    hipLaunchKernelGGL(up_rownorm_kernel<true>, dim3(R), dim3(kBlock), lds, static_cast<hipStream_t>(stream), src, h, w, H, W, out);
  else
    hipLaunchKernelGGL(up_rownorm_kernel<false>, dim3(R), dim3(kBlock), lds, static_cast<hipStream_t>(stream), src, h, w, H, W, out);
  return xai_launch_status();
}

XAI_EXPORT int xai_rownorm_f32(const float* x, int R, int64_t P, float* out, xai_stream_t stream) {
  XAI_REQUIRE_PTR(x); XAI_REQUIRE_PTR(out);
  XAI_REQUIRE(R > 0 && P > 0, XAI_E_SHAPE);
  XAI_REQUIRE(R <= 65535, XAI_E_UNSUPPORTED);
  int slices = 1;
  if (x != out) {                                       // in place: a slice must not be rewritten while a neighbour still scans
    const int64_t want = xai_ceil_div(2 * xai_cu_count(), R);
    const int64_t most = xai_ceil_div(P, 4 * kBlock);   // at least 4 elements per lane and slice
    slices = static_cast<int>(want < most ? want : most);
    if (slices < 1) slices = 1;
    if (slices > 32) slices = 32;
  }
  const unsigned grid = static_cast<unsigned>(kXcds * slices * xai_ceil_div(R, kXcds));
  if (P % 4 == 0 && xai_aligned16(x) && xai_aligned16(out))
    hipLaunchKernelGGL(rownorm_kernel<true>, dim3(grid), dim3(kBlock), 0, static_cast<hipStream_t>(stream), x, R, P, slices, out);
  else
    hipLaunchKernelGGL(rownorm_kernel<false>, dim3(grid), dim3(kBlock), 0, static_cast<hipStream_t>(stream), x, R, P, slices, out);
  return xai_launch_status();
}

XAI_EXPORT int xai_cluster_sum_f32(const float* rows, const int32_t* members, const int32_t* offs, int K, int64_t P, float* out,
                                   xai_stream_t stream) {
  XAI_REQUIRE_PTR(rows); XAI_REQUIRE_PTR(members); XAI_REQUIRE_PTR(offs); XAI_REQUIRE_PTR(out);
  XAI_REQUIRE(K > 0 && P > 0, XAI_E_SHAPE);
  XAI_REQUIRE(K <= 65535, XAI_E_UNSUPPORTED);
  const bool vec = (P % 4 == 0) && xai_aligned16(rows) && xai_aligned16(out);
  if (vec) {
    dim3 grid(static_cast<unsigned>(xai_ceil_div(P / 4, kBlock)), K);
    hipLaunchKernelGGL(cluster_sum_kernel<true>, grid, dim3(kBlock), 0, static_cast<hipStream_t>(stream), rows, members, offs, P, out);
  } else {
    dim3 grid(static_cast<unsigned>(xai_ceil_div(P, kBlock)), K);
    hipLaunchKernelGGL(cluster_sum_kernel<false>, grid, dim3(kBlock), 0, static_cast<hipStream_t>(stream), rows, members, offs, P, out);
  }
  return xai_launch_status();
}

XAI_EXPORT int xai_causal_apply_f32(const float* x, const float* masks, const float* noise, int N, int C, int64_t HW, float noise_scale,
                                    float* stack, xai_stream_t stream) {
  XAI_REQUIRE_PTR(x); XAI_REQUIRE_PTR(masks); XAI_REQUIRE_PTR(noise); XAI_REQUIRE_PTR(stack);
  XAI_REQUIRE(N > 0 && C > 0 && HW > 0, XAI_E_SHAPE);
  XAI_REQUIRE(N <= 65535, XAI_E_UNSUPPORTED);
  if (HW % 4 == 0 && xai_aligned16(x) && xai_aligned16(masks) && xai_aligned16(noise) && xai_aligned16(stack)) {
    dim3 grid(static_cast<unsigned>(xai_ceil_div(HW, kBlock * 4)), N);
    hipLaunchKernelGGL(causal_apply_kernel_v4, grid, dim3(kBlock), 0, static_cast<hipStream_t>(stream), x, masks, noise, N, C, HW,
                       noise_scale, stack);
  } else {
    dim3 grid(static_cast<unsigned>(xai_ceil_div(HW, kBlock)), N);
    hipLaunchKernelGGL(causal_apply_kernel, grid, dim3(kBlock), 0, static_cast<hipStream_t>(stream), x, masks, noise, N, C, HW,
                       noise_scale, stack);
  }
  return xai_launch_status();
}
