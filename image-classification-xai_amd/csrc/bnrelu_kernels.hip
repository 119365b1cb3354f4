// Classifier-side fusion (opt-in, xai_engine/prepare.py): eval-mode BatchNorm2d + ReLU (+ residual add) as ONE
// element-wise kernel per direction instead of PyTorch's three to five.
//
//   forward   y  = relu( bn(x) [+ identity] ),   bn(x) = ((x - mean) * invstd) * weight + bias
//   backward  g1 = y > 0 ? gy : 0;   gx = (g1 * weight) * invstd;   g_identity = g1
//
// NCHW, flat indexing: a lane owns 4 consecutive elements when HW % 4 == 0 (they share a channel).  `variant` selects
// among arithmetically equivalent orderings of the BN expression (bit 0: rsqrt instead of 1/sqrt; bits 1-2: grouping;
// bit 3: fused multiply-add for the last step); xai_engine/prepare.py uses the one that reproduces PyTorch-ROCm's own
// eval-mode kernels bit for bit (found by profiles/experiments/exp_bn_variants.py).
#include "xai_common.h"

namespace {

constexpr int kBlock = 256;

// every activation / gradient tensor here is read exactly once by these kernels: non-temporal loads keep them from pushing
// the just-written outputs (which the next convolution reads at once) out of L2 / Infinity Cache
typedef float fx4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4_nt(const float* p) {
  const fx4 v = __builtin_nontemporal_load(reinterpret_cast<const fx4*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ float inv_std(float var, float eps, int variant) {
  return (variant & 1) ? rsqrtf(var + eps) : 1.f / sqrtf(var + eps);
}

__device__ __forceinline__ float bn_value(float x, float mean, float invstd, float w, float b, int variant) {
  const int grouping = (variant >> 1) & 3;
  const bool fma = (variant >> 3) & 1;
  if (grouping == 0) {                       // ((x - mean) * invstd) * w + b
    const float h = (x - mean) * invstd;
    return fma ? __builtin_fmaf(h, w, b) : h * w + b;
  }
  if (grouping == 1) {                       // (x - mean) * (w * invstd) + b
    const float s = w * invstd;
    return fma ? __builtin_fmaf(x - mean, s, b) : (x - mean) * s + b;
  }
  const float s = w * invstd;                // x * s + (b - mean * s)
  const float t = b - mean * s;
  return fma ? __builtin_fmaf(x, s, t) : x * s + t;
}

__device__ __forceinline__ float bn_grad(float g, float invstd, float w, int variant) {
  const int grouping = (variant >> 1) & 3;
  if (grouping == 0) return (g * w) * invstd;
  if (grouping == 1) return g * (w * invstd);
  return (g * invstd) * w;
}

// second BN (bn2.*, nullable as a set): the identity operand is itself a raw convolution output that still needs its own
// eval-mode BatchNorm -- the down-sample branch of a residual block: y = relu(bn(x) + bn2(identity)).
struct BnParams {
  const float* w;
  const float* b;
  const float* mean;
  const float* var;
  float eps;
};

template <bool VEC, bool ADD, bool RELU>
__global__ __launch_bounds__(kBlock) void bn_act_fwd_kernel(const float* __restrict__ x, const float* __restrict__ idt,
                                                            const float* __restrict__ w, const float* __restrict__ b,
                                                            const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                                            BnParams bn2, int variant, int C, int HW, int64_t n, float* __restrict__ y) {
  const int64_t i = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) * (VEC ? 4 : 1);
  if (i >= n) return;
  const int c = static_cast<int>((i / HW) % C);
  const float m = mean[c], is = inv_std(var[c], eps, variant), wc = w[c], bc = b[c];
  const bool second = ADD && bn2.w != nullptr;
  float m2 = 0.f, is2 = 1.f, w2 = 1.f, b2 = 0.f;
  if (second) { m2 = bn2.mean[c]; is2 = inv_std(bn2.var[c], bn2.eps, variant); w2 = bn2.w[c]; b2 = bn2.b[c]; }
  if (VEC) {
    const float4 v = ld4_nt(x + i);
    float4 o = make_float4(bn_value(v.x, m, is, wc, bc, variant), bn_value(v.y, m, is, wc, bc, variant),
                           bn_value(v.z, m, is, wc, bc, variant), bn_value(v.w, m, is, wc, bc, variant));
    if (ADD) {
      float4 a = ld4_nt(idt + i);
      if (second)
        a = make_float4(bn_value(a.x, m2, is2, w2, b2, variant), bn_value(a.y, m2, is2, w2, b2, variant),
                        bn_value(a.z, m2, is2, w2, b2, variant), bn_value(a.w, m2, is2, w2, b2, variant));
      o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
    }
    if (RELU) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    st4(y + i, o);
  } else {
    float o = bn_value(x[i], m, is, wc, bc, variant);
    if (ADD) o += second ? bn_value(idt[i], m2, is2, w2, b2, variant) : idt[i];
    if (RELU) o = fmaxf(o, 0.f);
    y[i] = o;
  }
}

// gy2 (nullable): a second incoming gradient of the same output, summed first -- the residual join (the block output feeds
// the next block's first convolution AND its identity path), which autograd would otherwise add in a kernel of its own.
template <bool VEC, bool ADD>
__global__ __launch_bounds__(kBlock) void bn_relu_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ gy2,
                                                             const float* __restrict__ y, const float* __restrict__ w,
                                                             const float* __restrict__ var, float eps, BnParams bn2, int variant,
                                                             int C, int HW, int64_t n, float* __restrict__ gx,
                                                             float* __restrict__ gid) {
  const int64_t i = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) * (VEC ? 4 : 1);
  if (i >= n) return;
  const int c = static_cast<int>((i / HW) % C);
  const float is = inv_std(var[c], eps, variant), wc = w[c];
  const bool second = ADD && bn2.w != nullptr;          // g_identity then goes through the identity operand's own BatchNorm
  float is2 = 1.f, w2 = 1.f;
  if (second) { is2 = inv_std(bn2.var[c], bn2.eps, variant); w2 = bn2.w[c]; }
  if (VEC) {
    float4 g = ld4_nt(gy + i);
    if (gy2 != nullptr) {
      const float4 h = ld4_nt(gy2 + i);
      g.x += h.x; g.y += h.y; g.z += h.z; g.w += h.w;
    }
    const float4 o = ld4_nt(y + i);
    const float4 g1 = make_float4(o.x > 0.f ? g.x : 0.f, o.y > 0.f ? g.y : 0.f, o.z > 0.f ? g.z : 0.f, o.w > 0.f ? g.w : 0.f);
    st4(gx + i, make_float4(bn_grad(g1.x, is, wc, variant), bn_grad(g1.y, is, wc, variant), bn_grad(g1.z, is, wc, variant),
                            bn_grad(g1.w, is, wc, variant)));
    if (ADD)
      st4(gid + i, second ? make_float4(bn_grad(g1.x, is2, w2, variant), bn_grad(g1.y, is2, w2, variant), bn_grad(g1.z, is2, w2, variant),
                                        bn_grad(g1.w, is2, w2, variant))
                          : g1);
  } else {
    float g = gy[i];
    if (gy2 != nullptr) g += gy2[i];
    const float g1 = y[i] > 0.f ? g : 0.f;
    gx[i] = bn_grad(g1, is, wc, variant);
    if (ADD) gid[i] = second ? bn_grad(g1, is2, w2, variant) : g1;
  }
}

}  // namespace

XAI_EXPORT int xai_bn_act_fwd_f32(const float* x, const float* identity, const float* weight, const float* bias, const float* mean,
                                  const float* var, float eps, const float* weight2, const float* bias2, const float* mean2,
                                  const float* var2, float eps2, int variant, int relu, int N, int C, int HW, float* y,
                                  xai_stream_t stream) {
  XAI_REQUIRE_PTR(x); XAI_REQUIRE_PTR(weight); XAI_REQUIRE_PTR(bias); XAI_REQUIRE_PTR(mean); XAI_REQUIRE_PTR(var); XAI_REQUIRE_PTR(y);
  XAI_REQUIRE(N > 0 && C > 0 && HW > 0 && variant >= 0 && variant < 16, XAI_E_SHAPE);
  if (weight2 != nullptr) {
    XAI_REQUIRE_PTR(identity); XAI_REQUIRE_PTR(bias2); XAI_REQUIRE_PTR(mean2); XAI_REQUIRE_PTR(var2);
  }
  const BnParams bn2{weight2, bias2, mean2, var2, eps2};
  const int64_t n = static_cast<int64_t>(N) * C * HW;
  const bool vec = HW % 4 == 0 && xai_aligned16(x) && xai_aligned16(y) && (identity == nullptr || xai_aligned16(identity));
  const unsigned grid = static_cast<unsigned>(xai_ceil_div(n, static_cast<int64_t>(kBlock) * (vec ? 4 : 1)));
  hipStream_t st = static_cast<hipStream_t>(stream);
#define XAI_BN_FWD(V, A, R) \
  hipLaunchKernelGGL((bn_act_fwd_kernel<V, A, R>), dim3(grid), dim3(kBlock), 0, st, x, identity, weight, bias, mean, var, eps, bn2, variant, C, HW, n, y)
  if (identity != nullptr) {
    XAI_REQUIRE(relu != 0, XAI_E_UNSUPPORTED);
    if (vec) XAI_BN_FWD(true, true, true); else XAI_BN_FWD(false, true, true);
  } else if (relu) {
    if (vec) XAI_BN_FWD(true, false, true); else XAI_BN_FWD(false, false, true);
  } else {
    if (vec) XAI_BN_FWD(true, false, false); else XAI_BN_FWD(false, false, false);
  }
#undef XAI_BN_FWD
  return xai_launch_status();
}

XAI_EXPORT int xai_bn_relu_bwd_f32(const float* gy, const float* gy2, const float* y, const float* weight, const float* var, float eps,
                                   const float* weight2, const float* var2, float eps2, int variant, int N, int C, int HW, float* gx,
                                   float* g_identity, xai_stream_t stream) {
  XAI_REQUIRE_PTR(gy); XAI_REQUIRE_PTR(y); XAI_REQUIRE_PTR(weight); XAI_REQUIRE_PTR(var); XAI_REQUIRE_PTR(gx);
  XAI_REQUIRE(N > 0 && C > 0 && HW > 0 && variant >= 0 && variant < 16, XAI_E_SHAPE);
  const int64_t n = static_cast<int64_t>(N) * C * HW;
  if (weight2 != nullptr) {
    XAI_REQUIRE_PTR(g_identity); XAI_REQUIRE_PTR(var2);
  }
  const BnParams bn2{weight2, nullptr, nullptr, var2, eps2};
  const bool vec = HW % 4 == 0 && xai_aligned16(gy) && xai_aligned16(y) && xai_aligned16(gx) &&
                   (g_identity == nullptr || xai_aligned16(g_identity)) && (gy2 == nullptr || xai_aligned16(gy2));
  const unsigned grid = static_cast<unsigned>(xai_ceil_div(n, static_cast<int64_t>(kBlock) * (vec ? 4 : 1)));
  hipStream_t st = static_cast<hipStream_t>(stream);
#define XAI_BN_BWD(V, A) \
  hipLaunchKernelGGL((bn_relu_bwd_kernel<V, A>), dim3(grid), dim3(kBlock), 0, st, gy, gy2, y, weight, var, eps, bn2, variant, C, HW, n, gx, g_identity)
  if (g_identity != nullptr) {
    if (vec) XAI_BN_BWD(true, true); else XAI_BN_BWD(false, true);
  } else {
    if (vec) XAI_BN_BWD(true, false); else XAI_BN_BWD(false, false);
  }
#undef XAI_BN_BWD
  return xai_launch_status();
}

// ---- MaxPool2d backward (stem) ------------------------------------------------------------------------------------
// gx[plane][h][w] = sum of gy[plane][ph][pw] over the pooling windows (ph, pw ascending) whose arg-max index (from
// PyTorch's own forward, int64 = h * W + w) is this position -- the loop and the accumulation order of PyTorch's
// max_pool_backward_nchw, so the result is bit-identical; one lane per input element, coalesced stores, the window
// reads hit L1 / L2 (PyTorch's kernel takes 611 us for the benchmark's 100 x 64 x 112 x 112 stem activation).
namespace {

// grid = (ceil(H*W / 256), planes), 32-bit index arithmetic.  NW = windows per axis that can cover one input position
// (ceil(kernel / stride)); for NW <= 2 all candidate indices AND gradients are loaded unconditionally up front and then
// selected -- with the load of gy behind the index comparison, every lane walked a chain of up to 8 dependent memory
// latencies and the kernel took as long as PyTorch's (570 us for the benchmark's stem activation).
template <int NW>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ gy, const int64_t* __restrict__ idx, int H, int W,
                                                          int PH, int PW, int k, int stride, int pad, float* __restrict__ gx) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= H * W) return;
  const int h = p / W, w = p - h * W;
  const int ph0 = (h + pad < k) ? 0 : (h + pad - k) / stride + 1;
  const int ph1 = min((h + pad) / stride + 1, PH);
  const int pw0 = (w + pad < k) ? 0 : (w + pad - k) / stride + 1;
  const int pw1 = min((w + pad) / stride + 1, PW);
  const int64_t off = static_cast<int64_t>(blockIdx.y) * PH * PW;
  const int64_t* ip = idx + off;
  const float* gp = gy + off;
  float g = 0.f;
  if (NW > 0) {
    constexpr int M = NW > 0 ? NW : 1;
    int hit[M][M];
    float val[M][M];
#pragma unroll
    for (int a = 0; a < NW; ++a)
#pragma unroll
      for (int b = 0; b < NW; ++b) {
        const int ph = ph0 + a, pw = pw0 + b;
        const bool in = ph < ph1 && pw < pw1;
        const int q = in ? ph * PW + pw : 0;
        hit[a][b] = in ? static_cast<int>(ip[q]) : -1;
        val[a][b] = gp[q];
      }
#pragma unroll
    for (int a = 0; a < NW; ++a)
#pragma unroll
      for (int b = 0; b < NW; ++b)
        if (hit[a][b] == p) g += val[a][b];
  } else {
    for (int ph = ph0; ph < ph1; ++ph)
      for (int pw = pw0; pw < pw1; ++pw) {
        const int q = ph * PW + pw;
        if (static_cast<int>(ip[q]) == p) g += gp[q];
      }
  }
  gx[static_cast<int64_t>(blockIdx.y) * H * W + p] = g;
}

// Tiled form of the same gather: a workgroup owns kBwdRows input rows of one plane; the pooled rows whose windows can
// reach them (indices truncated to int32, and gradients) are staged in LDS with coalesced loads, 8 in flight per lane, and
// every lane then walks its (at most NW x NW) candidate windows in LDS in the same (ph, pw) order.
constexpr int kBwdRows = 16;

template <int NW>
__global__ __launch_bounds__(256) void maxpool_bwd_tiled_kernel(const float* __restrict__ gy, const int64_t* __restrict__ idx, int H, int W,
                                                                int PH, int PW, int k, int stride, int pad, float* __restrict__ gx) {
  extern __shared__ int lds_i[];                         // [rows][PW] indices, then [rows][PW] gradients
  const int h0 = blockIdx.x * kBwdRows;
  const int h_last = min(h0 + kBwdRows, H) - 1;
  const int pr0 = (h0 + pad < k) ? 0 : (h0 + pad - k) / stride + 1;           // first pooled row that can cover row h0
  const int pr1 = min((h_last + pad) / stride + 1, PH);                         // one past the last that can cover h_last
  const int rows = pr1 - pr0;
  float* lds_g = reinterpret_cast<float*>(lds_i + rows * PW);
  const int64_t off = static_cast<int64_t>(blockIdx.y) * PH * PW + static_cast<int64_t>(pr0) * PW;
  const int total = rows * PW;
  for (int base = threadIdx.x; base < total; base += 256 * 8) {
    int iv[8];
    float gv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * 256;
      iv[u] = i < total ? static_cast<int>(idx[off + i]) : -1;
      gv[u] = i < total ? gy[off + i] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * 256;
      if (i < total) { lds_i[i] = iv[u]; lds_g[i] = gv[u]; }
    }
  }
  __syncthreads();
  const int n_rows = h_last - h0 + 1;
  for (int q = threadIdx.x; q < n_rows * W; q += 256) {
    const int r = q / W, w = q - r * W;
    const int h = h0 + r, p = h * W + w;
    const int ph0 = (h + pad < k) ? 0 : (h + pad - k) / stride + 1;
    const int ph1 = min((h + pad) / stride + 1, PH);
    const int pw0 = (w + pad < k) ? 0 : (w + pad - k) / stride + 1;
    const int pw1 = min((w + pad) / stride + 1, PW);
    float g = 0.f;
#pragma unroll
    for (int a = 0; a < NW; ++a)
#pragma unroll
      for (int b = 0; b < NW; ++b) {
        const int ph = ph0 + a, pw = pw0 + b;
        if (ph < ph1 && pw < pw1) {
          const int t = (ph - pr0) * PW + pw;
          if (lds_i[t] == p) g += lds_g[t];
        }
      }
    gx[static_cast<int64_t>(blockIdx.y) * H * W + p] = g;
  }
}

}  // namespace

XAI_EXPORT int xai_maxpool_bwd_f32(const float* gy, const int64_t* indices, int planes, int H, int W, int PH, int PW, int kernel,
                                   int stride, int pad, float* gx, xai_stream_t stream) {
  XAI_REQUIRE_PTR(gy); XAI_REQUIRE_PTR(indices); XAI_REQUIRE_PTR(gx);
  XAI_REQUIRE(planes > 0 && H > 0 && W > 0 && PH > 0 && PW > 0 && kernel > 0 && stride > 0 && pad >= 0, XAI_E_SHAPE);
  XAI_REQUIRE(static_cast<int64_t>(H) * W <= INT32_MAX, XAI_E_UNSUPPORTED);
  const int nw = (kernel + stride - 1) / stride;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int pooled_rows = (kBwdRows + kernel - 2) / stride + 2;                   // upper bound of the pooled rows one tile needs
  const size_t lds = static_cast<size_t>(pooled_rows) * PW * 8;
  // planes ride on grid.y (<= 65535): larger batches (1024 images x 64 stem channels and up) go in slabs of planes
  constexpr int kMaxPlanes = 65535;
  for (int p0 = 0; p0 < planes; p0 += kMaxPlanes) {
    const int np = planes - p0 < kMaxPlanes ? planes - p0 : kMaxPlanes;
    const float* gy_s = gy + static_cast<int64_t>(p0) * PH * PW;
    const int64_t* idx_s = indices + static_cast<int64_t>(p0) * PH * PW;
    float* gx_s = gx + static_cast<int64_t>(p0) * H * W;
    if (nw <= 2 && lds <= 48 * 1024) {
      dim3 tgrid(static_cast<unsigned>(xai_ceil_div(H, kBwdRows)), np);
      if (nw == 1)
        hipLaunchKernelGGL(maxpool_bwd_tiled_kernel<1>, tgrid, dim3(256), lds, st, gy_s, idx_s, H, W, PH, PW, kernel, stride, pad, gx_s);
      else
        hipLaunchKernelGGL(maxpool_bwd_tiled_kernel<2>, tgrid, dim3(256), lds, st, gy_s, idx_s, H, W, PH, PW, kernel, stride, pad, gx_s);
    } else {
      dim3 grid(static_cast<unsigned>(xai_ceil_div(static_cast<int64_t>(H) * W, 256)), np);
      if (nw == 1)
        hipLaunchKernelGGL(maxpool_bwd_kernel<1>, grid, dim3(256), 0, st, gy_s, idx_s, H, W, PH, PW, kernel, stride, pad, gx_s);
      else if (nw == 2)
        hipLaunchKernelGGL(maxpool_bwd_kernel<2>, grid, dim3(256), 0, st, gy_s, idx_s, H, W, PH, PW, kernel, stride, pad, gx_s);
      else
        hipLaunchKernelGGL(maxpool_bwd_kernel<0>, grid, dim3(256), 0, st, gy_s, idx_s, H, W, PH, PW, kernel, stride, pad, gx_s);
    }
    const int rc = xai_launch_status();
    if (rc != XAI_OK) return rc;
  }
  return XAI_OK;
}

// ---- inference-only stem: max_pool( relu( bn(x) ) ) in one pass --------------------------------------------------------
// One lane per pooled output: the (up to k x k) inputs of its window go through the BN expression and ReLU in registers and
// only the maximum is written -- the 4x larger un-pooled activation is never stored.  Forward-only loops (RISE, the
// insertion / deletion sequences) run the classifier without autograd, so nothing downstream needs that activation.
// Values are those of PyTorch's three kernels (a maximum has no rounding); NaN wins like in max_pool_forward_nchw.
namespace {

__global__ __launch_bounds__(256) void bn_relu_maxpool_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ b, const float* __restrict__ mean,
                                                              const float* __restrict__ var, float eps, int variant, int C, int H, int W,
                                                              int PH, int PW, int k, int stride, int pad, float* __restrict__ y) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= PH * PW) return;
  const int plane = blockIdx.y;
  const int c = plane % C;
  const float m = mean[c], is = inv_std(var[c], eps, variant), wc = w[c], bc = b[c];
  const int ph = q / PW, pw = q - ph * PW;
  const int h0 = max(ph * stride - pad, 0), h1 = min(ph * stride - pad + k, H);
  const int w0 = max(pw * stride - pad, 0), w1 = min(pw * stride - pad + k, W);
  const float* src = x + static_cast<int64_t>(plane) * H * W;
  float best = -INFINITY;
  for (int h = h0; h < h1; ++h)
    for (int ww = w0; ww < w1; ++ww) {
      const float v = fmaxf(bn_value(src[h * W + ww], m, is, wc, bc, variant), 0.f);
      if (v > best || v != v) best = v;
    }
  y[static_cast<int64_t>(plane) * PH * PW + q] = best;
}

// Tiled form: a workgroup owns kPoolRows pooled rows of one plane; the input rows they cover are read once, coalesced, put
// through BN + ReLU and parked in LDS (out-of-range positions as -inf), then every lane takes the maximum of its window
// from LDS.  The direct form above reads each input up to (k/stride)^2 times with a stride-2 lane pattern and ran at
// 1.6 TB/s on the benchmark's stem (803 MB for 250 images).
constexpr int kPoolRows = 8;

__global__ __launch_bounds__(256) void bn_relu_maxpool_tiled_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                    const float* __restrict__ b, const float* __restrict__ mean,
                                                                    const float* __restrict__ var, float eps, int variant, int C, int H,
                                                                    int W, int PH, int PW, int k, int stride, int pad,
                                                                    float* __restrict__ y) {
  extern __shared__ float tile[];                       // [rows][W + 2 * pad]
  const int plane = blockIdx.y;
  const int c = plane % C;
  const float m = mean[c], is = inv_std(var[c], eps, variant), wc = w[c], bc = b[c];
  const int ph0 = blockIdx.x * kPoolRows;
  const int n_out = min(kPoolRows, PH - ph0);
  const int rows = (n_out - 1) * stride + k;
  const int h_first = ph0 * stride - pad;               // input row of tile row 0 (may be negative)
  const int TWp = W + 2 * pad;
  const float* src = x + static_cast<int64_t>(plane) * H * W;
  // 8 loads in flight per lane before the first LDS store (a store between two loads serialises them on the HBM latency)
  for (int base = threadIdx.x; base < rows * TWp; base += 256 * 8) {
    float v[8];
    bool in[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * 256;
      const int r = i / TWp, cx = i - r * TWp;
      const int h = h_first + r, ww = cx - pad;
      in[u] = i < rows * TWp && h >= 0 && h < H && ww >= 0 && ww < W;
      v[u] = in[u] ? src[h * W + ww] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * 256;
      if (i < rows * TWp) tile[i] = in[u] ? fmaxf(bn_value(v[u], m, is, wc, bc, variant), 0.f) : -INFINITY;
    }
  }
  __syncthreads();
  for (int q = threadIdx.x; q < n_out * PW; q += 256) {
    const int pr = q / PW, pw = q - pr * PW;
    const float* t = tile + (pr * stride) * TWp + pw * stride;
    float best = -INFINITY;
    for (int a = 0; a < k; ++a)
      for (int bb = 0; bb < k; ++bb) {
        const float v = t[a * TWp + bb];
        if (v > best || v != v) best = v;
      }
    y[(static_cast<int64_t>(plane) * PH + ph0 + pr) * PW + pw] = best;
  }
}

}  // namespace

XAI_EXPORT int xai_bn_relu_maxpool_fwd_f32(const float* x, const float* weight, const float* bias, const float* mean, const float* var,
                                           float eps, int variant, int N, int C, int H, int W, int PH, int PW, int kernel, int stride,
                                           int pad, float* y, xai_stream_t stream) {
  XAI_REQUIRE_PTR(x); XAI_REQUIRE_PTR(weight); XAI_REQUIRE_PTR(bias); XAI_REQUIRE_PTR(mean); XAI_REQUIRE_PTR(var); XAI_REQUIRE_PTR(y);
  XAI_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && PH > 0 && PW > 0 && kernel > 0 && stride > 0 && pad >= 0 && variant >= 0 && variant < 16,
              XAI_E_SHAPE);
  XAI_REQUIRE(static_cast<int64_t>(N) * C <= 65535 && static_cast<int64_t>(H) * W <= INT32_MAX, XAI_E_UNSUPPORTED);
  const size_t lds = static_cast<size_t>((kPoolRows - 1) * stride + kernel) * (W + 2 * pad) * sizeof(float);
  if (lds <= 48 * 1024) {
    dim3 grid(static_cast<unsigned>(xai_ceil_div(PH, kPoolRows)), N * C);
    hipLaunchKernelGGL(bn_relu_maxpool_tiled_kernel, grid, dim3(256), lds, static_cast<hipStream_t>(stream), x, weight, bias, mean, var, eps,
                       variant, C, H, W, PH, PW, kernel, stride, pad, y);
  } else {
    dim3 grid(static_cast<unsigned>(xai_ceil_div(static_cast<int64_t>(PH) * PW, 256)), N * C);
    hipLaunchKernelGGL(bn_relu_maxpool_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), x, weight, bias, mean, var, eps, variant,
                       C, H, W, PH, PW, kernel, stride, pad, y);
  }
  return xai_launch_status();
}
