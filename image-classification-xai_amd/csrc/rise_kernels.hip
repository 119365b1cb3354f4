// K4/K5: RISE random-mask application and score-weighted accumulation for gfx950.
//
// A mask is never read from memory: it is the order-1 ("bilinear", mirror boundary,
// half-pixel grid) up-sampling of an s x s binary grid to (s+1)*cell, cropped at a per-mask
// shift -- scipy.ndimage.zoom(grid, up/s, order=1, mode='mirror', grid_mode=True), which is
// what skimage.transform.resize(order=1, mode='reflect') runs.  Tap positions/weights are
// computed in fp64 (they are exact-to-fp32 then), the 4-tap blend in fp32.
//
// K4 (write-bound, C*H*W*4 B per mask): grid = (pixel tiles, masks); a lane produces 4 pixels x C channels with
//    16-byte stores.  s == 8 (the RISE default): the grid is one 64-bit word per mask in SGPRs, no LDS; other s: the
//    mask's s*s grid bytes sit in LDS.  The kernel is bound by the HBM write stream, not by the mask arithmetic: with
//    the arithmetic removed it is no faster, and a separable LDS-staged form (column-interpolated rows + one vertical
//    lerp per pixel) is 4-7 % slower (csrc/tune/tune_rise.hip, profiles/r02_tune_rise.txt).
// K5 (compute/LDS-bound, almost no HBM traffic): grid = (pixel tiles, mask slices); tap tables
//    for every up-sampled row/column live in LDS, mask grids are staged in LDS 256 at a time,
//    a lane owns one pixel and accumulates score*mask in fp64; one fp64 atomic per pixel per
//    slice merges the slices.
#include "xai_common.h"

namespace {

constexpr int kBlock = 256;

struct Tap { int i0, i1; float t; };

// value(j) = (1-t)*g[i0] + t*g[i1] for output index j of an n_in -> n_out up-sampling
// `ratio` = double(n_in) / double(n_out), computed once on the host (an fp64 divide per tap made K4 compute-bound)
__device__ __forceinline__ Tap make_tap(int j, int n_in, double ratio) {
  double c = (j + 0.5) * ratio - 0.5;
  if (c < 0) c = -c;                                   // mirror about sample 0
  const int i0 = static_cast<int>(floor(c));
  int i1 = i0 + 1;
  if (i1 >= n_in) i1 = 2 * n_in - 2 - i1;              // mirror about sample n_in-1
  if (i1 < 0) i1 = 0;                                  // n_in == 1
  return Tap{i0, i1, static_cast<float>(c - i0)};
}

// lo/hi = min/max of the mask's grid: skimage.transform.resize clips its output to the input range
// (clip=True), which also absorbs the last-bit overshoot of the fp32 blend
__device__ __forceinline__ float blend(const uint8_t* g, int s, const Tap& r, const Tap& c, float lo, float hi) {
  const float wr0 = 1.f - r.t, wr1 = r.t, wc0 = 1.f - c.t, wc1 = c.t;
  float v = g[r.i0 * s + c.i0] ? wr0 * wc0 : 0.f;
  v += g[r.i0 * s + c.i1] ? wr0 * wc1 : 0.f;
  v += g[r.i1 * s + c.i0] ? wr1 * wc0 : 0.f;
  v += g[r.i1 * s + c.i1] ? wr1 * wc1 : 0.f;
  return fminf(fmaxf(v, lo), hi);
}

__global__ __launch_bounds__(kBlock) void rise_apply_kernel(const uint8_t* __restrict__ grid, const int32_t* __restrict__ shift, int s,
                                                            int cell_h, int cell_w, double rh, double rw, const float* __restrict__ image, int C, int H,
                                                            int W, float* __restrict__ masked, float* __restrict__ masks) {
  extern __shared__ uint8_t g[];                        // [s][s]
  const int n = blockIdx.y;
  int one = 0, zero = 0;
  for (int i = threadIdx.x; i < s * s; i += kBlock) {
    const uint8_t b = grid[static_cast<int64_t>(n) * s * s + i];
    g[i] = b;
    one |= (b != 0); zero |= (b == 0);
  }
  const float hi = __syncthreads_or(one) ? 1.f : 0.f;
  const float lo = __syncthreads_or(zero) ? 0.f : 1.f;
  __syncthreads();
  const int64_t hw = static_cast<int64_t>(H) * W;
  const int64_t p = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (p >= hw) return;
  const int y = static_cast<int>(static_cast<uint32_t>(p) / static_cast<uint32_t>(W)), x = static_cast<int>(p - static_cast<int64_t>(y) * W);
  const Tap tr = make_tap(y + shift[2 * n], s, rh);
  const Tap tc = make_tap(x + shift[2 * n + 1], s, rw);
  const float m = blend(g, s, tr, tc, lo, hi);
  if (masks) masks[static_cast<int64_t>(n) * hw + p] = m;
  if (masked) {
    float* o = masked + static_cast<int64_t>(n) * C * hw + p;
    for (int c = 0; c < C; ++c) o[c * hw] = image[c * hw + p] * m;
  }
}

// 4 pixels per lane along x (W % 4 == 0, 16-byte aligned planes)
__global__ __launch_bounds__(kBlock) void rise_apply_kernel_v4(const uint8_t* __restrict__ grid, const int32_t* __restrict__ shift,
                                                               int s, int cell_h, int cell_w, double rh, double rw, const float* __restrict__ image,
                                                               int C, int H, int W, float* __restrict__ masked, float* __restrict__ masks) {
  extern __shared__ uint8_t g[];
  const int n = blockIdx.y;
  int one = 0, zero = 0;
  for (int i = threadIdx.x; i < s * s; i += kBlock) {
    const uint8_t b = grid[static_cast<int64_t>(n) * s * s + i];
    g[i] = b;
    one |= (b != 0); zero |= (b == 0);
  }
  const float hi = __syncthreads_or(one) ? 1.f : 0.f;
  const float lo = __syncthreads_or(zero) ? 0.f : 1.f;
  __syncthreads();
  const int64_t hw = static_cast<int64_t>(H) * W;
  const int64_t p = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) * 4;
  if (p >= hw) return;
  const int y = static_cast<int>(static_cast<uint32_t>(p) / static_cast<uint32_t>(W)), x = static_cast<int>(p - static_cast<int64_t>(y) * W);
  const int sx = shift[2 * n + 1];
  const Tap tr = make_tap(y + shift[2 * n], s, rh);
  float4 m;
  m.x = blend(g, s, tr, make_tap(x + sx, s, rw), lo, hi);
  m.y = blend(g, s, tr, make_tap(x + 1 + sx, s, rw), lo, hi);
  m.z = blend(g, s, tr, make_tap(x + 2 + sx, s, rw), lo, hi);
  m.w = blend(g, s, tr, make_tap(x + 3 + sx, s, rw), lo, hi);
  if (masks) st4(masks + static_cast<int64_t>(n) * hw + p, m);
  if (masked) {
    float* o = masked + static_cast<int64_t>(n) * C * hw + p;
    for (int c = 0; c < C; ++c) {
      const float4 v = ld4(image + c * hw + p);
      st4(o + c * hw, make_float4(v.x * m.x, v.y * m.y, v.z * m.z, v.w * m.w));
    }
  }
}

// s == 8 fast path (the RISE default): the whole 8x8 grid is one 64-bit word, fetched with wave-uniform
// (scalar) loads and packed with SALU -- no LDS staging, no barrier, so the short-lived workgroups (12 KB
// written each) start storing immediately.
__device__ __forceinline__ unsigned long long pack_grid8(const uint8_t* g) {
  const unsigned long long* w = reinterpret_cast<const unsigned long long*>(g);     // 64-byte grids are 8-byte aligned
  unsigned long long bits = 0;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const unsigned long long row = w[r] & 0x0101010101010101ull;                    // byte c -> bit 8c
    bits |= ((row * 0x0102040810204080ull) >> 56) << (8 * r);                       // gather the 8 bits, byte 0 -> bit 0
  }
  return bits;
}

// the two grid rows a pixel row touches, as bytes (bit c = column c): two 64-bit shifts per lane, the per-pixel
// tests below are 32-bit
__device__ __forceinline__ uint2 grid_rows8(unsigned long long bits, const Tap& r) {
  return make_uint2(static_cast<uint32_t>(bits >> (r.i0 * 8)) & 0xFFu, static_cast<uint32_t>(bits >> (r.i1 * 8)) & 0xFFu);
}

__device__ __forceinline__ float blend8(uint2 rows, const Tap& r, const Tap& c, float lo, float hi) {
  const float wr0 = 1.f - r.t, wr1 = r.t, wc0 = 1.f - c.t, wc1 = c.t;
  float v = ((rows.x >> c.i0) & 1u) ? wr0 * wc0 : 0.f;
  v += ((rows.x >> c.i1) & 1u) ? wr0 * wc1 : 0.f;
  v += ((rows.y >> c.i0) & 1u) ? wr1 * wc0 : 0.f;
  v += ((rows.y >> c.i1) & 1u) ? wr1 * wc1 : 0.f;
  return fminf(fmaxf(v, lo), hi);
}

// NT: non-temporal stores for batches that cannot stay cache-resident anyway (see xai_rise_apply_f32); C3: the usual
// three channels -- the lane's three image quads are loaded before the mask arithmetic, so their latency hides behind it.
typedef float rise_f4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ void rise_store(float* p, float4 v) {
  if (NT) { const rise_f4 t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, reinterpret_cast<rise_f4*>(p)); }
  else st4(p, v);
}

template <bool NT, bool C3>
__global__ __launch_bounds__(kBlock) void rise_apply_kernel_s8(const uint8_t* __restrict__ grid, const int32_t* __restrict__ shift,
                                                               int cell_h, int cell_w, double rh, double rw, const float* __restrict__ image, int C,
                                                               int H, int W, float* __restrict__ masked, float* __restrict__ masks) {
  const int n = blockIdx.y;
  const int64_t hw = static_cast<int64_t>(H) * W;
  const int64_t p = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) * 4;
  if (p >= hw) return;
  float4 img[3];
  if (C3 && masked) {
#pragma unroll
    for (int c = 0; c < 3; ++c) img[c] = ld4(image + c * hw + p);
  }
  const unsigned long long bits = pack_grid8(grid + static_cast<int64_t>(n) * 64);
  const float hi = bits != 0ull ? 1.f : 0.f;
  const float lo = bits == ~0ull ? 1.f : 0.f;
  const int y = static_cast<int>(static_cast<uint32_t>(p) / static_cast<uint32_t>(W)), x = static_cast<int>(p - static_cast<int64_t>(y) * W);
  const int sx = shift[2 * n + 1];
  const Tap tr = make_tap(y + shift[2 * n], 8, rh);
  const uint2 rows = grid_rows8(bits, tr);
  float4 m;
  m.x = blend8(rows, tr, make_tap(x + sx, 8, rw), lo, hi);
  m.y = blend8(rows, tr, make_tap(x + 1 + sx, 8, rw), lo, hi);
  m.z = blend8(rows, tr, make_tap(x + 2 + sx, 8, rw), lo, hi);
  m.w = blend8(rows, tr, make_tap(x + 3 + sx, 8, rw), lo, hi);
  if (masks) rise_store<NT>(masks + static_cast<int64_t>(n) * hw + p, m);
  if (masked) {
    float* o = masked + static_cast<int64_t>(n) * C * hw + p;
    if (C3) {
#pragma unroll
      for (int c = 0; c < 3; ++c) rise_store<NT>(o + c * hw, make_float4(img[c].x * m.x, img[c].y * m.y, img[c].z * m.z, img[c].w * m.w));
    } else {
      for (int c = 0; c < C; ++c) {
        const float4 v = ld4(image + c * hw + p);
        rise_store<NT>(o + c * hw, make_float4(v.x * m.x, v.y * m.y, v.z * m.z, v.w * m.w));
      }
    }
  }
}

constexpr int kStage = 256;   // masks staged in LDS per round

__global__ __launch_bounds__(kBlock) void rise_accum_kernel(const uint8_t* __restrict__ grid, const int32_t* __restrict__ shift,
                                                            const float* __restrict__ scores, int n_masks, int per_slice, int s,
                                                            int cell_h, int cell_w, double rh, double rw, int H, int W, double scale,
                                                            double* __restrict__ acc_out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
  const int up_h = (s + 1) * cell_h, up_w = (s + 1) * cell_w;
  Tap* rtap = reinterpret_cast<Tap*>(lds_raw);                      // [up_h]
  Tap* ctap = rtap + up_h;                                          // [up_w]
  float* sc = reinterpret_cast<float*>(ctap + up_w);                // [kStage]
  int* sh = reinterpret_cast<int*>(sc + kStage);                    // [kStage][2]
  float* lim = reinterpret_cast<float*>(sh + 2 * kStage);           // [kStage][2] clip range of each mask
  uint8_t* gs = reinterpret_cast<uint8_t*>(lim + 2 * kStage);       // [kStage][s*s]
  for (int i = threadIdx.x; i < up_h; i += kBlock) rtap[i] = make_tap(i, s, rh);
  for (int i = threadIdx.x; i < up_w; i += kBlock) ctap[i] = make_tap(i, s, rw);
  const int64_t hw = static_cast<int64_t>(H) * W;
  const int64_t p = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const bool live = p < hw;
  const int y = live ? static_cast<int>(static_cast<uint32_t>(p) / static_cast<uint32_t>(W)) : 0, x = live ? static_cast<int>(p - static_cast<int64_t>(y) * W) : 0;
  const int n_lo = blockIdx.y * per_slice, n_hi = min(n_lo + per_slice, n_masks);
  const int ss = s * s;
  double acc = 0.0;
  for (int base = n_lo; base < n_hi; base += kStage) {
    const int cnt = min(kStage, n_hi - base);
    __syncthreads();
    for (int i = threadIdx.x; i < cnt; i += kBlock) {
      sc[i] = scores[base + i];
      sh[2 * i] = shift[2 * (base + i)];
      sh[2 * i + 1] = shift[2 * (base + i) + 1];
    }
    for (int i = threadIdx.x; i < cnt * ss; i += kBlock) gs[i] = grid[static_cast<int64_t>(base) * ss + i];
    __syncthreads();
    for (int i = threadIdx.x; i < cnt; i += kBlock) {
      int one = 0, zero = 0;
      for (int j = 0; j < ss; ++j) { one |= (gs[i * ss + j] != 0); zero |= (gs[i * ss + j] == 0); }
      lim[2 * i] = zero ? 0.f : 1.f;
      lim[2 * i + 1] = one ? 1.f : 0.f;
    }
    __syncthreads();
    if (live) {
      for (int m = 0; m < cnt; ++m) {
        const Tap tr = rtap[y + sh[2 * m]];
        const Tap tc = ctap[x + sh[2 * m + 1]];
        acc += static_cast<double>(sc[m]) * static_cast<double>(blend(gs + m * ss, s, tr, tc, lim[2 * m], lim[2 * m + 1]));
      }
    }
  }
  if (live) atomicAdd(acc_out + p, acc * scale);
}

// s == 8 accumulate: grids staged as one 64-bit word per mask (packed while staging), score and both shifts
// as one 8-byte record -- two broadcast LDS reads per mask instead of ~10 byte/word reads.
__global__ __launch_bounds__(kBlock) void rise_accum_kernel_s8(const uint8_t* __restrict__ grid, const int32_t* __restrict__ shift,
                                                               const float* __restrict__ scores, int n_masks, int per_slice,
                                                               int cell_h, int cell_w, double rh, double rw, int H, int W, double scale,
                                                               double* __restrict__ acc_out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw8[];
  const int up_h = 9 * cell_h, up_w = 9 * cell_w;
  unsigned long long* bits = reinterpret_cast<unsigned long long*>(lds_raw8);   // [kStage]
  float2* meta = reinterpret_cast<float2*>(bits + kStage);                      // [kStage] {score, packed shifts}
  Tap* rtap = reinterpret_cast<Tap*>(meta + kStage);                            // [up_h]
  Tap* ctap = rtap + up_h;                                                      // [up_w]
  for (int i = threadIdx.x; i < up_h; i += kBlock) rtap[i] = make_tap(i, 8, rh);
  for (int i = threadIdx.x; i < up_w; i += kBlock) ctap[i] = make_tap(i, 8, rw);
  const int64_t hw = static_cast<int64_t>(H) * W;
  const int64_t p = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const bool live = p < hw;
  const int y = live ? static_cast<int>(static_cast<uint32_t>(p) / static_cast<uint32_t>(W)) : 0, x = live ? static_cast<int>(p - static_cast<int64_t>(y) * W) : 0;
  const int n_lo = blockIdx.y * per_slice, n_hi = min(n_lo + per_slice, n_masks);
  double acc = 0.0;
  for (int base = n_lo; base < n_hi; base += kStage) {
    const int cnt = min(kStage, n_hi - base);
    __syncthreads();
    for (int i = threadIdx.x; i < cnt; i += kBlock) {
      bits[i] = pack_grid8(grid + static_cast<int64_t>(base + i) * 64);
      meta[i] = make_float2(scores[base + i], __int_as_float((shift[2 * (base + i)] & 0xFFFF) | (shift[2 * (base + i) + 1] << 16)));
    }
    __syncthreads();
    if (live) {
      for (int m = 0; m < cnt; ++m) {
        const unsigned long long b = bits[m];
        const float2 mt = meta[m];
        const int sh = __float_as_int(mt.y);
        const Tap tr = rtap[y + (sh & 0xFFFF)];
        const Tap tc = ctap[x + (sh >> 16)];
        const float hi = b != 0ull ? 1.f : 0.f, lo = b == ~0ull ? 1.f : 0.f;
        acc += static_cast<double>(mt.x) * static_cast<double>(blend8(grid_rows8(b, tr), tr, tc, lo, hi));
      }
    }
  }
  if (live) atomicAdd(acc_out + p, acc * scale);
}

}  // namespace

XAI_EXPORT int xai_rise_apply_f32(const uint8_t* grid, const int32_t* shift, int n_masks, int s, int cell_h, int cell_w,
                                  const float* image, int C, int H, int W, float* masked_out, float* masks_out,
                                  xai_stream_t stream) {
  XAI_REQUIRE_PTR(grid); XAI_REQUIRE_PTR(shift);
  XAI_REQUIRE(masked_out != nullptr || masks_out != nullptr, XAI_E_NULL);
  XAI_REQUIRE(masked_out == nullptr || image != nullptr, XAI_E_NULL);
  XAI_REQUIRE(n_masks > 0 && s > 0 && cell_h > 0 && cell_w > 0 && C > 0 && H > 0 && W > 0, XAI_E_SHAPE);
  XAI_REQUIRE(H + cell_h - 1 <= (s + 1) * cell_h && W + cell_w - 1 <= (s + 1) * cell_w, XAI_E_SHAPE);   // crop stays inside
  XAI_REQUIRE(s <= 64 && n_masks <= 65535, XAI_E_UNSUPPORTED);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t hw = static_cast<int64_t>(H) * W;
  XAI_REQUIRE(hw < (int64_t(1) << 31), XAI_E_UNSUPPORTED);
  const double rh = static_cast<double>(s) / static_cast<double>((s + 1) * cell_h), rw = static_cast<double>(s) / static_cast<double>((s + 1) * cell_w);
  const bool vec = (W % 4 == 0) && xai_aligned16(image) && xai_aligned16(masked_out) && xai_aligned16(masks_out);
  if (vec && s == 8 && (reinterpret_cast<uintptr_t>(grid) & 7u) == 0) {
    dim3 g(static_cast<unsigned>(xai_ceil_div(hw, kBlock * 4)), n_masks);
    // Store policy by what the classifier will find: a batch that fits the 256 MiB Infinity Cache with room to spare is
    // stored normally (the convolution that consumes it next reads it from cache); a larger one cannot stay resident
    // anyway and is streamed with non-temporal stores, which on this part write 13-15 % faster
    // (csrc/tune/tune_rise.hip, profiles/r02_tune_rise.txt: 1000 masks 104.9 -> 92.4 us).
    const int64_t out_bytes = static_cast<int64_t>(n_masks) * hw * 4 * ((masked_out ? C : 0) + (masks_out ? 1 : 0));
    const bool nt = out_bytes > (int64_t(128) << 20);
    const bool c3 = (C == 3);
#define XAI_RISE_S8(NT, C3) \
    hipLaunchKernelGGL((rise_apply_kernel_s8<NT, C3>), g, dim3(kBlock), 0, st, grid, shift, cell_h, cell_w, rh, rw, image, C, H, W, masked_out, masks_out)
    if (nt) { if (c3) XAI_RISE_S8(true, true); else XAI_RISE_S8(true, false); }
    else    { if (c3) XAI_RISE_S8(false, true); else XAI_RISE_S8(false, false); }
#undef XAI_RISE_S8
  } else if (vec) {
    dim3 g(static_cast<unsigned>(xai_ceil_div(hw, kBlock * 4)), n_masks);
    hipLaunchKernelGGL(rise_apply_kernel_v4, g, dim3(kBlock), s * s, st, grid, shift, s, cell_h, cell_w, rh, rw, image, C, H, W, masked_out, masks_out);
  } else {
    dim3 g(static_cast<unsigned>(xai_ceil_div(hw, kBlock)), n_masks);
    hipLaunchKernelGGL(rise_apply_kernel, g, dim3(kBlock), s * s, st, grid, shift, s, cell_h, cell_w, rh, rw, image, C, H, W, masked_out, masks_out);
  }
  return xai_launch_status();
}

XAI_EXPORT int xai_rise_accum_f64(const uint8_t* grid, const int32_t* shift, const float* scores, int n_masks, int s, int cell_h,
                                  int cell_w, int H, int W, double scale, double* acc, xai_stream_t stream) {
  XAI_REQUIRE_PTR(grid); XAI_REQUIRE_PTR(shift); XAI_REQUIRE_PTR(scores); XAI_REQUIRE_PTR(acc);
  XAI_REQUIRE(n_masks > 0 && s > 0 && cell_h > 0 && cell_w > 0 && H > 0 && W > 0, XAI_E_SHAPE);
  XAI_REQUIRE(H + cell_h - 1 <= (s + 1) * cell_h && W + cell_w - 1 <= (s + 1) * cell_w, XAI_E_SHAPE);
  const int up_h = (s + 1) * cell_h, up_w = (s + 1) * cell_w;
  const double rh = static_cast<double>(s) / static_cast<double>(up_h), rw = static_cast<double>(s) / static_cast<double>(up_w);
  const size_t lds = static_cast<size_t>(up_h + up_w) * sizeof(Tap) + kStage * (3 * sizeof(float) + 2 * sizeof(int)) +
                     static_cast<size_t>(kStage) * s * s;
  XAI_REQUIRE(s <= 64 && lds <= 64 * 1024, XAI_E_UNSUPPORTED);
  const int64_t hw = static_cast<int64_t>(H) * W;
  const int64_t tiles = xai_ceil_div(hw, kBlock);
  int slices = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(xai_ceil_div(n_masks, 64), xai_ceil_div(2048, tiles))));
  const int per = static_cast<int>(xai_ceil_div(n_masks, slices));
  slices = static_cast<int>(xai_ceil_div(n_masks, per));
  dim3 g(static_cast<unsigned>(tiles), slices);
  if (s == 8 && (reinterpret_cast<uintptr_t>(grid) & 7u) == 0 && cell_h < 32768 && cell_w < 32768) {
    const size_t lds8 = kStage * (sizeof(unsigned long long) + sizeof(float2)) + static_cast<size_t>(up_h + up_w) * sizeof(Tap);
    hipLaunchKernelGGL(rise_accum_kernel_s8, g, dim3(kBlock), lds8, static_cast<hipStream_t>(stream), grid, shift, scores, n_masks, per,
                       cell_h, cell_w, rh, rw, H, W, scale, acc);
    return xai_launch_status();
  }
  hipLaunchKernelGGL(rise_accum_kernel, g, dim3(kBlock), lds, static_cast<hipStream_t>(stream), grid, shift, scores, n_masks, per, s,
                     cell_h, cell_w, rh, rw, H, W, scale, acc);
  return xai_launch_status();
}
