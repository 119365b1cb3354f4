// Integrated-Gradients kernels for gfx950 (MI355X): path interpolation (K1), Left-IG cutoff,
// Riemann accumulation (K2, the roofline kernel), gradient filing, streaming accumulation, IDGI.
//
// All of them are HBM-bound element-wise / strided-reduction work with no data reuse, so nothing is
// staged through LDS; what matters is (see csrc/tune/*.hip for the measurements behind each choice):
//   * 16-byte lanes, 1 KiB contiguous per wave-instruction;
//   * K2: a balanced grid (2 workgroups per CU, equal item ranges -- the per-CU load rate saturates
//     before HBM does), step-outer streaming so the chip walks contiguous [img][step] rows, non-temporal
//     loads of the once-read gradient stream;
//   * writes (K1): many short concurrent row streams (2 rows per lane) beat long per-lane streams;
//   * producers of data that is read much later (gradient filing) use non-temporal stores so that their
//     dirty lines do not sit in the Infinity Cache when the reader starts.
// Compiled with -ffp-contract=off: products and sums round separately, exactly like the torch
// expressions they replace.
#include "xai_common.h"

#include <hip/hip_ext.h>

namespace {

constexpr int kBlock = 256;

// ---- tiny vector algebra so that every kernel exists in a float4 and a scalar flavour ----
template <int W> struct Vec;
typedef float fx4 __attribute__((ext_vector_type(4)));
template <> struct Vec<4> {
  using T = float4;
  static __device__ __forceinline__ T load(const float* p) { return *reinterpret_cast<const float4*>(p); }
  // streamed-once data: non-temporal (nt) load, does not displace x/out lines in L2 / Infinity Cache
  static __device__ __forceinline__ T load_nt(const float* p) {
    const fx4 v = __builtin_nontemporal_load(reinterpret_cast<const fx4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
  }
  static __device__ __forceinline__ void store(float* p, T v) { *reinterpret_cast<float4*>(p) = v; }
  static __device__ __forceinline__ void store_nt(float* p, T v) {
    const fx4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<fx4*>(p));
  }
  static __device__ __forceinline__ T splat(float s) { return make_float4(s, s, s, s); }
};
template <> struct Vec<1> {
  using T = float;
  static __device__ __forceinline__ T load(const float* p) { return *p; }
  static __device__ __forceinline__ T load_nt(const float* p) { return __builtin_nontemporal_load(p); }
  static __device__ __forceinline__ void store(float* p, T v) { *p = v; }
  static __device__ __forceinline__ void store_nt(float* p, T v) { __builtin_nontemporal_store(v, p); }
  static __device__ __forceinline__ T splat(float s) { return s; }
};
__device__ __forceinline__ float4 vadd(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 vsub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 vmul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 vdiv(float4 a, float4 b) { return make_float4(a.x / b.x, a.y / b.y, a.z / b.z, a.w / b.w); }
__device__ __forceinline__ float4 vabs(float4 a) { return make_float4(fabsf(a.x), fabsf(a.y), fabsf(a.z), fabsf(a.w)); }
__device__ __forceinline__ float vadd(float a, float b) { return a + b; }
__device__ __forceinline__ float vsub(float a, float b) { return a - b; }
__device__ __forceinline__ float vmul(float a, float b) { return a * b; }
__device__ __forceinline__ float vdiv(float a, float b) { return a / b; }
__device__ __forceinline__ float vabs(float a) { return fabsf(a); }

// =========================================================================== K1 interpolation
// grid = (element tiles, step chunks, images).  A lane keeps x and (x-b) of its column in
// registers and writes `steps_per_chunk` rows; alphas come through the scalar cache.  NT: non-temporal stores, for
// outputs that cannot stay cache-resident for the classifier anyway (same policy and evidence as K4, rise_kernels.hip).
template <int W, bool NT>
__global__ __launch_bounds__(kBlock) void ig_interp_kernel(const float* __restrict__ x, const float* __restrict__ base,
                                                           float base_scalar, const float* __restrict__ alphas,
                                                           int64_t alpha_img_stride, int n_alpha, int64_t n_elem,
                                                           int steps_per_chunk, float* __restrict__ out) {
  using V = Vec<W>;
  const int64_t e = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) * W;
  if (e >= n_elem) return;
  const int img = blockIdx.z;
  const int s0 = blockIdx.y * steps_per_chunk;
  const int s1 = min(s0 + steps_per_chunk, n_alpha);
  const typename V::T xv = V::load(x + img * n_elem + e);
  const typename V::T bv = base ? V::load(base + img * n_elem + e) : V::splat(base_scalar);
  const typename V::T dv = vsub(xv, bv);
  const float* al = alphas + img * alpha_img_stride;
  float* o = out + (static_cast<int64_t>(img) * n_alpha + s0) * n_elem + e;
  for (int s = s0; s < s1; ++s, o += n_elem) {
    const typename V::T v = vadd(bv, vmul(V::splat(al[s]), dv));
    if (NT) V::store_nt(o, v); else V::store(o, v);
  }
}

// =========================================================================== Left-IG cutoff
__global__ __launch_bounds__(kWave) void ig_cutoff_kernel(const float* __restrict__ logits, int n_steps, float alpha_star,
                                                          int32_t* __restrict__ n_use) {
  const float* lg = logits + static_cast<int64_t>(blockIdx.x) * n_steps;
  const int lane = threadIdx.x;
  float m = -INFINITY;
  for (int s = lane; s < n_steps; s += kWave) m = fmaxf(m, lg[s]);
  m = wave_max(m);
  const float thr = m * alpha_star;
  int first = INT32_MAX;
  for (int s = lane; s < n_steps; s += kWave)
    if (lg[s] > thr) { first = s; break; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) first = min(first, __shfl_xor(first, off, kWave));
  if (lane == 0) {
    int cut = (first == INT32_MAX) ? 1 : first;   // nothing above the threshold -> 1
    if (cut == 0) cut = 1;                        // "avoid rare case where no attribution is returned"
    n_use[blockIdx.x] = (alpha_star == 1.0f) ? n_steps : cut;
  }
}

// =========================================================================== K2 accumulation
// grid = (pixel tiles, images).  A lane owns W consecutive pixels and walks channel-major
// through the first n_use step rows: per channel a sequential fp32 sum (s ascending), divided
// by n_use, times (x - b); the channel sum feeds the optional |.| map.
template <int W, bool WEIGHTED>
__global__ __launch_bounds__(kBlock) void ig_accum_kernel(const float* __restrict__ grads, int n_steps,
                                                          const int32_t* __restrict__ n_use_dev, int n_use_host,
                                                          const float* __restrict__ w1, const float* __restrict__ w2,
                                                          const float* __restrict__ x, const float* __restrict__ base,
                                                          float base_scalar, int C, int64_t hw,
                                                          float* __restrict__ out, float* __restrict__ out_abs) {
  using V = Vec<W>;
  using T = typename V::T;
  constexpr int U = 8;
  const int64_t p = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) * W;
  if (p >= hw) return;
  const int img = blockIdx.y;
  int n_use = n_use_dev ? n_use_dev[img] : n_use_host;
  n_use = max(1, min(n_use, n_steps));
  const T denom = V::splat(static_cast<float>(WEIGHTED ? n_steps : n_use));
  const int64_t row = static_cast<int64_t>(C) * hw;               // floats between two steps
  const float* gi = grads + static_cast<int64_t>(img) * n_steps * row + p;
  const float* wa = WEIGHTED ? w1 + static_cast<int64_t>(img) * n_steps : nullptr;
  const float* wb = (WEIGHTED && w2) ? w2 + static_cast<int64_t>(img) * n_steps : nullptr;
  T tot = V::splat(0.f);
  for (int c = 0; c < C; ++c) {
    const float* g = gi + c * hw;
    T acc = V::splat(0.f);
    int s = 0;
    for (; s + U <= n_use; s += U) {
      T v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = V::load_nt(g + (s + u) * row);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (WEIGHTED) {
          v[u] = vmul(v[u], V::splat(wa[s + u]));
          if (wb) v[u] = vmul(v[u], V::splat(wb[s + u]));
        }
        acc = vadd(acc, v[u]);
      }
    }
    for (; s < n_use; ++s) {
      T v = V::load_nt(g + s * row);
      if (WEIGHTED) {
        v = vmul(v, V::splat(wa[s]));
        if (wb) v = vmul(v, V::splat(wb[s]));
      }
      acc = vadd(acc, v);
    }
    const int64_t at = (static_cast<int64_t>(img) * C + c) * hw + p;
    const T bv = base ? V::load(base + at) : V::splat(base_scalar);
    const T o = vmul(vdiv(acc, denom), vsub(V::load(x + at), bv));
    V::store(out + at, o);
    tot = vadd(tot, o);
  }
  if (out_abs) V::store(out_abs + static_cast<int64_t>(img) * hw + p, vabs(tot));
}

// K2, production mapping for C in {1,3}, hw % 4 == 0 ("step-outer, balanced"):
//   * work item = 4 consecutive pixels of one image, all C channels; every workgroup owns an equal
//     contiguous range of items and the grid is sized to 2 workgroups per CU, so each CU moves the
//     same number of bytes (the per-CU load rate, ~11 B/clk, is what saturates first);
//   * a lane keeps ITEMS x C float4 accumulators in registers and the workgroup walks the step
//     rows TOGETHER: at any moment it streams contiguous runs of one [img][s] row instead of every
//     lane striding 4*C*hw bytes per load -- DRAM-page friendly (tune/tune_accum.hip: 6.65 TB/s vs
//     5.3 TB/s for the lane-strided mapping on 32 x 50 x 3x224x224);
//   * ITEMS*C independent nt loads in flight per lane and step.
// Per (pixel, channel) the sum still runs over s ascending in fp32: bit-identical to ig_accum_kernel.
template <int BLOCK, int ITEMS, int SU, int C, bool WEIGHTED>
__global__ __launch_bounds__(BLOCK) void ig_accum_stream_kernel(const float* __restrict__ grads, int n_steps,
                                                                const int32_t* __restrict__ n_use_dev, int n_use_host,
                                                                const float* __restrict__ w1, const float* __restrict__ w2,
                                                                const float* __restrict__ x, const float* __restrict__ base,
                                                                float base_scalar, int64_t hw, int n_img,
                                                                float* __restrict__ out, float* __restrict__ out_abs) {
  using V = Vec<4>;
  const int64_t hw4 = hw >> 2;
  const int64_t items = static_cast<int64_t>(n_img) * hw4;
  const int64_t per = (items + gridDim.x - 1) / gridDim.x;
  const int64_t lo = static_cast<int64_t>(blockIdx.x) * per;
  const int64_t hi = min(lo + per, items);
  const int64_t row = static_cast<int64_t>(C) * hw;                 // floats between two steps
  for (int64_t first = lo; first < hi; first += static_cast<int64_t>(BLOCK) * ITEMS) {
    const float* gp[ITEMS];
    const float* wa[ITEMS];
    const float* wb[ITEMS];
    int64_t at[ITEMS], img_of[ITEMS];
    int nu[ITEMS];
    bool live[ITEMS];
    float4 acc[ITEMS][C];
    int n_min = n_steps, n_max = 0;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      const int64_t it = first + static_cast<int64_t>(i) * BLOCK + threadIdx.x;
      live[i] = it < hi;
      const int64_t itc = live[i] ? it : lo;           // idle slots shadow a valid item (loads stay in bounds, nothing is stored)
      const int64_t img = itc / hw4;
      const int64_t p = (itc - img * hw4) << 2;
      img_of[i] = img;
      gp[i] = grads + img * n_steps * row + p;
      at[i] = img * row + p;
      const int n = n_use_dev ? n_use_dev[img] : n_use_host;
      nu[i] = max(1, min(n, n_steps));
      n_min = min(n_min, nu[i]);
      n_max = max(n_max, nu[i]);
      wa[i] = WEIGHTED ? w1 + img * n_steps : nullptr;
      wb[i] = (WEIGHTED && w2) ? w2 + img * n_steps : nullptr;
#pragma unroll
      for (int c = 0; c < C; ++c) acc[i][c] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // common prefix: no predicates, SU*ITEMS*C loads issued back to back, added in step order
    int s = 0;
    for (; s + SU <= n_min; s += SU) {
      float4 v[SU][ITEMS][C];
#pragma unroll
      for (int u = 0; u < SU; ++u)
#pragma unroll
        for (int i = 0; i < ITEMS; ++i)
#pragma unroll
          for (int c = 0; c < C; ++c) v[u][i][c] = V::load_nt(gp[i] + (s + u) * row + c * hw);
#pragma unroll
      for (int u = 0; u < SU; ++u)
#pragma unroll
        for (int i = 0; i < ITEMS; ++i)
#pragma unroll
          for (int c = 0; c < C; ++c) {
            float4 t = v[u][i][c];
            if (WEIGHTED) {
              t = vmul(t, V::splat(wa[i][s + u]));
              if (wb[i]) t = vmul(t, V::splat(wb[i][s + u]));
            }
            acc[i][c] = vadd(acc[i][c], t);
          }
    }
    for (; s < n_min; ++s) {
#pragma unroll
      for (int i = 0; i < ITEMS; ++i)
#pragma unroll
        for (int c = 0; c < C; ++c) {
          float4 t = V::load_nt(gp[i] + s * row + c * hw);
          if (WEIGHTED) {
            t = vmul(t, V::splat(wa[i][s]));
            if (wb[i]) t = vmul(t, V::splat(wb[i][s]));
          }
          acc[i][c] = vadd(acc[i][c], t);
        }
    }
    // ragged tail: only when the lane's items straddle images with different Left-IG cutoffs
    for (s = n_min; s < n_max; ++s) {
#pragma unroll
      for (int i = 0; i < ITEMS; ++i)
        if (s < nu[i]) {
#pragma unroll
          for (int c = 0; c < C; ++c) {
            float4 t = V::load_nt(gp[i] + s * row + c * hw);
            if (WEIGHTED) {
              t = vmul(t, V::splat(wa[i][s]));
              if (wb[i]) t = vmul(t, V::splat(wb[i][s]));
            }
            acc[i][c] = vadd(acc[i][c], t);
          }
        }
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      if (!live[i]) continue;
      const float4 denom = V::splat(static_cast<float>(WEIGHTED ? n_steps : nu[i]));
      float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const int64_t a = at[i] + c * hw;
        const float4 bv = base ? V::load(base + a) : V::splat(base_scalar);
        const float4 o = vmul(vdiv(acc[i][c], denom), vsub(V::load(x + a), bv));
        V::store(out + a, o);
        tot = vadd(tot, o);
      }
      if (out_abs) V::store(out_abs + img_of[i] * hw + (at[i] - img_of[i] * row), vabs(tot));
    }
  }
}

// Streaming copy of a pass's step gradients into the [img][step] buffer with NON-TEMPORAL stores:
// a plain-store copy leaves up to 256 MiB of dirty lines in the Infinity Cache whose write-back
// then competes with the accumulation kernel's reads (measured: 963 MB read 200 us after a plain
// 60 MB copy vs 155 us after an nt-store copy -- tune/tune_copy_then_accum.hip).
__global__ __launch_bounds__(kBlock) void store_stream_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t n4) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  const fx4* s4 = reinterpret_cast<const fx4*>(src);
  fx4* d4 = reinterpret_cast<fx4*>(dst);
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n4; i += stride)
    __builtin_nontemporal_store(s4[i], d4 + i);
}
__global__ __launch_bounds__(kBlock) void store_stream_scalar_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride)
    __builtin_nontemporal_store(src[i], dst + i);
}

// streaming form: acc += sum over the batch rows
template <int W>
__global__ __launch_bounds__(kBlock) void ig_accum_add_kernel(const float* __restrict__ grads, int n_batch,
                                                              float* __restrict__ acc, int64_t n_elem) {
  using V = Vec<W>;
  using T = typename V::T;
  constexpr int U = 8;
  const int64_t e = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) * W;
  if (e >= n_elem) return;
  const float* g = grads + e;
  T a = V::load(acc + e);
  int b = 0;
  for (; b + U <= n_batch; b += U) {
    T v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = V::load_nt(g + (b + u) * n_elem);
#pragma unroll
    for (int u = 0; u < U; ++u) a = vadd(a, v[u]);
  }
  for (; b < n_batch; ++b) a = vadd(a, V::load_nt(g + b * n_elem));
  V::store(acc + e, a);
}

template <int W>
__global__ __launch_bounds__(kBlock) void ig_finish_kernel(const float* __restrict__ acc, int n_steps,
                                                           const float* __restrict__ x, const float* __restrict__ base,
                                                           float base_scalar, int C, int64_t hw, float* __restrict__ out,
                                                           float* __restrict__ out_abs) {
  using V = Vec<W>;
  using T = typename V::T;
  const int64_t p = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) * W;
  if (p >= hw) return;
  const int img = blockIdx.y;
  const T denom = V::splat(static_cast<float>(n_steps));
  T tot = V::splat(0.f);
  for (int c = 0; c < C; ++c) {
    const int64_t at = (static_cast<int64_t>(img) * C + c) * hw + p;
    const T bv = base ? V::load(base + at) : V::splat(base_scalar);
    const T o = vmul(vdiv(V::load(acc + at), denom), vsub(V::load(x + at), bv));
    V::store(out + at, o);
    tot = vadd(tot, o);
  }
  if (out_abs) V::store(out_abs + static_cast<int64_t>(img) * hw + p, vabs(tot));
}

// =========================================================================== IDGI
// one 1024-thread workgroup per row: lane-strided float4 loads, fp32 partials, wave shuffle
// reduce, 16 wave partials through LDS -- a fixed tree, so the result is reproducible.
__global__ __launch_bounds__(1024) void sumsq_kernel(const float* __restrict__ g, int64_t n_elem, float* __restrict__ out) {
  __shared__ float part[16];
  const float* row = g + static_cast<int64_t>(blockIdx.x) * n_elem;
  float acc = 0.f;
  const bool vec = ((n_elem & 3) == 0) && ((reinterpret_cast<uintptr_t>(row) & 15u) == 0);
  if (vec) {
    for (int64_t i = threadIdx.x * 4; i < n_elem; i += 1024 * 4) {
      const float4 v = ld4(row + i);
      acc += v.x * v.x; acc += v.y * v.y; acc += v.z * v.z; acc += v.w * v.w;
    }
  } else {
    for (int64_t i = threadIdx.x; i < n_elem; i += 1024) acc += row[i] * row[i];
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x < 64) {
    float v = threadIdx.x < 16 ? part[threadIdx.x] : 0.f;
    v = wave_sum(v);
    if (threadIdx.x == 0) out[blockIdx.x] = v;
  }
}

template <int W>
__global__ __launch_bounds__(kBlock) void idgi_accum_kernel(const float* __restrict__ grads, int n_steps,
                                                            const float* __restrict__ logits, const float* __restrict__ sumsq,
                                                            int64_t n_elem, float* __restrict__ out) {
  using V = Vec<W>;
  using T = typename V::T;
  const int64_t e = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) * W;
  if (e >= n_elem) return;
  T acc = V::splat(0.f);
  for (int s = 0; s + 1 < n_steps; ++s) {
    const T g = V::load(grads + s * n_elem + e);
    const float d = logits[s + 1] - logits[s];
    acc = vadd(acc, vdiv(vmul(vmul(g, g), V::splat(d)), V::splat(sumsq[s])));
  }
  V::store(out + e, acc);
}

inline bool can_vec4(int64_t n, std::initializer_list<const void*> ptrs) {
  if (n & 3) return false;
  for (const void* p : ptrs)
    if (p && !xai_aligned16(p)) return false;
  return true;
}

}  // namespace

// ============================================================================== C ABI
XAI_EXPORT int xai_ig_interp_f32(const float* x, const float* baseline, float baseline_scalar, const float* alphas,
                                 int64_t alpha_img_stride, int n_img, int n_alpha, int64_t n_elem, float* out,
                                 xai_stream_t stream) {
  XAI_REQUIRE_PTR(x); XAI_REQUIRE_PTR(alphas); XAI_REQUIRE_PTR(out);
  XAI_REQUIRE(n_img > 0 && n_alpha > 0 && n_elem > 0 && alpha_img_stride >= 0, XAI_E_SHAPE);
  XAI_REQUIRE(n_img <= 65535, XAI_E_UNSUPPORTED);
  hipStream_t st = static_cast<hipStream_t>(stream);
  // two step rows per lane: on this part HBM writes like MANY concurrent row streams (5.98 TB/s at 2 rows per
  // lane vs 5.42 TB/s at 50, tune/tune_write.hip); x and b re-reads hit L2
  const bool vec = can_vec4(n_elem, {x, baseline, out});
  const int64_t tiles = xai_ceil_div(n_elem, kBlock * (vec ? 4 : 1));
  int per, chunks;
  const bool hbm_sized = static_cast<int64_t>(n_img) * n_alpha * n_elem * 4 >= (int64_t(256) << 20);
  if (hbm_sized) {
    per = n_alpha >= 2 ? 2 : 1;                                   // HBM-sized output: many short streams, non-temporal
  } else {                                                        // cache-sized output: just enough workgroups to fill the chip
    const int c0 = static_cast<int>(std::min<int64_t>(n_alpha, std::max<int64_t>(1, xai_ceil_div(2048, tiles * n_img))));
    per = static_cast<int>(xai_ceil_div(n_alpha, c0));
  }
  chunks = static_cast<int>(xai_ceil_div(n_alpha, per));
  XAI_REQUIRE(chunks <= 65535, XAI_E_UNSUPPORTED);
  dim3 grid(static_cast<unsigned>(tiles), chunks, n_img);
#define XAI_INTERP(W, NT) \
  hipLaunchKernelGGL((ig_interp_kernel<W, NT>), grid, dim3(kBlock), 0, st, x, baseline, baseline_scalar, alphas, alpha_img_stride, n_alpha, n_elem, per, out)
  if (vec) { if (hbm_sized) XAI_INTERP(4, true); else XAI_INTERP(4, false); }
  else     { if (hbm_sized) XAI_INTERP(1, true); else XAI_INTERP(1, false); }
#undef XAI_INTERP
  return xai_launch_status();
}

XAI_EXPORT int xai_ig_cutoff_f32(const float* logits, int n_img, int n_steps, float alpha_star, int32_t* n_use,
                                 xai_stream_t stream) {
  XAI_REQUIRE_PTR(logits); XAI_REQUIRE_PTR(n_use);
  XAI_REQUIRE(n_img > 0 && n_steps > 0, XAI_E_SHAPE);
  hipLaunchKernelGGL(ig_cutoff_kernel, dim3(n_img), dim3(kWave), 0, static_cast<hipStream_t>(stream), logits, n_steps, alpha_star, n_use);
  return xai_launch_status();
}

// The launch of K2.  ev0 / ev1 (both or neither): the kernel's own start / stop timestamps are recorded into the caller's
// events by the dispatch itself (hipExtLaunchKernelGGL) -- a timing without the ~5 us of dispatch latency that two events
// bracketing a launch include.
static int ig_accum_impl(const float* grads, int n_img, int n_steps, const int32_t* n_use_dev, int n_use_host,
                         const float* step_w1, const float* step_w2, const float* x, const float* baseline,
                         float baseline_scalar, int C, int64_t hw, float* out_chw, float* out_abs_hw,
                         hipEvent_t ev0, hipEvent_t ev1, xai_stream_t stream) {
  XAI_REQUIRE_PTR(grads); XAI_REQUIRE_PTR(x); XAI_REQUIRE_PTR(out_chw);
  XAI_REQUIRE(n_img > 0 && n_steps > 0 && C > 0 && hw > 0, XAI_E_SHAPE);
  XAI_REQUIRE(n_use_dev != nullptr || (n_use_host >= 1 && n_use_host <= n_steps), XAI_E_SHAPE);
  XAI_REQUIRE(step_w1 != nullptr || step_w2 == nullptr, XAI_E_SHAPE);
  XAI_REQUIRE(n_img <= 65535, XAI_E_UNSUPPORTED);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool vec = can_vec4(hw, {grads, x, baseline, out_chw, out_abs_hw});
  if (vec && (C == 3 || C == 1)) {
    // step-outer balanced mapping.  Big problems: 2 workgroups of 256 lanes per CU, 4 items per lane
    // and round; small ones: one wave per 64 items so that a single image still spreads over the chip.
    const int64_t items = static_cast<int64_t>(n_img) * (hw / 4);
    const int cus = xai_cu_count();
    const bool big = items >= static_cast<int64_t>(cus) * 2 * 256 * 2;
#define XAI_STREAM(BLK, IT, SU, CC, WT, GRID) \
  do { if (ev0) hipExtLaunchKernelGGL((ig_accum_stream_kernel<BLK, IT, SU, CC, WT>), dim3(GRID), dim3(BLK), 0, st, ev0, ev1, 0, grads, n_steps, \
                                      n_use_dev, n_use_host, step_w1, step_w2, x, baseline, baseline_scalar, hw, n_img, out_chw, out_abs_hw); \
       else hipLaunchKernelGGL((ig_accum_stream_kernel<BLK, IT, SU, CC, WT>), dim3(GRID), dim3(BLK), 0, st, grads, n_steps, n_use_dev, n_use_host, \
                               step_w1, step_w2, x, baseline, baseline_scalar, hw, n_img, out_chw, out_abs_hw); } while (0)
  // small (cache-resident, latency-bound) problems: few lanes, so each keeps 5 steps x C loads in flight
#define XAI_STREAM_C(CC, WT) \
  do { if (big) XAI_STREAM(256, 4, 1, CC, WT, static_cast<unsigned>(cus * 2)); \
       else XAI_STREAM(64, 1, 5, CC, WT, static_cast<unsigned>(xai_ceil_div(items, 64))); } while (0)
    if (C == 3) { if (step_w1) XAI_STREAM_C(3, true); else XAI_STREAM_C(3, false); }
    else        { if (step_w1) XAI_STREAM_C(1, true); else XAI_STREAM_C(1, false); }
#undef XAI_STREAM_C
#undef XAI_STREAM
    return xai_launch_status();
  }
  dim3 grid(static_cast<unsigned>(xai_ceil_div(hw, kBlock * (vec ? 4 : 1))), n_img);
#define XAI_ACCUM(W, WT) \
  do { if (ev0) hipExtLaunchKernelGGL((ig_accum_kernel<W, WT>), grid, dim3(kBlock), 0, st, ev0, ev1, 0, grads, n_steps, n_use_dev, n_use_host, \
                                      step_w1, step_w2, x, baseline, baseline_scalar, C, hw, out_chw, out_abs_hw); \
       else hipLaunchKernelGGL((ig_accum_kernel<W, WT>), grid, dim3(kBlock), 0, st, grads, n_steps, n_use_dev, n_use_host, step_w1, \
                               step_w2, x, baseline, baseline_scalar, C, hw, out_chw, out_abs_hw); } while (0)
  if (vec) { if (step_w1) XAI_ACCUM(4, true); else XAI_ACCUM(4, false); }
  else     { if (step_w1) XAI_ACCUM(1, true); else XAI_ACCUM(1, false); }
#undef XAI_ACCUM
  return xai_launch_status();
}

XAI_EXPORT int xai_ig_accum_f32(const float* grads, int n_img, int n_steps, const int32_t* n_use_dev, int n_use_host,
                                const float* step_w1, const float* step_w2, const float* x, const float* baseline,
                                float baseline_scalar, int C, int64_t hw, float* out_chw, float* out_abs_hw,
                                xai_stream_t stream) {
  return ig_accum_impl(grads, n_img, n_steps, n_use_dev, n_use_host, step_w1, step_w2, x, baseline, baseline_scalar, C, hw, out_chw,
                       out_abs_hw, nullptr, nullptr, stream);
}

XAI_EXPORT int xai_ig_accum_timed_f32(const float* grads, int n_img, int n_steps, const int32_t* n_use_dev, int n_use_host,
                                      const float* step_w1, const float* step_w2, const float* x, const float* baseline,
                                      float baseline_scalar, int C, int64_t hw, float* out_chw, float* out_abs_hw,
                                      void* start_event, void* stop_event, xai_stream_t stream) {
  XAI_REQUIRE_PTR(start_event); XAI_REQUIRE_PTR(stop_event);
  return ig_accum_impl(grads, n_img, n_steps, n_use_dev, n_use_host, step_w1, step_w2, x, baseline, baseline_scalar, C, hw, out_chw,
                       out_abs_hw, static_cast<hipEvent_t>(start_event), static_cast<hipEvent_t>(stop_event), stream);
}

XAI_EXPORT int xai_ig_store_grads_f32(const float* src, float* dst, int64_t n_elem, xai_stream_t stream) {
  XAI_REQUIRE_PTR(src); XAI_REQUIRE_PTR(dst);
  XAI_REQUIRE(n_elem > 0, XAI_E_SHAPE);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned grid = static_cast<unsigned>(xai_cu_count() * 8);
  if (can_vec4(n_elem, {src, dst}))
    hipLaunchKernelGGL(store_stream_kernel, dim3(grid), dim3(kBlock), 0, st, src, dst, n_elem / 4);
  else
    hipLaunchKernelGGL(store_stream_scalar_kernel, dim3(grid), dim3(kBlock), 0, st, src, dst, n_elem);
  return xai_launch_status();
}

XAI_EXPORT int xai_ig_accum_add_f32(const float* grads, int n_batch, float* acc, int64_t n_elem, xai_stream_t stream) {
  XAI_REQUIRE_PTR(grads); XAI_REQUIRE_PTR(acc);
  XAI_REQUIRE(n_batch > 0 && n_elem > 0, XAI_E_SHAPE);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool vec = can_vec4(n_elem, {grads, acc});
  dim3 grid(static_cast<unsigned>(xai_ceil_div(n_elem, kBlock * (vec ? 4 : 1))));
  if (vec) hipLaunchKernelGGL(ig_accum_add_kernel<4>, grid, dim3(kBlock), 0, st, grads, n_batch, acc, n_elem);
  else     hipLaunchKernelGGL(ig_accum_add_kernel<1>, grid, dim3(kBlock), 0, st, grads, n_batch, acc, n_elem);
  return xai_launch_status();
}

XAI_EXPORT int xai_ig_finish_f32(const float* acc, int n_img, int n_steps, const float* x, const float* baseline,
                                 float baseline_scalar, int C, int64_t hw, float* out_chw, float* out_abs_hw,
                                 xai_stream_t stream) {
  XAI_REQUIRE_PTR(acc); XAI_REQUIRE_PTR(x); XAI_REQUIRE_PTR(out_chw);
  XAI_REQUIRE(n_img > 0 && n_steps > 0 && C > 0 && hw > 0, XAI_E_SHAPE);
  XAI_REQUIRE(n_img <= 65535, XAI_E_UNSUPPORTED);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool vec = can_vec4(hw, {acc, x, baseline, out_chw, out_abs_hw});
  dim3 grid(static_cast<unsigned>(xai_ceil_div(hw, kBlock * (vec ? 4 : 1))), n_img);
  if (vec) hipLaunchKernelGGL(ig_finish_kernel<4>, grid, dim3(kBlock), 0, st, acc, n_steps, x, baseline, baseline_scalar, C, hw, out_chw, out_abs_hw);
  else     hipLaunchKernelGGL(ig_finish_kernel<1>, grid, dim3(kBlock), 0, st, acc, n_steps, x, baseline, baseline_scalar, C, hw, out_chw, out_abs_hw);
  return xai_launch_status();
}

XAI_EXPORT int xai_sumsq_f32(const float* grads, int n_rows, int64_t n_elem, float* sumsq, xai_stream_t stream) {
  XAI_REQUIRE_PTR(grads); XAI_REQUIRE_PTR(sumsq);
  XAI_REQUIRE(n_rows > 0 && n_elem > 0, XAI_E_SHAPE);
  hipLaunchKernelGGL(sumsq_kernel, dim3(n_rows), dim3(1024), 0, static_cast<hipStream_t>(stream), grads, n_elem, sumsq);
  return xai_launch_status();
}

XAI_EXPORT int xai_idgi_accum_f32(const float* grads, int n_steps, const float* logits, const float* sumsq,
                                  int64_t n_elem, float* out, xai_stream_t stream) {
  XAI_REQUIRE_PTR(grads); XAI_REQUIRE_PTR(logits); XAI_REQUIRE_PTR(sumsq); XAI_REQUIRE_PTR(out);
  XAI_REQUIRE(n_steps > 1 && n_elem > 0, XAI_E_SHAPE);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool vec = can_vec4(n_elem, {grads, out});
  dim3 grid(static_cast<unsigned>(xai_ceil_div(n_elem, kBlock * (vec ? 4 : 1))));
  if (vec) hipLaunchKernelGGL(idgi_accum_kernel<4>, grid, dim3(kBlock), 0, st, grads, n_steps, logits, sumsq, n_elem, out);
  else     hipLaunchKernelGGL(idgi_accum_kernel<1>, grid, dim3(kBlock), 0, st, grads, n_steps, logits, sumsq, n_elem, out);
  return xai_launch_status();
}
