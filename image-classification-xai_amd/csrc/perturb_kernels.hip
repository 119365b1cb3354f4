// Insertion/deletion loop kernels for gfx950: flip-step map, perturbed-batch builder (K6),
// density segment sums (K10) and the per-row softmax statistics (K9).
//
// K6 is write-bound (n_batch * C * hw * 4 B out, 2*C*hw*4 + hw*4 B in): a lane keeps the
// start/finish values and the flip step of its 4 pixels in registers and emits one 16-byte
// store per (step, channel); wave-instructions are 1 KiB contiguous.
#include "xai_common.h"

namespace {

constexpr int kBlock = 256;

__global__ __launch_bounds__(kBlock) void flip_steps_kernel(const int32_t* __restrict__ rank, int64_t hw, int descending,
                                                            int step_size, int32_t* __restrict__ flip) {
  const int64_t p = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (p >= hw) return;
  const int64_t pos = descending ? (hw - 1 - rank[p]) : rank[p];
  flip[p] = static_cast<int32_t>(pos / step_size);
}

// grid = (pixel tiles, step chunks, channel groups): gridDim.z == 1 -> a lane handles all channels,
// gridDim.z == C -> one channel per lane (more, shorter write streams for HBM-sized batches)
template <int W>
__global__ __launch_bounds__(kBlock) void perturb_kernel(const float* __restrict__ start, const float* __restrict__ finish,
                                                         const int32_t* __restrict__ flip, int C, int64_t hw, int first_step,
                                                         int n_batch, int per_chunk, float* __restrict__ out) {
  const int64_t p = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) * W;
  if (p >= hw) return;
  const int k0 = blockIdx.y * per_chunk;
  const int k1 = min(k0 + per_chunk, n_batch);
  const int64_t img = static_cast<int64_t>(C) * hw;
  const int c_lo = gridDim.z == 1 ? 0 : blockIdx.z, c_hi = gridDim.z == 1 ? C : blockIdx.z + 1;
  if constexpr (W == 4) {
    const int4 f = *reinterpret_cast<const int4*>(flip + p);
    for (int c = c_lo; c < c_hi; ++c) {
      const float4 s = ld4(start + c * hw + p);
      const float4 e = ld4(finish + c * hw + p);
      float* o = out + k0 * img + c * hw + p;
      for (int k = k0; k < k1; ++k, o += img) {
        const int t = first_step + k;
        st4(o, make_float4(f.x <= t ? e.x : s.x, f.y <= t ? e.y : s.y, f.z <= t ? e.z : s.z, f.w <= t ? e.w : s.w));
      }
    }
  } else {
    const int f = flip[p];
    for (int c = c_lo; c < c_hi; ++c) {
      const float s = start[c * hw + p], e = finish[c * hw + p];
      float* o = out + k0 * img + c * hw + p;
      for (int k = k0; k < k1; ++k, o += img) *o = (f <= first_step + k) ? e : s;
    }
  }
}

// One wave per step: gather the step's pixels through `order`, fp32 lane partials, shuffle
// reduce.  The extra last block sums the whole map (fixed tree: 1024 lanes -> 16 waves).
__global__ __launch_bounds__(1024) void segment_sums_kernel(const float* __restrict__ sal, const int32_t* __restrict__ order,
                                                            int64_t hw, int descending, int step_size, int n_steps,
                                                            float* __restrict__ seg, float* __restrict__ total) {
  __shared__ float part[16];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (static_cast<int>(blockIdx.x) == (n_steps + 15) / 16) {   // the "total" block
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < hw; i += 1024) acc += sal[i];
    acc = wave_sum(acc);
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (threadIdx.x < 64) {
      float v = threadIdx.x < 16 ? part[threadIdx.x] : 0.f;
      v = wave_sum(v);
      if (threadIdx.x == 0) *total = v;
    }
    return;
  }
  const int t = blockIdx.x * 16 + wave;
  if (t >= n_steps) return;
  const int64_t lo = static_cast<int64_t>(t) * step_size;
  const int64_t hi = min(lo + step_size, hw);
  float acc = 0.f;
  for (int64_t i = lo + lane; i < hi; i += kWave) {
    const int64_t j = descending ? (hw - 1 - i) : i;
    acc += sal[order[j]];
  }
  acc = wave_sum(acc);
  if (lane == 0) seg[t] = acc;
}

// argmax order: NaN beats every number (torch.max propagates NaN), ties go to the lower index.
__device__ __forceinline__ bool beats(float v, int i, float m, int mi) {
  const bool vn = v != v, mn = m != m;
  if (vn != mn) return vn;
  if (vn) return i < mi;
  return v > m || (v == m && i < mi);
}

// One wave per row of logits.
__global__ __launch_bounds__(kBlock) void softmax_stats_kernel(const float* __restrict__ logits, int B, int K,
                                                               const int32_t* __restrict__ target_dev, int target_host,
                                                               float* __restrict__ p_target, float* __restrict__ entropy,
                                                               int32_t* __restrict__ argmax) {
  const int row = blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
  if (row >= B) return;
  const int lane = threadIdx.x & 63;
  const float* z = logits + static_cast<int64_t>(row) * K;
  float m = -INFINITY;
  int mi = INT32_MAX;
  for (int k = lane; k < K; k += kWave) {
    const float v = z[k];
    if (beats(v, k, m, mi)) { m = v; mi = k; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float om = __shfl_xor(m, off, kWave);
    const int oi = __shfl_xor(mi, off, kWave);
    if (beats(om, oi, m, mi)) { m = om; mi = oi; }
  }
  float sum = 0.f;
  for (int k = lane; k < K; k += kWave) sum += expf(z[k] - m);
  sum = wave_sum(sum);
  int tgt = target_dev ? *target_dev : target_host;
  if (tgt < 0) tgt = mi;
  float ent = 0.f;
  if (entropy) {
    for (int k = lane; k < K; k += kWave) {
      const float p = expf(z[k] - m) / sum;
      ent += p * log2f(p);                       // 0 * -inf = NaN: kept, the reference has it too
    }
    ent = wave_sum(ent);
  }
  if (lane == 0) {
    if (p_target) p_target[row] = (tgt < K) ? expf(z[tgt] - m) / sum : NAN;
    if (entropy) entropy[row] = -ent;
    if (argmax) argmax[row] = mi;
  }
}

}  // namespace

XAI_EXPORT int xai_flip_steps_i32(const int32_t* rank, int64_t hw, int descending, int step_size, int32_t* flip_step,
                                  xai_stream_t stream) {
  XAI_REQUIRE_PTR(rank); XAI_REQUIRE_PTR(flip_step);
  XAI_REQUIRE(hw > 0 && step_size > 0, XAI_E_SHAPE);
  hipLaunchKernelGGL(flip_steps_kernel, dim3(static_cast<unsigned>(xai_ceil_div(hw, kBlock))), dim3(kBlock), 0,
                     static_cast<hipStream_t>(stream), rank, hw, descending, step_size, flip_step);
  return xai_launch_status();
}

XAI_EXPORT int xai_perturb_batch_f32(const float* start, const float* finish, const int32_t* flip_step, int C, int64_t hw,
                                     int first_step, int n_batch, float* out, xai_stream_t stream) {
  XAI_REQUIRE_PTR(start); XAI_REQUIRE_PTR(finish); XAI_REQUIRE_PTR(flip_step); XAI_REQUIRE_PTR(out);
  XAI_REQUIRE(C > 0 && hw > 0 && n_batch > 0 && first_step >= 0, XAI_E_SHAPE);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool vec = (hw % 4 == 0) && xai_aligned16(start) && xai_aligned16(finish) && xai_aligned16(flip_step) && xai_aligned16(out);
  const int64_t tiles = xai_ceil_div(hw, kBlock * (vec ? 4 : 1));
  int per, chunks, zdim = 1;
  if (static_cast<int64_t>(n_batch) * C * hw * 4 >= (int64_t(64) << 20) && C <= 64) {
    per = n_batch >= 2 ? 2 : 1;                  // HBM-sized batch: one channel x two step images per lane
    zdim = C;
  } else {
    const int c0 = static_cast<int>(std::min<int64_t>(n_batch, std::max<int64_t>(1, xai_ceil_div(2048, tiles))));
    per = static_cast<int>(xai_ceil_div(n_batch, c0));
  }
  chunks = static_cast<int>(xai_ceil_div(n_batch, per));
  XAI_REQUIRE(chunks <= 65535, XAI_E_UNSUPPORTED);
  dim3 grid(static_cast<unsigned>(tiles), chunks, zdim);
  if (vec) hipLaunchKernelGGL(perturb_kernel<4>, grid, dim3(kBlock), 0, st, start, finish, flip_step, C, hw, first_step, n_batch, per, out);
  else     hipLaunchKernelGGL(perturb_kernel<1>, grid, dim3(kBlock), 0, st, start, finish, flip_step, C, hw, first_step, n_batch, per, out);
  return xai_launch_status();
}

XAI_EXPORT int xai_segment_sums_f32(const float* sal, const int32_t* order, int64_t hw, int descending, int step_size,
                                    int n_steps, float* seg, float* total, xai_stream_t stream) {
  XAI_REQUIRE_PTR(sal); XAI_REQUIRE_PTR(order); XAI_REQUIRE_PTR(seg); XAI_REQUIRE_PTR(total);
  XAI_REQUIRE(hw > 0 && step_size > 0 && n_steps > 0, XAI_E_SHAPE);
  XAI_REQUIRE(static_cast<int64_t>(n_steps) * step_size >= hw && static_cast<int64_t>(n_steps - 1) * step_size < hw, XAI_E_SHAPE);
  hipLaunchKernelGGL(segment_sums_kernel, dim3((n_steps + 15) / 16 + 1), dim3(1024), 0, static_cast<hipStream_t>(stream), sal,
                     order, hw, descending, step_size, n_steps, seg, total);
  return xai_launch_status();
}

XAI_EXPORT int xai_softmax_stats_f32(const float* logits, int B, int K, const int32_t* target_dev, int target_host,
                                     float* p_target, float* entropy_bits, int32_t* argmax, xai_stream_t stream) {
  XAI_REQUIRE_PTR(logits);
  XAI_REQUIRE(B > 0 && K > 0, XAI_E_SHAPE);
  XAI_REQUIRE(target_dev != nullptr || target_host < K, XAI_E_SHAPE);
  hipLaunchKernelGGL(softmax_stats_kernel, dim3((B + 3) / 4), dim3(kBlock), 0, static_cast<hipStream_t>(stream), logits, B, K,
                     target_dev, target_host, p_target, entropy_bits, argmax);
  return xai_launch_status();
}
