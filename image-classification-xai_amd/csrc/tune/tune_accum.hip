// Stand-alone tuning harness for the IG accumulation kernel (not part of libxai_hip.so).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tune_accum.hip -o tune_accum && ./tune_accum
// Sweeps block size / unroll / load policy / work mapping on the BASELINE shape
// (32 images x 50 steps x 3x224x224 fp32 = 963 MB, larger than the 256 MiB Infinity Cache)
// and prints achieved algorithmic GB/s per variant (median of 20 launches, HIP events).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <bool NT> __device__ __forceinline__ float4 ldg(const float* p) {
  if constexpr (NT) {
    float4 v;
    v.x = __builtin_nontemporal_load(p); v.y = __builtin_nontemporal_load(p + 1);
    v.z = __builtin_nontemporal_load(p + 2); v.w = __builtin_nontemporal_load(p + 3);
    return v;
  } else {
    return *reinterpret_cast<const float4*>(p);
  }
}
typedef float fx4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ float4 ldg4(const float4* p) {
  if constexpr (NT) {
    const fx4 v = __builtin_nontemporal_load(reinterpret_cast<const fx4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
  } else {
    return *p;
  }
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// mapping A: lane = (img, p4), loops channels then steps (current production mapping)
template <int BLOCK, int U, bool NT>
__global__ __launch_bounds__(BLOCK) void accum_a(const float* __restrict__ g, int S, const float* __restrict__ x, int C, long hw,
                                                 float* __restrict__ out, float* __restrict__ out_abs) {
  const long p = ((long)blockIdx.x * BLOCK + threadIdx.x) * 4;
  if (p >= hw) return;
  const int img = blockIdx.y;
  const long row = (long)C * hw;
  const float* gi = g + (long)img * S * row + p;
  float4 tot = make_float4(0, 0, 0, 0);
  for (int c = 0; c < C; ++c) {
    const float4* gc = reinterpret_cast<const float4*>(gi + c * hw);
    float4 acc = make_float4(0, 0, 0, 0);
    int s = 0;
    for (; s + U <= S; s += U) {
      float4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = ldg4<NT>(gc + (s + u) * (row / 4));
#pragma unroll
      for (int u = 0; u < U; ++u) acc = add4(acc, v[u]);
    }
    for (; s < S; ++s) acc = add4(acc, ldg4<NT>(gc + s * (row / 4)));
    const long at = ((long)img * C + c) * hw + p;
    const float4 xv = *reinterpret_cast<const float4*>(x + at);
    const float n = (float)S;
    float4 o = make_float4(acc.x / n * xv.x, acc.y / n * xv.y, acc.z / n * xv.z, acc.w / n * xv.w);
    *reinterpret_cast<float4*>(out + at) = o;
    tot = add4(tot, o);
  }
  *reinterpret_cast<float4*>(out_abs + (long)img * hw + p) = make_float4(fabsf(tot.x), fabsf(tot.y), fabsf(tot.z), fabsf(tot.w));
}

// mapping B: lane = (img, c, p4) -- channel in the grid, no fused abs (upper bound for finer tiles)
template <int BLOCK, int U, bool NT>
__global__ __launch_bounds__(BLOCK) void accum_b(const float* __restrict__ g, int S, const float* __restrict__ x, int C, long hw,
                                                 float* __restrict__ out) {
  const long n_el = (long)C * hw;
  const long e = ((long)blockIdx.x * BLOCK + threadIdx.x) * 4;
  if (e >= n_el) return;
  const int img = blockIdx.y;
  const float4* gc = reinterpret_cast<const float4*>(g + (long)img * S * n_el + e);
  float4 acc = make_float4(0, 0, 0, 0);
  int s = 0;
  for (; s + U <= S; s += U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ldg4<NT>(gc + (s + u) * (n_el / 4));
#pragma unroll
    for (int u = 0; u < U; ++u) acc = add4(acc, v[u]);
  }
  for (; s < S; ++s) acc = add4(acc, ldg4<NT>(gc + s * (n_el / 4)));
  const long at = (long)img * n_el + e;
  const float4 xv = *reinterpret_cast<const float4*>(x + at);
  const float n = (float)S;
  *reinterpret_cast<float4*>(out + at) = make_float4(acc.x / n * xv.x, acc.y / n * xv.y, acc.z / n * xv.z, acc.w / n * xv.w);
}

// mapping P: balanced persistent grid -- every workgroup owns an equal contiguous range of
// (img, p4) items; grid = CUs * (2048 / BLOCK) so that every CU holds exactly the same number
// of lanes and the per-CU load rate (the real limiter, ~11 B/clk/CU) is evenly used.
template <int BLOCK, int U, bool NT>
__global__ __launch_bounds__(BLOCK, 2048 / BLOCK * BLOCK / 256) void accum_p(const float* __restrict__ g, int S, const float* __restrict__ x,
                                                                            int C, long hw, int n_img, float* __restrict__ out,
                                                                            float* __restrict__ out_abs) {
  const long hw4 = hw / 4;
  const long items = (long)n_img * hw4;
  const long per = (items + gridDim.x - 1) / gridDim.x;
  const long lo = (long)blockIdx.x * per;
  const long hi = lo + per < items ? lo + per : items;
  const long row = (long)C * hw;
  for (long it = lo + threadIdx.x; it < hi; it += BLOCK) {
    const long img = it / hw4;
    const long p = (it - img * hw4) * 4;
    const float* gi = g + img * S * row + p;
    float4 tot = make_float4(0, 0, 0, 0);
    for (int c = 0; c < C; ++c) {
      const float4* gc = reinterpret_cast<const float4*>(gi + c * hw);
      float4 acc = make_float4(0, 0, 0, 0);
      int s = 0;
      for (; s + U <= S; s += U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ldg4<NT>(gc + (s + u) * (row / 4));
#pragma unroll
        for (int u = 0; u < U; ++u) acc = add4(acc, v[u]);
      }
      for (; s < S; ++s) acc = add4(acc, ldg4<NT>(gc + s * (row / 4)));
      const long at = (img * C + c) * hw + p;
      const float4 xv = *reinterpret_cast<const float4*>(x + at);
      const float n = (float)S;
      float4 o = make_float4(acc.x / n * xv.x, acc.y / n * xv.y, acc.z / n * xv.z, acc.w / n * xv.w);
      *reinterpret_cast<float4*>(out + at) = o;
      tot = add4(tot, o);
    }
    *reinterpret_cast<float4*>(out_abs + img * hw + p) = make_float4(fabsf(tot.x), fabsf(tot.y), fabsf(tot.z), fabsf(tot.w));
  }
}

// mapping Q: balanced grid, step-outer: a lane keeps ITEMS x 3 channel accumulators in registers and
// the whole workgroup walks the step rows together, so at any moment it streams long contiguous
// runs of ONE [img][s] row (DRAM-page friendly), instead of every lane hopping 602 KB per load.
template <int BLOCK, int ITEMS, int SU, bool NT>
__global__ __launch_bounds__(BLOCK) void accum_q(const float* __restrict__ g, int S, const float* __restrict__ x, long hw, int n_img,
                                                 float* __restrict__ out, float* __restrict__ out_abs) {
  constexpr int C = 3;
  const long hw4 = hw / 4;
  const long items = (long)n_img * hw4;
  const long per = (items + gridDim.x - 1) / gridDim.x;
  const long lo = (long)blockIdx.x * per;
  const long hi = lo + per < items ? lo + per : items;
  const long row4 = (long)C * hw4;
  for (long base = lo; base < hi; base += (long)BLOCK * ITEMS) {
    const float4* gp[ITEMS];
    long at[ITEMS];
    bool live[ITEMS];
    float4 acc[ITEMS][C];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      const long it = base + (long)i * BLOCK + threadIdx.x;
      live[i] = it < hi;
      const long itc = live[i] ? it : lo;
      const long img = itc / hw4;
      const long p4 = itc - img * hw4;
      gp[i] = reinterpret_cast<const float4*>(g) + img * S * row4 + p4;
      at[i] = img * row4 + p4;
#pragma unroll
      for (int c = 0; c < C; ++c) acc[i][c] = make_float4(0, 0, 0, 0);
    }
    int s = 0;
    for (; s + SU <= S; s += SU) {
      float4 v[SU][ITEMS][C];
#pragma unroll
      for (int u = 0; u < SU; ++u)
#pragma unroll
        for (int i = 0; i < ITEMS; ++i)
#pragma unroll
          for (int c = 0; c < C; ++c) v[u][i][c] = ldg4<NT>(gp[i] + (s + u) * row4 + c * hw4);
#pragma unroll
      for (int u = 0; u < SU; ++u)
#pragma unroll
        for (int i = 0; i < ITEMS; ++i)
#pragma unroll
          for (int c = 0; c < C; ++c) acc[i][c] = add4(acc[i][c], v[u][i][c]);
    }
    for (; s < S; ++s)
#pragma unroll
      for (int i = 0; i < ITEMS; ++i)
#pragma unroll
        for (int c = 0; c < C; ++c) acc[i][c] = add4(acc[i][c], ldg4<NT>(gp[i] + s * row4 + c * hw4));
    const float n = (float)S;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      if (!live[i]) continue;
      float4 tot = make_float4(0, 0, 0, 0);
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float4 xv = reinterpret_cast<const float4*>(x)[at[i] + c * hw4];
        const float4 a = acc[i][c];
        const float4 o = make_float4(a.x / n * xv.x, a.y / n * xv.y, a.z / n * xv.z, a.w / n * xv.w);
        reinterpret_cast<float4*>(out)[at[i] + c * hw4] = o;
        tot = add4(tot, o);
      }
      const long img = at[i] / row4;
      reinterpret_cast<float4*>(out_abs)[img * hw4 + (at[i] - img * row4)] = make_float4(fabsf(tot.x), fabsf(tot.y), fabsf(tot.z), fabsf(tot.w));
    }
  }
}

// C: pure streaming read of the whole buffer (grid-stride), the read-bandwidth ceiling
template <int BLOCK, int U, bool NT>
__global__ __launch_bounds__(BLOCK) void read_all(const float4* __restrict__ g, long n4, float* __restrict__ sink) {
  const long stride = (long)gridDim.x * BLOCK;
  long i = (long)blockIdx.x * BLOCK + threadIdx.x;
  float4 acc = make_float4(0, 0, 0, 0);
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ldg4<NT>(g + i + u * stride);
#pragma unroll
    for (int u = 0; u < U; ++u) acc = add4(acc, v[u]);
  }
  for (; i < n4; i += stride) acc = add4(acc, ldg4<NT>(g + i));
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

template <typename F> double time_ms(F&& launch, int iters = 20) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) launch();
  std::vector<float> t;
  for (int i = 0; i < iters; ++i) {
    CK(hipEventRecord(a, 0)); launch(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms);
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

int main() {
  const int B = 32, S = 50, C = 3; const long hw = 224 * 224, N = C * hw;
  float *g, *x, *out, *out_abs;
  CK(hipMalloc(&g, sizeof(float) * B * S * N)); CK(hipMalloc(&x, sizeof(float) * B * N));
  CK(hipMalloc(&out, sizeof(float) * B * N)); CK(hipMalloc(&out_abs, sizeof(float) * B * hw));
  CK(hipMemset(g, 0x3c, sizeof(float) * B * S * N)); CK(hipMemset(x, 0x3c, sizeof(float) * B * N));
  const double bytes = (double)B * (S + 2) * 4 * N + (double)B * hw * 4;
  auto report = [&](const char* name, double ms) { printf("%-44s %8.1f us  %7.1f GB/s  frac(8TB/s)=%.3f\n", name, ms * 1e3, bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0); };

#define RUN_A(BLOCK, U, NT) report("A block=" #BLOCK " U=" #U " nt=" #NT, time_ms([&] { \
    hipLaunchKernelGGL((accum_a<BLOCK, U, NT>), dim3((hw / 4 + BLOCK - 1) / BLOCK, B), dim3(BLOCK), 0, 0, g, S, x, C, hw, out, out_abs); }))
#define RUN_B(BLOCK, U, NT) report("B block=" #BLOCK " U=" #U " nt=" #NT, time_ms([&] { \
    hipLaunchKernelGGL((accum_b<BLOCK, U, NT>), dim3((N / 4 + BLOCK - 1) / BLOCK, B), dim3(BLOCK), 0, 0, g, S, x, C, hw, out); }))
#define RUN_C(BLOCK, U, NT, GRID) report("C read-all block=" #BLOCK " U=" #U " nt=" #NT " grid=" #GRID, time_ms([&] { \
    hipLaunchKernelGGL((read_all<BLOCK, U, NT>), dim3(GRID), dim3(BLOCK), 0, 0, (const float4*)g, (long)B * S * N / 4, out); }))

#define RUN_P(BLOCK, U, NT, GRID) report("P block=" #BLOCK " U=" #U " nt=" #NT " grid=" #GRID, time_ms([&] { \
    hipLaunchKernelGGL((accum_p<BLOCK, U, NT>), dim3(GRID), dim3(BLOCK), 0, 0, g, S, x, C, hw, B, out, out_abs); }))
#define RUN_Q(BLOCK, ITEMS, SU, NT, GRID) report("Q block=" #BLOCK " items=" #ITEMS " SU=" #SU " nt=" #NT " grid=" #GRID, time_ms([&] { \
    hipLaunchKernelGGL((accum_q<BLOCK, ITEMS, SU, NT>), dim3(GRID), dim3(BLOCK), 0, 0, g, S, x, hw, B, out, out_abs); }))
  RUN_Q(1024, 1, 1, true, 512); RUN_Q(1024, 1, 2, true, 512); RUN_Q(1024, 2, 1, true, 256); RUN_Q(1024, 2, 2, true, 256);
  RUN_Q(512, 1, 2, true, 1024); RUN_Q(512, 2, 1, true, 512); RUN_Q(512, 2, 2, true, 512); RUN_Q(512, 4, 1, true, 256);
  RUN_Q(256, 2, 2, true, 1024); RUN_Q(256, 4, 1, true, 512); RUN_Q(256, 1, 2, true, 2048); RUN_Q(1024, 1, 1, false, 512);
  RUN_Q(1024, 1, 5, true, 512); RUN_Q(512, 1, 5, true, 1024); RUN_Q(256, 1, 5, true, 2048); RUN_Q(256, 8, 1, true, 256);
  RUN_P(1024, 4, true, 512); RUN_P(1024, 8, true, 512); RUN_P(1024, 2, true, 512); RUN_P(1024, 5, true, 512); RUN_P(1024, 10, true, 512);
  RUN_P(512, 4, true, 1024); RUN_P(512, 8, true, 1024); RUN_P(256, 8, true, 2048); RUN_P(256, 4, true, 2048);
  RUN_P(1024, 4, false, 512); RUN_P(1024, 8, true, 256); RUN_P(1024, 5, true, 1024); RUN_P(512, 5, true, 2048);
  RUN_A(256, 8, false); RUN_A(256, 8, true); RUN_A(256, 4, false); RUN_A(256, 16, false); RUN_A(256, 16, true);
  RUN_A(128, 8, false); RUN_A(128, 8, true); RUN_A(64, 8, false); RUN_A(64, 8, true); RUN_A(512, 8, false); RUN_A(512, 8, true);
  RUN_A(64, 16, true); RUN_A(128, 16, true); RUN_A(64, 25, true); RUN_A(128, 25, false); RUN_A(256, 25, false); RUN_A(256, 10, false);
  RUN_B(256, 8, false); RUN_B(256, 8, true); RUN_B(128, 8, true); RUN_B(64, 8, true); RUN_B(256, 16, true); RUN_B(256, 25, false); RUN_B(256, 10, true);
  RUN_C(256, 8, false, 2048); RUN_C(256, 8, true, 2048); RUN_C(256, 8, true, 4096); RUN_C(256, 4, true, 8192); RUN_C(512, 8, true, 1024);
  RUN_C(256, 16, true, 2048); RUN_C(256, 8, true, 1024); RUN_C(1024, 4, true, 512);
  return 0;
}
