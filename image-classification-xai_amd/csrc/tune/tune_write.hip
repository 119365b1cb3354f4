// Write-side ceiling and mapping study for the write-bound kernels (K1 interp, K6 perturb, K4 rise_apply):
// 32 images x 50 rows x 150528 floats (963 MB) written from a register-resident source.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float fx4 __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ void st(fx4* p, fx4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// A: K1's current mapping -- lane = float4 column, loops its step chunk (stride N between stores)
template <bool NT>
__global__ __launch_bounds__(256) void w_lane_strided(const fx4* __restrict__ x, fx4* __restrict__ out, long n4, int S, int per) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n4) return;
  const int img = blockIdx.z, s0 = blockIdx.y * per, s1 = min(s0 + per, S);
  const fx4 v = x[img * n4 + e];
  fx4* o = out + ((long)img * S + s0) * n4 + e;
  for (int s = s0; s < s1; ++s, o += n4) st<NT>(o, v * (float)s);
}
// B: balanced grid, every workgroup owns a contiguous range of OUTPUT rows' columns and walks steps together
template <int BLOCK, int ITEMS, bool NT>
__global__ __launch_bounds__(BLOCK) void w_stream(const fx4* __restrict__ x, fx4* __restrict__ out, long n4, int S, int n_img) {
  const long items = (long)n_img * n4;
  const long per = (items + gridDim.x - 1) / gridDim.x;
  const long lo = blockIdx.x * per, hi = lo + per < items ? lo + per : items;
  for (long first = lo; first < hi; first += (long)BLOCK * ITEMS) {
    fx4 v[ITEMS]; fx4* o[ITEMS]; bool live[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      const long it = first + (long)i * BLOCK + threadIdx.x;
      live[i] = it < hi;
      const long itc = live[i] ? it : lo;
      const long img = itc / n4, e = itc - img * n4;
      v[i] = x[itc];
      o[i] = out + img * S * n4 + e;
    }
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int i = 0; i < ITEMS; ++i)
        if (live[i]) st<NT>(o[i] + (long)s * n4, v[i] * (float)s);
  }
}
// C: pure contiguous fill (ceiling)
template <bool NT>
__global__ __launch_bounds__(256) void w_fill(fx4* __restrict__ out, long n4) {
  const long stride = (long)gridDim.x * 256;
  const fx4 v = {1, 2, 3, 4};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) st<NT>(out + i, v);
}

template <typename F> double time_ms(F&& launch) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) launch();
  std::vector<float> t;
  for (int i = 0; i < 15; ++i) { CK(hipEventRecord(a, 0)); launch(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms); }
  std::sort(t.begin(), t.end()); return t[7];
}

int main() {
  const int B = 32, S = 50; const long n4 = 150528 / 4;
  fx4 *x, *out; CK(hipMalloc(&x, B * n4 * 16)); CK(hipMalloc(&out, (long)B * S * n4 * 16)); CK(hipMemset(x, 0, B * n4 * 16));
  const double bytes = (double)B * (S + 1) * n4 * 16;
  auto rep = [&](const char* n, double ms) { printf("%-46s %8.1f us %8.1f GB/s  frac=%.3f\n", n, ms * 1e3, bytes / ms / 1e6, bytes / ms / 8e9); };
  rep("A lane-strided per=10 plain", time_ms([&] { hipLaunchKernelGGL(w_lane_strided<false>, dim3(147, 5, B), dim3(256), 0, 0, x, out, n4, S, 10); }));
  rep("A lane-strided per=10 nt", time_ms([&] { hipLaunchKernelGGL(w_lane_strided<true>, dim3(147, 5, B), dim3(256), 0, 0, x, out, n4, S, 10); }));
  rep("A lane-strided per=50 plain", time_ms([&] { hipLaunchKernelGGL(w_lane_strided<false>, dim3(147, 1, B), dim3(256), 0, 0, x, out, n4, S, 50); }));
  rep("A lane-strided per=2 plain", time_ms([&] { hipLaunchKernelGGL(w_lane_strided<false>, dim3(147, 25, B), dim3(256), 0, 0, x, out, n4, S, 2); }));
#define RB(BLOCK, ITEMS, NT, GRID) rep("B stream block=" #BLOCK " items=" #ITEMS " nt=" #NT " grid=" #GRID, time_ms([&] { hipLaunchKernelGGL((w_stream<BLOCK, ITEMS, NT>), dim3(GRID), dim3(BLOCK), 0, 0, x, out, n4, S, B); }))
  RB(256, 4, false, 512); RB(256, 4, true, 512); RB(256, 2, false, 1024); RB(256, 8, false, 256); RB(512, 2, false, 512); RB(1024, 1, false, 512);
  RB(256, 1, false, 2048); RB(256, 4, false, 1024); RB(256, 4, true, 1024); RB(1024, 1, true, 512); RB(256, 8, true, 256);
  rep("C fill plain grid=2048", time_ms([&] { hipLaunchKernelGGL(w_fill<false>, dim3(2048), dim3(256), 0, 0, out, (long)B * S * n4); }));
  rep("C fill nt grid=2048", time_ms([&] { hipLaunchKernelGGL(w_fill<true>, dim3(2048), dim3(256), 0, 0, out, (long)B * S * n4); }));
  rep("C fill plain grid=512", time_ms([&] { hipLaunchKernelGGL(w_fill<false>, dim3(512), dim3(256), 0, 0, out, (long)B * S * n4); }));
  rep("C fill nt grid=8192", time_ms([&] { hipLaunchKernelGGL(w_fill<true>, dim3(8192), dim3(256), 0, 0, out, (long)B * S * n4); }));
  return 0;
}
