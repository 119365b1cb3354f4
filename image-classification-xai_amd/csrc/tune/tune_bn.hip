// Mapping study for the classifier-fusion element-wise kernels: y = relu(fma((x - m) * is, w, b)) over 100 x 256 x 56 x 56
// (321 MB in, 321 MB out).  One float4 per lane vs ITEMS per lane vs a persistent balanced grid; nt loads / stores.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float fx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ fx4 bn4(fx4 v, float m, float is, float w, float b) {
  fx4 o;
  o.x = fmaxf(__builtin_fmaf((v.x - m) * is, w, b), 0.f); o.y = fmaxf(__builtin_fmaf((v.y - m) * is, w, b), 0.f);
  o.z = fmaxf(__builtin_fmaf((v.z - m) * is, w, b), 0.f); o.w = fmaxf(__builtin_fmaf((v.w - m) * is, w, b), 0.f);
  return o;
}

template <int ITEMS, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_items(const fx4* __restrict__ x, const float* __restrict__ p, int C, int HW4, long n4, fx4* __restrict__ y) {
  const long base = (long)blockIdx.x * 256 * ITEMS + threadIdx.x;
  fx4 v[ITEMS]; int c[ITEMS];
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const long i = base + k * 256;
    if (i < n4) { v[k] = NTL ? __builtin_nontemporal_load(x + i) : x[i]; c[k] = (int)((i / HW4) % C); }
  }
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const long i = base + k * 256;
    if (i < n4) {
      const fx4 o = bn4(v[k], p[c[k]], rsqrtf(p[C + c[k]] + 1e-5f), p[2 * C + c[k]], p[3 * C + c[k]]);
      if (NTS) __builtin_nontemporal_store(o, y + i); else y[i] = o;
    }
  }
}

template <int ITEMS, bool NTL>
__global__ __launch_bounds__(256) void k_persist(const fx4* __restrict__ x, const float* __restrict__ p, int C, int HW4, long n4, fx4* __restrict__ y) {
  const long per = (n4 + gridDim.x - 1) / gridDim.x;
  const long lo = blockIdx.x * per, hi = lo + per < n4 ? lo + per : n4;
  for (long first = lo; first < hi; first += 256 * ITEMS) {
    fx4 v[ITEMS]; int c[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const long i = first + k * 256 + threadIdx.x;
      if (i < hi) { v[k] = NTL ? __builtin_nontemporal_load(x + i) : x[i]; c[k] = (int)((i / HW4) % C); }
    }
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const long i = first + k * 256 + threadIdx.x;
      if (i < hi) y[i] = bn4(v[k], p[c[k]], rsqrtf(p[C + c[k]] + 1e-5f), p[2 * C + c[k]], p[3 * C + c[k]]);
    }
  }
}

template <typename F> double time_us(F&& launch) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) launch();
  std::vector<float> t;
  for (int i = 0; i < 11; ++i) { CK(hipEventRecord(a, 0)); for (int j = 0; j < 5; ++j) launch(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms / 5); }
  std::sort(t.begin(), t.end()); return t[5] * 1e3;
}

int main() {
  const int N = 100, C = 256, HW = 56 * 56;
  const long n4 = (long)N * C * HW / 4;
  fx4 *x, *y; float* p;
  CK(hipMalloc(&x, n4 * 16)); CK(hipMalloc(&y, n4 * 16)); CK(hipMalloc(&p, 4 * C * 4));
  CK(hipMemset(x, 0, n4 * 16));
  std::vector<float> hp(4 * C, 1.f); CK(hipMemcpy(p, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
  const double mb = 2.0 * n4 * 16 / 1e6;
  int cus = 256; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  auto rep = [&](const char* name, double us) { printf("%-40s %8.1f us  %7.1f GB/s\n", name, us, mb / us * 1e3); };
#define RUN_ITEMS(I, L, S) rep("items " #I " ntl=" #L " nts=" #S, time_us([&] { hipLaunchKernelGGL((k_items<I, L, S>), dim3((unsigned)((n4 + 256 * I - 1) / (256 * I))), dim3(256), 0, 0, x, p, C, HW / 4, n4, y); }))
  RUN_ITEMS(1, false, false); RUN_ITEMS(1, true, false); RUN_ITEMS(1, true, true);
  RUN_ITEMS(2, true, false); RUN_ITEMS(4, true, false); RUN_ITEMS(8, true, false);
#define RUN_P(I, G) rep("persistent items " #I " grid " #G "xCUs", time_us([&] { hipLaunchKernelGGL((k_persist<I, true>), dim3(G * cus), dim3(256), 0, 0, x, p, C, HW / 4, n4, y); }))
  RUN_P(4, 2); RUN_P(4, 4); RUN_P(4, 8); RUN_P(8, 2); RUN_P(2, 8);
  return 0;
}
