// Does a copy with non-temporal stores leave the Infinity Cache clean enough that the following
// streaming-read kernel runs at full speed?  (bench.py copies each pass's gradients into the
// [img][step] buffer right before xai_ig_accum_f32 reads all 963 MB of it.)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float fx4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 0 plain, 1 nt store, 2 nt load + nt store
__global__ __launch_bounds__(256) void copy_k(const fx4* __restrict__ src, fx4* __restrict__ dst, long n4) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    fx4 v = MODE == 2 ? __builtin_nontemporal_load(src + i) : src[i];
    if (MODE >= 1) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
  }
}

__global__ __launch_bounds__(256) void read_k(const fx4* __restrict__ g, long n4, float* sink) {   // stand-in for the accum kernel
  const long per = (n4 + gridDim.x - 1) / gridDim.x;
  const long lo = blockIdx.x * per, hi = lo + per < n4 ? lo + per : n4;
  fx4 acc = {0, 0, 0, 0};
  for (long i = lo + threadIdx.x; i < hi; i += 256 * 4) {
    fx4 a = __builtin_nontemporal_load(g + i);
    fx4 b = i + 256 < hi ? __builtin_nontemporal_load(g + i + 256) : fx4{0, 0, 0, 0};
    fx4 c = i + 512 < hi ? __builtin_nontemporal_load(g + i + 512) : fx4{0, 0, 0, 0};
    fx4 d = i + 768 < hi ? __builtin_nontemporal_load(g + i + 768) : fx4{0, 0, 0, 0};
    acc += a + b + c + d;
  }
  if (acc.x + acc.y + acc.z + acc.w == 1234.5f) *sink = acc.x;
}

int main() {
  const long n4 = 32L * 50 * 150528 / 4, chunk4 = 2L * 50 * 150528 / 4;
  fx4 *buf, *src; float* sink;
  CK(hipMalloc(&buf, n4 * 16)); CK(hipMalloc(&src, chunk4 * 16)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(buf, 0, n4 * 16)); CK(hipMemset(src, 0, chunk4 * 16));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto run = [&](const char* name, int mode, int chunks) {
    std::vector<float> t, tc;
    for (int it = 0; it < 12; ++it) {
      CK(hipEventRecord(a, 0));
      for (int c = 16 - chunks; c < 16; ++c) {
        if (mode == 0) hipLaunchKernelGGL(copy_k<0>, dim3(2048), dim3(256), 0, 0, src, buf + c * chunk4, chunk4);
        if (mode == 1) hipLaunchKernelGGL(copy_k<1>, dim3(2048), dim3(256), 0, 0, src, buf + c * chunk4, chunk4);
        if (mode == 2) hipLaunchKernelGGL(copy_k<2>, dim3(2048), dim3(256), 0, 0, src, buf + c * chunk4, chunk4);
      }
      CK(hipEventRecord(b, 0));
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(read_k, dim3(512), dim3(256), 0, 0, buf, n4, sink);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
      CK(hipEventElapsedTime(&ms, a, b)); tc.push_back(ms);
    }
    std::sort(t.begin(), t.end()); std::sort(tc.begin(), tc.end());
    printf("%-34s copies %7.1f us   read 963MB %7.1f us  %7.1f GB/s\n", name, tc[6] * 1e3, t[6] * 1e3, n4 * 16.0 / t[6] / 1e6);
  };
  run("no copy before", -1, 0);
  run("plain copy, last chunk", 0, 1); run("nt-store copy, last chunk", 1, 1); run("nt-ld+st copy, last chunk", 2, 1);
  run("plain copy, all 16 chunks", 0, 16); run("nt-store copy, all 16 chunks", 1, 16); run("nt-ld+st copy, all 16 chunks", 2, 16);
  return 0;
}
