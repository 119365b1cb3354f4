// Does a hipMemsetAsync captured into a hipGraph take effect on every replay?  (rank_kernels.hip zero-fills its histograms
// with a kernel because, captured through torch.cuda.graph, a memset node appeared not to.)
//   hipcc -O2 --offload-arch=gfx950 repro_graph_memset.hip -o repro_graph_memset && ./repro_graph_memset
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void add_one(int* p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] += 1;
}

static int run(hipStreamCaptureMode mode, const char* name, size_t n) {
  int* buf;
  CK(hipMalloc(&buf, n * sizeof(int)));
  CK(hipMemset(buf, 0, n * sizeof(int)));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, mode));
  CK(hipMemsetAsync(buf, 0, n * sizeof(int), s));
  hipLaunchKernelGGL(add_one, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, buf, (int)n);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int rep = 1; rep <= 3; ++rep) {
    CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    int first = -1, last = -1;
    CK(hipMemcpy(&first, buf, sizeof(int), hipMemcpyDeviceToHost));
    CK(hipMemcpy(&last, buf + n - 1, sizeof(int), hipMemcpyDeviceToHost));
    printf("%-28s n=%-9zu replay %d: buf[0]=%d buf[n-1]=%d  (%s)\n", name, n, rep, first, last,
           first == 1 && last == 1 ? "memset replayed" : "memset NOT replayed");
  }
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(s)); CK(hipFree(buf));
  return 0;
}

int main() {
  for (size_t n : {(size_t)256, (size_t)2048 * 256, (size_t)64 << 20})
    for (auto m : {hipStreamCaptureModeGlobal, hipStreamCaptureModeThreadLocal, hipStreamCaptureModeRelaxed})
      if (run(m, m == hipStreamCaptureModeGlobal ? "capture mode global" : m == hipStreamCaptureModeThreadLocal ? "capture mode thread-local" : "capture mode relaxed", n)) return 1;
  return 0;
}
