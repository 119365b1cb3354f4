// Third stage of the captured-memset investigation: which memset nodes does the roc-7.0.2 HIP runtime (the libamdhip64 the
// PyTorch 2.10+rocm7.0 wheel bundles and every torch process therefore runs on) fail to re-execute after the first launch?
// One graph = hipMemsetAsync(buf + offset, 0, bytes) followed by a kernel that adds 1 to every byte-quad; before each of 4
// launches the region is filled with 0xAB.  A launch is right when every word reads 1 afterwards.
//   hipcc -O2 --offload-arch=gfx950 repro_graph_memset3.hip -o repro_graph_memset3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void add_one(unsigned* p, size_t n) {
  const size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
  if (i < n) p[i] += 1u;
}

int main() {
  int rt = 0; CK(hipRuntimeGetVersion(&rt));
  printf("hipRuntimeGetVersion = %d\n", rt);
  hipStream_t s; CK(hipStreamCreate(&s));
  const size_t sizes[] = {32, 1024, 4096, 4100, 4128, 8192, 65536, 65568, 200704, 200736, 1 << 20, (1 << 20) + 32};
  const size_t offs[] = {0, 512, 4096};
  for (size_t bytes : sizes)
    for (size_t off : offs) {
      char* base; CK(hipMalloc(&base, bytes + off + 4096));
      unsigned* p = reinterpret_cast<unsigned*>(base + off);
      const size_t n = bytes / 4;
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
      CK(hipMemsetAsync(p, 0, bytes, s));
      hipLaunchKernelGGL(add_one, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n);
      CK(hipStreamEndCapture(s, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      std::vector<unsigned> h(n);
      char verdict[5] = "....";
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemset(p, 0xAB, bytes)); CK(hipDeviceSynchronize());
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h.data(), p, bytes, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (unsigned v : h) bad += v != 1u;
        verdict[rep] = bad == 0 ? 'k' : (bad == n ? 'X' : 'x');
      }
      printf("memset %8zu B at base+%-5zu launches 1-4: %s   (k = zero-fill took effect, X = no word zeroed, x = some words not zeroed)\n", bytes, off, verdict);
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipFree(base));
    }
  return 0;
}
