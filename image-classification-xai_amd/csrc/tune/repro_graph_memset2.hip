// Second stage of the captured-memset investigation.  profiles/experiments/exp_graph_memset.py showed
// that the hipMemsetAsync variant of xai_rank_f32 gives wrong sorts on EVERY replay through torch.cuda.graph whose
// workspace was dirtied, whatever the capture mode and wherever the workspace was allocated, while the first-stage
// stand-alone repro (repro_graph_memset.hip: memset + one plain kernel) replays correctly.  This program takes torch out:
// it links the very same sort (rank_kernels.hip built with -DXAI_RANK_ZERO_WITH_MEMSET) and drives it through plain HIP
// stream capture, varying what torch does differently from the first repro:
//   instantiate : hipGraphInstantiate(...)  |  hipGraphInstantiateWithFlags(AutoFreeOnLaunch)  (torch's call)
//   launch on   : the capture stream  |  another stream  |  the null stream
//   memset size : the sort's own front words (4128 B at hw = 1024; 200 736 B at hw = 50 176)
//   workspace   : its own hipMalloc  |  an interior pointer of a larger allocation (what a caching allocator hands out)
//   streams     : blocking  |  hipStreamNonBlocking (torch's pool streams)
// Every replay gets a fresh map; before odd replays the workspace is filled with 0xAB, so only a replayed zero-fill can
// make the sort right.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -DXAI_RANK_ZERO_WITH_MEMSET -I../../../include -I.. \
//         repro_graph_memset2.hip ../rank_kernels.hip ../abi.hip -o repro_graph_memset2
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <numeric>
#include <vector>
#include "xai_hip.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return -1; } } while (0)

static int trial(int64_t hw, bool auto_free, int launch_on, size_t ws_offset, bool nonblocking) {
  const size_t ws_bytes = xai_rank_workspace_bytes(1, hw);
  float* sal; int32_t *order, *rank; void* ws;
  // ws_offset != 0: the workspace is an INTERIOR pointer of a larger allocation, as every tensor of a caching allocator is
  void* ws_base;
  CK(hipMalloc(&sal, hw * 4)); CK(hipMalloc(&order, hw * 4)); CK(hipMalloc(&rank, hw * 4)); CK(hipMalloc(&ws_base, ws_bytes + ws_offset));
  CK(hipMemset(ws_base, 0xCD, ws_bytes + ws_offset));
  ws = static_cast<char*>(ws_base) + ws_offset;
  hipStream_t cap, other;
  if (nonblocking) { CK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&other, hipStreamNonBlocking)); }
  else { CK(hipStreamCreate(&cap)); CK(hipStreamCreate(&other)); }
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(cap, hipStreamCaptureModeGlobal));
  if (xai_rank_f32(sal, 1, hw, order, rank, ws, ws_bytes, cap) != 0) { printf("xai_rank_f32 failed\n"); return -1; }
  CK(hipStreamEndCapture(cap, &g));
  size_t n_nodes = 0;
  CK(hipGraphGetNodes(g, nullptr, &n_nodes));
  std::vector<hipGraphNode_t> nodes(n_nodes);
  CK(hipGraphGetNodes(g, nodes.data(), &n_nodes));
  int n_memset = 0, n_kernel = 0;
  for (auto nd : nodes) { hipGraphNodeType t; CK(hipGraphNodeGetType(nd, &t)); n_memset += t == hipGraphNodeTypeMemset; n_kernel += t == hipGraphNodeTypeKernel; }
  size_t n_edges = 0;
  CK(hipGraphGetEdges(g, nullptr, nullptr, &n_edges));
  if (auto_free) CK(hipGraphInstantiateWithFlags(&ge, g, hipGraphInstantiateFlagAutoFreeOnLaunch));
  else CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipStream_t ls = launch_on == 0 ? cap : launch_on == 1 ? other : nullptr;
  int wrong = 0;
  size_t canary_zeroed = 0;
  std::vector<float> h(hw); std::vector<int32_t> got(hw), want(hw);
  uint32_t st = 777u + static_cast<uint32_t>(hw);
  for (int rep = 0; rep < 6; ++rep) {
    for (auto& v : h) { st = st * 1664525u + 1013904223u; v = static_cast<float>(static_cast<int32_t>(st >> 8) - (1 << 23)) / 1024.f; }
    CK(hipMemcpy(sal, h.data(), hw * 4, hipMemcpyHostToDevice));
    if (rep & 1) CK(hipMemset(ws, 0xAB, ws_bytes));
    if (ws_offset) CK(hipMemset(ws_base, 0xCD, ws_offset));           // canary in front of the workspace
    CK(hipDeviceSynchronize());
    CK(hipGraphLaunch(ge, ls));
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(got.data(), order, hw * 4, hipMemcpyDeviceToHost));
    std::iota(want.begin(), want.end(), 0);
    std::stable_sort(want.begin(), want.end(), [&](int a, int b) { return h[a] < h[b]; });
    wrong += got != want;
    if (ws_offset) {                                                   // did the replayed memset land at the allocation's base instead?
      std::vector<unsigned char> front(ws_offset);
      CK(hipMemcpy(front.data(), ws_base, ws_offset, hipMemcpyDeviceToHost));
      size_t zeroed = 0;
      for (unsigned char b : front) zeroed += b == 0;
      canary_zeroed = std::max(canary_zeroed, zeroed);
    }
  }
  printf("hw=%-6lld nodes: %d memset + %d kernel, %zu edges | instantiate=%-17s launch on %-14s %s ws at base+%-6zu: %d of 6 replays wrong; %zu canary bytes before the workspace zeroed\n", (long long)hw, n_memset,
         n_kernel, n_edges, auto_free ? "AutoFreeOnLaunch" : "plain", launch_on == 0 ? "capture stream" : launch_on == 1 ? "other stream" : "null stream",
         nonblocking ? "(non-blocking streams)" : "(blocking streams)    ", ws_offset, wrong, canary_zeroed);
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(cap)); CK(hipStreamDestroy(other));
  CK(hipFree(sal)); CK(hipFree(order)); CK(hipFree(rank)); CK(hipFree(ws_base));
  return wrong;
}

int main() {
#ifdef XAI_RANK_ZERO_WITH_MEMSET
  printf("zero-fill = hipMemsetAsync\n");
#else
  printf("zero-fill = rank_zero_kernel\n");
#endif
  for (int64_t hw : {int64_t(1024), int64_t(50176)})
    for (int af = 0; af < 2; ++af)
      for (int lo = 0; lo < 3; ++lo)
        if (trial(hw, af, lo, 0, false) < 0) return 1;
  // what a caching allocator adds: interior pointers, non-blocking streams
  for (int64_t hw : {int64_t(1024), int64_t(50176)})
    for (size_t off : {size_t(512), size_t(1) << 20})
      for (int nb = 0; nb < 2; ++nb)
        if (trial(hw, true, 2, off, nb) < 0) return 1;
  return 0;
}
