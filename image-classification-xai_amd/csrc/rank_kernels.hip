// K8: stable ascending argsort of a saliency map (one 1024-lane workgroup per map) for gfx950.
//
// LSD radix sort, 4-bit digits, 8 passes, keys + indices ping-pong through an L2-resident
// scratch (16 B per pixel).  Every lane owns a contiguous run of the input; per pass it counts
// its digits into its private LDS column cnt[digit][lane] (bank = lane % 32, conflict-free),
// the 16 x 1024 counters are exclusive-scanned digit-major with wave shuffles, and the lane
// scatters its run in order -- so equal keys keep their input order (NumPy kind='stable').
// Key order is NumPy's: -0.0 == +0.0, NaN sorts last.  The last pass writes `order` and its
// inverse `rank` directly.
#include <mutex>
#include "xai_common.h"

namespace {

constexpr int RT = 1024;         // lanes per workgroup
constexpr int RD = 16;           // digits per pass (4 bits)
constexpr int kPasses = 8;
constexpr size_t kLdsBytes = (RD * RT + 16) * sizeof(uint32_t);

__device__ __forceinline__ uint32_t sort_key(float v) {
  if (v != v) return 0xFFFFFFFFu;                       // NaN last
  uint32_t u = __float_as_uint(v);
  if (u == 0x80000000u) u = 0u;                         // -0.0 ties with +0.0
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(RT) void rank_kernel(const float* __restrict__ sal_all, int64_t hw, int32_t* __restrict__ order_all,
                                                  int32_t* __restrict__ rank_all, uint32_t* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  uint32_t* cnt = lds;              // [RD][RT]
  uint32_t* wtot = lds + RD * RT;   // [16]
  const int seg = blockIdx.x;
  const float* sal = sal_all + seg * hw;
  int32_t* order = order_all + seg * hw;
  int32_t* rank = rank_all + seg * hw;
  uint32_t* K0 = ws + static_cast<int64_t>(seg) * 4 * hw;
  uint32_t* K1 = K0 + hw;
  uint32_t* I0 = K1 + hw;
  uint32_t* I1 = I0 + hw;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t ipt = (hw + RT - 1) / RT;
  const int64_t lo = min(static_cast<int64_t>(t) * ipt, hw);
  const int64_t hi = min(lo + ipt, hw);

  for (int pass = 0; pass < kPasses; ++pass) {
    const int shift = pass * 4;
    const uint32_t* kin = (pass & 1) ? K1 : K0;
    const uint32_t* iin = (pass & 1) ? I1 : I0;
    uint32_t* kout = (pass & 1) ? K0 : K1;
    uint32_t* iout = (pass & 1) ? I0 : I1;
#pragma unroll
    for (int d = 0; d < RD; ++d) cnt[d * RT + t] = 0;
    for (int64_t i = lo; i < hi; ++i) {
      const uint32_t key = pass == 0 ? sort_key(sal[i]) : kin[i];
      cnt[((key >> shift) & 15u) * RT + t] += 1;
    }
    // exclusive scan of cnt in (digit, lane) order
    uint32_t base = 0;
    for (int d = 0; d < RD; ++d) {
      const uint32_t v = cnt[d * RT + t];
      uint32_t incl = v;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t n = __shfl_up(incl, off, kWave);
        if (lane >= off) incl += n;
      }
      if (lane == 63) wtot[wave] = incl;
      __syncthreads();
      uint32_t before = 0, total = 0;
#pragma unroll
      for (int w = 0; w < 16; ++w) {
        const uint32_t x = wtot[w];
        before += (w < wave) ? x : 0u;
        total += x;
      }
      cnt[d * RT + t] = base + before + incl - v;
      base += total;
      __syncthreads();
    }
    // stable scatter of this lane's run
    for (int64_t i = lo; i < hi; ++i) {
      const uint32_t key = pass == 0 ? sort_key(sal[i]) : kin[i];
      const uint32_t idx = pass == 0 ? static_cast<uint32_t>(i) : iin[i];
      const uint32_t slot = ((key >> shift) & 15u) * RT + t;
      const uint32_t pos = cnt[slot];
      cnt[slot] = pos + 1;
      if (pass == kPasses - 1) {
        order[pos] = static_cast<int32_t>(idx);
        rank[idx] = static_cast<int32_t>(pos);
      } else {
        kout[pos] = key;
        iout[pos] = idx;
      }
    }
    __syncthreads();   // workgroup-scope release/acquire of the scratch (same CU, same L1)
  }
}

std::once_flag g_attr_once;
int g_attr_status = 0;

}  // namespace

XAI_EXPORT size_t xai_rank_workspace_bytes(int n_seg, int64_t hw) {
  if (n_seg <= 0 || hw <= 0) return 0;
  return static_cast<size_t>(n_seg) * static_cast<size_t>(hw) * 4u * sizeof(uint32_t);
}

XAI_EXPORT int xai_rank_f32(const float* sal, int n_seg, int64_t hw, int32_t* order, int32_t* rank, void* ws, size_t ws_bytes,
                            xai_stream_t stream) {
  XAI_REQUIRE_PTR(sal); XAI_REQUIRE_PTR(order); XAI_REQUIRE_PTR(rank); XAI_REQUIRE_PTR(ws);
  XAI_REQUIRE(n_seg > 0 && hw > 0, XAI_E_SHAPE);
  XAI_REQUIRE(hw < (int64_t(1) << 31), XAI_E_UNSUPPORTED);
  XAI_REQUIRE(ws_bytes >= xai_rank_workspace_bytes(n_seg, hw), XAI_E_SHAPE);
  std::call_once(g_attr_once, [] {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rank_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(kLdsBytes));
    g_attr_status = (e == hipSuccess) ? 0 : static_cast<int>(e);
  });
  if (g_attr_status) return g_attr_status;
  hipLaunchKernelGGL(rank_kernel, dim3(n_seg), dim3(RT), kLdsBytes, static_cast<hipStream_t>(stream), sal, hw, order, rank,
                     static_cast<uint32_t*>(ws));
  return xai_launch_status();
}
