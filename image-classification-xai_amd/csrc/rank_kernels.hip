// K8: stable ascending argsort of a saliency map (one 1024-lane workgroup per map) for gfx950.
//
// LSD radix sort, 4-bit digits, keys + indices ping-pong through an L2-resident scratch (16 B per
// pixel).  Every lane owns a contiguous run of the input; per pass
//   count   : it counts its digits into its private LDS column cnt[digit][lane] (bank = lane % 32,
//             conflict-free), loads issued 8 at a time;
//   scan    : the 16 x 1024 counters are exclusive-scanned digit-major -- 16 independent wave
//             shuffle scans per lane, the 16 x 16 wave totals scanned by one wave: 2 barriers;
//   scatter : the lane scatters its run in order, so equal keys keep their input order
//             (NumPy kind='stable').
// A pass whose digit is the same for every key (sign/exponent nibbles of a non-negative map, for
// instance) is skipped.  Key order is NumPy's: -0.0 == +0.0, NaN sorts last.  A final sweep
// writes `order` and its inverse `rank`.
#include <mutex>
#include "xai_common.h"

namespace {

constexpr int RT = 1024;         // lanes per workgroup
constexpr int RD = 16;           // digits per pass (4 bits)
constexpr int RW = RT / 64;      // waves
constexpr int kPasses = 8;
constexpr size_t kLdsBytes = (RD * RT + RD * RW + RD * RW + 4) * sizeof(uint32_t);

__device__ __forceinline__ uint32_t sort_key(float v) {
  if (v != v) return 0xFFFFFFFFu;                       // NaN last
  uint32_t u = __float_as_uint(v);
  if (u == 0x80000000u) u = 0u;                         // -0.0 ties with +0.0
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(RT) void rank_kernel(const float* __restrict__ sal_all, int64_t hw, int32_t* __restrict__ order_all,
                                                  int32_t* __restrict__ rank_all, uint32_t* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  uint32_t* cnt = lds;                        // [RD][RT]
  uint32_t* wtot = lds + RD * RT;             // [RD][RW] wave totals
  uint32_t* wex = wtot + RD * RW;             // [RD][RW] exclusive prefix of wtot in (digit, wave) order
  uint32_t* flag = wex + RD * RW;             // [0] = 1 when the pass is the identity
  const int seg = blockIdx.x;
  const float* sal = sal_all + seg * hw;
  int32_t* order = order_all + seg * hw;
  int32_t* rank = rank_all + seg * hw;
  uint32_t* kbuf[2] = {ws + static_cast<int64_t>(seg) * 4 * hw, ws + static_cast<int64_t>(seg) * 4 * hw + hw};
  uint32_t* ibuf[2] = {kbuf[1] + hw, kbuf[1] + 2 * hw};
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t ipt = (hw + RT - 1) / RT;
  const int64_t lo = min(static_cast<int64_t>(t) * ipt, hw);
  const int64_t hi = min(lo + ipt, hw);
  const uint32_t n = static_cast<uint32_t>(hw);

  int cur = -1;                               // -1: keys still come from `sal`, indices are the positions
  for (int pass = 0; pass < kPasses; ++pass) {
    const int shift = pass * 4;
    const uint32_t* kin = cur < 0 ? nullptr : kbuf[cur];
    const uint32_t* iin = cur < 0 ? nullptr : ibuf[cur];
    const int nxt = cur < 0 ? 0 : cur ^ 1;
    // ---- count
#pragma unroll
    for (int d = 0; d < RD; ++d) cnt[d * RT + t] = 0;
    {
      int64_t i = lo;
      for (; i + 8 <= hi; i += 8) {
        uint32_t k[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) k[u] = kin ? kin[i + u] : sort_key(sal[i + u]);
#pragma unroll
        for (int u = 0; u < 8; ++u) cnt[((k[u] >> shift) & 15u) * RT + t] += 1;
      }
      for (; i < hi; ++i) {
        const uint32_t k = kin ? kin[i] : sort_key(sal[i]);
        cnt[((k >> shift) & 15u) * RT + t] += 1;
      }
    }
    // ---- scan (digit-major): wave-level inclusive scans of all 16 digits, then the 256 wave totals
    uint32_t c[RD], incl[RD];
#pragma unroll
    for (int d = 0; d < RD; ++d) {
      c[d] = cnt[d * RT + t];
      uint32_t v = c[d];
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(v, off, kWave);
        if (lane >= off) v += up;
      }
      incl[d] = v;
      if (lane == 63) wtot[d * RW + wave] = v;
    }
    __syncthreads();
    if (wave == 0) {                           // 256 totals, 4 per lane, in (digit, wave) order
      uint32_t a[4], s = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) { a[j] = wtot[lane * 4 + j]; s += a[j]; }
      uint32_t v = s;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(v, off, kWave);
        if (lane >= off) v += up;
      }
      uint32_t run = v - s;
#pragma unroll
      for (int j = 0; j < 4; ++j) { wex[lane * 4 + j] = run; run += a[j]; }
      // a digit (= 4 consecutive lanes' worth of totals) holding all n keys makes the pass the identity
      const uint32_t dsum = s + __shfl_xor(s, 1, kWave) + __shfl_xor(s + __shfl_xor(s, 1, kWave), 2, kWave);
      const bool all_in_one = __any(dsum == n);
      if (lane == 0) flag[0] = all_in_one ? 1u : 0u;
    }
    __syncthreads();
    if (flag[0]) continue;                     // uniform: every lane reads the same word after the barrier
#pragma unroll
    for (int d = 0; d < RD; ++d) cnt[d * RT + t] = wex[d * RW + wave] + incl[d] - c[d];
    // ---- stable scatter of this lane's run
    uint32_t* kout = kbuf[nxt];
    uint32_t* iout = ibuf[nxt];
    {
      int64_t i = lo;
      for (; i + 4 <= hi; i += 4) {
        uint32_t k[4], x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          k[u] = kin ? kin[i + u] : sort_key(sal[i + u]);
          x[u] = iin ? iin[i + u] : static_cast<uint32_t>(i + u);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t slot = ((k[u] >> shift) & 15u) * RT + t;
          const uint32_t pos = cnt[slot];
          cnt[slot] = pos + 1;
          kout[pos] = k[u];
          iout[pos] = x[u];
        }
      }
      for (; i < hi; ++i) {
        const uint32_t k = kin ? kin[i] : sort_key(sal[i]);
        const uint32_t x = iin ? iin[i] : static_cast<uint32_t>(i);
        const uint32_t slot = ((k >> shift) & 15u) * RT + t;
        const uint32_t pos = cnt[slot];
        cnt[slot] = pos + 1;
        kout[pos] = k;
        iout[pos] = x;
      }
    }
    cur = nxt;
    __syncthreads();   // workgroup-scope release/acquire of the scratch (same CU, same L1)
  }
  // ---- emit order and its inverse (coalesced sweep)
  const uint32_t* ifin = cur < 0 ? nullptr : ibuf[cur];
  for (int64_t i = t; i < hw; i += RT) {
    const uint32_t x = ifin ? ifin[i] : static_cast<uint32_t>(i);
    order[i] = static_cast<int32_t>(x);
    rank[x] = static_cast<int32_t>(i);
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
