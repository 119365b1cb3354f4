// K8: stable ascending argsort of saliency maps for gfx950 -- multi-workgroup LSD radix sort.
//
// 8-bit digits, 4 passes over (key, index) pairs that ping-pong through an L2-resident scratch.
// Work is tiled by 1024 keys (grid = tiles x maps, so one 224x224 map already spreads over 49 CUs
// and a batch of maps over the whole chip); a sort is 10 launches (zero-fill, hist, 4 x (scan, scatter)):
//   hist    : (pass 0 only) per-tile digit histogram in LDS (ds_add) -> hist[0][tile][digit];
//             the histograms of passes 1..3 are accumulated by the previous pass's scatter with
//             global atomics on the DESTINATION tile (counts do not depend on arrival order)
//   scan    : one 256-lane workgroup per map: lane = digit, running sum over tiles, wave-shuffle
//             exclusive scan over digits -> offs[tile][digit]; flags an identity pass (every key in
//             one bin, e.g. the sign/exponent byte of a non-negative map)
//   scatter : each wave owns 256 consecutive keys, 4 rounds of 64 (coalesced dword loads); the
//             stable rank inside a round comes from 8 ballots (lanes with the same digit),
//             across rounds from a per-wave LDS counter row, across waves from a 4-row prefix;
//             position = offs[digit][tile] + rank.  Equal keys keep their input order
//             (NumPy kind='stable').  The last pass writes `order` and its inverse `rank` directly.
// Key order is NumPy's: -0.0 == +0.0, NaN sorts last.
#include "xai_common.h"

namespace {

constexpr int kTile = 1024;      // keys per workgroup
constexpr int kBlock = 256;      // lanes per workgroup (4 waves x 4 rounds x 64 keys)
constexpr int kBins = 256;

__device__ __forceinline__ uint32_t sort_key(float v) {
  if (v != v) return 0xFFFFFFFFu;                       // NaN last
  uint32_t u = __float_as_uint(v);
  if (u == 0x80000000u) u = 0u;                         // -0.0 ties with +0.0
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct SegPtrs {
  uint32_t* key[2];
  uint32_t* idx[2];
  uint32_t* hist[4]; // [pass][n_tiles][kBins]
  uint32_t* offs;    // [n_tiles][kBins]
  uint32_t* flag;    // [4] identity flag per pass
};

// scratch layout: front (zeroed by ONE memset per sort): [n_seg][8] identity flags, [n_seg][4][tiles][256]
// histograms; then per map: key0 key1 idx0 idx1 offs
__host__ __device__ inline size_t front_words(int n_seg, int n_tiles) {
  return static_cast<size_t>(n_seg) * (8u + 4u * kBins * static_cast<size_t>(n_tiles));
}
__host__ __device__ inline size_t seg_words(int64_t hw, int n_tiles) {
  return static_cast<size_t>(4 * hw) + static_cast<size_t>(kBins) * n_tiles;
}

__device__ __forceinline__ SegPtrs seg_ptrs(uint32_t* ws, int seg, int n_seg, int64_t hw, int n_tiles) {
  SegPtrs p;
  p.flag = ws + static_cast<size_t>(seg) * 8u;
  uint32_t* h = ws + static_cast<size_t>(n_seg) * 8u + static_cast<size_t>(seg) * 4u * kBins * n_tiles;
#pragma unroll
  for (int i = 0; i < 4; ++i) p.hist[i] = h + static_cast<size_t>(i) * kBins * n_tiles;
  uint32_t* b = ws + front_words(n_seg, n_tiles) + static_cast<size_t>(seg) * seg_words(hw, n_tiles);
  p.key[0] = b; p.key[1] = b + hw; p.idx[0] = b + 2 * hw; p.idx[1] = b + 3 * hw;
  p.offs = b + 4 * hw;
  return p;
}

// zero the front of the scratch (flags + histograms).  A kernel, NOT hipMemsetAsync: on the HIP runtime the PyTorch
// 2.10+rocm7.0 wheel bundles (torch/lib/libamdhip64.so, roc-7.0.2, hipRuntimeGetVersion 70051831 -- the runtime every torch
// process runs on, whatever /opt/rocm holds) a memset node of a hipGraph takes effect on the FIRST hipGraphLaunch only: every
// later launch leaves all words untouched, for any size, offset, capture mode or instantiate flag, with or without torch in
// the process; on ROCm 7.2.0's own runtime (70226015) the same programs replay correctly
// (csrc/tune/repro_graph_memset{2,3}.hip; profiles/r02_repro_graph_memset*.txt, r02_exp_graph_memset_torch.json).
// The C ABI promises graph-capturable launches, so no entry point of this library may enqueue a memset.
__global__ __launch_bounds__(kBlock) void rank_zero_kernel(uint32_t* __restrict__ w, size_t n_words) {
  const size_t stride = static_cast<size_t>(gridDim.x) * kBlock;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x; i < n_words; i += stride) w[i] = 0u;
}

template <bool FIRST>
__device__ __forceinline__ uint32_t load_key(const float* sal, const uint32_t* kin, int64_t i) {
  return FIRST ? sort_key(sal[i]) : kin[i];
}

template <bool FIRST>
__global__ __launch_bounds__(kBlock) void rank_hist_kernel(const float* __restrict__ sal_all, int64_t hw, int n_tiles, int pass,
                                                           uint32_t* __restrict__ ws) {
  __shared__ uint32_t h[kBins];
  const int seg = blockIdx.y, tile = blockIdx.x;
  const SegPtrs p = seg_ptrs(ws, seg, gridDim.y, hw, n_tiles);
  const float* sal = sal_all + seg * hw;
  const uint32_t* kin = p.key[pass & 1];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = static_cast<int64_t>(tile) * kTile;
#pragma unroll
  for (int j = 0; j < kTile / kBlock; ++j) {
    const int64_t i = base + j * kBlock + threadIdx.x;
    if (i < hw) atomicAdd(&h[(load_key<FIRST>(sal, kin, i) >> (8 * pass)) & 255u], 1u);
  }
  __syncthreads();
  p.hist[pass][tile * kBins + threadIdx.x] = h[threadIdx.x];
}

__global__ __launch_bounds__(kBins) void rank_scan_kernel(int64_t hw, int n_tiles, int pass, uint32_t* __restrict__ ws) {
  __shared__ uint32_t wsum[kBins / 64];
  const SegPtrs p = seg_ptrs(ws, blockIdx.x, gridDim.x, hw, n_tiles);
  const uint32_t* __restrict__ hist = p.hist[pass];
  uint32_t* __restrict__ offs = p.offs;
  const int d = threadIdx.x, lane = d & 63, wave = d >> 6;
  uint32_t total = 0;                                     // coalesced: lane = digit, row = tile
  int t = 0;
  for (; t + 8 <= n_tiles; t += 8) {
    uint32_t c[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) c[u] = hist[(t + u) * kBins + d];
#pragma unroll
    for (int u = 0; u < 8; ++u) total += c[u];
  }
  for (; t < n_tiles; ++t) total += hist[t * kBins + d];
  uint32_t incl = total;                                  // exclusive scan of the 256 digit totals
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t up = __shfl_up(incl, off, kWave);
    if (lane >= off) incl += up;
  }
  if (lane == 63) wsum[wave] = incl;
  const bool one_bin = __any(total == static_cast<uint32_t>(hw));
  __syncthreads();
  uint32_t run = incl - total;
  for (int w = 0; w < wave; ++w) run += wsum[w];
  for (t = 0; t + 8 <= n_tiles; t += 8) {                 // within a digit: tiles in order
    uint32_t c[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) c[u] = hist[(t + u) * kBins + d];
#pragma unroll
    for (int u = 0; u < 8; ++u) { offs[(t + u) * kBins + d] = run; run += c[u]; }
  }
  for (; t < n_tiles; ++t) { offs[t * kBins + d] = run; run += hist[t * kBins + d]; }
  if (one_bin && lane == 0) p.flag[pass] = 1u;            // the wave that holds the full bin (flags were zeroed by the launcher)
}

template <bool FIRST, bool LAST>
__global__ __launch_bounds__(kBlock) void rank_scatter_kernel(const float* __restrict__ sal_all, int64_t hw, int n_tiles, int pass,
                                                              uint32_t* __restrict__ ws, int32_t* __restrict__ order_all,
                                                              int32_t* __restrict__ rank_all) {
  __shared__ uint32_t cw[kBlock / 64][kBins];             // per-wave running count of each digit
  __shared__ uint32_t gbase[kBins];                       // offs[digit][tile]
  const int seg = blockIdx.y, tile = blockIdx.x;
  const SegPtrs p = seg_ptrs(ws, seg, gridDim.y, hw, n_tiles);
  const float* sal = sal_all + seg * hw;
  const uint32_t* kin = p.key[pass & 1];
  const uint32_t* iin = p.idx[pass & 1];
  uint32_t* kout = p.key[(pass & 1) ^ 1];
  uint32_t* iout = p.idx[(pass & 1) ^ 1];
  int32_t* order = order_all + seg * hw;
  int32_t* rank = rank_all + seg * hw;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const bool identity = p.flag[pass] != 0;                // uniform for the whole map
  const int64_t base = static_cast<int64_t>(tile) * kTile + wave * 256;

  uint32_t key[4], idx[4], dig[4], loc[4];
  bool live[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t i = base + j * 64 + lane;
    live[j] = i < hw;
    key[j] = live[j] ? load_key<FIRST>(sal, kin, i) : 0u;
    idx[j] = live[j] ? (FIRST ? static_cast<uint32_t>(i) : iin[i]) : 0u;
  }
  if (identity) {                                         // every key shares this digit: order unchanged
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t i = base + j * 64 + lane;
      if (!live[j]) continue;
      if (LAST) { order[i] = static_cast<int32_t>(idx[j]); rank[idx[j]] = static_cast<int32_t>(i); }
      else {
        kout[i] = key[j]; iout[i] = idx[j];
        atomicAdd(p.hist[pass + 1] + (i >> 10) * kBins + ((key[j] >> (8 * (pass + 1))) & 255u), 1u);
      }
    }
    return;
  }
#pragma unroll
  for (int w = 0; w < kBlock / 64; ++w) cw[w][t] = 0;
  gbase[t] = p.offs[tile * kBins + t];
  __syncthreads();
  volatile uint32_t* myrow = cw[wave];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    dig[j] = (key[j] >> (8 * pass)) & 255u;
    // lanes of this round holding the same digit (idle lanes match nobody)
    unsigned long long m = __ballot(live[j]);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const unsigned long long set = __ballot(live[j] && ((dig[j] >> b) & 1u));
      m &= ((dig[j] >> b) & 1u) ? set : ~set;
    }
    const unsigned long long below = m & ((1ull << lane) - 1ull);
    const uint32_t before = live[j] ? myrow[dig[j]] : 0u;
    loc[j] = before + static_cast<uint32_t>(__popcll(below));
    const bool leader = live[j] && (m >> lane) == 1ull;   // highest lane of its group
    if (leader) myrow[dig[j]] = loc[j] + 1u;
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  // digit-wise exclusive prefix over the 4 waves (lane = digit)
  {
    uint32_t run = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) { const uint32_t c = cw[w][t]; cw[w][t] = run; run += c; }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (!live[j]) continue;
    const uint32_t pos = gbase[dig[j]] + cw[wave][dig[j]] + loc[j];
    if (LAST) { order[pos] = static_cast<int32_t>(idx[j]); rank[idx[j]] = static_cast<int32_t>(pos); }
    else {
      kout[pos] = key[j]; iout[pos] = idx[j];
      // next pass's histogram, filed under the tile this key lands in
      atomicAdd(p.hist[pass + 1] + (pos >> 10) * kBins + ((key[j] >> (8 * (pass + 1))) & 255u), 1u);
    }
  }
}

inline int tiles_of(int64_t hw) { return static_cast<int>((hw + kTile - 1) / kTile); }

}  // namespace

XAI_EXPORT size_t xai_rank_workspace_bytes(int n_seg, int64_t hw) {
  if (n_seg <= 0 || hw <= 0) return 0;
  const int nt = tiles_of(hw);
  return (front_words(n_seg, nt) + static_cast<size_t>(n_seg) * seg_words(hw, nt)) * sizeof(uint32_t);
}

XAI_EXPORT int xai_rank_f32(const float* sal, int n_seg, int64_t hw, int32_t* order, int32_t* rank, void* ws, size_t ws_bytes,
                            xai_stream_t stream) {
  XAI_REQUIRE_PTR(sal); XAI_REQUIRE_PTR(order); XAI_REQUIRE_PTR(rank); XAI_REQUIRE_PTR(ws);
  XAI_REQUIRE(n_seg > 0 && hw > 0, XAI_E_SHAPE);
  XAI_REQUIRE(hw < (int64_t(1) << 31) && n_seg <= 65535, XAI_E_UNSUPPORTED);
  XAI_REQUIRE(ws_bytes >= xai_rank_workspace_bytes(n_seg, hw), XAI_E_SHAPE);
  XAI_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 3u) == 0, XAI_E_SHAPE);
  hipStream_t st = static_cast<hipStream_t>(stream);
  uint32_t* w = static_cast<uint32_t*>(ws);
  const int n_tiles = tiles_of(hw);
  // zero the identity flags and the four histograms of every map
  const size_t fw = front_words(n_seg, n_tiles);
#ifdef XAI_RANK_ZERO_WITH_MEMSET   // scratch builds of profiles/experiments/exp_graph_memset.py only; never defined by the Makefile
  if (hipMemsetAsync(w, 0, fw * sizeof(uint32_t), st) != hipSuccess) return xai_launch_status();
#else
  hipLaunchKernelGGL(rank_zero_kernel, dim3(static_cast<unsigned>(std::min<size_t>(1024, (fw + kBlock - 1) / kBlock))), dim3(kBlock), 0, st, w, fw);
#endif
  const dim3 grid(n_tiles, n_seg), blk(kBlock);
  hipLaunchKernelGGL(rank_hist_kernel<true>, grid, blk, 0, st, sal, hw, n_tiles, 0, w);
  for (int pass = 0; pass < 4; ++pass) {
    hipLaunchKernelGGL(rank_scan_kernel, dim3(n_seg), dim3(kBins), 0, st, hw, n_tiles, pass, w);
    if (pass == 0)      hipLaunchKernelGGL((rank_scatter_kernel<true, false>), grid, blk, 0, st, sal, hw, n_tiles, pass, w, order, rank);
    else if (pass == 3) hipLaunchKernelGGL((rank_scatter_kernel<false, true>), grid, blk, 0, st, sal, hw, n_tiles, pass, w, order, rank);
    else                hipLaunchKernelGGL((rank_scatter_kernel<false, false>), grid, blk, 0, st, sal, hw, n_tiles, pass, w, order, rank);
  }
  return xai_launch_status();
}
