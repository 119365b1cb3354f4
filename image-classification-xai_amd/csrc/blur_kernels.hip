// K7: zero-padded separable blur (the insertion substrate) for gfx950.
//
// One workgroup per THx64 output tile of one channel plane (TH = 16, or 32 for batches): the (TH+2r)x(64+2r)
// input halo is staged in LDS with zero fill, the horizontal pass writes a (TH+2r)x64 LDS intermediate, the
// vertical pass writes the tile -- one HBM read and one HBM write per pixel, the 2r-row
// intermediate never leaves the CU.  Taps are applied in ascending order, one explicit FMA per tap.
#include "xai_common.h"

namespace {

constexpr int TW = 64, kBlock = 256;
static_assert(TW == 64, "the index arithmetic below assumes 64-wide tiles");

// Both passes slide a 4-wide register window: a lane owns 4 adjacent outputs (along x in the horizontal pass, along y
// in the vertical one) and reads klen + 3 LDS values for them instead of 4 * klen.  Every output adds its taps in
// ascending order with one fused multiply-add per tap -- the same sequence blur_1d_kernel runs, so the fused and the
// two-pass form agree bit for bit.  (FMA on purpose: the reference is a dense 961-tap conv2d whose summation order
// cannot be reproduced by any separable form; parity is the 1e-5 bar, and the multiply-add count is what bounds this
// kernel.)
//
// KLEN = 31 (the reference's gkern(31, 31), evaluatePerturbation.py:456): the horizontal pass reads each lane's 36-dword row
// segment with nine 16-byte LDS reads and runs the 31 taps out of registers.  With the sliding window of dword reads
// (KLEN = 0, any odd length) lanes sit 4 dwords apart, so quads q and q+8 of a row share a bank: a 2-way conflict on
// 34 ds_read_b32 per lane, 2 108 of the 2 652 LDS cycles of a tile.  The 16-byte reads are conflict-free with a row pitch
// of 96 dwords once odd rows visit their quads in the order q ^ 8 (a 16-lane LDS group holds quads {0-3, 12-15} of one
// row and {4-11} of the next: the rotation puts the second row on banks 16-47, the first stays on 0-15 / 48-63).
template <int TH, int KLEN>
__global__ __launch_bounds__(kBlock) void blur_sep_kernel(const float* __restrict__ x, const float* __restrict__ k1d, int klen,
                                                          int H, int W, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int r = klen / 2;
  const int IH = TH + 2 * r, IW = TW + 2 * r;
  // KLEN == 0: odd row pitch -- a wave reads 4 rows x 16 quads with a 4-float stride inside a row, and an odd pitch puts the
  // four rows on the four bank residues mod 4 instead of on the same one.  KLEN == 31: pitch 96 (see above)
  const int P = KLEN ? 96 : (IW | 1);
  float* tin = lds;                            // [IH][P]
  float* tmid = lds + ((IH * P + 3) & ~3);     // [IH][TW], 16-byte aligned
  const int64_t plane = static_cast<int64_t>(blockIdx.z) * H * W;
  const int y0 = blockIdx.y * TH, x0 = blockIdx.x * TW;
  // halo staging: a wave per row, lanes along the row (no integer division, coalesced row reads).  Loads are issued in
  // groups of 32 (16 rows x 2 column steps) into registers before any of them is stored to LDS: with a store between
  // every two loads the staging was one HBM latency per element and dominated the kernel; with groups of 8 it was still
  // four latencies per tile (62 halo rows at 31 taps / 16 rows per round), now it is one.
  {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int kRows = 16, kWaves = kBlock >> 6;
    for (int base = wave; base < IH; base += kRows * kWaves) {
      float v[kRows][2];
#pragma unroll
      for (int k = 0; k < kRows; ++k) {
        const int ly = base + k * kWaves;
        const int gy = y0 + ly - r;
        const bool row_in = ly < IH && gy >= 0 && gy < H;
        const float* src = x + plane + static_cast<int64_t>(row_in ? gy : 0) * W;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int lx = lane + 64 * c;
          const int gx = x0 + lx - r;
          v[k][c] = (row_in && lx < IW && gx >= 0 && gx < W) ? src[gx] : 0.f;
        }
      }
#pragma unroll
      for (int k = 0; k < kRows; ++k) {
        const int ly = base + k * kWaves;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int lx = lane + 64 * c;
          if (ly < IH && lx < IW) tin[ly * P + lx] = v[k][c];
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < IH * (TW / 4); i += kBlock) {
    const int ly = i >> 4;                             // TW / 4 == 16 quads per row
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int q;
    if (KLEN) {
      q = (i & 15) ^ ((ly & 1) << 3);
      const float* row = tin + ly * P + 4 * q;
      float w[KLEN + 5];                               // KLEN + 3 values are used; 36 = nine quads for KLEN = 31
      // explicit LDS address space + 16-byte vector type: left to itself the compiler re-splits these loads into
      // ds_read2_b32 / ds_read2_b64 (which bring the 2-way conflict back)
      typedef float lds_f4 __attribute__((ext_vector_type(4)));
      const __attribute__((address_space(3))) lds_f4* seg = (const __attribute__((address_space(3))) lds_f4*)row;
#pragma unroll
      for (int v = 0; v < (KLEN + 5) / 4; ++v) {
        const lds_f4 t = __builtin_nontemporal_load(seg + v);
        w[4 * v] = t.x; w[4 * v + 1] = t.y; w[4 * v + 2] = t.z; w[4 * v + 3] = t.w;
      }
#pragma unroll
      for (int j = 0; j < KLEN; ++j) {
        const float k = k1d[j];
        a0 = __builtin_fmaf(k, w[j], a0); a1 = __builtin_fmaf(k, w[j + 1], a1); a2 = __builtin_fmaf(k, w[j + 2], a2); a3 = __builtin_fmaf(k, w[j + 3], a3);
      }
    } else {
      q = i & 15;
      const float* row = tin + ly * P + 4 * q;
      float w0 = row[0], w1 = row[1], w2 = row[2];
#pragma unroll 8
      for (int j = 0; j < klen; ++j) {
        const float w3 = row[j + 3];
        const float k = k1d[j];
        a0 = __builtin_fmaf(k, w0, a0); a1 = __builtin_fmaf(k, w1, a1); a2 = __builtin_fmaf(k, w2, a2); a3 = __builtin_fmaf(k, w3, a3);
        w0 = w1; w1 = w2; w2 = w3;
      }
    }
    st4(tmid + ly * TW + 4 * q, make_float4(a0, a1, a2, a3));
  }
  __syncthreads();
  for (int i = threadIdx.x; i < (TH / 4) * TW; i += kBlock) {
    const int g = i >> 6, lx = i & 63;                 // TW == 64
    const int gx = x0 + lx;
    const float* col = tmid + (4 * g) * TW + lx;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (KLEN) {                                          // all KLEN + 3 column values in flight at once, taps from registers
      float w[KLEN + 3];
#pragma unroll
      for (int t = 0; t < KLEN + 3; ++t) w[t] = col[t * TW];
#pragma unroll
      for (int j = 0; j < KLEN; ++j) {
        const float k = k1d[j];
        a0 = __builtin_fmaf(k, w[j], a0); a1 = __builtin_fmaf(k, w[j + 1], a1); a2 = __builtin_fmaf(k, w[j + 2], a2); a3 = __builtin_fmaf(k, w[j + 3], a3);
      }
    } else {
      float w0 = col[0], w1 = col[TW], w2 = col[2 * TW];
#pragma unroll 8
      for (int j = 0; j < klen; ++j) {
        const float w3 = col[(j + 3) * TW];
        const float k = k1d[j];
        a0 = __builtin_fmaf(k, w0, a0); a1 = __builtin_fmaf(k, w1, a1); a2 = __builtin_fmaf(k, w2, a2); a3 = __builtin_fmaf(k, w3, a3);
        w0 = w1; w1 = w2; w2 = w3;
      }
    }
    if (gx >= W) continue;
    const int gy = y0 + 4 * g;
    float* o = out + plane + static_cast<int64_t>(gy) * W + gx;
    if (gy < H) o[0] = a0;
    if (gy + 1 < H) o[W] = a1;
    if (gy + 2 < H) o[2 * static_cast<int64_t>(W)] = a2;
    if (gy + 3 < H) o[3 * static_cast<int64_t>(W)] = a3;
  }
}

// One 1-D zero-padded pass along W (axis 1) or H (axis 0) for kernels too long for the fused tile version
// (the growing-kernel search of the reference's MDA branch goes up to 101 taps): lane = output pixel, taps
// through the scalar cache, rows of neighbouring lanes coalesce.
__global__ __launch_bounds__(kBlock) void blur_1d_kernel(const float* __restrict__ x, const float* __restrict__ k1d, int klen, int axis,
                                                         int H, int W, float* __restrict__ out) {
  const int64_t plane = static_cast<int64_t>(blockIdx.y) * H * W;
  const int64_t p = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (p >= static_cast<int64_t>(H) * W) return;
  const int y = static_cast<int>(p / W), xx = static_cast<int>(p - static_cast<int64_t>(y) * W);
  const int r = klen / 2;
  float acc = 0.f;
  if (axis == 1) {
    for (int j = 0; j < klen; ++j) {
      const int gx = xx + j - r;
      acc = __builtin_fmaf(k1d[j], (gx >= 0 && gx < W) ? x[plane + static_cast<int64_t>(y) * W + gx] : 0.f, acc);
    }
  } else {
    for (int j = 0; j < klen; ++j) {
      const int gy = y + j - r;
      acc = __builtin_fmaf(k1d[j], (gy >= 0 && gy < H) ? x[plane + static_cast<int64_t>(gy) * W + xx] : 0.f, acc);
    }
  }
  out[plane + p] = acc;
}

}  // namespace

XAI_EXPORT int xai_blur_1d_f32(const float* x, const float* k1d, int klen, int axis, int B, int C, int H, int W, float* out,
                               xai_stream_t stream) {
  XAI_REQUIRE_PTR(x); XAI_REQUIRE_PTR(k1d); XAI_REQUIRE_PTR(out);
  XAI_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && klen > 0 && (klen & 1) && (axis == 0 || axis == 1) && x != out, XAI_E_SHAPE);
  XAI_REQUIRE(static_cast<int64_t>(B) * C <= 65535, XAI_E_UNSUPPORTED);
  dim3 grid(static_cast<unsigned>(xai_ceil_div(static_cast<int64_t>(H) * W, kBlock)), B * C);
  hipLaunchKernelGGL(blur_1d_kernel, grid, dim3(kBlock), 0, static_cast<hipStream_t>(stream), x, k1d, klen, axis, H, W, out);
  return xai_launch_status();
}

XAI_EXPORT int xai_blur_sep_f32(const float* x, const float* k1d, int klen, int B, int C, int H, int W, float* out,
                                xai_stream_t stream) {
  XAI_REQUIRE_PTR(x); XAI_REQUIRE_PTR(k1d); XAI_REQUIRE_PTR(out);
  XAI_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && klen > 0 && (klen & 1), XAI_E_SHAPE);
  XAI_REQUIRE(klen <= 63 && static_cast<int64_t>(B) * C <= 65535, XAI_E_UNSUPPORTED);
  const int r = klen / 2;
  // 16-row tiles while that is what it takes to give every CU a couple of workgroups (one image: 168 tiles), 32-row tiles
  // (a third less halo per output) for batches
  const int64_t tiles16 = static_cast<int64_t>(B) * C * ((W + TW - 1) / TW) * ((H + 15) / 16);
  const int pitch = klen == 31 ? 96 : ((TW + 2 * r) | 1);
  const size_t lds32 = static_cast<size_t>((32 + 2 * r) * pitch + 3 + (32 + 2 * r) * TW) * sizeof(float);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (tiles16 >= 8 * static_cast<int64_t>(xai_cu_count()) && lds32 <= 64 * 1024) {
    constexpr int TH = 32;
    const size_t lds = lds32;
    dim3 grid((W + TW - 1) / TW, (H + TH - 1) / TH, B * C);
    if (klen == 31) hipLaunchKernelGGL((blur_sep_kernel<TH, 31>), grid, dim3(kBlock), lds, st, x, k1d, klen, H, W, out);
    else            hipLaunchKernelGGL((blur_sep_kernel<TH, 0>), grid, dim3(kBlock), lds, st, x, k1d, klen, H, W, out);
  } else {
    constexpr int TH = 16;
    const size_t lds = static_cast<size_t>((TH + 2 * r) * pitch + 3 + (TH + 2 * r) * TW) * sizeof(float);
    dim3 grid((W + TW - 1) / TW, (H + TH - 1) / TH, B * C);
    if (klen == 31) hipLaunchKernelGGL((blur_sep_kernel<TH, 31>), grid, dim3(kBlock), lds, st, x, k1d, klen, H, W, out);
    else            hipLaunchKernelGGL((blur_sep_kernel<TH, 0>), grid, dim3(kBlock), lds, st, x, k1d, klen, H, W, out);
  }
  return xai_launch_status();
}
