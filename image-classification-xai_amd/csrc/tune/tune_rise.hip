// Mapping study for K4 rise_apply (s = 8, 3 x 224 x 224): masked[n][c][p] = image[c][p] * mask_n[p], 4N bytes written per
// mask.  V0 = the round-1 kernel (one 1024-pixel tile of one mask per workgroup; per lane 1 row tap + 4 column taps in
// fp64, 16 bit tests, 4-product blend).  V1 = separable: a workgroup owns ROWS image rows of one mask, stages the 8
// column-interpolated grid rows colrow[r][x] = (1-tc(x)) g[r][c0(x)] + tc(x) g[r][c1(x)] (8 x W fp32) and the ROWS row taps
// in LDS once, then every pixel quad is one vertical lerp of two b128 LDS reads.  Prints time, achieved GB/s of the
// 4N-byte algorithmic traffic, and the largest difference of the produced masks from V0's.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Tap { int i0, i1; float t; };
__device__ __forceinline__ Tap make_tap(int j, int n_in, double ratio) {
  double c = (j + 0.5) * ratio - 0.5;
  if (c < 0) c = -c;
  const int i0 = static_cast<int>(floor(c));
  int i1 = i0 + 1;
  if (i1 >= n_in) i1 = 2 * n_in - 2 - i1;
  if (i1 < 0) i1 = 0;
  return Tap{i0, i1, static_cast<float>(c - i0)};
}
__device__ __forceinline__ unsigned long long pack_grid8(const uint8_t* g) {
  const unsigned long long* w = reinterpret_cast<const unsigned long long*>(g);
  unsigned long long bits = 0;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const unsigned long long row = w[r] & 0x0101010101010101ull;
    bits |= ((row * 0x0102040810204080ull) >> 56) << (8 * r);
  }
  return bits;
}
__device__ __forceinline__ float blend8(uint2 rows, const Tap& r, const Tap& c, float lo, float hi) {
  const float wr0 = 1.f - r.t, wr1 = r.t, wc0 = 1.f - c.t, wc1 = c.t;
  float v = ((rows.x >> c.i0) & 1u) ? wr0 * wc0 : 0.f;
  v += ((rows.x >> c.i1) & 1u) ? wr0 * wc1 : 0.f;
  v += ((rows.y >> c.i0) & 1u) ? wr1 * wc0 : 0.f;
  v += ((rows.y >> c.i1) & 1u) ? wr1 * wc1 : 0.f;
  return fminf(fmaxf(v, lo), hi);
}
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

__global__ __launch_bounds__(256) void v0(const uint8_t* __restrict__ grid, const int32_t* __restrict__ shift, double rh, double rw,
                                          const float* __restrict__ image, int C, int H, int W, float* __restrict__ masked) {
  const int n = blockIdx.y;
  const unsigned long long bits = pack_grid8(grid + static_cast<int64_t>(n) * 64);
  const float hi = bits != 0ull ? 1.f : 0.f, lo = bits == ~0ull ? 1.f : 0.f;
  const int64_t hw = static_cast<int64_t>(H) * W;
  const int64_t p = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) * 4;
  if (p >= hw) return;
  const int y = static_cast<int>(static_cast<uint32_t>(p) / static_cast<uint32_t>(W)), x = static_cast<int>(p - static_cast<int64_t>(y) * W);
  const int sx = shift[2 * n + 1];
  const Tap tr = make_tap(y + shift[2 * n], 8, rh);
  const uint2 rows = make_uint2(static_cast<uint32_t>(bits >> (tr.i0 * 8)) & 0xFFu, static_cast<uint32_t>(bits >> (tr.i1 * 8)) & 0xFFu);
  float4 m;
  m.x = blend8(rows, tr, make_tap(x + sx, 8, rw), lo, hi);
  m.y = blend8(rows, tr, make_tap(x + 1 + sx, 8, rw), lo, hi);
  m.z = blend8(rows, tr, make_tap(x + 2 + sx, 8, rw), lo, hi);
  m.w = blend8(rows, tr, make_tap(x + 3 + sx, 8, rw), lo, hi);
  float* o = masked + static_cast<int64_t>(n) * C * hw + p;
  for (int c = 0; c < C; ++c) {
    const float4 v = ld4(image + c * hw + p);
    st4(o + c * hw, make_float4(v.x * m.x, v.y * m.y, v.z * m.z, v.w * m.w));
  }
}

// V0 variants for the ceiling question: NOMASK = the same stores with the mask arithmetic removed (m = 1): what the write
// pattern alone allows; NT = non-temporal stores; QPL = pixel quads per lane (tile = 1024 * QPL px); MPW = consecutive masks
// per workgroup on the same tile (image quads stay in registers).
typedef float fx4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ void stq(float* p, float4 v) {
  if (NT) { fx4 t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, reinterpret_cast<fx4*>(p)); }
  else st4(p, v);
}
template <bool NOMASK, bool NT, int QPL, int MPW>
__global__ __launch_bounds__(256) void v0x(const uint8_t* __restrict__ grid, const int32_t* __restrict__ shift, int n_masks, double rh, double rw,
                                           const float* __restrict__ image, int H, int W, float* __restrict__ masked) {
  constexpr int C = 3;
  const int64_t hw = static_cast<int64_t>(H) * W;
  int64_t p[QPL]; float4 img[QPL][C]; bool live[QPL];
#pragma unroll
  for (int q = 0; q < QPL; ++q) {
    p[q] = ((static_cast<int64_t>(blockIdx.x) * QPL + q) * 256 + threadIdx.x) * 4;
    live[q] = p[q] < hw;
#pragma unroll
    for (int c = 0; c < C; ++c) img[q][c] = live[q] ? ld4(image + c * hw + p[q]) : make_float4(0, 0, 0, 0);
  }
#pragma unroll
  for (int k = 0; k < MPW; ++k) {
    const int n = blockIdx.y * MPW + k;
    if (n >= n_masks) break;
    const unsigned long long bits = pack_grid8(grid + static_cast<int64_t>(n) * 64);
    const float hi = bits != 0ull ? 1.f : 0.f, lo = bits == ~0ull ? 1.f : 0.f;
    const int sy = shift[2 * n], sx = shift[2 * n + 1];
#pragma unroll
    for (int q = 0; q < QPL; ++q) {
      if (!live[q]) continue;
      float4 m = make_float4(1.f, 1.f, 1.f, 1.f);
      if (!NOMASK) {
        const int y = static_cast<int>(static_cast<uint32_t>(p[q]) / static_cast<uint32_t>(W)), x = static_cast<int>(p[q] - static_cast<int64_t>(y) * W);
        const Tap tr = make_tap(y + sy, 8, rh);
        const uint2 rows = make_uint2(static_cast<uint32_t>(bits >> (tr.i0 * 8)) & 0xFFu, static_cast<uint32_t>(bits >> (tr.i1 * 8)) & 0xFFu);
        m.x = blend8(rows, tr, make_tap(x + sx, 8, rw), lo, hi);
        m.y = blend8(rows, tr, make_tap(x + 1 + sx, 8, rw), lo, hi);
        m.z = blend8(rows, tr, make_tap(x + 2 + sx, 8, rw), lo, hi);
        m.w = blend8(rows, tr, make_tap(x + 3 + sx, 8, rw), lo, hi);
      }
      float* o = masked + static_cast<int64_t>(n) * C * hw + p[q];
#pragma unroll
      for (int c = 0; c < C; ++c) stq<NT>(o + c * hw, make_float4(img[q][c].x * m.x, img[q][c].y * m.y, img[q][c].z * m.z, img[q][c].w * m.w));
    }
  }
}

// V1: separable, LDS-staged.  grid = (ceil(H / ROWS), masks); dynamic LDS = 8 * W floats + ROWS taps.
// MPW = masks per workgroup (same pixel rows, consecutive masks; the image rows stay L1/L2-hot between them).
template <int BLOCK, int MPW>
__global__ __launch_bounds__(BLOCK) void v1(const uint8_t* __restrict__ grid, const int32_t* __restrict__ shift, int n_masks, int rows_per_wg,
                                            double rh, double rw, const float* __restrict__ image, int C, int H, int W,
                                            float* __restrict__ masked) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int64_t hw = static_cast<int64_t>(H) * W;
  const int y0 = blockIdx.x * rows_per_wg, y1 = min(y0 + rows_per_wg, H);
  const int W4 = W >> 2, quads = (y1 - y0) * W4;
  for (int k = 0; k < MPW; ++k) {
    const int n = blockIdx.y * MPW + k;
    if (n >= n_masks) break;
    float* colrow = lds;                                          // [8][W]
    Tap* rtap = reinterpret_cast<Tap*>(lds + 8 * W);              // [rows_per_wg]
    const unsigned long long bits = pack_grid8(grid + static_cast<int64_t>(n) * 64);
    const float hi = bits != 0ull ? 1.f : 0.f, lo = bits == ~0ull ? 1.f : 0.f;
    const int sy = shift[2 * n], sx = shift[2 * n + 1];
    if (k) __syncthreads();
    for (int x = threadIdx.x; x < W; x += BLOCK) {
      const Tap tc = make_tap(x + sx, 8, rw);
      const float w1 = tc.t, w0 = 1.f - tc.t;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const uint32_t row = static_cast<uint32_t>(bits >> (8 * r)) & 0xFFu;
        float v = ((row >> tc.i0) & 1u) ? w0 : 0.f;
        v += ((row >> tc.i1) & 1u) ? w1 : 0.f;
        colrow[r * W + x] = v;
      }
    }
    for (int r = threadIdx.x; r < y1 - y0; r += BLOCK) rtap[r] = make_tap(y0 + r + sy, 8, rh);
    __syncthreads();
    float* o = masked + static_cast<int64_t>(n) * C * hw + static_cast<int64_t>(y0) * W;
    const float* im = image + static_cast<int64_t>(y0) * W;
    for (int q = threadIdx.x; q < quads; q += BLOCK) {
      const int r = static_cast<int>(static_cast<uint32_t>(q) / static_cast<uint32_t>(W4)), x = (q - r * W4) << 2;
      const Tap tr = rtap[r];
      const float4 a = ld4(colrow + tr.i0 * W + x), b = ld4(colrow + tr.i1 * W + x);
      const float w1 = tr.t, w0 = 1.f - tr.t;
      float4 m;
      m.x = fminf(fmaxf(w0 * a.x + w1 * b.x, lo), hi);
      m.y = fminf(fmaxf(w0 * a.y + w1 * b.y, lo), hi);
      m.z = fminf(fmaxf(w0 * a.z + w1 * b.z, lo), hi);
      m.w = fminf(fmaxf(w0 * a.w + w1 * b.w, lo), hi);
      const int64_t off = static_cast<int64_t>(q) << 2;
      for (int c = 0; c < C; ++c) {
        const float4 v = ld4(im + c * hw + off);
        st4(o + c * hw + off, make_float4(v.x * m.x, v.y * m.y, v.z * m.z, v.w * m.w));
      }
    }
  }
}

template <typename F> double time_ms(F&& launch) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) launch();
  std::vector<float> t;
  for (int i = 0; i < 15; ++i) { CK(hipEventRecord(a, 0)); launch(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms); }
  std::sort(t.begin(), t.end()); return t[7];
}

int main() {
  const int C = 3, H = 224, W = 224, s = 8, cell = 28;
  const int64_t hw = (int64_t)H * W;
  const double rh = (double)s / ((s + 1) * cell), rw = rh;
  for (int N : {50, 250, 1000}) {
    std::vector<uint8_t> g((size_t)N * 64); std::vector<int32_t> sh((size_t)N * 2);
    uint32_t st = 12345u + N;
    auto rnd = [&] { st = st * 1664525u + 1013904223u; return st >> 8; };
    for (auto& b : g) b = rnd() & 1;
    for (auto& v : sh) v = rnd() % cell;
    std::vector<float> img((size_t)C * hw);
    for (auto& v : img) v = (float)(rnd() % 2001) / 1000.f - 1.f;
    uint8_t* dg; int32_t* dsh; float *dimg, *out0, *out1;
    CK(hipMalloc(&dg, g.size())); CK(hipMalloc(&dsh, sh.size() * 4)); CK(hipMalloc(&dimg, img.size() * 4));
    CK(hipMalloc(&out0, (size_t)N * C * hw * 4)); CK(hipMalloc(&out1, (size_t)N * C * hw * 4));
    CK(hipMemcpy(dg, g.data(), g.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dsh, sh.data(), sh.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dimg, img.data(), img.size() * 4, hipMemcpyHostToDevice));
    const double bytes = (double)N * C * hw * 4;
    std::vector<float> ref((size_t)N * C * hw), got(ref.size());
    auto rep = [&](const char* name, double ms, bool check) {
      double worst = -1;
      if (check) {
        CK(hipMemcpy(got.data(), out1, got.size() * 4, hipMemcpyDeviceToHost));
        worst = 0;
        for (size_t i = 0; i < got.size(); ++i) worst = std::max(worst, (double)std::fabs(got[i] - ref[i]));
      }
      printf("N=%-5d %-44s %8.1f us %8.1f GB/s frac=%.3f  max|diff vs V0|=%.3g\n", N, name, ms * 1e3, bytes / ms / 1e6, bytes / ms / 8e9, worst);
    };
    double t0 = time_ms([&] { hipLaunchKernelGGL(v0, dim3((unsigned)((hw + 1023) / 1024), N), dim3(256), 0, 0, dg, dsh, rh, rw, dimg, C, H, W, out0); });
    CK(hipMemcpy(ref.data(), out0, ref.size() * 4, hipMemcpyDeviceToHost));
    rep("V0 round-1 kernel (1024 px/wg)", t0, false);
#define RUN1(BLOCK, MPW, ROWS)                                                                                                        \
  {                                                                                                                                    \
    CK(hipMemset(out1, 0xFF, (size_t)N* C* hw * 4));                                                                                   \
    const size_t ldsb = (size_t)8 * W * 4 + (size_t)ROWS * sizeof(Tap);                                                                \
    double t = time_ms([&] { hipLaunchKernelGGL((v1<BLOCK, MPW>), dim3((H + ROWS - 1) / ROWS, (N + MPW - 1) / MPW), dim3(BLOCK), ldsb, 0, dg, dsh, N, ROWS, rh, rw, dimg, C, H, W, out1); }); \
    rep("V1 separable block=" #BLOCK " masks/wg=" #MPW " rows=" #ROWS, t, true);                                                       \
  }
#define RUN0(NOMASK, NT, QPL, MPW)                                                                                                    \
  {                                                                                                                                    \
    CK(hipMemset(out1, 0xFF, (size_t)N* C* hw * 4));                                                                                   \
    double t = time_ms([&] { hipLaunchKernelGGL((v0x<NOMASK, NT, QPL, MPW>), dim3((unsigned)((hw + 1024 * QPL - 1) / (1024 * QPL)), (N + MPW - 1) / MPW), dim3(256), 0, 0, dg, dsh, N, rh, rw, dimg, H, W, out1); }); \
    rep("V0x nomask=" #NOMASK " nt=" #NT " quads/lane=" #QPL " masks/wg=" #MPW, t, !NOMASK);                                            \
  }
    RUN0(false, false, 1, 1) RUN0(true, false, 1, 1) RUN0(false, true, 1, 1) RUN0(true, true, 1, 1)
    RUN0(false, false, 2, 1) RUN0(true, false, 2, 1) RUN0(false, false, 1, 2) RUN0(true, false, 1, 2) RUN0(false, true, 1, 2)
    RUN0(false, false, 1, 4) RUN0(false, false, 2, 2) RUN0(false, true, 2, 1)
    RUN1(256, 1, 8) RUN1(256, 1, 16) RUN1(256, 1, 32) RUN1(256, 1, 56) RUN1(256, 1, 112) RUN1(256, 1, 224)
    RUN1(512, 1, 32) RUN1(512, 1, 56) RUN1(1024, 1, 112)
    RUN1(256, 2, 16) RUN1(256, 2, 32) RUN1(256, 4, 16) RUN1(256, 4, 32)
    CK(hipFree(dg)); CK(hipFree(dsh)); CK(hipFree(dimg)); CK(hipFree(out0)); CK(hipFree(out1));
  }
  return 0;
}
