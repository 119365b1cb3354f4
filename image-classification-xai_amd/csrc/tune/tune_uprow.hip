// Mapping study for K11 up_rownorm (768 feature maps 14x14 -> 224x224, min-max normalised, 154 MB written).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tune_uprow.hip -o tune_uprow && ./tune_uprow
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float fx4 __attribute__((ext_vector_type(4)));

struct Tap { int i0, i1; float l0, l1; };
__device__ __forceinline__ Tap make_tap(int o, int n_in, float ratio) {
  const float f = fmaxf(ratio * (o + 0.5f) - 0.5f, 0.f);
  Tap t; t.i0 = (int)f; t.i1 = t.i0 + (t.i0 < n_in - 1 ? 1 : 0); t.l1 = f - t.i0; t.l0 = 1.f - t.l1; return t;
}
__device__ __forceinline__ float wmin(float v) { for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64)); return v; }
__device__ __forceinline__ float wmax(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64)); return v; }
template <int BLOCK> __device__ __forceinline__ void bminmax(float& lo, float& hi, float* red) {
  lo = wmin(lo); hi = wmax(hi);
  const int wave = threadIdx.x >> 6, nw = BLOCK >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { red[wave] = lo; red[nw + wave] = hi; }
  __syncthreads();
  lo = red[0]; hi = red[nw];
  for (int i = 1; i < nw; ++i) { lo = fminf(lo, red[i]); hi = fmaxf(hi, red[nw + i]); }
}

// V0: lane = column, rows walked twice.  NT / DIV switches; SPLIT workgroups share one map's rows in the write pass.
template <bool NT, bool DIV, int SPLIT>
__global__ __launch_bounds__(256) void v_rows(const float* __restrict__ src, int h, int w, int H, int W, float* __restrict__ out) {
  extern __shared__ float tmp[];
  __shared__ float s[4096];
  __shared__ float red[8];
  const int map = blockIdx.x / SPLIT, part = blockIdx.x % SPLIT;
  const float* row = src + (long)map * h * w;
  for (int i = threadIdx.x; i < h * w; i += 256) s[i] = row[i];
  __syncthreads();
  const float rh = (float)h / (float)H, rw = (float)w / (float)W;
  for (int ox = threadIdx.x; ox < W; ox += 256) {
    const Tap tx = make_tap(ox, w, rw);
    for (int y = 0; y < h; ++y) tmp[y * W + ox] = s[y * w + tx.i0] * tx.l0 + s[y * w + tx.i1] * tx.l1;
  }
  __syncthreads();
  float lo = INFINITY, hi = -INFINITY;
#pragma unroll 4
  for (int oy = 0; oy < H; ++oy) {
    const Tap ty = make_tap(oy, h, rh);
    const float* t0 = tmp + ty.i0 * W; const float* t1 = tmp + ty.i1 * W;
    for (int ox = threadIdx.x; ox < W; ox += 256) { const float v = t0[ox] * ty.l0 + t1[ox] * ty.l1; lo = fminf(lo, v); hi = fmaxf(hi, v); }
  }
  bminmax<256>(lo, hi, red);
  const float span = hi - lo, inv = 1.f / span;
  float* dst = out + (long)map * H * W;
  const int per = (H + SPLIT - 1) / SPLIT, y0 = part * per, y1 = min(y0 + per, H);
#pragma unroll 4
  for (int oy = y0; oy < y1; ++oy) {
    const Tap ty = make_tap(oy, h, rh);
    const float* t0 = tmp + ty.i0 * W; const float* t1 = tmp + ty.i1 * W;
    for (int ox = threadIdx.x; ox < W; ox += 256) {
      const float v = t0[ox] * ty.l0 + t1[ox] * ty.l1;
      const float r = DIV ? (v - lo) / span : (v - lo) * inv;
      if (NT) __builtin_nontemporal_store(r, dst + oy * W + ox); else dst[oy * W + ox] = r;
    }
  }
}

// V1: lane = 4 consecutive pixels of the flattened map (W % 4 == 0), float4 stores; the min/max pass reads the h x W tile only
// through the h..H row taps as well.  SPLIT as above.
template <bool NT, int SPLIT>
__global__ __launch_bounds__(256) void v_quads(const float* __restrict__ src, int h, int w, int H, int W, float* __restrict__ out) {
  extern __shared__ float tmp[];
  __shared__ float s[4096];
  __shared__ float red[8];
  const int map = blockIdx.x / SPLIT, part = blockIdx.x % SPLIT;
  const float* row = src + (long)map * h * w;
  for (int i = threadIdx.x; i < h * w; i += 256) s[i] = row[i];
  __syncthreads();
  const float rh = (float)h / (float)H, rw = (float)w / (float)W;
  for (int ox = threadIdx.x; ox < W; ox += 256) {
    const Tap tx = make_tap(ox, w, rw);
    for (int y = 0; y < h; ++y) tmp[y * W + ox] = s[y * w + tx.i0] * tx.l0 + s[y * w + tx.i1] * tx.l1;
  }
  __syncthreads();
  const int W4 = W / 4, Q = H * W4;
  float lo = INFINITY, hi = -INFINITY;
#pragma unroll 2
  for (int q = threadIdx.x; q < Q; q += 256) {
    const int oy = q / W4, x4 = (q - oy * W4) * 4;
    const Tap ty = make_tap(oy, h, rh);
    const fx4 a = *(const fx4*)(tmp + ty.i0 * W + x4), b = *(const fx4*)(tmp + ty.i1 * W + x4);
    const fx4 v = a * ty.l0 + b * ty.l1;
    lo = fminf(fminf(lo, v.x), fminf(v.y, fminf(v.z, v.w)));
    hi = fmaxf(fmaxf(hi, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
  }
  bminmax<256>(lo, hi, red);
  const float span = hi - lo;
  fx4* dst = (fx4*)(out + (long)map * H * W);
  const int per = (Q + SPLIT - 1) / SPLIT, q0 = part * per, q1 = min(q0 + per, Q);
#pragma unroll 2
  for (int q = q0 + threadIdx.x; q < q1; q += 256) {
    const int oy = q / W4, x4 = (q - oy * W4) * 4;
    const Tap ty = make_tap(oy, h, rh);
    const fx4 a = *(const fx4*)(tmp + ty.i0 * W + x4), b = *(const fx4*)(tmp + ty.i1 * W + x4);
    fx4 v = a * ty.l0 + b * ty.l1;
    v.x = (v.x - lo) / span; v.y = (v.y - lo) / span; v.z = (v.z - lo) / span; v.w = (v.w - lo) / span;
    if (NT) __builtin_nontemporal_store(v, dst + q); else dst[q] = v;
  }
}

// V2: V1 with the (row, column) of a lane's next quad tracked incrementally (no integer division per quad).
template <int BLOCK, bool NT, int UNROLL, int MODE = 0>
__global__ __launch_bounds__(BLOCK) void v_quads_inc(const float* __restrict__ src, int h, int w, int H, int W, float* __restrict__ out) {
  extern __shared__ float tmp[];
  __shared__ float s[4096];
  __shared__ float red[2 * (BLOCK / 64)];
  const int map = blockIdx.x;
  const float* row = src + (long)map * h * w;
  for (int i = threadIdx.x; i < h * w; i += BLOCK) s[i] = row[i];
  __syncthreads();
  const float rh = (float)h / (float)H, rw = (float)w / (float)W;
  for (int ox = threadIdx.x; ox < W; ox += BLOCK) {
    const Tap tx = make_tap(ox, w, rw);
    for (int y = 0; y < h; ++y) tmp[y * W + ox] = s[y * w + tx.i0] * tx.l0 + s[y * w + tx.i1] * tx.l1;
  }
  __syncthreads();
  const int W4 = W / 4, Q = H * W4;
  const int d_row = BLOCK / W4, d_col = BLOCK % W4;
  float lo = INFINITY, hi = -INFINITY;
  if (MODE == 2) { lo = 0.f; hi = 1.f; }
  if (MODE != 2) {
    int oy = threadIdx.x / W4, c = threadIdx.x % W4;
#pragma unroll UNROLL
    for (int q = threadIdx.x; q < Q; q += BLOCK) {
      const Tap ty = make_tap(oy, h, rh);
      const fx4 a = *(const fx4*)(tmp + ty.i0 * W + 4 * c), b = *(const fx4*)(tmp + ty.i1 * W + 4 * c);
      const fx4 v = a * ty.l0 + b * ty.l1;
      lo = fminf(fminf(lo, v.x), fminf(v.y, fminf(v.z, v.w)));
      hi = fmaxf(fmaxf(hi, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
      oy += d_row; c += d_col;
      if (c >= W4) { c -= W4; ++oy; }
    }
  }
  lo = wmin(lo); hi = wmax(hi);
  const int wave = threadIdx.x >> 6, nw = BLOCK >> 6;
  if ((threadIdx.x & 63) == 0) { red[wave] = lo; red[nw + wave] = hi; }
  __syncthreads();
  lo = red[0]; hi = red[nw];
  for (int i = 1; i < nw; ++i) { lo = fminf(lo, red[i]); hi = fmaxf(hi, red[nw + i]); }
  const float span = hi - lo;
  fx4* dst = (fx4*)(out + (long)map * H * W);
  if (MODE == 1) { if (threadIdx.x == 0) out[(long)map * H * W] = span; return; }
  int oy = threadIdx.x / W4, c = threadIdx.x % W4;
#pragma unroll UNROLL
  for (int q = threadIdx.x; q < Q; q += BLOCK) {
    const Tap ty = make_tap(oy, h, rh);
    const fx4 a = *(const fx4*)(tmp + ty.i0 * W + 4 * c), b = *(const fx4*)(tmp + ty.i1 * W + 4 * c);
    fx4 v = a * ty.l0 + b * ty.l1;
    v.x = (v.x - lo) / span; v.y = (v.y - lo) / span; v.z = (v.z - lo) / span; v.w = (v.w - lo) / span;
    if (NT) __builtin_nontemporal_store(v, dst + q); else dst[q] = v;
    oy += d_row; c += d_col;
    if (c >= W4) { c -= W4; ++oy; }
  }
}

// V3: V2 + per-row tap table in LDS + one correctly rounded reciprocal per map and a Markstein step per element
//     (q = a*y; r = fma(-q, b, a); q' = fma(r, y, q) is the correctly rounded a/b when y = RN(1/b)).
template <int BLOCK, bool NT, int UNROLL>
__global__ __launch_bounds__(BLOCK) void v_quads_tab(const float* __restrict__ src, int h, int w, int H, int W, float* __restrict__ out) {
  extern __shared__ float tmp[];                 // [h][W] then H taps of 4 words
  __shared__ float s[4096];
  __shared__ float red[2 * (BLOCK / 64)];
  fx4* taps = (fx4*)(tmp + h * W);
  const int map = blockIdx.x;
  const float* row = src + (long)map * h * w;
  for (int i = threadIdx.x; i < h * w; i += BLOCK) s[i] = row[i];
  const float rh = (float)h / (float)H, rw = (float)w / (float)W;
  for (int oy = threadIdx.x; oy < H; oy += BLOCK) {
    const Tap t = make_tap(oy, h, rh);
    fx4 e; e.x = __int_as_float(t.i0 * W); e.y = __int_as_float(t.i1 * W); e.z = t.l0; e.w = t.l1;
    taps[oy] = e;
  }
  __syncthreads();
  for (int ox = threadIdx.x; ox < W; ox += BLOCK) {
    const Tap tx = make_tap(ox, w, rw);
    for (int y = 0; y < h; ++y) tmp[y * W + ox] = s[y * w + tx.i0] * tx.l0 + s[y * w + tx.i1] * tx.l1;
  }
  __syncthreads();
  const int W4 = W / 4, Q = H * W4;
  const int d_row = BLOCK / W4, d_col = BLOCK % W4;
  float lo = INFINITY, hi = -INFINITY;
  {
    int oy = threadIdx.x / W4, c = threadIdx.x % W4;
#pragma unroll UNROLL
    for (int q = threadIdx.x; q < Q; q += BLOCK) {
      const fx4 t = taps[oy];
      const fx4 a = *(const fx4*)(tmp + __float_as_int(t.x) + 4 * c), b = *(const fx4*)(tmp + __float_as_int(t.y) + 4 * c);
      const fx4 v = a * t.z + b * t.w;
      lo = fminf(fminf(lo, v.x), fminf(v.y, fminf(v.z, v.w)));
      hi = fmaxf(fmaxf(hi, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
      oy += d_row; c += d_col;
      if (c >= W4) { c -= W4; ++oy; }
    }
  }
  lo = wmin(lo); hi = wmax(hi);
  const int wave = threadIdx.x >> 6, nw = BLOCK >> 6;
  if ((threadIdx.x & 63) == 0) { red[wave] = lo; red[nw + wave] = hi; }
  __syncthreads();
  lo = red[0]; hi = red[nw];
  for (int i = 1; i < nw; ++i) { lo = fminf(lo, red[i]); hi = fmaxf(hi, red[nw + i]); }
  const float span = hi - lo, y = 1.f / span;
  fx4* dst = (fx4*)(out + (long)map * H * W);
  int oy = threadIdx.x / W4, c = threadIdx.x % W4;
#pragma unroll UNROLL
  for (int q = threadIdx.x; q < Q; q += BLOCK) {
    const fx4 t = taps[oy];
    const fx4 a = *(const fx4*)(tmp + __float_as_int(t.x) + 4 * c), b = *(const fx4*)(tmp + __float_as_int(t.y) + 4 * c);
    fx4 v = a * t.z + b * t.w;
    float* e = (float*)&v;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float num = e[k] - lo;
      const float q0 = num * y;
      const float r = __builtin_fmaf(-q0, span, num);
      e[k] = __builtin_fmaf(r, y, q0);
    }
    if (NT) __builtin_nontemporal_store(v, dst + q); else dst[q] = v;
    oy += d_row; c += d_col;
    if (c >= W4) { c -= W4; ++oy; }
  }
}

// ceiling: plain fill of the same bytes
__global__ __launch_bounds__(256) void fill(fx4* __restrict__ out, long n4) {
  const fx4 v = {1, 2, 3, 4};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) __builtin_nontemporal_store(v, out + i);
}

template <typename F> double time_us(F&& launch) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) launch();
  std::vector<float> t;
  for (int i = 0; i < 15; ++i) { CK(hipEventRecord(a, 0)); for (int j = 0; j < 10; ++j) launch(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms / 10); }
  std::sort(t.begin(), t.end()); return t[7] * 1e3;
}

int main() {
  const int R = 768, h = 14, w = 14, H = 224, W = 224;
  float *src, *out, *ref;
  CK(hipMalloc(&src, (size_t)R * h * w * 4)); CK(hipMalloc(&out, (size_t)R * H * W * 4)); CK(hipMalloc(&ref, (size_t)R * H * W * 4));
  std::vector<float> hs((size_t)R * h * w);
  for (size_t i = 0; i < hs.size(); ++i) hs[i] = (float)((i * 2654435761u) % 10007) / 10007.f - 0.5f;
  CK(hipMemcpy(src, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
  const size_t lds = (size_t)h * W * 4;
  const double mb = (double)R * H * W * 4 / 1e6;
  auto rep = [&](const char* name, double us) { printf("%-44s %8.1f us  %7.1f GB/s\n", name, us, mb / us * 1e3); };
  hipLaunchKernelGGL((v_rows<true, true, 1>), dim3(R), dim3(256), lds, 0, src, h, w, H, W, ref);
  CK(hipDeviceSynchronize());
  std::vector<float> a((size_t)R * H * W), b((size_t)R * H * W);
  CK(hipMemcpy(a.data(), ref, a.size() * 4, hipMemcpyDeviceToHost));
  auto check = [&](const char* name) {
    CK(hipDeviceSynchronize()); CK(hipMemcpy(b.data(), out, b.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0; for (size_t i = 0; i < a.size(); ++i) bad += a[i] != b[i];
    if (bad) printf("   %s: %zu elements differ from the baseline mapping\n", name, bad);
  };
  rep("fill (ceiling)", time_us([&] { hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, (fx4*)out, (long)R * H * W / 4); }));
  rep("rows nt div", time_us([&] { hipLaunchKernelGGL((v_rows<true, true, 1>), dim3(R), dim3(256), lds, 0, src, h, w, H, W, out); })); check("rows nt div");
  rep("rows plain-store div", time_us([&] { hipLaunchKernelGGL((v_rows<false, true, 1>), dim3(R), dim3(256), lds, 0, src, h, w, H, W, out); })); check("rows plain");
  rep("rows nt reciprocal (not bit-exact)", time_us([&] { hipLaunchKernelGGL((v_rows<true, false, 1>), dim3(R), dim3(256), lds, 0, src, h, w, H, W, out); }));
  rep("rows nt div split2", time_us([&] { hipLaunchKernelGGL((v_rows<true, true, 2>), dim3(R * 2), dim3(256), lds, 0, src, h, w, H, W, out); })); check("rows split2");
  rep("rows nt div split4", time_us([&] { hipLaunchKernelGGL((v_rows<true, true, 4>), dim3(R * 4), dim3(256), lds, 0, src, h, w, H, W, out); })); check("rows split4");
  rep("quads nt", time_us([&] { hipLaunchKernelGGL((v_quads<true, 1>), dim3(R), dim3(256), lds, 0, src, h, w, H, W, out); })); check("quads nt");
  rep("quads plain-store", time_us([&] { hipLaunchKernelGGL((v_quads<false, 1>), dim3(R), dim3(256), lds, 0, src, h, w, H, W, out); })); check("quads plain");
  rep("quads nt split2", time_us([&] { hipLaunchKernelGGL((v_quads<true, 2>), dim3(R * 2), dim3(256), lds, 0, src, h, w, H, W, out); })); check("quads split2");
  rep("quads nt split4", time_us([&] { hipLaunchKernelGGL((v_quads<true, 4>), dim3(R * 4), dim3(256), lds, 0, src, h, w, H, W, out); })); check("quads split4");
  rep("quads-inc 256 nt u1", time_us([&] { hipLaunchKernelGGL((v_quads_inc<256, true, 1>), dim3(R), dim3(256), lds, 0, src, h, w, H, W, out); })); check("qi 256 u1");
  rep("quads-inc 256 nt u2", time_us([&] { hipLaunchKernelGGL((v_quads_inc<256, true, 2>), dim3(R), dim3(256), lds, 0, src, h, w, H, W, out); })); check("qi 256 u2");
  rep("quads-inc 256 nt u4", time_us([&] { hipLaunchKernelGGL((v_quads_inc<256, true, 4>), dim3(R), dim3(256), lds, 0, src, h, w, H, W, out); })); check("qi 256 u4");
  rep("quads-inc 512 nt u2", time_us([&] { hipLaunchKernelGGL((v_quads_inc<512, true, 2>), dim3(R), dim3(512), lds, 0, src, h, w, H, W, out); })); check("qi 512 u2");
  rep("quads-inc 1024 nt u2", time_us([&] { hipLaunchKernelGGL((v_quads_inc<1024, true, 2>), dim3(R), dim3(1024), lds, 0, src, h, w, H, W, out); })); check("qi 1024 u2");
  rep("quads-inc 512 plain u2", time_us([&] { hipLaunchKernelGGL((v_quads_inc<512, false, 2>), dim3(R), dim3(512), lds, 0, src, h, w, H, W, out); })); check("qi 512 plain");
  rep("quads-inc 256 pass-1 only", time_us([&] { hipLaunchKernelGGL((v_quads_inc<256, true, 2, 1>), dim3(R), dim3(256), lds, 0, src, h, w, H, W, out); }));
  rep("quads-inc 256 pass-2 only", time_us([&] { hipLaunchKernelGGL((v_quads_inc<256, true, 2, 2>), dim3(R), dim3(256), lds, 0, src, h, w, H, W, out); }));
  const size_t lds2 = lds + (size_t)H * 16;
  rep("quads-tab 256 nt u1", time_us([&] { hipLaunchKernelGGL((v_quads_tab<256, true, 1>), dim3(R), dim3(256), lds2, 0, src, h, w, H, W, out); })); check("qt 256 u1");
  rep("quads-tab 256 nt u2", time_us([&] { hipLaunchKernelGGL((v_quads_tab<256, true, 2>), dim3(R), dim3(256), lds2, 0, src, h, w, H, W, out); })); check("qt 256 u2");
  rep("quads-tab 512 nt u2", time_us([&] { hipLaunchKernelGGL((v_quads_tab<512, true, 2>), dim3(R), dim3(512), lds2, 0, src, h, w, H, W, out); })); check("qt 512 u2");
  rep("quads-tab 256 plain u2", time_us([&] { hipLaunchKernelGGL((v_quads_tab<256, false, 2>), dim3(R), dim3(256), lds2, 0, src, h, w, H, W, out); })); check("qt 256 plain");
  return 0;
}
