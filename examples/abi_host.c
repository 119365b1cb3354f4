/* A plain-C host of the C ABI (include/xai_hip.h): no torch, no C++ -- device memory from the HIP runtime, the library's entry
 * points called exactly as a cgo / JNI / ctypes binding would call them, results checked against the expressions of the
 * reference they replace.  Shows the boundary of DESIGN.md section 1: extern "C", device pointers + extents + a stream.
 *
 *   gcc -std=c99 -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/abi_host.c \
 *       -Limage-classification-xai_amd/xai_engine/lib -lxai_hip -L/opt/rocm/lib -lamdhip64 -lm \
 *       -Wl,-rpath,$PWD/image-classification-xai_amd/xai_engine/lib -Wl,-rpath,/opt/rocm/lib -o /tmp/abi_host && /tmp/abi_host
 *
 * 1. IG accumulate (saliencyMethods.py:53,70): out = mean_s(grads) * (x - baseline), 2 images x 7 steps x 3 x 24 x 24
 * 2. insertion/deletion step images (MASTestFunctions.py:209,249-257): rank a map, flip-step per pixel, materialise a batch
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "xai_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define XK(x) do { int r_ = (x); if (r_ != 0) { printf("%s -> %d (%s)\n", #x, r_, xai_strerror(r_)); return 1; } } while (0)

static uint32_t lcg_state = 12345u;
static float rnd(void) { lcg_state = lcg_state * 1664525u + 1013904223u; return (float)((int32_t)(lcg_state >> 8) - (1 << 23)) / (float)(1 << 22); }

int main(void) {
  /* major must match the header this host was compiled against; the library may be a later minor (entry points only get added) */
  if (xai_version() != XAI_ABI_VERSION || xai_version_minor() < XAI_ABI_MINOR) {
    printf("libxai_hip.so has ABI %d.%d, this host was built against %d.%d\n", xai_version(), xai_version_minor(), XAI_ABI_VERSION, XAI_ABI_MINOR);
    return 1;
  }
  hipStream_t st;
  CK(hipStreamCreate(&st));

  /* ---- 1. IG accumulate --------------------------------------------------------------------------------------------- */
  enum { N_IMG = 2, S = 7, C = 3, HW = 24 * 24 };
  const size_t n_g = (size_t)N_IMG * S * C * HW, n_x = (size_t)N_IMG * C * HW;
  float *h_g = malloc(n_g * 4), *h_x = malloc(n_x * 4), *h_out = malloc(n_x * 4), *h_abs = malloc((size_t)N_IMG * HW * 4);
  for (size_t i = 0; i < n_g; ++i) h_g[i] = rnd();
  for (size_t i = 0; i < n_x; ++i) h_x[i] = rnd();
  float *d_g, *d_x, *d_out, *d_abs;
  CK(hipMalloc((void**)&d_g, n_g * 4)); CK(hipMalloc((void**)&d_x, n_x * 4)); CK(hipMalloc((void**)&d_out, n_x * 4)); CK(hipMalloc((void**)&d_abs, (size_t)N_IMG * HW * 4));
  CK(hipMemcpy(d_g, h_g, n_g * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_x, h_x, n_x * 4, hipMemcpyHostToDevice));
  const float baseline = 0.25f;
  XK(xai_ig_accum_f32(d_g, N_IMG, S, NULL, S, NULL, NULL, d_x, NULL, baseline, C, HW, d_out, d_abs, st));
  CK(hipStreamSynchronize(st));
  CK(hipMemcpy(h_out, d_out, n_x * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h_abs, d_abs, (size_t)N_IMG * HW * 4, hipMemcpyDeviceToHost));
  double worst = 0, scale = 0;
  for (int i = 0; i < N_IMG; ++i)
    for (int p = 0; p < HW; ++p) {
      float chsum = 0.f;
      for (int c = 0; c < C; ++c) {
        float acc = 0.f;                                   /* fp32, steps ascending: the order the kernel promises */
        for (int s = 0; s < S; ++s) acc += h_g[(((size_t)i * S + s) * C + c) * HW + p];
        const float want = acc / (float)S * (h_x[((size_t)i * C + c) * HW + p] - baseline);
        const double d = fabs((double)want - h_out[((size_t)i * C + c) * HW + p]);
        if (d > worst) worst = d;
        if (fabs(want) > scale) scale = fabs(want);
        chsum += h_out[((size_t)i * C + c) * HW + p];
      }
      if (fabs(fabsf(chsum) - h_abs[(size_t)i * HW + p]) > 1e-6 * (1 + fabs(chsum))) { printf("abs map differs at %d %d\n", i, p); return 1; }
    }
  printf("ig_accum: max |diff| / max |want| = %.3g\n", worst / scale);
  if (worst / scale > 2e-6) return 1;

  /* ---- 2. rank -> flip steps -> one batch of deletion images ------------------------------------------------------------ */
  enum { STEP = 64, N_STEPS = HW / STEP };
  float *h_sal = malloc(HW * 4), *h_start = malloc((size_t)C * HW * 4), *h_imgs = malloc((size_t)N_STEPS * C * HW * 4);
  int32_t *h_order = malloc(HW * 4), *h_flip = malloc(HW * 4);
  for (int p = 0; p < HW; ++p) h_sal[p] = rnd();
  for (int i = 0; i < C * HW; ++i) h_start[i] = 1.f + rnd();
  float *d_sal, *d_start, *d_finish, *d_imgs; int32_t *d_order, *d_rank, *d_flip; void* d_ws;
  const size_t ws_bytes = xai_rank_workspace_bytes(1, HW);
  CK(hipMalloc((void**)&d_sal, HW * 4)); CK(hipMalloc((void**)&d_start, (size_t)C * HW * 4)); CK(hipMalloc((void**)&d_finish, (size_t)C * HW * 4));
  CK(hipMalloc((void**)&d_imgs, (size_t)N_STEPS * C * HW * 4)); CK(hipMalloc((void**)&d_order, HW * 4)); CK(hipMalloc((void**)&d_rank, HW * 4));
  CK(hipMalloc((void**)&d_flip, HW * 4)); CK(hipMalloc(&d_ws, ws_bytes));
  CK(hipMemcpy(d_sal, h_sal, HW * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_start, h_start, (size_t)C * HW * 4, hipMemcpyHostToDevice));
  CK(hipMemset(d_finish, 0, (size_t)C * HW * 4));                                   /* substrate = zeros (deletion) */
  XK(xai_rank_f32(d_sal, 1, HW, d_order, d_rank, d_ws, ws_bytes, st));
  XK(xai_flip_steps_i32(d_rank, HW, /*descending=*/1, STEP, d_flip, st));
  XK(xai_perturb_batch_f32(d_start, d_finish, d_flip, C, HW, /*first_step=*/0, N_STEPS, d_imgs, st));
  CK(hipStreamSynchronize(st));
  CK(hipMemcpy(h_order, d_order, HW * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h_flip, d_flip, HW * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(h_imgs, d_imgs, (size_t)N_STEPS * C * HW * 4, hipMemcpyDeviceToHost));
  for (int i = 1; i < HW; ++i)                                                      /* ascending, stable */
    if (h_sal[h_order[i - 1]] > h_sal[h_order[i]] || (h_sal[h_order[i - 1]] == h_sal[h_order[i]] && h_order[i - 1] > h_order[i])) { printf("order wrong at %d\n", i); return 1; }
  for (int k = 0; k < N_STEPS; ++k) {                                                /* step k: the (k+1)*STEP most salient pixels are zero, the rest untouched */
    int zeros = 0;
    for (int p = 0; p < HW; ++p) {
      const int gone = h_flip[p] <= k;
      zeros += gone;
      for (int c = 0; c < C; ++c) {
        const float v = h_imgs[((size_t)k * C + c) * HW + p];
        if (v != (gone ? 0.f : h_start[(size_t)c * HW + p])) { printf("step image %d wrong at pixel %d\n", k, p); return 1; }
      }
    }
    if (zeros != (k + 1) * STEP) { printf("step %d removed %d pixels\n", k, zeros); return 1; }
  }
  /* argument errors come back as codes, never as crashes */
  if (xai_ig_accum_f32(NULL, 1, 1, NULL, 1, NULL, NULL, d_x, NULL, 0.f, C, HW, d_out, NULL, st) != XAI_E_NULL) return 1;
  printf("perturb: %d step images exact; abi host ok\n", N_STEPS);
  return 0;
}
