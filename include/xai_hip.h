/*
 * xai_hip.h -- C ABI of libxai_hip.so, the MI355X (gfx950) kernels behind the
 * saliency-attribution hot path of chasewalker26/Image-Classification-XAI.
 *
 * The reference is pure Python and has NO FFI for this path: its "interface" is a set of
 * torch / NumPy expressions inside util/attribution_methods and util/test_methods.  Each
 * entry point below names the reference expression (file:line, relative to the reference
 * root) it replaces; INTEGRATION.md shows the ctypes stub a maintainer would add at that
 * line.  The Python modules under image-classification-xai_amd/util/ keep the reference's
 * call signatures and are the only intended callers.
 *
 * Conventions (all entry points)
 *   - every pointer is a DEVICE pointer owned by the caller (tensor.data_ptr() of a
 *     contiguous tensor); the library never allocates, frees or retains memory;
 *   - `stream` is the caller's hipStream_t (torch.cuda.current_stream().cuda_stream);
 *     launches are asynchronous on it and graph-capturable (no sync, no malloc inside);
 *   - float data is IEEE fp32, index data int32, layouts are dense row-major
 *     (NCHW for images, [image][step][C][H*W] for step batches);
 *   - return value: 0 = success; <0 = argument error (XAI_E_*); >0 = hipError_t of the launch;
 *   - no global mutable state beyond a per-device compute-unit count published once through a std::atomic;
 *     re-entrant from several host threads on different streams.
 */
#ifndef XAI_HIP_H
#define XAI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* xai_stream_t; /* hipStream_t */

/* xai_version() = XAI_ABI_VERSION: bumped when an existing prototype or its documented meaning changes.
 * xai_version_minor() = XAI_ABI_MINOR: bumped whenever entry points are added or an accepted argument range grows, so a
 * host can tell an older libxai_hip.so from this one without probing symbols:
 *   1 = the round-1 set;  2 = + xai_ig_accum_timed_f32, xai_maxpool_bwd_f32 accepts more than 65 535 planes;
 *   3 = + xai_version_minor, xai_masked_sums_f32 */
#define XAI_ABI_VERSION 1
#define XAI_ABI_MINOR 3
#define XAI_OK 0
#define XAI_E_NULL (-1)        /* required pointer is NULL                      */
#define XAI_E_SHAPE (-2)       /* non-positive / inconsistent extent, or misaligned */
#define XAI_E_UNSUPPORTED (-3) /* extent beyond what the kernel was built for   */

int xai_version(void);
int xai_version_minor(void);
/* static string for a code returned by any entry point (never NULL) */
const char* xai_strerror(int code);

/* ---- Integrated Gradients -------------------------------------------------------- */

/* K1  out[i][s][e] = b[i][e] + alpha[i][s] * (x[i][e] - b[i][e])   (mul, then add: no FMA)
 * replaces  saliencyMethods.py:38,44  (also :113, :169, :245)
 *   x, baseline : [n_img][n_elem]; baseline may be NULL -> every element = baseline_scalar
 *   alphas      : n_alpha values per image, image i at alphas + i*alpha_img_stride
 *                 (alpha_img_stride = 0 shares one schedule)
 *   out         : [n_img][n_alpha][n_elem] */
int xai_ig_interp_f32(const float* x, const float* baseline, float baseline_scalar,
                      const float* alphas, int64_t alpha_img_stride, int n_img, int n_alpha,
                      int64_t n_elem, float* out, xai_stream_t stream);

/* Left-IG cutoff: n_use[i] = first s with logit[i][s] > alpha_star * max_s logit[i][s];
 * none -> 1; 0 -> 1.   replaces  saliencyMethods.py:48-65   (alpha_star == 1 -> n_steps) */
int xai_ig_cutoff_f32(const float* logits, int n_img, int n_steps, float alpha_star,
                      int32_t* n_use, xai_stream_t stream);

/* K2  out[i][c][p] = ( sum_{s<n_use_i} w1[i][s]*w2[i][s]*g[i][s][c][p] ) / denom_i * (x - b)
 *     out_abs[i][p] = | sum_c out[i][c][p] |                     (optional, fused a5)
 * replaces  saliencyMethods.py:53,67,70 (IG / Left-IG), :125-135 (IDG, with w1 = slopes,
 *           w2 = alpha sub-step, denom = n_steps) and evaluatePerturbation.py:181 (abs-sum)
 *   grads      : [n_img][n_steps][C][hw]
 *   n_use_dev  : per-image count on the device (NULL -> n_use_host for every image);
 *                denom_i = n_use_i (mean over the prefix)
 *   step_w1/2  : [n_img][n_steps] optional per-step weights (NULL -> 1)
 *   out_chw    : [n_img][C][hw];  out_abs_hw : [n_img][hw] or NULL */
int xai_ig_accum_f32(const float* grads, int n_img, int n_steps, const int32_t* n_use_dev,
                     int n_use_host, const float* step_w1, const float* step_w2,
                     const float* x, const float* baseline, float baseline_scalar, int C,
                     int64_t hw, float* out_chw, float* out_abs_hw, xai_stream_t stream);

/* K2, the same launch (saliencyMethods.py:53,67,70), with the kernel's OWN start / stop timestamps recorded into two
 * caller-owned hipEvent_t (created with timing enabled) by the dispatch itself (hipExtLaunchKernel): hipEventElapsedTime of
 * the pair is the kernel's duration, without the dispatch latency that two events bracketing a launch include.  Used by
 * bench.py for roofline.achieved.  start_event / stop_event : hipEvent_t, both required */
int xai_ig_accum_timed_f32(const float* grads, int n_img, int n_steps, const int32_t* n_use_dev,
                           int n_use_host, const float* step_w1, const float* step_w2,
                           const float* x, const float* baseline, float baseline_scalar, int C,
                           int64_t hw, float* out_chw, float* out_abs_hw, void* start_event,
                           void* stop_event, xai_stream_t stream);

/* dst[e] = src[e] with non-temporal stores: files one classifier pass's step gradients into
 * the [n_img][n_steps][C][hw] buffer without leaving dirty lines in the Infinity Cache
 * replaces  `gradients[start:end] = ...` at saliencyMethods.py:46 */
int xai_ig_store_grads_f32(const float* src, float* dst, int64_t n_elem, xai_stream_t stream);

/* K2, streaming form: acc[e] += sum_{b<n_batch} grads[b][e]   (no [steps][N] buffer kept)
 * replaces  saliencyMethods.py:46 + :53 when alpha_star == 1 */
int xai_ig_accum_add_f32(const float* grads, int n_batch, float* acc, int64_t n_elem,
                         xai_stream_t stream);

/* finish of the streaming form: out = acc / n_steps * (x - b); optional abs-sum as above */
int xai_ig_finish_f32(const float* acc, int n_img, int n_steps, const float* x,
                      const float* baseline, float baseline_scalar, int C, int64_t hw,
                      float* out_chw, float* out_abs_hw, xai_stream_t stream);

/* IDGI  sumsq[s] = sum_e g[s][e]^2 ;  out[e] = sum_{s<n_steps-1} g[s][e]^2 * d[s] / sumsq[s]
 * replaces  saliencyMethods.py:172-179    (d[s] = logit[s+1]-logit[s], computed in-kernel) */
int xai_sumsq_f32(const float* grads, int n_rows, int64_t n_elem, float* sumsq,
                  xai_stream_t stream);
int xai_idgi_accum_f32(const float* grads, int n_steps, const float* logits,
                       const float* sumsq, int64_t n_elem, float* out, xai_stream_t stream);

/* ---- Grad-CAM --------------------------------------------------------------------- */

/* K3a  w[c] = mean_hw grad[b][c]; cam[b][p] = (relu) sum_c w[c]*act[b][c][p]
 * replaces  captum 0.7.0 LayerGradCam.attribute as called at evaluatePerturbation.py:149-151
 *   act, grad : [B][C][h*w];  cam : [B][h*w];  h*w <= 1024
 *   ws : optional device scratch of xai_gradcam_workspace_bytes(B,C,h,w) bytes; with it the channels
 *        of one image are reduced by several workgroups (partials summed in a fixed order);
 *        NULL -> one workgroup per image */
size_t xai_gradcam_workspace_bytes(int B, int C, int h, int w);
int xai_gradcam_f32(const float* act, const float* grad, int B, int C, int h, int w, int relu,
                    float* cam, void* ws, size_t ws_bytes, xai_stream_t stream);

/* K3b  bilinear up-sample, align_corners = False; dst = scale * up(src), |.| if take_abs
 * replaces  torchvision Resize((H,W), antialias=True) at evaluatePerturbation.py:153 and
 *           (scale = 3, take_abs = 1) the x ones(3,H,W) + abs-sum of :153,:181 */
int xai_bilinear_up_f32(const float* src, int B, int h, int w, int H, int W, float scale,
                        int take_abs, float* dst, xai_stream_t stream);

/* ---- RISE ------------------------------------------------------------------------- */

/* K4  mask_n = crop(upsample(grid_n, (s+1)*cell), shift_n, HxW);  masked_n = image * mask_n
 * replaces  generate_emap.py:72-80 (skimage resize order=1 'reflect' + shift crop) and :91
 *   grid : [n][s][s] uint8 {0,1};  shift : [n][2] int32 (row shift, col shift)
 *   image : [C][H][W];  masked_out : [n][C][H][W] or NULL;  masks_out : [n][H][W] or NULL */
int xai_rise_apply_f32(const uint8_t* grid, const int32_t* shift, int n_masks, int s,
                       int cell_h, int cell_w, const float* image, int C, int H, int W,
                       float* masked_out, float* masks_out, xai_stream_t stream);

/* K5  acc[p] += scale * sum_n scores[n] * mask_n[p]     (masks regenerated from the grid;
 *     fp64 accumulator that may be carried over several calls; the caller rounds to fp32)
 * replaces  generate_emap.py:99-100  (scale = 1/N/p1) */
int xai_rise_accum_f64(const uint8_t* grid, const int32_t* shift, const float* scores,
                       int n_masks, int s, int cell_h, int cell_w, int H, int W, double scale,
                       double* acc, xai_stream_t stream);

/* ---- insertion / deletion loop ------------------------------------------------------ */

/* K8  stable ascending argsort of each row (NumPy kind='stable': -0 == +0, NaN last) and its
 *     inverse permutation.   replaces  MASTestFunctions.py:209,212 (np.argsort / np.flip)
 *   sal : [n_seg][hw];  order, rank : [n_seg][hw] int32;  descending order = reverse of
 *   `order`.  ws : device scratch of at least xai_rank_workspace_bytes(n_seg, hw) bytes */
size_t xai_rank_workspace_bytes(int n_seg, int64_t hw);
int xai_rank_f32(const float* sal, int n_seg, int64_t hw, int32_t* order, int32_t* rank, void* ws,
                 size_t ws_bytes, xai_stream_t stream);

/* flip_step[p] = (descending ? hw-1-rank[p] : rank[p]) / step_size : the 0-based step at
 * which pixel p switches from `start` to `finish`   (MASTestFunctions.py:251) */
int xai_flip_steps_i32(const int32_t* rank, int64_t hw, int descending, int step_size,
                       int32_t* flip_step, xai_stream_t stream);

/* K6  out[k][c][p] = flip_step[p] <= first_step + k ? finish[c][p] : start[c][p]
 * replaces  MASTestFunctions.py:249-257 (and the same loop in RISE:181, AIC:179, PNP:141,
 *           MONO:173): the NumPy fancy-index copy + images[i] = start
 *   start, finish : [C][hw];  out : [n_batch][C][hw] */
int xai_perturb_batch_f32(const float* start, const float* finish, const int32_t* flip_step,
                          int C, int64_t hw, int first_step, int n_batch, float* out,
                          xai_stream_t stream);

/* K10 seg[t] = float32 sum of sal over the pixels of step t, i.e. positions
 *     [t*step_size, (t+1)*step_size) of the ascending `order` (read back to front when
 *     descending); total = float32 sum of the whole map; n_steps = ceil(hw/step_size)
 * replaces  MASTestFunctions.py:227,259   (density numerators) */
int xai_segment_sums_f32(const float* sal, const int32_t* order, int64_t hw, int descending,
                         int step_size, int n_steps, float* seg, float* total,
                         xai_stream_t stream);

/* K7  separable zero-padded blur: out = k1d (x) k1d applied per channel
 * replaces  conv2d(x, gkern(klen, nsig), padding=klen//2) at evaluatePerturbation.py:459
 *   x, out : [B][C][H][W];  k1d : [klen] on the device, klen odd <= 63 */
int xai_blur_sep_f32(const float* x, const float* k1d, int klen, int B, int C, int H, int W,
                     float* out, xai_stream_t stream);

/* one zero-padded 1-D pass of the same blur along W (axis = 1) or H (axis = 0), any odd klen; two calls
 * (through a caller buffer, x != out) give the separable blur for the long kernels of the growing-kernel
 * search in the MDA branch, evaluatePerturbation.py:244-257 (klen += 4 up to 101) */
int xai_blur_1d_f32(const float* x, const float* k1d, int klen, int axis, int B, int C, int H,
                    int W, float* out, xai_stream_t stream);

/* K9  per row of logits: softmax[target], -sum p log2 p, argmax
 * replaces  MASTestFunctions.py:273-276, AICTestFunctions.py:191-192
 *   target_dev : device int32 (NULL -> target_host; target_host < 0 -> each row's argmax)
 *   any of p_target / entropy_bits / argmax may be NULL */
int xai_softmax_stats_f32(const float* logits, int B, int K, const int32_t* target_dev,
                          int target_host, float* p_target, float* entropy_bits,
                          int32_t* argmax, xai_stream_t stream);

/* ---- feature-map maskers of the RISE family (ViT-CX) ---------------------------------- */

/* K11 masks[r] = minmax_row( bilinear_up(src[r], (H,W), align_corners = False) ), one launch, the up-sampled
 *     maps never exist un-normalised in memory
 * replaces  ViT_CX/ViT_CX.py:82-84 (transforms.Resize(input_size, antialias=True), then norm_matrix :29-34)
 *   src : [R][h*w], h*w <= 4096, h*W <= 8192, H <= 1024;  out : [R][H*W];  a constant row gives 0/0 = NaN as in the reference */
int xai_up_rownorm_f32(const float* src, int R, int h, int w, int H, int W, float* out,
                       xai_stream_t stream);

/* K12 out[r] = (x[r] - min x[r]) / (max x[r] - min x[r])      (in place allowed)
 * replaces  norm_matrix, ViT_CX/ViT_CX.py:29-34, as applied to the cluster sums at :109 */
int xai_rownorm_f32(const float* x, int R, int64_t P, float* out, xai_stream_t stream);

/* K13 out[k] = sum of rows[members[m]] for m in [offs[k], offs[k+1]), added in that order starting from 0
 * replaces  the `mask_clustering[cluster_labels[i]] += mask[i]` loop, ViT_CX/ViT_CX.py:105-106
 *   rows : [R][P];  members : int32 row ids grouped by cluster (ascending inside a cluster);
 *   offs : [K+1] int32;  out : [K][P] */
int xai_cluster_sum_f32(const float* rows, const int32_t* members, const int32_t* offs, int K,
                        int64_t P, float* out, xai_stream_t stream);

/* K14 add = (noise[n][c][p] * noise_scale) * (1 - masks[n][p]);
 *     stack[n][c][p] = x[c][p] * masks[n][p] + add;   stack[N + n][c][p] = x[c][p] + add
 * replaces  ViT_CX/causal_score.py:27-47 (masks_inverse, random_whole * 0.1, the per-mask loop, torch.cat)
 *   x : [C][HW];  masks : [N][HW];  noise : [N][C][HW] standard normal draws;  stack : [2N][C][HW] */
int xai_causal_apply_f32(const float* x, const float* masks, const float* noise, int N, int C,
                         int64_t HW, float noise_scale, float* stack, xai_stream_t stream);

/* K16 out_weighted[p] = (sum_n weights[n] * rows[n][p]) / N;  out_plain[p] = (sum_n rows[n][p]) / N   -- ONE read of the
 *     stored mask stack (n ascending, fp32, each product rounded before it is added)
 * replaces  (scores * masks).sum / masks.sum of TIS.generate_saliency, util/attribution_methods/TIS.py:331-366, and the
 *           matmul(p_final, masks / masks.sum(0)) of ViT_CX/causal_score.py:54-61 restricted to the one class row consumed
 *   rows : [N][P];  weights : [N];  out_weighted, out_plain : [P] */
int xai_masked_sums_f32(const float* rows, const float* weights, int N, int64_t P, float* out_weighted,
                        float* out_plain, xai_stream_t stream);

/* ---- opt-in classifier-side fusion (xai_engine/prepare.py: fuse_bn_relu) --------------- */

/* y = act( bn(x) [+ identity] ), eval-mode BatchNorm2d with running statistics; act = ReLU when relu != 0 (required with
 * an identity).  Not a replacement of a reference expression: the reference's classifiers are torchvision modules
 * (XAI_Survey/evaluations/evaluatePerturbation.py:627-640) whose BatchNorm2d / ReLU / residual add run as separate
 * PyTorch kernels; this fuses them.   x, identity, y : [N][C][HW];  weight, bias, mean, var : [C]
 *   variant : ordering of the arithmetically equivalent BN expression (see csrc/bnrelu_kernels.hip)
 *   weight2 .. eps2 (nullable as a set): the identity operand is a raw convolution output with its own eval-mode
 *   BatchNorm (the down-sample branch): y = relu( bn(x) + bn2(identity) ) */
int xai_bn_act_fwd_f32(const float* x, const float* identity, const float* weight, const float* bias,
                       const float* mean, const float* var, float eps, const float* weight2,
                       const float* bias2, const float* mean2, const float* var2, float eps2, int variant,
                       int relu, int N, int C, int HW, float* y, xai_stream_t stream);

/* backward of the ReLU form, reached through the autograd.grad of saliencyMethods.py:213 (getGradientsParallel):
 * g = gy [+ gy2];  g1 = y > 0 ? g : 0;  gx = g1 * weight * invstd;  g_identity (nullable) = g1
 *   gy2 (nullable): the second gradient of a block output that feeds both the next convolution and the next identity path
 *   weight2, var2, eps2 (nullable as a set): g_identity = g1 * weight2 * invstd2, the gradient of the identity operand
 *   before its own BatchNorm */
int xai_bn_relu_bwd_f32(const float* gy, const float* gy2, const float* y, const float* weight, const float* var,
                        float eps, const float* weight2, const float* var2, float eps2, int variant, int N,
                        int C, int HW, float* gx, float* g_identity, xai_stream_t stream);

/* MaxPool2d backward from the forward's arg-max indices (int64, h * W + w within a plane), windows added in (ph, pw)
 * ascending order like PyTorch's max_pool_backward_nchw -> bit-identical; the stem of the classifiers instantiated at
 * XAI_Survey/evaluations/evaluatePerturbation.py:627-640.   gy, indices : [planes][PH*PW];  gx : [planes][H*W];
 * any plane count (more than 65 535 planes are launched in slabs) */
int xai_maxpool_bwd_f32(const float* gy, const int64_t* indices, int planes, int H, int W, int PH, int PW,
                        int kernel, int stride, int pad, float* gx, xai_stream_t stream);

/* inference-only stem: y = max_pool2d( relu( bn(x) ), kernel, stride, pad ) in one pass (no autograd; the un-pooled
 * activation is never written); same classifiers, evaluatePerturbation.py:627-640, as run by the forward-only loops of
 * MASTestFunctions.py:273 and generate_emap.py:91-97.   x : [N][C][H*W];  y : [N][C][PH*PW] */
int xai_bn_relu_maxpool_fwd_f32(const float* x, const float* weight, const float* bias, const float* mean,
                                const float* var, float eps, int variant, int N, int C, int H, int W, int PH,
                                int PW, int kernel, int stride, int pad, float* y, xai_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* XAI_HIP_H */
