"""ViT-CX (causal explanation from ViT feature maps) on the HIP kernels.

Reference: util/attribution_methods/ViT_CX/ViT_CX.py:61-117 and ViT_CX/causal_score.py:9-61, called by the
harness as `result, _ = ViT_CX(model, input_tensor, model.blocks[-1].norm1, gpu_batch=1, device=device)`
(XAI_Survey/evaluations/evaluatePerturbation.py:231-235).

What runs where
  classifier forward (softmax of 2N masked / noised images)        PyTorch-ROCm, batches of `gpu_batch`
  feature maps (D,h,w) -> D normalised (H,W) masks                  K11  xai_up_rownorm_f32 (one launch, written once)
  D x D cosine similarity                                           torch.mm on the device (rocBLAS, a plain library GEMM)
  complete-linkage agglomerative clustering of the D x D matrix     host, scikit-learn -- exactly the reference's call (:94-95)
  cluster sums in the reference's `+=` order, min-max normalise     K13 xai_cluster_sum_f32, K12 xai_rownorm_f32
  x*m + noise*(1-m)  and  x + noise*(1-m)  for every mask           K14 xai_causal_apply_f32 (written straight into the 2N stack)
  sal = p_final @ (masks / sum masks) / N for the target class      K2 weighted form (masked.causal_saliency)

Differences from the reference, all deliberate
  * one forward pass serves both `y_hat` and the feature-map hook; the reference's extra CAM backward
    (get_feature_map -> BaseCAM.forward, loss.backward) only exists to fire the hooks and its gradients are unused.
  * only the target class row of `sal` is formed (the reference computes all CL rows and keeps one, ViT_CX.py:114);
    `causal_score.forward(..., target_category=None)` still returns all rows (one rocBLAS GEMM).
  * the Gaussian noise is drawn exactly like the reference by default (torch.randn on the HOST generator, then
    uploaded: same values for the same torch.manual_seed); `device_noise=True` draws on the device instead
    (different stream, no 2N x 602 KB host pass).
  * torchvision's Resize(antialias=True) == F.interpolate(bilinear, antialias=True); when up-sampling its taps are
    the plain bilinear ones up to rounding (<= 1.2e-7), which is what K11 evaluates.
"""
import numpy as np
import torch
import torch.nn as nn

from . import kernels as K
from .ig import hip_device
from .masked import causal_saliency


def get_cos_similar_matrix(v1, v2):
    """ViT_CX.py:22-28 (NaN from zero rows -> 0)."""
    num = torch.mm(v1, torch.transpose(v2, 0, 1))
    denom = torch.linalg.norm(v1, dim=1).reshape(-1, 1) * torch.linalg.norm(v2, dim=1)
    res = num / denom
    res[torch.isnan(res)] = 0
    return res


def norm_matrix(act):
    """ViT_CX.py:29-34: per-row min-max normalisation (K12)."""
    return K.rownorm(act.contiguous())


def reshape_function_vit(tensor):
    """ViT_CX.py:41-46: (B, 1+n, D) tokens -> (B, D, sqrt n, sqrt n), class token dropped."""
    side = int(np.sqrt(tensor.shape[1] - 1))
    result = tensor[:, 1:, :].reshape(tensor.size(0), side, side, tensor.size(2))
    return result.transpose(2, 3).transpose(1, 2)


def cluster_members(labels):
    """labels (R,) of 0..K-1 -> (members int32 (R,) grouped by label, ascending inside a label; offs int32 (K+1,))."""
    labels = np.asarray(labels, dtype=np.int64)
    members = np.argsort(labels, kind="stable").astype(np.int32)
    offs = np.concatenate([[0], np.cumsum(np.bincount(labels))]).astype(np.int32)
    return members, offs


def cluster_masks(mask, distance_threshold):
    """mask (D,P) on the device -> (K,P) normalised cluster masks (ViT_CX.py:89-109)."""
    from sklearn.cluster import AgglomerativeClustering
    similarity = get_cos_similar_matrix(mask, mask)
    distance = 1 - similarity
    cluster = AgglomerativeClustering(n_clusters=None, distance_threshold=distance_threshold, metric="precomputed", linkage="complete")
    cluster.fit(distance.cpu())
    members, offs = cluster_members(cluster.labels_)
    dev = mask.device
    sums = K.cluster_sum(mask, torch.from_numpy(members).to(dev), torch.from_numpy(offs).to(dev))
    return K.rownorm(sums), cluster.labels_            # out of place: lets the kernel split a row over several workgroups


class causal_score(nn.Module):
    """causal_score.py:9-61.  forward(x, masks_input, class_p, target_category=None, noise=None):
    target_category given -> (H,W) row of that class on the device; None -> all (CL,H,W) rows like the reference."""

    def __init__(self, model, input_size, gpu_batch=100, device="cuda:0", device_noise=False):
        super().__init__()
        self.model = model
        self.input_size = tuple(int(v) for v in input_size)
        self.gpu_batch = gpu_batch
        self.device = device
        self.device_noise = device_noise

    def draw_noise(self, N, C):
        H, W = self.input_size
        dev = hip_device(self.device)
        if self.device_noise:
            return torch.randn((N, C, H, W), device=dev)
        return torch.randn([N, C, H, W]).pin_memory().to(dev, non_blocking=True)      # the reference's host draw (:27)

    @torch.no_grad()
    def forward(self, x, masks_input, class_p, target_category=None, noise=None):
        dev = hip_device(self.device)
        H, W = self.input_size
        x = x[0].to(dev, torch.float32).contiguous()
        masks = masks_input.to(dev, torch.float32).reshape(-1, H * W).contiguous()
        N = masks.shape[0]
        self.N, self.masks = N, masks.view(N, 1, H, W)
        if noise is None:
            noise = self.draw_noise(N, x.shape[0])
        stack = K.causal_apply(x, masks, noise.to(dev, torch.float32).contiguous(), 0.1)
        p_whole = torch.cat([self.model(stack[i:i + self.gpu_batch]).detach() for i in range(0, 2 * N, self.gpu_batch)])
        class_p = float(class_p) if not torch.is_tensor(class_p) else class_p.to(dev)
        if target_category is not None:
            t = int(target_category)
            p_final = p_whole[:N, t] - p_whole[N:, t] + class_p
            return causal_saliency(p_final.contiguous(), masks).view(H, W)
        p_final = p_whole[:N].transpose(0, 1) - p_whole[N:].transpose(0, 1) + class_p
        masks_divide = masks / torch.sum(masks, dim=0)
        return (torch.matmul(p_final, masks_divide) / N).view(-1, H, W)


def feature_maps(model_softmax, image, target_layer, reshape_function):
    """One forward pass: (softmax scores (CL,), reshaped feature maps (D,h,w)) -- ViT_CX.py:68-80 without the unused backward."""
    kept = []
    handle = target_layer.register_forward_hook(lambda m, i, o: kept.append(o.detach()))
    try:
        with torch.no_grad():
            y_hat = model_softmax(image)
    finally:
        handle.remove()
    fmap = kept[0]
    if reshape_function is not None:
        fmap = reshape_function(fmap)
    return y_hat[0], fmap[0].float().contiguous()


def ViT_CX(model, image, target_layer, target_category=None, distance_threshold=0.1, reshape_function=reshape_function_vit,
           gpu_batch=50, device="cuda:0", *, noise=None, device_noise=False, return_feature_map=True):
    """-> (sal (H,W), feature_map (D,H,W)) as CPU tensors, like the reference (ViT_CX.py:117).
    Keyword-only extras: `noise` (N,3,H,W) to fix the draw, `device_noise`, `return_feature_map=False` to skip
    materialising and downloading the D x H x W up-sampled maps (the harness discards them)."""
    dev = hip_device(device)
    image = image.to(dev, torch.float32)
    model_softmax = nn.Sequential(model, nn.Softmax(dim=1))
    y_hat, fmap = feature_maps(model_softmax, image, target_layer, reshape_function)
    if target_category is None:
        target_category = int(torch.argmax(y_hat))
    target_category = int(target_category)
    class_p = y_hat[target_category]
    H, W = int(image.shape[2]), int(image.shape[3])
    mask = K.up_rownorm(fmap, H, W)                                         # (D, H*W)
    mask_clustering_norm, _ = cluster_masks(mask, distance_threshold)
    scorer = causal_score(model_softmax, (H, W), gpu_batch=gpu_batch, device=device, device_noise=device_noise)
    sal = scorer(image, mask_clustering_norm, class_p, target_category=target_category, noise=noise)
    feature_map = K.bilinear_up(fmap, H, W).cpu() if return_feature_map else None
    return sal.cpu(), feature_map
