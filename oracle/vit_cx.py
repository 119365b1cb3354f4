"""Oracle: ViT-CX mask construction and causal score (test infrastructure only).

Restates util/attribution_methods/ViT_CX/ViT_CX.py:22-34 (cosine similarity, norm_matrix), :41-46
(reshape_function_vit), :82-109 (resize, normalise, cluster, cluster sums, normalise) and
ViT_CX/causal_score.py:17-61 (noise, masked / noised stacks, score contraction).

PINNED by tests/golden/vit_cx.npz (tests/golden/make_golden.py: vitcx_fixture), vectors produced by calling the
reference's own `norm_matrix`, `get_cos_similar_matrix`, `reshape_function_vit` and `causal_score.forward`
(on the CPU, seeded host RNG).  UNPINNED at one boundary: `ViT_CX()` itself cannot run here because it
resizes with torchvision (`transforms.Resize(input_size, antialias=True)`, ViT_CX.py:66,82), absent from this
image; torchvision implements that call on a float tensor as
`torch.nn.functional.interpolate(x, size, mode='bilinear', align_corners=False, antialias=True)`, which is what
`resize_maps` calls.  The two-line cluster-sum loop (:105-106) lives inside that function and is restated as is.
"""
import numpy as np
import torch

F32 = np.float32


def norm_matrix(act):
    """ViT_CX.py:29-34."""
    act = np.asarray(act, dtype=F32)
    lo = act.min(axis=1, keepdims=True)
    hi = act.max(axis=1, keepdims=True)
    with np.errstate(divide="ignore", invalid="ignore"):
        return ((act - lo) / (hi - lo)).astype(F32)


def cos_similar_matrix(v1, v2):
    """ViT_CX.py:22-28 (float32 torch.mm on the CPU, NaN -> 0)."""
    a, b = torch.as_tensor(np.asarray(v1, dtype=F32)), torch.as_tensor(np.asarray(v2, dtype=F32))
    num = torch.mm(a, b.T)
    denom = torch.linalg.norm(a, dim=1).reshape(-1, 1) * torch.linalg.norm(b, dim=1)
    res = num / denom
    res[torch.isnan(res)] = 0
    return res.numpy()


def reshape_function_vit(tokens):
    """ViT_CX.py:41-46: (B,1+n,D) -> (B,D,side,side)."""
    tokens = np.asarray(tokens)
    side = int(np.sqrt(tokens.shape[1] - 1))
    r = tokens[:, 1:, :].reshape(tokens.shape[0], side, side, tokens.shape[2])
    return np.ascontiguousarray(r.transpose(0, 3, 1, 2))


def resize_maps(fmap, H, W):
    """(D,h,w) -> (D,H,W): what torchvision Resize((H,W), antialias=True) does to a float tensor."""
    t = torch.as_tensor(np.asarray(fmap, dtype=F32))[None]
    return torch.nn.functional.interpolate(t, size=(H, W), mode="bilinear", align_corners=False, antialias=True)[0].numpy()


def cluster_labels(mask, distance_threshold):
    """ViT_CX.py:89-96: complete-linkage clustering of 1 - cosine similarity (scikit-learn, as the reference)."""
    from sklearn.cluster import AgglomerativeClustering
    distance = 1 - cos_similar_matrix(mask, mask)
    c = AgglomerativeClustering(n_clusters=None, distance_threshold=distance_threshold, metric="precomputed", linkage="complete")
    c.fit(distance)
    return c.labels_


def cluster_sums(mask, labels):
    """ViT_CX.py:101-106: zeros, then `+=` row by row in ascending row order (float32)."""
    mask = np.asarray(mask, dtype=F32)
    out = np.zeros((len(set(labels)), mask.shape[1]), dtype=F32)
    for i in range(len(mask)):
        out[labels[i]] += mask[i]
    return out


def masks_from_feature_maps(fmap, H, W, distance_threshold=0.1):
    """(D,h,w) feature maps -> (K, H*W) normalised cluster masks, labels  (ViT_CX.py:82-109)."""
    mask = norm_matrix(resize_maps(fmap, H, W).reshape(fmap.shape[0], H * W))
    labels = cluster_labels(mask, distance_threshold)
    return norm_matrix(cluster_sums(mask, labels)), labels, mask


def causal_stack(x, masks, noise):
    """causal_score.py:24-47: (2N,C,H,W) = [x*m + (noise*0.1)*(1-m)] ++ [x + (noise*0.1)*(1-m)], float32."""
    x = np.asarray(x, dtype=F32)                          # (C,H,W)
    C, H, W = x.shape
    m = np.asarray(masks, dtype=F32).reshape(-1, 1, H, W)
    inv = (F32(1) - m).astype(F32)
    add = ((np.asarray(noise, dtype=F32) * F32(0.1)).astype(F32) * inv).astype(F32)
    masked = ((x[None] * m).astype(F32) + add).astype(F32)
    plain = (x[None] + add).astype(F32)
    return np.concatenate([masked, plain], axis=0)


def causal_saliency(p_whole, masks, class_p, H, W):
    """causal_score.py:54-61: p_whole (2N,CL) softmax rows -> (CL,H,W)."""
    N = len(masks)
    m = torch.as_tensor(np.asarray(masks, dtype=F32)).reshape(N, 1, H, W)
    p = torch.as_tensor(np.asarray(p_whole, dtype=F32))
    masks_divide = m / torch.sum(m, axis=0)
    p_final = p[:N].transpose(0, 1) - p[N:].transpose(0, 1) + F32(class_p)
    sal = torch.matmul(p_final, masks_divide.view(N, H * W)).view(p.shape[1], H, W)
    return (sal / N).numpy()


def causal_score(softmax_fn, x, masks, class_p, noise, gpu_batch=50):
    """softmax_fn: (B,C,H,W) float32 array -> (B,CL) probabilities."""
    C, H, W = np.asarray(x).shape[-3:]
    stack = causal_stack(np.asarray(x).reshape(C, H, W), masks, noise)
    p = np.concatenate([softmax_fn(stack[i:i + gpu_batch]) for i in range(0, len(stack), gpu_batch)])
    return causal_saliency(p, masks, class_p, H, W)
