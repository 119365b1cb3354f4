"""Oracle: Transformer Input Sampling (test infrastructure only).

Restates util/attribution_methods/TIS.py: encoder activations :96-132, binary masks :157-190, token-sampling
scores :244-329, saliency :331-365, input masking (ablation branch) :192-242.

PINNED by tests/golden/tis.npz (tests/golden/make_golden.py: tis_fixture: the reference's own TIS methods on its mini
hooked ViT).  UNPINNED: `generate_raw_masks` (:134-155) calls the third-party `fast_pytorch_kmeans.KMeans`, absent
from this image and from the reference's requirements.txt (no version); `kmeans_centroids` restates that library's
published Lloyd iteration (random data points as initial centroids, assignment by largest
2 x.c - |x|^2 - |c|^2, centroid = mean of its points, empty cluster -> 0, stop when the squared centroid shift is
<= tol or after max_iter rounds) and nothing here can check it against the library.
"""
import numpy as np
import torch

F32 = np.float32


def encoder_activations(model, x):
    """TIS.py:96-132: forward hooks on every block, outputs concatenated on the feature axis -> (pred, (1,1+n,depth*D))."""
    kept = []
    hooks = [blk.register_forward_hook(lambda m, i, o: kept.append(o.detach())) for blk in model.blocks]
    try:
        with torch.no_grad():
            pred = int(torch.argmax(model(torch.as_tensor(x))))
    finally:
        for h in hooks:
            h.remove()
    return pred, torch.cat(kept, dim=-1).numpy()


def kmeans_centroids(points, n_clusters, rng=np.random, max_iter=100, tol=1e-4):
    """points (n, d) -> centroids (n_clusters, d); see the module docstring (parity unpinned)."""
    x = np.asarray(points, dtype=F32)
    c = x[rng.choice(len(x), size=[n_clusters], replace=False)].copy()
    for _ in range(max_iter):
        sim = 2 * x @ c.T - (x * x).sum(1, keepdims=True) - (c * c).sum(1)[None]
        closest = sim.argmax(1)
        new = np.zeros_like(c)
        for k in range(n_clusters):
            sel = closest == k
            if sel.any():
                new[k] = x[sel].mean(0)
        err = ((new - c) ** 2).sum()
        c = new
        if err <= tol:
            break
    return c


def binary_masks(raw_masks, tokens_ratio):
    """TIS.py:157-190: per ratio, per raw mask: the int(ratio*n) largest entries -> 1 (indices in top-k order)."""
    raw = torch.as_tensor(np.asarray(raw_masks, dtype=F32))
    ratios = [tokens_ratio] if isinstance(tokens_ratio, float) else list(tokens_ratio)
    masks, indices = [], []
    for r in ratios:
        for row in raw:
            idx = row.topk(int(r * row.numel()))[1]
            m = torch.zeros_like(row)
            m[idx] = 1
            masks.append(m.numpy())
            indices.append(idx.numpy())
    return np.stack(masks), indices


def scores(model, x, class_idx, indices, batch_size):
    """TIS.py:244-329: a hook on model.pos_drop replaces the one token sequence by one sampled sequence per mask."""
    state = {"cur": None}

    def hook(_, __, output):
        if state["cur"] is not None:
            cls, tokens = output[:, 0].unsqueeze(1), output[:, 1:]
            return torch.cat([torch.cat([cls, tokens[:, torch.as_tensor(i)]], dim=1) for i in state["cur"]])
    h = model.pos_drop.register_forward_hook(hook)
    out = []
    try:
        with torch.no_grad():
            for b in range(0, len(indices), batch_size):
                state["cur"] = indices[b:b + batch_size]
                out.append(torch.softmax(model(torch.as_tensor(x)), dim=1)[:, class_idx])
    finally:
        h.remove()
    return torch.cat(out).numpy()


def saliency(score, masks, h, w, normalise):
    """TIS.py:331-365: sum_n s_n m_n / sum_n m_n, optionally min-max normalised."""
    m = np.asarray(masks, dtype=F32).T                     # (n_tokens, N)
    raw = (np.asarray(score, dtype=F32) * m).sum(-1, dtype=F32)
    sal = (raw / m.sum(-1, dtype=F32)).reshape(h, w)
    if normalise:
        sal = sal - sal.min()
        sal = sal / sal.max()
    return sal


def mask_input_zero(x, indices, patch):
    """TIS.py:192-242 with baseline='zero': nearest-neighbour up-sampled token mask times the image."""
    x = np.asarray(x, dtype=F32)
    nh, nw = x.shape[2] // patch, x.shape[3] // patch
    out = []
    for idx in indices:
        m = np.zeros(nh * nw, dtype=F32)
        m[np.asarray(idx)] = 1
        m = np.repeat(np.repeat(m.reshape(nh, nw), patch, axis=0), patch, axis=1)
        out.append(x * m + np.zeros_like(x) * (1 - m))
    return np.concatenate(out, axis=0)
