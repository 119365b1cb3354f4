"""Transformer Input Sampling on the HIP engine (reference util/attribution_methods/TIS.py; called by the harness as
`TIS(model, batch_size=64)(input_tensor.to(device), class_idx=target_class)`, evaluatePerturbation.py:236-239).

The per-pixel work of TIS is its score-weighted mask sum (`generate_saliency`, :331-365): K2's weighted form through
masked.tis_saliency.  Everything else is classifier-side PyTorch-ROCm on the device: block hooks for the encoder
activations, the k-means over activation channels (distance GEMMs on rocBLAS), batched top-k for the binary masks and
the token-sampling hook on `model.pos_drop` (one batched gather instead of the reference's per-mask Python loop).

`fast_pytorch_kmeans` (the reference's k-means) is not vendored by the reference and not in its requirements.txt; the
Lloyd iteration below restates its published algorithm with the initial centroids drawn from NumPy's global RNG like
that library does -- parity unpinned.  `raw_masks=` lets a caller supply centroids from anywhere.
"""
import math

import numpy as np
import torch

from .ig import hip_device
from .masked import tis_saliency


def kmeans_centroids(points, n_clusters, max_iter=100, tol=1e-4):
    """points (n,d) on the device -> centroids (n_clusters,d).  Euclidean Lloyd iteration; empty clusters become 0."""
    n = points.shape[0]
    first = np.random.choice(n, size=[n_clusters], replace=False)
    c = points[torch.from_numpy(first).to(points.device)].clone()
    x_sq = (points * points).sum(1, keepdim=True)
    for _ in range(max_iter):
        sim = 2 * points @ c.T - x_sq - (c * c).sum(1)[None]
        closest = sim.argmax(1)
        onehot = torch.zeros((n_clusters, n), dtype=points.dtype, device=points.device)
        onehot[closest, torch.arange(n, device=points.device)] = 1
        new = onehot @ points / onehot.sum(-1)[:, None]
        new[new != new] = 0
        err = (new - c).pow(2).sum()
        c = new
        if float(err) <= tol:
            break
    return c


class TIS:
    def __init__(self, model, n_masks=1024, batch_size=128, tokens_ratio=0.5, normalise=True, verbose=False,
                 ablation_study=False, *, raw_masks=None):
        self.model = model
        self.batch_size = batch_size
        self.n_masks = n_masks
        self.normalise = normalise
        self.verbose = verbose
        self.ablation_study = ablation_study
        self.tokens_ratio = [tokens_ratio] if isinstance(tokens_ratio, float) else tokens_ratio
        self.raw_masks = raw_masks
        self.cur_mask_indices = None

    @torch.no_grad()
    def __call__(self, x, class_idx=None):
        assert x.dim() == 3 or (x.dim() == 4 and x.shape[0] == 1), "Only one image can be processed at a time"
        if x.dim() == 3:
            x = x.unsqueeze(dim=0)
        hip_device(x.device)
        predicted_class, encoder_activations = self.get_encoder_activations(x)
        if class_idx is None:
            class_idx = predicted_class
        raw_masks = self.raw_masks if self.raw_masks is not None else self.generate_raw_masks(encoder_activations)
        masks, indices = self.generate_binary_masks(raw_masks.to(x.device))
        scores = self.generate_scores(x, class_idx, indices)
        return self.generate_saliency(x, scores, masks)

    @torch.no_grad()
    def get_encoder_activations(self, x):
        """TIS.py:96-132 -> (predicted class 0-d tensor, (1, 1+n_tokens, depth*dim))."""
        kept = []
        hooks = [layer.register_forward_hook(lambda m, i, o: kept.append(o.detach())) for layer in self.model.blocks]
        try:
            predicted_class = torch.argmax(self.model(x))
        finally:
            for h in hooks:
                h.remove()
        return predicted_class, torch.cat(kept, dim=-1)

    def generate_raw_masks(self, encoder_activations):
        """TIS.py:134-155: k-means over the activation channels (each a vector over the tokens)."""
        channels = encoder_activations.squeeze(0)[1:].T.contiguous().float()
        return kmeans_centroids(channels, self.n_masks)

    def generate_binary_masks(self, raw_masks):
        """TIS.py:157-190 -> (masks (N, n_tokens) float32, list of (n_masks, k) index tensors, one per ratio)."""
        masks, indices = [], []
        for ratio in self.tokens_ratio:
            k = int(ratio * raw_masks.shape[1])
            idx = raw_masks.topk(k, dim=1)[1]
            masks.append(torch.zeros_like(raw_masks).scatter_(1, idx, 1.0))
            indices.append(idx)
        return torch.cat(masks), indices

    def mask_input(self, x, indices, baseline="random"):
        """TIS.py:192-242 (ablation branch): token mask, nearest up-sampling, x*m + baseline*(1-m) per mask."""
        ph, pw = self.model.patch_embed.proj.kernel_size
        nh, nw = x.shape[2] // ph, x.shape[3] // pw
        m = torch.zeros((indices.shape[0], nh * nw), dtype=x.dtype, device=x.device).scatter_(1, indices, 1.0)
        m = m.view(-1, 1, nh, nw).repeat_interleave(ph, dim=2).repeat_interleave(pw, dim=3)
        if baseline == "random":
            base = torch.rand((indices.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        elif baseline == "zero":
            base = torch.zeros((1,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        else:
            print("Baseline not recognised")
            exit(1)
        return x * m + base * (1 - m)

    @torch.no_grad()
    def generate_scores(self, x, class_idx, indices):
        """TIS.py:244-329: softmax score of `class_idx` for every mask; masks of one ratio are batched together."""
        state = {"cur": None}

        def tokens_sampling_hook_fn(_, __, output):
            if state["cur"] is not None:
                idx = state["cur"]                                            # (B, k)
                cls = output[:, :1].expand(idx.shape[0], -1, -1)
                return torch.cat([cls, output[0, 1:][idx]], dim=1)
        hook = None if self.ablation_study else self.model.pos_drop.register_forward_hook(tokens_sampling_hook_fn)
        scores = []
        try:
            for idx_all in indices:
                for b in range(math.ceil(len(idx_all) / self.batch_size)):
                    idx = idx_all[b * self.batch_size:(b + 1) * self.batch_size]
                    if self.ablation_study:
                        result = self.model(self.mask_input(x, idx))
                    else:
                        state["cur"] = idx
                        result = self.model(x)
                    scores.append(torch.softmax(result, dim=1)[:, class_idx])
        finally:
            if hook is not None:
                hook.remove()
        return torch.cat(scores)

    def generate_saliency(self, x, scores, masks):
        """TIS.py:331-365 -> (h, w) on the device."""
        ph, pw = self.model.patch_embed.patch_size
        h, w = x.shape[-2] // ph, x.shape[-1] // pw
        return tis_saliency(scores.float().contiguous(), masks.float().contiguous(), normalise=self.normalise).reshape(h, w)
