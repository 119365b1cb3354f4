"""Attention-space attributions for hooked ViTs (reference
util/attribution_methods/VIT_LRP/ViT_explanation_generator.py: Baselines :135-386).

The model contract is the reference's: `model(x, register_hook=True)` records the last block's
attention map and, on backward, its gradient (`blocks[-1].attn.get_attention_map()` /
`.get_attn_gradients()`, ViT_ig.py:73-111).

`Baselines.IG` is a K2-shaped accumulation: the reference runs `steps` sequential
forward/backward passes on `input * alpha` and sums the (1,heads,S,S) attention gradients, of which
only the CLS row survives (`.mean(1)[:, 0, :]`, :381).  Here the `steps` scaled inputs go through
the classifier as ONE batch, the CLS rows (steps, heads, S) are reduced over the step axis by
xai_ig_accum_f32, and the clamp / head-mean finish on 2 364 numbers.
"""
import numpy as np
import torch

from . import kernels as K
from .ig import hip_device


class Baselines:
    def __init__(self, model):
        self.model = model
        self.model.eval()

    def _last_attn(self):
        return self.model.blocks[-1].attn

    def generate_raw_attn(self, input, device, layer=-1):
        """Head-mean CLS attention of one block (reference :139-144)."""
        with torch.no_grad():
            self.model(input.to(device))
        attn = self.model.blocks[layer].attn.get_attention_map().mean(1)[0, 0, 1:]
        side = int(np.sqrt(attn.shape[-1]))
        return attn.reshape(-1, side, side)

    def generate_grad(self, input, target_class, device, layer=-1):
        """Clamped head-mean CLS attention gradient (reference :146-157)."""
        x = input.to(device).detach().requires_grad_(True)
        output = self.model(x, register_hook=True)
        output[0][target_class].sum().backward()
        grad = self.model.blocks[layer].attn.get_attn_gradients().mean(1)[:, 0, 1:].clamp(0)
        side = int(np.sqrt(grad.shape[-1]))
        return grad.reshape(-1, side, side)

    def IG(self, input, target_class, steps=20, device="cuda:0"):
        """Attention-space Integrated Gradients (reference :358-386) -> (1, side, side)."""
        dev = hip_device(device)
        x = input.to(dev, torch.float32)
        alphas = torch.from_numpy(np.linspace(0, 1, steps)).to(dev, torch.float32)    # float64 linspace rounded once
        scaled = (x * alphas.reshape(steps, 1, 1, 1)).detach().requires_grad_(True)   # (steps,C,H,W): input * alpha
        output = self.model(scaled, register_hook=True)
        output[:, target_class].sum().backward()
        g = self._last_attn().get_attn_gradients()                                    # (steps, heads, S, S)
        heads, S = g.shape[1], g.shape[-1]
        cls_rows = g[:, :, 0, :].contiguous().reshape(1, steps, heads, S)              # the only rows that are used
        ones = torch.ones((1, heads, S), dtype=torch.float32, device=dev)
        mean = K.ig_accum(cls_rows, ones, 0.0)[0]                                      # sum_s / steps, (heads, S)
        w = mean.clamp(min=0).mean(0)                                                  # (S,)
        side = int(np.sqrt(S - 1))
        return w[1:].reshape(-1, side, side)
