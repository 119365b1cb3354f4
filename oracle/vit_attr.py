"""Oracle: attention-space IG of a hooked ViT (test infrastructure only).
Restates Baselines.IG, util/attribution_methods/VIT_LRP/ViT_explanation_generator.py:358-386:
sequential passes on input*alpha, full (heads,S,S) gradient sum, /steps, clamp, head mean, CLS row."""
import numpy as np
import torch


def attention_ig(model, x, target, steps=20):
    dev = next(model.parameters()).device
    total = None
    for alpha in np.linspace(0, 1, steps):
        scaled = (torch.from_numpy(np.asarray(x, dtype=np.float32)).to(dev) * alpha).requires_grad_(True)
        out = model(scaled, register_hook=True)
        out[0][int(target)].sum().backward()
        g = model.blocks[-1].attn.get_attn_gradients().detach().cpu().numpy()
        total = g.copy() if total is None else total + g
    w = np.maximum(total / np.float32(steps), 0).mean(axis=1)[:, 0, :]
    side = int(np.sqrt(w.shape[-1] - 1))
    return w[:, 1:].reshape(-1, side, side)
