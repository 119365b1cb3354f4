"""Shared test helpers: the tiny classifier the golden vectors were made with."""
import numpy as np
import torch
import torch.nn as nn


class TinyNet(nn.Module):
    """Conv(3,8,3,p=1) -> ReLU -> AdaptiveAvgPool(4) -> Flatten -> Linear(128,10); weights come
    from the golden .npz so no RNG has to agree across machines."""

    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(3, 8, 3, padding=1)
        self.act = nn.ReLU()
        self.pool = nn.AdaptiveAvgPool2d(4)
        self.fc = nn.Linear(128, 10)

    def forward(self, x):
        return self.fc(torch.flatten(self.pool(self.act(self.conv(x))), 1))


def tiny_from(npz, device="cpu"):
    m = TinyNet().eval()
    sd = {k: torch.from_numpy(npz["w_" + k.replace(".", "_")]) for k in m.state_dict()}
    m.load_state_dict(sd)
    for p in m.parameters():
        p.requires_grad_(False)
    return m.to(device)


def logits_fn_of(model):
    dev = next(model.parameters()).device

    def fn(batch):
        with torch.no_grad():
            return model(torch.from_numpy(np.ascontiguousarray(batch, dtype=np.float32)).to(dev)).cpu().numpy()
    return fn


def vit_mini_from(npz, device="cpu"):
    """The reference's mini hooked ViT (tests/golden/vit_mini.npz) rebuilt with the build's own
    architecture; parameter names are identical, so the state dict loads as is."""
    from xai_engine.zoo import VisionTransformer
    m = VisionTransformer(img=32, patch=8, dim=32, depth=2, heads=4, num_classes=10).eval()
    sd = {k: torch.from_numpy(npz["w_" + k]) for k in m.state_dict()}
    m.load_state_dict(sd)
    for p in m.parameters():
        p.requires_grad_(False)
    return m.to(device)
