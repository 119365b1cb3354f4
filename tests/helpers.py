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


CONV_NOISE = 4e-6      # max |conv_oneDNN - conv_MIOpen| of the tiny net over all fixtures: 3.8e-6 (profiles/r02_gate_flips.json), rounded up


def ill_conditioned_footprint(npz, x, baseline, steps, noise=CONV_NOISE):
    """(H,W) bool mask of the input pixels whose IG value depends on a ReLU gate that two correct fp32 convolutions may
    set differently: the 3x3 footprints of every pre-activation of the tiny net (all `steps` path points, float64 on the
    host) that lies within `noise` of zero.  profiles/r02_gate_flips.json shows that the reference's CPU run and the
    GPU disagree on exactly such gates (one of 20 070 400 for ig_224/ig_tensor_baseline, |z| = 5.6e-8) and nowhere else."""
    import torch.nn.functional as F
    x = torch.as_tensor(x, dtype=torch.float32)
    base = baseline if torch.is_tensor(baseline) else torch.full_like(x, float(baseline))
    pts = base + torch.linspace(0, 1, steps).reshape(-1, 1, 1, 1) * (x - base)
    z = F.conv2d(pts.double(), torch.from_numpy(npz["w_conv_weight"]).double(), torch.from_numpy(npz["w_conv_bias"]).double(), padding=1)
    near = (z.abs() <= noise).any(1, keepdim=True).any(0, keepdim=True).float()
    return F.max_pool2d(near, 3, 1, 1)[0, 0].bool().numpy()
