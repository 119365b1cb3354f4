"""Which ordering of the eval-mode BatchNorm expression reproduces PyTorch-ROCm's own kernels bit for bit?
Forward: F.batch_norm(training=False) (+ add) + relu.  Backward: autograd of the same."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-classification-xai_amd"))
import torch, torch.nn.functional as F
from xai_engine import kernels as K
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
for shape in ((4, 16, 28, 28), (3, 8, 7, 7), (2, 64, 56, 56)):
    N, C, H, W = shape
    x = torch.randn(shape, device=dev, generator=g) * 2 + 0.3
    idt = torch.randn(shape, device=dev, generator=g)
    w = torch.rand(C, device=dev, generator=g) + 0.5
    b = torch.randn(C, device=dev, generator=g)
    mean = torch.randn(C, device=dev, generator=g)
    var = torch.rand(C, device=dev, generator=g) + 0.2
    gy = torch.randn(shape, device=dev, generator=g)
    eps = 1e-5
    for add in (False, True):
        xr = x.clone().requires_grad_(True)
        ir = idt.clone().requires_grad_(True)
        bn = F.batch_norm(xr, mean, var, w, b, False, 0.0, eps)
        y = F.relu(bn + ir if add else bn)
        y.backward(gy)
        fwd_ok, bwd_ok = [], []
        for v in range(16):
            if ((v >> 1) & 3) == 3:
                continue
            yy = K.bn_act_fwd(x, idt if add else None, w, b, mean, var, eps, v, relu=True)
            if torch.equal(yy, y.detach()):
                fwd_ok.append(v)
            gx, gid = K.bn_relu_bwd(gy, y.detach(), w, var, eps, v, want_identity=add)
            if torch.equal(gx, xr.grad) and (not add or torch.equal(gid, ir.grad)):
                bwd_ok.append(v)
        plain = [v for v in range(16) if ((v >> 1) & 3) != 3 and torch.equal(K.bn_act_fwd(x, None, w, b, mean, var, eps, v, relu=False), bn.detach())]
        print(shape, "add" if add else "   ", "forward bit-exact variants:", fwd_ok, " backward:", bwd_ok, " plain bn:", plain, flush=True)
