"""Stem max-pool backward at the benchmark's shape (100 x 64 x 112 x 112): PyTorch's kernel vs xai_maxpool_bwd_f32."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-classification-xai_amd"))
import torch, torch.nn.functional as F
from xai_engine import kernels as K
dev = "cuda:0"
x = torch.randn(100, 64, 112, 112, device=dev).relu_().requires_grad_(True)
y, idx = F.max_pool2d(x, 3, 2, 1, 1, False, True)
gy = torch.randn_like(y)
def t(fn, n=10):
    for _ in range(2): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
print("pytorch max_pool2d backward: %.1f us" % t(lambda: torch.autograd.grad(y, x, gy, retain_graph=True)))
print("xai_maxpool_bwd_f32:         %.1f us" % t(lambda: K.maxpool_bwd(gy, idx, 112, 112, 3, 2, 1)))
print("equal:", torch.equal(torch.autograd.grad(y, x, gy, retain_graph=True)[0], K.maxpool_bwd(gy, idx, 112, 112, 3, 2, 1)))
