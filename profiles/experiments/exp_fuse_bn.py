"""Fused BN+ReLU(+add) classifier vs the original: bit-identity and fwd+bwd time at the benchmark's batch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-classification-xai_amd"))
import torch
from xai_engine.zoo import resnet50
from xai_engine.prepare import fuse_bn_relu, use_tuned_miopen_db
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = use_tuned_miopen_db(0)
m = resnet50(seed=0).to(dev)
with torch.no_grad():                                   # non-trivial BN statistics, like a trained network
    g = torch.Generator().manual_seed(1)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.copy_(torch.randn(mod.num_features, generator=g) * 0.1)
            mod.running_var.copy_(torch.rand(mod.num_features, generator=g) + 0.5)
            mod.weight.copy_(torch.rand(mod.num_features, generator=g) + 0.5)
            mod.bias.copy_(torch.randn(mod.num_features, generator=g) * 0.1)
x = torch.randn(100, 3, 224, 224, device=dev)
f = fuse_bn_relu(m, verify=x[:8])
print("every fused call site bit-identical to the PyTorch kernels on 8 images: yes")
for name, net in (("original", m), ("fused", f)):
    for _ in range(2):
        xi = x.clone().requires_grad_(True); out = net(xi); torch.autograd.grad(out[:, 3].sum(), xi)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        xi = x.clone().requires_grad_(True); out = net(xi); (gr,) = torch.autograd.grad(out[:, 3].sum(), xi)
    torch.cuda.synchronize(); print(f"{name}: fwd+bwd batch 100 = {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms, peak mem {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB", flush=True)
    if name == "original":
        ref = (out.detach().clone(), gr.clone())
    else:
        print("batch-100 fused vs original (MIOpen is not run-to-run deterministic): logits max rel diff",
              float((out.detach() - ref[0]).abs().max() / ref[0].abs().max()))
with torch.no_grad():
    for name, net in (("original", m), ("fused", f)):
        net(x[:50]); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            net(x[:50])
        torch.cuda.synchronize(); print(f"{name}: forward batch 50 = {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms")
