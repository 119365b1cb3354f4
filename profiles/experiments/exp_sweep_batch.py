"""Sweep time per image against the metric batch size (the reference's `max_batch_size`, 50 in its harness): fused 3-sequence
sweep, ResNet-50 with the fused BN/ReLU classifier, deterministic solvers and immediate mode.  Same image and map at every size;
the max difference of the ten numbers from the batch-50 run shows what the other batch composition costs in agreement."""
import sys, time
sys.path.insert(0, 'image-classification-xai_amd')
import torch, numpy as np
from xai_engine.zoo import resnet50
from xai_engine.prepare import fuse_bn_relu
from xai_engine.sweep import PerturbationSweep, KEYS
dev = torch.device('cuda:0')
m0 = resnet50(0).to(dev)
m = fuse_bn_relu(m0, verify=torch.randn(2, 3, 224, 224, device=dev), fork_residual=True)
x = torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(1000))
sal = np.abs(np.random.default_rng(0).standard_normal((224, 224))).astype(np.float32)
for det in (True, False):
    torch.backends.cudnn.deterministic = det
    ref = None
    for bs in (25, 50, 75, 112, 225):
        sw = PerturbationSweep(m, 224, dev, batch_size=bs)
        sw.run(x, sal); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            c = sw.run(x, sal)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        if bs == 50: ref = c
    for bs in (25, 50, 75, 112, 225):
        sw = PerturbationSweep(m, 224, dev, batch_size=bs)
        sw.run(x, sal); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            c = sw.run(x, sal)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        print(f"deterministic={det!s:5s} batch {bs:4d}: {dt*1e3:7.1f} ms per image-sweep; max |diff| of the ten numbers vs batch 50 = {max(abs(c[k]-ref[k]) for k in KEYS):.2e}", flush=True)
