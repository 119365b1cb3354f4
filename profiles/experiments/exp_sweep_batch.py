import sys, time
sys.path.insert(0, 'image-classification-xai_amd')
import torch, numpy as np
from xai_engine.zoo import resnet50
from xai_engine.sweep import PerturbationSweep, KEYS
dev = torch.device('cuda:0')
m = resnet50(0).to(dev)
x = torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(1000))
sal = np.abs(np.random.default_rng(0).standard_normal((224, 224))).astype(np.float32)
ref = None
for bs in (50, 75, 112, 225):
    sw = PerturbationSweep(m, 224, dev, batch_size=bs)
    sw.run(x, sal); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        c = sw.run(x, sal)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    if ref is None: ref = c
    print(f"batch {bs:4d}: {dt*1e3:7.1f} ms per image-sweep; max |diff| vs batch 50 = {max(abs(c[k]-ref[k]) for k in KEYS):.2e}")
