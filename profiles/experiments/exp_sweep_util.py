import sys, time
sys.path.insert(0, 'image-classification-xai_amd')
import torch, numpy as np
from xai_engine.zoo import resnet50
from xai_engine.sweep import PerturbationSweep
dev = torch.device('cuda:0')
m = resnet50(0).to(dev)
sw = PerturbationSweep(m, 224, dev, batch_size=50)
x = torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(1000))
sal = np.abs(np.random.default_rng(0).standard_normal((224, 224))).astype(np.float32)
sw.run(x, sal); torch.cuda.synchronize()
# wall per image, and GPU-busy time measured with events around the device part
import xai_engine.sweep as S
t0 = time.perf_counter()
for _ in range(5):
    sw.run(x, sal)
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 5
# host-only part: time curves math on cached arrays
from xai_engine import curves
r = np.random.default_rng(1).random(225)
t0 = time.perf_counter()
for _ in range(20):
    a = curves.monotone_normalise(r, 0.1, 0.9, False); b = curves.monotone_normalise(r, 0.1, 0.9, True)
    d = curves.density_curve(np.random.rand(224).astype(np.float32), np.float32(100.0), True)
    curves.mas_correct(a, d, 'ins'); curves.mas_correct(b, d, 'del')
    curves.monotone_normalise(r, 0, 1, False); curves.monotone_normalise(r, 0, 1, True)
    from scipy.stats import spearmanr
    spearmanr(np.linspace(0, 1, 225), r); spearmanr(np.linspace(1, 0, 225), r)
host = (time.perf_counter() - t0) / 20
print(f"wall per image {wall*1e3:.1f} ms; host curve math {host*1e3:.2f} ms")
