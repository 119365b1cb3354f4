import sys, time
sys.path.insert(0, 'image-classification-xai_amd')
import torch, numpy as np
from xai_engine.zoo import resnet50
from xai_engine.sweep import PerturbationSweep
from xai_engine.prepare import use_tuned_miopen_db
torch.backends.cudnn.benchmark = use_tuned_miopen_db(0)
dev = torch.device('cuda:0')
m = resnet50(0).to(dev)
sw = PerturbationSweep(m, 224, dev, batch_size=50)
x = torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(1000))
sal = np.abs(np.random.default_rng(0).standard_normal((224, 224))).astype(np.float32)
sw.run(x, sal); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(4):
    sw.run(x, sal)
torch.cuda.synchronize()
print("ms per image", (time.perf_counter() - t0) / 4 * 1e3)
