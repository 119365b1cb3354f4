import sys, time
sys.path.insert(0, 'image-classification-xai_amd')
import torch
from xai_engine.zoo import resnet50
dev = torch.device('cuda:0')
m = resnet50(0).to(dev)
for bs in (25, 50, 64, 100, 125, 128, 200, 250, 256, 500):
    x = torch.randn(bs, 3, 224, 224, device=dev)
    with torch.no_grad():
        for _ in range(2): m(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = max(2, 400 // bs)
        for _ in range(n): m(x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"forward batch {bs:4d}: {dt*1e3:8.2f} ms  {bs/dt:9.0f} img/s")
