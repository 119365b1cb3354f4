"""Cold start of a sweep: seconds for image 1, 2, 3 in a fresh process (MIOpen search for shapes missing from the shipped find-db)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-classification-xai_amd"))
import numpy as np, torch
from xai_engine.zoo import resnet50
from xai_engine.prepare import use_tuned_miopen_db
from xai_engine.sweep import sweep_images
from xai_engine.ig import IG
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = use_tuned_miopen_db(0) if (len(sys.argv) < 2 or sys.argv[1] == "db") else False
m = resnet50(seed=0).to(dev)
imgs = [torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(i)) for i in range(4)]
attr = lambda x, t: IG(x, m, 50, 50, 1, 0, dev, t).sum(0).abs()
for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sweep_images(imgs[i:i + 1], m, dev, attr)
    torch.cuda.synchronize(); print(f"image {i}: {time.perf_counter() - t0:.3f} s", flush=True)
