"""Where the cold start of one rank of a sweep goes (it is a constant per rank, so it caps the strong scaling of a 1000-image job:
125 images x 95 ms = 12 s of steady work per rank at 8 GPUs).  Fresh process; seconds for: importing torch, building the classifier,
the first classifier pass of each shape the sweep uses (1, 24, 50 images forward; 50 images forward + backward), the first and
second image through IG + the ten metrics.   usage: exp_cold_start.py [deterministic|finddb|immediate]"""
import os, sys, time
t_start = time.perf_counter()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-classification-xai_amd"))
import torch
t_import = time.perf_counter()
mode = sys.argv[1] if len(sys.argv) > 1 else "deterministic"
from xai_engine.zoo import resnet50
from xai_engine.prepare import use_tuned_miopen_db, fuse_bn_relu
from xai_engine.sweep import sweep_images
from xai_engine.ig import IG
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = use_tuned_miopen_db(0) if mode == "finddb" else False
torch.backends.cudnn.deterministic = mode == "deterministic"


def lap(label, t0):
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"{label:58s} {t1 - t0:7.3f} s", flush=True)
    return t1


print(f"mode {mode}")
print(f"{'import torch':58s} {t_import - t_start:7.3f} s")
t = time.perf_counter()
m = resnet50(seed=0).to(dev)
t = lap("ResNet-50 built and on the device (first HIP call)", t)
m = fuse_bn_relu(m, verify=torch.randn(2, 3, 224, 224, device=dev), fork_residual=True)
t = lap("fuse_bn_relu incl. call-site verification (2 images fwd+bwd)", t)
for n in (1, 24, 50):
    with torch.no_grad():
        m(torch.randn(n, 3, 224, 224, device=dev))
    t = lap(f"first forward of {n} images", t)
x = torch.randn(50, 3, 224, 224, device=dev, requires_grad=True)
torch.autograd.grad(m(x)[:, 0].sum(), x)
t = lap("first forward + backward of 50 images", t)
imgs = [torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(i)) for i in range(3)]
attr = lambda x, tt: IG(x, m, 50, 50, 1, 0, dev, tt).sum(0).abs()
for i in range(3):
    sweep_images(imgs[i:i + 1], m, dev, attr)
    t = lap(f"image {i}: IG + ten metrics", t)
print(f"{'total':58s} {time.perf_counter() - t_start:7.3f} s")
