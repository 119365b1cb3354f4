"""Which ResNet-50 forward convolution runs MIOpen's naive kernel at the sweep's batch sizes, and what the alternatives cost."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-classification-xai_amd"))
import torch
from xai_engine.zoo import resnet50
from xai_engine.prepare import use_tuned_miopen_db
mode = sys.argv[1] if len(sys.argv) > 1 else "db"
dev = torch.device("cuda:0")
if mode == "db":
    torch.backends.cudnn.benchmark = use_tuned_miopen_db(0)
elif mode == "find":
    torch.backends.cudnn.benchmark = True
else:
    torch.backends.cudnn.benchmark = False
m = resnet50(seed=0).to(dev)
convs = [(n, mod) for n, mod in m.named_modules() if isinstance(mod, torch.nn.Conv2d)]
for bs in (50, 25):
    x = torch.randn(bs, 3, 224, 224, device=dev)
    shapes = {}
    hooks = [mod.register_forward_hook(lambda mod, i, o, n=n: shapes.__setitem__(n, tuple(i[0].shape))) for n, mod in convs]
    with torch.no_grad():
        for _ in range(3):
            m(x)
    for h in hooks:
        h.remove()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        for _ in range(10):
            m(x)
    torch.cuda.synchronize()
    print(f"[{mode}] batch {bs}: forward {1e3 * (time.perf_counter() - t0) / 10:.3f} ms")
    rows = []
    for n, mod in convs:
        xi = torch.randn(shapes[n], device=dev)
        with torch.no_grad():
            for _ in range(3):
                mod(xi)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                mod(xi)
            b.record(); torch.cuda.synchronize()
        rows.append((a.elapsed_time(b) / 10, n, shapes[n], tuple(mod.weight.shape), mod.stride))
    rows.sort(reverse=True)
    for r in rows[:6]:
        print(f"   {r[0] * 1e3:8.1f} us  {r[1]:28s} in {r[2]} w {r[3]} stride {r[4]}")
