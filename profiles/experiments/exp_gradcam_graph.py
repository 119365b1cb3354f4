"""Grad-CAM, one image: eager launches vs one hipGraph replay (CapturedGradCam)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-classification-xai_amd"))
import torch
from xai_engine.zoo import resnet50
from xai_engine.gradcam import gradcam_saliency, CapturedGradCam
from xai_engine.prepare import use_tuned_miopen_db
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = use_tuned_miopen_db(0)
m = resnet50(seed=0).to(dev)
xs = [torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(i)).to(dev) for i in range(8)]
with torch.no_grad():
    ts = [m(x).argmax(1)[0] for x in xs]
for _ in range(3):
    gradcam_saliency(m, m.layer4, xs[0], ts[0], (224, 224))
torch.cuda.synchronize(); t0 = time.perf_counter()
eager = [gradcam_saliency(m, m.layer4, x, t, (224, 224)) for x, t in zip(xs, ts)]
torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 8
cap = CapturedGradCam(m, m.layer4, xs[0], (224, 224))
cap(xs[0], ts[0])
torch.cuda.synchronize(); t0 = time.perf_counter()
graph = [cap(x, t) for x, t in zip(xs, ts)]
torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 8
same = all(torch.equal(a, b) for a, b in zip(eager, graph))
print(f"eager {te * 1e3:.3f} ms/image   hipGraph replay {tg * 1e3:.3f} ms/image   bit-identical: {same}")
for i, (a, b) in enumerate(zip(eager, graph)):
    print(i, int(ts[i]), float((a - b).abs().max() / a.abs().max()))
again = [cap(x, t) for x, t in zip(xs, ts)]
print("replay twice identical:", all(torch.equal(a, b) for a, b in zip(graph, again)))
