"""Round 3 soak: is the multi-stream sweep bit-reproducible run after run?  ResNet-50 (fused classifier as in bench.py), 6 images, methods
grad / gc / ig, sweep_images(streams=3) repeated REPS times, every run compared EXACTLY with the one-stream totals."""
import json, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, os.path.join(R, "image-classification-xai_amd")]
import torch
import xai_engine
from xai_engine.zoo import resnet50
from xai_engine.prepare import fuse_bn_relu
from xai_engine.sweep import sweep_images, get_CNN_attr, KEYS

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = False, True
xai_engine.load_library()
REPS = int(os.environ.get("XAI_EXP_REPS", "12"))
plain = resnet50(seed=0).to(dev)
imgs = [torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(1000 + i)) for i in range(6)]
model = fuse_bn_relu(plain, verify=torch.cat(imgs[:2]).to(dev), fork_residual=True)
for method in ("grad", "gc", "ig"):
    td = {"models": [model], "img_hw": 224, "batch_size": 50, "device": str(dev), "device_maps": True, "attr_func": method}
    fn = lambda x, t: get_CNN_attr(x, None, t, td)
    ref, _, _ = sweep_images(imgs, model, dev, fn, img_hw=224, batch_size=50, streams=1)
    again, _, _ = sweep_images(imgs, model, dev, fn, img_hw=224, batch_size=50, streams=1)
    diffs = []
    for r in range(REPS):
        got, _, _ = sweep_images(imgs, model, dev, fn, img_hw=224, batch_size=50, streams=3)
        diffs.append(max(abs(got[k] - ref[k]) for k in KEYS))
    print(json.dumps({"method": method, "one_stream_twice_max_diff": max(abs(again[k] - ref[k]) for k in KEYS), "runs_on_3_streams": REPS,
                      "runs_differing_from_one_stream": sum(d != 0 for d in diffs), "max_diff": max(diffs)}), flush=True)
