"""Round 3: HBM traffic of the library's IG kernels per attribution, buffered flow vs streaming flow.
usage (under rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace): python3 exp_ig_flows.py streaming|buffered
Runs exactly 8 one-image attributions (IG 50 steps, batch 50, ResNet-50 224^2, alpha_star = 1); profiles/pmc_ig_flows.py sums the
counters of the library's kernels over the run and divides by 8."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, os.path.join(R, "image-classification-xai_amd")]
import torch
import xai_engine
from xai_engine.ig import IG, ig_batch
from xai_engine.zoo import resnet50

flow = sys.argv[1]
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = False, True
xai_engine.load_library()
model = resnet50(seed=0).to(dev)
xs = torch.randn(8, 3, 224, 224, generator=torch.Generator().manual_seed(2)).to(dev)
with torch.no_grad():
    ts = model(xs).argmax(1)
for i in range(8):
    if flow == "streaming":
        out = IG(xs[i:i + 1], model, 50, 50, 1, 0, dev, ts[i])                       # the product's one-image API since round 3
    else:
        out = ig_batch(xs[i:i + 1], model, ts[i:i + 1], steps=50, images_per_pass=1, buffered=True)[0]   # round 2's flow: file, then reduce
torch.cuda.synchronize()
print(flow, float(out.abs().sum()))
