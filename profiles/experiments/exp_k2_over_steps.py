"""Round 3: the stamped K2 time of every timed step of bench.py's workload over a long run (does it drift under sustained load?)."""
import json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, os.path.join(R, "image-classification-xai_amd")]
import torch
import xai_engine
from xai_engine.ig import ig_batch
from xai_engine.zoo import resnet50
from xai_engine.prepare import fuse_bn_relu

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = False, True
xai_engine.load_library()
streams = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
plain = resnet50(seed=0).to(dev)
x = torch.randn(32, 3, 224, 224, generator=torch.Generator().manual_seed(2)).to(dev)
model = fuse_bn_relu(plain, verify=x[:2], fork_residual=True)
with torch.no_grad():
    t = plain(x).argmax(1)
grads = torch.empty((32, 50, 3, 224, 224), device=dev)
ev = []
for _ in range(2):
    ig_batch(x, model, t, steps=50, images_per_pass=1, want_abs=True, grads_buffer=grads, streams=streams)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    ig_batch(x, model, t, steps=50, images_per_pass=1, want_abs=True, grads_buffer=grads, event_sink=ev, streams=streams)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
us = [e[2].elapsed_time(e[3]) * 1e3 for e in ev]
print(json.dumps({"streams": streams, "steps": steps, "attr_per_s": 32 * steps / dt, "k2_us_per_step": [round(u, 1) for u in us],
                  "k2_us_mean": sum(us) / len(us), "frac_mean": 1008336896 / (sum(us) / len(us) * 1e-6) / 8e12}))
