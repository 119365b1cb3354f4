"""The whole one-image IG attribution (K1, classifier forward / backward, filing, K2) as one hipGraph replay: does it
reproduce the eager result, and is it faster?  (Every tensor the captured kernels read must outlive the graph -- a first
version of this script let the `alphas` tensor go out of scope after capture and "found" replays that were off by 2 %.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-classification-xai_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from xai_engine import kernels as K
from xai_engine.ig import IG, _path, _prep


class CapturedIG:
    """The whole one-image IG (K1, classifier forward/backward, filing, K2) as one hipGraph on static buffers."""

    def __init__(self, model, example_input, steps, batch_size):
        self.dev, self.x, self.base = _prep(example_input.detach().clone(), 0, example_input.device)
        self.target = torch.zeros((), dtype=torch.int64, device=self.dev)
        self.alphas = torch.linspace(0, 1, steps).to(self.dev)          # read by the captured K1: must live as long as the graph

        def run():
            grads, logits = _path(self.x, self.base, self.alphas, model, batch_size, self.target)
            return K.ig_accum(grads, self.x, self.base)[0]
        side = torch.cuda.Stream(self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            for _ in range(2):
                run()
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = run()

    def __call__(self, input, target_class):
        self.x.copy_(input)
        self.target.copy_(torch.tensor(int(target_class)))
        self.graph.replay()
        return self.out.clone()
from xai_engine.zoo import resnet50
from xai_engine.prepare import use_tuned_miopen_db
DEV = "cuda:0"
mode = sys.argv[1] if len(sys.argv) > 1 else "db"
torch.backends.cudnn.benchmark = use_tuned_miopen_db(0) if mode == "db" else False
model = resnet50(seed=0).to(DEV)
xs = [torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(i)).to(DEV) for i in range(4)]
def rel(a, b): return float((a - b).abs().max() / b.abs().max())
cap = CapturedIG(model, xs[0], 50, 50)
for rep in range(2):
    for xi, t in zip(xs, (5, 700, 33, 5)):
        want = IG(xi, model, 50, 50, 1, 0, DEV, torch.tensor(t, device=DEV))
        print(mode, rep, t, "graph vs eager", rel(cap(xi, t), want), flush=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(8):
    cap(xs[i % 4], 5)
torch.cuda.synchronize(); print("graph ms", (time.perf_counter() - t0) / 8 * 1e3)
t5 = torch.tensor(5, device=DEV)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(8):
    IG(xs[i % 4], model, 50, 50, 1, 0, DEV, t5)
torch.cuda.synchronize(); print("eager ms", (time.perf_counter() - t0) / 8 * 1e3)
