"""Round 3: WHICH captured node misbehaves when two hipGraphs of a classifier pass replay concurrently on two streams
(profiles/r03_exp_ig_graph_streams*.jsonl)?  Every distinct convolution of a batch-50 ResNet-50 pass (plus the fc GEMM, the max-pool and
a BatchNorm/ReLU) is captured twice -- slot A and slot B, forward + backward-data on static buffers, each on its own capture
stream -- and the two graphs are replayed together on two streams with different inputs; outputs are compared with eager runs."""
import json, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, os.path.join(R, "image-classification-xai_amd")]
import torch
from xai_engine.zoo import resnet50

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = False, (sys.argv[1:] or ["deterministic"])[0] == "deterministic"
BATCH = int(os.environ.get("XAI_EXP_BATCH", "50"))
EAGER = os.environ.get("XAI_EXP_EAGER") == "1"          # launch the two slots EAGERLY on the two streams instead of replaying graphs
ONLY = os.environ.get("XAI_EXP_ONLY")                   # restrict to one module name
REPS = int(os.environ.get("XAI_EXP_REPS", "8"))
model = resnet50(seed=0).to(dev)
shapes = {}
hooks = []
for name, mod in model.named_modules():
    if isinstance(mod, (torch.nn.Conv2d, torch.nn.Linear, torch.nn.MaxPool2d)) or name in ("bn1", "layer1.0.bn3"):
        hooks.append(mod.register_forward_hook(lambda m, i, o, name=name: shapes.setdefault(name, tuple(i[0].shape)) and None))
with torch.no_grad():
    model(torch.randn(BATCH, 3, 224, 224, device=dev))
for h in hooks:
    h.remove()
mods = dict(model.named_modules())
seen, todo = set(), []
for name, shp in shapes.items():
    m = mods[name]
    key = (type(m).__name__, shp, tuple(getattr(m, "weight", torch.zeros(0)).shape), getattr(m, "stride", None))
    if key not in seen:
        seen.add(key); todo.append(name)


class Slot:
    def __init__(self, mod, shape):
        self.mod = mod
        self.x = torch.randn(shape, device=dev).requires_grad_(True)
        with torch.no_grad():
            self.gy = torch.randn_like(mod(self.x))
        self.stream = torch.cuda.Stream(dev)
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self.stream):
            for _ in range(2):
                self.run()
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=self.stream):
            self.y, self.gx = self.run()

    def run(self):
        y = self.mod(self.x)
        (gx,) = torch.autograd.grad(y, self.x, self.gy)
        return y.detach(), gx


gen = torch.Generator(device=dev).manual_seed(1)
for name in todo:
    if ONLY and name != ONLY:
        continue
    mod, shp = mods[name], shapes[name]
    A, B = Slot(mod, shp), Slot(mod, shp)
    bad = {"fwd": 0, "bwd": 0}
    worst = 0.0
    trials = int(os.environ.get("XAI_EXP_TRIALS", "6"))
    for t in range(trials):
        want = []
        for s in (A, B):
            with torch.no_grad():
                s.x.copy_(torch.randn(shp, device=dev, generator=gen)); s.gy.copy_(torch.randn(s.gy.shape, device=dev, generator=gen))
            y, gx = s.run()                                   # eager, serial
            want.append((y.clone(), gx.clone()))
        torch.cuda.synchronize()
        got = {}
        for _ in range(REPS):                                 # the two slots together on their two streams, several times
            for s in (A, B):
                with torch.cuda.stream(s.stream):
                    if EAGER:
                        got[id(s)] = s.run()
                    else:
                        s.graph.replay()
        torch.cuda.synchronize()
        for s, (wy, wg) in zip((A, B), want):
            y, gx = got[id(s)] if EAGER else (s.y, s.gx)
            if not torch.equal(y, wy):
                bad["fwd"] += 1; worst = max(worst, float((y - wy).abs().max() / wy.abs().max()))
            if not torch.equal(gx, wg):
                bad["bwd"] += 1; worst = max(worst, float((gx - wg).abs().max() / wg.abs().max()))
    print(json.dumps({"flow": "eager on two streams" if EAGER else "two graphs replayed on two streams", "module": name, "type": type(mod).__name__, "input": list(shp), "weight": list(getattr(mod, "weight", torch.zeros(0)).shape),
                      "stride": list(getattr(mod, "stride", [])) if not isinstance(getattr(mod, "stride", None), int) else getattr(mod, "stride"),
                      "mismatching_outputs_of": 2 * trials, **bad, "worst_rel_inf": worst}), flush=True)
    del A, B
