"""Round 3: does MIOpen's immediate mode serve the FIRST call of a shape differently from later calls?  In one fresh process (deterministic
solvers, benchmark off): ResNet-50 forward + input gradient of a 50-image batch, and of a 1-image batch, five times in a row on the main
thread; then the same five calls on a NEW host thread (a fresh MIOpen handle in a process whose on-disk caches are warm by then).
Every call is compared bit for bit with the LAST call of the main thread; hashes of that call are printed so that two PROCESSES (the first
on a box whose MIOpen user db / kernel cache are empty, the second right after it) can be compared too."""
import json, os, sys, threading
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, os.path.join(R, "image-classification-xai_amd")]
import torch
from xai_engine.zoo import resnet50

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = False, True
model = resnet50(seed=0).to(dev)
xs = {50: torch.randn(50, 3, 224, 224, generator=torch.Generator().manual_seed(1)).to(dev), 1: torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(2)).to(dev)}


def call(b):
    x = xs[b].detach().requires_grad_(True)
    out = model(x)
    (g,) = torch.autograd.grad(out[:, 3].sum(), x)
    return out.detach().clone(), g.clone()


def series(tag, results):
    for b in (50, 1):
        results[(tag, b)] = [call(b) for _ in range(5)]
    torch.cuda.synchronize()


res = {}
series("main", res)
t = threading.Thread(target=series, args=("new_thread", res)); t.start(); t.join()
for b in (50, 1):
    ref = res[("main", b)][-1]
    for tag in ("main", "new_thread"):
        row = [{"logits_equal": bool(torch.equal(o, ref[0])), "grad_equal": bool(torch.equal(g, ref[1])),
                "grad_rel_inf": float((g - ref[1]).abs().max() / ref[1].abs().max())} for o, g in res[(tag, b)]]
        print(json.dumps({"batch": b, "thread": tag, "calls_vs_last_main_call": row}), flush=True)
import hashlib
for b in (50, 1):
    o, g = res[("main", b)][-1]
    print(json.dumps({"batch": b, "sha256_logits": hashlib.sha256(o.cpu().numpy().tobytes()).hexdigest()[:16],
                      "sha256_grad": hashlib.sha256(g.cpu().numpy().tobytes()).hexdigest()[:16]}), flush=True)
print("user find-db after the run:", os.listdir(os.path.expanduser("~/.config/miopen")) if os.path.isdir(os.path.expanduser("~/.config/miopen")) else "absent")
