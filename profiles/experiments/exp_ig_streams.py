"""Round 3: do consecutive classifier passes of IG overlap usefully on several HIP streams?
ig_batch(32 images, 50 steps) with images_per_pass in {1, 2} x streams in {1, 2, 3, 4}, in the parity configuration
(immediate mode, deterministic solvers) and with the shipped find-db; outputs compared bit for bit with streams=1."""
import json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, os.path.join(R, "image-classification-xai_amd")]
import torch
import xai_engine
from xai_engine.ig import ig_batch
from xai_engine.zoo import resnet50
from xai_engine.prepare import fuse_bn_relu, use_tuned_miopen_db

mode = sys.argv[1] if len(sys.argv) > 1 else "deterministic"
dev = torch.device("cuda:0")
if mode == "finddb":
    assert use_tuned_miopen_db(0)
    torch.backends.cudnn.benchmark = True
else:
    torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = False, True
xai_engine.load_library()
plain = resnet50(seed=0).to(dev)
x = torch.randn(32, 3, 224, 224, generator=torch.Generator().manual_seed(2)).to(dev)
model = fuse_bn_relu(plain, verify=x[:2], fork_residual=True)
with torch.no_grad():
    targets = plain(x).argmax(1)
grads = torch.empty((32, 50, 3, 224, 224), device=dev)
rows = []
ref = {}
for ipp in (1, 2):
    for ns in (1, 2, 3, 4):
        f = lambda: ig_batch(x, model, targets, steps=50, images_per_pass=ipp, want_abs=True, grads_buffer=grads, streams=ns)
        out = f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            out = f()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        if ns == 1:
            ref[ipp] = out[0].clone()
        same = bool(torch.equal(out[0], ref[ipp]))
        rel = float((out[0] - ref[ipp]).abs().max() / ref[ipp].abs().max())
        rows.append({"mode": mode, "images_per_pass": ipp, "streams": ns, "ms_per_step": dt * 1e3, "attr_per_s": 32 / dt, "bit_identical_to_1_stream": same,
                     "rel_inf_vs_1_stream": rel})
        print(json.dumps(rows[-1]), flush=True)
