"""Round 3: WHAT differs when three fresh stream workers meet their first pass at the same moment on a cold box?  Each of three workers runs
the same ResNet-50 forward + input gradient of the same 50 images three times, all three workers starting together (no `first_alone`);
module outputs and module-output gradients are hashed per call; afterwards (everything warm) the main thread does the same once.  Printed:
for every worker and call, the first module whose output (forward order) / gradient (backward order) differs from the warm reference."""
import hashlib, json, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, os.path.join(R, "image-classification-xai_amd")]
import torch
from xai_engine.streams import workers
from xai_engine.zoo import resnet50

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = False, True
BATCH = int(os.environ.get("XAI_EXP_BATCH", "50"))
model = resnet50(seed=0).to(dev)
x0 = torch.randn(BATCH, 3, 224, 224, generator=torch.Generator().manual_seed(1)).to(dev)
if os.environ.get("XAI_EXP_FUSED") == "1":           # the classifier as bench.py runs it (must be prepared before anything is cold-started...
    import xai_engine
    from xai_engine.prepare import fuse_bn_relu
    xai_engine.load_library()
    model = fuse_bn_relu(model, verify=torch.randn(2, 3, 224, 224, device=dev), fork_residual=True)      # ... which itself warms the MAIN thread only)
names = [n for n, m in model.named_modules() if isinstance(m, (torch.nn.Conv2d, torch.nn.Linear))]      # modules that are still CALLED in the fused classifier
mods = dict(model.named_modules())


def sha(t):
    return hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()[:12]


def one_call():
    """-> ({module: hash of its output}, {module: hash of the gradient w.r.t. its output}) for ONE pass in the calling thread"""
    import threading
    me = threading.get_ident()
    outs, grads, hooks = {}, {}, []
    for n in names:
        def fwd(m, i, o, n=n):
            if threading.get_ident() != me:
                return
            outs[n] = o
            if o.requires_grad:
                o.register_hook(lambda g, n=n: grads.__setitem__(n, g))
        hooks.append(mods[n].register_forward_hook(fwd))
    x = x0.clone().requires_grad_(True)
    out = model(x)
    (gx,) = torch.autograd.grad(out[:, 3].sum(), x)
    for h in hooks:
        h.remove()
    torch.cuda.current_stream(dev).synchronize()
    return {n: sha(o) for n, o in outs.items()}, dict({n: sha(g) for n, g in grads.items()}, input=sha(gx))


ws = workers(dev, 3)
futs = [w.submit(lambda: [one_call() for _ in range(3)]) for w in ws]            # all three start together, cold
cold = [f.result() for f in futs]
ref_out, ref_grad = one_call()                                                      # warm, main thread
bwd_order = ["input"] + names                                                       # input gradient last in time; report the deepest-first order below
for k, calls in enumerate(cold):
    for c, (o, g) in enumerate(calls):
        first_fwd = next((n for n in names if o.get(n) != ref_out.get(n)), None)
        diff_bwd = [n for n in reversed(names) if g.get(n) != ref_grad.get(n)]
        print(json.dumps({"worker": k, "call": c, "first_module_whose_output_differs": first_fwd,
                          "first_module_in_backward_order_whose_output_gradient_differs": diff_bwd[0] if diff_bwd else None,
                          "modules_with_differing_gradients": len(diff_bwd), "input_gradient_differs": g["input"] != ref_grad["input"]}), flush=True)
