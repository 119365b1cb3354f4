"""Round 3: with passes overlapped on several streams the host's enqueue rate becomes the limit (4 streams slower than 3 in
exp_ig_streams.py).  Does replaying each pass as ONE hipGraph (K1 + classifier forward/backward on static buffers, one graph
per stream slot) lift it?  Reports host enqueue time and wall time per step, eager vs graph, bit-compared with eager/1 stream."""
import json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, os.path.join(R, "image-classification-xai_amd")]
import torch
import xai_engine
from xai_engine import kernels as K
from xai_engine.ig import ig_batch
_SIDE = []


def _side_streams(dev, n):
    """n HIP streams driven from THIS host thread (what round 3's first multi-stream version did; see xai_engine/streams.py for why
    the product now uses one host thread per stream)."""
    while len(_SIDE) < n:
        _SIDE.append(torch.cuda.Stream(dev))
    return _SIDE[:n]
from xai_engine.zoo import resnet50
from xai_engine.prepare import fuse_bn_relu, use_tuned_miopen_db

mode = sys.argv[1] if len(sys.argv) > 1 else "deterministic"
OWN = len(sys.argv) > 2 and sys.argv[2] == "own"      # capture every slot's graph on its OWN stream (torch.cuda.graph's default capture stream is shared)
dev = torch.device("cuda:0")
if mode == "finddb":
    assert use_tuned_miopen_db(0)
    torch.backends.cudnn.benchmark = True
else:
    torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = False, True
xai_engine.load_library()
plain = resnet50(seed=0).to(dev)
B, S = 32, 50
x = torch.randn(B, 3, 224, 224, generator=torch.Generator().manual_seed(2)).to(dev)
model = plain if os.environ.get("XAI_EXP_PLAIN") else fuse_bn_relu(plain, verify=x[:2], fork_residual=True)
ONLY_IPP = [int(v) for v in os.environ.get("XAI_EXP_IPP", "1,2").split(",")]
with torch.no_grad():
    targets = plain(x).argmax(1)
grads = torch.empty((B, S, 3, 224, 224), device=dev)
alphas = torch.linspace(0, 1, S).to(dev)


class Slot:
    def __init__(self, k):
        self.x = torch.zeros((k, 3, 224, 224), device=dev)
        self.t = torch.zeros(k, dtype=torch.int64, device=dev)
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):
                self.run()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, **({"stream": side} if OWN else {})):
            self.g = self.run()

    def run(self):
        imgs = K.ig_interp(self.x, 0.0, alphas)
        flat = imgs.view(-1, 3, 224, 224).requires_grad_(True)
        out = model(flat)
        scores = out.gather(1, self.t.repeat_interleave(S).unsqueeze(1)).squeeze(1)
        (g,) = torch.autograd.grad(scores, flat, grad_outputs=torch.ones_like(scores))
        return g.contiguous()


def graph_step(slots, ipp):
    main = torch.cuda.current_stream(dev)
    side = _side_streams(dev, len(slots))
    ready = torch.cuda.Event(); ready.record(main)
    for st in side:
        st.wait_event(ready)
    for i, lo in enumerate(range(0, B, ipp)):
        sl = slots[i % len(slots)]
        with torch.cuda.stream(side[i % len(slots)]):
            sl.x.copy_(x[lo:lo + ipp], non_blocking=True)
            sl.t.copy_(targets[lo:lo + ipp], non_blocking=True)
            sl.graph.replay()
            K.store_grads(sl.g, grads[lo:lo + ipp])
    for st in side:
        main.wait_stream(st)
    return K.ig_accum(grads, x, 0.0, want_abs=True)


def timeit(f, n=3):
    out = f(); torch.cuda.synchronize()
    t0 = time.perf_counter(); host = 0.0
    for _ in range(n):
        h0 = time.perf_counter(); out = f(); host += time.perf_counter() - h0
    torch.cuda.synchronize()
    return out, (time.perf_counter() - t0) / n, host / n


ref = {}
for ipp in ONLY_IPP:
    out, dt, host = timeit(lambda: ig_batch(x, model, targets, steps=S, images_per_pass=ipp, want_abs=True, grads_buffer=grads))
    ref[ipp] = out[0].clone()
    print(json.dumps({"mode": mode, "flow": "eager", "images_per_pass": ipp, "streams": 1, "ms_per_step": dt * 1e3, "host_enqueue_ms": host * 1e3, "attr_per_s": B / dt}), flush=True)
    slots = []
    for ns in (1, 2, 3):
        while len(slots) < ns:
            slots.append(Slot(ipp))
        out, dt, host = timeit(lambda: graph_step(slots[:ns], ipp))
        print(json.dumps({"mode": mode, "flow": "graph", "capture_on_own_stream": OWN, "images_per_pass": ipp, "streams": ns, "ms_per_step": dt * 1e3, "host_enqueue_ms": host * 1e3, "attr_per_s": B / dt,
                          "bit_identical_to_eager_1_stream": bool(torch.equal(out[0], ref[ipp])),
                          "rel_inf": float((out[0] - ref[ipp]).abs().max() / ref[ipp].abs().max())}), flush=True)
        if ns in (2, 3, 4):
            out, dt, host = timeit(lambda: ig_batch(x, model, targets, steps=S, images_per_pass=ipp, want_abs=True, grads_buffer=grads, streams=ns))
            print(json.dumps({"mode": mode, "flow": "eager", "images_per_pass": ipp, "streams": ns, "ms_per_step": dt * 1e3, "host_enqueue_ms": host * 1e3, "attr_per_s": B / dt}), flush=True)
    del slots
