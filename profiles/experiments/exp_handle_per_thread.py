"""Round 3: are PyTorch-ROCm's library handles (and so their workspaces) per HOST THREAD?  The backward-data convolution of
ResNet-50's layer4.0.conv3 at batch 50 (the one launch that corrupts under concurrency, profiles/r03_exp_eager_concurrency_*.jsonl) is
called directly -- torch.nn.grad.conv2d_input runs it in the CALLING thread -- on two streams at once:
  one_thread     both streams driven by one host thread  (one MIOpen / rocBLAS handle)
  two_threads    one host thread per stream              (xai_engine.streams.Worker: a handle set per thread)
  two_threads_turns   as above, each call inside backward_turn (the device-side event chain)"""
import json, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, os.path.join(R, "image-classification-xai_amd")]
import torch
from xai_engine.streams import workers, join, backward_turn

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = False, True
w = torch.randn(2048, 512, 1, 1, device=dev) * 0.05
shape = (50, 512, 7, 7)
gen = torch.Generator(device=dev).manual_seed(3)


def op(gy):
    return torch.nn.grad.conv2d_input(shape, w, gy)


op(torch.randn(50, 2048, 7, 7, device=dev)); torch.cuda.synchronize()
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
ws = workers(dev, 2)
for w_ in ws:                                                     # warm every thread's handles
    w_.submit(lambda: op(torch.randn(50, 2048, 7, 7, device=dev))).result()
torch.cuda.synchronize()
for mode in ("one_thread", "two_threads", "two_threads_turns"):
    bad, trials = 0, 100
    for t in range(trials):
        gys = [torch.randn(50, 2048, 7, 7, device=dev, generator=gen) for _ in range(2)]
        want = [op(g) for g in gys]
        torch.cuda.synchronize()
        if mode == "one_thread":
            got = [None, None]
            for _ in range(4):
                for k in range(2):
                    with torch.cuda.stream(streams[k]):
                        got[k] = op(gys[k])
        else:
            def job(k):
                out = None
                for _ in range(4):
                    if mode == "two_threads_turns":
                        with backward_turn(dev):
                            out = op(gys[k])
                    else:
                        out = op(gys[k])
                return out
            futs = [ws[k].submit(lambda k=k: job(k)) for k in range(2)]
            got = [f.result() for f in futs]
        torch.cuda.synchronize()
        bad += sum(0 if torch.equal(a, b) else 1 for a, b in zip(got, want))
    print(json.dumps({"mode": mode, "wrong_results": bad, "of": 2 * trials}), flush=True)
