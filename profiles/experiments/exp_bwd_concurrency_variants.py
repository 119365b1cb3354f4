"""Round 3: the backward-data of ResNet-50's layer4.0.conv3 at batch 50 goes wrong when two of them run at once on two streams
(profiles/r03_exp_eager_concurrency_*.jsonl).  WHO has to be separate for it to go right?  Variants, 100 trials x 2 results each:
  engine_one_caller        autograd.grad from ONE host thread on two streams (backward nodes run on autograd's device thread)
  engine_two_callers       autograd.grad from TWO host threads, one per stream (backward nodes still on autograd's ONE device thread)
  engine_two_callers_turns as above, each grad inside backward_turn (host lock + device event chain)
  inline_one_caller        torch.autograd.set_multithreading_enabled(False): backward nodes run on the CALLING thread; one caller, two streams
  inline_two_callers       the same with one caller thread per stream (each thread its own MIOpen / rocBLAS handles)
  direct_one_caller        aten.convolution_backward called directly from one thread on two streams (no autograd at all)
  direct_one_caller_turns  the same, each call inside backward_turn: the second stream waits (on the device) for the first one's kernels
Every variant runs REPS launches per stream and trial and EVERY result is compared (not only the last one)."""
import json, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, os.path.join(R, "image-classification-xai_amd")]
import torch
from xai_engine.streams import workers, backward_turn
from xai_engine.zoo import resnet50

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = False, True
conv = resnet50(seed=0).layer4[0].conv3.to(dev)
shape = (50, 512, 7, 7)
gen = torch.Generator(device=dev).manual_seed(3)
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
ws = workers(dev, 2)
TRIALS = int(os.environ.get("XAI_EXP_TRIALS", "60"))


def via_autograd(x, gy):
    xr = x.detach().requires_grad_(True)
    y = conv(xr)
    (gx,) = torch.autograd.grad(y, xr, gy)
    return gx


def via_autograd_inline(x, gy):
    with torch.autograd.set_multithreading_enabled(False):
        return via_autograd(x, gy)


def direct(x, gy):
    return torch.ops.aten.convolution_backward(gy, x, conv.weight, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])[0]


def with_turn(fn):
    def run(x, gy):
        with backward_turn(dev):
            return fn(x, gy)
    return run


REPS = int(os.environ.get("XAI_EXP_REPS", "12"))


def one_caller(fn, xs, gys):
    got = [[], []]
    for _ in range(REPS):
        for k in range(2):
            with torch.cuda.stream(streams[k]):
                got[k].append(fn(xs[k], gys[k]))
    return got


def two_callers(fn, xs, gys):
    def job(k):
        return [fn(xs[k], gys[k]) for _ in range(REPS)]
    futs = [ws[k].submit(lambda k=k: job(k)) for k in range(2)]
    return [f.result() for f in futs]


variants = [("engine_one_caller", one_caller, via_autograd), ("engine_two_callers", two_callers, via_autograd),
            ("engine_two_callers_turns", two_callers, with_turn(via_autograd)), ("inline_one_caller", one_caller, via_autograd_inline),
            ("inline_two_callers", two_callers, via_autograd_inline), ("direct_one_caller", one_caller, direct),
            ("direct_one_caller_turns", one_caller, with_turn(direct))]
x0, g0 = torch.randn(shape, device=dev), torch.randn(50, 2048, 7, 7, device=dev)
for _, runner, fn in variants:                                    # warm every thread's handles and every code path
    runner(fn, [x0, x0], [g0, g0])
torch.cuda.synchronize()
for name, runner, fn in variants:
    bad = 0
    for t in range(TRIALS):
        xs = [torch.randn(shape, device=dev, generator=gen) for _ in range(2)]
        gys = [torch.randn(50, 2048, 7, 7, device=dev, generator=gen) for _ in range(2)]
        want = [via_autograd(xs[k], gys[k]) for k in range(2)]     # serial, default stream
        torch.cuda.synchronize()
        got = runner(fn, xs, gys)
        torch.cuda.synchronize()
        bad += sum(0 if torch.equal(a, want[k]) else 1 for k in range(2) for a in got[k])
    print(json.dumps({"variant": name, "wrong_results": bad, "of": 2 * TRIALS * REPS}), flush=True)
