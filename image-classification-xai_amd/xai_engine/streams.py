"""Running independent classifier passes side by side on several HIP streams -- and the rule that makes it safe.

Why: one classifier pass at the reference's batch sizes (50 interpolants, 50 step images) is a dependent chain of several
hundred launches, many of them too small to fill 256 CUs; a second and a third chain on their own streams fill the gaps
(IG +22 %, the ten-metric sweep +29 % on one MI355X) without changing a kernel, a shape or a summation order.

Why not simply `with torch.cuda.stream(s_k):` from one host thread: on PyTorch-ROCm ONE HOST THREAD DRIVING TWO STREAMS IS NOT
SAFE.  Measured (profiles/r03_exp_bwd_concurrency_variants.jsonl, r03_exp_eager_concurrency_*.jsonl): of the 27 distinct layers of a
batch-50 ResNet-50 pass exactly one -- the backward-data of `layer4.0.conv3` (1x1, 512 -> 2048 on 7x7), which MIOpen runs as a
rocBLAS split-K GEMM (`Cijk_...` + `Cijk_S_PostGSU3`) -- returns a WRONG gradient in 92-96 % of the launches when one thread issues it
alternately on two streams, through autograd's engine thread, with autograd run inline, or as a bare `aten.convolution_backward`;
hipGraph replays of such launches inherit it.  The same launches are right every time (0 of 1440) when a device-side event makes
the second stream wait for the first, or when each stream is driven by ITS OWN host thread.  PyTorch keeps its MIOpen / rocBLAS
handles -- and with them the libraries' scratch memory -- per host thread, not per stream: two streams fed through one handle can
run the split-K GEMM at the same time on the same scratch buffer.  Root cause confirmed by switching it off
(profiles/r03_exp_bwd_concurrency_root_cause.txt): with ROCBLAS_STREAM_ORDER_ALLOC=1 (rocBLAS takes its workspace stream-ordered per
call instead of from the handle's chunk) or MIOPEN_DEBUG_CONV_GEMM=0 (MIOpen stops using its rocBLAS GEMM solvers) the one-thread,
two-stream form is right 480 times of 480; by default it is wrong 440-459 times of 480.  In a whole pass the window is ~100 us of ~13 ms, so an
end-to-end comparison against the one-stream result can pass for a long time (the first version of this module did, bit for bit,
over thousands of passes); that is luck, not safety.  Hence the rule:

    EVERYTHING that touches a stream is launched by that stream's own host thread (`Worker`): forward passes, this library's
    kernels, copies -- and the BACKWARD nodes too: a worker runs autograd inline (`torch.autograd.set_multithreading_enabled(False)`,
    a thread-local switch), so its backward passes execute on the worker, through the worker's handles, instead of on autograd's
    one device thread that all callers share.

The rule is structural: it does not depend on which solver MIOpen picks for which shape.  The drivers (`ig.ig_batch`, `rise.rise`,
`sweep.sweep_images`) apply it whenever `streams > 1`.  Code that runs backward passes from threads that are NOT workers (a caller's
own threads, the main thread) goes through autograd's shared device thread; for those, `backward_turn` makes backward passes of a
device take turns (a host lock around the enqueue plus a device-side event chain) -- measured unnecessary (0 of 1440 wrong without
it) but not provably so, and it costs nothing when there is one caller.
"""
import concurrent.futures
import contextlib
import queue
import threading

import torch

_WORKERS = {}            # device index -> [Worker, ...]
_WORKERS_LOCK = threading.Lock()
_TURN = {}               # device index -> _Turn
_TURN_LOCK = threading.Lock()
_local = threading.local()
_COLD_LOCK = threading.Lock()        # one cold start (`first_alone`) at a time in the process
CAPTURE_LOCK = threading.Lock()     # ONE hipGraph capture at a time in the process, whoever captures: two threads inside capture_end crash the runtime


class Worker(threading.Thread):
    """A host thread bound to one HIP stream of one device: everything submitted to it is launched from this thread (its own
    MIOpen / rocBLAS handles) on this stream (torch's current stream is thread-local)."""

    def __init__(self, dev, index):
        super().__init__(daemon=True, name=f"xai-stream-{dev.index}-{index}")
        self.dev = dev
        self.stream = None
        self.warm = set()                               # kinds of work this thread has already run once, alone (see `first_alone`)
        self._q = queue.SimpleQueue()
        self._up = threading.Event()
        self._boot_error = None
        self.start()
        self._up.wait()
        if self._boot_error is not None:
            raise self._boot_error

    def run(self):
        try:
            torch.cuda.set_device(self.dev)
            self.stream = torch.cuda.Stream(self.dev)
        except BaseException as e:                      # no device: report to the creator instead of dying silently
            self._boot_error = e
            self._up.set()
            return
        self._up.set()
        _local.inline_autograd = True
        # backward nodes of this thread's graphs run ON this thread (own handles), not on autograd's shared device thread
        with torch.cuda.stream(self.stream), torch.autograd.set_multithreading_enabled(False):
            while True:
                item = self._q.get()
                if item is None:
                    return
                fn, fut = item
                if not fut.set_running_or_notify_cancel():
                    continue
                try:
                    fut.set_result(fn())
                except BaseException as e:              # handed to whoever waits for the future
                    fut.set_exception(e)

    def submit(self, fn):
        fut = concurrent.futures.Future()
        self._q.put((fn, fut))
        return fut


def on_worker():
    """True on a stream worker's thread."""
    return isinstance(threading.current_thread(), Worker)


def workers(dev, n):
    """The first `n` stream workers of a device (created on first use, kept for the life of the process)."""
    dev = torch.device(dev)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    dev = torch.device("cuda", idx)
    with _WORKERS_LOCK:
        have = _WORKERS.setdefault(idx, [])
        while len(have) < n:
            have.append(Worker(dev, len(have)))
        return have[:n]


def first_alone(worker, kind, fn):
    """Run `fn` on `worker`; if the worker has not run this `kind` of work before, run it ALONE: wait until it has been enqueued AND has
    finished on the device before anything else is handed to any worker by the caller.  Why: the first pass of a kind through a thread's
    fresh library handles is where MIOpen settles, PER HANDLE, which solver serves which shape; when three threads do that at the same
    moment on a box whose caches are cold, one of them can settle on another (valid, deterministic) solver for a layer and keeps it for
    the life of its handle -- measured: of three workers started together, one computed `layer4.1.conv2` (batch 1) differently from the
    other two and from the main thread in its first AND every later call (profiles/r03_exp_cold_start_which_layer_fused_batch1.jsonl);
    end to end a 3-stream sweep started that way differs from its own repetitions by 4e-8 ... 9e-7, a 1-stream sweep started cold does
    not differ at all (profiles/r03_exp_cold_start_streams.txt).  Started one after the other, fresh handles settle like the main
    thread's.  -> a future."""
    if kind is None or kind in worker.warm:
        return worker.submit(fn)
    with _COLD_LOCK:                                     # two CALLER threads warming two workers at once would be the same concurrent start
        fut = worker.submit(fn)
        concurrent.futures.wait([fut])
        worker.stream.synchronize()
        worker.warm.add(kind)
    return fut


def run_on_streams(dev, n, jobs, kind=None):
    """Run the callables `jobs` round-robin on `n` stream workers (job i on worker i % n) and return their results in order.
    Every job starts after the work the calling thread has queued on its current stream so far; when this returns, the calling
    thread's current stream waits for everything the jobs queued.  `kind`: a hashable name for what the jobs do (driver, model,
    shapes); the first job of a kind on each worker runs alone (`first_alone`).  Exceptions of jobs are re-raised here (the first
    one, after all jobs have been collected -- nothing is left running on a worker)."""
    if on_worker():            # nested use (a sweep's attr_fn calling ig_batch): this thread IS a stream; its jobs run on it, in order
        return [job() for job in jobs]
    dev = torch.device(dev)
    main = torch.cuda.current_stream(dev)
    ws = workers(dev, n)
    ready = torch.cuda.Event()
    ready.record(main)

    def wrap(job):
        def run():
            torch.cuda.current_stream(dev).wait_event(ready)
            return job()
        return run

    futs = [first_alone(ws[i % n], kind, wrap(job)) for i, job in enumerate(jobs)]
    concurrent.futures.wait(futs)
    join(dev, ws, main)
    return [f.result() for f in futs]


def join(dev, ws, main=None):
    """The current stream of the calling thread (or `main`) waits for everything queued so far on the workers' streams."""
    main = main if main is not None else torch.cuda.current_stream(dev)
    for w in ws:
        ev = torch.cuda.Event()
        ev.record(w.stream)
        main.wait_event(ev)


class _Turn:
    def __init__(self):
        self.lock = threading.RLock()
        self.last = None                                 # event after the last kernel of the previous backward pass


@contextlib.contextmanager
def backward_turn(dev):
    """`with backward_turn(device): torch.autograd.grad(...)`: backward passes that go through autograd's shared device thread take
    turns (module docstring).  A no-op on a stream `Worker` (its backward runs inline, on its own handles).  Re-entrant (an
    attribution that takes a turn may call helpers that take one); inside a hipGraph capture nothing is chained (the capture as a
    whole, and every replay of a graph that contains backward kernels, is what takes the turn)."""
    if getattr(_local, "inline_autograd", False):
        yield
        return
    dev = torch.device(dev)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    with _TURN_LOCK:
        turn = _TURN.setdefault(idx, _Turn())
    depth = getattr(_local, "depth", {})
    _local.depth = depth
    with turn.lock:
        outer = depth.get(idx, 0) == 0
        depth[idx] = depth.get(idx, 0) + 1
        cur = torch.cuda.current_stream(torch.device("cuda", idx))
        capturing = torch.cuda.is_current_stream_capturing()
        try:
            if outer and not capturing and turn.last is not None:
                cur.wait_event(turn.last)
            yield
        finally:
            depth[idx] -= 1
            if outer and not capturing:
                ev = torch.cuda.Event()
                ev.record(cur)
                turn.last = ev
