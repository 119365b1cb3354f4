#!/usr/bin/env python3
"""Root-causing the round-1 observation that K8's captured hipMemsetAsync "once failed to take effect on replay"
(VERDICT r1 item 6).  A SCRATCH build of the library with the memset variant restored (-DXAI_RANK_ZERO_WITH_MEMSET,
built into /tmp, never shipped) is driven through torch.cuda.graph exactly like the failing test
(tests/test_gpu_kernels.py::test_kernels_run_on_the_callers_stream_and_are_graph_capturable), and the failing conditions
are switched on ONE AT A TIME:

  alloc     workspace allocated inside the capture (torch's graph-private pool) | before it (ordinary pool, static)
  mode      torch.cuda.graph(capture_error_mode = global | thread_local | relaxed)
  consumer  the memset is followed by kernels doing global atomics on the zeroed words (the real sort) -- always true here;
            the stand-alone control with a plain consumer is csrc/tune/repro_graph_memset.hip (replays fine)
  reuse     replays per graph (the pointer is re-used by every replay) and fresh inputs per replay

For every cell: N captures x R replays, each replay checked against np.argsort(kind='stable'); prints failures per cell,
and the DOT dump of one captured graph (node list + edges: is the memset node there, and does the first kernel depend on it?).
"""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "image-classification-xai_amd")
sys.path.insert(0, PKG)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def build_variant(define):
    out = os.path.join(tempfile.gettempdir(), f"libxai_rank_{'memset' if define else 'kernel'}.so")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-fvisibility=hidden",
           f"-I{ROOT}/include", f"-I{PKG}/csrc", os.path.join(PKG, "csrc", "rank_kernels.hip"), os.path.join(PKG, "csrc", "abi.hip"), "-o", out]
    if define:
        cmd.insert(1, "-DXAI_RANK_ZERO_WITH_MEMSET")
    subprocess.run(cmd, check=True)
    lib = C.CDLL(out)
    lib.xai_rank_workspace_bytes.restype = C.c_size_t
    lib.xai_rank_workspace_bytes.argtypes = [C.c_int, C.c_int64]
    lib.xai_rank_f32.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    return lib


def rank(lib, sal, ws=None):
    n_seg, hw = sal.shape
    if ws is None:
        ws = torch.empty(lib.xai_rank_workspace_bytes(n_seg, hw), dtype=torch.uint8, device=sal.device)
    order = torch.empty((n_seg, hw), dtype=torch.int32, device=sal.device)
    rk = torch.empty((n_seg, hw), dtype=torch.int32, device=sal.device)
    rc = lib.xai_rank_f32(sal.data_ptr(), n_seg, hw, order.data_ptr(), rk.data_ptr(), ws.data_ptr(), ws.numel(),
                          torch.cuda.current_stream(sal.device).cuda_stream)
    assert rc == 0, rc
    return order, rk, ws


def trial(lib, alloc_inside, mode, n_captures, n_replays, hw, n_seg, dot_path=None):
    dev = torch.device("cuda", 0)
    bad = 0
    first_bad = None
    for cap in range(n_captures):
        rng = np.random.default_rng(1000 * cap + hw)
        sal = torch.from_numpy(rng.standard_normal((n_seg, hw)).astype(np.float32)).to(dev)
        ws = None if alloc_inside else torch.empty(lib.xai_rank_workspace_bytes(n_seg, hw), dtype=torch.uint8, device=dev)
        graph = torch.cuda.CUDAGraph()
        if dot_path and cap == 0:
            graph.enable_debug_mode()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side, capture_error_mode=mode):
                order, _, ws_used = rank(lib, sal, ws)
        if dot_path and cap == 0:
            try:
                graph.debug_dump(dot_path)
            except Exception as e:                      # noqa: BLE001
                open(dot_path, "w").write(f"debug_dump failed: {e}\n")
        for rep in range(n_replays):
            sal.copy_(torch.from_numpy(rng.standard_normal((n_seg, hw)).astype(np.float32)).to(dev))
            if rep % 2:
                ws_used.fill_(0xAB)                    # dirty histogram words between replays: only the captured zero-fill can clear them
            graph.replay()
            torch.cuda.synchronize()
            want = np.argsort(sal.cpu().numpy(), axis=1, kind="stable")
            if not np.array_equal(order.cpu().numpy(), want):
                bad += 1
                first_bad = first_bad or (cap, rep)
        del graph
    return bad, first_bad


def main():
    out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "memset")
    os.makedirs(out_dir, exist_ok=True)
    results = []
    for variant, define in (("hipMemsetAsync", True), ("zero kernel", False)):
        lib = build_variant(define)
        for hw, n_seg in ((1024, 1), (50176, 1), (50176, 3)):
            for alloc_inside in (True, False):
                for mode in ("global", "thread_local", "relaxed"):
                    dot = os.path.join(out_dir, f"graph_{'memset' if define else 'kernel'}_{hw}_{n_seg}.dot") if (alloc_inside and mode == "global") else None
                    bad, first = trial(lib, alloc_inside, mode, n_captures=6, n_replays=6, hw=hw, n_seg=n_seg, dot_path=dot)
                    row = {"zero_fill": variant, "hw": hw, "n_seg": n_seg, "workspace": "allocated inside the capture" if alloc_inside else "allocated before the capture",
                           "capture_error_mode": mode, "replays_checked": 36, "wrong_replays": bad, "first_wrong (capture, replay)": first}
                    results.append(row)
                    print(json.dumps(row), flush=True)
    json.dump(results, open(os.path.join(out_dir, "exp_graph_memset.json"), "w"), indent=1)
    for f in sorted(os.listdir(out_dir)):
        if f.endswith(".dot"):
            txt = open(os.path.join(out_dir, f)).read()
            print(f"--- {f}: {txt.count('->')} edges, {txt.lower().count('memset')} mentions of memset")


if __name__ == "__main__":
    main()
