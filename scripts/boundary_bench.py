#!/usr/bin/env python3
"""Per-boundary cost of dependent kernel launches on MI355X: eager C loop vs a hipGraph captured by plain HIP calls
vs a torch CUDAGraph (what TrainStep replays).  Settles profiles/README.md's "4.8 us per dependent kernel" against
MI355X_MICROARCH.md's 1.1-1.9 us (row `boundary`).

    python scripts/boundary_bench.py            # table on stdout + gpurun_out/boundary.json
"""
import ctypes as C
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "boundary_bench.hip")
LIB = os.path.join(HERE, "_build", "libboundary_bench.so")


def build():
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", SRC, "-o", LIB])
    return LIB


def main():
    build()
    if "--build-only" in sys.argv:
        return
    import torch
    torch.cuda.init()  # torch's own HIP runtime must be the one this process uses: load it before the library
    lib = C.CDLL(LIB)
    lib.bb_time.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_float)]
    lib.bb_chain.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]

    def c_time(kind, n, blocks, nbytes, mode, reps=20):
        out = C.c_float(0)
        rc = lib.bb_time(kind, n, blocks, nbytes, mode, reps, C.byref(out))
        assert rc == 0, rc
        return out.value

    def torch_graph_time(kind, n, blocks, nbytes, reps=20):
        a = torch.ones(max(nbytes, 16), dtype=torch.uint8, device="cuda")
        b = torch.ones(max(nbytes, 16), dtype=torch.uint8, device="cuda")
        cnt = torch.zeros(64, dtype=torch.int32, device="cuda")

        def run():
            rc = lib.bb_chain(kind, n, blocks, a.data_ptr(), b.data_ptr(), nbytes, cnt.data_ptr(),
                              torch.cuda.current_stream().cuda_stream)
            assert rc == 0
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            run()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            run()
        g.replay()
        torch.cuda.synchronize()
        st = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            g.replay()
        e1.record(st)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / (reps * n)

    rows = []

    def row(name, kind, n, blocks, nbytes):
        r = {"chain": name, "n": n, "eager_us": round(c_time(kind, n, blocks, nbytes, 0), 3),
             "hipgraph_us": round(c_time(kind, n, blocks, nbytes, 1), 3),
             "torch_graph_us": round(torch_graph_time(kind, n, blocks, nbytes), 3)}
        rows.append(r)
        print(f'{name:46s} n={n:4d}  eager {r["eager_us"]:8.2f}  hipGraph {r["hipgraph_us"]:8.2f}  '
              f'torch graph {r["torch_graph_us"]:8.2f}   us per link', flush=True)

    for n in (16, 64, 256):
        for blocks in (1, 256, 1024):
            row(f"trivial, {blocks} workgroups", 0, n, blocks, 0)
    MB = 1 << 20
    for mb in (1, 4, 13, 26, 64):
        row(f"copy {mb} MB (plain stores)", 1, 64, 0, mb * MB)
        row(f"copy {mb} MB (nontemporal stores)", 2, 64, 0, mb * MB)
        row(f"copy {mb} MB + trivial 256 WG (2 launches/link)", 3, 64, 256, mb * MB)
    # one launch moving 64 x S bytes: the no-boundary reference for the copy chains
    for mb in (64, 64 * 13):
        row(f"ONE copy launch of {mb} MB", 1, 1, 0, mb * MB)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "boundary.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
