#!/usr/bin/env python3
"""Direct-to-LDS streaming rate per CU (scripts/lds_stream_probe.hip): what sets the ~45 GB/s per CU of the GEMM kernels?

    python scripts/lds_stream_probe.py        # table on stdout + gpurun_out/lds_stream_probe.json
"""
import ctypes as C
import json
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "lds_stream_probe.hip")
LIB = os.path.join(HERE, "_build", "liblds_stream_probe.so")

VARIANTS = {
    0: "contiguous 1 KiB pieces, 8 waves, ring 3, barrier, reads",
    1: "8 rows x 128 B pieces, 8 waves, ring 3, barrier, reads   (the GEMM kernels' form)",
    2: "  same, no barrier",
    3: "  same, barrier, no fragment reads",
    4: "  same, no barrier, no reads",
    5: "  ring 2",
    6: "  ring 4",
    7: "  4 waves x 6 pieces",
    8: "16 rows x 64 B pieces (BK = 32)",
    9: "8 rows x 128 B, ring 6, no barrier, no reads (pure stream)",
    10: "8 rows x 128 B, 48 KiB stages (6 pieces per wave)",
    11: "8 rows x 128 B, 16 waves x 3 pieces (48 KiB stages), ring 3",
    12: "8 rows x 128 B, 16 waves x 3 pieces (48 KiB stages), ring 2",
    13: "8 rows x 128 B, 16 waves x 2 pieces (32 KiB stages), ring 3",
    14: "GEMM form, fragment reads = 2 x the staged bytes",
    15: "GEMM form, fragment reads = 4 x the staged bytes (what a 128 x 64 tile reads)",
    16: "GEMM form, fragment reads = 8 x the staged bytes",
}


def main():
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-Wno-unused-value", SRC, "-o", LIB])
    import torch
    torch.cuda.init()
    lib = C.CDLL(LIB)
    lib.lsp_run.argtypes = [C.c_int, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                            C.POINTER(C.c_float)]
    buf = torch.zeros(1 << 30, dtype=torch.uint8, device="cuda")  # 1 GiB source
    sink = torch.zeros(1024, dtype=torch.float32, device="cuda")
    out = {}
    nsteps = 64
    for blocks, lds_pad, occ in ((256, 70 * 1024, "1 WG/CU"), (256, 0, "256 WGs, no LDS pad"), (512, 0, "2 WG/CU")):
        for res, wg_stride in (("shared 2 MiB source (L2-resident)", 0), ("own 1.5 MiB per workgroup (MALL / HBM)", 1536 * 1024)):
            for v, name in VARIANTS.items():
                for row_stride in ((1024, 4096) if v in (1, 8) else (1024,)):
                    ppw = {7: 6, 10: 6, 13: 2}.get(v, 3)
                    nw = 4 if v == 7 else (16 if 11 <= v <= 13 else 8)
                    stage = nw * ppw * 1024
                    ms = C.c_float(0)
                    rc = lib.lsp_run(v, buf.data_ptr(), wg_stride, nsteps, row_stride, lds_pad, blocks, 20,
                                     sink.data_ptr(), C.byref(ms))
                    if rc != 0:
                        print("variant", v, "failed", rc)
                        continue
                    total = blocks * nsteps * stage
                    gbps_cu = total / (ms.value * 1e-3) / 1e9 / 256
                    key = f"{occ} | {res} | {name} | row stride {row_stride}"
                    out[key] = round(gbps_cu, 1)
                    print(f"{gbps_cu:7.1f} GB/s per CU  {total / (ms.value * 1e-3) / 1e12:5.2f} TB/s  {ms.value * 1e3:8.1f} us   {key}", flush=True)
    os.makedirs(os.path.join(os.path.dirname(HERE), "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(os.path.dirname(HERE), "gpurun_out", "lds_stream_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
