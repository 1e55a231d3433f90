"""GEMM main-loop variants at the DiT shapes ("buy clock, not cycles"): builds tools/probes/gemm_variants_probe.hip once per knob and times every
variant — interleaved rounds, median — on the qkv (15076 x 9216 x 3072) and ff-up (15076 x 12288 x 3072, GELU) shapes, next to the vendor GEMM behind
torch.nn.functional.linear (reference point only).  With --pmc-run NAME it runs ONLY that variant a few times (one rocprofv3 --pmc pass per variant:
cycles, LDS / VMEM / MFMA instruction counts).  Measurement only."""
import argparse
import ctypes
import json
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "probes", "gemm_variants_probe.hip")
VARIANTS = {"shipped": [], "ksps2": ["-DAETHER_GEMM_KSPS=2"], "snake": ["-DAETHER_GEMM_MFMA_ORDER=1"], "nt_outer": ["-DAETHER_GEMM_MFMA_ORDER=2"], "noprio": ["-DAETHER_GEMM_SETPRIO=0"]}
SHAPES = {"qkv": (15076, 9216, 3072, 0), "ff1": (15076, 12288, 3072, 1)}


def build(name):
    so = os.path.join(HERE, "probes", f"gemm_variant_{name}.so")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(SRC), os.path.getmtime(os.path.join(HERE, "..", "aether_amd", "csrc", "gemm_kernel.hpp"))):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", *VARIANTS[name], SRC, "-o", so])
    lib = ctypes.CDLL(so)
    lib.run_gemm_variant.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--pmc-run", default=None)
    ap.add_argument("--out", default="gpurun_out/gemm_variants.json")
    a = ap.parse_args()
    libs = {n: build(n) for n in VARIANTS}
    if a.build_only:
        return
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    res = {}
    for sname, (M, N, K, epi) in SHAPES.items():
        A = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
        W = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
        bias = torch.randn(N, generator=g, device=dev)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)

        def run(name):
            if name == "vendor":
                torch.nn.functional.linear(A, W, bias.to(torch.bfloat16))
            else:
                rc = libs[name].run_gemm_variant(A.data_ptr(), K, W.data_ptr(), K, out.data_ptr(), N, M, N, K, bias.data_ptr(), epi, None)
                assert rc == 0

        if a.pmc_run:
            for _ in range(4):
                run(a.pmc_run)
            torch.cuda.synchronize()
            continue
        ref = None
        names = list(VARIANTS) + ["vendor"]
        times = {n: [] for n in names}
        for n in names:
            run(n); run(n)
            torch.cuda.synchronize()
            if n == "shipped":
                ref = out.clone()
            elif n != "vendor":
                assert torch.equal(out, ref), f"{n}: result differs from the shipped loop"      # same K order, same epilogue: bit-identical
        for _ in range(5):
            for n in names:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    run(n)
                e1.record()
                torch.cuda.synchronize()
                times[n].append(e0.elapsed_time(e1) / 10)
        res[sname] = {n: {"us": round(sorted(t)[2] * 1e3, 1), "TFLOPs": round(2.0 * M * N * K / (sorted(t)[2] * 1e-3) / 1e12, 1)} for n, t in times.items()}
        print(sname, json.dumps(res[sname]), flush=True)
        del A, W, out
    if not a.pmc_run:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
