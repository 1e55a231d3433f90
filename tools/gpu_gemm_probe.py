"""GEMM-only probe at a DiT shape (default: ff-up, [15076, 3072] x [12288, 3072]^T + GELU) for rocprofv3 --pmc passes."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aether_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--flags", type=int, default=1)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--M", type=int, default=15076)
    ap.add_argument("--N", type=int, default=12288)
    ap.add_argument("--K", type=int, default=3072)
    ap.add_argument("--split-n", type=int, default=1, help="run the GEMM as this many launches over equal column slices of W / C (N = 3072, 3 slices: 59 x 4 = 236 "
                    "tiles per launch = ONE exact round on 236 CUs, the vendor kernel's grid, instead of 708 tiles = 2.77 rounds of 256)")
    ap.add_argument("--epi", type=int, default=ops.AETHER_EPI_BIAS_GELU)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    A = torch.randn(a.M, a.K, generator=g, device=dev).to(torch.bfloat16)
    W = (torch.randn(a.N, a.K, generator=g, device=dev) * a.K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(a.N, generator=g, device=dev)
    out = torch.empty(a.M, a.N, dtype=torch.bfloat16, device=dev)
    ws = torch.empty(16 << 20, dtype=torch.float32, device=dev)
    ns = a.N // a.split_n

    def fn():
        for i in range(a.split_n):
            ops.gemm_bf16(A, W[i * ns:(i + 1) * ns], bias[i * ns:(i + 1) * ns], a.epi, out=out[:, i * ns:(i + 1) * ns], flags=a.flags, splitk_ws=ws)
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    print(json.dumps({"flags": a.flags, "split_n": a.split_n, "M": a.M, "N": a.N, "K": a.K, "ms": ms, "tflops": 2.0 * a.M * a.N * a.K / (ms * 1e-3) / 1e12}))


if __name__ == "__main__":
    main()
