"""Reference point only (not a product path): torch.nn.functional.linear (hipBLASLt / rocBLAS behind PyTorch-ROCm) at the DiT GEMM
shapes, next to aether_gemm_bf16 in the same process, interleaved rounds, random operands."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aether_amd import ops  # noqa: E402


def timeit(fn, iters=8, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
ws = torch.empty(16 << 20, dtype=torch.float32, device=dev)
for name, (M, N, K) in {"qkv": (15076, 9216, 3072), "out": (15076, 3072, 3072), "ff1": (15076, 12288, 3072), "ff2": (15076, 3072, 12288),
                        "8192^3": (8192, 8192, 8192)}.items():
    A = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
    b32 = torch.randn(N, generator=g, device=dev)
    b16 = b32.to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    res = {"ours": [], "torch": []}
    for _ in range(3):
        res["ours"].append(timeit(lambda: ops.gemm_bf16(A, W, b32, ops.AETHER_EPI_BIAS, out=out, flags=1, splitk_ws=ws)))
        res["torch"].append(timeit(lambda: torch.nn.functional.linear(A, W, b16)))
    fl = 2.0 * M * N * K
    print(json.dumps({"gemm": name, "ours_tflops": round(fl / sorted(res["ours"])[1] / 1e12, 1), "torch_linear_tflops": round(fl / sorted(res["torch"])[1] / 1e12, 1)}), flush=True)
