"""Kernel-level microbenchmarks at the BASELINE shapes (run on the MI355X through gpurun).

Times each hot kernel in isolation with HIP events on the launch stream and writes gpurun_out/microbench.json:
TFLOP/s for the GEMMs / attention against the 2.5 PFLOP/s dense bf16 MFMA peak, GB/s for the streaming kernels
against 8 TB/s.  Random (not zero) operands, as the CDNA guide requires for quotable numbers.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from aether_amd import ops  # noqa: E402
from aether_amd._lib import ATTN_Q_SCALE  # noqa: E402


def timeit(fn, iters=10, warmup=3):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/microbench.json")
    ap.add_argument("--S", type=int, default=15076)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--only", default="", help="comma list of sections: gemm,attn,stream")
    ap.add_argument("--attn-rounds", type=int, default=3)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    res = {"device": torch.cuda.get_device_name(0), "S": args.S, "results": []}
    S, D, FF, H = args.S, 3072, 12288, 48

    def rnd(*shape, scale=1.0, dtype=torch.bfloat16):
        return (torch.randn(*shape, generator=g, device=dev, dtype=torch.float32) * scale).to(dtype)

    only = set(x for x in args.only.split(",") if x)
    splitk_ws = torch.empty(16 << 20, dtype=torch.float32, device=dev)   # 64 MiB scratch for the split-K tail launch
    # ---- GEMMs -------------------------------------------------------------------------------
    for name, (M, N, K, epi) in ({} if (only and "gemm" not in only) else {
        "gemm_qkv": (S, 3 * D, D, ops.AETHER_EPI_BIAS),
        "gemm_out": (S, D, D, ops.AETHER_EPI_BIAS_GATE_RES),
        "gemm_ff1": (S, FF, D, ops.AETHER_EPI_BIAS_GELU),
        "gemm_ff2": (S, D, FF, ops.AETHER_EPI_BIAS_GATE_RES),
        "gemm_4096_cube": (4096, 4096, 4096, ops.AETHER_EPI_BIAS),
        "gemm_8192_cube": (8192, 8192, 8192, ops.AETHER_EPI_BIAS),
    }).items():
        A, W, bias = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N, dtype=torch.float32)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        R = rnd(M, N) if epi == ops.AETHER_EPI_BIAS_GATE_RES else None
        gate = rnd(1, 2 * N, dtype=torch.float32) if R is not None else None
        variants = ((1, True), (1, False))     # the ping-pong loop with and without the split-K tail launch
        kw = dict(R=R, gate_vid=gate[:, :N], gate_txt=gate[:, N:], rows_per_batch=M, n_text=226) if R is not None else {}
        times = {v: [] for v in variants}
        for _ in range(3):                      # interleaved rounds (guide rule 24): median reported
            for v in variants:
                flags, use_ws = v
                times[v].append(timeit(lambda: ops.gemm_bf16(A, W, bias, epi, out=out, flags=flags, splitk_ws=splitk_ws if use_ws else None, **kw)))
        for (flags, use_ws), ts in times.items():
            t = sorted(ts)[1]
            tf = 2.0 * M * N * K / t / 1e12
            res["results"].append({"kernel": name, "flags": flags, "tail_split_k": use_ws, "M": M, "N": N, "K": K, "ms": t * 1e3, "TFLOPs": tf,
                                   "frac_mfma_peak": tf / 2500.0})
            print(res["results"][-1], flush=True)
        del A, W, out, R

    # ---- attention ---------------------------------------------------------------------------
    # q, k shaped like LayerNorm(64) outputs (norm 8, the statistics the DiT feeds the kernel), q pre-scaled by
    # log2(e)/8.  Every kernel variant, with the exact online soft-max ("exact") and with the score bound supplied
    # ("bounded": the no-maximum path).  Interleaved rounds in one process (guide rule 24): median and min reported.
    for B in (() if (only and "attn" not in only) else ((1,) if args.quick else (1, 2))):
        def unit8(*shape):
            x = torch.randn(*shape, generator=g, device=dev, dtype=torch.float32)
            return x / x.norm(dim=-1, keepdim=True) * 8.0
        q = (unit8(B, H, S, 64) * ATTN_Q_SCALE).to(torch.bfloat16)
        k = unit8(B, H, S, 64).to(torch.bfloat16)
        Spad = (S + 63) // 64 * 64
        vt = rnd(B, H, 64, Spad)
        vt[..., S:] = 0
        variants = [(f, False) for f in (1, 1 | 32)]       # default (optimistic tile-pair sweep) and the conservative path alone
        times = {v: [] for v in variants}
        for rnd_i in range(args.attn_rounds):
            for v in variants:
                f, bnd = v
                times[v].append(timeit(lambda: ops.flash_attn_fwd(q, k, vt, flags=f), iters=3, warmup=1))
        for (f, bnd), ts in times.items():
            ts = sorted(ts)
            med, mn = ts[len(ts) // 2], ts[0]
            fl = 4.0 * B * H * S * S * 64
            res["results"].append({"kernel": "flash_attn", "flags": f, "softmax": "conservative" if f & 32 else "optimistic", "B": B, "H": H, "S": S,
                                   "ms_median": med * 1e3, "ms_min": mn * 1e3, "TFLOPs": fl / med / 1e12, "TFLOPs_best": fl / mn / 1e12,
                                   "frac_mfma_peak": fl / med / 1e12 / 2500.0})
            print(res["results"][-1], flush=True)
        del q, k, vt

    # ---- streaming kernels -------------------------------------------------------------------
    if only and "stream" not in only:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        json.dump(res, open(args.out, "w"), indent=1)
        return
    x = rnd(S, D)
    w, b = rnd(D, dtype=torch.float32), rnd(D, dtype=torch.float32)
    mod = rnd(1, 6 * D, dtype=torch.float32)
    y = torch.empty_like(x)
    t = timeit(lambda: ops.layernorm_modulate(x, w, b, 1e-5, mod[:, :D], mod[:, D:2 * D], mod[:, 3 * D:4 * D], mod[:, 4 * D:5 * D],
                                              rows_per_batch=S, n_text=226, out=y))
    res["results"].append({"kernel": "layernorm_modulate", "ms": t * 1e3, "GBps": 2 * S * D * 2 / t / 1e9, "frac_hbm_peak": 2 * S * D * 2 / t / 8e12})
    print(res["results"][-1], flush=True)
    qkv = rnd(1, S, 3 * D)
    nw, nb = rnd(64, dtype=torch.float32), rnd(64, dtype=torch.float32)
    cos, sin = rnd(S - 226, 64, dtype=torch.float32), rnd(S - 226, 64, dtype=torch.float32)
    for _ in (0,):
        t = timeit(lambda: ops.qk_norm_rope(qkv, H, 226, nw, nb, nw, nb, 1e-6, cos, sin, ATTN_Q_SCALE))
        res["results"].append({"kernel": "qk_norm_rope+v_transpose", "ms": t * 1e3, "GBps": 2 * S * 3 * D * 2 / t / 1e9,
                               "frac_hbm_peak": 2 * S * 3 * D * 2 / t / 8e12, "note": "includes torch.empty/zeros of the outputs"})
        print(res["results"][-1], flush=True)
    Wada = rnd(42 * 12 * D + 2 * D, 512, scale=0.05)
    temb = rnd(1, 512, dtype=torch.float32)
    t = timeit(lambda: ops.gemv_rows(temb, Wada, None, 1, 0))
    res["results"].append({"kernel": "adaln_gemv_all_layers", "ms": t * 1e3, "GBps": Wada.numel() * 2 / t / 1e9, "frac_hbm_peak": Wada.numel() * 2 / t / 8e12})
    print(res["results"][-1], flush=True)

    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    t0 = time.time()
    main()
    print(f"microbench wall {time.time() - t0:.1f}s")
