"""Attention-only probe at the BASELINE shape (B=1, 48 heads, S=15 076): TF/s of aether_flash_attn_fwd for a flags value, with the
per-tile bounds exactly as aether_qk_norm_rope emits them.  Used under rocprofv3 --pmc (tools/profile_attn.sh)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aether_amd import ops  # noqa: E402
from aether_amd._lib import ATTN_Q_SCALE  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--flags", type=int, default=1)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--S", type=int, default=15076)
    ap.add_argument("--gain", type=float, default=1.0, help="q/k norm gain (1: unit-scale LayerNorm outputs)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    H, S = 48, a.S
    Spad = (S + 63) // 64 * 64
    unitish = lambda: torch.nn.functional.layer_norm(torch.randn(1, H, S, 64, generator=g, device=dev), (64,)) * a.gain  # noqa: E731
    q, k = (unitish() * ATTN_Q_SCALE).to(torch.bfloat16), unitish().to(torch.bfloat16)
    vt = torch.zeros(1, H, 64, Spad, dtype=torch.bfloat16, device=dev)
    vt[..., :S] = torch.randn(1, H, 64, S, generator=g, device=dev).to(torch.bfloat16)
    for _ in range(2):
        ops.flash_attn_fwd(q, k, vt, flags=a.flags)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        ops.flash_attn_fwd(q, k, vt, flags=a.flags)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    print(json.dumps({"flags": a.flags, "gain": a.gain, "ms": ms, "tflops": 4.0 * S * S * 64 * H / (ms * 1e-3) / 1e12}))


if __name__ == "__main__":
    main()
