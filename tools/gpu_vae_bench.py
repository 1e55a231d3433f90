"""Full-size VAE timing on MI355X (41x480x720 clip, tiled + frame-batched exactly as the reference runs it).
Writes gpurun_out/vae_bench.json.  Algorithmic work: encode 175 TFLOP tiled, decode 369 TFLOP tiled."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aether_amd.vae import AetherVAE  # noqa: E402


def main():
    import argparse
    from aether_amd import _lib
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", choices=["encode", "decode"], default=None, help="time one direction only (per-direction rocprofv3 traces)")
    ap.add_argument("--lanes", type=int, default=2, choices=[1, 2], help="1 = no AETHER_VAE_TWO_LANES: kernels of one call do not overlap, so a kernel trace attributes time per kernel")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--out", default="gpurun_out/vae_bench.json")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    flags = _lib.AETHER_GEMM_WIDE_STORE | (_lib.AETHER_VAE_TWO_LANES if args.lanes == 2 else 0)
    vae = AetherVAE(device=dev, flags=flags).init_random_weights(0)
    vae.enable_tiling(); vae.enable_slicing()
    g = torch.Generator(device=dev).manual_seed(0)
    yy, xx = torch.meshgrid(torch.arange(480, device=dev).float(), torch.arange(720, device=dev).float(), indexing="ij")
    video = torch.stack([torch.stack([torch.sin(0.02 * xx + 0.1 * t + c) * torch.cos(0.015 * yy) for c in range(3)]) for t in range(41)], 1)
    video = (video * 0.8 + 0.05 * torch.randn(video.shape, generator=g, device=dev))[None].to(torch.bfloat16)
    res = {}
    for name, fn, flop in (("encode_41x480x720", lambda: vae.encode(video).latent_dist.parameters, 175e12),
                           ("decode_11x60x90", None, 369e12)):
        if args.only and not name.startswith(args.only):
            continue
        if fn is None:
            z = (torch.randn(1, 16, 11, 60, 90, generator=g, device=dev)).to(torch.bfloat16)
            fn = lambda: vae.decode(z).sample  # noqa: E731
        out = fn()
        torch.cuda.synchronize()
        times = []
        for _ in range(args.reps):
            t0 = time.perf_counter()
            out = fn()
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        t = min(times)
        res[name] = {"lanes": args.lanes, "seconds": t, "TFLOPs_algorithmic": flop / t / 1e12, "frac_mfma_peak": flop / t / 2.5e15,
                     "out_shape": list(out.shape), "finite": bool(torch.isfinite(out.float()).all()), "out_std": float(out.float().std())}
        print(name, res[name], flush=True)
    res["peak_mem_GB"] = torch.cuda.max_memory_allocated() / 1e9
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
