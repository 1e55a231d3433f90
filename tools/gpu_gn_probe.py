"""GroupNorm / causal-front kernels of the VAE at the decoder's real shapes (41x480x720 clip, tiles 240x360): time per
call and effective HBM rate (algorithmic bytes: partial = one read; apply = one read + one write).  MI355X only."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from aether_amd.vae import AetherVAE
    dev = torch.device("cuda:0")
    vae = AetherVAE(device=dev).init_random_weights(1)
    norm_by_c = {}
    for blk in vae.dec.up:
        for r in blk.resnets:
            norm_by_c.setdefault(r.norm1.gamma.numel(), r.norm1)
            norm_by_c.setdefault(r.norm2.gamma.numel(), r.norm2)
    shapes = [(4, 3, 30, 45, 512), (2, 4, 60, 90, 512), (2, 4, 60, 90, 256), (2, 8, 120, 180, 256), (1, 8, 240, 360, 256), (1, 8, 240, 360, 128)]
    out = []
    for shp in shapes:
        NB, T, H, W, Cc = shp
        x = torch.randn(shp, device=dev).to(torch.bfloat16)
        zq = torch.randn(NB, max(T // 4, 1) + (1 if T == 3 else 0), 30, 45, 16, device=dev).to(torch.bfloat16)
        norm = norm_by_c[Cc]
        cache = {}

        def full():
            vol = vae._norm_to_padded(x, norm, 2, 1, True, zq, 1e-6)
            vae._causal_front(vol, cache, "k")
            return vol

        for _ in range(3):
            full()
        torch.cuda.synchronize()
        prof = {}
        # events around each stage
        def timed(label, fn, n=20):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record(); torch.cuda.synchronize()
            prof[label] = e0.elapsed_time(e1) / n * 1e3      # us
        G = vae.config.norm_num_groups
        V = T * H * W
        nblk = max(1, min(256, (V + 127) // 128))
        part = torch.empty(NB * 4096 * 2 * Cc, dtype=torch.float32, device=dev)
        stats = torch.empty(NB * G * 2, dtype=torch.float32, device=dev)
        affine = torch.empty(NB * 2 * Cc, dtype=torch.float32, device=dev)
        lib, st = vae._lib, vae._stream()
        timed("stats(partial+finalize)", lambda: lib.aether_groupnorm_stats(x.data_ptr(), NB, V, Cc, G, 1e-6, norm.gamma.data_ptr(), norm.beta.data_ptr(),
                                                                            part.data_ptr(), nblk, stats.data_ptr(), affine.data_ptr(), st))
        timed("norm_to_padded(stats+cond+apply)", lambda: vae._norm_to_padded(x, norm, 2, 1, True, zq, 1e-6))
        vol = vae._norm_to_padded(x, norm, 2, 1, True, zq, 1e-6)
        timed("causal_front", lambda: vae._causal_front(vol, cache, "k"))
        nbytes = x.numel() * 2
        prof["stats_GBps"] = nbytes / prof["stats(partial+finalize)"] / 1e3
        apply_us = prof["norm_to_padded(stats+cond+apply)"] - prof["stats(partial+finalize)"]
        prof["apply+cond_us"] = apply_us
        prof["apply_GBps"] = 2 * nbytes / apply_us / 1e3
        out.append(dict(shape=shp, MB=nbytes / 1e6, **{k: round(v, 1) for k, v in prof.items()}))
        print(json.dumps(out[-1]), flush=True)


if __name__ == "__main__":
    main()
