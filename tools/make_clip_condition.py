"""The sampled + scaled video latents of the full-size reconstruction fixtures (clip of tools/fullsize_cases.py, CPU generator seed CLIP_SEED), computed ONCE
with the fp32 CPU oracle VAE in the build container (tiled 41-frame encode, ~10 min) and stored as exact bf16 bits in tests/golden/fullsize_clip_condition.npz,
so that tools/make_fullsize_golden_gpu.py can run the reconstruction trajectories on the device without spending its GPU lease on a host-side VAE encode."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fullsize_cases as fc  # noqa: E402


def main():
    torch.set_num_threads(int(os.environ.get("AETHER_GOLDEN_THREADS", os.cpu_count() or 8)))
    vae = fc.build_oracle_vae()
    v = fc.video_as_model_input(fc.clip_video()).to(torch.bfloat16)
    gen = torch.Generator().manual_seed(fc.CLIP_SEED)
    t0 = time.perf_counter()
    with torch.no_grad():
        dist = vae.encode(v.unsqueeze(0).permute(0, 2, 1, 3, 4).float()).latent_dist
    z = dist.mean + dist.std * torch.randn(dist.mean.shape, generator=gen, dtype=torch.bfloat16).float()         # oracle.pipeline.sample's `enc`, compute_dtype = fp32
    lat = (vae.config.scaling_factor * z.to(torch.bfloat16).permute(0, 2, 1, 3, 4))
    meta = dict(seconds_cpu=time.perf_counter() - t0, clip_seed=fc.CLIP_SEED, vae_seed=fc.VAE_SEED, sum=float(lat.double().sum()), abs_sum=float(lat.double().abs().sum()))
    np.savez_compressed(os.path.join(fc.GOLDEN_DIR, "fullsize_clip_condition.npz"), video_latents_bits=fc.bf16_bits(lat), meta=json.dumps(meta))
    print(meta)


if __name__ == "__main__":
    main()
