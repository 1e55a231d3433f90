"""A/B of the VAE launch plan's two lanes (AETHER_VAE_TWO_LANES) at the BASELINE geometry on MI355X: encode of a 41x480x720 clip, decode of an
11x60x90 latent and the pipeline's decode pair, one lane (tile batches of 4/2/2/1 on the caller's stream) against two lanes (batches of two on
two streams).  Writes gpurun_out/vae_lanes_ab.json."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aether_amd import _lib  # noqa: E402
from aether_amd.vae import AetherVAE  # noqa: E402


def timed(fn, reps=3):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps, out


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    x = (torch.rand(1, 3, 41, 480, 720, generator=g, device=dev) * 2 - 1).to(torch.bfloat16)
    z = torch.randn(1, 16, 11, 60, 90, generator=g, device=dev).to(torch.bfloat16)
    z2 = torch.randn(1, 16, 11, 60, 90, generator=g, device=dev).to(torch.bfloat16)
    res, outs = {}, {}
    for name, flags in (("one_lane", _lib.AETHER_GEMM_WIDE_STORE), ("two_lanes", _lib.AETHER_GEMM_WIDE_STORE | _lib.AETHER_VAE_TWO_LANES)):
        vae = AetherVAE(device=dev, flags=flags).init_random_weights(1)
        vae.enable_slicing(); vae.enable_tiling()
        te, oe = timed(lambda: vae.encode(x).latent_dist.mode())
        td, od = timed(lambda: vae.decode(z).sample)
        tp, _ = timed(lambda: vae.decode_pair(z, z2))
        res[name] = {"encode_s": te, "decode_s": td, "decode_pair_s": tp, "encode_mfma_frac": 175.0 / te / 2500, "decode_mfma_frac": 369.0 / td / 2500,
                     "pair_mfma_frac": 738.0 / tp / 2500, "workspace_GB": vae._workspace.numel() / 1e9}
        outs[name] = (oe.float(), od.float())
        print(name, res[name], flush=True)
        del vae
        torch.cuda.empty_cache()
    a, b = outs["one_lane"], outs["two_lanes"]
    res["two_lanes_vs_one_lane"] = {"encode_rel_l2": float((a[0] - b[0]).norm() / a[0].norm()), "decode_rel_l2": float((a[1] - b[1]).norm() / a[1].norm()),
                                    "encode_bit_identical": bool(torch.equal(a[0], b[0])), "decode_bit_identical": bool(torch.equal(a[1], b[1]))}
    print(res["two_lanes_vs_one_lane"])
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/vae_lanes_ab.json", "w"), indent=1)


if __name__ == "__main__":
    main()
