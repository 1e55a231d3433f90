import os, sys, torch
sys.path.insert(0, "/root/repo")
from aether_amd import _lib
from aether_amd.vae import AetherVAE
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
x = (torch.rand(1, 3, 41, 480, 720, generator=g, device=dev) * 2 - 1).to(torch.bfloat16)
z = torch.randn(1, 16, 11, 60, 90, generator=g, device=dev).to(torch.bfloat16)
def timed(fn, reps=3):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps
vae = AetherVAE(device=dev).init_random_weights(1); vae.enable_slicing(); vae.enable_tiling()
print("gmax", os.environ.get("AETHER_VAE_GMAX_EXPERIMENT"), "encode", timed(lambda: vae.encode(x).latent_dist.mode()), "decode", timed(lambda: vae.decode(z).sample), flush=True)
