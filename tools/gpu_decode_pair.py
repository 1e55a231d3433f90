"""Two full-size decodes (11 x 60 x 90 latents -> 41 x 480 x 720) one after the other against AetherVAE.decode_pair (two HIP streams)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aether_amd.vae import AetherVAE
dev = torch.device("cuda:0")
vae = AetherVAE(device=dev).init_random_weights(1)
vae.enable_slicing(); vae.enable_tiling()
g = torch.Generator(device=dev).manual_seed(0)
za = torch.randn(1, 16, 11, 60, 90, generator=g, device=dev).to(torch.bfloat16)
zb = torch.randn(1, 16, 11, 60, 90, generator=g, device=dev).to(torch.bfloat16)
for _ in range(3):
    ra, rb = vae.decode(za).sample, vae.decode(zb).sample
for _ in range(3):
    pa, pb = vae.decode_pair(za, zb)
torch.cuda.synchronize()
assert torch.equal(ra, pa) and torch.equal(rb, pb)
res = {}
for name, fn in (("sequential", lambda: (vae.decode(za).sample, vae.decode(zb).sample)), ("pair_two_streams", lambda: vae.decode_pair(za, zb)),
                 ("sequential_again", lambda: (vae.decode(za).sample, vae.decode(zb).sample)), ("pair_again", lambda: vae.decode_pair(za, zb))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    res[name] = (time.perf_counter() - t0) / 3
res["tflops_pair"] = 2 * 369.0 / res["pair_two_streams"]
res["tflops_sequential"] = 2 * 369.0 / res["sequential"]
print(json.dumps(res))
