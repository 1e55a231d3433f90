import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aether_amd import _lib
from aether_amd.transformer import AetherTransformer3D
from aether_amd.rope import resize_crop_region_for_grid, rotary_tables_3d
dev = torch.device("cuda:0")
base = _lib.AETHER_GEMM_WIDE_STORE | _lib.AETHER_GEMM_PINGPONG | _lib.AETHER_ATTN_PAIR_PIPELINE | _lib.AETHER_ATTN_QREG
F, H, W = 3, 60, 90
g = torch.Generator(device=dev).manual_seed(0)
hidden = torch.randn(1, F, 96, H, W, generator=g, device=dev).to(torch.bfloat16)
text = (torch.randn(1, 226, 4096, generator=g, device=dev) * 0.1).to(torch.bfloat16)
t = torch.tensor([499.0], device=dev)
rope = rotary_tables_3d(64, resize_crop_region_for_grid((30, 45), 45, 30), (30, 45), F, 1.0, device=dev)
outs = {}
for name, fl in (("two_pass", base), ("fused", base | _lib.AETHER_DIT_FUSE_QKV_PREP), ("two_pass_again", base), ("two_pass_conservative_attn", base | _lib.AETHER_ATTN_EXACT_MAX),
                 ("fused_conservative_attn", base | _lib.AETHER_DIT_FUSE_QKV_PREP | _lib.AETHER_ATTN_EXACT_MAX)):
    m = AetherTransformer3D({"num_layers": 1, "sample_frames": 9}, device=dev, flags=fl).init_random_weights(0)
    outs[name] = m(hidden_states=hidden, encoder_hidden_states=text, timestep=t, image_rotary_emb=rope)[0].float()
    torch.cuda.synchronize()
ref = outs["two_pass"]
for k, v in outs.items():
    d = (v - ref)
    per_tok = d.reshape(F, 56, 30, 2, 45, 2).permute(0, 2, 4, 1, 3, 5).reshape(F * 30 * 45, -1).norm(dim=1)     # token order (frame, row, col)
    nz = (per_tok > 0).nonzero().flatten()
    print(f"{k}: rel-L2 vs two_pass {(d.norm() / ref.norm()).item():.3e}; tokens that differ {nz.numel()} of {per_tok.numel()}"
          + (f" (first {nz[0].item()}, last {nz[-1].item()}); worst token {per_tok.argmax().item()}" if nz.numel() else ""))
