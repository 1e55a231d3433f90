"""Positional-table choice of the transformer at the BASELINE geometry (S = 226 + 14 850 tokens, real width, 2 blocks) against the fp32 CPU
oracle.  (The 2-block / 17-frame parity cases that lived here are superseded by the full-depth, whole-clip fixtures of
tests/test_fullsize_parity_gpu.py and tests/test_fullsize_guided_gpu.py.)
"""
import math
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

def _metrics(out: torch.Tensor, ref: torch.Tensor):
    d = (out.double() - ref.double())
    return {"rel_l2": (d.norm() / ref.double().norm()).item(), "linf": d.abs().max().item(), "ref_max": ref.abs().max().item(),
            "linf_rel": (d.abs().max() / ref.abs().max()).item()}


def _psnr(a: torch.Tensor, b: torch.Tensor) -> float:
    mse = ((a.double() - b.double()) ** 2).mean().item()
    return 10 * math.log10(1.0 / max(mse, 1e-20))


@pytest.mark.parametrize("frames,table", [(13, "learned"), (11, "sincos")])
def test_dit_positional_table_choice(cuda, hip_lib, frames, table):
    """use_learned_positional_embeddings with sample_frames = 49 (the CogVideoX-5b-I2V base): a 49-frame clip (13 latent
    frames) adds the learned table; AetherV1's 41-frame clips (11 latent frames) get the 3-D sin-cos table of the actual size
    (diffusers CogVideoXPatchEmbed.forward; never a slice of the learned table)."""
    from aether_amd.transformer import AetherTransformer3D
    from oracle.dit import DitConfig, OracleTransformer3D, init_random_
    from oracle.rope import crop_region_for_grid, rope_3d
    cfg = DitConfig(num_attention_heads=8, num_layers=2, text_embed_dim=128, time_embed_dim=64, max_text_seq_length=20,
                    sample_width=12, sample_height=8, sample_frames=49, use_learned_positional_embeddings=True)
    oracle = init_random_(OracleTransformer3D(cfg), seed=5)
    sd = {k: v.to(torch.bfloat16) for k, v in oracle.state_dict().items()}
    oracle.load_state_dict({k: v.float() for k, v in sd.items()})
    g = torch.Generator().manual_seed(0)
    hidden = torch.randn(1, frames, 96, 8, 12, generator=g).to(torch.bfloat16)
    text = (torch.randn(1, 20, 128, generator=g) * 0.1).to(torch.bfloat16)
    t = torch.tensor([500], dtype=torch.int64)
    rope = rope_3d(64, crop_region_for_grid((4, 6), 6, 4), (4, 6), frames)
    with torch.no_grad():
        ref = oracle(hidden.float(), text.float(), t, image_rotary_emb=rope)[0]
    native = AetherTransformer3D({k: getattr(cfg, k) for k in cfg.__dataclass_fields__}, device=cuda).load_state_dict(sd)
    out = native(hidden_states=hidden.to(cuda), encoder_hidden_states=text.to(cuda), timestep=t.to(cuda), ofs=None,
                 image_rotary_emb=(rope[0].to(cuda), rope[1].to(cuda)), return_dict=False)[0]
    torch.cuda.synchronize()
    assert native._pos_current == ("learned" if table == "learned" else ("sincos", frames))
    m = _metrics(out.cpu().float(), ref)
    print(f"\npositional table '{table}': rel-L2 {m['rel_l2']:.3e}  L-inf {100 * m['linf_rel']:.2f} % of max|ref|")
    assert m["rel_l2"] <= 1.0e-2 and m["linf_rel"] <= 0.06, m
    # the learned table only exists at the sample resolution (diffusers raises the same way)
    with pytest.raises(ValueError, match="different resolution"):
        native(hidden_states=hidden[:, :, :, :6].to(cuda), encoder_hidden_states=text.to(cuda), timestep=t.to(cuda), ofs=None,
               image_rotary_emb=(rope[0][: frames * 18].to(cuda), rope[1][: frames * 18].to(cuda)), return_dict=False)


def test_dit_positional_table_bounds(cuda, hip_lib):
    """aether_dit_forward refuses a positional table shorter than text + video tokens instead of reading past it."""
    from aether_amd import _lib
    from aether_amd.transformer import AetherTransformer3D
    m = AetherTransformer3D({"num_layers": 1, "num_attention_heads": 8, "text_embed_dim": 128, "time_embed_dim": 64, "max_text_seq_length": 20,
                             "sample_width": 12, "sample_height": 8, "sample_frames": 9, "use_learned_positional_embeddings": True},
                            device=cuda).init_random_weights(0)
    x = torch.zeros(1, 3, 96, 8, 12, dtype=torch.bfloat16, device=cuda)
    args = dict(hidden_states=x, encoder_hidden_states=torch.zeros(1, 20, 128, device=cuda), timestep=torch.zeros(1, device=cuda),
                image_rotary_emb=(torch.zeros(72, 64), torch.zeros(72, 64)))
    m(**args)                                              # 20 + 72 rows registered: fine
    short = torch.zeros(50, 512, dtype=torch.bfloat16, device=cuda)
    _lib.check(m._lib.aether_dit_set_pos_embedding(m._handle, short.data_ptr(), 50), "set_pos_embedding")
    m._pos_tables[m._pos_current] = short                  # keep the wrapper from re-registering the right one
    m._select_pos_table = lambda *a: None
    with pytest.raises(ValueError, match="fewer rows"):
        m(**args)


def _video(frames, H, W):
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    g = np.random.default_rng(0)
    v = np.stack([np.stack([0.5 + 0.4 * np.sin(0.02 * xx + 0.1 * t + c) * np.cos(0.015 * yy) for c in range(3)], 0) for t in range(frames)], 1)
    v = v + 0.03 * g.standard_normal(v.shape).astype(np.float32)
    return torch.from_numpy(v.astype(np.float32))[None] * 2 - 1            # [1, 3, F, H, W] in [-1, 1]
