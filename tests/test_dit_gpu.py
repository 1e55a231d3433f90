"""Whole-transformer parity: aether_dit_forward (HIP, bf16 activations) against the fp32 CPU oracle on identical
bf16-representable weights and inputs.  Tolerance is calibrated, not guessed: the oracle itself is re-run with bf16
weights/activations (the reference's dtype, /root/reference/scripts/demo.py:218,226) and the native path must be no
further from the fp32 oracle than 1.5x that distance (+ a 2e-3 floor)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _small_cfg(**kw):
    from oracle.dit import DitConfig
    base = dict(num_attention_heads=8, attention_head_dim=64, in_channels=96, out_channels=56, num_layers=2, text_embed_dim=128,
                time_embed_dim=64, max_text_seq_length=20, sample_width=12, sample_height=8, sample_frames=9)
    base.update(kw)
    return DitConfig(**base)


def _inputs(cfg, B, F, H, W, seed=0):
    g = torch.Generator().manual_seed(seed)
    hidden = torch.randn(B, F, cfg.in_channels, H, W, generator=g).to(torch.bfloat16)
    text = (torch.randn(B, cfg.max_text_seq_length, cfg.text_embed_dim, generator=g) * 0.1).to(torch.bfloat16)
    t = torch.tensor([999, 499][:B], dtype=torch.int64)
    return hidden, text, t


def _rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm()).item()


def _run_pair(cfg, B, F, H, W, cuda, fps=12, flags=None):
    from aether_amd.transformer import AetherTransformer3D
    from oracle.dit import OracleTransformer3D, init_random_
    from oracle.rope import rope_3d, crop_region_for_grid
    oracle = init_random_(OracleTransformer3D(cfg), seed=1)
    sd = {k: v.to(torch.bfloat16) for k, v in oracle.state_dict().items()}      # bf16-representable weights for all
    oracle.load_state_dict({k: v.float() for k, v in sd.items()})
    hidden, text, t = _inputs(cfg, B, F, H, W)
    p = cfg.patch_size
    crops = crop_region_for_grid((H // p, W // p), cfg.sample_width // p, cfg.sample_height // p)
    rope = rope_3d(64, crops, (H // p, W // p), F, fps_factor=12 / fps)
    ref = oracle(hidden.float(), text.float(), t, image_rotary_emb=rope)[0]
    o16 = OracleTransformer3D(cfg).to(torch.bfloat16)
    o16.load_state_dict(sd)
    ref16 = o16(hidden, text, t, image_rotary_emb=rope)[0]
    kw = {} if flags is None else {"flags": flags}
    native = AetherTransformer3D(vars(cfg) if not hasattr(cfg, "__dataclass_fields__") else {k: getattr(cfg, k) for k in cfg.__dataclass_fields__}, device=cuda, **kw)
    native.load_state_dict(sd)
    out = native(hidden_states=hidden.to(cuda), encoder_hidden_states=text.to(cuda), timestep=t.to(cuda), ofs=None,
                 image_rotary_emb=(rope[0].to(cuda), rope[1].to(cuda)), attention_kwargs=None, return_dict=False)[0]
    torch.cuda.synchronize()
    return out.cpu(), ref, ref16


@pytest.mark.parametrize("B,learned", [(1, False), (2, False), (2, True)])
def test_dit_forward_small(cuda, hip_lib, B, learned):
    cfg = _small_cfg(use_learned_positional_embeddings=learned)
    out, ref, ref16 = _run_pair(cfg, B, 3, 8, 12, cuda)
    assert out.shape == ref.shape == (B, 3, 56, 8, 12)
    assert torch.isfinite(out.float()).all()
    e_native, e_bf16 = _rel(out, ref), _rel(ref16, ref)
    print(f"rel-L2 native vs fp32 oracle {e_native:.3e}; bf16 oracle vs fp32 oracle {e_bf16:.3e}")
    assert e_native < 1.5 * e_bf16 + 2e-3


def test_dit_forward_ragged_tokens(cuda, hip_lib):
    """Token count not a multiple of any tile size (S = 20 + 5*3*5 = 95; 10x6 latent -> 5x3 patches)."""
    cfg = _small_cfg(sample_width=6, sample_height=10, sample_frames=17)
    out, ref, ref16 = _run_pair(cfg, 1, 5, 10, 6, cuda, fps=8)
    assert _rel(out, ref) < 1.5 * _rel(ref16, ref) + 2e-3


def test_dit_block_full_width(cuda, hip_lib):
    """One block at the real width (48 heads x 64, FF 12288, 226 text tokens) and a 3x60x90 latent
    (4 050 video tokens): every production tile shape, the ragged S tail, both row types."""
    from oracle.dit import DitConfig
    cfg = DitConfig(num_layers=1, sample_frames=9)
    out, ref, ref16 = _run_pair(cfg, 1, 3, 60, 90, cuda)
    e_native, e_bf16 = _rel(out, ref), _rel(ref16, ref)
    print(f"full-width block: native {e_native:.3e}  bf16-oracle {e_bf16:.3e}")
    assert e_native < 1.5 * e_bf16 + 2e-3


def test_dit_rejects_bad_inputs(cuda, hip_lib):
    from aether_amd.transformer import AetherTransformer3D
    with pytest.raises(ValueError):
        AetherTransformer3D({"attention_head_dim": 128}, device=cuda)
    m = AetherTransformer3D({"num_layers": 1, "num_attention_heads": 8, "text_embed_dim": 128, "time_embed_dim": 64,
                             "max_text_seq_length": 20}, device=cuda).init_random_weights(0)
    x = torch.zeros(1, 3, 95, 8, 12, dtype=torch.bfloat16, device=cuda)          # wrong channel count
    with pytest.raises(ValueError):
        m(x, torch.zeros(1, 20, 128, device=cuda), torch.zeros(1, device=cuda), image_rotary_emb=(torch.zeros(72, 64), torch.zeros(72, 64)))
