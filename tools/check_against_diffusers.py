#!/usr/bin/env python
"""The TRUE parity pin, for the day `diffusers` is importable (it is not in the build container).

Builds depth/width-reduced diffusers modules (CogVideoXTransformer3DModel, AutoencoderKLCogVideoX, CogVideoXDPMScheduler),
copies their randomly initialised state dicts into the oracle restatements of this repo (oracle/dit.py, oracle/vae.py;
same key names) and into aether_amd's scheduler, and compares outputs on seeded inputs on the CPU in fp32.
Every "[UPSTREAM-UNVERIFIED]" item of SURVEY.md Appendix A (RoPE pair layout, AdaLN chunk order, tile-blend extents,
first-frame rules of the resamplers, DPM noise-draw count ...) is exercised by one of the three checks.

    python tools/check_against_diffusers.py        # exit 0 = all restatements match diffusers, 3 = diffusers missing
    python tools/check_against_diffusers.py --self-test
        runs the SAME script body against a stand-in `diffusers` module whose three classes are thin adapters around this repo's own
        restatements (constructor keywords, call signatures and attribute names as the script uses them on the real package): proves
        nothing about diffusers, but every line of the check executes, so a typo does not surface on the day diffusers is available
        (tests/test_host_cpu.py runs it).
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _install_standin():
    """A module named `diffusers` exposing the three classes this script imports, backed by the oracle restatements."""
    import types

    from aether_amd.scheduler import CogVideoXDPMScheduler as OurScheduler
    from oracle.dit import DitConfig, OracleTransformer3D
    from oracle.vae import OracleVAE, VaeConfig

    class CogVideoXTransformer3DModel(OracleTransformer3D):
        def __init__(self, **kw):
            super().__init__(DitConfig(**kw))

    class AutoencoderKLCogVideoX(OracleVAE):
        def __init__(self, down_block_types=None, up_block_types=None, **kw):
            assert len(down_block_types) == len(up_block_types) == len(kw["block_out_channels"])
            super().__init__(VaeConfig(**kw))

    mod = types.ModuleType("diffusers")
    mod.CogVideoXTransformer3DModel, mod.AutoencoderKLCogVideoX, mod.CogVideoXDPMScheduler = CogVideoXTransformer3DModel, AutoencoderKLCogVideoX, OurScheduler
    mod.__version__ = "stand-in (oracle restatements)"
    sys.modules["diffusers"] = mod


def main(self_test: bool = False):
    if self_test:
        _install_standin()
    try:
        import diffusers  # noqa: F401
        from diffusers import AutoencoderKLCogVideoX, CogVideoXDPMScheduler, CogVideoXTransformer3DModel
    except Exception as e:  # pragma: no cover
        print(f"diffusers is not importable here ({e}); parity stays UNPINNED (oracle/__init__.py).")
        return 3
    from aether_amd.rope import resize_crop_region_for_grid, rotary_tables_3d
    from aether_amd.scheduler import CogVideoXDPMScheduler as OurScheduler
    from oracle.dit import DitConfig, OracleTransformer3D
    from oracle.vae import OracleVAE, VaeConfig

    torch.manual_seed(0)
    ok = True

    # ---- transformer ---------------------------------------------------------------------------------------
    kw = dict(num_attention_heads=4, attention_head_dim=64, in_channels=96, out_channels=56, num_layers=2, text_embed_dim=64,
              time_embed_dim=32, max_text_seq_length=8, sample_width=12, sample_height=8, sample_frames=9,
              use_rotary_positional_embeddings=True, use_learned_positional_embeddings=False)
    ref = CogVideoXTransformer3DModel(**kw).eval()
    for p in ref.parameters():
        torch.nn.init.normal_(p, std=0.05)
    mine = OracleTransformer3D(DitConfig(**kw))
    missing = mine.load_state_dict(ref.state_dict(), strict=False)
    print("transformer key diff:", missing)
    x = torch.randn(2, 3, 96, 8, 12)
    txt = torch.randn(2, 8, 64)
    t = torch.tensor([999, 499])
    rope = rotary_tables_3d(64, resize_crop_region_for_grid((4, 6), 6, 4), (4, 6), 3)
    with torch.no_grad():
        a = ref(hidden_states=x, encoder_hidden_states=txt, timestep=t, image_rotary_emb=rope, return_dict=False)[0]
        b = mine(x, txt, t, image_rotary_emb=rope)[0]
    err = (a - b).abs().max().item()
    print(f"transformer max|diff| = {err:.3e}")
    ok &= err < 1e-4

    # ---- VAE (tiled) -------------------------------------------------------------------------------------------
    vkw = dict(block_out_channels=(32, 64, 64, 64), layers_per_block=1, sample_height=96, sample_width=240)
    vref = AutoencoderKLCogVideoX(down_block_types=("CogVideoXDownBlock3D",) * 4, up_block_types=("CogVideoXUpBlock3D",) * 4, **vkw).eval()
    vmine = OracleVAE(VaeConfig(**vkw))
    print("vae key diff:", vmine.load_state_dict(vref.state_dict(), strict=False))
    vref.enable_tiling()
    vmine.enable_tiling()
    vid = torch.randn(1, 3, 17, 96, 240).clamp(-1, 1)
    z = torch.randn(1, 16, 5, 12, 30)
    with torch.no_grad():
        e1 = vref.encode(vid).latent_dist.parameters
        e2 = vmine.encode(vid).latent_dist.parameters
        d1 = vref.decode(z).sample
        d2 = vmine.decode(z).sample
    print(f"vae encode max|diff| = {(e1 - e2).abs().max().item():.3e}, decode max|diff| = {(d1 - d2).abs().max().item():.3e}")
    ok &= (e1 - e2).abs().max().item() < 1e-4 and (d1 - d2).abs().max().item() < 1e-4

    # ---- scheduler (values AND random-draw order) ----------------------------------------------------------------
    cfg = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", prediction_type="v_prediction",
               timestep_spacing="trailing", rescale_betas_zero_snr=True, snr_shift_scale=1.0, set_alpha_to_one=True, clip_sample=False)
    s1, s2 = CogVideoXDPMScheduler(**cfg), OurScheduler(**cfg)
    for n in (4, 50):
        s1.set_timesteps(n)
        s2.set_timesteps(n)
        ok &= s1.timesteps.tolist() == s2.timesteps.tolist()
        g1, g2 = torch.Generator().manual_seed(1), torch.Generator().manual_seed(1)
        x1 = x2 = torch.randn(1, 3, 8, 4, 6)
        o1 = o2 = None
        for i, tt in enumerate(s1.timesteps):
            v = torch.randn(1, 3, 8, 4, 6, generator=torch.Generator().manual_seed(100 + i))
            tb = s1.timesteps[i - 1] if i > 0 else None
            x1, o1 = s1.step(v, o1, tt, tb, x1, generator=g1, return_dict=False)
            x2, o2 = s2.step(v, o2, tt, tb, x2, generator=g2, return_dict=False)
        err = (x1 - x2).abs().max().item()
        same_rng = torch.equal(torch.randn(3, generator=g1), torch.randn(3, generator=g2))
        print(f"scheduler n={n}: max|diff| = {err:.3e}, generators in lock-step: {same_rng}")
        ok &= err < 1e-5 and same_rng
    print(("SELF-TEST PLUMBING OK (proves nothing about diffusers)" if self_test else "ALL MATCH") if ok else
          "MISMATCH — see SURVEY.md Appendix A for the item to revisit")
    if self_test:
        sys.modules.pop("diffusers", None)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main(self_test="--self-test" in sys.argv))
