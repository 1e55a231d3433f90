"""Parity at the BASELINE geometry (41 frames x 480x720 -> latent 11x60x90, S = 226 + 14 850 tokens, real widths), reported the
way BASELINE.json's north_star states it: LATENTS L-infinity and PIXEL PSNR against the fp32 CPU oracle on identical
bf16-representable weights, inputs and seeds.  Stated thresholds:

  * DiT noise prediction (the latent-space output of one transformer forward), 2 of the 42 blocks at full width and the FULL
    sequence, B = 1 and B = 2:  rel-L2 <= 8e-3,  L-inf <= 2 % of max|ref|   (measured 4.0e-3 / 0.5 %);
  * VAE encode of a 480x720 clip (tiled 9 tiles -> 4/2/2/1 tile batches, 8-frame chunks threaded through the conv caches):
    posterior-mean latents  L-inf <= 4 % of max|ref|, rel-L2 <= 2e-2   (measured on the whole 17-frame output: 1.2e-2 / 1.4 %);
  * VAE decode of a 60x90 latent (tiled, frame-chunked, caches threaded):  pixel PSNR >= 38 dB (pixels in [0, 1]).

The oracle's CPU time bounds what can be compared (fp32, one pass): the DiT cases take ~10-30 s each on the GPU box's host
cores, the VAE cases ~1 min; the 42-block / 50-step trajectory at this size would take hours on CPU and is covered by the
scaled-down end-to-end tests (tests/test_pipeline_gpu.py) plus the finite/shape run (tests/test_fullsize_gpu.py).
"""
import math
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# VAE cases: the NATIVE path runs the whole 480x720 geometry (9 tiles as 4/2/2/1 batches, frame chunks with threaded caches); the
# fp32 CPU oracle — 0.5 TFLOP/s of conv3d on the box's host cores — evaluates only the tiles needed to pin a region of the
# output that covers an unblended tile interior, a horizontal blend seam between two tiles of the 4-batch, and a tile of another
# batch (the narrow right column).  The whole 17-frame encode output was compared once (129.6 s of CPU):
# profiles/r02_parity_full_geometry.log.
ENC_FRAMES, DEC_LATENT_FRAMES = 17, 5          # encoder chunks (0,9) (9,17); decoder chunks (0,3) (3,5): caches threaded


def _metrics(out: torch.Tensor, ref: torch.Tensor):
    d = (out.double() - ref.double())
    return {"rel_l2": (d.norm() / ref.double().norm()).item(), "linf": d.abs().max().item(), "ref_max": ref.abs().max().item(),
            "linf_rel": (d.abs().max() / ref.abs().max()).item()}


def _psnr(a: torch.Tensor, b: torch.Tensor) -> float:
    mse = ((a.double() - b.double()) ** 2).mean().item()
    return 10 * math.log10(1.0 / max(mse, 1e-20))


@pytest.mark.parametrize("B", [1, 2])
def test_dit_full_sequence_two_blocks(cuda, hip_lib, B):
    """S = 15 076 (the BASELINE token count), width 3072, 48 heads, FF 12 288, 226 text rows; 2 blocks."""
    from aether_amd.transformer import AetherTransformer3D
    from oracle.dit import DitConfig, OracleTransformer3D, init_random_
    from oracle.rope import prepare_rope
    cfg = DitConfig(num_layers=2)
    oracle = init_random_(OracleTransformer3D(cfg), seed=11)
    sd = {k: v.to(torch.bfloat16) for k, v in oracle.state_dict().items()}
    oracle.load_state_dict({k: v.float() for k, v in sd.items()})
    g = torch.Generator().manual_seed(B)
    hidden = torch.randn(B, 11, 96, 60, 90, generator=g).to(torch.bfloat16)
    text = (torch.randn(B, 226, 4096, generator=g) * 0.1).to(torch.bfloat16)
    t = torch.tensor([999, 259][:B], dtype=torch.int64)
    rope = prepare_rope(480, 720, 11, 12)
    t0 = time.perf_counter()
    with torch.no_grad():
        ref = oracle(hidden.float(), text.float(), t, image_rotary_emb=rope)[0]
    t_cpu = time.perf_counter() - t0
    native = AetherTransformer3D({k: getattr(cfg, k) for k in cfg.__dataclass_fields__}, device=cuda).load_state_dict(sd)
    out = native(hidden_states=hidden.to(cuda), encoder_hidden_states=text.to(cuda), timestep=t.to(cuda), ofs=None,
                 image_rotary_emb=(rope[0].to(cuda), rope[1].to(cuda)), return_dict=False)[0]
    torch.cuda.synchronize()
    assert out.shape == (B, 11, 56, 60, 90) and torch.isfinite(out.float()).all()
    m = _metrics(out.cpu().float(), ref)
    print(f"\nDiT S=15076 B={B} 2 blocks vs fp32 oracle ({t_cpu:.1f} s CPU): rel-L2 {m['rel_l2']:.3e}  latents L-inf {m['linf']:.4f} "
          f"({100 * m['linf_rel']:.2f} % of max|ref| {m['ref_max']:.3f})")
    assert m["rel_l2"] <= 8.0e-3 and m["linf_rel"] <= 0.02, m


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


@pytest.fixture(scope="module")
def vae_pair(cuda, hip_lib):
    from aether_amd.vae import AetherVAE
    from oracle.vae import OracleVAE, VaeConfig, init_random_
    oracle = init_random_(OracleVAE(VaeConfig()), seed=2)
    sd = {k: v.to(torch.bfloat16) for k, v in oracle.state_dict().items()}
    oracle.load_state_dict({k: v.float() for k, v in sd.items()})
    oracle.enable_tiling(); oracle.enable_slicing()
    native = AetherVAE(device=cuda).load_state_dict(sd)
    native.enable_tiling(); native.enable_slicing()
    return oracle, native


def _oracle_row0(oracle, net, x, bs, tile_h, tile_w, stride_w, blend_w, limit_h, limit_w):
    """Row 0 of the reference's tiled pass (no vertical blend there): every tile of the first tile row through `net`, blended
    horizontally and cropped exactly as AutoencoderKLCogVideoX.tiled_encode / tiled_decode do."""
    W = x.shape[-1]
    row = [oracle._run_chunks(net, x[:, :, :, :tile_h, j:j + tile_w].contiguous(), bs) for j in range(0, W, stride_w)]
    out = []
    for j, tile in enumerate(row):
        if j > 0:
            tile = oracle._blend_h(row[j - 1], tile, blend_w)
        out.append(tile[:, :, :, :limit_h, :limit_w])
    return torch.cat(out, dim=4)


def test_vae_encode_full_clip(cuda, vae_pair):
    """17 x 480 x 720, tiling on (9 overlapping 240x360 tiles as 4/2/2/1 batches natively), 8-frame chunks with the causal-conv
    caches threaded.  Compared: latent rows 0..24 x all 90 columns (tile row 0: unblended interior of tile (0,0), the seams
    (0,0)|(0,1) and (0,1)|(0,2), the narrow tile (0,2) of the 2-batch)."""
    oracle, native = vae_pair
    x = _video(ENC_FRAMES, 480, 720).to(torch.bfloat16)
    t0 = time.perf_counter()
    with torch.no_grad():
        ref_rows = _oracle_row0(oracle, oracle.encoder, x.float(), 8, 240, 360, 288, 9, 25, 36)     # [1, 32, 5, 25, 90]
    t_cpu = time.perf_counter() - t0
    got = native.encode(x.to(cuda)).latent_dist
    torch.cuda.synchronize()
    mean = got.mode().cpu().float()
    assert mean.shape == (1, 16, (ENC_FRAMES - 1) // 4 + 1, 60, 90) and ref_rows.shape[-2:] == (25, 90)
    m = _metrics(mean[:, :, :, :25], ref_rows[:, :16])
    print(f"\nVAE encode {ENC_FRAMES}x480x720 tiled vs fp32 oracle (tile row 0, {t_cpu:.1f} s CPU): posterior mean rel-L2 {m['rel_l2']:.3e}  "
          f"latents L-inf {m['linf']:.4f} ({100 * m['linf_rel']:.2f} % of max|ref| {m['ref_max']:.3f})")
    assert m["rel_l2"] <= 2.0e-2 and m["linf_rel"] <= 0.04, m
    logvar = got.logvar.cpu().float()
    ml = _metrics(logvar[:, :, :, :25], ref_rows[:, 16:].clamp(-30.0, 20.0))
    assert ml["linf_rel"] <= 0.04, ml


def test_vae_decode_full_resolution(cuda, vae_pair):
    """5 x 60 x 90 latent -> 17 x 480 x 720 pixels, tiled (9 latent tiles 30x45, strides 25x36) and frame-chunked ((0,3) (3,5),
    caches threaded).  Compared (the decoder oracle costs ~55 s of CPU per full tile): pixel rows 0..199 of tile (0,0) (columns
    0..287, member of the 4-batch) and of the narrow tile (0,2) right of its blend seam (columns 648..719, the 2-batch).  The seams
    themselves are compared at full scale on the encoder side and bit for bit against the Python walk (tests/test_vae_gpu.py)."""
    oracle, native = vae_pair
    g = torch.Generator().manual_seed(4)
    z = (torch.randn(1, 16, DEC_LATENT_FRAMES, 60, 90, generator=g) * 0.8).to(torch.bfloat16)
    t0 = time.perf_counter()
    with torch.no_grad():
        zf = z.float()
        ref_a = oracle._run_chunks(oracle.decoder, zf[:, :, :, :30, 0:45].contiguous(), 2)[:, :, :, :200, :288]
        ref_c = oracle._run_chunks(oracle.decoder, zf[:, :, :, :30, 72:90].contiguous(), 2)[:, :, :, :200, 72:144]
    t_cpu = time.perf_counter() - t0
    out = native.decode(z.to(cuda)).sample
    torch.cuda.synchronize()
    out = out.cpu().float()
    assert out.shape == (1, 3, 4 * (DEC_LATENT_FRAMES - 1) + 1, 480, 720)
    got = torch.cat([out[:, :, :, :200, :288], out[:, :, :, :200, 648:720]], dim=4)
    ref = torch.cat([ref_a, ref_c], dim=4)
    # pixels as the pipeline post-processes them (P:932: x/2 + 0.5 clamped to [0, 1])
    pix_n, pix_o = (got / 2 + 0.5).clamp(0, 1), (ref / 2 + 0.5).clamp(0, 1)
    m = _metrics(got, ref)
    psnr = _psnr(pix_n, pix_o)
    print(f"\nVAE decode {DEC_LATENT_FRAMES}x60x90 -> {out.shape[2]}x480x720 tiled vs fp32 oracle (tiles (0,0) and (0,2), {t_cpu:.1f} s CPU): rel-L2 {m['rel_l2']:.3e}  "
          f"L-inf {m['linf']:.4f}  pixel PSNR {psnr:.1f} dB")
    assert psnr >= 38.0 and m["rel_l2"] <= 2e-2, (psnr, m)
