"""FULL-DEPTH, FULL-GEOMETRY parity on MI355X against fixtures the fp32 CPU oracle wrote offline (tools/make_fullsize_golden.py,
~1 h of CPU in the build container; tests/golden/fullsize_{dit,clip}.npz):

  * ONE transformer forward, all 42 blocks, S = 15 076 tokens, B = 1 (the call of P:865-875);
  * the tiled 41-frame VAE encode (9 tiles, 5 frame chunks) and the tiled 11 -> 41-frame decode (9 tiles, 5 chunks), whole output;
  * the whole reconstruction call (P:690-965) with the reference's default 4 steps, CPU generator seed 42: final latents
    (latents L-inf / rel-L2) and the decoded rgb / disparity (pixel PSNR).

Weights and inputs are regenerated here from the seeds of tools/fullsize_cases.py (CPU generators, bf16-representable), so both
sides saw identical bits.  The oracle is this repo's restatement of diffusers (PARITY UNPINNED against diffusers itself);
what these tests bound is the drift of the bf16 HIP path from an fp32 evaluation of the same arithmetic at the real depth and size.
Thresholds: the DiT must stay within the distance of the bf16 ORACLE itself from the same fixture (1.50e-2 / 2.24 %,
profiles/r03_bf16_oracle_calibration.json); the VAE / clip bounds are ~1.3x the values measured on MI355X (profiles/r03_parity_fullsize.log).
"""
import gc
import json
import os
import sys
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import fullsize_cases as fc  # noqa: E402


def _load(name):
    path = os.path.join(fc.GOLDEN_DIR, name)
    assert os.path.exists(path), f"{path} missing: run tools/make_fullsize_golden.py"
    z = np.load(path)
    return z, json.loads(str(z["meta"]))


@pytest.fixture(scope="module")
def native_dit(fullsize_modules):
    return fullsize_modules[0]


@pytest.fixture(scope="module")
def native_vae(fullsize_modules):
    return fullsize_modules[1]


def test_dit_42_blocks_full_sequence(cuda, native_dit):
    z, meta = _load("fullsize_dit.npz")
    hidden, text, t = fc.dit_inputs()
    assert abs(float(hidden.float().sum()) - meta["input_sum"]) < 1e-3 * max(1.0, abs(meta["input_sum"])), "seeded inputs differ from the fixture's"
    rope = fc.rope_tables()
    out = native_dit(hidden_states=hidden.to(cuda), encoder_hidden_states=text.to(cuda), timestep=t.to(cuda), ofs=None,
                     image_rotary_emb=(rope[0].to(cuda), rope[1].to(cuda)), return_dict=False)[0]
    torch.cuda.synchronize()
    ref = torch.from_numpy(z["out"].astype(np.float32))
    assert out.shape == ref.shape and torch.isfinite(out.float()).all()
    m = fc.metrics(out.cpu().float(), ref)
    print(f"\n[fullsize] DiT {meta['layers']} blocks, S = {fc.TEXT_LEN + fc.LAT_F * fc.LAT_H * fc.LAT_W // 4}, B = 1 vs fp32 oracle "
          f"({meta['seconds_cpu']:.0f} s of CPU offline): rel-L2 {m['rel_l2']:.3e}  latents L-inf {m['linf']:.4f} "
          f"({100 * m['linf_rel']:.2f} % of max|ref| {m['ref_max']:.3f})")
    # measured 1.08e-2 / 1.73 %; the bound is the bf16 oracle's own distance to this fixture (1.50e-2 / 2.24 %): no worse than the reference dtype
    assert m["rel_l2"] <= 1.50e-2 and m["linf_rel"] <= 0.023, m


def test_vae_encode_whole_clip(cuda, native_vae):
    z, _ = _load("fullsize_clip.npz")
    x = fc.video_as_model_input(fc.clip_video()).to(torch.bfloat16).permute(1, 0, 2, 3)[None]          # [1,3,F,H,W]
    dist = native_vae.encode(x.to(cuda)).latent_dist
    torch.cuda.synchronize()
    mean, logvar = dist.mode().cpu().float(), dist.logvar.cpu().float()
    ref_mean = torch.from_numpy(z["posterior_mean"].astype(np.float32))
    ref_logvar = torch.from_numpy(z["posterior_logvar_s2"].astype(np.float32))
    assert mean.shape == ref_mean.shape
    m, ml = fc.metrics(mean, ref_mean), fc.metrics(logvar[..., ::2, ::2], ref_logvar)
    print(f"\n[fullsize] VAE encode {fc.FRAMES}x{fc.HEIGHT}x{fc.WIDTH}, all tiles and chunks, vs fp32 oracle: posterior mean rel-L2 {m['rel_l2']:.3e}  "
          f"latents L-inf {m['linf']:.4f} ({100 * m['linf_rel']:.2f} % of max|ref| {m['ref_max']:.3f}); log-variance L-inf {100 * ml['linf_rel']:.2f} %")
    assert m["rel_l2"] <= 1.5e-2 and m["linf_rel"] <= 0.024, m       # measured 1.13e-2 / 1.80 %; log-variance 1.35 %
    assert ml["linf_rel"] <= 0.018, ml


def test_vae_decode_whole_clip(cuda, native_vae):
    """Decode of the ORACLE's final rgb latents (exact bf16 bits from the fixture), exactly as `decode_latents` calls it (P:931)."""
    z, _ = _load("fullsize_clip.npz")
    lat = fc.from_bf16_bits(z["final_latents_bits"])[:, :, :16]                                      # [1,11,16,60,90]
    zin = (1 / 0.7 * lat.to(cuda).permute(0, 2, 1, 3, 4))
    out = native_vae.decode(zin).sample
    torch.cuda.synchronize()
    s = fc.DEC_STRIDE
    got = out.cpu().float()[:, :, :, ::s, ::s]
    ref = torch.from_numpy(z["rgb_decoded_s8"].astype(np.float32))
    assert got.shape == ref.shape and out.shape[2:] == (fc.FRAMES, fc.HEIGHT, fc.WIDTH)
    m = fc.metrics(got, ref)
    p = fc.psnr((got / 2 + 0.5).clamp(0, 1), (ref / 2 + 0.5).clamp(0, 1))
    print(f"\n[fullsize] VAE decode {fc.LAT_F}x{fc.LAT_H}x{fc.LAT_W} -> {fc.FRAMES}x{fc.HEIGHT}x{fc.WIDTH}, all tiles and chunks (every {s}th row/column compared): "
          f"rel-L2 {m['rel_l2']:.3e}  L-inf {m['linf']:.4f}  pixel PSNR {p:.1f} dB")
    assert p >= 47.2 and m["rel_l2"] <= 8.1e-3, (p, m)              # measured 49.5 dB / 6.2e-3 (1.3x the error = -2.3 dB)


def test_reconstruction_clip_four_steps(cuda, native_dit, native_vae):
    """The whole `__call__` (P:690-965): encode -> sample -> 4 x (cat, 42-block forward, DPM step) -> two decodes; CPU generator."""
    from aether.pipelines.aetherv1_pipeline_cogvideox import AetherV1PipelineCogVideoX
    from aether_amd.scheduler import CogVideoXDPMScheduler
    z, meta = _load("fullsize_clip.npz")
    pipe = AetherV1PipelineCogVideoX(tokenizer=None, text_encoder=None, vae=native_vae, scheduler=CogVideoXDPMScheduler(),
                                     transformer=native_dit, empty_prompt_embeds=fc.prompt_embeds())
    pipe.set_progress_bar_config(disable=True)
    video = fc.clip_video()
    assert abs(float(video.astype(np.float64).sum()) - meta["video_sum"]) < 1e-6 * meta["video_sum"]
    t0 = time.perf_counter()
    out = pipe(task="reconstruction", video=video, height=fc.HEIGHT, width=fc.WIDTH, num_frames=fc.FRAMES, fps=12,
               num_inference_steps=fc.CLIP_STEPS, generator=torch.Generator().manual_seed(fc.CLIP_SEED))
    dt = time.perf_counter() - t0
    lat = pipe._final_latents.cpu().float()
    ref_lat = fc.from_bf16_bits(z["final_latents_bits"]).float()
    ml = fc.metrics(lat, ref_lat)
    s = fc.DEC_STRIDE
    rgb, disp = torch.from_numpy(out.rgb)[:, ::s, ::s], torch.from_numpy(out.disparity)[:, ::s, ::s]
    ref_rgb, ref_disp = torch.from_numpy(z["rgb_s8"].astype(np.float32)), torch.from_numpy(z["disparity_s8"].astype(np.float32))
    p_rgb, m_disp = fc.psnr(rgb, ref_rgb), fc.metrics(disp, ref_disp)
    from einops import rearrange                  # the raymap output is the camera channels of the final latents, un-folded (P:942-945)
    ref_ray = rearrange(ref_lat[:, :, 32:], "b t (n c) h w -> b (n t) c h w", n=4)[0, -fc.FRAMES:]
    m_ray = fc.metrics(torch.from_numpy(out.raymap), ref_ray)
    print(f"\n[fullsize] reconstruction, {fc.CLIP_STEPS} steps, {fc.FRAMES}x{fc.HEIGHT}x{fc.WIDTH} ({dt:.1f} s here, {meta['seconds_cpu_total']:.0f} s of CPU offline): "
          f"final latents rel-L2 {ml['rel_l2']:.3e}  L-inf {ml['linf']:.4f} ({100 * ml['linf_rel']:.2f} % of max|ref| {ml['ref_max']:.2f}); "
          f"rgb PSNR {p_rgb:.1f} dB; disparity rel-L2 {m_disp['rel_l2']:.3e}; raymap rel-L2 {m_ray['rel_l2']:.3e}")
    assert out.rgb.shape == (fc.FRAMES, fc.HEIGHT, fc.WIDTH, 3) and np.isfinite(out.rgb).all()
    # measured: latents 1.03e-2 / 1.53 %, rgb 39.1 dB, disparity 2.2e-2 (a squared quantity: twice the relative error), raymap 9.6e-3
    assert ml["rel_l2"] <= 1.35e-2 and ml["linf_rel"] <= 0.02, ml
    assert p_rgb >= 36.8 and m_disp["rel_l2"] <= 2.9e-2 and m_ray["rel_l2"] <= 1.25e-2, (p_rgb, m_disp, m_ray)
