"""CPU checks of the host-side pieces against fixtures produced by the REFERENCE itself (tools/make_golden.py imports
/root/reference in the build container; the .npz files are committed) and against closed-form known answers."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_center_crop_matches_reference_imcrop_center():
    from aether_amd.preprocess import center_crop_frames
    z = np.load(os.path.join(GOLD, "imcrop_center.npz"))
    n = len([k for k in z.files if k.startswith("in_")])
    assert n >= 7
    for i in range(n):
        th, tw = z[f"tgt_{i}"]
        got = np.stack(center_crop_frames(list(z[f"in_{i}"].astype(np.float32)), int(th), int(tw)))
        ref = z[f"out_{i}"].astype(np.float32)
        assert got.shape == ref.shape, i
        assert np.array_equal(got, ref), i          # pure indexing: bit exact


def test_rope_tables_known_answers():
    """Positions are plain integers at fps 12 (t 0..10, h 0..29, w 0..44), channel split 16|24|24 (P:103-105)."""
    from aether_amd.rope import resize_crop_region_for_grid, rotary_tables_3d
    crop = resize_crop_region_for_grid((30, 45), 45, 30)
    assert crop == ((0, 0), (30, 45))
    cos, sin = rotary_tables_3d(64, crop, (30, 45), 11, 1.0)
    assert cos.shape == sin.shape == (14850, 64) and cos.dtype == torch.float32
    tok = lambda t, h, w: (t * 30 + h) * 45 + w  # noqa: E731
    i = tok(7, 13, 29)
    f_t = [10000.0 ** (-2 * k / 16) for k in range(8)]
    f_hw = [10000.0 ** (-2 * k / 24) for k in range(12)]
    exp_ang = np.repeat(np.array([7 * f for f in f_t] + [13 * f for f in f_hw] + [29 * f for f in f_hw]), 2)
    assert np.allclose(cos[i].numpy(), np.cos(exp_ang), atol=2e-6)
    assert np.allclose(sin[i].numpy(), np.sin(exp_ang), atol=2e-6)
    # Aether's fps_factor = 12 / fps scales ONLY the temporal positions (P:81-90)
    cos8, _ = rotary_tables_3d(64, crop, (30, 45), 11, 12 / 8)
    assert np.allclose(cos8[i, :16].numpy(), np.cos(np.repeat(np.array([7 * 1.5 * f for f in f_t]), 2)), atol=2e-6)
    assert torch.equal(cos8[:, 16:], cos[:, 16:])


def test_rope_matches_oracle_restatement():
    from aether_amd.rope import resize_crop_region_for_grid, rotary_tables_3d
    from oracle.rope import prepare_rope
    for fps, (H, W, F) in [(12, (480, 720, 11)), (24, (480, 720, 5)), (10, (240, 368, 9))]:
        c0, s0 = prepare_rope(H, W, F, fps)
        crop = resize_crop_region_for_grid((H // 16, W // 16), 45, 30)
        c1, s1 = rotary_tables_3d(64, crop, (H // 16, W // 16), F, 12 / fps)
        assert torch.equal(c0, c1) and torch.equal(s0, s1)


def test_scheduler_timesteps_and_schedule():
    from aether_amd.scheduler import CogVideoXDPMScheduler
    s = CogVideoXDPMScheduler()
    s.set_timesteps(50)
    assert s.timesteps.tolist() == list(range(999, 0, -20))                 # 999, 979, ..., 19 (trailing)
    s.set_timesteps(4)
    assert s.timesteps.tolist() == [999, 749, 499, 249]
    ac = s.alphas_cumprod
    assert ac.dtype == torch.float64 and ac[-1].item() == 0.0                # zero terminal SNR
    assert abs(ac[0].item() - (1 - 0.00085)) < 1e-12                         # first value preserved by the rescale
    assert (ac[1:] < ac[:-1]).all()


def _np_dpm_step(ac, n, v, old_x0, t, t_back, x, noise1, noise2):
    """Independent float64 numpy restatement of the SDE DPM-Solver++(2M) update."""
    prev_t = t - 1000 // n
    a_t, a_prev = ac[t], (ac[prev_t] if prev_t >= 0 else np.float64(1.0))
    x0 = np.sqrt(a_t) * x - np.sqrt(1 - a_t) * v
    with np.errstate(divide="ignore", invalid="ignore"):
        lam = lambda a: np.log(np.sqrt(a / (1 - a)))  # noqa: E731
        h = lam(a_prev) - lam(a_t)
        m1 = np.sqrt((1 - a_prev) / (1 - a_t)) * np.exp(-h)
        m2 = np.expm1(-2 * h) * np.sqrt(a_prev)
        mn = np.sqrt(1 - a_prev) * np.sqrt(1 - np.exp(-2 * h))
        if old_x0 is None or prev_t < 0:
            return m1 * x - m2 * x0 + mn * noise1, x0
        r = (lam(a_t) - lam(ac[t_back])) / h
        d = (1 + 1 / (2 * r)) * x0 - (1 / (2 * r)) * old_x0
    return m1 * x - m2 * d + mn * noise2, x0


@pytest.mark.parametrize("n", [4, 50])
def test_scheduler_step_matches_numpy_and_rng_order(n):
    from aether_amd.scheduler import CogVideoXDPMScheduler
    s = CogVideoXDPMScheduler()
    s.set_timesteps(n)
    ts = s.timesteps.tolist()
    ac = s.alphas_cumprod.numpy()
    shape = (1, 3, 8, 4, 6)
    g = torch.Generator().manual_seed(7)
    g_ref = torch.Generator().manual_seed(7)
    x = torch.randn(shape, dtype=torch.float64)
    x_np, old, old_np = x.numpy().copy(), None, None
    draws = 0
    for i, t in enumerate(ts):
        v = torch.randn(shape, generator=torch.Generator().manual_seed(100 + i), dtype=torch.float64)
        x, old = s.step(v, old, t, ts[i - 1] if i > 0 else None, x, generator=g)
        n1 = torch.randn(shape, generator=g_ref, dtype=torch.float64).numpy()
        two = (old_np is not None) and (t - 1000 // n >= 0)
        n2 = torch.randn(shape, generator=g_ref, dtype=torch.float64).numpy() if two else None
        draws += 2 if two else 1
        x_np, old_np = _np_dpm_step(ac, n, v.numpy(), old_np, t, ts[i - 1] if i > 0 else None, x_np, n1, n2)
        assert np.allclose(x.numpy(), x_np, rtol=1e-9, atol=1e-9), i
    assert draws == 1 + 2 * (n - 2) + 1                     # 1 draw on the first and last step, 2 in between
    # generators in lock-step <=> same number of draws consumed
    assert torch.equal(torch.randn(4, generator=g), torch.randn(4, generator=g_ref))
    # last step lands exactly on the predicted clean sample (alpha_prev = 1, no noise)
    assert np.allclose(x.numpy(), old.numpy())


def test_scheduler_dtype_promotion_like_reference():
    """bf16 latents + fp32 model output -> fp32 prev_sample, noise drawn in the sample's dtype (P:876, P:916)."""
    from aether_amd.scheduler import CogVideoXDPMScheduler
    s = CogVideoXDPMScheduler()
    s.set_timesteps(4)
    x = torch.randn(1, 2, 4, 4, 4).to(torch.bfloat16)
    v = torch.randn(1, 2, 4, 4, 4)
    prev, x0 = s.step(v, None, 999, None, x, generator=torch.Generator().manual_seed(0))
    assert prev.dtype == torch.float32 and x0.dtype == torch.float32
    assert torch.isfinite(prev).all()


def test_video_processor_roundtrip_and_pil_crop():
    import PIL.Image
    from aether_amd.video_processor import VideoProcessor
    vp = VideoProcessor()
    frames = np.random.default_rng(0).random((5, 16, 24, 3), dtype=np.float32)
    t = vp.preprocess(list(frames), 16, 24)
    assert t.shape == (5, 3, 16, 24) and float(t.min()) >= -1 and float(t.max()) <= 1
    back = vp.postprocess_video(t.permute(1, 0, 2, 3)[None], output_type="np")
    assert back.shape == (1, 5, 16, 24, 3) and np.allclose(back[0], frames, atol=1e-6)
    img = PIL.Image.fromarray((np.random.default_rng(1).random((50, 50, 3)) * 255).astype(np.uint8))
    out = vp.preprocess(img, 16, 24, resize_mode="crop")
    assert out.shape == (1, 3, 16, 24)


def test_raymap_packing_is_outer_factor():
    """P:666-670 / P:942-945: latent frame t packs raw frames {t, T+t, 2T+t, 3T+t}; the output un-fold is its inverse."""
    from einops import rearrange
    r = torch.arange(44 * 6).reshape(1, 44, 6, 1, 1).float()
    packed = rearrange(r, "b (n t) c h w -> b t (n c) h w", n=4)
    assert packed.shape == (1, 11, 24, 1, 1)
    assert packed[0, 3, :6, 0, 0].tolist() == r[0, 3, :, 0, 0].tolist()
    assert packed[0, 3, 6:12, 0, 0].tolist() == r[0, 14, :, 0, 0].tolist()
    assert torch.equal(rearrange(packed, "b t (n c) h w -> b (n t) c h w", n=4), r)


def test_vae_state_dict_spec_matches_oracle_keys():
    """The native VAE's notion of the diffusers key layout equals the oracle module's state dict."""
    from aether_amd.vae import AetherVAE
    from oracle.vae import OracleVAE, VaeConfig
    for kw in (dict(), dict(block_out_channels=(64, 128, 128, 128), layers_per_block=1)):
        spec = AetherVAE(kw, device="cpu").state_dict_spec()
        ref = {k: tuple(v.shape) for k, v in OracleVAE(VaeConfig(**kw)).state_dict().items()}
        assert spec == ref


def test_source_digest_ignores_comments_and_white_space():
    """Profiles are stamped with a digest of the kernel CODE (aether_amd/build.py): a comment or re-indentation must not make a
    measured profile look stale, a code change must."""
    from aether_amd.build import _code_only, source_digest
    a = "int f(int x) {\n    return x + 1;   // add one\n}\n"
    b = "/* header */ int f(int x) { return x + 1; }"
    c = "int f(int x) { return x + 2; }"
    assert _code_only(a) == _code_only(b) != _code_only(c)
    d = source_digest()
    assert len(d) == 16 and d == source_digest()


def test_check_against_diffusers_self_test(capsys):
    """tools/check_against_diffusers.py is the true parity pin and has never met the real package (diffusers is not installable in the build
    container): its whole body runs here against a stand-in module backed by the oracle restatements, so constructor keywords, call
    signatures, state-dict hand-over and the scheduler loop are known to execute."""
    import importlib
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    mod = importlib.import_module("check_against_diffusers")
    assert mod.main(self_test=True) == 0
    out = capsys.readouterr().out
    assert "transformer max|diff| = 0.000e+00" in out and "SELF-TEST PLUMBING OK" in out and "generators in lock-step: True" in out
    assert "diffusers" not in sys.modules or getattr(sys.modules["diffusers"], "__version__", "").startswith("stand-in") is False


@pytest.mark.parametrize("nb,steps", [(1, 4), (2, 50)])
def test_fused_step_arithmetic_restated_on_the_cpu(nb, steps):
    """The op sequence aether_dpm_step (csrc/sampler.hip) implements, restated op by op in torch on the CPU from the scalars of
    `CogVideoXDPMScheduler._coefficients`, against `scheduler.step` fed the way the pipeline feeds it (P:877-916) over a whole schedule: bit-identical
    latents and x0 at every step — which pins the decomposition (coefficients, operation order, which draw is used, where bf16 roundings fall).
    One rounding differs between PyTorch's two back ends and is restated here the CPU way: a float64 scalar times a bf16 TENSOR rounds the
    scalar to bf16 first on the CPU, to fp32 on CUDA (measured: 14 % of the products differ in the last bit).  The HIP kernel follows CUDA —
    the back end the reference runs on — and is compared with the CUDA sequence on the GPU (tests/test_kernels_gpu.py, bit-identical)."""
    import torch
    from aether_amd.scheduler import CogVideoXDPMScheduler
    f32 = lambda v: torch.as_tensor(v, dtype=torch.float64).to(torch.float32)  # noqa: E731  scalar operand as PyTorch rounds it
    bf = lambda x: x.to(torch.bfloat16).to(torch.float32)  # noqa: E731          one bf16 rounding
    sbf = lambda v: bf(f32(v))  # noqa: E731                                      scalar operand of a bf16-tensor op on the CPU back end
    shape = (1, 2, 56, 6, 8)
    g = torch.Generator().manual_seed(5)
    lat0 = torch.randn(shape, generator=g).to(torch.bfloat16)
    preds = [(torch.randn((nb,) + shape[1:], generator=g) * 1.3).to(torch.bfloat16) for _ in range(steps)]
    scale = 3.7
    sched = CogVideoXDPMScheduler()
    sched.set_timesteps(steps)
    ts = sched.timesteps.tolist()
    ga, gb = torch.Generator().manual_seed(11), torch.Generator().manual_seed(11)
    lat_a, old_a = lat0.clone(), None
    lat_b, old_b = lat0.clone(), None
    for i, t in enumerate(ts):
        tb = ts[i - 1] if i > 0 else None
        # (a) the reference sequence
        npred = preds[i].float()
        if nb == 2:
            u, c = npred.chunk(2)
            npred = u + scale * (c - u)
        lat_a, old_a = sched.step(npred, old_a, t, tb, lat_a, generator=ga, return_dict=False)
        lat_a = lat_a.to(torch.bfloat16)
        # (b) what the kernel computes
        k = sched._coefficients(t, tb, old_b is not None)
        noise = torch.randn(shape, generator=gb, dtype=torch.bfloat16)
        if k["second"]:
            noise = torch.randn(shape, generator=gb, dtype=torch.bfloat16)
        s = lat_b.float()
        mo = preds[i][:1].float()
        if nb == 2:
            mo = mo + f32(scale) * (preds[i][1:].float() - mo)
        x0 = bf(sbf(k["a_sqrt"]) * s) - f32(k["b_sqrt"]) * mo
        d = f32(k["m3"]) * x0 - f32(k["m4"]) * old_b if k["second"] else x0
        prev = (bf(sbf(k["m1"]) * s) - f32(k["m2"]) * d) + bf(sbf(k["m_noise"]) * noise.float())
        lat_b, old_b = prev.to(torch.bfloat16), x0
        assert torch.equal(lat_a, lat_b), f"latents differ at step {i}"
        assert torch.equal(old_a, old_b), f"x0 differs at step {i}"


def test_fullsize_cases_reproduce_the_fixture_inputs():
    """tools/fullsize_cases.py builds the inputs of the full-depth fixtures from seeds (CPU generators, numpy): what it builds HERE must be what
    tools/make_fullsize_golden.py saw when it wrote tests/golden/fullsize_*.npz (the GPU tests rebuild them once more on the GPU box)."""
    import json
    import os
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fullsize_cases as fc
    meta = json.loads(str(np.load(os.path.join(fc.GOLDEN_DIR, "fullsize_dit.npz"))["meta"]))
    hidden, text, t = fc.dit_inputs()
    assert hidden.shape == (1, 11, 96, 60, 90) and text.shape == (1, 226, 4096) and int(t[0]) == meta["timestep"] == 999
    assert abs(float(hidden.float().sum()) - meta["input_sum"]) < 1e-3 * max(1.0, abs(meta["input_sum"]))
    assert meta["layers"] == 42 and meta["dit_seed"] == fc.DIT_SEED and meta["input_seed"] == fc.DIT_INPUT_SEED
    z = np.load(os.path.join(fc.GOLDEN_DIR, "fullsize_clip.npz"))
    cm = json.loads(str(z["meta"]))
    video = fc.clip_video()
    assert video.shape == (41, 480, 720, 3) and abs(float(video.astype(np.float64).sum()) - cm["video_sum"]) < 1e-6 * cm["video_sum"]
    assert cm["steps"] == fc.CLIP_STEPS == 4 and cm["clip_seed"] == fc.CLIP_SEED and cm["vae_seed"] == fc.VAE_SEED
    assert z["final_latents_bits"].shape == (1, 11, 56, 60, 90) and z["posterior_mean"].shape == (1, 16, 11, 60, 90)
    assert z["rgb_s8"].shape == (41, 480 // fc.DEC_STRIDE, 720 // fc.DEC_STRIDE, 3)
    lat = fc.from_bf16_bits(z["final_latents_bits"]).float()
    assert bool(np.isfinite(lat.numpy()).all()) and 1.0 < float(lat.abs().max()) < 20.0
