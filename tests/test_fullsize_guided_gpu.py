"""BASELINE configs[2] / configs[3] on their NAMED inputs, full depth and full geometry, on MI355X against fixtures the fp32 CPU oracle
wrote offline (tools/make_fullsize_golden.py prediction planning traj; tests/golden/fullsize_{prediction,planning,traj}.npz):

  * prediction: assets/example_obs/car.png + a forward-right raymap (camera_pose_to_raymap, the README recipe — the reference's own .npy
    is not part of the mount); planning: assets/example_obs_goal/01_obs.png + 01_goal.png.  480 x 720, 41 frames, all 42 blocks, CPU
    generator, 2 guided steps with the reference's dynamic classifier-free guidance (P:832-899): image(/goal) posterior and sampled
    latents (what `prepare_latents` builds the condition from: single-frame tiled VAE encode P:557-563, zero padding P:633-650, raymap
    packing P:652-670), the B = 2 noise prediction of step 0 (unconditional / conditional branch, 5 664-workgroup attention launch),
    final latents (latents L-inf / rel-L2) and the decoded rgb / disparity (pixel PSNR);
  * a 10-step reconstruction trajectory of the clip of test_fullsize_parity_gpu.py: drift against the oracle per step (4 -> 10 steps).

Weights and inputs are regenerated here from the seeds of tools/fullsize_cases.py; the three images travel as tests/golden/named_inputs.npz.
The oracle is this repo's restatement of diffusers (PARITY UNPINNED against diffusers itself).  Thresholds: ~1.3 x the values
measured on MI355X (profiles/r04_parity_fullsize.log).
"""
import gc
import json
import os
import sys
import time

import numpy as np
import pytest
import torch
from einops import rearrange

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import fullsize_cases as fc  # noqa: E402


def _load(name):
    path = os.path.join(fc.GOLDEN_DIR, name)
    assert os.path.exists(path), f"{path} missing: run tools/make_fullsize_golden.py prediction planning traj"
    z = np.load(path)
    return z, json.loads(str(z["meta"]))


@pytest.fixture(scope="module")
def modules(fullsize_modules):
    return fullsize_modules


def _pipeline(dit, vae):
    from aether.pipelines.aetherv1_pipeline_cogvideox import AetherV1PipelineCogVideoX
    from aether_amd.scheduler import CogVideoXDPMScheduler
    pipe = AetherV1PipelineCogVideoX(tokenizer=None, text_encoder=None, vae=vae, scheduler=CogVideoXDPMScheduler(), transformer=dit,
                                     empty_prompt_embeds=fc.prompt_embeds())
    pipe.set_progress_bar_config(disable=True)
    return pipe


def _guided_inputs(task):
    case = fc.GUIDED_CASES[task]
    image = fc.named_image(case["image"])
    goal = fc.named_image(case["goal"]) if case["goal"] else None
    raymap = fc.forward_right_raymap() if case["raymap"] else None
    return image, goal, raymap


# measured on MI355X (round 4, profiles/r04_parity_fullsize.log); bounds = ~1.3 x measured.  prediction: posterior 1.12e-2; B = 2 forward 1.45e-2 /
# 2.45 % (both branches); 2 guided steps: latents 2.20e-2 / 3.00 %, rgb 33.1 dB, disparity 4.5e-2 — guidance u + g (c - u) with g up to 4 amplifies
# the difference of two bf16 predictions, which is why the guided trajectory sits above the un-guided one (1.03e-2 after 4 steps)
GUIDED_BOUNDS = {
    # the B = 2 forward must stay within the bf16 ORACLE's own distance to the fixture (profiles/r04_bf16_oracle_calibration_guided.json:
    # prediction 1.84e-2 / 2.83 %, planning 1.84e-2 / 2.40 %): no worse than the reference dtype
    "prediction": dict(post_rel=1.45e-2, fwd_rel=1.84e-2, fwd_linf=0.0283, lat_rel=2.86e-2, lat_linf=0.039, psnr=30.8, disp_rel=5.85e-2),
    # planning: posteriors 1.07e-2 / 1.11e-2; B = 2 forward 1.38e-2 / 1.96 %; 2 guided steps: latents 2.46e-2 / 4.00 %, rgb 32.3 dB, disparity 4.4e-2
    "planning": dict(post_rel=1.44e-2, fwd_rel=1.79e-2, fwd_linf=0.024, lat_rel=3.2e-2, lat_linf=0.052, psnr=30.0, disp_rel=5.7e-2),
}


@pytest.mark.parametrize("task", ["prediction", "planning"])
def test_guided_condition_and_b2_forward(cuda, modules, task):
    """(1) `prepare_latents` on the named inputs: single-frame tiled VAE encode of the observation (/ goal), posterior mean against the oracle's;
    (2) ONE guided transformer call at B = 2, all 42 blocks, on the ORACLE's exact condition latents (bf16 bits from the fixture) and the
    exact initial noise (replayed from the CPU generator): unconditional and conditional noise prediction against the oracle's."""
    dit, vae = modules
    z, meta = _load(f"fullsize_{task}.npz")
    bd = GUIDED_BOUNDS[task]
    image, goal, raymap = _guided_inputs(task)
    pipe = _pipeline(dit, vae)
    img_t, goal_t, _, ray_t = pipe.preprocess_inputs(image=image, goal=goal, video=None, raymap=raymap, height=fc.HEIGHT, width=fc.WIDTH, num_frames=fc.FRAMES)
    # ---- (1) posterior of the observation (/ goal) --------------------------------------------------------------------------------
    names = [("image", img_t)] + ([("goal", goal_t)] if goal is not None else [])
    for nm, x in names:
        dist = vae.encode(x.unsqueeze(2)).latent_dist
        m = fc.metrics(dist.mode().cpu().float(), torch.from_numpy(z[f"{nm}_posterior_mean"].astype(np.float32)))
        print(f"\n[fullsize] {task}: {nm} posterior mean (1 frame, 9 tiles as 4/2/2/1 batches) vs fp32 oracle: rel-L2 {m['rel_l2']:.3e}  L-inf {100 * m['linf_rel']:.2f} % of max")
        assert m["rel_l2"] <= bd["post_rel"], m
    # ---- (2) B = 2 forward on the oracle's condition --------------------------------------------------------------------------------
    gen = torch.Generator().manual_seed(fc.GUIDED_SEED)
    for _ in names:                                                      # the posterior draws come first (P:552-631)
        torch.randn((1, 16, 1, fc.LAT_H, fc.LAT_W), generator=gen, dtype=torch.bfloat16)
    latents = torch.randn((1, fc.LAT_F, 56, fc.LAT_H, fc.LAT_W), generator=gen, dtype=torch.bfloat16)
    assert abs(float(latents.double().sum()) - float(z["initial_latents_sum"])) < 1e-6 * max(1.0, abs(float(z["initial_latents_sum"]))), "replayed noise differs"
    img_lat = fc.from_bf16_bits(z["image_latents_bits"])                                               # [1,1,16,60,90]
    pad = torch.zeros(1, fc.LAT_F - (2 if goal is not None else 1), 16, fc.LAT_H, fc.LAT_W, dtype=torch.bfloat16)
    parts = [img_lat, pad] + ([fc.from_bf16_bits(z["goal_latents_bits"])] if goal is not None else [])
    rgb_cond = torch.cat(parts, dim=1)
    if raymap is not None:
        rm = torch.from_numpy(raymap)[None].to(torch.bfloat16)
        rm = torch.cat([rm[:, : 4 - rm.shape[1] % 4], rm], dim=1)
        cam = rearrange(rm, "b (n t) c h w -> b t (n c) h w", n=4)
    else:
        cam = torch.zeros(1, fc.LAT_F, 24, fc.LAT_H, fc.LAT_W, dtype=torch.bfloat16)
    cond = torch.cat([rgb_cond, cam], dim=2)
    assert abs(float(cond.double().sum()) - meta["condition_sum"]) < 1e-6 * max(1.0, meta["condition_abs_sum"]), "rebuilt condition differs from the oracle's"
    un = cond.clone()
    if task == "planning":
        un[:, :, :16] = 0
    else:
        un[:, :1, :16] = 0
    model_in = torch.cat([torch.cat([latents] * 2), torch.cat([un, cond])], dim=2).to(cuda)
    rope = fc.rope_tables()
    out = dit(hidden_states=model_in, encoder_hidden_states=fc.prompt_embeds().repeat(2, 1, 1).to(cuda), timestep=torch.tensor([999, 999], device=cuda),
              ofs=None, image_rotary_emb=(rope[0].to(cuda), rope[1].to(cuda)), return_dict=False)[0]
    torch.cuda.synchronize()
    ref = torch.from_numpy(z["noise_pred0_s2"].astype(np.float32))
    got = out.cpu().float()[..., ::2, ::2]
    for b, nm in enumerate(("unconditional", "conditional")):
        m = fc.metrics(got[b], ref[b])
        print(f"[fullsize] {task}: B = 2 forward, 42 blocks, {nm} branch vs fp32 oracle: rel-L2 {m['rel_l2']:.3e}  L-inf {100 * m['linf_rel']:.2f} % of max {m['ref_max']:.2f}")
        assert m["rel_l2"] <= bd["fwd_rel"] and m["linf_rel"] <= bd["fwd_linf"], (nm, m)


@pytest.mark.parametrize("task", ["prediction", "planning"])
def test_guided_call_two_steps(cuda, modules, task):
    """The whole guided `__call__` (P:690-965) through the drop-in entry point on the named inputs: encode(s) -> condition -> 2 x (cat,
    B = 2 forward, dynamic-CFG combine, SDE-DPM++ step) -> two decodes."""
    dit, vae = modules
    z, meta = _load(f"fullsize_{task}.npz")
    bd = GUIDED_BOUNDS[task]
    image, goal, raymap = _guided_inputs(task)
    pipe = _pipeline(dit, vae)
    # ONE VAE workspace per pipeline geometry (AetherVAE.reserve_workspace, called at the top of `__call__`): a guided call — 1-frame image / goal
    # encodes, then two 11-frame decodes — must neither re-allocate it nor drop a captured hipGraph, whatever ran before (round 5 regrew 18 GB here)
    import warnings
    vae.reserve_workspace(fc.FRAMES, fc.HEIGHT, fc.WIDTH)
    ws_ptr, graphs_before = vae._workspace.data_ptr(), set(vae._graphs)
    t0 = time.perf_counter()
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        out = pipe(task=task, image=image, goal=goal, raymap=raymap, height=fc.HEIGHT, width=fc.WIDTH, num_frames=fc.FRAMES, fps=12,
                   num_inference_steps=fc.GUIDED_STEPS, generator=torch.Generator().manual_seed(fc.GUIDED_SEED))
    dt = time.perf_counter() - t0
    assert not [w for w in caught if "workspace grows" in str(w.message)], [str(w.message) for w in caught]
    assert vae._workspace.data_ptr() == ws_ptr and graphs_before <= set(vae._graphs), "the VAE workspace was re-allocated inside a guided call"
    assert isinstance(vae._graphs.get((True, fc.LAT_F, fc.LAT_H, fc.LAT_W, True)), tuple), "the second decode of the call should have been captured as a hipGraph"
    lat, ref_lat = pipe._final_latents.cpu().float(), fc.from_bf16_bits(z["final_latents_bits"]).float()
    ml = fc.metrics(lat, ref_lat)
    s = fc.DEC_STRIDE
    rgb, disp = torch.from_numpy(out.rgb)[:, ::s, ::s], torch.from_numpy(out.disparity)[:, ::s, ::s]
    p_rgb = fc.psnr(rgb, torch.from_numpy(z["rgb_s8"].astype(np.float32)))
    m_disp = fc.metrics(disp, torch.from_numpy(z["disparity_s8"].astype(np.float32)))
    print(f"\n[fullsize] {task} ({meta['inputs']}), {fc.GUIDED_STEPS} guided steps, dynamic CFG ({dt:.1f} s here, {meta['seconds_cpu_total']:.0f} s of CPU offline): "
          f"final latents rel-L2 {ml['rel_l2']:.3e}  L-inf {ml['linf']:.4f} ({100 * ml['linf_rel']:.2f} % of max|ref| {ml['ref_max']:.2f}); rgb PSNR {p_rgb:.1f} dB; "
          f"disparity rel-L2 {m_disp['rel_l2']:.3e}")
    assert out.rgb.shape == (fc.FRAMES, fc.HEIGHT, fc.WIDTH, 3) and np.isfinite(out.rgb).all() and np.isfinite(out.disparity).all()
    assert ml["rel_l2"] <= bd["lat_rel"] and ml["linf_rel"] <= bd["lat_linf"], ml
    assert p_rgb >= bd["psnr"] and m_disp["rel_l2"] <= bd["disp_rel"], (p_rgb, m_disp)


def _trajectory(pipe, steps, keep=None, task="reconstruction", seed=None, scales=None, **inputs):
    """Run a whole call with a scheduler that records the latents every step hands back (every 6th row / column) and, into `scales`, the
    guidance scale each fused step received.  Default: the reconstruction call on the clip of test_fullsize_parity_gpu.py."""
    from aether_amd.scheduler import CogVideoXDPMScheduler
    rec = {}

    class Spy(CogVideoXDPMScheduler):
        n = 0

        def _rec(self, res):
            if keep is None or Spy.n in keep:
                rec[Spy.n] = res[0][:, :, :, ::6, ::6].float().cpu()
            Spy.n += 1
            return res

        def step(self, *a, **kw):
            return self._rec(super().step(*a, **kw))

        def step_fused(self, *a, **kw):
            if scales is not None:
                scales.append(kw.get("guidance_scale"))
            return self._rec(super().step_fused(*a, **kw))

    # the pipeline decides from `step`'s SIGNATURE whether to hand it the generator (prepare_extra_step_kwargs, P:801): a bare (*a, **kw) wrapper
    # would silently send every scheduler draw to the global generator
    import inspect
    Spy.step.__signature__ = inspect.signature(CogVideoXDPMScheduler.step)
    pipe.scheduler = Spy()
    assert "generator" in pipe.prepare_extra_step_kwargs(torch.Generator(), 0.0)
    if task == "reconstruction":
        inputs = dict(video=fc.clip_video())
    out = pipe(task=task, height=fc.HEIGHT, width=fc.WIDTH, num_frames=fc.FRAMES, fps=12, num_inference_steps=steps,
               generator=torch.Generator().manual_seed(fc.CLIP_SEED if seed is None else seed), **inputs)
    return out, rec


# ---- BASELINE configs[2] / configs[3] at the step count BASELINE QUOTES: 50 guided steps, dynamic CFG on the n = 50 schedule ----------------------
# Fixtures: the fp32 ORACLE transformer run for 2 x 50 forwards per task with torch on an MI355X (tools/make_fullsize_golden_gpu.py: fp32 GEMMs, explicit
# fp32 soft-max attention; pinned against the CPU-generated fixtures to fp32 round-off, profiles/r05_gpu_oracle_pin.json, and against the first steps of
# the 10.5-hour CPU run of the same call, profiles/r05_prediction50_cpu_vs_device_oracle.json), final decodes by the fp32 CPU oracle VAE
# (tools/make_fullsize_golden.py decode50).  Bounds: the bf16 ORACLE's own distance to the same fixture along the same trajectory
# (profiles/r05_bf16_oracle_calibration_guided50.json) where it is larger than 1.3 x measured, else 1.3 x measured (profiles/r05_parity_fullsize.log).
# measured on MI355X (profiles/r05_parity_fullsize.log): prediction: 9.1e-4 after step 0, 5.9e-3 after 15, 9.2e-3 after 25, 1.35e-2 after 45, final 1.331e-2 /
# 2.26 %; planning: final 1.386e-2 / 2.63 %, largest along the trajectory 1.45e-2.  The bf16 ORACLE on the same trajectories: final 1.424e-2 / 2.58 % and
# 1.470e-2 / 2.96 % (transformer alone in bf16: 1.423e-2 / 1.469e-2) — `ref_rel` / `ref_linf`: the native path must not be further from the fp32 oracle
# than the reference dtype itself is.  `lat_rel` = 1.1 x the largest value measured along the trajectory.
GUIDED50_BOUNDS = {
    # decoded clip: prediction 36.5 dB, disparity 2.54e-2; planning 36.3 dB, 2.91e-2 (bounds: - 2 dB, x 1.2-1.3)
    "prediction": dict(lat_rel=1.49e-2, ref_rel=1.424e-2, ref_linf=0.0258, psnr=34.5, disp_rel=3.3e-2),
    "planning": dict(lat_rel=1.60e-2, ref_rel=1.470e-2, ref_linf=0.0296, psnr=34.0, disp_rel=3.5e-2),
}


@pytest.mark.parametrize("task", ["prediction", "planning"])
def test_guided_call_fifty_steps(cuda, modules, task):
    """The whole guided `__call__` (P:690-965) with the reference's default 50 steps (P:257-261) on the named inputs: latents along the trajectory
    (P:827-921), the guidance scale every step used against the reference's literal formula on the n = 50 schedule (P:880-893), final latents
    (L-inf / rel-L2), decoded rgb PSNR and disparity."""
    path = os.path.join(fc.GOLDEN_DIR, f"fullsize_{task}50.npz")
    if not os.path.exists(path):
        pytest.skip(f"{path} not generated (tools/make_fullsize_golden_gpu.py {task}50)")
    dit, vae = modules
    z, meta = _load(f"fullsize_{task}50.npz")
    bd = GUIDED50_BOUNDS[task]
    image, goal, raymap = _guided_inputs(task)
    pipe = _pipeline(dit, vae)
    kept, steps, scales = list(meta["kept_steps"]), int(meta["steps"]), []
    t0 = time.perf_counter()
    out, rec = _trajectory(pipe, steps, set(kept), task=task, seed=fc.GUIDED_SEED, scales=scales, image=image, goal=goal, raymap=raymap)
    dt = time.perf_counter() - t0
    # the n = 50 guidance-scale sequence: 1 + 3 (1 - cos(pi ((n - t) / n)^5)) / 2 with the TIMESTEP t in 999 ... 19, as the reference writes it
    assert len(scales) == steps and np.allclose(np.array(scales, np.float64), np.array(meta["guidance_scales"]), rtol=0, atol=1e-12), "guidance-scale sequence"
    errs = [fc.metrics(rec[i].to(torch.bfloat16).float(), fc.from_bf16_bits(z["step_latents_s6"][k]).float())["rel_l2"] for k, i in enumerate(kept)]
    fin = fc.metrics(pipe._final_latents.cpu().float(), fc.from_bf16_bits(z["final_latents_bits"]).float())
    msg = (f"\n[fullsize] {task} ({meta['inputs']}), {steps} guided steps, dynamic CFG (scale {min(scales):.2f} .. {max(scales):.2f}; {dt:.1f} s here): latents rel-L2 vs "
           "fp32 oracle after steps " + ", ".join(f"{i}: {e:.2e}" for i, e in zip(kept, errs))
           + f"; final: rel-L2 {fin['rel_l2']:.3e}  L-inf {fin['linf']:.4f} ({100 * fin['linf_rel']:.2f} % of max|ref| {fin['ref_max']:.2f})")
    s = fc.DEC_STRIDE
    p_rgb = m_disp = None
    if "rgb_s8" in z.files:
        p_rgb = fc.psnr(torch.from_numpy(out.rgb)[:, ::s, ::s], torch.from_numpy(z["rgb_s8"].astype(np.float32)))
        m_disp = fc.metrics(torch.from_numpy(out.disparity)[:, ::s, ::s], torch.from_numpy(z["disparity_s8"].astype(np.float32)))
        msg += f"; rgb PSNR {p_rgb:.1f} dB; disparity rel-L2 {m_disp['rel_l2']:.3e}"
    from einops import rearrange                  # the raymap output is the camera channels of the final latents, un-folded (P:942-945)
    ref_ray = rearrange(fc.from_bf16_bits(z["final_latents_bits"]).float()[:, :, 32:], "b t (n c) h w -> b (n t) c h w", n=4)[0, -fc.FRAMES:]
    m_ray = fc.metrics(torch.from_numpy(out.raymap), ref_ray)
    print(msg + f"; raymap rel-L2 {m_ray['rel_l2']:.3e}")
    assert out.rgb.shape == (fc.FRAMES, fc.HEIGHT, fc.WIDTH, 3) and np.isfinite(out.rgb).all() and np.isfinite(out.disparity).all()
    assert max(errs) <= bd["lat_rel"] and fin["rel_l2"] <= bd["ref_rel"] and fin["linf_rel"] <= bd["ref_linf"], (errs, fin)
    if p_rgb is not None and bd["psnr"] is not None:
        assert p_rgb >= bd["psnr"] and m_disp["rel_l2"] <= bd["disp_rel"], (p_rgb, m_disp)


SEVENTEEN_BOUNDS = dict(rel=1.57e-2, linf=0.0224)    # measured on MI355X: 1.204e-2 / 1.72 % (profiles/r06_parity_fullsize.log); bounds 1.3 x


def test_seventeen_frame_clip_full_size(cuda, modules):
    """The shortest clip the reference admits (17 frames -> 5 latent frames, S = 226 + 6 750) at full width and depth: ONE 42-block forward against the
    fp32 oracle at this geometry (tests/golden/fullsize_dit17.npz: tools/make_fullsize_golden_gpu.py dit17, the oracle run by torch on an MI355X), then
    reconstruction (B = 1) and planning (B = 2), two steps each: the shape-specific paths the 41-frame tests do not reach (GEMM tail launches at
    another M, the persistent LayerNorm grid, VAE frame chunking 9 + 8 and lane assignment) — finite outputs of the right shape, and the calls are
    deterministic (same seed -> same bits)."""
    dit, vae = modules
    z17, meta17 = _load("fullsize_dit17.npz")
    from oracle.rope import prepare_rope
    lat_f = int(meta17["latent_frames"])
    g = torch.Generator().manual_seed(int(meta17["input_seed"]))
    hidden = torch.randn(1, lat_f, 96, fc.LAT_H, fc.LAT_W, generator=g).to(torch.bfloat16)
    text = (torch.randn(1, fc.TEXT_LEN, fc.TEXT_DIM, generator=g) * 0.1).to(torch.bfloat16)
    assert abs(float(hidden.float().sum()) - meta17["input_sum"]) < 1e-3 * max(1.0, abs(meta17["input_sum"]))
    rope = prepare_rope(fc.HEIGHT, fc.WIDTH, lat_f, 12)
    out17 = dit(hidden_states=hidden.to(cuda), encoder_hidden_states=text.to(cuda), timestep=torch.tensor([int(meta17["timestep"])], device=cuda), ofs=None,
                image_rotary_emb=(rope[0].to(cuda), rope[1].to(cuda)), return_dict=False)[0]
    m17 = fc.metrics(out17.cpu().float(), torch.from_numpy(z17["out"].astype(np.float32)))
    print(f"\n[fullsize] DiT 42 blocks at 17 frames (S = {fc.TEXT_LEN + lat_f * fc.LAT_H * fc.LAT_W // 4}), B = 1 vs the fp32 oracle: rel-L2 {m17['rel_l2']:.3e}  "
          f"L-inf {100 * m17['linf_rel']:.2f} % of max|ref| {m17['ref_max']:.3f}")
    assert m17["rel_l2"] <= SEVENTEEN_BOUNDS["rel"] and m17["linf_rel"] <= SEVENTEEN_BOUNDS["linf"], m17
    pipe = _pipeline(dit, vae)
    F = 17
    video = fc.clip_video()[:F]
    for task, kw in (("reconstruction", dict(video=video)), ("planning", dict(image=video[0], goal=video[-1]))):
        outs = [pipe(task=task, height=fc.HEIGHT, width=fc.WIDTH, num_frames=F, num_inference_steps=2, fps=12,
                     generator=torch.Generator().manual_seed(7), **kw) for _ in range(2)]
        o = outs[0]
        assert o.rgb.shape == (F, fc.HEIGHT, fc.WIDTH, 3) and o.disparity.shape == (F, fc.HEIGHT, fc.WIDTH) and o.raymap.shape == (F, 6, fc.LAT_H, fc.LAT_W)
        assert np.isfinite(o.rgb).all() and np.isfinite(o.disparity).all() and np.isfinite(o.raymap).all() and float(o.rgb.std()) > 1e-3
        assert np.array_equal(o.rgb, outs[1].rgb) and np.array_equal(o.raymap, outs[1].raymap), f"{task}: two runs with one seed differ"


# ---- the reconstruction trajectories under DEVICE semantics ------------------------------------------------------------------------------------------
# These are the PRIMARY reconstruction fixtures.  The CPU-generated trajectory fixtures of rounds 2-4 (fullsize_clip's final latents, _traj, _traj50; the 2-step
# _prediction / _planning above) contain an artefact of torch-CPU: in
# `m1 * sample` and `m_noise * noise` (bf16 tensor times python / 0-dim scalar, diffusers' DPM update) the CPU kernels round the SCALAR to bf16 before the
# multiplication, a CUDA / HIP device keeps it in fp32.  The reference runs on the device (D:218), the native `aether_dpm_step` is bit-identical to the
# device op sequence — so every step of a CPU-oracle trajectory carries a perturbation of up to 2^-9 on two coefficients that neither the reference nor the
# native path has (profiles/r05_prediction50_cpu_vs_device_oracle.json: the same fp32 oracle on CPU and on the device differs by 2.6e-3 after TWO steps,
# 20 % of the bf16 latents by one ulp).  These fixtures are the same call, same oracle, same seed, with the transformer and the scheduler update executed by
# torch on the device (tools/make_fullsize_golden_gpu.py recon4 recon50); decodes by the fp32 CPU oracle VAE.
# measured on MI355X (profiles/r05_parity_fullsize.log): 4 steps: 3.6e-3 / 6.6e-3 / 9.7e-3 / 1.06e-2 along the trajectory, final 1.017e-2 / 1.53 % (CPU-semantics
# fixture: 1.032e-2); 10 steps: final 8.99e-3 (9.49e-3; fixture not committed: 7 MB for one more point of the same curve); 50 steps: 7.4e-4 after step 0, 3.3e-3
# after 10, 5.2e-3 after 20, 7.8e-3 after 35, 1.04e-2 after 50, final 1.047e-2 / 2.76 % — against 1.354e-2 for the CPU-semantics fixture: a quarter of what round 4
# reported as the 50-step drift was the oracle's own CPU artefact.  Decoded clips: 4 steps rgb 39.2 dB / disparity 2.16e-2, 50 steps rgb 38.5 dB / disparity 2.11e-2
# (CPU-semantics fixtures: 39.1 / 2.19e-2 and 36.9 / 2.58e-2).  Bounds ~1.3 x measured (PSNR: - 2 dB).
# `ref_rel`: the bf16 ORACLE's own final-latent distance on the same trajectories (profiles/r05_bf16_oracle_calibration_recon.json: 1.342e-2 after 4 steps, 1.135e-2 after 50): the native path must not be above it.
# `fin_rel` = the smaller of that and 1.3 x measured; the raymap output (camera channels of the final latents, P:942-945): 9.6e-3 measured after 4 steps.
# The 10- and 50-step CPU-semantics fixtures (fullsize_traj / _traj50) and the tests that asserted against them were retired in round 6: their bounds had to sit
# 30 % above a distance of which a quarter was the fixture's own artefact.
RECON_DEVICE_BOUNDS = {4: dict(lat_rel=1.38e-2, lat_linf=0.0199, fin_rel=1.32e-2, psnr=37.2, disp_rel=2.8e-2, ray_rel=1.25e-2),
                       50: dict(lat_rel=1.36e-2, lat_linf=0.036, fin_rel=1.135e-2, psnr=36.5, disp_rel=2.75e-2, ray_rel=1.4e-2)}


@pytest.mark.parametrize("steps", [4, 50])
def test_reconstruction_against_device_oracle(cuda, modules, steps):
    """configs[1] (50 steps) and the reference default (4 steps) against the fp32 oracle under the reference's (device) semantics of
    the scheduler update: latents after every kept step, final latents, decoded rgb / disparity when the fixture carries them."""
    path = os.path.join(fc.GOLDEN_DIR, f"fullsize_recon{steps}_device.npz")
    if not os.path.exists(path):
        pytest.skip(f"{path} not generated (tools/make_fullsize_golden_gpu.py recon{steps})")
    dit, vae = modules
    z, meta = _load(f"fullsize_recon{steps}_device.npz")
    pipe = _pipeline(dit, vae)
    kept = list(meta["kept_steps"])
    out, rec = _trajectory(pipe, steps, set(kept))
    errs = [fc.metrics(rec[i].to(torch.bfloat16).float(), fc.from_bf16_bits(z["step_latents_s6"][k]).float())["rel_l2"] for k, i in enumerate(kept)]
    fin = fc.metrics(pipe._final_latents.cpu().float(), fc.from_bf16_bits(z["final_latents_bits"]).float())
    msg = (f"\n[fullsize] reconstruction, {steps} steps, vs the fp32 oracle under DEVICE semantics: latents rel-L2 after steps " + ", ".join(f"{i}: {e:.2e}" for i, e in zip(kept, errs))
           + f"; final: rel-L2 {fin['rel_l2']:.3e}  L-inf {fin['linf']:.4f} ({100 * fin['linf_rel']:.2f} % of max|ref| {fin['ref_max']:.2f})")
    s = fc.DEC_STRIDE
    p_rgb = m_disp = None
    if "rgb_s8" in z.files:
        p_rgb = fc.psnr(torch.from_numpy(out.rgb)[:, ::s, ::s], torch.from_numpy(z["rgb_s8"].astype(np.float32)))
        m_disp = fc.metrics(torch.from_numpy(out.disparity)[:, ::s, ::s], torch.from_numpy(z["disparity_s8"].astype(np.float32)))
        msg += f"; rgb PSNR {p_rgb:.1f} dB; disparity rel-L2 {m_disp['rel_l2']:.3e}"
    from einops import rearrange                  # the raymap output is the camera channels of the final latents, un-folded (P:942-945)
    ref_ray = rearrange(fc.from_bf16_bits(z["final_latents_bits"]).float()[:, :, 32:], "b t (n c) h w -> b (n t) c h w", n=4)[0, -fc.FRAMES:]
    m_ray = fc.metrics(torch.from_numpy(out.raymap), ref_ray)
    print(msg + f"; raymap rel-L2 {m_ray['rel_l2']:.3e}")
    assert np.isfinite(out.rgb).all() and np.isfinite(out.disparity).all()
    bd = RECON_DEVICE_BOUNDS[steps]
    if bd is not None:
        assert max(errs) <= bd["lat_rel"] and fin["rel_l2"] <= bd["fin_rel"] and fin["linf_rel"] <= bd["lat_linf"], (errs, fin)
        assert out.raymap.shape == (fc.FRAMES, 6, fc.LAT_H, fc.LAT_W) and m_ray["rel_l2"] <= bd["ray_rel"], m_ray
        if p_rgb is not None and bd.get("psnr") is not None:
            assert p_rgb >= bd["psnr"] and m_disp["rel_l2"] <= bd["disp_rel"], (p_rgb, m_disp)
