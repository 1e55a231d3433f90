"""Generates the FULL-DEPTH, FULL-GEOMETRY parity fixtures (tests/golden/fullsize_{dit,clip}.npz) with the fp32 CPU oracle.

Run offline in the build container (no GPU needed; ~1 h on 8 vCPUs, 30 GB of RAM):

    python tools/make_fullsize_golden.py            # both stages
    python tools/make_fullsize_golden.py dit        # one stage

  dit   ONE transformer forward, all 42 blocks, S = 15 076 tokens, B = 1 (P:865-875) on seeded inputs
        -> the noise prediction [1, 11, 56, 60, 90] as float16 (6.7 MB).
  clip  the whole reconstruction call (P:690-965) at 41 x 480 x 720 with the reference's default 4 steps and a CPU generator
        (seed 42): tiled 41-frame VAE encode (all 9 tiles, all 5 frame chunks) -> posterior sample -> 4 x (42-block forward +
        SDE-DPM++ step) -> two tiled 11 -> 41-frame decodes.  Recorded: the posterior (mean float16, log-variance sub-sampled),
        the final latents (exact bf16 bits), per-step latents (sub-sampled), per-step noise-prediction norms, the decoded rgb
        clip and the disparity (every 8th row / column of every frame).  12 MB.

The three modules are the oracle restatements (oracle/dit.py, oracle/vae.py, PARITY UNPINNED against diffusers itself) in fp32
with bf16-representable weights; every random draw and every tensor passed between modules keeps the reference dtype (bf16),
`compute_dtype=float32` of oracle.pipeline.sample.  tests/test_fullsize_parity_gpu.py rebuilds the same weights / inputs from
the seeds in tools/fullsize_cases.py on the GPU box and compares the HIP path with these files.
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fullsize_cases as fc  # noqa: E402


def log(msg):
    print(f"[{time.strftime('%H:%M:%S')}] {msg}", flush=True)


def stage_dit(dit):
    hidden, text, t = fc.dit_inputs()
    rope = fc.rope_tables()
    t0 = time.perf_counter()
    with torch.no_grad():
        out = dit(hidden.float(), text.float(), t, image_rotary_emb=rope)[0]
    dt = time.perf_counter() - t0
    log(f"dit: 42-block forward at S=15076 took {dt:.1f} s; |out| max {out.abs().max():.3f} rms {out.pow(2).mean().sqrt():.4f}")
    assert torch.isfinite(out).all()
    meta = dict(seconds_cpu=dt, threads=torch.get_num_threads(), torch=torch.__version__, dit_seed=fc.DIT_SEED,
                input_seed=fc.DIT_INPUT_SEED, timestep=int(t[0]), out_max=float(out.abs().max()), out_rms=float(out.pow(2).mean().sqrt()),
                input_sum=float(hidden.float().sum()), layers=len(dit.transformer_blocks))
    np.savez(os.path.join(fc.GOLDEN_DIR, "fullsize_dit.npz"), out=out.numpy().astype(np.float16), meta=json.dumps(meta))
    log("dit: wrote tests/golden/fullsize_dit.npz")


def stage_clip(dit):
    from aether_amd.scheduler import CogVideoXDPMScheduler
    from oracle.pipeline import sample
    vae = fc.build_oracle_vae()
    video = fc.clip_video()
    v = fc.video_as_model_input(video)
    times = {}
    mark = [time.perf_counter()]
    step_lat = []

    def on_step(i, latents):
        now = time.perf_counter()
        times[f"step{i}"] = now - mark[0]
        mark[0] = now
        step_lat.append(latents[:, :, :, ::3, ::3].clone())
        log(f"clip: step {i} done ({times[f'step{i}']:.1f} s)")

    trace = {"on_step": on_step}
    t0 = time.perf_counter()
    rgb, disp, rm = sample("reconstruction", dit, vae, CogVideoXDPMScheduler(), fc.prompt_embeds(), video=v, height=fc.HEIGHT,
                           width=fc.WIDTH, num_frames=fc.FRAMES, num_inference_steps=fc.CLIP_STEPS,
                           generator=torch.Generator().manual_seed(fc.CLIP_SEED), rope=fc.rope_tables(),
                           compute_dtype=torch.float32, trace=trace)
    total = time.perf_counter() - t0
    log(f"clip: whole reconstruction call took {total:.1f} s")
    mean, logvar = trace["posterior"][0]
    s = fc.DEC_STRIDE
    meta = dict(seconds_cpu_total=total, step_seconds=times, threads=torch.get_num_threads(), torch=torch.__version__,
                dit_seed=fc.DIT_SEED, vae_seed=fc.VAE_SEED, clip_seed=fc.CLIP_SEED, steps=fc.CLIP_STEPS,
                noise_pred_rms=[float(p.pow(2).mean().sqrt()) for p in trace["noise_pred"]],
                noise_pred_max=[float(p.abs().max()) for p in trace["noise_pred"]],
                video_sum=float(video.astype(np.float64).sum()))
    meta["fixture_layout"] = f"decoded / post-processed arrays keep every {s}th row and column of every frame (DEC_STRIDE); per-step latents every 6th"
    np.savez(os.path.join(fc.GOLDEN_DIR, "fullsize_clip.npz"),
             posterior_mean=mean.numpy().astype(np.float16),                              # [1,16,11,60,90]
             posterior_logvar_s2=logvar[..., ::2, ::2].numpy().astype(np.float16),
             final_latents_bits=fc.bf16_bits(trace["final_latents"]),                     # [1,11,56,60,90], exact (the raymap output is a re-arrangement of it)
             step_latents_s6=np.stack([fc.bf16_bits(x[..., ::2, ::2]) for x in step_lat]),
             rgb_decoded_s8=trace["decoded"][0][:, :, :, ::s, ::s].numpy().astype(np.float16),    # raw decoder output [1,3,41,60,90]
             rgb_s8=rgb[:, ::s, ::s].numpy().astype(np.float16),                           # post-processed outputs (P:925-949)
             disparity_s8=disp[:, ::s, ::s].numpy().astype(np.float16),
             meta=json.dumps(meta))
    log("clip: wrote tests/golden/fullsize_clip.npz")


def stage_guided(dit, task):
    """BASELINE configs[2] / configs[3] on their NAMED inputs at the BASELINE geometry: the whole guided call (P:690-965) with
    GUIDED_STEPS steps, dynamic classifier-free guidance (P:880-893) and a CPU generator.  Recorded: the image (/ goal) posterior and
    its sampled latents (what `prepare_latents` builds `condition_latents` from, P:557-650), the B = 2 noise prediction of step 0
    (unconditional, conditional: all 42 blocks at batch 2, every 2nd row / column), the per-step guidance scale, the final latents
    (exact bf16 bits) and the decoded rgb / disparity (every 8th row / column)."""
    from aether_amd.scheduler import CogVideoXDPMScheduler
    from oracle.pipeline import sample
    case = fc.GUIDED_CASES[task]
    vae = fc.build_oracle_vae()
    image = fc.image_as_model_input(fc.named_image(case["image"]))
    goal = fc.image_as_model_input(fc.named_image(case["goal"])) if case["goal"] else None
    raymap = torch.from_numpy(fc.forward_right_raymap())[None] if case["raymap"] else None
    times, mark, step_lat = {}, [time.perf_counter()], []

    def on_step(i, latents):
        now = time.perf_counter()
        times[f"step{i}"] = now - mark[0]
        mark[0] = now
        step_lat.append(latents[:, :, :, ::6, ::6].clone())
        log(f"{task}: step {i} done ({times[f'step{i}']:.1f} s)")

    trace = {"on_step": on_step}
    t0 = time.perf_counter()
    rgb, disp, rm = sample(task, dit, vae, CogVideoXDPMScheduler(), fc.prompt_embeds(), image=image, goal=goal, raymap=raymap,
                           height=fc.HEIGHT, width=fc.WIDTH, num_frames=fc.FRAMES, num_inference_steps=fc.GUIDED_STEPS,
                           generator=torch.Generator().manual_seed(fc.GUIDED_SEED), rope=fc.rope_tables(),
                           compute_dtype=torch.float32, trace=trace)
    total = time.perf_counter() - t0
    log(f"{task}: whole guided call took {total:.1f} s")
    s = fc.DEC_STRIDE
    cond = trace["condition_latents"]                                                  # [1,11,40,60,90] bf16
    meta = dict(seconds_cpu_total=total, step_seconds=times, threads=torch.get_num_threads(), torch=torch.__version__, task=task,
                dit_seed=fc.DIT_SEED, vae_seed=fc.VAE_SEED, seed=fc.GUIDED_SEED, steps=fc.GUIDED_STEPS, inputs=case,
                noise_pred_rms=[float(p.pow(2).mean().sqrt()) for p in trace["noise_pred"]],
                noise_pred_max=[float(p.abs().max()) for p in trace["noise_pred"]],
                condition_sum=float(cond.double().sum()), condition_abs_sum=float(cond.double().abs().sum()),
                raymap_sum=(float(raymap.double().sum()) if raymap is not None else None))
    arrays = dict(
        image_posterior_mean=trace["posterior"][0][0].numpy().astype(np.float16),      # [1,16,1,60,90]
        image_latents_bits=fc.bf16_bits(cond[:, :1, :16]),                             # sampled + scaled, exact
        noise_pred0_s2=trace["noise_pred"][0][..., ::2, ::2].numpy().astype(np.float16),   # [2,11,56,30,45]: (uncond, cond)
        initial_latents_sum=np.float64(trace["initial_latents"].double().sum().item()),
        final_latents_bits=fc.bf16_bits(trace["final_latents"]),
        step_latents_s6=np.stack([fc.bf16_bits(x) for x in step_lat]),
        rgb_s8=rgb[:, ::s, ::s].numpy().astype(np.float16), disparity_s8=disp[:, ::s, ::s].numpy().astype(np.float16),
        meta=json.dumps(meta))
    if goal is not None:
        arrays["goal_posterior_mean"] = trace["posterior"][1][0].numpy().astype(np.float16)
        arrays["goal_latents_bits"] = fc.bf16_bits(cond[:, -1:, :16])
    np.savez_compressed(os.path.join(fc.GOLDEN_DIR, f"fullsize_{task}.npz"), **arrays)
    log(f"{task}: wrote tests/golden/fullsize_{task}.npz")


def stage_guided_long(dit, task, steps, keep, name):
    """BASELINE configs[2] / configs[3] at the step count BASELINE QUOTES: the whole guided call (P:690-965) with `steps` guided steps (P:827-921),
    dynamic classifier-free guidance evaluated on the n = `steps` schedule (P:880-893) and a CPU generator (seed GUIDED_SEED), on the named inputs
    of `stage_guided`.  2 x `steps` B = 1-equivalent 42-block fp32 forwards: ~10.5 h of CPU for 50 steps.  Because that is most of a build
    session, the run CHECKPOINTS: after every step it rewrites <name>.partial.npz (kept-step latents every 6th row / column so far, the latents
    of the last finished step every 2nd row / column, per-step guidance scale and noise-prediction norms) so a run that is cut short still
    leaves a fixture of the first k steps OF THE n-STEP SCHEDULE; the complete file adds the final latents (exact bf16 bits) and the decoded
    rgb / disparity (every 8th row / column)."""
    import math
    from aether_amd.scheduler import CogVideoXDPMScheduler
    from oracle.pipeline import sample
    case = fc.GUIDED_CASES[task]
    vae = fc.build_oracle_vae()
    image = fc.image_as_model_input(fc.named_image(case["image"]))
    goal = fc.image_as_model_input(fc.named_image(case["goal"])) if case["goal"] else None
    raymap = torch.from_numpy(fc.forward_right_raymap())[None] if case["raymap"] else None
    sched = CogVideoXDPMScheduler()
    sched.set_timesteps(steps)
    ts = [int(t) for t in sched.timesteps]
    scales = [1 + 3.0 * ((1 - math.cos(math.pi * ((steps - t) / steps) ** 5.0)) / 2) for t in ts]      # P:886-893, guidance_scale 3.0 (P:262-266)
    out_dir = os.environ.get("AETHER_GOLDEN_PARTIAL_DIR", fc.GOLDEN_DIR)
    partial = os.path.join(out_dir, name.replace(".npz", ".partial.npz"))
    times, mark, step_lat, rms, mx = {}, [time.perf_counter()], {}, [], []
    trace = {}

    def meta_now(done, total=None):
        return dict(task=task, steps=steps, steps_done=done, timesteps=ts, guidance_scales=scales, kept_steps=sorted(step_lat), inputs=case,
                    seed=fc.GUIDED_SEED, dit_seed=fc.DIT_SEED, vae_seed=fc.VAE_SEED, step_seconds=times, seconds_cpu_total=total,
                    threads=torch.get_num_threads(), torch=torch.__version__, noise_pred_rms=rms, noise_pred_max=mx)

    def on_step(i, latents):
        now = time.perf_counter()
        times[f"step{i}"] = now - mark[0]
        mark[0] = now
        for p in trace["noise_pred"]:                                  # [2, ...] (unconditional, conditional); keep the norms, drop the tensor
            rms.append([float(p[b].pow(2).mean().sqrt()) for b in range(p.shape[0])])
            mx.append(float(p.abs().max()))
        trace["noise_pred"].clear()
        if i in keep:
            step_lat[i] = fc.bf16_bits(latents[:, :, :, ::6, ::6])
        kept = sorted(step_lat)
        np.savez_compressed(partial + ".tmp.npz", step_latents_s6=np.stack([step_lat[k] for k in kept]) if kept else np.zeros(0),
                            last_latents_s2_bits=fc.bf16_bits(latents[..., ::2, ::2]), meta=json.dumps(meta_now(i + 1)))
        os.replace(partial + ".tmp.npz", partial)
        log(f"{task}{steps}: step {i} (t = {ts[i]}, guidance {scales[i]:.4f}) done ({times[f'step{i}']:.1f} s); checkpoint {os.path.basename(partial)}")

    trace["on_step"] = on_step
    t0 = time.perf_counter()
    rgb, disp, rm = sample(task, dit, vae, sched, fc.prompt_embeds(), image=image, goal=goal, raymap=raymap, height=fc.HEIGHT, width=fc.WIDTH,
                           num_frames=fc.FRAMES, num_inference_steps=steps, generator=torch.Generator().manual_seed(fc.GUIDED_SEED),
                           rope=fc.rope_tables(), compute_dtype=torch.float32, trace=trace)
    total = time.perf_counter() - t0
    s = fc.DEC_STRIDE
    kept = sorted(step_lat)
    cond = trace["condition_latents"]
    meta = meta_now(steps, total)
    meta.update(condition_sum=float(cond.double().sum()), condition_abs_sum=float(cond.double().abs().sum()))
    np.savez_compressed(os.path.join(fc.GOLDEN_DIR, name), step_latents_s6=np.stack([step_lat[k] for k in kept]),
                        final_latents_bits=fc.bf16_bits(trace["final_latents"]),
                        initial_latents_sum=np.float64(trace["initial_latents"].double().sum().item()),
                        rgb_s8=rgb[:, ::s, ::s].numpy().astype(np.float16), disparity_s8=disp[:, ::s, ::s].numpy().astype(np.float16),
                        meta=json.dumps(meta))
    log(f"{task}{steps}: whole guided call took {total:.1f} s; wrote tests/golden/{name}")


def stage_decode_long(name):
    """Finish a long guided fixture whose trajectory was computed elsewhere (tools/make_fullsize_golden_gpu.py stores the final latents, exact bf16
    bits): the two final decodes of P:925-940 with the fp32 CPU oracle VAE, post-processed as `oracle.pipeline.sample` does, every 8th row / column."""
    path = os.path.join(fc.GOLDEN_DIR, name)
    z = dict(np.load(path))
    meta = json.loads(str(z["meta"]))
    vae = fc.build_oracle_vae()
    sf = vae.config.scaling_factor
    lat = fc.from_bf16_bits(z["final_latents_bits"])                                   # [1,11,56,60,90] bf16
    t0 = time.perf_counter()

    def dec(x):
        with torch.no_grad():
            return vae.decode((1 / sf * x.permute(0, 2, 1, 3, 4)).float()).sample.to(torch.bfloat16)

    rgb = dec(lat[:, :, :16])
    rgb = (rgb[0].permute(1, 0, 2, 3) / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).float()
    log(f"{name}: rgb decode done ({time.perf_counter() - t0:.0f} s)")
    disp = dec(lat[:, :, 16:32]).mean(dim=1)
    disp = torch.square(disp * 0.5 + 0.5).float()[0]
    s = fc.DEC_STRIDE
    z["rgb_s8"], z["disparity_s8"] = rgb[:, ::s, ::s].numpy().astype(np.float16), disp[:, ::s, ::s].numpy().astype(np.float16)
    meta.update(decoded=True, decode_seconds_cpu=time.perf_counter() - t0)
    z["meta"] = json.dumps(meta)
    np.savez_compressed(path, **z)
    log(f"{name}: both decodes took {meta['decode_seconds_cpu']:.0f} s; rewrote tests/golden/{name}")


def stage_traj(dit, steps=None, name="fullsize_traj.npz", with_decodes=False, keep_steps=None):
    """The reconstruction call of `stage_clip` (same clip, same seed -> same posterior sample and initial latents) with `steps` steps: per-step
    latents (every 6th row / column; all steps, or `keep_steps`) and the final latents (every 2nd) — how the drift against the fp32 oracle
    grows with the number of steps.  `steps` = TRAJ_STEPS (10, no decodes) for the growth law; `steps` = 50 with decodes = BASELINE configs[1]
    itself ("4D reconstruction, 41x480x720, 50 steps"): final latents and the decoded rgb / disparity of the headline configuration."""
    from aether_amd.scheduler import CogVideoXDPMScheduler
    from oracle.pipeline import sample
    steps = steps or fc.TRAJ_STEPS

    class NoDecode:
        """The decodes of P:931,936 are irrelevant to the latent trajectory: replace them by zeros of the right shape."""
        def __init__(self, vae):
            self.vae, self.config = vae, vae.config

        def encode(self, x):
            return self.vae.encode(x)

        def decode(self, z):
            import types
            return types.SimpleNamespace(sample=torch.zeros(z.shape[0], 3, (z.shape[2] - 1) * 4 + 1, z.shape[3] * 8, z.shape[4] * 8, dtype=z.dtype))

    vae = fc.build_oracle_vae()
    if not with_decodes:
        vae = NoDecode(vae)
    v = fc.video_as_model_input(fc.clip_video())
    step_lat, times, mark = {}, {}, [time.perf_counter()]

    def on_step(i, latents):
        now = time.perf_counter()
        times[f"step{i}"] = now - mark[0]
        mark[0] = now
        if keep_steps is None or i in keep_steps:
            step_lat[i] = latents[:, :, :, ::6, ::6].clone()
        log(f"traj{steps}: step {i} done ({times[f'step{i}']:.1f} s)")

    trace = {"on_step": on_step}
    t0 = time.perf_counter()
    rgb, disp, rm = sample("reconstruction", dit, vae, CogVideoXDPMScheduler(), fc.prompt_embeds(), video=v, height=fc.HEIGHT, width=fc.WIDTH,
                           num_frames=fc.FRAMES, num_inference_steps=steps, generator=torch.Generator().manual_seed(fc.CLIP_SEED),
                           rope=fc.rope_tables(), compute_dtype=torch.float32, trace=trace)
    total = time.perf_counter() - t0
    kept = sorted(step_lat)
    meta = dict(seconds_cpu_total=total, step_seconds=times, threads=torch.get_num_threads(), torch=torch.__version__, steps=steps,
                dit_seed=fc.DIT_SEED, vae_seed=fc.VAE_SEED, clip_seed=fc.CLIP_SEED, kept_steps=kept, with_decodes=with_decodes,
                noise_pred_rms=[float(p.pow(2).mean().sqrt()) for p in trace["noise_pred"]])
    arrays = dict(step_latents_s6=np.stack([fc.bf16_bits(step_lat[i]) for i in kept]),
                  final_latents_s2_bits=fc.bf16_bits(trace["final_latents"][..., ::2, ::2]), meta=json.dumps(meta))
    if with_decodes:
        s = fc.DEC_STRIDE
        arrays["rgb_s8"] = rgb[:, ::s, ::s].numpy().astype(np.float16)
        arrays["disparity_s8"] = disp[:, ::s, ::s].numpy().astype(np.float16)
    np.savez_compressed(os.path.join(fc.GOLDEN_DIR, name), **arrays)
    log(f"traj: {steps}-step reconstruction trajectory took {total:.1f} s; wrote tests/golden/{name}")


def main():
    stages = sys.argv[1:] or ["dit", "clip"]
    torch.set_num_threads(int(os.environ.get("AETHER_GOLDEN_THREADS", os.cpu_count() or 8)))
    if stages[0] == "decode50":                                        # decode50 <file.npz> ...: needs the VAE only
        for name in stages[1:]:
            stage_decode_long(name)
        return
    log(f"building the 42-block oracle transformer (seed {fc.DIT_SEED}) ...")
    t0 = time.perf_counter()
    dit, _ = fc.build_oracle_dit()
    log(f"... {time.perf_counter() - t0:.1f} s")
    if "dit" in stages:
        stage_dit(dit)
    if "clip" in stages:
        stage_clip(dit)
    for task in ("prediction", "planning"):
        if task in stages:
            stage_guided(dit, task)
    if "prediction50" in stages:
        stage_guided_long(dit, "prediction", fc.HEADLINE_STEPS, set(fc.GUIDED_LONG_KEEP), "fullsize_prediction50.npz")
    if "planning10" in stages:
        stage_guided_long(dit, "planning", fc.TRAJ_STEPS, set(range(fc.TRAJ_STEPS)), "fullsize_planning10.npz")
    if "traj" in stages:
        stage_traj(dit)
    if "traj50" in stages:
        stage_traj(dit, steps=fc.HEADLINE_STEPS, name="fullsize_traj50.npz", with_decodes=True, keep_steps=fc.HEADLINE_KEEP)


if __name__ == "__main__":
    main()
