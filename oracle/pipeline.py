"""ORACLE (test infrastructure only): straight-line restatement of the reference's sampling procedure,
/root/reference/aether/pipelines/aetherv1_pipeline_cogvideox.py:690-965 (`__call__`) and P:514-688 (`prepare_latents`),
for inputs that are ALREADY preprocessed tensors.  One function, no helpers, every statement in reference order, so
the (re-structured) product pipeline can be compared with it output-for-output on the same modules and seed.
The orchestration it restates IS pinned: the product pipeline, which equals this restatement bit for bit (tests/test_pipeline_cpu.py),
also equals the reference's own pipeline class executed against a stub diffusers (tests/test_reference_pins_cpu.py).  The three modules
it drives (oracle/dit.py, oracle/vae.py, the scheduler) remain PARITY UNPINNED (see oracle/__init__.py)."""
from __future__ import annotations

import math

import torch
from einops import rearrange


@torch.no_grad()
def sample(task, transformer, vae, scheduler, prompt_embeds, *, image=None, goal=None, video=None, raymap=None, height, width,
           num_frames, num_inference_steps=None, guidance_scale=None, use_dynamic_cfg=False, generator=None, fps=12,
           rope=None, dtype=torch.bfloat16, device="cpu", compute_dtype=None, trace=None, vae_device=None, video_latents=None):
    """image/goal: [1,3,H,W] in [-1,1]; video: [F,3,H,W]; raymap: [1,F,6,h,w]. Returns (rgb, disparity, raymap) tensors.
    compute_dtype (calibration only): run the three modules in this dtype (e.g. fp32 weights) while every random draw
    and every inter-module tensor keeps the reference dtype `dtype`, so runs at different precision see the SAME noise.
    trace (fixture generation only): a dict that receives the intermediates a full-size fixture records — the video posterior,
    `condition_latents`, every step's noise prediction, the final latents and the raw decoder outputs.
    device / vae_device (fixture generation on an accelerator, tools/make_fullsize_golden_gpu.py): where the transformer / the VAE live; every random
    draw is still made on the generator's device (CPU) and moved, as `randn_tensor` does (P:683), so the noise is the same on every device.
    video_latents (same tool): the sampled + scaled video latents [1, f, 16, h, w] of an EARLIER run of this function on the same video and seed
    (trace["condition_latents"][:, :, :16]); the posterior draw is still made, so the generator stays aligned, but the 41-frame encode is skipped."""
    cd = compute_dtype or dtype
    dev = torch.device(device)
    vdev = torch.device(vae_device) if vae_device is not None else dev
    on_cpu = dev.type == "cpu" and vdev.type == "cpu"
    defaults_steps = {"reconstruction": 4, "prediction": 50, "planning": 50}                        # P:257-261
    defaults_g = {"reconstruction": 1.0, "prediction": 3.0, "planning": 3.0}                        # P:262-266
    defaults_dyn = {"reconstruction": False, "prediction": True, "planning": True}                  # P:267-271
    num_inference_steps = num_inference_steps or defaults_steps[task]
    guidance_scale = guidance_scale or defaults_g[task]
    use_dynamic_cfg = use_dynamic_cfg or defaults_dyn[task]
    do_cfg = guidance_scale > 1.0                                                                     # P:777
    scheduler.set_timesteps(num_inference_steps, device=device)                                      # P:780-783
    timesteps = scheduler.timesteps
    sf = vae.config.scaling_factor
    lat_frames = (num_frames - 1) // 4 + 1                                                            # P:535
    shape = (1, lat_frames, 56, height // 8, width // 8)                                              # P:536-542

    def enc(x):                                                                                       # P:557-576
        dist = vae.encode(x.to(vdev, cd)).latent_dist
        if trace is not None:
            trace.setdefault("posterior", []).append((dist.mean.float().clone(), dist.logvar.float().clone()))
        if cd == dtype and on_cpu:
            z = dist.sample(generator)
        else:   # same bf16 noise as the reference-dtype run, higher-precision mean/std
            z = dist.mean + dist.std * torch.randn(dist.mean.shape, generator=generator, dtype=dtype).to(vdev, cd)
        return (sf * z.to(dtype).permute(0, 2, 1, 3, 4)).to(dev)

    if image is not None:
        image_latents = enc(image.to(dtype).unsqueeze(2))
    if goal is not None:
        goal_latents = enc(goal.to(dtype).unsqueeze(2))
    if video is not None and video_latents is not None:
        torch.randn((1, 16, lat_frames, height // 8, width // 8), generator=generator, dtype=dtype)    # the posterior draw of `enc`, discarded
        video_latents = video_latents.to(dev, dtype)
    elif video is not None:
        video_latents = enc(video.to(dtype).unsqueeze(0).permute(0, 2, 1, 3, 4))
    if image is not None and goal is None:                                                            # P:633-640
        pad = torch.zeros(1, lat_frames - 1, *image_latents.shape[2:], dtype=dtype, device=dev)
        cond = torch.cat([image_latents, pad], dim=1)
    elif goal is not None:                                                                            # P:641-648
        pad = torch.zeros(1, lat_frames - 2, *image_latents.shape[2:], dtype=dtype, device=dev)
        cond = torch.cat([image_latents, pad, goal_latents], dim=1)
    else:
        cond = video_latents
    if raymap is not None:                                                                            # P:652-670
        raymap = raymap.to(dev, dtype)
        if raymap.shape[1] % 4 != 0:
            raymap = torch.cat([raymap[:, : 4 - raymap.shape[1] % 4], raymap], dim=1)
        cam = rearrange(raymap, "b (n t) c h w -> b t (n c) h w", n=4)
    else:
        cam = torch.zeros(1, lat_frames, 24, height // 8, width // 8, dtype=dtype, device=dev)                    # P:672-680
    cond = torch.cat([cond, cam], dim=2)                                                              # P:682
    latents = torch.randn(shape, generator=generator, dtype=dtype).to(dev) * scheduler.init_noise_sigma       # P:683-686
    if trace is not None:
        trace["condition_latents"], trace["initial_latents"], trace["noise_pred"] = cond.clone(), latents.clone(), []
        on_step = trace.get("on_step")

    old_x0 = None
    g_now = guidance_scale
    for i, t in enumerate(timesteps):                                                                 # P:827
        lat_in = torch.cat([latents] * 2) if do_cfg else latents                                      # P:832-834
        if do_cfg:                                                                                    # P:839-855
            un = cond.clone()
            if task == "planning":
                un[:, :, :16] = 0
            elif task == "prediction":
                un[:, :1, :16] = 0
            else:
                raise ValueError(f"Task {task} not supported for classifier-free guidance.")
            c_in = torch.cat([un, cond])
        else:
            c_in = cond
        lat_in = torch.cat([lat_in, c_in], dim=2)                                                     # P:857-859
        pred = transformer(hidden_states=lat_in.to(cd), encoder_hidden_states=prompt_embeds.repeat(lat_in.shape[0], 1, 1).to(dev, cd),
                           timestep=t.expand(lat_in.shape[0]).to(dev), ofs=None, image_rotary_emb=rope, return_dict=False)[0].float()
        if trace is not None:
            trace["noise_pred"].append(pred.clone())
        if use_dynamic_cfg:                                                                           # P:879-893
            g_now = 1 + guidance_scale * ((1 - math.cos(math.pi * ((num_inference_steps - t.item()) / num_inference_steps) ** 5.0)) / 2)
        if do_cfg:                                                                                    # P:895-899
            u, c = pred.chunk(2)
            pred = u + g_now * (c - u)
        latents, old_x0 = scheduler.step(pred, old_x0, t, timesteps[i - 1] if i > 0 else None, latents,
                                         generator=generator, return_dict=False)                      # P:907-915
        latents = latents.to(dtype)                                                                   # P:916
        if trace is not None and on_step is not None:
            on_step(i, latents)
    if trace is not None:
        trace["final_latents"] = latents.clone()

    def dec(z):                                                                                       # decode_latents
        out = vae.decode((1 / sf * z.permute(0, 2, 1, 3, 4)).to(vdev, cd)).sample
        if trace is not None:
            trace.setdefault("decoded", []).append(out.float().clone())
        return out.to(dev, dtype)

    rgb = dec(latents[:, :, :16])                                                                     # P:925-934
    rgb = (rgb[0].permute(1, 0, 2, 3) / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).float()
    disp = dec(latents[:, :, 16:32]).mean(dim=1)                                                      # P:936-940
    disp = torch.square(disp * 0.5 + 0.5).float()[0]
    rm = rearrange(latents[:, :, 32:], "b t (n c) h w -> b (n t) c h w", n=4)[:, -rgb.shape[0]:].float()[0]   # P:942-949
    return rgb, disp, rm
