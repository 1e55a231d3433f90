"""The fp32 ORACLE transformer executed with plain torch on the MI355X (fp32 GEMMs, explicit fp32 soft-max attention — NOT this repo's kernels)
to write the long guided fixtures in minutes instead of 10.5 h of host CPU per task:

    gpurun -- python tools/make_fullsize_golden_gpu.py pin prediction50 planning50 calib

  pin           re-computes what two CPU-generated fixtures hold — tests/golden/fullsize_dit.npz (one 42-block forward) and the step-0 B = 2 noise
                prediction + 2-step final latents of tests/golden/fullsize_prediction.npz — and reports the distance device-oracle <-> CPU-oracle
                (fp32 round-off of a different summation order through 42 blocks: measured 2.1e-4 rel-L2 on the forward, profiles/r05_gpu_oracle_pin.json — fifty
                times below the bf16 distances the parity tests measure).
                The CPU run of the SAME 50-step call (tools/make_fullsize_golden.py prediction50, checkpointing) pins the first steps of the long
                trajectories the same way (tools/compare_partial_fixture.py).
  prediction50  BASELINE configs[2] at the quoted step count: car.png + forward-right raymap, 50 guided steps, dynamic CFG on the n = 50 schedule
  planning50    BASELINE configs[3]: 01_obs.png + 01_goal.png, 50 guided steps
                (P:690-965, P:827-921, P:880-899).  The VAE (single-frame encodes) runs on the host CPU exactly as in the CPU generator; the two final
                decodes are NOT run here: the fixture stores the final latents (exact bf16 bits) and `tools/make_fullsize_golden.py decode50 <task>`
                decodes them with the fp32 CPU oracle VAE in the build container.
  calib         the same two calls with the oracle in the REFERENCE dtype (bf16 weights / activations, torch's own bf16 kernels on this device): its
                distance to the fp32 fixture along the trajectory is what the reference dtype itself costs over 50 guided steps — the calibration
                of the test bounds (profiles/r05_bf16_oracle_calibration_guided50.json).

Writes into gpurun_out/fixtures/ (merged back by gpurun); copy the .npz files into tests/golden/.  Test infrastructure: imports oracle/.
"""
from __future__ import annotations

import json
import math
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fullsize_cases as fc  # noqa: E402

OUT = os.path.join(fc.ROOT, "gpurun_out", "fixtures")
DEV = torch.device(os.environ.get("AETHER_ORACLE_DEVICE", "cuda:0"))      # "cpu" only for the plumbing dry run


def log(msg):
    print(f"[{time.strftime('%H:%M:%S')}] {msg}", flush=True)


def _explicit_attention(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False):
    """soft-max(q k^T / sqrt(d)) v written out, 8 heads at a time ([B, 8, S, S] fp32 scores = 7.5 GB at S = 15 302): fp32 GEMMs, fp32 soft-max."""
    assert attn_mask is None and not is_causal and dropout_p == 0.0
    if q.dtype != torch.float32:
        return _sdpa(q, k, v)                                            # reference dtype: torch's own kernel, what diffusers would call
    out = torch.empty_like(q)
    scale = 1.0 / math.sqrt(q.shape[-1])
    for h in range(0, q.shape[1], 8):
        s = torch.matmul(q[:, h:h + 8] * scale, k[:, h:h + 8].transpose(-1, -2))
        s = torch.softmax(s, dim=-1)
        out[:, h:h + 8] = torch.matmul(s, v[:, h:h + 8])
        del s
    return out


_sdpa = F.scaled_dot_product_attention


def _sync():
    if DEV.type == "cuda":
        torch.cuda.synchronize()


def _devname():
    return torch.cuda.get_device_name(0) if DEV.type == "cuda" else "cpu (dry run)"


def setup():
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    torch.set_float32_matmul_precision("highest")
    import oracle.dit as od
    od.F.scaled_dot_product_attention = _explicit_attention            # oracle.dit's `F` IS torch.nn.functional: restored by nobody, this process only runs the oracle
    os.makedirs(OUT, exist_ok=True)


def _rope():
    r = fc.rope_tables()
    return r[0].to(DEV), r[1].to(DEV)


def stage_pin(dit):
    res = {}
    hidden, text, t = fc.dit_inputs()
    _sync()
    t0 = time.perf_counter()
    with torch.no_grad():
        out = dit(hidden.float().to(DEV), text.float().to(DEV), t.to(DEV), image_rotary_emb=_rope())[0]
    _sync()
    dt = time.perf_counter() - t0
    ref = torch.from_numpy(np.load(os.path.join(fc.GOLDEN_DIR, "fullsize_dit.npz"))["out"].astype(np.float32))
    # the CPU fixture is stored as float16: compare at that resolution too
    got = out.cpu()
    res["dit_forward_b1"] = {"seconds_device": dt, **fc.metrics(got, ref), "vs_fixture_rounded_to_f16": fc.metrics(got.half().float(), ref)}
    log(f"pin: B = 1 forward {dt:.2f} s on the device: {json.dumps(res['dit_forward_b1'])}")
    # ---- step 0 of the guided prediction fixture (B = 2) + its 2-step final latents -------------------------------------------------------
    from calibrate_fullsize_bf16 import guided_forward_inputs
    model_in, ref2 = guided_forward_inputs("prediction")
    t0 = time.perf_counter()
    with torch.no_grad():
        o2 = dit(model_in.float().to(DEV), fc.prompt_embeds().repeat(2, 1, 1).float().to(DEV), torch.tensor([999, 999], device=DEV), image_rotary_emb=_rope())[0]
    _sync()
    dt = time.perf_counter() - t0
    o2 = o2.cpu()[..., ::2, ::2]
    res["prediction_step0_b2"] = {"seconds_device": dt, "unconditional": fc.metrics(o2[0], ref2[0]), "conditional": fc.metrics(o2[1], ref2[1])}
    log(f"pin: B = 2 forward {dt:.2f} s: {json.dumps(res['prediction_step0_b2'])}")
    z = np.load(os.path.join(fc.GOLDEN_DIR, "fullsize_prediction.npz"))
    tr = run_guided(dit, "prediction", fc.GUIDED_STEPS, set(range(fc.GUIDED_STEPS)), torch.float32)
    fin = fc.from_bf16_bits(z["final_latents_bits"]).float()
    got = tr["final_latents"].cpu().float()
    m = fc.metrics(got, fin)
    m["bf16_values_that_differ"] = int((got != fin).sum())
    m["of"] = fin.numel()
    res["prediction_2_steps_final_latents"] = m
    log(f"pin: 2-step guided final latents (bf16 values) device-oracle vs CPU-oracle: {json.dumps(m)}")
    with open(os.path.join(OUT, "gpu_oracle_pin.json"), "w") as f:
        json.dump({"case": "fp32 oracle (oracle/dit.py) run with torch on MI355X vs the fixtures the same oracle wrote on the host CPU", **res}, f, indent=1)


class _HostVAE:
    """The VAE stays on the host CPU (single-frame encodes: seconds).  Decodes are done later, in the build container, from the stored latents.
    dtype = bfloat16: the encoder runs in the reference dtype (bf16 weights — exact, they are bf16-representable — and bf16 activations)."""
    def __init__(self, vae, dtype=torch.float32):
        self.vae, self.config, self.dtype = (vae if dtype == torch.float32 else vae.to(dtype)), vae.config, dtype

    def encode(self, x):
        return self.vae.encode(x.to(self.dtype))

    def decode(self, z):
        import types
        return types.SimpleNamespace(sample=torch.zeros(z.shape[0], 3, (z.shape[2] - 1) * 4 + 1, z.shape[3] * 8, z.shape[4] * 8, dtype=z.dtype))


_VAE = {}


def host_vae(dtype=torch.float32):
    if dtype not in _VAE:
        _VAE[dtype] = _HostVAE(fc.build_oracle_vae(), dtype)
    return _VAE[dtype]


def run_guided(dit, task, steps, keep, compute_dtype, vae_dtype=torch.float32):
    from aether_amd.scheduler import CogVideoXDPMScheduler
    from oracle.pipeline import sample
    case = fc.GUIDED_CASES[task]
    image = fc.image_as_model_input(fc.named_image(case["image"]))
    goal = fc.image_as_model_input(fc.named_image(case["goal"])) if case["goal"] else None
    raymap = torch.from_numpy(fc.forward_right_raymap())[None] if case["raymap"] else None
    step_lat, times, mark, rms, mx = {}, [], [time.perf_counter()], [], []
    trace = {}

    def on_step(i, latents):
        _sync()
        now = time.perf_counter()
        times.append(now - mark[0])
        mark[0] = now
        for p in trace["noise_pred"]:
            rms.append([float(p[b].pow(2).mean().sqrt()) for b in range(p.shape[0])])
            mx.append(float(p.abs().max()))
        trace["noise_pred"].clear()
        if i in keep:
            step_lat[i] = fc.bf16_bits(latents[:, :, :, ::6, ::6].cpu())
        if i % 10 == 0 or i == steps - 1:
            log(f"{task}{steps} [{compute_dtype}]: step {i} done ({times[-1]:.2f} s)")

    trace["on_step"] = on_step
    # the fp32 host VAE encodes the observation in BOTH precisions: the calibration varies the dtype of the 2 x `steps` transformer calls only, on
    # identical condition latents.  `sample(compute_dtype=float32)` hands fp32 tensors to the transformer; for the reference dtype the wrapper casts
    # them to bf16, so the transformer runs bf16 in / bf16 weights / bf16 out.

    class AsBf16:
        def __call__(self, hidden_states, encoder_hidden_states, timestep, **kw):
            return (dit(hidden_states=hidden_states.to(torch.bfloat16), encoder_hidden_states=encoder_hidden_states.to(torch.bfloat16), timestep=timestep, **kw)[0],)

    t0 = time.perf_counter()
    sample(task, dit if compute_dtype == torch.float32 else AsBf16(), host_vae(vae_dtype), CogVideoXDPMScheduler(), fc.prompt_embeds(), image=image, goal=goal,
           raymap=raymap, height=fc.HEIGHT, width=fc.WIDTH, num_frames=fc.FRAMES, num_inference_steps=steps,
           generator=torch.Generator().manual_seed(fc.GUIDED_SEED), rope=_rope(), compute_dtype=torch.float32, trace=trace, device=DEV, vae_device="cpu")
    trace.update(step_lat=step_lat, step_seconds=times, noise_pred_rms=rms, noise_pred_max=mx, seconds_total=time.perf_counter() - t0)
    return trace


def guidance_scales(steps):
    from aether_amd.scheduler import CogVideoXDPMScheduler
    s = CogVideoXDPMScheduler()
    s.set_timesteps(steps)
    ts = [int(t) for t in s.timesteps]
    return ts, [1 + 3.0 * ((1 - math.cos(math.pi * ((steps - t) / steps) ** 5.0)) / 2) for t in ts]         # P:886-893


def stage_long(dit, task, steps, name):
    keep = set(fc.GUIDED_LONG_KEEP) if steps == fc.HEADLINE_STEPS else set(range(steps))
    tr = run_guided(dit, task, steps, keep, torch.float32)
    ts, scales = guidance_scales(steps)
    kept = sorted(tr["step_lat"])
    cond = tr["condition_latents"].cpu()
    meta = dict(task=task, steps=steps, steps_done=steps, timesteps=ts, guidance_scales=scales, kept_steps=kept, inputs=fc.GUIDED_CASES[task], seed=fc.GUIDED_SEED,
                dit_seed=fc.DIT_SEED, vae_seed=fc.VAE_SEED, step_seconds=tr["step_seconds"], seconds_total=tr["seconds_total"], torch=torch.__version__,
                generated_on=f"fp32 oracle transformer with torch on {_devname()} (tools/make_fullsize_golden_gpu.py), VAE encodes on the host CPU",
                noise_pred_rms=tr["noise_pred_rms"], noise_pred_max=tr["noise_pred_max"], condition_sum=float(cond.double().sum()),
                condition_abs_sum=float(cond.double().abs().sum()), decoded=False)
    np.savez_compressed(os.path.join(OUT, name), step_latents_s6=np.stack([tr["step_lat"][k] for k in kept]), final_latents_bits=fc.bf16_bits(tr["final_latents"].cpu()),
                        initial_latents_sum=np.float64(tr["initial_latents"].double().sum().item()), meta=json.dumps(meta))
    log(f"{task}{steps}: {tr['seconds_total']:.0f} s on the device; wrote gpurun_out/fixtures/{name}")
    return tr


def stage_calib(dit32, runs):
    """bf16 oracle along the same trajectories vs the fp32 results of `runs` ({task: trace})."""
    dit = dit32.to(torch.bfloat16)                                        # in place: bf16-representable weights, exact; run this stage LAST
    res = {}
    for task, tr32 in runs.items():
        steps = len(tr32["step_seconds"])
        tr = run_guided(dit, task, steps, set(tr32["step_lat"]), torch.bfloat16)
        per = {int(k): fc.metrics(fc.from_bf16_bits(tr["step_lat"][k]).float(), fc.from_bf16_bits(tr32["step_lat"][k]).float()) for k in sorted(tr["step_lat"])}
        fin = fc.metrics(tr["final_latents"].cpu().float(), tr32["final_latents"].cpu().float())
        res[task] = {"steps": steps, "seconds_device": tr["seconds_total"], "per_step_rel_l2": {k: v["rel_l2"] for k, v in per.items()},
                     "per_step_linf_rel": {k: v["linf_rel"] for k, v in per.items()}, "final_latents": fin}
        log(f"calib {task}: bf16 oracle vs fp32 oracle after {steps} guided steps: {json.dumps(fin)}")
        np.savez_compressed(os.path.join(OUT, f"bf16_oracle_{task}{steps}_final_latents.npz"), final_latents_bits=fc.bf16_bits(tr["final_latents"].cpu()))
        with open(os.path.join(OUT, "bf16_oracle_calibration_guided50.json"), "w") as f:
            json.dump({"case": "the ORACLE transformer in the reference dtype (bf16, torch's own kernels on MI355X) vs the same oracle in fp32, whole guided "
                               "trajectories on the named inputs, dynamic CFG, identical noise and condition latents", **res}, f, indent=1)


def stage_recon(dit, steps, name, keep, compute_dtype=torch.float32):
    """The RECONSTRUCTION call of tools/make_fullsize_golden.py (stage_clip / stage_traj: same clip, seed CLIP_SEED, same posterior sample — the sampled
    video latents come from tests/golden/fullsize_clip_condition.npz, written by the fp32 CPU oracle VAE in the build container) with the fp32 oracle
    transformer on the device: B = 1, no guidance, `steps` steps.  Same oracle code as the CPU fixtures; the difference is where torch executes the
    scheduler's element-wise update: a CUDA / HIP device keeps the python scalars of `m1 * sample` and `m_noise * noise` in fp32 (what the reference,
    which runs on the device, does: D:218), torch-CPU rounds them to bf16 first — a per-step perturbation of up to 2^-9 that is in every CPU-generated
    trajectory fixture and in neither the reference nor the native path."""
    from aether_amd.scheduler import CogVideoXDPMScheduler
    from oracle.pipeline import sample
    zc = np.load(os.path.join(fc.GOLDEN_DIR, "fullsize_clip_condition.npz"))
    video_latents = fc.from_bf16_bits(zc["video_latents_bits"])
    v = fc.video_as_model_input(fc.clip_video())
    step_lat, times, mark = {}, [], [time.perf_counter()]
    trace = {}

    vae = host_vae()                                                  # only its config and the (unused here) decode stub are touched

    def on_step(i, latents):
        _sync()
        now = time.perf_counter()
        times.append(now - mark[0])
        mark[0] = now
        trace["noise_pred"].clear()
        if i in keep:
            step_lat[i] = fc.bf16_bits(latents[:, :, :, ::6, ::6].cpu())

    trace["on_step"] = on_step

    class AsBf16:                                                      # the reference dtype: bf16 in, bf16 weights, bf16 out (see run_guided)
        def __call__(self, hidden_states, encoder_hidden_states, timestep, **kw):
            return (dit(hidden_states=hidden_states.to(torch.bfloat16), encoder_hidden_states=encoder_hidden_states.to(torch.bfloat16), timestep=timestep, **kw)[0],)

    t0 = time.perf_counter()
    sample("reconstruction", dit if compute_dtype == torch.float32 else AsBf16(), vae, CogVideoXDPMScheduler(), fc.prompt_embeds(), video=v, height=fc.HEIGHT, width=fc.WIDTH, num_frames=fc.FRAMES,
           num_inference_steps=steps, generator=torch.Generator().manual_seed(fc.CLIP_SEED), rope=_rope(), compute_dtype=torch.float32, trace=trace,
           device=DEV, vae_device="cpu", video_latents=video_latents)
    total = time.perf_counter() - t0
    kept = sorted(step_lat)
    if name is None:                                                   # calibration run: hand the trajectory back, write nothing
        return dict(step_lat=step_lat, final_latents=trace["final_latents"].cpu(), seconds_total=total)
    meta = dict(task="reconstruction", steps=steps, kept_steps=kept, clip_seed=fc.CLIP_SEED, dit_seed=fc.DIT_SEED, vae_seed=fc.VAE_SEED, step_seconds=times,
                seconds_total=total, torch=torch.__version__, decoded=False,
                generated_on=f"fp32 oracle transformer with torch on {_devname()} (tools/make_fullsize_golden_gpu.py); video latents from the fp32 CPU oracle VAE")
    np.savez_compressed(os.path.join(OUT, name), step_latents_s6=np.stack([step_lat[k] for k in kept]), final_latents_bits=fc.bf16_bits(trace["final_latents"].cpu()),
                        meta=json.dumps(meta))
    log(f"reconstruction, {steps} steps: {total:.0f} s on the device; wrote gpurun_out/fixtures/{name}")


def stage_windows3(dit):
    """BASELINE configs[4] END TO END at its own geometry, oracle side: a 72-frame clip cut into three 41-frame windows with starts [0, 24, 31]
    (overlaps 17 and 34: the two overlap lengths of the reference's [0, 24, ..., 144, 151]); every window is an independent reconstruction call
    (D:613-631: fresh generator, same seed) of the fp32 oracle — transformer AND VAE executed by torch on this device (fp32 GEMMs / fp32 convolutions,
    explicit fp32 attention), 4 steps (the reference default, P:257-261) — and the three outputs are merged by the host merge, which the REFERENCE's own
    blend pins at this geometry (tests/golden/blend_fullsize.npz), with the CLI's default Kalman smoothing (D:173-179).  Stored: per-window final latents
    (exact bf16 bits, every 2nd latent pixel), the merged rgb / disparity / point maps on a pixel lattice, all 72 poses, the fitted disparity scales."""
    from aether_amd import geometry as G
    from aether_amd.scheduler import CogVideoXDPMScheduler
    from aether_amd.windows import WindowResult, blend_and_merge_window_results
    from oracle.pipeline import sample
    starts = fc.windows3_starts()
    total = starts[-1] + fc.FRAMES
    video = fc.long_video(total)
    vae = fc.build_oracle_vae().to(DEV)
    t0 = time.perf_counter()
    wins, finals, secs = [], [], []
    for s0 in starts:
        t1 = time.perf_counter()
        trace = {}
        rgb, disp, rm = sample("reconstruction", dit, vae, CogVideoXDPMScheduler(), fc.prompt_embeds(), video=fc.video_as_model_input(video[s0:s0 + fc.FRAMES]),
                               height=fc.HEIGHT, width=fc.WIDTH, num_frames=fc.FRAMES, num_inference_steps=fc.WINDOWS3_STEPS,
                               generator=torch.Generator().manual_seed(fc.CLIP_SEED), rope=_rope(), compute_dtype=torch.float32, trace=trace, device=DEV, vae_device=DEV)
        _sync()
        wins.append(WindowResult(s0, rgb.cpu().numpy(), disp.cpu().numpy(), rm.cpu().numpy()))
        finals.append(fc.bf16_bits(trace["final_latents"].cpu()))
        secs.append(time.perf_counter() - t1)
        log(f"windows3: window at {s0}: {secs[-1]:.0f} s (encode + {fc.WINDOWS3_STEPS} steps + 2 decodes, fp32 oracle on the device)")
        del trace
    scales, real = [], G.compute_scale

    def recording(*a, **k):
        scales.append(real(*a, **k))
        return scales[-1]
    G.compute_scale = recording
    try:
        m_rgb, m_disp, m_poses, m_pm = blend_and_merge_window_results([WindowResult(w.start, w.rgb, w.disparity, w.raymap.copy()) for w in wins], height=fc.HEIGHT,
                                                                      width=fc.WIDTH, smooth_camera=True, smooth_method="kalman")
    finally:
        G.compute_scale = real
    s = fc.DEC_STRIDE
    meta = dict(task="reconstruction, sliding windows", starts=starts, total_frames=total, steps=fc.WINDOWS3_STEPS, seed=fc.CLIP_SEED, window_seconds=secs,
                seconds_total=time.perf_counter() - t0, torch=torch.__version__, video_sum=float(video.astype(np.float64).sum()),
                generated_on=f"fp32 oracle transformer + fp32 oracle VAE with torch on {_devname()} (tools/make_fullsize_golden_gpu.py windows3); host merge, kalman smoothing")
    # per-window final latents on every 2nd latent pixel (exact bf16 bits), merged arrays on a pixel lattice: rgb / disparity every 8th, point maps every 16th
    np.savez_compressed(os.path.join(OUT, "fullsize_windows3.npz"), final_latents_s2_bits=np.stack(finals)[..., ::2, ::2], rgb_s8=m_rgb[:, ::s, ::s].astype(np.float16),
                        disparity_s8=m_disp[:, ::s, ::s].astype(np.float32), pointmaps_s16=m_pm[:, ::2 * s, ::2 * s].astype(np.float32), poses=m_poses,
                        scales=np.array(scales), meta=json.dumps(meta))
    log(f"windows3: scales {scales}; wrote gpurun_out/fixtures/fullsize_windows3.npz in {meta['seconds_total']:.0f} s")


def stage_dit17(dit):
    """One fp32 oracle forward at the SHORTEST clip the reference admits (17 frames -> 5 latent frames, S = 226 + 6 750), B = 1: the geometry of
    tests/test_fullsize_guided_gpu.py::test_seventeen_frame_clip_full_size, which had no oracle."""
    from oracle.rope import prepare_rope
    lat_f = (17 - 1) // 4 + 1 if fc.FRAMES == 41 else 3
    frames = (lat_f - 1) * 4 + 1
    g = torch.Generator().manual_seed(fc.DIT_INPUT_SEED + 17)
    hidden = torch.randn(1, lat_f, 96, fc.LAT_H, fc.LAT_W, generator=g).to(torch.bfloat16)
    text = (torch.randn(1, fc.TEXT_LEN, fc.TEXT_DIM, generator=g) * 0.1).to(torch.bfloat16)
    t = torch.tensor([749], dtype=torch.int64)
    rope = prepare_rope(fc.HEIGHT, fc.WIDTH, lat_f, 12)
    _sync()
    t0 = time.perf_counter()
    with torch.no_grad():
        out = dit(hidden_states=hidden.to(DEV, torch.float32), encoder_hidden_states=text.to(DEV, torch.float32), timestep=t.to(DEV), ofs=None,
                  image_rotary_emb=(rope[0].to(DEV), rope[1].to(DEV)), return_dict=False)[0]
    _sync()
    np.savez_compressed(os.path.join(OUT, "fullsize_dit17.npz"), out=out.float().cpu().numpy().astype(np.float32),
                        meta=json.dumps(dict(frames=frames, latent_frames=lat_f, timestep=749, input_seed=fc.DIT_INPUT_SEED + 17, input_sum=float(hidden.float().sum()),
                                             seconds=time.perf_counter() - t0, generated_on=f"fp32 oracle transformer with torch on {_devname()}")))
    log(f"dit17: one {frames}-frame forward in {time.perf_counter() - t0:.1f} s; wrote gpurun_out/fixtures/fullsize_dit17.npz")


def stage_calib_recon(dit32, step_counts):
    """The reconstruction trajectories with the oracle transformer in the REFERENCE dtype (bf16) against the committed device-semantics fixtures
    (tests/golden/fullsize_recon<n>_device.npz): what bf16 itself costs along the headline configuration's 50 steps."""
    dit = dit32 if next(dit32.parameters()).dtype == torch.bfloat16 else dit32.to(torch.bfloat16)
    res = {}
    for n in step_counts:
        z = np.load(os.path.join(fc.GOLDEN_DIR, f"fullsize_recon{n}_device.npz"))
        kept = list(json.loads(str(z["meta"]))["kept_steps"])
        tr = stage_recon(dit, n, None, set(kept), torch.bfloat16)
        per = {int(i): fc.metrics(fc.from_bf16_bits(tr["step_lat"][i]).float(), fc.from_bf16_bits(z["step_latents_s6"][k]).float()) for k, i in enumerate(kept)}
        fin = fc.metrics(tr["final_latents"].float(), fc.from_bf16_bits(z["final_latents_bits"]).float())
        res[f"reconstruction_{n}_steps"] = {"seconds_device": tr["seconds_total"], "per_step_rel_l2": {k: v["rel_l2"] for k, v in per.items()}, "final_latents": fin}
        log(f"calib_recon {n}: bf16 oracle vs fp32 device-semantics fixture: {json.dumps(fin)}")
        np.savez_compressed(os.path.join(OUT, f"bf16_oracle_recon{n}_final_latents.npz"), final_latents_bits=fc.bf16_bits(tr["final_latents"]))
        with open(os.path.join(OUT, "bf16_oracle_calibration_recon.json"), "w") as f:
            json.dump({"case": "the ORACLE transformer in the reference dtype (bf16, torch's own kernels on MI355X) vs the same oracle in fp32 (device semantics), reconstruction "
                               "trajectories of the full-size clip, identical noise and video latents", **res}, f, indent=1)


def stage_calib_full(dit32, tasks):
    """The WHOLE guided call in the reference dtype — bf16 VAE encode of the observation (host CPU, torch's bf16 kernels) AND bf16 transformer — against
    the committed fp32 fixtures (tests/golden/fullsize_<task>50.npz): what `D:218,226` + `P:297` (everything loaded and run as bf16) costs end to end
    over 50 guided steps.  This is the number the native path's full-call distance is comparable with (its VAE encode is bf16 too)."""
    dit = dit32 if next(dit32.parameters()).dtype == torch.bfloat16 else dit32.to(torch.bfloat16)
    res = {}
    for task in tasks:
        z = np.load(os.path.join(fc.GOLDEN_DIR, f"fullsize_{task}50.npz"))
        meta = json.loads(str(z["meta"]))
        kept, steps = list(meta["kept_steps"]), int(meta["steps"])
        tr = run_guided(dit, task, steps, set(kept), torch.bfloat16, vae_dtype=torch.bfloat16)
        per = {int(i): fc.metrics(fc.from_bf16_bits(tr["step_lat"][i]).float(), fc.from_bf16_bits(z["step_latents_s6"][k]).float()) for k, i in enumerate(kept)}
        fin = fc.metrics(tr["final_latents"].cpu().float(), fc.from_bf16_bits(z["final_latents_bits"]).float())
        cond = tr["condition_latents"].cpu()
        res[task] = {"steps": steps, "seconds_device": tr["seconds_total"], "per_step_rel_l2": {k: v["rel_l2"] for k, v in per.items()},
                     "per_step_linf_rel": {k: v["linf_rel"] for k, v in per.items()}, "final_latents": fin,
                     "condition_sum_bf16_vae": float(cond.double().sum()), "condition_sum_fp32_vae": meta["condition_sum"]}
        log(f"calib_full {task}: all-bf16 oracle vs fp32 fixture after {steps} guided steps: {json.dumps(fin)}")
        np.savez_compressed(os.path.join(OUT, f"bf16_full_oracle_{task}{steps}_final_latents.npz"), final_latents_bits=fc.bf16_bits(tr["final_latents"].cpu()))
        with open(os.path.join(OUT, "bf16_full_oracle_calibration_guided50.json"), "w") as f:
            json.dump({"case": "the ORACLE in the reference dtype END TO END (bf16 VAE encode of the observation on the host CPU + bf16 transformer with torch's kernels on "
                               "MI355X) vs the fp32 fixtures, whole guided trajectories on the named inputs, dynamic CFG, identical noise", **res}, f, indent=1)


def main():
    """Stages run in the order given; each long stage is skipped (and says so) when its estimate does not fit into what is left of
    AETHER_ORACLE_BUDGET_S (default 1700 s), so one bounded gpurun lease always ends with whatever was finished on disk."""
    stages = sys.argv[1:]
    budget = float(os.environ.get("AETHER_ORACLE_BUDGET_S", "1700"))
    start = time.perf_counter()
    setup()
    dit, _ = fc.build_oracle_dit()
    log(f"built the fp32 oracle transformer in {time.perf_counter() - start:.0f} s; moving to {_devname()}")
    dit = dit.to(DEV)
    runs, per_step = {}, None
    for st in stages:
        left = budget - (time.perf_counter() - start)
        if st == "pin":
            stage_pin(dit)
            continue
        if st == "calib":
            continue
        if st == "windows3":
            stage_windows3(dit)
            continue
        if st == "dit17":
            stage_dit17(dit)
            continue
        if st.startswith("recon"):                                 # recon4 / recon10 / recon50: the reconstruction fixtures under device semantics
            n = int(st[5:])
            keep = set(fc.HEADLINE_KEEP) if n == fc.HEADLINE_STEPS else set(range(n))
            stage_recon(dit, n, f"fullsize_recon{n}_device.npz", keep)
            continue
        if st == "calib_recon":                                    # converts the transformer to bf16 in place: after every fp32 stage
            stage_calib_recon(dit, [4, 50])
            continue
        if st == "calib_full":                                     # needs the committed fixtures; converts the transformer to bf16 in place: LAST stage
            stage_calib_full(dit, ["prediction", "planning"])
            continue
        task = "prediction" if st.startswith("prediction") else "planning"
        n = int(st[len(task):])
        if per_step is not None and n * per_step > left:
            log(f"SKIPPED {st}: needs ~{n * per_step:.0f} s, {left:.0f} s of the budget left")
            continue
        runs[task] = stage_long(dit, task, n, f"fullsize_{task}{n}.npz")
        per_step = float(np.median(runs[task]["step_seconds"]))
    if "calib" in stages and runs:
        left = budget - (time.perf_counter() - start)
        fits = {}
        for task, tr in runs.items():                               # the reference dtype is assumed to take at most half the fp32 time per step
            need = 0.5 * sum(tr["step_seconds"])
            if need < left:
                fits[task] = tr
                left -= need
            else:
                log(f"SKIPPED calib {task}: needs ~{need:.0f} s, {left:.0f} s of the budget left")
        if fits:
            stage_calib(dit, fits)
    log(f"done in {time.perf_counter() - start:.0f} s")


if __name__ == "__main__":
    main()
