#!/usr/bin/env python
"""Headline benchmark: denoise-steps/s of the AetherV1 sampling loop on a 41-frame 480x720 clip
(latent 11x60x90 -> 226 + 14 850 tokens, 42-layer / 3072-wide DiT, bf16), BASELINE.json configs[1]:
"4D reconstruction, 41x480x720, 50 steps, bf16, 1xMI355X" (B = 1 through the transformer, no CFG).

One "step" = exactly what the reference's loop body does per iteration (P:827-916): concat noisy latents with the
condition latents, one transformer forward, fp32 cast, one CogVideoXDPMScheduler.step (two generator draws), cast
back to bf16 — the element-wise tail as the drop-in pipeline runs it (one HIP kernel, bit-identical to the PyTorch ops).  Inputs (synthetic, seeded) and the random-init weights are resident in HBM before the timed region.
N > 1: one process per GPU, each rank denoises its own independent window (the reference's sliding windows are
independent pipeline calls, scripts/demo.py:613-631) -> weak scaling, no data-path collective.

Prints ONE JSON line on rank 0 (contract in the task statement), with `roofline` for the dominant kernel class
(HIP-event timing recorded on the launch stream inside aether_dit_forward during the timed steps) and `cpu_baseline`
(the fp32 oracle timed on this host's cores on a bounded sample, rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0   # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md:42
HBM_PEAK_GBPS = 8000.0      # spec, MI355X_MICROARCH.md:35


def flops_per_launch(cls: str, B: int, S: int, D: int, FF: int) -> float:
    """ALGORITHMIC flops of one launch of a kernel class (multiply-add = 2)."""
    M = B * S
    return {
        "attention": 4.0 * B * S * S * D,          # QK^T + PV over all heads: 4*S^2*64*H
        "gemm_qkv": 2.0 * M * 3 * D * D,
        "gemm_out": 2.0 * M * D * D,
        "gemm_ff1": 2.0 * M * FF * D,
        "gemm_ff2": 2.0 * M * D * FF,
    }.get(cls, 0.0)


def pmc_traffic(kernel_class: str, calls_per_forward: int = 42):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same command
    (tools/profile_dit.sh: FETCH_SIZE and WRITE_SIZE in separate passes; read side doubled per the gfx950 correction of
    /opt/skills/guides/MI355X_MICROARCH.md §HBM; the write side is the raw counter).  Only a summary stamped with the digest
    of the CURRENT kernel sources counts (tools/summarize_rocprof.py writes `csrc_sha16`): a profile of older kernels is
    reported as stale -> (None, reason)."""
    import glob

    from aether_amd.build import source_digest
    needle = {"attention": "flash_attn", "gemm_qkv": "gemm_bf16_kernel<2, 4, 4, 2, 0", "gemm_ff1": "gemm_bf16_kernel<2, 4, 4, 2, 1",
              "gemm_ff2": "gemm_bf16_kernel<2, 4, 4, 2, 2", "gemm_out": "gemm_bf16_kernel<2, 4, 4, 2, 2"}.get(kernel_class)
    digest, stale = source_digest(), None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_dit_step*.json")), reverse=True):
        try:
            with open(path) as f:
                doc = json.load(f)
            pmc = doc["pmc"]
        except (OSError, KeyError, ValueError):
            continue
        if doc.get("csrc_sha16") != digest:
            stale = stale or os.path.relpath(path, ROOT)
            continue
        # one logical launch may be several kernels (tail launches): sum them; the PMC passes profile exactly one forward
        tot = 0.0
        for name, e in pmc.items():
            if needle and needle in name and "hbm_read_bytes_per_launch_corrected" in e:
                tot += (e["hbm_read_bytes_per_launch_corrected"] + e.get("hbm_write_bytes_per_launch_raw", 0.0)) * e.get("launches", 1)
        if tot > 0.0:
            # the same summary's kernel trace (rocprofv3 --kernel-trace --stats): time of one LOGICAL launch = total time of the kernels of the class /
            # (forwards traced x logical launches per forward).  A logical launch may be several physical kernels (main + tail launch of a GEMM:
            # `calls` counts those, not launches), so the forwards come from the one kernel that runs exactly once per layer, the attention.
            # ff-down and the out-projection share one instantiation (<2,4,4,2,EPI_BIAS_GATE_RES>): the trace cannot tell them apart -> no profile time.
            stats = doc.get("kernel_stats", [])
            attn_calls = sum(k["calls"] for k in stats if "flash_attn" in k["kernel"])
            forwards = attn_calls / calls_per_forward if attn_calls else 0
            us = [k for k in stats if needle and needle in k["kernel"]]
            prof_us = None
            if forwards and us and kernel_class not in ("gemm_ff2", "gemm_out"):
                prof_us = sum(k["total_ms"] for k in us) * 1e3 / (forwards * calls_per_forward)
            if kernel_class in ("gemm_ff2", "gemm_out"):         # their PMC rows are one row too: not attributable to either class
                return None, os.path.relpath(path, ROOT) + ": ff-down and out-projection share one kernel instantiation (one PMC row)", None
            return tot / calls_per_forward, os.path.relpath(path, ROOT), prof_us
    return None, (f"stale: {stale} was taken from other kernel sources (digest now {digest})" if stale else "no PMC summary committed"), None


def cpu_baseline(cfg_overrides, S_video_shape, seconds_budget=30.0):
    """fp32 oracle (oracle/dit.py) on the host cores: ONE transformer block at the full token count, extrapolated to
    the 42-block forward (blocks are identical; embeddings/final layers are <0.1 % of the flops)."""
    from oracle.dit import Block, DitConfig

    cfg = DitConfig(**cfg_overrides)
    torch.manual_seed(0)
    blk = Block(cfg).float().eval()
    F_, H_, W_ = S_video_shape
    n_vid = F_ * (H_ // 2) * (W_ // 2)
    h = torch.randn(1, n_vid, cfg.inner_dim)
    e = torch.randn(1, cfg.max_text_seq_length, cfg.inner_dim)
    temb = torch.randn(1, cfg.time_embed_dim)
    ang = torch.rand(n_vid, 32)
    rope = (ang.cos().repeat_interleave(2, 1), ang.sin().repeat_interleave(2, 1))
    with torch.no_grad():
        t0 = time.perf_counter()
        blk(h, e, temb, rope)
        dt = time.perf_counter() - t0
    steps_per_s = 1.0 / (dt * cfg.num_layers)
    return {"value": steps_per_s, "unit": "denoise-steps/s", "cores": torch.get_num_threads(), "host_logical_cores": os.cpu_count(),
            "cores_note": "torch's intra-op thread count = the host's physical cores; the SMT siblings are not used",
            "kind": "port", "method": "sampled-extrapolated",
            "sample": f"1 of {cfg.num_layers} DiT blocks at full size (B=1, S={n_vid + cfg.max_text_seq_length}, fp32 torch-CPU "
                      f"oracle, {dt:.1f} s), extrapolated x{cfg.num_layers}; host has {os.cpu_count()} logical cores"}


def offline_cpu_figures():
    """The UN-sampled CPU figures of the same oracle: what the full-size fixture runs took in the build container (8 vCPUs; recorded in the
    fixtures' metadata by tools/make_fullsize_golden.py) — one whole 42-block forward, the whole 4-step reconstruction clip, the guided calls."""
    import numpy as np
    out = {"host": "build container, 8 vCPU (torch-CPU fp32 oracle)"}
    gold = os.path.join(ROOT, "tests", "golden")
    for key, fn, field in (("forward_42_blocks_s", "fullsize_dit.npz", "seconds_cpu"), ("reconstruction_clip_4_steps_s", "fullsize_clip.npz", "seconds_cpu_total"),
                           ("prediction_2_guided_steps_s", "fullsize_prediction.npz", "seconds_cpu_total"),
                           ("planning_2_guided_steps_s", "fullsize_planning.npz", "seconds_cpu_total")):
        try:
            out[key] = round(float(json.loads(str(np.load(os.path.join(gold, fn))["meta"]))[field]), 1)
        except (OSError, KeyError, ValueError):
            pass
    if "forward_42_blocks_s" in out:
        out["denoise_steps_per_s"] = round(1.0 / out["forward_42_blocks_s"], 5)
    return out


def cpu_baseline_vae():
    """fp32 oracle VAE (oracle/vae.py) on the host cores, bounded sample: ONE 240x360 tile x ONE 8-frame chunk of the encoder and
    ONE 30x45 latent tile x ONE 2-latent-frame chunk of the decoder, extrapolated by the tile x chunk count of the 41x480x720
    clip (9 tiles; 5.125 encoder chunks of 8 frames; 5.5 decoder chunks of 2 latent frames).  Baseline only."""
    from oracle.vae import OracleVAE, VaeConfig, init_random_

    vae = init_random_(OracleVAE(VaeConfig()), seed=1).float().eval()
    with torch.no_grad():
        x = torch.randn(1, 3, 8, 240, 360)
        t0 = time.perf_counter()
        vae.encoder(x)
        te = time.perf_counter() - t0
        z = torch.randn(1, 16, 2, 30, 45)
        t0 = time.perf_counter()
        vae.decoder(z)
        td = time.perf_counter() - t0
    return {"encode_s_per_clip": te * 9 * 41 / 8, "decode_s_per_clip": td * 9 * 11 / 2, "cores": torch.get_num_threads(), "kind": "port", "method": "sampled-extrapolated",
            "sample": f"fp32 torch-CPU oracle: one 8x240x360 encoder tile-chunk ({te:.1f} s) x 46.1, one 2x30x45 decoder tile-chunk ({td:.1f} s) x 49.5"}


def mfma_power_cap_leg():
    """What a bare stream of v_mfma_f32_32x32x16_bf16 reaches on THIS box under its power cap (tools/probes/mfma_power_probe.hip,
    built by __graft_entry__.build(); 2 x 25 ms): random operands like the bench's data, and all-zero operands (the datasheet peak
    `roofline.peak` is priced against).  Context for `roofline.frac`, not a replacement for it; None if the probe is not built."""
    import ctypes
    so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "probes", "mfma_power_probe.so")
    if not os.path.exists(so):
        return None
    lib = ctypes.CDLL(so)
    lib.run_mfma_probe.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    dev = torch.device("cuda", torch.cuda.current_device())
    out = torch.zeros(1024, dtype=torch.float32, device=dev)
    sink = torch.zeros(4, dtype=torch.float32, device=dev)
    res = {"kernel": "bare v_mfma_f32_32x32x16_bf16 stream, 128x128 register tile, one wave per SIMD, every CU", "source": "tools/probes/mfma_power_probe.hip"}
    iters = 40000
    for name, data in (("random_operands", torch.randn(32 * 64 * 8, device=dev).bfloat16()), ("zero_operands", torch.zeros(32 * 64 * 8, device=dev).bfloat16())):
        for _ in range(2):
            if lib.run_mfma_probe(0, 4, data.data_ptr(), iters, out.data_ptr(), sink.data_ptr(), 256, None) != 0:
                return None
            torch.cuda.synchronize()
        t = out[:512].view(256, 2).double().cpu()
        cyc, ns = float(t[:, 0].median()), float(t[:, 1].median()) * 10.0
        res[name] = {"tflops": round(256 * 4 * iters * 32 * 32768 / ns / 1e3, 1), "clock_ghz": round(cyc / ns, 3)}
    return res


def make_vae(dev):
    """The ONE VAE of the process (what an application holds): built and its workspace reserved right after the transformer, before the timed
    legs churn the heap — the VAE leg and the clip legs share it, as a pipeline shares its VAE across calls."""
    from aether_amd.vae import AetherVAE

    vae = AetherVAE(device=dev).init_random_weights(1)
    vae.enable_slicing()
    vae.enable_tiling()
    return vae


def vae_leg(dev, vae, reps=3):
    """VAE encode of a 41x480x720 clip and decode of its 11x60x90 latent exactly as the pipeline calls them (tiling + slicing on):
    seconds (HIP events on the launch stream), algorithmic TFLOP with the reference's tiling (175 / 369) and the
    fraction of the 2.5 PF/s dense bf16 MFMA peak."""
    g = torch.Generator(device=dev).manual_seed(0)
    x = (torch.rand(1, 3, 41, 480, 720, generator=g, device=dev) * 2 - 1).to(torch.bfloat16)
    z = torch.randn(1, 16, 11, 60, 90, generator=g, device=dev).to(torch.bfloat16)
    out = {}

    def samples(fn, n):
        """n individually timed calls (HIP events on the launch stream) after 3 untimed ones: eager first call (sizes the workspace), hipGraph
        capture on the second, one replay."""
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ms = []
        for _ in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        return ms

    for name, fn, tflop in (("encode", lambda: vae.encode(x).latent_dist.mode(), 175.0), ("decode", lambda: vae.decode(z).sample, 369.0)):
        ms = samples(fn, max(reps, 5))
        sec = sorted(ms)[len(ms) // 2] * 1e-3                     # median
        out[name] = {"seconds": sec, "algorithmic_tflop": tflop, "tflops": tflop / sec, "mfma_frac": tflop / sec / MFMA_PEAK_TFLOPS,
                     "samples_ms": [round(v, 1) for v in ms]}
    # the pipeline's two final decodes (rgb + disparity latents, P:931,936) as it issues them (AetherVAE.decode_pair: with the two-lane launch plan
    # two calls in a row, each with its tile batches on two streams)
    z2 = torch.randn(1, 16, 11, 60, 90, generator=g, device=dev).to(torch.bfloat16)
    for _ in range(3):
        vae.decode_pair(z, z2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        vae.decode_pair(z, z2)
    torch.cuda.synchronize()
    sec = (time.perf_counter() - t0) / reps
    out["decode_pair"] = {"seconds": sec, "algorithmic_tflop": 2 * 369.0, "tflops": 2 * 369.0 / sec, "mfma_frac": 2 * 369.0 / sec / MFMA_PEAK_TFLOPS,
                                      "note": "both decodes of one pipeline call (two-lane launch plan: the tile batches of each decode on two HIP streams)"}
    return out


def clip_wall_clock(transformer, dev, steps, vae):
    """Wall-clock of ONE whole pipeline call (BASELINE metric, second half): reconstruction of a synthetic 41x480x720 clip
    through the drop-in entry point — VAE encode (tiled, 9 tiles x 5 frame chunks), `steps` denoise steps, two VAE decodes,
    D2H of rgb/disparity/raymap — random-init weights, device generator seeded like scripts/demo.py:629."""
    import numpy as np

    from aether.pipelines.aetherv1_pipeline_cogvideox import AetherV1PipelineCogVideoX
    from aether_amd.scheduler import CogVideoXDPMScheduler

    g = torch.Generator().manual_seed(0)
    prompt = (torch.randn(1, 226, 4096, generator=g) * 0.1).to(torch.bfloat16)
    pipe = AetherV1PipelineCogVideoX(tokenizer=None, text_encoder=None, vae=vae, scheduler=CogVideoXDPMScheduler(),
                                     transformer=transformer, empty_prompt_embeds=prompt)
    pipe.set_progress_bar_config(disable=True)
    yy, xx = np.mgrid[0:480, 0:720].astype(np.float32)
    video = np.stack([np.stack([0.5 + 0.4 * np.sin(0.02 * xx + 0.1 * t + c) * np.cos(0.015 * yy) for c in range(3)], -1)
                      for t in range(41)]).astype(np.float32)
    out = {}
    for label, n in (("warmup", 1), (f"reconstruction_{steps}_steps", steps), ("reconstruction_4_steps_reference_default", 4)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = pipe(task="reconstruction", video=video, height=480, width=720, num_frames=41, num_inference_steps=n, fps=12,
                   generator=torch.Generator(device=dev).manual_seed(42))
        torch.cuda.synchronize()
        if label != "warmup":
            out[label] = {"seconds": time.perf_counter() - t0, "steps": n}
        assert res.rgb.shape == (41, 480, 720, 3) and np.isfinite(res.rgb).all() and np.isfinite(res.disparity).all()
    # BASELINE configs[2] / configs[3] on their NAMED inputs, exactly the sequence scripts/demo.py runs for them (D:564-606): the guided call
    # (50 steps, B = 2 through the transformer, dynamic classifier-free guidance: the per-task defaults P:257-272) followed by the 4-step
    # post-reconstruction of the generated clip (`--post_reconstruction`, D:581-606: disparity and raymap come from that second call).
    named = os.path.join(ROOT, "tests", "golden", "named_inputs.npz")
    if os.path.exists(named):
        import PIL.Image

        from aether_amd.geometry import forward_right_raymap
        z = np.load(named)
        img = lambda k: PIL.Image.fromarray(z[k])  # noqa: E731
        cases = (("prediction", dict(image=img("car"), raymap=forward_right_raymap()), "car.png + forward-right raymap (camera_pose_to_raymap)"),
                 ("planning", dict(image=img("obs01"), goal=img("goal01")), "01_obs.png + 01_goal.png"))
        for task, kw, what in cases:
            for label, n in (("warmup", 1), ("timed", steps)):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                gen_out = pipe(task=task, height=480, width=720, num_frames=41, fps=12, num_inference_steps=n,
                               generator=torch.Generator(device=dev).manual_seed(42), **kw)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                rec = pipe(task="reconstruction", video=gen_out.rgb, height=480, width=720, num_frames=41, fps=12, num_inference_steps=4,
                           guidance_scale=1.0, use_dynamic_cfg=False, generator=torch.Generator(device=dev).manual_seed(42))
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                assert gen_out.rgb.shape == (41, 480, 720, 3) and np.isfinite(gen_out.rgb).all() and np.isfinite(rec.disparity).all()
                if label == "timed":
                    out[f"{task}_{steps}_steps_cfg_plus_post_reconstruction"] = {
                        "seconds": t2 - t0, "guided_call_seconds": t1 - t0, "post_reconstruction_seconds": t2 - t1, "steps": n,
                        "inputs": what, "guidance": "dynamic CFG, scale 3.0, B = 2 through the DiT (per-task defaults P:257-272)"}
    out["unit"] = "s per 41f 480x720 clip (VAE encode + steps + 2 decodes + D2H), 1 GPU"
    return out


def windows_run(args, dev, rank, world, dist, transformer=None):
    """BASELINE configs[4]: long-video reconstruction — 192 synthetic frames = 8 sliding 41-frame windows (stride 24, starts
    0..144 + 151, scripts/demo.py:235-251), window w on rank w mod N, one gather of the device-resident outputs to rank 0 per ROUND of
    windows (RCCL; with N = 1 the same gathers run in a one-rank nccl group so the code path is exercised), rank 0 merging round j on the
    device while round j + 1 is computed (aether_amd.windows.run_windows_merged); merged arrays float32 into pinned host buffers
    (scripts/demo.py's default; the reference's float64 with --float64_outputs).  Returns the result dict on rank 0, None elsewhere."""
    import numpy as np

    from aether.pipelines.aetherv1_pipeline_cogvideox import AetherV1PipelineCogVideoX
    from aether_amd.scheduler import CogVideoXDPMScheduler
    from aether_amd.transformer import AetherTransformer3D
    from aether_amd.vae import AetherVAE
    from aether_amd.windows import get_window_starts, run_windows_merged

    if transformer is None:
        transformer = AetherTransformer3D({"num_layers": args.layers}, device=dev).init_random_weights(seed=0)
    vae = AetherVAE(device=dev).init_random_weights(1)
    vae.enable_slicing(); vae.enable_tiling()
    g = torch.Generator().manual_seed(0)
    prompt = (torch.randn(1, 226, 4096, generator=g) * 0.1).to(torch.bfloat16)
    pipe = AetherV1PipelineCogVideoX(tokenizer=None, text_encoder=None, vae=vae, scheduler=CogVideoXDPMScheduler(),
                                     transformer=transformer, empty_prompt_embeds=prompt)
    pipe.set_progress_bar_config(disable=True)
    pipe.keep_outputs_on_device = True
    n_frames = 192
    yy, xx = np.mgrid[0:480, 0:720].astype(np.float32)
    video = np.stack([np.stack([0.5 + 0.4 * np.sin(0.02 * xx + 0.1 * t + c) * np.cos(0.015 * yy) for c in range(3)], -1)
                      for t in range(n_frames)]).astype(np.float32)
    starts = get_window_starts(n_frames, 41, 24)

    def call_window(s0):
        return pipe(task="reconstruction", video=video[s0:s0 + 41], height=480, width=720, num_frames=41,
                    num_inference_steps=args.window_steps, fps=12, generator=torch.Generator(device=dev).manual_seed(42))

    call_window(0)                                           # warm-up (allocations, first-touch, hipGraph capture)
    if rank == 0:                                            # the page-locked output buffers exist before the clock starts (like every workspace)
        from aether_amd import windows as W_
        for tag, shp in (("rgb", (n_frames, 480, 720, 3)), ("disparity", (n_frames, 480, 720)), ("pointmaps", (n_frames, 480, 720, 3))):
            W_._pinned(tag, shp, torch.float32)
    dist.barrier(); torch.cuda.synchronize()
    tm = {}
    t0 = time.perf_counter()
    merged = run_windows_merged(call_window, starts, height=480, width=720, gather_device=dev, out_dtype=np.float32, pinned=True,
                                force_collective=True, timings=tm)
    torch.cuda.synchronize()
    t_total = time.perf_counter() - t0
    out = None
    if rank == 0:
        rgb, disp, poses, pointmaps = merged
        assert rgb.shape == (n_frames, 480, 720, 3) and np.isfinite(disp).all() and np.isfinite(pointmaps).all()
        out = {"seconds_per_192_frame_clip": t_total, "windows_and_gather": tm["windows_and_gather"], "merge_tail_incl_d2h": tm["merge_tail"],
               "windows": len(starts), "window_starts": starts, "sampler_steps_per_window": args.window_steps, "n_gpus": world,
               "windows_per_rank": -(-len(starts) // world), "merged_dtype": "float32 (pinned host buffers)",
               "gather": f"one dist.gather(dst=0) of device tensors per round of {world} window(s), backend {dist.get_backend()}; "
                         "rank 0 merges round j on a side stream while round j+1 is computed",
               "workload": "configs[4]: long-video reconstruction, 8 windows x 41 frames, stride 24"}
    # N >= 2: ONE reconstruction clip on two ranks — the two final decodes (40 % of the reference-default 4-step clip) on ranks 0 / 1
    # (AetherV1PipelineCogVideoX.enable_decode_parallel: replicated encode + loop with equal seeds, rank 0 decodes rgb, rank 1 disparity, one all-gather
    # of the decoded bf16 clips; bit-identical to one rank).  Compare with `clip.reconstruction_4_steps_reference_default` of the N = 1 line.
    if world >= 2:
        pair = dist.new_group([0, 1])                          # collective over all ranks
        single = None
        if rank < 2:
            pipe.keep_outputs_on_device = False
            pipe.enable_decode_parallel(pair)
            clip = video[:41]

            def one():
                return pipe(task="reconstruction", video=clip, height=480, width=720, num_frames=41, num_inference_steps=args.window_steps, fps=12,
                            generator=torch.Generator(device=dev).manual_seed(42))
            one()                                              # warm-up
            dist.barrier(group=pair); torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = one()
            torch.cuda.synchronize(); dist.barrier(group=pair)
            dt = time.perf_counter() - t0
            pipe.disable_decode_parallel()
            assert res.rgb.shape == (41, 480, 720, 3) and np.isfinite(res.rgb).all() and np.isfinite(res.disparity).all()
            single = {"seconds": dt, "steps": args.window_steps, "ranks": 2,
                      "workload": "configs[1]-style single clip (41x480x720) over 2 ranks: replicated encode + sampling loop, the two final decodes split (P:931 on rank 0, P:936 on rank 1), one all-gather"}
        if out is not None:
            out["single_clip_two_ranks_decode_parallel"] = single
    dist.barrier()
    del pipe, vae
    torch.cuda.empty_cache()
    return out


def windows_mode(args, dev, rank, world, dist):
    """`bench.py --windows`: BASELINE configs[4] end to end instead of the step benchmark; prints its own JSON line."""
    own_group = dist is None
    if own_group:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    d = windows_run(args, dev, rank, world, dist)
    if rank == 0:
        print(json.dumps({
            "metric": "wall-clock per 192-frame 480x720 clip (8 sliding 41f windows + temporal blend)", "value": d["seconds_per_192_frame_clip"], "unit": "s",
            "n_gpus": world, "steps": 1, "warmup": 1, "ms_per_step": d["seconds_per_192_frame_clip"] * 1e3, "higher_is_better": False, "scaling": "strong",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic (192 smooth frames; random-init weights)",
            "config": {"workload": d["workload"], "valid": args.layers == 42}, "windows": d}), flush=True)
    dist.destroy_process_group()


def cfg_parallel_leg(args, dev, rank, dist, model, steps):
    """N = 2 only: ONE guided denoise step split over the two ranks: rank 0 evaluates the unconditional branch, rank 1
    the conditional one at batch 1, ONE all-gather of the bf16 noise prediction (6.65 MB per rank) per step over RCCL, identical combine + DPM
    step on both.  Times `steps` steps of exactly the pipeline's loop body (aether_amd pipeline `_gather_pair`)."""
    from aether_amd.scheduler import CogVideoXDPMScheduler, randn_tensor
    from aether_amd.rope import resize_crop_region_for_grid, rotary_tables_3d
    c = model.config
    F_, H_, W_ = 11, 60, 90
    gen = torch.Generator(device=dev).manual_seed(42)          # the SAME seed on both ranks: the latents stay replicated without a broadcast
    prompt = (torch.randn(1, c.max_text_seq_length, c.text_embed_dim, generator=gen, device=dev) * 0.1).to(torch.bfloat16)
    cond = torch.randn(2, F_, 40, H_, W_, generator=gen, device=dev).to(torch.bfloat16)[rank:rank + 1]
    latents = randn_tensor((1, F_, 56, H_, W_), generator=gen, device=dev, dtype=torch.bfloat16)
    rope = rotary_tables_3d(64, resize_crop_region_for_grid((H_ // 2, W_ // 2), c.sample_width // 2, c.sample_height // 2), (H_ // 2, W_ // 2), F_, 1.0, device=dev)
    sched = CogVideoXDPMScheduler()
    sched.set_timesteps(50, device=dev)
    ts = sched.timesteps.tolist()
    old = None

    def one(i, latents, old):
        model_in = torch.cat([latents, cond], dim=2)
        pred = model(hidden_states=model_in, encoder_hidden_states=prompt, timestep=sched.timesteps[i].expand(1), ofs=None, image_rotary_emb=rope,
                     attention_kwargs=None, return_dict=False)[0]
        parts = [torch.empty_like(pred) for _ in range(2)]
        dist.all_gather(parts, pred.contiguous())
        return sched.step_fused(torch.cat(parts), old, ts[i], ts[i - 1] if i > 0 else None, latents, guidance_scale=3.0, generator=gen)

    latents, old = one(0, latents, old)
    dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(1, steps + 1):
        latents, old = one(i, latents, old)
    dist.barrier(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    chk = latents.float().sum().reshape(1)
    both = [torch.empty_like(chk) for _ in range(2)]
    dist.all_gather(both, chk)
    return {"steps_per_s": steps / dt, "ms_per_step": dt / steps * 1e3, "ranks_hold_identical_latents": bool(torch.equal(both[0], both[1])),
            "workload": "configs[2]/[3]: ONE guided step over 2 ranks (one guidance branch each at B = 1, all-gather of noise_pred over RCCL, DPM step)"}


def windows_leg(args):
    """BASELINE configs[4] on this GPU (`bench.py --windows`: 192 frames = 8 windows, gather in a one-rank nccl group, device merge) as a child
    process — RCCL prints a version banner on stdout when its group comes up, and this process's stdout must stay ONE JSON line."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--windows", "--window-steps", str(args.window_steps), "--layers", str(args.layers)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    for ln in reversed(r.stdout.splitlines()):
        if ln.startswith("{"):
            return json.loads(ln)["windows"]
    return {"error": (r.stderr or r.stdout)[-400:]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--layers", type=int, default=42, help="debug only; anything but 42 marks the line invalid")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-clip", action="store_true", help="skip the end-to-end clip wall-clock leg (pipeline incl. VAE)")
    ap.add_argument("--clip-steps", type=int, default=50, help="sampler steps of the clip leg (reference default for reconstruction: 4)")
    ap.add_argument("--cfg", action="store_true", help="B=2 (prediction/planning CFG) instead of reconstruction B=1")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the attention-path legs and the VAE leg (profiling runs)")
    ap.add_argument("--windows", action="store_true", help="BASELINE configs[4] end to end instead of the step benchmark: 192-frame clip, "
                    "8 windows (stride 24) sharded over the ranks, gather to rank 0 over RCCL, device merge; reports s per clip")
    ap.add_argument("--dit-flags-or", type=int, default=0, help="A/B: OR these AETHER_* bits into the transformer's default flags")
    ap.add_argument("--dit-flags-clear", type=int, default=0, help="A/B: clear these AETHER_* bits from the transformer's default flags")
    ap.add_argument("--legs-timeout", type=int, default=420, help="N > 1: seconds after which rank 0 prints the headline without the extra legs")
    ap.add_argument("--window-steps", type=int, default=4, help="sampler steps per window in --windows mode (reference default: 4)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # one process per GPU: re-launch this script under torch.distributed.run (the driver does this itself for N > 1)
        import socket
        if torch.cuda.device_count() < args.gpus and os.environ.get("AETHER_BENCH_ONE_DEVICE") != "1":
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {torch.cuda.device_count()} HIP device(s) visible")
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch.distributed as dist
        # plumbing check of the N > 1 path on a ONE-GPU box (AETHER_BENCH_ONE_DEVICE=1): every rank on cuda:0, exchange over gloo (RCCL refuses
        # two ranks on one device); never set by the driver — its N > 1 runs are one rank per GPU over RCCL
        one_dev = os.environ.get("AETHER_BENCH_ONE_DEVICE") == "1"
        dev_index = 0 if one_dev else local_rank
        torch.cuda.set_device(dev_index)
        from datetime import timedelta
        if one_dev:
            dist.init_process_group("gloo", timeout=timedelta(minutes=8))
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index), timeout=timedelta(minutes=8))
    else:
        dist = None
        dev_index = 0
        torch.cuda.set_device(0)
    dev = torch.device("cuda", dev_index)
    if args.windows:
        return windows_mode(args, dev, rank, world, dist)

    from aether_amd.rope import resize_crop_region_for_grid, rotary_tables_3d
    from aether_amd.scheduler import CogVideoXDPMScheduler, randn_tensor
    from aether_amd.transformer import AetherTransformer3D

    B = 2 if args.cfg else 1
    F_, H_, W_ = 11, 60, 90
    model = AetherTransformer3D({"num_layers": args.layers}, device=dev).init_random_weights(seed=0)
    if args.dit_flags_or or args.dit_flags_clear:
        model.set_flags((model._flags | args.dit_flags_or) & ~args.dit_flags_clear)
    c = model.config
    D, FF = model.inner_dim, 4 * model.inner_dim
    S = c.max_text_seq_length + F_ * (H_ // 2) * (W_ // 2)

    gen = torch.Generator(device=dev).manual_seed(42 + rank)          # scripts/demo.py:93-97 seed, one window per rank
    prompt = (torch.randn(1, c.max_text_seq_length, c.text_embed_dim, generator=gen, device=dev) * 0.1).to(torch.bfloat16)
    cond = torch.randn(1, F_, 40, H_, W_, generator=gen, device=dev).to(torch.bfloat16)     # 16 latent + 24 raymap ch (P:682)
    latents = randn_tensor((1, F_, 56, H_, W_), generator=gen, device=dev, dtype=torch.bfloat16)
    crop = resize_crop_region_for_grid((H_ // 2, W_ // 2), c.sample_width // 2, c.sample_height // 2)
    rope = rotary_tables_3d(64, crop, (H_ // 2, W_ // 2), F_, 1.0, device=dev)
    sched = CogVideoXDPMScheduler()
    sched.set_timesteps(50, device=dev)
    timesteps = sched.timesteps
    ts_host = timesteps.tolist()

    state = {"latents": latents, "old_x0": None, "i": 0, "B": B}

    def step():
        B = state["B"]
        i = state["i"] % len(ts_host)
        if i == 0:
            state["old_x0"] = None
        t = timesteps[i]
        lat = state["latents"]
        model_in = torch.cat([lat] * 2) if B == 2 else lat
        cnd = torch.cat([cond] * 2) if B == 2 else cond
        model_in = torch.cat([model_in, cnd], dim=2)                                          # P:857-859
        noise_pred = model(hidden_states=model_in, encoder_hidden_states=prompt.repeat(B, 1, 1), timestep=t.expand(B),
                           ofs=None, image_rotary_emb=rope, attention_kwargs=None, return_dict=False)[0]
        # the element-wise tail exactly as the drop-in pipeline runs it (P:877-916): ONE kernel, bit-identical to the PyTorch sequence
        lat, state["old_x0"] = sched.step_fused(noise_pred, state["old_x0"], ts_host[i], ts_host[i - 1] if i > 0 else None, lat,
                                                guidance_scale=3.0 if B == 2 else None, generator=gen)
        state["latents"] = lat
        state["i"] += 1

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n_warm, n_steps):
        for _ in range(n_warm):
            step()
        model.set_profile(True)
        barrier()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step()
        barrier()
        dt = time.perf_counter() - t0
        pr = model.get_profile()
        model.set_profile(False)
        return dt, pr

    elapsed, prof = timed(args.warmup, args.steps)          # THE measurement: default flags, exactly --steps steps
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    assert torch.isfinite(state["latents"].float()).all(), "non-finite latents"
    def build_line():
        steps_per_s = world * args.steps / elapsed
        total_ms = sum(ms for ms, _ in prof.values())
        dom = max((k for k in prof if flops_per_launch(k, B, S, D, FF) > 0), key=lambda k: prof[k][0])
        dom_ms, dom_n = prof[dom]
        fl = flops_per_launch(dom, B, S, D, FF)
        ach = fl / (dom_ms / dom_n * 1e-3) / 1e12
        traffic, traffic_src, prof_us = pmc_traffic(dom, c.num_layers)
        profile_frac = None if (prof_us is None or B != 1) else fl / (prof_us * 1e-6) / 1e12 / MFMA_PEAK_TFLOPS
        assert profile_frac is None or 0.0 < profile_frac <= 1.0, f"profile_frac {profile_frac}: the committed trace was mis-attributed"
        line = {
            "metric": "denoise-steps/s (41f 480x720 clip, 11x60x90 latent, B=%d through the DiT)" % B,
            "value": steps_per_s, "unit": "denoise-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic (seeded latents/conditions/prompt embeds; random-init weights, 5.57 B params)",
            "config": {"workload": ("configs[2]-style CFG step" if B == 2 else "configs[1]: 4D reconstruction 41x480x720, bf16, 1 window per GPU"),
                       "tokens": S, "layers": c.num_layers, "width": D, "heads": c.num_attention_heads, "batch_through_dit": B,
                       "scheduler": "CogVideoXDPMScheduler(50 steps)", "valid": args.layers == 42},
            "mfma_frac_whole_step": steps_per_s / world * B * 260.8e12 / (MFMA_PEAK_TFLOPS * 1e12),
            "roofline": {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_unit": "HBM bytes per launch (rocprofv3 PMC)",
                         "traffic_source": traffic_src,
                         # the same fraction from the committed rocprofv3 kernel trace (average launch duration under the profiler, B = 1) — what a
                         # reader recomputes from profiles/; `frac` is this run's own HIP-event average
                         "profile_avg_launch_ms": None if prof_us is None else prof_us / 1e3,
                         "profile_frac": profile_frac,
                         "csrc_sha16": __import__("aether_amd.build", fromlist=["x"]).source_digest(), "avg_launch_ms": dom_ms / dom_n, "launches": dom_n,
                         "algorithmic_flops_per_launch": fl},
            "kernel_ms_per_step": {k: round(ms / args.steps, 3) for k, (ms, _) in prof.items()},
            "kernel_tflops": {k: round(flops_per_launch(k, B, S, D, FF) * n / (ms * 1e-3) / 1e12, 1) for k, (ms, n) in prof.items()
                              if flops_per_launch(k, B, S, D, FF) > 0 and ms > 0},
            "gpu_kernel_ms_per_step_total": total_ms / args.steps,
        }
        return line, steps_per_s, ach

    # N > 1: the replica steps above say nothing about the one thing that shards WITH an exchange — BASELINE configs[4] (8 windows over the N
    # ranks, RCCL gathers, device merge) runs in the SAME process group and is attached to the rank-0 line; N = 2 adds the guided step split
    # over the two ranks.  Both legs are outside the timed region of `value`.
    multi = {}
    watchdog = None
    if dist is not None and not args.no_extra_legs and rank == 0:
        # a leg in which a PEER rank hangs or dies inside a collective blocks this rank until the process group's timeout aborts the process —
        # without the headline.  Rank 0 therefore arms a timer: if the legs have not returned in time it prints the headline line (the timed
        # steps above are complete) with the failure recorded, and leaves.
        import threading
        print_lock, printed = threading.Lock(), [False]        # exactly ONE headline line: whoever takes the lock first prints it

        def give_up():
            with print_lock:
                if printed[0]:
                    return
                printed[0] = True
                ln = build_line()[0]
                ln["extra_legs_error"] = "timeout: the N > 1 legs (windows / guided split) did not finish in %d s; headline printed by the watchdog" % args.legs_timeout
                print(json.dumps(ln), flush=True)
            os._exit(3)                                        # non-zero: the launcher sees that the legs hung (the line itself is complete)
        watchdog = threading.Timer(args.legs_timeout, give_up)
        watchdog.daemon = True
        watchdog.start()
    if dist is not None and not args.no_extra_legs:
        # a failure in an extra leg must not cost the headline line: it is recorded instead (the process group's timeout bounds a leg in
        # which only some ranks failed)
        try:
            if world == 2:
                multi["cfg_parallel_step"] = cfg_parallel_leg(args, dev, rank, dist, model, args.steps)
            multi["windows"] = windows_run(args, dev, rank, world, dist, transformer=model)
        except Exception as e:  # noqa: BLE001
            multi["extra_legs_error"] = f"{type(e).__name__}: {e}"[:400]
    if watchdog is not None:
        watchdog.cancel()

    if rank == 0 and watchdog is not None:
        with print_lock:                                       # the watchdog may have started printing: then it owns the line and ends the process
            if printed[0]:
                time.sleep(30)
                os._exit(3)
            printed[0] = True
    if rank == 0:
        line, steps_per_s, ach = build_line()
        line.update({k: v for k, v in multi.items() if v is not None})
        if world == 1 and not args.no_extra_legs:
            # ONE VAE for the process, built before the other legs churn the heap, shared by the VAE leg and the clip legs (as a pipeline shares
            # its VAE across calls).  Measured in round 4: a VAE whose 9 / 15 GiB workspace was allocated AFTER the clip and windows legs ran the
            # SAME captured encode graph 40 % slower on every sample (265 ms against 188-195 ms before those legs, after a re-allocation, and in a
            # fresh process; profiles/r04_bench_vae_order.json) — address-dependent, most likely the page size the driver could still find for a
            # multi-GB block.
            shared_vae = make_vae(dev)
            line["vae"] = vae_leg(dev, shared_vae)
        if world == 1 and not args.no_extra_legs:
            # The attention soft-max is exact on every path (include/aether_hip.h).  Legs, each `--steps` timed steps: the default
            # (optimistic shift-0 tile-pair sweep), the CONSERVATIVE path alone (true-maximum shift from the first tile, a-posteriori
            # check per tile: no dependence on the data or the weights = the floor), and the worst case of the default path: q/k-norm
            # weights x10 (log2-domain scores of several hundred: every workgroup's optimistic sweep is thrown away and redone).
            from aether_amd import _lib as L_
            att = lambda pr: round(flops_per_launch("attention", B, S, D, FF) * pr["attention"][1] / (pr["attention"][0] * 1e-3) / 1e12, 1)  # noqa: E731
            paths = {"default": {"steps_per_s": steps_per_s, "attention_tflops": line["kernel_tflops"].get("attention")}}
            fl0 = model._flags
            model.set_flags(fl0 | L_.AETHER_ATTN_EXACT_MAX)
            dt, pr = timed(1, args.steps)
            paths["conservative_path_data_independent"] = {"steps_per_s": args.steps / dt, "attention_tflops": att(pr)}
            model.set_flags(fl0)
            saved = {k: model._weights[k].clone() for k in ("qn_w", "kn_w")}          # restored bit-exactly below (x10 then /10 is not)
            model._weights["qn_w"].mul_(10.0); model._weights["kn_w"].mul_(10.0)
            dt, pr = timed(1, args.steps)
            paths["worst_case_every_workgroup_redoes_qk_norm_x10"] = {"steps_per_s": args.steps / dt, "attention_tflops": att(pr)}
            for k, v in saved.items():
                model._weights[k].copy_(v)
            line["attention_paths"] = paths
            # `value` is measured on seeded RANDOM weights, where every workgroup stays on the optimistic shift-0 sweep; the conservative path does not
            # look at the data (no fast path to leave), so this is what any checkpoint gets at least
            line["value_data_independent"] = paths["conservative_path_data_independent"]["steps_per_s"]
            # BASELINE configs[2] / [3] (prediction / planning): classifier-free guidance = B = 2 through the transformer + the combine
            state.update(B=2, i=0, old_x0=None)
            dt, pr = timed(1, args.steps)
            state.update(B=B, i=0, old_x0=None)
            line["cfg_step"] = {"steps_per_s": args.steps / dt, "ms_per_step": dt / args.steps * 1e3, "batch_through_dit": 2,
                                "mfma_frac_whole_step": args.steps / dt * 2 * 260.8e12 / (MFMA_PEAK_TFLOPS * 1e12),
                                "workload": "configs[2]/[3]: guided step (cat, B=2 forward, CFG combine, DPM step)",
                                "attention_tflops": round(flops_per_launch("attention", 2, S, D, FF) * pr["attention"][1] / (pr["attention"][0] * 1e-3) / 1e12, 1)}
        if world == 1 and not args.no_extra_legs:
            cap = mfma_power_cap_leg()
            if cap is not None:
                cap["whole_step_frac_of_random_operand_stream"] = round(steps_per_s * B * 260.8 / cap["random_operands"]["tflops"], 4)
                cap["dominant_kernel_frac_of_random_operand_stream"] = round(ach / cap["random_operands"]["tflops"], 4)
            line["mfma_power_cap"] = cap
        if world == 1 and not args.no_clip:
            line["clip"] = clip_wall_clock(model, dev, args.clip_steps, shared_vae if not args.no_extra_legs else make_vae(dev))
        if world == 1 and not args.no_clip and not args.no_extra_legs:
            line["windows"] = windows_leg(args)
        if world == 1 and not args.no_cpu_baseline:
            model = None
            torch.cuda.empty_cache()
            line["cpu_baseline"] = cpu_baseline({"num_layers": args.layers}, (F_, H_, W_))
            line["cpu_baseline"]["covers"] = "the DiT steps only (>= 93 % of a 50-step clip); VAE: cpu_baseline_vae"
            line["cpu_baseline"]["unsampled_offline"] = offline_cpu_figures()
            if not args.no_extra_legs:
                line["cpu_baseline_vae"] = cpu_baseline_vae()
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
