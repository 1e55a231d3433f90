#!/usr/bin/env python
"""AetherV1 inference demo on MI355X — same command line as /root/reference/scripts/demo.py (flags and defaults of
D:52-203), same task flow (D:524-646): prediction / planning (+ the default 4-step post-reconstruction, D:589-606),
reconstruction with sliding 41-frame windows (D:607-631).

What differs: the three model slots are the MI355X-native modules (HIP kernels behind include/aether_hip.h), and long
videos can be sharded over the GPUs of a node — launch with
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 scripts/demo.py --task reconstruction ...
and window w runs on rank w mod N (aether_amd/windows.py); rank 0 blends and saves.

Extra, optional flags (the reference's flags are unchanged): --empty_prompt_embeds (a .pt with the cached T5 embedding
of "" — the only thing the text encoder is ever used for, P:290-297), --synthetic_weights (seeded random weights, for
smoke runs without checkpoints).  The window merge (disparity scale fitting, camera alignment, pose smoothing, point maps:
D:254-422) runs on the host like the reference's (aether_amd/windows.py + aether_amd/geometry.py, pinned against the
reference's own functions by tests/golden/blend.npz); merged rgb / disparity / poses / point maps are written as .npz + a
preview frame — mp4 / GLB export (D:425-521: imageio, trimesh) is outside this repo's scope.
"""
from __future__ import annotations

import argparse
import os
import random
import sys

import numpy as np
import PIL.Image
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from aether.pipelines.aetherv1_pipeline_cogvideox import AetherV1PipelineCogVideoX  # noqa: E402
from aether_amd.export import colorize_depth, flip_for_export, output_stem, write_video  # noqa: E402
from aether_amd.windows import WindowResult, blend_and_merge_window_results, get_window_starts, run_windows, run_windows_merged  # noqa: E402


def seed_all(seed: int = 0) -> None:
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def parse_args(argv=None) -> argparse.Namespace:
    p = argparse.ArgumentParser(description="AetherV1-CogvideoX Inference Demo", epilog="Multi-GPU (python -m torch.distributed.run --nproc-per-node N scripts/demo.py ...): "
                                "reconstruction of a long video shards its sliding windows over all N ranks; a single guided clip (prediction / planning) uses "
                                "ranks 0 and 1 for the two guidance branches and leaves ranks >= 2 idle.")
    p.add_argument("--task", type=str, required=True, choices=["reconstruction", "prediction", "planning"],
                   help="Task to perform: 'reconstruction', 'prediction' or 'planning'.")
    p.add_argument("--video", type=str, default=None, help="Path to a video file. Only used for 'reconstruction' task.")
    p.add_argument("--image", type=str, default=None, help="Path to an image file. Only used for 'prediction' and 'planning' tasks.")
    p.add_argument("--goal", type=str, default=None, help="Path to a goal image file. Only used for 'planning' task.")
    p.add_argument("--raymap_action", type=str, default=None,
                   help="Path to a raymap action file. Should be a numpy array of shape (num_frame, 6, latent_height, latent_width).")
    p.add_argument("--output_dir", type=str, default="outputs", help="Path to save the outputs.")
    p.add_argument("--seed", type=int, default=42, help="Random seed.")
    p.add_argument("--fps", type=int, default=12, choices=[8, 10, 12, 15, 24], help="Frames per second. Options: 8, 10, 12, 15, 24.")
    p.add_argument("--num_inference_steps", type=int, default=None,
                   help="Number of inference steps. If not specified, will use the default number of steps for the task.")
    p.add_argument("--guidance_scale", type=float, default=None,
                   help="Guidance scale. If not specified, will use the default guidance scale for the task.")
    p.add_argument("--use_dynamic_cfg", action="store_true", default=True, help="Use dynamic cfg.")
    p.add_argument("--height", type=int, default=480, help="Height of the output video.")
    p.add_argument("--width", type=int, default=720, help="Width of the output video.")
    p.add_argument("--num_frames", type=int, default=41, help="Number of frames to predict.")
    p.add_argument("--max_depth", type=float, default=100.0, help="Maximum depth of the scene in meters.")
    p.add_argument("--rtol", type=float, default=0.2, help="Relative tolerance for depth edge detection.")
    p.add_argument("--cogvideox_pretrained_model_name_or_path", type=str, default="THUDM/CogVideoX-5b-I2V",
                   help="Name or path of the CogVideoX model to use.")
    p.add_argument("--aether_pretrained_model_name_or_path", type=str, default="AetherWorldModel/AetherV1",
                   help="Name or path of the Aether model to use.")
    p.add_argument("--smooth_camera", action="store_true", default=True, help="Smooth the camera trajectory.")
    p.add_argument("--smooth_method", type=str, default="kalman", choices=["kalman", "simple"], help="Smooth method.")
    p.add_argument("--sliding_window_stride", type=int, default=24,
                   help="Sliding window stride (window size equals to num_frames). Only used for 'reconstruction' task.")
    p.add_argument("--post_reconstruction", action="store_true", default=True,
                   help="Run reconstruction after prediction for better quality. Only used for 'prediction' and 'planning' tasks.")
    p.add_argument("--pointcloud_save_frame_interval", type=int, default=10, help="Pointcloud save frame interval.")
    p.add_argument("--align_pointmaps", action="store_true", default=False, help="Align pointmaps.")
    # ---- additions (not in the reference) ----
    p.add_argument("--empty_prompt_embeds", type=str, default=None, help="[aether_amd] .pt file with the T5 embedding of the empty prompt [1,226,4096].")
    p.add_argument("--synthetic_weights", action="store_true", default=False, help="[aether_amd] seeded random weights instead of checkpoints (smoke runs).")
    p.add_argument("--synthetic_layers", type=int, default=42, help="[aether_amd] depth of the synthetic transformer.")
    p.add_argument("--float64_outputs", action="store_true", default=False,
                   help="[aether_amd] merged arrays of a long clip as float64 like the reference's numpy merge (default float32: half the D2H bytes).")
    return p.parse_args(argv)


def _device() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("scripts/demo.py needs an MI355X (aether_amd has no CPU fallback)")
    return torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))


def build_pipeline(args: argparse.Namespace, device: torch.device) -> AetherV1PipelineCogVideoX:
    """Same five slots as D:206-232; vae/scheduler/transformer are the native modules."""
    from aether_amd.scheduler import CogVideoXDPMScheduler
    from aether_amd.transformer import AetherTransformer3D
    from aether_amd.vae import AetherVAE

    tokenizer = text_encoder = None
    prompt_embeds = None
    if args.synthetic_weights:
        vae = AetherVAE(device=device).init_random_weights(args.seed)
        transformer = AetherTransformer3D({"num_layers": args.synthetic_layers}, device=device).init_random_weights(args.seed)
        scheduler = CogVideoXDPMScheduler()
        g = torch.Generator().manual_seed(args.seed)
        prompt_embeds = torch.randn(1, 226, 4096, generator=g) * 0.1
    else:
        cog, aether = args.cogvideox_pretrained_model_name_or_path, args.aether_pretrained_model_name_or_path
        vae = AetherVAE.from_pretrained(cog, subfolder="vae", torch_dtype=torch.bfloat16, device=device)
        scheduler = CogVideoXDPMScheduler.from_pretrained(cog, subfolder="scheduler")
        transformer = AetherTransformer3D.from_pretrained(aether, subfolder="transformer", torch_dtype=torch.bfloat16, device=device)
        if args.empty_prompt_embeds is not None:
            prompt_embeds = torch.load(args.empty_prompt_embeds, map_location="cpu")
        else:
            from transformers import AutoTokenizer, T5EncoderModel
            tokenizer = AutoTokenizer.from_pretrained(cog, subfolder="tokenizer")
            text_encoder = T5EncoderModel.from_pretrained(cog, subfolder="text_encoder")
    pipeline = AetherV1PipelineCogVideoX(tokenizer=tokenizer, text_encoder=text_encoder, vae=vae, scheduler=scheduler,
                                         transformer=transformer, empty_prompt_embeds=prompt_embeds)
    pipeline.vae.enable_slicing()
    pipeline.vae.enable_tiling()
    pipeline.to(device)
    return pipeline


def read_video(path: str) -> np.ndarray:
    """[N,H,W,3] float32 in [0,1].  mp4 needs imageio (as in the reference, D:542); .npy/.npz arrays and a directory of
    image files are accepted as well (imageio/ffmpeg are not part of the ROCm image)."""
    if os.path.isdir(path):
        files = sorted(f for f in os.listdir(path) if f.lower().endswith((".png", ".jpg", ".jpeg")))
        return np.stack([np.asarray(PIL.Image.open(os.path.join(path, f)).convert("RGB")) for f in files]).astype(np.float32) / 255.0
    if path.endswith(".npy"):
        v = np.load(path)
    elif path.endswith(".npz"):
        z = np.load(path)
        v = z[z.files[0]]
    else:
        import imageio.v3 as iio
        v = iio.imread(path)
    return v.astype(np.float32) / 255.0 if v.dtype == np.uint8 else v.astype(np.float32)


def merge(args, results, device=None):
    """D:633-639 / D:436-449: merged (rgb, disparity, poses, pointmaps) of one or more windows; the per-pixel part runs on
    `device` (the reference does it in float64 numpy on the host: ~20 s for a 192-frame clip)."""
    return blend_and_merge_window_results(results, height=args.height, width=args.width, align_pointmaps=args.align_pointmaps,
                                          smooth_camera=args.smooth_camera, smooth_method=args.smooth_method, device=device)


def save_output(args, rgb, disparity, poses=None, pointmap=None, **extra):
    """D:425-521: <output_dir>/<task>_<input name>_rgb.mp4 and _disparity.mp4 (colour-mapped); every array (point maps and
    poses with the reference's export flips) additionally goes into <...>.npz.  The GLB scenes need trimesh (absent here)."""
    os.makedirs(args.output_dir, exist_ok=True)
    filename = os.path.join(args.output_dir, output_stem(args.task, args.video, args.image, args.goal))
    arrays = dict(rgb=rgb, disparity=disparity, **{k: v for k, v in extra.items() if v is not None})
    if pointmap is not None and poses is not None:
        arrays["pointmap"], arrays["poses"] = flip_for_export(pointmap, poses)
    np.savez_compressed(f"{filename}.npz", **arrays)
    written = [f"{filename}.npz",
               write_video(f"{filename}_rgb.mp4", (np.clip(rgb, 0, 1) * 255).astype(np.uint8), fps=12),
               write_video(f"{filename}_disparity.mp4", (colorize_depth(disparity) * 255).astype(np.uint8), fps=12)]
    print("Saved outputs to " + ", ".join(written))


def main(argv=None) -> None:
    os.environ["TOKENIZERS_PARALLELISM"] = "false"
    args = parse_args(argv)
    seed_all(args.seed)
    if args.num_inference_steps is None:
        args.num_inference_steps = 4 if args.task == "reconstruction" else 50
    if args.guidance_scale is None:
        args.guidance_scale = 1.0 if args.task == "reconstruction" else 3.0

    world = int(os.environ.get("WORLD_SIZE", "1"))
    device = _device()
    torch.cuda.set_device(device)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=device)
    rank = int(os.environ.get("RANK", "0"))
    pipeline = build_pipeline(args, device)
    if rank != 0:
        pipeline.set_progress_bar_config(disable=True)

    image = goal = video = None
    if args.task == "reconstruction":
        assert args.video is not None, "Video is required for reconstruction task."
        assert args.image is None, "Image is not required for reconstruction task."
        assert args.goal is None, "Goal is not required for reconstruction task."
        video = read_video(args.video)
    elif args.task == "prediction":
        assert args.image is not None, "Image is required for prediction task."
        assert args.goal is None, "Goal is not required for prediction task."
        image = PIL.Image.open(args.image)
    else:
        assert args.image is not None, "Image is required for planning task."
        assert args.goal is not None, "Goal is required for planning task."
        image, goal = PIL.Image.open(args.image), PIL.Image.open(args.goal)
    raymap = np.load(args.raymap_action) if args.raymap_action is not None else None

    common = dict(height=args.height, width=args.width, num_frames=args.num_frames, fps=args.fps)
    if args.task != "reconstruction":
        # A single clip.  With two or more ranks the two guidance branches run on ranks 0 and 1:
        # one all-gather of the bf16 noise prediction per step; further ranks have nothing to do.
        pair = None
        if world >= 2 and args.guidance_scale > 1.0:
            import torch.distributed as dist
            pair = dist.new_group([0, 1])                 # collective over all ranks
            if rank < 2:
                pipeline.enable_cfg_parallel(pair)
        if rank == 0 or (pair is not None and rank == 1):
            output = pipeline(task=args.task, image=image, video=None, goal=goal, raymap=raymap,
                              num_inference_steps=args.num_inference_steps, guidance_scale=args.guidance_scale,
                              use_dynamic_cfg=args.use_dynamic_cfg, generator=torch.Generator(device=device).manual_seed(args.seed),
                              return_dict=True, **common)
        if rank == 0:
            geo = output
            if args.post_reconstruction:
                geo = pipeline(task="reconstruction", video=output.rgb, num_inference_steps=4, guidance_scale=1.0, use_dynamic_cfg=False,
                               generator=torch.Generator(device=device).manual_seed(args.seed), **common)
            # like the reference's save_output (D:436-449): a single window goes through the same merge to get poses / point maps
            _, _, poses, pointmaps = merge(args, [WindowResult(0, output.rgb, geo.disparity, geo.raymap.copy())], device)
            save_output(args, rgb=output.rgb, disparity=geo.disparity, raymap=geo.raymap, poses=poses, pointmap=pointmaps)
    else:
        starts = get_window_starts(len(video), args.num_frames, args.sliding_window_stride)

        def call_window(s):
            return pipeline(task=args.task, image=None, goal=None, video=video[s:s + args.num_frames],
                            raymap=raymap[s:s + args.num_frames] if raymap is not None else None,
                            num_inference_steps=args.num_inference_steps, guidance_scale=1.0, use_dynamic_cfg=False,
                            generator=torch.Generator(device=device).manual_seed(args.seed), **common)

        # window outputs never leave HBM between the pipeline, the gather to rank 0 (RCCL) and the device merge
        pipeline.keep_outputs_on_device = not args.align_pointmaps
        if len(starts) == 1 and world >= 2:
            # ONE window on several GPUs: the two final decodes (40 % of a 4-step clip) go to ranks 0 and 1 (AetherV1PipelineCogVideoX.
            # enable_decode_parallel: replicated encode + loop, one all-gather of the decoded videos; bit-identical to one rank).  Rank 1 makes
            # the same call as rank 0's call inside the window driver below and drops the result.
            import torch.distributed as dist
            pair = dist.new_group([0, 1])                 # collective over all ranks
            if rank < 2:
                pipeline.enable_decode_parallel(pair)
            if rank == 1:
                call_window(starts[0])
        if args.align_pointmaps:
            # the reference's point-map alignment branch stays on the host (numpy): gather everything, then merge
            results = run_windows(call_window, starts, gather_device=device, keep_on_device=False)
            merged = merge(args, results, device) if results is not None else None
        else:
            # rounds of one window per rank; rank 0 merges round j on a side stream while round j + 1 is computed (aether_amd.windows).
            # pinned=True: `merged` are views of process-wide page-locked buffers, consumed by save_output right below (a later merge of the
            # same shape would overwrite them; aether_amd.windows.release_pinned_buffers() frees them)
            merged = run_windows_merged(call_window, starts, height=args.height, width=args.width, gather_device=device,
                                        smooth_camera=args.smooth_camera, smooth_method=args.smooth_method,
                                        out_dtype=np.float64 if args.float64_outputs else np.float32, pinned=True)
        if merged is not None:
            rgb, disparity, poses, pointmaps = merged
            save_output(args, rgb=rgb, disparity=disparity, poses=poses, pointmap=pointmaps, window_starts=np.asarray(starts))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
