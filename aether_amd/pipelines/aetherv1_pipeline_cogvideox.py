"""Drop-in for `aether.pipelines.aetherv1_pipeline_cogvideox` of InternRobotics/Aether on MI355X.

Same public surface as the reference module (/root/reference/aether/pipelines/aetherv1_pipeline_cogvideox.py, "P:"):
`AetherV1PipelineCogVideoX(tokenizer, text_encoder, vae, scheduler, transformer)` (P:274-281), its
`__call__` keyword set and defaults (P:691-711), the per-task defaults (P:257-272), `check_inputs` error strings
(P:362-449), `AetherV1PipelineOutput(rgb, disparity, raymap)` (P:248-252) and the module-level helpers.

What differs is what sits underneath: the reference inherits from diffusers' CogVideoXImageToVideoPipeline and calls
diffusers modules; here the three module slots are duck-typed and are normally filled with
`aether_amd.transformer.AetherTransformer3D`, `aether_amd.vae.AetherVAE` and `aether_amd.scheduler.
CogVideoXDPMScheduler`, whose arithmetic runs in the hand-written HIP kernels of libaether_hip.so.  The few
members of the diffusers base class that the reference relies on (P:290,459,474,535,572,801,824,931-934,952) are
provided by `_PipelineBase` below.  Random draws happen in the same order, shape, dtype and device as in the
reference (posterior sample(s) -> initial latents -> per-step scheduler noise) so a seeded generator reproduces it.
"""
from __future__ import annotations

import inspect
import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple, Union

import numpy as np
import PIL.Image
import torch
from einops import rearrange

from ..preprocess import center_crop_frames, crop_window
from ..rope import resize_crop_region_for_grid, rotary_tables_3d
from ..scheduler import CogVideoXDPMScheduler, randn_tensor
from ..video_processor import VideoProcessor

__all__ = ["AetherV1PipelineCogVideoX", "AetherV1PipelineOutput", "get_3d_rotary_pos_embed",
           "get_resize_crop_region_for_grid", "retrieve_timesteps", "retrieve_latents"]


def get_resize_crop_region_for_grid(src, tgt_width, tgt_height):
    """P:148-163."""
    return resize_crop_region_for_grid(src, tgt_width, tgt_height)


def get_3d_rotary_pos_embed(embed_dim, crops_coords, grid_size, temporal_size, theta: int = 10000, use_real: bool = True,
                            grid_type: str = "linspace", max_size: Optional[Tuple[int, int]] = None,
                            device: Optional[torch.device] = None, fps_factor: Optional[float] = 1.0):
    """Signature of P:25-36.  Only the branch AetherV1 exercises (CogVideoX-1.0 "linspace" grid) is implemented."""
    if use_real is not True:
        raise ValueError(" `use_real = False` is not currently supported for get_3d_rotary_pos_embed")
    if grid_type != "linspace":
        raise ValueError("Invalid value passed for `grid_type`." if grid_type != "slice" else
                         "aether_amd: grid_type='slice' (CogVideoX 1.5) is not implemented")
    return rotary_tables_3d(embed_dim, crops_coords, tuple(grid_size), temporal_size, fps_factor, float(theta), device)


def retrieve_timesteps(scheduler, num_inference_steps: Optional[int] = None, device=None,
                       timesteps: Optional[List[int]] = None, sigmas: Optional[List[float]] = None, **kwargs):
    """P:167-229: lets the scheduler build its schedule and hands back (timesteps, num_inference_steps)."""
    if timesteps is not None and sigmas is not None:
        raise ValueError("Only one of `timesteps` or `sigmas` can be passed. Please choose one to set custom values")
    accepted = set(inspect.signature(scheduler.set_timesteps).parameters.keys())
    for name, value in (("timesteps", timesteps), ("sigmas", sigmas)):
        if value is None:
            continue
        if name not in accepted:
            what = "timestep" if name == "timesteps" else "sigmas"
            raise ValueError(f"The current scheduler class {scheduler.__class__}'s `set_timesteps` does not support custom"
                             f" {what} schedules. Please check whether you are using the correct scheduler.")
        scheduler.set_timesteps(device=device, **{name: value}, **kwargs)
        return scheduler.timesteps, len(scheduler.timesteps)
    scheduler.set_timesteps(num_inference_steps, device=device, **kwargs)
    return scheduler.timesteps, num_inference_steps


def retrieve_latents(encoder_output, generator: Optional[torch.Generator] = None, sample_mode: str = "sample"):
    """P:233-245."""
    dist = getattr(encoder_output, "latent_dist", None)
    if dist is not None and sample_mode == "sample":
        return dist.sample(generator)
    if dist is not None and sample_mode == "argmax":
        return dist.mode()
    if hasattr(encoder_output, "latents"):
        return encoder_output.latents
    raise AttributeError("Could not access latents of provided encoder_output")


@dataclass
class AetherV1PipelineOutput:
    rgb: np.ndarray
    disparity: np.ndarray
    raymap: np.ndarray


class _NullProgress:
    def __init__(self, total=None):
        self.total, self.n = total, 0

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def update(self, k=1):
        self.n += k


class _PipelineBase:
    """The slice of diffusers' DiffusionPipeline / CogVideoXImageToVideoPipeline the reference depends on."""

    def __init__(self, tokenizer, text_encoder, vae, scheduler, transformer):
        self.tokenizer, self.text_encoder = tokenizer, text_encoder
        self.vae, self.scheduler, self.transformer = vae, scheduler, transformer
        n_down = len(vae.config.block_out_channels) - 1 if vae is not None else 3
        self.vae_scale_factor_spatial = 2 ** n_down
        self.vae_scale_factor_temporal = vae.config.temporal_compression_ratio if vae is not None else 4
        self.vae_scaling_factor_image = vae.config.scaling_factor if vae is not None else 0.7
        self.video_processor = VideoProcessor(vae_scale_factor=self.vae_scale_factor_spatial)
        self._device = torch.device(getattr(transformer, "device", "cpu"))
        self._progress_bar_config: Dict = {}

    # -- device handling ---------------------------------------------------------------------------
    @property
    def _execution_device(self) -> torch.device:
        return self._device

    @property
    def device(self) -> torch.device:
        return self._device

    def to(self, device=None, dtype=None):
        if device is not None:
            self._device = torch.device(device)
            for m in (self.vae, self.transformer, self.text_encoder):
                if m is not None and hasattr(m, "to"):
                    m.to(device)
        return self

    def maybe_free_model_hooks(self):
        return None

    def set_progress_bar_config(self, **kwargs):
        self._progress_bar_config = kwargs

    def progress_bar(self, iterable=None, total=None):
        if self._progress_bar_config.get("disable", False):
            return _NullProgress(total)
        try:
            from tqdm.auto import tqdm
            return tqdm(total=total, **self._progress_bar_config) if iterable is None else tqdm(iterable, **self._progress_bar_config)
        except Exception:
            return _NullProgress(total)

    @property
    def guidance_scale(self):
        return self._guidance_scale

    @property
    def num_timesteps(self):
        return self._num_timesteps

    @property
    def interrupt(self):
        return self._interrupt

    def prepare_extra_step_kwargs(self, generator, eta):
        accepted = set(inspect.signature(self.scheduler.step).parameters.keys())
        extra = {}
        if "eta" in accepted:
            extra["eta"] = eta
        if "generator" in accepted:
            extra["generator"] = generator
        return extra

    # -- text ------------------------------------------------------------------------------------
    def encode_prompt(self, prompt, negative_prompt=None, do_classifier_free_guidance: bool = True,
                      num_videos_per_prompt: int = 1, prompt_embeds=None, negative_prompt_embeds=None,
                      max_sequence_length: int = 226, device=None, dtype=None):
        """T5 embedding of the prompt, padded to 226 tokens, no attention mask (diffusers' CogVideoX `_get_t5_prompt_embeds`).
        AetherV1 only ever encodes the empty prompt, once, at construction (P:290-297)."""
        if prompt_embeds is None:
            if self.tokenizer is None or self.text_encoder is None:
                raise ValueError("encode_prompt needs a tokenizer and a text_encoder (or pass `empty_prompt_embeds` to the pipeline)")
            prompt = [prompt] if isinstance(prompt, str) else prompt
            ids = self.tokenizer(prompt, padding="max_length", max_length=max_sequence_length, truncation=True,
                                 add_special_tokens=True, return_tensors="pt").input_ids
            enc_dev = next(self.text_encoder.parameters()).device
            prompt_embeds = self.text_encoder(ids.to(enc_dev))[0]
            b, s, _ = prompt_embeds.shape
            prompt_embeds = prompt_embeds.repeat(1, num_videos_per_prompt, 1).view(b * num_videos_per_prompt, s, -1)
        return prompt_embeds, negative_prompt_embeds

    # -- VAE -------------------------------------------------------------------------------------
    def decode_latents(self, latents: torch.Tensor) -> torch.Tensor:
        latents = latents.permute(0, 2, 1, 3, 4)  # [B, C, F, H, W]
        latents = 1 / self.vae_scaling_factor_image * latents
        return self.vae.decode(latents).sample


class AetherV1PipelineCogVideoX(_PipelineBase):
    _supported_tasks = ["reconstruction", "prediction", "planning"]
    _default_num_inference_steps = {"reconstruction": 4, "prediction": 50, "planning": 50}
    _default_guidance_scale = {"reconstruction": 1.0, "prediction": 3.0, "planning": 3.0}
    _default_use_dynamic_cfg = {"reconstruction": False, "prediction": True, "planning": True}
    _base_fps = 12
    _num_output_channels = 56  # 16 rgb + 16 disparity + 24 raymap latent channels (P:539, P:925-929)
    # Extension (not in the reference): leave rgb / disparity / raymap as float32 torch tensors on the execution device — same
    # values, no D2H copy — for callers that keep working on the GPU (sliding-window gather + merge, aether_amd/windows.py).
    keep_outputs_on_device = False
    # Extension: run the element-wise tail of every denoise step (P:877-916) as one HIP kernel (aether_dpm_step) when the scheduler is this
    # repo's CogVideoXDPMScheduler on an MI355X.  Bit-identical to the PyTorch sequence; False keeps the reference's op-by-op form.
    fuse_step_tail = True
    # Extension: hand the two final VAE decodes (rgb, disparity: P:931,936) to `AetherVAE.decode_pair` — two calls in a row under the two-lane
    # launch plan (each decode runs its tile batches on two streams); two HIP streams over a twin context (one more VAE workspace) without it.
    # Bit-identical outputs.  False = the reference's two `decode_latents` calls.
    decode_concurrently = True

    def __init__(self, tokenizer, text_encoder, vae, scheduler, transformer, empty_prompt_embeds: Optional[torch.Tensor] = None):
        super().__init__(tokenizer=tokenizer, text_encoder=text_encoder, vae=vae, scheduler=scheduler, transformer=transformer)
        if empty_prompt_embeds is None:
            empty_prompt_embeds, _ = self.encode_prompt(prompt="", negative_prompt=None, do_classifier_free_guidance=False,
                                                        num_videos_per_prompt=1, prompt_embeds=None)
        self.empty_prompt_embeds = empty_prompt_embeds.to(dtype=torch.bfloat16)

    # -------------------------------------------------------------------------------------------------
    def _prepare_rotary_positional_embeddings(self, height: int, width: int, num_frames: int, device, fps: Optional[int] = None):
        cfg = self.transformer.config
        p = cfg.patch_size
        grid_h = height // (self.vae_scale_factor_spatial * p)
        grid_w = width // (self.vae_scale_factor_spatial * p)
        if cfg.patch_size_t is not None:
            raise ValueError("aether_amd: patch_size_t (CogVideoX 1.5) models are not supported")
        crops = get_resize_crop_region_for_grid((grid_h, grid_w), cfg.sample_width // p, cfg.sample_height // p)
        return get_3d_rotary_pos_embed(embed_dim=cfg.attention_head_dim, crops_coords=crops, grid_size=(grid_h, grid_w),
                                       temporal_size=num_frames, device=device, fps_factor=self._base_fps / fps)

    # -------------------------------------------------------------------------------------------------
    def check_inputs(self, task, image, video, goal, raymap, height, width, num_frames, fps):
        def _is_img(x):
            return isinstance(x, (torch.Tensor, np.ndarray, PIL.Image.Image))

        if task not in self._supported_tasks:
            raise ValueError(f"`task` has to be one of {self._supported_tasks}.")
        if image is None and video is None:
            raise ValueError("`image` or `video` has to be provided.")
        if image is not None and video is not None:
            raise ValueError("`image` and `video` cannot both be provided.")
        if image is not None:
            if task == "reconstruction":
                raise ValueError("`image` is not supported for `reconstruction` task.")
            if not _is_img(image):
                raise ValueError("`image` has to be of type `torch.Tensor` or `np.ndarray` or `PIL.Image.Image` but is"
                                 f" {type(image)}")
        if goal is not None:
            if task != "planning":
                raise ValueError("`goal` is only supported for `planning` task.")
            if not _is_img(goal):
                raise ValueError("`goal` has to be of type `torch.Tensor` or `np.ndarray` or `PIL.Image.Image` but is"
                                 f" {type(goal)}")
        if video is not None:
            if task != "reconstruction":
                raise ValueError("`video` is only supported for `reconstruction` task.")
            pil_list = isinstance(video, list) and all(isinstance(v, PIL.Image.Image) for v in video)
            if not isinstance(video, (torch.Tensor, np.ndarray)) and not pil_list:
                raise ValueError("`video` has to be of type `torch.Tensor` or `np.ndarray` or `List[PIL.Image.Image]` but is"
                                 f" {type(video)}")
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if num_frames is None:
            raise ValueError("`num_frames` is required.")
        if num_frames not in [17, 25, 33, 41]:
            raise ValueError("`num_frames` has to be one of [17, 25, 33, 41].")
        if fps not in [8, 10, 12, 15, 24]:
            raise ValueError("`fps` has to be one of [8, 10, 12, 15, 24].")
        if raymap is not None and not isinstance(raymap, (torch.Tensor, np.ndarray)):
            raise ValueError("`raymap` has to be of type `torch.Tensor` or `np.ndarray`.")
        if raymap is not None:
            s = self.vae_scale_factor_spatial
            if tuple(raymap.shape[-4:]) != (num_frames, 6, height // s, width // s):
                raise ValueError(f"`raymap` shape is not correct. "
                                 f"Expected {num_frames, 6, height // s, width // s}, "
                                 f"got {raymap.shape}.")

    # -------------------------------------------------------------------------------------------------
    def _preprocess_image(self, image, height, width):
        if isinstance(image, torch.Tensor):
            image = image.cpu().numpy()
        if image.dtype == np.uint8:
            image = image.astype(np.float32) / 255.0
        frames = [image] if image.ndim == 3 else image
        frames = center_crop_frames(frames, height, width)
        return self.video_processor.preprocess(frames, height, width)

    def _preprocess_frames_on_device(self, frames, height, width, dev):
        """Same values as `_preprocess_image(...).to(dev, bfloat16)` for an [N,H,W,C] / [H,W,C] array, computed on the device from ONE
        upload of the raw frames.  On an MI355X uint8 / float32 clips of ANY size go through `aether_preprocess_frames` (/255,
        imcrop_center's window incl. zero fill, nearest resize with PyTorch's index rule, 2x-1, NHWC->NCHW, bf16 — one kernel).
        Elsewhere (CPU tests) and for float64 clips the no-resize case runs as torch operations (the same IEEE operations as the
        host path).  Returns None when neither applies (the caller falls back to the reference's host path)."""
        if isinstance(frames, torch.Tensor) or frames.dtype not in (np.uint8, np.float32, np.float64) or frames.ndim not in (3, 4):
            return None
        arr = frames[None] if frames.ndim == 3 else frames
        top, left, ch, cw = crop_window(arr.shape[1], arr.shape[2], height, width)
        dev = torch.device(dev)
        if dev.type == "cuda" and arr.dtype in (np.uint8, np.float32) and ch > 0 and cw > 0:
            from .. import _lib
            src = torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
            out = torch.empty(arr.shape[0], arr.shape[3], height, width, dtype=torch.bfloat16, device=dev)
            _lib.check(_lib.load().aether_preprocess_frames(src.data_ptr(), int(arr.dtype == np.uint8), arr.shape[0], arr.shape[1], arr.shape[2],
                                                            arr.shape[3], top, left, ch, cw, height, width, out.data_ptr(),
                                                            torch.cuda.current_stream(dev).cuda_stream), "aether_preprocess_frames")
            return out
        if (ch, cw) != (height, width) or top < 0 or left < 0 or top + ch > arr.shape[1] or left + cw > arr.shape[2]:
            return None
        t = torch.from_numpy(arr[:, top:top + ch, left:left + cw]).to(dev)
        if t.dtype == torch.uint8:
            # a true fp32 division like the host path (`x.astype(float32) / 255.0`): dividing by a Python scalar makes the device
            # kernel multiply by the rounded reciprocal, which is 1 ulp off for 126 of the 256 pixel values
            t = torch.div(t.to(torch.float32), torch.full((), 255.0, dtype=torch.float32, device=t.device))
        t = 2.0 * t.permute(0, 3, 1, 2) - 1.0
        return t.to(torch.bfloat16).contiguous()

    def preprocess_inputs(self, image, goal, video, raymap, height, width, num_frames):
        dev = self._execution_device

        def _one(x, pil_ok):
            if x is None:
                return None
            if pil_ok(x):
                y = self.video_processor.preprocess(x, height, width, resize_mode="crop")
            else:
                y = self._preprocess_frames_on_device(x, height, width, dev)
                if y is not None:
                    return y
                y = self._preprocess_image(x, height, width)
            return y.to(device=dev, dtype=torch.bfloat16)

        is_pil = lambda x: isinstance(x, PIL.Image.Image)  # noqa: E731
        is_pil_list = lambda x: isinstance(x, list) and all(isinstance(v, PIL.Image.Image) for v in x)  # noqa: E731
        image, goal, video = _one(image, is_pil), _one(goal, is_pil), _one(video, is_pil_list)
        if raymap is not None:
            if isinstance(raymap, np.ndarray):
                raymap = torch.from_numpy(raymap).to(dev, dtype=torch.bfloat16)
            if raymap.ndim == 4:
                raymap = raymap.unsqueeze(0).to(dev, dtype=torch.bfloat16)
        return image, goal, video, raymap

    # -------------------------------------------------------------------------------------------------
    def _encode_frames(self, frames_bcfhw: torch.Tensor, generator, batch_size: int, dtype) -> torch.Tensor:
        """VAE-encode each batch item separately (as P:554-569 does), sample the posterior with the caller's generator,
        return [B, F_latent, 16, h, w] scaled by the VAE scaling factor (P:571-576)."""
        if isinstance(generator, list):
            lat = [retrieve_latents(self.vae.encode(frames_bcfhw[i].unsqueeze(0)), generator[i]) for i in range(batch_size)]
        else:
            lat = [retrieve_latents(self.vae.encode(x.unsqueeze(0)), generator) for x in frames_bcfhw]
        lat = torch.cat(lat, dim=0).to(dtype).permute(0, 2, 1, 3, 4)
        if not self.vae.config.invert_scale_latents:
            return self.vae_scaling_factor_image * lat
        return 1 / self.vae_scaling_factor_image * lat

    @torch.no_grad()
    def prepare_latents(self, image=None, goal=None, video=None, raymap=None, batch_size: int = 1, num_frames: int = 13,
                        height: int = 60, width: int = 90, dtype=None, device=None, generator=None):
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective batch"
                             f" size of {batch_size}. Make sure the batch size matches the length of the generators.")
        s, ts = self.vae_scale_factor_spatial, self.vae_scale_factor_temporal
        lat_frames = (num_frames - 1) // ts + 1
        shape = (batch_size, lat_frames, self._num_output_channels, height // s, width // s)

        # conditions, in the reference's RNG order: image, goal, video (P:552-631)
        image_latents = goal_latents = video_latents = None
        if image is not None:
            image_latents = self._encode_frames(image.unsqueeze(2), generator, batch_size, dtype)
        if goal is not None:
            goal_latents = self._encode_frames(goal.unsqueeze(2), generator, batch_size, dtype)
        if video is not None:
            if video.ndim == 4:
                video = video.unsqueeze(0)
            video_latents = self._encode_frames(video.permute(0, 2, 1, 3, 4), generator, batch_size, dtype)

        if image is not None and goal is None:      # prediction: first frame known, the rest zero (P:633-640)
            pad = torch.zeros((batch_size, lat_frames - image_latents.shape[1], *image_latents.shape[2:]), device=device, dtype=dtype)
            condition_latents = torch.cat([image_latents, pad], dim=1)
        elif goal is not None:                      # planning: first and last frame known (P:641-648)
            gap = lat_frames - goal_latents.shape[1] - image_latents.shape[1]
            pad = torch.zeros((batch_size, gap, *image_latents.shape[2:]), device=device, dtype=dtype)
            condition_latents = torch.cat([image_latents, pad, goal_latents], dim=1)
        elif video is not None:                     # reconstruction (P:649-650)
            condition_latents = video_latents

        if raymap is not None:
            rem = raymap.shape[1] % ts
            if rem != 0:                            # front-pad by repeating the first frames (P:653-665)
                raymap = torch.cat([raymap[:, : ts - rem], raymap], dim=1)
            # n is the OUTER factor: latent frame t packs raw frames {t, T+t, 2T+t, 3T+t} (P:666-670)
            camera_conditions = rearrange(raymap, "b (n t) c h w -> b t (n c) h w", n=ts)
        else:
            camera_conditions = torch.zeros(batch_size, lat_frames, 24, height // s, width // s, device=device, dtype=dtype)

        condition_latents = torch.cat([condition_latents, camera_conditions], dim=2)
        latents = randn_tensor(shape, device=device, generator=generator, dtype=dtype)
        latents = latents * self.scheduler.init_noise_sigma
        return latents, condition_latents

    # -------------------------------------------------------------------------------------------------
    # ---- two-rank classifier-free-guidance split (not in the reference, which is single-process) -----------
    _cfg_group = None
    _cfg_rank = 0

    def enable_cfg_parallel(self, group=None) -> None:
        """Prediction / planning run the transformer on a batch of two (unconditional, conditional: P:832-859).  With a
        two-rank process group, rank 0 of the group evaluates the unconditional branch and rank 1 the conditional one, each at
        batch 1; the bf16 `noise_pred` halves (6.65 MB at 41x480x720) are exchanged with one all-gather per step and BOTH ranks
        then perform the identical guidance combination and scheduler step, so the latents stay replicated without a
        broadcast.  After the loop rank 0 decodes the rgb latents and rank 1 the disparity latents (one more all-gather).
        Callers must give both ranks the same inputs and a generator with the same seed.  No effect when guidance is off."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("enable_cfg_parallel needs an initialised torch.distributed process group")
        if dist.get_world_size(group) != 2:
            raise ValueError(f"cfg-parallel needs a process group of exactly two ranks, got {dist.get_world_size(group)}")
        self._cfg_group = group if group is not None else dist.group.WORLD
        self._cfg_rank = dist.get_rank(group)

    def disable_cfg_parallel(self) -> None:
        self._cfg_group, self._cfg_rank = None, 0

    # ---- two-rank split of the two final decodes for ANY task (single-clip latency; not in the reference) -------------------------
    _dec_group = None
    _dec_rank = 0

    def enable_decode_parallel(self, group=None) -> None:
        """The two final VAE decodes (rgb latents P:931, disparity latents P:936) are independent and are 40 % of the reference-default 4-step
        reconstruction clip.  With a two-rank group BOTH ranks make every call with identical inputs and equally seeded generators — the
        encode and the sampling loop run replicated, so the latents are bit-identical on both — then rank 0 decodes the rgb latents, rank 1
        the disparity latents, and one all-gather (85 MB of bf16 per rank at 41 x 480 x 720) gives both ranks both videos.  Same kernels on
        the same latents (the final latents are broadcast from the pair's first rank before the split, so the rgb / disparity / raymap triple is
        consistent even if the ranks were seeded differently): outputs are bit-identical to the one-rank call.  Unlike `enable_cfg_parallel` this changes reconstruction calls
        too, so a call made by one rank only would wait for its peer forever."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("enable_decode_parallel needs an initialised torch.distributed process group")
        if dist.get_world_size(group) != 2:
            raise ValueError(f"decode-parallel needs a process group of exactly two ranks, got {dist.get_world_size(group)}")
        self._dec_group = group if group is not None else dist.group.WORLD
        self._dec_rank = dist.get_rank(group)

    def disable_decode_parallel(self) -> None:
        self._dec_group, self._dec_rank = None, 0

    def _gather_pair(self, x: torch.Tensor, group=None) -> torch.Tensor:
        """[1, ...] on each rank of the pair -> [2, ...] in group-rank order on both (one part per rank of the group: a one-rank group —
        the RCCL configuration a one-GPU box can exercise — returns its own part)."""
        import torch.distributed as dist
        group = self._cfg_group if group is None else group
        parts = [torch.empty_like(x, memory_format=torch.contiguous_format) for _ in range(dist.get_world_size(group))]
        dist.all_gather(parts, x.contiguous(), group=group)
        return torch.cat(parts)

    def _unconditional(self, task: str, condition_latents: torch.Tensor, goal) -> torch.Tensor:
        """Classifier-free branch: drop the observed RGB latents (P:839-855)."""
        nz = self.vae.config.latent_channels
        uncond = condition_latents.clone()
        if task == "planning":
            assert goal is not None
            uncond[:, :, :nz] = 0
        elif task == "prediction":
            uncond[:, :1, :nz] = 0
        else:
            raise ValueError(f"Task {task} not supported for classifier-free guidance.")
        return torch.cat([uncond, condition_latents])

    @torch.no_grad()
    def __call__(self, task: Optional[str] = None, image=None, video=None, goal=None,
                 raymap: Optional[Union[torch.Tensor, np.ndarray]] = None, height: Optional[int] = None,
                 width: Optional[int] = None, num_frames: Optional[int] = None, num_inference_steps: Optional[int] = None,
                 timesteps: Optional[List[int]] = None, guidance_scale: Optional[float] = None, use_dynamic_cfg: bool = False,
                 num_videos_per_prompt: int = 1, eta: float = 0.0,
                 generator: Optional[Union[torch.Generator, List[torch.Generator]]] = None, return_dict: bool = True,
                 attention_kwargs: Optional[Dict] = None, fps: Optional[int] = None):
        if task is None:
            task = "reconstruction" if video is not None else ("planning" if goal is not None else "prediction")
        tcfg = self.transformer.config
        height = height or tcfg.sample_height * self.vae_scale_factor_spatial
        width = width or tcfg.sample_width * self.vae_scale_factor_spatial
        num_frames = num_frames or tcfg.sample_frames
        fps = fps or self._base_fps
        num_videos_per_prompt = 1

        self.check_inputs(task=task, image=image, video=video, goal=goal, raymap=raymap, height=height, width=width,
                          num_frames=num_frames, fps=fps)
        image, goal, video, raymap = self.preprocess_inputs(image=image, goal=goal, video=video, raymap=raymap, height=height,
                                                            width=width, num_frames=num_frames)
        self._guidance_scale = guidance_scale
        self._current_timestep = None
        self._attention_kwargs = attention_kwargs
        self._interrupt = False
        batch_size = 1
        device = self._execution_device
        # one VAE workspace for every task at this geometry (clip encode, one-frame image / goal encode, latent-clip decode): reserved before the
        # first encode so that neither this call nor a later call of another task re-allocates it (and drops captured hipGraphs)
        if hasattr(self.vae, "reserve_workspace"):
            self.vae.reserve_workspace(num_frames, height, width)

        prompt_embeds = self.empty_prompt_embeds.to(device)
        num_inference_steps = num_inference_steps or self._default_num_inference_steps[task]
        guidance_scale = guidance_scale or self._default_guidance_scale[task]
        use_dynamic_cfg = use_dynamic_cfg or self._default_use_dynamic_cfg[task]
        do_cfg = guidance_scale > 1.0

        timesteps, num_inference_steps = retrieve_timesteps(self.scheduler, num_inference_steps, device, timesteps)
        self._num_timesteps = len(timesteps)

        latents, condition_latents = self.prepare_latents(image, goal, video, raymap, batch_size * num_videos_per_prompt,
                                                          num_frames, height, width, prompt_embeds.dtype, device, generator)
        extra_step_kwargs = self.prepare_extra_step_kwargs(generator, eta)
        rope = (self._prepare_rotary_positional_embeddings(height, width, latents.size(1), device, fps=fps)
                if tcfg.use_rotary_positional_embeddings else None)
        ofs_emb = None if tcfg.ofs_embed_dim is None else latents.new_full((1,), fill_value=2.0)

        # host copy of the schedule: the dynamic-CFG scalar needs t as a Python number every step (P:886)
        t_host = [int(t) for t in timesteps.tolist()]
        dpm = isinstance(self.scheduler, CogVideoXDPMScheduler) or type(self.scheduler).__name__ == "CogVideoXDPMScheduler"
        n_warm = max(len(timesteps) - num_inference_steps * self.scheduler.order, 0)
        latent_condition = self._unconditional(task, condition_latents, goal) if do_cfg else condition_latents
        split = do_cfg and self._cfg_group is not None            # this rank evaluates one guidance branch at batch 1
        if split:
            latent_condition = latent_condition[self._cfg_rank:self._cfg_rank + 1]
        text = prompt_embeds.repeat(2 if (do_cfg and not split) else 1, 1, 1)
        # the element-wise tail of a step as one HIP kernel: this repo's DPM scheduler, v-prediction, bf16 latents on an MI355X, and a
        # guidance scale that is not a per-sample tensor; anything else takes the reference's PyTorch sequence
        fused_tail = (self.fuse_step_tail and dpm and hasattr(self.scheduler, "step_fused") and latents.is_cuda and latents.dtype == torch.bfloat16
                      and getattr(self.scheduler.config, "prediction_type", None) == "v_prediction" and latents.shape[0] == 1
                      and latents.numel() % 8 == 0)            # the kernel moves 16-byte pieces; other sizes take the PyTorch sequence

        with self.progress_bar(total=num_inference_steps) as bar:
            old_x0 = None
            for i, t in enumerate(timesteps):
                if self.interrupt:
                    continue
                self._current_timestep = t
                model_in = torch.cat([latents] * 2) if (do_cfg and not split) else latents
                model_in = self.scheduler.scale_model_input(model_in, t)
                model_in = torch.cat([model_in, latent_condition], dim=2)                        # P:857-859 -> 96 channels
                noise_pred = self.transformer(hidden_states=model_in, encoder_hidden_states=text,
                                              timestep=t.expand(model_in.shape[0]), ofs=ofs_emb, image_rotary_emb=rope,
                                              attention_kwargs=attention_kwargs, return_dict=False)[0]
                if split:
                    noise_pred = self._gather_pair(noise_pred)                                   # (unconditional, conditional)
                if use_dynamic_cfg:
                    # the reference feeds the raw timestep value (999 ... 19) here, literally (P:880-893)
                    frac = (num_inference_steps - t_host[i]) / num_inference_steps
                    self._guidance_scale = 1 + guidance_scale * ((1 - math.cos(math.pi * frac ** 5.0)) / 2)
                if fused_tail and noise_pred.dtype == torch.bfloat16:
                    # P:877-916 (fp32 cast, guidance combine, scheduler.step, cast back) as ONE kernel — bit-identical to the sequence below
                    latents, old_x0 = self.scheduler.step_fused(noise_pred, old_x0, t_host[i], t_host[i - 1] if i > 0 else None, latents,
                                                                guidance_scale=self.guidance_scale if do_cfg else None,
                                                                generator=extra_step_kwargs.get("generator"))
                else:
                    noise_pred = noise_pred.float()
                    if do_cfg:
                        uncond, cond = noise_pred.chunk(2)
                        noise_pred = uncond + self.guidance_scale * (cond - uncond)
                    if not dpm:
                        latents = self.scheduler.step(noise_pred, t, latents, **extra_step_kwargs, return_dict=False)[0]
                    else:
                        latents, old_x0 = self.scheduler.step(noise_pred, old_x0, t, timesteps[i - 1] if i > 0 else None, latents,
                                                              **extra_step_kwargs, return_dict=False)
                    latents = latents.to(prompt_embeds.dtype)
                if i == len(timesteps) - 1 or ((i + 1) > n_warm and (i + 1) % self.scheduler.order == 0):
                    bar.update()
        self._current_timestep = None
        self._final_latents = latents             # extension: what the loop ended on (parity tests compare it; P:921)

        if self._dec_group is not None and not split:
            # decode-parallel: both ranks must decode the SAME latents.  Replicated calls with equal seeds give that by construction; a caller that
            # passed generator=None or different seeds would otherwise get rgb from one trajectory and disparity from another, silently.  One
            # broadcast of the final latents from the pair's first rank (6.7 MB of bf16 at 41 x 480 x 720, against the 85 MB all-gather of the
            # decoded clips) makes the triple consistent whatever the ranks drew; with equal seeds it moves the bits the peer already has.
            import torch.distributed as dist
            latents = latents.contiguous()
            dist.broadcast(latents, src=dist.get_global_rank(self._dec_group, 0), group=self._dec_group)
            self._final_latents = latents
        nz = self.vae.config.latent_channels
        rgb_latents, disparity_latents, camera_latents = latents[:, :, :nz], latents[:, :, nz:2 * nz], latents[:, :, 2 * nz:]
        if split:
            rgb_decoded, disparity_decoded = self._gather_pair(
                self.decode_latents(rgb_latents if self._cfg_rank == 0 else disparity_latents)).split(1)
        elif self._dec_group is not None:
            rgb_decoded, disparity_decoded = self._gather_pair(
                self.decode_latents(rgb_latents if self._dec_rank == 0 else disparity_latents), self._dec_group).split(1)
        elif self.decode_concurrently and hasattr(self.vae, "decode_pair") and latents.is_cuda:
            # the two decodes of P:931,936 through aether_amd.vae.AetherVAE.decode_pair: same kernels, bit-identical results
            inv = 1 / self.vae_scaling_factor_image
            try:
                rgb_decoded, disparity_decoded = self.vae.decode_pair(inv * rgb_latents.permute(0, 2, 1, 3, 4), inv * disparity_latents.permute(0, 2, 1, 3, 4))
            except torch.OutOfMemoryError:
                # the second decode context needs a workspace of its own: without room for it the two decodes run one after the other
                # (the reference's order), and the pair is not tried again on this pipeline
                self.decode_concurrently = False
                if hasattr(self.vae, "_twin"):
                    self.vae._twin = None
                torch.cuda.empty_cache()
                rgb_decoded, disparity_decoded = self.decode_latents(rgb_latents), self.decode_latents(disparity_latents)
        else:
            rgb_decoded, disparity_decoded = self.decode_latents(rgb_latents), self.decode_latents(disparity_latents)
        disparity_video = disparity_decoded.mean(dim=1, keepdim=False)
        disparity_video = torch.square(disparity_video * 0.5 + 0.5).float()
        if self.keep_outputs_on_device:
            rgb_video = self.video_processor.postprocess_video(video=rgb_decoded, output_type="pt")          # [B,F,C,H,W], device
            rgb_video = rgb_video.permute(0, 1, 3, 4, 2).float().contiguous()                                 # [B,F,H,W,C] like "np"
        else:
            rgb_video = self.video_processor.postprocess_video(video=rgb_decoded, output_type="np")
            disparity_video = disparity_video.cpu().numpy()
        raymap_out = rearrange(camera_latents, "b t (n c) h w -> b (n t) c h w", n=4)[:, -rgb_video.shape[1]:, :, :]
        raymap_out = raymap_out.float() if self.keep_outputs_on_device else raymap_out.float().cpu().numpy()
        self.maybe_free_model_hooks()
        if not return_dict:
            return rgb_video, disparity_video, raymap_out
        return AetherV1PipelineOutput(rgb=rgb_video.squeeze(0), disparity=disparity_video.squeeze(0), raymap=raymap_out.squeeze(0))
