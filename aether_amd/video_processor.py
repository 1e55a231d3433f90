"""Minimal image/video pre- and post-processing with the semantics of diffusers' `VideoProcessor`
(`VaeImageProcessor.preprocess` / `postprocess_video`) that the reference calls at
/root/reference/aether/pipelines/aetherv1_pipeline_cogvideox.py:459,474-496,932 (UPSTREAM-UNVERIFIED):
numpy/tensor inputs -> [N,C,H,W] in [-1,1] (nearest resize if the size differs); PIL inputs with
resize_mode="crop" -> resize-to-cover (Lanczos) + centre crop; outputs -> float32 numpy [B,F,H,W,C] in [0,1]."""
from __future__ import annotations

from typing import List, Union

import numpy as np
import PIL.Image
import torch
import torch.nn.functional as F


class VideoProcessor:
    def __init__(self, vae_scale_factor: int = 8, do_resize: bool = True, do_normalize: bool = True):
        self.vae_scale_factor, self.do_resize, self.do_normalize = vae_scale_factor, do_resize, do_normalize

    @staticmethod
    def _resize_and_crop(img: PIL.Image.Image, width: int, height: int) -> PIL.Image.Image:
        ratio, src_ratio = width / height, img.width / img.height
        src_w = width if ratio > src_ratio else img.width * height // img.height
        src_h = height if ratio <= src_ratio else img.height * width // img.width
        resized = img.resize((src_w, src_h), resample=PIL.Image.LANCZOS)
        out = PIL.Image.new("RGB", (width, height))
        out.paste(resized, box=(width // 2 - src_w // 2, height // 2 - src_h // 2))
        return out

    def preprocess(self, image, height: int = None, width: int = None, resize_mode: str = "default") -> torch.Tensor:
        if isinstance(image, (PIL.Image.Image, np.ndarray, torch.Tensor)):
            image = [image]
        first = image[0]
        if isinstance(first, PIL.Image.Image):
            if self.do_resize:
                if resize_mode == "crop":
                    image = [self._resize_and_crop(i, width, height) for i in image]
                elif resize_mode == "default":
                    image = [i.resize((width, height), resample=PIL.Image.LANCZOS) for i in image]
                else:
                    raise ValueError(f"resize_mode {resize_mode} is not supported")
            arr = np.stack([np.array(i.convert("RGB")).astype(np.float32) / 255.0 for i in image], axis=0)
            t = torch.from_numpy(arr.transpose(0, 3, 1, 2))
        elif isinstance(first, np.ndarray):
            arr = np.concatenate(image, axis=0) if first.ndim == 4 else np.stack(image, axis=0)
            if arr.ndim == 3:
                arr = arr[..., None]
            t = torch.from_numpy(arr.transpose(0, 3, 1, 2))
            if self.do_resize and tuple(t.shape[-2:]) != (height, width):
                t = F.interpolate(t, size=(height, width))
        elif isinstance(first, torch.Tensor):
            t = torch.cat(image, dim=0) if first.ndim == 4 else torch.stack(image, dim=0)
            if self.do_resize and tuple(t.shape[-2:]) != (height, width):
                t = F.interpolate(t, size=(height, width))
        else:
            raise ValueError(f"Input is in incorrect format: {type(first)}")
        if self.do_normalize:
            t = 2.0 * t - 1.0
        return t

    def postprocess_video(self, video: torch.Tensor, output_type: str = "np") -> Union[np.ndarray, torch.Tensor, List]:
        """video [B,C,F,H,W] in [-1,1] -> [B,F,H,W,C] float32 in [0,1] ("np") or [B,F,C,H,W] tensor ("pt")."""
        outs = []
        for b in range(video.shape[0]):
            frames = (video[b].permute(1, 0, 2, 3) / 2 + 0.5).clamp(0, 1)
            # permute + bf16->fp32 (exact) on the device, then ONE contiguous D2H copy (upstream converts on the host after the copy)
            outs.append(frames.permute(0, 2, 3, 1).float().contiguous().cpu().numpy() if output_type == "np" else frames)
        if output_type == "np":
            return outs[0][None] if len(outs) == 1 else np.stack(outs)
        if output_type == "pt":
            return torch.stack(outs)
        raise ValueError(f"{output_type} does not exist. Please choose one of ['np', 'pt']")
