"""3-D rotary tables for the AetherV1 DiT (host side, computed once per pipeline call).

Same arithmetic, in the same fp32 order, as the reference's `get_3d_rotary_pos_embed`
(/root/reference/aether/pipelines/aetherv1_pipeline_cogvideox.py:25-144, "linspace" grid) with Aether's
`fps_factor` on the temporal axis (P:81-90), `get_resize_crop_region_for_grid` (P:148-163) and the channel split
16 | 24 | 24 of head_dim 64 (P:103-105).  Output: cos, sin fp32 [T*H*W, head_dim], token order (t, h, w),
adjacent pairs sharing one angle (diffusers get_1d_rotary_pos_embed(use_real=True) repeat-interleave layout).
"""
from __future__ import annotations

from typing import Tuple

import torch


def resize_crop_region_for_grid(src: Tuple[int, int], tgt_width: int, tgt_height: int):
    h, w = src
    if h / w > tgt_height / tgt_width:
        new_h, new_w = tgt_height, int(round(tgt_height / h * w))
    else:
        new_h, new_w = int(round(tgt_width / w * h)), tgt_width
    top = int(round((tgt_height - new_h) / 2.0))
    left = int(round((tgt_width - new_w) / 2.0))
    return (top, left), (top + new_h, left + new_w)


def _axis_table(dim: int, positions: torch.Tensor, theta: float):
    inv = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float32, device=positions.device)[: dim // 2] / dim))
    angles = torch.outer(positions, inv)
    return angles.cos().repeat_interleave(2, dim=1).float(), angles.sin().repeat_interleave(2, dim=1).float()


def rotary_tables_3d(head_dim: int, crop, grid_hw: Tuple[int, int], frames: int, fps_factor: float = 1.0,
                     theta: float = 10000.0, device=None):
    (top, left), (bottom, right) = crop
    gh, gw = grid_hw
    kw = dict(device=device, dtype=torch.float32)
    pos_h = torch.linspace(top, bottom * (gh - 1) / gh, gh, **kw)
    pos_w = torch.linspace(left, right * (gw - 1) / gw, gw, **kw)
    pos_t = torch.linspace(0, frames * (frames - 1) / frames, frames, **kw) * fps_factor
    d_t, d_hw = head_dim // 4, head_dim // 8 * 3
    tabs = [_axis_table(d_t, pos_t, theta), _axis_table(d_hw, pos_h, theta), _axis_table(d_hw, pos_w, theta)]
    out = []
    for k in (0, 1):  # cos, sin
        t = tabs[0][k].view(frames, 1, 1, d_t).expand(frames, gh, gw, d_t)
        h = tabs[1][k].view(1, gh, 1, d_hw).expand(frames, gh, gw, d_hw)
        w = tabs[2][k].view(1, 1, gw, d_hw).expand(frames, gh, gw, d_hw)
        out.append(torch.cat([t, h, w], dim=-1).reshape(frames * gh * gw, head_dim).contiguous())
    return out[0], out[1]
