"""ORACLE (test infrastructure only): 3-D rotary tables exactly as the reference builds them.
PINNED: equal to the outputs of the reference's own get_3d_rotary_pos_embed / get_resize_crop_region_for_grid (P:25-163) on the
fixtures of tests/golden/pipeline.npz (tests/test_reference_pins_cpu.py), given the published formula of diffusers' get_1d_rotary_pos_embed.

Follows /root/reference/aether/pipelines/aetherv1_pipeline_cogvideox.py:25-144 (`get_3d_rotary_pos_embed`,
whose only change w.r.t. diffusers is `fps_factor` scaling the temporal positions, P:35,81-90), P:148-163
(`get_resize_crop_region_for_grid`) and P:299-348 (`_prepare_rotary_positional_embeddings`), plus diffusers'
`get_1d_rotary_pos_embed(dim, pos, theta, use_real=True)` which the reference imports at P:16 and calls at
P:108-111 (restated from diffusers 0.32 models/embeddings.py; UPSTREAM-UNVERIFIED).
"""
from __future__ import annotations

import torch


def rope_1d(dim: int, pos: torch.Tensor, theta: float = 10000.0):
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float32)[: dim // 2] / dim))
    ang = torch.outer(pos.float(), freqs)
    return ang.cos().repeat_interleave(2, dim=1).float(), ang.sin().repeat_interleave(2, dim=1).float()


def crop_region_for_grid(src, tgt_width, tgt_height):
    """P:148-163."""
    h, w = src
    if h / w > tgt_height / tgt_width:
        rh, rw = tgt_height, int(round(tgt_height / h * w))
    else:
        rw, rh = tgt_width, int(round(tgt_width / w * h))
    top, left = int(round((tgt_height - rh) / 2.0)), int(round((tgt_width - rw) / 2.0))
    return (top, left), (top + rh, left + rw)


def rope_3d(embed_dim, crops_coords, grid_size, temporal_size, theta=10000.0, fps_factor=1.0):
    """P:25-144, grid_type == "linspace" branch. Returns cos, sin of shape [T*H*W, embed_dim]."""
    (top, left), (bottom, right) = crops_coords
    gh, gw = grid_size
    grid_h = torch.linspace(top, bottom * (gh - 1) / gh, gh, dtype=torch.float32)
    grid_w = torch.linspace(left, right * (gw - 1) / gw, gw, dtype=torch.float32)
    grid_t = torch.linspace(0, temporal_size * (temporal_size - 1) / temporal_size, temporal_size, dtype=torch.float32) * fps_factor
    dim_t, dim_h, dim_w = embed_dim // 4, embed_dim // 8 * 3, embed_dim // 8 * 3
    parts = []
    for which in (0, 1):
        ft = rope_1d(dim_t, grid_t, theta)[which][:, None, None, :].expand(-1, gh, gw, -1)
        fh = rope_1d(dim_h, grid_h, theta)[which][None, :, None, :].expand(temporal_size, -1, gw, -1)
        fw = rope_1d(dim_w, grid_w, theta)[which][None, None, :, :].expand(temporal_size, gh, -1, -1)
        parts.append(torch.cat([ft, fh, fw], dim=-1).reshape(temporal_size * gh * gw, -1))
    return parts[0], parts[1]


def prepare_rope(height, width, latent_frames, fps, *, patch_size=2, sample_height=60, sample_width=90, head_dim=64,
                 vae_scale_factor_spatial=8, base_fps=12):
    """P:299-332 (patch_size_t is None, "CogVideoX 1.0" branch)."""
    gh = height // (vae_scale_factor_spatial * patch_size)
    gw = width // (vae_scale_factor_spatial * patch_size)
    crops = crop_region_for_grid((gh, gw), sample_width // patch_size, sample_height // patch_size)
    return rope_3d(head_dim, crops, (gh, gw), latent_frames, fps_factor=base_fps / fps)
