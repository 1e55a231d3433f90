"""ORACLE (test infrastructure only — never imported by the product path in aether_amd/).

CPU restatement, in plain PyTorch, of the diffusion transformer the reference calls at
/root/reference/aether/pipelines/aetherv1_pipeline_cogvideox.py:865-875, i.e. diffusers'
`CogVideoXTransformer3DModel.forward` (third-party dependency `diffusers>=0.32.2`,
/root/reference/requirements.txt:4 — NOT vendored under /root/reference and not installable here).

PARITY UNPINNED: the reference ships no tests/golden vectors for this path and diffusers cannot be imported in
the build container, so the algorithm below is restated from the published diffusers 0.32 sources
(models/transformers/cogvideox_transformer_3d.py, models/embeddings.py, models/normalization.py,
models/attention_processor.py — see SURVEY.md Appendix A.1) and is pinned only by hand-computed known answers
(tests/test_oracle_dit.py) and by the reference's own RoPE formula (P:25-144, restated in oracle/rope.py).
Module and parameter names reproduce diffusers' state-dict keys so a real checkpoint loads unchanged.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass
class DitConfig:
    """Field names follow transformer/config.json of a diffusers CogVideoX checkpoint."""
    num_attention_heads: int = 48
    attention_head_dim: int = 64
    in_channels: int = 96            # 56 noisy + 16 cond latent + 24 raymap  (P:539,682,857-859)
    out_channels: int = 56           # P:539, split at P:925-929
    num_layers: int = 42
    patch_size: int = 2
    patch_size_t: Optional[int] = None
    text_embed_dim: int = 4096
    time_embed_dim: int = 512
    max_text_seq_length: int = 226
    sample_width: int = 90
    sample_height: int = 60
    sample_frames: int = 41
    temporal_compression_ratio: int = 4
    norm_eps: float = 1e-5
    use_rotary_positional_embeddings: bool = True
    use_learned_positional_embeddings: bool = False
    ofs_embed_dim: Optional[int] = None
    flip_sin_to_cos: bool = True
    freq_shift: int = 0
    activation_fn: str = "gelu-approximate"
    timestep_activation_fn: str = "silu"
    attention_bias: bool = True
    spatial_interpolation_scale: float = 1.875     # CogVideoXTransformer3DModel.__init__ defaults
    temporal_interpolation_scale: float = 1.0

    @property
    def inner_dim(self) -> int:
        return self.num_attention_heads * self.attention_head_dim


def timestep_sinusoid(t: torch.Tensor, dim: int, flip_sin_to_cos: bool = True, freq_shift: float = 0.0) -> torch.Tensor:
    """diffusers.models.embeddings.get_timestep_embedding (max_period 10000, scale 1)."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device)
    exponent = exponent / (half - freq_shift)
    emb = t[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


def apply_rotary_emb(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """diffusers apply_rotary_emb(use_real=True, use_real_unbind_dim=-1): adjacent-pair rotation, fp32 math.
    x [B,H,S,D]; cos,sin [S,D]."""
    cos = cos[None, None].to(x.device)
    sin = sin[None, None].to(x.device)
    x_real, x_imag = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    x_rot = torch.stack([-x_imag, x_real], dim=-1).flatten(3)
    return (x.float() * cos + x_rot.float() * sin).to(x.dtype)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_dim: int, dim: int):
        super().__init__()
        self.linear_1 = nn.Linear(in_dim, dim)
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


def sincos_1d(dim: int, pos: torch.Tensor) -> torch.Tensor:
    """diffusers get_1d_sincos_pos_embed_from_grid (float64, [sin | cos], flip_sin_to_cos=False)."""
    omega = torch.arange(dim // 2, dtype=torch.float64) / (dim / 2.0)
    omega = 1.0 / 10000 ** omega
    out = torch.outer(pos.reshape(-1).double(), omega)
    return torch.cat([torch.sin(out), torch.cos(out)], dim=1)


def sincos_pos_embed_3d(dim: int, width: int, height: int, frames: int, spatial_scale: float, temporal_scale: float) -> torch.Tensor:
    """diffusers get_3d_sincos_pos_embed(embed_dim, (width, height), frames, ...) -> [frames, height*width, dim]:
    per token [temporal dim/4 | spatial 3dim/4], the spatial part = [emb(grid_w) | emb(grid_h)] (meshgrid 'xy', w first)."""
    ds, dt = 3 * dim // 4, dim // 4
    gh = torch.arange(height, dtype=torch.float32) / spatial_scale
    gw = torch.arange(width, dtype=torch.float32) / spatial_scale
    mw, mh = torch.meshgrid(gw, gh, indexing="xy")            # both [height, width]
    spatial = torch.cat([sincos_1d(ds // 2, mw), sincos_1d(ds // 2, mh)], dim=1)          # [H*W, ds]
    temporal = sincos_1d(dt, torch.arange(frames, dtype=torch.float32) / temporal_scale)   # [T, dt]
    return torch.cat([temporal[:, None, :].expand(frames, height * width, dt), spatial[None].expand(frames, height * width, ds)], dim=-1)


class PatchEmbed(nn.Module):
    """CogVideoXPatchEmbed, patch_size_t=None branch."""

    def __init__(self, cfg: DitConfig):
        super().__init__()
        self.cfg = cfg
        D, p = cfg.inner_dim, cfg.patch_size
        self.proj = nn.Conv2d(cfg.in_channels, D, kernel_size=(p, p), stride=p, bias=True)
        self.text_proj = nn.Linear(cfg.text_embed_dim, D)
        if cfg.use_learned_positional_embeddings:
            n_patches = (cfg.sample_height // p) * (cfg.sample_width // p) * ((cfg.sample_frames - 1) // cfg.temporal_compression_ratio + 1)
            self.pos_embedding = nn.Parameter(torch.zeros(1, cfg.max_text_seq_length + n_patches, D))

    def forward(self, text_embeds, image_embeds):
        text_embeds = self.text_proj(text_embeds)
        B, Fr, Cc, H, W = image_embeds.shape
        x = image_embeds.reshape(-1, Cc, H, W)
        x = self.proj(x)
        x = x.view(B, Fr, *x.shape[1:])
        x = x.flatten(3).transpose(2, 3)   # [B, F, H*W/p^2, D]
        x = x.flatten(1, 2)                # token order (frame, row, col)
        embeds = torch.cat([text_embeds, x], dim=1).contiguous()
        cfg = self.cfg
        if cfg.use_learned_positional_embeddings or not cfg.use_rotary_positional_embeddings:
            # diffusers CogVideoXPatchEmbed.forward (0.32) [UPSTREAM-UNVERIFIED]: the learned table needs the sample resolution;
            # when the FRAME count differs from sample_frames (AetherV1 runs 41 frames on a base whose sample_frames is 49)
            # it does not slice the learned table: it recomputes the 3-D sin-cos table for the actual size and adds that.
            if cfg.use_learned_positional_embeddings and (cfg.sample_width != W or cfg.sample_height != H):
                raise ValueError("It is currently not possible to generate videos at a different resolution that the defaults.")
            pre_frames = (Fr - 1) * cfg.temporal_compression_ratio + 1
            if cfg.sample_height != H or cfg.sample_width != W or cfg.sample_frames != pre_frames or not cfg.use_learned_positional_embeddings:
                p = cfg.patch_size
                pe = sincos_pos_embed_3d(cfg.inner_dim, W // p, H // p, Fr, cfg.spatial_interpolation_scale, cfg.temporal_interpolation_scale)
                pos = torch.zeros(1, cfg.max_text_seq_length + pe.shape[0] * pe.shape[1], cfg.inner_dim, dtype=pe.dtype)
                pos[:, cfg.max_text_seq_length:] = pe.flatten(0, 1)
            else:
                pos = self.pos_embedding
            embeds = embeds + pos.to(embeds.dtype)
        return embeds


class LayerNormZero(nn.Module):
    """CogVideoXLayerNormZero: chunk order shift, scale, gate, enc_shift, enc_scale, enc_gate."""

    def __init__(self, cond_dim: int, dim: int, eps: float):
        super().__init__()
        self.silu = nn.SiLU()
        self.linear = nn.Linear(cond_dim, 6 * dim)
        self.norm = nn.LayerNorm(dim, eps=eps, elementwise_affine=True)

    def forward(self, h, e, temb):
        shift, scale, gate, eshift, escale, egate = self.linear(self.silu(temb)).chunk(6, dim=1)
        h = self.norm(h) * (1 + scale)[:, None, :] + shift[:, None, :]
        e = self.norm(e) * (1 + escale)[:, None, :] + eshift[:, None, :]
        return h, e, gate[:, None, :], egate[:, None, :]


class Attention(nn.Module):
    """Attention(qk_norm="layer_norm", bias=True) driven by CogVideoXAttnProcessor2_0."""

    def __init__(self, dim: int, heads: int, head_dim: int):
        super().__init__()
        self.heads, self.head_dim = heads, head_dim
        self.to_q = nn.Linear(dim, dim)
        self.to_k = nn.Linear(dim, dim)
        self.to_v = nn.Linear(dim, dim)
        self.norm_q = nn.LayerNorm(head_dim, eps=1e-6, elementwise_affine=True)
        self.norm_k = nn.LayerNorm(head_dim, eps=1e-6, elementwise_affine=True)
        self.to_out = nn.ModuleList([nn.Linear(dim, dim), nn.Dropout(0.0)])

    def forward(self, h, e, rope):
        n_text = e.shape[1]
        x = torch.cat([e, h], dim=1)
        B, S, _ = x.shape
        q = self.to_q(x).view(B, S, self.heads, self.head_dim).transpose(1, 2)
        k = self.to_k(x).view(B, S, self.heads, self.head_dim).transpose(1, 2)
        v = self.to_v(x).view(B, S, self.heads, self.head_dim).transpose(1, 2)
        q = self.norm_q(q)
        k = self.norm_k(k)
        if rope is not None:
            cos, sin = rope
            q = torch.cat([q[:, :, :n_text], apply_rotary_emb(q[:, :, n_text:], cos, sin)], dim=2)
            k = torch.cat([k[:, :, :n_text], apply_rotary_emb(k[:, :, n_text:], cos, sin)], dim=2)
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False)
        o = o.transpose(1, 2).reshape(B, S, self.heads * self.head_dim)
        o = self.to_out[0](o)
        return o[:, n_text:], o[:, :n_text]


class GELUProj(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, inner)

    def forward(self, x):
        return F.gelu(self.proj(x), approximate="tanh")


class FeedForward(nn.Module):
    def __init__(self, dim: int, mult: int = 4):
        super().__init__()
        self.net = nn.ModuleList([GELUProj(dim, dim * mult), nn.Dropout(0.0), nn.Linear(dim * mult, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class Block(nn.Module):
    def __init__(self, cfg: DitConfig):
        super().__init__()
        D = cfg.inner_dim
        self.norm1 = LayerNormZero(cfg.time_embed_dim, D, cfg.norm_eps)
        self.attn1 = Attention(D, cfg.num_attention_heads, cfg.attention_head_dim)
        self.norm2 = LayerNormZero(cfg.time_embed_dim, D, cfg.norm_eps)
        self.ff = FeedForward(D, 4)

    def forward(self, h, e, temb, rope):
        n_text = e.shape[1]
        nh, ne, gate, egate = self.norm1(h, e, temb)
        ah, ae = self.attn1(nh, ne, rope)
        h = h + gate * ah
        e = e + egate * ae
        nh, ne, gate, egate = self.norm2(h, e, temb)
        ff = self.ff(torch.cat([ne, nh], dim=1))
        h = h + gate * ff[:, n_text:]
        e = e + egate * ff[:, :n_text]
        return h, e


class AdaLayerNorm(nn.Module):
    """diffusers AdaLayerNorm(chunk_dim=1): shift first, then scale."""

    def __init__(self, cond_dim: int, dim: int, eps: float):
        super().__init__()
        self.silu = nn.SiLU()
        self.linear = nn.Linear(cond_dim, 2 * dim)
        self.norm = nn.LayerNorm(dim, eps=eps, elementwise_affine=True)

    def forward(self, x, temb):
        shift, scale = self.linear(self.silu(temb)).chunk(2, dim=1)
        return self.norm(x) * (1 + scale[:, None, :]) + shift[:, None, :]


class OracleTransformer3D(nn.Module):
    def __init__(self, cfg: DitConfig):
        super().__init__()
        self.config = cfg
        D = cfg.inner_dim
        self.patch_embed = PatchEmbed(cfg)
        self.time_embedding = TimestepEmbedding(D, cfg.time_embed_dim)
        self.transformer_blocks = nn.ModuleList([Block(cfg) for _ in range(cfg.num_layers)])
        self.norm_final = nn.LayerNorm(D, eps=cfg.norm_eps, elementwise_affine=True)
        self.norm_out = AdaLayerNorm(cfg.time_embed_dim, D, cfg.norm_eps)
        self.proj_out = nn.Linear(D, cfg.patch_size * cfg.patch_size * cfg.out_channels)

    @torch.no_grad()
    def forward(self, hidden_states, encoder_hidden_states, timestep, ofs=None, image_rotary_emb: Optional[Tuple] = None,
                attention_kwargs=None, return_dict: bool = False):
        cfg = self.config
        B, Fr, Cc, H, W = hidden_states.shape
        t_emb = timestep_sinusoid(timestep, cfg.inner_dim, cfg.flip_sin_to_cos, cfg.freq_shift).to(hidden_states.dtype)
        emb = self.time_embedding(t_emb)
        x = self.patch_embed(encoder_hidden_states, hidden_states)
        n_text = encoder_hidden_states.shape[1]
        e, h = x[:, :n_text], x[:, n_text:]
        for blk in self.transformer_blocks:
            h, e = blk(h, e, emb, image_rotary_emb)
        x = torch.cat([e, h], dim=1)
        x = self.norm_final(x)[:, n_text:]
        x = self.norm_out(x, emb)
        x = self.proj_out(x)
        p = cfg.patch_size
        out = x.reshape(B, Fr, H // p, W // p, -1, p, p).permute(0, 1, 4, 2, 5, 3, 6).flatten(5, 6).flatten(3, 4)
        return (out,)


def init_random_(model: nn.Module, seed: int = 0, std: float = 0.02) -> nn.Module:
    """Seeded synthetic weights (real checkpoints are unavailable): N(0, std) matrices scaled like a trained
    model, non-zero AdaLN linears/biases so every gate and modulation path is exercised, norm weights near 1."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p_ in model.named_parameters():
            if p_.dim() >= 2:
                fan_in = p_[0].numel()
                p_.copy_(torch.randn(p_.shape, generator=g) * min(std * 2.0, 1.0 / math.sqrt(fan_in)))
            elif "norm" in name and name.endswith("weight"):
                p_.copy_(1.0 + 0.1 * torch.randn(p_.shape, generator=g))
            else:
                p_.copy_(0.05 * torch.randn(p_.shape, generator=g))
            if ".linear.weight" in name and ("norm1" in name or "norm2" in name or "norm_out" in name):
                p_.mul_(4.0)  # make shift/scale/gate O(0.1..1)
            if "pos_embedding" in name:
                p_.copy_(0.05 * torch.randn(p_.shape, generator=g))
    return model
