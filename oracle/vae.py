"""ORACLE (test infrastructure only — never imported by the product path).

CPU restatement, in plain PyTorch, of diffusers' `AutoencoderKLCogVideoX` as the reference uses it:
`vae.encode(...)` + `retrieve_latents` at /root/reference/aether/pipelines/aetherv1_pipeline_cogvideox.py:233-245,
557-618, `decode_latents` at P:931,936, with `enable_slicing()` / `enable_tiling()` switched on by every entry point
(/root/reference/scripts/demo.py:229-230).  The algorithm lives in the un-vendored third-party package
`diffusers>=0.32.2` (models/autoencoders/autoencoder_kl_cogvideox.py); it is restated here from the published
source as summarised in SURVEY.md Appendix A.2.  PARITY UNPINNED (no reference tests / golden vectors; diffusers is
not importable in the build container).  Parameter names reproduce diffusers' state-dict keys.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass
class VaeConfig:
    in_channels: int = 3
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 256, 512)
    latent_channels: int = 16
    layers_per_block: int = 3
    norm_eps: float = 1e-6
    norm_num_groups: int = 32
    temporal_compression_ratio: int = 4
    sample_height: int = 480
    sample_width: int = 720
    scaling_factor: float = 0.7
    invert_scale_latents: bool = False
    use_quant_conv: bool = False
    use_post_quant_conv: bool = False


class CausalConv3d(nn.Module):
    """CogVideoXCausalConv3d, pad_mode="first": k_t-1 frames on the FRONT only (cache or replicated first frame),
    zero padding in space inside the conv; returns (out, new_cache = last k_t-1 input frames)."""

    def __init__(self, cin, cout, kernel_size):
        super().__init__()
        k = (kernel_size,) * 3 if isinstance(kernel_size, int) else kernel_size
        self.kt = k[0]
        self.conv = nn.Conv3d(cin, cout, k, stride=1, padding=(0, (k[1] - 1) // 2, (k[2] - 1) // 2))

    def forward(self, x, conv_cache=None):
        if self.kt > 1:
            front = [conv_cache] if conv_cache is not None else [x[:, :, :1]] * (self.kt - 1)
            x = torch.cat(front + [x], dim=2)
        new_cache = x[:, :, -self.kt + 1:].clone() if self.kt > 1 else None
        return self.conv(x), new_cache


class SpatialNorm3D(nn.Module):
    def __init__(self, f_channels, zq_channels, groups):
        super().__init__()
        self.norm_layer = nn.GroupNorm(groups, f_channels, eps=1e-6, affine=True)
        self.conv_y = CausalConv3d(zq_channels, f_channels, 1)
        self.conv_b = CausalConv3d(zq_channels, f_channels, 1)

    def forward(self, f, zq):
        if f.shape[2] > 1 and f.shape[2] % 2 == 1:
            z_first = F.interpolate(zq[:, :, :1], size=f[:, :, :1].shape[-3:])
            z_rest = F.interpolate(zq[:, :, 1:], size=f[:, :, 1:].shape[-3:])
            zq = torch.cat([z_first, z_rest], dim=2)
        else:
            zq = F.interpolate(zq, size=f.shape[-3:])
        return self.norm_layer(f) * self.conv_y(zq)[0] + self.conv_b(zq)[0]


class ResnetBlock3D(nn.Module):
    def __init__(self, cin, cout, groups, eps, spatial_norm_dim=None):
        super().__init__()
        self.cin, self.cout = cin, cout
        if spatial_norm_dim is None:
            self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
            self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        else:
            self.norm1 = SpatialNorm3D(cin, spatial_norm_dim, groups)
            self.norm2 = SpatialNorm3D(cout, spatial_norm_dim, groups)
        self.conv1 = CausalConv3d(cin, cout, 3)
        self.conv2 = CausalConv3d(cout, cout, 3)
        if cin != cout:
            self.conv_shortcut = nn.Conv3d(cin, cout, 1)

    def forward(self, x, zq=None, conv_cache=None):
        conv_cache = conv_cache or {}
        new_cache = {}
        h = self.norm1(x, zq) if zq is not None else self.norm1(x)
        h, new_cache["conv1"] = self.conv1(F.silu(h), conv_cache.get("conv1"))
        h = self.norm2(h, zq) if zq is not None else self.norm2(h)
        h, new_cache["conv2"] = self.conv2(F.silu(h), conv_cache.get("conv2"))
        if self.cin != self.cout:
            x = self.conv_shortcut(x)
        return h + x, new_cache


class Downsample3D(nn.Module):
    def __init__(self, c, compress_time):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=0)
        self.compress_time = compress_time

    def forward(self, x):
        if self.compress_time:
            B, C, T, H, W = x.shape
            x = x.permute(0, 3, 4, 1, 2).reshape(B * H * W, C, T)
            if T % 2 == 1:
                first, rest = x[..., 0], x[..., 1:]
                if rest.shape[-1] > 0:
                    rest = F.avg_pool1d(rest, kernel_size=2, stride=2)
                x = torch.cat([first[..., None], rest], dim=-1)
            else:
                x = F.avg_pool1d(x, kernel_size=2, stride=2)
            x = x.reshape(B, H, W, C, x.shape[-1]).permute(0, 3, 4, 1, 2)
        x = F.pad(x, (0, 1, 0, 1), mode="constant", value=0)
        B, C, T, H, W = x.shape
        x = self.conv(x.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W))
        return x.reshape(B, T, *x.shape[1:]).permute(0, 2, 1, 3, 4)


class Upsample3D(nn.Module):
    def __init__(self, c, compress_time):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=1, padding=1)
        self.compress_time = compress_time

    def forward(self, x):
        if self.compress_time:
            if x.shape[2] > 1 and x.shape[2] % 2 == 1:
                first = F.interpolate(x[:, :, 0], scale_factor=2.0)[:, :, None]
                rest = F.interpolate(x[:, :, 1:], scale_factor=2.0)
                x = torch.cat([first, rest], dim=2)
            elif x.shape[2] > 1:
                x = F.interpolate(x, scale_factor=2.0)
            else:
                x = F.interpolate(x.squeeze(2), scale_factor=2.0)[:, :, None]
        else:
            B, C, T, H, W = x.shape
            x = F.interpolate(x.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W), scale_factor=2.0)
            x = x.reshape(B, T, C, *x.shape[2:]).permute(0, 2, 1, 3, 4)
        B, C, T, H, W = x.shape
        x = self.conv(x.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W))
        return x.reshape(B, T, *x.shape[1:]).permute(0, 2, 1, 3, 4)


class _ResStack(nn.Module):
    def __init__(self, cin, cout, n, groups, eps, spatial_norm_dim=None):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock3D(cin if i == 0 else cout, cout, groups, eps, spatial_norm_dim) for i in range(n)])

    def run(self, x, zq, conv_cache):
        conv_cache = conv_cache or {}
        new_cache = {}
        for i, r in enumerate(self.resnets):
            x, new_cache[f"resnet_{i}"] = r(x, zq, conv_cache.get(f"resnet_{i}"))
        return x, new_cache


class DownBlock3D(_ResStack):
    def __init__(self, cin, cout, n, groups, eps, add_downsample, compress_time):
        super().__init__(cin, cout, n, groups, eps)
        self.downsamplers = nn.ModuleList([Downsample3D(cout, compress_time)]) if add_downsample else None

    def forward(self, x, zq=None, conv_cache=None):
        x, c = self.run(x, None, conv_cache)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
        return x, c


class UpBlock3D(_ResStack):
    def __init__(self, cin, cout, n, groups, eps, zdim, add_upsample, compress_time):
        super().__init__(cin, cout, n, groups, eps, zdim)
        self.upsamplers = nn.ModuleList([Upsample3D(cout, compress_time)]) if add_upsample else None

    def forward(self, x, zq, conv_cache=None):
        x, c = self.run(x, zq, conv_cache)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x, c


class MidBlock3D(_ResStack):
    def forward(self, x, zq=None, conv_cache=None):
        return self.run(x, zq, conv_cache)


class Encoder3D(nn.Module):
    def __init__(self, cfg: VaeConfig):
        super().__init__()
        ch, g, eps = cfg.block_out_channels, cfg.norm_num_groups, cfg.norm_eps
        tlevel = int(np.log2(cfg.temporal_compression_ratio))
        self.conv_in = CausalConv3d(cfg.in_channels, ch[0], 3)
        blocks, cout = [], ch[0]
        for i in range(len(ch)):
            cin, cout = cout, ch[i]
            blocks.append(DownBlock3D(cin, cout, cfg.layers_per_block, g, eps, add_downsample=i != len(ch) - 1, compress_time=i < tlevel))
        self.down_blocks = nn.ModuleList(blocks)
        self.mid_block = MidBlock3D(ch[-1], ch[-1], 2, g, eps)
        self.norm_out = nn.GroupNorm(g, ch[-1], eps=1e-6)
        self.conv_out = CausalConv3d(ch[-1], 2 * cfg.latent_channels, 3)

    def forward(self, x, conv_cache=None):
        conv_cache = conv_cache or {}
        new_cache = {}
        x, new_cache["conv_in"] = self.conv_in(x, conv_cache.get("conv_in"))
        for i, b in enumerate(self.down_blocks):
            x, new_cache[f"down_block_{i}"] = b(x, None, conv_cache.get(f"down_block_{i}"))
        x, new_cache["mid_block"] = self.mid_block(x, None, conv_cache.get("mid_block"))
        x, new_cache["conv_out"] = self.conv_out(F.silu(self.norm_out(x)), conv_cache.get("conv_out"))
        return x, new_cache


class Decoder3D(nn.Module):
    def __init__(self, cfg: VaeConfig):
        super().__init__()
        ch, g, eps, z = list(reversed(cfg.block_out_channels)), cfg.norm_num_groups, cfg.norm_eps, cfg.latent_channels
        tlevel = int(np.log2(cfg.temporal_compression_ratio))
        self.conv_in = CausalConv3d(z, ch[0], 3)
        self.mid_block = MidBlock3D(ch[0], ch[0], 2, g, eps, z)
        blocks, cout = [], ch[0]
        for i in range(len(ch)):
            cin, cout = cout, ch[i]
            blocks.append(UpBlock3D(cin, cout, cfg.layers_per_block + 1, g, eps, z, add_upsample=i != len(ch) - 1, compress_time=i < tlevel))
        self.up_blocks = nn.ModuleList(blocks)
        self.norm_out = SpatialNorm3D(ch[-1], z, g)
        self.conv_out = CausalConv3d(ch[-1], cfg.out_channels, 3)

    def forward(self, z, conv_cache=None):
        conv_cache = conv_cache or {}
        new_cache = {}
        x, new_cache["conv_in"] = self.conv_in(z, conv_cache.get("conv_in"))
        x, new_cache["mid_block"] = self.mid_block(x, z, conv_cache.get("mid_block"))
        for i, b in enumerate(self.up_blocks):
            x, new_cache[f"up_block_{i}"] = b(x, z, conv_cache.get(f"up_block_{i}"))
        x, new_cache["conv_out"] = self.conv_out(F.silu(self.norm_out(x, z)), conv_cache.get("conv_out"))
        return x, new_cache


class DiagonalGaussian:
    def __init__(self, params):
        self.parameters = params
        self.mean, logvar = torch.chunk(params, 2, dim=1)
        self.logvar = torch.clamp(logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator: Optional[torch.Generator] = None):
        noise = torch.randn(self.mean.shape, generator=generator, device=self.parameters.device, dtype=self.parameters.dtype)
        return self.mean + self.std * noise

    def mode(self):
        return self.mean


class _Out:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class OracleVAE(nn.Module):
    def __init__(self, cfg: VaeConfig = VaeConfig()):
        super().__init__()
        self.config = cfg
        self.encoder = Encoder3D(cfg)
        self.decoder = Decoder3D(cfg)
        self.use_tiling = False
        self.use_slicing = False
        self.num_latent_frames_batch_size = 2
        self.num_sample_frames_batch_size = 8
        down = 2 ** (len(cfg.block_out_channels) - 1)
        self.tile_sample_min_height = cfg.sample_height // 2
        self.tile_sample_min_width = cfg.sample_width // 2
        self.tile_latent_min_height = int(self.tile_sample_min_height / down)
        self.tile_latent_min_width = int(self.tile_sample_min_width / down)
        self.tile_overlap_factor_height = 1 / 6
        self.tile_overlap_factor_width = 1 / 5

    def enable_tiling(self):
        self.use_tiling = True

    def enable_slicing(self):
        self.use_slicing = True

    # ---- frame batching (conv caches threaded chunk to chunk) ----------------------------------------
    @staticmethod
    def _chunks(n, bs):
        nb, rem = max(n // bs, 1), n % bs
        return [(bs * k + (0 if k == 0 else rem), bs * (k + 1) + rem) for k in range(nb)]

    def _run_chunks(self, net, x, bs):
        cache, outs = None, []
        for s, e in self._chunks(x.shape[2], bs):
            y, cache = net(x[:, :, s:e], conv_cache=cache)
            outs.append(y)
        return torch.cat(outs, dim=2)

    # ---- tiling ---------------------------------------------------------------------------------------
    @staticmethod
    def _blend_v(a, b, extent):
        extent = min(a.shape[3], b.shape[3], extent)
        for y in range(extent):
            b[:, :, :, y, :] = a[:, :, :, -extent + y, :] * (1 - y / extent) + b[:, :, :, y, :] * (y / extent)
        return b

    @staticmethod
    def _blend_h(a, b, extent):
        extent = min(a.shape[4], b.shape[4], extent)
        for x in range(extent):
            b[:, :, :, :, x] = a[:, :, :, :, -extent + x] * (1 - x / extent) + b[:, :, :, :, x] * (x / extent)
        return b

    def _tiled(self, net, x, bs, tile_h, tile_w, stride_h, stride_w, blend_h, blend_w, limit_h, limit_w):
        H, W = x.shape[-2:]
        rows = [[self._run_chunks(net, x[:, :, :, i:i + tile_h, j:j + tile_w], bs) for j in range(0, W, stride_w)]
                for i in range(0, H, stride_h)]
        out_rows = []
        for i, row in enumerate(rows):
            out = []
            for j, tile in enumerate(row):
                if i > 0:
                    tile = self._blend_v(rows[i - 1][j], tile, blend_h)
                if j > 0:
                    tile = self._blend_h(row[j - 1], tile, blend_w)
                out.append(tile[:, :, :, :limit_h, :limit_w])
            out_rows.append(torch.cat(out, dim=4))
        return torch.cat(out_rows, dim=3)

    @torch.no_grad()
    def encode(self, x):
        x = x.contiguous()   # CPU bf16 convolutions pick stride-dependent kernels: canonicalise the layout at entry
        H, W = x.shape[-2:]
        if self.use_tiling and (W > self.tile_sample_min_width or H > self.tile_sample_min_height):
            sh = int(self.tile_sample_min_height * (1 - self.tile_overlap_factor_height))
            sw = int(self.tile_sample_min_width * (1 - self.tile_overlap_factor_width))
            bh = int(self.tile_latent_min_height * self.tile_overlap_factor_height)
            bw = int(self.tile_latent_min_width * self.tile_overlap_factor_width)
            h = self._tiled(self.encoder, x, self.num_sample_frames_batch_size, self.tile_sample_min_height,
                            self.tile_sample_min_width, sh, sw, bh, bw, self.tile_latent_min_height - bh, self.tile_latent_min_width - bw)
        else:
            h = self._run_chunks(self.encoder, x, self.num_sample_frames_batch_size)
        return _Out(latent_dist=DiagonalGaussian(h))

    @torch.no_grad()
    def decode(self, z):
        z = z.contiguous()
        H, W = z.shape[-2:]
        if self.use_tiling and (W > self.tile_latent_min_width or H > self.tile_latent_min_height):
            sh = int(self.tile_latent_min_height * (1 - self.tile_overlap_factor_height))
            sw = int(self.tile_latent_min_width * (1 - self.tile_overlap_factor_width))
            bh = int(self.tile_sample_min_height * self.tile_overlap_factor_height)
            bw = int(self.tile_sample_min_width * self.tile_overlap_factor_width)
            d = self._tiled(self.decoder, z, self.num_latent_frames_batch_size, self.tile_latent_min_height,
                            self.tile_latent_min_width, sh, sw, bh, bw, self.tile_sample_min_height - bh, self.tile_sample_min_width - bw)
        else:
            d = self._run_chunks(self.decoder, z, self.num_latent_frames_batch_size)
        return _Out(sample=d)


def init_random_(model: nn.Module, seed: int = 0) -> nn.Module:
    """Seeded synthetic VAE weights (real ones are unavailable): fan-in scaled convs so activations stay O(1)
    through ~40 layers, GroupNorm weights 1 +- 0.1."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() >= 3:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g) * (1.0 / fan_in) ** 0.5)
            elif name.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.02 * torch.randn(p.shape, generator=g))
    return model
