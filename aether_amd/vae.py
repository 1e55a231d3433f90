"""MI355X-native stand-in for diffusers' `AutoencoderKLCogVideoX` at the reference's call sites:
`vae.encode(x).latent_dist.sample(generator)` (/root/reference/aether/pipelines/aetherv1_pipeline_cogvideox.py:233-245,
557-618), `vae.decode(z).sample` through `decode_latents` (P:931,936), `.config.{latent_channels, invert_scale_latents,
scaling_factor, block_out_channels, temporal_compression_ratio}` (P:571,843,925-929), `enable_slicing()` /
`enable_tiling()` (/root/reference/scripts/demo.py:229-230) and `from_pretrained(..., subfolder="vae")` (D:215-219).

Host side only (Python, like the reference): it walks the encoder / decoder graph, frame chunks and spatial tiles
and enqueues the HIP kernels of csrc/vae_kernels.hip + the MFMA GEMM.  MI355X-first choices:
  * channels-last activations [NB, T, H, W, C]: every 3x3x3 causal convolution is an implicit GEMM on the MFMA
    kernel (rows = voxels, K = 27 taps x C) with NO im2col buffer; spatial zero padding and the causal front frames
    (conv cache) are materialised by the producer in zero-bordered volumes, so the GEMM loop has no bounds checks;
  * the reference's spatial tiles of equal shape are BATCHED (NB = 4/2/2/1 instead of 9 sequential passes): 288 GB of
    HBM makes the 9x activation footprint irrelevant and the low-resolution layers get 4x more rows per launch;
  * GroupNorm statistics are a deterministic two-level reduction; normalise + affine + SpatialNorm3D + SiLU are one
    pass that writes straight into the next convolution's padded input volume.
There is no PyTorch fallback for the arithmetic: without libaether_hip.so / a gfx950 device construction fails.
Tile blending / cropping and the posterior sample are small element-wise device ops kept in PyTorch (plumbing).
"""
from __future__ import annotations

import ctypes as C
import json
import math
import os
from types import SimpleNamespace
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib
from .scheduler import randn_tensor

_CONFIG_DEFAULTS = dict(
    in_channels=3, out_channels=3, block_out_channels=(128, 256, 256, 512), latent_channels=16, layers_per_block=3,
    act_fn="silu", norm_eps=1e-6, norm_num_groups=32, temporal_compression_ratio=4, sample_height=480, sample_width=720,
    scaling_factor=0.7, shift_factor=None, latents_mean=None, latents_std=None, force_upcast=True, use_quant_conv=False,
    use_post_quant_conv=False, invert_scale_latents=False,
)


class DiagonalGaussianDistribution:
    """diffusers.models.autoencoders.vae.DiagonalGaussianDistribution (mean | logvar along dim 1, logvar clamped to
    [-30, 20]); `sample` draws randn(mean.shape) from the caller's generator in the parameters' dtype (RNG parity)."""

    def __init__(self, parameters: torch.Tensor):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        noise = randn_tensor(self.mean.shape, generator=generator, device=self.parameters.device, dtype=self.parameters.dtype)
        return self.mean + self.std * noise

    def mode(self) -> torch.Tensor:
        return self.mean


class _Conv:
    """Packed convolution: weight bf16 [Cout_pad, K], bias fp32 [Cout_pad].
    K order of the implicit-GEMM convolutions (cin a multiple of 64): (dt, dh, channel block of 64, dw, 64 channels) — dw is the
    innermost tap index, so three consecutive K steps of a workgroup read the SAME rows of the input volume shifted by one
    voxel and the second and third hit L2 (the plain (dt, dh, dw, cin) order separates them by cin/64 K steps, i.e. by
    cin/64 x 2 MB of other tiles' traffic per XCD).  The thin first convolutions (explicit im2col, `pad_k_to`) keep (dt, dh, dw, cin)."""

    def __init__(self, weight: torch.Tensor, bias: torch.Tensor, device, pad_k_to: Optional[int] = None):
        w = weight.detach().float()
        self.cout, self.cin = w.shape[0], w.shape[1]
        self.ksize = tuple(w.shape[2:])
        self.blocked = pad_k_to is None and self.cin % 64 == 0 and len(self.ksize) >= 2
        if self.blocked:
            cb = self.cin // 64
            w5 = w if w.dim() == 5 else w.unsqueeze(2)                       # [cout, cin, kt, kh, kw] (kt = 1 for conv2d)
            kt, kh, kw = w5.shape[2:]
            w2 = w5.reshape(self.cout, cb, 64, kt, kh, kw).permute(0, 3, 4, 1, 5, 2).reshape(self.cout, -1)   # (dt, dh, cb, dw, 64)
        else:
            perm = (0,) + tuple(range(2, w.dim())) + (1,)
            w2 = w.permute(*perm).reshape(self.cout, -1)
        cout_pad = (self.cout + 31) // 32 * 32
        k_pad = pad_k_to or w2.shape[1]
        wp = torch.zeros(cout_pad, k_pad)
        wp[: self.cout, : w2.shape[1]] = w2
        bp = torch.zeros(cout_pad)
        bp[: self.cout] = bias.detach().float()
        self.cout_pad = cout_pad
        self.w = wp.to(device=device, dtype=torch.bfloat16).contiguous()
        self.b = bp.to(device=device, dtype=torch.float32).contiguous()


class _Norm:
    def __init__(self, sd, prefix, device, spatial: bool):
        f32 = lambda t: t.detach().to(device=device, dtype=torch.float32).contiguous()  # noqa: E731
        self.spatial = spatial
        if spatial:
            self.gamma, self.beta = f32(sd[prefix + "norm_layer.weight"]), f32(sd[prefix + "norm_layer.bias"])
            self.wy = f32(sd[prefix + "conv_y.conv.weight"].flatten(1))
            self.by = f32(sd[prefix + "conv_y.conv.bias"])
            self.wb = f32(sd[prefix + "conv_b.conv.weight"].flatten(1))
            self.bb = f32(sd[prefix + "conv_b.conv.bias"])
        else:
            self.gamma, self.beta = f32(sd[prefix + "weight"]), f32(sd[prefix + "bias"])


class _Resnet:
    def __init__(self, sd, prefix, device, spatial: bool):
        self.norm1 = _Norm(sd, prefix + "norm1.", device, spatial)
        self.norm2 = _Norm(sd, prefix + "norm2.", device, spatial)
        self.conv1 = _Conv(sd[prefix + "conv1.conv.weight"], sd[prefix + "conv1.conv.bias"], device)
        self.conv2 = _Conv(sd[prefix + "conv2.conv.weight"], sd[prefix + "conv2.conv.bias"], device)
        self.shortcut = None
        if prefix + "conv_shortcut.weight" in sd:
            self.shortcut = _Conv(sd[prefix + "conv_shortcut.weight"], sd[prefix + "conv_shortcut.bias"], device)
        self.name = prefix


class AetherVAE:
    def __init__(self, config: Optional[dict] = None, device="cuda", flags: int = _lib.AETHER_GEMM_WIDE_STORE | _lib.AETHER_VAE_TWO_LANES):
        cfg = dict(_CONFIG_DEFAULTS)
        cfg.update(config or {})
        cfg["block_out_channels"] = tuple(cfg["block_out_channels"])
        self.config = SimpleNamespace(**cfg)
        c = self.config
        if c.use_quant_conv or c.use_post_quant_conv:
            raise ValueError("aether_amd: quant_conv / post_quant_conv are not implemented (CogVideoX VAEs do not use them)")
        if any(ch % 64 for ch in c.block_out_channels) or c.latent_channels > 16:
            raise ValueError("aether_amd: block_out_channels must be multiples of 64 and latent_channels <= 16")
        self.device = torch.device(device)
        self.dtype = torch.bfloat16
        self._lib = _lib.load()
        self._flags = flags
        self.use_tiling = False
        self.use_slicing = False
        self.num_latent_frames_batch_size = 2
        self.num_sample_frames_batch_size = 8
        down = 2 ** (len(c.block_out_channels) - 1)
        self.tile_sample_min_height = c.sample_height // 2
        self.tile_sample_min_width = c.sample_width // 2
        self.tile_latent_min_height = int(self.tile_sample_min_height / down)
        self.tile_latent_min_width = int(self.tile_sample_min_width / down)
        self.tile_overlap_factor_height = 1 / 6
        self.tile_overlap_factor_width = 1 / 5
        self._pool: Dict[tuple, torch.Tensor] = {}
        self._taps: Dict[tuple, torch.Tensor] = {}
        self._splitk_ws: Optional[torch.Tensor] = None       # fp32 scratch for split-K partial tiles (allocated on first use)
        self.splitk_ws_bytes = 96 << 20
        self._tap_reuse_max_waste = 1.06                      # padded-plane / output-plane ratio up to which the tap-reuse conv runs (0: never)
        self._loaded = False
        # encode()/decode() are ONE C call each (aether_vae_encode / aether_vae_decode: the launch plan lives in csrc/vae_plan.hip).
        # use_c_plan = False walks the same graph from Python through the per-kernel entry points (tests, A/B: bit-identical).
        self.use_c_plan = True
        self._handle = None
        self._workspace: Optional[torch.Tensor] = None
        self._ws_bytes: Optional[int] = None
        self._graphs: Dict = {}
        self.use_graphs = os.environ.get("AETHER_VAE_GRAPHS", "1") != "0"    # replay encode / decode from a captured hipGraph

    # ------------------------------------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, path, subfolder: Optional[str] = "vae", torch_dtype=torch.bfloat16, device="cuda", **_):
        from safetensors.torch import load_file
        root = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(root, "config.json")) as f:
            cfg = {k: v for k, v in json.load(f).items() if not k.startswith("_") and k in _CONFIG_DEFAULTS}
        m = cls(cfg, device=device)
        m.load_state_dict(load_file(os.path.join(root, "diffusion_pytorch_model.safetensors")))
        return m

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    @property
    def tap_reuse_max_waste(self) -> float:
        return self._tap_reuse_max_waste

    @tap_reuse_max_waste.setter
    def tap_reuse_max_waste(self, value: float):
        """The C launch plan copies this tunable into its handle at registration: changing it after the weights are loaded re-registers the plan
        (new handle, workspace pool and hipGraphs start over), so the C plan and the Python walk can never run with different settings."""
        self._tap_reuse_max_waste = float(value)
        if getattr(self, "_loaded", False):
            self._register_c_plan()

    def enable_tiling(self):
        self.use_tiling = True

    def disable_tiling(self):
        self.use_tiling = False

    def enable_slicing(self):
        self.use_slicing = True

    def disable_slicing(self):
        self.use_slicing = False

    # ------------------------------------------------------------------------------------------------
    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        c, dev = self.config, self.device
        ch = c.block_out_channels
        nlev = len(ch)
        tlevel = int(math.log2(c.temporal_compression_ratio))
        e = SimpleNamespace()
        e.conv_in = _Conv(sd["encoder.conv_in.conv.weight"], sd["encoder.conv_in.conv.bias"], dev, pad_k_to=(27 * c.in_channels + 63) // 64 * 64)
        e.down = []
        for i in range(nlev):
            res = [_Resnet(sd, f"encoder.down_blocks.{i}.resnets.{j}.", dev, False) for j in range(c.layers_per_block)]
            ds = None
            if i != nlev - 1:
                ds = _Conv(sd[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"], sd[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"], dev)
            e.down.append(SimpleNamespace(resnets=res, down=ds, compress_time=i < tlevel))
        e.mid = [_Resnet(sd, f"encoder.mid_block.resnets.{j}.", dev, False) for j in range(2)]
        e.norm_out = _Norm(sd, "encoder.norm_out.", dev, False)
        e.conv_out = _Conv(sd["encoder.conv_out.conv.weight"], sd["encoder.conv_out.conv.bias"], dev)
        d = SimpleNamespace()
        d.conv_in = _Conv(sd["decoder.conv_in.conv.weight"], sd["decoder.conv_in.conv.bias"], dev, pad_k_to=(27 * c.latent_channels + 63) // 64 * 64)
        d.mid = [_Resnet(sd, f"decoder.mid_block.resnets.{j}.", dev, True) for j in range(2)]
        d.up = []
        for i in range(nlev):
            res = [_Resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}.", dev, True) for j in range(c.layers_per_block + 1)]
            us = None
            if i != nlev - 1:
                us = _Conv(sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"], sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"], dev)
            d.up.append(SimpleNamespace(resnets=res, up=us, compress_time=i < tlevel))
        d.norm_out = _Norm(sd, "decoder.norm_out.", dev, True)
        d.conv_out = _Conv(sd["decoder.conv_out.conv.weight"], sd["decoder.conv_out.conv.bias"], dev)
        self.enc, self.dec = e, d
        self._loaded = True
        self._register_c_plan()
        return self

    def _register_c_plan(self):
        """Hand every packed convolution / norm to the C launch plan under its diffusers module path."""
        c = self.config
        if self._handle is not None:
            self._lib.aether_vae_destroy(self._handle)
        cfg = _lib.AetherVaeConfig(in_channels=c.in_channels, out_channels=c.out_channels, latent_channels=c.latent_channels,
                                   layers_per_block=c.layers_per_block, num_levels=len(c.block_out_channels), norm_num_groups=c.norm_num_groups,
                                   temporal_compression_ratio=c.temporal_compression_ratio, sample_height=c.sample_height,
                                   sample_width=c.sample_width, norm_eps=c.norm_eps, tap_reuse_max_waste=self.tap_reuse_max_waste, flags=self._flags)
        h = self._lib.aether_vae_create(C.byref(cfg))
        if not h:
            raise ValueError("aether_vae_create: " + self._lib.aether_last_error().decode())
        self._handle = h
        self._workspace = None
        self._graphs.clear()
        self._twin = None          # a twin context (decode_pair) holds the previous handle's weight pointers: rebuilt on the next pair

        def conv(name, cv: _Conv):
            kt, kh, kw = cv.ksize if len(cv.ksize) == 3 else ((1,) + tuple(cv.ksize) if len(cv.ksize) == 2 else (1, 1, 1))
            _lib.check(self._lib.aether_vae_set_conv(h, name.encode(), cv.w.data_ptr(), cv.b.data_ptr(), cv.cout, cv.cout_pad, cv.cin, kt, kh, kw,
                                                     cv.w.shape[1], int(cv.blocked)), f"aether_vae_set_conv({name})")

        def norm(name, n: _Norm):
            sp = [n.wy.data_ptr(), n.by.data_ptr(), n.wb.data_ptr(), n.bb.data_ptr()] if n.spatial else [None] * 4
            _lib.check(self._lib.aether_vae_set_norm(h, name.encode(), n.gamma.data_ptr(), n.beta.data_ptr(), *sp), f"aether_vae_set_norm({name})")

        def resnet(r: _Resnet):
            norm(r.name + "norm1", r.norm1); norm(r.name + "norm2", r.norm2)
            conv(r.name + "conv1", r.conv1); conv(r.name + "conv2", r.conv2)
            if r.shortcut is not None:
                conv(r.name + "conv_shortcut", r.shortcut)

        e, d = self.enc, self.dec
        conv("encoder.conv_in", e.conv_in); conv("encoder.conv_out", e.conv_out); norm("encoder.norm_out", e.norm_out)
        for i, blk in enumerate(e.down):
            for r in blk.resnets:
                resnet(r)
            if blk.down is not None:
                conv(f"encoder.down_blocks.{i}.downsamplers.0", blk.down)
        for r in e.mid:
            resnet(r)
        conv("decoder.conv_in", d.conv_in); conv("decoder.conv_out", d.conv_out); norm("decoder.norm_out", d.norm_out)
        for r in d.mid:
            resnet(r)
        for i, blk in enumerate(d.up):
            for r in blk.resnets:
                resnet(r)
            if blk.up is not None:
                conv(f"decoder.up_blocks.{i}.upsamplers.0", blk.up)

    def __del__(self):
        try:
            if self._handle is not None:
                self._lib.aether_vae_destroy(self._handle)
        except Exception:
            pass

    def reserve_workspace(self, num_frames: int, height: int, width: int) -> int:
        """Size the launch-plan workspace ONCE for everything a pipeline call of this geometry runs (P:557-618 then P:931,936 inside one `__call__`):
        the encode of the `num_frames` clip (reconstruction), the encode of ONE frame (the image / goal of prediction and planning) and the decode of
        the resulting latent clip — the maximum of the three, so that no later call of any task re-allocates and drops captured hipGraphs.
        A no-op when the workspace already covers it.  Returns the workspace size in bytes."""
        if not (self.use_c_plan and self._loaded):
            return 0
        down = 2 ** (len(self.config.block_out_channels) - 1)
        ct = self.config.temporal_compression_ratio
        zT, zH, zW = (num_frames - 1) // ct + 1, height // down, width // down
        need = 0
        for decode, (T, H, W) in ((0, (num_frames, height, width)), (0, (1, height, width)), (1, (zT, zH, zW))):
            n = self._lib.aether_vae_workspace_bytes(self._handle, decode, T, H, W, int(self.use_tiling))
            if n == 0:
                raise RuntimeError("aether_vae_workspace_bytes: " + self._lib.aether_last_error().decode())
            need = max(need, int(n))
        # each query counts the tap-offset tables already generated in this workspace plus those of ITS direction: the slack (added only when the
        # workspace is (re)allocated, never to the comparison) covers the tables of the other direction and of further geometries
        if self._workspace is not None and self._workspace.numel() >= need:
            return self._workspace.numel()
        need += 8 << 20
        if self._graphs:
            import warnings
            warnings.warn(f"AetherVAE: workspace grows to {need / 2**30:.1f} GiB; {len(self._graphs)} captured hipGraph(s) dropped", stacklevel=2)
        self._workspace = None
        self._graphs.clear()
        torch.cuda.empty_cache()
        self._ws_bytes = max(need, 0 if self._ws_bytes is None else self._ws_bytes)
        self._workspace = torch.empty(self._ws_bytes, dtype=torch.uint8, device=self.device)
        return self._ws_bytes

    def _run_c_plan(self, x: torch.Tensor, decode: bool) -> torch.Tensor:
        """x [1, C, T, H, W] bf16 on the device -> [1, C_out, T_out, H_out, W_out] through ONE C call (replayed from a hipGraph
        when `use_graphs`: the call only enqueues — ~4 700 kernels per decode — so its capture is valid for a fixed geometry,
        workspace and static input / output buffers)."""
        src = x[0].contiguous()
        _, T, H, W = src.shape
        shp = [C.c_int() for _ in range(4)]
        _lib.check(self._lib.aether_vae_output_shape(self._handle, int(decode), T, H, W, *[C.byref(v) for v in shp]), "aether_vae_output_shape")
        oshape = (1, *[v.value for v in shp])
        need = self._lib.aether_vae_workspace_bytes(self._handle, int(decode), T, H, W, int(self.use_tiling))
        if need == 0:
            raise RuntimeError("aether_vae_workspace_bytes: " + self._lib.aether_last_error().decode())
        if self._workspace is None or self._workspace.numel() < need:
            # one workspace for both directions (nothing in it persists from call to call except the tiny tap-offset tables, which the library
            # regenerates when the pointer changes), sized at once for this call AND for the mirror call of the other direction — the decode of
            # the latent this encode produces, or the encode of the clip this decode produces — so that a pipeline's first encode is not followed
            # by a re-allocation (and the loss of its captured graph) at the first decode
            down = 2 ** (len(self.config.block_out_channels) - 1)
            ct = self.config.temporal_compression_ratio
            mirror = ((T - 1) * ct + 1, H * down, W * down) if decode else ((T - 1) // ct + 1, H // down, W // down)
            other = self._lib.aether_vae_workspace_bytes(self._handle, int(not decode), *mirror, int(self.use_tiling)) if min(mirror) > 0 else 0
            if self._graphs:
                # the captured graphs hold the old workspace's addresses: they are dropped and re-captured (one eager call + one capture per geometry)
                import warnings
                warnings.warn(f"AetherVAE: workspace grows to {max(int(need), int(other)) / 2**30:.1f} GiB; {len(self._graphs)} captured hipGraph(s) dropped", stacklevel=3)
            self._workspace = None
            self._graphs.clear()
            torch.cuda.empty_cache()
            # each query counts the tap-offset tables (a few hundred bytes per distinct convolution shape) of ITS direction only: the slack covers the
            # other direction's tables, and those of further geometries of the same size class
            self._ws_bytes = max(int(need), int(other), 0 if self._ws_bytes is None else self._ws_bytes) + (8 << 20)
            self._workspace = torch.empty(self._ws_bytes, dtype=torch.uint8, device=self.device)
        fn = self._lib.aether_vae_decode if decode else self._lib.aether_vae_encode
        what = "aether_vae_decode" if decode else "aether_vae_encode"

        def call(src_t, out_t):
            _lib.check(fn(self._handle, src_t.data_ptr(), T, H, W, int(self.use_tiling), out_t.data_ptr(), self._workspace.data_ptr(),
                          self._workspace.numel(), self._stream()), what)

        if not self.use_graphs or torch.cuda.is_current_stream_capturing():
            out = torch.empty(oshape, dtype=torch.bfloat16, device=self.device)
            call(src, out)
            return out
        key = (decode, T, H, W, bool(self.use_tiling))
        ent = self._graphs.get(key)
        if ent is None:
            # first call of this geometry: eager (it also zeroes / fills what the pool needs); the second one is captured — but only
            # after the eager call SUCCEEDED (a failed first call must not let the next one capture the pool's one-time set-up)
            out = torch.empty(oshape, dtype=torch.bfloat16, device=self.device)
            call(src, out)
            self._graphs[key] = "seen"
            return out
        if ent == "seen":
            s_in, s_out = torch.empty_like(src), torch.empty(oshape, dtype=torch.bfloat16, device=self.device)
            g = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                with torch.cuda.graph(g, stream=side):
                    call(s_in, s_out)
            torch.cuda.current_stream(self.device).wait_stream(side)
            ent = self._graphs[key] = (g, s_in, s_out)
        g, s_in, s_out = ent
        s_in.copy_(src)
        g.replay()
        return s_out.clone()

    def state_dict_spec(self) -> Dict[str, tuple]:
        """diffusers state-dict keys and shapes of an AutoencoderKLCogVideoX with this config."""
        c = self.config
        ch, z, nl = c.block_out_channels, c.latent_channels, c.layers_per_block
        spec: Dict[str, tuple] = {}

        def conv3(name, cin, cout, k=3):
            spec[name + ".conv.weight"] = (cout, cin, k, k, k)
            spec[name + ".conv.bias"] = (cout,)

        def resnet(prefix, cin, cout, spatial):
            for n, cc in (("norm1", cin), ("norm2", cout)):
                if spatial:
                    spec[f"{prefix}.{n}.norm_layer.weight"] = (cc,)
                    spec[f"{prefix}.{n}.norm_layer.bias"] = (cc,)
                    conv3(f"{prefix}.{n}.conv_y", z, cc, 1)
                    conv3(f"{prefix}.{n}.conv_b", z, cc, 1)
                else:
                    spec[f"{prefix}.{n}.weight"] = (cc,)
                    spec[f"{prefix}.{n}.bias"] = (cc,)
            conv3(prefix + ".conv1", cin, cout)
            conv3(prefix + ".conv2", cout, cout)
            if cin != cout:
                spec[prefix + ".conv_shortcut.weight"] = (cout, cin, 1, 1, 1)
                spec[prefix + ".conv_shortcut.bias"] = (cout,)

        conv3("encoder.conv_in", c.in_channels, ch[0])
        cout = ch[0]
        for i in range(len(ch)):
            cin, cout = cout, ch[i]
            for j in range(nl):
                resnet(f"encoder.down_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout, False)
            if i != len(ch) - 1:
                spec[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"] = (cout, cout, 3, 3)
                spec[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"] = (cout,)
        for j in range(2):
            resnet(f"encoder.mid_block.resnets.{j}", ch[-1], ch[-1], False)
        spec["encoder.norm_out.weight"] = (ch[-1],)
        spec["encoder.norm_out.bias"] = (ch[-1],)
        conv3("encoder.conv_out", ch[-1], 2 * z)
        rch = list(reversed(ch))
        conv3("decoder.conv_in", z, rch[0])
        for j in range(2):
            resnet(f"decoder.mid_block.resnets.{j}", rch[0], rch[0], True)
        cout = rch[0]
        for i in range(len(rch)):
            cin, cout = cout, rch[i]
            for j in range(nl + 1):
                resnet(f"decoder.up_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout, True)
            if i != len(rch) - 1:
                spec[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (cout, cout, 3, 3)
                spec[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (cout,)
        spec["decoder.norm_out.norm_layer.weight"] = (rch[-1],)
        spec["decoder.norm_out.norm_layer.bias"] = (rch[-1],)
        conv3("decoder.norm_out.conv_y", z, rch[-1], 1)
        conv3("decoder.norm_out.conv_b", z, rch[-1], 1)
        conv3("decoder.conv_out", rch[-1], c.out_channels)
        return spec

    def init_random_weights(self, seed: int = 0):
        """Seeded synthetic weights (benchmarks; the real CogVideoX VAE checkpoint is not available offline):
        fan-in-scaled convolutions so activations stay O(1) through the ~40 layers, norm weights 1 +- 0.1."""
        g = torch.Generator().manual_seed(seed)
        sd = {}
        for k, shape in self.state_dict_spec().items():
            if len(shape) >= 3:
                fan_in = 1
                for s in shape[1:]:
                    fan_in *= s
                sd[k] = torch.randn(shape, generator=g) * fan_in ** -0.5
            elif k.endswith("weight"):
                sd[k] = 1.0 + 0.1 * torch.randn(shape, generator=g)
            else:
                sd[k] = 0.02 * torch.randn(shape, generator=g)
        return self.load_state_dict(sd)

    # ------------------------------------------------------------------------------------------------
    # low-level helpers (each enqueues exactly one HIP entry point)
    # ------------------------------------------------------------------------------------------------
    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _padded(self, shape: tuple) -> torch.Tensor:
        """Zero-bordered volume, reused for every convolution with this input shape (producers rewrite the whole
        interior; borders are never written, so they stay zero)."""
        buf = self._pool.get(shape)
        if buf is None:
            buf = torch.zeros(shape, dtype=torch.bfloat16, device=self.device)
            self._pool[shape] = buf
        return buf

    def _tap_table(self, kt, kh, kw, iH, iW, iC) -> torch.Tensor:
        key = (kt, kh, kw, iH, iW, iC)
        t = self._taps.get(key)
        if t is None:
            # same K order as _Conv packs the weights in: (dt, dh, channel block, dw)
            offs = [((dt * iH + dh) * iW + dw) * iC + cb * 64 for dt in range(kt) for dh in range(kh) for cb in range(iC // 64)
                    for dw in range(kw)]
            t = torch.tensor(offs, dtype=torch.int32, device=self.device)
            self._taps[key] = t
        return t

    def _conv(self, vol: torch.Tensor, conv: _Conv, out_thw: Tuple[int, int, int], stride: int = 1, residual=None) -> torch.Tensor:
        NB, iT, iH, iW, iC = vol.shape
        oT, oH, oW = out_thw
        kt, kh, kw = conv.ksize if len(conv.ksize) == 3 else (1,) + conv.ksize
        taps = self._tap_table(kt, kh, kw, iH, iW, iC)
        out = torch.empty(NB, oT, oH, oW, conv.cout_pad, dtype=torch.bfloat16, device=self.device)
        if self._splitk_ws is None and self.splitk_ws_bytes:
            self._splitk_ws = torch.empty(self.splitk_ws_bytes // 4, dtype=torch.float32, device=self.device)
        flags = self._flags
        # tap-reuse kernel: needs the (dt, dh, channel block, dw) K order and pays where the one-voxel border is a small share
        # of the plane it enumerates (240x360: 1.4 %, 120x180: 2.8 %, 60x90: 5.6 %; at 30x45 the split-K plain kernel wins)
        if (conv.blocked and stride == 1 and kh == 3 and kw == 3 and iH == oH + 2 and iW == oW + 2
                and iH * iW <= self.tap_reuse_max_waste * oH * oW):
            flags |= _lib.AETHER_CONV_TAP_REUSE
        rc = self._lib.aether_conv_gemm_bf16(vol.data_ptr(), NB, iT, iH, iW, iC, oT, oH, oW, stride, taps.data_ptr(), taps.numel(),
                                             conv.w.data_ptr(), conv.cout_pad, out.data_ptr(), conv.cout_pad, conv.b.data_ptr(),
                                             _lib.ptr(residual), conv.cout_pad if residual is not None else 0,
                                             _lib.ptr(self._splitk_ws), self.splitk_ws_bytes if self._splitk_ws is not None else 0,
                                             flags, self._stream())
        _lib.check(rc, "aether_conv_gemm_bf16")
        return out

    def _linear(self, x: torch.Tensor, conv: _Conv) -> torch.Tensor:
        """1x1x1 convolution (conv_shortcut) or a pre-gathered im2col matrix: plain GEMM over voxels."""
        rows = x.numel() // x.shape[-1]
        out = torch.empty(*x.shape[:-1], conv.cout_pad, dtype=torch.bfloat16, device=self.device)
        rc = self._lib.aether_gemm_bf16(x.data_ptr(), x.shape[-1], conv.w.data_ptr(), conv.w.shape[1], out.data_ptr(), conv.cout_pad,
                                        rows, conv.cout_pad, conv.w.shape[1], conv.b.data_ptr(), _lib.AETHER_EPI_BIAS, None, 0, None, None,
                                        0, 0, 0, None, 0, self._flags, self._stream())
        _lib.check(rc, "aether_gemm_bf16")
        return out

    def _norm_to_padded(self, x: torch.Tensor, norm: _Norm, pad_t: int, pad_hw: int, silu: bool, zq=None, eps=None, causal=None) -> torch.Tensor:
        """GroupNorm (+SpatialNorm3D) + SiLU of x [NB,T,H,W,C] into a zero-bordered volume [NB,T+pad_t,H+2p,W+2p,C].
        causal = (cache dict, key): the same launch also writes the two causal front frames (from cache[key] or copies of frame 0)
        and stores the chunk's last two frames back as cache[key] (aether_groupnorm_apply_causal; what the C launch plan uses)."""
        NB, T, H, W, Cc = x.shape
        G = self.config.norm_num_groups
        V = T * H * W
        nblk = max(1, min(256, (V + 127) // 128))     # partial-sum blocks per tile (per-group doubles; one workgroup per tile merges them in a fixed order)
        part = torch.empty(NB * nblk * 2 * Cc, dtype=torch.float32, device=self.device)
        stats = torch.empty(NB * G * 2, dtype=torch.float32, device=self.device)
        affine = torch.empty(NB * 2 * Cc, dtype=torch.float32, device=self.device)
        eps = 1e-6 if (eps is None or norm.spatial) else eps
        _lib.check(self._lib.aether_groupnorm_stats(x.data_ptr(), NB, V, Cc, G, float(eps), norm.gamma.data_ptr(), norm.beta.data_ptr(),
                                                    part.data_ptr(), nblk, stats.data_ptr(), affine.data_ptr(), self._stream()),
                   "aether_groupnorm_stats")
        vol = self._padded((NB, T + pad_t, H + 2 * pad_hw, W + 2 * pad_hw, Cc))
        cond, tmap, zT, zH, zW = None, None, 0, 0, 0
        if norm.spatial:
            _, zT, zH, zW, zC = zq.shape
            cond = torch.empty(NB * zT * zH * zW * 2 * Cc, dtype=torch.float32, device=self.device)
            _lib.check(self._lib.aether_spatial_cond(zq.data_ptr(), NB, zT * zH * zW, zC, Cc, norm.wy.data_ptr(), norm.by.data_ptr(),
                                                     norm.wb.data_ptr(), norm.bb.data_ptr(), cond.data_ptr(), self._stream()),
                       "aether_spatial_cond")
            tmap = (C.c_int * T)(*_nearest_time_map(T, zT))
        if causal is not None:
            cache, key = causal
            assert pad_t == 2
            prev = cache.get(key)
            nxt = torch.empty((NB, 2) + tuple(vol.shape[2:]), dtype=vol.dtype, device=vol.device)
            _lib.check(self._lib.aether_groupnorm_apply_causal(x.data_ptr(), NB, T, H, W, Cc, affine.data_ptr(), int(silu), vol.data_ptr(), vol.shape[2],
                                                               vol.shape[3], pad_hw, pad_hw, _lib.ptr(cond), zT, zH, zW, tmap, _lib.ptr(prev),
                                                               nxt.data_ptr(), self._stream()), "aether_groupnorm_apply_causal")
            cache[key] = nxt
            return vol
        if norm.spatial:
            rc = self._lib.aether_groupnorm_apply(x.data_ptr(), NB, T, H, W, Cc, affine.data_ptr(), int(silu), vol.data_ptr(), vol.shape[1],
                                                  vol.shape[2], vol.shape[3], pad_t, pad_hw, pad_hw, cond.data_ptr(), zT, zH, zW, tmap,
                                                  self._stream())
        else:
            rc = self._lib.aether_groupnorm_apply(x.data_ptr(), NB, T, H, W, Cc, affine.data_ptr(), int(silu), vol.data_ptr(), vol.shape[1],
                                                  vol.shape[2], vol.shape[3], pad_t, pad_hw, pad_hw, None, 0, 0, 0, None, self._stream())
        _lib.check(rc, "aether_groupnorm_apply")
        return vol

    def _resample(self, x: torch.Tensor, mode: int, out_shape: tuple, offs: Tuple[int, int, int]) -> torch.Tensor:
        NB, T, H, W, Cc = x.shape
        vol = self._padded(out_shape)
        rc = self._lib.aether_resample_pad(x.data_ptr(), NB, T, H, W, Cc, mode, vol.data_ptr(), out_shape[1], out_shape[2], out_shape[3],
                                           offs[0], offs[1], offs[2], self._stream())
        _lib.check(rc, "aether_resample_pad")
        return vol

    def _causal_front(self, vol: torch.Tensor, cache: Dict, key: str):
        """Fill the two causal front frames of a padded conv input from the cache of the previous chunk (or by
        replicating the first frame), then remember this chunk's last two input frames (CogVideoXCausalConv3d): one launch."""
        prev = cache.get(key)
        NB, Tp = vol.shape[0], vol.shape[1]
        nxt = torch.empty((NB, 2) + tuple(vol.shape[2:]), dtype=vol.dtype, device=vol.device)
        _lib.check(self._lib.aether_causal_front(vol.data_ptr(), NB, Tp, vol[0, 0].numel(), _lib.ptr(prev), nxt.data_ptr(), self._stream()),
                   "aether_causal_front")
        cache[key] = nxt

    def _causal_conv(self, x, norm, conv, cache, key, silu=True, zq=None, residual=None, eps=None):
        NB, T, H, W, _ = x.shape
        vol = self._norm_to_padded(x, norm, 2, 1, silu, zq, eps)
        self._causal_front(vol, cache, key)
        return self._conv(vol, conv, (T, H, W), 1, residual)

    def _resnet(self, x, r: _Resnet, cache, zq=None):
        eps = self.config.norm_eps
        h = self._causal_conv(x, r.norm1, r.conv1, cache, r.name + "conv1", True, zq, None, eps)
        skip = x if r.shortcut is None else self._linear(x, r.shortcut)
        return self._causal_conv(h, r.norm2, r.conv2, cache, r.name + "conv2", True, zq, skip, eps)

    # ------------------------------------------------------------------------------------------------
    # encoder / decoder over one frame chunk of NB equally shaped tiles
    # ------------------------------------------------------------------------------------------------
    def _im2col(self, src: torch.Tensor, conv: _Conv, crops: List[Tuple[int, int]], t0: int, T: int, H: int, W: int, first: bool):
        """src [C, T_all, H_all, W_all] bf16 (any strides); one A matrix [NB, T*H*W, Kpad] for all crops."""
        Kpad = conv.w.shape[1]
        A = torch.empty(len(crops), T * H * W, Kpad, dtype=torch.bfloat16, device=self.device)
        sC, sT, sH, sW = src.stride()
        for i, (y0, x0) in enumerate(crops):
            rc = self._lib.aether_im2col_first(src.data_ptr(), sC, sT, sH, sW, src.shape[0], t0, int(first), y0, x0, T, H, W,
                                               A[i].data_ptr(), Kpad, self._stream())
            _lib.check(rc, "aether_im2col_first")
        return A

    def _encode_chunk(self, video: torch.Tensor, crops, t0, T, H, W, first, cache):
        e = self.enc
        NB = len(crops)
        A = self._im2col(video, e.conv_in, crops, t0, T, H, W, first)
        x = self._linear(A, e.conv_in).view(NB, T, H, W, e.conv_in.cout_pad)
        for bi, blk in enumerate(e.down):
            for r in blk.resnets:
                x = self._resnet(x, r, cache)
            if blk.down is not None:
                NBx, Tx, Hx, Wx, Cx = x.shape
                Tn = (Tx // 2 + 1 if Tx % 2 else Tx // 2) if blk.compress_time else Tx
                vol = self._resample(x, 1 if blk.compress_time else 0, (NBx, Tn, Hx + 1, Wx + 1, Cx), (0, 0, 0))
                x = self._conv(vol, blk.down, (Tn, Hx // 2, Wx // 2), stride=2)
        for r in e.mid:
            x = self._resnet(x, r, cache)
        x = self._causal_conv(x, e.norm_out, e.conv_out, cache, "encoder.conv_out", True, None, None, 1e-6)
        return x[..., : 2 * self.config.latent_channels]      # [NB, T', h, w, 32]

    def _decode_chunk(self, z: torch.Tensor, crops, t0, T, H, W, first, cache):
        d = self.dec
        NB = len(crops)
        # channels-last latent volume of this chunk (SpatialNorm3D conditions every norm on it)
        zq = torch.stack([z[:, t0:t0 + T, y0:y0 + H, x0:x0 + W] for (y0, x0) in crops], 0).permute(0, 2, 3, 4, 1).contiguous()
        A = self._im2col(z, d.conv_in, crops, t0, T, H, W, first)
        x = self._linear(A, d.conv_in).view(NB, T, H, W, d.conv_in.cout_pad)
        for r in d.mid:
            x = self._resnet(x, r, cache, zq)
        for blk in d.up:
            for r in blk.resnets:
                x = self._resnet(x, r, cache, zq)
            if blk.up is not None:
                NBx, Tx, Hx, Wx, Cx = x.shape
                if blk.compress_time:
                    Tn = (2 * Tx - 1 if Tx % 2 else 2 * Tx) if Tx > 1 else 1
                    mode = 3
                else:
                    Tn, mode = Tx, 2
                vol = self._resample(x, mode, (NBx, Tn, 2 * Hx + 2, 2 * Wx + 2, Cx), (0, 1, 1))
                x = self._conv(vol, blk.up, (Tn, 2 * Hx, 2 * Wx), stride=1)
        x = self._causal_conv(x, d.norm_out, d.conv_out, cache, "decoder.conv_out", True, zq, None, 1e-6)
        return x[..., : self.config.out_channels]              # [NB, T_out, H_out, W_out, 3]

    # ------------------------------------------------------------------------------------------------
    @staticmethod
    def _chunks(n: int, bs: int):
        nb, rem = max(n // bs, 1), n % bs
        return [(bs * k + (0 if k == 0 else rem), min(bs * (k + 1) + rem, n)) for k in range(nb)]

    def _run_tiles(self, src: torch.Tensor, origins: List[Tuple[int, int]], tile_h: int, tile_w: int, bs: int, chunk_fn):
        """src [C, T, H, W]; returns {origin: tensor [1, C_out, T_out, h, w]} for every tile, batching tiles of equal shape."""
        _, T, H, W = src.shape
        groups: Dict[Tuple[int, int], List[Tuple[int, int]]] = {}
        for (y0, x0) in origins:
            groups.setdefault((min(tile_h, H - y0), min(tile_w, W - x0)), []).append((y0, x0))
        out = {}
        batches = [((th, tw), crops[i:i + 4]) for (th, tw), crops in groups.items() for i in range(0, len(crops), 4)]      # as the C launch plan: up to four per batch
        for (th, tw), crops in batches:
            cache: Dict = {}
            pieces = []
            for k, (s, e) in enumerate(self._chunks(T, bs)):
                pieces.append(chunk_fn(src, crops, s, e - s, th, tw, k == 0, cache))
            full = torch.cat(pieces, dim=1)                     # [NB, T_out, h, w, C]
            for i, o in enumerate(crops):
                out[o] = full[i].permute(3, 0, 1, 2).unsqueeze(0)
        return out

    @staticmethod
    def _blend(a: torch.Tensor, b: torch.Tensor, extent: int, dim: int) -> torch.Tensor:
        """diffusers blend_v (dim 3) / blend_h (dim 4): linear cross-fade of b's leading `extent` rows/cols with a's
        trailing ones, IN PLACE on b, each product rounded to bf16 before the sum exactly like the python loop."""
        extent = min(a.shape[dim], b.shape[dim], extent)
        if extent == 0:
            return b
        shape = [1] * 5
        shape[dim] = extent
        y = [i / extent for i in range(extent)]
        w_b = torch.tensor(y, dtype=torch.float32, device=b.device).view(shape)
        w_a = torch.tensor([1 - v for v in y], dtype=torch.float32, device=b.device).view(shape)
        a_tail = a.narrow(dim, a.shape[dim] - extent, extent)
        b_head = b.narrow(dim, 0, extent)
        b_head.copy_((a_tail.float() * w_a).to(b.dtype) + (b_head.float() * w_b).to(b.dtype))
        return b

    def _assemble(self, tiles, rows_y, cols_x, blend_h, blend_w, limit_h, limit_w):
        grid = [[tiles[(y, x)] for x in cols_x] for y in rows_y]
        out_rows = []
        for i, row in enumerate(grid):
            res = []
            for j, tile in enumerate(row):
                if i > 0:
                    tile = self._blend(grid[i - 1][j], tile, blend_h, 3)
                if j > 0:
                    tile = self._blend(row[j - 1], tile, blend_w, 4)
                res.append(tile[:, :, :, :limit_h, :limit_w])
            out_rows.append(torch.cat(res, dim=4))
        return torch.cat(out_rows, dim=3)

    # ------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def _encode(self, x: torch.Tensor) -> torch.Tensor:
        _, _, T, H, W = x.shape
        src = x[0]
        bs = self.num_sample_frames_batch_size
        if self.use_tiling and (W > self.tile_sample_min_width or H > self.tile_sample_min_height):
            sh = int(self.tile_sample_min_height * (1 - self.tile_overlap_factor_height))
            sw = int(self.tile_sample_min_width * (1 - self.tile_overlap_factor_width))
            bh = int(self.tile_latent_min_height * self.tile_overlap_factor_height)
            bw = int(self.tile_latent_min_width * self.tile_overlap_factor_width)
            rows_y, cols_x = list(range(0, H, sh)), list(range(0, W, sw))
            tiles = self._run_tiles(src, [(y, xx) for y in rows_y for xx in cols_x], self.tile_sample_min_height,
                                    self.tile_sample_min_width, bs, self._encode_chunk)
            return self._assemble(tiles, rows_y, cols_x, bh, bw, self.tile_latent_min_height - bh, self.tile_latent_min_width - bw)
        return self._run_tiles(src, [(0, 0)], H, W, bs, self._encode_chunk)[(0, 0)]

    @torch.no_grad()
    def encode(self, x: torch.Tensor, return_dict: bool = True):
        self._check_ready(x, self.config.in_channels)
        x = x.to(device=self.device, dtype=torch.bfloat16)
        run = (lambda s: self._run_c_plan(s, False)) if self.use_c_plan else self._encode
        h = torch.cat([run(s) for s in x.split(1)])               # batch items are independent (slicing or not)
        posterior = DiagonalGaussianDistribution(h)
        return SimpleNamespace(latent_dist=posterior) if return_dict else (posterior,)

    @torch.no_grad()
    def _decode(self, z: torch.Tensor) -> torch.Tensor:
        _, _, T, H, W = z.shape
        src = z[0]
        bs = self.num_latent_frames_batch_size
        if self.use_tiling and (W > self.tile_latent_min_width or H > self.tile_latent_min_height):
            sh = int(self.tile_latent_min_height * (1 - self.tile_overlap_factor_height))
            sw = int(self.tile_latent_min_width * (1 - self.tile_overlap_factor_width))
            bh = int(self.tile_sample_min_height * self.tile_overlap_factor_height)
            bw = int(self.tile_sample_min_width * self.tile_overlap_factor_width)
            rows_y, cols_x = list(range(0, H, sh)), list(range(0, W, sw))
            tiles = self._run_tiles(src, [(y, xx) for y in rows_y for xx in cols_x], self.tile_latent_min_height,
                                    self.tile_latent_min_width, bs, self._decode_chunk)
            return self._assemble(tiles, rows_y, cols_x, bh, bw, self.tile_sample_min_height - bh, self.tile_sample_min_width - bw)
        return self._run_tiles(src, [(0, 0)], H, W, bs, self._decode_chunk)[(0, 0)]

    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = True):
        self._check_ready(z, self.config.latent_channels)
        z = z.to(device=self.device, dtype=torch.bfloat16)
        run = (lambda s: self._run_c_plan(s, True)) if self.use_c_plan else self._decode
        dec = torch.cat([run(z[i:i + 1]) for i in range(z.shape[0])])
        return SimpleNamespace(sample=dec) if return_dict else (dec,)

    # ---- two decodes at once (extension; the reference decodes the rgb and the disparity latents one after the other, P:931,936) ----------
    def _make_twin(self) -> "AetherVAE":
        """A second launch context over the SAME packed weights: its own C handle, workspace and hipGraphs (nothing is copied but the
        registration of the weight pointers), so two decodes can be in flight on two HIP streams."""
        twin = AetherVAE.__new__(AetherVAE)
        twin.__dict__.update(self.__dict__)
        twin._handle, twin._workspace, twin._ws_bytes, twin._graphs = None, None, None, {}
        twin._pool, twin._taps, twin._splitk_ws = {}, {}, None
        twin._twin = None
        twin._register_c_plan()
        return twin

    @torch.no_grad()
    def decode_pair(self, z_a: torch.Tensor, z_b: torch.Tensor):
        """`(decode(z_a).sample, decode(z_b).sample)` — the pipeline's two final decodes (rgb and disparity latents, P:931,936).
        With the two-lane launch plan (AETHER_VAE_TWO_LANES, the default) every decode already runs its tile batches on two streams, and the pair is
        simply the two calls one after the other: 0.74-0.78 s per pair at 41 x 480 x 720, where two ONE-lane decodes on two streams took 0.765 s and
        two TWO-lane decodes on two streams (four streams in all) 0.836 s (profiles/r04_vae_lanes_ab.json) — and no second workspace.
        Without the lanes flag the two decodes are enqueued on two HIP streams over a twin launch context (a second C handle over the same packed
        weights with a workspace and hipGraphs of its own: + 28.8 GB).  Same kernels, same order within each decode either way: results are
        bit-identical to the sequential calls (tests/test_vae_gpu.py::test_decode_pair_is_bit_identical)."""
        if self._flags & _lib.AETHER_VAE_TWO_LANES:
            return self.decode(z_a).sample, self.decode(z_b).sample
        if getattr(self, "_twin", None) is None:
            self._twin = self._make_twin()
            # a HIGH-PRIORITY stream: HIP gives priority streams hardware queues of their own, whereas a normal pool stream may share the
            # queue of the caller's stream (4 hardware queues, assigned round robin) — and two graphs on one queue do not overlap at all
            # (measured: the same pair 0.74 s in one process, 0.81 s = sequential in another, depending on how many streams existed before)
            self._side_stream = torch.cuda.Stream(device=self.device, priority=-1)
        if self._twin._flags != self._flags:          # the flags are baked into the C handle: a changed value needs a new twin
            self._twin = self._make_twin()
        for k in ("use_tiling", "use_slicing", "use_graphs", "use_c_plan"):
            setattr(self._twin, k, getattr(self, k))
        cur = torch.cuda.current_stream(self.device)
        self._side_stream.wait_stream(cur)
        with torch.cuda.stream(self._side_stream):
            out_b = self._twin.decode(z_b).sample
        out_a = self.decode(z_a).sample
        cur.wait_stream(self._side_stream)
        out_b.record_stream(cur)
        return out_a, out_b

    def _check_ready(self, x, channels):
        if not self._loaded:
            raise RuntimeError("AetherVAE: weights not loaded")
        if x.dim() != 5 or x.shape[1] != channels:
            raise ValueError(f"expected a [B,{channels},T,H,W] tensor, got {tuple(x.shape)}")
        if not torch.cuda.is_available():
            raise RuntimeError("AetherVAE needs an MI355X (no CPU fallback)")


def _nearest_time_map(T: int, zT: int) -> List[int]:
    """Source latent frame of output frame t under CogVideoXSpatialNorm3D's F.interpolate(mode="nearest"):
    when T > 1 and odd the first frame maps to latent frame 0 and the rest are resized separately."""
    def nearest(n_out, n_in):
        scale = n_in / n_out
        return [min(int(math.floor(i * scale)), n_in - 1) for i in range(n_out)]
    if T > 1 and T % 2 == 1:
        return [0] + [1 + s for s in nearest(T - 1, zT - 1)] if zT > 1 else [0] * T
    return nearest(T, zT)
