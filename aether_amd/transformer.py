"""MI355X-native stand-in for diffusers' `CogVideoXTransformer3DModel` at the reference's call site
(/root/reference/aether/pipelines/aetherv1_pipeline_cogvideox.py:865-875 and the `.config` reads at
P:308-318,326,338,545-548,722-728,808,815).

Host side only: weights are repacked once into the layout `aether_dit_forward` (include/aether_hip.h) expects and
registered on a C handle; `__call__` is ONE ctypes call that enqueues the whole forward on torch's current stream.
There is no PyTorch fallback — without libaether_hip.so / a gfx950 device the constructor raises.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from types import SimpleNamespace
from typing import Dict, Optional

import torch

from . import _lib

_CONFIG_DEFAULTS = dict(
    num_attention_heads=48, attention_head_dim=64, in_channels=96, out_channels=56, num_layers=42, patch_size=2,
    patch_size_t=None, text_embed_dim=4096, time_embed_dim=512, max_text_seq_length=226, sample_width=90,
    sample_height=60, sample_frames=41, temporal_compression_ratio=4, norm_eps=1e-5,
    use_rotary_positional_embeddings=True, use_learned_positional_embeddings=False, ofs_embed_dim=None,
    flip_sin_to_cos=True, freq_shift=0, activation_fn="gelu-approximate", timestep_activation_fn="silu",
    attention_bias=True, norm_elementwise_affine=True, spatial_interpolation_scale=1.875, temporal_interpolation_scale=1.0,
)


def sincos_position_table(dim: int, n_text: int, frames: int, height: int, width: int, spatial_scale: float = 1.875,
                          temporal_scale: float = 1.0) -> torch.Tensor:
    """[n_text + frames*height*width, dim] float64: zeros for the text rows, then diffusers' get_3d_sincos_pos_embed for a
    (height x width) patch grid — per token [temporal dim/4 | width 3dim/8 | height 3dim/8], each part [sin | cos] of
    pos / 10000^(2i/d) with spatial positions divided by `spatial_scale`.  This is what diffusers' CogVideoXPatchEmbed adds
    when the clip's frame count differs from `sample_frames` (it does NOT slice the learned table) [UPSTREAM-UNVERIFIED]."""
    def axis(d, pos):
        ang = torch.outer(pos.double(), 1.0 / 10000 ** (torch.arange(d // 2, dtype=torch.float64) / (d / 2.0)))
        return torch.cat([ang.sin(), ang.cos()], dim=1)
    dt, dsp = dim // 4, 3 * dim // 4
    et = axis(dt, torch.arange(frames, dtype=torch.float32) / temporal_scale)          # [F, dt]
    ew = axis(dsp // 2, torch.arange(width, dtype=torch.float32) / spatial_scale)      # [W, 3dim/8]
    eh = axis(dsp // 2, torch.arange(height, dtype=torch.float32) / spatial_scale)     # [H, 3dim/8]
    tab = torch.cat([et[:, None, None, :].expand(frames, height, width, dt), ew[None, None, :, :].expand(frames, height, width, dsp // 2),
                     eh[None, :, None, :].expand(frames, height, width, dsp // 2)], dim=-1).reshape(frames * height * width, dim)
    return torch.cat([torch.zeros(n_text, dim, dtype=torch.float64), tab], dim=0)


class AetherTransformer3D:
    """Duck-types the members the reference pipeline touches: `config`, `__call__(hidden_states=, encoder_hidden_states=,
    timestep=, ofs=, image_rotary_emb=, attention_kwargs=, return_dict=False)[0]`, `from_pretrained`, `to`, `dtype`."""

    def __init__(self, config: Optional[dict] = None, device: str = "cuda", flags: int = _lib.AETHER_GEMM_WIDE_STORE):
        cfg = dict(_CONFIG_DEFAULTS)
        cfg.update(config or {})
        self.config = SimpleNamespace(**cfg)
        self._check_config()
        self.device = torch.device(device)
        self.dtype = torch.bfloat16
        self._flags = flags
        self._weights: Dict[str, torch.Tensor] = {}
        self._handle = None
        self._workspace = None
        self._lib = _lib.load()

    # ------------------------------------------------------------------------------------------
    def _check_config(self):
        c = self.config
        if c.attention_head_dim != 64:
            raise ValueError("aether_amd: attention_head_dim must be 64")
        if c.patch_size_t is not None:
            raise ValueError("aether_amd: patch_size_t (CogVideoX 1.5 layout) is not implemented")
        if c.ofs_embed_dim is not None:
            raise ValueError("aether_amd: ofs embedding is not implemented (AetherV1 passes ofs=None, P:813-817)")
        if not c.use_rotary_positional_embeddings:
            raise ValueError("aether_amd: only the rotary variant is implemented")
        if c.activation_fn != "gelu-approximate" or c.timestep_activation_fn != "silu":
            raise ValueError("aether_amd: unsupported activation")
        # values the kernels hard-code (aether_timestep_sinusoid: [cos | sin], no frequency shift; biased linears; affine norms)
        if not c.flip_sin_to_cos or c.freq_shift != 0:
            raise ValueError("aether_amd: only flip_sin_to_cos=True, freq_shift=0 timestep features are implemented")
        if not c.attention_bias or not c.norm_elementwise_affine:
            raise ValueError("aether_amd: attention_bias=False / norm_elementwise_affine=False checkpoints are not supported")

    @property
    def inner_dim(self) -> int:
        return self.config.num_attention_heads * self.config.attention_head_dim

    # ------------------------------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, path, subfolder: Optional[str] = "transformer", torch_dtype=torch.bfloat16, device="cuda", **_):
        """Loads a diffusers-format folder: config.json + (sharded) diffusion_pytorch_model*.safetensors
        (same call as /root/reference/scripts/demo.py:223-227)."""
        from safetensors.torch import load_file

        root = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(root, "config.json")) as f:
            cfg = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        model = cls(cfg, device=device)
        index = os.path.join(root, "diffusion_pytorch_model.safetensors.index.json")
        sd = {}
        if os.path.exists(index):
            with open(index) as f:
                files = sorted(set(json.load(f)["weight_map"].values()))
        else:
            files = ["diffusion_pytorch_model.safetensors"]
        for fn in files:
            sd.update(load_file(os.path.join(root, fn)))
        model.load_state_dict(sd)
        return model

    def to(self, *args, **kwargs):
        return self

    def eval(self):
        return self

    # ------------------------------------------------------------------------------------------
    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        """Repack a diffusers-keyed state dict into the stacked device tensors of the C weight table."""
        c = self.config
        L, D = c.num_layers, self.inner_dim
        dev = self.device

        def bf(t):
            return t.detach().to(device=dev, dtype=torch.bfloat16).contiguous()

        def f32(t):
            return t.detach().to(device=dev, dtype=torch.float32).contiguous()

        def stack(fmt, conv):
            return conv(torch.stack([sd[fmt.format(i)] for i in range(L)], dim=0))

        w = {}
        w["patch_w"] = bf(sd["patch_embed.proj.weight"].reshape(D, -1))  # (c, dy, dx) column order
        w["patch_b"] = f32(sd["patch_embed.proj.bias"])
        w["text_w"] = bf(sd["patch_embed.text_proj.weight"])
        w["text_b"] = f32(sd["patch_embed.text_proj.bias"])
        if c.use_learned_positional_embeddings:
            w["pos_emb"] = bf(sd["patch_embed.pos_embedding"].reshape(-1, D))
        w["time_w1"] = bf(sd["time_embedding.linear_1.weight"])
        w["time_b1"] = f32(sd["time_embedding.linear_1.bias"])
        w["time_w2"] = bf(sd["time_embedding.linear_2.weight"])
        w["time_b2"] = f32(sd["time_embedding.linear_2.bias"])
        ada_w, ada_b = [], []
        for i in range(L):
            for n in ("norm1", "norm2"):
                ada_w.append(sd[f"transformer_blocks.{i}.{n}.linear.weight"])
                ada_b.append(sd[f"transformer_blocks.{i}.{n}.linear.bias"])
        ada_w.append(sd["norm_out.linear.weight"])
        ada_b.append(sd["norm_out.linear.bias"])
        w["adaln_w"] = bf(torch.cat(ada_w, dim=0))
        w["adaln_b"] = f32(torch.cat(ada_b, dim=0))
        tb = "transformer_blocks.{}."
        w["ln1_w"] = stack(tb + "norm1.norm.weight", f32)
        w["ln1_b"] = stack(tb + "norm1.norm.bias", f32)
        w["ln2_w"] = stack(tb + "norm2.norm.weight", f32)
        w["ln2_b"] = stack(tb + "norm2.norm.bias", f32)
        w["qkv_w"] = bf(torch.stack([torch.cat([sd[f"transformer_blocks.{i}.attn1.to_{x}.weight"] for x in "qkv"], 0) for i in range(L)], 0))
        w["qkv_b"] = f32(torch.stack([torch.cat([sd[f"transformer_blocks.{i}.attn1.to_{x}.bias"] for x in "qkv"], 0) for i in range(L)], 0))
        w["qn_w"] = stack(tb + "attn1.norm_q.weight", f32)
        w["qn_b"] = stack(tb + "attn1.norm_q.bias", f32)
        w["kn_w"] = stack(tb + "attn1.norm_k.weight", f32)
        w["kn_b"] = stack(tb + "attn1.norm_k.bias", f32)
        w["o_w"] = stack(tb + "attn1.to_out.0.weight", bf)
        w["o_b"] = stack(tb + "attn1.to_out.0.bias", f32)
        w["ff1_w"] = stack(tb + "ff.net.0.proj.weight", bf)
        w["ff1_b"] = stack(tb + "ff.net.0.proj.bias", f32)
        w["ff2_w"] = stack(tb + "ff.net.2.weight", bf)
        w["ff2_b"] = stack(tb + "ff.net.2.bias", f32)
        w["normf_w"] = f32(sd["norm_final.weight"])
        w["normf_b"] = f32(sd["norm_final.bias"])
        w["normo_w"] = f32(sd["norm_out.norm.weight"])
        w["normo_b"] = f32(sd["norm_out.norm.bias"])
        w["proj_w"] = bf(sd["proj_out.weight"])
        w["proj_b"] = f32(sd["proj_out.bias"])
        self._install(w)
        return self

    def _install(self, w: Dict[str, torch.Tensor]):
        c = self.config
        if self._handle is not None:
            self._lib.aether_dit_destroy(self._handle)
        cfg = _lib.AetherDitConfig(
            num_layers=c.num_layers, num_heads=c.num_attention_heads, head_dim=c.attention_head_dim,
            in_channels=c.in_channels, out_channels=c.out_channels, patch_size=c.patch_size, text_dim=c.text_embed_dim,
            time_embed_dim=c.time_embed_dim, ff_mult=4, max_text_len=c.max_text_seq_length, norm_eps=c.norm_eps,
            qk_norm_eps=1e-6, use_pos_embedding=int(bool(c.use_learned_positional_embeddings)), flags=self._flags)
        h = self._lib.aether_dit_create(C.byref(cfg))
        if not h:
            raise ValueError("aether_dit_create: " + self._lib.aether_last_error().decode())
        self._handle = h
        self._weights = w
        self._pos_tables = {}
        self._pos_current = None
        for name, t in w.items():
            if name == "pos_emb":
                continue                      # chosen per call (`_select_pos_table`)
            _lib.check(self._lib.aether_dit_set_weight(h, name.encode(), t.data_ptr()), f"set_weight({name})")

    def _select_pos_table(self, F: int, H: int, W: int):
        """diffusers CogVideoXPatchEmbed.forward [UPSTREAM-UNVERIFIED, SURVEY.md A.1]: the learned table is used only at the
        sample resolution AND `sample_frames` frames; at the sample resolution with another frame count (AetherV1's 41 frames on a
        base with sample_frames 49) the 3-D sin-cos table of the actual size is added instead; another resolution raises."""
        c = self.config
        if not c.use_learned_positional_embeddings:
            return
        if (c.sample_height, c.sample_width) != (H, W):
            raise ValueError("It is currently not possible to generate videos at a different resolution that the defaults. "
                             "This should only be the case with 'THUDM/CogVideoX-5b-I2V'.")
        key = "learned" if (F - 1) * c.temporal_compression_ratio + 1 == c.sample_frames else ("sincos", F)
        if key not in self._pos_tables:
            if key == "learned":
                tab = self._weights["pos_emb"]
            else:
                p = c.patch_size
                tab = sincos_position_table(self.inner_dim, c.max_text_seq_length, F, H // p, W // p, c.spatial_interpolation_scale,
                                            c.temporal_interpolation_scale).to(device=self.device, dtype=torch.bfloat16).contiguous()
            self._pos_tables[key] = tab
        if self._pos_current != key:
            tab = self._pos_tables[key]
            _lib.check(self._lib.aether_dit_set_pos_embedding(self._handle, tab.data_ptr(), tab.shape[0]), "set_pos_embedding")
            self._pos_current = key

    def init_random_weights(self, seed: int = 0, std: float = 0.02):
        """Seeded synthetic weights generated directly in HBM in the packed layout (benchmarks: real AetherV1
        weights are not available offline).  Scales mimic a trained model: N(0, min(2·std, fan_in^-1/2)) matrices,
        LayerNorm weights 1 ± 0.1, small biases, AdaLN linears ×4 so gates/modulations are O(0.1-1)."""
        c = self.config
        L, D, T = c.num_layers, self.inner_dim, c.time_embed_dim
        FF, Kp, Np = 4 * D, c.in_channels * c.patch_size ** 2, c.out_channels * c.patch_size ** 2
        g = torch.Generator(device=self.device).manual_seed(seed)

        def mat(*shape, fan_in, mult=1.0):
            s = min(2.0 * std, fan_in ** -0.5) * mult
            return (torch.randn(*shape, generator=g, device=self.device, dtype=torch.float32) * s).to(torch.bfloat16)

        def vec(*shape, mean=0.0, s=0.05):
            return mean + s * torch.randn(*shape, generator=g, device=self.device, dtype=torch.float32)

        w = {
            "patch_w": mat(D, Kp, fan_in=Kp), "patch_b": vec(D), "text_w": mat(D, c.text_embed_dim, fan_in=c.text_embed_dim),
            "text_b": vec(D), "time_w1": mat(T, D, fan_in=D), "time_b1": vec(T), "time_w2": mat(T, T, fan_in=T), "time_b2": vec(T),
            "adaln_w": mat(L * 12 * D + 2 * D, T, fan_in=T, mult=4.0), "adaln_b": vec(L * 12 * D + 2 * D),
            "ln1_w": vec(L, D, mean=1.0, s=0.1), "ln1_b": vec(L, D), "ln2_w": vec(L, D, mean=1.0, s=0.1), "ln2_b": vec(L, D),
            "qkv_w": mat(L, 3 * D, D, fan_in=D), "qkv_b": vec(L, 3 * D), "qn_w": vec(L, 64, mean=1.0, s=0.1), "qn_b": vec(L, 64),
            "kn_w": vec(L, 64, mean=1.0, s=0.1), "kn_b": vec(L, 64), "o_w": mat(L, D, D, fan_in=D), "o_b": vec(L, D),
            "ff1_w": mat(L, FF, D, fan_in=D), "ff1_b": vec(L, FF), "ff2_w": mat(L, D, FF, fan_in=FF), "ff2_b": vec(L, D),
            "normf_w": vec(D, mean=1.0, s=0.1), "normf_b": vec(D), "normo_w": vec(D, mean=1.0, s=0.1), "normo_b": vec(D),
            "proj_w": mat(Np, D, fan_in=D), "proj_b": vec(Np),
        }
        if c.use_learned_positional_embeddings:
            p = c.patch_size
            n_tok = c.max_text_seq_length + (c.sample_height // p) * (c.sample_width // p) * ((c.sample_frames - 1) // c.temporal_compression_ratio + 1)
            w["pos_emb"] = mat(n_tok, D, fan_in=400)
        self._install(w)
        return self

    def set_flags(self, flags: int):
        """Replace the kernel flags (AETHER_GEMM_* | AETHER_ATTN_*) of the C handle, e.g. `| AETHER_ATTN_EXACT_MAX`."""
        self._flags = flags
        _lib.check(self._lib.aether_dit_set_flags(self._handle, int(flags)), "aether_dit_set_flags")

    def set_profile(self, enable: bool):
        """Bracket every kernel enqueue of the forward with HIP events on the launch stream (bench.py's roofline leg)."""
        _lib.check(self._lib.aether_dit_set_profile(self._handle, int(enable)), "aether_dit_set_profile")

    def get_profile(self):
        """{class: (milliseconds, launches)} accumulated since the last call; synchronises on the recorded events."""
        ms = (C.c_float * len(_lib.PROF_CLASSES))()
        n = (C.c_int * len(_lib.PROF_CLASSES))()
        _lib.check(self._lib.aether_dit_get_profile(self._handle, ms, n), "aether_dit_get_profile")
        return {name: (float(ms[i]), int(n[i])) for i, name in enumerate(_lib.PROF_CLASSES)}

    def num_parameters(self) -> int:
        return sum(t.numel() for t in self._weights.values())

    def __del__(self):
        try:
            if self._handle is not None:
                self._lib.aether_dit_destroy(self._handle)
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------
    def _get_workspace(self, B, F, H, W):
        need = self._lib.aether_dit_workspace_bytes(self._handle, B, F, H, W)
        if self._workspace is None or self._workspace.numel() < need:
            self._workspace = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._workspace, need

    @torch.no_grad()
    def __call__(self, hidden_states, encoder_hidden_states, timestep, ofs=None, image_rotary_emb=None,
                 attention_kwargs=None, return_dict: bool = False, timestep_cond=None):
        if self._handle is None:
            raise RuntimeError("AetherTransformer3D: weights not loaded")
        if ofs is not None:
            raise ValueError("aether_amd: ofs embedding is not implemented")
        if image_rotary_emb is None:
            raise ValueError("aether_amd: image_rotary_emb is required (rotary variant)")
        c = self.config
        if hidden_states.dim() != 5 or hidden_states.shape[2] != c.in_channels:
            raise ValueError(f"hidden_states must be [B,F,{c.in_channels},H,W], got {tuple(hidden_states.shape)}")
        B, F, _, H, W = hidden_states.shape
        x = hidden_states.to(device=self.device, dtype=torch.bfloat16).contiguous()
        txt = encoder_hidden_states.to(device=self.device, dtype=torch.bfloat16).contiguous()
        if txt.shape != (B, c.max_text_seq_length, c.text_embed_dim):
            raise ValueError(f"encoder_hidden_states must be [{B},{c.max_text_seq_length},{c.text_embed_dim}], got {tuple(txt.shape)}")
        t = timestep.to(device=self.device, dtype=torch.float32).reshape(-1).contiguous()
        if t.numel() != B:
            raise ValueError("timestep must have one entry per batch element")
        cos, sin = image_rotary_emb
        cos = cos.to(device=self.device, dtype=torch.float32).contiguous()
        sin = sin.to(device=self.device, dtype=torch.float32).contiguous()
        n_vid = F * (H // c.patch_size) * (W // c.patch_size)
        if cos.shape != (n_vid, 64) or sin.shape != (n_vid, 64):
            raise ValueError(f"image_rotary_emb must be two [{n_vid},64] tensors, got {tuple(cos.shape)}")
        self._select_pos_table(F, H, W)
        out = torch.empty(B, F, c.out_channels, H, W, dtype=torch.bfloat16, device=self.device)
        ws, need = self._get_workspace(B, F, H, W)
        rc = self._lib.aether_dit_forward(self._handle, x.data_ptr(), txt.data_ptr(), t.data_ptr(), cos.data_ptr(),
                                          sin.data_ptr(), out.data_ptr(), B, F, H, W, ws.data_ptr(), need,
                                          _lib.current_stream())
        _lib.check(rc, "aether_dit_forward")
        if return_dict:
            return SimpleNamespace(sample=out)
        return (out,)

    forward = __call__
