"""Shared definitions of the FULL-GEOMETRY parity cases (BASELINE configs[1]: 41 x 480 x 720 clip, latent 11 x 60 x 90,
S = 226 + 14 850 tokens, 42 blocks, width 3072; VAE at its real widths).  Used by

  * tools/make_fullsize_golden.py — runs the fp32 CPU oracle ONCE, offline, in the build container (~1 h on 8 vCPUs) and writes
    tests/golden/fullsize_*.npz;
  * tests/test_fullsize_parity_gpu.py — regenerates the same seeded weights and inputs on the GPU box, runs the HIP path and
    compares with those fixtures.

Everything here is deterministic on the CPU (torch CPU generators, numpy) so both sides see identical bf16-representable weights
and inputs.  Test infrastructure: imports oracle/, never imported by aether_amd/.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
DIT_SEED, VAE_SEED, CLIP_SEED, DIT_INPUT_SEED = 11, 2, 42, 1
FRAMES, HEIGHT, WIDTH = 41, 480, 720
LAT_F, LAT_H, LAT_W = 11, 60, 90
DIT_KW: dict = {}
VAE_KW: dict = {}
TEXT_LEN, TEXT_DIM = 226, 4096
if os.environ.get("AETHER_FULLSIZE_DRYRUN"):
    # plumbing check of the generator + the test on a toy geometry (seconds on a CPU); never written into tests/golden
    FRAMES, HEIGHT, WIDTH = 17, 96, 240
    LAT_F, LAT_H, LAT_W = 5, 12, 30
    TEXT_LEN, TEXT_DIM = 20, 128
    DIT_KW = dict(num_attention_heads=8, num_layers=2, text_embed_dim=128, time_embed_dim=64, max_text_seq_length=20,
                  sample_width=LAT_W, sample_height=LAT_H, sample_frames=FRAMES)
    VAE_KW = dict(block_out_channels=(64, 128, 128, 128), layers_per_block=1, sample_height=HEIGHT, sample_width=WIDTH)
    GOLDEN_DIR = os.environ["AETHER_FULLSIZE_DRYRUN"]
CLIP_STEPS = 4                               # the reference's default for reconstruction (P:257-261)
GUIDED_STEPS = 2                             # guided steps of the prediction / planning fixtures (2 x B = 2 = four 42-block forwards)
GUIDED_SEED = 42
TRAJ_STEPS = 10                              # the longer reconstruction trajectory (drift against the step count: 4 -> 10)
HEADLINE_STEPS = 50                          # BASELINE configs[1] itself: reconstruction, 50 steps
HEADLINE_KEEP = (0, 1, 2, 3, 4, 6, 9, 14, 19, 24, 29, 34, 39, 44, 49)   # steps whose latents the 50-step fixture keeps
GUIDED_LONG_KEEP = tuple(range(15)) + (19, 24, 29, 34, 39, 44, 49)   # steps whose latents the 50-guided-step fixture keeps
DEC_STRIDE = 8                               # decoded pixels kept in the fixture: every 8th row / column, all frames


def _empty_module(ctor):
    """Build an nn.Module without paying for its default initialisation (5.6 B parameters): meta tensors -> uninitialised CPU
    storage; init_random_ then writes EVERY parameter from its own generator."""
    with torch.device("meta"):
        m = ctor()
    return m.to_empty(device="cpu")


def _round_to_bf16_(model):
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(p.to(torch.bfloat16).float())
    return model


def build_oracle_dit(num_layers: int = 42):
    """fp32 oracle transformer whose weights are exactly representable in bf16 (what the native module stores)."""
    from oracle.dit import DitConfig, OracleTransformer3D, init_random_
    cfg = DitConfig(**{"num_layers": num_layers, **DIT_KW})
    return _round_to_bf16_(init_random_(_empty_module(lambda: OracleTransformer3D(cfg)), seed=DIT_SEED)), cfg


def build_oracle_vae():
    from oracle.vae import OracleVAE, VaeConfig, init_random_
    vae = _round_to_bf16_(init_random_(_empty_module(lambda: OracleVAE(VaeConfig(**VAE_KW))), seed=VAE_SEED))
    vae.enable_tiling()
    vae.enable_slicing()
    return vae


def bf16_state_dict(model):
    return {k: v.to(torch.bfloat16) for k, v in model.state_dict().items()}


def dit_inputs():
    """One transformer call at the BASELINE shape, B = 1 (the `transformer(...)` call of P:865-875)."""
    g = torch.Generator().manual_seed(DIT_INPUT_SEED)
    hidden = torch.randn(1, LAT_F, 96, LAT_H, LAT_W, generator=g).to(torch.bfloat16)
    text = (torch.randn(1, TEXT_LEN, TEXT_DIM, generator=g) * 0.1).to(torch.bfloat16)
    t = torch.tensor([999], dtype=torch.int64)
    return hidden, text, t


def prompt_embeds():
    return (torch.randn(1, TEXT_LEN, TEXT_DIM, generator=torch.Generator().manual_seed(0)) * 0.1).to(torch.bfloat16)


def clip_video() -> np.ndarray:
    """[41, 480, 720, 3] float32 in [0, 1]: smooth moving sinusoid field + a little noise (image-like GroupNorm statistics)."""
    yy, xx = np.mgrid[0:HEIGHT, 0:WIDTH].astype(np.float32)
    g = np.random.default_rng(0)
    v = np.stack([np.stack([0.5 + 0.4 * np.sin(0.02 * xx + 0.1 * t + c) * np.cos(0.015 * yy) for c in range(3)], -1) for t in range(FRAMES)], 0)
    v = v + 0.03 * g.standard_normal(v.shape).astype(np.float32)
    return np.clip(v, 0.0, 1.0).astype(np.float32)


WINDOWS3_STARTS = (0, 24, 31)               # 72 frames: overlaps 17 AND 34, the two overlap lengths of get_window_starts(192, 41, 24) = [0, ..., 144, 151]
WINDOWS3_TOTAL = WINDOWS3_STARTS[-1] + FRAMES if not os.environ.get("AETHER_FULLSIZE_DRYRUN") else None
WINDOWS3_STEPS = 4


def windows3_starts():
    """Starts of the three-window case at the current geometry (dry run, 17-frame windows: [0, 10, 13] — overlaps 7 and 14)."""
    return list(WINDOWS3_STARTS) if FRAMES == 41 else [0, 10, 13]


def long_video(total: int) -> np.ndarray:
    """[total, H, W, 3] float32 in [0, 1]: the moving sinusoid field of `clip_video`, frame by frame (every frame draws its own noise from its own
    generator, so the first frames do not depend on `total`)."""
    yy, xx = np.mgrid[0:HEIGHT, 0:WIDTH].astype(np.float32)
    out = np.empty((total, HEIGHT, WIDTH, 3), np.float32)
    for t in range(total):
        f = np.stack([0.5 + 0.4 * np.sin(0.02 * xx + 0.1 * t + c) * np.cos(0.015 * yy) for c in range(3)], -1)
        f = f + 0.03 * np.random.default_rng(1000 + t).standard_normal(f.shape).astype(np.float32)
        out[t] = np.clip(f, 0.0, 1.0)
    return out


def video_as_model_input(video: np.ndarray) -> torch.Tensor:
    """What `preprocess_inputs` (P:462-512) hands to the VAE for a clip that needs no crop / resize: [F, 3, H, W] in [-1, 1]."""
    return torch.from_numpy(video).permute(0, 3, 1, 2) * 2 - 1


def rope_tables():
    from oracle.rope import prepare_rope
    return prepare_rope(HEIGHT, WIDTH, LAT_F, 12)


def bf16_bits(t: torch.Tensor) -> np.ndarray:
    """Exact storage of a bf16 tensor in an .npz (numpy has no bfloat16): its 16-bit patterns."""
    return t.to(torch.bfloat16).contiguous().view(torch.int16).numpy().view(np.uint16)


def from_bf16_bits(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16)


def metrics(out: torch.Tensor, ref: torch.Tensor) -> dict:
    d = out.double() - ref.double()
    return {"rel_l2": (d.norm() / ref.double().norm()).item(), "linf": d.abs().max().item(), "ref_max": ref.abs().max().item(),
            "linf_rel": (d.abs().max() / ref.abs().max()).item()}


def psnr(a: torch.Tensor, b: torch.Tensor) -> float:
    import math
    mse = ((a.double() - b.double()) ** 2).mean().item()
    return 10 * math.log10(1.0 / max(mse, 1e-20))


# ---- BASELINE configs[2] / configs[3]: the NAMED inputs (car.png + a forward-right raymap; 01_obs.png / 01_goal.png) -----------------
NAMED_INPUTS = os.path.join(ROOT, "tests", "golden", "named_inputs.npz")     # written by tools/make_named_inputs.py from /root/reference/assets


def named_image(name: str):
    """PIL image of one of the reference's example observations ('car', 'obs01', 'goal01'; 480 x 720 RGB)."""
    import PIL.Image
    z = np.load(NAMED_INPUTS)
    a = z[name]
    if os.environ.get("AETHER_FULLSIZE_DRYRUN"):
        a = np.ascontiguousarray(a[::5, ::3][:HEIGHT, :WIDTH])
    return PIL.Image.fromarray(a)


def image_as_model_input(img) -> torch.Tensor:
    """What `preprocess_inputs` (P:476-496) hands on for a PIL image that already has the target size: [1, 3, H, W] in [-1, 1]."""
    a = np.asarray(img.convert("RGB")).astype(np.float32) / 255.0
    return 2.0 * torch.from_numpy(a.transpose(2, 0, 1))[None] - 1.0


def forward_right_raymap() -> np.ndarray:
    """[41, 6, 60, 90] float32: the README's recipe for `--raymap_action` (camera_pose_to_raymap, U:867-961) on a forward-right
    trajectory in the first frame's camera coordinates (the reference's own assets/example_raymaps/raymap_forward_right.npy is
    not part of the mount: .MISSING_LARGE_BLOBS): eased translation 0.6 forward (+z) and 0.3 to the right (+x), yaw to the right
    up to 15 degrees, 60-degree horizontal field of view."""
    from aether_amd.geometry import forward_right_raymap as make
    return make(FRAMES, HEIGHT, WIDTH)


GUIDED_CASES = {
    "prediction": dict(image="car", goal=None, raymap=True),        # BASELINE configs[2]
    "planning": dict(image="obs01", goal="goal01", raymap=False),   # BASELINE configs[3]
}
