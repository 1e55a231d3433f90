"""The rows SURVEY.md §8 marks "next", driven through the NATIVE pipeline on MI355X:

  f1  checkpoint loader (D:206-228): a diffusers-layout folder (sharded transformer safetensors + index, VAE, scheduler config) written
      from seeded weights -> `from_pretrained` of all three modules on the GPU -> forward / encode / decode against the fp32 oracle
      loaded from the SAME files;
  f4  evaluation-harness drivers (evaluation/video_depth/launch_aether.py:81-287, evaluation/rel_pose/launch_aether.py:124-250):
      `process_with_sliding_window` (two overlapping 480x720 crops of a 480x900 clip, merged) and `process_video_with_sliding_window`
      (two overlapping temporal windows, merged) with the native pipeline against the same drivers around the oracle pipeline;
  e   RCCL inside pytest: `run_windows(force_collective=True)` (payload packed on the device, `dist.gather` over nccl) and the guidance
      split's all-gather in the only RCCL configuration a one-GPU box allows, a one-rank group.

Reduced geometry: small modules, 17-frame windows and — because the CPU oracle needs minutes per 480 x 720 call — a 96 x 240 unit size
(the drivers' 480 x 720 constants are patched for the test; the reference's sizes are pinned on the CPU by tests/test_eval_windows_cpu.py).
"""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

UH, UW = 96, 240                      # unit (crop) size of the drivers in this test
TKW = dict(num_attention_heads=8, num_layers=2, text_embed_dim=128, time_embed_dim=64, max_text_seq_length=20, sample_width=UW // 8,
           sample_height=UH // 8, sample_frames=17)
VKW = dict(block_out_channels=(64, 128, 128, 128), layers_per_block=1, sample_height=UH, sample_width=UW)


@pytest.fixture()
def small_units(monkeypatch):
    import aether_amd.eval_windows as ew
    monkeypatch.setattr(ew, "TARGET_H", UH)
    monkeypatch.setattr(ew, "TARGET_W", UW)


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-12))


@pytest.fixture(scope="module")
def modules(cuda, hip_lib):
    from aether_amd.transformer import AetherTransformer3D
    from aether_amd.vae import AetherVAE
    from oracle.dit import DitConfig, OracleTransformer3D, init_random_ as init_dit
    from oracle.vae import OracleVAE, VaeConfig, init_random_ as init_vae
    dit32 = init_dit(OracleTransformer3D(DitConfig(**TKW)), seed=1)
    dsd = {k: v.to(torch.bfloat16) for k, v in dit32.state_dict().items()}
    dit32.load_state_dict({k: v.float() for k, v in dsd.items()})
    vae32 = init_vae(OracleVAE(VaeConfig(**VKW)), seed=2)
    vsd = {k: v.to(torch.bfloat16) for k, v in vae32.state_dict().items()}
    vae32.load_state_dict({k: v.float() for k, v in vsd.items()})
    vae32.enable_tiling(); vae32.enable_slicing()
    ndit = AetherTransformer3D(TKW, device=cuda).load_state_dict(dsd)
    nvae = AetherVAE(VKW, device=cuda).load_state_dict(vsd)
    nvae.enable_tiling(); nvae.enable_slicing()
    prompt = (torch.randn(1, 20, 128, generator=torch.Generator().manual_seed(0)) * 0.1).to(torch.bfloat16)
    return dict(dit32=dit32, vae32=vae32, ndit=ndit, nvae=nvae, prompt=prompt, dsd=dsd, vsd=vsd)


# ------------------------------------------------------------------------------------------------------------------ f1
def test_from_pretrained_runs_on_the_gpu(cuda, hip_lib, tmp_path, modules):
    """D:206-228: AutoencoderKLCogVideoX / CogVideoXDPMScheduler / CogVideoXTransformer3DModel `.from_pretrained(root, subfolder=...)`."""
    from safetensors.torch import load_file, save_file
    from test_checkpoint_cpu import _write_transformer
    from aether_amd.scheduler import CogVideoXDPMScheduler
    from aether_amd.transformer import AetherTransformer3D
    from aether_amd.vae import AetherVAE
    from oracle.dit import DitConfig, OracleTransformer3D
    from oracle.rope import crop_region_for_grid, rope_3d
    from oracle.vae import OracleVAE, VaeConfig
    root = str(tmp_path)
    tkw = dict(TKW, sample_width=12, sample_height=8, sample_frames=9, use_learned_positional_embeddings=True)
    from oracle.dit import init_random_ as init_dit
    sd = {k: v.to(torch.bfloat16) for k, v in init_dit(OracleTransformer3D(DitConfig(**tkw)), 3).state_dict().items()}
    _write_transformer(root, sd, tkw, shards=3)
    vkw = dict(block_out_channels=[64, 128, 128, 128], layers_per_block=1, sample_height=96, sample_width=240)
    os.makedirs(os.path.join(root, "vae"))
    json.dump(dict(vkw, _class_name="AutoencoderKLCogVideoX", scaling_factor=0.7, latent_channels=16), open(os.path.join(root, "vae", "config.json"), "w"))
    small_vsd = {k: v.to(torch.bfloat16) for k, v in __import__("oracle.vae", fromlist=["x"]).init_random_(OracleVAE(VaeConfig(**vkw)), 4).state_dict().items()}
    save_file({k: v.contiguous() for k, v in small_vsd.items()}, os.path.join(root, "vae", "diffusion_pytorch_model.safetensors"))
    os.makedirs(os.path.join(root, "scheduler"))
    json.dump({"_class_name": "CogVideoXDDIMScheduler", "beta_end": 0.012, "beta_schedule": "scaled_linear", "beta_start": 0.00085,
               "clip_sample": False, "num_train_timesteps": 1000, "prediction_type": "v_prediction", "rescale_betas_zero_snr": True,
               "set_alpha_to_one": True, "snr_shift_scale": 1.0, "steps_offset": 0, "timestep_spacing": "trailing"},
              open(os.path.join(root, "scheduler", "scheduler_config.json"), "w"))

    # ---- the three loads of D:215-227 --------------------------------------------------------------------------------------------
    vae = AetherVAE.from_pretrained(root, subfolder="vae", torch_dtype=torch.bfloat16)
    sched = CogVideoXDPMScheduler.from_pretrained(root, subfolder="scheduler")
    dit = AetherTransformer3D.from_pretrained(root, subfolder="transformer", torch_dtype=torch.bfloat16)
    vae.enable_slicing(); vae.enable_tiling()
    sched.set_timesteps(4, device=cuda)
    assert sched.timesteps.tolist() == [999, 749, 499, 249]

    # ---- oracle loaded from the SAME files ------------------------------------------------------------------------------------------
    files = sorted(f for f in os.listdir(os.path.join(root, "transformer")) if f.endswith(".safetensors"))
    assert len(files) == 3
    osd = {}
    for f in files:
        osd.update(load_file(os.path.join(root, "transformer", f)))
    odit = OracleTransformer3D(DitConfig(**tkw))
    odit.load_state_dict({k: v.float() for k, v in osd.items()})
    ovae = OracleVAE(VaeConfig(**vkw))
    ovae.load_state_dict({k: v.float() for k, v in load_file(os.path.join(root, "vae", "diffusion_pytorch_model.safetensors")).items()})
    ovae.enable_tiling()

    g = torch.Generator().manual_seed(0)
    hidden = torch.randn(1, 3, 96, 8, 12, generator=g).to(torch.bfloat16)            # 9 frames = sample_frames: the LEARNED table is used
    text = (torch.randn(1, 20, 128, generator=g) * 0.1).to(torch.bfloat16)
    t = torch.tensor([499], dtype=torch.int64)
    rope = rope_3d(64, crop_region_for_grid((4, 6), 6, 4), (4, 6), 3)
    with torch.no_grad():
        ref = odit(hidden.float(), text.float(), t, image_rotary_emb=rope)[0]
    out = dit(hidden_states=hidden.to(cuda), encoder_hidden_states=text.to(cuda), timestep=t.to(cuda), ofs=None,
              image_rotary_emb=(rope[0].to(cuda), rope[1].to(cuda)), return_dict=False)[0]
    assert dit._pos_current == "learned"
    e = _rel(out.float().cpu().numpy(), ref.numpy())
    x = (torch.rand(1, 3, 9, 96, 240, generator=g) * 2 - 1).to(torch.bfloat16)
    z = torch.randn(1, 16, 3, 12, 30, generator=g).to(torch.bfloat16)
    with torch.no_grad():
        ref_e, ref_d = ovae.encode(x.float()).latent_dist.mean, ovae.decode(z.float()).sample
    ee = _rel(vae.encode(x.to(cuda)).latent_dist.mean.float().cpu().numpy(), ref_e.numpy())
    ed = _rel(vae.decode(z.to(cuda)).sample.float().cpu().numpy(), ref_d.numpy())
    print(f"\nfrom_pretrained on the GPU vs the oracle loaded from the same files: DiT rel-L2 {e:.3e}, VAE encode {ee:.3e}, decode {ed:.3e}")
    assert e < 1.5e-2 and ee < 2e-2 and ed < 2e-2, (e, ee, ed)


# ------------------------------------------------------------------------------------------------------------------ f4
class _OraclePipe:
    """The oracle sampler behind the call signature the evaluation drivers use (EV:229-240, EP:142-153)."""

    def __init__(self, m):
        self.m = m
        self._execution_device = torch.device("cpu")

    def __call__(self, video, num_inference_steps, num_frames, generator, return_dict, fps):
        from aether_amd.scheduler import CogVideoXDPMScheduler
        from oracle.pipeline import sample
        from oracle.rope import prepare_rope
        v = torch.from_numpy(np.ascontiguousarray(video, dtype=np.float32)).permute(0, 3, 1, 2) * 2 - 1
        rgb, disp, ray = sample("reconstruction", self.m["dit32"], self.m["vae32"], CogVideoXDPMScheduler(), self.m["prompt"], video=v, height=UH,
                                width=UW, num_frames=num_frames, num_inference_steps=num_inference_steps, generator=generator,
                                rope=prepare_rope(UH, UW, (num_frames - 1) // 4 + 1, fps), compute_dtype=torch.float32)
        return rgb.numpy()[None], disp.numpy()[None], ray.numpy()[None]


def _native_pipe(m):
    from aether.pipelines.aetherv1_pipeline_cogvideox import AetherV1PipelineCogVideoX
    from aether_amd.scheduler import CogVideoXDPMScheduler
    pipe = AetherV1PipelineCogVideoX(tokenizer=None, text_encoder=None, vae=m["nvae"], scheduler=CogVideoXDPMScheduler(), transformer=m["ndit"],
                                     empty_prompt_embeds=m["prompt"])
    pipe.set_progress_bar_config(disable=True)
    return pipe


def _clip(t, h, w):
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    g = np.random.default_rng(1)
    v = np.stack([np.stack([0.5 + 0.4 * np.sin(0.02 * xx + 0.15 * k + c) * np.cos(0.017 * yy) for c in range(3)], -1) for k in range(t)])
    return np.clip(v + 0.02 * g.standard_normal(v.shape).astype(np.float32), 0, 1).astype(np.float32)[None]


def test_depth_driver_native_vs_oracle(cuda, modules, small_units):
    """EV:81-287 on a 25 x 96 x 300 clip with 96 x 240 units: two crops overlapping by 180 columns x two 17-frame windows = 4 units, each a
    full pipeline call (CPU generator of the same seed -> identical draws on both sides), disparities chained along the width and then
    along time by the least-squares scale + cross-fade."""
    from aether_amd.eval_windows import plan_depth_windows, process_with_sliding_window
    T_, H_, W_ = 25, UH, 300
    clip = _clip(T_, H_, W_)
    assert len(plan_depth_windows(T_, H_, W_, 17).units) == 4
    cpu = torch.device("cpu")
    ref_rgb, ref_disp = process_with_sliding_window(_OraclePipe(modules), clip, 2, 17, 7, device=cpu)
    rgb, disp = process_with_sliding_window(_native_pipe(modules), clip, 2, 17, 7, device=cpu)
    assert disp.shape == ref_disp.shape == (T_, H_, W_) and rgb.shape == ref_rgb.shape == (17, UH, UW, 3)
    e_rgb, e_disp = _rel(rgb, ref_rgb), _rel(disp, ref_disp)
    print(f"\ndepth driver (4 units) native vs fp32 oracle: rgb rel-L2 {e_rgb:.3e}, merged disparity rel-L2 {e_disp:.3e}")
    assert np.isfinite(disp).all() and e_rgb < 2e-2 and e_disp < 3e-2, (e_rgb, e_disp)


def test_pose_driver_native_vs_oracle(cuda, modules, small_units):
    """EP:124-250 on 21 frames: two 17-frame windows (starts 0 and 4), raymap -> poses per window, scale / similarity alignment over
    the 13 shared frames, cross-fades.  Colour and disparity are compared; the poses come out of random-weight raymaps (an
    ill-conditioned fit on noise), so they are only checked for shape, finiteness and rigidity."""
    from aether_amd.eval_windows import pose_window_starts, process_video_with_sliding_window
    assert pose_window_starts(21) == ([0, 4], 17)
    clip = _clip(21, UH, UW)
    cpu = torch.device("cpu")
    ref = process_video_with_sliding_window(_OraclePipe(modules), clip, 2, 3, device=cpu)
    got = process_video_with_sliding_window(_native_pipe(modules), clip, 2, 3, device=cpu)
    assert got["range"] == ref["range"] == (0, 21) and got["poses"].shape == (21, 4, 4) and got["focals"].shape == (21,)
    e_rgb, e_disp = _rel(got["rgb"], ref["rgb"]), _rel(got["disparity"], ref["disparity"])
    print(f"\npose driver (2 windows) native vs fp32 oracle: rgb rel-L2 {e_rgb:.3e}, merged disparity rel-L2 {e_disp:.3e}")
    assert e_rgb < 2e-2 and e_disp < 3e-2, (e_rgb, e_disp)
    assert np.isfinite(got["poses"]).all() and np.allclose(got["poses"][:, 3], [0, 0, 0, 1])
    R = got["poses"][:, :3, :3]
    assert np.allclose(np.einsum("nij,nkj->nik", R, R), np.eye(3), atol=1e-5)


# ------------------------------------------------------------------------------------------------------------------ e (RCCL, one rank)
_NCCL_WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, %(root)r)
    from types import SimpleNamespace
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from aether_amd.windows import blend_and_merge_window_results, get_window_starts, run_windows, run_windows_merged
    from aether_amd.geometry import camera_pose_to_raymap
    H, W, F = 48, 72, 17
    rng = np.random.default_rng(0)
    video = rng.random((31, H, W, 3), dtype=np.float32)
    K = np.tile(np.array([[60.0, 0, W / 2], [0, 60.0, H / 2], [0, 0, 1.0]]), (F, 1, 1))
    poses = np.tile(np.eye(4), (F, 1, 1)); poses[:, 2, 3] = np.linspace(0, 1, F)
    ray = camera_pose_to_raymap(camera_pose=poses, intrinsic=K, H=H, W=W).astype(np.float32)
    def call(s):                                   # a pipeline call that leaves its outputs on the device (keep_outputs_on_device)
        w = torch.from_numpy(video[s:s + F]).to(dev)
        return SimpleNamespace(rgb=w, disparity=w.mean(-1) + 0.2, raymap=torch.from_numpy(ray).to(dev))
    starts = get_window_starts(31, F, 7)
    plain = run_windows(call, starts)                                                    # no collective
    coll = run_windows(call, starts, gather_device=dev, keep_on_device=True, force_collective=True)    # dist.gather over nccl, device payload
    assert len(coll) == len(plain) == len(starts) == 3
    for a, b in zip(coll, plain):
        assert a.start == b.start and a.rgb.is_cuda and a.disparity.is_cuda
        assert np.array_equal(a.rgb.cpu().numpy(), b.rgb) and np.array_equal(a.disparity.cpu().numpy(), b.disparity) and np.array_equal(a.raymap, b.raymap)
    host = blend_and_merge_window_results(plain, height=H, width=W, smooth_camera=False)
    devm = blend_and_merge_window_results(coll, height=H, width=W, smooth_camera=False, device=dev)
    for x, y, tol in zip(host, devm, (1e-6, 1e-5, 1e-9, 1e-4)):
        assert np.allclose(np.asarray(x), np.asarray(y), rtol=tol, atol=tol), float(np.abs(np.asarray(x) - np.asarray(y)).max())
    # the pipelined form (one nccl gather per round of windows, rank 0 merging round j on a side stream while round j + 1 runs): the same values,
    # as float64 and as float32 into pinned host buffers
    t = {}
    m64 = run_windows_merged(call, starts, height=H, width=W, gather_device=dev, smooth_camera=False, force_collective=True, timings=t)
    m32 = run_windows_merged(call, starts, height=H, width=W, gather_device=dev, smooth_camera=False, force_collective=True, out_dtype=np.float32, pinned=True)
    assert "merge_tail" in t
    for x, y, z in zip(devm, m64, m32):
        assert np.array_equal(np.asarray(x), np.asarray(y)), float(np.abs(np.asarray(x) - np.asarray(y)).max())
        assert np.array_equal(np.asarray(y).astype(np.float32) if z.dtype == np.float32 else np.asarray(y), z)
    # the guidance split's exchange (pipeline._gather_pair): an all_gather of bf16 device tensors over nccl
    from aether.pipelines.aetherv1_pipeline_cogvideox import AetherV1PipelineCogVideoX
    pipe = AetherV1PipelineCogVideoX.__new__(AetherV1PipelineCogVideoX)
    pipe._cfg_group = dist.group.WORLD
    x = torch.randn(1, 3, 56, 6, 9, device=dev).to(torch.bfloat16)
    got = pipe._gather_pair(x)
    assert got.shape[0] == 1 and torch.equal(got, x)
    dist.barrier(); dist.destroy_process_group()
    print("NCCL-ONE-RANK-OK")
''')


def test_rccl_paths_in_a_one_rank_group(cuda, hip_lib, tmp_path):
    script = tmp_path / "nccl_worker.py"
    script.write_text(_NCCL_WORKER % dict(root=ROOT))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0 and "NCCL-ONE-RANK-OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
