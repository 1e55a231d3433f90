"""BASELINE config 1 ("plumbing, no GPU"): the drop-in pipeline end to end on CPU.  The module slots are filled with
the (depth/width-reduced) fp32 ORACLE transformer and VAE — allowed here because this is a test — and the product
pipeline's outputs are compared, bit for bit, with oracle/pipeline.py's straight-line restatement of the reference's
`__call__` on the same seed.  Also pins the error strings of check_inputs (they are API, P:362-449)."""
import numpy as np
import pytest
import torch

H, W, F = 96, 240, 17   # scaled-down 480x720: tiles 48x120, strides 40x96, latent 12x30 (tiling composes exactly)


@pytest.fixture(scope="module")
def parts():
    from aether_amd.scheduler import CogVideoXDPMScheduler
    from oracle.dit import DitConfig, OracleTransformer3D, init_random_ as init_dit
    from oracle.vae import OracleVAE, VaeConfig, init_random_ as init_vae
    torch.manual_seed(0)
    tcfg = DitConfig(num_attention_heads=2, num_layers=1, text_embed_dim=64, time_embed_dim=32, max_text_seq_length=8,
                     sample_width=W // 8, sample_height=H // 8, sample_frames=F)
    dit = init_dit(OracleTransformer3D(tcfg), seed=1).to(torch.bfloat16)
    vcfg = VaeConfig(block_out_channels=(32, 64, 64, 64), layers_per_block=1, sample_height=H, sample_width=W)
    vae = init_vae(OracleVAE(vcfg), seed=2).to(torch.bfloat16)
    vae.enable_tiling()
    vae.enable_slicing()
    prompt = (torch.randn(1, 8, 64) * 0.1).to(torch.bfloat16)
    return dit, vae, CogVideoXDPMScheduler, prompt


def _pipe(parts):
    from aether.pipelines.aetherv1_pipeline_cogvideox import AetherV1PipelineCogVideoX
    dit, vae, Sched, prompt = parts
    p = AetherV1PipelineCogVideoX(tokenizer=None, text_encoder=None, vae=vae, scheduler=Sched(), transformer=dit,
                                  empty_prompt_embeds=prompt)
    p.set_progress_bar_config(disable=True)
    return p


def _video():
    g = np.random.default_rng(3)
    yy, xx = np.mgrid[0:H, 0:W]
    return np.stack([np.stack([0.5 + 0.5 * np.sin(0.1 * xx + 0.2 * t + c) * np.cos(0.07 * yy) for c in range(3)], -1)
                     for t in range(F)]).astype(np.float32) * 0.9 + 0.05 * g.random((F, H, W, 3), dtype=np.float32)


def _oracle_run(parts, task, **kw):
    from aether_amd.rope import resize_crop_region_for_grid, rotary_tables_3d
    from oracle.pipeline import sample
    dit, vae, Sched, prompt = parts
    rope = rotary_tables_3d(64, resize_crop_region_for_grid((H // 16, W // 16), W // 16, H // 16), (H // 16, W // 16), (F - 1) // 4 + 1, 1.0)
    return sample(task, dit, vae, Sched(), prompt, height=H, width=W, num_frames=F, rope=rope, **kw)


def test_reconstruction_matches_straight_line_oracle(parts):
    pipe = _pipe(parts)
    video = _video()
    out = pipe(task="reconstruction", video=video, height=H, width=W, num_frames=F, fps=12,
               generator=torch.Generator().manual_seed(42))
    assert out.rgb.shape == (F, H, W, 3) and out.disparity.shape == (F, H, W) and out.raymap.shape == (F, 6, H // 8, W // 8)
    assert out.rgb.dtype == np.float32 and 0 <= out.rgb.min() and out.rgb.max() <= 1
    v = torch.from_numpy(video).permute(0, 3, 1, 2) * 2 - 1
    rgb, disp, rm = _oracle_run(parts, "reconstruction", video=v, generator=torch.Generator().manual_seed(42))
    assert np.array_equal(out.rgb, rgb.numpy()) and np.array_equal(out.disparity, disp.numpy()) and np.array_equal(out.raymap, rm.numpy())
    again = pipe(task="reconstruction", video=video, height=H, width=W, num_frames=F, generator=torch.Generator().manual_seed(42))
    assert np.array_equal(out.rgb, again.rgb)                      # seeded -> reproducible
    other = pipe(task="reconstruction", video=video, height=H, width=W, num_frames=F, generator=torch.Generator().manual_seed(43))
    assert not np.array_equal(out.rgb, other.rgb)


@pytest.mark.parametrize("task", ["prediction", "planning"])
def test_cfg_tasks_match_straight_line_oracle(parts, task):
    pipe = _pipe(parts)
    img, goal = _video()[0], _video()[-1]
    raymap = np.random.default_rng(5).standard_normal((F, 6, H // 8, W // 8)).astype(np.float32)
    kw = dict(image=img, raymap=raymap, goal=goal if task == "planning" else None)
    out = pipe(task=task, height=H, width=W, num_frames=F, num_inference_steps=3, fps=12, generator=torch.Generator().manual_seed(1), **kw)
    t = lambda a: torch.from_numpy(a).permute(2, 0, 1)[None] * 2 - 1  # noqa: E731
    rgb, disp, rm = _oracle_run(parts, task, image=t(img), goal=t(goal) if task == "planning" else None,
                                raymap=torch.from_numpy(raymap)[None], num_inference_steps=3,
                                generator=torch.Generator().manual_seed(1))
    assert np.array_equal(out.rgb, rgb.numpy()) and np.array_equal(out.disparity, disp.numpy()) and np.array_equal(out.raymap, rm.numpy())


def test_task_inference_and_defaults(parts):
    pipe = _pipe(parts)
    assert pipe._default_num_inference_steps == {"reconstruction": 4, "prediction": 50, "planning": 50}
    assert pipe._default_guidance_scale == {"reconstruction": 1.0, "prediction": 3.0, "planning": 3.0}
    assert pipe._supported_tasks == ["reconstruction", "prediction", "planning"] and pipe._base_fps == 12
    assert pipe.vae_scale_factor_spatial == 8 and pipe.vae_scale_factor_temporal == 4 and pipe.vae_scaling_factor_image == 0.7


def test_check_inputs_messages(parts):
    pipe = _pipe(parts)
    img = _video()[0]
    cases = [
        (dict(task="foo", image=img), "`task` has to be one of ['reconstruction', 'prediction', 'planning']."),
        (dict(task="prediction"), "`image` or `video` has to be provided."),
        (dict(task="prediction", image=img, video=_video()), "`image` and `video` cannot both be provided."),
        (dict(task="reconstruction", image=img), "`image` is not supported for `reconstruction` task."),
        (dict(task="prediction", image=img, goal=img), "`goal` is only supported for `planning` task."),
        (dict(task="prediction", video=_video()), "`video` is only supported for `reconstruction` task."),
        (dict(task="prediction", image=img, height=60), "`height` and `width` have to be divisible by 8 but are 60 and 240."),
        (dict(task="prediction", image=img, num_frames=16), "`num_frames` has to be one of [17, 25, 33, 41]."),
        (dict(task="prediction", image=img, fps=30), "`fps` has to be one of [8, 10, 12, 15, 24]."),
        (dict(task="prediction", image=img, raymap="x"), "`raymap` has to be of type `torch.Tensor` or `np.ndarray`."),
    ]
    for kw, msg in cases:
        kw.setdefault("height", H); kw.setdefault("width", W); kw.setdefault("num_frames", F)
        with pytest.raises(ValueError) as e:
            pipe(**kw)
        assert str(e.value) == msg, (kw.keys(), str(e.value))
    with pytest.raises(ValueError, match="`raymap` shape is not correct"):
        pipe(task="prediction", image=img, raymap=np.zeros((F, 6, 4, 4), np.float32), height=H, width=W, num_frames=F)


def test_device_preprocess_fast_path_matches_host_path(parts):
    """uint8 / float32 clips whose centred crop already has the target size take the upload-then-normalise path; it must
    give the bits of the generic host path (crop copy, stack, 2x-1, bf16), and decline everything else."""
    pipe = _pipe(parts)
    g = np.random.default_rng(11)
    dev = torch.device("cpu")
    for arr in (_video(), (g.random((F, H, W, 3)) * 255).astype(np.uint8), _video()[0],
                g.random((5, H, W + 40, 3), dtype=np.float32),                 # wider source: window at left = 20
                (g.random((3, H + 16, W, 3)) * 255).astype(np.uint8)):         # taller source: window at top = 8
        fast = pipe._preprocess_frames_on_device(arr, H, W, dev)
        assert fast is not None and fast.dtype == torch.bfloat16 and fast.is_contiguous()
        slow = pipe._preprocess_image(arr, H, W).to(torch.bfloat16)
        assert fast.shape == slow.shape and torch.equal(fast, slow)
    assert pipe._preprocess_frames_on_device(g.random((2, H // 2, W // 2, 3), dtype=np.float32), H, W, dev) is None   # needs a resize
    f64 = g.random((2, H, W, 3))                                          # float64 clips (cv2.resize(...) / 255.0 in the evaluation loaders)
    assert torch.equal(pipe._preprocess_frames_on_device(f64, H, W, dev), pipe._preprocess_image(f64, H, W).to(torch.bfloat16))
    assert pipe._preprocess_frames_on_device(g.random((2, H, W, 3)).astype(np.float16), H, W, dev) is None            # other dtypes: host path
    assert pipe._preprocess_frames_on_device(torch.zeros(2, H, W, 3), H, W, dev) is None


_CFG_WORKER = '''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import test_pipeline_cpu as T
torch.set_num_threads(2)
world = int(os.environ.get("WORLD_SIZE", "1"))
if world > 1:
    dist.init_process_group("gloo")
parts = T.parts.__wrapped__() if hasattr(T.parts, "__wrapped__") else T.parts.__pytest_wrapped__.obj()
pipe = T._pipe(parts)
if world > 1:
    pipe.enable_cfg_parallel()
img, goal = T._video()[0], T._video()[-1]
out = {}
for task, kw in (("prediction", dict(image=img)), ("planning", dict(image=img, goal=goal))):
    r = pipe(task=task, height=T.H, width=T.W, num_frames=T.F, num_inference_steps=3, use_dynamic_cfg=True,
             generator=torch.Generator().manual_seed(5), **kw)
    out[task + "_rgb"], out[task + "_disparity"], out[task + "_raymap"] = r.rgb, r.disparity, r.raymap
# guidance off: the pair must not communicate (rank 1 may not even make the call)
if world == 1 or dist.get_rank() == 0:
    r = pipe(task="reconstruction", video=T._video(), height=T.H, width=T.W, num_frames=T.F, num_inference_steps=2,
             generator=torch.Generator().manual_seed(5))
    out["rec_rgb"] = r.rgb
# decode-parallel: BOTH ranks make the reconstruction call; rank 0 decodes rgb, rank 1 disparity, one all-gather
if world > 1:
    pipe.disable_cfg_parallel(); pipe.enable_decode_parallel()
r = pipe(task="reconstruction", video=T._video(), height=T.H, width=T.W, num_frames=T.F, num_inference_steps=2,
         generator=torch.Generator().manual_seed(5))
out["dec_rgb"], out["dec_disparity"], out["dec_raymap"] = r.rgb, r.disparity, r.raymap
np.savez(%(out)r + (".%%d" %% (dist.get_rank() if world > 1 else 0)), **out)
if world > 1:
    dist.barrier(); dist.destroy_process_group()
'''


def test_cfg_parallel_two_ranks_gloo(tmp_path):
    """one guidance branch per rank, exchanged every step.  Both ranks must hold the same outputs bit for bit,
    and they must agree with the single-process batch-of-two run (to bf16 rounding: a batch-1 and a batch-2 matmul may block
    their reductions differently)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for world in (1, 2):
        out = str(tmp_path / f"w{world}")
        script = tmp_path / f"cfg_worker{world}.py"
        script.write_text(_CFG_WORKER % dict(root=root, out=out))
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
        cmd = ([sys.executable, str(script)] if world == 1 else
               [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                "--master-port", "29519", str(script)])
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        outs[world] = [np.load(f"{out}.{k}.npz") for k in range(world)]
    single, (r0, r1) = outs[1][0], outs[2]
    for k in r1.files:
        assert np.array_equal(r0[k], r1[k]), f"ranks disagree on {k}"
    for k in single.files:
        a, b = single[k].astype(np.float64), r0[k].astype(np.float64)
        assert a.shape == b.shape and np.abs(a - b).max() <= 2e-2 * max(1.0, np.abs(a).max()), (k, np.abs(a - b).max())
    assert np.array_equal(single["rec_rgb"], r0["rec_rgb"])
    # decode-parallel reconstruction: the same kernels on the same (replicated) latents -> bit-identical to one rank, on both ranks
    for k in ("dec_rgb", "dec_disparity", "dec_raymap"):
        assert np.array_equal(single[k], r0[k]) and np.array_equal(single[k], r1[k]), k
