"""Window driver (D:235-251, 607-631) and its multi-process sharding, on CPU with the gloo backend (world_size 2)."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_window_starts_known_answers():
    from aether_amd.windows import get_window_starts
    assert get_window_starts(41, 41, 24) == [0]
    assert get_window_starts(128, 41, 24) == [0, 24, 48, 72, 87]                       # moviegen.mp4 (128 frames)
    assert get_window_starts(192, 41, 24) == [0, 24, 48, 72, 96, 120, 144, 151]        # veo2.mp4 (192 frames): 8 windows
    s = get_window_starts(209, 41, 24)
    assert len(s) == 8 and s[-1] == 168
    assert get_window_starts(30, 41, 24) == []                                          # shorter than one window
    assert get_window_starts(65, 41, 24) == [0, 24]


def test_shard_round_robin():
    from aether_amd.windows import shard
    items = list(range(8))
    for world in (1, 2, 4, 8):
        parts = [shard(items, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == items and all(len(p) == 8 // world for p in parts)
    assert shard(list(range(5)), 1, 2) == [1, 3]


def test_demo_cli_surface_matches_reference_defaults():
    """Every flag of D:52-203 with its default (the reference's store_true/default=True quirks included)."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import importlib
    demo = importlib.import_module("demo")
    a = demo.parse_args(["--task", "reconstruction"])
    expect = dict(video=None, image=None, goal=None, raymap_action=None, output_dir="outputs", seed=42, fps=12,
                  num_inference_steps=None, guidance_scale=None, use_dynamic_cfg=True, height=480, width=720, num_frames=41,
                  max_depth=100.0, rtol=0.2, cogvideox_pretrained_model_name_or_path="THUDM/CogVideoX-5b-I2V",
                  aether_pretrained_model_name_or_path="AetherWorldModel/AetherV1", smooth_camera=True, smooth_method="kalman",
                  sliding_window_stride=24, post_reconstruction=True, pointcloud_save_frame_interval=10, align_pointmaps=False)
    for k, v in expect.items():
        assert getattr(a, k) == v, k
    with pytest.raises(SystemExit):
        demo.parse_args(["--task", "segmentation"])
    with pytest.raises(SystemExit):
        demo.parse_args(["--task", "prediction", "--fps", "30"])


_WORKER = textwrap.dedent('''
    import os, sys, json
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, %(root)r)
    from types import SimpleNamespace
    from aether_amd.windows import get_window_starts, run_windows, blend_rgb
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo")
    video = np.random.default_rng(0).random((%(frames)d, 6, 8, 3), dtype=np.float32)
    def call(s):      # stand-in for one pipeline call: deterministic function of the window and a seeded generator
        g = torch.Generator().manual_seed(42)
        noise = torch.randn(17, 6, 8, generator=g).numpy()
        w = video[s:s + 17]
        return SimpleNamespace(rgb=np.sqrt(w) + 0.01 * noise[..., None], disparity=w.mean(-1) * noise, raymap=np.tile(w[:, :1, :1, :1].reshape(17, 1, 1, 1), (1, 6, 2, 3)))
    starts = get_window_starts(len(video), 17, 7)
    res = run_windows(call, starts)
    if res is not None:
        np.savez(%(out)r, rgb=blend_rgb(res, len(video)), disparity=np.stack([r.disparity for r in res]), raymap=np.stack([r.raymap for r in res]),
                 starts=np.asarray([r.start for r in res]))
    if world > 1:
        kept = run_windows(call, starts, keep_on_device=True)       # tensors are kept only on a CUDA gather device: under gloo everything is numpy
        if res is not None:
            assert all(isinstance(k.rgb, np.ndarray) and isinstance(k.disparity, np.ndarray) and isinstance(k.raymap, np.ndarray) for k in kept)
            assert all(np.array_equal(k.rgb, r.rgb) and np.array_equal(k.disparity, r.disparity)
                       and np.array_equal(k.raymap, r.raymap) and k.start == r.start for k, r in zip(kept, res))
        dist.barrier(); dist.destroy_process_group()
''')


@pytest.mark.parametrize("frames", [66, 41])
def test_gloo_world2_gather_is_bit_identical(tmp_path, frames):
    outs = []
    for world in (1, 2):
        out = str(tmp_path / f"w{world}.npz")
        script = tmp_path / f"worker{world}.py"
        script.write_text(_WORKER % dict(root=ROOT, frames=frames, out=out))
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
        if world == 1:
            cmd = [sys.executable, str(script)]
        else:
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                   "--master-port", "29517", str(script)]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(out))
    a, b = outs
    assert list(a["starts"]) == list(b["starts"]) and len(a["starts"]) >= 4
    for k in ("rgb", "disparity", "raymap"):
        assert np.array_equal(a[k], b[k]), k


def test_export_names_colour_map_and_flips(tmp_path):
    """save_output's pieces (D:425-521): file stems, colorize_depth against the reference's own output, the export flips."""
    from types import SimpleNamespace
    from aether_amd.export import colorize_depth, flip_for_export, output_stem
    assert output_stem("reconstruction", "assets/example_videos/moviegen.mp4", None, None) == "reconstruction_moviegen"
    assert output_stem("prediction", None, "assets/example_obs/car.png", None) == "prediction_car"
    assert output_stem("planning", None, "a/01_obs.png", "b/01_goal.v2.png") == "planning_01_obs_01_goal"
    z = np.load(os.path.join(ROOT, "tests", "golden", "export.npz"))
    assert np.array_equal(colorize_depth(z["disparity"]), z["colorized"])
    rng = np.random.default_rng(0)
    pm, poses = rng.random((2, 3, 4, 3)), rng.random((2, 4, 4))
    fpm, fposes = flip_for_export(pm, poses)
    assert np.array_equal(fpm, pm * np.array([-1, -1, 1.0]))
    sign = np.array([[1, 1, -1, -1], [1, 1, -1, -1], [-1, -1, 1, 1], [1, 1, 1, 1.0]])      # rows and columns 0, 1 negated, then t_x, t_y
    assert np.array_equal(fposes, poses * sign)
    # the CLI's writer: every array in one .npz under the reference's stem, the videos as mp4 (imageio) or first-frame PNGs
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import importlib
    demo = importlib.import_module("demo")
    args = SimpleNamespace(output_dir=str(tmp_path), task="prediction", video=None, image="x/car.png", goal=None)
    rgb, disp = rng.random((2, 6, 8, 3)).astype(np.float32), (0.1 + rng.random((2, 6, 8))).astype(np.float32)
    demo.save_output(args, rgb=rgb, disparity=disp, poses=poses, pointmap=rng.random((2, 6, 8, 3)), raymap=None)
    out = np.load(tmp_path / "prediction_car.npz")
    assert set(out.files) == {"rgb", "disparity", "pointmap", "poses"} and np.array_equal(out["poses"], fposes)
    assert any(f.name.startswith("prediction_car_rgb") for f in tmp_path.iterdir())
    assert any(f.name.startswith("prediction_car_disparity") for f in tmp_path.iterdir())


_MERGED_WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, %(root)r)
    from types import SimpleNamespace
    from aether_amd.windows import WindowResult, blend_and_merge_window_results, run_windows, run_windows_merged
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo")
    G = np.load(os.path.join(%(root)r, "tests", "golden", "blend.npz"))
    H, W = (int(v) for v in G["hw"])
    starts = [int(s) for s in G["starts"]]
    def call(s):      # the three overlapping windows of the reference-made blend fixture (valid raymaps: the merge fits cameras on them)
        k = starts.index(s)
        return SimpleNamespace(rgb=G[f"rgb_{k}"].astype(np.float32), disparity=G[f"disparity_{k}"].copy(), raymap=G[f"raymap_{k}"].copy())
    t = {}
    merged = run_windows_merged(call, starts, height=H, width=W, smooth_camera=False, timings=t, force_collective=%(force)r)
    assert "windows_and_gather" in t
    if merged is not None:
        np.savez(%(out)r, rgb=merged[0], disparity=merged[1], poses=merged[2], pointmaps=merged[3])
        f32 = None
    res = run_windows(call, starts)
    if res is not None:
        ref = blend_and_merge_window_results(res, height=H, width=W, smooth_camera=False, device="cpu")
        for a, b in zip(merged, ref):
            assert a.dtype == np.float64 and np.array_equal(a, b)
    m32 = run_windows_merged(call, starts, height=H, width=W, smooth_camera=False, out_dtype=np.float32)
    if m32 is not None:
        assert m32[0].dtype == np.float32 and m32[3].dtype == np.float32 and np.array_equal(m32[0], merged[0].astype(np.float32))
    if world > 1:
        dist.barrier(); dist.destroy_process_group()
''')


@pytest.mark.parametrize("world", [1, 2, 4])
def test_gloo_round_wise_gather_and_incremental_merge(tmp_path, world):
    """`run_windows_merged` (what scripts/demo.py and bench.py run for a long clip): one gather per ROUND of windows, rank 0 merges a round
    while the next one is computed.  3 windows on 1 / 2 / 4 ranks (uneven rounds; more ranks than windows): bit-identical to the gather-at-the-
    end path, and equal to the reference's own merge of the same windows (tests/golden/blend.npz)."""
    out = str(tmp_path / f"m{world}.npz")
    script = tmp_path / f"mworker{world}.py"
    script.write_text(_MERGED_WORKER % dict(root=ROOT, out=out, force=False))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, str(script)] if world == 1 else [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                                                            "--master-addr", "127.0.0.1", "--master-port", "29519", str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    got, gold = np.load(out), np.load(os.path.join(ROOT, "tests", "golden", "blend.npz"))
    for k, ref in (("rgb", "plain_rgb"), ("disparity", "plain_disparity"), ("poses", "plain_poses"), ("pointmaps", "plain_pointmaps")):
        err = np.abs(got[k] - gold[ref]).max() / np.abs(gold[ref]).max()
        assert err < 1e-5, (k, err)


_EIGHT_WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, %(root)r)
    from types import SimpleNamespace
    from aether_amd import geometry as G
    from aether_amd.windows import blend_and_merge_window_results, get_window_starts, run_windows, run_windows_merged
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist.init_process_group("gloo")
    F, H, W, total = 9, 16, 24, 42
    starts = get_window_starts(total, F, 5)                      # [0, 5, ..., 30, 33]: 8 windows, overlaps 4 and (tail) 6 — the shape of [0, 24, ..., 144, 151]
    assert len(starts) == 8 and starts[-2:] == [30, 33]
    t = np.linspace(0, 1, total)
    K = np.array([[20.0, 0, W / 2], [0, 20.0, H / 2], [0, 0, 1.0]])
    yy, xx = np.mgrid[0:H, 0:W]
    def call(s):                                                 # a deterministic function of the start only: every rank can make every window
        k = starts.index(s)
        pose = np.tile(np.eye(4), (F, 1, 1))
        a = 0.4 * (t[s:s + F] - t[s])
        pose[:, 0, 0], pose[:, 0, 2], pose[:, 2, 0], pose[:, 2, 2] = np.cos(a), np.sin(a), -np.sin(a), np.cos(a)
        pose[:, 0, 3], pose[:, 2, 3] = 0.3 * (t[s:s + F] - t[s]) * (1 + 0.1 * k), 1.1 * (t[s:s + F] - t[s]) * (1 + 0.1 * k)
        ray = G.camera_pose_to_raymap(pose.astype(np.float32), np.tile(K * np.array([[8], [8], [1.0]]), (F, 1, 1)), H=H * 8, W=W * 8)
        disp = np.stack([(0.3 + 0.25 * np.sin(0.3 * xx + 0.2 * (s + f)) * np.cos(0.2 * yy)) * (1 + 0.15 * k) for f in range(F)]).astype(np.float32)
        rgb = np.stack([np.stack([(xx * 3 + yy * 5 + (s + f) * 7 + c) %% 17 / 17.0 for c in range(3)], -1) for f in range(F)]).astype(np.float32)
        return SimpleNamespace(rgb=rgb, disparity=disp, raymap=ray.astype(np.float32))
    merged = run_windows_merged(call, starts, height=H * 8, width=W * 8, smooth_camera=True, smooth_method="kalman")
    res = run_windows(call, starts)
    if dist.get_rank() == 0:
        assert merged is not None and len(res) == 8 and [r.start for r in res] == starts
        serial = blend_and_merge_window_results([SimpleNamespace(start=s, **vars(call(s))) for s in starts], height=H * 8, width=W * 8, smooth_camera=True,
                                                smooth_method="kalman", device="cpu")
        for a, b, what in zip(merged, serial, ("rgb", "disparity", "poses", "pointmaps")):
            assert a.shape[0] == total and np.array_equal(a, b), what
        open(%(out)r, "w").write("ok")
    else:
        assert merged is None and res is None
    dist.barrier(); dist.destroy_process_group()
''')


def test_gloo_eight_ranks_eight_windows_one_round(tmp_path):
    """The N = 8 configuration of BASELINE configs[4] — 8 windows on 8 ranks, ONE round, one gather of eight payloads, rank 0 merging all eight —
    over gloo: bit-identical to the serial merge of the same windows (default Kalman smoothing)."""
    out = str(tmp_path / "eight.ok")
    script = tmp_path / "eight_worker.py"
    script.write_text(_EIGHT_WORKER % dict(root=ROOT, out=out))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1", "--master-port", "29523", str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and os.path.exists(out), r.stderr[-3000:]
