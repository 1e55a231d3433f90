"""Evaluation-harness windows against outputs of the reference's own functions
(tests/golden/eval_windows.npz, produced by tools/make_golden.py from evaluation/video_depth/launch_aether.py:81-287 and
evaluation/rel_pose/launch_aether.py:173-250), plus the window plans' known answers and the rank sharding on gloo."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from make_golden import EvalFakePipeline, eval_pattern   # noqa: E402  (pure numpy helpers; nothing of the reference is touched)


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "eval_windows.npz"))


def test_window_plans_known_answers():
    from aether_amd.eval_windows import max_window_frames, plan_depth_windows, pose_window_starts
    assert [max_window_frames(n) for n in (200, 41, 40, 33, 30, 17)] == [41, 41, 33, 33, 25, 17]
    p = plan_depth_windows(110, 480, 720, 110)
    assert p.window_frames == 41 and p.crops == [(0, 480)] and not p.horizontal and [a for a, _ in p.times] == [0, 8, 16, 24, 32, 40, 48, 56, 64, 69]
    p = plan_depth_windows(25, 480, 900, 17)
    assert p.horizontal and p.crops == [(0, 720), (180, 900)] and p.times == [(0, 17), (8, 25)] and len(p.units) == 4
    assert (p.units[1].w0, p.units[1].w1, p.units[1].t0) == (180, 900, 0) and (p.units[2].w0, p.units[2].t0) == (0, 8)
    p = plan_depth_windows(17, 1000, 720, 17)                       # three crops down the height, stride (1000-480)//2
    assert not p.horizontal and p.crops == [(0, 480), (260, 740), (520, 1000)] and p.times == [(0, 17)]
    p = plan_depth_windows(41, 436, 1024, 41)                       # Sintel-like frame, narrower than the target height
    assert p.crops == [(0, 720), (304, 1024)]
    with pytest.raises(AssertionError):
        plan_depth_windows(41, 600, 900, 41)                        # the reference tiles one axis only
    assert pose_window_starts(110) == ([0, 32, 64, 69], 41)
    assert pose_window_starts(41) == ([0], 41) and pose_window_starts(30) == ([0, 5], 25) and pose_window_starts(73) == ([0, 32], 41)


@pytest.mark.parametrize("tag", ["wide", "tall", "plain"])
def test_depth_windows_match_reference(gold, tag):
    import torch
    from aether_amd.eval_windows import process_with_sliding_window
    t, h, w, total = (int(v) for v in gold[f"{tag}_dims"])
    rgb, disp = process_with_sliding_window(EvalFakePipeline(), eval_pattern(t, h, w), 4, total, 7, device=torch.device("cpu"))
    assert list(rgb.shape) == list(gold[f"{tag}_rgb_shape"]) and list(disp.shape) == list(gold[f"{tag}_disp_shape"])
    assert disp.shape == (t, h, w)
    assert abs(float(rgb.sum(dtype=np.float64)) - float(gold[f"{tag}_rgb_sum"])) <= 1e-6 * float(gold[f"{tag}_rgb_sum"])
    np.testing.assert_allclose(np.asarray(disp, np.float64)[::3, ::7, ::11], gold[f"{tag}_disp_sub"], rtol=2e-6, atol=1e-7)
    assert abs(float(np.asarray(disp, np.float64).sum()) - float(gold[f"{tag}_disp_sum"])) <= 2e-6 * float(gold[f"{tag}_disp_sum"])


def test_pose_window_blend_matches_reference(gold):
    from aether_amd.eval_windows import blend_window_outputs
    wins = [{key: gold[f"pose_in_{k}_{key}"].copy() for key in ("rgb", "disparity", "poses", "focals")} for k in range(3)]
    for k, wd in enumerate(wins):
        wd["range"] = tuple(int(v) for v in gold[f"pose_in_{k}_range"])
    res = blend_window_outputs(wins, smooth=lambda p: p)            # the fixture was made with the smoother replaced by the identity
    assert tuple(res["range"]) == tuple(int(v) for v in gold["pose_out_range"]) == (0, 19)
    for key in ("rgb", "disparity", "focals"):
        np.testing.assert_allclose(res[key], gold[f"pose_out_{key}"], rtol=2e-6, atol=1e-7, err_msg=key)
    assert res["poses"].shape == (19, 4, 4)
    np.testing.assert_allclose(res["poses"], gold["pose_out_poses"], rtol=1e-6, atol=1e-8)
    # default smoother: the Kalman restatement runs and keeps a rigid trajectory
    wins = [{key: gold[f"pose_in_{k}_{key}"].copy() for key in ("rgb", "disparity", "poses", "focals")} for k in range(3)]
    for k, wd in enumerate(wins):
        wd["range"] = tuple(int(v) for v in gold[f"pose_in_{k}_range"])
    sm = blend_window_outputs(wins)["poses"]
    assert sm.shape == (19, 4, 4) and np.allclose(sm[:, 3], [0, 0, 0, 1])
    assert np.allclose(np.einsum("nij,nkj->nik", sm[:, :3, :3], sm[:, :3, :3]), np.eye(3), atol=1e-6)
    assert np.abs(sm[:, :3, 3] - gold["pose_out_poses"][:, :3, 3]).max() < 0.2


def test_pose_driver_end_to_end_with_fake_pipeline():
    """process_video_with_sliding_window: window starts, per-window raymap decoding with the Kalman smoother, merge."""
    import torch
    from aether_amd.eval_windows import process_video_with_sliding_window
    from aether_amd.geometry import camera_pose_to_raymap
    t, H, W = 30, 48, 72
    tt = np.linspace(0, 1, t)
    world = np.tile(np.eye(4), (t, 1, 1))
    world[:, 0, 3], world[:, 2, 3] = 0.3 * tt, 1.0 * tt
    K = np.array([[60.0, 0, W / 2], [0, 60.0, H / 2], [0, 0, 1.0]])
    calls = []

    def pipe(video, num_inference_steps, num_frames, generator, return_dict, fps):
        s = int(round(float(video[0, 0, 0, 0]) * 1000))             # the clip carries its frame index in the first pixel
        calls.append((s, num_frames))
        rel = np.linalg.inv(world[s]) @ world[s:s + num_frames]
        ray = camera_pose_to_raymap(camera_pose=rel, intrinsic=np.tile(K, (num_frames, 1, 1)), H=H, W=W).astype(np.float32)
        disp = np.full((num_frames, H, W), 0.5, np.float32)
        return np.asarray(video, np.float32)[None], disp[None], ray[None]

    video = np.zeros((1, t, H, W, 3), np.float32)
    video[0, :, 0, 0, 0] = np.arange(t) / 1000
    res = process_video_with_sliding_window(pipe, video, 4, 0, device=torch.device("cpu"))
    assert calls == [(0, 25), (5, 25)] and res["range"] == (0, 30)
    assert res["poses"].shape == (30, 4, 4) and res["rgb"].shape == (30, H, W, 3) and res["focals"].shape == (30,)
    assert np.allclose(res["focals"], 60.0, rtol=2e-2)
    step = np.linalg.norm(np.diff(res["poses"][:, :3, 3], axis=0), axis=1)
    assert np.all(step > 0) and step.std() < 0.35 * step.mean()    # one smooth forward trajectory, no jump at the seam


_WORKER = '''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tools"))
from make_golden import EvalFakePipeline, eval_pattern
from aether_amd.eval_windows import plan_depth_windows, process_with_sliding_window
world = int(os.environ.get("WORLD_SIZE", "1"))
if world > 1:
    dist.init_process_group("gloo")
t, h, w, total = 25, 480, 900, 17
plan = plan_depth_windows(t, h, w, total)
class Pipe(EvalFakePipeline):            # the stand-in's scale must depend on the UNIT, not on this rank's call count
    def __call__(self, video, **kw):
        key = (round(float(video[0, 0, 0, 0]) * 101), round(float(video[0, 0, 1, 0]) * 101), round(float(video[0, 1, 0, 0]) * 101))
        self.calls = hash(key) %% 7
        return super().__call__(video, **kw)
res = process_with_sliding_window(Pipe(), eval_pattern(t, h, w), 4, total, 7, device=torch.device("cpu"))
if res is not None:
    np.savez(%(out)r, rgb=res[0], disparity=res[1])
else:
    assert dist.get_rank() != 0
if world > 1:
    dist.barrier(); dist.destroy_process_group()
'''


def test_depth_units_shard_over_ranks_gloo(tmp_path):
    outs = []
    for world in (1, 2):
        out = str(tmp_path / f"w{world}.npz")
        script = tmp_path / f"worker{world}.py"
        script.write_text(_WORKER % dict(root=ROOT, out=out))
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", PYTHONHASHSEED="0")
        cmd = ([sys.executable, str(script)] if world == 1 else
               [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                "--master-port", "29525", str(script)])
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append(np.load(out))
    assert np.array_equal(outs[0]["rgb"], outs[1]["rgb"]) and np.array_equal(outs[0]["disparity"], outs[1]["disparity"])


def test_window_plans_cover_everything():
    """Size-independent properties of the plans: every frame and every pixel row/column is inside some unit, consecutive units
    overlap by at least the reference's minimum (60 rows / 90 columns; one frame), unit shapes are what the pipeline accepts."""
    from hypothesis import given, settings, strategies as st
    from aether_amd.eval_windows import plan_depth_windows, pose_window_starts
    from aether_amd.windows import get_window_starts

    @settings(max_examples=200, deadline=None)
    @given(st.integers(17, 400), st.integers(480, 1500), st.booleans())
    def depth(t, extent, horizontal):
        h, w = (480, max(extent, 720)) if horizontal else (extent, 720)
        p = plan_depth_windows(t, h, w, t)
        assert p.window_frames in (17, 25, 33, 41) and len(p.units) == len(p.times) * len(p.crops)
        assert p.times[0][0] == 0 and p.times[-1][1] == t and all(b - a == p.window_frames for a, b in p.times)
        assert all(nb[0] < pa[1] for pa, nb in zip(p.times, p.times[1:]))                    # temporal windows overlap
        ext, tgt, min_ov = (w, 720, 90) if p.horizontal else (h, 480, 60)
        # reference quirk kept: the crop stride is floored, so up to len(crops) - 2 trailing rows / columns stay uncovered
        assert p.crops[0][0] == 0 and ext - max(len(p.crops) - 2, 0) <= p.crops[-1][1] <= ext and all(b - a == tgt for a, b in p.crops)
        assert all(pa[1] - nb[0] >= min_ov for pa, nb in zip(p.crops, p.crops[1:]))
        assert all((u.t1 - u.t0, u.h1 - u.h0, u.w1 - u.w0) == (p.window_frames, 480, 720) for u in p.units)

    @settings(max_examples=200, deadline=None)
    @given(st.integers(17, 600))
    def pose(t):
        starts, nf = pose_window_starts(t)
        assert nf in (17, 25, 33, 41) and starts[0] == 0 and starts[-1] + nf == t and starts == sorted(set(starts))
        assert all(b - a <= 32 and a + nf > b for a, b in zip(starts, starts[1:]))           # consecutive windows overlap

    @settings(max_examples=200, deadline=None)
    @given(st.integers(41, 600), st.integers(1, 40))
    def demo(n, stride):
        starts = get_window_starts(n, 41, stride)
        assert starts[0] == 0 and starts[-1] + 41 == n and starts == sorted(set(starts))
        assert all(b - a <= stride for a, b in zip(starts, starts[1:]))

    depth(); pose(); demo()
