"""Generates tests/golden/*.npz by running the parts of the REFERENCE that are importable in the build container
(pure numpy / torch-CPU helpers of /root/reference/aether/utils).  Run here only (the GPU box has no /root/reference):

    python tools/make_golden.py

Fixtures are small and committed; tests only read the .npz files.
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main():
    sys.path.insert(0, REF)
    # `plyfile` is absent here and only used by an export helper: stub it so postprocess_utils imports
    sys.modules.setdefault("plyfile", types.SimpleNamespace(PlyData=None, PlyElement=None))
    from aether.utils.preprocess_utils import imcrop_center
    os.makedirs(OUT, exist_ok=True)

    # --- imcrop_center (preprocess_utils.py:4-39) on frames of assorted aspect ratios -------------------------
    rng = np.random.default_rng(0)
    cases = {}
    for i, (h, w, th, tw) in enumerate([(48, 72, 480, 720), (50, 72, 480, 720), (48, 90, 480, 720), (37, 41, 480, 720),
                                        (64, 64, 60, 90), (30, 100, 17, 5), (91, 33, 8, 24)]):
        img = rng.random((1, h, w, 3), dtype=np.float32).astype(np.float16).astype(np.float32)
        out = imcrop_center(list(img), th, tw)
        cases[f"in_{i}"] = img.astype(np.float16)
        cases[f"out_{i}"] = np.stack(out).astype(np.float16)
        cases[f"tgt_{i}"] = np.array([th, tw])
    np.savez_compressed(os.path.join(OUT, "imcrop_center.npz"), **cases)

    # --- camera_pose_to_raymap / raymap_to_poses (postprocess_utils.py:919-961, 219-280) ----------------------
    from aether.utils import postprocess_utils as U
    n = 9
    t = np.linspace(0, 1, n)
    poses = np.tile(np.eye(4), (n, 1, 1))
    ang = 0.3 * t
    poses[:, 0, 0], poses[:, 0, 2], poses[:, 2, 0], poses[:, 2, 2] = np.cos(ang), np.sin(ang), -np.sin(ang), np.cos(ang)
    poses[:, 0, 3], poses[:, 2, 3] = 0.4 * t, 1.5 * t          # forward-right trajectory
    K = np.array([[400.0, 0, 360.0], [0, 400.0, 240.0], [0, 0, 1.0]])
    raymap = U.camera_pose_to_raymap(camera_pose=poses.copy(), intrinsic=np.tile(K, (n, 1, 1)))   # (N, 6, 60, 90)
    rec_poses, fov_x, fov_y = U.raymap_to_poses(raymap.copy(), ray_o_scale_inv=0.1)
    np.savez_compressed(os.path.join(OUT, "raymap.npz"), poses=poses, K=K, raymap=raymap.astype(np.float32),
                        rec_poses=rec_poses, fov_x=np.asarray(fov_x), fov_y=np.asarray(fov_y))
    make_blend_golden(U)
    make_eval_golden(U)
    # --- colorize_depth (postprocess_utils.py:49-56), the colour map of the disparity video ---------------------------------
    d = (np.random.default_rng(5).random((3, 6, 8)) * 0.9).astype(np.float32)
    d[0, 0, 0] = 0.0
    np.savez_compressed(os.path.join(OUT, "export.npz"), disparity=d, colorized=U.colorize_depth(d))
    print("wrote", os.listdir(OUT))


def _reference_blend_function(U):
    """The reference's blend_and_merge_window_results (scripts/demo.py:254-422) cannot be imported (imageio / rootutils are
    absent and the module runs rootutils at import), so its SOURCE SEGMENT is read from /root/reference at generation time and
    executed against the reference's own helper functions.  Nothing of it is stored in this repository but its outputs."""
    import argparse
    import ast
    from typing import List, Tuple

    import torch
    src = open(os.path.join(REF, "scripts", "demo.py")).read()
    node = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "blend_and_merge_window_results")
    ns = dict(np=np, torch=torch, List=List, Tuple=Tuple, argparse=argparse, AetherV1PipelineOutput=object,
              postprocess_pointmap=U.postprocess_pointmap, compute_scale=U.compute_scale, raymap_to_poses=U.raymap_to_poses,
              align_camera_extrinsics=U.align_camera_extrinsics, apply_transformation=U.apply_transformation,
              interpolate_poses=U.interpolate_poses, get_intrinsics=U.get_intrinsics, project=U.project)
    exec(compile(ast.Module(body=[node], type_ignores=[]), "reference_demo_blend", "exec"), ns)
    return ns["blend_and_merge_window_results"]


def make_blend_golden(U):
    """Three overlapping 9-frame windows (starts 0, 5, 10 of a 19-frame clip, 24x32 pixels, raymaps 3x4) whose cameras are
    expressed in per-window frames (different origin, orientation and scale) and whose disparities carry per-window scales:
    inputs + the reference's merged outputs for (align_pointmaps, smooth_camera, smooth_method) in
    {(False, False, -), (True, False, -), (False, True, simple)}, plus unit-level outputs of the helpers."""
    import types as _t
    rng = np.random.default_rng(7)
    H, W, F, starts, N = 24, 32, 9, [0, 5, 10], 19
    t = np.linspace(0, 1, N)
    world = np.tile(np.eye(4), (N, 1, 1))
    ang = 0.5 * t
    world[:, 0, 0], world[:, 0, 2], world[:, 2, 0], world[:, 2, 2] = np.cos(ang), np.sin(ang), -np.sin(ang), np.cos(ang)
    world[:, 0, 3], world[:, 1, 3], world[:, 2, 3] = 0.5 * t, 0.05 * np.sin(6 * t), 1.2 * t
    K = np.array([[28.0, 0, W / 2], [0, 28.0, H / 2], [0, 0, 1.0]])
    yy, xx = np.mgrid[0:H, 0:W]
    wins = []
    for k, s0 in enumerate(starts):
        rel = np.linalg.inv(world[s0]) @ world[s0:s0 + F]                 # each window predicts in its own first-frame system
        rel[:, :3, 3] *= (1.0, 0.7, 1.3)[k]                                # ... and at its own scale
        rel = rel + 1e-3 * rng.standard_normal(rel.shape) * np.array([1, 1, 1, 0])[:, None]   # prediction noise (rows 0-2)
        raymap = U.camera_pose_to_raymap(camera_pose=rel.copy(), intrinsic=np.tile(K, (F, 1, 1)), H=H, W=W).astype(np.float32)
        disp = np.stack([0.35 + 0.3 * np.sin(0.2 * xx + 0.3 * (s0 + f)) * np.cos(0.25 * yy) for f in range(F)])
        disp = ((0.9, 1.2, 0.75)[k] * disp + 0.01 * rng.random(disp.shape)).clip(0.02, 1).astype(np.float32)
        rgb = rng.random((F, H, W, 3), dtype=np.float32).astype(np.float16).astype(np.float32)   # stored as float16, exactly
        wins.append((rgb, disp, raymap))
    blend = _reference_blend_function(U)
    out = {"starts": np.array(starts), "hw": np.array([H, W])}
    for k, (rgb, disp, raymap) in enumerate(wins):
        out[f"rgb_{k}"], out[f"disparity_{k}"], out[f"raymap_{k}"] = rgb.astype(np.float16), disp, raymap
    for tag, (ap, sc, smeth) in {"plain": (False, False, "simple"), "aligned": (True, False, "simple"), "smooth": (False, True, "simple")}.items():
        results = [_t.SimpleNamespace(rgb=r.copy(), disparity=d.copy(), raymap=m.copy()) for r, d, m in wins]
        args = _t.SimpleNamespace(align_pointmaps=ap, smooth_camera=sc, smooth_method=smeth, width=W, height=H)
        m_rgb, m_disp, m_poses, m_pm = blend(results, list(starts), args)
        out[f"{tag}_disparity"], out[f"{tag}_poses"], out[f"{tag}_pointmaps"] = m_disp, m_poses, m_pm
        if tag == "plain":
            out["plain_rgb"] = m_rgb          # the colour cross-fade does not depend on the geometry options
    # unit level
    pm = U.postprocess_pointmap(wins[1][1].copy(), wins[1][2].copy(), vae_downsample_scale=8, ray_o_scale_inv=0.1)
    out["unit_pointmap"], out["unit_pose"], out["unit_K"] = pm["pointmap"], pm["camera_pose"], pm["intrinsics"]
    import torch
    p_a, _, _ = U.raymap_to_poses(wins[1][2].copy(), ray_o_scale_inv=0.1)
    p_b, _, _ = U.raymap_to_poses(wins[0][2].copy(), ray_o_scale_inv=0.1)
    aR, aT, aS = U.align_camera_extrinsics(torch.from_numpy(p_a[:4]), torch.from_numpy(p_b[-4:]))
    out["unit_align_R"], out["unit_align_T"], out["unit_align_s"] = aR.numpy(), aT.numpy(), np.array(float(aS))
    out["unit_applied"] = U.apply_transformation(torch.from_numpy(p_a), aR, aT, aS, return_extri=True).numpy()
    out["unit_interp"] = np.stack([U.interpolate_poses(p_a[0], p_b[3], wgt) for wgt in (0.0, 0.3, 1.0)])
    out["unit_smooth_gauss"] = U.smooth_poses(p_a.copy(), 5, "gaussian")
    out["unit_smooth_savgol"] = U.smooth_poses(p_a.copy(), 5, "savgol")
    out["unit_scale"] = np.array(U.compute_scale(wins[1][1][:4].reshape(1, -1, W), wins[0][1][-4:].reshape(1, -1, W),
                                                 wins[1][1][:4].reshape(1, -1, W) > 0.1))
    out["unit_project"] = U.project(1 / np.clip(wins[0][1][2].astype(np.float64), 1e-8, 1e8), K, p_b[2])
    # poses / scalars stay float64; images and point maps are stored as float32 (tests compare at 1e-5)
    small = {k: (v.astype(np.float32) if isinstance(v, np.ndarray) and v.dtype == np.float64 and v.size > 1000 else v) for k, v in out.items()}
    np.savez_compressed(os.path.join(OUT, "blend.npz"), **small)


def _reference_function(path, name, ns):
    """Executes ONE top-level function of a reference script (whose module cannot be imported here: imageio / accelerate / cv2
    imports, CUDA generators) in the namespace `ns`; nothing of it is stored in this repository but its outputs."""
    import ast
    node = next(n for n in ast.parse(open(os.path.join(REF, path)).read()).body if isinstance(n, ast.FunctionDef) and n.name == name)
    exec(compile(ast.Module(body=[node], type_ignores=[]), "reference_" + name, "exec"), ns)
    return ns[name]


def eval_pattern(t, h, w):
    """Exactly reproducible clip (integer arithmetic only) shared by this generator and tests/test_eval_windows_cpu.py."""
    tt, yy, xx = np.meshgrid(np.arange(t), np.arange(h), np.arange(w), indexing="ij")
    base = ((xx * 7 + yy * 13 + tt * 29) % 101).astype(np.float32) / np.float32(101)
    third = ((xx + 2 * yy + 3 * tt) % 50).astype(np.float32) / np.float32(50)
    return np.stack([base, base * np.float32(0.5) + np.float32(0.25), third], -1)[None]


class EvalFakePipeline:
    """Stand-in for the pipeline: a deterministic function of the crop and of the call index (so every unit has its own
    disparity scale and the merge has something to align)."""
    def __init__(self):
        self.calls = 0

    def __call__(self, video, num_inference_steps, num_frames, generator, return_dict, fps):
        k, self.calls = self.calls, self.calls + 1
        v = np.ascontiguousarray(video, dtype=np.float32)
        assert v.shape[0] == num_frames and v.shape[1:3] == (480, 720) and not return_dict and fps == 12
        disp = ((np.float32(0.15) + np.float32(0.8) * v.mean(-1, dtype=np.float32)) * np.float32(1.0 + 0.07 * ((k * 5) % 7))).astype(np.float32)
        return v[None], disp[None], np.zeros((1, v.shape[0], 6, 60, 90), np.float32)


def make_eval_golden(U):
    """Evaluation-harness windows: the reference's process_with_sliding_window (video depth) on two
    procedurally generated clips, and its blend_window_outputs (relative pose, with the Kalman smoother — which needs the
    absent filterpy — replaced by the identity) on three small overlapping windows."""
    import math
    import types as _t

    import torch
    torch_shim = _t.SimpleNamespace(Generator=lambda device=None: torch.Generator())      # the reference asks for device="cuda"
    fn = _reference_function("evaluation/video_depth/launch_aether.py", "process_with_sliding_window",
                             dict(np=np, math=math, torch=torch_shim, compute_scale=U.compute_scale))
    out = {}
    for tag, (t, h, w, total) in {"wide": (25, 480, 900, 17), "tall": (17, 600, 720, 17), "plain": (33, 480, 720, 30)}.items():
        rgb, disp = fn(EvalFakePipeline(), eval_pattern(t, h, w), 4, total, 7)
        out[f"{tag}_dims"] = np.array([t, h, w, total])
        out[f"{tag}_rgb_shape"], out[f"{tag}_rgb_sum"] = np.array(rgb.shape), np.array(rgb.sum(dtype=np.float64))
        out[f"{tag}_disp_shape"], out[f"{tag}_disp_sum"] = np.array(disp.shape), np.array(np.asarray(disp, np.float64).sum())
        out[f"{tag}_disp_sub"] = np.asarray(disp, np.float64)[::3, ::7, ::11]

    blend = _reference_function("evaluation/rel_pose/launch_aether.py", "blend_window_outputs",
                                dict(np=np, torch=torch, compute_scale=U.compute_scale, align_camera_extrinsics=U.align_camera_extrinsics,
                                     apply_transformation=U.apply_transformation, interpolate_poses=U.interpolate_poses,
                                     smooth_trajectory=lambda poses, window_size=5: poses))
    rng = np.random.default_rng(11)
    N, F, starts = 19, 9, [0, 5, 10]
    tt = np.linspace(0, 1, N)
    world = np.tile(np.eye(4), (N, 1, 1))
    ang = 0.6 * tt
    world[:, 0, 0], world[:, 0, 2], world[:, 2, 0], world[:, 2, 2] = np.cos(ang), np.sin(ang), -np.sin(ang), np.cos(ang)
    world[:, 0, 3], world[:, 1, 3], world[:, 2, 3] = 0.4 * tt, 0.03 * np.sin(5 * tt), 1.1 * tt
    wins = []
    for k, s0 in enumerate(starts):
        rel = np.linalg.inv(world[s0]) @ world[s0:s0 + F]
        rel[:, :3, 3] *= (1.0, 0.8, 1.25)[k]
        rel[:, :3, 3] += 1e-3 * rng.standard_normal((F, 3))
        wins.append({"rgb": rng.random((F, 6, 8, 3)), "disparity": ((0.9, 1.3, 0.7)[k] * (0.2 + 0.6 * rng.random((F, 6, 8)))).astype(np.float32),
                     "poses": rel[:, :3, :4].copy(), "focals": 500 + 20 * rng.random(F), "range": (s0, s0 + F)})
    for k, wd in enumerate(wins):
        for key in ("rgb", "disparity", "poses", "focals"):
            out[f"pose_in_{k}_{key}"] = wd[key].copy()
        out[f"pose_in_{k}_range"] = np.array(wd["range"])
    res = blend([{k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in wd.items()} for wd in wins])
    for key in ("rgb", "disparity", "poses", "focals"):
        out[f"pose_out_{key}"] = res[key]
    out["pose_out_range"] = np.array(res["range"])
    np.savez_compressed(os.path.join(OUT, "eval_windows.npz"), **out)


if __name__ == "__main__":
    main()
