"""Spatio-temporal sliding windows of the reference's evaluation harnesses, as independent units.

Two drivers, with the reference's window geometry and merge arithmetic:

* video depth (/root/reference/evaluation/video_depth/launch_aether.py:81-287, `process_with_sliding_window`): windows of
  up to 41 frames every 8 frames; frames larger than 480x720 are covered by 480x720 crops along ONE axis that overlap by at
  least 60 rows / 90 columns; disparities are merged crop by crop (least-squares scale against what is already merged, linear
  cross-fade over the overlap), then window by window in time the same way.  Quirk kept: the returned rgb is the FIRST
  unit's rgb only (the reference never merges colour here).
* relative pose (/root/reference/evaluation/rel_pose/launch_aether.py:124-250, `process_video_with_sliding_window` and
  `blend_window_outputs`): windows of up to 41 frames every 32 frames; per window the raymap is decoded to smoothed camera
  poses; windows are chained by a disparity scale, a similarity alignment of the cameras over the overlap, pose interpolation
  across it and linear cross-fades of rgb / disparity / focal, then the whole trajectory is smoothed.

Every unit is a full, independent pipeline call with a fresh generator of the same seed, so units shard over ranks exactly
like the demo's windows (`windows.run_windows`: unit i -> rank i mod N, one all_gather of the finished outputs, merge on
rank 0; other ranks return None).  The merges are sequential host numpy, as in the reference.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import geometry as G
from .windows import run_windows

TARGET_H, TARGET_W = 480, 720


def max_window_frames(total_frames: int, longest: int = 41) -> int:
    """Largest of 41, 33, 25, 17, ... that fits the clip (the pipeline accepts 17/25/33/41 frames)."""
    n = longest
    while n > total_frames:
        n -= 8
    return n


# ---- video depth ------------------------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class DepthUnit:
    t0: int
    t1: int
    h0: int
    h1: int
    w0: int
    w1: int


@dataclass
class DepthPlan:
    window_frames: int
    horizontal: bool                       # crops step along the width (else along the height)
    times: List[Tuple[int, int]]           # (t0, t1) of every temporal window
    crops: List[Tuple[int, int]]           # (start, end) of every crop along the tiled axis
    units: List[DepthUnit]                 # time-major: units[k * len(crops) + i]


def plan_depth_windows(t: int, h: int, w: int, total_frames: int, spatial_overlap: Tuple[int, int] = (60, 90),
                       temporal_stride: int = 8) -> DepthPlan:
    """EV:87-150.  `t, h, w` are the clip's dimensions, `total_frames` bounds the window length.
    Reference behaviour kept: the crop stride is `(extent - target) // (n - 1)`, so with three or more crops up to n - 2
    trailing rows / columns are not covered and the merged disparity is that much smaller than the clip."""
    nf = max_window_frames(total_frames)
    n_h = 1 if h <= TARGET_H else math.ceil((h - TARGET_H) / (TARGET_H - spatial_overlap[0])) + 1
    n_w = 1 if w <= TARGET_W else math.ceil((w - TARGET_W) / (TARGET_W - spatial_overlap[1])) + 1
    if n_h != 1 and n_w != 1:
        raise AssertionError((n_h, n_w))      # the reference tiles one axis only
    horizontal = n_w > 1
    n, extent, target = (n_w, w, TARGET_W) if horizontal else (n_h, h, TARGET_H)
    stride = (extent - target) // (n - 1) if n > 1 else 0
    crops = []
    for i in range(n):
        a = int(i * stride)
        b = a + target
        if b > extent:
            a, b = extent - target, extent
        crops.append((a, b))
    starts = list(range(0, t - nf, temporal_stride)) + [t - nf]
    times = [(s, min(s + nf, t)) for s in starts]
    units = [DepthUnit(t0, t1, 0, TARGET_H, a, b) if horizontal else DepthUnit(t0, t1, a, b, 0, TARGET_W)
             for (t0, t1) in times for (a, b) in crops]
    return DepthPlan(nf, horizontal, times, crops, units)


def _chain(pieces: Sequence[np.ndarray], ranges: Sequence[Tuple[int, int]], axis: int) -> np.ndarray:
    """The reference's pairwise merge along `axis` (EV:175-250 for crops, EV:258-283 for time): piece k is scaled onto what is
    merged so far by a least-squares fit over their overlap, then cross-faded over it with linspace(1, 0, overlap)."""
    def sl(lo, hi):
        idx = [slice(None)] * 3
        idx[axis] = slice(lo, hi)
        return tuple(idx)

    merged = pieces[0]
    width = pieces[0].shape[-1]
    for k in range(1, len(pieces)):
        (a, b), prev_end = ranges[k], ranges[k - 1][1]
        overlap = prev_end - a
        head, tail = pieces[k][sl(None, overlap)], merged[sl(merged.shape[axis] - overlap, None)]
        # operands flattened the way the reference reshapes them (the fit is a plain sum, so only the element set matters)
        shape = (1, -1, overlap) if axis == 2 else ((1, overlap, -1) if axis == 1 else (1, -1, width))
        scale = G.compute_scale(head.reshape(shape), tail.reshape(shape), np.ones_like(tail).reshape(shape))
        aligned = scale * pieces[k]
        out_shape = list(merged.shape)
        out_shape[axis] = b
        out = np.ones(out_shape)
        wshape = [1, 1, 1]
        wshape[axis] = overlap
        weight = np.linspace(1, 0, overlap).reshape(wshape)
        out[sl(None, a)] = merged[sl(None, a)]
        out[sl(prev_end, None)] = aligned[sl(prev_end - a, None)]
        out[sl(a, prev_end)] = merged[sl(a, prev_end)] * weight + aligned[sl(None, overlap)] * (1 - weight)
        merged = out
    return merged


def merge_depth_windows(plan: DepthPlan, rgb: Sequence[np.ndarray], disparity: Sequence[np.ndarray]) -> Tuple[np.ndarray, np.ndarray]:
    """rgb / disparity of every unit, in `plan.units` order -> (rgb of the first unit, merged disparity [t, h, w])."""
    nc = len(plan.crops)
    per_time = [_chain(disparity[k * nc:(k + 1) * nc], plan.crops, 2 if plan.horizontal else 1) for k in range(len(plan.times))]
    return rgb[0], _chain(per_time, plan.times, 0)


def process_with_sliding_window(pipeline, obs_image: np.ndarray, num_inference_step: int, total_frames: int, seed: int,
                                device: Optional[torch.device] = None):
    """Signature and results of the reference's function (EV:81-287).  obs_image [1, t, h, w, 3] float in [0, 1].
    Returns (rgb [frames, 480, 720, 3] of the first unit, disparity [t, h, w]); None on ranks other than 0 when a process
    group is initialised (units are sharded over the ranks)."""
    b, t, h, w, _ = obs_image.shape
    assert b == 1, "Only batch size 1 is supported"
    plan = plan_depth_windows(t, h, w, total_frames)
    dev = device if device is not None else getattr(pipeline, "_execution_device", torch.device("cuda"))

    def call(i: int):
        u = plan.units[i]
        rgb, disp, ray = pipeline(video=obs_image[0, u.t0:u.t1, u.h0:u.h1, u.w0:u.w1, :], num_inference_steps=num_inference_step,
                                  num_frames=u.t1 - u.t0, generator=torch.Generator(device=dev).manual_seed(seed), return_dict=False,
                                  fps=12)
        return SimpleNamespace(rgb=rgb[0], disparity=disp[0], raymap=ray[0])

    results = run_windows(call, list(range(len(plan.units))))
    if results is None:
        return None
    return merge_depth_windows(plan, [r.rgb for r in results], [r.disparity for r in results])


# ---- relative pose ----------------------------------------------------------------------------------------------------------
def pose_window_starts(t: int, temporal_stride: int = 32) -> Tuple[List[int], int]:
    """EP:128-138: (starts, frames per window)."""
    nf = max_window_frames(t)
    starts = list(range(0, t - nf, temporal_stride))
    if not starts or starts[-1] != t - nf:
        starts.append(t - nf)
    return starts, nf


def blend_window_outputs(window_outputs: List[Dict], smooth: Optional[Callable[[np.ndarray], np.ndarray]] = None) -> Dict:
    """EP:173-250.  Each entry: rgb [f,H,W,3], disparity [f,H,W], poses [f,3,4], focals [f], range (t0, t1).  Like the
    reference this consumes its input (the first entry becomes the result, disparities are rescaled in place).
    `smooth` post-processes the [n,4,4] trajectory (default: the Kalman smoother of geometry.smooth_trajectory, window 5)."""
    final = window_outputs[0]
    for i in range(1, len(window_outputs)):
        prev, curr = final, window_outputs[i]
        t0_curr = curr["range"][0]
        overlap = prev["range"][1] - t0_curr
        width = curr["disparity"].shape[-1]
        scale = G.compute_scale(curr["disparity"][:overlap].reshape(1, -1, width), prev["disparity"][-overlap:].reshape(1, -1, width), 1.0)
        curr["disparity"] *= scale
        rel_r, rel_t, rel_s = G.align_camera_extrinsics(curr["poses"][:overlap], prev["poses"][-overlap:])
        aligned = G.apply_transformation(curr["poses"], rel_r, rel_t, rel_s)
        weights = np.linspace(1, 0, overlap)
        # NB the index t0_curr + k addresses the MERGED trajectory, whose first frame is frame 0 of the video
        blended_poses = np.array([G.interpolate_poses(prev["poses"][t0_curr + k], aligned[k], wk)[:3, :4] for k, wk in enumerate(weights)])
        for key in ("rgb", "disparity", "poses", "focals"):
            keep = prev[key].shape[0] - overlap
            if key == "poses":
                mid, new = blended_poses, aligned[overlap:]
            else:
                wgt = weights.reshape((overlap,) + (1,) * (prev[key].ndim - 1))
                mid, new = prev[key][-overlap:] * wgt + curr[key][:overlap] * (1 - wgt), curr[key][overlap:]
            final[key] = np.concatenate((prev[key][:keep], mid, new), axis=0)
        final["range"] = (prev["range"][0], curr["range"][-1])
    n = final["poses"].shape[0]
    poses = np.concatenate([final["poses"], np.zeros((n, 1, 4))], axis=1)
    poses[:, -1, 3] = 1.0
    final["poses"] = (smooth if smooth is not None else (lambda p: G.smooth_trajectory(p, window_size=5)))(poses)
    return final


def process_video_with_sliding_window(pipeline, video_frames: np.ndarray, num_inference_steps: int, seed: int,
                                      device: Optional[torch.device] = None, smooth=None):
    """EP:124-170.  video_frames [1, t, 480, 720, 3].  Returns the merged dict of blend_window_outputs (None on ranks > 0)."""
    t = video_frames.shape[1]
    starts, nf = pose_window_starts(t)
    dev = device if device is not None else getattr(pipeline, "_execution_device", torch.device("cuda"))

    def call(s: int):
        rgb, disp, ray = pipeline(video=video_frames[0, s:s + nf], num_inference_steps=num_inference_steps, num_frames=nf,
                                  generator=torch.Generator(device=dev).manual_seed(seed), return_dict=False, fps=12)
        return SimpleNamespace(rgb=rgb[0], disparity=disp[0], raymap=ray[0])

    results = run_windows(call, starts)
    if results is None:
        return None
    outputs = []
    for r in results:
        pcd = G.postprocess_pointmap(r.disparity, r.raymap, smooth_camera=True, smooth_method="kalman")
        K = pcd["intrinsics"]
        outputs.append({"rgb": r.rgb, "disparity": r.disparity, "poses": pcd["camera_pose"][:, :3, :4],
                        "focals": (K[:, 0, 0] + K[:, 1, 1]) / 2, "range": (r.start, r.start + nf)})
    return blend_window_outputs(outputs, smooth)
