"""`aether.utils.postprocess_utils` — the names reference user code imports (scripts/demo.py:25-35, demo_gradio.py,
evaluation/*/launch_aether.py) on top of aether_amd.geometry / aether_amd.export, with the reference's calling conventions:
the camera-alignment helpers take and return torch tensors there (U:516-607), everything else numpy.
Scope: the helpers on the sliding-window / evaluation-window path.  The reference's file-export and mesh
utilities (depth_to_disparity, get_raymap_from_camera_parameters, save_ply, save_pointmap, depth_edge, align_rigid, get_pixel;
aether.utils.visualize_utils.predictions_to_glb) are out of scope and raise a clear error when asked for."""
import numpy as np
import torch

from aether_amd import geometry as _G
from aether_amd.export import colorize_depth  # noqa: F401
from aether_amd.geometry import (  # noqa: F401
    adaptive_pose_smoothing,
    camera_pose_to_raymap,
    compute_scale as _compute_scale,
    detect_static_sequence,
    fov_to_focal,
    get_intrinsics,
    get_rays,
    interpolate_poses,
    postprocess_pointmap,
    project,
    raymap_to_poses,
    signed_log1p,
    signed_log1p_inverse,
    slerp,
    smooth_poses,
    smooth_trajectory,
)


def _np(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def compute_scale(prediction, target, mask):
    """U:847-864 (numpy arrays, torch tensors or a scalar mask)."""
    return _compute_scale(_np(prediction), _np(target), _np(mask))


def align_camera_extrinsics(cameras_src, cameras_tgt, estimate_scale: bool = True, eps: float = 1e-9):
    """U:516-568: (align_t_R [1,3,3], align_t_T [1,3], align_t_s) as torch tensors / float like the reference."""
    R, T, s = _G.align_camera_extrinsics(_np(cameras_src), _np(cameras_tgt), estimate_scale, eps)
    return torch.from_numpy(np.ascontiguousarray(R)), torch.from_numpy(np.ascontiguousarray(T)), s


def apply_transformation(cameras_src, align_t_R, align_t_T, align_t_s, return_extri: bool = True):
    """U:571-607."""
    out = torch.from_numpy(_G.apply_transformation(_np(cameras_src), _np(align_t_R), _np(align_t_T), float(align_t_s)))
    return out if return_extri else (out[..., :3], out[..., 3])


_OUT_OF_SCOPE = ("depth_to_disparity", "get_raymap_from_camera_parameters", "save_ply", "save_pointmap", "depth_edge", "align_rigid", "get_pixel")


def __getattr__(name):
    if name in _OUT_OF_SCOPE:
        raise AttributeError(f"aether.utils.postprocess_utils.{name} is a file-export / mesh helper of the reference that the MI355X hot-path "
                             "build does not provide (SURVEY.md §2, out of scope); use the reference's own module for it")
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
