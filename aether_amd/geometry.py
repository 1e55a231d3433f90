"""Host-side geometry used when sliding windows are merged: raymap -> camera poses / field of view,
disparity -> point map, disparity scale fit, camera alignment between windows, pose interpolation and smoothing.

Restates, in plain numpy / scipy, the helpers of the reference that `blend_and_merge_window_results`
(/root/reference/scripts/demo.py:254-422) calls from /root/reference/aether/utils/postprocess_utils.py (cited per function as
U:line).  This is ≤ 41 poses and a few masked reductions per window — host work in the reference and here (float64, like the
reference); it is pinned against the reference's own functions by tests/golden/blend.npz (tools/make_golden.py imports them).
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import numpy as np


# ---- raymap <-> camera ------------------------------------------------------------------------------------------------
def signed_log1p_inverse(x: np.ndarray) -> np.ndarray:
    """U:31-46: x = sign(y) * (exp(|y|) - 1)."""
    return np.sign(x) * (np.exp(np.abs(x)) - 1)


def fov_to_focal(fovx, fovy, h: int, w: int):
    """U:97-101: mean of the horizontal and vertical focal lengths (the FoV arguments are HALF angles)."""
    return (w * 0.5 / np.tan(fovx) + h * 0.5 / np.tan(fovy)) / 2


def get_intrinsics(batch_size: int, h: int, w: int, fovx=None, fovy=None, focal=None):
    """U:147-161: pinhole matrices with the principal point at the image centre; returns (K [B,3,3], focal)."""
    if focal is None:
        focal = fov_to_focal(fovx, fovy, h, w)
    K = np.zeros((batch_size, 3, 3))
    K[:, 0, 0] = focal
    K[:, 1, 1] = focal
    K[:, 0, 2] = w * 0.5
    K[:, 1, 2] = h * 0.5
    K[:, 2, 2] = 1.0
    return K, focal


def get_rays(pose: np.ndarray, h: int, w: int, focal):
    """U:104-144: per-pixel ray origins / directions (float32, as the reference computes them in torch.float32) of cameras
    `pose` [T,4,4] (camera-to-world); pixel centres at +0.5, unit-depth directions, no normalisation."""
    T = pose.shape[0]
    K, focal = get_intrinsics(T, h, w, focal=focal)
    f = np.asarray(focal, dtype=np.float32).reshape(-1)             # a scalar focal broadcasts over the frames
    if f.shape[0] == 1 and T > 1:
        f = np.repeat(f, T)
    xs = (np.arange(w, dtype=np.float32) - np.float32(w * 0.5) + np.float32(0.5))
    ys = (np.arange(h, dtype=np.float32) - np.float32(h * 0.5) + np.float32(0.5))
    p32 = pose.astype(np.float32)
    rays_d = np.empty((T, h, w, 3), np.float32)
    for t in range(T):                                              # d = x·R[:,0] + y·R[:,1] + R[:,2], one frame at a time
        R = p32[t, :3, :3]
        np.multiply((xs / f[t])[None, :, None], R[None, None, :, 0], out=rays_d[t])
        rays_d[t] += (ys / f[t])[:, None, None] * R[None, None, :, 1]
        rays_d[t] += R[None, None, :, 2]
    rays_o = np.broadcast_to(p32[:, None, None, :3, 3], rays_d.shape).copy()
    return rays_o, rays_d, K


def signed_log1p(x: np.ndarray) -> np.ndarray:
    return np.sign(x) * np.log1p(np.abs(x))


def camera_pose_to_raymap(camera_pose: np.ndarray, intrinsic: np.ndarray, ray_o_scale_factor: float = 10.0, dmax: float = 1.0,
                          H: int = 480, W: int = 720, vae_downsample: int = 8, align_corners: bool = False) -> np.ndarray:
    """U:867-961 (the README recipe for `--raymap_action`): camera-to-world poses [N,4,4] + intrinsics [N,3,3] -> raymap
    [N,6,H/8,W/8] float32: channels 0-2 = R·((u-cu)/fu, (v-cv)/fv, 1) of the full-resolution pixel grid, bilinearly resized
    by 1/vae_downsample; channels 3-5 = signed_log1p(t · dmax · ray_o_scale_factor), constant over the frame.
    The reference resizes a full-resolution ray image; the ray direction is affine in (u, v) and bilinear resizing reproduces
    affine functions, so the same values come from evaluating it at the resize's source coordinates (i + 0.5)·s - 0.5
    (align_corners=False) or i·(n_in - 1)/(n_out - 1) (True).  Unlike the reference (float32 input only) this never writes into
    the caller's `camera_pose`."""
    pose = np.asarray(camera_pose, np.float32)
    Kf = np.asarray(intrinsic, np.float32)
    N = pose.shape[0]
    h, w = (H, W) if vae_downsample == 1 else (int(math.floor(H / vae_downsample)), int(math.floor(W / vae_downsample)))

    def source(n_out, n_in):
        i = np.arange(n_out, dtype=np.float32)
        if vae_downsample == 1:
            return i
        if align_corners:
            return i * np.float32((n_in - 1) / (n_out - 1)) if n_out > 1 else np.zeros(1, np.float32)
        return (i + np.float32(0.5)) * np.float32(vae_downsample) - np.float32(0.5)

    u, v = source(w, W), source(h, H)
    x = (u[None, :] - Kf[:, 0, 2, None]) / Kf[:, 0, 0, None]                       # [N, w]
    y = (v[None, :] - Kf[:, 1, 2, None]) / Kf[:, 1, 1, None]                       # [N, h]
    R = pose[:, :3, :3]
    d = (R[:, :, 0, None, None] * x[:, None, None, :] + R[:, :, 1, None, None] * y[:, None, :, None]
         + R[:, :, 2, None, None])                                                  # [N, 3, h, w]
    t = signed_log1p(pose[:, :3, 3] * np.float32(dmax) * np.float32(ray_o_scale_factor)).astype(np.float32)
    o = np.broadcast_to(t[:, :, None, None], (N, 3, h, w))
    return np.concatenate([d.astype(np.float32), o], axis=1)


def forward_right_raymap(frames: int = 41, height: int = 480, width: int = 720, forward: float = 0.6, right: float = 0.3, yaw_deg: float = 15.0,
                         fov_x_deg: float = 60.0) -> np.ndarray:
    """A `--raymap_action` input made the way the reference's README prescribes ("Inference with your own raymap action": camera poses in the
    first frame's camera coordinates -> `camera_pose_to_raymap`, U:867-961): eased translation `forward` along +z and `right` along +x with a
    yaw to the right, pinhole intrinsics from the horizontal field of view.  Stands in for assets/example_raymaps/raymap_forward_right.npy,
    which the reference repository keeps as a large blob.  Returns [frames, 6, height/8, width/8] float32."""
    s = np.linspace(0.0, 1.0, frames, dtype=np.float64)
    ease = s * s * (3 - 2 * s)
    yaw = np.deg2rad(yaw_deg) * ease
    pose = np.tile(np.eye(4, dtype=np.float32), (frames, 1, 1))
    pose[:, 0, 0], pose[:, 0, 2], pose[:, 2, 0], pose[:, 2, 2] = np.cos(yaw), np.sin(yaw), -np.sin(yaw), np.cos(yaw)
    pose[:, 0, 3], pose[:, 2, 3] = right * ease, forward * ease
    f = width / 2 / np.tan(np.deg2rad(fov_x_deg / 2))
    K = np.tile(np.array([[f, 0, width / 2], [0, f, height / 2], [0, 0, 1]], np.float32), (frames, 1, 1))
    return camera_pose_to_raymap(pose, K, H=height, W=width)


def raymap_to_poses(raymap: np.ndarray, camera_pose: Optional[np.ndarray] = None, ray_o_scale_inv: float = 1.0,
                    return_intrinsics: bool = True):
    """U:219-280.  raymap [T,6,h,w]: channels 0-2 ray directions, 3-5 signed-log1p ray origins.  Returns
    (camera_pose [T,4,4], fov_x [T], fov_y [T]) with HALF-angle fields of view.
    NOTE (reference behaviour, kept): the origin channels of `raymap` are decoded IN PLACE."""
    T = raymap.shape[0]
    if (not return_intrinsics) and camera_pose is not None:
        return camera_pose, None, None
    raymap[:, 3:] = signed_log1p_inverse(raymap[:, 3:])
    ray_o = np.transpose(raymap[:, 3:], (0, 2, 3, 1)) * ray_o_scale_inv
    ray_d = np.transpose(raymap[:, :3], (0, 2, 3, 1))
    origin = ray_o.reshape(T, -1, 3).mean(axis=1)
    image_centre = (ray_o + ray_d).reshape(T, -1, 3).mean(axis=1)
    z_dir = image_centre - origin
    focal = np.linalg.norm(z_dir, axis=-1)
    hh, ww = raymap.shape[-2], raymap.shape[-1]
    # image width / height in world units from the first and last column / row of ray directions (pixel centres: n-1 gaps)
    x_span = ray_d[:, :, -1:, :].reshape(T, -1, 3).mean(axis=1) - ray_d[:, :, :1, :].reshape(T, -1, 3).mean(axis=1)
    w_real = np.linalg.norm(np.cross(x_span, z_dir), axis=-1) / (ww - 1) * ww
    fov_x = np.arctan(w_real / (2 * focal))
    y_span = ray_d[:, :1, :, :].reshape(T, -1, 3).mean(axis=1) - ray_d[:, -1:, :, :].reshape(T, -1, 3).mean(axis=1)
    h_real = np.linalg.norm(np.cross(y_span, z_dir), axis=-1) / (hh - 1) * hh
    fov_y = np.arctan(h_real / (2 * focal))
    if camera_pose is None:
        x_dir = x_span.copy()
        y_dir = np.cross(z_dir, x_dir)
        x_dir = np.cross(y_dir, z_dir)
        x_dir = x_dir / np.linalg.norm(x_dir, axis=-1, keepdims=True)
        y_dir = y_dir / np.linalg.norm(y_dir, axis=-1, keepdims=True)
        z_unit = z_dir / np.linalg.norm(z_dir, axis=-1, keepdims=True)
        camera_pose = np.zeros((T, 4, 4))
        camera_pose[:, :3, 0], camera_pose[:, :3, 1], camera_pose[:, :3, 2] = x_dir, y_dir, z_unit
        camera_pose[:, :3, 3] = origin
        camera_pose[:, 3, 3] = 1.0
    return camera_pose, fov_x, fov_y


def project(depth: np.ndarray, intrinsic: np.ndarray, pose: np.ndarray) -> np.ndarray:
    """U:393-403: back-project a depth map [H,W] through K into world points [H,W,3] (pixel centres at +0.5)."""
    H, W = depth.shape
    u, v = np.meshgrid(np.arange(W), np.arange(H))
    pix = np.stack([u.reshape(-1) + 0.5, v.reshape(-1) + 0.5, np.ones(H * W)], axis=0).astype(np.float32)
    cam = (np.linalg.inv(intrinsic) @ pix) * depth.reshape(-1)
    world = pose[:3, :4] @ np.concatenate([cam, np.ones((1, cam.shape[1]))], axis=0)
    return world.T.reshape(H, W, 3)


# ---- pose smoothing -----------------------------------------------------------------------------------------------------
def detect_static_sequence(poses: np.ndarray, threshold: float = 0.01):
    """U:354-365: mean frame-to-frame translation / rotation (Frobenius) change below `threshold`."""
    dt = np.linalg.norm(np.diff(poses[:, :3, 3], axis=0), axis=1).mean()
    dr = np.linalg.norm(np.diff(poses[:, :3, :3], axis=0), axis=(1, 2)).mean()
    return bool(dt < threshold and dr < threshold), dt, dr


def _sign_consistent_quats(poses: np.ndarray) -> np.ndarray:
    from scipy.spatial.transform import Rotation as R
    q = R.from_matrix(poses[:, :3, :3]).as_quat()
    for i in range(1, len(q)):
        if np.dot(q[i], q[i - 1]) < 0:
            q[i] = -q[i]
    return q


def _poses_from(quats: np.ndarray, trans: np.ndarray) -> np.ndarray:
    from scipy.spatial.transform import Rotation as R
    out = np.tile(np.eye(4), (len(quats), 1, 1))
    out[:, :3, :3] = R.from_quat(quats).as_matrix()
    out[:, :3, 3] = trans
    return out


def smooth_poses(poses: np.ndarray, window_size: int = 5, method: str = "gaussian") -> np.ndarray:
    """U:686-748: temporal smoothing of translations and (sign-consistent) quaternions; gaussian sigma = window/6."""
    from scipy.ndimage import gaussian_filter1d
    from scipy.signal import savgol_filter
    assert window_size % 2 == 1, "window_size must be odd"
    trans, quats = poses[:, :3, 3], _sign_consistent_quats(poses)
    if method == "gaussian":
        sigma = window_size / 6.0
        st = gaussian_filter1d(trans, sigma, axis=0, mode="nearest")
        sq = gaussian_filter1d(quats, sigma, axis=0, mode="nearest")
    elif method == "savgol":
        order = min(window_size - 1, 3)
        st = savgol_filter(trans, window_size, order, axis=0, mode="nearest")
        sq = savgol_filter(quats, window_size, order, axis=0, mode="nearest")
    elif method == "ma":
        k = np.ones(window_size) / window_size
        st = np.stack([np.convolve(trans[:, i], k, mode="same") for i in range(3)], axis=1)
        sq = np.stack([np.convolve(quats[:, i], k, mode="same") for i in range(4)], axis=1)
    else:
        raise ValueError(f"unknown smoothing method {method!r}")
    sq = sq / np.linalg.norm(sq, axis=1, keepdims=True)
    return _poses_from(sq, st)


def adaptive_pose_smoothing(poses: np.ndarray, trans_diff: float, rot_diff: float, base_window: int = 5) -> np.ndarray:
    """U:368-378: (near-)static sequences get a wider gaussian window, up to 41 frames."""
    window = min(41, max(base_window, int(base_window * (0.1 / max(trans_diff + rot_diff, 1e-6)))))
    return smooth_poses(poses, window_size=window, method="gaussian")


def smooth_trajectory(poses: np.ndarray, window_size: int = 5) -> np.ndarray:
    """U:751-844: gaussian pre-smoothing, a constant-velocity Kalman filter over the translations (the reference builds it
    with filterpy.kalman.KalmanFilter(dim_x=6, dim_z=3), F = [[I, I], [0, I]], H = [I 0], Q = 0.1 I, R = 0.1 I, P0 = I; the
    predict / update equations below are filterpy's, Joseph-form covariance update included), and a locally weighted
    quaternion average (gaussian weights, sigma = window/4) for the rotations.
    Pinned by tests/golden/blend_kalman.npz and blend_fullsize.npz: the reference's own smooth_trajectory run against a stand-in for the absent
    filterpy (tools/make_blend_golden.py)."""
    from scipy.spatial.transform import Rotation as R
    N = poses.shape[0]
    F = np.eye(6)
    F[:3, 3:] = np.eye(3)
    Hm = np.hstack([np.eye(3), np.zeros((3, 3))])
    Q, Rm, P = 0.1 * np.eye(6), 0.1 * np.eye(3), np.eye(6)
    pre = smooth_poses(poses, window_size, method="gaussian")[:, :3, 3]
    x = np.zeros(6)
    x[:3] = pre[0]
    filtered = np.zeros((N, 3))
    filtered[0] = pre[0]
    I6 = np.eye(6)
    for i in range(1, N):
        x = F @ x
        P = F @ P @ F.T + Q
        y = pre[i] - Hm @ x
        S = Hm @ P @ Hm.T + Rm
        K = P @ Hm.T @ np.linalg.inv(S)
        x = x + K @ y
        IKH = I6 - K @ Hm
        P = IKH @ P @ IKH.T + K @ Rm @ K.T
        filtered[i] = x[:3]
    quats = R.from_matrix(poses[:, :3, :3]).as_quat()
    half = window_size // 2
    sq = np.zeros_like(quats)
    for i in range(N):
        lo, hi = max(0, i - half), min(N, i + half + 1)
        wts = np.exp(-0.5 * ((np.arange(lo, hi) - i) / (half / 2)) ** 2)
        wts = wts / wts.sum()
        acc = np.zeros(4)
        for j, wt in zip(range(lo, hi), wts):
            acc += wt * (-quats[j] if np.dot(quats[j], quats[i]) < 0 else quats[j])
        sq[i] = acc / np.linalg.norm(acc)
    return _poses_from(sq, filtered)


# ---- disparity -> point map ---------------------------------------------------------------------------------------------
def postprocess_pointmap(disparity: np.ndarray, raymap: np.ndarray, vae_downsample_scale: int = 8,
                         camera_pose: Optional[np.ndarray] = None, focal=None, ray_o_scale_inv: float = 1.0,
                         smooth_camera: bool = False, smooth_method: str = "simple", with_pointmap: bool = True) -> dict:
    """U:283-351.  disparity [T,H,W] in [0,1], raymap [T,6,H/8,W/8] -> pointmap = depth * ray_d + ray_o (world space),
    depth = 1 / clip(disparity, 1e-3, 1).  `raymap` is decoded in place by raymap_to_poses (see there).
    `with_pointmap=False` returns only camera_pose / intrinsics (callers that need just the cameras skip the per-pixel work)."""
    camera_pose, fov_x, fov_y = raymap_to_poses(raymap, camera_pose=camera_pose, ray_o_scale_inv=ray_o_scale_inv,
                                                return_intrinsics=(focal is not None))
    H, W = int(raymap.shape[2] * vae_downsample_scale), int(raymap.shape[3] * vae_downsample_scale)
    if focal is None:
        focal = fov_to_focal(fov_x, fov_y, H, W)
    if smooth_camera:
        static, dt, dr = detect_static_sequence(camera_pose)
        if static:
            camera_pose = adaptive_pose_smoothing(camera_pose, dt, dr)
        elif smooth_method == "simple":
            camera_pose = smooth_poses(camera_pose, window_size=5, method="gaussian")
        elif smooth_method == "kalman":
            camera_pose = smooth_trajectory(camera_pose, window_size=5)
    if not with_pointmap:
        return {"camera_pose": camera_pose, "intrinsics": get_intrinsics(camera_pose.shape[0], H, W, focal=focal)[0]}
    depth = np.clip(1.0 / np.clip(disparity, 1e-3, 1), 0, 1e8)
    ray_o, ray_d, K = get_rays(camera_pose, H, W, focal)
    return {"pointmap": depth[..., None] * ray_d + ray_o, "camera_pose": camera_pose, "intrinsics": K, "ray_o": ray_o,
            "ray_d": ray_d, "depth": depth}


# ---- window-to-window alignment -------------------------------------------------------------------------------------------
def compute_scale(prediction: np.ndarray, target: np.ndarray, mask: np.ndarray) -> float:
    """U:847-864: least-squares scale s minimising |mask (s·prediction − target)|² (float32 accumulation like the reference;
    0 when the masked prediction is all zero)."""
    p = np.asarray(prediction, np.float32)
    t = np.asarray(target, np.float32)
    m = np.asarray(mask, bool).astype(np.float32)
    num = float(np.sum(m * p * t, dtype=np.float32))
    den = float(np.sum(m * p * p, dtype=np.float32))
    return num / den if den != 0 else 0.0


def align_camera_extrinsics(cameras_src: np.ndarray, cameras_tgt: np.ndarray, estimate_scale: bool = True, eps: float = 1e-9):
    """U:516-568: similarity (R, T, s) that maps the source cameras [B,·,4] ([R|t] rows 0-2 are used) onto the targets:
    rotation from the SVD of the mean relative rotation, scale from the covariance of the projected translations."""
    R_src, R_tgt = cameras_src[:, :3, :3], cameras_tgt[:, :3, :3]
    rr = np.einsum("bji,bjk->bik", R_tgt, R_src).mean(axis=0)            # mean of R_tgtᵀ R_src
    U, _, Vh = np.linalg.svd(rr)
    align_R = Vh.T @ U.T
    T_src, T_tgt = cameras_src[:, :3, 3], cameras_tgt[:, :3, 3]
    A = np.einsum("bj,bjk->bk", T_src, R_src)
    B = np.einsum("bj,bjk->bk", T_tgt, R_src)
    A_mu, B_mu = A.mean(axis=0, keepdims=True), B.mean(axis=0, keepdims=True)
    if estimate_scale and A.shape[0] > 1:
        Ac, Bc = A - A_mu, B - B_mu
        s = float((Ac * Bc).mean() / max((Ac ** 2).mean(), eps))
    else:
        s = 1.0
    return align_R[None], B_mu - s * A_mu, s


def apply_transformation(cameras_src: np.ndarray, align_R: np.ndarray, align_T: np.ndarray, align_s: float) -> np.ndarray:
    """U:571-607 with return_extri=True: aligned [R|t] ([B,·,4], as many rows as the input had) of the source cameras."""
    R_src, T_src = cameras_src[:, :, :3], cameras_src[:, :, 3]
    aligned_R = R_src @ align_R[0]
    aligned_T = np.einsum("bij,j->bi", R_src, align_T[0]) + T_src * align_s
    return np.concatenate([aligned_R, aligned_T[..., None]], axis=-1)


def slerp(q1: np.ndarray, q2: np.ndarray, t: float) -> np.ndarray:
    """U:610-647: shortest-path spherical interpolation; linear + renormalise when the quaternions are within ~1.8 degrees."""
    dot = float(np.sum(q1 * q2))
    if dot < 0.0:
        q2, dot = -q2, -dot
    if dot > 0.9995:
        r = q1 + t * (q2 - q1)
        return r / np.linalg.norm(r)
    theta0 = np.arccos(dot)
    theta = theta0 * t
    s1 = np.sin(theta) / np.sin(theta0)
    return (np.cos(theta) - dot * s1) * q1 + s1 * q2


def interpolate_poses(pose1: np.ndarray, pose2: np.ndarray, weight: float) -> np.ndarray:
    """U:650-683: `weight` is pose1's share: rotation slerp(q1, q2, 1 - weight), translation linear."""
    from scipy.spatial.transform import Rotation as R
    q = slerp(R.from_matrix(pose1[:3, :3]).as_quat(), R.from_matrix(pose2[:3, :3]).as_quat(), 1 - weight)
    out = np.eye(4)
    out[:3, :3] = R.from_quat(q).as_matrix()
    out[:3, 3] = weight * pose1[:3, 3] + (1 - weight) * pose2[:3, 3]
    return out


def focals_from_fov(n: int, h: int, w: int, fov_x, fov_y) -> np.ndarray:
    """(K[0,0] + K[1,1]) / 2 of get_intrinsics, as D:374-383 computes the per-frame focal of a window."""
    K, _ = get_intrinsics(n, h, w, fovx=fov_x, fovy=fov_y)
    return (K[:, 0, 0] + K[:, 1, 1]) / 2


def pose_rows(poses: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Convenience: rotations [B,3,3] and translations [B,3] of camera-to-world matrices."""
    return poses[:, :3, :3], poses[:, :3, 3]
