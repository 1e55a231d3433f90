"""Window merge against the REFERENCE's own outputs: tests/golden/blend.npz holds three overlapping
windows and what /root/reference's blend_and_merge_window_results (scripts/demo.py:254-422) + aether/utils/postprocess_utils.py
produced for them (tools/make_golden.py, run in the build container where the reference is importable).  This row of the
hot-path table IS pinned by the reference; tolerance 1e-5 relative (fixtures stored as float32, float32 reductions inside
compute_scale / get_rays differ in summation order between numpy and torch)."""
import os

import numpy as np
import pytest

from aether_amd import geometry as G
from aether_amd.windows import WindowResult, blend_and_merge_window_results

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "blend.npz"))
# the default `--smooth_method kalman` (D:173-179) of the reference, run against a stand-in for the absent filterpy (tools/make_blend_golden.py)
KALMAN = np.load(os.path.join(os.path.dirname(__file__), "golden", "blend_kalman.npz"))
H, W = (int(v) for v in GOLD["hw"])


def _close(a, b, what, rtol=1e-5):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)
    assert err < rtol, f"{what}: max error {err:.3e} of the tensor's scale"


def _windows():
    return [WindowResult(int(s), GOLD[f"rgb_{k}"].astype(np.float32), GOLD[f"disparity_{k}"].copy(), GOLD[f"raymap_{k}"].copy())
            for k, s in enumerate(GOLD["starts"])]


@pytest.mark.parametrize("tag,kw", [("plain", dict(align_pointmaps=False, smooth_camera=False)),
                                    ("aligned", dict(align_pointmaps=True, smooth_camera=False)),
                                    ("smooth", dict(align_pointmaps=False, smooth_camera=True, smooth_method="simple")),
                                    ("kalman", dict(align_pointmaps=False, smooth_camera=True, smooth_method="kalman")),
                                    ("kalman_aligned", dict(align_pointmaps=True, smooth_camera=True, smooth_method="kalman"))])
@pytest.mark.parametrize("device", [None, "cpu"])     # None: host numpy; "cpu": the torch path that runs on the GPU in scripts/demo.py
def test_blend_matches_reference(tag, kw, device):
    gold = KALMAN if tag.startswith("kalman") else GOLD
    rgb, disp, poses, pm = blend_and_merge_window_results(_windows(), height=H, width=W, device=device, **kw)
    assert all(isinstance(a, np.ndarray) and a.dtype == np.float64 for a in (rgb, disp, poses, pm))
    assert rgb.shape == (19, H, W, 3) and disp.shape == (19, H, W) and poses.shape == (19, 4, 4) and pm.shape == (19, H, W, 3)
    _close(rgb, GOLD["plain_rgb"], f"{tag} rgb")
    _close(disp, gold[f"{tag}_disparity"], f"{tag} disparity")
    _close(poses, gold[f"{tag}_poses"], f"{tag} poses")
    _close(pm, gold[f"{tag}_pointmaps"], f"{tag} pointmaps")


def test_kalman_smoothing_matches_reference():
    """`smooth_trajectory` (U:751-844) as the reference computes it (its own source, with the published predict / update equations of
    filterpy.kalman.KalmanFilter standing in for the absent package): a window's decoded cameras, and a noisy walk with large rotations at
    window sizes 5 and 9."""
    _close(G.smooth_trajectory(KALMAN["unit_in_a"].copy(), 5), KALMAN["unit_out_a"], "window cameras", 1e-10)
    _close(G.smooth_trajectory(KALMAN["unit_in_b"].copy(), 5), KALMAN["unit_out_b"], "walk, window 5", 1e-10)
    _close(G.smooth_trajectory(KALMAN["unit_in_b"].copy(), 9), KALMAN["unit_out_b_w9"], "walk, window 9", 1e-10)


def test_rgb_only_blend_agrees():
    from aether_amd.windows import blend_rgb
    _close(blend_rgb(_windows(), 19), GOLD["plain_rgb"], "blend_rgb", 1e-6)


def test_blend_keeps_the_reference_quirks():
    wins = _windows()
    before = wins[1].raymap.copy()
    _, _, poses, _ = blend_and_merge_window_results(wins, height=H, width=W, align_pointmaps=False, smooth_camera=False)
    assert not np.array_equal(wins[1].raymap[:, 3:], before[:, 3:])           # origins decoded in place (U:226)
    assert np.array_equal(wins[1].raymap[:, :3], before[:, :3])
    # outside the overlaps the aligned poses of windows >= 1 carry the similarity's scale in [3,3] (U:597-603 on 4x4 inputs)
    assert abs(poses[0, 3, 3] - 1.0) < 1e-12 and abs(poses[-1, 3, 3] - 1.0) > 1e-3


def test_geometry_units_match_reference():
    d1, r1, r0 = GOLD["disparity_1"].copy(), GOLD["raymap_1"].copy(), GOLD["raymap_0"].copy()
    pm = G.postprocess_pointmap(d1, r1.copy(), vae_downsample_scale=8, ray_o_scale_inv=0.1)
    _close(pm["pointmap"], GOLD["unit_pointmap"], "pointmap")
    _close(pm["camera_pose"], GOLD["unit_pose"], "camera_pose", 1e-9)
    _close(pm["intrinsics"], GOLD["unit_K"], "intrinsics", 1e-9)
    p_a, _, _ = G.raymap_to_poses(r1.copy(), ray_o_scale_inv=0.1)
    p_b, _, _ = G.raymap_to_poses(r0.copy(), ray_o_scale_inv=0.1)
    aR, aT, aS = G.align_camera_extrinsics(p_a[:4], p_b[-4:])
    _close(aR, GOLD["unit_align_R"], "align R", 1e-9)
    _close(aT, GOLD["unit_align_T"], "align T", 1e-9)
    assert abs(aS - float(GOLD["unit_align_s"])) < 1e-9 * abs(aS)
    _close(G.apply_transformation(p_a, aR, aT, aS), GOLD["unit_applied"], "apply_transformation", 1e-9)
    _close(np.stack([G.interpolate_poses(p_a[0], p_b[3], w) for w in (0.0, 0.3, 1.0)]), GOLD["unit_interp"], "interpolate", 1e-9)
    _close(G.smooth_poses(p_a.copy(), 5, "gaussian"), GOLD["unit_smooth_gauss"], "smooth gaussian", 1e-9)
    _close(G.smooth_poses(p_a.copy(), 5, "savgol"), GOLD["unit_smooth_savgol"], "smooth savgol", 1e-9)
    s = G.compute_scale(d1[:4].reshape(1, -1, W), GOLD["disparity_0"][-4:].reshape(1, -1, W), d1[:4].reshape(1, -1, W) > 0.1)
    assert abs(s - float(GOLD["unit_scale"])) < 1e-5 * abs(s)
    K = np.array([[28.0, 0, W / 2], [0, 28.0, H / 2], [0, 0, 1.0]])
    _close(G.project(1 / np.clip(GOLD["disparity_0"][2].astype(np.float64), 1e-8, 1e8), K, p_b[2]), GOLD["unit_project"], "project")


def test_kalman_branch_runs_and_is_smooth():
    """Properties of the Kalman branch (U:751-844; its values are pinned by test_kalman_smoothing_matches_reference): keeps valid rotations and
    does not move a smooth trajectory far."""
    p, _, _ = G.raymap_to_poses(GOLD["raymap_1"].copy(), ray_o_scale_inv=0.1)
    s = G.smooth_trajectory(p.copy(), 5)
    assert s.shape == p.shape and np.isfinite(s).all()
    assert np.allclose(np.einsum("nij,nkj->nik", s[:, :3, :3], s[:, :3, :3]), np.eye(3), atol=1e-9)
    assert np.abs(s[:, :3, 3] - p[:, :3, 3]).max() < 0.5 * np.abs(np.diff(p[:, :3, 3], axis=0)).max() * len(p)


def test_camera_pose_to_raymap_matches_reference_and_round_trips():
    """The README recipe for `--raymap_action` (U:919-961): encoder against the reference's output, decoder against its
    raymap_to_poses on that raymap, and the caller's poses left untouched."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "raymap.npz"))
    n = len(z["poses"])
    poses32 = z["poses"].astype(np.float32)
    keep = poses32.copy()
    ray = G.camera_pose_to_raymap(poses32, np.tile(z["K"], (n, 1, 1)))
    assert ray.shape == (n, 6, 60, 90) and ray.dtype == np.float32 and np.array_equal(poses32, keep)
    np.testing.assert_allclose(ray, z["raymap"], atol=1e-6, rtol=0)
    rec, fov_x, fov_y = G.raymap_to_poses(ray.copy(), ray_o_scale_inv=0.1)
    np.testing.assert_allclose(rec, z["rec_poses"], atol=2e-6)
    np.testing.assert_allclose(fov_x, z["fov_x"], rtol=1e-5)
    np.testing.assert_allclose(fov_y, z["fov_y"], rtol=1e-5)
    np.testing.assert_allclose(rec[:, :3, 3], z["poses"][:, :3, 3], atol=2e-3)       # the trajectory survives encode -> decode
    # full resolution / align_corners variants stay consistent with the 1/8 grid (affine in the pixel coordinates)
    full = G.camera_pose_to_raymap(poses32, np.tile(z["K"], (n, 1, 1)), vae_downsample=1)
    assert full.shape == (n, 6, 480, 720)
    np.testing.assert_allclose(0.5 * (full[:, :3, 3::8, 3::8] + full[:, :3, 4::8, 4::8]), ray[:, :3], atol=1e-6)


@pytest.mark.gpu
def test_blend_on_the_gpu_matches_reference():
    """The device merge on the MI355X (float64 torch kernels) against the reference's outputs."""
    import torch
    for tag, kw in (("plain", dict(smooth_camera=False)), ("smooth", dict(smooth_camera=True, smooth_method="simple")),
                    ("kalman", dict(smooth_camera=True, smooth_method="kalman"))):
        gold = KALMAN if tag == "kalman" else GOLD
        rgb, disp, poses, pm = blend_and_merge_window_results(_windows(), height=H, width=W, device=torch.device("cuda:0"), **kw)
        _close(rgb, GOLD["plain_rgb"], f"{tag} rgb")
        _close(disp, gold[f"{tag}_disparity"], f"{tag} disparity")
        _close(poses, gold[f"{tag}_poses"], f"{tag} poses")
        _close(pm, gold[f"{tag}_pointmaps"], f"{tag} pointmaps")


def test_device_merge_accepts_gathered_tensors():
    """run_windows(keep_on_device=True) hands rgb / disparity over as torch tensors: same merged values."""
    import torch
    wins = [WindowResult(w.start, torch.from_numpy(w.rgb), torch.from_numpy(w.disparity), w.raymap) for w in _windows()]
    got = blend_and_merge_window_results(wins, height=H, width=W, smooth_camera=False, device="cpu")
    ref = blend_and_merge_window_results(_windows(), height=H, width=W, smooth_camera=False, device="cpu")
    assert all(np.array_equal(a, b) for a, b in zip(got, ref))


def test_reference_import_paths_and_calling_conventions():
    """`from aether.utils.postprocess_utils import ...` (scripts/demo.py:25-35, evaluation/*/launch_aether.py) resolves, the
    camera-alignment helpers speak torch like the reference's, and give the reference's values."""
    import torch
    from aether.utils.postprocess_utils import (align_camera_extrinsics, apply_transformation, camera_pose_to_raymap, colorize_depth,  # noqa: F401
                                                compute_scale, get_intrinsics, interpolate_poses, postprocess_pointmap, project,
                                                raymap_to_poses, smooth_trajectory)
    from aether.utils.preprocess_utils import imcrop_center
    p_a, _, _ = raymap_to_poses(GOLD["raymap_1"].copy(), ray_o_scale_inv=0.1)
    p_b, _, _ = raymap_to_poses(GOLD["raymap_0"].copy(), ray_o_scale_inv=0.1)
    R, T, s = align_camera_extrinsics(torch.from_numpy(p_a[:4]), torch.from_numpy(p_b[-4:]))
    assert isinstance(R, torch.Tensor) and R.shape == (1, 3, 3) and T.shape == (1, 3)
    _close(R.numpy(), GOLD["unit_align_R"], "R", 1e-9)
    _close(T.numpy(), GOLD["unit_align_T"], "T", 1e-9)
    out = apply_transformation(torch.from_numpy(p_a), R, T, s, return_extri=True)
    assert isinstance(out, torch.Tensor)
    _close(out.numpy(), GOLD["unit_applied"], "applied", 1e-9)
    aR, aT = apply_transformation(torch.from_numpy(p_a), R, T, s, return_extri=False)
    assert aR.shape == (p_a.shape[0], 4, 3) and aT.shape == (p_a.shape[0], 4)
    d1, d0 = GOLD["disparity_1"][:4].reshape(1, -1, W), GOLD["disparity_0"][-4:].reshape(1, -1, W)
    assert abs(compute_scale(torch.from_numpy(d1), torch.from_numpy(d0), torch.from_numpy(d1 > 0.1)) - float(GOLD["unit_scale"])) < 1e-5
    img = np.random.default_rng(0).random((50, 72, 3), dtype=np.float32)
    assert imcrop_center([img], 480, 720)[0].shape == (48, 72, 3)


def test_deferred_camera_algebra_matches_immediate():
    """WindowMerger fetches a DEVICE raymap asynchronously and runs the window's camera algebra when its copy has landed — possibly several `add` calls
    later (run_windows_merged: rank 0 must not wait on the host for a merge).  The queue is exercised here without a GPU: events that report "not yet" for a
    while, windows added meanwhile; poses, focal lengths and every merged array must equal the immediate form's."""
    from aether_amd.windows import WindowMerger

    class LateEvent:                                       # a copy that lands after `n` polls
        def __init__(self, n):
            self.n = n

        def query(self):
            self.n -= 1
            return self.n < 0

        def synchronize(self):
            self.n = -1

    def run(deferred):
        wins = _windows()
        m = WindowMerger(total_frames=19, window_frames=wins[0].rgb.shape[0], frame_hw=(H, W), height=H, width=W, device="cpu", smooth_camera=True,
                         smooth_method="simple")
        if deferred:
            real = m._queue_cameras

            def queue(raymap, t0, ov, first):              # what the CUDA branch appends: (host buffer, event, ...); here the "copy" lands 3 polls late
                m._pending.append((__import__("torch").from_numpy(raymap.astype(np.float32)), LateEvent(3), t0, ov, first))
            m._queue_cameras = queue
            del real
        for w in wins:
            m.add(w)
        if deferred:
            assert len(m._pending) == len(wins), "nothing may have been drained while the copies were in flight"
        return m.finish()

    for a, b, what in zip(run(False), run(True), ("rgb", "disparity", "poses", "pointmaps")):
        assert np.array_equal(a, b), what


def test_kalman_translations_match_an_independent_per_axis_filter():
    """Independent of the stand-in filter behind blend_kalman.npz: F, H, Q, R and P0 of the reference's filter
    are isotropic, so the 6-state filter decouples into three independent (position, velocity) filters.  A separately written 2-state scalar recursion
    (textbook covariance update P = (I - K H) P instead of the Joseph form) fed with the same gaussian-pre-smoothed translations must reproduce
    smooth_trajectory's translations."""
    rng = np.random.default_rng(5)
    n = 23
    poses = np.tile(np.eye(4), (n, 1, 1))
    poses[:, :3, 3] = np.cumsum(rng.normal(0.02, 0.05, (n, 3)), axis=0)
    got = G.smooth_trajectory(poses.copy(), 5)[:, :3, 3]
    pre = G.smooth_poses(poses.copy(), 5, method="gaussian")[:, :3, 3]
    want = np.zeros_like(pre)
    for ax in range(3):
        z = pre[:, ax]
        pos, vel = z[0], 0.0
        p11, p12, p22 = 1.0, 0.0, 1.0                       # P0 = I
        want[0, ax] = z[0]
        for i in range(1, n):
            pos, vel = pos + vel, vel                        # x = F x
            p11, p12, p22 = p11 + 2 * p12 + p22 + 0.1, p12 + p22, p22 + 0.1      # P = F P F' + Q
            s = p11 + 0.1                                    # S = H P H' + R
            k1, k2 = p11 / s, p12 / s
            y = z[i] - pos
            pos, vel = pos + k1 * y, vel + k2 * y
            p11, p12, p22 = (1 - k1) * p11, (1 - k1) * p12, p22 - k2 * p12       # P = (I - K H) P
            want[i, ax] = pos
    assert np.abs(got - want).max() < 1e-10
