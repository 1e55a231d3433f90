"""Generates the two window-merge fixtures that tools/make_golden.py's toy `blend.npz` does not cover (run in the build container only,
where /root/reference exists):

    python tools/make_blend_golden.py            # both
    python tools/make_blend_golden.py kalman     # tests/golden/blend_kalman.npz only
    python tools/make_blend_golden.py fullsize   # tests/golden/blend_fullsize.npz only

* `blend_fullsize.npz` — BASELINE configs[4] at ITS OWN geometry: three 41 x 480 x 720 windows with starts [0, 24, 31] of a 72-frame clip, i.e.
  the two overlap lengths the reference's `get_window_starts(192, 41, 24)` = [0, 24, ..., 144, 151] produces (17 frames, and the 34-frame tail
  overlap), merged by the REFERENCE's own `blend_and_merge_window_results` (scripts/demo.py:254-422, its source segment executed against the
  reference's own aether/utils/postprocess_utils.py) with camera smoothing off, "simple" and "kalman" (the CLI default, D:173-179).  Stored:
  a pixel lattice of the merged rgb / disparity / point maps, per-frame float64 sums of the whole arrays, all 72 poses, and the disparity scale
  the reference fitted for windows 1 and 2 (captured at its `compute_scale` call).  The INPUTS are not stored: `blend_fullsize_inputs()` below
  rebuilds them from integer arithmetic (+ the window cameras kept in the fixture), on the GPU box too.
* `blend_kalman.npz` — the toy windows of `blend.npz` merged with `smooth_method="kalman"`, plus `smooth_trajectory` (U:751-844) itself on two
  trajectories.  `filterpy` (imported at U:759) is absent from this image, so a stand-in module is injected for the duration of the run: the
  predict / update equations of `filterpy.kalman.KalmanFilter` as published (Labbe, "Kalman and Bayesian Filters in Python"; filterpy 1.4.5
  kalman_filter.py: x = Fx, P = a²FPFᵀ + Q; y = z − Hx, S = HPHᵀ + R, K = PHᵀS⁻¹, x += Ky, P = (I − KH)P(I − KH)ᵀ + KRKᵀ).  Everything else —
  the gaussian pre-smoothing, the filter's matrices, the quaternion averaging, the static-sequence switch — is the reference's own code.

Test infrastructure; nothing of the reference is stored but its outputs.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
OUT = os.path.join(ROOT, "tests", "golden")

FS_STARTS, FS_FRAMES, FS_H, FS_W, FS_TOTAL = (0, 24, 31), 41, 480, 720, 72
FS_LATTICE = (slice(None, None, 3), slice(None, None, 12), slice(None, None, 12))      # frames 0,3,..69 x rows 0,12,.. x cols 0,12,..
FS_WINDOW_SCALES = (1.0, 1.25, 0.8)


# ---------------------------------------------------------------------------------------------------------------------------------------------
# filterpy stand-in (test infrastructure): the published KalmanFilter.predict / update, nothing else of the package
# ---------------------------------------------------------------------------------------------------------------------------------------------
class _KalmanFilter:
    def __init__(self, dim_x, dim_z, dim_u=0):
        self.dim_x, self.dim_z = dim_x, dim_z
        self.x = np.zeros((dim_x, 1))
        self.P, self.Q, self.F = np.eye(dim_x), np.eye(dim_x), np.eye(dim_x)
        self.H, self.R = np.zeros((dim_z, dim_x)), np.eye(dim_z)
        self._alpha_sq = 1.0
        self._I = np.eye(dim_x)

    def predict(self):
        self.x = np.dot(self.F, self.x)
        self.P = self._alpha_sq * np.dot(np.dot(self.F, self.P), self.F.T) + self.Q

    def update(self, z):
        z = np.asarray(z, float)
        z = z.reshape(self.dim_z) if self.x.ndim == 1 else z.reshape(self.dim_z, 1)       # filterpy.common.reshape_z
        y = z - np.dot(self.H, self.x)
        PHT = np.dot(self.P, self.H.T)
        S = np.dot(self.H, PHT) + self.R
        K = np.dot(PHT, np.linalg.inv(S))
        self.x = self.x + np.dot(K, y)
        I_KH = self._I - np.dot(K, self.H)
        self.P = np.dot(np.dot(I_KH, self.P), I_KH.T) + np.dot(np.dot(K, self.R), K.T)


def install_filterpy_stand_in():
    pkg, kal = types.ModuleType("filterpy"), types.ModuleType("filterpy.kalman")
    kal.KalmanFilter = _KalmanFilter
    pkg.kalman = kal
    sys.modules["filterpy"], sys.modules["filterpy.kalman"] = pkg, kal


# ---------------------------------------------------------------------------------------------------------------------------------------------
# inputs of the full-size case: integer arithmetic + IEEE basic operations only (identical on every machine), cameras from the fixture
# ---------------------------------------------------------------------------------------------------------------------------------------------
def blend_fullsize_window(k: int, cams: np.ndarray, K: np.ndarray):
    """Window `k` of the 72-frame clip: (rgb [41,480,720,3] f32, disparity [41,480,720] f32, raymap [41,6,60,90] f32).  The scene content is a
    function of the GLOBAL frame index (overlapping windows see the same scene), each window predicts it at its own disparity scale and in its own
    camera frame (`cams` [41,4,4]: what tools/make_blend_golden.py drew and stored), with a small window-specific error pattern on top."""
    from aether_amd import geometry as G
    s0 = FS_STARTS[k]
    f, y, x = np.meshgrid(np.arange(FS_FRAMES, dtype=np.int64), np.arange(FS_H, dtype=np.int64), np.arange(FS_W, dtype=np.int64), indexing="ij")
    g = f + s0
    tri = np.abs(((x * 3 + y * 5 + g * 11) % 512) - 256).astype(np.float32) / np.float32(256)          # triangle wave in [0, 1]
    err = ((x * 7 + y * 13 + f * 29 + k * 31) % 101).astype(np.float32) / np.float32(101)
    disp = (np.float32(0.06) + np.float32(0.62) * tri + np.float32(0.01) * err) * np.float32(FS_WINDOW_SCALES[k])   # some pixels below the 0.1 mask threshold
    base = ((x * 7 + y * 13 + g * 29) % 101).astype(np.float32) / np.float32(101)
    third = ((x + 2 * y + 3 * g + 17 * k) % 50).astype(np.float32) / np.float32(50)
    rgb = np.stack([base, base * np.float32(0.5) + np.float32(0.25), third], -1)
    del f, y, x, g, tri, err, base, third
    ray = G.camera_pose_to_raymap(cams.astype(np.float32), np.tile(K, (FS_FRAMES, 1, 1)), H=FS_H, W=FS_W)
    ff, c, yy, xx = np.meshgrid(np.arange(FS_FRAMES), np.arange(3), np.arange(FS_H // 8), np.arange(FS_W // 8), indexing="ij")
    ray[:, :3] += np.float32(2e-3) * (((xx * 5 + yy * 11 + ff * 3 + c * 7 + k * 13) % 64).astype(np.float32) / np.float32(64) - np.float32(0.5))
    return rgb, disp.astype(np.float32), ray.astype(np.float32)


def _draw_fullsize_cameras():
    """72 world cameras on a gently curving forward-right path; each window expresses its 41 in its own first-frame system, at its own
    translation scale, with a little prediction noise (as AetherV1 windows do: every window is an independent pipeline call)."""
    rng = np.random.default_rng(23)
    t = np.linspace(0, 1, FS_TOTAL)
    world = np.tile(np.eye(4), (FS_TOTAL, 1, 1))
    ang = 0.45 * t * t
    world[:, 0, 0], world[:, 0, 2], world[:, 2, 0], world[:, 2, 2] = np.cos(ang), np.sin(ang), -np.sin(ang), np.cos(ang)
    world[:, 0, 3], world[:, 1, 3], world[:, 2, 3] = 0.35 * t, 0.04 * np.sin(5 * t), 1.4 * t
    cams = []
    for k, s0 in enumerate(FS_STARTS):
        rel = np.linalg.inv(world[s0]) @ world[s0:s0 + FS_FRAMES]
        rel[:, :3, 3] *= (1.0, 0.8, 1.3)[k]
        rel[:, :3, 3] += 2e-3 * rng.standard_normal((FS_FRAMES, 3))
        cams.append(rel)
    K = np.array([[520.0, 0, FS_W / 2], [0, 520.0, FS_H / 2], [0, 0, 1.0]])
    return np.stack(cams), K


def make_fullsize(U, blend):
    import types as _t
    cams, K = _draw_fullsize_cameras()
    out = {"starts": np.array(FS_STARTS), "dims": np.array([FS_FRAMES, FS_H, FS_W, FS_TOTAL]), "cams": cams, "K": K,
           "lattice_step": np.array([s.step for s in FS_LATTICE])}
    wins = [blend_fullsize_window(k, cams[k], K) for k in range(len(FS_STARTS))]
    out["input_sums"] = np.array([[a.sum(dtype=np.float64) for a in w] for w in wins])       # the tests check their regenerated inputs against these
    for tag, (sc, smeth) in {"plain": (False, "simple"), "simple": (True, "simple"), "kalman": (True, "kalman")}.items():
        scales = []

        def recording_scale(*a, _f=U.compute_scale, **kw):
            s = _f(*a, **kw)
            scales.append(float(s))
            return s
        blend.__globals__["compute_scale"] = recording_scale
        results = [_t.SimpleNamespace(rgb=r, disparity=d, raymap=m.copy()) for r, d, m in wins]     # the merge decodes raymaps in place
        args = _t.SimpleNamespace(align_pointmaps=False, smooth_camera=sc, smooth_method=smeth, width=FS_W, height=FS_H)
        m_rgb, m_disp, m_poses, m_pm = blend(results, list(FS_STARTS), args)
        blend.__globals__["compute_scale"] = U.compute_scale
        assert m_rgb.shape == (FS_TOTAL, FS_H, FS_W, 3) and m_pm.shape == (FS_TOTAL, FS_H, FS_W, 3) and len(scales) == 2
        out[f"{tag}_poses"] = m_poses
        out[f"{tag}_pointmaps"] = m_pm[FS_LATTICE].astype(np.float32)
        out[f"{tag}_pointmaps_frame_sums"] = m_pm.sum(axis=(1, 2), dtype=np.float64)
        out[f"{tag}_scales"] = np.array(scales)
        if tag == "plain":                                         # colour and disparity do not depend on the camera options
            out["rgb"], out["rgb_frame_sums"] = m_rgb[FS_LATTICE].astype(np.float32), m_rgb.sum(axis=(1, 2, 3), dtype=np.float64)
            out["disparity"], out["disparity_frame_sums"] = m_disp[FS_LATTICE].astype(np.float32), m_disp.sum(axis=(1, 2), dtype=np.float64)
        else:
            assert np.array_equal(m_disp[FS_LATTICE].astype(np.float32), out["disparity"])
        print(f"[blend_fullsize] {tag}: scales {scales}", flush=True)
        del m_rgb, m_disp, m_pm, results
    np.savez_compressed(os.path.join(OUT, "blend_fullsize.npz"), **out)
    print("wrote blend_fullsize.npz", os.path.getsize(os.path.join(OUT, "blend_fullsize.npz")) // 1024, "KiB")


def make_kalman(U, blend):
    import types as _t
    G = np.load(os.path.join(OUT, "blend.npz"))
    H, W = (int(v) for v in G["hw"])
    starts = [int(s) for s in G["starts"]]
    out = {}
    for tag, ap in (("kalman", False), ("kalman_aligned", True)):
        results = [_t.SimpleNamespace(rgb=G[f"rgb_{k}"].astype(np.float32), disparity=G[f"disparity_{k}"].copy(), raymap=G[f"raymap_{k}"].copy())
                   for k in range(len(starts))]
        args = _t.SimpleNamespace(align_pointmaps=ap, smooth_camera=True, smooth_method="kalman", width=W, height=H)
        m_rgb, m_disp, m_poses, m_pm = blend(results, starts, args)
        out[f"{tag}_disparity"], out[f"{tag}_poses"], out[f"{tag}_pointmaps"] = m_disp.astype(np.float32), m_poses, m_pm.astype(np.float32)
    # smooth_trajectory itself (U:751-844): a window's decoded cameras, and a noisy 23-camera walk with large rotations
    p, _, _ = U.raymap_to_poses(G["raymap_1"].copy(), ray_o_scale_inv=0.1)
    out["unit_in_a"], out["unit_out_a"] = p, U.smooth_trajectory(p.copy(), window_size=5)
    from scipy.spatial.transform import Rotation as R
    rng = np.random.default_rng(5)
    n = 23
    walk = np.tile(np.eye(4), (n, 1, 1))
    walk[:, :3, 3] = np.cumsum(rng.normal(0.02, 0.05, (n, 3)), axis=0)
    walk[:, :3, :3] = R.from_rotvec(np.cumsum(rng.normal(0.0, 0.25, (n, 3)), axis=0)).as_matrix()
    out["unit_in_b"], out["unit_out_b"], out["unit_out_b_w9"] = walk, U.smooth_trajectory(walk.copy(), window_size=5), U.smooth_trajectory(walk.copy(), window_size=9)
    np.savez_compressed(os.path.join(OUT, "blend_kalman.npz"), **out)
    print("wrote blend_kalman.npz", os.path.getsize(os.path.join(OUT, "blend_kalman.npz")) // 1024, "KiB")


def main():
    import make_golden as MG
    sys.path.insert(0, MG.REF)
    sys.modules.setdefault("plyfile", types.SimpleNamespace(PlyData=None, PlyElement=None))
    install_filterpy_stand_in()
    from aether.utils import postprocess_utils as U
    assert U.__file__.startswith(MG.REF), U.__file__
    blend = MG._reference_blend_function(U)
    what = sys.argv[1] if len(sys.argv) > 1 else "both"
    if what in ("both", "kalman"):
        make_kalman(U, blend)
    if what in ("both", "fullsize"):
        make_fullsize(U, blend)


if __name__ == "__main__":
    main()
