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
    print("wrote", os.listdir(OUT))


if __name__ == "__main__":
    main()
