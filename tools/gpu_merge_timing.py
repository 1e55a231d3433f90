"""Wall-clock of the device-side window merge for a 192-frame clip (8 windows of 41x480x720, stride 24), inputs already on
the GPU as run_windows(keep_on_device=True) leaves them.  MI355X only."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aether_amd.geometry import camera_pose_to_raymap  # noqa: E402
from aether_amd.windows import WindowResult, blend_and_merge_window_results, get_window_starts  # noqa: E402

dev = torch.device("cuda:0")
H, W, F, N = 480, 720, 41, 192
t = np.linspace(0, 1, N)
world = np.tile(np.eye(4), (N, 1, 1))
world[:, 0, 3], world[:, 2, 3] = 0.5 * t, 1.5 * t
K = np.array([[400.0, 0, 360.0], [0, 400.0, 240.0], [0, 0, 1.0]])
g = torch.Generator(device=dev).manual_seed(0)
res = []
for s in get_window_starts(N, F, 24):
    rel = np.linalg.inv(world[s]) @ world[s:s + F]
    res.append(WindowResult(s, torch.rand(F, H, W, 3, generator=g, device=dev), 0.2 + 0.6 * torch.rand(F, H, W, generator=g, device=dev),
                            camera_pose_to_raymap(rel, np.tile(K, (F, 1, 1)))))
torch.cuda.synchronize()
t0 = time.perf_counter()
out = blend_and_merge_window_results(res, height=H, width=W, smooth_camera=True, smooth_method="kalman", device=dev)
print({"windows": len(res), "merge_incl_D2H_s": round(time.perf_counter() - t0, 3), "shapes": [o.shape for o in out]})
