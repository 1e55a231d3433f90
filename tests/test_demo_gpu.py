"""scripts/demo.py end to end on MI355X (the reference's CLI, /root/reference/scripts/demo.py): a long-video reconstruction — two
sliding windows, device-resident window outputs, device merge (HIP kernels), files written under the reference's names — and a
planning clip with classifier-free guidance + the 4-step post-reconstruction.  Synthetic weights, reduced geometry (96x240, 17
frames, 2 transformer blocks) so the test takes seconds."""
import importlib
import os
import sys

import numpy as np
import PIL.Image
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def demo():
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    return importlib.import_module("demo")


def _frames(n, h, w):
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    return np.stack([np.stack([0.5 + 0.4 * np.sin(0.05 * xx + 0.2 * t + c) * np.cos(0.04 * yy) for c in range(3)], -1) for t in range(n)])


def test_long_video_reconstruction_cli(cuda, hip_lib, demo, tmp_path):
    video = (_frames(25, 96, 240) * 255).astype(np.uint8)                       # 25 frames, windows of 17 at stride 24 -> starts [0, 8]
    np.save(tmp_path / "clip.npy", video)
    demo.main(["--task", "reconstruction", "--video", str(tmp_path / "clip.npy"), "--height", "96", "--width", "240", "--num_frames", "17",
               "--num_inference_steps", "2", "--synthetic_weights", "--synthetic_layers", "2", "--smooth_method", "simple",
               "--output_dir", str(tmp_path / "out")])
    z = np.load(tmp_path / "out" / "reconstruction_clip.npz")
    assert z["rgb"].shape == (25, 96, 240, 3) and z["disparity"].shape == (25, 96, 240)
    assert z["pointmap"].shape == (25, 96, 240, 3) and z["poses"].shape == (25, 4, 4)
    assert list(z["window_starts"]) == [0, 8]
    for k in ("rgb", "disparity", "pointmap", "poses"):
        assert np.isfinite(z[k]).all(), k
    assert 0 <= z["rgb"].min() and z["rgb"].max() <= 1


def test_planning_cli(cuda, hip_lib, demo, tmp_path):
    f = (_frames(17, 96, 240) * 255).astype(np.uint8)
    PIL.Image.fromarray(f[0]).save(tmp_path / "obs.png")
    PIL.Image.fromarray(f[-1]).save(tmp_path / "goal.png")
    demo.main(["--task", "planning", "--image", str(tmp_path / "obs.png"), "--goal", str(tmp_path / "goal.png"), "--height", "96", "--width", "240",
               "--num_frames", "17", "--num_inference_steps", "3", "--synthetic_weights", "--synthetic_layers", "2", "--smooth_method", "simple",
               "--output_dir", str(tmp_path / "out")])
    z = np.load(tmp_path / "out" / "planning_obs_goal.npz")
    assert z["rgb"].shape == (17, 96, 240, 3) and z["disparity"].shape == (17, 96, 240) and np.isfinite(z["pointmap"]).all()
