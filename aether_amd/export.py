"""What `scripts/demo.py` writes (reference save_output, /root/reference/scripts/demo.py:425-521): file naming, the colour map
of the disparity video, and the axis flips applied to point maps and camera poses before export.  Video containers (mp4,
through imageio) and the GLB scene (trimesh) need packages this image does not have: they are written when importable,
otherwise the same pixels go out as .npz + PNG."""
from __future__ import annotations

import os
from typing import Optional, Tuple

import numpy as np


def output_stem(task: str, video: Optional[str], image: Optional[str], goal: Optional[str]) -> str:
    """D:484-491: reconstruction_<video>, prediction_<image>, planning_<image>_<goal> (names cut at the first dot)."""
    base = lambda p: p.split("/")[-1].split(".")[0]  # noqa: E731
    if task == "reconstruction":
        return f"reconstruction_{base(video)}"
    if task == "prediction":
        return f"prediction_{base(image)}"
    if task == "planning":
        return f"planning_{base(image)}_{base(goal)}"
    raise ValueError(f"unknown task {task}")


def colorize_depth(depth: np.ndarray, cmap: str = "Spectral") -> np.ndarray:
    """aether/utils/postprocess_utils.py:49-56: (max - d) / (max - min) over the positive entries, clipped, through matplotlib's
    colour map; returns float RGB in [0, 1]."""
    import matplotlib
    pos = depth[depth > 0]
    lo, hi = pos.min(), pos.max()
    return matplotlib.colormaps[cmap](((hi - depth) / (hi - lo)).clip(0, 1), bytes=False)[..., 0:3]


def flip_for_export(pointmap: np.ndarray, poses: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """D:455-481: X and Y of the world frame are negated for viewers (points, camera positions, and rows AND columns 0, 1 of
    the camera rotation — so the rotation's upper-left 2x2 block ends up unchanged)."""
    pm = pointmap.copy()
    pm[..., 1] = -pm[..., 1]
    pm[..., 0] = -pm[..., 0]
    ps = poses.copy()
    ps[..., 1, :3] = -ps[..., 1, :3]
    ps[..., 0, :3] = -ps[..., 0, :3]
    ps[..., :3, 1] = -ps[..., :3, 1]
    ps[..., :3, 0] = -ps[..., :3, 0]
    ps[..., 1, 3] = -ps[..., 1, 3]
    ps[..., 0, 3] = -ps[..., 0, 3]
    return pm, ps


def write_video(path_mp4: str, frames_u8: np.ndarray, fps: int = 12) -> str:
    """`iio.imwrite(path, frames, fps=12)` as D:494-503 when imageio is importable; else the first frame as PNG (the frames
    themselves are in the .npz written next to it).  Returns the path written."""
    try:
        import imageio.v3 as iio
    except ImportError:
        import PIL.Image
        png = os.path.splitext(path_mp4)[0] + "_frame0.png"
        PIL.Image.fromarray(frames_u8[0]).save(png)
        return png
    iio.imwrite(path_mp4, frames_u8, fps=fps)
    return path_mp4
