"""Aspect-preserving centre crop of numpy frames before resizing — same results as the reference's
`imcrop_center` / `crop` (/root/reference/aether/utils/preprocess_utils.py:4-39), written as index arithmetic."""
from __future__ import annotations

from typing import List, Sequence

import numpy as np


def crop_window(h: int, w: int, target_h: int, target_w: int):
    """(top, left, crop_h, crop_w) of the largest centred window with the target aspect ratio."""
    if target_h / target_w > h / w:          # source too wide: trim left/right
        cw = int(h / target_h * target_w)
        return 0, int((w - h / target_h * target_w) / 2), h, cw
    ch = int(w / target_w * target_h)        # source too tall: trim top/bottom
    return int((h - w / target_w * target_h) / 2), 0, ch, w


def center_crop_frames(frames: Sequence[np.ndarray], target_h: int, target_w: int) -> List[np.ndarray]:
    out = []
    for img in frames:
        top, left, ch, cw = crop_window(img.shape[0], img.shape[1], target_h, target_w)
        canvas = np.zeros((ch, cw, *img.shape[2:]), dtype=img.dtype)
        # clip the window to the image (the reference zero-fills whatever falls outside)
        y0, x0 = max(top, 0), max(left, 0)
        y1, x1 = min(top + ch, img.shape[0]), min(left + cw, img.shape[1])
        canvas[y0 - top:y1 - top, x0 - left:x1 - left] = img[y0:y1, x0:x1]
        out.append(canvas)
    return out
