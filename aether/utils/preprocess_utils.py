"""`aether.utils.preprocess_utils` — the reference's names (/root/reference/aether/utils/preprocess_utils.py:4-39) on top of
aether_amd.preprocess."""
from aether_amd.preprocess import center_crop_frames


def imcrop_center(img_list, crop_p_h, crop_p_w):
    return center_crop_frames(img_list, crop_p_h, crop_p_w)
