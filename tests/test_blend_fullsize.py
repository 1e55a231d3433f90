"""BASELINE configs[4] (long-video reconstruction: sliding 41-frame windows with temporal blend) at ITS OWN geometry, against the REFERENCE.

`get_window_starts(192, 41, 24)` (scripts/demo.py:235-251) = [0, 24, ..., 144, 151]: consecutive windows overlap by 17 frames, the tail window
by 34.  tests/golden/blend_fullsize.npz holds what the reference's own `blend_and_merge_window_results` (D:254-422, run by
tools/make_blend_golden.py in the build container) produced for three 41 x 480 x 720 windows with starts [0, 24, 31] — both overlap lengths —
with camera smoothing off, "simple" and "kalman" (the CLI default): a pixel lattice of the merged rgb / disparity / point maps, per-frame float64
sums of the whole arrays, all 72 poses and the two fitted disparity scales.  The inputs are rebuilt here from integer arithmetic.

CPU: the host (numpy, float64) merge.  `-m gpu`: `run_windows_merged` — the HIP merge kernels of csrc/merge_kernels.hip (masked scale-fit
reduction over 17 and 34 x 480 x 720 pixels, fused scale + cross-fade, back-projection) behind the incremental WindowMerger.
Tolerances: 1e-5 of each tensor's scale on the pixel lattices and the point-map sums (the reference sums the scale fit in float32, the kernels in
float64: ~1e-7), 1e-6 on the per-frame rgb / disparity sums and the fitted scales, 1e-7 on the poses."""
import os
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_blend_golden as MB  # noqa: E402  (pure-numpy input builder; nothing of the reference is touched at test time)

from aether_amd.windows import WindowResult, blend_and_merge_window_results, get_window_starts, run_windows_merged  # noqa: E402

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "blend_fullsize.npz"))
TAGS = {"plain": dict(smooth_camera=False), "simple": dict(smooth_camera=True, smooth_method="simple"),
        "kalman": dict(smooth_camera=True, smooth_method="kalman")}


def test_fixture_geometry_is_the_baseline_configs():
    starts = get_window_starts(192, 41, 24)
    overlaps = sorted({a + 41 - b for a, b in zip(starts[:-1], starts[1:])})
    assert starts[-2:] == [144, 151] and overlaps == [17, 34]
    mine = [int(s) for s in GOLD["starts"]]
    assert sorted({a + 41 - b for a, b in zip(mine[:-1], mine[1:])}) == [17, 34] and tuple(GOLD["dims"]) == (41, 480, 720, 72)


@pytest.fixture(scope="module")
def windows():
    wins = [MB.blend_fullsize_window(k, GOLD["cams"][k], GOLD["K"]) for k in range(3)]
    sums = np.array([[a.sum(dtype=np.float64) for a in w] for w in wins])
    # rgb / disparity are integer arithmetic + IEEE basic operations: identical everywhere; the raymap goes through float32 log1p (1 ulp between hosts)
    assert np.array_equal(sums[:, :2], GOLD["input_sums"][:, :2])
    np.testing.assert_allclose(sums[:, 2], GOLD["input_sums"][:, 2], rtol=1e-6)
    return wins


def _check(tag, merged, scales=None):
    rgb, disp, poses, pm = merged
    lat = MB.FS_LATTICE

    def close(a, b, what, rtol=1e-5):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        assert a.shape == b.shape, (what, a.shape, b.shape)
        err = np.abs(a - b).max() / np.abs(b).max()
        assert err < rtol, f"{tag} {what}: max error {err:.3e} of the tensor's scale"
    assert rgb.shape == (72, 480, 720, 3) and disp.shape == (72, 480, 720) and pm.shape == (72, 480, 720, 3) and poses.shape == (72, 4, 4)
    close(rgb[lat], GOLD["rgb"], "rgb lattice")
    close(rgb.sum(axis=(1, 2, 3), dtype=np.float64), GOLD["rgb_frame_sums"], "rgb per-frame sums", 1e-6)
    close(disp[lat], GOLD["disparity"], "disparity lattice")
    close(disp.sum(axis=(1, 2), dtype=np.float64), GOLD["disparity_frame_sums"], "disparity per-frame sums", 1e-6)
    close(poses, GOLD[f"{tag}_poses"], "poses", 1e-7)
    close(pm[lat], GOLD[f"{tag}_pointmaps"], "point-map lattice")
    close(pm.sum(axis=(1, 2), dtype=np.float64), GOLD[f"{tag}_pointmaps_frame_sums"], "point-map per-frame sums")
    if scales is not None:
        close(np.asarray(scales), GOLD[f"{tag}_scales"], "fitted disparity scales", 1e-6)


@pytest.mark.parametrize("tag", list(TAGS))
def test_host_merge_matches_reference_at_baseline_geometry(windows, tag):
    from aether_amd import geometry as G
    scales, real = [], G.compute_scale

    def recording(*a, **k):
        scales.append(real(*a, **k))
        return scales[-1]
    G.compute_scale = recording
    try:
        merged = blend_and_merge_window_results([WindowResult(int(s), r, d, m.copy()) for s, (r, d, m) in zip(GOLD["starts"], windows)],
                                                height=480, width=720, **TAGS[tag])
    finally:
        G.compute_scale = real
    assert [round(float(s), 3) for s in scales] == [0.8, 1.25]          # 1 / the windows' own factors (1.25, 0.8 against 1.0): sanity of the case
    _check(tag, merged, scales)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(TAGS))
def test_hip_merge_matches_reference_at_baseline_geometry(windows, tag, cuda, hip_lib):
    """Windows handed over as DEVICE tensors (what the pipeline leaves with keep_outputs_on_device), merged incrementally by the HIP kernels."""
    import torch
    by_start = {int(s): w for s, w in zip(GOLD["starts"], windows)}

    def call_window(start):
        r, d, m = by_start[start]
        return types.SimpleNamespace(rgb=torch.from_numpy(r).to(cuda), disparity=torch.from_numpy(d).to(cuda), raymap=torch.from_numpy(m.copy()).to(cuda))
    merged = run_windows_merged(call_window, [int(s) for s in GOLD["starts"]], height=480, width=720, gather_device=cuda, **TAGS[tag])
    _check(tag, merged)


@pytest.mark.gpu
def test_hip_scale_fit_at_both_overlap_lengths(windows, cuda, hip_lib):
    """The scale the HIP reduction fits over the 17- and the 34-frame overlap (aether_merge_scale_fit) against the reference's captured values."""
    import torch
    from aether_amd.windows import WindowMerger
    m = WindowMerger(total_frames=72, window_frames=41, frame_hw=(480, 720), height=480, width=720, device=cuda, smooth_camera=False)
    got = []
    for s, (r, d, ray) in zip(GOLD["starts"], windows):
        m.add(WindowResult(int(s), torch.from_numpy(r).to(cuda), torch.from_numpy(d).to(cuda), ray.copy()))
        if s:
            got.append(float(m.scratch[4098].item()))
    np.testing.assert_allclose(got, GOLD["plain_scales"], rtol=1e-6)
    _check("plain", m.finish())
