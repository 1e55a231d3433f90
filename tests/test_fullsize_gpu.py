"""Full-size smoke of the drop-in pipeline on MI355X (real-size random-weight DiT + VAE, 480x720): the shortest and the longest
frame count the reference admits, reconstruction (B = 1) and planning (B = 2, CFG), one step each — finite outputs of the right
shape.  Reaches the paths the scaled-down parity tests cannot (GEMM tail launches, the attention two-launch split at 2832 / 5664
workgroups, the VAE's 4/2/2/1 tile batches and frame chunking at 480x720)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_full_size_pipeline_shapes(cuda, hip_lib):
    from gpu_shape_sweep import sweep
    ok, results = sweep(frame_counts=(17, 41), steps=1)
    assert ok, results
