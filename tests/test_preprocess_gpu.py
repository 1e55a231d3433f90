"""Device preprocessing: aether_preprocess_frames against the reference-shaped host path
(`_preprocess_image`: uint8 -> /255, imcrop_center, VideoProcessor.preprocess with its nearest resize, 2x-1, bf16), bit for bit,
for every uint8 value, for windows that leave the frame (zero fill) and for up- and down-sampling sizes."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pipe(cuda):
    from aether.pipelines.aetherv1_pipeline_cogvideox import AetherV1PipelineCogVideoX
    p = AetherV1PipelineCogVideoX.__new__(AetherV1PipelineCogVideoX)
    from aether_amd.video_processor import VideoProcessor
    p.video_processor = VideoProcessor(vae_scale_factor=8)
    return p


@pytest.mark.parametrize("shape,target", [((2, 48, 72, 3), (48, 72)), ((1, 50, 72, 3), (48, 72)), ((3, 37, 41, 3), (48, 72)), ((2, 100, 60, 3), (16, 40)),
                                          ((1, 480, 720, 3), (480, 720)), ((2, 240, 368, 3), (480, 720)), ((1, 91, 33, 3), (8, 24))])
@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_device_preprocessing_equals_host_path(cuda, hip_lib, shape, target, dtype):
    g = np.random.default_rng(shape[1] * 7 + shape[2])
    frames = g.integers(0, 256, size=shape, dtype=np.uint8) if dtype == np.uint8 else g.random(shape, dtype=np.float32)
    pipe = _pipe(cuda)
    h, w = target
    host = pipe._preprocess_image(frames.copy(), h, w).to(torch.bfloat16)                     # the reference-shaped host path
    dev = pipe._preprocess_frames_on_device(frames, h, w, cuda)
    torch.cuda.synchronize()
    assert dev is not None and dev.shape == host.shape == (shape[0], 3, h, w)
    assert torch.equal(dev.cpu(), host), (dev.cpu().float() - host.float()).abs().max()


def test_all_uint8_values(cuda, hip_lib):
    """Every pixel value through /255, 2x-1 and the bf16 rounding (the reciprocal-multiply shortcut differs for 126 of them)."""
    frames = np.arange(256, dtype=np.uint8).reshape(1, 16, 16, 1).repeat(3, axis=3)
    pipe = _pipe(cuda)
    host = pipe._preprocess_image(frames.copy(), 16, 16).to(torch.bfloat16)
    dev = pipe._preprocess_frames_on_device(frames, 16, 16, cuda)
    assert torch.equal(dev.cpu(), host)
