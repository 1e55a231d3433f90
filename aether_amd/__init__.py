"""aether_amd — MI355X (gfx950) native implementation of the AetherV1 latent-video denoising hot path.

Scope: the diffusion-transformer sampling loop and the CogVideoX 3D-causal VAE behind the reference's
`aether.pipelines.aetherv1_pipeline_cogvideox.AetherV1PipelineCogVideoX` entry point.  Compute runs in
hand-written HIP kernels (aether_amd/csrc, C ABI in include/aether_hip.h); PyTorch only owns memory and streams.
"""
__version__ = "0.1.0"
