"""`aether.pipelines.aetherv1_pipeline_cogvideox` — re-export of the MI355X-native implementation."""
from aether_amd.pipelines.aetherv1_pipeline_cogvideox import *  # noqa: F401,F403
from aether_amd.pipelines.aetherv1_pipeline_cogvideox import (  # noqa: F401
    AetherV1PipelineCogVideoX,
    AetherV1PipelineOutput,
    get_3d_rotary_pos_embed,
    get_resize_crop_region_for_grid,
    retrieve_latents,
    retrieve_timesteps,
)
