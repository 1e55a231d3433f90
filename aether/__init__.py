"""Import-path alias so reference user code (`from aether.pipelines.aetherv1_pipeline_cogvideox import ...`,
/root/reference/scripts/demo.py:22-25) runs unchanged on top of aether_amd."""
