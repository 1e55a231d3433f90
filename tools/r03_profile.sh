#!/bin/bash
# Round-3 rocprofv3 evidence in ONE gpurun call: DiT step (kernel trace + separate PMC passes) and the VAE (kernel trace).
# Summaries land in gpurun_out/r03_dit_step_summary.{json,md} and gpurun_out/r03_vae_summary.{json,md}: copy them to profiles/.
set -u
bash tools/profile_dit.sh r03_dit_step
bash tools/profile_vae.sh r03_vae
ls -la gpurun_out/*summary* 2>/dev/null
