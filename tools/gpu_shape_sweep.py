"""Full-size shape sweep on MI355X: every frame count the reference admits (17 / 25 / 33 / 41 -> 5 / 7 / 9 / 11 latent frames,
S = 226 + f*1350 tokens) through the drop-in pipeline with the real-size random-weight DiT and VAE, reconstruction (B = 1) and
planning (B = 2), two steps each: outputs must be finite and of the right shape.  Catches shape-specific paths that the
scaled-down parity tests cannot reach (GEMM tail launches, attention two-launch split, VAE frame chunking)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def sweep(frame_counts=(17, 25, 33, 41), steps=2):
    from aether.pipelines.aetherv1_pipeline_cogvideox import AetherV1PipelineCogVideoX
    from aether_amd.scheduler import CogVideoXDPMScheduler
    from aether_amd.transformer import AetherTransformer3D
    from aether_amd.vae import AetherVAE
    dev = torch.device("cuda:0")
    model = AetherTransformer3D({}, device=dev).init_random_weights(seed=0)
    vae = AetherVAE(device=dev).init_random_weights(1)
    vae.enable_slicing(); vae.enable_tiling()
    g = torch.Generator().manual_seed(0)
    prompt = (torch.randn(1, 226, 4096, generator=g) * 0.1).to(torch.bfloat16)
    pipe = AetherV1PipelineCogVideoX(tokenizer=None, text_encoder=None, vae=vae, scheduler=CogVideoXDPMScheduler(),
                                     transformer=model, empty_prompt_embeds=prompt)
    pipe.set_progress_bar_config(disable=True)
    yy, xx = np.mgrid[0:480, 0:720].astype(np.float32)
    ok = True
    results = []
    for F in frame_counts:
        video = np.stack([np.stack([0.5 + 0.4 * np.sin(0.02 * xx + 0.1 * t + c) * np.cos(0.015 * yy) for c in range(3)], -1)
                          for t in range(F)]).astype(np.float32)
        for task, kw in (("reconstruction", dict(video=video)), ("planning", dict(image=video[0], goal=video[-1]))):
            t0 = time.perf_counter()
            out = pipe(task=task, height=480, width=720, num_frames=F, num_inference_steps=steps, fps=12,
                       generator=torch.Generator(device=dev).manual_seed(42), **kw)
            torch.cuda.synchronize()
            good = (out.rgb.shape == (F, 480, 720, 3) and out.disparity.shape == (F, 480, 720) and out.raymap.shape == (F, 6, 60, 90)
                    and np.isfinite(out.rgb).all() and np.isfinite(out.disparity).all() and np.isfinite(out.raymap).all())
            ok &= bool(good)
            results.append({"frames": F, "task": task, "ok": bool(good), "seconds": round(time.perf_counter() - t0, 2),
                            "rgb_std": float(out.rgb.std())})
            print(results[-1], flush=True)
    return ok, results


def main():
    ok, _ = sweep()
    print("SHAPE SWEEP", "OK" if ok else "FAILED")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
