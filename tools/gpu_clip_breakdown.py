"""Wall-clock anatomy of one reconstruction clip (41x480x720) through the drop-in pipeline on MI355X: where the time of
the reference's default 4-step call goes (VAE encode, transformer forwards, scheduler/cat glue, the two decodes, host
pre/post-processing and the D2H copies).  Synchronises around every module call, so the total is slightly pessimistic."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from aether.pipelines.aetherv1_pipeline_cogvideox import AetherV1PipelineCogVideoX
    from aether_amd.scheduler import CogVideoXDPMScheduler
    from aether_amd.transformer import AetherTransformer3D
    from aether_amd.vae import AetherVAE
    dev = torch.device("cuda:0")
    model = AetherTransformer3D({}, device=dev).init_random_weights(seed=0)
    vae = AetherVAE(device=dev).init_random_weights(1)
    vae.enable_slicing(); vae.enable_tiling()
    acc = {}

    def timed(name, fn):
        def wrapper(*a, **k):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r = fn(*a, **k)
            torch.cuda.synchronize(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
            return r
        return wrapper

    vae.encode = timed("vae_encode", vae.encode)
    vae.decode = timed("vae_decode_x2", vae.decode)
    class Timed:                                  # the transformer object with a timed __call__, everything else delegated
        def __init__(self, m):
            self._m, self._call = m, timed("transformer", m.__call__)

        def __call__(self, *a, **k):
            return self._call(*a, **k)

        def __getattr__(self, name):
            return getattr(self._m, name)

    g = torch.Generator().manual_seed(0)
    prompt = (torch.randn(1, 226, 4096, generator=g) * 0.1).to(torch.bfloat16)
    pipe = AetherV1PipelineCogVideoX(tokenizer=None, text_encoder=None, vae=vae, scheduler=CogVideoXDPMScheduler(),
                                     transformer=Timed(model), empty_prompt_embeds=prompt)
    pipe.set_progress_bar_config(disable=True)
    yy, xx = np.mgrid[0:480, 0:720].astype(np.float32)
    video = np.stack([np.stack([0.5 + 0.4 * np.sin(0.02 * xx + 0.1 * t + c) * np.cos(0.015 * yy) for c in range(3)], -1)
                      for t in range(41)]).astype(np.float32)
    res = {}
    for label, n in (("warmup", 1), ("4_steps", 4), ("50_steps", 50)):
        acc.clear()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pipe(task="reconstruction", video=video, height=480, width=720, num_frames=41, num_inference_steps=n, fps=12,
             generator=torch.Generator(device=dev).manual_seed(42))
        torch.cuda.synchronize(); total = time.perf_counter() - t0
        if label != "warmup":
            d = dict(acc); d["total"] = total; d["other (pre/post-processing, scheduler glue, D2H)"] = total - sum(acc.values())
            res[label] = {k: round(v, 4) for k, v in d.items()}
            print(label, res[label], flush=True)
    # BASELINE configs[2] / [3]: prediction (image + raymap) and planning (image + goal): 50 steps with CFG (B = 2 through the
    # DiT) followed, like scripts/demo.py:589-606, by the 4-step post-reconstruction of the predicted clip
    img = video[0]
    goal = video[-1]
    tt = np.linspace(0, 1, 41, dtype=np.float32)[:, None, None, None]
    yy8, xx8 = np.mgrid[0:60, 0:90].astype(np.float32)
    rays = np.stack([xx8 / 90 - 0.5, yy8 / 60 - 0.5, np.ones_like(xx8)], 0)[None]
    raymap = np.concatenate([rays + 0.1 * tt * np.array([1, 0, 0], np.float32)[None, :, None, None],
                             tt * np.array([0.3, 0.0, 1.0], np.float32)[None, :, None, None] * np.ones_like(rays)], 1).astype(np.float32)
    for task, kw in (("prediction", dict(image=img, raymap=raymap)), ("planning", dict(image=img, goal=goal))):
        acc.clear()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = pipe(task=task, height=480, width=720, num_frames=41, fps=12, generator=torch.Generator(device=dev).manual_seed(42), **kw)
        pipe(task="reconstruction", video=out.rgb, height=480, width=720, num_frames=41, num_inference_steps=4, guidance_scale=1.0,
             use_dynamic_cfg=False, fps=12, generator=torch.Generator(device=dev).manual_seed(42))
        torch.cuda.synchronize(); total = time.perf_counter() - t0
        d = dict(acc); d["total"] = total; d["other (pre/post-processing, scheduler glue, D2H)"] = total - sum(acc.values())
        res[f"{task}_50_steps_cfg_plus_post_reconstruction"] = {k: round(v, 4) for k, v in d.items()}
        print(task, res[f"{task}_50_steps_cfg_plus_post_reconstruction"], flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/clip_breakdown.json", "w"), indent=1)


if __name__ == "__main__":
    main()
