"""Calibration of the full-depth parity thresholds: the SAME oracle transformer evaluated in the reference dtype (bf16 weights, bf16
activations, PyTorch CPU kernels) on the inputs of the 42-block fixture, compared with the fp32 fixture (tests/golden/fullsize_dit.npz).
The distance bf16-oracle <-> fp32-oracle is what "the reference dtype" itself costs at this depth and size; the HIP path's distance to the
same fixture (tests/test_fullsize_parity_gpu.py) should sit at or below it.  ~1 h of CPU on 8 vCPUs (bf16 matmuls are emulated here).

    nice python tools/calibrate_fullsize_bf16.py          # writes profiles/r03_bf16_oracle_calibration.json
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fullsize_cases as fc  # noqa: E402


def main():
    torch.set_num_threads(os.cpu_count() or 8)
    t0 = time.perf_counter()
    dit, _ = fc.build_oracle_dit()
    dit = dit.to(torch.bfloat16)                      # weights are bf16-representable: exact
    hidden, text, t = fc.dit_inputs()
    rope = fc.rope_tables()
    print(f"built in {time.perf_counter() - t0:.0f} s; running the bf16 forward ...", flush=True)
    t0 = time.perf_counter()
    with torch.no_grad():
        out = dit(hidden, text, t, image_rotary_emb=rope)[0].float()
    dt = time.perf_counter() - t0
    ref = torch.from_numpy(np.load(os.path.join(fc.GOLDEN_DIR, "fullsize_dit.npz"))["out"].astype(np.float32))
    m = fc.metrics(out, ref)
    res = {"case": "42 blocks, S = 15 076, B = 1: bf16 oracle (torch CPU) vs the fp32 oracle fixture", "seconds_cpu": dt, **m}
    print(json.dumps(res), flush=True)
    with open(os.path.join(fc.ROOT, "profiles", "r03_bf16_oracle_calibration.json"), "w") as f:
        json.dump(res, f, indent=1)


def guided_forward_inputs(task):
    """The B = 2 transformer input of step 0 of a guided fixture (tests/golden/fullsize_<task>.npz), rebuilt exactly as
    tests/test_fullsize_guided_gpu.py::test_guided_condition_and_b2_forward rebuilds it: the oracle's sampled image (/ goal) latents from the
    fixture, zero padding, raymap packing, the unconditional branch, and the initial noise replayed from the CPU generator."""
    from einops import rearrange
    z = np.load(os.path.join(fc.GOLDEN_DIR, f"fullsize_{task}.npz"))
    case = fc.GUIDED_CASES[task]
    gen = torch.Generator().manual_seed(fc.GUIDED_SEED)
    for _ in range(2 if case["goal"] else 1):
        torch.randn((1, 16, 1, fc.LAT_H, fc.LAT_W), generator=gen, dtype=torch.bfloat16)
    latents = torch.randn((1, fc.LAT_F, 56, fc.LAT_H, fc.LAT_W), generator=gen, dtype=torch.bfloat16)
    assert abs(float(latents.double().sum()) - float(z["initial_latents_sum"])) < 1e-6 * max(1.0, abs(float(z["initial_latents_sum"])))
    parts = [fc.from_bf16_bits(z["image_latents_bits"]), torch.zeros(1, fc.LAT_F - (2 if case["goal"] else 1), 16, fc.LAT_H, fc.LAT_W, dtype=torch.bfloat16)]
    if case["goal"]:
        parts.append(fc.from_bf16_bits(z["goal_latents_bits"]))
    if case["raymap"]:
        rm = torch.from_numpy(fc.forward_right_raymap())[None].to(torch.bfloat16)
        rm = torch.cat([rm[:, : 4 - rm.shape[1] % 4], rm], dim=1)
        cam = rearrange(rm, "b (n t) c h w -> b t (n c) h w", n=4)
    else:
        cam = torch.zeros(1, fc.LAT_F, 24, fc.LAT_H, fc.LAT_W, dtype=torch.bfloat16)
    cond = torch.cat([torch.cat(parts, dim=1), cam], dim=2)
    un = cond.clone()
    if task == "planning":
        un[:, :, :16] = 0
    else:
        un[:, :1, :16] = 0
    return torch.cat([torch.cat([latents] * 2), torch.cat([un, cond])], dim=2), torch.from_numpy(z["noise_pred0_s2"].astype(np.float32))


def guided(tasks):
    """bf16 oracle of the guided B = 2 forward (42 blocks) against the fp32 fixture: the reference dtype's own distance for configs[2] / [3]."""
    torch.set_num_threads(int(os.environ.get("AETHER_GOLDEN_THREADS", os.cpu_count() or 8)))
    dit, _ = fc.build_oracle_dit()
    dit = dit.to(torch.bfloat16)
    rope = fc.rope_tables()
    res = {}
    for task in tasks:
        model_in, ref = guided_forward_inputs(task)
        t0 = time.perf_counter()
        with torch.no_grad():
            out = dit(model_in, fc.prompt_embeds().repeat(2, 1, 1), torch.tensor([999, 999]), image_rotary_emb=rope)[0].float()[..., ::2, ::2]
        dt = time.perf_counter() - t0
        res[task] = {"seconds_cpu": dt, "unconditional": fc.metrics(out[0], ref[0]), "conditional": fc.metrics(out[1], ref[1])}
        print(task, json.dumps(res[task]), flush=True)
        with open(os.path.join(fc.ROOT, "profiles", "r04_bf16_oracle_calibration_guided.json"), "w") as f:
            json.dump({"case": "42 blocks, S = 15 076, B = 2 (step 0 of the guided fixtures): bf16 oracle (torch CPU) vs the fp32 oracle fixture, every 2nd row / column", **res}, f, indent=1)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "guided":
        guided(sys.argv[2:] or ["prediction", "planning"])
    else:
        main()
