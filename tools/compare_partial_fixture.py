"""Cross-check of the device-generated long guided fixture (tools/make_fullsize_golden_gpu.py) against the CHECKPOINTS of the CPU run of the same
call (tools/make_fullsize_golden.py prediction50 -> <name>.partial.npz): kept-step latents (every 6th row / column, bf16 values) of the steps the CPU
run finished.  Both are the fp32 oracle; they differ by fp32 summation order (and by a few bf16 roundings of the condition latents, whose fp32 VAE
encode ran on two different host CPUs), re-quantised to bf16 after every step.

    python tools/compare_partial_fixture.py /tmp/aether_fixture/fullsize_prediction50.partial.npz tests/golden/fullsize_prediction50.npz out.json
"""
import json
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
import fullsize_cases as fc  # noqa: E402


def main():
    part, full = np.load(sys.argv[1]), np.load(sys.argv[2])
    mp, mf = json.loads(str(part["meta"])), json.loads(str(full["meta"]))
    assert mp["timesteps"] == mf["timesteps"] and np.allclose(mp["guidance_scales"], mf["guidance_scales"], rtol=0, atol=1e-12)
    res = {"case": "CPU fp32 oracle (checkpoints of the 10.5-hour run) vs the device fp32 oracle fixture, same call", "cpu_steps_done": mp["steps_done"],
           "cpu_seconds_per_step": [round(v, 1) for v in mp["step_seconds"].values()], "per_step": {}}
    for k, i in enumerate(mp["kept_steps"]):
        j = mf["kept_steps"].index(i)
        a, b = fc.from_bf16_bits(part["step_latents_s6"][k]).float(), fc.from_bf16_bits(full["step_latents_s6"][j]).float()
        m = fc.metrics(a, b)
        res["per_step"][i] = {"rel_l2": m["rel_l2"], "linf_rel": m["linf_rel"], "bf16_values_that_differ": float((a != b).float().mean())}
    res["noise_pred_rms_cpu"] = mp["noise_pred_rms"]
    res["noise_pred_rms_device"] = mf["noise_pred_rms"][:len(mp["noise_pred_rms"])]
    print(json.dumps(res, indent=1))
    if len(sys.argv) > 3:
        json.dump(res, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
