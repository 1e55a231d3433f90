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


if __name__ == "__main__":
    main()
