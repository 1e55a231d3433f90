"""Writes tests/golden/named_inputs.npz: the three example observations BASELINE configs[2] / configs[3] name
(/root/reference/assets/example_obs/car.png, assets/example_obs_goal/01_obs.png, 01_goal.png; all 720 x 480 RGB) as uint8 arrays,
so that the full-size guided fixtures (tools/make_fullsize_golden.py) and the GPU tests — which cannot read /root/reference — see
the same pixels.  Input DATA of the reference's demo commands (README: "Action-conditioned video prediction", "Goal-conditioned
visual planning"), not code.  Run in the build container:  python tools/make_named_inputs.py"""
import os

import numpy as np
import PIL.Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = {"car": "example_obs/car.png", "obs01": "example_obs_goal/01_obs.png", "goal01": "example_obs_goal/01_goal.png"}

if __name__ == "__main__":
    arrays = {k: np.asarray(PIL.Image.open(os.path.join("/root/reference/assets", v)).convert("RGB")) for k, v in SRC.items()}
    for k, a in arrays.items():
        assert a.shape == (480, 720, 3) and a.dtype == np.uint8, (k, a.shape)
    out = os.path.join(ROOT, "tests", "golden", "named_inputs.npz")
    np.savez_compressed(out, **arrays)
    print(out, os.path.getsize(out), "bytes")
