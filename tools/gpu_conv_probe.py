"""Implicit-GEMM convolution at the VAE's real layer shapes: the plain gathered kernel (gemm_kernel.hpp, one staged A tile
per K tile) against the tap-reuse kernel (conv3_kernel.hpp, one staged A tile per three dw taps).  TFLOP/s count the useful
outputs only (the tap-reuse kernel also computes the dropped border rows).  Run on the MI355X through gpurun."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aether_amd.vae import AetherVAE, _Conv  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    vae = AetherVAE(device=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    shapes = ((4, 8, 240, 360, 128, 128, 3), (4, 8, 240, 360, 256, 128, 3), (4, 8, 240, 360, 256, 256, 1), (4, 8, 120, 180, 256, 256, 3),
              (1, 8, 240, 360, 128, 128, 3), (4, 4, 60, 90, 512, 256, 3), (4, 3, 30, 45, 512, 512, 3))
    for (NB, T, H, W, C, Cout, kt) in shapes:
        vol = (torch.randn(NB, T + (2 if kt == 3 else 0), H + 2, W + 2, C, generator=g, device=dev) * 0.5).to(torch.bfloat16)
        wshape = (Cout, C, 3, 3, 3) if kt == 3 else (Cout, C, 3, 3)
        conv = _Conv((torch.randn(wshape, generator=g, device=dev) * (9 * kt * C) ** -0.5).cpu(), torch.zeros(Cout), dev)
        flop = 2.0 * NB * T * H * W * Cout * 9 * kt * C
        row = {"shape": [NB, T, H, W, C, Cout], "kt": kt}
        ref = None
        for mode, waste in (("plain", 0.0), ("tap_reuse", 100.0)):
            vae.tap_reuse_max_waste = waste
            for _ in range(2):
                out = vae._conv(vol, conv, (T, H, W), 1, None)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5):
                out = vae._conv(vol, conv, (T, H, W), 1, None)
            e.record()
            torch.cuda.synchronize()
            ms = s.elapsed_time(e) / 5
            row[mode + "_ms"], row[mode + "_TFLOPs"] = round(ms, 3), round(flop / ms / 1e9, 1)
            if ref is None:
                ref = out.clone()
            else:
                row["max_abs_diff"] = float((out.float() - ref.float()).abs().max())
        print(json.dumps(row), flush=True)
        del vol, out, ref


if __name__ == "__main__":
    main()
