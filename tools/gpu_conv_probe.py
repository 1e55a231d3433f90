"""Where does the implicit-GEMM convolution lose time?  Times the decoder's full-resolution 128->128 causal conv
(4 tiles x 8 frames x 240 x 360 voxels, K = 27 x 128) and the 256->256 level with (a) the real tap table, (b) every tap
pointing at tap 0 (same arithmetic, A tile always cache resident: upper bound of what better A-operand locality / tap
reuse could give).  Run on the MI355X through gpurun; prints TFLOP/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aether_amd import _lib  # noqa: E402
from aether_amd.vae import AetherVAE, _Conv  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    vae = AetherVAE(device=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    for (NB, T, H, W, C, Cout) in ((4, 8, 240, 360, 128, 128), (4, 8, 120, 180, 256, 256), (4, 3, 60, 90, 512, 512)):
        vol = (torch.randn(NB, T + 2, H + 2, W + 2, C, generator=g, device=dev) * 0.5).to(torch.bfloat16)
        w = (torch.randn(Cout, C, 3, 3, 3, generator=g, device=dev) * (27 * C) ** -0.5).to(torch.bfloat16)
        conv = _Conv(w.cpu(), torch.zeros(Cout), dev)
        flop = 2.0 * NB * T * H * W * Cout * 27 * C
        for mode in ("real taps", "all taps -> tap 0"):
            if mode != "real taps":
                key = (3, 3, 3, H + 2, W + 2, C)
                vae._tap_table(*key)
                t = vae._taps[key]
                vae._taps[key] = (t % 64 if False else torch.tensor([cb * 64 for _ in range(27) for cb in range(C // 64)], dtype=torch.int32, device=dev))
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
            print({"shape": [NB, T, H, W, C, Cout], "mode": mode, "ms": round(ms, 3), "TFLOPs": round(flop / ms / 1e9, 1)}, flush=True)
        vae._taps.clear()
        del vol, out


if __name__ == "__main__":
    main()
