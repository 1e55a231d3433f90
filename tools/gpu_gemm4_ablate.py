"""Ablation of the four-wave GEMM loop (tools/probes/gemm4_ablate.hip): the main loop with the MFMAs (1), the LDS-DMA (2), the
fragment reads (4) dropped, the LDS-DMA as plain register loads (8), or every workgroup streaming tile (0,0) (32); always with the
in-kernel timers (16): shader clock, share of the loop spent in s_waitcnt / s_barrier.  Results are wrong by construction, only the
times matter.  --pads: leading dimension of A and W = K + pad elements (L2 channel spread of the row stride)."""
import ctypes, json, os, subprocess, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(here))
from aether_amd import ops
so, src = os.path.join(here, "probes", "gemm4_ablate.so"), os.path.join(here, "probes", "gemm4_ablate.hip")
hdr = os.path.join(os.path.dirname(here), "aether_amd", "csrc", "gemm4_kernel.hpp")
if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", src, "-o", so])
lib = ctypes.CDLL(so)
lib.run_gemm4_ablate.argtypes = [ctypes.c_int] + [ctypes.c_void_p, ctypes.c_int] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 3
dev = torch.device("cuda:0")
def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
out = []
ABLS = (16, 21, 22) if "--quick" in sys.argv else (16, 17, 18, 19, 20, 21, 22, 23, 24, 28, 29, 48, 53)
PADS = (0, 64, 192, 576) if "--pads" in sys.argv else (0,)
DROPS = ((1, "mfma"), (2, "dma"), (4, "ldsread"), (8, "lds-dma->vgpr-loads"), (32, "distinct tiles (all WGs stream tile 0,0)"))
for name, (M, N, K) in {"8192_cube": (8192, 8192, 8192), "ff1": (15076, 12288, 3072), "ff2": (15076, 3072, 12288)}.items():
    for pad in PADS:
        A = torch.randn(M, K + pad, device=dev).bfloat16()[:, :K]; W = (torch.randn(N, K + pad, device=dev) * K ** -0.5).bfloat16()[:, :K]
        tm = torch.zeros(2048, dtype=torch.float32, device=dev)
        bias = torch.randn(N, device=dev); C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        tiles = ((M + 255) // 256) * ((N + 255) // 256); nk = K // 64
        rounds = -(-tiles // 256)
        for abl in ABLS:
            def run():
                rc = lib.run_gemm4_ablate(abl, A.data_ptr(), A.stride(0), W.data_ptr(), W.stride(0), C.data_ptr(), C.stride(0), M, N, K, bias.data_ptr(),
                                          tm.data_ptr(), None)
                assert rc == 0, rc
            ms = sorted(timeit(run) for _ in range(3))[1]
            t = tm[:512].view(256, 2).double().cpu(); w = tm[512:1024].view(256, 2).double().cpu()
            r = {"shape": name, "ld_pad": pad, "kernel": "four-wave", "drop": "+".join(n for b, n in DROPS if abl & b) or "none",
                 "ms": round(ms, 4), "TF_equiv": round(2 * M * N * K / ms / 1e9, 1), "us_per_64k_tile_round": round(ms * 1e3 / rounds / nk, 4),
                 "GHz": round(float((t[:, 0] / (t[:, 1] * 10.0)).median()), 3), "wait_frac": round(float((w[:, 0] / t[:, 0]).median()), 3),
                 "barrier_frac": round(float((w[:, 1] / t[:, 0]).median()), 3)}
            print(r); out.append(r)
        for flags, kname in ((5, "ping-pong (product default)"), (1029, "four-wave (product, flag 1024)")):
            ms = sorted(timeit(lambda: ops.gemm_bf16(A, W, bias, ops.AETHER_EPI_BIAS, out=C, flags=flags)) for _ in range(3))[1]
            r = {"shape": name, "ld_pad": pad, "kernel": kname, "drop": "none", "ms": round(ms, 4), "TF_equiv": round(2 * M * N * K / ms / 1e9, 1)}
            print(r); out.append(r)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/gemm4_ablate.json", "w"), indent=1)
