"""Bare MFMA streams under the power cap (tools/probes/mfma_power_probe.hip): TF/s, shader clock and cycles per MFMA for
32x32x16 and 16x16x32 bf16, random and all-zero operands.  Measurement only."""
import ctypes, json, os, subprocess, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
so, src = os.path.join(here, "probes", "mfma_power_probe.so"), os.path.join(here, "probes", "mfma_power_probe.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", src, "-o", so])
lib = ctypes.CDLL(so)
lib.run_mfma_probe.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
out = torch.zeros(1024, dtype=torch.float32, device=dev); sink = torch.zeros(4, dtype=torch.float32, device=dev)
res = []
for dname, data in (("random N(0,1)", torch.randn(32 * 64 * 8, device=dev).bfloat16()), ("zeros", torch.zeros(32 * 64 * 8, device=dev).bfloat16())):
    for shape, sname, nw, mfma_per_it, flop in ((0, "32x32x16, 128x128 tile, 1 wave/SIMD", 4, 32, 32768), (1, "16x16x32, 128x128 tile, 1 wave/SIMD", 4, 128, 16384),
                                               (2, "32x32x16, 128x64 tile, 2 waves/SIMD", 8, 16, 32768),
                                               (3, "32x32x16, 128x128 tile, snake order (one operand changes per MFMA)", 4, 32, 32768),
                                               (4, "32x32x16, 128x128 tile, the same operand pair for every MFMA", 4, 32, 32768),
                                               (5, "32x32x16, 128x128 tile, throttled: s_sleep 6 per 32 MFMAs", 4, 32, 32768),
                                               (6, "32x32x16, 128x128 tile, throttled: s_sleep 4 per 32 MFMAs", 4, 32, 32768)):
        iters = 40000 if shape != 1 else 10000
        iters = iters if shape != 2 else 40000
        for _ in range(2):
            rc = lib.run_mfma_probe(shape, nw, data.data_ptr(), iters, out.data_ptr(), sink.data_ptr(), 256, None); assert rc == 0
            torch.cuda.synchronize()
        t = out[:512].view(256, 2).double().cpu()
        cyc, ns = float(t[:, 0].median()), float(t[:, 1].median()) * 10.0
        n_mfma = iters * mfma_per_it
        r = {"operands": dname, "stream": sname, "TFps": round(256 * nw * n_mfma * flop / ns / 1e3, 1), "GHz": round(cyc / ns, 3),
             "cycles_per_mfma_per_simd": round(cyc / (n_mfma * (nw / 4)), 2), "ms": round(ns / 1e6, 2)}
        print(r); res.append(r)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/mfma_power_probe.json", "w"), indent=1)
