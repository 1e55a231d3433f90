"""Per-CU load throughput by instruction kind (tools/probes/load_probe.hip): bytes per shader cycle and GB/s one CU pulls from an
L2-resident (2 MiB) or L1-resident (16 KiB) region, one workgroup per CU, 4 / 8 / 16 waves.  Measurement only."""
import ctypes, json, os, subprocess, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "probes", "load_probe.so")
src = os.path.join(here, "probes", "load_probe.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", src, "-o", so])
lib = ctypes.CDLL(so)
lib.run_load_probe.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint, ctypes.c_int, ctypes.c_uint, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
buf = torch.randint(0, 255, (256 << 20,), dtype=torch.uint8, device=dev)
out = torch.zeros(1024, dtype=torch.float32, device=dev)
NAMES = {0: "global_load_dwordx4 -> VGPR", 1: "buffer_load_dwordx4 lds (LDS-DMA)", 2: "buffer_load_dwordx4 -> VGPR", 3: "global_load_dwordx2 -> VGPR",
         4: "buffer_load_dword lds (LDS-DMA 4 B/lane)", 5: "LDS-DMA x4, GEMM row pattern (8 rows x 128 B, stride 16 KiB)"}
res = []
for region, rname in ((16 << 10, "L1 16KiB"), (2 << 20, "L2 2MiB"), (256 << 20, "HBM/MALL 256MiB")):
    for mode in (0, 2, 1, 3, 4, 5):
        for nw in (4, 8, 16):
            bpl = 8 if mode == 3 else 4 if mode == 4 else 16
            iters = 2000 if region <= (2 << 20) else 400
            for _ in range(2):
                rc = lib.run_load_probe(mode, nw, buf.data_ptr(), region, iters, 16384, out.data_ptr(), 256, None)
                assert rc == 0, rc
                torch.cuda.synchronize()
            t = out[:512].view(256, 2).double().cpu()
            cyc, ns = float(t[:, 0].median()), float(t[:, 1].median()) * 10.0
            nbytes = iters * nw * 8 * 64 * bpl
            r = {"region": rname, "instr": NAMES[mode], "waves_per_cu": nw, "B_per_clk_per_CU": round(nbytes / cyc, 1), "GBps_per_CU": round(nbytes / ns, 1),
                 "chip_TBps": round(nbytes / ns * 256 / 1e3, 2), "GHz": round(cyc / ns, 2)}
            print(r); res.append(r)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/load_probe.json", "w"), indent=1)
