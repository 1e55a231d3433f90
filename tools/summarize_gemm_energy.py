"""profiles/r06_gemm_energy.txt: per GEMM main-loop variant and shape — us (rocprofv3 kernel trace), cycles per XCD (GRBM_GUI_ACTIVE / 8), effective GHz
(= cycles per XCD / duration), LDS / VMEM / MFMA instructions, MFMA-pipe busy share — from gpurun_out/<dir>/{trace,pmc}_<variant> (tools/gpu_r06_gemm_energy.sh)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
variants = ["shipped", "ksps2", "snake", "nt_outer", "noprio", "vendor"]
shapes = ["qkv 15076x9216x3072", "ff1 15076x12288x3072 (+GELU)"]
flops = [2.0 * 15076 * 9216 * 3072, 2.0 * 15076 * 12288 * 3072]


def is_gemm(name, v):
    return ("Cijk" in name) if v == "vendor" else ("gemm_bf16_kernel" in name)


rows = []
for v in variants:
    tr = glob.glob(os.path.join(src, f"trace_{v}", "**", "*_kernel_trace.csv"), recursive=True)
    pm = glob.glob(os.path.join(src, f"pmc_{v}", "**", "*_counter_collection.csv"), recursive=True)
    if not tr or not pm:
        continue
    durs = []
    with open(tr[0]) as f:
        for r in sorted(csv.DictReader(f), key=lambda r: int(r["Start_Timestamp"])):
            if is_gemm(r["Kernel_Name"], v):
                durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    per_dispatch = defaultdict(dict)
    with open(pm[0]) as f:
        for r in csv.DictReader(f):
            if is_gemm(r["Kernel_Name"], v):
                per_dispatch[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    disp = [per_dispatch[k] for k in sorted(per_dispatch)]
    n = len(durs) // 2                        # first half of the dispatches: qkv, second half: ff1 (4 runs each)
    m = len(disp) // 2
    for si in range(2):
        d = sorted(durs[si * n:(si + 1) * n])[n // 2] if n else float("nan")
        cs = disp[si * m:(si + 1) * m][-1] if m else {}
        cyc = cs.get("GRBM_GUI_ACTIVE", float("nan")) / 8
        busy = cs.get("SQ_VALU_MFMA_BUSY_CYCLES", float("nan")) / max(cs.get("SQ_BUSY_CU_CYCLES", float("nan")), 1) / 4
        rows.append(dict(variant=v, shape=shapes[si], us=round(d, 1), TFLOPs=round(flops[si] / d / 1e6, 1), cycles_per_xcd=round(cyc), GHz=round(cyc / d / 1e3, 3),
                         lds_insts=cs.get("SQ_INSTS_LDS"), vmem_insts=cs.get("SQ_INSTS_VMEM"), mfma_insts=cs.get("SQ_INSTS_MFMA"), mfma_busy=round(busy, 3)))
with open(dst, "w") as f:
    f.write("# GEMM main-loop variants at the DiT shapes: duration, cycles, effective clock, instruction counts (tools/gpu_r06_gemm_energy.sh; one MI355X, one lease)\n")
    f.write("# variant: shipped = ping-pong loop, KSPS 1, mt-outer MFMA order, s_setprio on, LDS-staged epilogue; each other row changes ONE knob; vendor = the GEMM behind F.linear\n")
    f.write("# (under rocprofv3 the kernels run back to back with idle gaps: clocks are a little higher than in a 42-layer step)\n\n")
    f.write("| variant | shape | us | TFLOP/s | cycles/XCD | GHz | LDS insts | VMEM insts | MFMA insts | MFMA busy |\n|---|---|---|---|---|---|---|---|---|---|\n")
    for r in rows:
        f.write(f"| {r['variant']} | {r['shape']} | {r['us']} | {r['TFLOPs']} | {r['cycles_per_xcd']:.4g} | {r['GHz']} | {r['lds_insts']:.4g} | {r['vmem_insts']:.4g} | {r['mfma_insts']:.4g} | {r['mfma_busy']} |\n")
json.dump(rows, open(os.path.splitext(dst)[0] + ".json", "w"), indent=1)
print(open(dst).read())
