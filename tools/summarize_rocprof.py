"""Condense rocprofv3 CSV output (gpurun_out/prof/...) into a small JSON + markdown summary for profiles/.

Usage: python tools/summarize_rocprof.py gpurun_out/prof profiles/r01_dit_step
Reads   <dir>/trace/**/*_kernel_stats.csv        (rocprofv3 --kernel-trace --stats)
        <dir>/pmc_*/**/*_counter_collection.csv  (one rocprofv3 --pmc pass each)
HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are in KiB-like units of
the TCC EA request counters (bytes = value * 1024 as in the guide's recipe) and, on gfx950, FETCH_SIZE under-counts a
wide coalesced streaming read by exactly 2x -> the read side is doubled.  WRITE_SIZE is uncalibrated (reported raw).
"""
from __future__ import annotations

import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = name.replace("void ", "")
    for pre in ("aether::",):
        if pre in name:
            name = name[name.index(pre):]
            break
    return name[:90]


def main(src: str, dst_prefix: str):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from aether_amd.build import source_digest
    out = {"source": src, "csrc_sha16": source_digest(), "kernel_stats": [], "pmc": {}}
    stats = glob.glob(os.path.join(src, "trace", "**", "*_kernel_stats.csv"), recursive=True)
    if stats:
        with open(stats[0]) as f:
            for row in csv.DictReader(f):
                out["kernel_stats"].append({"kernel": short(row["Name"]), "calls": int(row["Calls"]),
                                            "total_ms": float(row["TotalDurationNs"]) / 1e6,
                                            "avg_us": float(row["AverageNs"]) / 1e3, "pct": float(row["Percentage"]),
                                            "min_us": float(row["MinNs"]) / 1e3, "max_us": float(row["MaxNs"]) / 1e3})
    for ccsv in glob.glob(os.path.join(src, "pmc_*", "**", "*_counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
        regs = {}
        with open(ccsv) as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                if "aether::" not in k:
                    continue
                a = acc[k][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
                regs[k] = {"vgpr": int(row["VGPR_Count"]), "agpr": int(row["Accum_VGPR_Count"]), "sgpr": int(row["SGPR_Count"]),
                           "lds": int(row["LDS_Block_Size"]), "wg": int(row["Workgroup_Size"]), "grid": int(row["Grid_Size"])}
        for k, counters in acc.items():
            e = out["pmc"].setdefault(k, {"resources": regs[k]})
            for cname, (tot, n) in counters.items():
                e[cname + "_avg_per_launch"] = tot / n
                e.setdefault("launches", n)
    for k, e in out["pmc"].items():
        if "FETCH_SIZE_avg_per_launch" in e:
            e["hbm_read_bytes_per_launch_corrected"] = e["FETCH_SIZE_avg_per_launch"] * 1024 * 2
        if "WRITE_SIZE_avg_per_launch" in e:
            e["hbm_write_bytes_per_launch_raw"] = e["WRITE_SIZE_avg_per_launch"] * 1024
        if "TCC_HIT_sum_avg_per_launch" in e and "TCC_MISS_sum_avg_per_launch" in e:
            e["l2_hit_rate"] = e["TCC_HIT_sum_avg_per_launch"] / max(e["TCC_HIT_sum_avg_per_launch"] + e["TCC_MISS_sum_avg_per_launch"], 1.0)
    os.makedirs(os.path.dirname(dst_prefix) or ".", exist_ok=True)
    with open(dst_prefix + ".json", "w") as f:
        json.dump(out, f, indent=1)
    with open(dst_prefix + ".md", "w") as f:
        f.write(f"# rocprofv3 summary ({src})\n\n## kernel trace (--kernel-trace --stats)\n\n")
        f.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
        for r in out["kernel_stats"][:16]:
            f.write(f"| `{r['kernel']}` | {r['calls']} | {r['total_ms']:.2f} | {r['avg_us']:.1f} | {r['pct']:.2f} |\n")
        f.write("\n## PMC (separate passes), averages per launch\n\n")
        for k, e in out["pmc"].items():
            f.write(f"* `{k}`: " + ", ".join(f"{a}={b:.4g}" if isinstance(b, float) else f"{a}={b}" for a, b in e.items()) + "\n")
    print("wrote", dst_prefix + ".json/.md")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
