"""Per-SHAPE view of a rocprofv3 kernel trace: dispatches grouped by (kernel, grid size) — the per-kernel averages of `--stats` hide that one
instantiation serves launches of very different sizes (the VAE's tile batches of 4 / 2 / 2 / 1) — plus the idle time between consecutive dispatches
(meaningful for a ONE-lane plan, where nothing overlaps).

Usage: python tools/summarize_trace_by_grid.py gpurun_out/<name>/trace profiles/<out>.md [needle ...]
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "")
    if "aether::" in name:
        name = name[name.index("aether::"):]
    return name[:78]


def main(src, dst, needles):
    files = glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True)
    assert files, f"no *_kernel_trace.csv under {src}"
    rows = []
    with open(files[0]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1),
                         int(r["Workgroup_Size_X"])))
    rows.sort()
    groups = defaultdict(lambda: [0, 0.0])
    busy, gaps, big_gaps = 0.0, 0.0, 0
    for i, (s, e, k, g, w) in enumerate(rows):
        a = groups[(k, g // max(w, 1))]
        a[0] += 1
        a[1] += (e - s) / 1e3
        busy += (e - s) / 1e3
        if i:
            gap = (s - rows[i - 1][1]) / 1e3
            if 0 < gap < 200:                    # the pauses between the timed repetitions are not launch gaps
                gaps += gap
            elif gap >= 200:
                big_gaps += 1
    with open(dst, "w") as f:
        f.write(f"# kernel trace by (kernel, workgroups per launch): {src}\n\n{len(rows)} dispatches, {busy / 1e3:.1f} ms inside kernels, {gaps / 1e3:.1f} ms between consecutive "
                f"dispatches (gaps < 200 us; {big_gaps} longer pauses excluded)\n\n| kernel | workgroups | launches | total ms | avg us |\n|---|---|---|---|---|\n")
        for (k, g), (n, us) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
            if needles and not any(x in k for x in needles):
                continue
            if us / 1e3 < 0.3:
                continue
            f.write(f"| `{k}` | {g} | {n} | {us / 1e3:.2f} | {us / n:.1f} |\n")
    print(open(dst).read()[:6000])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3:])
