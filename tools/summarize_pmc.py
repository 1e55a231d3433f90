"""Average per-launch PMC counters of the aether:: kernels found under <dir>/pmc*/ (rocprofv3 counter_collection csv)."""
import collections
import csv
import glob
import os
import sys

src = sys.argv[1]
for d in sorted(glob.glob(os.path.join(src, "pmc*"))):
    if not os.path.isdir(d):
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "aether::" not in k:
                continue
            a = acc[k[k.index("aether::"):][:70]][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
    for k, cs in acc.items():
        print(os.path.basename(d), k)
        for c, (tot, n) in sorted(cs.items()):
            print(f"    {c:28s} {tot / n:16.4g}  (x{n})")
