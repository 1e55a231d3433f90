"""Per-shape summary of tools/profile_vendor_gemm.sh: the vendor kernel and ours, dispatches grouped in launch order (33 per shape)."""
import collections, csv, glob, os, sys
src = sys.argv[1]
SHAPES = ["qkv", "out", "ff1", "ff2", "cube"]
OURS = {"1048576": "qkv main", "116736": "qkv tail", "362496": "out+ff2", "1441792": "ff1 main", "98304": "ff1 tail", "524288": "cube"}
for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    d = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    v = [r for r in rows if "Cijk" in r["Kernel_Name"]]
    if v:
        r = v[0]
        print("vendor kernel:", r["Kernel_Name"][:160], "vgpr", r["VGPR_Count"], "agpr", r["Accum_VGPR_Count"], "lds", r["LDS_Block_Size"], "wg", r["Workgroup_Size_X"])
    for i, n in enumerate(SHAPES):
        s = sorted(d(r) for r in v[i * 33:(i + 1) * 33])
        if s: print(f"trace vendor {n:5s} grid {v[i * 33]['Grid_Size_X']:>8s} median {s[len(s) // 2]:8.1f} us")
    g = collections.defaultdict(list)
    for r in rows:
        if "gemm_bf16_kernel" in r["Kernel_Name"]: g[r["Grid_Size_X"]].append(d(r))
    for k, vv in g.items(): print(f"trace ours   {OURS.get(k, k):9s} median {sorted(vv)[len(vv) // 2]:8.1f} us")
for p in ("pmcA", "pmcB"):
    for f in glob.glob(os.path.join(src, p, "**", "*counter_collection.csv"), recursive=True):
        disp = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "Cijk" in k or "gemm_bf16_kernel" in k:
                disp.setdefault((r["Dispatch_Id"], "vendor" if "Cijk" in k else "ours", r["Grid_Size"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
        def show(tag, grp):
            avg = {c: sum(x[c] for x in grp) / len(grp) for c in grp[0]}
            cyc = avg.get("GRBM_GUI_ACTIVE", 0) / 8
            s = f"{p} {tag:16s} cycles/XCD {cyc:10.4g}"
            if "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
                s += f"  mfma_util {avg['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc:.3f}  lds_active/cyc/CU {avg['SQ_LDS_IDX_ACTIVE'] / 256 / cyc:.3f}  vmem {avg['SQ_INSTS_VMEM']:.4g} lds_insts {avg['SQ_INSTS_LDS']:.4g} mfma_insts {avg['SQ_INSTS_MFMA']:.4g} waves {avg['SQ_WAVES']:.0f}"
            if "TCC_HIT_sum" in avg:
                s += f"  l2_hit {avg['TCC_HIT_sum'] / (avg['TCC_HIT_sum'] + avg['TCC_MISS_sum']):.3f}  tcc_req {avg['TCC_HIT_sum'] + avg['TCC_MISS_sum']:.4g}  miss_MB(128 B each) {avg['TCC_MISS_sum'] * 128 / 1e6:.1f}"
            print(s)
        ven = [v for k, v in disp.items() if k[1] == "vendor"]
        for i, n in enumerate(SHAPES):
            if ven[i * 33:(i + 1) * 33]: show("vendor " + n, ven[i * 33:(i + 1) * 33])
        ours = collections.defaultdict(list)
        for k, v in disp.items():
            if k[1] == "ours": ours[k[2]].append(v)
        for gsz, grp in ours.items(): show("ours " + OURS.get(gsz, gsz), grp)
