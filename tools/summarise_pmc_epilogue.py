#!/usr/bin/env python3
"""Per-launch averages of the PMC groups tools/prof_epilogue_pmc.sh collected, for the epilogue kernels.
   python tools/summarise_pmc_epilogue.py gpurun_out/epi_pmc"""
import collections
import csv
import glob
import sys

KERNELS = ("pica2_small_kernel", "hfst_small_kernel", "pica2_kernel", "hfst_kernel", "gram_fp4_kernel")
rows = collections.defaultdict(lambda: collections.defaultdict(dict))  # kernel -> launch ordinal -> counter -> value
for f in sorted(glob.glob(sys.argv[1] + "/g*/**/p_counter_collection.csv", recursive=True)):
    seen = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        k = next((k for k in KERNELS if k + "(" in r["Kernel_Name"] or k + "<" in r["Kernel_Name"]), None)
        if not k:
            continue
        d = seen[k].setdefault(r["Dispatch_Id"], len(seen[k]))
        e = rows[k][d]
        e[r["Counter_Name"]] = float(r["Counter_Value"])
        e["us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        e["vgpr"] = r.get("VGPR_Count"); e["lds"] = r.get("LDS_Block_Size"); e["grid"] = r.get("Grid_Size"); e["wg"] = r.get("Workgroup_Size")
for k in KERNELS:
    for d, e in sorted(rows[k].items()):
        print(k, "launch", d, " ".join(f"{n}={v:.6g}" if isinstance(v, float) else f"{n}={v}" for n, v in e.items()))
