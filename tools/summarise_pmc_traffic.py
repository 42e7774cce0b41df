#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes of bench.py (FETCH_SIZE, WRITE_SIZE; separate runs, MI355X_MICROARCH.md)
into the per-kernel HBM-traffic JSON that bench.py's `roofline.traffic` cites.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d A -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d B -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
    python tools/summarise_pmc_traffic.py A/p_counter_collection.csv B/p_counter_collection.csv out.json
"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"commands": ["rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline",
                    "rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"],
       "note": "FETCH_SIZE / WRITE_SIZE are in KiB. gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports exactly 1/2 of "
               "the bytes of wide coalesced streaming reads -> read bytes = FETCH_SIZE*1024*2; WRITE_SIZE is exact -> write bytes = "
               "WRITE_SIZE*1024. Separate --pmc passes (TCC slots).",
       "workload": {"n_hap": 465, "window_sites": 50000, "windows_per_gpu": 4854}, "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    f = sum(fetch[k]) / len(fetch[k]) if fetch.get(k) else 0.0
    w = sum(write[k]) / len(write[k]) if write.get(k) else 0.0
    short = k.split("(")[0].replace("void ", "")
    out["kernels"][short] = {"FETCH_SIZE_avg_KiB": f, "FETCH_SIZE_dispatches": len(fetch.get(k, [])), "WRITE_SIZE_avg_KiB": w,
                             "WRITE_SIZE_dispatches": len(write.get(k, [])), "hbm_read_bytes_corrected": f * 1024 * 2,
                             "hbm_write_bytes": w * 1024, "hbm_bytes_per_launch": f * 1024 * 2 + w * 1024}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: v["hbm_bytes_per_launch"] for k, v in out["kernels"].items() if "scan_tiles" in k}))
