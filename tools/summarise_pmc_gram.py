#!/usr/bin/env python3
"""rocprofv3 PMC passes of tools/bench_pairwise.py -> per-launch summary of the Gram kernel.

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE \\
        --kernel-trace --output-format csv -d A -o p -- python3 tools/bench_pairwise.py --windows 4096 --big-sites 200000
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d B -o p -- python3 tools/bench_pairwise.py --windows 4096 --big-sites 200000
    python tools/summarise_pmc_gram.py A/p_counter_collection.csv B/p_counter_collection.csv out.json [windows_per_launch window_bytes]

MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8 XCDs) x 1024 SIMDs); clock = GRBM_GUI_ACTIVE / 8 / duration;
HBM read bytes = FETCH_SIZE x 1024 x 2 (gfx950 correction for wide coalesced reads, MI355X_MICROARCH.md)."""
import collections
import csv
import json
import sys


def dispatches(path, kernel):
    d = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if kernel in r["Kernel_Name"]:
            e = d.setdefault(r["Dispatch_Id"], {"duration_us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                                "vgpr": int(r["VGPR_Count"]), "grid": int(r["Grid_Size"])})
            e[r["Counter_Name"]] = float(r["Counter_Value"])
    return list(d.values())


mf = dispatches(sys.argv[1], "gram_fp4_kernel")
fe = dispatches(sys.argv[2], "gram_fp4_kernel")
nw = int(sys.argv[4]) if len(sys.argv) > 4 else None
wb = float(sys.argv[5]) if len(sys.argv) > 5 else None
out = {"kernel": "gram_fp4_kernel", "note": __doc__.split("MfmaUtil", 1)[1].strip().replace("\n", " ").join(["MfmaUtil ", ""]), "mfma": [], "hbm": []}
for e in mf:
    cyc = e["GRBM_GUI_ACTIVE"] / 8
    out["mfma"].append({"duration_us": e["duration_us"], "MfmaUtil_pct": 100 * e["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024),
                        "clock_GHz": cyc / e["duration_us"] / 1e3,
                        "resident_waves_per_SIMD": e["SQ_WAVE_CYCLES"] / (cyc * 1024 / 4) / 4 if False else e["SQ_WAVE_CYCLES"] / (cyc * 1024),
                        "wait_any_frac": e["SQ_WAIT_ANY"] / e["SQ_WAVE_CYCLES"], "wait_inst_any_frac": e["SQ_WAIT_INST_ANY"] / e["SQ_WAVE_CYCLES"],
                        "active_valu_frac": e["SQ_ACTIVE_INST_VALU"] / e["SQ_WAVE_CYCLES"], "vgpr": e["vgpr"]})
for e in fe:
    rd = e["FETCH_SIZE"] * 1024 * 2
    row = {"duration_us": e["duration_us"], "FETCH_SIZE_KiB": e["FETCH_SIZE"], "hbm_read_bytes_corrected": rd,
           "hbm_read_TBps": rd / e["duration_us"] / 1e6}
    if nw and wb and e["duration_us"] > 4000:
        row["hbm_read_bytes_per_window"] = rd / nw
        row["x_window_bytes"] = rd / nw / wb
    out["hbm"].append(row)
json.dump(out, open(sys.argv[3], "w"), indent=1)
big = [m for m in out["mfma"] if m["duration_us"] > 4000]
print(json.dumps({"MfmaUtil_pct": [round(m["MfmaUtil_pct"], 1) for m in big], "clock_GHz": [round(m["clock_GHz"], 2) for m in big],
                  "hbm": [{k: (round(v, 3) if isinstance(v, float) else v) for k, v in h.items()} for h in out["hbm"] if h["duration_us"] > 4000]}))
