#!/usr/bin/env python3
"""Per-dispatch table from a rocprofv3 --pmc counter_collection.csv:
duration, effective clock (GRBM_GUI_ACTIVE / 8 XCDs / duration) and MFMA-pipe
utilisation (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles)).
    python tools/pmc_dispatch_table.py <csv> [name-filter]"""
import collections
import csv
import sys

rows = collections.OrderedDict()
flt = sys.argv[2] if len(sys.argv) > 2 else "gemm"
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if flt not in r["Kernel_Name"]:
            continue
        d = rows.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"][:40],
                                                    "dur": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
print(f"{'id':>5} {'kernel':40s} {'us':>8} {'GHz':>6} {'mfma%':>6} {'wait_any%':>9} {'wait_inst%':>10} {'valu%':>6}")
for i, d in rows.items():
    cyc = d.get("GRBM_GUI_ACTIVE", 0) / 8.0
    ghz = cyc / (d["dur"] * 1e3) if d["dur"] > 0 else 0
    mf = 100.0 * d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024.0 * cyc) if cyc else 0
    wc = d.get("SQ_WAVE_CYCLES", 0)
    wa = 100.0 * d.get("SQ_WAIT_ANY", 0) / wc if wc else 0
    wi = 100.0 * d.get("SQ_WAIT_INST_ANY", 0) / wc if wc else 0
    va = 100.0 * d.get("SQ_ACTIVE_INST_VALU", 0) / wc if wc else 0
    print(f"{i:5d} {d['name']:40s} {d['dur']:8.1f} {ghz:6.3f} {mf:6.1f} {wa:9.1f} {wi:10.1f} {va:6.1f}")
