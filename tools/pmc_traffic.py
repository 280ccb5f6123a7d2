#!/usr/bin/env python3
"""Write profiles/pmc_traffic.json from the rocprofv3 --pmc passes of tools/profile_round.sh:

    python tools/pmc_traffic.py <counter_collection.csv ...> > profiles/pmc_traffic.json

Per GEMM class of the bench step (0 layer forward, 1 data gradient, 2 weight gradient; the kernel names are recorded):
  traffic   = (2 FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch  (FETCH_SIZE doubled: on gfx950 it tallies 128-byte requests
              as 64 bytes, MI355X_MICROARCH.md section HBM; both counters are in KiB)
  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)
and the hash of the library sources the profiled run was built from (bench.py: csrc_sha()).  bench.py reports the
numbers only while that hash equals the one of the tree it runs in."""
import collections
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FWD_EPI = {1, 4, 5, 6, 10}


def klass(name):
    m = re.search(r"gemm\w*_nt_kernel<(\d+)", name)
    if m:
        return 0 if int(m.group(1)) in FWD_EPI else 1
    if re.search(r"gemm\w*_tn\w*_kernel", name):
        return 2
    return None


def main():
    from bench import csrc_sha
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    names = collections.defaultdict(set)
    for path in sys.argv[1:]:
        with open(path) as f:
            for r in csv.DictReader(f):
                k = klass(r.get("Kernel_Name", ""))
                if k is None:
                    continue
                a = acc[k][r.get("Counter_Name", "")]
                a[0] += float(r.get("Counter_Value", 0) or 0)
                a[1] += 1
                names[k].add(re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", ""))
    out = {"_comment": " ".join(__doc__.split()), "csrc_sha": csrc_sha(), "traffic": {}, "mfma_busy": {},
           "kernels": {}, "launches_profiled": {}}
    for k in sorted(acc):
        c = {n: v[0] / v[1] for n, v in acc[k].items() if v[1]}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            out["traffic"][str(k)] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("GRBM_GUI_ACTIVE"):
            out["mfma_busy"][str(k)] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * c["GRBM_GUI_ACTIVE"] / 8)
        out["kernels"][str(k)] = sorted(names[k])
        out["launches_profiled"][str(k)] = max(v[1] for v in acc[k].values())
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
