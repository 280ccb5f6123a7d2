#!/usr/bin/env python3
"""Mean counter value per dispatch for every (kernel, counter) from rocprofv3 --pmc sqlite outputs (ROCm 7.2
writes rocpd .db files by default):  python tools/pmc_db_summary.py gpurun_out/pmc_x3/*/*/*.db [name-filter]"""
import collections
import sqlite3
import sys

paths = [p for p in sys.argv[1:] if p.endswith(".db")]
flt = [p for p in sys.argv[1:] if not p.endswith(".db")]
acc = collections.defaultdict(lambda: [0.0, 0])
dur = collections.defaultdict(lambda: [0.0, 0])
for path in paths:
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    t = lambda pre: [x for x in tabs if x.startswith(pre)][0]
    kd, ks, pe, pi = t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol"), t("rocpd_pmc_event"), t("rocpd_info_pmc")
    q = (f"select s.kernel_name, p.name, d.id, sum(e.value), d.end - d.start from {pe} e join {pi} p on e.pmc_id = p.id "
         f"join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id group by d.id, p.name")
    for name, ctr, did, val, dt in c.execute(q):
        if flt and not any(f in name for f in flt):
            continue
        a = acc[(name[:64], ctr)]
        a[0] += val
        a[1] += 1
        b = dur[name[:64]]
        b[0] += dt
        b[1] += 1
for k in sorted({k for k, _ in acc}):
    items = {c: a for (kk, c), a in acc.items() if kk == k}
    n = max(a[1] for a in items.values())
    print(f"{k}  (dispatches {n}, mean {dur[k][0] / dur[k][1] / 1e3:.1f} us under the profiler)")
    for c, a in sorted(items.items()):
        print(f"    {c:36s} {a[0] / a[1]:18.1f}")
