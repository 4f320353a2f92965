#!/usr/bin/env python3
"""Dev: launch-ordered durations of the kernels whose demangled name contains a substring.   usage: tools/rocpd_list.py <trace.db> <substring> [max]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import rocpd_tools as r
db, T = r._open(sys.argv[1])
names = {x[0]: x[1] for x in db.execute(f'select id, kernel_name from "{T("info_kernel_symbol")}"')}
dm = r._demangle(sorted(set(names.values())))
rows = list(db.execute(f'select start, end - start, kernel_id, grid_size_x, workgroup_size_x from "{T("kernel_dispatch")}" order by start'))
n = 0
for st, dur, kid, gx, wx in rows:
    k = dm.get(names.get(kid, "?"), "?")
    if sys.argv[2] in k:
        print(f"{dur / 1e3:9.1f} us  [{gx // max(wx, 1)} x {wx}]  {k[:100]}")
        n += 1
        if len(sys.argv) > 3 and n >= int(sys.argv[3]):
            break
