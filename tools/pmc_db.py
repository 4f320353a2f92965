#!/usr/bin/env python3
"""Dev: per-kernel means of the counters in a rocprofv3 --pmc rocpd database (pmc_results.db)."""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
T = lambda stem: [t for t in tabs if t.startswith("rocpd_" + stem)][0]
cols = lambda t: [r[1] for r in db.execute(f'pragma table_info("{t}")')]
pmc, ev, disp, sym, info = T("pmc_event"), T("event"), T("kernel_dispatch"), T("info_kernel_symbol"), T("info_pmc")
if "-v" in sys.argv:
    for t in (pmc, ev, disp, sym, info):
        print(t, cols(t))
names = {r[0]: r[1] for r in db.execute(f'select id, kernel_name from "{sym}"')}
cname = {r[0]: r[1] for r in db.execute(f'select id, name from "{info}"')}
d2k = {}
for r in db.execute(f'select event_id, kernel_id, end - start from "{disp}"'):
    d2k[r[0]] = (names.get(r[1], "?"), r[2])
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
seen = set()
for eid, pid, val in db.execute(f'select event_id, pmc_id, value from "{pmc}"'):
    if eid not in d2k:
        continue
    k, d = d2k[eid]
    acc[k][cname.get(pid, str(pid))].append((eid, val))
    if eid not in seen:
        seen.add(eid)
        dur[k].append(d)
for k, cs in acc.items():
    if "sa::" not in k and "_ZN2sa" not in k and "-a" not in sys.argv:
        continue
    print(k[:110], f"launches={len(dur[k])} avg_us={sum(dur[k]) / len(dur[k]) / 1e3:.1f}")
    per = {}
    for c, lst in cs.items():
        by = defaultdict(float)
        for eid, v in lst:
            by[eid] += v       # summed over dimensions (XCDs / SEs)
        per[c] = sum(by.values()) / len(by)
    wc = per.get("SQ_WAVE_CYCLES", 0) or 1.0
    for c, v in sorted(per.items()):
        print(f"    {c:28s} {v:18.1f}   /WAVE_CYCLES={v / wc:7.3f}")
