#!/usr/bin/env python3
"""Summaries of rocprofv3 rocpd databases (the sqlite files `rocprofv3 -d DIR -o NAME` leaves as DIR/NAME_results.db), dev tool.

    python tools/rocpd_tools.py stats   <trace.db> [--by-grid]             per-kernel launch count / total / avg / min / max / share
                                                                           (--by-grid: one row per (kernel, launch size): per-level numbers)
    python tools/rocpd_tools.py pmc     <pmc.db> [--by-grid]               per-kernel mean of every collected counter
    python tools/rocpd_tools.py traffic <fetch.db> <write.db> [out.json] [provenance] [--by-grid]   HBM-side bytes per launch: (2 x FETCH_SIZE + WRITE_SIZE) KiB
                                                                           (gfx950: 128-B read requests are tallied at 64 B, MI355X_MICROARCH.md)
Kernel names are demangled and trimmed to the spelling `sa_last_conv_kernel()` reports (e.g. `conv_fprop_halo256_kernel<unsigned short, true, 8>`)."""
import json
import re
import sqlite3
import subprocess
import sys
from collections import defaultdict


def _open(path):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    return db, (lambda stem: [t for t in tabs if t.startswith("rocpd_" + stem)][0])


def _demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(n[:-3] if n.endswith(".kd") else n for n in names), capture_output=True, text=True).stdout.split("\n")
    except OSError:
        out = list(names)
    res = {}
    for n, d in zip(names, out):
        d = re.sub(r"^void ", "", d)
        d = re.sub(r"\((sa::|HIP_|float|unsigned|long|int|void|at::|std::|c10::).*$", "", d) if "(" in d else d
        res[n] = d.replace("sa::", "").strip()
    return res


def _dispatches(db, T, by_grid=False):
    """event id -> (kernel name, duration ns).  by_grid: the name carries the launch size as ` [blocks x threads]`, so launches of one kernel instance on
    different problem sizes (80x112x80 vs 40x56x40 levels, training vs eval) are separate rows."""
    names = {r[0]: r[1] for r in db.execute(f'select id, kernel_name from "{T("info_kernel_symbol")}"')}
    dm = _demangle(sorted(set(names.values())))
    res = {}
    for ev, kid, dur, gx, gy, gz, wx, wy, wz in db.execute(
            f'select event_id, kernel_id, end - start, grid_size_x, grid_size_y, grid_size_z, workgroup_size_x, workgroup_size_y, workgroup_size_z from "{T("kernel_dispatch")}"'):
        k = dm[names.get(kid, "?")] if names.get(kid) in dm else "?"
        if by_grid:
            blocks = (gx // max(wx, 1)) * (gy // max(wy, 1)) * (gz // max(wz, 1))
            k = f"{k} [{blocks} x {wx * wy * wz}]"
        res[ev] = (k, dur)
    return res


def stats(path, by_grid=False):
    db, T = _open(path)
    acc = defaultdict(list)
    for k, d in _dispatches(db, T, by_grid).values():
        acc[k].append(d)
    tot = sum(sum(v) for v in acc.values())
    print(f"{'kernel':96s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        print(f"{k[:96]:96s} {len(v):7d} {sum(v) / 1e6:10.3f} {sum(v) / len(v) / 1e3:10.2f} {min(v) / 1e3:10.2f} {max(v) / 1e3:10.2f} {100 * sum(v) / tot:6.2f}")


def _counters(path, by_grid=False):
    db, T = _open(path)
    disp = _dispatches(db, T, by_grid)
    cname = {r[0]: r[1] for r in db.execute(f'select id, name from "{T("info_pmc")}"')}
    per = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))   # kernel -> counter -> event -> value summed over instances
    for eid, pid, val in db.execute(f'select event_id, pmc_id, value from "{T("pmc_event")}"'):
        if eid in disp:
            per[disp[eid][0]][cname.get(pid, str(pid))][eid] += val
    res = {k: {c: (sum(ev.values()) / len(ev), len(ev)) for c, ev in cs.items()} for k, cs in per.items()}
    dur = defaultdict(list)
    for k, d in disp.values():
        dur[k].append(d)
    for k in res:      # mean launch duration of the SAME pass (counter passes run slower than a plain trace)
        res[k]["_duration_ns"] = (sum(dur[k]) / len(dur[k]), len(dur[k]))
    return res


def pmc(path, by_grid=False):
    for k, cs in _counters(path, by_grid).items():
        print(k[:110])
        wc = cs.get("SQ_WAVE_CYCLES", (0, 0))[0] or 1.0
        dur = cs.pop("_duration_ns", (0.0, 0))[0]
        for c, (v, n) in sorted(cs.items()):
            print(f"    {c:28s} launches={n:4d} mean={v:18.1f}  /WAVE_CYCLES={v / wc:7.3f}")
        gui = cs.get("GRBM_GUI_ACTIVE", (0, 0))[0]
        if gui and dur >= 100e3:      # (short launches: the counter window is dominated by the dispatch itself)
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES over the 1 024 SIMDs
            line = f"    launch {dur / 1e3:.1f} us in this pass -> shader clock ~ {gui / 8 / dur:.2f} GHz"
            busy = cs.get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0))[0]
            if busy:
                line += f"; MFMA busy {busy / (gui / 8 * 1024):.3f} of the launch's cycles"
            print(line)


def traffic(fetch_db, write_db, out=None, provenance="", by_grid=False, sources="vqvae"):
    """by_grid: rows per (kernel, launch size); the JSON then ALSO keeps the per-kernel means under the plain names (what bench.py looks up)."""
    f, w = _counters(fetch_db, by_grid), _counters(write_db, by_grid)
    if by_grid:
        f0, w0 = _counters(fetch_db), _counters(write_db)
        for k in f0:
            f.setdefault(k, f0[k])
        for k in w0:
            w.setdefault(k, w0[k])
    rec = {}
    print(f"{'kernel':84s} {'launches':>8s} {'FETCH_SIZE KiB':>15s} {'WRITE_SIZE KiB':>15s} {'HBM GB/launch':>14s}")
    for k in sorted(f, key=lambda k: -(2 * f[k].get("FETCH_SIZE", (0, 0))[0] + w.get(k, {}).get("WRITE_SIZE", (0, 0))[0]) * f[k].get("FETCH_SIZE", (0, 1))[1]):
        fs, n = f[k].get("FETCH_SIZE", (0.0, 0))
        ws = w.get(k, {}).get("WRITE_SIZE", (0.0, 0))[0]
        b = (2.0 * fs + ws) * 1024.0
        rec[k] = {"launches": n, "fetch_size_kib": round(fs, 1), "write_size_kib": round(ws, 1), "hbm_bytes_per_launch": int(b)}
        print(f"{k[:84]:84s} {n:8d} {fs:15.1f} {ws:15.1f} {b / 1e9:14.3f}")
    if out:
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench                       # the digest of the kernel sources the counters were collected on: bench.py reports the traffic only while it matches
        srcs = bench.PERFORMER_KERNEL_SOURCES if sources == "performer" else bench.VQVAE_KERNEL_SOURCES
        with open(out, "w") as fh:
            json.dump({"_provenance": provenance, "csrc_sha1": bench.csrc_digest(srcs), "sources": sources, "kernels": rec}, fh, indent=1)


if __name__ == "__main__":
    cmd = sys.argv[1]
    if cmd == "stats":
        stats(sys.argv[2], by_grid="--by-grid" in sys.argv[3:])
    elif cmd == "pmc":
        pmc(sys.argv[2], by_grid="--by-grid" in sys.argv[3:])
    elif cmd == "traffic":
        args = [x for x in sys.argv[2:] if x != "--by-grid" and not x.startswith("--sources=")]
        src = [x.split("=", 1)[1] for x in sys.argv[2:] if x.startswith("--sources=")]
        traffic(args[0], args[1], args[2] if len(args) > 2 else None, args[3] if len(args) > 3 else "", by_grid="--by-grid" in sys.argv[2:],
                sources=src[0] if src else "vqvae")
