#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace`):
    rocpd_stats.py <results.db> [--after-last <kernel substring>] [--between <start substring> <end substring> <occurrence>]
Without options: all dispatches.  --window a b: only dispatches whose index (in start order) lies in [a, b)."""
import argparse
import re
import sqlite3
from collections import defaultdict


def load(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    names = {r[0]: r[1] for r in cur.execute(f"select id, kernel_name from {sym}")}
    rows = [dict(name=names.get(k, str(k)), start=s, end=e, grid=(gx, gy, gz), wg=(wx, wy, wz), lds=lds)
            for k, s, e, gx, gy, gz, wx, wy, wz, lds in cur.execute(
                f"select kernel_id, start, end, grid_size_x, grid_size_y, grid_size_z, workgroup_size_x, workgroup_size_y, "
                f"workgroup_size_z, group_segment_size from {disp} order by start")]
    return rows


def short(name):
    name = re.sub(r"^_ZN\d+_GLOBAL__N_1\d+", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:64]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--window", nargs=2, type=int)
    ap.add_argument("--last-chain", help="only the dispatches after the second-to-last and up to the last dispatch of this kernel")
    ap.add_argument("--gridz", type=int, help="with --last-chain: the marker kernel must have this many workgroups in z")
    args = ap.parse_args()
    rows = load(args.db)
    if args.last_chain:
        marks = [i for i, r in enumerate(rows) if args.last_chain in r["name"] and
                 (args.gridz is None or r["grid"][2] // max(1, r["wg"][2]) == args.gridz)]
        allm = [i for i, r in enumerate(rows) if args.last_chain in r["name"]]
        end = marks[-1]
        prev = max([i for i in allm if i < end], default=-1)
        rows = rows[prev + 1:end + 1]
    if args.window:
        rows = rows[args.window[0]:args.window[1]]
    agg = defaultdict(lambda: [0, 0.0])
    for r in rows:
        a = agg[short(r["name"])]
        a[0] += 1
        a[1] += (r["end"] - r["start"]) / 1e3
    span = (rows[-1]["end"] - rows[0]["start"]) / 1e3 if rows else 0.0
    busy = sum(v[1] for v in agg.values())
    print(f"{len(rows)} dispatches, span {span:.1f} us, sum of kernel durations {busy:.1f} us ({100 * busy / max(span, 1e-9):.0f} % of the span)")
    for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{name:64s} calls={n:5d} total_us={t:10.1f} avg_us={t / n:8.1f} pct={100 * t / busy:5.1f}")


if __name__ == "__main__":
    main()
