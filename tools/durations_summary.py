#!/usr/bin/env python3
"""Compress a kernel_durations.py listing: consecutive launches of the same kernel and grid -> one line."""
import re
import sys

prev, acc = None, []


def flush():
    if prev:
        print(f"{prev[0][:66]:66s} grid {prev[1]:>8s} n {len(acc):4d} avg {sum(acc) / len(acc):9.1f} us")


for line in open(sys.argv[1]):
    m = re.match(r"(.*?)\s+([\d.]+) us\s+grid (\d*)", line)
    if not m:
        continue
    key = (m.group(1).strip(), m.group(3))
    if key != prev:
        flush()
        prev, acc = key, []
    acc.append(float(m.group(2)))
flush()
