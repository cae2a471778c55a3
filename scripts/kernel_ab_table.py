#!/usr/bin/env python3
"""Rows of replay-window kernel statistics (scripts/replay_window_stats.py) that match a pattern, side by side.
    kernel_ab_table.py PATTERN label=stats.csv [label=stats.csv ...]"""
import csv
import re
import sys

pat = re.compile(sys.argv[1])
cols, tab = [], {}
for arg in sys.argv[2:]:
    label, path = arg.rsplit("::=", 1) if "::=" in arg else (arg.split("=", 1) if "=" in arg else (arg, arg))
    cols.append(label)
    for r in csv.reader(open(path)):
        if r[0] == "kernel":
            continue
        if r[0].startswith("#"):
            tab.setdefault(r[0], {})[label] = r[1] or r[2]
        elif pat.search(r[0]):
            tab.setdefault(re.sub(r"\(.*", "", r[0])[:72], {})[label] = f"{r[1]} x {r[3]} = {r[2]}"
w = max(18, max(len(c) for c in cols) + 1)
print("".ljust(74) + "".join(c.rjust(w) + " |" for c in cols))
for k, v in tab.items():
    print(k.ljust(74) + "".join(v.get(c, "-").rjust(w) + " |" for c in cols))
