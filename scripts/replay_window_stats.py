#!/usr/bin/env python3
"""Per-kernel statistics of the REPLAYED steps only, out of a rocprofv3 --kernel-trace CSV.

    replay_window_stats.py kernel_trace.csv STEPS out.csv

The trace of a bench.py run holds set-up work (capture warm-ups, casts, copies, fills) in front of the timed graph
replays; `--stats` averages over all of it.  Here the step is found as the smallest period P of the kernel-name
sequence at the END of the trace (the last 3 P names repeat), the window is the last STEPS x P dispatches, and every
row is per step: launches, microseconds, share.  The last rows give the sum of kernel time per step and the wall span
per step (first start to last end of the window / STEPS): their difference is idle time between kernels.
"""
import csv
import sys
from collections import OrderedDict


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "")


def main():
    path, steps, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    period = None
    for skip in range(0, 64):  # a few dispatches may follow the last step (the copy behind loss.item(), a probe)
        n = len(names) - skip
        for p in range(8, n // 3 + 1):
            if names[n - p:n] == names[n - 2 * p:n - p] == names[n - 3 * p:n - 2 * p]:
                period = p
                break
        if period is not None:
            break
    if period is None:
        # kernels of several streams interleave differently from step to step (the data-parallel exchange beside
        # backward): delimit the steps by a kernel that runs exactly once per step instead
        delim = next((d for d in ("begin_step_kernel", "adam_tiled_kernel") if sum(d in x for x in names) >= 3), None)
        if delim is None:
            raise SystemExit("no periodic tail found in the kernel trace")
        marks = [i for i, x in enumerate(names) if delim in x]
        steps = min(steps, len(marks) - 1)
        lo, hi = marks[-1 - steps], marks[-1]
        win = rows[lo:hi]
        period = round(len(win) / steps, 1)
    else:
        rows, names = rows[:n], names[:n]
        steps = min(steps, n // period)
        while steps > 1 and names[n - steps * period:n - (steps - 1) * period] != names[n - period:]:
            steps -= 1
        win = rows[n - steps * period:]
    agg = OrderedDict()
    for r in win:
        a = agg.setdefault(short(r["Kernel_Name"]), [0, 0.0, 1e30, 0.0])
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    span = (max(int(r["End_Timestamp"]) for r in win) - int(win[0]["Start_Timestamp"])) / 1e3
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches_per_step", "us_per_step", "avg_us", "min_us", "max_us", "pct_of_kernel_time"])
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k, round(a[0] / steps, 2), round(a[1] / steps, 2), round(a[1] / a[0], 2), round(a[2], 2),
                        round(a[3], 2), round(100 * a[1] / total, 2)])
        w.writerow(["# steps in window", steps, "", "", "", "", ""])
        w.writerow(["# launches per step", period, "", "", "", "", ""])
        w.writerow(["# sum of kernel time per step (us)", "", round(total / steps, 1), "", "", "", ""])
        w.writerow(["# wall span per step (us)", "", round(span / steps, 1), "", "", "", ""])
    print(f"{out}: {steps} steps x {period} launches, kernel time {total / steps:.1f} us/step, span {span / steps:.1f} us/step")


if __name__ == "__main__":
    main()
