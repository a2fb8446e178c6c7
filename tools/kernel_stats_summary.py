#!/usr/bin/env python3
"""Condense a rocprofv3 `--kernel-trace --stats` kernel_stats CSV into the per-kernel table kept under profiles/.

    python tools/kernel_stats_summary.py gpurun_out/prof/x_kernel_stats.csv [steps]

Prints total time, calls, average and share per kernel (sorted by total); with `steps` also the per-step total."""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("# total kernel time %.1f us%s" % (tot / 1e3, " = %.3f ms per step" % (tot / 1e6 / steps) if steps else ""))
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
        t, n = float(r["TotalDurationNs"]), int(r["Calls"])
        if t / tot < 0.002:
            continue
        print("%9.1f us %6d x %8.2f us %5.1f%%  %s" % (t / 1e3, n, t / n / 1e3, 100 * t / tot, r["Name"][:150]))


if __name__ == "__main__":
    main()
