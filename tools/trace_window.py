"""Aggregate the tail of a rocprofv3 kernel trace: per-kernel time of the dispatches that started in the last WINDOW_MS
milliseconds of the trace (the timed steps of a bench run, after tuning and warm-up are over).

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --mode train --steps 20 --no-verify ...
    python tools/trace_window.py DIR WINDOW_MS STEPS > summary.txt
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root, window_ms, steps = sys.argv[1], float(sys.argv[2]), int(sys.argv[3])
    files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    assert files, f"no kernel trace under {root}"
    rows = []
    with open(files[0]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    last = rows[-1][1]
    lo = last - int(window_ms * 1e6)
    agg = defaultdict(lambda: [0, 0])
    busy = 0
    for s, e, n in rows:
        if s < lo:
            continue
        a = agg[n]
        a[0] += 1
        a[1] += e - s
        busy += e - s
    print(f"window {window_ms:.1f} ms = {steps} steps; kernel time in it {busy / 1e6:.2f} ms ({busy / 1e6 / steps:.3f} ms per step)")
    print(f"{'ms/step':>9s} {'calls/step':>10s} {'avg us':>9s}  kernel")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{t / 1e6 / steps:9.4f} {c / steps:10.1f} {t / 1e3 / c:9.2f}  {n[:150]}")


if __name__ == "__main__":
    main()
