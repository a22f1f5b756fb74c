"""Mean counter values per kernel from a rocprofv3 counter_collection csv.  usage: pmc_summary.py <csv> [kernel substring]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
agg = {}
for r in rows:
    if sub in r["Kernel_Name"]:
        a = agg.setdefault((r["Kernel_Name"][:70], r["Counter_Name"]), [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
for (k, c), (n, s) in sorted(agg.items()):
    print(f"{k:70s} {c:28s} n={n:3d} mean={s / n:14.1f}")
