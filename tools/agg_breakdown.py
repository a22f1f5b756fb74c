"""Aggregate a `bench.py --breakdown` table by op name: total ms, launches, mean GB/s / TFLOP/s."""
import collections, re, sys
tot = collections.defaultdict(lambda: [0.0, 0, 0.0, 0.0])
for line in open(sys.argv[1]):
    m = re.match(r"\s*([\d.]+) ms\s+([\d.]+) TFLOP/s\s+([\d.]+) GB/s\s+(?:\w+\s+[\d.]+\s+\w+\s+)?(\w+)\[", line)
    if not m:
        continue
    ms, tf, gb, name = float(m.group(1)), float(m.group(2)), float(m.group(3)), m.group(4)
    t = tot[name]
    t[0] += ms; t[1] += 1; t[2] += tf * ms; t[3] += gb * ms
all_ms = sum(t[0] for t in tot.values())
for name, t in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print(f"{name:20s} {t[0]:8.3f} ms {100 * t[0] / all_ms:5.1f}%  {t[1]:5d} launches  {t[2] / t[0]:7.2f} TFLOP/s  {t[3] / t[0]:8.1f} GB/s")
print(f"{'total traced':20s} {all_ms:8.3f} ms")
