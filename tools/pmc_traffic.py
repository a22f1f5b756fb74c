"""HBM-side bytes per launch of one kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).
usage: pmc_traffic.py <fetch csv> <write csv> <kernel-name substring>
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE counts KB assuming 32-B units while the 16-B-per-lane loads of
these kernels make 64-B requests on gfx950 (hence the factor 2); WRITE_SIZE is exact."""
import csv, sys


def mean(path, counter, sub):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and sub in r["Kernel_Name"]]
    return sum(v) / len(v), len(v)


f, nf = mean(sys.argv[1], "FETCH_SIZE", sys.argv[3])
w, nw = mean(sys.argv[2], "WRITE_SIZE", sys.argv[3])
print(f"fetch_size_kb {f:.1f} (n={nf})  write_size_kb {w:.1f} (n={nw})  bytes_per_launch {(2 * f + w) * 1024:.0f}")
