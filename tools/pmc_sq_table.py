"""Per-kernel table of the SQ counter pass (rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS), means per launch.
usage: pmc_sq_table.py counter_collection.csv [substring ...]   (kernels whose name contains any substring; default: the
MBConv, split-product GEMM, projection, depthwise and squeeze-excite kernels)"""
import csv, re, sys
from collections import defaultdict

subs = sys.argv[2:] or ["expand_dw", "gemm_bf16_kernel", "dwconv_strip", "se_gate", "gemm_x3dl", "gemm_x3t", "thin_project"]
disp = {}
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    if not any(s in name for s in subs):
        continue
    d = disp.setdefault(r["Dispatch_Id"], {"name": name, "us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
agg = defaultdict(list)
for d in disp.values():
    short = re.sub(r"^void icap::", "", d["name"])
    short = re.sub(r"\(.*$", "", short)
    agg[short].append(d)
print(f"{'kernel':66s} {'n':>3s} {'us':>8s} {'wait_any':>8s} {'wait_inst':>9s} {'act_any':>8s} {'act_valu':>8s} {'act_lds':>8s} {'mfma_busy':>9s}")
for k in sorted(agg):
    ds = agg[k]
    n = len(ds)
    m = lambda c: sum(d.get(c, 0.0) for d in ds) / n
    us, wc = m("us"), m("SQ_WAVE_CYCLES")
    if wc <= 0:
        continue
    print(f"{k[:66]:66s} {n:3d} {us:8.1f} {m('SQ_WAIT_ANY') / wc:8.3f} {m('SQ_WAIT_INST_ANY') / wc:9.3f} {m('SQ_ACTIVE_INST_ANY') / wc:8.3f} "
          f"{m('SQ_ACTIVE_INST_VALU') / wc:8.3f} {m('SQ_ACTIVE_INST_LDS') / wc:8.3f} {m('SQ_VALU_MFMA_BUSY_CYCLES') / (us * 1e-6 * 2.4e9 * 1024):9.3f}")
